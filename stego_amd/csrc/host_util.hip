// Per-device caches and measurement knobs (see host_util.h).
#include <atomic>
#include <chrono>
#include <thread>
#include <cstdlib>
#include <map>
#include <mutex>
#include <utility>

#include "host_util.h"

namespace stego {

namespace {
std::mutex g_mutex;
std::map<std::pair<int, const void*>, int> g_lds;       // (device, kernel) -> dynamic LDS limit already set
std::map<int, int> g_cus;                                // device -> compute units

int env_int(const char* name, int dflt)
{
    const char* v = std::getenv(name);
    return v && *v ? std::atoi(v) : dflt;
}

struct Knobs {
    std::atomic<int> v[KNOB_COUNT];
    Knobs()
    {
        v[KNOB_DEBUG] = env_int("STEGO_DEBUG", 0);
        v[KNOB_DEBUG_SAMPLE] = env_int("STEGO_DEBUG_SAMPLE", 0);
        v[KNOB_DEBUG_BWD] = env_int("STEGO_DEBUG_BWD", 0);
        v[KNOB_DEBUG_VIT] = env_int("STEGO_DEBUG_VIT", 0);
        v[KNOB_DEBUG_KNN] = env_int("STEGO_DEBUG_KNN", 0);
        v[KNOB_FWD_VARIANT] = env_int("STEGO_FWD_VARIANT", -1);
        v[KNOB_SHARED_DEVICE] = env_int("STEGO_SHARED_DEVICE", 0);
    }
};
Knobs g_knobs;      // initialised when the library is loaded
}  // namespace

hipError_t ensure_dynamic_lds(const void* kernel, int bytes)
{
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    std::lock_guard<std::mutex> lock(g_mutex);
    int& have = g_lds[std::make_pair(dev, kernel)];
    if (have >= bytes) return hipSuccess;
    e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e == hipSuccess) have = bytes;
    return e;
}

int device_cu_count()
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 256;
    std::lock_guard<std::mutex> lock(g_mutex);
    auto it = g_cus.find(dev);
    if (it != g_cus.end()) return it->second;
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    g_cus[dev] = n;
    return n;
}

int knob(int which) { return (which >= 0 && which < KNOB_COUNT) ? g_knobs.v[which].load(std::memory_order_relaxed) : 0; }

void set_knob(int which, int value)
{
    if (which >= 0 && which < KNOB_COUNT) g_knobs.v[which].store(value, std::memory_order_relaxed);
}

// ---- test hook: a kernel that just holds compute units (stego_debug_occupy)
__global__ void occupy_kernel(long long ticks, int* sink)
{
    extern __shared__ int occupy_lds[];
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    occupy_lds[threadIdx.x] = threadIdx.x;
    while ((long long)(__builtin_amdgcn_s_memrealtime() - t0) < ticks) __builtin_amdgcn_s_sleep(32);
    if (occupy_lds[threadIdx.x] == -1 && sink) sink[0] = 1;
}

hipError_t launch_occupy(int n_wg, int lds_bytes, int micros, hipStream_t stream)
{
    hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(&occupy_kernel), lds_bytes);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(occupy_kernel, dim3(n_wg), dim3(256), lds_bytes, stream, (long long)micros * 100, (int*)nullptr);
    return hipGetLastError();
}

// ---- the library's side stream (stego_corr_workspace_prepare_now)
struct SideChannel {
    hipStream_t stream = nullptr;
    int* flag_host = nullptr;
    int* flag_dev = nullptr;
    int seq = 0;
};
static std::mutex side_mutex;
static std::map<int, SideChannel> side_channels;

__global__ void side_publish_kernel(int* flag, int value)
{
    *reinterpret_cast<volatile int*>(flag) = value;
    __threadfence_system();
}

static hipError_t side_channel(SideChannel** out)
{
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    SideChannel& ch = side_channels[dev];
    if (!ch.stream) {
        if ((e = hipStreamCreateWithFlags(&ch.stream, hipStreamNonBlocking)) != hipSuccess) return e;
        if ((e = hipHostMalloc(reinterpret_cast<void**>(&ch.flag_host), 64, hipHostMallocDefault)) != hipSuccess) return e;
        ch.flag_host[0] = 0;
        if ((e = hipHostGetDevicePointer(reinterpret_cast<void**>(&ch.flag_dev), ch.flag_host, 0)) != hipSuccess) return e;
    }
    *out = &ch;
    return hipSuccess;
}

hipError_t side_begin(hipStream_t* stream)
{
    std::lock_guard<std::mutex> lock(side_mutex);
    SideChannel* ch = nullptr;
    hipError_t e = side_channel(&ch);
    if (e == hipSuccess) *stream = ch->stream;
    return e;
}

hipError_t side_finish()
{
    std::lock_guard<std::mutex> lock(side_mutex);
    SideChannel* ch = nullptr;
    hipError_t e = side_channel(&ch);
    if (e != hipSuccess) return e;
    const int want = ++ch->seq;
    hipLaunchKernelGGL(side_publish_kernel, dim3(1), dim3(1), 0, ch->stream, ch->flag_dev, want);
    if ((e = hipGetLastError()) != hipSuccess) return e;
    const auto t0 = std::chrono::steady_clock::now();
    while (*reinterpret_cast<volatile int*>(ch->flag_host) != want) {
        if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(10)) return hipErrorNotReady;
        std::this_thread::yield();
    }
    return hipSuccess;
}

}  // namespace stego
