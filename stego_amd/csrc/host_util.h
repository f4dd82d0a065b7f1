// Host-side helpers shared by the launchers: per-device caches (thread-safe), environment knobs read once.
#pragma once
#include <hip/hip_runtime.h>

namespace stego {

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is a per-device property of a kernel: remember what each
// (device, kernel) pair already has, under a mutex (one process may drive several GPUs from several threads, as the
// reference's DataParallel feature extraction does, precompute_knns.py:59).
hipError_t ensure_dynamic_lds(const void* kernel, int bytes);

// number of compute units of the current device (cached per device)
int device_cu_count();

// Measurement / ablation knobs: STEGO_DEBUG, STEGO_DEBUG_SAMPLE, STEGO_DEBUG_BWD, STEGO_DEBUG_VIT, STEGO_DEBUG_KNN, STEGO_FWD_VARIANT,
// and one deployment knob, STEGO_SHARED_DEVICE (1: other kernels - the collectives of a data-parallel job - run on the device while the
// loss does: the fused forward then launches no phase-1 helper workgroups and so leaves the compute units beyond its tiles free;
// a value > 8 is an explicit number of phase-1 owner workgroups, for measurements: 232 .. 256 time alike, 224 costs ~1 us).
// Read from the environment once, when the library is loaded; stego_debug_set() (C ABI, tools only) overrides them
// afterwards.  The product path never calls getenv.
enum { KNOB_DEBUG = 0, KNOB_DEBUG_SAMPLE, KNOB_DEBUG_BWD, KNOB_DEBUG_VIT, KNOB_DEBUG_KNN, KNOB_FWD_VARIANT, KNOB_SHARED_DEVICE, KNOB_COUNT };
int knob(int which);
void set_knob(int which, int value);

// test hook (stego_debug_occupy): n_wg workgroups of 256 threads with lds_bytes of LDS each spin for `micros`
hipError_t launch_occupy(int n_wg, int lds_bytes, int micros, hipStream_t stream);
// A stream of the library's own per device + a pinned flag word: side_begin() returns the stream (creating both on the first call for a
// device - NOT while a capture is under way), side_finish() returns when everything enqueued on it has run, by watching the flag a
// one-thread kernel sets: no HIP synchronisation API is called, so both are legal while the calling thread captures another stream.
hipError_t side_begin(hipStream_t* stream);
hipError_t side_finish();

}  // namespace stego
