"""Build libstego_corr.so (HIP kernels + C ABI) in-tree with hipcc for gfx950.

hipcc cross-compiles without a GPU, so this runs in the build container; the resulting
``stego_amd/lib/libstego_corr.so`` is git-ignored but travels to the GPU box with the
repo snapshot.  No torch headers are involved: the library is plain C ABI.

Every source is compiled to its own object (``stego_amd/lib/obj/*.o``, in parallel, only when it or a header
changed) and the objects are linked into the one shared library: a one-kernel edit rebuilds in seconds.
"""
import os
import shutil
import subprocess
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB_DIR = os.path.join(HERE, "lib")
OBJ_DIR = os.path.join(LIB_DIR, "obj")
LIB_PATH = os.path.join(LIB_DIR, "libstego_corr.so")
SOURCES = ["corr_sample.hip", "corr_fwd.hip", "corr_fused.hip", "corr_fused_odd.hip", "corr_fused_c192.hip", "corr_fused_half.hip", "corr_bwd.hip", "knn_topk.hip", "dense_corr.hip", "dense_stream.hip", "sample_sets.hip", "loss_pointwise.hip", "corr_wide.hip", "vit_forward.hip", "host_util.hip", "draws.hip", "head_fused.hip", "c_api.hip"]
HEADERS = ["corr_common.h", "corr_tile.h", "host_util.h", "corr_wide.h", "corr_fused.hip", os.path.join("..", "..", "include", "stego_corr.h"), os.path.join("..", "..", "include", "stego_vit.h"),
           os.path.join("..", "..", "include", "stego_head.h")]
CFLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]
LDFLAGS = ["--offload-arch=gfx950", "-fPIC", "-shared"]
# the one torch C++ extension (host only, g++): the autograd function of the loss and the generator's graph-safe Philox state, over the C ABI
# measurement only (bench.py's roofline.frac_of_skeleton): the traffic skeleton of the fused forward, a stand-alone binary beside the library
SKELETON_SRC = os.path.join(HERE, "..", "tools", "ubench", "fused_skeleton.hip")
SKELETON_PATH = os.path.join(LIB_DIR, "fused_skeleton.bin")
TORCHGLUE_SRC = os.path.join(CSRC, "torch_glue_ext.cpp")
TORCHGLUE_PATH = os.path.join(LIB_DIR, "_stego_torchglue.so")


def _hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found (need ROCm with gfx950 support)")
    return exe


def sources():
    return [os.path.join(CSRC, s) for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]


def _headers():
    return [p for p in (os.path.join(CSRC, h) for h in HEADERS) if os.path.exists(p)]


def _obj(src):
    return os.path.join(OBJ_DIR, os.path.splitext(os.path.basename(src))[0] + ".o")


def _obj_stale(src):
    o = _obj(src)
    if not os.path.exists(o):
        return True
    t = os.path.getmtime(o)
    return any(os.path.getmtime(d) > t for d in [src, os.path.abspath(__file__)] + _headers())


def is_stale():
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    return any(os.path.getmtime(d) > t for d in sources() + _headers())


def _compile(src, extra, verbose):
    cmd = [_hipcc()] + CFLAGS + list(extra) + ["-c", src, "-o", _obj(src) + ".tmp"]
    if verbose:
        print(" ".join(cmd))
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError("hipcc failed on %s:\n%s%s" % (src, res.stdout, res.stderr))
    os.replace(_obj(src) + ".tmp", _obj(src))
    return res.stderr


def build(force=False, verbose=False, extra_flags=(), out=None):
    """Compile every HIP source (objects cached per file) and link the shared library. Returns the library path.
    `extra_flags` (e.g. -DSTEGO_FUSED_TIMELINE) force a full rebuild into `out` without touching the object cache."""
    out = out or LIB_PATH
    if not force and not extra_flags and out == LIB_PATH and not is_stale():
        return LIB_PATH
    os.makedirs(OBJ_DIR, exist_ok=True)
    srcs = sources()
    if extra_flags:                       # one-off variant build: everything in one go, no cache
        cmd = [_hipcc()] + CFLAGS + list(extra_flags) + ["-shared"] + srcs + ["-o", out + ".tmp"]
        res = subprocess.run(cmd, capture_output=True, text=True)
        if res.returncode != 0:
            raise RuntimeError("hipcc failed:\n" + res.stdout + res.stderr)
        os.replace(out + ".tmp", out)
        return out
    todo = [s for s in srcs if force or _obj_stale(s)]
    with ThreadPoolExecutor(max_workers=min(8, max(1, len(todo)))) as ex:
        list(ex.map(lambda s: _compile(s, (), verbose), todo))
    cmd = [_hipcc()] + LDFLAGS + [_obj(s) for s in srcs] + ["-o", out + ".tmp"]
    if verbose:
        print(" ".join(cmd))
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError("hipcc link failed:\n" + res.stdout + res.stderr)
    os.replace(out + ".tmp", out)
    return out


def build_torchglue(force=False, verbose=False):
    """Compile csrc/torch_glue_ext.cpp against the installed torch (pybind11 module, no device code) -> lib/_stego_torchglue.so."""
    deps = [TORCHGLUE_SRC, os.path.join(CSRC, "..", "..", "include", "stego_corr.h")]
    import sysconfig
    import torch
    stamp = TORCHGLUE_PATH + ".torch"          # the torch build the extension was compiled against: another one makes it stale
    built_for = open(stamp).read().strip() if os.path.exists(stamp) else ""
    if not force and os.path.exists(TORCHGLUE_PATH) and built_for == torch.__version__ and \
            all(os.path.getmtime(TORCHGLUE_PATH) >= os.path.getmtime(d) for d in deps):
        return TORCHGLUE_PATH
    from torch.utils import cpp_extension
    os.makedirs(LIB_DIR, exist_ok=True)
    tlib = os.path.join(os.path.dirname(torch.__file__), "lib")
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    cmd = [shutil.which("g++") or "g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1",
           "-DTORCH_EXTENSION_NAME=_stego_torchglue", "-D_GLIBCXX_USE_CXX11_ABI=%d" % int(torch._C._GLIBCXX_USE_CXX11_ABI)]
    cmd += ["-I" + d for d in cpp_extension.include_paths() + [os.path.join(rocm, "include"), sysconfig.get_paths()["include"]]]
    cmd += [TORCHGLUE_SRC, "-L" + tlib, "-Wl,-rpath," + tlib, "-lc10", "-lc10_hip", "-ltorch_cpu", "-ltorch_hip", "-ltorch", "-ltorch_python", "-L" + os.path.join(rocm, "lib"), "-lamdhip64", "-ldl",
            "-o", TORCHGLUE_PATH + ".tmp"]
    if verbose:
        print(" ".join(cmd))
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError("g++ failed on %s:\n%s%s" % (TORCHGLUE_SRC, res.stdout, res.stderr))
    os.replace(TORCHGLUE_PATH + ".tmp", TORCHGLUE_PATH)
    with open(stamp, "w") as f:
        f.write(torch.__version__)
    return TORCHGLUE_PATH


# ---- sanitizer build of the HOST side ("race / memory checkers"): the same sources, host code instrumented with
# AddressSanitizer + UndefinedBehaviorSanitizer (the device code is what the product ships: clang ignores -fsanitize for gfx950 without
# xnack+).  tests/test_asan_host.py runs the C-ABI's host logic - descriptor validation, geometry, error paths, the symbol table - through it.
ASAN_DIR = os.path.join(LIB_DIR, "asan")
ASAN_LIB_PATH = os.path.join(ASAN_DIR, "libstego_corr.so")
ASAN_FLAGS = ["-fsanitize=address,undefined", "-fno-omit-frame-pointer", "-shared-libsan", "-g", "-Wno-option-ignored"]


def asan_runtime():
    """The clang AddressSanitizer runtime to LD_PRELOAD into an uninstrumented python (None if this toolchain has none)."""
    import glob
    hits = sorted(glob.glob(os.path.join(os.path.dirname(os.path.dirname(os.path.realpath(_hipcc()))), "lib", "llvm", "lib", "clang", "*", "lib", "linux",
                                         "libclang_rt.asan-x86_64.so")))
    return hits[-1] if hits else None


def build_asan(force=False, verbose=False):
    """lib/asan/libstego_corr.so: every source with ASAN_FLAGS (objects cached under lib/obj/asan)."""
    odir = os.path.join(OBJ_DIR, "asan")
    os.makedirs(odir, exist_ok=True)
    os.makedirs(ASAN_DIR, exist_ok=True)
    srcs = sources()
    deps = _headers() + [os.path.abspath(__file__)]

    def obj(src):
        return os.path.join(odir, os.path.splitext(os.path.basename(src))[0] + ".o")

    def stale(src):
        o = obj(src)
        return force or not os.path.exists(o) or any(os.path.getmtime(d) > os.path.getmtime(o) for d in [src] + deps)

    def one(src):
        cmd = [_hipcc()] + CFLAGS + ASAN_FLAGS + ["-c", src, "-o", obj(src) + ".tmp"]
        if verbose:
            print(" ".join(cmd))
        res = subprocess.run(cmd, capture_output=True, text=True)
        if res.returncode != 0:
            raise RuntimeError("hipcc (asan) failed on %s:\n%s%s" % (src, res.stdout, res.stderr))
        os.replace(obj(src) + ".tmp", obj(src))

    todo = [s for s in srcs if stale(s)]
    with ThreadPoolExecutor(max_workers=min(8, max(1, len(todo)))) as ex:
        list(ex.map(one, todo))
    if todo or not os.path.exists(ASAN_LIB_PATH):
        res = subprocess.run([_hipcc()] + LDFLAGS + ASAN_FLAGS + [obj(s) for s in srcs] + ["-o", ASAN_LIB_PATH + ".tmp"], capture_output=True, text=True)
        if res.returncode != 0:
            raise RuntimeError("hipcc link (asan) failed:\n" + res.stdout + res.stderr)
        os.replace(ASAN_LIB_PATH + ".tmp", ASAN_LIB_PATH)
    return ASAN_LIB_PATH


if __name__ == "__main__":
    print(build(force=True, verbose=True))
    print(build_torchglue(force=True, verbose=True))


def build_skeleton(force=False, verbose=False):
    """tools/ubench/fused_skeleton.hip -> stego_amd/lib/fused_skeleton.bin (git-ignored; travels with the snapshot like the library)."""
    src = os.path.abspath(SKELETON_SRC)
    if not os.path.exists(src):
        return None
    if not force and os.path.exists(SKELETON_PATH) and os.path.getmtime(SKELETON_PATH) >= os.path.getmtime(src):
        return SKELETON_PATH
    os.makedirs(LIB_DIR, exist_ok=True)
    cmd = [_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", src, "-o", SKELETON_PATH + ".tmp"]
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    os.replace(SKELETON_PATH + ".tmp", SKELETON_PATH)
    return SKELETON_PATH
