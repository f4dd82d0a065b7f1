"""Build libstego_corr.so (HIP kernels + C ABI) in-tree with hipcc for gfx950.

hipcc cross-compiles without a GPU, so this runs in the build container; the resulting
``stego_amd/lib/libstego_corr.so`` is git-ignored but travels to the GPU box with the
repo snapshot.  No torch headers are involved: the library is plain C ABI.
"""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB_DIR = os.path.join(HERE, "lib")
LIB_PATH = os.path.join(LIB_DIR, "libstego_corr.so")
SOURCES = ["corr_sample.hip", "corr_fwd.hip", "corr_fused.hip", "corr_bwd.hip", "knn_topk.hip", "dense_corr.hip", "vit_forward.hip", "host_util.hip", "draws.hip", "c_api.hip"]
HEADERS = ["corr_common.h", "corr_tile.h", "host_util.h", os.path.join("..", "..", "include", "stego_corr.h"), os.path.join("..", "..", "include", "stego_vit.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-munsafe-fp-atomics", "-fPIC", "-shared",
         "-Wall", "-Wno-unused-function"]


def _hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found (need ROCm with gfx950 support)")
    return exe


def sources():
    return [os.path.join(CSRC, s) for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]


def is_stale():
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = sources() + [os.path.join(CSRC, h) for h in HEADERS]
    return any(os.path.exists(d) and os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    """Compile every HIP source into one shared library. Returns the library path."""
    if not force and not is_stale():
        return LIB_PATH
    os.makedirs(LIB_DIR, exist_ok=True)
    tmp = LIB_PATH + ".tmp"
    cmd = [_hipcc()] + FLAGS + sources() + ["-o", tmp]
    if verbose:
        print(" ".join(cmd))
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError("hipcc failed:\n" + res.stdout + res.stderr)
    os.replace(tmp, LIB_PATH)
    return LIB_PATH


if __name__ == "__main__":
    print(build(force=True, verbose=True))
