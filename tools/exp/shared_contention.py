"""Does a kernel that holds a few CUs (a collective) stall the fused forward?  Default launch (one workgroup per CU, helpers) vs
STEGO_SHARED_DEVICE (one per tile), with n_wg stand-in workgroups of 64 KB LDS spinning 80 us on a second stream."""
import ctypes, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
from stego_amd import capi
_so = os.path.join(ROOT, "tools", "ubench", "lib", "liboccupy.so")
if not os.path.exists(_so):          # (git-ignored build product: hipcc --offload-arch=gfx950 -O3 -fPIC -shared tools/ubench/occupy.hip)
    import subprocess
    os.makedirs(os.path.dirname(_so), exist_ok=True)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-fPIC", "-shared",
                           os.path.join(ROOT, "tools", "ubench", "occupy.hip"), "-o", _so])
occ = ctypes.CDLL(_so)
occ.occupy_launch.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
dev = torch.device("cuda:0")
cfg = bench.Cfg()
C, H, W, K = bench.WORKLOADS["vits8_224"]
B, S, n_neg = 32, 11, 5
sets = [bench.make_inputs(B, C, H, W, K, S, n_neg, 1000 + i, dev) for i in range(4)]
desc = capi.make_desc(B, C, K, H, W, S, n_neg, cfg, (.18, .12, .46), capi.PREC_F16X3)
sink = torch.zeros(4, dtype=torch.int32, device=dev)
side = torch.cuda.Stream()
def fwd(d):
    return capi.corr_fwd(desc, d["feats"], d["feats_pos"], d["code"], d["code_pos"], d["coords1"], d["coords2"], d["perms"], True)
for shared in (0, 1):
    capi.set_shared_device(bool(shared))
    for n_wg in (0, 8, 32):
        for _ in range(3): fwd(sets[0])
        torch.cuda.synchronize()
        ts = []
        for rep in range(20):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            if n_wg:
                occ.occupy_launch(n_wg, 64 * 1024, 80, sink.data_ptr(), side.cuda_stream)
            torch.cuda._sleep(20000)          # let the stand-in get onto its CUs first (~10 us)
            e0.record(); fwd(sets[rep % 4]); e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3)
        ts.sort()
        print("shared_device=%d  stand-in workgroups=%2d  forward us: median %.1f  max %.1f" % (shared, n_wg, ts[len(ts) // 2], ts[-1]))
capi.set_shared_device(False)
