"""Timeline of the lists-first backward (corr_bwd_tile_build_kernel + corr_unsample_list_kernel, STEGO_DEBUG_BWD bit 8): builders, tiles, unsample workgroups.
usage: python tools/stamps_bwd_lists.py [batch]"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import bench
from stego_amd import capi
from ctypes import byref
dev = torch.device("cuda:0")
cfg = bench.Cfg()
C, H, W, K = bench.WORKLOADS[os.environ.get("WORKLOAD", "vits8_224")]
B, S, n_neg = (int(sys.argv[1]) if len(sys.argv) > 1 else 32), 11, 5
d = bench.make_inputs(B, C, H, W, K, S, n_neg, 1000, dev)
desc = capi.make_desc(B, C, K, H, W, S, n_neg, cfg, (.18, .12, .46), capi.PREC_F16X3)
lib = capi.load()
out = capi.corr_fwd(desc, d["feats"], d["feats_pos"], d["code"], d["code_pos"], d["coords1"], d["coords2"], d["perms"], True)
lm, icd, ecd, nl, ncd, saved = out
sw, sm, sctx = saved
g_intra = torch.tensor(0.67, device=dev); g_inter = torch.tensor(0.25, device=dev)
g_neg = torch.full((1,), 0.63 / (n_neg * B * S ** 4), device=dev)
capi.debug_set("STEGO_DEBUG_BWD", int(os.environ.get("SDB", "8")))
nws = lib.stego_corr_bwd_workspace_bytes(byref(desc))
ws = torch.zeros(nws, dtype=torch.uint8, device=dev)
dc = torch.empty(B, H, W, K, device=dev); dcp = torch.empty(B, H, W, K, device=dev)
for rep in range(3):
    rc = lib.stego_corr_bwd(byref(desc), d["perms"].data_ptr(), sw.data_ptr(), sm.data_ptr(), sctx.data_ptr(), icd.data_ptr(), ecd.data_ptr(),
                            ncd.data_ptr(), g_intra.data_ptr(), g_inter.data_ptr(), g_neg.data_ptr(), 0, None, None, None,
                            dc.data_ptr(), dcp.data_ptr(), ws.data_ptr(), ws.numel(), torch.cuda.current_stream().cuda_stream)
    assert rc == 0, rc
    torch.cuda.synchronize()
nt = (2 + n_neg) * B
LDK = ((K + 7) & ~7) + 4
dt_bytes = (nt * 2 * 128 * LDK * 4 + 255) & ~255
ts = ws[dt_bytes: dt_bytes + 65536].view(torch.int64).cpu().numpy()
n_build = min(B, 32)
n_pairs = B * H * ((W + 15) // 16)
n_uwg = min((n_pairs + 3) // 4, 1024)
uns = ts[: 2 * n_uwg].reshape(n_uwg, 2)
build = ts[2048: 2048 + 2 * n_build].reshape(n_build, 2)
tile = ts[4096: 4096 + 4 * min(nt, 1024)].reshape(-1, 4)
t0 = min(build[:, 0].min(), tile[:, 0].min())
us = lambda a: (a - t0) / 100.0
pc = lambda a: "%.2f %.2f %.2f" % tuple(np.percentile(us(a), [0, 50, 100]))
print("B=%d: %d builders + %d tiles, then %d unsample workgroups of 4 waves (us since the first workgroup of the first launch started; p0 / p50 / p100)" % (B, n_build, nt, n_uwg))
print("builder start      ", pc(build[:, 0]))
print("builder end        ", pc(build[:, 1]))
for k, n in enumerate(["tile start", "tile G + copies", "tile mfma", "tile end"]):
    print("%-19s" % n, pc(tile[:, k]))
print("unsample start     ", pc(uns[:, 0]))
print("unsample end       ", pc(uns[:, 1]))
print("unsample duration  ", "%.2f %.2f %.2f" % tuple(np.percentile((uns[:, 1] - uns[:, 0]) / 100.0, [0, 50, 100])))
