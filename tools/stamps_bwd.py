import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from stego_amd import capi
from ctypes import byref
dev = torch.device("cuda:0")
cfg = bench.Cfg()
C, H, W, K = bench.WORKLOADS["vits8_224"]
B, S, n_neg = 32, 11, 5
d = bench.make_inputs(B, C, H, W, K, S, n_neg, 1000, dev)
desc = capi.make_desc(B, C, K, H, W, S, n_neg, cfg, (.18, .12, .46), capi.PREC_F32 if os.environ.get("PREC") == "f32" else capi.PREC_F16X3)
lib = capi.load()
out = capi.corr_fwd(desc, d["feats"], d["feats_pos"], d["code"], d["code_pos"], d["coords1"], d["coords2"], d["perms"], True)
lm, icd, ecd, nl, ncd, saved = out
sw, sm, sctx = saved
g_intra = torch.tensor(0.67, device=dev); g_inter = torch.tensor(0.25, device=dev)
g_neg = torch.full((1,), 0.63 / (n_neg * B * S ** 4), device=dev)
capi.debug_set("STEGO_DEBUG_BWD", int(sys.argv[1]) if len(sys.argv) > 1 else 8)
nws = lib.stego_corr_bwd_workspace_bytes(byref(desc))
ws = torch.zeros(nws, dtype=torch.uint8, device=dev)
dc = torch.empty(B, H, W, K, device=dev); dcp = torch.empty(B, H, W, K, device=dev)
for rep in range(3):
    rc = lib.stego_corr_bwd(byref(desc), d["perms"].data_ptr(), sw.data_ptr(), sm.data_ptr(), sctx.data_ptr(), icd.data_ptr(), ecd.data_ptr(),
                            ncd.data_ptr(), g_intra.data_ptr(), g_inter.data_ptr(), g_neg.data_ptr(), 0, None, None, None,
                            dc.data_ptr(), dcp.data_ptr(), ws.data_ptr(), ws.numel(), torch.cuda.current_stream().cuda_stream)
    assert rc == 0, rc
    torch.cuda.synchronize()
import numpy as np
nblk = B * H + (B * H + 2) // 3
ts = ws[nws - 65536: nws - 65536 + 16 * nblk].view(torch.int64).cpu().numpy().reshape(nblk, 2)
t0 = ts[:, 0].min()
st, en = (ts[:, 0] - t0) / 100.0, (ts[:, 1] - t0) / 100.0
print("blocks", nblk, "span us", en.max())
for name, sl in (("heavy", slice(0, B * H)), ("light", slice(B * H, nblk))):
    print(name, "start p0/p50/p100 %.2f %.2f %.2f" % tuple(np.percentile(st[sl], [0, 50, 100])),
          "end p0/p50/p100 %.2f %.2f %.2f" % tuple(np.percentile(en[sl], [0, 50, 100])),
          "dur p0/p50/p100 %.2f %.2f %.2f" % tuple(np.percentile((en - st)[sl], [0, 50, 100])))

nt = (2 + n_neg) * B
tt = ws[nws - 65536 + 32768: nws - 65536 + 32768 + 32 * nt].view(torch.int64).cpu().numpy().reshape(nt, 4)
t0 = tt[:, 0].min()
rel = (tt - t0) / 100.0
for k, n in enumerate(["start", "G + copies", "mfma", "end"]):
    print("tile kernel %-11s p0/p50/p100 %.2f %.2f %.2f us" % ((n,) + tuple(np.percentile(rel[:, k], [0, 50, 100]))))
