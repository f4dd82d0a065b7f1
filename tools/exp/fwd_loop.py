#!/usr/bin/env python
"""Runs the forward N times eagerly (for rocprofv3 --kernel-trace / --pmc passes)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
from stego_amd import capi
dev = torch.device("cuda:0")
cfg = bench.Cfg()
wl = os.environ.get("WL", "vits8_224")
C, H, W, K = bench.WORKLOADS[wl]
B, S, n_neg = int(os.environ.get("B", 32)), 11, 5
sets = [bench.make_inputs(B, C, H, W, K, S, n_neg, 1000 + i, dev) for i in range(4)]
prec = capi.PREC_F32 if os.environ.get("PREC") == "f32" else capi.PREC_F16X3
desc = capi.make_desc(B, C, K, H, W, S, n_neg, cfg, (.18, .12, .46), prec)
if os.environ.get("VARIANT"): capi.debug_set("STEGO_FWD_VARIANT", int(os.environ["VARIANT"]))
if os.environ.get("DEBUG"): capi.debug_set("STEGO_DEBUG", int(os.environ["DEBUG"]))
for i in range(int(sys.argv[1]) if len(sys.argv) > 1 else 20):
    d = sets[i % 4]
    out = capi.corr_fwd(desc, d["feats"], d["feats_pos"], d["code"], d["code_pos"], d["coords1"], d["coords2"], d["perms"], True)
    if os.environ.get("BWD"):              # + the training backward (scalar upstreams)
        lm, icd, ecd, nl, ncd, saved = out
        gi = torch.tensor(0.67, device=dev); ge = torch.tensor(0.25, device=dev)
        gn = torch.full((1,), 0.63 / (n_neg * B * S ** 4), device=dev).expand(n_neg * B, S, S, S, S)
        capi.corr_bwd(desc, d["code"], d["code_pos"], d["coords1"], d["coords2"], d["perms"], saved, icd, ecd, ncd, gi, ge, gn, None, None, None)
torch.cuda.synchronize()
