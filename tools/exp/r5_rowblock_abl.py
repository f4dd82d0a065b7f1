"""Ablations of dense_rowblock_kernel<APANELS> at the loss's S = 16 shapes: time of the fd (C = 384) and cd (K = 70) launches with
A loads / stores / MFMAs removed (debug bits 16..18)."""
import os, sys, json
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from stego_amd import capi
dev = torch.device("cuda:0")
B, n_sets, S = 32, 7, int(sys.argv[1]) if len(sys.argv) > 1 else 16
P = S * S
for C in (384, 70):
    t = torch.randn(B, C, 28, 28, device=dev).contiguous(memory_format=torch.channels_last)
    coords = torch.rand(B, S, S, 2, device=dev) * 2 - 1
    idx = torch.randint(0, B, (5 * B,), device=dev)
    ps = capi.PanelSet(n_sets * B, C, P, dev)
    capi.sample_panels(ps, 0, t, coords)
    capi.sample_panels(ps, B, t, coords)
    capi.sample_panels(ps, 2 * B, t, coords, idx)
    for dbg, name in ((0, "full"), (1, "no A loads"), (2, "no stores"), (4, "no MFMA"), (6, "no stores, no MFMA"), (7, "nothing but the B stream"),
                      (8, "four stages (three copies in flight), one workgroup per CU"), (8 + 7, "four stages: nothing but the B stream")):
        capi.debug_set("STEGO_DEBUG", dbg << 16)
        for _ in range(3):
            capi.dense_corr_panels(ps, B, ps, n_sets * B, want_rowsum=True)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            capi.dense_corr_panels(ps, B, ps, n_sets * B, want_rowsum=True)
        e1.record(); torch.cuda.synchronize()
        print(json.dumps({"C": C, "variant": name, "us": round(e0.elapsed_time(e1) * 50, 1)}), flush=True)
    capi.debug_set("STEGO_DEBUG", 0)
