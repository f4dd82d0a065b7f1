#!/usr/bin/env python
"""Race hunt: gather-team self-check counters (debug 2097152): registers vs a synchronous re-load, LDS read-back."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
from stego_amd import capi
from ctypes import byref
dev = torch.device("cuda:0")
cfg = bench.Cfg()
C, H, W, K = bench.WORKLOADS["vits8_224"]
B, S, n_neg = 32, 11, 5
sets = [bench.make_inputs(B, C, H, W, K, S, n_neg, 1000 + i, dev) for i in range(4)]
lib = capi.load()
nt = (2 + n_neg) * B
capi.debug_set("STEGO_DEBUG", int(os.environ.get("DEBUG", 8388608)))
desc = capi.make_desc(B, C, K, H, W, S, n_neg, cfg, (.18, .12, .46), capi.PREC_F16X3)
f32 = dict(dtype=torch.float32, device=dev)
outs = [torch.empty(2, **f32), torch.empty(B, S**4, **f32), torch.empty(B, S**4, **f32), torch.empty(n_neg * B, S**4, **f32),
        torch.empty(n_neg * B, S**4, **f32), torch.empty(7 * B, S**4, **f32), torch.empty(7, **f32)]
ctx = torch.empty(lib.stego_corr_saved_ctx_bytes(byref(desc)), dtype=torch.uint8, device=dev)
ws = torch.zeros(lib.stego_corr_workspace_bytes(byref(desc)), dtype=torch.uint8, device=dev)
for rep in range(int(os.environ.get("REPS", 200))):
    d = sets[rep % 4]
    maps = [capi._map(d[k]) for k in ("feats", "feats_pos", "code", "code_pos")]
    rc = lib.stego_corr_fwd(byref(desc), *[byref(m) for m in maps], d["coords1"].data_ptr(), d["coords2"].data_ptr(), d["perms"].data_ptr(),
                            *[o.data_ptr() for o in outs], ctx.data_ptr(), ws.data_ptr(), ws.numel(), torch.cuda.current_stream().cuda_stream)
    assert rc == 0
torch.cuda.synchronize()
cnt = ws[nt * 16: nt * 16 + 64].view(torch.int32).cpu().tolist()
print("register mismatches vs re-load: %d   LDS read-back mismatches: %d   checks: %d" % (cnt[0], cnt[1], cnt[2]))
print("LDS verify one barrier later: %d mismatching items of %d lane-checks; lane mask lo %08x hi %08x" % (cnt[8], cnt[11], cnt[9] & 0xffffffff, cnt[10] & 0xffffffff))
print("tap table at the end of the loop: %d offset entries, %d weight entries differ of %d" % (cnt[12], cnt[13], cnt[14]))
print("checksum mismatches (lanes whose consumed gather registers differ from a synchronous re-gather): %d of %d lane-tiles; lane mask lo %08x hi %08x" % (cnt[4], cnt[7], cnt[5] & 0xffffffff, cnt[6] & 0xffffffff))
