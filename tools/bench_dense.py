#!/usr/bin/env python
"""Dense correspondence tensor (einsum 'bchw,bcij->bhwij' at full map resolution) on the HIP kernel,
next to torch's einsum on the same GPU (rocBLAS/hipBLASLt fp32) and on the host CPU (bounded sample).  One JSON line."""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from stego_amd import capi  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    out = []
    for name, (B, C, H) in {"vits8_224": (32, 384, 28), "vitb8_320": (32, 768, 40)}.items():
        g = torch.Generator(device=dev).manual_seed(7)
        a = torch.randn(B, H, H, C, device=dev, generator=g).permute(0, 3, 1, 2)      # channels-last views
        b = torch.randn(B, H, H, C, device=dev, generator=g).permute(0, 3, 1, 2)

        def t_ms(fn, iters=20):
            fn(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(iters):
                fn()
            e1.record(); torch.cuda.synchronize()
            return e0.elapsed_time(e1) / iters

        ms = t_ms(lambda: capi.dense_corr(a, b, normalize=True))
        an, bn = torch.nn.functional.normalize(a, dim=1, eps=1e-10), torch.nn.functional.normalize(b, dim=1, eps=1e-10)
        ms_torch = t_ms(lambda: torch.einsum("nchw,ncij->nhwij", torch.nn.functional.normalize(a, dim=1, eps=1e-10),
                                             torch.nn.functional.normalize(b, dim=1, eps=1e-10)))
        err = float((capi.dense_corr(a, b, normalize=True) - torch.einsum("nchw,ncij->nhwij", an, bn)).abs().max())
        flops = 2.0 * B * (H * H) ** 2 * C
        obytes = 4.0 * B * (H * H) ** 2
        ac, bc = an[:4].cpu(), bn[:4].cpu()
        torch.set_num_threads(16)
        t0 = time.perf_counter()
        torch.einsum("nchw,ncij->nhwij", ac, bc)
        cpu_s = time.perf_counter() - t0
        out.append({"workload": "%s B=%d C=%d %dx%d" % (name, B, C, H, H), "ms": ms, "pairs_per_s": B / (ms * 1e-3),
                    "effective_TFLOPs": flops / (ms * 1e-3) / 1e12, "output_write_GBps": obytes / (ms * 1e-3) / 1e9,
                    "torch_einsum_same_gpu_ms": ms_torch, "max_abs_diff_vs_torch": err,
                    "cpu_einsum_pairs_per_s_16thr": 4 / cpu_s})
    print(json.dumps({"metric": "dense correspondence tensor [B,hw,hw], norm + einsum", "results": out}))


if __name__ == "__main__":
    main()
