import torch, sys
sys.path.insert(0, "/root/repo")
from stego_amd import capi, modules as M
dev = torch.device("cuda:0")
gen = M._device_generator(dev)
for (B, S, n) in [(32, 11, 5), (4, 11, 5), (300, 5, 2)]:
    shape = [B, S, S, 2]
    torch.manual_seed(5)
    st = gen.get_state(); o0 = gen.get_offset()
    ref = M._torch_draws(shape, n, B, dev)
    print("B", B, "torch offset advance", gen.get_offset() - o0)
    for v in range(8):
        gen.set_state(st)
        c1, c2, p = capi.ref_draws(gen, shape, n, B, v, dev)
        adv = gen.get_offset() - o0
        e1 = int((c1 != ref[0] * 2 - 1).sum()); e2 = int((c2 != ref[1] * 2 - 1).sum())
        ep = int((p != M._unfix(torch.stack(ref[2:]))).sum())
        print("  variant", v, "adv", adv, "mismatch c1 c2 perms:", e1, e2, ep, "of", c1.numel(), p.numel())
    if B == 32:
        print(ref[0].flatten()[:4].tolist(), ((c1 + 1) / 2).flatten()[:4].tolist())
        print(ref[2].tolist())
