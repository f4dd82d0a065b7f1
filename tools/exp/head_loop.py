"""Runs the native head forward + backward a few times (for rocprofv3 --kernel-trace --stats)."""
import os, sys, torch, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
warnings.simplefilter("ignore")
from stego_amd import capi
dev = torch.device("cuda:0")
B, HW, C, K = int(os.environ.get("B", 64)), 784, 384, 70
g = torch.Generator(device=dev).manual_seed(5)
tokens = torch.randn(B, 1 + HW, C, device=dev, generator=g)[:, 1:, :]
m = [(torch.rand(B, C, device=dev, generator=g) > 0.1).float() / 0.9 for _ in range(3)]
w1 = torch.randn(K, C, device=dev) * 0.05; b1 = torch.randn(K, device=dev) * 0.05
w21 = torch.randn(C, C, device=dev) * 0.05; b21 = torch.randn(C, device=dev) * 0.05
w22 = torch.randn(K, C, device=dev) * 0.05; b22 = torch.randn(K, device=dev) * 0.05
G = torch.randn(B, HW, K, device=dev) / HW
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 10):
    code, feats, saved_h = capi.head_fwd(tokens, tuple(m), w1, b1, w21, b21, w22, b22, True, True)
    capi.head_bwd(tokens, (m[0], m[1], None), saved_h, w22, G, K)
torch.cuda.synchronize()
