"""Tiny driver for rocprofv3 PMC passes: a few launches of chosen GEMM configs / attention on cfg-2 shapes."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from visualcloze_amd import hip
dev = "cuda:0"
def rnd(*s, scale=1.0):
    return (torch.randn(*s, device=dev) * scale).to(torch.bfloat16)
cfgs = [int(c) for c in sys.argv[1].split(",")] if len(sys.argv) > 1 else [4, 20, 3, 19]
M, N, K = 3968, 9216, 3072
a, w, b = rnd(M, K), rnd(N, K, scale=K ** -0.5), rnd(N)
out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
for c in cfgs:
    p = hip.make_problem(a, w, b, out)
    for _ in range(3):
        hip.gemm(p, epi=0, tile_cfg=c)
torch.cuda.synchronize()
if "attn" in sys.argv:
    L, H = 3968, 24
    qkv = rnd(L, 3 * H * 128)
    vt = rnd(H, 128, L)
    o = torch.empty(L, H * 128, dtype=torch.bfloat16, device=dev)
    for v in (0, 1):
        for _ in range(3):
            hip.attention(qkv, vt, o, L, H, variant=v)
    torch.cuda.synchronize()
