"""Driver for rocprofv3 --pmc: 3 launches each of normal / A hot / W hot / both hot (see gemm_l2hot.py) for one tile cfg."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from visualcloze_amd import hip
dev = "cuda:0"
def rnd(*s, scale=1.0):
    return (torch.randn(*s, device=dev) * scale).to(torch.bfloat16)
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 36
M, N, K = 3968, 3072, 12288
a, w, b = rnd(M, K), rnd(N, K, scale=K ** -0.5), rnd(N)
a0, w0 = a[:1].expand(M, K), w[:1].expand(N, K)
out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
torch.cuda.synchronize()
for (aa, ww) in [(a, w), (a0, w), (a, w0), (a0, w0)]:
    p = hip.make_problem(aa, ww, b, out)
    for _ in range(3):
        hip.gemm(p, epi=0, tile_cfg=cfg)
    torch.cuda.synchronize()
