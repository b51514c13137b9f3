"""Is the GEMM main loop bound by the memory path or by the core?  Times the same launch with normal operands and
with row stride 0 (every row aliases one 128-B line per K-step, so all LDS-DMA requests hit L1/L2)."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from visualcloze_amd import hip
dev = "cuda:0"
def rnd(*s, scale=1.0):
    return (torch.randn(*s, device=dev) * scale).to(torch.bfloat16)
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
import os
SHAPES = [(3968, 3072, 12288), (3968, 9216, 3072), (3968, 12288, 3072)] if not os.environ.get("SWAP") else [(3072, 3968, 12288), (12288, 3968, 3072)]
for (M, N, K) in SHAPES:
    a, w, b = rnd(M, K), rnd(N, K, scale=K ** -0.5), rnd(N)
    a0, w0 = a[:1].expand(M, K), w[:1].expand(N, K)
    out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    for cfg in [int(c) for c in sys.argv[1].split(",")]:
        res = []
        modes = [(a, w, "normal"), (a0, w, "A hot"), (a, w0, "W hot"), (a0, w0, "both hot")]
        if os.environ.get("NORMAL_ONLY"):
            modes = [(a, w, "normal"), (a, w, "again")]
        for (aa, ww, tag) in modes:
            p = hip.make_problem(aa, ww, b, out)
            us = timeit(lambda: hip.gemm(p, epi=0, tile_cfg=cfg))
            res.append(f"{tag} {us:7.1f} us {2.0*M*N*K/us/1e6:7.1f} TF")
        print(f"M={M} N={N} K={K} cfg {cfg}: " + " | ".join(res), flush=True)
