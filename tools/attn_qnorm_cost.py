"""What the in-kernel query QKNorm + RoPE prologue of attn64_kernel costs per launch: the product's variant (12, bounded logits) on
the same operands with and without `q_norm`, back to back, for the BASELINE lengths.   python tools/attn_qnorm_cost.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from visualcloze_amd import hip
dev, H = "cuda:0", 24
for L in (3968, 3752, 4608, 6656, 7424, 3968, 1664):
    Lp = (L + 63) // 64 * 64
    q3 = torch.randn(L, 3 * H, 128, device=dev)
    qkv = (q3 / q3.pow(2).mean(-1, keepdim=True).sqrt()).to(torch.bfloat16).view(L, 3 * H * 128).contiguous()
    vt = torch.randn(H, 128, Lp, device=dev).to(torch.bfloat16)
    o = torch.empty(L, H * 128, dtype=torch.bfloat16, device=dev)
    scale = torch.ones(128, dtype=torch.bfloat16, device=dev)
    ang = torch.rand(L, 64, device=dev) * 6.28
    rope = torch.stack((torch.cos(ang), torch.sin(ang)), -1).float().contiguous()
    res = {}
    for tag, qn in (("plain", None), ("q_norm", (scale, None, 0, rope))):
        f = lambda: hip.attention(qkv, vt, o, L, H, variant=12, logit_bound=16.65, q_norm=qn)
        for _ in range(3): f()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        best = 1e9
        for r in range(5):
            e0.record()
            for _ in range(10): f()
            e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) * 100)
        res[tag] = best
    print(f"L={L}: without query norm {res['plain']:.1f} us, with {res['q_norm']:.1f} us -> prologue {res['q_norm'] - res['plain']:.1f} us per launch ({(res['q_norm'] / res['plain'] - 1) * 100:.1f} %)", flush=True)
