"""Attention throughput vs sequence length (work items per resident slot), to separate block-round quantisation from
the steady-state rate of the kernel.     python tools/attn_quant.py [variant ...]"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from visualcloze_amd import hip
dev = "cuda:0"
variants = [int(v) for v in sys.argv[1:]] or [3]
H = 24
BOUND = float(os.environ.get("VC_ATTN_BOUND", "0"))      # > 0: the bounded-logit instantiation of variants 8 / 12 (the product's at unit norm scales: 16.65)
for L in [int(x) for x in os.environ.get("VC_ATTN_L", "1664,2688,3968,4608,5376,6656,7424,8064").split(",")]:
    Lpad = (L + 63) // 64 * 64
    qkv = torch.randn(L, 3 * H * 128, device=dev).to(torch.bfloat16)
    if BOUND > 0:      # rows of bounded norm, as QKNorm leaves them
        q3 = qkv.view(L, 3 * H, 128).float()
        qkv = (q3 / q3.pow(2).mean(-1, keepdim=True).sqrt()).to(torch.bfloat16).view(L, 3 * H * 128)
    vt = torch.randn(H, 128, Lpad, device=dev).to(torch.bfloat16)
    o = torch.empty(L, H * 128, dtype=torch.bfloat16, device=dev)
    items = (L + 127) // 128 * H
    line = f"L={L:5d} items={items:5d} ({items / 512:.2f} per slot): "
    for v in variants:
        for _ in range(3):
            hip.attention(qkv, vt, o, L, H, variant=v, logit_bound=BOUND if v & 8 else 0.0)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        best = 1e9
        for r in range(5):
            e0.record()
            for _ in range(10):
                hip.attention(qkv, vt, o, L, H, variant=v, logit_bound=BOUND if v & 8 else 0.0)
            e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) * 100)
        fl = 4.0 * L * L * H * 128
        line += f" v{v}: {best:7.1f} us {fl / best / 1e6:6.0f} TF |"
    print(line, flush=True)
