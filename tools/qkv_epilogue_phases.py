"""Where the qkv GEMM's epilogue spends its time per block (profiling build, s_memtime stamps as tools/gemm_phases.py): the plain
bias epilogue, the natural-order V^T epilogue, and the head-permuted one with QKNorm + RoPE of q and k (the product's).
    make -C visualcloze_amd/csrc debug && VC_HIP_LIB=visualcloze_amd/lib/libvcloze_hip_dbg.so python tools/qkv_epilogue_phases.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from visualcloze_amd import hip  # noqa: E402

dev = "cuda:0"
L, D, H = 3968, 3072, 24
Lp = (L + 63) // 64 * 64


def rnd(*s, scale=1.0):
    return (torch.randn(*s, device=dev) * scale).to(torch.bfloat16)


a, w, b = rnd(L, D), rnd(3 * D, D, scale=D ** -0.5), rnd(3 * D)
perm = hip.qkv_head_permutation(H).to(dev)
wp, bp = w[perm].contiguous(), b[perm].contiguous()
qs, ks = (1 + 0.1 * torch.randn(128, device=dev)).to(torch.bfloat16), (1 + 0.1 * torch.randn(128, device=dev)).to(torch.bfloat16)
pos = torch.arange(L, dtype=torch.float64)[:, None] * torch.linspace(0.01, 1.0, 64, dtype=torch.float64)[None]
rope = torch.stack([torch.cos(pos), torch.sin(pos)], -1).float().to(dev).contiguous()
out = torch.empty(L, 3 * D, dtype=torch.bfloat16, device=dev)
vt = torch.zeros(1, H, 128, Lp, dtype=torch.bfloat16, device=dev)
cases = {
    "bias (plain)": (hip.EPI_BIAS, lambda: hip.make_problem(a, w, b, out)),
    "qkv natural order, V^T": (hip.EPI_QKV, lambda: hip.make_problem(a, w, b, out, vt=vt, vt_col0=2 * D, vt_rpb=L, vt_row0=0)),
    "qkv permuted, V^T only": (hip.EPI_QKV, lambda: hip.make_problem(a, wp, bp, out, vt=vt, vt_col0=2 * D, vt_rpb=L, vt_row0=0, kn_heads=H)),
    "qkv permuted, k norm": (hip.EPI_QKV, lambda: hip.make_problem(a, wp, bp, out, vt=vt, vt_col0=2 * D, vt_rpb=L, vt_row0=0, kn_heads=H,
                                                                      kn_scale=ks, kn_rope=rope)),
    "qkv permuted, q + k norm (product)": (hip.EPI_QKV, lambda: hip.make_problem(a, wp, bp, out, vt=vt, vt_col0=2 * D, vt_rpb=L, vt_row0=0,
                                                                                    kn_heads=H, kn_scale=ks, qn_scale=qs, qn_prescale=True, kn_rope=rope)),
}
nblk = ((L + 255) // 256) * ((3 * D + 191) // 192)
for name, (epi, mk) in cases.items():
    p = mk()
    ts = torch.zeros(8192 + nblk * 8, dtype=torch.int64, device=dev)
    for _ in range(3):
        hip.gemm(p, epi=epi, tile_cfg=36)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        hip.gemm(p, epi=epi, tile_cfg=36)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    hip.gemm(p, epi=epi, tile_cfg=36, debug_ts=ts)
    torch.cuda.synchronize()
    t = ts[8192:].view(nblk, 8).cpu().double()
    print(f"{name}: {us:.1f} us per launch (back to back), {nblk} blocks")
    for nm, i, j in [("prologue  entry->loop", 0, 1), ("main loop", 1, 2), ("epilogue pass 1", 2, 3), ("epilogue pass 2", 3, 4), ("whole block", 0, 4)]:
        v = t[:, j] - t[:, i]
        print(f"   {nm:24s} mean {float(v.mean()):9.0f}  min {float(v.min()):9.0f}  max {float(v.max()):9.0f}  (s_memtime ticks)")
