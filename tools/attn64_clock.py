"""Shader clock and cycles per KV tile of the one-wave-per-SIMD attention kernel under its real load (needs the profiling
build: make -C visualcloze_amd/csrc debug; VC_HIP_LIB=visualcloze_amd/lib/libvcloze_hip_dbg.so python tools/attn64_clock.py)."""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from visualcloze_amd import hip
dev = "cuda:0"
H = 24
for L in (3968, 6656):
    Lp = (L + 63) // 64 * 64
    qkv = torch.randn(L, 3 * H * 128, device=dev).to(torch.bfloat16)
    vt = torch.randn(H, 128, Lp, device=dev).to(torch.bfloat16)
    o = torch.empty(L, H * 128, dtype=torch.bfloat16, device=dev)
    for variant in (8, 12):
        ts = torch.zeros(256, 4, dtype=torch.int64, device=dev)
        for _ in range(3):
            hip.attention(qkv, vt, o, L, H, variant=variant)
        torch.cuda.synchronize()
        hip.lib().vc_debug_set_attn_ts(ctypes.c_void_p(ts.data_ptr()))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); hip.attention(qkv, vt, o, L, H, variant=variant); e1.record()
        torch.cuda.synchronize()
        hip.lib().vc_debug_set_attn_ts(ctypes.c_void_p(0))
        us = e0.elapsed_time(e1) * 1e3
        t = ts.cpu()
        t = t[t[:, 2] > 0]
        span = (t[:, 1].max() - t[:, 0].min()).item()
        per_tile = ((t[:, 1] - t[:, 0]).double() / t[:, 2].double())
        print(f"L={L} variant {variant}: {us:7.1f} us (event, incl. launch); first start -> last end {span} ticks = {span / us / 1e3:.2f} GHz if ticks are "
              f"shader cycles; ticks per tile: mean {per_tile.mean():.0f} min {per_tile.min():.0f} max {per_tile.max():.0f}; tiles per block {t[:, 2].double().mean():.1f}")
