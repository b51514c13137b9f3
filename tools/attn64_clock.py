"""Where a workgroup of the one-wave-per-SIMD attention kernel spends its ticks: per work item prologue (DMA + query load up to
the first barrier), first tile (not overlapped), steady-state loop, epilogue (stores complete) - from the s_memtime stamps of the
profiling build:   make -C visualcloze_amd/csrc debug; VC_HIP_LIB=visualcloze_amd/lib/libvcloze_hip_dbg.so python tools/attn64_clock.py
Optional arguments: lengths (default 3968 6656).  Launch form: the product's (bounded logits, prescaled queries = the stream
kernel); VC_CLOCK_FORM=runmax times the running-max template instead."""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from visualcloze_amd import hip
dev = "cuda:0"
H = 24
KW = {} if os.environ.get("VC_CLOCK_FORM") == "runmax" else dict(logit_bound=16.65, q_prescaled=True)
Ls = [int(x) for x in sys.argv[1:]] or [3968, 6656]
for L in Ls:
    Lp = (L + 63) // 64 * 64
    qkv = torch.randn(L, 3 * H * 128, device=dev).to(torch.bfloat16)
    vt = torch.randn(H, 128, Lp, device=dev).to(torch.bfloat16)
    o = torch.empty(L, H * 128, dtype=torch.bfloat16, device=dev)
    for variant in ([int(os.environ["VC_CLOCK_VARIANT"])] if "VC_CLOCK_VARIANT" in os.environ else [8, 12]):
        ts = torch.zeros(256, 32, dtype=torch.int64, device=dev)
        for _ in range(3):
            hip.attention(qkv, vt, o, L, H, variant=variant, **KW)
        torch.cuda.synchronize()
        hip.lib().vc_debug_set_attn_ts(ctypes.c_void_p(ts.data_ptr()))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); hip.attention(qkv, vt, o, L, H, variant=variant, **KW); e1.record()
        torch.cuda.synchronize()
        hip.lib().vc_debug_set_attn_ts(ctypes.c_void_p(0))
        us = e0.elapsed_time(e1) * 1e3
        t = ts.cpu()
        t = t[t[:, 2] > 0]
        span = (t[:, 1].max() - t[:, 0].min()).item()
        life = (t[:, 1] - t[:, 0]).double()
        print(f"L={L} variant {variant}: {us:7.1f} us (event, incl. merge launch); first start -> last end {span} ticks; workgroup life mean {life.mean():.0f} "
              f"min {life.min():.0f} max {life.max():.0f}; tiles per workgroup {t[:, 2].double().mean():.1f}; ticks per tile {(life / t[:, 2].double()).mean():.0f}")
        start_skew = (t[:, 0] - t[:, 0].min()).double()
        end_skew = (t[:, 1].max() - t[:, 1]).double()
        print(f"    start skew mean {start_skew.mean():.0f} max {start_skew.max():.0f}; idle before the last workgroup ends: mean {end_skew.mean():.0f} max {end_skew.max():.0f}")
        nseg = int(t[:, 3].max())
        for k in range(min(nseg, 3)):
            s = t[t[:, 3] > k][:, 8 + 8 * k: 16 + 8 * k].double()
            if not len(s): continue
            pro, first, loop, epi, tiles = s[:, 1] - s[:, 0], s[:, 2] - s[:, 1], s[:, 3] - s[:, 2], s[:, 4] - s[:, 3], s[:, 5]
            print(f"    item {k} ({len(s)} workgroups, {tiles.mean():.1f} tiles): prologue {pro.mean():.0f} (max {pro.max():.0f}) | first tile {first.mean():.0f} | "
                  f"loop {loop.mean():.0f} = {(loop / tiles).mean():.0f} per tile (min {(loop / tiles).min():.0f} max {(loop / tiles).max():.0f}) | epilogue {epi.mean():.0f} (max {epi.max():.0f})")
