"""Where a GEMM launch spends its time per block (profiling build): s_memtime at kernel entry, main-loop start, main-loop
end, after epilogue pass 1, at exit, for every block.  VC_HIP_LIB=.../libvcloze_hip_dbg.so python tools/gemm_phases.py"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from visualcloze_amd import hip
dev = "cuda:0"
def rnd(*s, scale=1.0):
    return (torch.randn(*s, device=dev) * scale).to(torch.bfloat16)
for (M, N, K, epi) in [(3968, 3072, 3072, 2), (3968, 9216, 3072, 0), (3968, 12288, 3072, 1), (3968, 3072, 12288, 2)]:
    a, w, b = rnd(M, K), rnd(N, K, scale=K ** -0.5), rnd(N)
    out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    res, gate = rnd(M, N), rnd(N)
    kw = dict(res=res, gate=gate, rows_per_batch=M) if epi == 2 else {}
    p = hip.make_problem(a, w, b, out, **kw)
    nblk = ((M + 255) // 256) * ((N + 191) // 192)
    ts = torch.zeros(8192 + nblk * 8, dtype=torch.int64, device=dev)
    for _ in range(3):
        hip.gemm(p, epi=epi, tile_cfg=36)
    torch.cuda.synchronize()
    hip.gemm(p, epi=epi, tile_cfg=36, debug_ts=ts)
    torch.cuda.synchronize()
    t = ts[8192:].view(nblk, 8).cpu().double()
    t0 = t[:, 0].min()
    rel = t - t0
    print(f"M={M} N={N} K={K} epi={epi}: {nblk} blocks, kernel span {float((t[:, 4].max() - t0)):.0f} cycles")
    for name, i, j in [("entry (rel. first block)", None, 0), ("prologue  entry->loop", 0, 1), ("main loop", 1, 2), ("epilogue pass 1", 2, 3), ("epilogue pass 2", 3, 4)]:
        v = rel[:, j] if i is None else (t[:, j] - t[:, i])
        print(f"   {name:26s} mean {float(v.mean()):9.0f}  min {float(v.min()):9.0f}  max {float(v.max()):9.0f}")
    # same-CU succession: HW_ID (cu_id bits 8-11, sh 12, se 13-15 on gfx9) + XCC_ID identify the CU; s_memtime is per XCD
    ids = ts[8192:].view(nblk, 8)[:, 5].cpu()
    key = [(int(x) >> 32, (int(x) & 0xffffffff) >> 8 & 0xff) for x in ids]
    bycu = {}
    for b in range(nblk):
        bycu.setdefault(key[b], []).append((float(t[b, 0]), float(t[b, 4]), b))
    gaps = []
    for k, lst in bycu.items():
        lst.sort()
        for i in range(1, len(lst)):
            gaps.append(lst[i][0] - lst[i - 1][1])
    if gaps:
        g = torch.tensor(gaps)
        print(f"   {len(bycu)} distinct CUs; exit -> next block's entry on the same CU: mean {float(g.mean()):.0f} median {float(g.median()):.0f} min {float(g.min()):.0f} max {float(g.max()):.0f} cycles ({len(gaps)} successions)")
    else:
        print(f"   {len(bycu)} distinct CUs, one block each")
