"""Per-segment timeline of the ping-pong GEMM main loop (s_memtime stamps of block 0, waves 0 and 4)."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from visualcloze_amd import hip
dev = "cuda:0"
def rnd(*s, scale=1.0):
    return (torch.randn(*s, device=dev) * scale).to(torch.bfloat16)
M, N, K = 3968, 9216, 3072
a, w, b = rnd(M, K), rnd(N, K, scale=K ** -0.5), rnd(N)
out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
for cfg in [int(c) for c in sys.argv[1].split(",")]:
    ts = torch.zeros(2, 4096, dtype=torch.int64, device=dev)
    p = hip.make_problem(a, w, b, out)
    for _ in range(2):
        hip.gemm(p, epi=0, tile_cfg=cfg)
    torch.cuda.synchronize()
    ts.zero_()
    hip.gemm(p, epi=0, tile_cfg=cfg, debug_ts=ts)
    torch.cuda.synchronize()
    t = ts.cpu()
    print("cfg", cfg)
    for g in range(2):
        v = t[g]
        n = int((v != 0).sum())
        v = v[:n]
        base = int(t[0][0])
        rel = [(int(x) - base) for x in v]
        # 6 stamps per segment pair: top, reads issued, waited, after bar, mfma issued, waited2
        print(f" group {g}: {n} stamps; per K-tile (12 stamps) deltas for K-tiles 8..12:")
        for kt in range(8, 13):
            row = rel[kt * 12:(kt + 1) * 12 + 1]
            if len(row) < 13: break
            d = [row[i + 1] - row[i] for i in range(12)]
            print(f"  kt{kt}: start {row[0]:7d} | M0: issue+reads {d[0]:4d} wait {d[1]:4d} bar {d[2]:4d} | C0: mfma {d[3]:4d} wait {d[4]:4d} bar {d[5]:4d} | "
                  f"M1: reads {d[6]:4d} wait {d[7]:4d} bar {d[8]:4d} | C1: mfma {d[9]:4d} wait {d[10]:4d} bar {d[11]:4d} | total {row[12]-row[0]}")
