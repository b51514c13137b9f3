"""A/B timing of GEMM variants inside ONE process: several builds of the library (make variant NAME=..) and tile cfgs are
interleaved round-robin so clock / thermal drift hits all of them alike.
    python tools/gemm_ab.py main:36 main:20 v1:36 ...      (lib 'main' = the product build)"""
import sys, os, ctypes as C, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from visualcloze_amd import hip
dev = "cuda:0"
hip.lib()
LIBDIR = os.path.dirname(hip.LIB_PATH)
libs = {}
def getlib(name):
    if name not in libs:
        path = hip.LIB_PATH if name == "main" else os.path.join(LIBDIR, f"libvcloze_hip_{name}.so")
        l = C.CDLL(path)
        l.vc_gemm.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        l.vc_gemm.restype = C.c_int
        libs[name] = l
    return libs[name]
def rnd(*s, scale=1.0):
    return (torch.randn(*s, device=dev) * scale).to(torch.bfloat16)
variants = [(v.split(":")[0], int(v.split(":")[1])) for v in sys.argv[1:]]
shapes = [(3968, 3072, 12288, 2), (3968, 9216, 3072, 0), (3968, 12288, 3072, 1), (3968, 3072, 3072, 2)]
stream = hip.cur_stream()
for (M, N, K, epi) in shapes:
    a, w, b = rnd(M, K), rnd(N, K, scale=K ** -0.5), rnd(N)
    out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    res, gate = rnd(M, N), rnd(N)
    args = hip.GemmArgs()
    args.nprob, args.epi = 1, epi
    args.p[0] = hip.make_problem(a, w, b, out, res=res if epi == 2 else None, gate=gate if epi == 2 else None, rows_per_batch=M if epi == 2 else None)
    tot = {v: 0.0 for v in variants}
    R, n = 6, 10
    for r in range(R + 1):
        for v in variants:
            l = getlib(v[0])
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(n):
                rc = l.vc_gemm(C.byref(args), v[1], stream)
                assert rc == 0, rc
            e1.record(); torch.cuda.synchronize()
            if r > 0:
                tot[v] += e0.elapsed_time(e1) * 1e3 / n
    print(f"M={M} N={N} K={K} epi={epi}: " + " | ".join(f"{v[0]}:{v[1]} {tot[v]/R:6.1f} us {2.0*M*N*K/(tot[v]/R)/1e6:6.0f} TF" for v in variants), flush=True)
