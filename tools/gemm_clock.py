"""Driver for a rocprofv3 --pmc GRBM_GUI_ACTIVE pass: launches of one tile cfg from several library builds, so that
cycles / duration gives the shader clock each variant ran at.   python tools/gemm_clock.py main:36 nodma:36 ..."""
import sys, os, ctypes as C, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from visualcloze_amd import hip
dev = "cuda:0"
hip.lib()
LIBDIR = os.path.dirname(hip.LIB_PATH)
def rnd(*s, scale=1.0):
    return (torch.randn(*s, device=dev) * scale).to(torch.bfloat16)
M, N, K = 3968, 3072, 12288
a, w, b = rnd(M, K), rnd(N, K, scale=K ** -0.5), rnd(N)
out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
args = hip.GemmArgs(); args.nprob, args.epi = 1, 0
args.p[0] = hip.make_problem(a, w, b, out)
stream = hip.cur_stream()
torch.cuda.synchronize()
for v in sys.argv[1:]:
    name, cfg = v.split(":")
    l = C.CDLL(hip.LIB_PATH if name == "main" else os.path.join(LIBDIR, f"libvcloze_hip_{name}.so"))
    l.vc_gemm.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    for _ in range(12):
        assert l.vc_gemm(C.byref(args), int(cfg), stream) == 0
    torch.cuda.synchronize()
