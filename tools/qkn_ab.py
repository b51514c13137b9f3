"""A/B timing of qknorm_rope_vt builds inside one process.   python tools/qkn_ab.py main u4 u8"""
import sys, os, ctypes as C, torch
PARTS = int(os.environ.get("VC_QKN_PARTS", "7"))   # 7 = q, k and V^T; 6 = k and V^T only
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from visualcloze_amd import hip
dev = "cuda:0"
hip.lib()
LIBDIR = os.path.dirname(hip.LIB_PATH)
L, H = int(os.environ.get("VC_QKN_L", "3968")), 24
Lp = (L + 63) // 64 * 64
qkv = (torch.randn(L, 3 * H * 128, device=dev)).to(torch.bfloat16)
qs, ks = torch.ones(128, dtype=torch.bfloat16, device=dev), torch.ones(128, dtype=torch.bfloat16, device=dev)
rope = torch.randn(L, 64, 2, device=dev)
vt = torch.empty(H, 128, Lp, dtype=torch.bfloat16, device=dev)
stream = hip.cur_stream()
libs = {}
for v in sys.argv[1:]:
    l = C.CDLL(hip.LIB_PATH if v == "main" else os.path.join(LIBDIR, f"libvcloze_hip_{v}.so"))
    l.vc_qknorm_rope_vt.restype = C.c_int
    libs[v] = l
def run(l):
    rc = l.vc_qknorm_rope_vt(C.c_void_p(qkv.data_ptr()), C.c_int64(qkv.stride(0)), C.c_int64(0), C.c_void_p(qs.data_ptr()), C.c_void_p(ks.data_ptr()),
                             C.c_void_p(0), C.c_void_p(0), C.c_int32(L), C.c_void_p(rope.data_ptr()), C.c_int64(0), C.c_void_p(vt.data_ptr()),
                             C.c_int32(1), C.c_int32(L), C.c_int32(Lp), C.c_int32(H), C.c_int32(PARTS), C.c_void_p(stream))
    assert rc == 0, rc
tot = {v: 0.0 for v in libs}
R, n = 6, 20
for r in range(R + 1):
    for v, l in libs.items():
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): run(l)
        e1.record(); torch.cuda.synchronize()
        if r > 0: tot[v] += e0.elapsed_time(e1) * 1e3 / n
nbytes = L * H * 128 * 2 * (2 * bin(PARTS & 3).count("1") + 2 * (PARTS >> 2 & 1)) + L * 512   # rows in + out, V in + V^T out, table
print(f"L={L} parts={PARTS}: " + " | ".join(f"{v} {tot[v]/R:6.1f} us {nbytes / (tot[v]/R) / 1e6:5.2f} TB/s" for v in libs))
