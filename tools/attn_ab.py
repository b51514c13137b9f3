"""A/B timing of attention variants (library builds x kernel variant) interleaved in one process.
    python tools/attn_ab.py main:1 main:0 nodma:1 main:12:b ...      (":b" = bounded logits, VcAttention.logit_bound = 16.65: the no-running-max template; ":bp" = also q_prescaled = 1: the
    queries are taken as normalised / rotated / scaled already - the product's launch form, the stream kernel of round 6)"""
import sys, os, ctypes as C, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from visualcloze_amd import hip
dev = "cuda:0"
hip.lib()
LIBDIR = os.path.dirname(hip.LIB_PATH)
libs = {}
def getlib(name):
    if name not in libs:
        l = C.CDLL(hip.LIB_PATH if name == "main" else os.path.join(LIBDIR, f"libvcloze_hip_{name}.so"))
        l.vc_attention.restype = C.c_int
        libs[name] = l
    return libs[name]
def rnd(*s, scale=1.0):
    return (torch.randn(*s, device=dev) * scale).to(torch.bfloat16)
variants = [(v.split(":")[0], int(v.split(":")[1]), v.split(":")[2] if v.count(":") > 1 else "") for v in sys.argv[1:]]
for L in (3968, 6656):
    H = 24
    Lpad = (L + 63) // 64 * 64
    qkv = rnd(L, 3 * H * 128)
    vt = rnd(H, 128, Lpad)
    o = torch.empty(L, H * 128, dtype=torch.bfloat16, device=dev)
    stream = hip.cur_stream()
    scr = hip.attention_scratch(dev)
    def run(l, var, mode=""):
        a = hip.Attention()
        a.logit_bound = 16.65 if "b" in mode else 0.0
        a.q_prescaled = 1 if "p" in mode else 0
        a.qkv, a.ld, a.bstride, a.vt, a.out, a.ldo, a.out_bstride = qkv.data_ptr(), qkv.stride(0), 0, vt.data_ptr(), o.data_ptr(), o.stride(0), 0
        a.B, a.L, a.Lpad, a.H, a.variant = 1, L, Lpad, H, var
        a.scratch, a.scratch_bytes = scr.data_ptr(), scr.numel()
        rc = l.vc_attention(C.byref(a), C.c_void_p(stream))
        assert rc == 0, rc
    refs = {}
    for v in variants:
        print("  check", v, flush=True)
        if "p" in v[2]:          # prescaled queries: another function of the operands - compared among themselves (first one = reference)
            o.zero_(); run(getlib(v[0]), v[1], v[2]); torch.cuda.synchronize()
            ref = refs.setdefault("p", o.clone())
        else:
            if "" not in refs:
                run(getlib("main"), 1); torch.cuda.synchronize(); refs[""] = o.clone()
            ref = refs[""]
            o.zero_(); run(getlib(v[0]), v[1], v[2]); torch.cuda.synchronize()
        if not torch.equal(o, ref):
            d = (o.float() - ref.float()).abs().max().item()
            rel = ((o.float() - ref.float()).norm() / ref.float().norm()).item()
            print(f"  differs {v}: max abs diff vs main:1 = {d:.4g}, rel-L2 {rel:.3g}, nan={bool(torch.isnan(o.float()).any())}")
    tot = {v: 0.0 for v in variants}
    R, n = 6, 10
    for r in range(R + 1):
        for v in variants:
            l = getlib(v[0])
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(n): run(l, v[1], v[2])
            e1.record(); torch.cuda.synchronize()
            if r > 0: tot[v] += e0.elapsed_time(e1) * 1e3 / n
    fl = 4.0 * L * L * H * 128
    print(f"L={L}: " + " | ".join(f"{v[0]}:{v[1]}{v[2]} {tot[v]/R:6.1f} us {fl/(tot[v]/R)/1e6:5.0f} TF" for v in variants), flush=True)
