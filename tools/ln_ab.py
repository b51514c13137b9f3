"""A/B timing of ln_modulate builds inside one process (rows per wave etc.).   python tools/ln_ab.py main rpw1 rpw4
Two row sets (the img / txt streams of a DoubleStreamBlock) in one launch, as the step graph calls it; checks every build
against the first one bit for bit."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from visualcloze_amd import hip  # noqa: E402

dev = "cuda:0"
hip.lib()
LIBDIR = os.path.dirname(hip.LIB_PATH)
T, N, D = 512, int(os.environ.get("VC_LN_N", "3456")), 3072
xi, xt = torch.randn(N, D, device=dev).to(torch.bfloat16), torch.randn(T, D, device=dev).to(torch.bfloat16)
mod = (0.1 * torch.randn(4, D, device=dev)).to(torch.bfloat16)
stream = hip.cur_stream()
libs = {}
for v in sys.argv[1:]:
    l = C.CDLL(hip.LIB_PATH if v == "main" else os.path.join(LIBDIR, f"libvcloze_hip_{v}.so"))
    l.vc_ln_modulate2.restype = C.c_int
    l.vc_ln_modulate2.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_int64, C.c_void_p]
    libs[v] = l


def run(l, yi, yt):
    a = hip.LnStream(xi.data_ptr(), D, yi.data_ptr(), D, mod[0].data_ptr(), mod[1].data_ptr(), N, N)
    b = hip.LnStream(xt.data_ptr(), D, yt.data_ptr(), D, mod[2].data_ptr(), mod[3].data_ptr(), T, T)
    rc = l.vc_ln_modulate2(C.byref(a), C.byref(b), 0, D, None, 0, stream)
    assert rc == 0, rc


outs = {}
for v, l in libs.items():
    yi, yt = torch.full_like(xi, float("nan")), torch.full_like(xt, float("nan"))
    run(l, yi, yt)
    torch.cuda.synchronize()
    outs[v] = (yi, yt)
first = next(iter(outs))
for v, (yi, yt) in outs.items():     # bit for bit, or how far apart (f32 rounding: e.g. another fma contraction of the variance sum)
    assert torch.isfinite(yi.float()).all(), v
    same = torch.equal(yi, outs[first][0]) and torch.equal(yt, outs[first][1])
    d = (yi.float() - outs[first][0].float())
    print(f"{v} vs {first}: {'bit-identical' if same else f'{(d != 0).float().mean().item():.2e} of the elements differ, max |d| {d.abs().max().item():.3e}'}")
tot = {v: 0.0 for v in libs}
R, n = 6, 20
yi, yt = torch.empty_like(xi), torch.empty_like(xt)
for r in range(R + 1):
    for v, l in libs.items():
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            run(l, yi, yt)
        e1.record()
        torch.cuda.synchronize()
        if r > 0:
            tot[v] += e0.elapsed_time(e1) * 1e3 / n
nbytes = (N + T) * D * 2 * 2
print(f"rows={N}+{T}: " + " | ".join(f"{v} {tot[v]/R:6.1f} us {nbytes / (tot[v]/R) / 1e6:5.2f} TB/s" for v in libs))
