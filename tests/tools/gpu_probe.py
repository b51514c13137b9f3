"""GPU-box diagnostic: run every HIP op against its torch reference over a sweep of shapes, print error
statistics and timings, never abort on a failing case.  Output goes to stdout (tee it into gpurun_out/)."""
import json
import math
import os
import sys
import time
import traceback

import torch

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
from tests import ref_ops as R  # noqa: E402
from visualcloze_amd import hip  # noqa: E402

dev = torch.device("cuda:0")
RESULTS = []


def rnd(*shape, scale=1.0, seed=0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(torch.bfloat16).to(dev)


def report(name, got, ref, tol=2e-2, extra=None):
    got, ref = got.float(), ref.float()
    err = (got - ref).abs()
    denom = ref.abs().max().item() + 1e-12
    rel_l2 = ((got - ref).norm() / (ref.norm() + 1e-12)).item()
    bad = int((err > tol * denom).sum().item())
    ok = bool(torch.isfinite(got).all().item()) and rel_l2 < tol and bad == 0
    rec = dict(name=name, ok=ok, max_abs=err.max().item(), ref_max=denom, rel_l2=rel_l2, n_bad=bad, n=got.numel())
    if extra:
        rec.update(extra)
    RESULTS.append(rec)
    print(("PASS " if ok else "FAIL ") + json.dumps(rec), flush=True)
    if not ok and bad:
        idx = (err > tol * denom).nonzero()[:8]
        for i in idx:
            t = tuple(i.tolist())
            print("     bad at", t, "got", got[t].item(), "ref", ref[t].item(), flush=True)
    return ok


def timeit(fn, iters=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = hip.Event(), hip.Event()
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    return e0.elapsed_ms(e1) / iters


def guard(fn):
    def w(*a, **k):
        try:
            return fn(*a, **k)
        except Exception:
            print("EXC in", fn.__name__, a, k)
            traceback.print_exc()
            RESULTS.append(dict(name=fn.__name__ + str(a), ok=False, exc=True))
    return w


@guard
def probe_gemm(M, N, K, epi, cfg, lda_pad=0, time_it=False, ldw_pad=0):
    a_full = rnd(M, K + lda_pad, seed=1)
    a = a_full[:, :K]
    if lda_pad < 0:
        a = rnd(1, K, seed=1).expand(M, K)      # stride-0 rows: the A operand stays cache-hot (traffic experiment)
    w = rnd(N, K + ldw_pad, scale=K ** -0.5, seed=2)[:, :K]
    bias = rnd(N, seed=3)
    res = rnd(M, N, seed=4)
    gate = rnd(N, seed=5)
    out = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device=dev)
    p = hip.make_problem(a, w, bias, out, res=res if epi == 2 else None, gate=gate if epi == 2 else None)
    hip.gemm(p, epi=epi, tile_cfg=cfg)
    torch.cuda.synchronize()
    ref = R.gemm_ref(a, w, bias, epi, res, gate)
    extra = None
    if time_it:
        ms = timeit(lambda: hip.gemm(p, epi=epi, tile_cfg=cfg))
        extra = dict(ms=ms, tflops=2.0 * M * N * K / ms / 1e9)
    report(f"gemm M{M} N{N} K{K} epi{epi} cfg{cfg} pad{lda_pad}/{ldw_pad}", out, ref, extra=extra)


@guard
def probe_gemm_grouped(cfg):
    M1, M2, N, K = 300, 136, 384, 256
    a1, a2 = rnd(M1, K, seed=1), rnd(M2, K, seed=2)
    w1, w2 = rnd(N, K, scale=K ** -0.5, seed=3), rnd(N, K, scale=K ** -0.5, seed=4)
    b1, b2 = rnd(N, seed=5), rnd(N, seed=6)
    big = torch.full((M1 + M2, N), float("nan"), dtype=torch.bfloat16, device=dev)
    p1 = hip.make_problem(a1, w1, b1, big[M2:])
    p2 = hip.make_problem(a2, w2, b2, big[:M2])
    hip.gemm([p1, p2], epi=0, tile_cfg=cfg)
    torch.cuda.synchronize()
    ref = torch.cat([R.gemm_ref(a2, w2, b2, 0), R.gemm_ref(a1, w1, b1, 0)])
    report(f"gemm grouped cfg{cfg}", big, ref)


@guard
def probe_ln(rows, D):
    x = rnd(rows, D, scale=2.0, seed=1) + 0.5
    sh, sc = rnd(D, seed=2), rnd(D, scale=0.3, seed=3)
    out = hip.ln_modulate(x, sh, sc)
    torch.cuda.synchronize()
    report(f"ln_modulate rows{rows} D{D}", out, R.ln_modulate_ref(x, sh, sc))


def rope_table(L):
    pos = torch.arange(L, dtype=torch.float64)[:, None] * torch.linspace(0.01, 1.0, 64, dtype=torch.float64)[None]
    return torch.stack([torch.cos(pos), torch.sin(pos)], -1).float().to(dev).contiguous()


@guard
def probe_qknorm_attn(L, H, extra_cols=0, kv_len=None, variant=0, time_it=False):
    ld = 3 * H * 128 + extra_cols
    qkv = rnd(L, ld, seed=7)
    qs, ks = (1 + 0.1 * rnd(128, seed=8)).to(torch.bfloat16), (1 + 0.1 * rnd(128, seed=9)).to(torch.bfloat16)
    rope = rope_table(L)
    Lpad = (L + 63) // 64 * 64
    vt = torch.full((H, 128, Lpad), float("nan"), dtype=torch.bfloat16, device=dev)
    qref, kref, vtref = R.qknorm_rope_ref(qkv, qs, ks, rope, H)
    work = qkv.clone()
    hip.qknorm_rope_vt(work, qs, ks, rope, vt, L, H)
    torch.cuda.synchronize()
    got = work[:, : 3 * H * 128].float().reshape(L, 3, H, 128)
    report(f"qknorm q L{L} H{H}", got[:, 0], qref)
    report(f"qknorm k L{L} H{H}", got[:, 1], kref)
    report(f"vt L{L} H{H}", vt[:, :, :L], vtref, tol=1e-6)
    if Lpad > L:
        report(f"vt pad zero L{L}", vt[:, :, L:], torch.zeros_like(vt[:, :, L:]).float(), tol=1e-6)
    out = torch.full((L, H * 128), float("nan"), dtype=torch.bfloat16, device=dev)
    kvl = None if kv_len is None else torch.tensor([kv_len], dtype=torch.int32, device=dev)
    hip.attention(work, vt, out, L, H, kv_len=kvl, variant=variant)
    torch.cuda.synchronize()
    v = qkv[:, 2 * H * 128: 3 * H * 128].float().reshape(L, H, 128)
    ref = R.attention_ref(got[:, 0], got[:, 1], v, kv_len)
    extra = None
    if time_it:
        ms = timeit(lambda: hip.attention(work, vt, out, L, H, kv_len=kvl, variant=variant))
        ms2 = timeit(lambda: hip.qknorm_rope_vt(work, qs, ks, rope, vt, L, H))
        extra = dict(ms=ms, tflops=4.0 * L * L * 128 * H / ms / 1e9, qknorm_ms=ms2)
    report(f"attention L{L} H{H} kv{kv_len} var{variant} ld{ld}", out, ref, extra=extra)


@guard
def probe_elementwise():
    import oracle.flux_oracle as O
    t = torch.tensor([0.0, 0.348, 1.0], device=dev)
    fr = O.temb_freqs().to(dev)
    out = torch.empty(3, 256, dtype=torch.bfloat16, device=dev)
    hip.timestep_embedding(t, fr, out)
    torch.cuda.synchronize()
    report("temb", out, O.timestep_embedding(t.cpu()).to(dev), tol=1e-2)
    g = torch.tensor([30.0], device=dev)
    out = torch.empty(1, 256, dtype=torch.bfloat16, device=dev)
    hip.timestep_embedding(g, fr, out, round_t_bf16=True)
    torch.cuda.synchronize()
    report("temb g30 bf16", out, O.timestep_embedding(g.cpu(), t_is_bf16=True).to(dev), tol=1e-2)
    x = rnd(1000, seed=1)
    report("silu", hip.silu(x), R.rb(torch.nn.functional.silu(x.float())))
    a, b, c = rnd(777, seed=2), rnd(777, seed=3), rnd(777, seed=4)
    report("add3", hip.add3(a, b, c), R.rb(R.rb(a.float() + b.float()) + c.float()))
    report("add2", hip.add3(a, b), R.rb(a.float() + b.float()))
    xx, cc = rnd(50, 64, seed=5), rnd(50, 320, seed=6)
    o = torch.empty(50, 384, dtype=torch.bfloat16, device=dev)
    hip.concat_cols(xx, cc, o)
    report("concat", o, torch.cat([xx, cc], -1), tol=1e-6)
    xs, v = rnd(999, seed=7), rnd(999, seed=8)
    dts = torch.tensor([0.1, 0.037, 0.2], device=dev)
    step = torch.tensor([1], dtype=torch.int32, device=dev)
    ref = R.rb(xs.float() + R.rb(R.rb(torch.tensor(0.037)).item() * (-v.float())))
    x2 = xs.clone()
    hip.euler_step(x2, v, dts, step)
    report("euler", x2, ref)
    hip.step_advance(step)
    torch.cuda.synchronize()
    print("step after advance:", step.item())


@guard
def probe_graph():
    st = torch.cuda.Stream()
    a, w, bias = rnd(256, 128, seed=1), rnd(128, 128, scale=0.1, seed=2), rnd(128, seed=3)
    out = torch.zeros(256, 128, dtype=torch.bfloat16, device=dev)
    gates = rnd(3, 128, seed=4)
    res = rnd(256, 128, seed=5)
    step = torch.zeros(1, dtype=torch.int32, device=dev)
    p = hip.make_problem(a, w, bias, out, res=res, gate=gates)
    torch.cuda.synchronize()
    with torch.cuda.stream(st):
        s = st.cuda_stream
        with hip.Graph(s) as g:
            hip.gemm(p, epi=2, step_ptr=step, gate_step_stride=128, stream=s)
            hip.step_advance(step, stream=s)
        for i in range(3):
            g.launch()
            st.synchronize()
            ref = R.gemm_ref(a, w, bias, 2, res, gates[i])
            report(f"graph replay {i}", out, ref)


def main():
    print("device:", torch.cuda.get_device_name(0), "lib:", hip.LIB_PATH)
    hip.require_gpu()
    for cfg in (1, 2, 3, 4, 5, 19, 20, 21, 34, 36):
        for epi in (0, 1, 2, 3):
            probe_gemm(128, 128, 64, epi, cfg)
        probe_gemm(200, 192, 128, 0, cfg)
        probe_gemm(37, 64, 256, 2, cfg, lda_pad=64)
        probe_gemm(513, 264, 384, 1, cfg)
        probe_gemm_grouped(cfg)
    if "--gemm-only" in sys.argv or "--attn-only" in sys.argv:
        return perf()
    probe_ln(10, 256); probe_ln(1000, 3072); probe_ln(7, 4096)
    probe_elementwise()
    for var in (0, 1):
        probe_qknorm_attn(64, 2, variant=var)
        probe_qknorm_attn(40, 2, variant=var)
        probe_qknorm_attn(200, 3, extra_cols=256, variant=var)
        probe_qknorm_attn(333, 2, kv_len=301, variant=var)
        probe_qknorm_attn(1664, 4, variant=var)
    probe_graph()
    perf()


def perf():
    if "--perf" in sys.argv:
        for cfg in (() if "--attn-only" in sys.argv else (0, 20, 5, 19)):
            for pa, pw in ((0, 0),):
                probe_gemm(3968, 9216, 3072, 0, cfg, time_it=True, lda_pad=pa, ldw_pad=pw)
            probe_gemm(3968, 3072, 3072, 2, cfg, time_it=True)
            probe_gemm(3968, 12288, 3072, 1, cfg, time_it=True)
            probe_gemm(3968, 3072, 15360, 2, cfg, time_it=True)
        for var in (() if "--gemm-only" in sys.argv else (0, 1)):
            probe_qknorm_attn(3968, 24, variant=var, time_it=True)
            probe_qknorm_attn(3968, 24, extra_cols=12288, variant=var, time_it=True)
    nfail = sum(1 for r in RESULTS if not r.get("ok"))
    print(f"SUMMARY: {len(RESULTS)} cases, {nfail} failed")
    with open(os.path.join(REPO, "gpurun_out", "probe_results.json"), "w") as f:
        json.dump(RESULTS, f, indent=1)


if __name__ == "__main__":
    os.makedirs(os.path.join(REPO, "gpurun_out"), exist_ok=True)
    main()
