"""-m gpu: parity at BASELINE.json's full sizes (D = 3072, H = 24, T = 512 text tokens), for EVERY geometry of
SURVEY.md §8's table:

    cfg 1   384-grid 1x2           N = 1152   L = 1664
    cfg 2   384-grid 2x3           N = 3456   L = 3968      (cfg 4 = the same per GPU)
    cfg 3   512-grid 2x3           N = 6144   L = 6656
    cfg 5   384-grid 3x4           N = 6912   L = 7424
    sdedit  cfg 5's upsample stage N = 4096   L = 4608      (one 1024x1024 target, unshifted strength-0.4 grid)
    g5x5    384-grid 5x5           N = 14400  L = 14912     (beyond BASELINE: the largest grid of the reference's UI, app.py:10-11)
and for two shapes the pipeline REALLY produces from non-square photographs (visualcloze.py:28-60,312-323: area ~ 384^2, sides
floored to multiples of 16, every row at the aspect of its first image) - off every 64 / 128 / 256 tile edge:
    p34     384-grid 2x3 of 3:4 portraits (320x432 px, 540 tokens each)                N = 3240   L = 3752
    mixed   the same grid with a 4:3 landscape second row (432x320 px: rows 54x120 and 40x162 latent)   L = 3752

Per geometry:
* a full-WIDTH Flux with 1 double + 1 single block against the oracle on the CPU (bf16 mode = same rounding points,
  fp32 mode = exact reference semantics);
* size-independent properties of the attention kernel at that L: softmax rows sum to one (V = 1 => O = 1), joint
  key/value permutation invariance;
* the hipGraph-replayed sampler == eager stepping on that geometry's time grid.
At cfg 2 additionally: the full 19+38-block model against the oracle (ONE evaluation; bf16-merged and fp32-ref),
fused == eager and determinism of the full model, GEMM vs torch + gate linearity, and the race screen.

Stated tolerances.  `floor` = rel-L2 between the oracle's own bf16 and fp32 runs on the same inputs = what bf16
execution costs ANY implementation, the reference included.
    HIP vs bf16 oracle (same rounding points, merged LoRA)   <= 1.5e-2      (1+1 blocks)    <= 1.5 * floor (full depth)
    HIP vs fp32 oracle                                        <= max(3e-2, 4 * floor) (1+1)  <= 2 * floor   (full depth)
"""
import os
import time

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
T, D, H = 512, 3072, 24
GEOMS = {                         # rows of the grid, latent (h, w) of one concatenated row
    "cfg1": (1, (48, 96)),
    "cfg2": (2, (48, 144)),
    "cfg3": (2, (64, 192)),
    "cfg5": (3, (48, 192)),
    "sdedit": (1, (128, 128)),
    "g5x5": (5, (48, 240)),                  # the largest grid the reference's UI offers (app.py:10-11): N = 14400, L = 14912
    "p34": [(54, 120), (54, 120)],           # per-row latent sizes
    "mixed": [(54, 120), (40, 162)],
}


def _rows(geom):
    g = GEOMS[geom]
    return list(g) if isinstance(g, list) else [g[1]] * g[0]
N2 = 3456
L2 = T + N2


def rel_l2(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return ((a - b).norm() / b.norm()).item()


def _inputs(geom="cfg2", seed=0):
    from bench import grid_img_ids
    g = torch.Generator().manual_seed(seed)
    r = lambda *s: torch.randn(*s, generator=g)  # noqa: E731
    ids = grid_img_ids(_rows(geom))
    N = ids.shape[0]
    return dict(x=r(1, N, 64), cond=r(1, N, 320), img_ids=ids[None], txt=r(1, T, 4096), txt_ids=torch.zeros(1, T, 3),
                y=r(1, 768), txt_mask=torch.ones(1, T, dtype=torch.int32), img_mask=torch.ones(1, N, dtype=torch.int32),
                guidance=torch.full((1,), 30.0))


def _build(depth, single, seed=1):
    from visualcloze_amd.model import FLUX_DEV_FILL, FluxLoraWrapper, FluxParams
    old = torch.get_default_dtype()
    torch.set_default_dtype(torch.bfloat16)
    try:
        with torch.device(DEV):
            m = FluxLoraWrapper(lora_rank=256, lora_scale=1.0,
                                params=FluxParams(**{**FLUX_DEV_FILL, "depth": depth, "depth_single_blocks": single}))
    finally:
        torch.set_default_dtype(old)
    g = torch.Generator(device=DEV).manual_seed(seed)
    with torch.no_grad():
        for name, p in m.named_parameters():
            if name.endswith("norm.scale"):
                p.fill_(1.0)
            else:
                p.normal_(0.0, 0.02, generator=g)
    return m.eval()


@pytest.fixture(scope="module")
def small_model():
    """full width, 1 DoubleStreamBlock + 1 SingleStreamBlock; shared by the per-geometry tests"""
    m = _build(1, 1)
    sd = {k: v.detach().float().cpu() for k, v in m.state_dict().items()}
    return m, sd


def _call(m, inp, t):
    img = torch.cat((inp["x"], inp["cond"]), -1)
    out = m(img.to(DEV, torch.bfloat16), img_ids=inp["img_ids"].to(DEV), txt=inp["txt"].to(DEV, torch.bfloat16),
            txt_ids=inp["txt_ids"].to(DEV), timesteps=t.to(DEV), y=inp["y"].to(DEV, torch.bfloat16),
            txt_mask=inp["txt_mask"].to(DEV), img_mask=inp["img_mask"].to(DEV), guidance=inp["guidance"].to(DEV))
    torch.cuda.synchronize()
    return out


def _oracle_pair(sd, G, inp, t):
    """(bf16-merged, fp32-ref) oracle outputs for f32 guidance"""
    import oracle.flux_oracle as O
    orig = O.compute_vec
    O.compute_vec = lambda *a, **k: orig(*a, **{**k, "guidance_is_bf16": False})
    try:
        args = (sd, G, torch.cat((inp["x"], inp["cond"]), -1).bfloat16().float(), inp["img_ids"],
                inp["txt"].bfloat16().float(), inp["txt_ids"], t, inp["y"].bfloat16().float(), inp["txt_mask"],
                inp["img_mask"], inp["guidance"])
        with torch.no_grad():
            return O.flux_forward(*args, P=O.Prec("bf16", "merged")), O.flux_forward(*args, P=O.Prec("fp32", "ref"))
    finally:
        O.compute_vec = orig


@pytest.mark.parametrize("geom", list(GEOMS))
def test_full_width_one_plus_one_blocks_vs_oracle(small_model, geom):
    import oracle.flux_oracle as O
    m, sd = small_model
    inp = _inputs(geom)
    t = torch.tensor([0.62])
    got = _call(m, inp, t)
    if geom == "g5x5":      # L = 14912: the oracle's L x L attention dominates (2 x 70 s on the box's host cores) - the fp32 /
        # reference-semantics leg only (with both: bf16 5.4e-3, fp32 9.07e-3, floor 9.07e-3, profiles/r05e_parity.log)
        with torch.no_grad():
            want_fp32 = O.flux_forward(sd, O.FluxGeometry(depth=1, depth_single_blocks=1), torch.cat((inp["x"], inp["cond"]), -1).bfloat16().float(),
                                       inp["img_ids"], inp["txt"].bfloat16().float(), inp["txt_ids"], t, inp["y"].bfloat16().float(),
                                       inp["txt_mask"], inp["img_mask"], inp["guidance"], P=O.Prec("fp32", "ref"))
        e32 = rel_l2(got, want_fp32)
        from tests.helpers import parity_log
        parity_log(f"[1+1 blocks, {geom}] L={T + inp['x'].shape[1]}: HIP vs fp32 oracle {e32:.3e}")
        assert e32 < 3e-2
        return
    want_bf16, want_fp32 = _oracle_pair(sd, O.FluxGeometry(depth=1, depth_single_blocks=1), inp, t)
    floor = rel_l2(want_bf16, want_fp32)          # what bf16 execution costs the oracle itself at this size
    e16, e32 = rel_l2(got, want_bf16), rel_l2(got, want_fp32)
    from tests.helpers import parity_log
    parity_log(f"[1+1 blocks, {geom}] L={T + inp['x'].shape[1]}: HIP vs bf16 oracle {e16:.3e}, vs fp32 oracle {e32:.3e}, oracle bf16-vs-fp32 floor {floor:.3e}")
    assert e16 < 1.5e-2
    assert e32 < max(3e-2, 4 * floor)


@pytest.mark.parametrize("qmax,kmax,single_max,bounded", [(1.5, 1.5, None, True), (3.0, 3.0, None, False), (1.5, 1.5, 3.0, "mixed")])
def test_full_width_non_unit_norm_scales_both_sides_of_the_logit_bound(qmax, kmax, single_max, bounded):
    """Which attention instantiation runs is a property of the WEIGHTS, block by block: the bounded-logit kernels (no running max;
    attn64s_kernel<true>, the stream form, with the queries finished by the qkv GEMM's epilogue) while
    16.65 * max|query_norm.scale| * max|key_norm.scale| of the BLOCK is <= 100 (model.prepare / flux_engine.hip resolve(); every
    other full-size test and the bench use unit scales and therefore always take it), the running-max template beyond - a real
    checkpoint may sit on either side, or on both: "mixed" gives the double block scales up to 1.5 and the single block scales up
    to 3, so ONE evaluation runs both templates (the C handle and the Python-ordered plan must pick the same ones: bit-equal).
    One full-width evaluation (cfg 2 geometry, 1 + 1 blocks) with NON-UNIT scales against the oracle (layers.py:63-84; bounds as
    test_full_width_one_plus_one_blocks_vs_oracle)."""
    import oracle.flux_oracle as O
    from tests.helpers import parity_log
    m = _build(1, 1, seed=7)
    g = torch.Generator(device=DEV).manual_seed(11)
    with torch.no_grad():
        for name, p in m.named_parameters():
            if name.endswith("norm.scale"):
                hi = qmax if "query_norm" in name else kmax
                if single_max is not None and name.startswith("single_blocks."):
                    hi = single_max
                p.copy_(0.5 + (hi - 0.5) * torch.rand(p.shape, device=DEV, generator=g).to(p.dtype))
                p[0] = hi
    sd = {k: v.detach().float().cpu() for k, v in m.state_dict().items()}
    inp = _inputs("cfg2", seed=9)
    t = torch.tensor([0.62])
    got = _call(m, inp, t)
    eng = m.engine()
    bd, bs = eng.W.logit_bounds["double_blocks.0"], eng.W.logit_bounds["single_blocks.0"]
    assert eng.W.logit_bound == max(bd, bs)
    if bounded == "mixed":
        assert 0.0 < bd <= 100.0 < bs, (bd, bs)
        m.use_handle = False                       # the Python-ordered plan picks its templates from model.prepare's per-block bounds,
        try:                                       # the C handle from its own read-back of the bound scales: the same bits
            twin = _call(m, inp, t)
        finally:
            m.use_handle = True
        assert torch.equal(twin, got)
    else:
        assert (0.0 < eng.W.logit_bound <= 100.0) == bounded, eng.W.logit_bound
    assert eng.attention_variant(eng.workspace(T, inp["x"].shape[1], 1, 1)) == 28      # 12 + 16: tail pieces combined in the launch where the stream form runs
    want_bf16, want_fp32 = _oracle_pair(sd, O.FluxGeometry(depth=1, depth_single_blocks=1), inp, t)
    floor, e16, e32 = rel_l2(want_bf16, want_fp32), rel_l2(got, want_bf16), rel_l2(got, want_fp32)
    parity_log(f"[1+1 blocks, cfg2, norm scales up to {qmax} / {kmax}" + (f" (single block {single_max})" if single_max else "") +
               f": logit bounds {bd:.1f} / {bs:.1f} -> attn64s_kernel<{str(bd <= 100).lower()}> / <{str(bs <= 100).lower()}>] "
               f"HIP vs bf16 oracle {e16:.3e}, vs fp32 oracle {e32:.3e}, oracle bf16-vs-fp32 floor {floor:.3e}")
    assert e16 < 1.5e-2 and e32 < max(3e-2, 4 * floor)
    del m
    torch.cuda.empty_cache()


def test_full_width_batch_of_two_equals_per_sample(small_model):
    """A per-GPU batch at full width (cfg 2 geometry, B = 2, different text / guidance per sample): one stacked launch
    sequence - batch-strided qkv rows, the key norm of both samples in the GEMM's epilogue with per-sample RoPE rows, V^T per
    sample, the batch dimension of the attention grid - equals the two B = 1 runs (same kernels; tile shapes per row block may
    differ, hence bf16 noise instead of bit equality)."""
    m, _ = small_model
    a, b = _inputs("cfg2", seed=21), _inputs("cfg2", seed=22)
    both = {k: torch.cat((a[k], b[k])) for k in a}
    both["guidance"] = torch.tensor([30.0, 3.5])
    b["guidance"] = torch.tensor([3.5])
    t = torch.tensor([0.8, 0.3])
    got = _call(m, both, t).float().cpu()
    for i, one in enumerate((a, b)):
        ref = _call(m, one, t[i:i + 1]).float().cpu()
        assert rel_l2(got[i:i + 1], ref) < 5e-3, i


@pytest.mark.parametrize("geom", list(GEOMS))
def test_attention_properties(geom):
    from visualcloze_amd import hip
    L = T + sum((h // 2) * (w // 2) for h, w in _rows(geom))
    Lp = (L + 63) // 64 * 64
    g = torch.Generator().manual_seed(5)
    qkv = (torch.randn(L, 3 * D, generator=g)).to(torch.bfloat16).to(DEV)
    out = torch.empty(L, D, dtype=torch.bfloat16, device=DEV)
    # V = 1  =>  every output element is a convex combination of ones
    vt = torch.zeros(H, 128, Lp, dtype=torch.bfloat16, device=DEV)
    vt[..., :L] = 1
    for variant in (0, 1, 2, 3, 7, 8, 12):
        out.fill_(float("nan"))
        hip.attention(qkv, vt, out, L, H, variant=variant)
        torch.cuda.synchronize()
        assert (out.float() - 1.0).abs().max().item() < 1e-2
    # permuting keys and values together leaves the result unchanged (up to summation order)
    vt[..., :L] = qkv[:, 2 * D:].reshape(L, H, 128).permute(1, 2, 0)
    hip.attention(qkv, vt, out, L, H, variant=12)
    perm = torch.randperm(L, generator=g).to(DEV)
    qkv2 = qkv.clone()
    qkv2[:, D:] = qkv[perm][:, D:]                 # permute k and v rows, keep q
    vt2 = torch.zeros_like(vt)
    vt2[..., :L] = qkv2[:, 2 * D:].reshape(L, H, 128).permute(1, 2, 0)
    out2 = torch.empty_like(out)
    hip.attention(qkv2, vt2, out2, L, H, variant=12)
    torch.cuda.synchronize()
    assert rel_l2(out2, out) < 1e-2
    hip.attention(qkv, vt, out, L, H, variant=7)
    # variants 0-3 agree bit for bit; 7 (tail items cut along the keys and merged) differs only by f32 summation order
    o1 = torch.empty_like(out)
    hip.attention(qkv, vt, o1, L, H, variant=1)
    for variant in (0, 2, 3):
        o3 = torch.empty_like(out)
        hip.attention(qkv, vt, o3, L, H, variant=variant)
        torch.cuda.synchronize()
        assert torch.equal(o3, o1)
    assert rel_l2(out, o1) < 4e-3            # (rows that differ do so by one bf16 ulp = 2^-8 relative)
    assert (out.float() - o1.float()).abs().max().item() <= 2 ** -7 * o1.float().abs().max().item()
    # the one-wave-per-SIMD kernel (8; 12 with the tail split): same arithmetic, its own summation order
    for variant in (8, 12):
        o8 = torch.full_like(out, float("nan"))
        hip.attention(qkv, vt, o8, L, H, variant=variant)
        torch.cuda.synchronize()
        assert rel_l2(o8, o1) < 6e-3        # (q * 128^-0.5 log2 e is rounded to bf16 once more in this kernel)
        assert (o8.float() - o1.float()).abs().max().item() <= 2 ** -6 * o1.float().abs().max().item()
    # every variant against the function itself (f32 torch softmax(q k^T / sqrt d) v on two heads)
    for hd in (0, H - 1):
        q = qkv[:, hd * 128:(hd + 1) * 128].float()
        k = qkv[:, D + hd * 128:D + (hd + 1) * 128].float()
        v = qkv[:, 2 * D + hd * 128:2 * D + (hd + 1) * 128].float()
        ref = torch.softmax(q @ k.t() * 128 ** -0.5, dim=-1) @ v
        for variant in (1, 7, 8, 12):
            o = torch.empty_like(out)
            hip.attention(qkv, vt, o, L, H, variant=variant)
            torch.cuda.synchronize()
            got = o[:, hd * 128:(hd + 1) * 128]
            assert rel_l2(got, ref) < 1e-2, (variant, hd)
            assert (got.float() - ref).abs().max().item() <= 2e-2 * ref.abs().max().item(), (variant, hd)


def _kw(inp):
    return dict(txt=inp["txt"].to(DEV, torch.bfloat16), txt_ids=inp["txt_ids"].to(DEV), txt_mask=inp["txt_mask"].to(DEV),
                y=inp["y"].to(DEV, torch.bfloat16), img_ids=inp["img_ids"].to(DEV), img_mask=inp["img_mask"].to(DEV),
                cond=inp["cond"].to(DEV, torch.bfloat16), guidance=inp["guidance"].to(DEV, torch.bfloat16))


@pytest.mark.parametrize("geom", list(GEOMS))
def test_fused_sampler_equals_eager_per_geometry(small_model, geom):
    """hipGraph replays with the device step counter == host-driven stepping of the same kernels, on the geometry's own
    time grid (shifted with mu(N); the SDEdit stage: strength 0.4, no shift - visualcloze.py:184-193)."""
    from visualcloze_amd.transport import Sampler, create_transport
    m, _ = small_model
    inp = _inputs(geom, seed=3)
    kw, x = _kw(inp), inp["x"].to(DEV, torch.bfloat16)
    opts = dict(sampling_method="euler", num_steps=4, do_shift=True, time_shifting_factor=1)
    if geom == "sdedit":
        opts.update(do_shift=False, time_shifting_factor=1.0, strength=0.4)
    fn = Sampler(create_transport()).sample_ode(**opts)
    fused = fn(x, m.forward, kw)
    fused2 = fn(x, m.forward, kw)
    eager = fn(x, lambda xx, **k: m.forward(xx, **k), kw)
    m.use_handle = False                          # the same plan ordered from Python over the op-level ABI
    try:
        fused_py = fn(x, m.forward, kw)
    finally:
        m.use_handle = True
    torch.cuda.synchronize()
    assert fused.shape == (1, 1, x.shape[1], 64) and torch.isfinite(fused.float()).all()
    assert torch.equal(fused, fused2)
    assert torch.equal(fused, fused_py)           # vc_flux_sample_euler == graph replays driven from engine.py, bit for bit
    assert rel_l2(fused, eager) < 5e-3            # same kernels; eager does the Euler update with torch


def test_gemm_full_size_vs_torch_and_gate_linearity():
    from visualcloze_amd import hip
    from tests import ref_ops as R
    g = torch.Generator().manual_seed(6)
    a = torch.randn(L2, D, generator=g).to(torch.bfloat16).to(DEV)
    w = (torch.randn(3 * D, D, generator=g) * D ** -0.5).to(torch.bfloat16).to(DEV)
    b = torch.randn(3 * D, generator=g).to(torch.bfloat16).to(DEV)
    out = hip.linear(a, w, b)
    torch.cuda.synchronize()
    ref = R.gemm_ref(a, w, b, 0)
    assert (out.float() - ref).abs().max().item() <= 2e-2 * ref.abs().max().item()
    # gate = 0  =>  the gated-residual epilogue returns the residual bit-exactly, in place
    res = torch.randn(L2, D, generator=g).to(torch.bfloat16).to(DEV)
    x = res.clone()
    w2 = w[:D].contiguous()
    hip.gemm(hip.make_problem(a, w2, b[:D], x, res=x, gate=torch.zeros(D, dtype=torch.bfloat16, device=DEV)),
             epi=hip.EPI_GATE_RES)
    torch.cuda.synchronize()
    assert torch.equal(x, res)


class _LazyF32(dict):
    """state dict that keeps the 13 B parameters as bf16 on the host and hands the oracle one f32 tensor at a time
    (the weights are bf16 values, so this is exact; a full f32 copy would be 52 GB)."""

    def __getitem__(self, k):
        return dict.__getitem__(self, k).float()

    def get(self, k, default=None):
        return self[k] if k in self else default


FULLDEPTH_FIXTURE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "fulldepth_cfg2_oracle.npz")
PROBE_KEYS = ("img_in.weight", "double_blocks.18.txt_mlp.2.lora_B.weight", "single_blocks.37.linear1.weight",
              "final_layer.adaLN_modulation.1.bias")


def _weights_probe(sd):
    """integer checksums of four parameters (bf16 bit patterns summed as int64): identifies the weight draw exactly"""
    return [int(dict.__getitem__(sd, k).contiguous().view(torch.int16).to(torch.int64).sum()) for k in PROBE_KEYS]


def test_full_depth_19_38_vs_oracle():
    """ONE evaluation of the full 19 + 38-block model (13.1 B parameters, L = 3968, LoRA r256) against the CPU oracle,
    both modes.  Error growth through 57 blocks is stated against the oracle's own bf16-vs-fp32 deviation.

    The oracle needs ~9 min on the box's 128 cores, so its two outputs for exactly these weights and inputs are a
    committed fixture (tests/golden/fulldepth_cfg2_oracle.npz, written by this very test with VC_SAVE_FULLDEPTH=<path>);
    the fixture is only used when integer checksums of the weights match the draw it was made from - otherwise, or with
    VC_LIVE_ORACLE=1, the oracle runs live."""
    import numpy as np
    import oracle.flux_oracle as O
    m = _build(19, 38)
    inp = _inputs("cfg2", seed=11)
    t = torch.tensor([0.62])
    got = _call(m, inp, t).float().cpu()
    sd = _LazyF32({k: v.detach().cpu() for k, v in m.state_dict().items()})
    del m
    torch.cuda.empty_cache()
    probe = _weights_probe(sd)
    fx = np.load(FULLDEPTH_FIXTURE) if os.path.exists(FULLDEPTH_FIXTURE) else None
    live = fx is None or os.environ.get("VC_LIVE_ORACLE") == "1" or [int(v) for v in fx["probe"]] != probe \
        or float(fx["x_sum"]) != inp["x"].double().sum().item()
    t0 = time.time()
    if live:
        want_bf16, want_fp32 = _oracle_pair(sd, O.FluxGeometry(), inp, t)
        how = f"oracle run live: {time.time() - t0:.0f} s on {torch.get_num_threads()} threads"
        if os.environ.get("VC_SAVE_FULLDEPTH"):
            assert torch.equal(want_bf16.to(torch.bfloat16).float(), want_bf16)     # bf16-mode outputs are bf16 values
            np.savez_compressed(os.environ["VC_SAVE_FULLDEPTH"], want_bf16_bits=want_bf16.to(torch.bfloat16).view(torch.int16).numpy(),
                                want_fp32=want_fp32.numpy(), probe=np.asarray(probe, np.int64),
                                x_sum=np.float64(inp["x"].double().sum().item()), t=t.numpy(),
                                oracle_seconds=time.time() - t0, threads=torch.get_num_threads())
    else:
        want_bf16 = torch.tensor(fx["want_bf16_bits"]).view(torch.bfloat16).float()
        want_fp32 = torch.tensor(fx["want_fp32"])
        how = f"oracle outputs from the committed fixture ({float(fx['oracle_seconds']):.0f} s on {int(fx['threads'])} threads when made)"
    floor = rel_l2(want_bf16, want_fp32)
    e16, e32 = rel_l2(got, want_bf16), rel_l2(got, want_fp32)
    from tests.helpers import parity_log
    parity_log(f"[full depth 19+38, cfg2] HIP vs bf16-merged oracle {e16:.3e}, vs fp32-ref oracle {e32:.3e}, "
               f"oracle bf16-vs-fp32 floor {floor:.3e}  ({how})")
    assert torch.isfinite(got).all()
    assert e16 < 1.5 * floor
    assert e32 < 2.0 * floor


TRAJ_FIXTURE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "fullwidth_traj.npz")


def _traj_module():
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_fullwidth_traj", os.path.join(os.path.dirname(TRAJ_FIXTURE), "make_fullwidth_traj.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _build_procedural(depth, single):
    """full width, LoRA r256, PROCEDURAL weights (tests/procedural.py, evaluated on the GPU: bit-identical to the numpy /
    torch-CPU values the committed oracle fixtures were made from)"""
    from tests.procedural import procedural_param
    from visualcloze_amd.model import FLUX_DEV_FILL, FluxLoraWrapper, FluxParams
    old = torch.get_default_dtype()
    torch.set_default_dtype(torch.bfloat16)
    try:
        with torch.device(DEV):
            m = FluxLoraWrapper(lora_rank=256, lora_scale=1.0, params=FluxParams(**{**FLUX_DEV_FILL, "depth": depth, "depth_single_blocks": single}))
    finally:
        torch.set_default_dtype(old)
    with torch.no_grad():
        for k, v in m.state_dict().items():
            v.copy_(procedural_param(k, v.shape, device=DEV, dtype=torch.float32))
    return m.eval()


@pytest.fixture(scope="module")
def procedural_small_model():
    return _build_procedural(1, 1)


@pytest.mark.parametrize("case", ["cfg2", "sdedit", "cfg5", "cfg5_50", "p34"])
def test_full_width_trajectory_vs_oracle(procedural_small_model, case):
    """The WHOLE loop at full width against the oracle's own trajectories (transport/integrators.py:106-120,
    transport/transport.py:384): cfg 2's 30-point shifted grid = 29 evaluations at L = 3968, and the SDEdit stage's 10
    points from strength 0.4 = 9 evaluations at L = 4608, and cfg 5's 29 evaluations at L = 7424 (the largest BASELINE geometry;
    its own fixture file) - and cfg 5 as BASELINE.json quotes it, 50 solver points = 49 evaluations (`cfg5_50`), and a NON-SQUARE grid
    the pipeline really produces (`p34`: 2x3 of 3:4 portraits, L = 3752, 29 evaluations) -, D = 3072,
    1 + 1 blocks.  The fused sampler's intermediate and
    FINAL latents are held to the bf16-merged oracle (same rounding points) and the fp32-ref oracle (exact reference
    semantics), with bounds stated against `floor` = the oracle's own bf16-vs-fp32 deviation on the same state:
        HIP vs bf16 oracle <= floor,   HIP vs fp32 oracle <= 1.5 * floor      (final state and every saved one;
    measured: 0.3 * floor and 1.0 * floor - the fused loop drifts from exact arithmetic exactly as far as the reference's own
    bf16 rounding does, and 3x less far from the oracle that rounds where it rounds)."""
    import numpy as np
    from tests.helpers import parity_log
    from visualcloze_amd.transport import Sampler, create_transport
    FT = _traj_module()
    c, inp = FT.CASES[case], FT.inputs(case)
    path = os.path.join(os.path.dirname(TRAJ_FIXTURE), c.get("file", os.path.basename(TRAJ_FIXTURE)))
    if not os.path.exists(path):
        pytest.skip(f"{os.path.basename(path)} not generated (tests/golden/make_fullwidth_traj.py --only {case})")
    fx = np.load(path)
    assert float(fx[f"{case}_x_sum"]) == inp["x"].double().sum().item()
    opts = dict(sampling_method="euler", num_steps=c["points"], do_shift=c["do_shift"], return_trajectory=True,
                time_shifting_factor=1 if c["do_shift"] else 1.0, strength=c["strength"])
    fn = Sampler(create_transport()).sample_ode(**opts)
    m = procedural_small_model
    kw = dict(_kw(inp), guidance=inp["guidance"].to(DEV))     # f32 guidance: the value both oracle modes were run with
    tr = fn(inp["x"].to(DEV, torch.bfloat16), m.forward, kw)
    torch.cuda.synchronize()
    assert tr.shape[0] == c["points"] and torch.isfinite(tr.float()).all()
    last = int(fx[f"{case}_keep"][-1])
    assert last == c["points"] - 1
    stride = int(fx["token_stride"])                  # intermediate states are stored for every stride-th token, the last whole
    for k in [int(v) for v in fx[f"{case}_keep"]]:
        b16 = torch.tensor(fx[f"{case}_bf16_{k}"]).view(torch.bfloat16).float()
        f32 = torch.tensor(fx[f"{case}_fp32_{k}"].astype("float32"))
        got = tr[k][:, ::int(fx[f"{case}_final_stride"]) if f"{case}_final_stride" in fx else 1] if k == last else tr[k][:, ::stride]
        floor, e16, e32 = rel_l2(b16, f32), rel_l2(got, b16), rel_l2(got, f32)
        parity_log(f"[trajectory 1+1 blocks, {case}] state {k}/{last}: HIP vs bf16 oracle {e16:.3e}, vs fp32 oracle {e32:.3e}, "
                   f"oracle bf16-vs-fp32 floor {floor:.3e}")
        assert e16 < 1.0 * floor and e32 < 1.5 * floor, (k, e16, e32, floor)


def test_full_width_blocks_vs_the_reference_itself(procedural_small_model):
    """The HIP blocks against the REFERENCE'S OWN outputs at FLUX width (tests/golden/fullwidth_reference.npz: the reference's
    FluxLoraWrapper, 1 DoubleStreamBlock + 1 SingleStreamBlock at hidden 3072 / 24 heads / L = 3968, fp32 on the CPU - not the
    oracle): `Flux.forward` whole, and the block outputs where the fixture samples them (engine taps).  Bounds as for the fp32
    oracle: <= max(3e-2, 4 x the bf16 floor); measured ~9e-3 = the floor itself."""
    import numpy as np
    from tests.helpers import parity_log
    FT = _traj_module()
    fx = np.load(os.path.join(os.path.dirname(TRAJ_FIXTURE), "fullwidth_reference.npz"))
    inp = FT.inputs("cfg2")
    assert float(fx["x_sum"]) == inp["x"].double().sum().item()
    m = procedural_small_model
    t = torch.tensor(fx["t"])
    got = _call(m, inp, t)
    e = rel_l2(got, torch.tensor(fx["flux"]))
    # the block outputs: the same evaluation through the Python-ordered plan with taps (bit-identical to the handle's)
    eng = m.engine()
    N = inp["x"].shape[1]
    ws = eng.workspace(T, N, 1, 1)
    kw = _kw(inp)
    eng.prepare_sample(ws, kw["txt"], kw["y"], inp["guidance"].to(DEV), False, kw["img_ids"], kw["txt_ids"],
                       t.to(DEV).float().reshape(1, 1), [ws.L])                      # (as model.forward's Python-ordered route)
    ws.XIN.copy_(torch.cat((inp["x"], inp["cond"]), -1).to(DEV, torch.bfloat16).reshape(N, -1))
    taps = {}
    eng.eval_once(ws, None, euler=False, taps=taps, concat=False)
    torch.cuda.synchronize()
    assert torch.equal(ws.V.reshape(got.shape), got)
    rs, cs = int(fx["row_stride"]), int(fx["col_stride"])
    errs = {"double_img": rel_l2(taps["double.0.img"][::rs, ::cs], torch.tensor(fx["double_img"])),
            "double_txt": rel_l2(taps["double.0.txt"][::rs, ::cs], torch.tensor(fx["double_txt"])),
            "single": rel_l2(taps["single.0"][::rs, ::cs], torch.tensor(fx["single"]))}
    parity_log(f"[1+1 blocks at full width, cfg2] HIP vs the REFERENCE's own fp32 run: Flux.forward {e:.3e}, DoubleStreamBlock img "
               f"{errs['double_img']:.3e} / txt {errs['double_txt']:.3e}, SingleStreamBlock {errs['single']:.3e}")
    assert e < 3e-2 and all(v < 3e-2 for v in errs.values()), (e, errs)


def test_full_width_hip_vs_the_reference_bf16_run(procedural_small_model):
    """The HIP path against the reference's OWN bf16 run at FLUX width (`flux_bf16` of fullwidth_reference.npz: bf16 parameters,
    bf16 inputs and guidance under torch.autocast("cpu", bf16) - SURVEY.md §8c's mode, written by make_fullwidth_reference.py):
    the un-merged LoRA mode (`lora_mode="ref"`: LinearLora.forward executed as lora.py:92-98 writes it) and the product's merged
    mode, both with bf16 guidance (1000 g -> 29952).  Bounds STATED: <= 1.5e-2 rel-L2 for `Flux.forward` in either mode (two bf16
    implementations of one function; the oracle's bf16 mode measures 4.9e-3 against the same run, merged LoRA adds one rounding
    per weight instead of three per activation)."""
    import numpy as np
    from tests.helpers import parity_log
    FT = _traj_module()
    fx = np.load(os.path.join(os.path.dirname(TRAJ_FIXTURE), "fullwidth_reference.npz"))
    if "flux_bf16" not in fx.files:
        pytest.skip("fullwidth_reference.npz predates round 6 (no bf16 run)")
    inp = FT.inputs("cfg2")
    ref = torch.tensor(fx["flux_bf16"].view(np.int16)).view(torch.bfloat16).float()
    m = procedural_small_model
    t = torch.tensor(fx["t"])

    def call():
        img = torch.cat((inp["x"], inp["cond"]), -1)
        out = m(img.to(DEV, torch.bfloat16), img_ids=inp["img_ids"].to(DEV), txt=inp["txt"].to(DEV, torch.bfloat16),
                txt_ids=inp["txt_ids"].to(DEV), timesteps=t.to(DEV), y=inp["y"].to(DEV, torch.bfloat16),
                txt_mask=inp["txt_mask"].to(DEV), img_mask=inp["img_mask"].to(DEV), guidance=inp["guidance"].to(DEV, torch.bfloat16))
        torch.cuda.synchronize()
        return out.float().cpu()
    e_merged = rel_l2(call(), ref.reshape(1, -1, 64))
    assert m.lora_mode == "merged"
    m.lora_mode = "ref"
    try:
        e_ref = rel_l2(call(), ref.reshape(1, -1, 64))
    finally:
        m.lora_mode = "merged"
    parity_log(f"[1+1 blocks at full width, cfg2] HIP vs the REFERENCE's own bf16 autocast run: lora_mode=ref {e_ref:.3e}, merged {e_merged:.3e}")
    assert e_ref < 1.5e-2 and e_merged < 1.5e-2, (e_ref, e_merged)


TIMES_FIXTURE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "fulldepth_times_oracle.npz")


@pytest.fixture(scope="module")
def procedural_full_model():
    return _build_procedural(19, 38)


def test_full_depth_at_both_ends_of_the_time_grid_vs_oracle(procedural_full_model):
    """The full 19 + 38-block model at the FIRST and the LAST Flux time of cfg 2's 30-point grid (t = 1.0 and
    t = 1 - bf16(t_28) ~ 0.09: the timestep embeddings furthest from the t = 0.62 of test_full_depth_19_38_vs_oracle),
    procedural weights, against the committed oracle outputs of tests/golden/make_fulldepth_times.py; bounds as there:
    HIP vs bf16-merged oracle <= 1.5 * floor, vs fp32-ref oracle <= 2 * floor."""
    import numpy as np
    from tests.helpers import parity_log
    FT = _traj_module()
    fx = np.load(TIMES_FIXTURE)
    inp = FT.inputs("cfg2")
    assert float(fx["x_sum"]) == inp["x"].double().sum().item()
    m = procedural_full_model
    for i, t in enumerate(fx["times"]):
        got = _call(m, inp, torch.tensor([float(t)], dtype=torch.float32)).float().cpu()
        b16 = torch.tensor(fx[f"bf16_{i}"]).view(torch.bfloat16).float()
        f32 = torch.tensor(fx[f"fp32_{i}"].astype("float32"))
        floor, e16, e32 = rel_l2(b16, f32), rel_l2(got, b16), rel_l2(got, f32)
        parity_log(f"[full depth 19+38, cfg2, procedural weights] t = {float(t):.4f}: HIP vs bf16-merged oracle {e16:.3e}, vs fp32-ref oracle "
                   f"{e32:.3e}, oracle bf16-vs-fp32 floor {floor:.3e}")
        assert torch.isfinite(got).all()
        assert e16 < 1.5 * floor and e32 < 2.0 * floor, (float(t), e16, e32, floor)


def test_full_depth_trajectory_vs_oracle(procedural_full_model):
    """The one combination the other fixtures leave open: a TRAJECTORY through the FULL-DEPTH model.  The first three solver
    steps of cfg 2's 30-point grid (19 + 38 blocks, 13.1 B procedural parameters, L = 3968), the state fed back after every
    evaluation (transport/integrators.py:99-120), against the oracle's own two trajectories
    (`tests/golden/make_fulldepth_times.py --traj 3` -> fulldepth_traj_oracle.npz; each mode steps ITS state).  The fused
    sampler runs the whole 29-evaluation loop; states 1..3 are compared.  Bounds as for single full-depth evaluations:
    HIP vs bf16-merged oracle <= 1.5 * floor, vs fp32-ref oracle <= 2 * floor, floor = the oracle's own bf16-vs-fp32
    deviation on that state."""
    import numpy as np
    from tests.helpers import parity_log
    from visualcloze_amd.transport import Sampler, create_transport
    path = os.path.join(os.path.dirname(TIMES_FIXTURE), "fulldepth_traj_oracle.npz")
    if not os.path.exists(path):
        pytest.skip("fulldepth_traj_oracle.npz not generated (tests/golden/make_fulldepth_times.py --traj 3, ~70 min of CPU)")
    FT = _traj_module()
    fx = np.load(path)
    inp = FT.inputs("cfg2")
    assert float(fx["x_sum"]) == inp["x"].double().sum().item()
    m = procedural_full_model
    fn = Sampler(create_transport()).sample_ode(sampling_method="euler", num_steps=30, do_shift=True, time_shifting_factor=1,
                                                 return_trajectory=True)
    kw = dict(_kw(inp), guidance=inp["guidance"].to(DEV))     # f32 guidance: the value both oracle modes were run with
    tr = fn(inp["x"].to(DEV, torch.bfloat16), m.forward, kw)
    torch.cuda.synchronize()
    assert tr.shape[0] == 30 and torch.isfinite(tr.float()).all()
    K = len(fx["t"]) - 1
    from visualcloze_amd.transport import model_times, solver_time_grid
    t = solver_time_grid(30, inp["x"].shape[1], 0, 1, True, 1)
    assert np.array_equal(t[:K + 1].numpy(), fx["t"])                                            # same grid points ...
    assert np.allclose(model_times(t, tr[0])[:K].double().numpy(), fx["bf16_model_t"], atol=0, rtol=0)   # ... same Flux times
    for k in range(1, K + 1):
        b16 = torch.tensor(fx[f"bf16_{k}"]).view(torch.bfloat16).float()
        f32 = torch.tensor(fx[f"fp32_{k}"].astype("float32"))
        floor, e16, e32 = rel_l2(b16, f32), rel_l2(tr[k], b16), rel_l2(tr[k], f32)
        parity_log(f"[trajectory through the FULL-DEPTH model 19+38, cfg2] state {k}/{K}: HIP vs bf16-merged oracle {e16:.3e}, vs fp32-ref "
                   f"oracle {e32:.3e}, oracle bf16-vs-fp32 floor {floor:.3e}")
        assert e16 < 1.5 * floor and e32 < 2.0 * floor, (k, e16, e32, floor)


@pytest.mark.parametrize("geom", ["cfg3", "cfg5", "p34"])
def test_full_depth_on_the_large_geometries_vs_oracle(procedural_full_model, geom):
    """The full 19 + 38-block model on the two LARGEST BASELINE geometries (cfg 3: 512-grid 2x3, L = 6656; cfg 5: 384-grid
    3x4, L = 7424 - other attention tails, other tile counts, other RoPE grids than cfg 2) and on a NON-SQUARE shape the pipeline
    really produces (p34: 2x3 grid of 3:4 portraits, L = 3752, off every tile edge), one evaluation at t = 0.62,
    procedural weights, against the committed oracle outputs of `tests/golden/make_fulldepth_times.py --geom <geom>`
    (every second image token); bounds as at cfg 2: <= 1.5 * floor vs the bf16-merged oracle, <= 2 * floor vs fp32-ref."""
    import importlib.util
    import numpy as np
    from tests.helpers import parity_log
    path = os.path.join(os.path.dirname(TIMES_FIXTURE), f"fulldepth_{geom}_oracle.npz")
    spec = importlib.util.spec_from_file_location("fdt", os.path.join(os.path.dirname(TIMES_FIXTURE), "make_fulldepth_times.py"))
    FD = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(FD)
    if not os.path.exists(path):
        pytest.skip(f"{os.path.basename(path)} not generated (tests/golden/make_fulldepth_times.py --geom {geom}, ~40 min of CPU)")
    fx = np.load(path)
    inp = FD.geom_inputs(geom)
    assert float(fx["x_sum"]) == inp["x"].double().sum().item()
    st = int(fx["token_stride"])
    got = _call(procedural_full_model, inp, torch.tensor([float(fx["times"][0])], dtype=torch.float32)).float().cpu()
    assert torch.isfinite(got).all()
    got = got[:, ::st]
    b16 = torch.tensor(fx["bf16_0"]).view(torch.bfloat16).float()
    f32 = torch.tensor(fx["fp32_0"].astype("float32"))
    floor, e16, e32 = rel_l2(b16, f32), rel_l2(got, b16), rel_l2(got, f32)
    parity_log(f"[full depth 19+38, {geom} (L = {512 + inp['x'].shape[1]}), procedural weights] t = {float(fx['times'][0]):.2f}: HIP vs "
               f"bf16-merged oracle {e16:.3e}, vs fp32-ref oracle {e32:.3e}, oracle bf16-vs-fp32 floor {floor:.3e}")
    assert e16 < 1.5 * floor and e32 < 2.0 * floor, (geom, e16, e32, floor)


def test_full_model_fused_equals_eager_and_is_deterministic():
    from visualcloze_amd.transport import Sampler, create_transport
    m = _build(19, 38)
    inp = _inputs("cfg2", 3)
    kw, x = _kw(inp), inp["x"].to(DEV, torch.bfloat16)
    fn = Sampler(create_transport()).sample_ode(sampling_method="euler", num_steps=3, do_shift=True, time_shifting_factor=1)
    fused = fn(x, m.forward, kw)
    fused2 = fn(x, m.forward, kw)
    eager = fn(x, lambda xx, **k: m.forward(xx, **k), kw)
    torch.cuda.synchronize()
    assert fused.shape == (1, 1, N2, 64) and torch.isfinite(fused.float()).all()
    assert torch.equal(fused, fused2)
    assert rel_l2(fused, eager) < 5e-3            # same kernels; eager does the Euler update with torch


def test_race_screen_repeated_launches_are_bit_identical():
    """The GEMM main loops (hand-placed barriers / vmcnt / lgkmcnt, LDS-DMA) and the attention ring must be
    deterministic: 40 back-to-back launches per config on full-size operands, under load from each other, have to
    reproduce the first result bit for bit (a read racing a DMA shows up as rare differing tiles).  36 (256x192 +
    loader waves) and 34 (256x128 + loader waves) are the tiles the cost model actually picks; 0 = its own choice."""
    from visualcloze_amd import hip
    g = torch.Generator().manual_seed(17)
    a = torch.randn(L2, D, generator=g).to(torch.bfloat16).to(DEV)
    w = (torch.randn(3 * D, D, generator=g) * D ** -0.5).to(torch.bfloat16).to(DEV)
    b = torch.randn(3 * D, generator=g).to(torch.bfloat16).to(DEV)
    for cfg in (36, 34, 0, 1, 5, 19, 20):
        for epi in ((hip.EPI_GELU, hip.EPI_BIAS) if cfg in (36, 34) else (hip.EPI_GELU,)):
            outs = [torch.empty(L2, 3 * D, dtype=torch.bfloat16, device=DEV) for _ in range(2)]
            ps = [hip.make_problem(a, w, b, o) for o in outs]
            hip.gemm(ps[0], epi=epi, tile_cfg=cfg)
            torch.cuda.synchronize()
            for it in range(40):
                hip.gemm(ps[1], epi=epi, tile_cfg=cfg)
                if it % 8 == 7:
                    torch.cuda.synchronize()
                    assert torch.equal(outs[0], outs[1]), f"cfg {cfg} epi {epi}: launch {it} differs"
    # gated-residual epilogue (in place: the residual is re-read) on the product tile, K = 12288 (ring wrap-around)
    a4 = torch.randn(L2, 4 * D, generator=g).to(torch.bfloat16).to(DEV)
    w4 = (torch.randn(D, 4 * D, generator=g) * (4 * D) ** -0.5).to(torch.bfloat16).to(DEV)
    gate = torch.randn(D, generator=g).to(torch.bfloat16).to(DEV)
    res = torch.randn(L2, D, generator=g).to(torch.bfloat16).to(DEV)
    first = None
    for it in range(24):
        x = res.clone()
        hip.gemm(hip.make_problem(a4, w4, b[:D], x, res=x, gate=gate), epi=hip.EPI_GATE_RES, tile_cfg=36)
        if it % 8 == 7:
            torch.cuda.synchronize()
            if first is None:
                first = x.clone()
            assert torch.equal(first, x), f"GATE_RES cfg 36: launch {it} differs"
    # the split-K remainder as the SDEdit stage runs it (L = 4608: 256 whole tiles + 32 tiles x 8 K-slices + the reduce launch),
    # chosen by the launcher itself, gate + residual in place: slices racing the reduce, or a reduce reading a stale partial of
    # the previous launch in the shared scratch, would show up here
    Ls = 4608
    a5 = torch.randn(Ls, 4 * D, generator=g).to(torch.bfloat16).to(DEV)
    res5 = torch.randn(Ls, D, generator=g).to(torch.bfloat16).to(DEV)
    ws = hip.splitk_workspace(DEV)
    a5b = (a5.float() * 0.5).to(torch.bfloat16)           # a second problem through the SAME scratch between the repeats
    first = None
    for it in range(24):
        x = res5.clone()
        y = res5.clone()
        hip.gemm(hip.make_problem(a5, w4, b[:D], x, res=x, gate=gate), epi=hip.EPI_GATE_RES, tile_cfg=0, splitk_ws=ws)
        hip.gemm(hip.make_problem(a5b, w4, b[:D], y, res=y, gate=gate), epi=hip.EPI_GATE_RES, tile_cfg=0, splitk_ws=ws)
        if it % 8 == 7:
            torch.cuda.synchronize()
            if first is None:
                first = (x.clone(), y.clone())
                one = res5.clone()
                hip.gemm(hip.make_problem(a5, w4, b[:D], one, res=one, gate=gate), epi=hip.EPI_GATE_RES, tile_cfg=hip.GEMM_NO_SPLITK, splitk_ws=ws)
                torch.cuda.synchronize()
                assert not torch.equal(one, x) and rel_l2(x, one) < 2e-3        # the split WAS taken; same function
            assert torch.equal(first[0], x) and torch.equal(first[1], y), f"split-K GATE_RES: launch {it} differs"
    qkv = torch.randn(L2, 3 * D, generator=g).to(torch.bfloat16).to(DEV)
    vt = qkv[:, 2 * D:].reshape(L2, H, 128).permute(1, 2, 0).contiguous()
    for variant in (0, 1, 2, 3, 7, 8, 12):
        o0 = torch.empty(L2, D, dtype=torch.bfloat16, device=DEV)
        o1 = torch.empty(L2, D, dtype=torch.bfloat16, device=DEV)
        hip.attention(qkv, vt, o0, L2, H, variant=variant)
        for it in range(40):
            hip.attention(qkv, vt, o1, L2, H, variant=variant)
        torch.cuda.synchronize()
        assert torch.equal(o0, o1), f"attention variant {variant} not deterministic"
        if variant == 12:      # tail items cut along the keys and merged: the same function as the uncut kernel
            o8 = torch.empty_like(o0)
            hip.attention(qkv, vt, o8, L2, H, variant=8)                # the same kernel, no item cut: f32 summation order only
            torch.cuda.synchronize()
            assert rel_l2(o0, o8) < 4e-3 and not torch.equal(o0, o8)
    # the stream form (bounded logits, prescaled queries: the product's launches), pieces combined by the merge kernel (12) and
    # inside the launch (28): 40 launches each, bit-reproducible and equal to each other
    outs = {}
    for variant in (12, 28):
        o0 = torch.empty(L2, D, dtype=torch.bfloat16, device=DEV)
        o1 = torch.empty(L2, D, dtype=torch.bfloat16, device=DEV)
        hip.attention(qkv, vt, o0, L2, H, variant=variant, q_prescaled=True, logit_bound=16.65)
        for it in range(40):
            hip.attention(qkv, vt, o1, L2, H, variant=variant, q_prescaled=True, logit_bound=16.65)
        torch.cuda.synchronize()
        assert torch.equal(o0, o1), f"stream-form attention, variant {variant}: not deterministic"
        outs[variant] = o0
    assert torch.equal(outs[12], outs[28])
