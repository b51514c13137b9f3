"""CPU: the oracle (oracle/flux_oracle.py) against the golden vectors produced by the REFERENCE ITSELF
(tests/golden/make_golden.py ran /root/reference under shims in the build container)."""
import numpy as np
import pytest
import torch

import oracle.flux_oracle as O
from tests.procedural import TINY, procedural_param, ptensor, tiny_inputs

G = O.FluxGeometry(**TINY)
F32 = O.Prec("fp32")


def close(a, b, tol=2e-5):
    a, b = torch.as_tensor(a).float(), torch.as_tensor(b).float()
    assert a.shape == b.shape, (a.shape, b.shape)
    err = (a - b).abs().max().item()
    assert err <= tol * max(1.0, b.abs().max().item()), err


def test_rope_table_and_apply(golden):
    ids = torch.tensor(golden["pe_ids"])
    cs = O.rope_cos_sin(ids, G.axes_dim, G.theta)                  # [B,L,64,2]
    pe = torch.tensor(golden["pe"])[:, 0]                          # [B,L,64,2,2] = [[cos,-sin],[sin,cos]]
    close(cs[..., 0], pe[..., 0, 0]); close(cs[..., 1], pe[..., 1, 0]); close(-cs[..., 1], pe[..., 0, 1])
    assert float(ids[0, -1, 0]) == 2.0                             # row index + 1 (sampling.py:57)
    q = torch.tensor(golden["rope_q_in"])
    close(O.apply_rope(q, cs, F32), golden["rope_q_out"])
    close(O.apply_rope(ptensor(q.shape, 12, q=6), cs, F32), golden["rope_k_out"])


def test_timestep_embedding(golden):
    close(O.timestep_embedding(torch.tensor(golden["temb_t"])), golden["temb"], 1e-4)
    close(O.timestep_embedding(torch.tensor([30.0])), golden["temb_g30"], 1e-3)


def test_qknorm_and_modulation(golden, tiny_sd):
    L = golden["pe_ids"].shape[1]
    q, k = ptensor((1, 2, L, 128), 11, q=6), ptensor((1, 2, L, 128), 12, q=6)
    pf = "double_blocks.0.img_attn.norm"
    close(O.rms_norm(q, tiny_sd[pf + ".query_norm.scale"], F32), golden["qknorm_q"])
    close(O.rms_norm(k, tiny_sd[pf + ".key_norm.scale"], F32), golden["qknorm_k"])
    vec = torch.tensor(golden["mod_vec"])
    out = torch.cat(O.modulation(tiny_sd, "double_blocks.0.img_mod", vec, 6, F32), dim=-1)
    close(out, golden["mod_out"])                                  # chunk order shift,scale,gate x2


def test_attention_full_and_ragged(golden):
    L = golden["pe_ids"].shape[1]
    cs = O.rope_cos_sin(torch.tensor(golden["pe_ids"]), G.axes_dim, G.theta).repeat(2, 1, 1, 1)
    q, k, v = (ptensor((2, 2, L, 128), s, q=6) for s in (21, 22, 23))
    rq, rk = O.apply_rope(q, cs, F32), O.apply_rope(k, cs, F32)
    close(O.sdpa(rq, rk, v, F32), golden["attn_full"])
    kv = [int(x) for x in golden["attn_ragged_mask"].sum(1)]
    got = O.sdpa(rq, rk, v, F32, kv_len=kv)
    close(got, golden["attn_ragged"])
    assert float(got[1, kv[1]:].abs().max()) == 0.0               # padded query rows -> zeros (pad_input)
    gm = torch.tensor(golden["attn_general_mask"])                 # holes anywhere: _upad_input gathers any mask
    got = O.sdpa(rq, rk, v, F32, kv_len=gm)
    close(got, golden["attn_general"])
    assert float(got[gm == 0].abs().max()) == 0.0


def test_lora_rank_clip(golden):
    keys = [str(k) for k in golden["lora_clip_keys"]]
    shapes = [tuple(int(x) for x in str(s).split(",")) for s in golden["lora_clip_shapes"]]
    assert dict(zip(keys, shapes))["lora_A.weight"] == (4, 12)     # rank clipped to min(in, out)
    sd = {"l." + k: procedural_param("lltest." + k, s) for k, s in zip(keys, shapes)}
    close(O.linear(sd, "l", torch.tensor(golden["lora_clip_in"]), F32, lora_scale=0.5), golden["lora_clip_out"])
    close(O.linear(sd, "l", torch.tensor(golden["lora_clip_in"]), O.Prec("fp32", "merged"), lora_scale=0.5),
          golden["lora_clip_out"])


def test_blocks(golden, tiny_sd):
    cs = O.rope_cos_sin(torch.tensor(golden["pe_ids"]), G.axes_dim, G.theta)
    img, txt = torch.tensor(golden["blk_img_in"]), torch.tensor(golden["blk_txt_in"])
    vec = torch.tensor(golden["mod_vec"])
    di, dt = O.double_block(tiny_sd, "double_blocks.0", img, txt, vec, cs, G, F32)
    close(di, golden["double0_img"]); close(dt, golden["double0_txt"])
    close(O.single_block(tiny_sd, "single_blocks.0", torch.cat((txt, img), 1), vec, cs, G, F32), golden["single0"])
    close(O.last_layer(tiny_sd, img, vec, F32), golden["last"])


def _fwd(sd, inp, t, P):
    return O.flux_forward(sd, G, torch.cat((inp["x"], inp["cond"]), -1), inp["img_ids"], inp["txt"], inp["txt_ids"], t,
                          inp["y"], inp["txt_mask"], inp["img_mask"], inp["guidance"], P=P)


def test_flux_forward(golden, tiny_sd):
    close(_fwd(tiny_sd, tiny_inputs(B=1), torch.tensor(golden["flux_b1_t"]), F32), golden["flux_b1"])
    close(_fwd(tiny_sd, tiny_inputs(B=1), torch.tensor(golden["flux_b1_t"]), O.Prec("fp32", "merged")), golden["flux_b1"])
    inp2 = tiny_inputs(B=2, seed=7)
    inp2["img_mask"][1, -12:] = 0
    close(_fwd(tiny_sd, inp2, torch.tensor(golden["flux_b2_t"]), F32), golden["flux_b2"])
    inp3 = tiny_inputs(B=2, seed=7)                                # non-prefix masks in both streams
    inp3["txt_mask"], inp3["img_mask"] = torch.tensor(golden["flux_general_txt_mask"]), torch.tensor(golden["flux_general_img_mask"])
    close(_fwd(tiny_sd, inp3, torch.tensor(golden["flux_b2_t"]), F32), golden["flux_general"])
    inp4 = tiny_inputs(B=2, seed=13)                               # sample 1 without any text (txt_mask all zeros)
    inp4["txt_mask"], inp4["img_mask"] = torch.tensor(golden["flux_notext_txt_mask"]), torch.tensor(golden["flux_notext_img_mask"])
    assert int(inp4["txt_mask"][1].sum()) == 0
    close(_fwd(tiny_sd, inp4, torch.tensor([0.8, 0.3]), F32), golden["flux_notext"])


def test_errors(tiny_sd):
    inp = tiny_inputs(B=1)
    with pytest.raises(ValueError):
        O.flux_forward(tiny_sd, G, inp["x"][0], inp["img_ids"], inp["txt"], inp["txt_ids"], torch.ones(1), inp["y"])
    with pytest.raises(ValueError):
        O.flux_forward(tiny_sd, G, torch.cat((inp["x"], inp["cond"]), -1), inp["img_ids"], inp["txt"], inp["txt_ids"],
                       torch.ones(1), inp["y"], guidance=None)


@pytest.mark.parametrize("n_tok", [1152, 3456, 6144, 6912])
@pytest.mark.parametrize("steps", [4, 30, 50])
def test_time_grids(golden, n_tok, steps):
    t = O.time_grid(steps, n_tok, do_shift=True, time_shifting_factor=1)
    close(t[:-1].double(), golden[f"grid_{n_tok}_{steps}_solver_t"], 1e-6)
    close((1 - t[:-1]).double(), golden[f"grid_{n_tok}_{steps}_model_t"], 1e-6)
    assert len(golden[f"grid_{n_tok}_{steps}_model_t"]) == steps - 1     # N points -> N-1 model evaluations
    assert abs(float(t[0])) < 1e-7 and float(t[-1]) == 1.0   # endpoints survive the shift via inf arithmetic


def test_time_grid_upsample(golden):
    t = O.time_grid(10, 4096, do_shift=False, time_shifting_factor=1.0, strength=0.4)
    close((1 - t[:-1]).double(), golden["grid_upsample_model_t"], 1e-6)


def test_sampler_trajectory(golden, tiny_sd):
    inp = tiny_inputs(B=1)
    kw = dict(P=F32)

    def model_fn(xin, tm):
        return O.flux_forward(tiny_sd, G, xin, inp["img_ids"], inp["txt"], inp["txt_ids"], tm, inp["y"], inp["txt_mask"],
                              inp["img_mask"], inp["guidance"], **kw)
    states, evals = O.sample_euler(model_fn, inp["x"], inp["cond"], O.time_grid(5, inp["x"].shape[1], True, 1))
    assert len(evals) == 4
    ref = golden["traj_states"]
    for i in range(5):
        close(states[i], ref[i], 5e-5)
    st2, _ = O.sample_euler(model_fn, inp["x"], inp["cond"], O.time_grid(4, inp["x"].shape[1], False, 1.0, strength=0.4))
    close(st2[-1], golden["traj_sdedit_last"], 5e-5)


def test_bf16_noise_floor(golden, tiny_sd):
    """How far bf16 execution moves the result (states the tolerance the GPU tests use)."""
    inp = tiny_inputs(B=1)
    ref = torch.tensor(golden["flux_b1"])
    orig = O.compute_vec
    O.compute_vec = lambda *a, **k: orig(*a, **{**k, "guidance_is_bf16": False})
    try:
        yb = _fwd(tiny_sd, inp, torch.tensor([0.7]), O.Prec("bf16", "ref"))
        ym = _fwd(tiny_sd, inp, torch.tensor([0.7]), O.Prec("bf16", "merged"))
    finally:
        O.compute_vec = orig
    rel = lambda a: ((a - ref).norm() / ref.norm()).item()  # noqa: E731
    assert 1e-3 < rel(yb) < 1.5e-2 and 1e-3 < rel(ym) < 1.5e-2
    # with guidance in bf16 (the production dtype) the oracle tracks the reference's own bf16 run
    yg = _fwd(tiny_sd, inp, torch.tensor([0.7]), O.Prec("bf16", "ref"))
    rb = torch.tensor(golden["flux_b1_ref_bf16"])
    assert ((yg - rb).norm() / rb.norm()).item() < 5e-2


def test_bf16_state_sampler_model_times_and_trajectory(golden, tiny_sd):
    """With the pipeline's bf16 state (visualcloze.py:399) torchdiffeq casts t to bf16 before the drift sees it:
    the reference's own bf16 run records Flux timesteps 1 - bf16(t_i).  The oracle's bf16 sampler reproduces that
    sequence exactly and tracks the reference's bf16 trajectory within bf16 noise."""
    inp = tiny_inputs(B=1)
    P = O.Prec("bf16", "ref")
    seen = golden["traj_bf16_model_t"]
    t = O.time_grid(5, inp["x"].shape[1], True, 1)
    assert np.array_equal((1 - t[:-1].to(torch.bfloat16).float()).double().numpy(), seen)
    assert not np.array_equal((1 - t[:-1]).double().numpy(), seen)        # the f32 grid would NOT match

    def model_fn(xin, tm):
        return O.flux_forward(tiny_sd, G, xin, inp["img_ids"], inp["txt"], inp["txt_ids"], tm, inp["y"], inp["txt_mask"],
                              inp["img_mask"], inp["guidance"], P=P)
    states, evals = O.sample_euler(model_fn, P.r(inp["x"]), P.r(inp["cond"]), t, P)
    assert np.array_equal(np.array(evals, dtype=np.float64), seen)
    ref = torch.tensor(golden["traj_bf16_states"])
    for i in range(1, 5):
        assert ((states[i] - ref[i]).norm() / ref[i].norm()).item() < 5e-2


def test_f32_state_sampler_keeps_the_state_in_f32(golden, tiny_sd):
    """An f32 state through the bf16 model (transport/integrators.py:119: odeint keeps y's dtype): the reference's own run
    records UNROUNDED Flux timesteps 1 - t_i and an f32 trajectory; the oracle's state_f32 mode reproduces the times
    exactly and tracks the states within bf16 noise."""
    inp = tiny_inputs(B=1)
    P = O.Prec("bf16", "ref")
    seen = golden["traj_f32state_model_t"]
    t = O.time_grid(5, inp["x"].shape[1], True, 1)
    assert np.array_equal((1 - t[:-1]).double().numpy(), seen)

    def model_fn(xin, tm):
        return O.flux_forward(tiny_sd, G, xin, inp["img_ids"], inp["txt"], inp["txt_ids"], tm, inp["y"], inp["txt_mask"],
                              inp["img_mask"], inp["guidance"], P=P)
    states, evals = O.sample_euler(model_fn, inp["x"], P.r(inp["cond"]), t, P, state_f32=True)
    assert np.array_equal(np.array(evals, dtype=np.float64), seen)
    ref = torch.tensor(golden["traj_f32state_states"])
    assert not torch.equal(P.r(states[-1]), states[-1])                    # the state really is finer than bf16
    for i in range(1, 5):
        assert ((states[i] - ref[i]).norm() / ref[i].norm()).item() < 5e-2


def test_oracle_at_full_width_against_the_reference_itself():
    """The oracle held to the REFERENCE at FLUX width once: `tests/golden/fullwidth_reference.npz` is the reference's own
    FluxLoraWrapper (1 DoubleStreamBlock + 1 SingleStreamBlock, hidden 3072, 24 heads, LoRA r256) run in fp32 on cfg 2's
    geometry (L = 512 + 3456) by tests/golden/make_fullwidth_reference.py; procedural weights and inputs, outputs only.  The
    fp32 / un-merged oracle must reproduce `Flux.forward` and the sampled block outputs to fp32 summation noise (K up to 15360
    per dot product: measured 2e-6 of the output's scale; bound 2e-5 as at the tiny geometry)."""
    import importlib.util
    import os
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    fx = np.load(os.path.join(here, "fullwidth_reference.npz"))
    spec = importlib.util.spec_from_file_location("make_fullwidth_traj", os.path.join(here, "make_fullwidth_traj.py"))
    FT = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(FT)
    inp = FT.inputs("cfg2")
    assert float(fx["x_sum"]) == inp["x"].double().sum().item()
    sd = {k: procedural_param(k, s, device="cpu").to(torch.bfloat16).float() for k, s in FT.key_shapes()}
    Gw = O.FluxGeometry(depth=1, depth_single_blocks=1)
    taps = {}
    orig = O.compute_vec
    O.compute_vec = lambda *a, **k: orig(*a, **{**k, "guidance_is_bf16": False})
    try:
        with torch.no_grad():
            y = O.flux_forward(sd, Gw, torch.cat((inp["x"], inp["cond"]), -1), inp["img_ids"], inp["txt"], inp["txt_ids"],
                               torch.tensor(fx["t"]), inp["y"], inp["txt_mask"], inp["img_mask"], inp["guidance"],
                               P=O.Prec("fp32", "ref"), taps=taps)
    finally:
        O.compute_vec = orig
    rs, cs_ = int(fx["row_stride"]), int(fx["col_stride"])
    close(y, fx["flux"])
    close(taps["double.0.img"][0, ::rs, ::cs_], fx["double_img"])
    close(taps["double.0.txt"][0, ::rs, ::cs_], fx["double_txt"])
    close(taps["single.0"][0, ::rs, ::cs_], fx["single"])
    err = ((y - torch.tensor(fx["flux"])).norm() / torch.tensor(fx["flux"]).norm()).item()
    print(f"oracle fp32 vs the reference at full width (1+1 blocks, L = 3968): rel-L2 {err:.2e}")
    assert err < 1e-5


def test_oracle_bf16_mode_at_full_width_against_the_reference_bf16_run():
    """The oracle's BF16 mode - WHERE `Prec.r` rounds - held to the reference's own bf16 run at FLUX width (the
    reference with bf16 parameters under `torch.autocast("cpu", torch.bfloat16)`, visualcloze.py:363; bf16 inputs and bf16
    guidance, visualcloze.py:399,413): `flux_bf16` / `*_bf16` of tests/golden/fullwidth_reference.npz, written by
    tests/golden/make_fullwidth_reference.py from the reference's FluxLoraWrapper (1 + 1 blocks, hidden 3072, 24 heads, LoRA
    r256 executed as lora.py:92-98 writes it, L = 3968).  Two bf16 implementations of one function differ by rounding noise
    that depends on summation order (mkldnn's bf16 GEMM vs the oracle's f32 matmul of bf16-rounded operands), so the bound
    is a tolerance, STATED: `Flux.forward` <= 1e-2, block outputs <= 5e-3 rel-L2 (measured 4.9e-3 and 1.7-2.2e-3; the
    bf16-vs-fp32 noise floor of the same evaluation is 9e-3 - the oracle's bf16 mode sits closer to the reference's bf16 run
    than bf16 sits to exact arithmetic).  bf16 guidance matters: 1000 * 30 rounds to 29952 before the sinusoid
    (`guidance_is_bf16`), which moves the output by 17 % - the fixture pins that too."""
    import importlib.util
    import os
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    fx = np.load(os.path.join(here, "fullwidth_reference.npz"))
    assert "flux_bf16" in fx.files, "regenerate tests/golden/fullwidth_reference.npz (make_fullwidth_reference.py, round 6)"
    spec = importlib.util.spec_from_file_location("make_fullwidth_traj", os.path.join(here, "make_fullwidth_traj.py"))
    FT = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(FT)
    inp = FT.inputs("cfg2")
    sd = {k: procedural_param(k, s, device="cpu").to(torch.bfloat16).float() for k, s in FT.key_shapes()}
    Gw = O.FluxGeometry(depth=1, depth_single_blocks=1)
    bf = lambda a: torch.tensor(a.view(np.int16)).view(torch.bfloat16).float()  # noqa: E731
    rel = lambda a, b: ((a - b).norm() / b.norm()).item()  # noqa: E731
    taps = {}
    with torch.no_grad():             # compute_vec's default: guidance_is_bf16 = True, the production dtype
        y = O.flux_forward(sd, Gw, torch.cat((inp["x"], inp["cond"]), -1), inp["img_ids"], inp["txt"], inp["txt_ids"],
                           torch.tensor(fx["t"]), inp["y"], inp["txt_mask"], inp["img_mask"], inp["guidance"],
                           P=O.Prec("bf16", "ref"), taps=taps)
    rs, cs_ = int(fx["row_stride"]), int(fx["col_stride"])
    ref = bf(fx["flux_bf16"]).reshape(y.shape)
    e = rel(y, ref)
    eb = {n: rel(taps[k][0, ::rs, ::cs_], bf(fx[n + "_bf16"])) for n, k in
          (("double_img", "double.0.img"), ("double_txt", "double.0.txt"), ("single", "single.0"))}
    floor = rel(ref, torch.tensor(fx["flux"]).reshape(y.shape))
    print(f"oracle bf16/ref vs the reference's bf16 autocast run at full width: Flux.forward {e:.2e}, blocks {eb}; "
          f"reference bf16-vs-fp32 (bf16 guidance included) {floor:.2e}")
    assert e < 1e-2 and all(v < 5e-3 for v in eb.values()), (e, eb)
    assert floor > 0.1                # the fixture really ran with bf16 guidance (29952, not 30000)
