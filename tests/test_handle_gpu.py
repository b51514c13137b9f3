"""The handle API (vc_flux_*, include/vcloze_hip.h): Flux.forward and the whole Euler loop as one C call
each.  The launch plan in csrc/flux_engine.hip must produce the SAME BITS as the Python-ordered plan (engine.FluxEngine)
over the op-level ABI - same kernels, same order, same operands - which the other GPU tests pin to the oracle and the
reference's golden vectors; here additionally against the golden vectors directly, plus the API's error behaviour."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


def rel_l2(a, b):
    a, b = torch.as_tensor(a).float().cpu(), torch.as_tensor(b).float().cpu()
    return ((a - b).norm() / b.norm()).item()


@pytest.fixture(scope="module")
def model():
    from tests.helpers import tiny_model
    return tiny_model()


def _fwd(m, inp, t, use_handle):
    m.use_handle = use_handle
    img = torch.cat((inp["x"], inp["cond"]), -1)
    out = m(img.to(DEV, torch.bfloat16), img_ids=inp["img_ids"].to(DEV), txt=inp["txt"].to(DEV, torch.bfloat16),
            txt_ids=inp["txt_ids"].to(DEV), timesteps=t.to(DEV), y=inp["y"].to(DEV, torch.bfloat16),
            txt_mask=inp["txt_mask"].to(DEV), img_mask=inp["img_mask"].to(DEV), guidance=inp["guidance"].to(DEV))
    torch.cuda.synchronize()
    m.use_handle = True
    return out


def _kw(inp):
    return dict(txt=inp["txt"].to(DEV, torch.bfloat16), txt_ids=inp["txt_ids"].to(DEV), txt_mask=inp["txt_mask"].to(DEV),
                y=inp["y"].to(DEV, torch.bfloat16), img_ids=inp["img_ids"].to(DEV), img_mask=inp["img_mask"].to(DEV),
                cond=inp["cond"].to(DEV, torch.bfloat16), guidance=inp["guidance"].to(DEV))


def _masks(kind, inp):
    if kind == "ragged":
        inp["img_mask"][1, -12:] = 0
    elif kind == "holes":
        inp["txt_mask"][0, [0, 5]] = 0
        inp["txt_mask"][1, -4:] = 0
        inp["img_mask"][0, [1, 2, 9]] = 0
    return inp


@pytest.mark.parametrize("variant", [None, 3, 12])
@pytest.mark.parametrize("masks", ["full", "ragged", "holes"])
def test_forward_handle_equals_python_plan_bitwise(model, golden, masks, variant):
    from tests.procedural import tiny_inputs
    m, _ = model
    m.engine().attn_variant = variant
    try:
        inp = _masks(masks, tiny_inputs(B=2, seed=7))
        t = torch.tensor([0.9, 0.25])
        a, b = _fwd(m, inp, t, True), _fwd(m, inp, t, False)
        assert m.handle() is not None
        assert torch.equal(a, b)
        if masks == "ragged":
            assert rel_l2(a, golden["flux_b2"]) < 3e-2          # the reference's own Flux.forward on these inputs
    finally:
        m.engine().attn_variant = None


def test_forward_handle_vs_golden_b1(model, golden):
    from tests.procedural import tiny_inputs
    m, _ = model
    got = _fwd(m, tiny_inputs(B=1), torch.tensor([0.7]), True)
    assert rel_l2(got, golden["flux_b1"]) < 3e-2


@pytest.mark.parametrize("masks", ["full", "holes"])
def test_sampler_handle_equals_python_plan_bitwise_and_golden(model, golden, masks):
    """vc_flux_sample_euler (one call per trajectory) against graph replays ordered from Python, state by state."""
    from tests.procedural import tiny_inputs
    from visualcloze_amd.transport import Sampler, create_transport
    m, _ = model
    B = 1 if masks == "full" else 2
    inp = _masks(masks, tiny_inputs(B=B) if B == 1 else tiny_inputs(B=2, seed=11))
    fn = Sampler(create_transport()).sample_ode(sampling_method="euler", num_steps=5, do_shift=True, time_shifting_factor=1)
    x = inp["x"].to(DEV, torch.bfloat16)
    x_before = x.clone()
    a = fn(x, m.forward, _kw(inp))
    m.use_handle = False
    try:
        b = fn(x, m.forward, _kw(inp))
    finally:
        m.use_handle = True
    assert torch.equal(x, x_before)                            # the caller's state is never updated in place
    assert a.shape == b.shape == (1, B) + tuple(inp["x"].shape[1:])
    assert torch.equal(a, b)
    if masks == "full":
        assert rel_l2(a[-1], golden["traj_states"][-1]) < 6e-2  # the reference's own sample_ode run (fp32)


def test_sampler_trajectory_and_piecewise_steps(model):
    """trajectory buffer = the state after every step; begin / steps(k) / steps(S-k) / end == sample_euler."""
    from tests.procedural import tiny_inputs
    from visualcloze_amd.transport import solver_time_grid
    m, _ = model
    inp = tiny_inputs(B=1)
    h = m.handle()
    kw = _kw(inp)
    S = 4
    t = solver_time_grid(S + 1, inp["x"].shape[1], 0.0, 1, True, 1)
    st = m.engine().stream
    with torch.cuda.stream(st):
        s = st.cuda_stream
        h.prepare(kw["txt"], kw["y"], kw["guidance"], False, kw["img_ids"], kw["txt_ids"], S, stream=s)
        x1 = inp["x"].to(DEV, torch.bfloat16).clone()
        traj = torch.empty((S,) + tuple(x1.shape), dtype=torch.bfloat16, device=DEV)
        h.sample_euler(x1, kw["cond"], t, True, s, trajectory=traj)
        x2 = inp["x"].to(DEV, torch.bfloat16).clone()
        h.sample_begin(x2, kw["cond"], t, True, s)
        h.sample_steps(1, s)
        mid = torch.empty_like(x2)
        h.sample_end(mid, s)
        h.sample_steps(S - 1, s)
        out = torch.empty_like(x2)
        h.sample_end(out, s)
    torch.cuda.synchronize()
    assert torch.equal(traj[-1], x1) and torch.equal(out, x1) and torch.equal(mid, traj[0])
    assert torch.equal(x2, inp["x"].to(DEV, torch.bfloat16))   # begin / steps never write the caller's x
    assert not torch.equal(traj[0], traj[1])
    with pytest.raises(Exception, match="more steps"):
        h.sample_steps(1, st.cuda_stream)


def test_profile_times_the_plans_own_launches_and_leaves_the_trajectory_alone(model):
    """vc_flux_profile (ABI 10, bench.py's roofline leg): the launch classes of whole evaluations issued by the handle's own
    plan - the counts and the arithmetic are the plan's, every class has a positive time, and a trajectory with a profile in
    its middle ends in the same bits as one without."""
    from tests.procedural import tiny_inputs
    from visualcloze_amd import hip
    from visualcloze_amd.transport import solver_time_grid
    m, _ = model
    inp = tiny_inputs(B=2)
    h = m.handle()
    kw = _kw(inp)
    S, E = 3, 2
    t = solver_time_grid(S + 1, inp["x"].shape[1], 0.0, 1, True, 1)
    st = m.engine().stream
    g = m.engine().g
    B, N, T = inp["x"].shape[0], inp["x"].shape[1], inp["txt"].shape[1]
    L, D, mlp = N + T, g.hidden_size, int(g.hidden_size * g.mlp_ratio)
    with torch.cuda.stream(st):
        s = st.cuda_stream
        h.prepare(kw["txt"], kw["y"], kw["guidance"], False, kw["img_ids"], kw["txt_ids"], S, stream=s)
        with pytest.raises(Exception, match="sample_begin"):
            h.profile(1, s)                                   # no sample in flight
        x1 = inp["x"].to(DEV, torch.bfloat16).clone()
        h.sample_euler(x1, kw["cond"], t, True, s)
        x2 = inp["x"].to(DEV, torch.bfloat16).clone()
        h.sample_begin(x2, kw["cond"], t, True, s)
        h.sample_steps(1, s)
        recs = h.profile(E, s)
        h.sample_steps(S - 1, s)
        out = torch.empty_like(x2)
        h.sample_end(out, s)
    torch.cuda.synchronize()
    assert torch.equal(out, x1)
    att = [r for r in recs if r["kind"] == hip.LAUNCH_ATTENTION]
    assert len(att) == 1 and att[0]["launches"] == E * (g.depth + g.depth_single_blocks)
    assert att[0]["flops"] == pytest.approx(att[0]["launches"] * 4.0 * L * L * D * B)
    gate = [r for r in recs if r["kind"] == hip.LAUNCH_GEMM and r["epi"] == hip.EPI_GATE_RES]
    assert sum(r["launches"] for r in gate) == E * (2 * g.depth + g.depth_single_blocks)
    assert sum(r["flops"] for r in gate) == pytest.approx(E * B * (g.depth * (2.0 * L * D * D + 2.0 * L * D * mlp) + g.depth_single_blocks * 2.0 * L * D * (D + mlp)))
    ln = [r for r in recs if r["kind"] == hip.LAUNCH_LN_MODULATE]
    assert sum(r["launches"] for r in ln) == E * (2 * g.depth + g.depth_single_blocks + 1)
    for r in recs:
        assert r["launches"] > 0 and 0 < r["min_us"] <= r["total_us"] / r["launches"] <= r["max_us"] < 1e5, r


def test_handle_error_behaviour(model):
    from tests.procedural import TINY, tiny_inputs
    from visualcloze_amd import hip
    from visualcloze_amd.handle import FluxHandle
    m, _ = model
    L = hip.lib()
    p = m.params
    D = p.hidden_size
    cfg = hip.FluxConfig(p.in_channels, p.out_channels, p.vec_in_dim, p.context_in_dim, D, p.num_heads, p.depth,
                         p.depth_single_blocks, int(D * p.mlp_ratio), 1, (C.c_int32 * 3)(*p.axes_dim), p.theta)
    h = C.c_void_p()
    assert L.vc_flux_create(C.byref(cfg), C.byref(h)) == 0
    try:
        ws = torch.empty(L.vc_flux_workspace_bytes(h, 1, 16, 24, 2) + 256, dtype=torch.uint8, device=DEV)
        base = (ws.data_ptr() + 255) & ~255
        inp = tiny_inputs(B=1)
        kw = _kw(inp)
        f32 = lambda a: np.ascontiguousarray(a.float().cpu().numpy())  # noqa: E731
        ii, ti, g = f32(inp["img_ids"]), f32(inp["txt_ids"]), f32(inp["guidance"])
        fp = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))  # noqa: E731
        args = hip.FluxInputs(1, 16, 24, 2, kw["txt"].data_ptr(), kw["y"].data_ptr(), fp(g), fp(ii), fp(ti), None, None, 0, 0)
        assert L.vc_flux_prepare(h, C.byref(args), base, ws.numel() - 256, None) == -3           # VC_ERR_STATE
        assert b"not bound" in L.vc_last_error()
        out = torch.empty(1, 24, 64, dtype=torch.bfloat16, device=DEV)
        t = np.asarray([0.5], np.float32)
        assert L.vc_flux_forward(h, out.data_ptr(), fp(t), 0, out.data_ptr(), None) == -3        # not prepared
        assert L.vc_flux_set_option(h, b"no_such_knob", 1) == -1
        # the optional split-K scratch changes the carve-up of every workspace: a first bind AFTER a workspace has been sized
        # (vc_flux_workspace_bytes above) is refused, not silently applied (advisor r05)
        skw = torch.empty(64, dtype=torch.float32, device=DEV)
        assert L.vc_flux_bind_weight(h, b"splitk_ws", skw.data_ptr(), None, 1, 64, 64) == -3 and b"BEFORE" in L.vc_last_error()
        assert L.vc_flux_mod_offset(h, b"double_blocks.0.txt_mod.lin") == 6 * D
        assert L.vc_flux_mod_offset(h, b"nope") == -1
        bad = hip.FluxConfig(p.in_channels, p.out_channels, p.vec_in_dim, p.context_in_dim, D + 8, p.num_heads, p.depth,
                             p.depth_single_blocks, int(D * p.mlp_ratio), 1, (C.c_int32 * 3)(*p.axes_dim), p.theta)
        h2 = C.c_void_p()
        assert L.vc_flux_create(C.byref(bad), C.byref(h2)) == -1 and b"head_dim 128" in L.vc_last_error()
    finally:
        assert L.vc_flux_destroy(h) == 0
    # a fully bound handle: wrong shapes / too small a workspace / too many steps are refused with a message
    fh = FluxHandle(m.params, m.engine().W, m.engine().dev)
    inp = tiny_inputs(B=1)
    kw = _kw(inp)
    fh.prepare(kw["txt"], kw["y"], kw["guidance"], False, kw["img_ids"], kw["txt_ids"], 2)
    x = inp["x"].to(DEV, torch.bfloat16).clone()
    from visualcloze_amd.transport import solver_time_grid
    with pytest.raises(hip.VclozeHipError, match="workspace holds"):
        fh.sample_euler(x, kw["cond"], solver_time_grid(6, 24, 0.0, 1, True, 1), True, m.engine().stream.cuda_stream)
    with pytest.raises(hip.VclozeHipError, match="guidance"):
        fh.prepare(kw["txt"], kw["y"], None, False, kw["img_ids"], kw["txt_ids"], 2)
    with pytest.raises(hip.VclozeHipError, match="kv_len"):
        fh.prepare(kw["txt"], kw["y"], kw["guidance"], False, kw["img_ids"], kw["txt_ids"], 2, kv_len=[41])
    # re-binding a weight invalidates the prepared state (and every captured step): forward refuses until prepared again
    fh.prepare(kw["txt"], kw["y"], kw["guidance"], False, kw["img_ids"], kw["txt_ids"], 2)
    W = m.engine().W
    fh._bind("img_in", W.w["img_in"], W.b["img_in"], *W.w["img_in"].shape)
    img = torch.cat((inp["x"], inp["cond"]), -1).to(DEV, torch.bfloat16)
    out = torch.empty(1, 24, 64, dtype=torch.bfloat16, device=DEV)
    with pytest.raises(hip.VclozeHipError, match="vc_flux_prepare first"):
        fh.forward(img, torch.tensor([0.5]), False, out)
    fh.prepare(kw["txt"], kw["y"], kw["guidance"], False, kw["img_ids"], kw["txt_ids"], 2)
    fh.forward(img, torch.tensor([0.5]), False, out)
    torch.cuda.synchronize()
    # a handle built DIRECTLY over head-permuted weights reads qkv_heads from them (advisor r03: it used to default to 0 and
    # scatter q/k/v into scrambled columns); with the engine's other options - qkv_heads omitted - same bits as the model's own
    assert torch.isfinite(out.float()).all()
    e = m.engine()
    fh.set_options(e.attn_variant, e.tile_cfg, e.fuse_qnorm, e.fuse_vt, None, e.fuse_knorm,
                   e.W.logit_bound if e.bounded_softmax else 0.0, e.mlp_first, e.splitk)
    fh.prepare(kw["txt"], kw["y"], kw["guidance"], False, kw["img_ids"], kw["txt_ids"], 2)
    fh.forward(img, torch.tensor([0.5]), False, out)
    torch.cuda.synchronize()
    want = m.forward(img, kw["img_ids"], kw["txt"], kw["txt_ids"], torch.tensor([0.5], device=DEV), kw["y"], guidance=kw["guidance"])
    assert m.engine().W.qkv_heads > 0 and torch.equal(out, want.to(out.dtype))
    with pytest.raises(hip.VclozeHipError, match="follows the weights"):
        fh.set_options(qkv_heads=0)


def test_step_graph_cache_eviction_and_recapture(model):
    """The handle keeps the captured step of the 4 most recent geometries (a two-stage pipeline alternates between two);
    cycling through 6 geometries twice evicts and re-captures every one of them - the second pass must reproduce the
    first bit for bit, and so must a pass through the Python-ordered plan."""
    from tests.procedural import tiny_inputs
    from visualcloze_amd.transport import Sampler, create_transport
    m, _ = model
    fn = Sampler(create_transport()).sample_ode(sampling_method="euler", num_steps=3, do_shift=True, time_shifting_factor=1)
    geoms = [((4, 12), (4, 12)), ((4, 12),), ((4, 8), (4, 8)), ((2, 12), (2, 12), (2, 12)), ((4, 16),), ((6, 12), (6, 12))]
    cases = []
    for i, rows in enumerate(geoms):
        inp = tiny_inputs(B=1, rows_hw=rows, seed=20 + i)
        cases.append((inp["x"].to(DEV, torch.bfloat16), _kw(inp)))
    first = [fn(x, m.forward, kw) for x, kw in cases]
    second = [fn(x, m.forward, kw) for x, kw in cases]
    m.use_handle = False
    try:
        third = [fn(x, m.forward, kw) for x, kw in cases]
    finally:
        m.use_handle = True
    torch.cuda.synchronize()
    for a, b, c in zip(first, second, third):
        assert torch.isfinite(a.float()).all() and torch.equal(a, b) and torch.equal(a, c)
    assert len({tuple(a.shape) for a in first}) >= 4


def test_host_copy_cache_hits_through_the_product_sampler(model):
    """handle._HostCopies (advisor r04): vc_flux_prepare takes the position ids and the guidance as HOST arrays; the D2H copies
    that would drain the stream once per sample are remembered while the caller hands over the same memory at the same version.
    The product path slices its arguments per chunk - a new view OBJECT of the same storage per call - so the key must be the
    memory: the second sample through `Sampler.sample_ode` must hit for all three arguments, an in-place edit of the ids must
    miss (and change the result), and a different tensor of equal contents must give equal bits."""
    from tests.procedural import tiny_inputs
    from visualcloze_amd.transport import Sampler, create_transport
    m, _ = model
    fn = Sampler(create_transport()).sample_ode(sampling_method="euler", num_steps=4, do_shift=True, time_shifting_factor=1)
    inp = tiny_inputs(B=1)
    kw, x = _kw(inp), inp["x"].to(DEV, torch.bfloat16)
    a = fn(x, m.forward, kw)
    hc = m.handle()._host
    h0, m0 = hc.hits, hc.misses
    b = fn(x, m.forward, kw)
    torch.cuda.synchronize()
    assert torch.equal(a, b)
    assert hc.hits - h0 == 3 and hc.misses == m0, (hc.hits - h0, hc.misses - m0)       # guidance, img_ids, txt_ids: no D2H copy
    kw2 = dict(kw, img_ids=kw["img_ids"].clone(), txt_ids=kw["txt_ids"].clone(), guidance=kw["guidance"].clone())
    c = fn(x, m.forward, kw2)                                                          # other memory, equal contents
    assert torch.equal(a, c) and hc.misses - m0 == 3
    kw2["img_ids"][0, :, 1] += 3.0                                                     # same memory, new version: positions moved
    d = fn(x, m.forward, kw2)
    torch.cuda.synchronize()
    assert not torch.equal(a, d)
