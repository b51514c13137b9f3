"""CPU: host-side mirror of the reference interface, the C-ABI surface, and the oracle-vs-host time grids."""
import ctypes
import os
import re

import pytest
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_abi_exports_every_declared_symbol():
    from visualcloze_amd import hip
    if not os.path.exists(hip.LIB_PATH):
        hip.build()
    lib = ctypes.CDLL(hip.LIB_PATH)
    hdr = open(os.path.join(REPO, "include", "vcloze_hip.h")).read()
    declared = set(re.findall(r"\b(vc_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(hip.SYMBOLS), declared ^ set(hip.SYMBOLS)
    for name in declared:
        assert hasattr(lib, name), name
    # ... and nothing else: every exported vc_* symbol of the product library is declared in the header
    import subprocess
    nm = subprocess.run(["nm", "-D", "--defined-only", hip.LIB_PATH], capture_output=True, text=True).stdout
    exported = {ln.split()[-1] for ln in nm.splitlines() if " T " in ln and ln.split()[-1].startswith("vc_")}
    assert exported == declared, exported ^ declared
    assert hip.lib().vc_abi_version() == hip.ABI_VERSION


def test_struct_layout_matches_header():
    from visualcloze_amd import hip
    assert ctypes.sizeof(hip.GemmProblem) == 10 * 8 + 12 * 8 + 16 * 4     # 10 pointers (ABI 8: + qn_scale), 12 int64 (incl. the batch strides), 16 int32 (incl. the V^T and QK-norm fields)
    assert ctypes.sizeof(hip.GemmArgs) == 4 * ctypes.sizeof(hip.GemmProblem) + 8 + 8 + 8 + 8 + 8 + 8 + 6 * 4   # + splitk_ws, its size, sk_*, batch (ABI 7), sk_stream + pad (ABI 8)
    assert ctypes.sizeof(hip.FluxConfig) == 14 * 4
    assert ctypes.sizeof(hip.FluxInputs) == 4 * 4 + 7 * 8 + 2 * 4
    # the library reports the same sizes (and hip.lib() refuses to load one that does not)
    assert ctypes.sizeof(hip.FluxLaunchClass) == 6 * 4 + 2 * 8 + 4 * 4           # vc_flux_profile (ABI 10)
    sizes = (ctypes.c_int32 * 7)()
    hip.lib().vc_struct_sizes(sizes)
    assert list(sizes) == [ctypes.sizeof(c) for c in (hip.GemmProblem, hip.GemmArgs, hip.LnStream, hip.Attention, hip.FluxConfig,
                                                       hip.FluxInputs, hip.FluxLaunchClass)]


def test_no_gpu_fails_loudly():
    from visualcloze_amd import hip
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(hip.VclozeHipError):
        hip.require_gpu()
    from visualcloze_amd.model import FluxLoraWrapper, FluxParams
    from tests.procedural import TINY
    m = FluxLoraWrapper(lora_rank=4, params=FluxParams(**TINY))
    with pytest.raises(hip.VclozeHipError):                       # no CPU fallback
        m(torch.zeros(1, 8, 384), torch.zeros(1, 8, 3), torch.zeros(1, 4, 128), torch.zeros(1, 4, 3),
          torch.ones(1), torch.zeros(1, 64), guidance=torch.ones(1))


def test_state_dict_contract(golden):
    """B3: parameter names, shapes and order equal the reference's FluxLoraWrapper."""
    from visualcloze_amd.model import FluxLoraWrapper, FluxParams
    from tests.procedural import TINY, TINY_RANK
    m = FluxLoraWrapper(lora_rank=TINY_RANK, lora_scale=1.0, params=FluxParams(**TINY))
    mine = [(k, tuple(v.shape)) for k, v in m.state_dict().items()]
    ref = list(zip([str(k) for k in golden["keys"]],
                   [tuple(int(x) for x in str(s).split(",")) for s in golden["shapes"]]))
    assert mine == ref
    lora_only = {k: v for k, v in m.state_dict().items() if "lora" in k}
    missing, unexpected = m.load_state_dict(lora_only, strict=False)    # visualcloze.py:111-112
    assert not unexpected and all("lora" not in k for k in missing)
    assert m.double_blocks[0].img_attn.qkv.lora_B.weight.abs().sum() == 0   # lora_B zero-init (lora.py:84-86)


def test_constructor_errors():
    from visualcloze_amd.model import Flux, FluxParams
    from tests.procedural import TINY
    with pytest.raises(ValueError):
        Flux(FluxParams(**{**TINY, "hidden_size": 250}))
    with pytest.raises(ValueError):
        Flux(FluxParams(**{**TINY, "axes_dim": [16, 56, 48]}))


@pytest.mark.parametrize("args", [dict(num_steps=30, n=3456, do_shift=True, tsf=1, strength=None),
                                  dict(num_steps=4, n=1152, do_shift=True, tsf=1, strength=None),
                                  dict(num_steps=10, n=4096, do_shift=False, tsf=1.0, strength=0.4)])
def test_host_time_grid_equals_oracle(args):
    import oracle.flux_oracle as O
    from visualcloze_amd.transport import solver_time_grid
    t0 = 0 if args["strength"] is None else args["strength"]
    mine = solver_time_grid(args["num_steps"], args["n"], t0, 1, args["do_shift"], args["tsf"])
    ref = O.time_grid(args["num_steps"], args["n"], args["do_shift"], args["tsf"], args["strength"])
    assert torch.equal(mine, ref)


def test_transport_surface():
    from visualcloze_amd.transport import Sampler, create_transport
    s = Sampler(create_transport("Linear", "velocity", do_shift=True))
    with pytest.raises(NotImplementedError):
        s.sample_ode(sampling_method="dopri5")
    with pytest.raises(NotImplementedError):
        create_transport("VP", "noise")
    with pytest.raises(AssertionError):
        s.sample_ode(sampling_method="euler", strength=1.0)
    fn = s.sample_ode(sampling_method="euler", num_steps=4, atol=1e-6, rtol=1e-3, reverse=False, do_shift=True,
                      time_shifting_factor=1)
    # foreign callable path (CPU-capable): 3 evaluations at 1 - t, cond concatenated, kwargs not mutated
    seen = []

    def model(x, timesteps, **kw):
        seen.append((float(timesteps[0]), x.shape[-1]))
        return torch.ones_like(x[..., :2])
    kw = dict(cond=torch.zeros(1, 1152, 3), foo=1)
    out = fn(torch.zeros(1, 1152, 2), model, kw)
    assert out.shape == (1, 1, 1152, 2) and len(seen) == 3 and "cond" in kw
    # first evaluation at 1 - t[0]; the reference's own shifted grid starts at 6e-8, not exactly 0 (golden vectors)
    assert abs(seen[0][0] - 1.0) < 1e-6 and seen[0][1] == 5
    assert torch.allclose(out, torch.full_like(out, -1.0), atol=1e-6)     # integral of -1 over [0,1]


def test_model_times_follow_the_state_dtype(golden):
    """B2: torchdiffeq hands the drift t.to(y.dtype); with a bf16 state Flux sees 1 - bf16(t_i) (pinned by the
    reference's own bf16 run, tests/golden traj_bf16_model_t), with an f32 state the f32 grid."""
    import numpy as np
    from visualcloze_amd.transport import model_times, solver_time_grid
    t = solver_time_grid(5, 24, 0, 1, True, 1)
    got = model_times(t, torch.zeros(1, 24, 64, dtype=torch.bfloat16))
    assert got.dtype == torch.float32 and np.array_equal(got.double().numpy(), golden["traj_bf16_model_t"])
    assert torch.equal(model_times(t, torch.zeros(1, 24, 64)), 1 - t[:-1])
    # foreign-callable path: same rounding
    from visualcloze_amd.transport import Sampler, create_transport
    fn = Sampler(create_transport()).sample_ode(sampling_method="euler", num_steps=5, do_shift=True, time_shifting_factor=1)
    seen = []
    fn(torch.zeros(1, 24, 2, dtype=torch.bfloat16), lambda x, timesteps, **kw: (seen.append(float(timesteps[0])), x * 0)[1], {})
    assert np.array_equal(np.array(seen), golden["traj_bf16_model_t"])


def test_mask_layout_valid_first_permutation():
    """MaskLayout: any (txt_mask, img_mask) -> per-stream stable valid-first row order + (kv_len, one masked gap); right-
    padded masks are the identity; img_rows_back undoes img_rows."""
    import torch
    from visualcloze_amd.model import MaskLayout
    B, T, N = 3, 6, 8
    tm = torch.ones(B, T, dtype=torch.int32)
    im = torch.ones(B, N, dtype=torch.int32)
    tm[0, 4:] = 0                      # right-padded
    im[0, 5:] = 0
    tm[1, [0, 3]] = 0                  # holes
    im[1, [2, 7]] = 0
    lay = MaskLayout(tm, im, B, T, N)
    assert lay.perm_t[0].tolist() == list(range(T)) and lay.perm_i[0].tolist() == list(range(N))
    assert lay.perm_t[1].tolist() == [1, 2, 4, 5, 0, 3] and lay.perm_i[1].tolist() == [0, 1, 3, 4, 5, 6, 2, 7]
    assert lay.kv_len(slice(0, 3)) == [T + 5, T + 6, T + 8]
    assert lay.kv_gap(slice(0, 3)) == [(4, T), (4, T), (0, 0)] and lay.kv_gap(slice(2, 3)) is None
    x = torch.arange(B * N * 2, dtype=torch.float32).reshape(B, N, 2)
    sl = slice(1, 3)
    px = lay.img_rows(x, sl)
    assert torch.equal(px[0, :, 0], x[1, lay.perm_i[1], 0]) and torch.equal(px[1], x[2])
    assert torch.equal(lay.img_rows_back(px, sl), x[sl])
    ids = torch.arange(B * T * 3).reshape(B, T, 3)
    assert torch.equal(lay.txt_rows(ids, slice(1, 2))[0], ids[1, lay.perm_t[1]])
    # prefix masks and no masks: nothing is moved
    tm2, im2 = torch.ones(2, T), torch.ones(2, N)
    im2[1, -3:] = 0
    lay2 = MaskLayout(tm2, im2, 2, T, N)
    assert lay2.perm_t is None and lay2.perm_i is None and lay2.kv_gap(slice(0, 2)) is None
    assert lay2.kv_len(slice(0, 2)) == [T + N, T + N - 3]
    lay3 = MaskLayout(None, None, 2, T, N)
    assert lay3.kv_len(slice(0, 2)) == [T + N] * 2 and lay3.kv_gap(slice(0, 2)) is None


def test_upsampling_host_glue():
    """visualcloze.py:165-179 (target size rule) and :437-439 (to_pil_image truncates to 8 bits)."""
    import torch
    from visualcloze_amd.pipeline import to_uint8_image, upsampling_size
    assert upsampling_size(None) == (1024, 1024)
    assert upsampling_size((2048, 1024)) == (1440, 720)          # area-limited at the aspect ratio, then // 16 * 16
    assert upsampling_size((500, 300)) == (496, 288)
    assert upsampling_size((1024, 1024)) == (1024, 1024)
    im = to_uint8_image(torch.tensor([0.0, 0.999, 1.0]).reshape(3, 1, 1).expand(3, 2, 4))
    assert im.size == (4, 2) and im.mode == "RGB" and im.getpixel((3, 1)) == (0, 254, 255)


def test_gemm_launch_plans_for_the_flux_shapes():
    """vc_gemm_plan = the tile choice and the optional row cut of vc_gemm, without a launch (no GPU needed): pins the cost
    model on the shapes of every BASELINE geometry.  256x192 + loader waves (tile 4, form 2) wherever it fills whole rounds
    (cfg 2: 256 / 768 / 1024 tiles = 1 / 3 / 4 rounds), 256x128 for the short cfg 1, and ONE cut: the N = 3072 launches at
    L = 6656 (416 tiles = 1.6 rounds -> 256 tiles + 240 narrower ones)."""
    import ctypes as C
    from visualcloze_amd import hip
    L = hip.lib()

    def plan(Ms, N, K, epi=0, tile_cfg=0, sk=False, n=6):
        a = hip.GemmArgs()
        a.nprob, a.epi = len(Ms), epi
        if sk:                                                  # a split-K scratch is on offer (never dereferenced here)
            a.splitk_ws, a.splitk_ws_bytes = 0x1000, hip.GEMM_SPLITK_WS_BYTES
        for i, M in enumerate(Ms):
            p = a.p[i]
            p.A = p.W = p.C = p.res = p.gate = 0x1000          # never dereferenced by the planner
            p.lda, p.ldw, p.ldc, p.ldres = K, K, N, N
            p.M, p.N, p.K, p.rows_per_batch = M, N, K, M
        out = (C.c_int32 * 8)()
        rc = L.vc_gemm_plan(C.byref(a), tile_cfg, out)
        assert rc == 0, L.vc_last_error()
        return list(out)[:n]
    T = 512
    # cfg 2 (N_img = 3456): exact rounds of the 256x192 loader-wave tile, never cut
    assert plan([3456, T], 3072, 3072, 2) == [0, 4, 2, 0, 0, 256]
    assert plan([3456, T], 9216, 3072) == [0, 4, 2, 0, 0, 768]
    assert plan([3968], 12288, 3072, 1) == [0, 4, 2, 0, 0, 1024]
    assert plan([3968], 3072, 15360, 2) == [0, 4, 2, 0, 0, 256]
    # cfg 1 (N_img = 1152): the 256x128 sibling for N = 3072 (168 tiles instead of 112)
    assert plan([1152, T], 3072, 3072, 2)[:3] == [0, 2, 2]
    # cfg 3 (N_img = 6144): N = 3072 is cut at 4096 rows -> 256 tiles of 256x192 + (8 + 2) x 24 = 240 tiles of 256x128
    assert plan([6144, T], 3072, 12288, 2) == [4096, 4, 2, 2, 2, 496]
    assert plan([6656], 3072, 15360, 2) == [4096, 4, 2, 2, 2, 496]
    assert plan([6144, T], 3072, 12288, 2, hip.GEMM_NO_SPLIT) == [0, 4, 2, 0, 0, 416]
    assert plan([6144, T], 9216, 3072) == [0, 4, 2, 0, 0, 1248]          # 4.9 rounds: nothing to gain
    # cfg 5 (N_img = 6912) and the SDEdit stage (4096): cuts the model rates below 10 %, or with a half-empty remainder, are not taken
    assert plan([6912, T], 12288, 3072, 1)[0] == 0 and plan([7424], 3072, 15360, 2)[0] == 0
    assert plan([4096, T], 9216, 3072)[0] == 0 and plan([4608], 3072, 15360, 2)[:3] == [0, 2, 2]
    # a fixed tile is taken literally; a forced cut is obeyed
    assert plan([3968], 3072, 3072, 0, 1)[:3] == [0, 1, 0] and plan([3968], 3072, 3072, 0, 36)[:3] == [0, 4, 2]
    assert plan([3968], 3072, 3072, 0, 3 << 8)[0] == 768
    # SPLIT-K REMAINDER (a scratch on offer): the SDEdit stage's deep-K N = 3072 launches are 288 tiles = one round of 256 + 32
    # tiles cut 8 ways along K (256 slices: one short round) instead of 432 narrower tiles; cfg 1's are 112 tiles x 2 slices;
    # never at K = 3072 (partial traffic outweighs 1 / S of a short tile), never where the tiles are whole rounds (cfg 2),
    # never for the qkv epilogue; two samples per GPU at the SDEdit stage (M = 9216: 576 tiles) -> 64 tiles x 4
    assert plan([4608], 3072, 15360, 2, sk=True, n=8) == [0, 4, 2, 0, 0, 288, 8, 32]
    assert plan([4096, T], 3072, 12288, 2, sk=True, n=8) == [0, 4, 2, 0, 0, 288, 8, 32]
    assert plan([4096, T], 3072, 3072, 2, sk=True, n=8)[6:] == [0, 0]
    assert plan([1664], 3072, 15360, 2, sk=True, n=8) == [0, 4, 2, 0, 0, 112, 2, 112]
    assert plan([1152, T], 3072, 12288, 2, sk=True, n=8) == [0, 4, 2, 0, 0, 112, 2, 112]
    assert plan([9216], 3072, 15360, 2, sk=True, n=8) == [0, 4, 2, 0, 0, 576, 4, 64]
    assert plan([3968], 3072, 15360, 2, sk=True, n=8) == [0, 4, 2, 0, 0, 256, 0, 0]
    assert plan([4608], 9216, 3072, 4, sk=True, n=8)[6:] == [0, 0]
    assert plan([4608], 3072, 15360, 2, hip.GEMM_NO_SPLITK, sk=True, n=8)[:3] + plan([4608], 3072, 15360, 2, hip.GEMM_NO_SPLITK, sk=True, n=8)[6:] == [0, 2, 2, 0, 0]
    assert plan([4608], 3072, 15360, 2, n=8)[6:] == [0, 0]                                   # no scratch, no split
    assert plan([777], 1024, 512, 1, hip.GEMM_SPLITK(3), sk=True, n=8) == [0, 4, 2, 0, 0, 24, 3, 24]     # forced (tests)
    # STREAM form of the remainder (more than half a round, where no uniform S fits): cfg 3's deep-K N = 3072 launches, 416 tiles =
    # 256 + 160 whose K-iterations are dealt out to 256 work items (0.625 tile each) instead of the row cut; NOT cfg 5's (464 =
    # 256 + 208: a second round at 81 % fill measures faster), not at K = 3072, not without a scratch; the 5x5 grid's (944 = 3 x 256
    # + 176) takes it too
    assert plan([6144, T], 3072, 12288, 2, sk=True, n=8) == [0, 4, 2, 0, 0, 416, -256, 160]
    assert plan([6656], 3072, 15360, 2, sk=True, n=8) == [0, 4, 2, 0, 0, 416, -256, 160]
    assert plan([6144, T], 3072, 3072, 2, sk=True, n=8) == [4096, 4, 2, 2, 2, 496, 0, 0]
    assert plan([7424], 3072, 15360, 2, sk=True, n=8) == [0, 4, 2, 0, 0, 464, 0, 0]
    assert plan([7424], 3072, 15360, 2, hip.GEMM_PREFER_STREAMK, sk=True, n=8) == [0, 4, 2, 0, 0, 464, -256, 208]
    assert plan([14912], 3072, 15360, 2, sk=True, n=8) == [0, 4, 2, 0, 0, 944, -256, 176]
    assert plan([6656], 3072, 15360, 2, hip.GEMM_NO_SPLITK, sk=True, n=8) == [4096, 4, 2, 2, 2, 496, 0, 0]
    assert plan([300], 264, 4096, 1, hip.GEMM_STREAMK, sk=True, n=8) == [0, 4, 2, 0, 0, 4, -21, 4]       # forced (tests): ~12 iterations per item
    # argument errors come back as codes, with a message
    a = hip.GemmArgs(); a.nprob = 1; a.p[0].M, a.p[0].N, a.p[0].K = 8, 8, 60
    assert L.vc_gemm_plan(C.byref(a), 0, (C.c_int32 * 8)()) == -1 and b"multiple of 64" in L.vc_last_error()


def test_procedural_torch_equals_numpy():
    """tests/procedural.py: the torch evaluation of the closed-form weights (what the GPU tests use at full width) is
    bit-identical to the numpy one (what the golden generators use), chunk boundary included."""
    import torch
    from tests.procedural import procedural_param, ptensor, ptensor_torch
    for key, shape in (("double_blocks.0.img_attn.qkv.weight", (96, 64)), ("x.norm.scale", (128,)), ("a.bias", (300,)),
                       ("single_blocks.0.linear1.lora_B.weight", (70, 33)), ("m.img_mod.lin.weight", (50, 20))):
        assert torch.equal(procedural_param(key, shape), procedural_param(key, shape, device="cpu"))
    n = (1 << 24) + 5
    assert torch.equal(ptensor((n,), 77), ptensor_torch((n,), 77))
    assert torch.equal(ptensor((33, 7), 5, q=6).to(torch.bfloat16), ptensor_torch((33, 7), 5, q=6, dtype=torch.bfloat16))


def test_compat_install_aliases_the_reference_import_names(tmp_path):
    """visualcloze_amd.compat.install(): the reference's own `from models.model import Flux, FluxLoraWrapper, FluxParams`
    (models/util.py:11) and `from transport import Sampler, create_transport` (visualcloze.py:12) resolve to the MI355X
    implementations with no source edit.  Exercised on a stand-in tree with the reference's import lines and its
    `load_flow_model` construction (models/util.py:384-404), in a child interpreter (sys.modules is process state), through
    both entry points: install() in code and `python -m visualcloze_amd.compat script.py`."""
    import subprocess
    import sys
    import textwrap
    (tmp_path / "models").mkdir()
    (tmp_path / "models" / "__init__.py").write_text("")
    (tmp_path / "models" / "util.py").write_text(textwrap.dedent("""
        import torch
        from models.model import Flux, FluxLoraWrapper, FluxParams
        PARAMS = FluxParams(in_channels=384, out_channels=64, vec_in_dim=32, context_in_dim=64, hidden_size=256, mlp_ratio=4.0, num_heads=2,
                            depth=1, depth_single_blocks=1, axes_dim=[16, 56, 56], theta=10_000, qkv_bias=True, guidance_embed=True)
        def load_flow_model(name, device="cpu", lora_rank=128, lora_scale=1.0):
            return FluxLoraWrapper(params=PARAMS, lora_rank=lora_rank, lora_scale=lora_scale).to(torch.bfloat16)
    """))
    (tmp_path / "pipeline_like.py").write_text(textwrap.dedent("""
        from models.util import load_flow_model
        from transport import Sampler, create_transport
        import visualcloze_amd.model as M, visualcloze_amd.transport as T
        m = load_flow_model("flux-dev-fill-lora", lora_rank=8)
        assert type(m) is M.FluxLoraWrapper and Sampler is T.Sampler and create_transport is T.create_transport
        fn = Sampler(create_transport("Linear", "velocity", do_shift=True)).sample_ode(sampling_method="euler", num_steps=4, atol=1e-6,
                                                                                     rtol=1e-3, reverse=False, do_shift=True, time_shifting_factor=1)
        assert callable(fn) and "double_blocks.0.img_attn.qkv.lora_A.weight" in m.state_dict()
        print("drop-in ok")
    """))
    env = dict(os.environ, PYTHONPATH=REPO)
    code = f"import sys; sys.path.insert(0, {str(tmp_path)!r}); import visualcloze_amd.compat as c; c.install(); import pipeline_like"
    for cmd in ([sys.executable, "-c", code], [sys.executable, "-m", "visualcloze_amd.compat", str(tmp_path / "pipeline_like.py")]):
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300, cwd=str(tmp_path))
        assert r.returncode == 0 and "drop-in ok" in r.stdout, r.stdout + r.stderr
    # too late: the name is taken by another module already
    code = (f"import sys, types; sys.modules['transport'] = types.ModuleType('transport'); import visualcloze_amd.compat as c\n"
            "try:\n    c.install()\nexcept RuntimeError as e:\n    print('refused:', e)")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert "refused:" in r.stdout and "already imported" in r.stdout, r.stdout + r.stderr


def test_board_sampler_reads_hwmon_nodes(tmp_path):
    """visualcloze_amd.board: power / cap / shader clock of the card with the requested PCI address, sampled by a thread while
    a `with` block runs (bench.py: the timed steps); a machine without the nodes reports "unavailable" instead of failing."""
    import time
    from visualcloze_amd.board import BoardSampler, find_hwmon
    root = tmp_path / "drm"
    for i, (pci, uw, hz) in enumerate((("0000:05:00.0", 250_000_000, 2_400_000_000), ("0000:c1:00.0", 1_398_000_000, 1_812_000_000))):
        dev = tmp_path / "devices" / pci
        hw = dev / "hwmon" / f"hwmon{i + 3}"
        hw.mkdir(parents=True)
        (hw / ("power1_average" if i == 0 else "power1_input")).write_text(f"{uw}\n")
        (hw / "power1_cap").write_text("1400000000\n")
        (hw / "freq1_input").write_text(f"{hz}\n")
        card = root / f"card{i}"
        card.mkdir(parents=True)
        (card / "device").symlink_to(dev, target_is_directory=True)
        (root / f"card{i}-DP-1").mkdir()
    assert find_hwmon("0000:C1:00.0", root=str(root))["pci"] == "0000:c1:00.0"
    assert find_hwmon(None, index=0, root=str(root))["pci"] == "0000:05:00.0"
    with BoardSampler("0000:c1:00.0", hz=200.0, root=str(root)) as b:
        time.sleep(0.05)
    r = b.summary()
    assert r["power_w_avg"] == 1398.0 and r["power_cap_w"] == 1400.0 and r["sclk_mhz_avg"] == 1812.0 and r["samples"] >= 3
    with BoardSampler(None, root=str(tmp_path / "nothing")) as b:
        pass
    assert b.summary()["source"] == "unavailable"


def test_qkv_head_permutation_layout():
    """hip.qkv_head_permutation (VcGemmProblem.kn_heads): a bijection of the 3 * 128 * H qkv rows whose every 192-row block is one whole
    query or key head followed by 64 V rows - the layout the qkv GEMM's epilogue relies on (one head to normalise per 192-column
    tile, half a value head to transpose), and its inverse is the `qkv_col` map of csrc/gemm.hip."""
    from visualcloze_amd import hip
    for H in (1, 2, 3, 24):
        D = 128 * H
        perm = hip.qkv_head_permutation(H).tolist()
        assert sorted(perm) == list(range(3 * D))
        for t in range(2 * H):
            blk = perm[192 * t:192 * t + 192]
            assert blk[:128] == list(range(128 * t, 128 * t + 128))              # q head t (t < H) or k head t - H: logical columns 128 t ...
            assert blk[128:] == list(range(2 * D + 64 * t, 2 * D + 64 * t + 64))  # V columns 64 t ... 64 t + 63
        # the device-side map permuted column -> logical column (gemm.hip: qkv_col)
        def qkv_col(n):
            t, j = divmod(n, 192)
            return 128 * t + j if j < 128 else 256 * H + 64 * t + j - 128
        assert [qkv_col(p) for p in range(3 * D)] == perm


def test_stream_remainder_partition_properties():
    """The stream form of the split-K remainder (VcGemmArgs.sk_stream; csrc/gemm.hip): n work items share I = rem * nk K-iterations,
    item p owning [p I / n, (p + 1) I / n).  Restated here as the device code computes it - the writer's (tile, k range, slot) per
    segment and the reducer's piece list per tile - and checked for what correctness needs: every iteration of every tile is
    covered exactly once, an item has at most two segments (n >= rem), the reducer enumerates exactly the slots the writers
    filled, in K order, the first piece starting at K = 0 (it carries the bias), and at most three pieces per tile where the
    launcher takes the form by itself (rem > n / 2)."""
    import random

    def writer(p, rem, nk, n):
        I = rem * nk
        it, end, seg, out = p * I // n, (p + 1) * I // n, 0, []
        while it < end:
            t = it // nk
            k0, k1 = it - t * nk, min(nk, end - t * nk)
            out.append((t, k0, k1, 2 * p + seg))
            it += k1 - k0
            seg += 1
        return out

    def reducer(r, rem, nk, n):
        I = rem * nk
        it0 = lambda q: q * I // n  # noqa: E731
        u0, u1 = r * nk, (r + 1) * nk
        q = u0 * n // I
        while q > 0 and it0(q) > u0:
            q -= 1
        while q + 1 < n and it0(q + 1) <= u0:
            q += 1
        slots = []
        while q < n and it0(q) < u1:
            b = it0(q)
            if b != it0(q + 1):
                slots.append(2 * q + (0 if b // nk == r else 1))
            q += 1
        return slots

    rng = random.Random(5)
    cases = [(160, 192, 256), (160, 240, 256), (208, 192, 256), (176, 240, 256), (4, 64, 21), (24, 32, 64), (33, 25, 68), (1, 4, 1), (240, 48, 256)]
    cases += [(rem, nk, n) for rem, nk, n in ((rng.randint(1, 256), rng.randint(1, 240), 0) for _ in range(200))]
    for rem, nk, n in cases:
        if n == 0:
            n = rng.randint(rem, max(rem, min(256, rem * nk)))
        segs = {}
        for p in range(n):
            w = writer(p, rem, nk, n)
            assert len(w) <= 2, (rem, nk, n, p, w)
            for t, k0, k1, slot in w:
                assert 0 <= t < rem and 0 <= k0 < k1 <= nk
                segs.setdefault(t, []).append((k0, k1, slot))
        for r in range(rem):
            pieces = sorted(segs[r])
            assert pieces[0][0] == 0 and pieces[-1][1] == nk                      # covered from K = 0 to the end ...
            assert all(a[1] == b[0] for a, b in zip(pieces, pieces[1:]))          # ... without gap or overlap
            assert reducer(r, rem, nk, n) == [s for _, _, s in pieces], (rem, nk, n, r)     # the reducer's list = the writers' slots, in K order
            if 2 * rem > n:
                assert len(pieces) <= 3


def test_bench_roofline_legs_from_a_handle_profile():
    """bench.py's roofline legs are arithmetic on vc_flux_profile records (ABI 10): with a synthetic profile of one evaluation's
    launch classes the GATE_RES leg sums its three shapes, the table keeps every class with the fraction of ITS bound, and an
    attention class of 57 launches of 4 L^2 D FLOPs at 170 us reads 0.455 of the 2.5 PFLOP/s peak."""
    import bench
    from visualcloze_amd import hip
    L, D, mlp = 3968, 3072, 12288
    mk = lambda kind, epi, n, k, launches, flops, nbytes, us: dict(kind=kind, epi=epi, n=n, k=k, launches=launches, flops=flops, bytes=nbytes,  # noqa: E731
                                                                 total_us=us * launches, min_us=us * 0.98, max_us=us * 1.05, evaluations=1)
    recs = [mk(hip.LAUNCH_LN_MODULATE, 0, 0, 0, 77, 0.0, 77 * 4.0 * L * D, 15.0),
            mk(hip.LAUNCH_GEMM, hip.EPI_QKV, 3 * D, D, 57, 57 * 2.0 * L * 3 * D * D, 0.0, 180.0),
            mk(hip.LAUNCH_ATTENTION, 28, 0, 0, 57, 57 * 4.0 * L * L * D, 0.0, 170.0),
            mk(hip.LAUNCH_GEMM, hip.EPI_GATE_RES, D, D, 19, 19 * 2.0 * L * D * D, 0.0, 66.0),
            mk(hip.LAUNCH_GEMM, hip.EPI_GATE_RES, D, mlp, 19, 19 * 2.0 * L * D * mlp, 0.0, 224.0),
            mk(hip.LAUNCH_GEMM, hip.EPI_GATE_RES, D, D + mlp, 38, 38 * 2.0 * L * D * (D + mlp), 0.0, 265.0)]
    rows = bench.launch_classes(recs)
    assert [r["launch"] for r in rows] == ["ln_modulate", "gemm QKV N=9216 K=3072", "attention variant 28", "gemm GATE_RES N=3072 K=3072",
                                          "gemm GATE_RES N=3072 K=12288", "gemm GATE_RES N=3072 K=15360"]
    assert rows[0]["per_eval"] == 77 and rows[0]["frac_hbm"] == pytest.approx(4.0 * L * D / 15e-6 / 8e12, rel=1e-3) and "frac_mfma" not in rows[0]
    assert rows[2]["frac_mfma"] == pytest.approx(4.0 * L * L * D / 170e-6 / 2.5e15, rel=1e-3) and 0.45 < rows[2]["frac_mfma"] < 0.46

    class Job:
        pass
    g = bench.roofline_gemm_handle(Job(), recs)
    fl = 19 * 2.0 * L * D * D + 19 * 2.0 * L * D * mlp + 38 * 2.0 * L * D * (D + mlp)
    us = 19 * 66.0 + 19 * 224.0 + 38 * 265.0
    assert g["launches_per_eval"] == 76 and g["flops_per_launch"] == pytest.approx(fl / 76)
    assert g["avg_launch_us"] == pytest.approx(us / 76, abs=0.01) and g["frac"] == pytest.approx(fl / (us * 1e-6) / 2.5e15, abs=1e-4)
    assert g["bound"] == "mfma" and g["unit"] == "TFLOP/s" and "vc_flux_profile" in g["timed"]
