"""-m gpu: the HIP kernels and the engine's block-level launch sequences against the PER-OP vectors the reference
itself produced (tests/golden/tiny_golden.npz, written by tests/golden/make_golden.py importing /root/reference).

Unlike tests/test_ops_gpu.py (kernels vs a torch restatement of the same op), every expected value here came out
of the reference's own functions: `timestep_embedding`, `QKNorm`, `apply_rope` / `EmbedND`, `Modulation`,
`attention` (full and ragged mask), `LinearLora` (rank clipped), `DoubleStreamBlock`, `SingleStreamBlock`,
`LastLayer`.  Inputs are the procedural tensors of tests/procedural.py (exactly representable in bf16), the
reference ran in fp32, the kernels compute in bf16 with f32 accumulation:
    leaf ops      |err| <= 2e-2 * max|ref|  and rel-L2 <= 1e-2        (bf16 eps = 7.8e-3)
    block level   rel-L2 <= 2e-2                                       (a block chains ~10 bf16-rounded ops)
"""
import numpy as np
import pytest
import torch

from tests.procedural import TINY, ptensor, tiny_inputs

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
H, HD = TINY["num_heads"], 128


def rel_l2(a, b):
    a, b = torch.as_tensor(a).float().cpu(), torch.as_tensor(b).float().cpu()
    return ((a - b).norm() / b.norm()).item()


def check(got, ref, tol=2e-2, l2=1e-2):
    got, ref = torch.as_tensor(got).float().cpu(), torch.as_tensor(ref).float().cpu()
    assert got.shape == ref.shape and torch.isfinite(got).all()
    assert (got - ref).abs().max().item() <= tol * ref.abs().max().item()
    assert rel_l2(got, ref) <= l2


def bf(t):
    return torch.as_tensor(t).to(DEV, torch.bfloat16).contiguous()


@pytest.fixture(scope="module")
def hip():
    from visualcloze_amd import hip as h
    h.require_gpu()
    return h


@pytest.fixture(scope="module")
def model():
    from tests.helpers import tiny_model
    return tiny_model()


def pe_to_cos_sin(pe):
    """golden `pe` [1,1,L,64,2,2] = [[cos,-sin],[sin,cos]] (math.py:107) -> the kernels' [L,64,2] (cos, sin) table."""
    pe = torch.as_tensor(pe)[0, 0]
    return torch.stack([pe[..., 0, 0], pe[..., 1, 0]], dim=-1).contiguous()


def pe_apply(pe, x):
    """apply the reference's own 2x2 rotation matrices to x [B,H,L,128] (only an einsum; no trig of ours)."""
    pe = torch.as_tensor(pe).float()[0]                       # [1,L,64,2,2]
    xp = torch.as_tensor(x).float().reshape(*x.shape[:-1], 64, 1, 2)
    out = pe[..., 0] * xp[..., 0] + pe[..., 1] * xp[..., 1]   # math.py:115-116
    return out.reshape(*x.shape)


def _qkv_rows(q, k, v):
    """[H,L,128] x3 -> the "L (K H D)" rows the QKV GEMM writes."""
    L = q.shape[1]
    return torch.stack([q, k, v], 0).permute(2, 0, 1, 3).reshape(L, 3 * H * HD)


# ------------------------------------------------------------------------------------------------ leaf ops
def test_timestep_embedding_vs_reference(hip, golden):
    from visualcloze_amd.model import Flux  # noqa: F401  (freqs table lives in the engine; rebuilt here the same way)
    import math
    fr = torch.exp(-math.log(10000) * torch.arange(0, 128, dtype=torch.float32) / 128).to(DEV)
    out = torch.empty(3, 256, dtype=torch.bfloat16, device=DEV)
    hip.timestep_embedding(torch.tensor(golden["temb_t"]).to(DEV), fr, out)
    check(out, golden["temb"], 1e-2)
    out = torch.empty(1, 256, dtype=torch.bfloat16, device=DEV)
    hip.timestep_embedding(torch.tensor([30.0], device=DEV), fr, out)
    check(out, golden["temb_g30"], 1e-2)


def test_qknorm_and_rope_vs_reference(hip, golden, tiny_sd):
    """vc_qknorm_rope_vt = RoPE(QKNorm(q)), RoPE(QKNorm(k)), V^T.  Expected: the reference's QKNorm outputs
    (`qknorm_q/k`, scales of double_blocks.0.img_attn.norm) rotated by the reference's own `pe` matrices."""
    L = golden["pe_ids"].shape[1]
    q, k, v = (ptensor((1, 2, L, 128), s, q=6)[0] for s in (11, 12, 13))
    assert torch.equal(q, torch.tensor(golden["rope_q_in"])[0])
    qkv = bf(_qkv_rows(q, k, v))
    rope = pe_to_cos_sin(golden["pe"]).to(DEV)
    vt = torch.zeros(H, 128, 64, dtype=torch.bfloat16, device=DEV)
    qs = bf(tiny_sd["double_blocks.0.img_attn.norm.query_norm.scale"])
    ks = bf(tiny_sd["double_blocks.0.img_attn.norm.key_norm.scale"])
    hip.qknorm_rope_vt(qkv, qs, ks, rope, vt, L, H)
    torch.cuda.synchronize()
    got = qkv.float().cpu().reshape(L, 3, H, 128).permute(1, 2, 0, 3)          # [3,H,L,128]
    check(got[0], pe_apply(golden["pe"], golden["qknorm_q"])[0])
    check(got[1], pe_apply(golden["pe"], golden["qknorm_k"])[0])
    assert torch.equal(vt[:, :, :L].float().cpu(), v.permute(0, 2, 1))          # V^T: pure data movement
    # RoPE alone against `apply_rope`'s own outputs: RMSNorm with unit scale is x / rms(x), so the kernel's row r
    # must equal rope_q_out[r] / rms(q[r])
    qkv = bf(_qkv_rows(q, k, v))
    ones = torch.ones(128, dtype=torch.bfloat16, device=DEV)
    hip.qknorm_rope_vt(qkv, ones, ones, rope, vt, L, H)
    torch.cuda.synchronize()
    got = qkv.float().cpu().reshape(L, 3, H, 128).permute(1, 2, 0, 3)
    for i, (x, key) in enumerate(((q, "rope_q_out"), (k, "rope_k_out"))):
        rrms = torch.rsqrt((x.double() ** 2).mean(-1, keepdim=True) + 1e-6).float()
        check(got[i], torch.tensor(golden[key])[0] * rrms)


def test_modulation_vs_reference(hip, golden, model):
    """Modulation.forward (layers.py:120-126) of double_blocks.0.img_mod: silu + the stacked modulation GEMM rows."""
    m, _ = model
    eng = m.engine()
    off = eng.W.mod_off["double_blocks.0.img_mod.lin"]
    n = 6 * TINY["hidden_size"]
    h = hip.silu(bf(golden["mod_vec"]))
    out = hip.linear(h, eng.W.mod_w[off:off + n], eng.W.mod_b[off:off + n])
    torch.cuda.synchronize()
    check(out, golden["mod_out"][0])        # (shift1, scale1, gate1, shift2, scale2, gate2): chunk order included


@pytest.mark.parametrize("variant", [0, 1, 2, 3, 7, 8, 12])
@pytest.mark.parametrize("case", ["attn_full", "attn_ragged"])
def test_attention_vs_reference(hip, golden, variant, case):
    """models/math.py:63-99 `attention(q, k, v, pe, attn_mask)` = apply_rope + flash_attn_varlen_func + pad_input.
    q, k are rotated with the reference's own pe matrices on the host, the kernel does softmax(QK^T/sqrt(d))V, the
    key mask (as kv_len) and the zeroed padded-query rows."""
    L = golden["pe_ids"].shape[1]
    q2, k2, v2 = (ptensor((2, 2, L, 128), s, q=6) for s in (21, 22, 23))
    qr, kr = pe_apply(golden["pe"], q2), pe_apply(golden["pe"], k2)
    qkv = bf(torch.cat([_qkv_rows(qr[b], kr[b], v2[b]) for b in range(2)]))     # [2L, 3*H*128], sample-major
    Lp = 64
    vt = torch.zeros(2, H, 128, Lp, dtype=torch.bfloat16, device=DEV)
    vt[..., :L] = bf(v2.permute(0, 1, 3, 2))
    out = torch.full((2 * L, H * 128), float("nan"), dtype=torch.bfloat16, device=DEV)
    kvl = None
    if case == "attn_ragged":
        mask = golden["attn_ragged_mask"]
        kvl = torch.tensor(mask.sum(1), dtype=torch.int32, device=DEV)
        assert all(mask[b, :int(mask[b].sum())].all() for b in range(2))        # prefix mask
    hip.attention(qkv, vt, out, L, H, kv_len=kvl, variant=variant, B=2)
    torch.cuda.synchronize()
    ref = torch.tensor(golden[case])                                           # [2, L, H*128]
    check(out.reshape(2, L, H * 128), ref)
    if kvl is not None:
        n1 = int(kvl[1])
        assert float(out.reshape(2, L, -1)[1, n1:].float().abs().sum()) == 0.0   # pad_input rows (math.py:96)
        assert float(ref[1, n1:].abs().sum()) == 0.0


@pytest.mark.parametrize("variant", [0, 3, 8, 12])
@pytest.mark.parametrize("split", [16, 24])
def test_attention_general_mask_vs_reference(hip, golden, variant, split):
    """A mask with holes anywhere (math.py:9-60 gathers arbitrary masks): the sequence is cut into two streams at
    `split`, MaskLayout moves each stream's valid rows first, the kernel masks [kv_len, L) and the gap (n_txt, split),
    the rows are scattered back and must equal the reference's `attention(..., attn_mask=general)`."""
    from visualcloze_amd.model import MaskLayout
    L = golden["pe_ids"].shape[1]
    mask = torch.tensor(golden["attn_general_mask"])
    lay = MaskLayout(mask[:, :split], mask[:, split:], 2, split, L - split)
    assert lay.perm_t is not None and lay.perm_i is not None
    sl = slice(0, 2)
    q2, k2, v2 = (ptensor((2, 2, L, 128), s, q=6) for s in (21, 22, 23))
    qr, kr = pe_apply(golden["pe"], q2), pe_apply(golden["pe"], k2)
    rows = torch.stack([_qkv_rows(qr[b], kr[b], v2[b]) for b in range(2)])      # [2, L, 3*H*128], caller order
    rows = torch.cat((lay.txt_rows(rows[:, :split], sl), lay.img_rows(rows[:, split:], sl)), 1)
    qkv = bf(rows.reshape(2 * L, -1))
    Lp = 64
    vt = torch.zeros(2, H, 128, Lp, dtype=torch.bfloat16, device=DEV)
    vt[..., :L] = qkv.reshape(2, L, 3, H, 128)[:, :, 2].permute(0, 2, 3, 1)
    out = torch.full((2 * L, H * 128), float("nan"), dtype=torch.bfloat16, device=DEV)
    kvl = torch.tensor(lay.kv_len(sl), dtype=torch.int32, device=DEV)
    gap = torch.tensor(lay.kv_gap(sl), dtype=torch.int32, device=DEV)
    hip.attention(qkv, vt, out, L, H, kv_len=kvl, variant=variant, B=2, kv_gap=gap)
    torch.cuda.synchronize()
    o = out.reshape(2, L, H * 128).float().cpu()
    inv_t = torch.argsort(lay.perm_t, dim=1)
    back = torch.cat((torch.gather(o[:, :split], 1, inv_t[..., None].expand(2, split, H * 128)),
                      lay.img_rows_back(o[:, split:], sl)), 1)
    ref = torch.tensor(golden["attn_general"])
    check(back, ref)
    assert float(back[mask == 0].abs().sum()) == 0.0 and float(ref[mask == 0].abs().sum()) == 0.0


def test_rank_clipped_lora_merge_vs_reference(hip, golden):
    """LinearLora (lora.py:34-98) with rank clipped to min(in, out) = 4 and scale 0.5: the product executes the merged
    weight; in = 12 and out = 4 are zero-padded to the GEMM's K % 64 / N % 8 granularity."""
    from tests.procedural import procedural_param
    from visualcloze_amd.model import Flux, Linear
    lin = Linear(12, 4, bias=True)
    lin.add_lora(8, 0.5)
    assert lin.rank == 4
    keys = [str(k) for k in golden["lora_clip_keys"]]
    shapes = [tuple(int(x) for x in str(s).split(",")) for s in golden["lora_clip_shapes"]]
    sd = {k: procedural_param("lltest." + k, s) for k, s in zip(keys, shapes)}
    assert set(sd) == set(lin.state_dict())
    lin.load_state_dict(sd)
    w, b = Flux.merged_linear(lin.to(DEV))
    wp = torch.zeros(8, 64, dtype=torch.bfloat16, device=DEV)
    bp = torch.zeros(8, dtype=torch.bfloat16, device=DEV)
    xp = torch.zeros(3, 64, dtype=torch.bfloat16, device=DEV)
    wp[:4, :12], bp[:4], xp[:, :12] = w, b, bf(golden["lora_clip_in"])
    out = hip.linear(xp, wp, bp)
    torch.cuda.synchronize()
    check(out[:, :4], golden["lora_clip_out"])


# ------------------------------------------------------------------------------------------------ block level
def _block_setup(model, golden):
    """Workspace of the tiny geometry with the reference's block inputs loaded: XI / XT = blk_img_in / blk_txt_in,
    MOD = every modulation Linear applied to silu(mod_vec) (the blocks were called with vec = mod_vec directly),
    ROPE from the same ids."""
    from visualcloze_amd import hip
    m, _ = model
    eng = m.engine()
    inp = tiny_inputs(B=1)
    T, N = inp["txt"].shape[1], inp["x"].shape[1]
    ws = eng.workspace(T, N, 1, 1)
    eng.prepare_sample(ws, bf(inp["txt"]), bf(inp["y"]), inp["guidance"].to(DEV), False, inp["img_ids"], inp["txt_ids"],
                       torch.tensor([0.5]), [T + N])
    h = hip.silu(bf(golden["mod_vec"]))
    hip.gemm(hip.make_problem(h, eng.W.mod_w, eng.W.mod_b, ws.MOD))
    rope = pe_to_cos_sin(golden["pe"])
    torch.cuda.synchronize()
    assert torch.allclose(ws.ROPE[0].cpu(), rope, atol=1e-6)      # host f64 table == the reference's EmbedND
    ws.XI.copy_(bf(golden["blk_img_in"][0]))
    ws.XT.copy_(bf(golden["blk_txt_in"][0]))
    return eng, ws, eng._ctx(ws, None, None), T, N


def test_double_block_vs_reference(model, golden):
    eng, ws, c, T, N = _block_setup(model, golden)
    eng.double_block(c, 0)
    torch.cuda.synchronize()
    assert rel_l2(ws.XI, golden["double0_img"][0]) < 2e-2
    assert rel_l2(ws.XT, golden["double0_txt"][0]) < 2e-2


def test_single_block_vs_reference(model, golden):
    eng, ws, c, T, N = _block_setup(model, golden)
    eng.join_streams(c)                          # cat((txt, img), 1)
    eng.single_block(c, 0)
    torch.cuda.synchronize()
    assert rel_l2(ws.X, golden["single0"][0]) < 2e-2


def test_last_layer_vs_reference(model, golden):
    eng, ws, c, T, N = _block_setup(model, golden)
    eng.join_streams(c)
    eng.last_layer(c)
    torch.cuda.synchronize()
    assert rel_l2(ws.V, golden["last"][0]) < 2e-2
