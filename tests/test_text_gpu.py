"""Text encoders on the GPU: glue kernels against torch references, tiny T5 / CLIP against the golden
vectors of transformers itself and the bf16 oracle, XXL-width T5 layers and L-width CLIP against the oracle."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import text_oracle as TO
from tests.procedural import TINY_CLIP, TINY_T5, procedural_text_param, ptensor, tiny_ids

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
G = np.load(os.path.join(os.path.dirname(__file__), "golden", "text_golden.npz"))


@pytest.fixture(scope="module")
def hip():
    from visualcloze_amd import hip as h
    h.require_gpu()
    return h


def rel_l2(a, b):
    a, b = torch.as_tensor(a).double().cpu(), torch.as_tensor(b).double().cpu()
    return float((a - b).norm() / b.norm())


def bf(t):
    return t.to(torch.bfloat16).to(DEV)


def test_embedding_is_exact(hip):
    tab = bf(ptensor((50, 64), 1, q=6))
    ids = torch.tensor([3, 0, 49, 7, 7, 12], dtype=torch.int32, device=DEV)
    out = torch.empty(6, 64, dtype=torch.bfloat16, device=DEV)
    hip.embedding(ids, tab, out)
    assert torch.equal(out, tab[ids.long()])


@pytest.mark.parametrize("rows,D", [(5, 128), (64, 768), (9, 4096)])
def test_rmsnorm_and_layernorm(hip, rows, D):
    x = bf(ptensor((rows, D), 2, q=5) + 0.125)
    w, b = bf(ptensor((D,), 3, q=8, kmax=64, offset=1.0)), bf(ptensor((D,), 4, q=8, kmax=32))
    y = torch.empty_like(x)
    hip.rmsnorm(x, w, y, 1e-6)
    xf = x.float()
    ref = (w.float() * (xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + 1e-6)).to(torch.bfloat16).float()).to(torch.bfloat16).float()
    assert (y.float() - ref).abs().max().item() <= 2 ** -7 * ref.abs().max().item()
    hip.layernorm(x, w, b, y, 1e-5)
    ref = F.layer_norm(xf, (D,), w.float(), b.float(), 1e-5)
    assert (y.float() - ref).abs().max().item() <= 2 ** -7 * ref.abs().max().item()


def test_elementwise_kernels(hip):
    a, b = bf(ptensor((40, 64), 5, q=5)), bf(ptensor((40, 64), 6, q=5))
    y = torch.empty_like(a)
    hip.mul(a, b, y); assert torch.equal(y, a * b)
    hip.add(a, b, y); assert torch.equal(y, a + b)
    hip.quick_gelu(a, y)
    ref = a * torch.sigmoid(1.702 * a)                     # torch's own bf16 rounding sequence
    assert (y.float() - ref.float()).abs().max().item() <= 2 ** -7 * ref.float().abs().max().item()


def test_softmax_bias_and_causal(hip):
    x, bias = bf(ptensor((32, 16), 7, q=4)), bf(ptensor((32, 16), 8, q=5))
    x0 = x.clone()
    hip.softmax_rows(x, 1.0, bias=bias)
    ref = torch.softmax((x0 + bias).float(), dim=-1)
    assert (x.float() - ref).abs().max().item() <= 8e-3
    x = x0.clone()
    hip.softmax_rows(x, 0.125, causal_period=16)           # two stacked [16, 16] causal blocks
    s = (x0 * 0.125).float().view(2, 16, 16) + torch.full((16, 16), float("-inf"), device=DEV).triu(1)
    ref = torch.softmax(s, dim=-1).view(32, 16)
    assert (x.float() - ref).abs().max().item() <= 8e-3
    assert float(x.view(2, 16, 16)[:, 0, 1:].abs().sum()) == 0.0


def load(model, sd):
    model.load_state_dict(sd)
    return model.to(DEV).to(torch.bfloat16)


@pytest.mark.parametrize("name", ["t5_a", "t5_b"])
def test_tiny_t5_matches_transformers_golden_and_oracle(hip, name):
    from visualcloze_amd.text import T5Config, T5EncoderModel
    m = T5EncoderModel(T5Config(**TINY_T5))
    sd = {k: procedural_text_param(k, v.shape) for k, v in m.state_dict().items()}
    sd["encoder.embed_tokens.weight"] = sd["shared.weight"]
    m = load(m, sd)
    ids = torch.tensor(G[name + "_ids"])
    out = m(ids[None].to(DEV))[0].float().cpu()
    ref32 = torch.tensor(G[name + "_fp32"])
    o16 = TO.t5_encode(sd, ids, TINY_T5, "bf16")
    noise = rel_l2(o16, ref32)
    assert rel_l2(out, ref32) <= 3.0 * noise + 2e-3, (rel_l2(out, ref32), noise)
    assert rel_l2(out, o16) <= 2.0 * noise + 2e-3, (rel_l2(out, o16), noise)
    assert torch.equal(out, m(ids[None].to(DEV))[0].float().cpu())


@pytest.mark.parametrize("name", ["clip_a", "clip_b"])
def test_tiny_clip_matches_transformers_golden_and_oracle(hip, name):
    from visualcloze_amd.text import CLIPTextConfig, CLIPTextModel
    m = CLIPTextModel(CLIPTextConfig(**TINY_CLIP))
    sd = {k: procedural_text_param(k, v.shape) for k, v in m.state_dict().items()}
    m = load(m, sd)
    ids = torch.tensor(G[name + "_ids"])
    pooled, hs = m(ids[None].to(DEV))
    pooled, hs = pooled[0].float().cpu(), hs[0].float().cpu()
    ref_h, ref_p = torch.tensor(G[name + "_hidden_fp32"]), torch.tensor(G[name + "_pooled_fp32"])
    p16, h16 = TO.clip_text(sd, ids, TINY_CLIP, "bf16")
    noise = rel_l2(h16, ref_h)
    assert rel_l2(hs, ref_h) <= 3.0 * noise + 2e-3, (rel_l2(hs, ref_h), noise)
    assert rel_l2(pooled, ref_p) <= 3.0 * rel_l2(p16, ref_p) + 4e-3


def test_xxl_width_t5_layers_match_oracle(hip):
    """t5-v1_1-xxl geometry (d_model 4096, 64 heads x 64, d_ff 10240, 512 tokens), 2 layers, small vocabulary."""
    from visualcloze_amd.text import T5Config, T5EncoderModel
    cfg = dict(vocab_size=512, d_model=4096, d_kv=64, d_ff=10240, num_layers=2, num_heads=64,
               relative_attention_num_buckets=32, relative_attention_max_distance=128, layer_norm_epsilon=1e-6)
    m = T5EncoderModel(T5Config(**cfg))
    sd = {k: procedural_text_param(k, v.shape) for k, v in m.state_dict().items()}
    sd["encoder.embed_tokens.weight"] = sd["shared.weight"]
    m = load(m, sd)
    ids = tiny_ids(512, 512, seed=77)
    out = m(ids[None].to(DEV))[0].float().cpu()
    assert out.shape == (512, 4096) and torch.isfinite(out).all()
    o32 = TO.t5_encode(sd, ids, cfg, "fp32")
    o16 = TO.t5_encode(sd, ids, cfg, "bf16")
    noise = rel_l2(o16, o32)
    assert rel_l2(out, o32) <= 3.0 * noise + 2e-3, (rel_l2(out, o32), noise)


def test_clip_l_width_matches_oracle(hip):
    """clip-vit-large-patch14 text geometry (768 wide, 12 heads, 77 tokens), 3 layers, small vocabulary."""
    from visualcloze_amd.text import CLIPTextConfig, CLIPTextModel
    cfg = dict(vocab_size=512, hidden_size=768, intermediate_size=3072, num_hidden_layers=3, num_attention_heads=12,
               max_position_embeddings=77, layer_norm_eps=1e-5, eos_token_id=511)
    m = CLIPTextModel(CLIPTextConfig(**cfg))
    sd = {k: procedural_text_param(k, v.shape) for k, v in m.state_dict().items()}
    m = load(m, sd)
    ids = tiny_ids(77, 512, seed=78, eos=511, eos_at=20)
    pooled, hs = m(ids[None].to(DEV))
    p32, h32 = TO.clip_text(sd, ids, cfg, "fp32")
    p16, h16 = TO.clip_text(sd, ids, cfg, "bf16")
    noise = rel_l2(h16, h32)
    assert hs.shape == (1, 77, 768) and pooled.shape == (1, 768)
    assert rel_l2(hs[0].float().cpu(), h32) <= 3.0 * noise + 2e-3, (rel_l2(hs[0].float().cpu(), h32), noise)
    assert rel_l2(pooled[0].float().cpu(), p32) <= 3.0 * rel_l2(p16, p32) + 4e-3
