"""CPU restatement of the T5 encoder and the CLIP text model — TEST INFRASTRUCTURE ONLY.

The reference reaches both through `HFEmbedder` (models/modules/conditioner.py:5-37): `T5EncoderModel(input_ids,
attention_mask=None).last_hidden_state` and `CLIPTextModel(input_ids).pooler_output`.  The arithmetic is transformers'
(third party, absent from /root/reference, version unpinned there): this file restates modeling_t5 (T5LayerNorm,
T5Attention with the bidirectional relative-position buckets, T5DenseGatedActDense with gelu_new) and modeling_clip
(CLIPTextEmbeddings, causal CLIPAttention, quick_gelu MLP, final LayerNorm, EOS pooling) from their published
definitions.  Pinned by tests/golden/text_golden.npz, produced by running the transformers build of the container on
tiny random-free configurations (tests/golden/make_text_golden.py).
Modes: fp32, or bf16 = every tensor the HF module materialises when it runs in bfloat16 is rounded.
"""
from __future__ import annotations

import math

import torch


def _r(x, bf16):
    return x.to(torch.bfloat16).to(torch.float32) if bf16 else x


def _lin(sd, key, x, bf16):
    y = x @ sd[key + ".weight"].float().t()
    if (key + ".bias") in sd:
        y = y + sd[key + ".bias"].float()
    return _r(y, bf16)


def t5_relative_buckets(L, num_buckets=32, max_distance=128):
    ctx = torch.arange(L)[:, None]
    mem = torch.arange(L)[None, :]
    rel = mem - ctx
    nb = num_buckets // 2
    buckets = (rel > 0).long() * nb
    rel = rel.abs()
    max_exact = nb // 2
    is_small = rel < max_exact
    large = max_exact + (torch.log(rel.float() / max_exact) / math.log(max_distance / max_exact) * (nb - max_exact)).long()
    large = torch.min(large, torch.full_like(large, nb - 1))
    return buckets + torch.where(is_small, rel, large)


def _t5_norm(sd, key, x, eps, bf16):
    var = x.pow(2).mean(-1, keepdim=True)
    return _r(sd[key + ".weight"].float() * _r(x * torch.rsqrt(var + eps), bf16), bf16)


def _gelu_new(x):
    return 0.5 * x * (1.0 + torch.tanh(math.sqrt(2.0 / math.pi) * (x + 0.044715 * torch.pow(x, 3.0))))


def t5_encode(sd: dict, ids: torch.Tensor, cfg: dict, mode: str = "fp32") -> torch.Tensor:
    """ids [L] -> last_hidden_state [L, d_model]"""
    bf16 = mode == "bf16"
    H, dh, eps = cfg["num_heads"], cfg["d_kv"], cfg.get("layer_norm_epsilon", 1e-6)
    L = ids.shape[0]
    x = sd["shared.weight"].float()[ids]
    b = t5_relative_buckets(L, cfg.get("relative_attention_num_buckets", 32), cfg.get("relative_attention_max_distance", 128))
    bias = sd["encoder.block.0.layer.0.SelfAttention.relative_attention_bias.weight"].float()[b].permute(2, 0, 1)   # [H, L, L]
    for i in range(cfg["num_layers"]):
        p = f"encoder.block.{i}.layer.0"
        n = _t5_norm(sd, p + ".layer_norm", x, eps, bf16)
        q, k, v = (_lin(sd, f"{p}.SelfAttention.{t}", n, bf16).view(L, H, dh).transpose(0, 1) for t in ("q", "k", "v"))
        s = _r(_r(q @ k.transpose(1, 2), bf16) + bias, bf16)                  # no 1/sqrt(d) in T5
        a = _r(torch.softmax(s, dim=-1), bf16)
        o = _r(a @ v, bf16).transpose(0, 1).reshape(L, H * dh)
        x = _r(x + _lin(sd, p + ".SelfAttention.o", o, bf16), bf16)
        p = f"encoder.block.{i}.layer.1"
        n = _t5_norm(sd, p + ".layer_norm", x, eps, bf16)
        g = _r(_gelu_new(_lin(sd, p + ".DenseReluDense.wi_0", n, bf16)), bf16)
        u = _lin(sd, p + ".DenseReluDense.wi_1", n, bf16)
        x = _r(x + _lin(sd, p + ".DenseReluDense.wo", _r(g * u, bf16), bf16), bf16)
    return _t5_norm(sd, "encoder.final_layer_norm", x, eps, bf16)


def _ln(sd, key, x, eps, bf16):
    return _r(torch.nn.functional.layer_norm(x, (x.shape[-1],), sd[key + ".weight"].float(), sd[key + ".bias"].float(), eps), bf16)


def clip_text(sd: dict, ids: torch.Tensor, cfg: dict, mode: str = "fp32"):
    """ids [L] -> (pooler_output [D], last_hidden_state [L, D]); sd keys carry the checkpoint's `text_model.` prefix."""
    bf16 = mode == "bf16"
    H, eps = cfg["num_attention_heads"], cfg.get("layer_norm_eps", 1e-5)
    L = ids.shape[0]
    P = "text_model."
    x = _r(sd[P + "embeddings.token_embedding.weight"].float()[ids] + sd[P + "embeddings.position_embedding.weight"].float()[:L], bf16)
    D = x.shape[-1]
    dh = D // H
    mask = torch.full((L, L), float("-inf")).triu(1)
    for i in range(cfg["num_hidden_layers"]):
        p = f"{P}encoder.layers.{i}"
        n = _ln(sd, p + ".layer_norm1", x, eps, bf16)
        q, k, v = (_lin(sd, f"{p}.self_attn.{t}_proj", n, bf16).view(L, H, dh).transpose(0, 1) for t in ("q", "k", "v"))
        s = _r(_r(q @ k.transpose(1, 2), bf16) * dh ** -0.5, bf16) + mask
        a = _r(torch.softmax(s, dim=-1), bf16)
        o = _r(a @ v, bf16).transpose(0, 1).reshape(L, D)
        x = _r(x + _lin(sd, p + ".self_attn.out_proj", o, bf16), bf16)
        n = _ln(sd, p + ".layer_norm2", x, eps, bf16)
        h1 = _lin(sd, p + ".mlp.fc1", n, bf16)
        h1 = _r(h1 * _r(torch.sigmoid(_r(1.702 * h1, bf16)), bf16), bf16)      # quick_gelu
        x = _r(x + _lin(sd, p + ".mlp.fc2", h1, bf16), bf16)
    hs = _ln(sd, P + "final_layer_norm", x, eps, bf16)
    eos = int((ids == cfg["eos_token_id"]).int().argmax())
    return hs[eos], hs
