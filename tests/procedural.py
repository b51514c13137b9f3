"""Procedural (closed-form, RNG-free) tensors shared by the golden-vector generator and the tests.

Every value is k * 2^-q with integer |k| <= 127, i.e. exactly representable in bfloat16, so the fp32
reference, the oracle and the bf16 HIP path all see bit-identical weights and inputs on any machine."""
from __future__ import annotations

import math
import zlib

import numpy as np
import torch


def _hash(idx: np.ndarray, seed: int) -> np.ndarray:
    h = (idx * np.uint64(2654435761) + np.uint64(seed) * np.uint64(40503) + np.uint64(12345)) & np.uint64(0xFFFFFFFF)
    h ^= h >> np.uint64(15)
    h = (h * np.uint64(2246822519)) & np.uint64(0xFFFFFFFF)
    h ^= h >> np.uint64(13)
    h = (h * np.uint64(3266489917)) & np.uint64(0xFFFFFFFF)
    h ^= h >> np.uint64(16)
    return h


def ptensor(shape, seed: int, q: int = 7, kmax: int = 127, offset: float = 0.0) -> torch.Tensor:
    n = int(np.prod(shape))
    h = _hash(np.arange(n, dtype=np.uint64), seed)
    k = (h % np.uint64(2 * kmax + 1)).astype(np.int64) - kmax
    return torch.tensor(k.astype(np.float32) * np.float32(2.0 ** -q) + np.float32(offset)).reshape(*shape)


def ptensor_torch(shape, seed: int, q: int = 7, kmax: int = 127, offset: float = 0.0, device="cpu",
                  dtype=torch.float32) -> torch.Tensor:
    """`ptensor` evaluated with torch integer ops on `device` (the 0.6 B parameters of a full-width model take minutes in
    numpy on one core, a second on the GPU): the same hash in int64 with 32-bit masks - bit-identical values, pinned by
    tests/test_host_cpu.py::test_procedural_torch_equals_numpy."""
    n = int(np.prod(shape))
    M = 0xFFFFFFFF
    out = torch.empty(n, dtype=dtype, device=device)
    step = 1 << 24
    for s0 in range(0, n, step):
        h = torch.arange(s0, min(n, s0 + step), dtype=torch.int64, device=device)
        h = (h * 2654435761 + (seed * 40503 + 12345)) & M
        h = h ^ (h >> 15)
        h = (h * 2246822519) & M
        h = h ^ (h >> 13)
        h = (h * 3266489917) & M
        h = h ^ (h >> 16)
        k = (h % (2 * kmax + 1) - kmax).to(torch.float32)
        out[s0:s0 + h.numel()] = (k * (2.0 ** -q) + offset).to(dtype)
    return out.reshape(*shape)


def key_seed(key: str) -> int:
    return zlib.crc32(key.encode()) & 0x7FFFFFFF


def procedural_param(key: str, shape, device=None, dtype=torch.float32) -> torch.Tensor:
    """Weight for state-dict entry `key` (reference naming).  `device`: evaluate with torch ops there
    (`ptensor_torch`, same values) instead of numpy on the host."""
    seed = key_seed(key)
    shape = tuple(shape)
    gen = ptensor if device is None else (lambda *a, **k: ptensor_torch(*a, device=device, dtype=dtype, **k))
    if key.endswith("norm.scale"):                       # RMSNorm scales ~ 1
        return gen(shape, seed, q=9, kmax=64, offset=1.0)
    if key.endswith(".bias"):
        return gen(shape, seed, q=9, kmax=32)
    fan_in = shape[-1]
    q = int(round(math.log2(73.0 * math.sqrt(fan_in))))  # unit-variance outputs for unit-variance inputs
    if ".lora_A." in key:
        return gen(shape, seed, q=q, kmax=127)
    if ".lora_B." in key:
        return gen(shape, seed, q=q + 2, kmax=127)   # LoRA path live at ~25 % of the base magnitude
    if "_mod.lin" in key or "modulation.lin" in key or "adaLN_modulation" in key:
        return gen(shape, seed, q=q + 2, kmax=127)   # keep shift/scale/gate ~ 0.25
    return gen(shape, seed, q=q, kmax=127)


def procedural_state_dict(key_shapes) -> dict:
    return {k: procedural_param(k, s) for k, s in key_shapes}


TINY = dict(in_channels=384, out_channels=64, vec_in_dim=64, context_in_dim=128, hidden_size=256, mlp_ratio=4.0,
            num_heads=2, depth=2, depth_single_blocks=2, axes_dim=[16, 56, 56], theta=10_000, qkv_bias=True,
            guidance_embed=True)
TINY_RANK = 8


def tiny_inputs(B: int = 1, rows_hw=((4, 12), (4, 12)), T: int = 16, seed: int = 1):
    """Synthetic Flux.forward inputs for the tiny geometry: a 2-row latent grid (row index+1 ids)."""
    from oracle.flux_oracle import grid_img_ids
    ids = grid_img_ids(list(rows_hw))
    N = ids.shape[0]
    x = ptensor((B, N, 64), seed + 1, q=6)
    cond = torch.cat([ptensor((B, N, 64), seed + 2, q=6),
                      (ptensor((B, N, 256), seed + 3, q=0, kmax=1).abs() > 0.5).float()], dim=-1)
    return dict(
        x=x, cond=cond, img_ids=ids[None].repeat(B, 1, 1), txt=ptensor((B, T, TINY["context_in_dim"]), seed + 4, q=6),
        txt_ids=torch.zeros(B, T, 3), y=ptensor((B, TINY["vec_in_dim"]), seed + 5, q=6),
        txt_mask=torch.ones(B, T, dtype=torch.int32), img_mask=torch.ones(B, N, dtype=torch.int32),
        guidance=torch.full((B,), 30.0),
    )


# ---- VAE decoder: tiny geometry with every channel count a multiple of 64 ----
TINY_AE = dict(resolution=16, in_channels=3, ch=64, out_ch=3, ch_mult=[1, 2], num_res_blocks=1, z_channels=4,
               scale_factor=0.3611, shift_factor=0.1159)


def procedural_ae_param(key: str, shape) -> torch.Tensor:
    """Weight for an AutoEncoder state-dict entry (conv [O,I,k,k] / GroupNorm affine), exactly representable in bf16."""
    seed = key_seed("ae:" + key)
    shape = tuple(shape)
    if ".norm" in key and key.endswith(".weight"):       # GroupNorm gamma ~ 1
        return ptensor(shape, seed, q=8, kmax=64, offset=1.0)
    if key.endswith(".bias"):
        return ptensor(shape, seed, q=8, kmax=32)
    fan_in = int(np.prod(shape[1:]))
    q = int(round(math.log2(73.0 * math.sqrt(fan_in))))
    return ptensor(shape, seed, q=q, kmax=127)


def tiny_ae_latent(h: int, w: int, seed: int = 5) -> torch.Tensor:
    return ptensor((1, TINY_AE["z_channels"], h, w), seed, q=5, kmax=96)      # |z| <= 3


def tiny_ae_image(H: int, W: int, seed: int = 7) -> torch.Tensor:
    return ptensor((1, TINY_AE["in_channels"], H, W), seed, q=7, kmax=127)    # pixels in [-1, 1]


def tiny_ae_noise(h: int, w: int, seed: int = 8) -> torch.Tensor:
    return ptensor((1, TINY_AE["z_channels"], h, w), seed, q=5, kmax=80)


# ---- text encoders: tiny geometries, head_dim 64 like the real models ----
TINY_T5 = dict(vocab_size=128, d_model=128, d_kv=64, d_ff=256, num_layers=2, num_heads=2,
               relative_attention_num_buckets=32, relative_attention_max_distance=128, layer_norm_epsilon=1e-6)
TINY_CLIP = dict(vocab_size=128, hidden_size=128, intermediate_size=256, num_hidden_layers=2, num_attention_heads=2,
                 max_position_embeddings=24, layer_norm_eps=1e-5, eos_token_id=127)


def procedural_text_param(key: str, shape) -> torch.Tensor:
    seed = key_seed("txt:" + key)
    shape = tuple(shape)
    if "layer_norm" in key and key.endswith(".weight"):
        return ptensor(shape, seed, q=8, kmax=64, offset=1.0)
    if key.endswith(".bias"):
        return ptensor(shape, seed, q=8, kmax=32)
    if "relative_attention_bias" in key:
        return ptensor(shape, seed, q=5, kmax=64)            # |bias| <= 2
    if "embedding" in key or key.startswith("shared") or "embed_tokens" in key:
        return ptensor(shape, seed, q=6, kmax=96)            # |e| <= 1.5
    fan_in = shape[-1]
    q = int(round(math.log2(73.0 * math.sqrt(fan_in))))
    return ptensor(shape, seed, q=q, kmax=127)


def tiny_ids(L: int, vocab: int, seed: int, eos: int | None = None, eos_at: int | None = None) -> torch.Tensor:
    h = _hash(np.arange(L, dtype=np.uint64), seed)
    ids = (h % np.uint64(vocab - 1)).astype(np.int64)       # never the last id, which the CLIP case reserves for EOS
    if eos is not None:
        ids[eos_at] = eos
        ids[eos_at + 1:] = 0
    return torch.tensor(ids)
