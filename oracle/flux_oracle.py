"""CPU oracle for the VisualCloze denoising path — TEST INFRASTRUCTURE ONLY.

A functional, from-scratch restatement (plain PyTorch on CPU, no nn.Module, weights addressed by the
reference's state-dict keys) of the reference algorithm for the one hot path this repo accelerates:
`Flux.forward` driven by the fixed-grid Euler sampler.  Only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg may import it; the product path (visualcloze_amd/) never does.

Pinning: there are no tests or golden vectors in the reference.  This oracle is pinned
against outputs of the reference ITSELF, imported in the build container under three shims
(tests/golden/make_golden.py) and committed as fixtures under tests/golden/ — see
tests/test_oracle_golden.py.  Two third-party pieces are absent from /root/reference and are
restated from their published semantics: flash-attn's `flash_attn_varlen_func` (= softmax(QK^T/sqrt(d))V
per unpadded sequence; call site models/math.py:85-95) and torchdiffeq's fixed-grid `euler`
(y_{i+1} = y_i + (t_{i+1}-t_i) f(t_i, y_i); call site transport/integrators.py:119).  Parity against the
diffusers `VisualClozePipeline` is UNPINNED (its source is not in /root/reference).

Precision modes (`Prec`):
  fp32  exact reference semantics with every tensor in float32 (what `model.float()` computes).
  bf16  float32 tensors rounded to bfloat16 at exactly the points where the reference materialises a
        bf16 tensor under torch.autocast("cuda", bfloat16) (visualcloze.py:363): Linear outputs,
        bf16*bf16 products, residual adds, casts — LayerNorm / RMSNorm / RoPE / softmax internals stay f32.
        This is the tight comparator for the HIP kernels (they round at the same points).
LoRA (`lora`): "ref" = W x + b + s*(B(A x) + b_B) with the reference's three roundings (lora.py:92-98);
"merged" = (W + s*B A) x + (b + s*b_B) with the merged weight rounded once — what the product executes.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional

import torch

Tensor = torch.Tensor


@dataclass
class FluxGeometry:
    """models/model.py:18-32 (FluxParams) — same field names."""
    in_channels: int = 384
    out_channels: int = 64
    vec_in_dim: int = 768
    context_in_dim: int = 4096
    hidden_size: int = 3072
    mlp_ratio: float = 4.0
    num_heads: int = 24
    depth: int = 19
    depth_single_blocks: int = 38
    axes_dim: List[int] = field(default_factory=lambda: [16, 56, 56])
    theta: int = 10_000
    qkv_bias: bool = True
    guidance_embed: bool = True


class Prec:
    def __init__(self, mode: str = "fp32", lora: str = "ref"):
        assert mode in ("fp32", "bf16") and lora in ("ref", "merged")
        self.mode, self.lora = mode, lora

    def r(self, x: Tensor) -> Tensor:
        """materialise as the activation dtype"""
        return x.to(torch.bfloat16).float() if self.mode == "bf16" else x


# ------------------------------------------------------------------------------------------------
# leaf math
# ------------------------------------------------------------------------------------------------
def timestep_embedding(t: Tensor, dim: int = 256, max_period: int = 10000, time_factor: float = 1000.0,
                       t_is_bf16: bool = False) -> Tensor:
    """layers.py:28-49.  `t_is_bf16`: the guidance tensor is created in bf16 (visualcloze.py:413), so
    `time_factor * t` rounds to bf16 and the result is cast back `.to(t)`."""
    t = t.float()
    if t_is_bf16:
        t = (time_factor * t.to(torch.bfloat16)).float()  # bf16 product
    else:
        t = time_factor * t
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(0, half, dtype=torch.float32) / half)
    args = t[:, None] * freqs[None]
    emb = torch.cat([torch.cos(args), torch.sin(args)], dim=-1)
    if t_is_bf16:
        emb = emb.to(torch.bfloat16).float()
    return emb


def temb_freqs(dim: int = 256, max_period: int = 10000) -> Tensor:
    half = dim // 2
    return torch.exp(-math.log(max_period) * torch.arange(0, half, dtype=torch.float32) / half)


def rope_cos_sin(ids: Tensor, axes_dim: List[int], theta: int) -> Tensor:
    """EmbedND + rope (layers.py:11-25, math.py:102-109): float64 angles, returned as [..., L, 64, 2] =
    (cos, sin) per rotary pair in float32 (the reference's 2x2 is [[cos,-sin],[sin,cos]])."""
    outs = []
    for i, d in enumerate(axes_dim):
        assert d % 2 == 0
        scale = torch.arange(0, d, 2, dtype=torch.float64) / d
        omega = 1.0 / (theta ** scale)
        ang = ids[..., i].to(torch.float64)[..., None] * omega
        outs.append(torch.stack([torch.cos(ang), torch.sin(ang)], dim=-1))
    return torch.cat(outs, dim=-2).float()


def apply_rope(x: Tensor, cs: Tensor, P: Prec) -> Tensor:
    """math.py:112-117 on x [B,H,L,D]: o0 = cos*x0 - sin*x1, o1 = sin*x0 + cos*x1 (f32), cast back."""
    B, H, L, D = x.shape
    xp = x.float().reshape(B, H, L, D // 2, 2)
    cos, sin = cs[:, None, :, :, 0], cs[:, None, :, :, 1]
    o0 = cos * xp[..., 0] + (-sin) * xp[..., 1]
    o1 = sin * xp[..., 0] + cos * xp[..., 1]
    return P.r(torch.stack([o0, o1], dim=-1).reshape(B, H, L, D))


def rms_norm(x: Tensor, scale: Tensor, P: Prec) -> Tensor:
    """layers.py:63-72: (x*rsqrt(mean(x^2)+1e-6)).to(dtype) * scale."""
    xf = x.float()
    rrms = torch.rsqrt(torch.mean(xf ** 2, dim=-1, keepdim=True) + 1e-6)
    return P.r(P.r(xf * rrms) * scale.float())


def layer_norm(x: Tensor) -> Tensor:
    """nn.LayerNorm(elementwise_affine=False, eps=1e-6); f32 output under CUDA autocast."""
    return torch.nn.functional.layer_norm(x.float(), (x.shape[-1],), eps=1e-6)


def modulate(x: Tensor, shift: Tensor, scale: Tensor, P: Prec) -> Tensor:
    """(1 + scale) * LN(x) + shift (layers.py:164,191,234,257).  `1 + scale` is a bf16 tensor under
    autocast; the product/sum are f32 and only round when the next Linear casts its input."""
    return P.r(P.r(1 + scale) * layer_norm(x) + shift)


def gelu_tanh(x: Tensor) -> Tensor:
    return torch.nn.functional.gelu(x, approximate="tanh")


def sdpa(q: Tensor, k: Tensor, v: Tensor, P: Prec, kv_len=None) -> Tensor:
    """flash_attn_varlen_func semantics behind `_upad_input` / `pad_input` (math.py:9-60,85-96): softmax(q k^T * d^-0.5) v
    over the UNMASKED keys of each batch element, for its unmasked query rows; masked query rows come back as zeros.
    kv_len: None (everything attends), a list of prefix lengths, or a bool / int mask [B, L] (any pattern).
    q,k,v: [B,H,L,D] -> [B,L,H*D]."""
    B, H, L, D = q.shape
    out = torch.zeros(B, L, H * D)
    for b in range(B):
        if kv_len is None:
            idx = torch.arange(L)
        elif torch.is_tensor(kv_len) and kv_len.dim() == 2:
            idx = torch.nonzero(kv_len[b], as_tuple=False).flatten()
        else:
            idx = torch.arange(int(kv_len[b]))
        n = idx.numel()
        qq, kk, vv = q[b][:, idx].float(), k[b][:, idx].float(), v[b][:, idx].float()
        hc = max(1, min(H, (1 << 28) // max(1, n * n)))     # heads per chunk: score tensors stay <= 1 GiB (L = 7424 fits)
        for h0 in range(0, H, hc):
            hs = slice(h0, min(H, h0 + hc))
            s = (qq[hs] @ kk[hs].transpose(-1, -2)) * (D ** -0.5)
            if P.mode == "bf16":
                # flash-attn keeps the un-normalised P in bf16 for the PV matmul and divides by the f32 row sum last
                m = s.max(dim=-1, keepdim=True).values
                e = torch.exp(s - m)
                o = (P.r(e) @ vv[hs]) / e.sum(dim=-1, keepdim=True)
            else:
                o = torch.softmax(s, dim=-1) @ vv[hs]
            out[b, idx, h0 * D:(h0 + o.shape[0]) * D] = P.r(o.permute(1, 0, 2).reshape(n, o.shape[0] * D))
    return out


# ------------------------------------------------------------------------------------------------
# Linear (+LoRA)
# ------------------------------------------------------------------------------------------------
def lora_merged_weight(sd: Dict[str, Tensor], prefix: str, lora_scale: float, P: Prec):
    w = sd[prefix + ".weight"].float()
    b = sd.get(prefix + ".bias")
    b = None if b is None else b.float()
    if prefix + ".lora_A.weight" in sd:
        a, bb = sd[prefix + ".lora_A.weight"].float(), sd[prefix + ".lora_B.weight"].float()
        w = P.r(w + lora_scale * (bb @ a))
        lb = sd.get(prefix + ".lora_B.bias")
        if lb is not None:
            b = P.r((b if b is not None else 0) + lora_scale * lb.float())
    return w, b


def linear(sd: Dict[str, Tensor], prefix: str, x: Tensor, P: Prec, lora_scale: float = 1.0) -> Tensor:
    """nn.Linear / LinearLora.forward (lora.py:92-98)."""
    x = P.r(x)
    if P.lora == "merged":
        w, b = lora_merged_weight(sd, prefix, lora_scale, P)
        y = x @ w.t()
        return P.r(y + b if b is not None else y)
    w = sd[prefix + ".weight"].float()
    b = sd.get(prefix + ".bias")
    y = x @ w.t()
    y = P.r(y + b.float() if b is not None else y)
    if prefix + ".lora_A.weight" in sd:
        h = P.r(x @ sd[prefix + ".lora_A.weight"].float().t())
        u = h @ sd[prefix + ".lora_B.weight"].float().t()
        lb = sd.get(prefix + ".lora_B.bias")
        u = P.r(u + lb.float() if lb is not None else u)
        y = P.r(y + P.r(u * lora_scale))
    return y


def mlp_embedder(sd, prefix, x, P, ls=1.0) -> Tensor:
    """layers.py:52-60."""
    h = linear(sd, prefix + ".in_layer", x, P, ls)
    return linear(sd, prefix + ".out_layer", P.r(torch.nn.functional.silu(h)), P, ls)


def modulation(sd, prefix, vec, n, P, ls=1.0):
    """layers.py:113-126: Linear(silu(vec)) chunked into n parts of [B,1,D]."""
    out = linear(sd, prefix + ".lin", P.r(torch.nn.functional.silu(vec)), P, ls)
    return out[:, None, :].chunk(n, dim=-1)


# ------------------------------------------------------------------------------------------------
# blocks
# ------------------------------------------------------------------------------------------------
def _split_heads(qkv: Tensor, H: int):
    B, L, _ = qkv.shape
    return qkv.reshape(B, L, 3, H, -1).permute(2, 0, 3, 1, 4)  # "B L (K H D) -> K B H L D"


def _qk_attn(q, k, v, sd, norm_prefix, cs, P, kv_len):
    q = rms_norm(q, sd[norm_prefix + ".query_norm.scale"], P)
    k = rms_norm(k, sd[norm_prefix + ".key_norm.scale"], P)
    q, k = apply_rope(q, cs, P), apply_rope(k, cs, P)
    return sdpa(q, k, v, P, kv_len)


def double_block(sd, pfx, img, txt, vec, cs, G: FluxGeometry, P: Prec, kv_len=None, ls=1.0):
    """DoubleStreamBlock.forward, layers.py:158-196."""
    H = G.num_heads
    im = modulation(sd, pfx + ".img_mod", vec, 6, P, ls)   # shift1, scale1, gate1, shift2, scale2, gate2
    tm = modulation(sd, pfx + ".txt_mod", vec, 6, P, ls)
    iq, ik, iv = _split_heads(linear(sd, pfx + ".img_attn.qkv", modulate(img, im[0], im[1], P), P, ls), H)
    tq, tk, tv = _split_heads(linear(sd, pfx + ".txt_attn.qkv", modulate(txt, tm[0], tm[1], P), P, ls), H)
    iq, ik = rms_norm(iq, sd[pfx + ".img_attn.norm.query_norm.scale"], P), rms_norm(ik, sd[pfx + ".img_attn.norm.key_norm.scale"], P)
    tq, tk = rms_norm(tq, sd[pfx + ".txt_attn.norm.query_norm.scale"], P), rms_norm(tk, sd[pfx + ".txt_attn.norm.key_norm.scale"], P)
    q, k, v = torch.cat((tq, iq), 2), torch.cat((tk, ik), 2), torch.cat((tv, iv), 2)
    attn = sdpa(apply_rope(q, cs, P), apply_rope(k, cs, P), v, P, kv_len)
    T = txt.shape[1]
    ta, ia = attn[:, :T], attn[:, T:]
    img = P.r(img + P.r(im[2] * linear(sd, pfx + ".img_attn.proj", ia, P, ls)))
    h = P.r(gelu_tanh(linear(sd, pfx + ".img_mlp.0", modulate(img, im[3], im[4], P), P, ls)))
    img = P.r(img + P.r(im[5] * linear(sd, pfx + ".img_mlp.2", h, P, ls)))
    txt = P.r(txt + P.r(tm[2] * linear(sd, pfx + ".txt_attn.proj", ta, P, ls)))
    h = P.r(gelu_tanh(linear(sd, pfx + ".txt_mlp.0", modulate(txt, tm[3], tm[4], P), P, ls)))
    txt = P.r(txt + P.r(tm[5] * linear(sd, pfx + ".txt_mlp.2", h, P, ls)))
    return img, txt


def single_block(sd, pfx, x, vec, cs, G: FluxGeometry, P: Prec, kv_len=None, ls=1.0):
    """SingleStreamBlock.forward, layers.py:232-245."""
    H, D = G.num_heads, G.hidden_size
    shift, scale, gate = modulation(sd, pfx + ".modulation", vec, 3, P, ls)
    y = linear(sd, pfx + ".linear1", modulate(x, shift, scale, P), P, ls)
    qkv, mlp = y[..., : 3 * D], y[..., 3 * D:]
    q, k, v = _split_heads(qkv, H)
    attn = _qk_attn(q, k, v, sd, pfx + ".norm", cs, P, kv_len)
    out = linear(sd, pfx + ".linear2", torch.cat((attn, P.r(gelu_tanh(mlp))), 2), P, ls)
    return P.r(x + P.r(gate * out))


def last_layer(sd, x, vec, P: Prec, ls=1.0):
    """LastLayer.forward, layers.py:255-259."""
    m = linear(sd, "final_layer.adaLN_modulation.1", P.r(torch.nn.functional.silu(vec)), P, ls)
    shift, scale = m.chunk(2, dim=1)
    return linear(sd, "final_layer.linear", modulate(x, shift[:, None, :], scale[:, None, :], P), P, ls)


def compute_vec(sd, timesteps, guidance, y, G: FluxGeometry, P: Prec, ls=1.0, guidance_is_bf16=True):
    """model.py:102-107."""
    vec = mlp_embedder(sd, "time_in", timestep_embedding(timesteps, 256), P, ls)
    if G.guidance_embed:
        if guidance is None:
            raise ValueError("Didn't get guidance strength for guidance distilled model.")
        g = timestep_embedding(guidance, 256, t_is_bf16=(guidance_is_bf16 and P.mode == "bf16"))
        vec = P.r(vec + mlp_embedder(sd, "guidance_in", g, P, ls))
    return P.r(vec + mlp_embedder(sd, "vector_in", y, P, ls))


def flux_forward(sd: Dict[str, Tensor], G: FluxGeometry, img, img_ids, txt, txt_ids, timesteps, y,
                 txt_mask=None, img_mask=None, guidance=None, P: Prec = Prec(), lora_scale: float = 1.0,
                 taps: Optional[dict] = None) -> Tensor:
    """Flux.forward, models/model.py:85-124; the joint mask is cat(txt_mask, img_mask), any pattern (the reference's
    callers only produce right-padded ones, models/sampling.py:41-46,68-70,98)."""
    if img.ndim != 3 or txt.ndim != 3:
        raise ValueError("Input img and txt tensors must have 3 dimensions.")
    ls = lora_scale
    B, N, _ = img.shape
    T = txt.shape[1]
    kv_len = None
    if img_mask is not None and txt_mask is not None:
        joint = torch.cat((txt_mask, img_mask), 1)
        kv_len = None if bool(joint.all()) else joint.bool()
    img = linear(sd, "img_in", img.float(), P, ls)
    vec = compute_vec(sd, timesteps.float(), None if guidance is None else guidance.float(), y.float(), G, P, ls)
    txt = linear(sd, "txt_in", txt.float(), P, ls)
    cs = rope_cos_sin(torch.cat((txt_ids, img_ids), dim=1).float(), G.axes_dim, G.theta)
    if taps is not None:
        taps["vec"], taps["img_in"], taps["txt_in"] = vec.clone(), img.clone(), txt.clone()
    for i in range(G.depth):
        img, txt = double_block(sd, f"double_blocks.{i}", img, txt, vec, cs, G, P, kv_len, ls)
        if taps is not None:
            taps[f"double.{i}.img"], taps[f"double.{i}.txt"] = img.clone(), txt.clone()
    x = torch.cat((txt, img), 1)
    for i in range(G.depth_single_blocks):
        x = single_block(sd, f"single_blocks.{i}", x, vec, cs, G, P, kv_len, ls)
        if taps is not None:
            taps[f"single.{i}"] = x.clone()
    return last_layer(sd, x[:, T:], vec, P, ls)


# ------------------------------------------------------------------------------------------------
# sampler: transport/transport.py:361-410, transport/integrators.py:82-120, transport/utils.py:33-44
# ------------------------------------------------------------------------------------------------
def time_shift(mu: float, sigma: float, t: Tensor) -> Tensor:
    t = 1 - t
    t = math.exp(mu) / (math.exp(mu) + (1 / t - 1) ** sigma)
    return 1 - t


def lin_mu(n_tokens: int, x1=256, y1=0.5, x2=4096, y2=1.15) -> float:
    m = (y2 - y1) / (x2 - x1)
    return m * n_tokens + (y1 - m * x1)


def time_grid(num_steps: int, n_tokens: int, do_shift: bool = True, time_shifting_factor=None,
              strength: Optional[float] = None) -> Tensor:
    """The solver's time points t (f32).  Flux is evaluated at 1 - t[i]; dt[i] = t[i+1] - t[i]."""
    t0, t1 = 0, 1  # check_interval for velocity + Linear path, transport.py:70-96
    if strength is not None:
        t0 = (t1 - t0) * strength + t0
    assert t0 < t1, "ODE sampler has to be in forward time"
    t = torch.linspace(t0, t1, num_steps)
    if time_shifting_factor:
        t = t / (t + time_shifting_factor - time_shifting_factor * t)
    if do_shift:
        t = time_shift(lin_mu(n_tokens), 1.0, t)
    return t


def sample_euler(model_fn, x: Tensor, cond: Optional[Tensor], t: Tensor, P: Prec = Prec(), state_f32: bool = False):
    """`Sampler.sample_ode(...)(x, model, kwargs)` with method="euler": returns the list of states.
    model_fn(x_cat, timesteps) -> velocity;  drift = -model(x || cond, 1 - t).
    state_f32 (bf16 mode only): the caller's state is f32 and STAYS f32 (transport/integrators.py:119: odeint keeps y's
    dtype) - the model sees 1 - t_i unrounded and bf16(x) (img_in's autocast), only dt * f is rounded to bf16."""
    states = [x]
    B = x.shape[0]
    evals = []
    for i in range(len(t) - 1):
        # torchdiffeq's _PerturbFunc hands the drift t.to(y.dtype): a bf16 state sees bf16(t_i) (dt stays f32-derived)
        ti = torch.ones(B) * (t[i] if state_f32 else P.r(t[i]))
        tm = torch.ones_like(ti) * (1 - ti)
        xin = torch.cat((x, cond), dim=-1) if cond is not None else x
        v = model_fn(xin, tm)
        assert v.shape == x.shape, "Output shape from ODE solver must match input shape"
        dt = t[i + 1] - t[i]
        # bf16 mode: torch casts the 0-dim f32 dt to the common dtype (bf16) before multiplying a bf16 tensor
        x = x + P.r(P.r(dt) * (-v)) if state_f32 else P.r(x + P.r(P.r(dt) * (-v)))
        states.append(x)
        evals.append(float(tm[0]))
    return states, evals


# ------------------------------------------------------------------------------------------------
# input packer (models/sampling.py:47-73) — boundary feeder for synthetic benches
# ------------------------------------------------------------------------------------------------
def grid_img_ids(rows_hw: List[tuple]) -> Tensor:
    """img_ids for a list of per-row LATENT sizes (h, w): axis0 = row index + 1, axis1 = y, axis2 = x."""
    out = []
    for j, (h, w) in enumerate(rows_hw):
        ids = torch.zeros(h // 2, w // 2, 3)
        ids[..., 0] = j + 1
        ids[..., 1] = ids[..., 1] + torch.arange(h // 2)[:, None]
        ids[..., 2] = ids[..., 2] + torch.arange(w // 2)[None, :]
        out.append(ids.reshape(-1, 3))
    return torch.cat(out, dim=0)


def pack_latent(lat: Tensor) -> Tensor:
    """"c (h ph) (w pw) -> (h w) (c ph pw)", ph = pw = 2 (models/sampling.py:61): [C,h,w] -> [(h/2)(w/2), 4C]."""
    C, h, w = lat.shape
    return lat.reshape(C, h // 2, 2, w // 2, 2).permute(1, 3, 0, 2, 4).reshape((h // 2) * (w // 2), C * 4)


def unpack_latent(tok: Tensor, h: int, w: int) -> Tensor:
    """"(h w) (c ph pw) -> c (h ph) (w pw)" (visualcloze.py:237,428): [(h/2)(w/2), 4C] -> [C,h,w]."""
    C = tok.shape[-1] // 4
    return tok.reshape(h // 2, w // 2, C, 2, 2).permute(2, 0, 3, 1, 4).reshape(C, h, w)


def pack_mask(mask: Tensor) -> Tensor:
    """visualcloze.py:381-382: pixel mask [H,W] -> 8x8 pixel-unshuffle -> 2x2 pack -> [(H/16)(W/16), 256]."""
    H, W = mask.shape
    m8 = mask.reshape(H // 8, 8, W // 8, 8).permute(1, 3, 0, 2).reshape(64, H // 8, W // 8)
    return pack_latent(m8)


def prepare_grid(samples: List[List[Tensor]]):
    """Tensor part of prepare_modified (models/sampling.py:37-100): per sample a list of row latents [1,16,h,w];
    returns img [B,Nmax,64], img_ids [B,Nmax,3], img_mask [B,Nmax] (int32), right-padded to the batch max."""
    toks, ids = [], []
    for rows in samples:
        toks.append(torch.cat([pack_latent(r.squeeze(0)) for r in rows], dim=0))
        ids.append(grid_img_ids([tuple(r.shape[-2:]) for r in rows]))
    n = max(t.shape[0] for t in toks)
    B = len(samples)
    img = torch.zeros(B, n, toks[0].shape[1], dtype=toks[0].dtype)
    img_ids = torch.zeros(B, n, 3)
    mask = torch.zeros(B, n, dtype=torch.int32)
    for b in range(B):
        k = toks[b].shape[0]
        img[b, :k], img_ids[b, :k], mask[b, :k] = toks[b], ids[b], 1
    return img, img_ids, mask
