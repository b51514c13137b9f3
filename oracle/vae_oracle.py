"""CPU restatement of the FLUX AutoEncoder (encode + decode) — TEST INFRASTRUCTURE ONLY (tests/, smoke, bench cpu leg).

Follows the reference's vendored twin models/modules/autoencoder.py: `swish` :21-22, `AttnBlock` :25-52,
`ResnetBlock.forward` :69-82, `Downsample.forward` :91-95, `Upsample.forward` :103-106, `Encoder.forward` :159-180,
`Decoder.forward` :237-259, `DiagonalGaussian.forward` :268-275, `AutoEncoder.encode/decode` :301-308.
Weights are addressed by the reference's state-dict keys (`encoder.*`, `decoder.*`).  Two modes:
  fp32  exact reference semantics in float32
  bf16  every tensor the reference materialises when the module runs in bfloat16 (conv / GroupNorm / sigmoid / product /
        add outputs, q k v, attention output) is rounded to bf16; accumulations stay f32
Pinned by tests/golden/vae_golden.npz, produced by importing the reference file itself (tests/golden/make_vae_golden.py).
The diffusers AutoencoderKL that visualcloze.py instantiates is NOT in /root/reference: parity with it is unpinned.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F


def _r(x: torch.Tensor, bf16: bool) -> torch.Tensor:
    return x.to(torch.bfloat16).to(torch.float32) if bf16 else x


def _conv(sd, key, x, bf16, padding, stride=1):
    w, b = sd[key + ".weight"].float(), sd[key + ".bias"].float()
    return _r(F.conv2d(x, w, b, stride=stride, padding=padding), bf16)


def _group_norm(sd, key, x, bf16, groups=32, eps=1e-6):
    b, c, h, w = x.shape
    xg = x.reshape(b, groups, -1).double()
    mean = xg.mean(-1, keepdim=True)
    var = xg.var(-1, unbiased=False, keepdim=True)
    xn = ((xg - mean) / torch.sqrt(var + eps)).float().reshape(b, c, h, w)
    g, be = sd[key + ".weight"].float().view(1, c, 1, 1), sd[key + ".bias"].float().view(1, c, 1, 1)
    return _r(xn * g + be, bf16)


def _swish(x, bf16):                       # x * sigmoid(x): the sigmoid is a tensor of its own in the reference
    return _r(x * _r(torch.sigmoid(x), bf16), bf16)


def _resnet(sd, p, x, bf16):
    h = _conv(sd, p + ".conv1", _swish(_group_norm(sd, p + ".norm1", x, bf16), bf16), bf16, 1)
    h = _conv(sd, p + ".conv2", _swish(_group_norm(sd, p + ".norm2", h, bf16), bf16), bf16, 1)
    if (p + ".nin_shortcut.weight") in sd:
        x = _conv(sd, p + ".nin_shortcut", x, bf16, 0)
    return _r(x + h, bf16)


def _attn(sd, p, x, bf16):
    b, c, hh, ww = x.shape
    hn = _group_norm(sd, p + ".norm", x, bf16)
    q, k, v = (_conv(sd, p + "." + n, hn, bf16, 0).reshape(b, c, hh * ww).transpose(1, 2) for n in ("q", "k", "v"))
    s = torch.softmax((q @ k.transpose(1, 2)) * (float(c) ** -0.5), dim=-1)
    o = _r(s @ v, bf16).transpose(1, 2).reshape(b, c, hh, ww)
    return _r(x + _conv(sd, p + ".proj_out", o, bf16, 0), bf16)


def decoder_forward(sd: dict, z: torch.Tensor, params: dict, mode: str = "fp32", taps: dict | None = None) -> torch.Tensor:
    """Decoder.forward (autoencoder.py:237-259). sd keys are relative to the decoder (`conv_in.weight`, ...)."""
    bf16 = mode == "bf16"
    nres, nblk = len(params["ch_mult"]), params["num_res_blocks"]
    h = _conv(sd, "conv_in", _r(z.float(), bf16), bf16, 1)
    if taps is not None: taps["conv_in"] = h
    h = _resnet(sd, "mid.block_1", h, bf16)
    if taps is not None: taps["mid.block_1"] = h
    h = _attn(sd, "mid.attn_1", h, bf16)
    if taps is not None: taps["mid.attn_1"] = h
    h = _resnet(sd, "mid.block_2", h, bf16)
    for lvl in reversed(range(nres)):
        for i in range(nblk + 1):
            h = _resnet(sd, f"up.{lvl}.block.{i}", h, bf16)
        if lvl != 0:
            h = F.interpolate(h, scale_factor=2.0, mode="nearest")
            h = _conv(sd, f"up.{lvl}.upsample.conv", h, bf16, 1)
        if taps is not None: taps[f"up.{lvl}"] = h
    h = _swish(_group_norm(sd, "norm_out", h, bf16), bf16)
    return _conv(sd, "conv_out", h, bf16, 1)


def decode(sd: dict, z: torch.Tensor, params: dict, mode: str = "fp32", taps: dict | None = None) -> torch.Tensor:
    """AutoEncoder.decode (autoencoder.py:306-308); sd has the `decoder.` prefix as in ae.safetensors."""
    bf16 = mode == "bf16"
    dsd = {k[len("decoder."):]: v for k, v in sd.items() if k.startswith("decoder.")}
    z = _r(_r(_r(z.float(), bf16) / params["scale_factor"], bf16) + params["shift_factor"], bf16)
    return decoder_forward(dsd, z, params, mode, taps)


def encoder_forward(sd: dict, x: torch.Tensor, params: dict, mode: str = "fp32") -> torch.Tensor:
    """Encoder.forward (autoencoder.py:159-180) -> moments [B, 2z, h, w]. sd keys relative to the encoder."""
    bf16 = mode == "bf16"
    nres, nblk = len(params["ch_mult"]), params["num_res_blocks"]
    h = _conv(sd, "conv_in", _r(x.float(), bf16), bf16, 1)
    for lvl in range(nres):
        for i in range(nblk):
            h = _resnet(sd, f"down.{lvl}.block.{i}", h, bf16)
        if lvl != nres - 1:
            h = F.pad(h, (0, 1, 0, 1), mode="constant", value=0)
            h = _conv(sd, f"down.{lvl}.downsample.conv", h, bf16, 0, stride=2)
    h = _resnet(sd, "mid.block_1", h, bf16)
    h = _attn(sd, "mid.attn_1", h, bf16)
    h = _resnet(sd, "mid.block_2", h, bf16)
    h = _swish(_group_norm(sd, "norm_out", h, bf16), bf16)
    return _conv(sd, "conv_out", h, bf16, 1)


def encode(sd: dict, x: torch.Tensor, params: dict, noise: torch.Tensor | None, mode: str = "fp32") -> torch.Tensor:
    """AutoEncoder.encode (autoencoder.py:301-304) with the DiagonalGaussian noise as an input (None -> the mean)."""
    bf16 = mode == "bf16"
    esd = {k[len("encoder."):]: v for k, v in sd.items() if k.startswith("encoder.")}
    mom = encoder_forward(esd, x, params, mode)
    mean, logvar = torch.chunk(mom, 2, dim=1)
    z = mean
    if noise is not None:
        std = _r(torch.exp(_r(0.5 * logvar, bf16)), bf16)
        z = _r(mean + _r(std * _r(noise.float(), bf16), bf16), bf16)
    return _r(params["scale_factor"] * _r(z - params["shift_factor"], bf16), bf16)
