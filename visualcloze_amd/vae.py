"""FLUX AutoEncoder (VAE encode / decode of row images) on MI355X.

Mirrors the reference's vendored twin `models/modules/autoencoder.py` (classes `AttnBlock` :25-52, `ResnetBlock` :55-82,
`Downsample` :85-95, `Upsample` :98-106, `Encoder` :109-180, `Decoder` :183-259, `DiagonalGaussian` :262-275,
`AutoEncoder` :277-311): same constructor arguments, same module tree and therefore the same `state_dict()` keys and
shapes, so `ae.safetensors` loads unchanged.  Execution is
NHWC bf16 on the HIP library: every 3x3 convolution is ONE launch of the bf16 MFMA GEMM whose loader waves gather the
taps straight from the NHWC map (`vc_conv3x3`: implicit GEMM, nearest-2x upsampling / stride-2 downsampling folded into
the addressing, fused bias / residual epilogue), GroupNorm+swish is `vc_groupnorm`,
the mid-block attention (one head, head_dim = C) is two GEMMs around `vc_softmax_rows`.  There is no CPU or torch
fallback: without the GPU library `decode` raises.

The diffusers `AutoencoderKL` that `visualcloze.py:100` actually instantiates is not part of /root/reference; parity is
pinned against the vendored twin above (tests/golden/make_vae_golden.py), not against diffusers.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List, Optional

import torch
from torch import nn

from . import hip


@dataclass
class AutoEncoderParams:            # autoencoder.py:8-18
    resolution: int = 256
    in_channels: int = 3
    ch: int = 128
    out_ch: int = 3
    ch_mult: List[int] = field(default_factory=lambda: [1, 2, 4, 4])
    num_res_blocks: int = 2
    z_channels: int = 16
    scale_factor: float = 0.3611
    shift_factor: float = 0.1159


FLUX_AE = dict(resolution=256, in_channels=3, ch=128, out_ch=3, ch_mult=[1, 2, 4, 4], num_res_blocks=2, z_channels=16,
               scale_factor=0.3611, shift_factor=0.1159)      # models/util.py:86-96


def _pad_to(n: int, m: int) -> int:
    return (n + m - 1) // m * m


class _Conv(nn.Module):
    """Parameter holder with nn.Conv2d's state-dict layout ([O, I, k, k] weight, [O] bias)."""

    def __init__(self, cin: int, cout: int, k: int):
        super().__init__()
        self.cin, self.cout, self.k = cin, cout, k
        self.weight = nn.Parameter(torch.empty(cout, cin, k, k))
        self.bias = nn.Parameter(torch.empty(cout))
        self._prep = None

    def prepared(self):
        """GEMM operands: W [O_pad8, k*k*I_pad64] bf16 with K ordered (dy, dx, c) like vc_im2col3x3's columns."""
        key = (self.weight._version, self.bias._version, self.weight.data_ptr(), self.weight.device)
        if self._prep is None or self._prep[0] != key:
            w = self.weight.detach().to(torch.bfloat16)
            cin_p, cout_p = _pad_to(self.cin, 64), _pad_to(self.cout, 8)
            wp = torch.zeros(cout_p, self.k * self.k, cin_p, dtype=torch.bfloat16, device=w.device)
            wp[: self.cout, :, : self.cin] = w.permute(0, 2, 3, 1).reshape(self.cout, self.k * self.k, self.cin)
            bp = torch.zeros(cout_p, dtype=torch.bfloat16, device=w.device)
            bp[: self.cout] = self.bias.detach().to(torch.bfloat16)
            self._prep = (key, wp.reshape(cout_p, -1).contiguous(), bp)
        return self._prep[1], self._prep[2]


class _GroupNorm(nn.Module):
    def __init__(self, c: int):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(c))
        self.bias = nn.Parameter(torch.empty(c))


class AttnBlock(nn.Module):          # autoencoder.py:25-52
    def __init__(self, in_channels: int):
        super().__init__()
        self.in_channels = in_channels
        self.norm = _GroupNorm(in_channels)
        self.q = _Conv(in_channels, in_channels, 1)
        self.k = _Conv(in_channels, in_channels, 1)
        self.v = _Conv(in_channels, in_channels, 1)
        self.proj_out = _Conv(in_channels, in_channels, 1)


class ResnetBlock(nn.Module):        # autoencoder.py:55-82
    def __init__(self, in_channels: int, out_channels: Optional[int]):
        super().__init__()
        self.in_channels = in_channels
        out_channels = in_channels if out_channels is None else out_channels
        self.out_channels = out_channels
        self.norm1 = _GroupNorm(in_channels)
        self.conv1 = _Conv(in_channels, out_channels, 3)
        self.norm2 = _GroupNorm(out_channels)
        self.conv2 = _Conv(out_channels, out_channels, 3)
        if self.in_channels != self.out_channels:
            self.nin_shortcut = _Conv(in_channels, out_channels, 1)


class Upsample(nn.Module):           # autoencoder.py:98-106
    def __init__(self, in_channels: int):
        super().__init__()
        self.conv = _Conv(in_channels, in_channels, 3)


class Downsample(nn.Module):         # autoencoder.py:85-95
    def __init__(self, in_channels: int):
        super().__init__()
        self.conv = _Conv(in_channels, in_channels, 3)


class _HipExec(nn.Module):
    """Execution helpers shared by Encoder and Decoder: NHWC bf16 activations [H*W, C] on the HIP library."""

    def _scratch(self, dev, name, shape, dtype=torch.bfloat16):
        pool = self.__dict__.setdefault("_pool", {})
        t = pool.get(name)
        n = 1
        for s in shape:
            n *= s
        if t is None or t.numel() < n or t.dtype != dtype or t.device != dev:
            t = torch.empty(n, dtype=dtype, device=dev)
            pool[name] = t
        return t[:n].view(*shape)

    def _act(self, dev, name, HW, C):
        """Activation map [HW + 1, C]: rows 0..HW-1 are the pixels, row HW is the zero row that vc_conv3x3 reads for
        every out-of-image tap (writers only touch rows < HW)."""
        t = self._scratch(dev, name, (HW + 1, C))
        t[HW].zero_()
        return t

    def _conv3(self, conv: _Conv, xz, H, W, out, up=False, down=False, res=None):
        """out[H*W, O_pad] = conv3x3(x) (+ res); (H, W) is the output map; xz is an `_act` map (zero row appended) of the
        same, half (`up`) or double (`down`) resolution.  One launch: the GEMM's loader waves gather the taps."""
        w, b = conv.prepared()
        gate = None if res is None else self._ones(xz.device, out.shape[1])     # x + h (autoencoder.py:82) = gate of ones
        hip.conv3x3(xz, w, b, out, H, W, up=up, down=down, res=res, gate=gate)

    def _ones(self, dev, n):
        pool = self.__dict__.setdefault("_pool", {})
        o = pool.get("ones%d" % n)
        if o is None or o.device != dev:
            o = torch.ones(n, dtype=torch.bfloat16, device=dev)       # filled once, never written again
            pool["ones%d" % n] = o
        return o

    def _conv1(self, conv: _Conv, x, out, res=None):
        w, b = conv.prepared()
        if res is None:
            hip.gemm(hip.make_problem(x, w, b, out), epi=hip.EPI_BIAS)
        else:
            hip.gemm(hip.make_problem(x, w, b, out, res=res, gate=self._ones(x.device, out.shape[1]), rows_per_batch=x.shape[0]),
                     epi=hip.EPI_GATE_RES)

    def _norm(self, gn: _GroupNorm, x, y, swish):
        sc = self._scratch(x.device, "gn", (hip.groupnorm_scratch_floats(x.shape[0]),), torch.float32)
        c = gn.__dict__.get("_bf16_affine")
        key = (gn.weight.data_ptr(), gn.weight._version, gn.bias.data_ptr(), gn.bias._version)
        if c is None or c[0] != key:        # converted once per weight load, not per call
            c = (key, gn.weight.detach().to(torch.bfloat16).contiguous(), gn.bias.detach().to(torch.bfloat16).contiguous())
            gn.__dict__["_bf16_affine"] = c
        hip.groupnorm(x, c[1], c[2], y, sc, swish=swish)

    def _resnet(self, blk: ResnetBlock, x, H, W, tag):
        """x: `_act` map [HW+1, Cin]; returns an `_act` map [HW+1, Cout]."""
        dev, HW = x.device, H * W
        t = self._act(dev, "t0", HW, blk.in_channels)
        self._norm(blk.norm1, x[:HW], t[:HW], True)
        h = self._scratch(dev, "t1", (HW, blk.out_channels))
        self._conv3(blk.conv1, t, H, W, h)
        t2 = self._act(dev, "t2", HW, blk.out_channels)
        self._norm(blk.norm2, h, t2[:HW], True)
        r = x[:HW]
        if blk.in_channels != blk.out_channels:
            sc = self._scratch(dev, "t3", (HW, blk.out_channels))
            self._conv1(blk.nin_shortcut, x[:HW], sc)
            r = sc
        out = self._act(dev, "x" + tag, HW, blk.out_channels)
        self._conv3(blk.conv2, t2, H, W, out[:HW], res=r)
        return out

    def _attn(self, blk: AttnBlock, xz, tag):
        """xz: `_act` map [L+1, C]; returns an `_act` map."""
        dev, L, Cc = xz.device, xz.shape[0] - 1, xz.shape[1]
        x = xz[:L]
        t = self._scratch(dev, "t0", (L, Cc))
        self._norm(blk.norm, x, t, False)
        Lk, Lp = _pad_to(L, 8), _pad_to(L, 64)               # N of the S GEMM / K of the P.V GEMM; pads stay zero
        q, v = self._scratch(dev, "aq", (L, Cc)), self._scratch(dev, "av", (L, Cc))
        k = self._scratch(dev, "ak", (Lk, Cc))
        if Lk != L:
            k[L:].zero_()                                    # zero key rows -> zero score columns L..Lk-1
        self._conv1(blk.q, t, q); self._conv1(blk.k, t, k[:L]); self._conv1(blk.v, t, v)
        s = self._scratch(dev, "as", (L, Lp))
        if Lp != L:
            s.zero_()
        hip.gemm(hip.make_problem(q, k, None, s[:, :Lk]), epi=hip.EPI_BIAS)                 # S = Q K^T  [L, L]
        hip.softmax_rows(s[:, :L], float(Cc) ** -0.5)
        vt = self._scratch(dev, "avt", (Cc, Lp))
        if Lp != L:
            vt.zero_()
        hip.transpose(v, vt[:, :L])
        o = self._scratch(dev, "ao", (L, Cc))
        hip.gemm(hip.make_problem(s, vt, None, o), epi=hip.EPI_BIAS)                        # O = P V
        out = self._act(dev, "x" + tag, L, Cc)
        self._conv1(blk.proj_out, o, out[:L], res=x)
        return out


def _level(cin: int, cout: int, n_blocks: int, resample_name: str, resample) -> nn.Module:
    """One resolution level of the encoder / decoder: `n_blocks` ResnetBlocks (the first maps cin -> cout), an empty `attn` list
    (the FLUX AutoEncoder has none outside `mid`) and, except at the last level of the walk, a Downsample / Upsample - registered
    in the order that gives the reference's state-dict key order (autoencoder.py:128-147, 212-232)."""
    lvl = nn.Module()
    lvl.block = nn.ModuleList(ResnetBlock(cin if j == 0 else cout, cout) for j in range(n_blocks))
    lvl.attn = nn.ModuleList()
    if resample is not None:
        setattr(lvl, resample_name, resample(cout))
    return lvl


def _mid(width: int) -> nn.Module:
    m = nn.Module()
    m.block_1, m.attn_1, m.block_2 = ResnetBlock(width, width), AttnBlock(width), ResnetBlock(width, width)
    return m


class Encoder(_HipExec):             # autoencoder.py:109-180
    def __init__(self, resolution: int, in_channels: int, ch: int, ch_mult: List[int], num_res_blocks: int, z_channels: int):
        super().__init__()
        self.ch, self.z_channels = ch, z_channels
        self.num_resolutions = len(ch_mult)
        self.num_res_blocks = num_res_blocks
        self.resolution, self.in_channels = resolution, in_channels
        self.in_ch_mult = (1,) + tuple(ch_mult)
        widths = [ch * m for m in self.in_ch_mult]                    # channel count entering level i = widths[i], leaving = widths[i + 1]
        self.conv_in = _Conv(in_channels, ch, 3)
        last = self.num_resolutions - 1
        self.down = nn.ModuleList(_level(widths[i], widths[i + 1], num_res_blocks, "downsample", None if i == last else Downsample)
                                  for i in range(self.num_resolutions))
        self.mid = _mid(widths[-1])
        self.norm_out = _GroupNorm(widths[-1])
        self.conv_out = _Conv(widths[-1], 2 * z_channels, 3)

    def _moments_one(self, img):
        """img [in_channels, H, W] -> (moments NHWC [h*w, pad8(2z)], h, w)   (Encoder.forward, :159-180)"""
        _, H, W = img.shape
        f = 2 ** (self.num_resolutions - 1)
        if H % f or W % f:
            raise ValueError(f"Encoder: image size {H}x{W} must be a multiple of {f}")
        h, w = H // f, W // f
        mom = self._scratch(img.device, "yout", (h * w, _pad_to(2 * self.z_channels, 8)))
        self._moments_into(img, mom)
        return mom, h, w

    def _moments_into(self, img, mom):
        dev = img.device
        _, H, W = img.shape
        x0 = self._act(dev, "zin", H * W, _pad_to(self.in_channels, 64))
        hip.nchw_to_nhwc(img, x0[:H * W], 1.0, 0.0)
        cur = self._act(dev, "xa", H * W, self.conv_in.cout)
        self._conv3(self.conv_in, x0, H, W, cur[:H * W])
        flip = ["b", "a"]
        k = 0
        for i_level in range(self.num_resolutions):
            for blk in self.down[i_level].block:
                cur = self._resnet(blk, cur, H, W, flip[k & 1]); k += 1
            if i_level != self.num_resolutions - 1:
                H, W = H // 2, W // 2
                nxt = self._act(dev, "x" + flip[k & 1], H * W, cur.shape[1]); k += 1
                self._conv3(self.down[i_level].downsample.conv, cur, H, W, nxt[:H * W], down=True)
                cur = nxt
        cur = self._resnet(self.mid.block_1, cur, H, W, flip[k & 1]); k += 1
        cur = self._attn(self.mid.attn_1, cur, flip[k & 1]); k += 1
        cur = self._resnet(self.mid.block_2, cur, H, W, flip[k & 1]); k += 1
        t = self._act(dev, "t0", H * W, cur.shape[1])
        self._norm(self.norm_out, cur[:H * W], t[:H * W], True)
        self._conv3(self.conv_out, t, H, W, mom)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """x [B, in_channels, H, W] -> moments [B, 2*z_channels, H/8, W/8] bf16 (the reference's Encoder.forward)."""
        hip.require_gpu()
        if x.dim() != 4 or x.shape[1] != self.in_channels:
            raise ValueError(f"Encoder expects [B, {self.in_channels}, H, W], got {tuple(x.shape)}")
        outs = []
        for xi in x:
            mom, h, w = self._moments_one(xi.contiguous())
            o = torch.empty(2 * self.z_channels, h, w, dtype=torch.bfloat16, device=x.device)
            hip.nhwc_to_nchw(mom, o)
            outs.append(o)
        return torch.stack(outs)


class Decoder(_HipExec):             # autoencoder.py:183-259
    def __init__(self, ch: int, out_ch: int, ch_mult: List[int], num_res_blocks: int, in_channels: int, resolution: int,
                 z_channels: int):
        super().__init__()
        self.ch, self.out_ch, self.z_channels = ch, out_ch, z_channels
        self.num_resolutions = len(ch_mult)
        self.num_res_blocks = num_res_blocks
        self.resolution, self.in_channels = resolution, in_channels
        self.ffactor = 2 ** (self.num_resolutions - 1)
        widths = [ch * m for m in ch_mult]                            # level i runs at widths[i]; the walk goes from the last level down to 0
        top = widths[-1]
        self.conv_in = _Conv(z_channels, top, 3)
        self.mid = _mid(top)
        enter = [top] + widths[:0:-1]                                 # channels entering levels n-1, n-2, ..., 0 (the previous level's width)
        levels = [_level(cin, widths[i], num_res_blocks + 1, "upsample", None if i == 0 else Upsample)
                  for cin, i in zip(enter, reversed(range(self.num_resolutions)))]
        self.up = nn.ModuleList(levels[::-1])                         # stored by level index, as the state dict names them
        self.norm_out = _GroupNorm(widths[0])
        self.conv_out = _Conv(widths[0], out_ch, 3)

    def forward(self, z: torch.Tensor) -> torch.Tensor:
        """z: [B, z_channels, h, w] (f32 or bf16) -> image [B, out_ch, 8h, 8w] bf16 (the reference's Decoder.forward)."""
        hip.require_gpu()
        if z.dim() != 4 or z.shape[1] != self.z_channels:
            raise ValueError(f"Decoder expects [B, {self.z_channels}, h, w], got {tuple(z.shape)}")
        outs = []
        for zi in z:
            outs.append(self._decode_one(zi.contiguous(), 1.0, 0.0))
        return torch.stack(outs)

    def _decode_one(self, z, div, add):
        _, h, w = z.shape
        f = self.ffactor
        img = torch.empty(self.out_ch, f * h, f * w, dtype=torch.bfloat16, device=z.device)
        self._decode_into(z, div, add, img)
        return img

    def _decode_into(self, z, div, add, img):
        dev = z.device
        _, h, w = z.shape
        H, W = h, w
        x0 = self._act(dev, "zin", H * W, _pad_to(self.z_channels, 64))
        hip.nchw_to_nhwc(z, x0[:H * W], div, add)
        cur = self._act(dev, "xa", H * W, self.conv_in.cout)
        self._conv3(self.conv_in, x0, H, W, cur[:H * W])
        flip = ["b", "a"]
        k = 0
        cur = self._resnet(self.mid.block_1, cur, H, W, flip[k & 1]); k += 1
        cur = self._attn(self.mid.attn_1, cur, flip[k & 1]); k += 1
        cur = self._resnet(self.mid.block_2, cur, H, W, flip[k & 1]); k += 1
        for i_level in reversed(range(self.num_resolutions)):
            for blk in self.up[i_level].block:
                cur = self._resnet(blk, cur, H, W, flip[k & 1]); k += 1
            if i_level != 0:
                H, W = 2 * H, 2 * W
                nxt = self._act(dev, "x" + flip[k & 1], H * W, cur.shape[1]); k += 1
                self._conv3(self.up[i_level].upsample.conv, cur, H, W, nxt[:H * W], up=True)
                cur = nxt
        t = self._act(dev, "t0", H * W, cur.shape[1])
        self._norm(self.norm_out, cur[:H * W], t[:H * W], True)
        y = self._scratch(dev, "yout", (H * W, _pad_to(self.out_ch, 8)))
        self._conv3(self.conv_out, t, H, W, y)
        hip.nhwc_to_nchw(y, img)


class AutoEncoder(nn.Module):
    """autoencoder.py:277-311.  `encode` takes the Gaussian noise as an argument (the reference draws
    `torch.randn_like(mean)` inside DiagonalGaussian, :272; pass `noise=None` to draw the same way here, or
    `sample=False` for the mean)."""

    def __init__(self, params: AutoEncoderParams):
        super().__init__()
        self.encoder = Encoder(resolution=params.resolution, in_channels=params.in_channels, ch=params.ch,
                               ch_mult=params.ch_mult, num_res_blocks=params.num_res_blocks, z_channels=params.z_channels)
        self.decoder = Decoder(resolution=params.resolution, in_channels=params.in_channels, ch=params.ch,
                               out_ch=params.out_ch, ch_mult=params.ch_mult, num_res_blocks=params.num_res_blocks,
                               z_channels=params.z_channels)
        self.scale_factor = params.scale_factor
        self.shift_factor = params.shift_factor

    def encode(self, x: torch.Tensor, noise: Optional[torch.Tensor] = None, sample: bool = True) -> torch.Tensor:
        hip.require_gpu()
        enc = self.encoder
        if x.dim() != 4 or x.shape[1] != enc.in_channels:
            raise ValueError(f"encode expects [B, {enc.in_channels}, H, W], got {tuple(x.shape)}")
        outs = []
        for i, xi in enumerate(x):
            mom, h, w = enc._moments_one(xi.contiguous())
            z = torch.empty(enc.z_channels, h, w, dtype=torch.bfloat16, device=x.device)
            n = None
            if sample:
                n = (noise[i] if noise is not None else torch.randn_like(z)).to(torch.bfloat16).contiguous()
            hip.gaussian_sample(mom, n, z, self.scale_factor, self.shift_factor)
            outs.append(z)
        return torch.stack(outs)

    def decode(self, z: torch.Tensor) -> torch.Tensor:
        hip.require_gpu()
        if z.dim() != 4 or z.shape[1] != self.decoder.z_channels:
            raise ValueError(f"decode expects [B, {self.decoder.z_channels}, h, w], got {tuple(z.shape)}")
        return torch.stack([self.decoder._decode_one(zi.contiguous(), self.scale_factor, self.shift_factor) for zi in z])

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return self.decode(self.encode(x))
