"""Host-side mirror of the reference's model interface for the denoising path.

`Flux`, `FluxLoraWrapper` and `FluxParams` keep the reference's constructor, `state_dict` key/shape contract
and `forward` keyword surface (models/model.py:18-175), so `visualcloze.py` /
`sample.py` can construct, `.to()`, `load_state_dict(strict=False)` and call `model.forward` unchanged.
The modules below are PARAMETER HOLDERS: every FLOP of `forward` runs in libvcloze_hip.so via
`FluxEngine`.  There is no torch fallback — without a GPU or without the library, forward raises.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import List, Optional

import torch
from torch import Tensor, nn

from . import hip
from .engine import FluxEngine, PreparedWeights


@dataclass
class FluxParams:
    in_channels: int
    out_channels: int
    vec_in_dim: int
    context_in_dim: int
    hidden_size: int
    mlp_ratio: float
    num_heads: int
    depth: int
    depth_single_blocks: int
    axes_dim: list
    theta: int
    qkv_bias: bool
    guidance_embed: bool


# "flux-dev-fill-lora" hyper-parameters (models/util.py:132-165)
FLUX_DEV_FILL = dict(in_channels=384, out_channels=64, vec_in_dim=768, context_in_dim=4096, hidden_size=3072,
                     mlp_ratio=4.0, num_heads=24, depth=19, depth_single_blocks=38, axes_dim=[16, 56, 56],
                     theta=10_000, qkv_bias=True, guidance_embed=True)


class _NoTorchForward(nn.Module):
    def forward(self, *a, **k):  # pragma: no cover
        raise hip.VclozeHipError(f"{type(self).__name__} is a parameter holder; the denoising path runs in "
                                 "libvcloze_hip.so through Flux.forward (no torch fallback)")


class Linear(_NoTorchForward):
    """nn.Linear-shaped holder (`weight [out,in]`, `bias [out]`); LoRA factors are added by `add_lora`."""

    def __init__(self, in_features: int, out_features: int, bias: bool = True):
        super().__init__()
        self.in_features, self.out_features = in_features, out_features
        self.weight = nn.Parameter(torch.empty(out_features, in_features))
        self.bias = nn.Parameter(torch.empty(out_features)) if bias else None
        self.scale = 1.0
        self.rank = 0
        bound = 1 / math.sqrt(in_features)
        with torch.no_grad():
            self.weight.uniform_(-bound, bound)
            if self.bias is not None:
                self.bias.uniform_(-bound, bound)

    def add_lora(self, max_rank: int, scale: float) -> None:
        """LinearLora.__init__ (models/modules/lora.py:34-90): rank clipped to min(in,out), lora_B has a bias,
        lora_B initialised to zero."""
        assert isinstance(scale, float), "scale must be a float"
        self.rank = min(max_rank, self.in_features, self.out_features)
        self.scale = scale
        self.lora_A = Linear(self.in_features, self.rank, bias=False)
        self.lora_B = Linear(self.rank, self.out_features, bias=True)
        with torch.no_grad():
            self.lora_A.to(self.weight.device, self.weight.dtype)
            self.lora_B.to(self.weight.device, self.weight.dtype)
            self.lora_B.weight.zero_()
            self.lora_B.bias.zero_()

    def set_scale(self, scale: float) -> None:
        assert isinstance(scale, float), "scalar value must be a float"
        self.scale = scale


class MLPEmbedder(_NoTorchForward):
    def __init__(self, in_dim: int, hidden_dim: int):
        super().__init__()
        self.in_layer = Linear(in_dim, hidden_dim)
        self.out_layer = Linear(hidden_dim, hidden_dim)


class RMSNorm(_NoTorchForward):
    def __init__(self, dim: int):
        super().__init__()
        self.scale = nn.Parameter(torch.ones(dim))


class QKNorm(_NoTorchForward):
    def __init__(self, dim: int):
        super().__init__()
        self.query_norm, self.key_norm = RMSNorm(dim), RMSNorm(dim)


class SelfAttention(_NoTorchForward):
    def __init__(self, dim: int, num_heads: int, qkv_bias: bool):
        super().__init__()
        self.qkv = Linear(dim, dim * 3, bias=qkv_bias)
        self.norm = QKNorm(dim // num_heads)
        self.proj = Linear(dim, dim)


class Modulation(_NoTorchForward):
    def __init__(self, dim: int, double: bool):
        super().__init__()
        self.is_double = double
        self.multiplier = 6 if double else 3
        self.lin = Linear(dim, self.multiplier * dim)


class _Seq(nn.Sequential, _NoTorchForward):
    pass


class _Empty(_NoTorchForward):
    """stands in for parameter-free members (nn.GELU / nn.SiLU / LayerNorm without affine) to keep indices"""


class DoubleStreamBlock(_NoTorchForward):
    def __init__(self, hidden_size: int, num_heads: int, mlp_ratio: float, qkv_bias: bool = False):
        super().__init__()
        mlp = int(hidden_size * mlp_ratio)
        self.img_mod = Modulation(hidden_size, True)
        self.img_attn = SelfAttention(hidden_size, num_heads, qkv_bias)
        self.img_mlp = _Seq(Linear(hidden_size, mlp), _Empty(), Linear(mlp, hidden_size))
        self.txt_mod = Modulation(hidden_size, True)
        self.txt_attn = SelfAttention(hidden_size, num_heads, qkv_bias)
        self.txt_mlp = _Seq(Linear(hidden_size, mlp), _Empty(), Linear(mlp, hidden_size))


class SingleStreamBlock(_NoTorchForward):
    def __init__(self, hidden_size: int, num_heads: int, mlp_ratio: float = 4.0):
        super().__init__()
        self.mlp_hidden_dim = int(hidden_size * mlp_ratio)
        self.linear1 = Linear(hidden_size, hidden_size * 3 + self.mlp_hidden_dim)
        self.linear2 = Linear(hidden_size + self.mlp_hidden_dim, hidden_size)
        self.norm = QKNorm(hidden_size // num_heads)
        self.modulation = Modulation(hidden_size, False)


class LastLayer(_NoTorchForward):
    def __init__(self, hidden_size: int, patch_size: int, out_channels: int):
        super().__init__()
        self.linear = Linear(hidden_size, patch_size * patch_size * out_channels)
        self.adaLN_modulation = _Seq(_Empty(), Linear(hidden_size, 2 * hidden_size))


class MaskLayout:
    """How (txt_mask, img_mask) map onto the kernels' masks.  flash-attn's varlen path (math.py:9-60) drops masked rows
    of the joint sequence wherever they are; attention does not care about key order (RoPE travels with the row), every
    other op of Flux.forward is per-row, so each stream's rows are reordered VALID FIRST (stable), which leaves
        [txt valid | txt masked | img valid | img masked]
    = a prefix length kv_len = T + n_img plus one masked gap (n_txt, T) per sample, and the result rows are scattered
    back.  Right-padded masks (all the reference's callers build, sampling.py:41-46,68-70,98) are the identity case."""

    def __init__(self, txt_mask, img_mask, B: int, T: int, N: int):
        self.B, self.T, self.N = B, T, N
        self.perm_t = self.perm_i = self.inv_i = None
        if txt_mask is None or img_mask is None:
            self.n_txt, self.n_img = [T] * B, [N] * B
            return
        tm = txt_mask.reshape(B, T).to("cpu") != 0
        im = img_mask.reshape(B, N).to("cpu") != 0
        self.n_txt = [int(v) for v in tm.sum(1)]
        self.n_img = [int(v) for v in im.sum(1)]
        pt = torch.sort((~tm).to(torch.uint8), dim=1, stable=True).indices
        pi = torch.sort((~im).to(torch.uint8), dim=1, stable=True).indices
        if not torch.equal(pt, torch.arange(T).expand(B, T)):
            self.perm_t = pt
        if not torch.equal(pi, torch.arange(N).expand(B, N)):
            self.perm_i = pi
            self.inv_i = torch.argsort(pi, dim=1)

    def kv_len(self, sl: slice) -> List[int]:
        return [self.T + n for n in self.n_img[sl]]

    def kv_gap(self, sl: slice):
        gaps = [(n, self.T) if n < self.T else (0, 0) for n in self.n_txt[sl]]
        return gaps if any(hi > lo for lo, hi in gaps) else None

    @staticmethod
    def _take(x: Tensor, perm, sl: slice) -> Tensor:
        x = x[sl]
        if perm is None:
            return x
        idx = perm[sl].to(x.device)
        return torch.gather(x, 1, idx.reshape(idx.shape + (1,) * (x.dim() - 2)).expand(x.shape))

    def txt_rows(self, x: Tensor, sl: slice) -> Tensor:
        """rows of a [B, T, ...] text-stream tensor (txt, txt_ids) in kernel order"""
        return self._take(x, self.perm_t, sl)

    def img_rows(self, x: Tensor, sl: slice) -> Tensor:
        """rows of a [B, N, ...] image-stream tensor (img / x, cond, img_ids) in kernel order"""
        return self._take(x, self.perm_i, sl)

    def img_rows_back(self, x: Tensor, sl: slice) -> Tensor:
        """kernel-order [bs, N, ...] image-stream rows back in the caller's order"""
        if self.inv_i is None:
            return x
        idx = self.inv_i[sl].to(x.device)
        return torch.gather(x, 1, idx.reshape(idx.shape + (1,) * (x.dim() - 2)).expand(x.shape))


def per_sample(v: Optional[Tensor], B: int) -> Optional[Tensor]:
    """A per-sample vector [B]; a single value (the pipeline's `guidance = torch.full((1,), cfg)`, visualcloze.py:413)
    broadcasts over the batch as it does in the reference's `vec + guidance_in(...)`."""
    if v is None:
        return None
    v = v.reshape(-1)
    if v.numel() == 1 and B > 1:
        v = v.expand(B)
    if v.numel() != B:
        raise ValueError(f"expected 1 or {B} per-sample values, got {v.numel()}")
    return v


class Flux(nn.Module):
    """Drop-in for models/model.py:35-151 (inference surface)."""

    def __init__(self, params: FluxParams):
        super().__init__()
        self.params = params
        self.in_channels, self.out_channels = params.in_channels, params.out_channels
        if params.hidden_size % params.num_heads != 0:
            raise ValueError(f"Hidden size {params.hidden_size} must be divisible by num_heads {params.num_heads}")
        pe_dim = params.hidden_size // params.num_heads
        if sum(params.axes_dim) != pe_dim:
            raise ValueError(f"Got {params.axes_dim} but expected positional dim {pe_dim}")
        if pe_dim != 128:
            raise ValueError("the gfx950 attention kernel is specialised for head_dim 128 (FLUX)")
        self.hidden_size, self.num_heads = params.hidden_size, params.num_heads
        self.img_in = Linear(self.in_channels, self.hidden_size)
        self.time_in = MLPEmbedder(256, self.hidden_size)
        self.vector_in = MLPEmbedder(params.vec_in_dim, self.hidden_size)
        self.guidance_in = MLPEmbedder(256, self.hidden_size) if params.guidance_embed else nn.Identity()
        self.txt_in = Linear(params.context_in_dim, self.hidden_size)
        self.double_blocks = nn.ModuleList(
            [DoubleStreamBlock(self.hidden_size, self.num_heads, params.mlp_ratio, params.qkv_bias)
             for _ in range(params.depth)])
        self.single_blocks = nn.ModuleList(
            [SingleStreamBlock(self.hidden_size, self.num_heads, params.mlp_ratio)
             for _ in range(params.depth_single_blocks)])
        self.final_layer = LastLayer(self.hidden_size, 1, self.out_channels)
        self._engine: Optional[FluxEngine] = None
        self._handle = None
        self._fingerprint = None
        # True: Flux.forward and the fused sampler run through the C handle API (vc_flux_*, csrc/flux_engine.hip), one
        # call per evaluation / per trajectory.  False (or lora_mode "ref"): the same launch plan ordered from Python
        # (engine.FluxEngine) over the op-level ABI - bit-identical results, kept for the parity mode and per-block taps.
        self.use_handle = True
        # "merged": W + s*B@A folded once (the product mode).  "ref": LinearLora.forward executed as the
        # reference does - base GEMM, two skinny GEMMs, three bf16 roundings (models/modules/lora.py:92-98) - so that
        # the merge's one-rounding deviation is a choice, not a necessity; slower (+8 % FLOPs, unfused epilogues).
        self.lora_mode = "merged"
        # the prepared qkv weights (merged mode) are stored HEAD-PERMUTED - every key head inside one 192-column GEMM tile -
        # so that QKNorm + RoPE of k can ride the projection's epilogue (hip.qkv_head_permutation, VcGemmProblem.kn_heads)
        self.qkv_permute = True

    # ------------------------------------------------------------------ weights -> engine
    def _linears(self):
        for name, m in self.named_modules():
            if isinstance(m, Linear) and ".lora_" not in name:
                yield name, m

    def _weights_fingerprint(self):
        return tuple((p.data_ptr(), p._version) for p in self.parameters()) + tuple(
            m.scale for _, m in self._linears()) + (self.lora_mode, self.qkv_permute)

    @staticmethod
    @torch.no_grad()
    def merged_linear(m: "Linear", consume: bool = False, out: Optional[Tensor] = None):
        """(W', b') of one Linear as the engine executes it: bf16(W + s*B@A), bf16(b + s*b_B) - LinearLora
        (models/modules/lora.py:92-98) with its LoRA pair folded in, exact in f32 and rounded to bf16 once.
        `out`: write W' into this [out, in] bf16 view instead of a new tensor (the stacked modulation matrix).
        `consume`: the module gives its storage away - W' overwrites m.weight's own memory (when that is a contiguous bf16
        tensor) and every parameter of the Linear is left EMPTY, so that preparing a sampling-only rank never holds the
        un-merged and the merged set at once (26.3 GB peak instead of 56.7)."""
        W = m.weight.detach()
        W32 = W.float()                   # NOT a copy when the parameter is f32 already: never update it in place
        B32 = None if m.bias is None else m.bias.detach().float()
        if m.rank:
            delta = m.scale * (m.lora_B.weight.detach().float() @ m.lora_A.weight.detach().float())
            if W32.data_ptr() == W.data_ptr() and not consume:
                W32 = W32 + delta                 # out of place: the module keeps its un-merged weight
            else:
                W32 += delta                      # our own f32 temporary (or storage the module gives away)
            del delta
            if m.lora_B.bias is not None:
                B32 = (B32 if B32 is not None else 0) + m.scale * m.lora_B.bias.detach().float()
        if out is None and consume and W.dtype == torch.bfloat16 and W.is_contiguous():
            out = W
        if out is None:
            out = W32.to(torch.bfloat16).contiguous()
        else:
            out.copy_(W32)
        del W32
        bias = None if B32 is None else B32.to(torch.bfloat16).contiguous()
        if consume:
            for p in m.parameters():
                p.data = torch.empty(0, dtype=p.dtype, device=p.device)
        return out, bias

    @torch.no_grad()
    def prepare(self, free_parameters: bool = False) -> FluxEngine:
        """(Re)build the engine's device weights: bf16, contiguous, LoRA merged as W + s*B@A, b + s*b_B
        (exact in f32, rounded to bf16 once — DESIGN.md §numerics).  One-time preprocessing; uses torch ops.
        `free_parameters`: release the module's own parameter storage afterwards (sampling-only ranks of a
        data-parallel job: 23.8 GB of merged weights stay, the 26.3 GB of un-merged parameters go; the module can
        then no longer be re-prepared or saved until weights are loaded again)."""
        hip.require_gpu()
        dev = next(self.parameters()).device
        if dev.type != "cuda":
            raise hip.VclozeHipError("Flux weights must live on the GPU (model.to('cuda')) — there is no CPU path")
        if getattr(self, "_params_freed", False):
            raise hip.VclozeHipError("parameters were released by prepare(free_parameters=True); load weights again first")
        if self.lora_mode not in ("merged", "ref"):
            raise ValueError(f"lora_mode must be 'merged' or 'ref', got {self.lora_mode!r}")
        w, b, ref = {}, {}, ({} if self.lora_mode == "ref" else None)
        bf = lambda t: None if t is None else t.detach().to(torch.bfloat16).contiguous()  # noqa: E731
        # all modulation projections are stacked: one GEMM yields every shift/scale/gate of a step.  Their merged rows
        # are written straight into the stacked matrix (no 6.5 GB concatenation copy).
        mods: List[str] = []
        for i in range(self.params.depth):
            mods += [f"double_blocks.{i}.img_mod.lin", f"double_blocks.{i}.txt_mod.lin"]
        mods += [f"single_blocks.{i}.modulation.lin" for i in range(self.params.depth_single_blocks)]
        mods.append("final_layer.adaLN_modulation.1")
        lin = dict(self._linears())
        off, o = {}, 0
        for n in mods:
            off[n] = o
            o += lin[n].out_features
        mod_w = torch.empty(o, self.hidden_size, dtype=torch.bfloat16, device=dev)
        mod_b = torch.zeros(o, dtype=torch.bfloat16, device=dev)
        consume = free_parameters and ref is None
        for name, m in lin.items():
            if ref is None:
                if name in off:
                    _, bb = self.merged_linear(m, consume, out=mod_w[off[name]:off[name] + m.out_features])
                    if bb is not None:
                        mod_b[off[name]:off[name] + m.out_features] = bb
                else:
                    w[name], b[name] = self.merged_linear(m, consume)
                continue
            w[name], b[name] = bf(m.weight), bf(m.bias)
            A = B = bB = sc = None
            if m.rank:
                rk = (m.rank + 63) // 64 * 64            # zero-pad the rank to the GEMM's K granularity
                A = torch.zeros(rk, m.in_features, dtype=torch.bfloat16, device=dev)
                B = torch.zeros(m.out_features, rk, dtype=torch.bfloat16, device=dev)
                A[:m.rank], B[:, :m.rank] = bf(m.lora_A.weight), bf(m.lora_B.weight)
                bB = bf(m.lora_B.bias)
                # the reference multiplies the bf16 lora output by a python float (lora.py:96: f32 arithmetic, ONE rounding);
                # this mode applies it as the bf16 gate vector of the gated-residual epilogue - identical iff the scale
                # itself is a bf16 value (1.0, 0.5, 0.75 ...); anything else would be rounded once more, so refuse it
                if float(torch.tensor(m.scale, dtype=torch.bfloat16)) != float(m.scale):
                    raise ValueError(f"lora_mode='ref': lora scale {m.scale} of {name} is not representable in bf16; the parity "
                                     "mode would round it (use the merged mode, which applies it exactly in f32)")
                sc = torch.full((m.out_features,), m.scale, dtype=torch.bfloat16, device=dev)
            from .engine import RefLinear
            ref[name] = RefLinear(w[name], b[name], A, B, bB, sc)
        for name, p in self.named_parameters():
            if name.endswith("norm.scale"):
                w[name] = p.detach().to(torch.bfloat16).contiguous().clone()
        if ref is not None:              # parity mode: the modulation Linears were collected un-stacked above
            for n in mods:
                mod_w[off[n]:off[n] + w[n].shape[0]] = w[n]
                if b[n] is not None:
                    mod_b[off[n]:off[n] + w[n].shape[0]] = b[n]
        D = self.hidden_size
        qkv_heads = 0
        if ref is None and self.qkv_permute:
            qkv_heads = self.num_heads
            perm = hip.qkv_head_permutation(qkv_heads).to(dev)
            names = [f"double_blocks.{i}.{st}_attn.qkv" for i in range(self.params.depth) for st in ("img", "txt")]
            names += [f"single_blocks.{i}.linear1" for i in range(self.params.depth_single_blocks)]
            for n in names:          # in place, one matrix at a time (a 57 MB temporary): linear1's rows [0, 3D) only
                w[n][:3 * D] = w[n][:3 * D][perm]
                if b[n] is not None:
                    b[n][:3 * D] = b[n][:3 * D][perm]
        for i in range(self.params.depth_single_blocks):
            n = f"single_blocks.{i}.linear1"
            w[n + ".qkv"], w[n + ".mlp"] = w[n][: 3 * D], w[n][3 * D:]
            b[n + ".qkv"], b[n + ".mlp"] = b[n][: 3 * D], b[n][3 * D:]
            if ref is not None:         # linear1's two column ranges share lora_A
                from .engine import RefLinear
                r = ref[n]
                for sfx, sl in ((".qkv", slice(0, 3 * D)), (".mlp", slice(3 * D, None))):
                    ref[n + sfx] = RefLinear(r.w[sl], r.b[sl], r.A, None if r.B is None else r.B[sl],
                                             None if r.bB is None else r.bB[sl], None if r.scale is None else r.scale[sl])
        for n in mods:
            w.pop(n, None)
            b.pop(n, None)
        half = 128
        freqs = torch.exp(-math.log(10000) * torch.arange(0, half, dtype=torch.float32) / half).to(dev)
        # |q|, |k| <= sqrt(128) * max|scale| after QKNorm (layers.py:75-84; RoPE is a rotation), so with c = 128^-0.5 * log2(e)
        # every attention logit obeys |c q.k| <= 128 c max|q scale| max|k scale| (+2 %: q and k pass three bf16 roundings each - norm, scale, RoPE -
        # worth (1 + 2^-9)^6 = 1.012 on the product; the kernel only compares the bound with its <= 100 cutoff)
        # The bound is kept per BLOCK (a double block: both streams share one attention): each launch gets its own, so outlier
        # scales in a few blocks of a checkpoint cost the running-max template there and nowhere else; `logit_bound` = the
        # largest of them, the cap handed to the C handle (which computes the per-block values itself: flux_engine.hip resolve()).
        amax = {n: float(t.float().abs().max()) for n, t in w.items() if n.endswith(("query_norm.scale", "key_norm.scale"))}
        blocks = sorted({n.split(".img_attn.")[0].split(".txt_attn.")[0].split(".norm.")[0] for n in amax})
        c_log = 1.02 * 11.313708498984761 * 1.4426950408889634          # 1.02 sqrt(128) log2(e), the constant of flux_engine.hip
        logit_bounds = {}
        for pf in blocks:
            qm = max(v for n, v in amax.items() if n.startswith(pf + ".") and n.endswith("query_norm.scale"))
            km = max(v for n, v in amax.items() if n.startswith(pf + ".") and n.endswith("key_norm.scale"))
            bnd = c_log * qm * km
            logit_bounds[pf] = float(torch.tensor(bnd, dtype=torch.float32)) if bnd == bnd else 1e30      # (f32, as the C plan holds it)
        logit_bound = max(logit_bounds.values())
        pw = PreparedWeights(w=w, b=b, mod_w=mod_w, mod_b=mod_b, mod_off=off, n_mod=o, temb_freqs=freqs, ref=ref, qkv_heads=qkv_heads,
                             logit_bound=logit_bound, logit_bounds=logit_bounds)
        self._engine = FluxEngine(self.params, pw, dev)
        self._handle = None
        if free_parameters:
            for p in self.parameters():
                p.data = torch.empty(0, dtype=p.dtype, device=p.device)
            self._params_freed = True
            torch.cuda.empty_cache()
        self._fingerprint = self._weights_fingerprint()
        return self._engine

    def invalidate_engine(self) -> None:
        """Forget the prepared (merged, bf16) weights; the next forward / sample re-prepares.  Call after writing the
        parameters through a path that does not bump `_version` (`.data`, dist.broadcast).  Refused on a model whose
        parameters were released by prepare(free_parameters=True): the engine then holds the ONLY copy of the weights and
        dropping it would leave the module unusable (its parameters are empty, load_state_dict cannot fill them)."""
        if getattr(self, "_params_freed", False):
            raise hip.VclozeHipError("invalidate_engine: this model's parameters were released by prepare(free_parameters="
                                     "True); the prepared engine is the only copy of the weights - build a new model to "
                                     "load other weights")
        self._engine = None
        self._handle = None
        self._fingerprint = None

    def engine(self) -> FluxEngine:
        if self._engine is None or self._fingerprint != self._weights_fingerprint():
            self.prepare()
        return self._engine

    def handle(self):
        """The C-side engine (`handle.FluxHandle`) over the prepared weights, or None when this model runs the
        Python-ordered plan (use_handle False, un-merged LoRA mode).  Test / A-B knobs set on `engine()` apply to both."""
        eng = self.engine()
        if not self.use_handle or eng.W.ref is not None:
            return None
        if self._handle is None or self._handle.W is not eng.W:
            from .handle import FluxHandle
            self._handle = FluxHandle(self.params, eng.W, eng.dev)
        self._handle.set_options(eng.attn_variant, eng.tile_cfg, eng.fuse_qnorm, eng.fuse_vt, eng.W.qkv_heads, eng.fuse_knorm,
                                 eng.W.logit_bound if eng.bounded_softmax else 0.0, eng.mlp_first, eng.splitk)
        return self._handle

    # ------------------------------------------------------------------ the B1 boundary
    @torch.no_grad()
    def forward(self, img: Tensor, img_ids: Tensor, txt: Tensor, txt_ids: Tensor, timesteps: Tensor, y: Tensor,
                txt_mask: Tensor = None, img_mask: Tensor = None, guidance: Optional[Tensor] = None) -> Tensor:
        if img.ndim != 3 or txt.ndim != 3:
            raise ValueError("Input img and txt tensors must have 3 dimensions.")
        if self.params.guidance_embed and guidance is None:
            raise ValueError("Didn't get guidance strength for guidance distilled model.")
        eng = self.engine()
        B, N, _ = img.shape
        T = txt.shape[1]
        dev = eng.dev
        out = torch.empty(B, N, self.out_channels, dtype=torch.bfloat16, device=dev)
        bf = lambda t: t.to(dev, torch.bfloat16).contiguous()  # noqa: E731
        gbf16 = guidance is not None and guidance.dtype == torch.bfloat16
        guidance = per_sample(guidance, B)
        timesteps = per_sample(timesteps, B)
        lay = MaskLayout(txt_mask, img_mask, B, T, N)
        h = self.handle()
        for b0 in range(0, B, eng.MAX_BATCH):          # samples of a chunk run as ONE stacked launch sequence
            bs = min(eng.MAX_BATCH, B - b0)
            sl = slice(b0, b0 + bs)
            if h is not None:                           # one C call per chunk: vc_flux_prepare + vc_flux_forward
                h.prepare(bf(lay.txt_rows(txt, sl)), bf(y[sl]), None if guidance is None else guidance[sl], gbf16,
                          lay.img_rows(img_ids, sl), lay.txt_rows(txt_ids, sl), 1, lay.kv_len(sl), lay.kv_gap(sl))
                o = torch.empty(bs, N, self.out_channels, dtype=torch.bfloat16, device=dev)
                h.forward(bf(lay.img_rows(img, sl)), timesteps[sl], timesteps.dtype == torch.bfloat16, o)
                out[sl].copy_(lay.img_rows_back(o, sl))
                continue
            ws = eng.workspace(T, N, 1, bs)
            eng.prepare_sample(ws, bf(lay.txt_rows(txt, sl)), bf(y[sl]), None if guidance is None else guidance[sl], gbf16,
                               lay.img_rows(img_ids, sl), lay.txt_rows(txt_ids, sl), timesteps[sl].float().reshape(1, bs),
                               lay.kv_len(sl), timesteps_is_bf16=timesteps.dtype == torch.bfloat16, kv_gap=lay.kv_gap(sl))
            ws.XIN.copy_(bf(lay.img_rows(img, sl)).reshape(bs * N, -1))
            eng.eval_once(ws, None, euler=False, concat=False)
            out[sl].copy_(lay.img_rows_back(ws.V.reshape(bs, N, -1), sl))
        return out.to(img.dtype) if img.dtype.is_floating_point else out


class FluxLoraWrapper(Flux):
    """models/model.py:154-175: Flux with a LoRA pair on EVERY Linear (`replace_linear_with_lora`)."""

    def __init__(self, lora_rank: int = 128, lora_scale: float = 1.0, *args, **kwargs) -> None:
        super().__init__(*args, **kwargs)
        self.lora_rank = lora_rank
        for _, m in list(self._linears()):
            m.add_lora(lora_rank, lora_scale)

    def set_lora_scale(self, scale: float) -> None:
        for _, m in self._linears():
            m.set_scale(scale=scale)
