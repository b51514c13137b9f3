"""Launch plan of one Flux evaluation on MI355X: the sequence of HIP kernels behind `Flux.forward`
(models/model.py:85-124) and, for the fused sampler, the per-sample precomputation + hipGraph of one
Euler step (transport/integrators.py:106-120).

This is the Python-ordered twin of `csrc/flux_engine.hip` (the plan behind the handle API `vc_flux_*`, which is what
`Flux.forward`, the fused sampler and bench.py run): same kernels, same order, same operands - bit-identical results
(tests/test_handle_gpu.py).  It stays for the un-merged LoRA parity mode (`lora_mode="ref"`), per-block taps and A/B runs.

Python here only ORDERS launches (once, under stream capture, for the sampler path); all arithmetic runs in
libvcloze_hip.so.  A per-GPU batch of B samples with the same (T, N) runs as ONE launch sequence: GEMM rows of all
samples are stacked (M = B*rows), modulation vectors / gates / RoPE tables / kv_len are indexed per sample
inside the kernels, attention gets a batch grid dimension.  Ragged batches = per-sample kv_len (+ one masked gap per
sample for masks with holes, model.MaskLayout).

HBM layout per geometry (B samples, T text tokens, N image tokens, L = T+N, D hidden, H heads), bf16 unless noted:
  XI   [B*N, D]      image residual stream of the DoubleStream blocks (samples stacked)
  XT   [B*T, D]      text residual stream of the DoubleStream blocks
  X    [B*L, D]      joint residual stream of the SingleStream blocks: per sample text rows first, then image rows
                     (= the reference's cat((txt, img), 1)); filled from XT/XI once per evaluation
  XH   [B*L, D]      LayerNorm+modulate output (GEMM A operand); XH[:B*N] / XH[B*N:] serve the two streams
  QKV  [B*L, 3D]     "B L (K H D)" rows in joint order; q and k arrive QK-normed + RoPE'd from the qkv GEMM's epilogue (q times
                     the softmax scale) where the one-wave-per-SIMD attention runs; the V third is not materialised (fuse_vt)
  VT   [B, H, 128, Lp]  V transposed per head, Lp = L rounded up to 64 (attention B operand is key-contiguous)
  CAT  [B*L, D+4D]   attn | gelu(mlp) = linear2's input (SingleStreamBlock); CAT[:, :D] is also the DoubleStream
                     attention output (joint order), whose rows the proj GEMMs read batch-strided
  HID  [B*L, 4D]     MLP hidden of the DoubleStream blocks (image rows first, then text rows)
  MOD  [S*B, n_mod]  every modulation vector of every block for every solver step s and sample b (row s*B+b;
                     one GEMM at prepare time: the modulations depend on (t, guidance, y) only, never on x)
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, Optional, Sequence, Tuple

import torch

from . import hip


@dataclass
class RefLinear:
    """One Linear of the UN-MERGED LoRA execution mode (LinearLora.forward, models/modules/lora.py:92-98):
    y = bf16(x W^T + b);  h = bf16(x A^T);  u = bf16(h B^T + b_B);  out = bf16(y + bf16(u * scale)).
    A / B are zero-padded along the rank to the GEMM's K granularity (64)."""
    w: torch.Tensor
    b: Optional[torch.Tensor]
    A: Optional[torch.Tensor]          # [rK, in]   (None: no LoRA pair on this Linear)
    B: Optional[torch.Tensor]          # [out, rK]
    bB: Optional[torch.Tensor]
    scale: Optional[torch.Tensor]      # [out] bf16, every element = lora scale


@dataclass
class LinCall:
    """A Linear to run: module path + operand views (+ the batch-strided row description of hip.make_problem)."""
    name: str
    a: torch.Tensor
    out: torch.Tensor
    kw: dict


@dataclass
class PreparedWeights:
    """bf16, contiguous, LoRA-merged device tensors keyed by reference module path."""
    w: Dict[str, torch.Tensor]
    b: Dict[str, Optional[torch.Tensor]]
    mod_w: torch.Tensor          # [n_mod, D] all modulation Linear weights stacked
    mod_b: torch.Tensor          # [n_mod]
    mod_off: Dict[str, int]      # module path -> column offset in MOD
    n_mod: int
    temb_freqs: torch.Tensor     # [128] f32
    ref: Optional[Dict[str, RefLinear]] = None    # lora_mode="ref": un-merged factors of every Linear (incl. the modulation ones)
    qkv_heads: int = 0           # H > 0: every qkv weight / bias (linear1's first 3D rows) is HEAD-PERMUTED (hip.qkv_head_permutation)
    logit_bound: float = 0.0     # 16.33 * max|query_norm.scale| * max|key_norm.scale| over all blocks: |q.k| 128^-0.5 log2(e) <= this
    logit_bounds: Optional[Dict[str, float]] = None      # the same per block ("double_blocks.3", "single_blocks.17"): what each launch gets


class Workspace:
    def __init__(self, T: int, N: int, D: int, H: int, mlp: int, in_ch: int, out_ch: int, n_mod: int,
                 steps: int, B: int, dev: torch.device):
        bf = dict(dtype=torch.bfloat16, device=dev)
        L = T + N
        self.T, self.N, self.L, self.steps, self.B = T, N, L, steps, B
        self.Lp = (L + 63) // 64 * 64
        self.XI = torch.empty(B * N, D, **bf)
        self.XT = torch.empty(B * T, D, **bf)
        self.X = torch.empty(B * L, D, **bf)
        self.XH = torch.empty(B * L, D, **bf)
        self.QKV = torch.empty(B * L, 3 * D, **bf)
        self.VT = torch.zeros(B, H, 128, self.Lp, **bf)
        self.CAT = torch.empty(B * L, D + mlp, **bf)
        self.HID = torch.empty(B * L, mlp, **bf)
        self.TXT0 = torch.empty(B * T, D, **bf)
        self.XIN = torch.empty(B * N, in_ch, **bf)
        self.V = torch.empty(B * N, out_ch, **bf)
        self.XS = torch.empty(B * N, out_ch, **bf)          # ODE state
        self.COND = torch.empty(B * N, in_ch - out_ch, **bf)
        self.MOD = torch.empty(steps * B, n_mod, **bf)
        self.TEMB = torch.empty(steps * B, 256, **bf)
        self.H1 = torch.empty(steps * B, D, **bf)
        self.TVEC = torch.empty(steps * B, D, **bf)
        self.GVEC = torch.empty(B, D, **bf)
        self.YVEC = torch.empty(B, D, **bf)
        self.VEC = torch.empty(steps * B, D, **bf)
        self.ROPE = torch.empty(B, L, 64, 2, dtype=torch.float32, device=dev)
        self.TS = torch.empty(steps * B, dtype=torch.float32, device=dev)
        self.DTS = torch.zeros(steps, dtype=torch.float32, device=dev)
        self.STEP = torch.zeros(1, dtype=torch.int32, device=dev)
        self.KVLEN = torch.full((B,), L, dtype=torch.int32, device=dev)
        self.KVGAP = torch.zeros(B, 2, dtype=torch.int32, device=dev)   # masked tail of the text stream, joint order
        self.ragged = False
        self.gapped = False
        self.graph: Optional[hip.Graph] = None
        self.graph_key = None


class FluxEngine:
    MAX_BATCH = 4   # samples per launch sequence; larger batches run in chunks (the GPU is full at B = 1-2 for L ~ 4000)

    def __init__(self, geom, weights: PreparedWeights, dev: torch.device):
        self.g = geom
        self.W = weights
        self.dev = dev
        self.D = geom.hidden_size
        self.H = geom.num_heads
        self.mlp = int(geom.hidden_size * geom.mlp_ratio)
        self._ws: Dict[tuple, Workspace] = {}
        # hip.attention variant; None = by size: the one-wave-per-SIMD kernel (12: 4 waves x 64 queries per work item, tail
        # items cut along the keys) once there is at least one 256-query item per CU, the 32-queries-per-wave kernel (3)
        # below that (measured: L = 1664 -> 52 vs 60 us; L = 3968 -> 196 vs 183; 6656 -> 495 vs 455; 7424 -> 679 vs 581)
        self.attn_variant = None
        self.n_cu = hip.device_cus(dev)
        self.attn_scratch = hip.attention_scratch(dev)
        # query QKNorm + RoPE: 2 = in the qkv GEMM's epilogue, with the softmax scale (where the key heads are normalised there
        # too); 1 = inside the attention kernel (variants 8 / 12), on the rows a wave loads; 0 = by the pre-pass
        self.fuse_qnorm = 2
        self.fuse_vt = weights.ref is None   # V^T written by the qkv GEMM's epilogue (EPI_QKV); the pre-pass is then K only
        # QKNorm + RoPE of the key heads inside the qkv GEMM's epilogue (head-permuted weights, VcGemmProblem.kn_scale): with
        # fuse_qnorm and fuse_vt no pre-pass kernel is left between the projection and the attention
        self.fuse_knorm = weights.qkv_heads > 0
        # the attention kernel may drop its running max when the model's QK-norm scales bound the logits (VcAttention.logit_bound)
        self.bounded_softmax = True
        # SingleStreamBlock launch order: qkv GEMM -> attention -> MLP-up GEMM -> linear2 (operands consumed right after they
        # are produced: the attention finds q / k / V^T where the projection left them, linear2 its 97 MB of gelu(mlp)); True =
        # MLP-up before the attention, the order of layers.py:236-243
        self.mlp_first = False
        self.tile_cfg = 0
        # f32 scratch for split-K GEMM remainders (VcGemmArgs.splitk_ws): where the 256x192 tiles are whole rounds of the CUs plus
        # a few, the launcher may cut those few along K (SDEdit stage, cfg 1); False = never
        self.splitk = True
        self.splitk_ws = hip.splitk_workspace(dev)
        self.stream = torch.cuda.Stream(device=dev)   # capture needs a non-default stream
        self._ref_scratch: Dict[tuple, torch.Tensor] = {}
        if weights.ref is not None:
            self.MAX_BATCH = 1     # the un-merged mode addresses plain rows (no batch-strided views)

    # ------------------------------------------------------------------ helpers
    def workspace(self, T: int, N: int, steps: int, B: int = 1) -> Workspace:
        key = (T, N, steps, B)
        ws = self._ws.get(key)
        if ws is None:
            if len(self._ws) > 8:
                self._ws.clear()
            ws = Workspace(T, N, self.D, self.H, self.mlp, self.g.in_channels, self.g.out_channels, self.W.n_mod,
                           steps, B, self.dev)
            self._ws[key] = ws
        return ws

    def _gemm(self, calls, epi=hip.EPI_BIAS, step_ptr=None, gate_step_stride=0, s=None):
        """Run one or several Linears (grouped into ONE launch in the merged mode) with the fused epilogue `epi`."""
        if isinstance(calls, LinCall):
            calls = [calls]
        if self.W.ref is not None:
            for c in calls:
                self._linear_ref(c, epi, step_ptr, gate_step_stride, s)
            return
        probs = [hip.make_problem(c.a, self.W.w[c.name], self.W.b[c.name], c.out, **c.kw) for c in calls]
        hip.gemm(probs, epi=epi, tile_cfg=self.tile_cfg, step_ptr=step_ptr, gate_step_stride=gate_step_stride, stream=s,
                 splitk_ws=self.splitk_ws if self.splitk else None)

    def _prob(self, name, a, out, **kw) -> LinCall:
        return LinCall(name, a, out, kw)

    def _scratch(self, tag, rows, cols):
        key = (tag, rows, cols)
        t = self._ref_scratch.get(key)
        if t is None:
            t = torch.empty(rows, cols, dtype=torch.bfloat16, device=self.dev)
            self._ref_scratch[key] = t
        return t

    def _linear_ref(self, c: LinCall, epi, step_ptr, gate_step_stride, s) -> None:
        """LinearLora.forward with the reference's three roundings (lora.py:92-98), then the epilogue as its own pass.
        A parity / debugging mode (3 GEMMs + 1 pass per Linear, +8 % FLOPs); one sample per launch sequence."""
        R = self.W.ref[c.name]
        kw = c.kw
        M = kw.get("M") or c.a.shape[0]
        a, out = c.a[:M], c.out[:M]
        N = R.w.shape[0]
        plain = epi == hip.EPI_BIAS
        y = out if (plain and R.A is None) else self._scratch("y", M, N)
        hip.gemm(hip.make_problem(a, R.w, R.b, y), tile_cfg=self.tile_cfg, stream=s)                  # base_out
        if R.A is not None:
            h = self._scratch("h", M, R.A.shape[0])
            hip.gemm(hip.make_problem(a, R.A, None, h), tile_cfg=self.tile_cfg, stream=s)              # lora_A(x)
            y2 = out if plain else self._scratch("y2", M, N)
            hip.gemm(hip.make_problem(h, R.B, R.bB, y2, res=y, gate=R.scale), epi=hip.EPI_GATE_RES,    # base_out + lora_B(.) * scale
                     tile_cfg=self.tile_cfg, stream=s)
            y = y2
        if epi == hip.EPI_GELU:
            hip.act2d(y, out, "gelu", stream=s)
        elif epi == hip.EPI_SILU:
            hip.act2d(y, out, "silu", stream=s)
        elif epi == hip.EPI_GATE_RES:
            hip.gate_residual(y, kw["res"][:M], kw["gate"], out, step_ptr=step_ptr, gate_step_stride=gate_step_stride, stream=s)

    def _lin(self, name, a, out, epi=hip.EPI_BIAS, s=None, **kw):
        self._gemm(self._prob(name, a, out, **kw), epi=epi, s=s)

    def _mod(self, ws: Workspace, name: str, idx: int) -> torch.Tensor:
        """row (step 0, sample 0) of the idx-th D-wide chunk of module `name`'s modulation output; kernels add
        sample * n_mod and step * B * n_mod themselves"""
        o = self.W.mod_off[name] + idx * self.D
        return ws.MOD[0, o:o + self.D]

    def step_graph(self, ws: Workspace, s: int) -> hip.Graph:
        """hipGraph of ONE solver step (Flux evaluation + Euler update + device step-counter increment).
        Everything step-dependent (modulation rows, dt) is indexed on the device by ws.STEP, so the same
        graph replays for every step of every sample batch with this geometry."""
        key = (ws.ragged, ws.gapped, self.attn_variant, self.tile_cfg, self.fuse_qnorm, self.fuse_vt, self.fuse_knorm, self.bounded_softmax, self.mlp_first, self.splitk)
        if ws.graph is None or ws.graph_key != key:
            xs = ws.XS.clone()
            self.eval_once(ws, ws.STEP, euler=True, s=s)      # warm-up: sets func attributes outside capture
            with hip.Graph(s) as g:
                self.eval_once(ws, ws.STEP, euler=True, s=s)  # recorded, not executed
            ws.XS.copy_(xs)
            ws.STEP.zero_()
            ws.graph, ws.graph_key = g, key
        return ws.graph

    # ------------------------------------------------------------------ per-batch precompute
    def prepare_sample(self, ws: Workspace, txt, y, guidance, guidance_is_bf16: bool, img_ids, txt_ids,
                       timesteps: torch.Tensor, kv_len: Sequence[int], s=None, timesteps_is_bf16: bool = False,
                       kv_gap: Optional[Sequence[Tuple[int, int]]] = None) -> None:
        """Everything that does not depend on x: txt_in(txt), the vec path and all modulations for every
        (step, sample) (model.py:102-108 + layers.py:120-126 for all 57+1 modules), the RoPE tables.
        txt [B,T,ctx], y [B,vec], guidance [B] or None, img_ids [B,N,3], txt_ids [B,T,3], timesteps [S] (shared by
        the batch) or [S,B], kv_len: B ints (<= L); kv_gap: B (lo, hi) pairs, a second masked row range per sample
        (the padded tail of the text stream once `model.mask_layout` has moved every stream's valid rows first)."""
        W, D, B, S = self.W, self.D, ws.B, ws.steps
        ts = timesteps.to(torch.float32).reshape(S, -1)
        ws.TS.copy_((ts.expand(S, B) if ts.shape[1] == 1 else ts).reshape(-1), non_blocking=True)
        kv = [int(v) for v in kv_len]
        ws.KVLEN.copy_(torch.tensor(kv, dtype=torch.int32), non_blocking=True)
        gap = [(0, 0)] * B if kv_gap is None else [(int(lo), int(hi)) for lo, hi in kv_gap]
        if any(not (0 <= lo <= hi <= v) for (lo, hi), v in zip(gap, kv)):
            raise hip.VclozeHipError(f"kv_gap {gap} must lie inside [0, kv_len) = {kv}")
        ws.KVGAP.copy_(torch.tensor(gap, dtype=torch.int32), non_blocking=True)
        ws.gapped = any(hi > lo for lo, hi in gap)
        ws.ragged = ws.gapped or any(v < ws.L for v in kv)
        # RoPE angles in float64 on the host exactly as math.py:102-109, stored as (cos, sin) f32 per sample
        ids = torch.cat((txt_ids.reshape(B, ws.T, 3), img_ids.reshape(B, ws.N, 3)), dim=1).to("cpu", torch.float64)
        cs = []
        for i, d in enumerate(self.g.axes_dim):
            omega = 1.0 / (self.g.theta ** (torch.arange(0, d, 2, dtype=torch.float64) / d))
            ang = ids[..., i:i + 1] * omega
            cs.append(torch.stack([torch.cos(ang), torch.sin(ang)], dim=-1))
        ws.ROPE.copy_(torch.cat(cs, dim=2).float(), non_blocking=True)
        # txt_in
        self._lin("txt_in", txt.reshape(B * ws.T, -1), ws.TXT0, s=s)
        # vec path: rows s*B+b
        # bf16 `timesteps` (not what the sampler passes: integrators.py:108 builds them in f32) make `1000 * t` a bf16
        # product in layers.py:38, exactly as for the bf16 guidance below
        hip.timestep_embedding(ws.TS, W.temb_freqs, ws.TEMB, round_t_bf16=timesteps_is_bf16, stream=s)
        self._lin("time_in.in_layer", ws.TEMB, ws.H1, epi=hip.EPI_SILU, s=s)
        self._lin("time_in.out_layer", ws.H1, ws.TVEC, s=s)
        if self.g.guidance_embed:
            if guidance is None:
                raise ValueError("Didn't get guidance strength for guidance distilled model.")
            g32 = guidance.reshape(-1)
            if g32.numel() == 1:           # the pipeline's guidance is shape (1,) and broadcasts (visualcloze.py:413)
                g32 = g32.expand(B)
            g32 = g32.reshape(B).to(self.dev, torch.float32).contiguous()
            ge = torch.empty(B, 256, dtype=torch.bfloat16, device=self.dev)
            gh = torch.empty(B, D, dtype=torch.bfloat16, device=self.dev)
            hip.timestep_embedding(g32, W.temb_freqs, ge, round_t_bf16=guidance_is_bf16, stream=s)
            self._lin("guidance_in.in_layer", ge, gh, epi=hip.EPI_SILU, s=s)
            self._lin("guidance_in.out_layer", gh, ws.GVEC, s=s)
        yh = torch.empty(B, D, dtype=torch.bfloat16, device=self.dev)
        self._lin("vector_in.in_layer", y.reshape(B, -1), yh, epi=hip.EPI_SILU, s=s)
        self._lin("vector_in.out_layer", yh, ws.YVEC, s=s)
        if self.g.guidance_embed:      # row s*B+b: (time[s] + guidance[b]) + vector[b]; b,c broadcast with period B*D
            hip.add3(ws.TVEC, ws.GVEC, ws.YVEC, out=ws.VEC, stream=s)
        else:
            hip.add3(ws.TVEC, ws.YVEC, None, out=ws.VEC, stream=s)
        hip.silu(ws.VEC, out=ws.H1, stream=s)
        if W.ref is None:
            hip.gemm(hip.make_problem(ws.H1, W.mod_w, W.mod_b, ws.MOD), tile_cfg=self.tile_cfg, stream=s,
                     splitk_ws=self.splitk_ws if self.splitk else None)
        else:                          # un-merged mode: every modulation Linear on its own column range of MOD
            for name, off in W.mod_off.items():
                n = W.ref[name].w.shape[0]
                self._gemm(self._prob(name, ws.H1, ws.MOD[:, off:off + n]), s=s)

    # ------------------------------------------------------------------ one evaluation
    class _Ctx:
        """Per-evaluation launch context: buffer views and the step-indexed modulation addressing."""

    def _ctx(self, ws: Workspace, step_ptr, s):
        c = FluxEngine._Ctx()
        D, T, N, L, B = self.D, ws.T, ws.N, ws.L, ws.B
        c.ws, c.step_ptr, c.s = ws, step_ptr, s
        c.nm = self.W.n_mod
        c.mss = B * c.nm                  # MOD step stride (elements); sample stride is nm
        c.XH_I, c.XH_T = ws.XH[:B * N], ws.XH[B * N:]
        c.HID_I, c.HID_T = ws.HID[:B * N], ws.HID[B * N:]
        c.ATT = ws.CAT[:, :D]
        c.kvl = ws.KVLEN if ws.ragged else None
        c.kvgap = ws.KVGAP if ws.gapped else None
        # joint-order views of the first sample's rows + batch strides (row m of a stream -> sample m // rows)
        c.qkv_i = dict(M=B * N, c_rpb=N, c_bstride=L * ws.QKV.stride(0))
        c.qkv_t = dict(M=B * T, c_rpb=T, c_bstride=L * ws.QKV.stride(0))
        c.att_i = dict(M=B * N, a_rpb=N, a_bstride=L * ws.CAT.stride(0))
        c.att_t = dict(M=B * T, a_rpb=T, a_bstride=L * ws.CAT.stride(0))
        return c

    def _ln(self, c, x, name, idx, out, rpb):
        hip.ln_modulate(x, self._mod(c.ws, name, idx), self._mod(c.ws, name, idx + 1), out=out, step_ptr=c.step_ptr,
                        mod_step_stride=c.mss, stream=c.s, rows_per_batch=rpb, mod_bstride=c.nm)

    def _ln2(self, c, name_i, name_t, idx):    # img + txt streams in one launch
        ws = c.ws
        hip.ln_modulate2([(ws.XI, self._mod(ws, name_i, idx), self._mod(ws, name_i, idx + 1), c.XH_I, ws.N),
                          (ws.XT, self._mod(ws, name_t, idx), self._mod(ws, name_t, idx + 1), c.XH_T, ws.T)],
                         step_ptr=c.step_ptr, mod_step_stride=c.mss, stream=c.s, mod_bstride=c.nm)

    def _gated(self, c, names, As, outs, gates, rpbs, a_views):
        ps = []
        for n_, a_, o_, g_, rpb, av in zip(names, As, outs, gates, rpbs, a_views):
            ps.append(self._prob(n_, a_, o_, res=o_, gate=g_, rows_per_batch=rpb, gate_bstride=c.nm, **av))
        self._gemm(ps, epi=hip.EPI_GATE_RES, step_ptr=c.step_ptr, gate_step_stride=c.mss, s=c.s)

    def attention_variant(self, ws: Workspace) -> int:
        if self.attn_variant is not None:
            return self.attn_variant
        # 28 = 12 + 16: tail pieces combined inside the launch where the stream form runs (hip.attention_scratch is zero-initialised);
        # fewer 256-query items than CUs: the same kernel without a split (8) down to half the CUs (cfg 1), the
        # 32-queries-per-wave kernel (3) below (csrc/flux_engine.hip attention_variant)
        items = ((ws.L + 255) // 256) * self.H * ws.B
        return 28 if items >= self.n_cu else 8 if 2 * items >= self.n_cu else 3

    def _block_bound(self, pf: str) -> float:
        """VcAttention.logit_bound of block `pf`: its own bound, capped by the model's (csrc/flux_engine.hip attention())"""
        if not self.bounded_softmax:
            return 0.0
        cap = float(self.W.logit_bound)
        if not cap > 0:
            return 0.0
        b = (self.W.logit_bounds or {}).get(pf, cap)
        if not b < 1e30:            # non-finite scales: no bound
            return 1e30
        return min(b, cap)

    def _attention(self, c, scales, split, pf=None):
        """QKNorm + RoPE (+ V^T) and the joint attention over ws.QKV -> CAT[:, :D] (layers.py:165-185 / 236-241).  With the
        one-wave-per-SIMD kernel (variants 8 / 12) the query rows are normalised where they are loaded, inside the
        attention kernel, and the pre-pass touches only K and V."""
        ws, s = c.ws, c.s
        q1, k1, q2, k2 = scales
        variant = self.attention_variant(ws)
        fused_q = bool(variant & 8) and bool(self.fuse_qnorm)
        q_done = self._qn_in_gemm(ws)      # by the projection's epilogue, prescaled (hip.make_problem(qn_prescale=True))
        parts = (0 if self._kn_in_gemm(ws) else hip.QKN_K) | (0 if fused_q else hip.QKN_Q) | (0 if self._vt_in_gemm() else hip.QKN_VT)
        if parts:
            hip.qknorm_rope_vt(ws.QKV, q1, k1, ws.ROPE, ws.VT, ws.L, self.H, stream=s, q_scale2=q2, k_scale2=k2, split=split, B=ws.B,
                               parts=parts)
        ev = getattr(self, "attn_events", None)     # bench.py: HIP events around the attention launch(es) IN SITU
        if ev is not None:
            e0 = hip.Event()
            e0.record(s)
        hip.attention(ws.QKV, ws.VT, c.ATT, ws.L, self.H, kv_len=c.kvl, variant=variant, stream=s, B=ws.B,
                      scratch=self.attn_scratch, q_norm=(q1, q2, split, ws.ROPE) if fused_q and not q_done else None, kv_gap=c.kvgap,
                      logit_bound=self._block_bound(pf), q_prescaled=q_done)
        if ev is not None:
            e1 = hip.Event()
            e1.record(s)
            ev.append((e0, e1))

    def _vt_in_gemm(self) -> bool:
        return self.fuse_vt and self.W.ref is None

    def _kn_in_gemm(self, ws: Workspace) -> bool:
        """key QKNorm + RoPE in the qkv GEMM's epilogue: where the one-wave-per-SIMD attention kernel runs (it normalises
        its own queries, so no pre-pass is left).  Small geometries (cfg 1) keep ONE pre-pass launch for q and k: the fused
        epilogue needs the 256x192 tile, which their short M does not fill (+1.2 ms of GEMM time at L = 1664)."""
        return self.fuse_knorm and self.W.qkv_heads > 0 and self.W.ref is None and bool(self.attention_variant(ws) & 8)

    def _qn_in_gemm(self, ws: Workspace) -> bool:
        """query QKNorm + RoPE + the softmax scale in the qkv GEMM's epilogue as well: the attention kernel loads finished rows"""
        return self._kn_in_gemm(ws) and int(self.fuse_qnorm) >= 2

    def _qkv_epi(self, ws: Workspace, rows: int, row0: int, k_scale=None, q_scale=None):
        """(epilogue, problem kwargs) of a qkv projection: with fuse_vt the V third goes straight to ws.VT, transposed; with
        head-permuted weights C receives the logical columns and, with fuse_knorm, the key heads leave normalised + rotated"""
        kw = {}
        if self._vt_in_gemm():
            kw.update(vt=ws.VT, vt_col0=2 * self.D)
        if self.W.qkv_heads and self.W.ref is None:
            kw.update(kn_heads=self.W.qkv_heads)
            if self._kn_in_gemm(ws):
                kw.update(kn_scale=k_scale, kn_rope=ws.ROPE)
            if self._qn_in_gemm(ws):
                kw.update(qn_scale=q_scale, qn_prescale=True)
        if not kw:
            return hip.EPI_BIAS, {}
        kw.update(vt_rpb=rows, vt_row0=row0)
        return hip.EPI_QKV, kw

    def double_block(self, c, i: int) -> None:
        """DoubleStreamBlock i (layers.py:158-196) on ws.XI / ws.XT, in place."""
        ws, s, Wn = c.ws, c.s, self.W.w
        T, N = ws.T, ws.N
        pf = f"double_blocks.{i}"
        im, tm = pf + ".img_mod.lin", pf + ".txt_mod.lin"
        self._ln2(c, im, tm, 0)
        epi, kv_i = self._qkv_epi(ws, N, T, Wn[pf + ".img_attn.norm.key_norm.scale"], Wn[pf + ".img_attn.norm.query_norm.scale"])
        _, kv_t = self._qkv_epi(ws, T, 0, Wn[pf + ".txt_attn.norm.key_norm.scale"], Wn[pf + ".txt_attn.norm.query_norm.scale"])
        self._gemm([self._prob(pf + ".img_attn.qkv", c.XH_I, ws.QKV[T:], **c.qkv_i, **kv_i),
                    self._prob(pf + ".txt_attn.qkv", c.XH_T, ws.QKV[:T], **c.qkv_t, **kv_t)], epi=epi, s=s)
        self._attention(c, (Wn[pf + ".txt_attn.norm.query_norm.scale"], Wn[pf + ".txt_attn.norm.key_norm.scale"],
                            Wn[pf + ".img_attn.norm.query_norm.scale"], Wn[pf + ".img_attn.norm.key_norm.scale"]), T, pf)
        self._gated(c, (pf + ".img_attn.proj", pf + ".txt_attn.proj"), (c.ATT[T:], c.ATT[:T]), (ws.XI, ws.XT),
                    (self._mod(ws, im, 2), self._mod(ws, tm, 2)), (N, T), (c.att_i, c.att_t))
        self._ln2(c, im, tm, 3)
        self._gemm([self._prob(pf + ".img_mlp.0", c.XH_I, c.HID_I), self._prob(pf + ".txt_mlp.0", c.XH_T, c.HID_T)],
                   epi=hip.EPI_GELU, s=s)
        self._gated(c, (pf + ".img_mlp.2", pf + ".txt_mlp.2"), (c.HID_I, c.HID_T), (ws.XI, ws.XT),
                    (self._mod(ws, im, 5), self._mod(ws, tm, 5)), (N, T), ({}, {}))

    def join_streams(self, c) -> None:
        """cat((txt, img), 1) per sample (model.py:116): XT / XI -> X."""
        ws, s = c.ws, c.s
        T, N, L = ws.T, ws.N, ws.L
        for b in range(ws.B):
            hip.copy(ws.X[b * L:b * L + T], ws.XT[b * T:(b + 1) * T], stream=s)
            hip.copy(ws.X[b * L + T:(b + 1) * L], ws.XI[b * N:(b + 1) * N], stream=s)

    def single_block(self, c, i: int) -> None:
        """SingleStreamBlock i (layers.py:232-245) on ws.X, in place."""
        ws, s, Wn, D = c.ws, c.s, self.W.w, self.D
        pf = f"single_blocks.{i}"
        mn = pf + ".modulation.lin"
        self._ln(c, ws.X, mn, 0, ws.XH, ws.L)
        epi, kv = self._qkv_epi(ws, ws.L, 0, Wn[pf + ".norm.key_norm.scale"], Wn[pf + ".norm.query_norm.scale"])
        self._lin(pf + ".linear1.qkv", ws.XH, ws.QKV, epi=epi, s=s, **kv)
        if self.mlp_first:      # (the reference's textual order; the product runs the attention right behind its projection)
            self._lin(pf + ".linear1.mlp", ws.XH, ws.CAT[:, D:], epi=hip.EPI_GELU, s=s)
        self._attention(c, (Wn[pf + ".norm.query_norm.scale"], Wn[pf + ".norm.key_norm.scale"], None, None), 0, pf)
        if not self.mlp_first:
            self._lin(pf + ".linear1.mlp", ws.XH, ws.CAT[:, D:], epi=hip.EPI_GELU, s=s)
        self._gated(c, (pf + ".linear2",), (ws.CAT,), (ws.X,), (self._mod(ws, mn, 2),), (ws.L,), ({},))

    def last_layer(self, c) -> None:
        """LastLayer (layers.py:248-259) on the image rows of ws.X -> ws.V."""
        ws, s = c.ws, c.s
        fm = "final_layer.adaLN_modulation.1"
        self._ln(c, ws.X, fm, 0, ws.XH, ws.L)   # text rows are normalised too (13 % of a 10 us kernel) and then skipped
        self._lin("final_layer.linear", ws.XH[ws.T:], ws.V, s=s, M=ws.B * ws.N, a_rpb=ws.N, a_bstride=ws.L * ws.XH.stride(0))

    def eval_once(self, ws: Workspace, step_ptr, euler: bool, s=None, taps: Optional[dict] = None,
                  concat: bool = True) -> None:
        """Flux.forward on ws.XS || ws.COND (or a caller-filled ws.XIN when not `concat`) -> ws.V, plus the
        Euler update of ws.XS when `euler`."""
        c = self._ctx(ws, step_ptr, s)

        def tap(name, t):
            if taps is not None:
                torch.cuda.synchronize()
                taps[name] = t.float().cpu().clone()

        if concat:
            hip.concat_cols(ws.XS, ws.COND, ws.XIN, stream=s)
        hip.copy(ws.XT, ws.TXT0, stream=s)
        self._lin("img_in", ws.XIN, ws.XI, s=s)
        tap("img_in", ws.XI); tap("txt_in", ws.XT)
        for i in range(self.g.depth):
            self.double_block(c, i)
            tap(f"double.{i}.img", ws.XI); tap(f"double.{i}.txt", ws.XT)
        self.join_streams(c)
        for i in range(self.g.depth_single_blocks):
            self.single_block(c, i)
            tap(f"single.{i}", ws.X)
        self.last_layer(c)
        if euler:
            hip.euler_step(ws.XS, ws.V, ws.DTS, step_ptr, stream=s)
            hip.step_advance(step_ptr, stream=s)
