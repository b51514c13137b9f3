"""Launch plan of one Flux evaluation on MI355X: the sequence of HIP kernels behind `Flux.forward`
(models/model.py:85-124) and, for the fused sampler, the per-sample precomputation + hipGraph of one
Euler step (transport/integrators.py:106-120).

Python here only ORDERS launches (once, under stream capture, for the sampler path); all arithmetic runs in
libvcloze_hip.so.  One sample (B=1) at a time: the grid's samples are independent, so a batch is a loop.

HBM layout per geometry (T text tokens, N image tokens, L = T+N, D hidden, H heads):
  X    [L, D]        residual stream, text rows first (the reference's cat((txt, img), 1) order)
  XH   [L, D]        LayerNorm+modulate output (GEMM A operand)
  QKV  [L, 3D]       "B L (K H D)" rows; q,k are QK-normed + RoPE'd in place
  VT   [H, 128, Lp]  V transposed per head, Lp = L rounded up to 64 (attention B operand is key-contiguous)
  CAT  [L, D+4D]     attn | gelu(mlp) — linear2's input (SingleStreamBlock), attn part doubles as the
                     DoubleStreamBlock attention output;  HID [L, 4D] = CAT[:, D:] view for the MLP hidden
  MOD  [S, n_mod]    every modulation vector of every block for every solver step s (one GEMM at prepare
                     time: the modulations depend on (t, guidance, y) only, never on x)
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, List, Optional

import torch

from . import hip


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


class Workspace:
    def __init__(self, T: int, N: int, D: int, H: int, mlp: int, in_ch: int, out_ch: int, n_mod: int,
                 steps: int, dev: torch.device):
        bf = dict(dtype=torch.bfloat16, device=dev)
        L = T + N
        self.T, self.N, self.L, self.steps = T, N, L, steps
        self.Lp = (L + 63) // 64 * 64
        self.X = torch.empty(L, D, **bf)
        self.XH = torch.empty(L, D, **bf)
        self.QKV = torch.empty(L, 3 * D, **bf)
        self.VT = torch.zeros(H, 128, self.Lp, **bf)
        self.CAT = torch.empty(L, D + mlp, **bf)
        self.TXT0 = torch.empty(T, D, **bf)
        self.XIN = torch.empty(N, in_ch, **bf)
        self.V = torch.empty(N, out_ch, **bf)
        self.XS = torch.empty(N, out_ch, **bf)          # ODE state
        self.COND = torch.empty(N, in_ch - out_ch, **bf)
        self.MOD = torch.empty(steps, n_mod, **bf)
        self.TEMB = torch.empty(steps, 256, **bf)
        self.H1 = torch.empty(steps, D, **bf)
        self.TVEC = torch.empty(steps, D, **bf)
        self.GVEC = torch.empty(1, D, **bf)
        self.YVEC = torch.empty(1, D, **bf)
        self.VEC = torch.empty(steps, D, **bf)
        self.ROPE = torch.empty(L, 64, 2, dtype=torch.float32, device=dev)
        self.TS = torch.empty(steps, dtype=torch.float32, device=dev)
        self.DTS = torch.zeros(steps, dtype=torch.float32, device=dev)
        self.STEP = torch.zeros(1, dtype=torch.int32, device=dev)
        self.KVLEN = torch.full((1,), L, dtype=torch.int32, device=dev)
        self.kv_len = L
        self.graph: Optional[hip.Graph] = None
        self.graph_key = None


class FluxEngine:
    def __init__(self, geom, weights: PreparedWeights, dev: torch.device):
        self.g = geom
        self.W = weights
        self.dev = dev
        self.D = geom.hidden_size
        self.H = geom.num_heads
        self.mlp = int(geom.hidden_size * geom.mlp_ratio)
        self._ws: Dict[tuple, Workspace] = {}
        self.attn_variant = 1
        self.tile_cfg = 0
        self.stream = torch.cuda.Stream(device=dev)   # capture needs a non-default stream

    # ------------------------------------------------------------------ helpers
    def workspace(self, T: int, N: int, steps: int) -> Workspace:
        key = (T, N, steps)
        ws = self._ws.get(key)
        if ws is None:
            if len(self._ws) > 8:
                self._ws.clear()
            ws = Workspace(T, N, self.D, self.H, self.mlp, self.g.in_channels, self.g.out_channels, self.W.n_mod,
                           steps, self.dev)
            self._ws[key] = ws
        return ws

    def _lin(self, name, a, out, epi=hip.EPI_BIAS, res=None, gate=None, step_ptr=None, gate_step_stride=0, s=None):
        p = hip.make_problem(a, self.W.w[name], self.W.b[name], out, res=res, gate=gate)
        hip.gemm(p, epi=epi, tile_cfg=self.tile_cfg, step_ptr=step_ptr, gate_step_stride=gate_step_stride, stream=s)

    def _lin2(self, names, As, outs, epi=hip.EPI_BIAS, ress=(None, None), gates=(None, None), step_ptr=None,
              gate_step_stride=0, s=None):
        ps = [hip.make_problem(a, self.W.w[n], self.W.b[n], o, res=r, gate=g)
              for n, a, o, r, g in zip(names, As, outs, ress, gates)]
        hip.gemm(ps, epi=epi, tile_cfg=self.tile_cfg, step_ptr=step_ptr, gate_step_stride=gate_step_stride, stream=s)

    def _mod(self, ws: Workspace, name: str, idx: int) -> torch.Tensor:
        """row 0 of the idx-th D-wide chunk of module `name`'s modulation output (step offset added in-kernel)"""
        o = self.W.mod_off[name] + idx * self.D
        return ws.MOD[0, o:o + self.D]

    def step_graph(self, ws: Workspace, s: int) -> hip.Graph:
        """hipGraph of ONE solver step (Flux evaluation + Euler update + device step-counter increment).
        Everything step-dependent (modulation rows, dt) is indexed on the device by ws.STEP, so the same
        graph replays for every step of every sample with this geometry."""
        key = (ws.kv_len < ws.L, self.attn_variant, self.tile_cfg)
        if ws.graph is None or ws.graph_key != key:
            xs = ws.XS.clone()
            self.eval_once(ws, ws.STEP, euler=True, s=s)      # warm-up: sets func attributes outside capture
            with hip.Graph(s) as g:
                self.eval_once(ws, ws.STEP, euler=True, s=s)  # recorded, not executed
            ws.XS.copy_(xs)
            ws.STEP.zero_()
            ws.graph, ws.graph_key = g, key
        return ws.graph

    # ------------------------------------------------------------------ per-sample precompute
    def prepare_sample(self, ws: Workspace, txt, y, guidance, guidance_is_bf16: bool, img_ids, txt_ids,
                       timesteps: torch.Tensor, kv_len: int, s=None) -> None:
        """Everything that does not depend on x: txt_in(txt), the vec path and all modulations for every
        step in `timesteps` (model.py:102-108 + layers.py:120-126 for all 57+1 modules), the RoPE table."""
        W, D = self.W, self.D
        S = timesteps.numel()
        assert S == ws.steps
        ws.TS.copy_(timesteps.to(torch.float32), non_blocking=True)
        ws.KVLEN.fill_(kv_len)
        ws.kv_len = kv_len
        # RoPE angles in float64 on the host exactly as math.py:102-109, stored as (cos, sin) f32
        ids = torch.cat((txt_ids.reshape(-1, 3), img_ids.reshape(-1, 3)), dim=0).to("cpu", torch.float64)
        cs = []
        for i, d in enumerate(self.g.axes_dim):
            omega = 1.0 / (self.g.theta ** (torch.arange(0, d, 2, dtype=torch.float64) / d))
            ang = ids[:, i:i + 1] * omega
            cs.append(torch.stack([torch.cos(ang), torch.sin(ang)], dim=-1))
        ws.ROPE.copy_(torch.cat(cs, dim=1).float(), non_blocking=True)
        # txt_in
        self._lin("txt_in", txt, ws.TXT0, s=s)
        # vec path
        hip.timestep_embedding(ws.TS, W.temb_freqs, ws.TEMB, stream=s)
        self._lin("time_in.in_layer", ws.TEMB, ws.H1, epi=hip.EPI_SILU, s=s)
        self._lin("time_in.out_layer", ws.H1, ws.TVEC, s=s)
        if self.g.guidance_embed:
            if guidance is None:
                raise ValueError("Didn't get guidance strength for guidance distilled model.")
            g32 = guidance.reshape(1).to(self.dev, torch.float32)
            ge = torch.empty(1, 256, dtype=torch.bfloat16, device=self.dev)
            gh = torch.empty(1, D, dtype=torch.bfloat16, device=self.dev)
            hip.timestep_embedding(g32, W.temb_freqs, ge, round_t_bf16=guidance_is_bf16, stream=s)
            self._lin("guidance_in.in_layer", ge, gh, epi=hip.EPI_SILU, s=s)
            self._lin("guidance_in.out_layer", gh, ws.GVEC, s=s)
        yh = torch.empty(1, D, dtype=torch.bfloat16, device=self.dev)
        self._lin("vector_in.in_layer", y.reshape(1, -1), yh, epi=hip.EPI_SILU, s=s)
        self._lin("vector_in.out_layer", yh, ws.YVEC, s=s)
        if self.g.guidance_embed:
            hip.add3(ws.TVEC, ws.GVEC, ws.YVEC, out=ws.VEC, stream=s)
        else:
            hip.add3(ws.TVEC, ws.YVEC, None, out=ws.VEC, stream=s)
        hip.silu(ws.VEC, out=ws.H1, stream=s)
        p = hip.make_problem(ws.H1, W.mod_w, W.mod_b, ws.MOD)
        hip.gemm(p, epi=hip.EPI_BIAS, tile_cfg=self.tile_cfg, stream=s)

    # ------------------------------------------------------------------ one evaluation
    def eval_once(self, ws: Workspace, step_ptr, euler: bool, s=None, taps: Optional[dict] = None,
                  concat: bool = True) -> None:
        """Flux.forward on ws.XS || ws.COND (or a caller-filled ws.XIN when not `concat`) -> ws.V, plus the
        Euler update of ws.XS when `euler`."""
        D, H, T, N, L = self.D, self.H, ws.T, ws.N, ws.L
        mss = self.W.n_mod  # MOD row stride = step stride
        X, XH, QKV, CAT = ws.X, ws.XH, ws.QKV, ws.CAT
        Xt, Xi = X[:T], X[T:]
        ATT = CAT[:, :D]
        HID = CAT[:, D:]
        kvl = ws.KVLEN if ws.kv_len < L else None

        def tap(name, t):
            if taps is not None:
                torch.cuda.synchronize()
                taps[name] = t.float().cpu().clone()

        if concat:
            hip.concat_cols(ws.XS, ws.COND, ws.XIN, stream=s)
        hip.copy(Xt, ws.TXT0, stream=s)
        self._lin("img_in", ws.XIN, Xi, s=s)
        tap("img_in", Xi); tap("txt_in", Xt)

        def ln(x, name, idx, out):
            hip.ln_modulate(x, self._mod(ws, name, idx), self._mod(ws, name, idx + 1), out=out, step_ptr=step_ptr,
                            mod_step_stride=mss, stream=s)

        def attn(qkv_view, norm_prefix, out_view):
            hip.qknorm_rope_vt(qkv_view, self.W.w[norm_prefix + ".query_norm.scale"],
                               self.W.w[norm_prefix + ".key_norm.scale"], ws.ROPE, ws.VT, L, H, stream=s)
            hip.attention(qkv_view, ws.VT, out_view, L, H, kv_len=kvl, variant=self.attn_variant, stream=s)

        for i in range(self.g.depth):
            pf = f"double_blocks.{i}"
            im, tm = pf + ".img_mod.lin", pf + ".txt_mod.lin"
            ln(Xi, im, 0, XH[T:]); ln(Xt, tm, 0, XH[:T])
            self._lin2((pf + ".img_attn.qkv", pf + ".txt_attn.qkv"), (XH[T:], XH[:T]), (QKV[T:], QKV[:T]), s=s)
            Wn = self.W.w
            hip.qknorm_rope_vt(QKV, Wn[pf + ".txt_attn.norm.query_norm.scale"], Wn[pf + ".txt_attn.norm.key_norm.scale"],
                               ws.ROPE, ws.VT, L, H, stream=s, q_scale2=Wn[pf + ".img_attn.norm.query_norm.scale"],
                               k_scale2=Wn[pf + ".img_attn.norm.key_norm.scale"], split=T)
            hip.attention(QKV, ws.VT, ATT, L, H, kv_len=kvl, variant=self.attn_variant, stream=s)
            self._lin2((pf + ".img_attn.proj", pf + ".txt_attn.proj"), (ATT[T:], ATT[:T]), (Xi, Xt),
                       epi=hip.EPI_GATE_RES, ress=(Xi, Xt), gates=(self._mod(ws, im, 2), self._mod(ws, tm, 2)),
                       step_ptr=step_ptr, gate_step_stride=mss, s=s)
            ln(Xi, im, 3, XH[T:]); ln(Xt, tm, 3, XH[:T])
            self._lin2((pf + ".img_mlp.0", pf + ".txt_mlp.0"), (XH[T:], XH[:T]), (HID[T:], HID[:T]), epi=hip.EPI_GELU, s=s)
            self._lin2((pf + ".img_mlp.2", pf + ".txt_mlp.2"), (HID[T:], HID[:T]), (Xi, Xt), epi=hip.EPI_GATE_RES,
                       ress=(Xi, Xt), gates=(self._mod(ws, im, 5), self._mod(ws, tm, 5)), step_ptr=step_ptr,
                       gate_step_stride=mss, s=s)
            tap(f"double.{i}.img", Xi); tap(f"double.{i}.txt", Xt)

        for i in range(self.g.depth_single_blocks):
            pf = f"single_blocks.{i}"
            mn = pf + ".modulation.lin"
            ln(X, mn, 0, XH)
            self._lin(pf + ".linear1.qkv", XH, QKV, s=s)
            self._lin(pf + ".linear1.mlp", XH, HID, epi=hip.EPI_GELU, s=s)
            attn(QKV, pf + ".norm", ATT)
            self._lin(pf + ".linear2", CAT, X, epi=hip.EPI_GATE_RES, res=X, gate=self._mod(ws, mn, 2),
                      step_ptr=step_ptr, gate_step_stride=mss, s=s)
            tap(f"single.{i}", X)

        fm = "final_layer.adaLN_modulation.1"
        ln(Xi, fm, 0, XH[T:])
        self._lin("final_layer.linear", XH[T:], ws.V, s=s)
        if euler:
            hip.euler_step(ws.XS, ws.V, ws.DTS, step_ptr, stream=s)
            hip.step_advance(step_ptr, stream=s)
