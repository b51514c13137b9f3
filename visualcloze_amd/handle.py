"""`FluxHandle`: the handle API of include/vcloze_hip.h (vc_flux_*) behind torch tensors — Flux.forward and the whole
fixed-grid Euler loop as ONE C call each.  The launch plan lives in csrc/flux_engine.hip; this class
binds the prepared (bf16, LoRA-merged) weights by reference-module path, owns the workspace tensors and converts the
host-side inputs (ids, timesteps, masks) to the plain arrays the ABI takes.  `engine.FluxEngine` is the same plan spelt
in Python over the op-level ABI; it stays for the un-merged LoRA parity mode and for per-block taps."""
from __future__ import annotations

import ctypes as C
from typing import Dict, Optional, Sequence, Tuple

import numpy as np
import torch

from . import hip


def _f32(a) -> np.ndarray:
    return np.ascontiguousarray(a.detach().to("cpu", torch.float32).numpy() if torch.is_tensor(a) else np.asarray(a, np.float32),
                                dtype=np.float32)


class _HostCopies:
    """Host copies of small DEVICE inputs (ids, guidance) that vc_flux_prepare takes as host arrays, remembered per argument
    while the caller keeps handing over the SAME MEMORY at the same version: a D2H copy synchronises the stream, i.e. drains
    every queued solver step - once per grid that is nothing, once per 3-evaluation sample (cfg 1) it left the GPU idle for
    2 - 6 % of the run.  The key is the memory, not the Python object: `Sampler.sample_ode` / `Flux.forward` slice their
    arguments per chunk (`MaskLayout._take`), which makes a NEW view object of the same storage on every call (advisor r04: keyed
    on identity the cache hit in bench.py only).  (storage address, offset, shape, strides, dtype, version counter - views share
    their base's); the entry holds the tensor, so the storage cannot be freed and its address recycled while the entry lives.
    These inputs are never written by the library's kernels (which would not bump the version)."""

    def __init__(self):
        self._c: Dict[str, tuple] = {}
        self.hits = self.misses = 0

    @staticmethod
    def _key(t) -> tuple:
        return (t.untyped_storage().data_ptr(), t.storage_offset(), tuple(t.shape), tuple(t.stride()), t.dtype, t._version)

    def f32(self, key: str, t) -> np.ndarray:
        if not torch.is_tensor(t) or not t.is_cuda:
            return _f32(t)
        k = self._key(t)
        e = self._c.get(key)
        if e is not None and e[0] == k:
            self.hits += 1
            return e[2]
        self.misses += 1
        a = _f32(t)
        self._c[key] = (k, t, a)
        return a


def _fp(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(C.POINTER(C.c_float))


def _ip(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(C.POINTER(C.c_int32))


class FluxHandle:
    MAX_BATCH = 4      # samples per launch sequence, as engine.FluxEngine

    def __init__(self, params, weights, dev: torch.device):
        self.params, self.dev, self.W = params, dev, weights
        D = params.hidden_size
        cfg = hip.FluxConfig(params.in_channels, params.out_channels, params.vec_in_dim, params.context_in_dim, D, params.num_heads,
                             params.depth, params.depth_single_blocks, int(D * params.mlp_ratio), int(bool(params.guidance_embed)),
                             (C.c_int32 * 3)(*params.axes_dim), int(params.theta))
        self.h = C.c_void_p()
        with torch.cuda.device(dev):
            hip._check(hip.lib().vc_flux_create(C.byref(cfg), C.byref(self.h)), "vc_flux_create")
        L = hip.lib()
        for name, off in weights.mod_off.items():          # the stacking order is part of the ABI
            if L.vc_flux_mod_offset(self.h, name.encode()) != off:
                raise hip.VclozeHipError(f"modulation row offset of {name} differs between model.prepare and libvcloze_hip.so")
        if L.vc_flux_mod_offset(self.h, None) != weights.n_mod:
            raise hip.VclozeHipError("stacked modulation size differs between model.prepare and libvcloze_hip.so")
        for name, w in weights.w.items():
            if name.endswith(".linear1.qkv") or name.endswith(".linear1.mlp"):
                continue                                    # row ranges of linear1, which is bound whole
            b = weights.b.get(name)
            rows, cols = (1, w.numel()) if w.dim() == 1 else tuple(w.shape)
            self._bind(name, w, b, rows, cols)
        self._bind("modulation", weights.mod_w, weights.mod_b, *weights.mod_w.shape)
        hip._check(L.vc_flux_bind_weight(self.h, b"timestep_freqs", weights.temb_freqs.data_ptr(), None, 1, 128, 128),
                   "vc_flux_bind_weight(timestep_freqs)")      # torch's own f32 table: bit-equal to the Python-ordered plan
        # ONE split-K scratch for every geometry of this handle (its launches are ordered on one stream), instead of 100 MB carved
        # into each of the up to nine cached workspaces (advisor r04)
        self._sk_ws = hip.splitk_workspace(dev)
        nf = self._sk_ws.numel() // 4
        hip._check(L.vc_flux_bind_weight(self.h, b"splitk_ws", self._sk_ws.data_ptr(), None, 1, nf, nf), "vc_flux_bind_weight(splitk_ws)")
        self._ws: Dict[tuple, torch.Tensor] = {}
        self._opts: Dict[str, int] = {}
        self.geom: Optional[Tuple[int, int, int, int]] = None
        self._host = _HostCopies()
        # the storage order of the bound qkv rows (head-permuted or natural) and the logit bound are PROPERTIES OF THE
        # WEIGHTS, not knobs: a handle built directly must un-permute exactly as model.handle()'s does
        self.set_options()

    def _bind(self, name, w, b, rows, cols):
        hip._bf16(w, name)
        if w.stride(-1) != 1 or (b is not None and not b.is_contiguous()):
            raise hip.VclozeHipError(f"{name}: weight rows / bias must be contiguous")
        hip._check(hip.lib().vc_flux_bind_weight(self.h, name.encode(), w.data_ptr(), hip._p(b), rows, cols,
                                                 w.stride(0) if w.dim() == 2 else cols), f"vc_flux_bind_weight({name})")

    def __del__(self):
        try:
            if self.h:
                hip.lib().vc_flux_destroy(self.h)
                self.h = C.c_void_p()
        except Exception:
            pass

    def set_options(self, attn_variant=None, tile_cfg=0, fuse_qnorm=2, fuse_vt=True, qkv_heads=None, fuse_knorm=False,
                    logit_bound=0.0, mlp_first=False, splitk=True) -> None:
        import math
        wq = int(getattr(self.W, "qkv_heads", 0) or 0)
        if qkv_heads is None:
            qkv_heads = wq
        elif int(qkv_heads) != wq:
            raise hip.VclozeHipError(f"set_options(qkv_heads={qkv_heads}): the bound qkv weights are stored with qkv_heads={wq} "
                                     "(model.prepare / hip.qkv_head_permutation); the option follows the weights")
        want = dict(attn_variant=-1 if attn_variant is None else int(attn_variant), tile_cfg=int(tile_cfg),
                    fuse_qnorm=int(fuse_qnorm), fuse_vt=int(bool(fuse_vt)), qkv_heads=int(qkv_heads),
                    fuse_knorm=int(bool(fuse_knorm)),
                    logit_bound_milli=int(math.ceil(logit_bound * 1000)) if 0 < logit_bound < 2e6 else 0,
                    mlp_first=int(bool(mlp_first)), splitk=int(bool(splitk)))
        for k, v in want.items():
            if self._opts.get(k) != v:
                hip._check(hip.lib().vc_flux_set_option(self.h, k.encode(), v), f"vc_flux_set_option({k})")
                self._opts[k] = v

    def workspace(self, B: int, T: int, N: int, S: int) -> torch.Tensor:
        key = (B, T, N, S)
        ws = self._ws.get(key)
        if ws is None:
            if len(self._ws) > 8:
                self._ws.clear()
            n = hip.lib().vc_flux_workspace_bytes(self.h, B, T, N, S)
            if n <= 0:
                raise hip.VclozeHipError(f"vc_flux_workspace_bytes({key}) = {n}")
            ws = torch.empty(n + 256, dtype=torch.uint8, device=self.dev)
            self._ws[key] = ws
        return ws

    def prepare(self, txt, y, guidance, guidance_is_bf16: bool, img_ids, txt_ids, max_steps: int,
                kv_len: Optional[Sequence[int]] = None, kv_gap: Optional[Sequence[Tuple[int, int]]] = None, stream=None) -> None:
        """txt [B,T,ctx] / y [B,vec] bf16 device tensors; guidance [B] (any device) or None; ids [B,N|T,3]; kv_len B ints;
        kv_gap B (lo, hi) pairs (model.MaskLayout)."""
        hip._bf16(txt, "txt"); hip._bf16(y, "y")
        B, T = txt.shape[0], txt.shape[1]
        N = img_ids.shape[-2]
        txt, y = txt.contiguous(), y.contiguous()
        g = None
        if guidance is not None:
            g = self._host.f32("guidance", guidance).reshape(-1)
            g = np.ascontiguousarray(np.broadcast_to(g, (B,)) if g.size == 1 else g.reshape(B))
        ii, ti = self._host.f32("img_ids", img_ids).reshape(B, N, 3), self._host.f32("txt_ids", txt_ids).reshape(B, T, 3)
        kv = None if kv_len is None or all(int(v) == T + N for v in kv_len) else np.asarray([int(v) for v in kv_len], np.int32)
        gp = None
        if kv_gap is not None and any(hi > lo for lo, hi in kv_gap):
            gp = np.asarray([[int(lo), int(hi)] for lo, hi in kv_gap], np.int32).reshape(-1)
            if kv is None:
                kv = np.full(B, T + N, np.int32)
        ws = self.workspace(B, T, N, max_steps)
        base = (ws.data_ptr() + 255) & ~255
        inp = hip.FluxInputs(B, T, N, max_steps, txt.data_ptr(), y.data_ptr(), _fp(g), _fp(ii), _fp(ti), _ip(kv), _ip(gp),
                             int(bool(guidance_is_bf16)), 0)
        hip._check(hip.lib().vc_flux_prepare(self.h, C.byref(inp), base, ws.numel() - (base - ws.data_ptr()),
                                             stream if stream is not None else hip.cur_stream()), "vc_flux_prepare")
        self.geom = (B, T, N, max_steps)
        self._keep = (txt, y, ws)

    def forward(self, img, timesteps, timesteps_is_bf16: bool, out, stream=None) -> None:
        """img [B,N,in] bf16 -> out [B,N,out] bf16 (both contiguous), timesteps: B values"""
        hip._bf16(img, "img"); hip._bf16(out, "out")
        if not (img.is_contiguous() and out.is_contiguous()):
            raise hip.VclozeHipError("vc_flux_forward: contiguous img / out expected")
        t = _f32(timesteps).reshape(-1)
        if t.size != self.geom[0]:
            raise hip.VclozeHipError(f"vc_flux_forward: {t.size} timesteps for a batch of {self.geom[0]}")
        hip._check(hip.lib().vc_flux_forward(self.h, img.data_ptr(), _fp(t), int(bool(timesteps_is_bf16)), out.data_ptr(),
                                             stream if stream is not None else hip.cur_stream()), "vc_flux_forward")

    @staticmethod
    def _state(x, state_is_bf16, what):
        import torch
        want = torch.bfloat16 if state_is_bf16 else torch.float32     # the ABI takes the state in the caller's dtype
        if x.dtype != want:
            raise hip.VclozeHipError(f"{what}: state_is_bf16={bool(state_is_bf16)} needs a {want} state tensor, got {x.dtype}")

    def sample_begin(self, x, cond, t_grid, state_is_bf16: bool, stream) -> None:
        self._state(x, state_is_bf16, "vc_flux_sample_begin"); hip._bf16(cond, "cond")
        if not (x.is_contiguous() and cond.is_contiguous()):
            raise hip.VclozeHipError("vc_flux_sample: contiguous x / cond expected")
        t = _f32(t_grid).reshape(-1)
        hip._check(hip.lib().vc_flux_sample_begin(self.h, x.data_ptr(), cond.data_ptr(), _fp(t), t.size, int(bool(state_is_bf16)), stream),
                   "vc_flux_sample_begin")

    def sample_steps(self, n: int, stream, trajectory=None) -> None:
        hip._check(hip.lib().vc_flux_sample_steps(self.h, n, hip._p(trajectory), stream), "vc_flux_sample_steps")

    def profile(self, evaluations: int, stream) -> list:
        """vc_flux_profile: HIP-event times of the launches of `evaluations` evaluations at the current step of the sample in
        flight, class by class - a list of dicts (kind, epi, n, k, launches, flops, bytes, total_us, min_us, max_us)."""
        import ctypes as C
        cap = 32
        out = (hip.FluxLaunchClass * cap)()
        n = C.c_int32(0)
        hip._check(hip.lib().vc_flux_profile(self.h, int(evaluations), out, cap, C.byref(n), stream), "vc_flux_profile")
        return [{k: getattr(out[i], k) for k, _ in hip.FluxLaunchClass._fields_ if not k.startswith("reserved")} for i in range(n.value)]

    def sample_end(self, x_out, stream) -> None:
        hip._check(hip.lib().vc_flux_sample_end(self.h, x_out.data_ptr(), stream), "vc_flux_sample_end")

    def sample_euler(self, x, cond, t_grid, state_is_bf16: bool, stream, trajectory=None) -> None:
        """x [B,N,C] in place (bf16, or f32 with state_is_bf16 False): x(t_grid[0]) -> x(t_grid[-1]); trajectory: optional
        [S,B,N,C] buffer of the state's dtype"""
        self._state(x, state_is_bf16, "vc_flux_sample_euler"); hip._bf16(cond, "cond")
        if trajectory is not None:
            self._state(trajectory, state_is_bf16, "vc_flux_sample_euler (trajectory)")
        if not (x.is_contiguous() and cond.is_contiguous()) or (trajectory is not None and not trajectory.is_contiguous()):
            raise hip.VclozeHipError("vc_flux_sample_euler: contiguous x / cond / trajectory expected")
        t = _f32(t_grid).reshape(-1)
        hip._check(hip.lib().vc_flux_sample_euler(self.h, x.data_ptr(), cond.data_ptr(), _fp(t), t.size, int(bool(state_is_bf16)),
                                                  hip._p(trajectory), stream), "vc_flux_sample_euler")
