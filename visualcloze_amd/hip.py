"""ctypes binding of libvcloze_hip.so (the C ABI in include/vcloze_hip.h).

torch is used here only for device memory and streams: every wrapper takes torch CUDA tensors, checks
dtype / contiguity on the host and hands raw device pointers to the library.  There is NO fallback:
a missing library or a missing GPU raises (`VclozeHipError`)."""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Optional

import torch  # noqa: F401  (must be imported first so the process shares torch's HIP runtime)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("VC_HIP_LIB") or os.path.join(_HERE, "lib", "libvcloze_hip.so")   # VC_HIP_LIB: profiling build
CSRC = os.path.join(_HERE, "csrc")

ABI_VERSION = 10               # VC_ABI_VERSION of include/vcloze_hip.h
GEMM_MAX_PROBLEMS = 4          # VC_GEMM_MAX_PROBLEMS: grouped problems per vc_gemm launch
EPI_BIAS, EPI_GELU, EPI_GATE_RES, EPI_SILU, EPI_QKV = 0, 1, 2, 3, 4
GEMM_NO_SPLIT = 64             # VC_GEMM_NO_SPLIT: tile_cfg value that keeps an auto-tiled vc_gemm one launch
GEMM_PERSIST = 128             # VC_GEMM_PERSIST: opt into the persistent tile loop of the loader-wave GEMM (multi-round launches)
GEMM_NO_SPLITK = 1 << 20       # VC_GEMM_NO_SPLITK: an auto-tiled call keeps its remainder tiles whole even with a splitk_ws
GEMM_STREAMK = 1 << 21         # VC_GEMM_STREAMK: force the stream form of the split-K remainder (tests, A/B runs)
GEMM_PREFER_STREAMK = 1 << 22  # VC_GEMM_PREFER_STREAMK: auto-tiled calls take the stream form wherever eligible (A/B runs)
GEMM_STREAMK_ANY_K = 1 << 23   # VC_GEMM_STREAMK_ANY_K: ... at any K
GEMM_SPLITK_WS_BYTES = 2 * 256 * 256 * 192 * 4     # VC_GEMM_SPLITK_WS_BYTES


def GEMM_SPLITK(S: int) -> int:
    """VC_GEMM_SPLITK(S): force the 256x192 loader-wave tile with the remainder tiles cut S ways along K (tests, A/B runs)"""
    return int(S) << 16


class VclozeHipError(RuntimeError):
    pass


class GemmProblem(C.Structure):
    _fields_ = [
        ("A", C.c_void_p), ("W", C.c_void_p), ("bias", C.c_void_p), ("C", C.c_void_p),
        ("res", C.c_void_p), ("gate", C.c_void_p),
        ("lda", C.c_int64), ("ldw", C.c_int64), ("ldc", C.c_int64), ("ldres", C.c_int64), ("gate_bstride", C.c_int64),
        ("a_bstride", C.c_int64), ("c_bstride", C.c_int64),
        ("M", C.c_int32), ("N", C.c_int32), ("K", C.c_int32), ("rows_per_batch", C.c_int32),
        ("a_rpb", C.c_int32), ("c_rpb", C.c_int32),
        ("tiles_m", C.c_int32), ("tiles_n", C.c_int32), ("tile_start", C.c_int32), ("m_begin", C.c_int32),
        ("vt", C.c_void_p), ("vt_bstride", C.c_int64),
        ("vt_col0", C.c_int32), ("vt_rpb", C.c_int32), ("vt_row0", C.c_int32), ("vt_lpad", C.c_int32),
        ("kn_scale", C.c_void_p), ("kn_rope", C.c_void_p), ("kn_rope_bstride", C.c_int64), ("kn_heads", C.c_int32), ("qn_prescale", C.c_int32),
        ("qn_scale", C.c_void_p),
        ("a_zstride", C.c_int64), ("w_zstride", C.c_int64), ("c_zstride", C.c_int64),
    ]


class GemmArgs(C.Structure):
    _fields_ = [
        ("p", GemmProblem * 4), ("nprob", C.c_int32), ("epi", C.c_int32),
        ("step_ptr", C.c_void_p), ("gate_step_stride", C.c_int64), ("debug_ts", C.c_void_p),
        ("splitk_ws", C.c_void_p), ("splitk_ws_bytes", C.c_int64),
        ("sk_full", C.c_int32), ("sk_rem", C.c_int32), ("sk_S", C.c_int32), ("batch", C.c_int32),
        ("sk_stream", C.c_int32), ("sk_pad_", C.c_int32),
    ]


class LnStream(C.Structure):
    """VcLnStream of include/vcloze_hip.h."""
    _fields_ = [("x", C.c_void_p), ("ldx", C.c_int64), ("y", C.c_void_p), ("ldy", C.c_int64), ("shift", C.c_void_p),
                ("scale", C.c_void_p), ("rows", C.c_int32), ("rows_per_batch", C.c_int32)]


class Attention(C.Structure):
    """VcAttention of include/vcloze_hip.h."""
    _fields_ = [("qkv", C.c_void_p), ("ld", C.c_int64), ("bstride", C.c_int64), ("vt", C.c_void_p), ("out", C.c_void_p),
                ("ldo", C.c_int64), ("out_bstride", C.c_int64), ("kv_len", C.c_void_p),
                ("B", C.c_int32), ("L", C.c_int32), ("Lpad", C.c_int32), ("H", C.c_int32), ("variant", C.c_int32), ("split", C.c_int32),
                ("scratch", C.c_void_p), ("scratch_bytes", C.c_int64),
                ("q_scale", C.c_void_p), ("q_scale2", C.c_void_p), ("rope", C.c_void_p), ("rope_bstride", C.c_int64),
                ("kv_gap", C.c_void_p), ("logit_bound", C.c_float), ("q_prescaled", C.c_int32)]


class FluxConfig(C.Structure):
    """VcFluxConfig of include/vcloze_hip.h."""
    _fields_ = [("in_channels", C.c_int32), ("out_channels", C.c_int32), ("vec_in_dim", C.c_int32), ("context_in_dim", C.c_int32),
                ("hidden_size", C.c_int32), ("num_heads", C.c_int32), ("depth", C.c_int32), ("depth_single_blocks", C.c_int32),
                ("mlp_hidden", C.c_int32), ("guidance_embed", C.c_int32), ("axes_dim", C.c_int32 * 3), ("theta", C.c_int32)]


class FluxInputs(C.Structure):
    """VcFluxInputs of include/vcloze_hip.h (txt / y device pointers, everything else host arrays)."""
    _fields_ = [("B", C.c_int32), ("T", C.c_int32), ("N", C.c_int32), ("max_steps", C.c_int32),
                ("txt", C.c_void_p), ("y", C.c_void_p), ("guidance", C.POINTER(C.c_float)), ("img_ids", C.POINTER(C.c_float)),
                ("txt_ids", C.POINTER(C.c_float)), ("kv_len", C.POINTER(C.c_int32)), ("kv_gap", C.POINTER(C.c_int32)),
                ("guidance_is_bf16", C.c_int32), ("_pad", C.c_int32)]


# every symbol include/vcloze_hip.h declares: name -> (restype, argtypes)
_vp, _i32, _i64 = C.c_void_p, C.c_int32, C.c_int64
class FluxLaunchClass(C.Structure):          # VcFluxLaunchClass (vc_flux_profile)
    _fields_ = [("kind", C.c_int32), ("epi", C.c_int32), ("n", C.c_int32), ("k", C.c_int32), ("launches", C.c_int32), ("reserved", C.c_int32),
                ("flops", C.c_double), ("bytes", C.c_double), ("total_us", C.c_float), ("min_us", C.c_float), ("max_us", C.c_float),
                ("reserved2", C.c_float)]


LAUNCH_GEMM, LAUNCH_ATTENTION, LAUNCH_LN_MODULATE = 1, 2, 3

SYMBOLS = {
    "vc_abi_version": (C.c_int, []),
    "vc_last_error": (C.c_char_p, []),
    "vc_struct_sizes": (None, [C.POINTER(C.c_int32)]),
    "vc_device_count": (C.c_int, []),
    "vc_device_info": (C.c_int, [C.c_int, C.c_char_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int64)]),
    "vc_gemm": (C.c_int, [C.POINTER(GemmArgs), C.c_int, _vp]),
    "vc_gemm_plan": (C.c_int, [C.POINTER(GemmArgs), C.c_int, C.POINTER(C.c_int32)]),
    "vc_ln_modulate": (C.c_int, [_vp, _i64, _vp, _i64, _vp, _vp, _i64, _i32, _i32, _i32, _vp, _i64, _vp]),
    "vc_ln_modulate2": (C.c_int, [_vp, _vp, _i64, _i32, _vp, _i64, _vp]),
    "vc_qknorm_rope_vt": (C.c_int, [_vp, _i64, _i64, _vp, _vp, _vp, _vp, _i32, _vp, _i64, _vp, _i32, _i32, _i32, _i32, _i32, _vp]),
    "vc_attention": (C.c_int, [C.POINTER(Attention), _vp]),
    "vc_attention_scratch_bytes": (_i64, []),
    "vc_timestep_embedding": (C.c_int, [_vp, _vp, _vp, _i32, _i32, _i32, _vp]),
    "vc_silu": (C.c_int, [_vp, _vp, _i64, _vp]),
    "vc_act2d": (C.c_int, [_vp, _i64, _vp, _i64, _i32, _i32, _i32, _vp]),
    "vc_gate_residual": (C.c_int, [_vp, _i64, _vp, _i64, _vp, _vp, _i64, _i32, _i32, _vp, _i64, _vp]),
    "vc_add3": (C.c_int, [_vp, _vp, _vp, _vp, _i64, _i64, _i64, _vp]),
    "vc_copy": (C.c_int, [_vp, _vp, _i64, _vp]),
    "vc_concat_cols": (C.c_int, [_vp, _i32, _vp, _i32, _vp, _i64, _vp]),
    "vc_euler_step": (C.c_int, [_vp, _vp, _vp, _vp, _i64, _vp]),
    "vc_euler_step_f32": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _i64, _vp]),
    "vc_step_advance": (C.c_int, [_vp, _vp]),
    "vc_sdedit_mix": (C.c_int, [_vp, _vp, C.c_float, _vp, _i64, _vp]),
    "vc_pack_latent": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _i64, _i32, _vp]),
    "vc_pack_mask": (C.c_int, [_vp, _vp, _i32, _i32, _i64, _i32, _vp]),
    "vc_unpack_latent": (C.c_int, [_vp, _i64, _i32, _vp, _i32, _i32, _i32, _vp]),
    "vc_im2col3x3": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _i32, _vp]),
    "vc_conv3x3": (C.c_int, [_vp, _vp, _vp, _vp, _i64, _vp, _i64, _vp, _i32, _i32, _i32, _i32, _i32, _vp]),
    "vc_groupnorm": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _i64, _i64, _i32, _i32, C.c_float, _i32, _vp]),
    "vc_softmax_rows": (C.c_int, [_vp, _i64, _i32, _i32, C.c_float, _vp, _i64, _i32, _vp]),
    "vc_embedding": (C.c_int, [_vp, _vp, _i64, _i32, _vp, _i32, _i32, _vp]),
    "vc_rmsnorm": (C.c_int, [_vp, _vp, _vp, _i32, _i32, C.c_float, _vp]),
    "vc_layernorm": (C.c_int, [_vp, _vp, _vp, _vp, _i32, _i32, C.c_float, _vp]),
    "vc_mul": (C.c_int, [_vp, _vp, _vp, _i64, _vp]),
    "vc_add": (C.c_int, [_vp, _vp, _vp, _i64, _vp]),
    "vc_quick_gelu": (C.c_int, [_vp, _vp, _i64, _vp]),
    "vc_transpose": (C.c_int, [_vp, _i64, _vp, _i64, _i32, _i32, _vp]),
    "vc_nchw_to_nhwc": (C.c_int, [_vp, _i32, _vp, _i32, _i32, _i64, C.c_float, C.c_float, _vp]),
    "vc_nhwc_to_nchw": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _i64, _vp]),
    "vc_gaussian_sample": (C.c_int, [_vp, _i32, _vp, _vp, _i32, _i64, C.c_float, C.c_float, _vp]),
    "vc_flux_create": (C.c_int, [C.POINTER(FluxConfig), C.POINTER(_vp)]),
    "vc_flux_destroy": (C.c_int, [_vp]),
    "vc_flux_bind_weight": (C.c_int, [_vp, C.c_char_p, _vp, _vp, _i32, _i32, _i64]),
    "vc_flux_mod_offset": (_i64, [_vp, C.c_char_p]),
    "vc_flux_set_option": (C.c_int, [_vp, C.c_char_p, _i32]),
    "vc_flux_workspace_bytes": (_i64, [_vp, _i32, _i32, _i32, _i32]),
    "vc_flux_prepare": (C.c_int, [_vp, C.POINTER(FluxInputs), _vp, _i64, _vp]),
    "vc_flux_forward": (C.c_int, [_vp, _vp, C.POINTER(C.c_float), _i32, _vp, _vp]),
    "vc_flux_sample_euler": (C.c_int, [_vp, _vp, _vp, C.POINTER(C.c_float), _i32, _i32, _vp, _vp]),
    "vc_flux_sample_begin": (C.c_int, [_vp, _vp, _vp, C.POINTER(C.c_float), _i32, _i32, _vp]),
    "vc_flux_sample_steps": (C.c_int, [_vp, _i32, _vp, _vp]),
    "vc_flux_sample_end": (C.c_int, [_vp, _vp, _vp]),
    "vc_flux_profile": (C.c_int, [_vp, _i32, C.POINTER(FluxLaunchClass), _i32, C.POINTER(_i32), _vp]),
    "vc_stream_create": (C.c_int, [C.POINTER(_vp)]),
    "vc_stream_destroy": (C.c_int, [_vp]),
    "vc_stream_sync": (C.c_int, [_vp]),
    "vc_graph_begin": (C.c_int, [_vp]),
    "vc_graph_end": (C.c_int, [_vp, C.POINTER(_vp)]),
    "vc_graph_launch": (C.c_int, [_vp, _vp]),
    "vc_graph_destroy": (C.c_int, [_vp]),
    "vc_event_create": (C.c_int, [C.POINTER(_vp)]),
    "vc_event_record": (C.c_int, [_vp, _vp]),
    "vc_event_elapsed_ms": (C.c_int, [_vp, _vp, C.POINTER(C.c_float)]),
    "vc_event_destroy": (C.c_int, [_vp]),
}

_lib = None


def build(force: bool = False) -> str:
    """Compile csrc/*.hip for gfx950 into lib/libvcloze_hip.so (hipcc cross-compiles without a GPU).  `make` is
    incremental and knows every dependency (the ABI header include/vcloze_hip.h included), so it is simply run."""
    if force:
        subprocess.run(["make", "-C", CSRC, "clean"], capture_output=True, text=True)
    r = subprocess.run(["make", "-C", CSRC, "-j8"], capture_output=True, text=True)
    if r.returncode != 0:
        raise VclozeHipError("building libvcloze_hip.so failed:\n" + r.stdout[-4000:] + r.stderr[-4000:])
    return LIB_PATH


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise VclozeHipError(
                f"{LIB_PATH} is missing — run `python -c 'import __graft_entry__ as g; g.build()'` "
                "(the HIP extension is mandatory; there is no CPU fallback)")
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(l, name)  # AttributeError if the ABI lost a symbol
            fn.restype, fn.argtypes = res, args
        if l.vc_abi_version() != ABI_VERSION:
            raise VclozeHipError(f"libvcloze_hip.so ABI version {l.vc_abi_version()} != {ABI_VERSION} expected by hip.py - rebuild")
        sizes = (C.c_int32 * 7)()
        l.vc_struct_sizes(sizes)          # a stale library whose structs disagree with these ctypes mirrors must not run
        if list(sizes) != [C.sizeof(GemmProblem), C.sizeof(GemmArgs), C.sizeof(LnStream), C.sizeof(Attention),
                           C.sizeof(FluxConfig), C.sizeof(FluxInputs), C.sizeof(FluxLaunchClass)]:
            raise VclozeHipError(f"libvcloze_hip.so struct sizes {list(sizes)} differ from the ctypes mirrors - rebuild")
        _lib = l
    return _lib


def _check(rc: int, what: str) -> None:
    if rc != 0:
        raise VclozeHipError(f"{what} failed ({rc}): {lib().vc_last_error().decode()}")


def require_gpu() -> None:
    if lib().vc_device_count() < 1:
        raise VclozeHipError("no ROCm device visible: " + lib().vc_last_error().decode())


def device_cus(device=None) -> int:
    """compute units of the device (256 on MI355X)"""
    idx = torch.device(device).index if device is not None else torch.cuda.current_device()
    n = C.c_int()
    _check(lib().vc_device_info(idx or 0, None, 0, C.byref(n), None), "vc_device_info")
    return n.value


def cur_stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _p(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def _bf16(t: torch.Tensor, name: str) -> None:
    if t.dtype != torch.bfloat16 or not t.is_cuda:
        raise VclozeHipError(f"{name}: expected a CUDA bf16 tensor, got {t.dtype} on {t.device}")


# ------------------------------------------------------------------------------------------------
# op wrappers (2-D row-major views; the last dim must be contiguous)
# ------------------------------------------------------------------------------------------------
def qkv_head_permutation(H: int) -> torch.Tensor:
    """Row order of a HEAD-PERMUTED qkv weight [3 * 128 * H, K] (VcGemmProblem.kn_heads): 2H blocks of 192 rows = [head t (128
    rows: query head t for t < H, key head t - H after) | V rows 64 t .. 64 t + 63] - every 192-column GEMM tile then holds one
    whole query or key head (QKNorm + RoPE in its epilogue) and half a value head.  perm[p] = the original row."""
    D = 128 * H
    idx = []
    for t in range(2 * H):
        idx += list(range(128 * t, 128 * t + 128)) + list(range(2 * D + 64 * t, 2 * D + 64 * t + 64))
    return torch.tensor(idx, dtype=torch.long)


def make_problem(a, w, bias, out, res=None, gate=None, rows_per_batch=None, gate_bstride=0, M=None,
                 a_rpb=0, a_bstride=0, c_rpb=0, c_bstride=0, vt=None, vt_col0=0, vt_rpb=0, vt_row0=0,
                 kn_heads=0, kn_scale=None, kn_rope=None, qn_scale=None, qn_prescale=False) -> GemmProblem:
    """`M` + (a_rpb, a_bstride) / (c_rpb, c_bstride) describe batch-strided rows: `a` / `out` are then views of the
    FIRST batch element's rows (row m of the problem lives at (m // rpb) * bstride + (m % rpb) * ld).
    EPI_QKV: `vt` [B, H, 128, Lpad] receives the columns >= vt_col0 transposed; this problem's rows are tokens
    vt_row0 .. vt_row0 + vt_rpb of each batch element's joint sequence."""
    for n, t in (("A", a), ("W", w), ("C", out)):
        _bf16(t, n)
        if t.dim() != 2 or t.stride(1) != 1:
            raise VclozeHipError(f"gemm {n}: need a 2-D tensor with contiguous last dim")
    K = a.shape[1]
    N = w.shape[0]
    if M is None:
        M = a.shape[0]
        if tuple(out.shape) != (M, N):
            raise VclozeHipError(f"gemm shape mismatch A{tuple(a.shape)} W{tuple(w.shape)} C{tuple(out.shape)}")
    if w.shape[1] != K or out.shape[1] != N:
        raise VclozeHipError(f"gemm shape mismatch A{tuple(a.shape)} W{tuple(w.shape)} C{tuple(out.shape)}")
    p = GemmProblem()
    p.A, p.W, p.C = a.data_ptr(), w.data_ptr(), out.data_ptr()
    p.bias = _p(bias)
    p.lda, p.ldw, p.ldc = a.stride(0), w.stride(0), out.stride(0)
    p.M, p.N, p.K = M, N, K
    p.rows_per_batch = rows_per_batch or M
    p.a_rpb, p.a_bstride, p.c_rpb, p.c_bstride = a_rpb, a_bstride, c_rpb, c_bstride
    if res is not None:
        _bf16(res, "res")
        p.res, p.ldres = res.data_ptr(), res.stride(0)
    if gate is not None:
        _bf16(gate, "gate")
        p.gate, p.gate_bstride = gate.data_ptr(), gate_bstride
    if vt is not None:
        _bf16(vt, "vt")
        if vt.dim() != 4 or not vt.is_contiguous():
            raise VclozeHipError("gemm vt: contiguous [B, H, 128, Lpad] expected")
        p.vt, p.vt_bstride, p.vt_lpad = vt.data_ptr(), vt.stride(0), vt.shape[-1]
        p.vt_col0, p.vt_rpb, p.vt_row0 = vt_col0, vt_rpb or M, vt_row0
    if kn_heads:
        p.kn_heads = kn_heads        # w / bias are head-permuted (qkv_head_permutation); C receives the logical columns
        if vt is None:
            p.vt_rpb, p.vt_row0 = vt_rpb or M, vt_row0
        if kn_scale is not None or qn_scale is not None:     # ... and QKNorm + RoPE of the key / query heads happen in the epilogue
            if kn_rope is None or kn_rope.dtype != torch.float32 or not kn_rope.is_contiguous():
                raise VclozeHipError("gemm kn_rope: contiguous f32 [B?, L, 64, 2] expected")
            p.kn_rope = kn_rope.data_ptr()
            p.kn_rope_bstride = kn_rope.shape[-3] * 128 if kn_rope.dim() == 4 else 0
            if kn_scale is not None:
                _bf16(kn_scale, "kn_scale")
                p.kn_scale = kn_scale.data_ptr()
            if qn_scale is not None:     # qn_prescale: the queries leave times 128^-0.5 * log2(e) (attention(q_prescaled=True))
                _bf16(qn_scale, "qn_scale")
                p.qn_scale, p.qn_prescale = qn_scale.data_ptr(), int(bool(qn_prescale))
    p._keep = (a, w, bias, out, res, gate, vt, kn_scale, kn_rope, qn_scale)   # the struct carries raw pointers: keep the operands alive with it
    return p


def splitk_workspace(device) -> torch.Tensor:
    """f32 scratch for the split-K remainder of vc_gemm (VcGemmArgs.splitk_ws): two 256x192 partial tiles per CU of one round"""
    return torch.empty(GEMM_SPLITK_WS_BYTES, dtype=torch.uint8, device=device)


def gemm(problems, epi=EPI_BIAS, tile_cfg=0, step_ptr=None, gate_step_stride=0, stream=None, debug_ts=None, splitk_ws=None,
         batch=0) -> None:
    """`batch` = Z > 1: Z independent GEMMs per problem in one launch; set the problem's a_zstride / w_zstride / c_zstride."""
    args = GemmArgs()
    args.batch = batch
    if splitk_ws is not None:
        args.splitk_ws, args.splitk_ws_bytes = splitk_ws.data_ptr(), splitk_ws.numel() * splitk_ws.element_size()
    if isinstance(problems, GemmProblem):
        problems = [problems]
    args.nprob = len(problems)
    for i, p in enumerate(problems):
        args.p[i] = p
    args.epi = epi
    args.step_ptr = _p(step_ptr)
    args.gate_step_stride = gate_step_stride
    args.debug_ts = _p(debug_ts)
    _check(lib().vc_gemm(C.byref(args), tile_cfg, stream if stream is not None else cur_stream()), "vc_gemm")


def linear(a, w, bias=None, out=None, epi=EPI_BIAS, res=None, gate=None, tile_cfg=0, stream=None):
    if out is None:
        out = torch.empty(a.shape[0], w.shape[0], dtype=torch.bfloat16, device=a.device)
    gemm(make_problem(a, w, bias, out, res=res, gate=gate), epi=epi, tile_cfg=tile_cfg, stream=stream)
    return out


def ln_modulate(x, shift, scale, out=None, step_ptr=None, mod_step_stride=0, stream=None, rows_per_batch=None,
                mod_bstride=0):
    _bf16(x, "x")
    rows, D = x.shape
    if out is None:
        out = torch.empty(rows, D, dtype=torch.bfloat16, device=x.device)
    _check(lib().vc_ln_modulate(x.data_ptr(), x.stride(0), out.data_ptr(), out.stride(0), shift.data_ptr(),
                                scale.data_ptr(), mod_bstride, rows, D, rows_per_batch or rows, _p(step_ptr), mod_step_stride,
                                stream if stream is not None else cur_stream()), "vc_ln_modulate")
    return out


def ln_modulate2(streams, step_ptr=None, mod_step_stride=0, stream=None, mod_bstride=0) -> None:
    """LayerNorm + modulate over one or two row sets in one launch; streams: [(x, shift, scale, out, rows_per_batch), ...]
    (the img and txt streams of a DoubleStreamBlock)."""
    if not 1 <= len(streams) <= 2:
        raise VclozeHipError("ln_modulate2 takes one or two row sets")
    segs, D = [], streams[0][0].shape[1]
    for (x, shift, scale, out, rpb) in streams:
        _bf16(x, "x"); _bf16(out, "out")
        if x.shape[1] != D or out.shape != x.shape:
            raise VclozeHipError("ln_modulate2: row sets must share D, and out must match x")
        segs.append(LnStream(x.data_ptr(), x.stride(0), out.data_ptr(), out.stride(0), shift.data_ptr(), scale.data_ptr(),
                             x.shape[0], rpb or x.shape[0]))
    _check(lib().vc_ln_modulate2(C.byref(segs[0]), C.byref(segs[1]) if len(segs) > 1 else None, mod_bstride, D,
                                 _p(step_ptr), mod_step_stride, stream if stream is not None else cur_stream()), "vc_ln_modulate2")


QKN_Q, QKN_K, QKN_VT, QKN_QPRE = 1, 2, 4, 8


def qknorm_rope_vt(qkv, q_scale, k_scale, rope, vt, L, H, stream=None, q_scale2=None, k_scale2=None, split=0, B=1,
                   parts=QKN_Q | QKN_K | QKN_VT):
    """qkv: [L, >=3*H*128] rows (q|k|v at column 0, H*128, 2*H*128); rope: [L,64,2] f32; vt: [H,128,Lpad].
    rows < split use (q_scale, k_scale), the rest (q_scale2, k_scale2) when given.  parts: which of q / k / V^T to do."""
    _bf16(qkv, "qkv")
    if rope.dtype != torch.float32 or not rope.is_contiguous():
        raise VclozeHipError("rope table must be contiguous f32 [B?,L,64,2]")
    Lpad = vt.shape[-1]
    # B > 1: qkv rows are sample-major ([B*L, ld]), rope [B,L,64,2], vt [B,H,128,Lpad]
    _check(lib().vc_qknorm_rope_vt(qkv.data_ptr(), qkv.stride(0), L * qkv.stride(0), q_scale.data_ptr(), k_scale.data_ptr(),
                                   _p(q_scale2), _p(k_scale2), split, rope.data_ptr(), L * 128 if rope.dim() == 4 else 0,
                                   vt.data_ptr(), B, L, Lpad, H, parts,
                                   stream if stream is not None else cur_stream()), "vc_qknorm_rope_vt")


_attn_scratch = {}


def attention_scratch(device) -> torch.Tensor:
    """The partial-result buffer of the tail-split attention (variants 7 / 12 / 28), one per device; launches on one stream
    serialise their use of it.  ZERO-initialised: variant 28 combines the pieces inside the launch through flag words at the
    end of the buffer that are zero before and after every launch (VcAttention.variant bit 16)."""
    key = str(device)
    if key not in _attn_scratch:
        _attn_scratch[key] = torch.zeros(lib().vc_attention_scratch_bytes(), dtype=torch.uint8, device=device)
    return _attn_scratch[key]


def attention(qkv, vt, out, L, H, kv_len=None, variant=0, stream=None, B=1, scratch=None, q_norm=None, kv_gap=None,
              logit_bound=0.0, q_prescaled=False):
    """out: [B*L, >=H*128] rows (B L (H D)), sample-major like qkv; kv_len: optional int32 device tensor [B];
    scratch: uint8 device buffer of >= vc_attention_scratch_bytes() for the tail split (taken from attention_scratch()
    when omitted); q_norm = (q_scale, q_scale2 | None, split, rope): QKNorm + RoPE of the RAW query rows inside the kernel
    (variants 8 / 12; the pre-pass then runs with parts = QKN_K | QKN_VT); q_prescaled: the q columns hold normalised, rotated
    queries times 128^-0.5 * log2(e) (make_problem(qn_prescale=True) / QKN_QPRE; variants 8 / 12)."""
    _bf16(qkv, "qkv"); _bf16(vt, "vt"); _bf16(out, "out")
    if scratch is None and variant & 4:
        scratch = attention_scratch(qkv.device)
    a = Attention()
    a.qkv, a.ld, a.bstride = qkv.data_ptr(), qkv.stride(0), L * qkv.stride(0)
    a.vt, a.out, a.ldo, a.out_bstride = vt.data_ptr(), out.data_ptr(), out.stride(0), L * out.stride(0)
    a.kv_len = _p(kv_len)
    a.kv_gap = _p(kv_gap)      # int32 [B, 2] device tensor: (lo, hi) of a second masked range (needs kv_len)
    a.B, a.L, a.Lpad, a.H, a.variant = B, L, vt.shape[-1], H, variant
    a.scratch, a.scratch_bytes = _p(scratch), scratch.numel() if scratch is not None else 0
    a.q_prescaled = int(bool(q_prescaled))
    a.logit_bound = float(logit_bound)   # > 0: |q.k| * 128^-0.5 * log2(e) <= logit_bound guaranteed (variants 8 / 12: no running max)
    if q_norm is not None:
        qs, qs2, split, rope = q_norm
        _bf16(qs, "q_scale")
        if rope.dtype != torch.float32 or not rope.is_contiguous():
            raise VclozeHipError("rope table must be contiguous f32 [B?,L,64,2]")
        a.q_scale, a.q_scale2, a.split = qs.data_ptr(), _p(qs2), split
        a.rope, a.rope_bstride = rope.data_ptr(), L * 128 if rope.dim() == 4 else 0
    _check(lib().vc_attention(C.byref(a), stream if stream is not None else cur_stream()), "vc_attention")


def timestep_embedding(t_f32, freqs_f32, out_bf16, round_t_bf16=False, stream=None):
    n, half = t_f32.numel(), freqs_f32.numel()
    _check(lib().vc_timestep_embedding(t_f32.data_ptr(), freqs_f32.data_ptr(), out_bf16.data_ptr(), n, half,
                                       int(round_t_bf16), stream if stream is not None else cur_stream()),
           "vc_timestep_embedding")


def silu(x, out=None, stream=None):
    out = torch.empty_like(x) if out is None else out
    _check(lib().vc_silu(x.data_ptr(), out.data_ptr(), x.numel(), stream if stream is not None else cur_stream()), "vc_silu")
    return out


def act2d(x, out, act, stream=None):
    """out = bf16(act(x)) on 2-D row views; act: "gelu" (tanh) or "silu"."""
    for t in (x, out):
        _bf16(t, "act2d")
        if t.dim() != 2 or t.stride(1) != 1 or t.shape != x.shape:
            raise VclozeHipError("act2d: 2-D views of one shape with contiguous rows expected")
    _check(lib().vc_act2d(x.data_ptr(), x.stride(0), out.data_ptr(), out.stride(0), x.shape[0], x.shape[1],
                          {"gelu": 0, "silu": 1}[act], stream if stream is not None else cur_stream()), "vc_act2d")
    return out


def gate_residual(y, res, gate, out, step_ptr=None, gate_step_stride=0, stream=None):
    """out = bf16(res + bf16(gate * y)) on 2-D row views (last dim contiguous); out may alias res."""
    for t in (y, res, out):
        _bf16(t, "gate_residual")
        if t.dim() != 2 or t.stride(1) != 1 or t.shape != y.shape:
            raise VclozeHipError("gate_residual: 2-D views of one shape with contiguous rows expected")
    _check(lib().vc_gate_residual(y.data_ptr(), y.stride(0), res.data_ptr(), res.stride(0), gate.data_ptr(), out.data_ptr(),
                                  out.stride(0), y.shape[0], y.shape[1], _p(step_ptr), gate_step_stride,
                                  stream if stream is not None else cur_stream()), "vc_gate_residual")


def add3(a, b, c=None, out=None, stream=None):
    out = torch.empty_like(a) if out is None else out
    _check(lib().vc_add3(a.data_ptr(), b.data_ptr(), _p(c), out.data_ptr(), a.numel(), b.numel(),
                         c.numel() if c is not None else 1, stream if stream is not None else cur_stream()), "vc_add3")
    return out


def copy(dst, src, stream=None):
    if dst.numel() * dst.element_size() != src.numel() * src.element_size() or not (dst.is_contiguous() and src.is_contiguous()):
        raise VclozeHipError("copy: size/contiguity mismatch")
    _check(lib().vc_copy(dst.data_ptr(), src.data_ptr(), src.numel() * src.element_size(),
                         stream if stream is not None else cur_stream()), "vc_copy")


def concat_cols(x, cond, out, stream=None):
    rows = x.shape[0]
    _check(lib().vc_concat_cols(x.data_ptr(), x.shape[1], cond.data_ptr(), cond.shape[1], out.data_ptr(), rows,
                                stream if stream is not None else cur_stream()), "vc_concat_cols")
    return out


def euler_step_f32(x32, shadow, v, dts, step_ptr=None, stream=None):
    """f32 master state: x32 += f32(bf16(bf16(dt) * (-v))); shadow = bf16(x32).  v None: refresh the shadow only."""
    _check(lib().vc_euler_step_f32(x32.data_ptr(), shadow.data_ptr(), _p(v), _p(dts), _p(step_ptr), x32.numel(),
                                   stream if stream is not None else cur_stream()), "vc_euler_step_f32")


def euler_step(x, v, dts, step_ptr=None, stream=None):
    _check(lib().vc_euler_step(x.data_ptr(), v.data_ptr(), dts.data_ptr(), _p(step_ptr), x.numel(),
                               stream if stream is not None else cur_stream()), "vc_euler_step")


def step_advance(step_ptr, stream=None):
    _check(lib().vc_step_advance(step_ptr.data_ptr(), stream if stream is not None else cur_stream()), "vc_step_advance")


def sdedit_mix(noise, latent, strength, out=None, stream=None):
    _bf16(noise, "noise"); _bf16(latent, "latent")
    out = torch.empty_like(noise) if out is None else out
    _check(lib().vc_sdedit_mix(noise.data_ptr(), latent.data_ptr(), float(strength), out.data_ptr(), noise.numel(),
                               stream if stream is not None else cur_stream()), "vc_sdedit_mix")
    return out


def pack_latent(latent, tokens, col0=0, stream=None):
    """latent [C,h,w] bf16 (contiguous) -> tokens[:, col0:col0+4C] (rows of stride tokens.stride(0))."""
    _bf16(latent, "latent"); _bf16(tokens, "tokens")
    Cc, h, w = latent.shape[-3:]
    if not latent.is_contiguous() or tokens.shape[0] != (h // 2) * (w // 2):
        raise VclozeHipError("pack_latent: contiguous [C,h,w] latent and (h/2)(w/2) token rows expected")
    if Cc > 64:
        raise VclozeHipError(f"pack_latent: C={Cc} > 64 channels (the tile staged through LDS holds 64)")
    _check(lib().vc_pack_latent(latent.data_ptr(), tokens.data_ptr(), Cc, h, w, tokens.stride(0), col0,
                                stream if stream is not None else cur_stream()), "vc_pack_latent")


def pack_mask(mask, tokens, col0=0, stream=None):
    """pixel mask [H,W] bf16 -> tokens[:, col0:col0+256]."""
    _bf16(mask, "mask"); _bf16(tokens, "tokens")
    H, W = mask.shape[-2:]
    if not mask.is_contiguous() or tokens.shape[0] != (H // 16) * (W // 16):
        raise VclozeHipError("pack_mask: contiguous [H,W] mask and (H/16)(W/16) token rows expected")
    if mask.data_ptr() % 16 or tokens.data_ptr() % 16:
        raise VclozeHipError("pack_mask: the mask and the token rows must be 16-byte aligned (a view at an odd storage offset is not)")
    _check(lib().vc_pack_mask(mask.data_ptr(), tokens.data_ptr(), H, W, tokens.stride(0), col0,
                              stream if stream is not None else cur_stream()), "vc_pack_mask")


def unpack_latent(tokens, latent, col0=0, stream=None):
    _bf16(latent, "latent"); _bf16(tokens, "tokens")
    Cc, h, w = latent.shape[-3:]
    if not latent.is_contiguous() or tokens.shape[0] != (h // 2) * (w // 2):
        raise VclozeHipError("unpack_latent: contiguous [C,h,w] latent and (h/2)(w/2) token rows expected")
    if Cc > 64:
        raise VclozeHipError(f"unpack_latent: C={Cc} > 64 channels (the tile staged through LDS holds 64)")
    _check(lib().vc_unpack_latent(tokens.data_ptr(), tokens.stride(0), col0, latent.data_ptr(), Cc, h, w,
                                  stream if stream is not None else cur_stream()), "vc_unpack_latent")


# ---- VAE decoder glue (activations NHWC bf16 [H*W, C]) ----
def im2col3x3(src, dst, H, W, up=False, down=False, stream=None):
    """(H, W) is the OUTPUT map; `up`: src is the half-resolution map (nearest 2x folded in); `down`: src is the
    double-resolution map (pad (0,1,0,1) + stride 2)."""
    _bf16(src, "src"); _bf16(dst, "dst")
    Cc = src.shape[1]
    hs, ws = (H >> 1, W >> 1) if up else (2 * H, 2 * W) if down else (H, W)
    if (up and down) or src.shape[0] != hs * ws or tuple(dst.shape) != (H * W, 9 * Cc) or not (src.is_contiguous() and dst.is_contiguous()):
        raise VclozeHipError("im2col3x3: src [Hs*Ws, C] and dst [H*W, 9C] contiguous expected")
    _check(lib().vc_im2col3x3(src.data_ptr(), dst.data_ptr(), H, W, Cc, 1 if up else 2 if down else 0,
                              stream if stream is not None else cur_stream()), "vc_im2col3x3")


def conv3x3(x, w, bias, out, H, W, up=False, down=False, res=None, gate=None, stream=None):
    """One-launch 3x3 convolution (implicit GEMM).  x: [Hs*Ws + 1, C] NHWC map whose last row is zero; w [O, 9C];
    out [H*W, O]; optional residual: out = res + gate * (conv + bias)."""
    _bf16(x, "x"); _bf16(w, "w"); _bf16(out, "out")
    Cc, O = x.shape[1], w.shape[0]
    hs, ws = (H >> 1, W >> 1) if up else (2 * H, 2 * W) if down else (H, W)
    if (up and down) or x.shape[0] != hs * ws + 1 or not x.is_contiguous() or w.shape[1] != 9 * Cc or not w.is_contiguous() \
            or tuple(out.shape) != (H * W, O) or out.stride(1) != 1:
        raise VclozeHipError("conv3x3: x [Hs*Ws + 1, C] (last row zero), w [O, 9C], out [H*W, O] expected")
    if res is not None:
        _bf16(res, "res"); _bf16(gate, "gate")
    _check(lib().vc_conv3x3(x.data_ptr(), w.data_ptr(), _p(bias), out.data_ptr(), out.stride(0), _p(res),
                            res.stride(0) if res is not None else 0, _p(gate), H, W, Cc, O, 1 if up else 2 if down else 0,
                            stream if stream is not None else cur_stream()), "vc_conv3x3")


def groupnorm(x, gamma, beta, y, scratch, groups=32, eps=1e-6, swish=False, stream=None):
    _bf16(x, "x"); _bf16(y, "y"); _bf16(gamma, "gamma"); _bf16(beta, "beta")
    HW, Cc = x.shape
    if not (x.is_contiguous() and y.is_contiguous()) or y.shape != x.shape or scratch.dtype != torch.float32:
        raise VclozeHipError("groupnorm: contiguous [HW, C] x, y and an f32 scratch tensor expected")
    _check(lib().vc_groupnorm(x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), y.data_ptr(), scratch.data_ptr(),
                              scratch.numel() * 4, HW, Cc, groups, eps, int(bool(swish)),
                              stream if stream is not None else cur_stream()), "vc_groupnorm")


def groupnorm_scratch_floats(HW, groups=32):
    return ((HW + 127) // 128 + 1) * 2 * groups


def softmax_rows(x, scale, bias=None, causal_period=0, stream=None):
    _bf16(x, "x")
    if x.dim() != 2 or x.stride(1) != 1:
        raise VclozeHipError("softmax_rows: 2-D tensor with contiguous rows expected")
    ldb = 0
    if bias is not None:
        _bf16(bias, "bias")
        if bias.dim() != 2 or bias.stride(1) != 1 or bias.shape != x.shape:
            raise VclozeHipError("softmax_rows: bias must match x")
        ldb = bias.stride(0)
    _check(lib().vc_softmax_rows(x.data_ptr(), x.stride(0), x.shape[0], x.shape[1], scale, _p(bias), ldb, causal_period,
                                 stream if stream is not None else cur_stream()), "vc_softmax_rows")


# ---- text-encoder glue ----
def embedding(ids, table, out, stream=None):
    _bf16(table, "table"); _bf16(out, "out")
    if ids.dtype != torch.int32 or not ids.is_cuda or ids.dim() != 1 or not ids.is_contiguous():
        raise VclozeHipError("embedding: ids must be a contiguous CUDA int32 vector")
    if table.stride(1) != 1 or not out.is_contiguous() or tuple(out.shape) != (ids.shape[0], table.shape[1]):
        raise VclozeHipError("embedding: out [L, D] contiguous expected")
    _check(lib().vc_embedding(ids.data_ptr(), table.data_ptr(), table.stride(0), table.shape[0], out.data_ptr(), ids.shape[0],
                              table.shape[1], stream if stream is not None else cur_stream()), "vc_embedding")


def rmsnorm(x, weight, y, eps, stream=None):
    _bf16(x, "x"); _bf16(weight, "weight"); _bf16(y, "y")
    if not (x.is_contiguous() and y.is_contiguous()) or x.shape != y.shape or weight.numel() != x.shape[1]:
        raise VclozeHipError("rmsnorm: contiguous [rows, D] x, y and weight [D] expected")
    _check(lib().vc_rmsnorm(x.data_ptr(), weight.data_ptr(), y.data_ptr(), x.shape[0], x.shape[1], eps,
                            stream if stream is not None else cur_stream()), "vc_rmsnorm")


def layernorm(x, weight, bias, y, eps, stream=None):
    _bf16(x, "x"); _bf16(weight, "weight"); _bf16(bias, "bias"); _bf16(y, "y")
    if not (x.is_contiguous() and y.is_contiguous()) or x.shape != y.shape or weight.numel() != x.shape[1]:
        raise VclozeHipError("layernorm: contiguous [rows, D] x, y and weight/bias [D] expected")
    _check(lib().vc_layernorm(x.data_ptr(), weight.data_ptr(), bias.data_ptr(), y.data_ptr(), x.shape[0], x.shape[1], eps,
                              stream if stream is not None else cur_stream()), "vc_layernorm")


def _ew(fn, name, a, b, y, stream):
    for t in (a, b, y):
        if t is not None:
            _bf16(t, name)
            if not t.is_contiguous() or t.numel() != y.numel():
                raise VclozeHipError(f"{name}: contiguous tensors of equal size expected")
    args = [a.data_ptr()] + ([b.data_ptr()] if b is not None else []) + [y.data_ptr(), y.numel(), stream if stream is not None else cur_stream()]
    _check(fn(*args), name)


def mul(a, b, y, stream=None):
    _ew(lib().vc_mul, "vc_mul", a, b, y, stream)


def add(a, b, y, stream=None):
    _ew(lib().vc_add, "vc_add", a, b, y, stream)


def quick_gelu(x, y, stream=None):
    _ew(lib().vc_quick_gelu, "vc_quick_gelu", x, None, y, stream)


def transpose(src, dst, stream=None):
    _bf16(src, "src"); _bf16(dst, "dst")
    if src.dim() != 2 or src.stride(1) != 1 or dst.stride(1) != 1 or tuple(dst.shape) != (src.shape[1], src.shape[0]):
        raise VclozeHipError("transpose: [R, C] -> [C, R] with contiguous rows expected")
    _check(lib().vc_transpose(src.data_ptr(), src.stride(0), dst.data_ptr(), dst.stride(0), src.shape[0], src.shape[1],
                              stream if stream is not None else cur_stream()), "vc_transpose")


def nchw_to_nhwc(src, dst, div=1.0, add=0.0, stream=None):
    _bf16(dst, "dst")
    if src.dtype not in (torch.float32, torch.bfloat16) or not (src.is_contiguous() and dst.is_contiguous()):
        raise VclozeHipError("nchw_to_nhwc: contiguous f32/bf16 [C, H, W] source expected")
    Cc, HW = src.shape[0], src[0].numel()
    if dst.shape[0] != HW or dst.shape[1] < Cc:
        raise VclozeHipError("nchw_to_nhwc: dst [H*W, Cp >= C] expected")
    _check(lib().vc_nchw_to_nhwc(src.data_ptr(), int(src.dtype == torch.float32), dst.data_ptr(), Cc, dst.shape[1], HW,
                                 div, add, stream if stream is not None else cur_stream()), "vc_nchw_to_nhwc")


def nhwc_to_nchw(src, dst, stream=None):
    _bf16(src, "src")
    if dst.dtype not in (torch.float32, torch.bfloat16) or not (src.is_contiguous() and dst.is_contiguous()):
        raise VclozeHipError("nhwc_to_nchw: contiguous f32/bf16 [C, H, W] destination expected")
    Cc, HW = dst.shape[0], dst[0].numel()
    if src.shape[0] != HW or src.shape[1] < Cc:
        raise VclozeHipError("nhwc_to_nchw: src [H*W, Cp >= C] expected")
    _check(lib().vc_nhwc_to_nchw(src.data_ptr(), dst.data_ptr(), int(dst.dtype == torch.float32), Cc, src.shape[1], HW,
                                 stream if stream is not None else cur_stream()), "vc_nhwc_to_nchw")


def gaussian_sample(moments, noise, out, scale, shift, stream=None):
    _bf16(moments, "moments"); _bf16(out, "out")
    Z, HW = out.shape[0], out[0].numel()
    if noise is not None:
        _bf16(noise, "noise")
        if noise.shape != out.shape or not noise.is_contiguous():
            raise VclozeHipError("gaussian_sample: noise must match out [Z, h, w]")
    if moments.shape[0] != HW or moments.shape[1] < 2 * Z or not (moments.is_contiguous() and out.is_contiguous()):
        raise VclozeHipError("gaussian_sample: moments [H*W, Cp >= 2Z] and contiguous out [Z, h, w] expected")
    _check(lib().vc_gaussian_sample(moments.data_ptr(), moments.shape[1], _p(noise), out.data_ptr(), Z, HW, scale, shift,
                                    stream if stream is not None else cur_stream()), "vc_gaussian_sample")


class Graph:
    """hipGraph captured from the launches issued on `stream` inside the with-block."""

    def __init__(self, stream: int):
        self.stream = stream
        self.exec = C.c_void_p()

    def __enter__(self):
        _check(lib().vc_graph_begin(self.stream), "vc_graph_begin")
        return self

    def __exit__(self, et, ev, tb):
        rc = lib().vc_graph_end(self.stream, C.byref(self.exec))
        if et is None:
            _check(rc, "vc_graph_end")
        return False

    def launch(self, stream: Optional[int] = None):
        _check(lib().vc_graph_launch(self.exec, stream if stream is not None else self.stream), "vc_graph_launch")

    def __del__(self):
        try:
            if self.exec:
                lib().vc_graph_destroy(self.exec)
        except Exception:
            pass


class Event:
    def __init__(self):
        self.h = C.c_void_p()
        _check(lib().vc_event_create(C.byref(self.h)), "vc_event_create")

    def record(self, stream=None):
        _check(lib().vc_event_record(self.h, stream if stream is not None else cur_stream()), "vc_event_record")

    def elapsed_ms(self, stop: "Event") -> float:
        ms = C.c_float()
        _check(lib().vc_event_elapsed_ms(self.h, stop.h, C.byref(ms)), "vc_event_elapsed_ms")
        return ms.value

    def __del__(self):
        try:
            lib().vc_event_destroy(self.h)
        except Exception:
            pass
