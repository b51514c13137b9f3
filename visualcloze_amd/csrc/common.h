// Shared device helpers for the gfx950 (CDNA4) kernels of the VisualCloze denoising path.
// Everything here is wave64 / MI355X-only by design (no portability layer).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef unsigned short bf16_t;  // raw bf16 bits in HBM
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

#define VC_DEV __device__ __forceinline__

// round-to-nearest-even f32 -> bf16 (v_cvt_pk_bf16_f32 on gfx950)
VC_DEV bf16_t f2bf(float f) { return __builtin_bit_cast(bf16_t, (__bf16)f); }
VC_DEV float bf2f(bf16_t u) { return __builtin_bit_cast(float, ((uint32_t)u) << 16); }
// round an f32 through bf16 (mimics a bf16 tensor materialised by the reference under autocast)
VC_DEV float rbf(float f) { return bf2f(f2bf(f)); }
VC_DEV uint32_t pack2bf(float lo, float hi) { return (uint32_t)f2bf(lo) | ((uint32_t)f2bf(hi) << 16); }
VC_DEV float lo_bf(uint32_t u) { return __builtin_bit_cast(float, u << 16); }
VC_DEV float hi_bf(uint32_t u) { return __builtin_bit_cast(float, u & 0xffff0000u); }

VC_DEV float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// GELU(tanh) as torch.nn.GELU(approximate="tanh") defines it, 0.5*x*(1+tanh(u)) with u = sqrt(2/pi)*(x+0.044715x^3),
// evaluated in the algebraically identical form x * sigmoid(2u) = x / (1 + 2^(-2u*log2 e)): one v_exp_f32 + one
// v_rcp_f32 (1 ulp) instead of an IEEE divide; the result is rounded to bf16 (8 bits) right after.
VC_DEV float gelu_tanh(float x) {
  const float k0 = 0.7978845608028654f;  // sqrt(2/pi)
  const float k1 = 0.044715f;
  const float u = k0 * (x + k1 * x * x * x);
  const float e = __builtin_amdgcn_exp2f(-2.0f * 1.4426950408889634f * u);
  return x * __builtin_amdgcn_rcpf(1.0f + e);
}
// two at a time: the polynomial part maps onto v_pk_mul_f32 / v_pk_fma_f32 / v_pk_add_f32 (one issue slot per PAIR), only
// the exp and the rcp stay per element.  arg = -2*log2(e)*u = x * (kA + kB*x^2).
typedef float f32x2 __attribute__((ext_vector_type(2)));
VC_DEV f32x2 gelu_tanh2(f32x2 x) {
  const float kA = -2.0f * 1.4426950408889634f * 0.7978845608028654f, kB = kA * 0.044715f;
  const f32x2 a = (x * x * kB + kA) * x;
  f32x2 d;
  d[0] = __builtin_amdgcn_exp2f(a[0]);
  d[1] = __builtin_amdgcn_exp2f(a[1]);
  d = d + 1.0f;
  f32x2 r;
  r[0] = __builtin_amdgcn_rcpf(d[0]);
  r[1] = __builtin_amdgcn_rcpf(d[1]);
  return x * r;
}
VC_DEV float silu_f(float x) {
  return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x));
}

// the value of another lane of the same 16-lane row, selected by a DPP control (folded into the consuming VALU instruction)
template <int CTRL>
VC_DEV float dpp_f32(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}

// QKNorm (layers.py:63-84: x * rsqrt(mean(x^2) + 1e-6) -> bf16, * scale -> bf16) and RoPE on the interleaved pairs
// (math.py:112-117) of the 8 consecutive elements a lane owns of a 128-wide head row; the row's 16 lanes are neighbours
// (lane & 15 = position in the row).  ONE definition for the pre-pass kernels (norm.hip) and the qkv GEMM's epilogue
// (gemm.hip), with floating-point contraction OFF and every fused multiply-add WRITTEN OUT: the call sites give the same bits
// whatever the compiler would fuse around them.  In the GEMM's epilogue this function is VALU-bound (12 waves per CU, 5.3
// chunks per lane and tile: tools/qkv_epilogue_phases.py), so it is written for instruction count (round 5): the sum of squares
// and the rotation as explicit FMAs, rsqrt as the hardware's v_rsq_f32 (1 ulp; torch.rsqrt is no IEEE division either) - 96
// instead of 130 instructions per chunk.  Its rounding POINTS are the reference's: (x * rrms) -> bf16, * scale -> bf16, the
// rotated value -> bf16.
//
// `post` multiplies the rotated value before its ONE rounding to bf16: 1.0f (exact: the reference's q / k) or VC_QK_PRESCALE,
// the softmax scale 128^-0.5 * log2(e) folded into the QUERY rows, which the one-wave-per-SIMD attention kernel then loads
// straight into its MFMA operand registers (VcAttention.q_prescaled).
#define VC_QK_PRESCALE (0.08838834764831845f * 1.4426950408889634f)
VC_DEV u32x4 qknorm_rope8(const u32x4 w, const float (&g)[8], const float (&cs)[8], const float post = 1.0f) {
#pragma clang fp contract(off)
  float x[8];
#pragma unroll
  for (int e = 0; e < 4; ++e) { x[2 * e] = lo_bf(w[e]); x[2 * e + 1] = hi_bf(w[e]); }
  float ss = x[0] * x[0];
#pragma unroll
  for (int e = 1; e < 8; ++e) ss = __builtin_fmaf(x[e], x[e], ss);
  // the row's 16 lanes: an xor butterfly (every lane ends with the same bits) as four DPP adds - lanes ^1 and ^2 by quad_perm,
  // then, all four lanes of a quad being equal, ^4 = row_half_mirror and ^8 = row_mirror.  (__shfl_xor compiles to
  // ds_bpermute_b32: four dependent LDS round trips per chunk in a loop that runs 5 chunks per lane and tile.)
  ss += dpp_f32<0xB1>(ss);      // quad_perm [1, 0, 3, 2]
  ss += dpp_f32<0x4E>(ss);      // quad_perm [2, 3, 0, 1]
  ss += dpp_f32<0x141>(ss);     // row_half_mirror
  ss += dpp_f32<0x140>(ss);     // row_mirror
  const float rrms = __builtin_amdgcn_rsqf(__builtin_fmaf(ss, 1.0f / 128.0f, 1e-6f));
#pragma unroll
  for (int e = 0; e < 8; ++e) x[e] = rbf(rbf(x[e] * rrms) * g[e]);
  u32x4 o;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const float co = cs[2 * e], si = cs[2 * e + 1];
    const float re = __builtin_fmaf(co, x[2 * e], -(si * x[2 * e + 1])), im = __builtin_fmaf(si, x[2 * e], co * x[2 * e + 1]);
    o[e] = pack2bf(re * post, im * post);
  }
  return o;
}

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
// async HBM -> LDS copy, 16 B per lane; LDS destination = wave-uniform base + lane*16
VC_DEV void glds16(const void* g, void* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)lds_wave_base, 16, 0, 0);
}

// the same with a SCALAR base and a 32-bit per-lane byte offset (`global_load_lds v_off, s[base]`), M0 written in the same
// statement (one wait state before the DMA reads it): where the builtin is handed base + (scalar + lane offset) hipcc adds the
// scalar part per piece in the vector ALU - on SIMDs whose issue slots feed the matrix pipe
VC_DEV void glds16_saddr(const char* sbase, uint32_t voff, void* lds_wave_base) {
  const uint32_t lds_off = (uint32_t)(uintptr_t)(lptr_t)lds_wave_base;
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase), "s"(lds_off) : "memory", "m0");
}

// XCD-aware bijective block remap: hardware places block b on XCD b%8; give every XCD a contiguous
// chunk of logical ids so neighbouring tiles share that XCD's L2 (speed only, never correctness).
VC_DEV int xcd_remap(int bid, int nb) {
  const int q = nb >> 3, r = nb & 7;
  const int xcd = bid & 7, idx = bid >> 3;
  const int start = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return start + idx;
}
