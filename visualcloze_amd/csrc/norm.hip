// HBM-bound normalisation kernels of the FLUX blocks on gfx950.
//
//  ln_modulate      LayerNorm(eps 1e-6, no affine) fused with AdaLN modulate, bf16 in/out, f32 math:
//                   y = bf16( bf16(1+scale) * LN(x) + shift )   (layers.py:163-164,191,195,234,257;
//                   under CUDA autocast LN returns f32 and `1 + scale` is a bf16 tensor)
//  qknorm_rope_vt   QKNorm (RMSNorm over head_dim 128, layers.py:63-84) + RoPE (math.py:112-117) in place on
//                   the q,k columns of the qkv rows, and V -> vt[b][h][d][Lpad] (keys >= L zero-filled).
// Both read/write 16 B per lane; one wave owns a whole row (LN) / 16 lanes own one head row (QKNorm).
#include "common.h"
#include "vcloze_internal.h"

namespace {

// One launch covers up to two row sets (the img and txt streams of a DoubleStreamBlock: different x / y / modulation
// rows, same D): rows [0, rows) belong to the first set, [rows, rows + B.rows) to the second.
struct LnStream2 {
  const bf16_t* x; long ldx; bf16_t* y; long ldy; const bf16_t* shift; const bf16_t* scale; int rows; int rows_per_batch;
};

template <int NCH>
__global__ __launch_bounds__(256) void ln_modulate_kernel(const bf16_t* __restrict__ x, long ldx, bf16_t* __restrict__ y,
                                                          long ldy, const bf16_t* __restrict__ shift,
                                                          const bf16_t* __restrict__ scale, long mod_bstride, int rows,
                                                          int D, int rows_per_batch, const int* __restrict__ step_ptr,
                                                          long mod_step_stride, const LnStream2 second) {
  const int lane = threadIdx.x & 63;
  int row = blockIdx.x * 4 + (threadIdx.x >> 6);   // wave-uniform
  if (row >= rows) {
    row -= rows;
    if (row >= second.rows) return;
    x = second.x; ldx = second.ldx; y = second.y; ldy = second.ldy; shift = second.shift; scale = second.scale;
    rows_per_batch = second.rows_per_batch;
  }
  long moff = (long)(row / rows_per_batch) * mod_bstride;
  if (step_ptr) moff += (long)(*step_ptr) * mod_step_stride;
  const bf16_t* xr = x + (long)row * ldx;
  float v[NCH][8];
  float sum = 0.f;
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int idx = c * 512 + lane * 8;
    if (idx < D) {
      const u32x4 w = *(const u32x4*)(xr + idx);
#pragma unroll
      for (int e = 0; e < 4; ++e) { v[c][2 * e] = lo_bf(w[e]); v[c][2 * e + 1] = hi_bf(w[e]); }
#pragma unroll
      for (int e = 0; e < 8; ++e) sum += v[c][e];
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[c][e] = 0.f;
    }
  }
  const float mean = wave_sum(sum) / (float)D;
  float ss = 0.f;
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int idx = c * 512 + lane * 8;
    if (idx < D) {
#pragma unroll
      for (int e = 0; e < 8; ++e) { const float d = v[c][e] - mean; ss += d * d; }
    }
  }
  const float rstd = 1.0f / sqrtf(wave_sum(ss) / (float)D + 1e-6f);
  bf16_t* yr = y + (long)row * ldy;
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int idx = c * 512 + lane * 8;
    if (idx < D) {
      const u32x4 sc = *(const u32x4*)(scale + moff + idx);
      const u32x4 sh = *(const u32x4*)(shift + moff + idx);
      u32x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float a0 = rbf(1.0f + lo_bf(sc[e])), a1 = rbf(1.0f + hi_bf(sc[e]));
        const float r0 = a0 * ((v[c][2 * e] - mean) * rstd) + lo_bf(sh[e]);
        const float r1 = a1 * ((v[c][2 * e + 1] - mean) * rstd) + hi_bf(sh[e]);
        o[e] = pack2bf(r0, r1);
      }
      *(u32x4*)(yr + idx) = o;
    }
  }
}

// grid (ceil(L/64), H, B), 256 threads
__global__ __launch_bounds__(256) void qknorm_rope_vt_kernel(bf16_t* __restrict__ qkv, long ld, long bstride,
                                                             const bf16_t* __restrict__ q_scale,
                                                             const bf16_t* __restrict__ k_scale,
                                                             const bf16_t* __restrict__ q_scale2,
                                                             const bf16_t* __restrict__ k_scale2, int split,
                                                             const float* __restrict__ rope, long rope_bstride,
                                                             bf16_t* __restrict__ vt, int L, int Lpad, int H, int parts) {
  __shared__ uint32_t tl[128 * 33];
  const int tid = threadIdx.x;
  const int t0 = blockIdx.x * 64, h = blockIdx.y, b = blockIdx.z;
  bf16_t* base = qkv + (long)b * bstride + h * 128;
  const int sub = tid & 15;  // 16 lanes x 8 elements = one head row

  // ---- phase 1: q and k rows ----
#pragma unroll 2
  for (int p = (parts & 1) ? 0 : 4; p < ((parts & 2) ? 8 : 4); ++p) {
    const int rowid = p * 16 + (tid >> 4);
    const int which = rowid >> 6;  // 0 = q, 1 = k
    const int tok = t0 + (rowid & 63);
    const bool ok = tok < L;
    bf16_t* ptr = base + (long)(ok ? tok : 0) * ld + which * (H * 128) + sub * 8;
    const u32x4 w = *(const u32x4*)ptr;
    const bf16_t* sc = (tok < split ? (which ? k_scale : q_scale) : (which ? k_scale2 : q_scale2)) + sub * 8;
    const u32x4 sw = *(const u32x4*)sc;
    float g[8];
#pragma unroll
    for (int e = 0; e < 4; ++e) { g[2 * e] = lo_bf(sw[e]); g[2 * e + 1] = hi_bf(sw[e]); }
    // RoPE on interleaved pairs; table [L][64][2] = (cos, sin)
    const float* rp = rope + (long)b * rope_bstride + (long)(ok ? tok : 0) * 128 + sub * 8;
    const f32x4 c0 = *(const f32x4*)rp;
    const f32x4 c1 = *(const f32x4*)(rp + 4);
    const float cs[8] = {c0[0], c0[1], c0[2], c0[3], c1[0], c1[1], c1[2], c1[3]};
    const u32x4 o = qknorm_rope8(w, g, cs, (which == 0 && (parts & 8)) ? VC_QK_PRESCALE : 1.0f);
    if (ok) *(u32x4*)ptr = o;
  }

  // ---- phase 2: V tile [64 tok][128 d] -> vt[d][t0 .. t0+63] ----
  if (!(parts & 4)) return;
  const bf16_t* vbase = base + 2 * (H * 128);
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int pc = i * 256 + tid;
    const int dch = pc & 15, tp = pc >> 4;  // token pair tp: tokens 2tp, 2tp+1
    const int ta = t0 + 2 * tp, tb = ta + 1;
    u32x4 wa = {0, 0, 0, 0}, wb = {0, 0, 0, 0};
    if (ta < L) wa = *(const u32x4*)(vbase + (long)ta * ld + dch * 8);
    if (tb < L) wb = *(const u32x4*)(vbase + (long)tb * ld + dch * 8);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const uint32_t lo = (wa[e] & 0xffffu) | (wb[e] << 16);
      const uint32_t hi = (wa[e] >> 16) | (wb[e] & 0xffff0000u);
      tl[(dch * 8 + 2 * e) * 33 + tp] = lo;
      tl[(dch * 8 + 2 * e + 1) * 33 + tp] = hi;
    }
  }
  __syncthreads();
  bf16_t* vout = vt + ((long)(b * H + h) * 128) * Lpad + t0;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = i * 256 + tid;
    const int d = c >> 3, part = c & 7;
    u32x4 w;
#pragma unroll
    for (int e = 0; e < 4; ++e) w[e] = tl[d * 33 + part * 4 + e];
    *(u32x4*)(vout + (long)d * Lpad + part * 8) = w;
  }
}

// QKNorm + RoPE of the q and / or k rows WITHOUT the V^T phase (V^T comes from the qkv GEMM's epilogue, VC_EPI_QKV): a
// thread owns (token, 8 of the 128 dims) and walks HG heads with them, so the (cos, sin) row of the token is read once
// per HG heads instead of once per head (the f32 table is twice the bytes of the bf16 row it rotates) and HG 16-B loads are in flight
// per lane.  Same arithmetic, in the same order, as phase 1 of qknorm_rope_vt_kernel: bit-identical results.
// grid (ceil(L/16), ceil(H/HG), B), 256 threads
template <int HG>
__global__ __launch_bounds__(256) void qknorm_rope_rows_kernel(bf16_t* __restrict__ qkv, long ld, long bstride,
                                                               const bf16_t* __restrict__ q_scale, const bf16_t* __restrict__ k_scale,
                                                               const bf16_t* __restrict__ q_scale2, const bf16_t* __restrict__ k_scale2,
                                                               int split, const float* __restrict__ rope, long rope_bstride, int L,
                                                               int H, int parts) {
  const int tid = threadIdx.x;
  const int tok = blockIdx.x * 16 + (tid >> 4), sub = tid & 15, h0 = blockIdx.y * HG, b = blockIdx.z;
  if (tok >= L) return;   // whole 16-lane groups leave together; the reductions below stay inside a group
  const float* rp = rope + (long)b * rope_bstride + (long)tok * 128 + sub * 8;
  const f32x4 c0 = *(const f32x4*)rp;
  const f32x4 c1 = *(const f32x4*)(rp + 4);
  const float cs[8] = {c0[0], c0[1], c0[2], c0[3], c1[0], c1[1], c1[2], c1[3]};
  bf16_t* row = qkv + (long)b * bstride + (long)tok * ld + sub * 8;
#pragma unroll
  for (int which = 0; which < 2; ++which) {      // 0 = q, 1 = k
    if (!(parts & (1 << which))) continue;
    const bf16_t* sc = (tok < split ? (which ? k_scale : q_scale) : (which ? k_scale2 : q_scale2)) + sub * 8;
    const u32x4 sw = *(const u32x4*)sc;
    float g[8];
#pragma unroll
    for (int e = 0; e < 4; ++e) { g[2 * e] = lo_bf(sw[e]); g[2 * e + 1] = hi_bf(sw[e]); }
    bf16_t* base = row + which * (H * 128) + h0 * 128;
    u32x4 w[HG];
#pragma unroll
    for (int hh = 0; hh < HG; ++hh) w[hh] = (h0 + hh < H) ? *(const u32x4*)(base + hh * 128) : u32x4{0u, 0u, 0u, 0u};
#pragma unroll
    for (int hh = 0; hh < HG; ++hh) {
      const u32x4 o = qknorm_rope8(w[hh], g, cs, (which == 0 && (parts & 8)) ? VC_QK_PRESCALE : 1.0f);
      if (h0 + hh < H) *(u32x4*)(base + hh * 128) = o;
    }
  }
}

}  // namespace

int vc_ln_modulate2_launch(const VcLnStream* a, const VcLnStream* b, int64_t mod_bstride, int32_t D, const int32_t* step_ptr,
                           int64_t mod_step_stride, hipStream_t s, char* err, int errlen) {
  if (!a) { snprintf(err, errlen, "ln_modulate: null stream description"); return VC_ERR_ARG; }
  const VcLnStream* set[2] = {a, b};
  for (int i = 0; i < 2; ++i) {
    const VcLnStream* q = set[i];
    if (!q) continue;
    if (!q->x || !q->y || !q->shift || !q->scale) { snprintf(err, errlen, "ln_modulate: null pointer"); return VC_ERR_ARG; }
    if (q->rows <= 0 || D <= 0 || q->rows_per_batch <= 0) { snprintf(err, errlen, "ln_modulate: empty input rows=%d D=%d", q->rows, D); return VC_ERR_ARG; }
    if (D % 8 || D > 4096 || q->ldx % 8 || q->ldy % 8 || mod_bstride % 8 || mod_step_stride % 8) {
      snprintf(err, errlen, "ln_modulate: D=%d must be a multiple of 8 and <= 4096 with 16-B aligned strides", D); return VC_ERR_ARG; }
  }
  LnStream2 second = {nullptr, 0, nullptr, 0, nullptr, nullptr, 0, 1};
  if (b) second = LnStream2{(const bf16_t*)b->x, (long)b->ldx, (bf16_t*)b->y, (long)b->ldy, (const bf16_t*)b->shift,
                            (const bf16_t*)b->scale, b->rows, b->rows_per_batch};
  const dim3 grid((a->rows + second.rows + 3) / 4), block(256);
  if (D <= 3072)
    hipLaunchKernelGGL(ln_modulate_kernel<6>, grid, block, 0, s, (const bf16_t*)a->x, (long)a->ldx, (bf16_t*)a->y, (long)a->ldy,
                       (const bf16_t*)a->shift, (const bf16_t*)a->scale, (long)mod_bstride, a->rows, D, a->rows_per_batch,
                       step_ptr, (long)mod_step_stride, second);
  else
    hipLaunchKernelGGL(ln_modulate_kernel<8>, grid, block, 0, s, (const bf16_t*)a->x, (long)a->ldx, (bf16_t*)a->y, (long)a->ldy,
                       (const bf16_t*)a->shift, (const bf16_t*)a->scale, (long)mod_bstride, a->rows, D, a->rows_per_batch,
                       step_ptr, (long)mod_step_stride, second);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { snprintf(err, errlen, "ln_modulate launch: %s", hipGetErrorString(e)); return VC_ERR_HIP; }
  return VC_OK;
}

int vc_ln_modulate_launch(const void* x, int64_t ldx, void* y, int64_t ldy, const void* shift, const void* scale,
                          int64_t mod_bstride, int32_t rows, int32_t D, int32_t rows_per_batch,
                          const int32_t* step_ptr, int64_t mod_step_stride, hipStream_t s, char* err, int errlen) {
  const VcLnStream a = {x, ldx, y, ldy, shift, scale, rows, rows_per_batch};
  return vc_ln_modulate2_launch(&a, nullptr, mod_bstride, D, step_ptr, mod_step_stride, s, err, errlen);
}

int vc_qknorm_rope_vt_launch(void* qkv, int64_t ld, int64_t bstride, const void* q_scale, const void* k_scale,
                             const void* q_scale2, const void* k_scale2, int32_t split, const float* rope, int64_t rope_bstride, void* vt, int32_t B, int32_t L, int32_t Lpad,
                             int32_t H, int32_t parts, hipStream_t s, char* err, int errlen) {
  if (!qkv || !q_scale || !k_scale || !rope || !vt) { snprintf(err, errlen, "qknorm_rope_vt: null pointer"); return VC_ERR_ARG; }
  if (parts <= 0 || parts > 15 || !(parts & 7) || ((parts & VC_QKN_QPRE) && !(parts & VC_QKN_Q))) {
    snprintf(err, errlen, "qknorm_rope_vt: parts=%d must be a non-empty subset of VC_QKN_Q | VC_QKN_K | VC_QKN_VT (+ VC_QKN_QPRE with VC_QKN_Q)", parts); return VC_ERR_ARG; }
  if (!q_scale2 || !k_scale2) { q_scale2 = q_scale; k_scale2 = k_scale; split = L; }
  if (B <= 0 || L <= 0 || H <= 0) { snprintf(err, errlen, "qknorm_rope_vt: empty problem"); return VC_ERR_ARG; }
  if (Lpad < L || Lpad % 64 || ld % 8 || bstride % 8) { snprintf(err, errlen, "qknorm_rope_vt: Lpad=%d must be a multiple of 64 >= L=%d; ld, bstride multiples of 8", Lpad, L); return VC_ERR_ARG; }
#ifndef VC_QKN_HG
#define VC_QKN_HG 2
#endif
#ifndef VC_QKN_TILE_ONLY          // analysis builds (tools/qkn_ab.py): the 64-token tile kernel for every `parts`
  if (!(parts & VC_QKN_VT)) {     // rows only: one (cos, sin) read per token for VC_QKN_HG heads
    constexpr int HG = VC_QKN_HG;
    hipLaunchKernelGGL(qknorm_rope_rows_kernel<HG>, dim3((L + 15) / 16, (H + HG - 1) / HG, B), dim3(256), 0, s, (bf16_t*)qkv, (long)ld,
                       (long)bstride, (const bf16_t*)q_scale, (const bf16_t*)k_scale, (const bf16_t*)q_scale2, (const bf16_t*)k_scale2,
                       split, rope, (long)rope_bstride, L, H, parts);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { snprintf(err, errlen, "qknorm_rope rows launch: %s", hipGetErrorString(e)); return VC_ERR_HIP; }
    return VC_OK;
  }
#endif
  const dim3 grid((L + 63) / 64, H, B), block(256);
  hipLaunchKernelGGL(qknorm_rope_vt_kernel, grid, block, 0, s, (bf16_t*)qkv, (long)ld, (long)bstride,
                     (const bf16_t*)q_scale, (const bf16_t*)k_scale, (const bf16_t*)q_scale2, (const bf16_t*)k_scale2, split,
                     rope, (long)rope_bstride, (bf16_t*)vt, L, Lpad, H, parts);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { snprintf(err, errlen, "qknorm_rope_vt launch: %s", hipGetErrorString(e)); return VC_ERR_HIP; }
  return VC_OK;
}
