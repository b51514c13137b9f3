// Text-encoder glue kernels (T5-XXL encoder and CLIP-L text model, reference call site
// models/modules/conditioner.py:5-37 -> transformers T5EncoderModel / CLIPTextModel).  The projections and the per-head
// attention products run on the bf16 MFMA GEMM; these are the HBM-bound pieces between them.  Rounding points follow the
// transformers modules run in bfloat16.
//   embedding     out[i, :] = table[ids[i], :]                                   (nn.Embedding)
//   rmsnorm       y = bf16(w * bf16(x * rsqrt(mean(x^2) + eps)))                 (T5LayerNorm: no mean, no bias, f32 stats)
//   layernorm     y = bf16(LN(x) * w + b), f32 statistics                         (nn.LayerNorm, CLIP)
//   mul / add     elementwise bf16                                                (T5 gated FF product; CLIP token + position)
//   quick_gelu    y = bf16(x * bf16(sigmoid(bf16(1.702 * x))))                    (CLIP hidden_act)
#include "common.h"
#include "vcloze_internal.h"

namespace {

__global__ void embedding_kernel(const int32_t* __restrict__ ids, const bf16_t* __restrict__ table, long ldt, int V,
                                 bf16_t* __restrict__ out, int L, int D) {
  const int cpr = D >> 3;
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)L * cpr) return;
  const int row = (int)(i / cpr), c8 = (int)(i % cpr);
  int id = ids[row];
  id = id < 0 ? 0 : (id >= V ? V - 1 : id);
  *(u32x4*)(out + (long)row * D + c8 * 8) = *(const u32x4*)(table + (long)id * ldt + c8 * 8);
}

// one wave per row; D <= 64 * 8 * NV elements, 16-B chunks strided across the wave
template <int AFFINE_LN>
__global__ __launch_bounds__(256) void rownorm_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ w,
                                                      const bf16_t* __restrict__ b, bf16_t* __restrict__ y, int rows, int D, float eps) {
  constexpr int NV = 8;                        // up to 8 chunks of 8 elements per lane -> D <= 4096
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int cpr = D >> 3;
  float v[NV][8];
  float s = 0.f, q = 0.f;
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    const int c8 = k * 64 + lane;
    if (c8 < cpr) {
      const u32x4 u = *(const u32x4*)(x + (long)row * D + c8 * 8);
#pragma unroll
      for (int e = 0; e < 4; ++e) { v[k][2 * e] = lo_bf(u[e]); v[k][2 * e + 1] = hi_bf(u[e]); }
#pragma unroll
      for (int e = 0; e < 8; ++e) { s += v[k][e]; q += v[k][e] * v[k][e]; }
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { s += __shfl_xor(s, o, 64); q += __shfl_xor(q, o, 64); }
  const float inv_d = 1.0f / (float)D;
  float mean = 0.f, rstd;
  if (AFFINE_LN) {
    mean = s * inv_d;
    float var = 0.f;                            // second pass over the registers: sum (x - mean)^2
#pragma unroll
    for (int k = 0; k < NV; ++k)
      if (k * 64 + lane < cpr)
#pragma unroll
        for (int e = 0; e < 8; ++e) { const float d = v[k][e] - mean; var += d * d; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) var += __shfl_xor(var, o, 64);
    rstd = 1.0f / sqrtf(var * inv_d + eps);
  } else {
    rstd = 1.0f / sqrtf(q * inv_d + eps);
  }
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    const int c8 = k * 64 + lane;
    if (c8 < cpr) {
      const u32x4 wu = *(const u32x4*)(w + c8 * 8);
      u32x4 bu = {0u, 0u, 0u, 0u};
      if (AFFINE_LN) bu = *(const u32x4*)(b + c8 * 8);
      u32x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float r[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const float xv = v[k][2 * e + h];
          const float wv = h ? hi_bf(wu[e]) : lo_bf(wu[e]);
          if (AFFINE_LN) r[h] = (xv - mean) * rstd * wv + (h ? hi_bf(bu[e]) : lo_bf(bu[e]));
          else r[h] = wv * rbf(xv * rstd);
        }
        o[e] = pack2bf(r[0], r[1]);
      }
      *(u32x4*)(y + (long)row * D + c8 * 8) = o;
    }
  }
}

// op 0: a*b, op 1: a+b, op 2: quick_gelu(a)
__global__ void ewise_kernel(const bf16_t* __restrict__ a, const bf16_t* __restrict__ b, bf16_t* __restrict__ y, long n8, int op) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n8) return;
  const u32x4 ua = *(const u32x4*)(a + i * 8);
  u32x4 ub = {0u, 0u, 0u, 0u};
  if (op != 2) ub = *(const u32x4*)(b + i * 8);
  u32x4 o;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    float r[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const float av = h ? hi_bf(ua[e]) : lo_bf(ua[e]);
      const float bv = h ? hi_bf(ub[e]) : lo_bf(ub[e]);
      if (op == 0) r[h] = av * bv;
      else if (op == 1) r[h] = av + bv;
      else {
        const float t = rbf(1.702f * av);
        r[h] = av * rbf(1.0f / (1.0f + expf(-t)));
      }
    }
    o[e] = pack2bf(r[0], r[1]);
  }
  *(u32x4*)(y + i * 8) = o;
}

}  // namespace

#define TXT_LAUNCH_CHECK(what)                                                                   \
  do { hipError_t e_ = hipGetLastError();                                                        \
       if (e_ != hipSuccess) { snprintf(err, errlen, what " launch: %s", hipGetErrorString(e_)); return VC_ERR_HIP; } } while (0)

int vc_embedding_launch(const int32_t* ids, const void* table, int64_t ldt, int V, void* out, int L, int D, hipStream_t s, char* err, int errlen) {
  if (!ids || !table || !out) { snprintf(err, errlen, "embedding: null pointer"); return VC_ERR_ARG; }
  if (L <= 0 || D <= 0 || D % 8 || V <= 0 || ldt < D || ldt % 8) { snprintf(err, errlen, "embedding: bad shape L=%d D=%d V=%d", L, D, V); return VC_ERR_ARG; }
  const long total = (long)L * (D >> 3);
  hipLaunchKernelGGL(embedding_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, ids, (const bf16_t*)table, (long)ldt, V, (bf16_t*)out, L, D);
  TXT_LAUNCH_CHECK("embedding");
  return VC_OK;
}

int vc_rownorm_launch(const void* x, const void* w, const void* b, void* y, int rows, int D, float eps, int affine_ln, hipStream_t s, char* err, int errlen) {
  if (!x || !w || !y || (affine_ln && !b)) { snprintf(err, errlen, "rmsnorm/layernorm: null pointer"); return VC_ERR_ARG; }
  if (rows <= 0 || D <= 0 || D % 8 || D > 4096) { snprintf(err, errlen, "rmsnorm/layernorm: rows=%d D=%d (D %% 8 == 0, D <= 4096)", rows, D); return VC_ERR_ARG; }
  if (affine_ln) hipLaunchKernelGGL(rownorm_kernel<1>, dim3((rows + 3) / 4), dim3(256), 0, s, (const bf16_t*)x, (const bf16_t*)w, (const bf16_t*)b, (bf16_t*)y, rows, D, eps);
  else hipLaunchKernelGGL(rownorm_kernel<0>, dim3((rows + 3) / 4), dim3(256), 0, s, (const bf16_t*)x, (const bf16_t*)w, (const bf16_t*)b, (bf16_t*)y, rows, D, eps);
  TXT_LAUNCH_CHECK("rmsnorm/layernorm");
  return VC_OK;
}

int vc_ewise_launch(const void* a, const void* b, void* y, int64_t n, int op, hipStream_t s, char* err, int errlen) {
  if (!a || !y || (op != 2 && !b)) { snprintf(err, errlen, "elementwise: null pointer"); return VC_ERR_ARG; }
  if (n <= 0 || n % 8) { snprintf(err, errlen, "elementwise: n=%ld must be a positive multiple of 8", (long)n); return VC_ERR_ARG; }
  const long n8 = n >> 3;
  hipLaunchKernelGGL(ewise_kernel, dim3((unsigned)((n8 + 255) / 256)), dim3(256), 0, s, (const bf16_t*)a, (const bf16_t*)b, (bf16_t*)y, n8, op);
  TXT_LAUNCH_CHECK("elementwise");
  return VC_OK;
}
