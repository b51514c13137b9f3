// VAE decoder support kernels (first slice): the FLUX AutoEncoder's Decoder
// (reference twin: models/modules/autoencoder.py:25-106,183-259) runs its convolutions and 1x1 projections on the bf16
// MFMA GEMM of gemm.hip; these kernels are the HBM-bound glue around it.  Activations are NHWC bf16: [H*W, C], C % 8 == 0.
//   im2col3x3     [Hs*Ws, C] -> [H*W, 9*C]: column (tap = dy*3+dx, c) of row (y, x) = src(y+dy-1, x+dx-1), zero outside;
//                 `up` = 1 folds F.interpolate(scale_factor=2, mode="nearest") (Upsample, :98-106) into the gather
//   groupnorm     nn.GroupNorm(32, C, eps=1e-6, affine) over [H*W, C] (+ swish, :21-22): f32 statistics by a deterministic
//                 two-level reduction, y = bf16(.) and, if asked, bf16(y * sigmoid(y))
//   softmax_rows  softmax(scale * x) per row, f32 internal, bf16 in place   (scaled_dot_product_attention, :47)
//   transpose     [R, Cc] -> [Cc, R] (V^T for the P.V GEMM)
//   nchw <-> nhwc layout changes at the two ends (latent in, image out), with the affine `z / scale + shift` (:306-307)
#include "common.h"
#include "vcloze_internal.h"

namespace {

// mode 0: same size, zero padding 1.  mode 1: source is the half-resolution map read at (yy>>1, xx>>1) (nearest 2x
// upsampling folded in).  mode 2: stride-2 convolution of the source padded by one zero row/column at the bottom/right
// (Downsample.forward, autoencoder.py:91-95): tap (dy, dx) of output (y, x) reads source (2y+dy, 2x+dx).
__global__ void im2col3x3_kernel(const bf16_t* __restrict__ src, bf16_t* __restrict__ dst, int H, int W, int C, int mode) {
  const int cpr = C >> 3;                                   // 16-B chunks per pixel
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long total = (long)H * W * 9 * cpr;
  if (i >= total) return;
  const int c8 = (int)(i % cpr);
  const long rt = i / cpr;
  const int tap = (int)(rt % 9);
  const long row = rt / 9;
  const int y = (int)(row / W), x = (int)(row % W);
  u32x4 v = {0u, 0u, 0u, 0u};
  if (mode == 2) {
    const int Hs = 2 * H, Ws = 2 * W;
    const int yy = 2 * y + tap / 3, xx = 2 * x + tap % 3;
    if (yy < Hs && xx < Ws) v = *(const u32x4*)(src + ((long)yy * Ws + xx) * C + c8 * 8);
  } else {
    const int yy = y + tap / 3 - 1, xx = x + tap % 3 - 1;
    if (yy >= 0 && yy < H && xx >= 0 && xx < W) {
      const int Ws = W >> mode;
      v = *(const u32x4*)(src + ((long)(yy >> mode) * Ws + (xx >> mode)) * C + c8 * 8);
    }
  }
  *(u32x4*)(dst + (row * 9 + tap) * C + c8 * 8) = v;
}

// ---- GroupNorm: partial sums per block of GN_ROWS rows, fixed-order finalize, apply ----
constexpr int GN_ROWS = 128;

__global__ __launch_bounds__(256) void gn_partial_kernel(const bf16_t* __restrict__ x, float* __restrict__ part, long HW, int C, int G) {
  // 256 % (C/8) == 0, so a thread always meets the same 8-channel chunk; it keeps (sum, sum of squares) for the up to
  // four groups inside that chunk in registers and the block adds the per-thread partials in a FIXED order: no float
  // atomics anywhere, the statistics are bit-reproducible
  __shared__ float ps[256][8];
  const int tid = threadIdx.x;
  const int cpr = C >> 3, cpg = C / G;
  const int sub = cpg >= 8 ? 8 : cpg;                     // channels of one group inside a chunk
  float sm[4] = {0.f, 0.f, 0.f, 0.f}, sq[4] = {0.f, 0.f, 0.f, 0.f};
  const long r0 = (long)blockIdx.x * GN_ROWS;
  const long r1 = min(r0 + GN_ROWS, HW);
  for (long i = r0 * cpr + tid; i < r1 * cpr; i += 256) {
    const u32x4 w = *(const u32x4*)(x + i * 8);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float v = (e & 1) ? hi_bf(w[e >> 1]) : lo_bf(w[e >> 1]);
      const int j = e / sub;
      sm[j] += v;
      sq[j] += v * v;
    }
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) { ps[tid][j] = sm[j]; ps[tid][4 + j] = sq[j]; }
  __syncthreads();
  if (tid < 2 * G) {
    const int g = tid >> 1, which = tid & 1;
    const int c_lo = (g * cpg) >> 3, c_hi = ((g + 1) * cpg - 1) >> 3;   // chunks the group touches
    const int j = cpg >= 8 ? 0 : ((g * cpg) & 7) / cpg;
    float acc = 0.f;
    for (int c = c_lo; c <= c_hi; ++c)
      for (int t = c; t < 256; t += cpr) acc += ps[t][which * 4 + j];
    part[(long)blockIdx.x * 2 * G + tid] = acc;
  }
}

// stats[2g] = mean, stats[2g+1] = rstd.  One 256-thread block per group: thread t adds partials t, t+256, ... in that
// order, then a fixed-shape tree over the 256 thread sums - deterministic, and parallel enough for the 1152 partial
// blocks of a 384x384 map.
__global__ __launch_bounds__(256) void gn_finalize_kernel(const float* __restrict__ part, float* __restrict__ stats, int nblk, int G, float inv_n, float eps) {
  __shared__ double rs[256], rq[256];
  const int g = blockIdx.x, t = threadIdx.x;
  double s = 0.0, q = 0.0;
  for (int b = t; b < nblk; b += 256) { s += part[(long)b * 2 * G + 2 * g]; q += part[(long)b * 2 * G + 2 * g + 1]; }
  rs[t] = s; rq[t] = q;
  __syncthreads();
#pragma unroll
  for (int w = 128; w > 0; w >>= 1) {
    if (t < w) { rs[t] += rs[t + w]; rq[t] += rq[t + w]; }
    __syncthreads();
  }
  if (t == 0) {
    const double mean = rs[0] * inv_n;
    const double var = rq[0] * inv_n - mean * mean;
    stats[2 * g] = (float)mean;
    stats[2 * g + 1] = (float)(1.0 / sqrt((var > 0.0 ? var : 0.0) + (double)eps));
  }
}

__global__ void gn_apply_kernel(const bf16_t* __restrict__ x, const float* __restrict__ stats, const bf16_t* __restrict__ gamma,
                                const bf16_t* __restrict__ beta, bf16_t* __restrict__ y, long HW, int C, int G, int swish) {
  const int cpr = C >> 3, cpg = C / G;
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= HW * cpr) return;
  const int c8 = (int)(i % cpr);
  const u32x4 w = *(const u32x4*)(x + i * 8);
  const u32x4 gw = *(const u32x4*)(gamma + c8 * 8);
  const u32x4 bw = *(const u32x4*)(beta + c8 * 8);
  u32x4 o;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    float r[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int c = c8 * 8 + 2 * e + k;
      const int g = c / cpg;
      const float xv = k ? hi_bf(w[e]) : lo_bf(w[e]);
      const float ga = k ? hi_bf(gw[e]) : lo_bf(gw[e]);
      const float be = k ? hi_bf(bw[e]) : lo_bf(bw[e]);
      float t = rbf((xv - stats[2 * g]) * stats[2 * g + 1] * ga + be);
      if (swish) t = t * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * t));
      r[k] = t;
    }
    o[e] = pack2bf(r[0], r[1]);
  }
  *(u32x4*)(y + i * 8) = o;
}

// one 256-thread block per row; cols <= 256 * MAXV (MAXV picked from cols at launch, up to 64).  v = bf16(scale * x) [+ bias, rounded to bf16 again]; causal: columns
// j > (row % causal_period) are masked (CLIP text).  softmax in f32, result bf16 in place.
template <int MAXV>   // register slots per thread: cols <= 256 * MAXV
__global__ __launch_bounds__(256) void softmax_rows_kernel(bf16_t* __restrict__ x, long ld, int cols, float scale,
                                                           const bf16_t* __restrict__ bias, long ldb, int causal_period) {
  __shared__ float red[8];
  bf16_t* row = x + (long)blockIdx.x * ld;
  const bf16_t* brow = bias ? bias + (long)blockIdx.x * ldb : nullptr;
  const int tid = threadIdx.x;
  const int limit = causal_period > 0 ? min(cols, (int)(blockIdx.x % causal_period) + 1) : cols;
  float v[MAXV];
  float mx = -INFINITY;
#pragma unroll
  for (int k = 0; k < MAXV; ++k) {
    const int c = k * 256 + tid;
    float t = -INFINITY;
    if (c < limit) {
      t = bf2f(row[c]) * scale;
      if (scale != 1.0f) t = rbf(t);
      if (brow) t = rbf(t + bf2f(brow[c]));
    }
    v[k] = t;
    mx = fmaxf(mx, t);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
  if ((tid & 63) == 0) red[tid >> 6] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < MAXV; ++k) {
    v[k] = __builtin_amdgcn_exp2f((v[k] - mx) * 1.4426950408889634f);
    s += v[k];
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if ((tid & 63) == 0) red[4 + (tid >> 6)] = s;
  __syncthreads();
  const float inv = 1.0f / (red[4] + red[5] + red[6] + red[7]);
#pragma unroll
  for (int k = 0; k < MAXV; ++k) {
    const int c = k * 256 + tid;
    if (c < cols) row[c] = f2bf(v[k] * inv);
  }
}

__global__ void transpose_kernel(const bf16_t* __restrict__ src, long lds_, bf16_t* __restrict__ dst, long ldd, int R, int Cc) {
  __shared__ bf16_t t[32][33];
  const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 256 threads: 8 rows per pass
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int r = r0 + ty + k * 8, c = c0 + tx;
    t[ty + k * 8][tx] = (r < R && c < Cc) ? src[(long)r * lds_ + c] : (bf16_t)0;
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int c = c0 + ty + k * 8, r = r0 + tx;
    if (c < Cc && r < R) dst[(long)c * ldd + r] = t[tx][ty + k * 8];
  }
}

// dst[(y*W+x)*Cp + c] = bf16(src[c][y][x] / div + add) for c < C (a bf16 source rounds the quotient to bf16 first, as
// torch does for `z / scale_factor + shift_factor` on a bf16 tensor), 0 for C <= c < Cp
__global__ void nchw_to_nhwc_kernel(const void* __restrict__ src, int src_f32, bf16_t* __restrict__ dst, int C, int Cp, long HW, float div, float add) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= HW * Cp) return;
  const int c = (int)(i % Cp);
  const long p = i / Cp;
  float v = 0.f;
  if (c < C) {
    v = src_f32 ? ((const float*)src)[(long)c * HW + p] / div : rbf(bf2f(((const bf16_t*)src)[(long)c * HW + p]) / div);
    v = v + add;
  }
  dst[i] = f2bf(v);
}

// dst[c][p] = src[p*Cp + c] for c < C
__global__ void nhwc_to_nchw_kernel(const bf16_t* __restrict__ src, void* __restrict__ dst, int dst_f32, int C, int Cp, long HW) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= HW * C) return;
  const int c = (int)(i / HW);
  const long p = i % HW;
  const bf16_t v = src[p * Cp + c];
  if (dst_f32) ((float*)dst)[i] = bf2f(v); else ((bf16_t*)dst)[i] = v;
}

// DiagonalGaussian.forward + AutoEncoder.encode (autoencoder.py:268-275, :301-304): moments [HW, Cp] NHWC with mean in
// channels [0, Z) and logvar in [Z, 2Z); out[c][p] = scale * ((mean + exp(0.5*logvar) * noise[c][p]) - shift), every
// intermediate rounded to bf16 like the reference's bf16 tensors; noise == nullptr -> the mean (sample=False).
__global__ void gaussian_sample_kernel(const bf16_t* __restrict__ mom, int Cp, const bf16_t* __restrict__ noise,
                                       bf16_t* __restrict__ out, int Z, long HW, float scale, float shift) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= HW * Z) return;
  const int c = (int)(i / HW);
  const long p = i % HW;
  float z = bf2f(mom[p * Cp + c]);
  if (noise) {
    const float lv = bf2f(mom[p * Cp + Z + c]);
    const float sd = rbf(expf(rbf(0.5f * lv)));
    z = rbf(z + rbf(sd * bf2f(noise[i])));
  }
  out[i] = f2bf(scale * rbf(z - shift));
}

}  // namespace

#define VAE_LAUNCH_CHECK(what)                                                                   \
  do { hipError_t e_ = hipGetLastError();                                                        \
       if (e_ != hipSuccess) { snprintf(err, errlen, what " launch: %s", hipGetErrorString(e_)); return VC_ERR_HIP; } } while (0)

int vc_im2col3x3_launch(const void* src, void* dst, int H, int W, int C, int up, hipStream_t s, char* err, int errlen) {
  if (!src || !dst) { snprintf(err, errlen, "im2col3x3: null pointer"); return VC_ERR_ARG; }
  if (H <= 0 || W <= 0 || C <= 0 || C % 8 || up < 0 || up > 2 || (up == 1 && ((H | W) & 1))) {
    snprintf(err, errlen, "im2col3x3: bad shape H=%d W=%d C=%d mode=%d (C %% 8 == 0; even H, W when upsampling)", H, W, C, up); return VC_ERR_ARG; }
  const long total = (long)H * W * 9 * (C >> 3);
  hipLaunchKernelGGL(im2col3x3_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, (const bf16_t*)src, (bf16_t*)dst, H, W, C, up);
  VAE_LAUNCH_CHECK("im2col3x3");
  return VC_OK;
}

int vc_groupnorm_launch(const void* x, const void* gamma, const void* beta, void* y, void* scratch, int64_t scratch_bytes,
                        int64_t HW, int C, int G, float eps, int swish, hipStream_t s, char* err, int errlen) {
  if (!x || !gamma || !beta || !y || !scratch) { snprintf(err, errlen, "groupnorm: null pointer"); return VC_ERR_ARG; }
  if (HW <= 0 || C <= 0 || G <= 0 || G > 64 || C % 8 || C % G || 256 % (C / 8) || C / G < 2 || (C / G < 8 && 8 % (C / G)) || (C / G >= 8 && (C / G) % 8)) {
    snprintf(err, errlen, "groupnorm: unsupported shape HW=%ld C=%d G=%d (C/8 must divide 256; C/G in {2,4,8,16,...})", (long)HW, C, G); return VC_ERR_ARG; }
  const int nblk = (int)((HW + GN_ROWS - 1) / GN_ROWS);
  const int64_t need = ((int64_t)nblk + 1) * 2 * G * (int64_t)sizeof(float);
  if (scratch_bytes < need) { snprintf(err, errlen, "groupnorm: scratch too small (%ld < %ld bytes)", (long)scratch_bytes, (long)need); return VC_ERR_ARG; }
  float* stats = (float*)scratch;
  float* part = stats + 2 * G;
  hipLaunchKernelGGL(gn_partial_kernel, dim3(nblk), dim3(256), 0, s, (const bf16_t*)x, part, (long)HW, C, G);
  hipLaunchKernelGGL(gn_finalize_kernel, dim3(G), dim3(256), 0, s, part, stats, nblk, G, 1.0f / ((float)HW * (float)(C / G)), eps);
  const long chunks = HW * (C >> 3);
  hipLaunchKernelGGL(gn_apply_kernel, dim3((unsigned)((chunks + 255) / 256)), dim3(256), 0, s, (const bf16_t*)x, stats,
                     (const bf16_t*)gamma, (const bf16_t*)beta, (bf16_t*)y, (long)HW, C, G, swish);
  VAE_LAUNCH_CHECK("groupnorm");
  return VC_OK;
}

int vc_softmax_rows_launch(void* x, int64_t ld, int rows, int cols, float scale, const void* bias, int64_t ldb, int causal_period,
                           hipStream_t s, char* err, int errlen) {
  if (!x) { snprintf(err, errlen, "softmax_rows: null pointer"); return VC_ERR_ARG; }
  if (rows <= 0 || cols <= 0 || cols > 256 * 64 || ld < cols || (bias && ldb < cols) || causal_period < 0) {
    snprintf(err, errlen, "softmax_rows: bad shape rows=%d cols=%d (cols <= 16384)", rows, cols); return VC_ERR_ARG; }
  auto fn = cols <= 512 ? softmax_rows_kernel<2> : cols <= 2048 ? softmax_rows_kernel<8> : cols <= 8192 ? softmax_rows_kernel<32> : softmax_rows_kernel<64>;
  hipLaunchKernelGGL(fn, dim3(rows), dim3(256), 0, s, (bf16_t*)x, (long)ld, cols, scale, (const bf16_t*)bias, (long)ldb, causal_period);
  VAE_LAUNCH_CHECK("softmax_rows");
  return VC_OK;
}

int vc_transpose_launch(const void* src, int64_t lds_, void* dst, int64_t ldd, int R, int Cc, hipStream_t s, char* err, int errlen) {
  if (!src || !dst) { snprintf(err, errlen, "transpose: null pointer"); return VC_ERR_ARG; }
  if (R <= 0 || Cc <= 0 || lds_ < Cc || ldd < R) { snprintf(err, errlen, "transpose: bad shape R=%d C=%d", R, Cc); return VC_ERR_ARG; }
  hipLaunchKernelGGL(transpose_kernel, dim3((Cc + 31) / 32, (R + 31) / 32), dim3(256), 0, s, (const bf16_t*)src, (long)lds_, (bf16_t*)dst, (long)ldd, R, Cc);
  VAE_LAUNCH_CHECK("transpose");
  return VC_OK;
}

int vc_nchw_to_nhwc_launch(const void* src, int src_f32, void* dst, int C, int Cp, int64_t HW, float div, float add, hipStream_t s, char* err, int errlen) {
  if (!src || !dst) { snprintf(err, errlen, "nchw_to_nhwc: null pointer"); return VC_ERR_ARG; }
  if (C <= 0 || Cp < C || HW <= 0 || div == 0.0f) { snprintf(err, errlen, "nchw_to_nhwc: bad shape C=%d Cp=%d HW=%ld (div != 0)", C, Cp, (long)HW); return VC_ERR_ARG; }
  const long total = HW * Cp;
  hipLaunchKernelGGL(nchw_to_nhwc_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, src, src_f32, (bf16_t*)dst, C, Cp, (long)HW, div, add);
  VAE_LAUNCH_CHECK("nchw_to_nhwc");
  return VC_OK;
}

int vc_nhwc_to_nchw_launch(const void* src, void* dst, int dst_f32, int C, int Cp, int64_t HW, hipStream_t s, char* err, int errlen) {
  if (!src || !dst) { snprintf(err, errlen, "nhwc_to_nchw: null pointer"); return VC_ERR_ARG; }
  if (C <= 0 || Cp < C || HW <= 0) { snprintf(err, errlen, "nhwc_to_nchw: bad shape C=%d Cp=%d HW=%ld", C, Cp, (long)HW); return VC_ERR_ARG; }
  const long total = HW * C;
  hipLaunchKernelGGL(nhwc_to_nchw_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, (const bf16_t*)src, dst, dst_f32, C, Cp, (long)HW);
  VAE_LAUNCH_CHECK("nhwc_to_nchw");
  return VC_OK;
}

int vc_gaussian_sample_launch(const void* moments, int Cp, const void* noise, void* out, int Z, int64_t HW, float scale, float shift,
                              hipStream_t s, char* err, int errlen) {
  if (!moments || !out) { snprintf(err, errlen, "gaussian_sample: null pointer"); return VC_ERR_ARG; }
  if (Z <= 0 || Cp < 2 * Z || HW <= 0) { snprintf(err, errlen, "gaussian_sample: bad shape Z=%d Cp=%d HW=%ld", Z, Cp, (long)HW); return VC_ERR_ARG; }
  const long total = HW * Z;
  hipLaunchKernelGGL(gaussian_sample_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, (const bf16_t*)moments, Cp,
                     (const bf16_t*)noise, (bf16_t*)out, Z, (long)HW, scale, shift);
  VAE_LAUNCH_CHECK("gaussian_sample");
  return VC_OK;
}
