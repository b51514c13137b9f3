// Small elementwise kernels of the denoising path (all HBM/launch-bound, bf16 storage, f32 math).
//   timestep_embedding  layers.py:28-49      silu / add3   MLPEmbedder + vec sum, model.py:102-107
//   concat_cols         transport.py:193-196 (x || cond)   euler_step   torchdiffeq fixed-grid Euler update
#include "common.h"
#include "vcloze_internal.h"

namespace {

__global__ void temb_kernel(const float* __restrict__ t, const float* __restrict__ freqs, bf16_t* __restrict__ out,
                            int n, int half, int round_t) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n * half) return;
  const int b = i / half, k = i % half;
  float tt = 1000.0f * t[b];
  if (round_t) tt = rbf(1000.0f * rbf(t[b]));  // reference multiplies a bf16 tensor: product rounds to bf16
  const float arg = tt * freqs[k];
  out[(long)b * 2 * half + k] = f2bf(cosf(arg));
  out[(long)b * 2 * half + half + k] = f2bf(sinf(arg));
}

__global__ void silu_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ y, long n) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) y[i] = f2bf(silu_f(bf2f(x[i])));
}

// y[i] = bf16(bf16(a[i] + b[i % bn]) + c[i % cn]): b, c broadcast over rows (vec = time + guidance + vector)
__global__ void add3_kernel(const bf16_t* __restrict__ a, const bf16_t* __restrict__ b, const bf16_t* __restrict__ c,
                            bf16_t* __restrict__ y, long n, long bn, long cn) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float v = rbf(bf2f(a[i]) + bf2f(b[i % bn]));
  if (c) v = v + bf2f(c[i % cn]);
  y[i] = f2bf(v);
}

// 16-B chunks; cx, cc multiples of 8
__global__ void concat_cols_kernel(const u32x4* __restrict__ x, int cxc, const u32x4* __restrict__ cond, int ccc,
                                   u32x4* __restrict__ out, long rows) {
  const int w = cxc + ccc;
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows * w) return;
  const long r = i / w;
  const int c = (int)(i % w);
  out[i] = (c < cxc) ? x[r * cxc + c] : cond[r * ccc + (c - cxc)];
}

__global__ void euler_kernel(bf16_t* __restrict__ x, const bf16_t* __restrict__ v, const float* __restrict__ dts,
                             const int* __restrict__ step_ptr, long n) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  // dt is a 0-dim f32 tensor in the reference; multiplying a bf16 tensor by it casts dt to the common dtype
  // (bf16) first, so the effective step is bf16(dt) (verified against torch: 100 % bitwise agreement)
  const float dt = rbf(dts[step_ptr ? *step_ptr : 0]);
  const float dy = rbf(dt * (-bf2f(v[i])));
  x[i] = f2bf(bf2f(x[i]) + dy);
}

// The same update for a caller whose ODE state is f32 (integrators.py:119 keeps the state's dtype): y1 = y0 + dt * f0 with
// f0 bf16 is f32(y0) + f32(bf16(bf16(dt) * f0)) in torch's type promotion; the bf16 shadow is what img_in reads next
// (its Linear rounds the f32 input to bf16 under autocast, visualcloze.py:363).  v == nullptr: refresh the shadow only.
__global__ void euler_f32_kernel(float* __restrict__ x32, bf16_t* __restrict__ shadow, const bf16_t* __restrict__ v,
                                 const float* __restrict__ dts, const int* __restrict__ step_ptr, long n) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float x = x32[i];
  if (v) {
    const float dt = rbf(dts[step_ptr ? *step_ptr : 0]);
    x += rbf(dt * (-bf2f(v[i])));
    x32[i] = x;
  }
  shadow[i] = f2bf(x);
}

// SDEdit start state (visualcloze.py:221): x0 = bf16(bf16(noise*(1-s)) + bf16(latent*s)), s a python float
__global__ void sdedit_mix_kernel(const bf16_t* __restrict__ noise, const bf16_t* __restrict__ latent, float s,
                                  bf16_t* __restrict__ out, long n) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  out[i] = f2bf(rbf(bf2f(noise[i]) * (1.0f - s)) + rbf(bf2f(latent[i]) * s));
}

// y[m, n] = bf16(act(x[m, n])) on row views: act 0 = GELU(tanh), 1 = SiLU
__global__ void act2d_kernel(const bf16_t* __restrict__ x, long ldx, bf16_t* __restrict__ y, long ldy, int rows, int cols, int act) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)rows * cols) return;
  const int m = (int)(i / cols), n = (int)(i % cols);
  const float v = bf2f(x[m * ldx + n]);
  y[m * ldy + n] = f2bf(act == 0 ? gelu_tanh(v) : silu_f(v));
}

// out[m, n] = bf16(res[m, n] + bf16(gate[n] * y[m, n])): the gated residual of layers.py:190-195,245 as its own pass
// (un-merged LoRA mode, where y = base + lora is only complete after a second GEMM)
__global__ void gate_residual_kernel(const bf16_t* __restrict__ y, long ldy, const bf16_t* __restrict__ res, long ldres,
                                     const bf16_t* __restrict__ gate, bf16_t* __restrict__ out, long ldo, int rows, int cols,
                                     const int* __restrict__ step_ptr, long gate_step_stride) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)rows * cols) return;
  const int m = (int)(i / cols), n = (int)(i % cols);
  const bf16_t* g = gate + (step_ptr ? (long)(*step_ptr) * gate_step_stride : 0);
  out[m * ldo + n] = f2bf(bf2f(res[m * ldres + n]) + rbf(bf2f(g[n]) * bf2f(y[m * ldy + n])));
}

__global__ void step_advance_kernel(int* step_ptr) { if (threadIdx.x == 0 && blockIdx.x == 0) *step_ptr += 1; }

}  // namespace

#define VC_CHECK_LAUNCH(name)                                                                     \
  do {                                                                                            \
    hipError_t e_ = hipGetLastError();                                                            \
    if (e_ != hipSuccess) { snprintf(err, errlen, name " launch: %s", hipGetErrorString(e_)); return VC_ERR_HIP; } \
    return VC_OK;                                                                                 \
  } while (0)

int vc_temb_launch(const float* t, const float* freqs, void* out, int n, int half, int round_t, hipStream_t s, char* err, int errlen) {
  if (!t || !freqs || !out || n <= 0 || half <= 0) { snprintf(err, errlen, "timestep_embedding: bad args"); return VC_ERR_ARG; }
  hipLaunchKernelGGL(temb_kernel, dim3((n * half + 255) / 256), dim3(256), 0, s, t, freqs, (bf16_t*)out, n, half, round_t);
  VC_CHECK_LAUNCH("timestep_embedding");
}
int vc_silu_launch(const void* x, void* y, int64_t n, hipStream_t s, char* err, int errlen) {
  if (!x || !y || n <= 0) { snprintf(err, errlen, "silu: bad args"); return VC_ERR_ARG; }
  hipLaunchKernelGGL(silu_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, (const bf16_t*)x, (bf16_t*)y, (long)n);
  VC_CHECK_LAUNCH("silu");
}
int vc_act2d_launch(const void* x, int64_t ldx, void* y, int64_t ldy, int32_t rows, int32_t cols, int32_t act, hipStream_t s,
                    char* err, int errlen) {
  if (!x || !y || rows <= 0 || cols <= 0 || act < 0 || act > 1) { snprintf(err, errlen, "act2d: bad args"); return VC_ERR_ARG; }
  const long n = (long)rows * cols;
  hipLaunchKernelGGL(act2d_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, (const bf16_t*)x, (long)ldx, (bf16_t*)y,
                     (long)ldy, rows, cols, act);
  VC_CHECK_LAUNCH("act2d");
}
int vc_gate_residual_launch(const void* y, int64_t ldy, const void* res, int64_t ldres, const void* gate, void* out, int64_t ldo,
                            int32_t rows, int32_t cols, const int32_t* step_ptr, int64_t gate_step_stride, hipStream_t s,
                            char* err, int errlen) {
  if (!y || !res || !gate || !out || rows <= 0 || cols <= 0) { snprintf(err, errlen, "gate_residual: bad args"); return VC_ERR_ARG; }
  const long n = (long)rows * cols;
  hipLaunchKernelGGL(gate_residual_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, (const bf16_t*)y, (long)ldy,
                     (const bf16_t*)res, (long)ldres, (const bf16_t*)gate, (bf16_t*)out, (long)ldo, rows, cols, step_ptr,
                     (long)gate_step_stride);
  VC_CHECK_LAUNCH("gate_residual");
}
int vc_add3_launch(const void* a, const void* b, const void* c, void* y, int64_t n, int64_t bn, int64_t cn, hipStream_t s, char* err, int errlen) {
  if (!a || !b || !y || n <= 0 || bn <= 0 || (c && cn <= 0)) { snprintf(err, errlen, "add3: bad args"); return VC_ERR_ARG; }
  hipLaunchKernelGGL(add3_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, (const bf16_t*)a, (const bf16_t*)b, (const bf16_t*)c, (bf16_t*)y, (long)n, (long)bn, (long)(c ? cn : 1));
  VC_CHECK_LAUNCH("add3");
}
int vc_concat_cols_launch(const void* x, int cx, const void* cond, int cc, void* out, int64_t rows, hipStream_t s, char* err, int errlen) {
  if (!x || !cond || !out || rows <= 0 || cx <= 0 || cc <= 0 || cx % 8 || cc % 8) { snprintf(err, errlen, "concat_cols: need cx, cc positive multiples of 8"); return VC_ERR_ARG; }
  const long n = rows * ((cx + cc) / 8);
  hipLaunchKernelGGL(concat_cols_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, (const u32x4*)x, cx / 8, (const u32x4*)cond, cc / 8, (u32x4*)out, (long)rows);
  VC_CHECK_LAUNCH("concat_cols");
}
int vc_euler_launch(void* x, const void* v, const float* dts, const int32_t* step_ptr, int64_t n, hipStream_t s, char* err, int errlen) {
  if (!x || !v || !dts || n <= 0) { snprintf(err, errlen, "euler_step: bad args"); return VC_ERR_ARG; }
  hipLaunchKernelGGL(euler_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, (bf16_t*)x, (const bf16_t*)v, dts, step_ptr, (long)n);
  VC_CHECK_LAUNCH("euler_step");
}
int vc_euler_f32_launch(float* x32, void* shadow, const void* v, const float* dts, const int32_t* step_ptr, int64_t n, hipStream_t s,
                        char* err, int errlen) {
  if (!x32 || !shadow || (v && !dts) || n <= 0) { snprintf(err, errlen, "euler_step_f32: bad args"); return VC_ERR_ARG; }
  hipLaunchKernelGGL(euler_f32_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, x32, (bf16_t*)shadow, (const bf16_t*)v, dts,
                     step_ptr, (long)n);
  VC_CHECK_LAUNCH("euler_step_f32");
}
int vc_step_advance_launch(int32_t* step_ptr, hipStream_t s, char* err, int errlen) {
  if (!step_ptr) { snprintf(err, errlen, "step_advance: null"); return VC_ERR_ARG; }
  hipLaunchKernelGGL(step_advance_kernel, dim3(1), dim3(64), 0, s, step_ptr);
  VC_CHECK_LAUNCH("step_advance");
}
int vc_sdedit_mix_launch(const void* noise, const void* latent, float strength, void* out, int64_t n, hipStream_t s, char* err, int errlen) {
  if (!noise || !latent || !out || n <= 0) { snprintf(err, errlen, "sdedit_mix: bad args"); return VC_ERR_ARG; }
  hipLaunchKernelGGL(sdedit_mix_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, (const bf16_t*)noise, (const bf16_t*)latent, strength, (bf16_t*)out, (long)n);
  VC_CHECK_LAUNCH("sdedit_mix");
}
