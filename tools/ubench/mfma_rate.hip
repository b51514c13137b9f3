// Micro-benchmark: issue interval of v_mfma_f32_32x32x16_bf16 / 16x16x32_bf16 for independent and dependent chains,
// one wave per SIMD and two waves per SIMD (second wave: same MFMA stream, or a VALU/exp stream).
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_rate.hip -o tools/ubench/mfma_rate.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

template <int MODE>   // 0: 32x32x16, 4 independent accs; 1: 32x32x16 one acc (dependent); 2: 16x16x32 x8 independent; 3: 32x32x16 2 accs
__global__ __launch_bounds__(512) void k(unsigned long long* out, float* sink, int iters, int partner) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(float)(lane + i); b[i] = (__bf16)(float)(lane * 3 + i); }
  if (wave >= 4 && partner == 0) return;                 // one wave per SIMD
  if (wave >= 4 && partner == 2) {                       // partner = VALU / exp stream
    float x = lane * 0.001f, y = 0.f;
    for (int it = 0; it < iters * 40; ++it) {
#pragma unroll
      for (int j = 0; j < 16; ++j) { y += __builtin_amdgcn_exp2f(x); x = x * 0.999f + 0.0001f; }
    }
    if (y == 12345.f) sink[0] = y;
    return;
  }
  f32x16 c0 = {}, c1 = {}, c2 = {}, c3 = {};
  bf16x8 av[16], bv[8];
  for (int f = 0; f < 16; ++f) for (int i = 0; i < 8; ++i) av[f][i] = (__bf16)(float)(lane + i + f);
  for (int f = 0; f < 8; ++f) for (int i = 0; i < 8; ++i) bv[f][i] = (__bf16)(float)(lane * 3 + i + f);
  f32x16 cc[4] = {};
  f32x4 d[8] = {};
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c3, 0, 0, 0);
      }
    } else if (MODE == 4) {      // attention P pattern: 16 distinct A frags, B changes every 4, 4 accumulators
#pragma unroll
      for (int f = 0; f < 16; ++f) cc[f & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[f], bv[f >> 2], cc[f & 3], 0, 0, 0);
      asm volatile("" : "+v"(av[0]), "+v"(av[5]), "+v"(bv[1]));
    } else if (MODE == 5) {      // attention Q pattern: 16 distinct A frags, B changes every 2, 2 accumulators
#pragma unroll
      for (int f = 0; f < 16; ++f) cc[f & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[f], bv[f >> 1], cc[f & 1], 0, 0, 0);
      asm volatile("" : "+v"(av[0]), "+v"(av[5]), "+v"(bv[1]));
    } else if (MODE == 1) {
#pragma unroll
      for (int j = 0; j < 16; ++j) c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0);
    } else if (MODE == 3) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c1, 0, 0, 0);
      }
    } else {
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int q = 0; q < 8; ++q) d[q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, d[q], 0, 0, 0);
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = c0[0] + c1[1] + c2[2] + c3[3];
  for (int q = 0; q < 8; ++q) s += d[q][0];
  for (int q = 0; q < 4; ++q) s += cc[q][q];
  if (s == 12345.f) sink[0] = s;
  if (blockIdx.x == 0 && lane == 0 && wave == 0) out[0] = t1 - t0;
}

int main() {
  unsigned long long* out; float* sink;
  hipMalloc(&out, 64); hipMalloc(&sink, 64);
  const int iters = 2000;
  const char* names[6] = {"32x32x16 4 indep accs", "32x32x16 1 acc (dependent)", "16x16x32 8 indep accs", "32x32x16 2 accs", "attn P pattern", "attn Q pattern"};
  for (int partner = 0; partner < 3; ++partner)
    for (int mode = 0; mode < 6; ++mode) {
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      hipEventRecord(e0);
      if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(512), 0, 0, out, sink, iters, partner);
      if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(256), dim3(512), 0, 0, out, sink, iters, partner);
      if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(256), dim3(512), 0, 0, out, sink, iters, partner);
      if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(256), dim3(512), 0, 0, out, sink, iters, partner);
      if (mode == 4) hipLaunchKernelGGL(k<4>, dim3(256), dim3(512), 0, 0, out, sink, iters, partner);
      if (mode == 5) hipLaunchKernelGGL(k<5>, dim3(256), dim3(512), 0, 0, out, sink, iters, partner);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      unsigned long long cyc; hipMemcpy(&cyc, out, 8, hipMemcpyDeviceToHost);
      const double flop_per = mode == 2 ? 16384.0 : 32768.0;
      const double nwaves = 256.0 * (partner == 1 ? 8 : 4);
      printf("partner %d (%s) %-28s: %.1f memtime-ticks per MFMA, kernel %.3f ms -> %.0f TFLOP/s chip-wide\n", partner,
             partner == 0 ? "none" : partner == 1 ? "same MFMA stream" : "VALU exp stream", names[mode], (double)cyc / (iters * 16.0), ms,
             nwaves * iters * 16.0 * flop_per / (ms * 1e-3) / 1e12);
    }
  return 0;
}
