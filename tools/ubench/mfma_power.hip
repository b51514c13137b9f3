// Micro-benchmark: what the board's power management lets the matrix pipe sustain, by MFMA shape and operand data.
// Register-resident operands only (no LDS, no memory): every wave cycles through NA A-fragments x NB B-fragments into NA*NB
// independent accumulators, for ~1.5 s per configuration so that DVFS settles; the rate of the last second is reported.
//   shape: v_mfma_f32_16x16x32_bf16 (the GEMM's) vs v_mfma_f32_32x32x16_bf16 (attention's): same FLOPs per pass, the
//          32x32 form reads half the operand registers per FLOP
//   data:  zeros / random bf16 in [-2, 2) / the same random fragment for every MFMA (accumulators toggle, operands do not)
//   order: of the NA x NB (a_i, b_j) products inside a K-step: i-major (b changes at every MFMA, a every NB-th), serpentine
//          (j runs back and forth: one operand is shared by EVERY pair of consecutive MFMAs), diagonal (both change always)
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_power.hip -o tools/ubench/mfma_power.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <chrono>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) unsigned short u16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

__device__ inline unsigned hash(unsigned x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
__device__ inline bf16x8 frag(unsigned seed, int data) {
  u16x8 u;
  for (int i = 0; i < 8; ++i) {
    const unsigned h = hash(seed * 8u + i);
    // sign | exponent 124..127 | 7 random mantissa bits: |v| in [0.125, 2)
    u[i] = data == 0 ? 0 : (unsigned short)(((h & 1u) << 15) | ((124u + ((h >> 1) & 3u)) << 7) | ((h >> 3) & 0x7fu));
  }
  return __builtin_bit_cast(bf16x8, u);
}

template <int ORDER, int NA, int NB>
__device__ constexpr int pair_i(int q) { return ORDER == 2 ? q % NA : q / NB; }
template <int ORDER, int NA, int NB>
__device__ constexpr int pair_j(int q) {
  return ORDER == 0 ? q % NB : ORDER == 1 ? (((q / NB) & 1) ? NB - 1 - q % NB : q % NB) : (q % NA + q / NA) % NB;
}

template <int SHAPE, int ORDER>   // SHAPE 0: 16x16x32, 1: 32x32x16
__global__ __launch_bounds__(512) void k(float* sink, int iters, int data, int waves_per_simd) {
  const int wave = threadIdx.x >> 6;
  if (wave >= 4 * waves_per_simd) return;
  const unsigned id = blockIdx.x * 512u + threadIdx.x;
  constexpr int NA = 4, NB = 2;
  bf16x8 a[NA], b[NB];
  for (int i = 0; i < NA; ++i) a[i] = frag(data == 2 ? 1u : id * 16u + i, data);
  for (int i = 0; i < NB; ++i) b[i] = frag(data == 2 ? 2u : id * 16u + 8 + i, data);
  float s = 0.f;
  if (SHAPE == 1) {
    f32x16 c[NA * NB] = {};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int q = 0; q < NA * NB; ++q) {
          const int i = pair_i<ORDER, NA, NB>(q), j = pair_j<ORDER, NA, NB>(q);
          c[i * NB + j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], c[i * NB + j], 0, 0, 0);
        }
      if ((it & 63) == 63)      // keep the accumulators bounded (values, not zeros)
#pragma unroll
        for (int q = 0; q < NA * NB; ++q) c[q] *= 0.5f;
    }
    for (int q = 0; q < NA * NB; ++q) s += c[q][q];
  } else {
    f32x4 c[NA * NB * 2] = {};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int r = 0; r < 4; ++r)       // 32 x 16x16x32 = the FLOPs of 16 x 32x32x16
#pragma unroll
        for (int q = 0; q < NA * NB; ++q) {
          const int i = pair_i<ORDER, NA, NB>(q), j = pair_j<ORDER, NA, NB>(q);
          c[(r & 1) * NA * NB + i * NB + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], c[(r & 1) * NA * NB + i * NB + j], 0, 0, 0);
        }
      if ((it & 63) == 63)
#pragma unroll
        for (int q = 0; q < NA * NB * 2; ++q) c[q] *= 0.5f;
    }
    for (int q = 0; q < NA * NB * 2; ++q) s += c[q][q & 3];
  }
  if (s == 12345.678f) sink[0] = s;
}

int main() {
  float* sink;
  hipMalloc(&sink, 64);
  const char* shapes[2] = {"16x16x32", "32x32x16"};
  const char* datas[3] = {"zero operands", "random operands", "one random fragment pair (operands constant)"};
  const char* orders[3] = {"i-major", "serpentine", "diagonal"};
  const int iters = 20000;                       // 16 x 32768 FLOP x iters per wave
  for (int wps = 1; wps <= 2; ++wps)
    for (int data = 0; data < 3; ++data)
      for (int order = 0; order < (data == 1 ? 3 : 1); ++order)
      for (int shape = 0; shape < 2; ++shape) {
        auto launch = [&]() {
          const dim3 g(256), b(512);
          if (shape == 0 && order == 0) hipLaunchKernelGGL((k<0, 0>), g, b, 0, 0, sink, iters, data, wps);
          if (shape == 0 && order == 1) hipLaunchKernelGGL((k<0, 1>), g, b, 0, 0, sink, iters, data, wps);
          if (shape == 0 && order == 2) hipLaunchKernelGGL((k<0, 2>), g, b, 0, 0, sink, iters, data, wps);
          if (shape == 1 && order == 0) hipLaunchKernelGGL((k<1, 0>), g, b, 0, 0, sink, iters, data, wps);
          if (shape == 1 && order == 1) hipLaunchKernelGGL((k<1, 1>), g, b, 0, 0, sink, iters, data, wps);
          if (shape == 1 && order == 2) hipLaunchKernelGGL((k<1, 2>), g, b, 0, 0, sink, iters, data, wps);
        };
        const double flop = 256.0 * 4 * wps * (double)iters * 16.0 * 32768.0;
        double last = 0;
        const auto t_begin = std::chrono::steady_clock::now();
        int n = 0;
        while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count() < 1.5) {
          hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
          hipEventRecord(e0);
          for (int r = 0; r < 4; ++r) launch();
          hipEventRecord(e1); hipEventSynchronize(e1);
          float ms; hipEventElapsedTime(&ms, e0, e1);
          last = 4 * flop / (ms * 1e-3) / 1e12;
          hipEventDestroy(e0); hipEventDestroy(e1);
          ++n;
        }
        printf("%d wave(s)/SIMD  %-9s %-46s %-10s: %7.0f TFLOP/s sustained (after %d x 4 launches)\n", wps, shapes[shape], datas[data],
               data == 1 ? orders[order] : "", last, n);
        fflush(stdout);
      }
  return 0;
}
