// Micro-benchmark: per-CU HBM/L2 -> LDS throughput of global_load_lds_dwordx4 vs plain global_load_dwordx4
// (+ds_write_b128), with the GEMM's access pattern (8 rows x 128 B per wave-instruction, row stride 6144 B).
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench/dma_bw.hip -o /tmp/dma_bw ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

template <int MODE, int PIECES>  // MODE 0: LDS-DMA, 1: global_load -> ds_write_b128, 2: global_load only
__global__ __launch_bounds__(512) void k(const char* __restrict__ src, long region, int iters, unsigned* sink, int share) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // each block streams its own window of `region` bytes, tile after tile (56 pieces of 1 KiB per "K-tile")
  // `share` CUs of one XCD (block b runs on XCD b%8) stream the SAME window -> L2 hits, like GEMM operand panels
  const long sid = (blockIdx.x & 7) + 8 * ((blockIdx.x >> 3) / share);
  const long base = (sid * 7919 * 57344) % (region - (long)PIECES * 8192 * 64);
  unsigned acc = 0;
  for (int it = 0; it < iters; ++it) {
    const char* p = src + base + (long)(it % 48) * 128;   // walk along K like a GEMM
    u32x4 r[PIECES];
#pragma unroll
    for (int i = 0; i < PIECES; ++i) {
      const int c = i * 512 + tid;             // 16-B chunk id within the tile
      const long off = (long)(c >> 3) * 6144 + (c & 7) * 16;
      if (MODE == 0)
        __builtin_amdgcn_global_load_lds((gptr_t)(p + off), (lptr_t)(smem + (it & 1) * PIECES * 8192 + (i * 512 + wave * 64) * 16), 16, 0, 0);
      else
        r[i] = *(const u32x4*)(p + off);
    }
    if (MODE == 1) {
#pragma unroll
      for (int i = 0; i < PIECES; ++i) *(u32x4*)(smem + (it & 1) * PIECES * 8192 + (i * 512 + tid) * 16) = r[i];
    } else if (MODE == 2) {
#pragma unroll
      for (int i = 0; i < PIECES; ++i) acc += r[i][0];
    }
    __syncthreads();
    if (MODE != 2) acc += *(unsigned*)(smem + tid * 4);
  }
  if (acc == 0x12345) sink[0] = acc;
}

int main() {
  const long region = 512L << 20;
  char* d; unsigned* sink;
  hipMalloc(&d, region); hipMalloc(&sink, 64);
  hipMemset(d, 1, region);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 2000;
  for (int share = 1; share <= 32; share *= (share == 1 ? 4 : 2))
  for (int mode = 0; mode < 3; ++mode) {
    for (int rep = 0; rep < 1; ++rep) {
      hipEventRecord(e0);
      constexpr int P = 7;
      const int lds = 2 * P * 8192;
      if (mode == 0) { hipFuncSetAttribute((const void*)k<0, P>, hipFuncAttributeMaxDynamicSharedMemorySize, lds); hipLaunchKernelGGL((k<0, P>), dim3(256), dim3(512), lds, 0, d, region, iters, sink, share); }
      if (mode == 1) { hipFuncSetAttribute((const void*)k<1, P>, hipFuncAttributeMaxDynamicSharedMemorySize, lds); hipLaunchKernelGGL((k<1, P>), dim3(256), dim3(512), lds, 0, d, region, iters, sink, share); }
      if (mode == 2) { hipFuncSetAttribute((const void*)k<2, P>, hipFuncAttributeMaxDynamicSharedMemorySize, lds); hipLaunchKernelGGL((k<2, P>), dim3(256), dim3(512), lds, 0, d, region, iters, sink, share); }
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      const double bytes = 256.0 * iters * P * 8192.0;
      printf("share %2d mode %d rep %d: %.3f ms  %.2f TB/s  %.1f GB/s per CU (%s)\n", share, mode, rep, ms, bytes / ms / 1e9, bytes / 256 / ms / 1e6,
             mode == 0 ? "global_load_lds" : mode == 1 ? "global_load + ds_write_b128" : "global_load only");
    }
  }
  printf("err=%s\n", hipGetErrorString(hipGetLastError()));
  return 0;
}
