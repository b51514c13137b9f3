// Micro-benchmark: sustained per-CU L2 -> LDS rate of global_load_lds_dwordx4 as a function of the number of
// tiles kept in flight (ring depth NS) and of the row-piece length (ROWB bytes contiguous per row).
// Unlike dma_bw.hip there is no full drain per tile: iteration i waits only for tile i-(NS-1).
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench/dma_depth.hip -o tools/ubench/dma_depth.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

template <int NS, int PIECES, int ROWB, int NT>
__global__ __launch_bounds__(NT) void k(const char* __restrict__ src, long region, int iters, unsigned* sink, int share) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  constexpr int STAGE = PIECES * NT * 16;
  constexpr int CPR = ROWB / 16;                       // 16-B chunks per row piece
  const long sid = (blockIdx.x & 7) + 8 * ((blockIdx.x >> 3) / share);
  const long base = (sid * 7919 * 57344) % (region - (16L << 20));
  unsigned acc = 0;
  auto issue = [&](int it) {
    const char* p = src + base + (long)(it % 48) * ROWB;
#pragma unroll
    for (int i = 0; i < PIECES; ++i) {
      const int c = i * NT + tid;
      const long off = (long)(c / CPR) * 6144 + (c % CPR) * 16;
      __builtin_amdgcn_global_load_lds((gptr_t)(p + off), (lptr_t)(smem + (it % NS) * STAGE + (i * NT + wave * 64) * 16), 16, 0, 0);
    }
  };
#pragma unroll
  for (int s = 0; s < NS - 1; ++s) issue(s);
  for (int it = NS - 1; it < iters; ++it) {
    issue(it);
    if (NS == 1) __builtin_amdgcn_s_waitcnt(0x0070 | 0x3F00);            // vmcnt(0)
    else __asm__ volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 1) * PIECES));
    __builtin_amdgcn_s_barrier();
    acc += *(unsigned*)(smem + ((it + 1) % NS) * STAGE + tid * 4);
    __builtin_amdgcn_s_barrier();
  }
  if (acc == 0x12345) sink[0] = acc;
}

template <int NS, int PIECES, int ROWB, int NT>
void run(const char* d, long region, unsigned* sink, int share) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 3000;
  const int lds = NS * PIECES * NT * 16;
  hipFuncSetAttribute((const void*)k<NS, PIECES, ROWB, NT>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<NS, PIECES, ROWB, NT>), dim3(256), dim3(NT), lds, 0, d, region, iters, sink, share);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double bytes = 256.0 * iters * PIECES * NT * 16.0;
    if (rep == 1)
      printf("share %2d NS %d stage %3d KB rowB %3d NT %d: %.3f ms  %.2f TB/s  %.1f GB/s per CU\n", share, NS, PIECES * NT * 16 / 1024, ROWB, NT, ms,
             bytes / ms / 1e9, bytes / 256 / ms / 1e6);
  }
}

int main() {
  const long region = 512L << 20;
  char* d; unsigned* sink;
  hipMalloc(&d, region); hipMalloc(&sink, 64);
  hipMemset(d, 1, region);
  for (int share : {1, 8, 32}) {
    run<1, 4, 128, 512>(d, region, sink, share);
    run<2, 4, 128, 512>(d, region, sink, share);
    run<3, 4, 128, 512>(d, region, sink, share);
    run<4, 4, 128, 512>(d, region, sink, share);
    run<2, 8, 128, 512>(d, region, sink, share);
    run<2, 8, 256, 512>(d, region, sink, share);
    run<4, 4, 256, 512>(d, region, sink, share);
    run<4, 4, 64, 512>(d, region, sink, share);
    run<2, 16, 128, 256>(d, region, sink, share);
    run<4, 8, 128, 256>(d, region, sink, share);
  }
  printf("err=%s\n", hipGetErrorString(hipGetLastError()));
  return 0;
}
