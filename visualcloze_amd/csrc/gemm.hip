// bf16 MFMA GEMM for the FLUX linear layers on gfx950:  C[M,N] = epi( A[M,K] . W[N,K]^T + bias )
//
// Replaces every nn.Linear on the hot path (reference: models/modules/layers.py:55-57,93-95,118,
// 142-155,220-222,252-253; LoRA-merged weights, models/modules/lora.py:92-98) with fused epilogues:
//   EPI_BIAS      y = bf16(acc + b)                                   (qkv, linear1-qkv, img_in, ...)
//   EPI_GELU      y = bf16(gelu_tanh(bf16(acc + b)))                  (mlp.0 + nn.GELU("tanh"))
//   EPI_GATE_RES  y = bf16(res + bf16(gate * bf16(acc + b)))          (x + gate * proj(...), layers.py:190-195,245)
//   EPI_SILU      y = bf16(silu(bf16(acc + b)))                       (MLPEmbedder in_layer + SiLU)
// The bf16() rounding points are the ones the reference materialises under torch.autocast(bf16).
//
// Structure: BMxBNx64 block tile, WMxWN waves, v_mfma_f32_16x16x32_bf16, both operands K-contiguous.
// HBM->LDS by global_load_lds (16 B/lane, LDS image lane-linear) with the 16-B-slot XOR swizzle applied
// on the SOURCE address and again on the ds_read_b128 address (conflict-free for 128-B rows).
// Double-buffered LDS, one barrier per K-tile. Operands are swapped in the MFMA (D = W_frag x A_frag)
// so each lane ends up with 4 consecutive n of one row m -> 8-byte epilogue stores.
// Up to two problems per launch ("grouped": img + txt streams of a DoubleStreamBlock share a grid).
#include "common.h"
#include "vcloze_internal.h"

namespace {

constexpr int BK = 64;

template <int BM, int BN, int WM, int WN, int EPI>
__global__ __launch_bounds__(WM* WN * 64) void gemm_bf16_kernel(const VcGemmArgs args) {
  constexpr int NT = WM * WN * 64;
  constexpr int TM = BM / WM, TN = BN / WN;
  constexpr int MI = TM / 16, NI = TN / 16;
  constexpr int A_BYTES = BM * BK * 2, B_BYTES = BN * BK * 2;
  constexpr int STAGE_BYTES = A_BYTES + B_BYTES;
  constexpr int A_IT = BM * 8 / NT, B_IT = BN * 8 / NT;
  static_assert(BM * 8 % NT == 0 && BN * 8 % NT == 0, "staging split");
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;

  // ---- which tile ----
  int id = xcd_remap(blockIdx.x, gridDim.x);
  const int pi = (args.nprob > 1 && id >= args.p[1].tile_start) ? 1 : 0;
  const VcGemmProblem P = pi ? args.p[1] : args.p[0];
  id -= P.tile_start;
  constexpr int GROUP_M = 8;
  const int in_group = GROUP_M * P.tiles_n;
  const int group = id / in_group;
  const int first_m = group * GROUP_M;
  const int gsz = min(P.tiles_m - first_m, GROUP_M);
  const int tm = first_m + (id % in_group) % gsz;
  const int tn = (id % in_group) / gsz;
  const int m0 = tm * BM, n0 = tn * BN;
  const int M = P.M, N = P.N, K = P.K;

  const bf16_t* __restrict__ Ab = (const bf16_t*)P.A;
  const bf16_t* __restrict__ Wb = (const bf16_t*)P.W;

  // ---- staging source offsets (elements), one per 16-B chunk this thread copies ----
  uint32_t a_off[A_IT], b_off[B_IT];
#pragma unroll
  for (int i = 0; i < A_IT; ++i) {
    const int c = i * NT + tid;
    const int row = c >> 3, slot = (c & 7) ^ (row & 7);
    const int grow = min(m0 + row, M - 1);
    a_off[i] = (uint32_t)grow * (uint32_t)P.lda + slot * 8;
  }
#pragma unroll
  for (int i = 0; i < B_IT; ++i) {
    const int c = i * NT + tid;
    const int row = c >> 3, slot = (c & 7) ^ (row & 7);
    const int grow = min(n0 + row, N - 1);
    b_off[i] = (uint32_t)grow * (uint32_t)K + slot * 8;
  }

  auto stage = [&](int buf, int k0) {
    char* sa = smem + buf * STAGE_BYTES;
    char* sb = sa + A_BYTES;
#pragma unroll
    for (int i = 0; i < A_IT; ++i) glds16(Ab + a_off[i] + k0, sa + (i * NT + wave * 64) * 16);
#pragma unroll
    for (int i = 0; i < B_IT; ++i) glds16(Wb + b_off[i] + k0, sb + (i * NT + wave * 64) * 16);
  };

  // ---- fragment read offsets ----
  const int fr = lane & 15, fq = lane >> 4;
  const int sw0 = ((fq ^ (lane & 7)) << 4);  // kk=0 slot; kk=1 is sw0 ^ 64
  const int a_rd = (wm * TM + fr) * 128 + sw0;
  const int b_rd = A_BYTES + (wn * TN + fr) * 128 + sw0;

  f32x4 acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nk = K / BK;
  stage(0, 0);
  __syncthreads();

  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < nk) stage(cur ^ 1, (kt + 1) * BK);
    const char* base = smem + cur * STAGE_BYTES;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      bf16x8 af[MI], bfr[NI];
#pragma unroll
      for (int i = 0; i < MI; ++i) af[i] = *(const bf16x8*)(base + ((a_rd + i * 16 * 128) ^ (kk * 64)));
#pragma unroll
      for (int j = 0; j < NI; ++j) bfr[j] = *(const bf16x8*)(base + ((b_rd + j * 16 * 128) ^ (kk * 64)));
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
    }
    __syncthreads();
  }

  // ---- epilogue: lane holds C[m = ..+fr][n = ..+fq*4 .. +3] ----
  const bf16_t* __restrict__ bias = (const bf16_t*)P.bias;
  bf16_t* __restrict__ C = (bf16_t*)P.C;
  const bf16_t* __restrict__ res = (const bf16_t*)P.res;
  const bf16_t* __restrict__ gate = (const bf16_t*)P.gate;
  long gate_step = 0;
  if (EPI == VC_EPI_GATE_RES && args.step_ptr) gate_step = (long)(*args.step_ptr) * args.gate_step_stride;

#pragma unroll
  for (int i = 0; i < MI; ++i) {
    const int m = m0 + wm * TM + i * 16 + fr;
    if (m >= M) continue;
    const bf16_t* grow_ptr = nullptr;
    if (EPI == VC_EPI_GATE_RES) grow_ptr = gate + gate_step + (long)(m / P.rows_per_batch) * P.gate_bstride;
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      const int n = n0 + wn * TN + j * 16 + fq * 4;
      if (n >= N) continue;
      float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
      if (bias) {
        const u32x2 bb = *(const u32x2*)(bias + n);
        v[0] += lo_bf(bb[0]); v[1] += hi_bf(bb[0]); v[2] += lo_bf(bb[1]); v[3] += hi_bf(bb[1]);
      }
      if (EPI == VC_EPI_GELU) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = gelu_tanh(rbf(v[e]));
      } else if (EPI == VC_EPI_SILU) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = silu_f(rbf(v[e]));
      } else if (EPI == VC_EPI_GATE_RES) {
        const u32x2 gg = *(const u32x2*)(grow_ptr + n);
        const u32x2 rr = *(const u32x2*)(res + (long)m * P.ldres + n);
        const float g[4] = {lo_bf(gg[0]), hi_bf(gg[0]), lo_bf(gg[1]), hi_bf(gg[1])};
        const float r[4] = {lo_bf(rr[0]), hi_bf(rr[0]), lo_bf(rr[1]), hi_bf(rr[1])};
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = r[e] + rbf(g[e] * rbf(v[e]));
      }
      u32x2 o;
      o[0] = pack2bf(v[0], v[1]);
      o[1] = pack2bf(v[2], v[3]);
      *(u32x2*)(C + (long)m * P.ldc + n) = o;
    }
  }
}

template <int BM, int BN, int WM, int WN>
hipError_t launch_cfg(const VcGemmArgs& a, int total_tiles, hipStream_t s) {
  constexpr int NT = WM * WN * 64;
  constexpr int LDS = 2 * (BM + BN) * BK * 2;
  void (*fn)(const VcGemmArgs) = nullptr;
  switch (a.epi) {
    case VC_EPI_BIAS: fn = gemm_bf16_kernel<BM, BN, WM, WN, VC_EPI_BIAS>; break;
    case VC_EPI_GELU: fn = gemm_bf16_kernel<BM, BN, WM, WN, VC_EPI_GELU>; break;
    case VC_EPI_GATE_RES: fn = gemm_bf16_kernel<BM, BN, WM, WN, VC_EPI_GATE_RES>; break;
    case VC_EPI_SILU: fn = gemm_bf16_kernel<BM, BN, WM, WN, VC_EPI_SILU>; break;
    default: return hipErrorInvalidValue;
  }
  static bool attr_done[4] = {false, false, false, false};
  if (!attr_done[a.epi]) {
    hipError_t e = hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    if (e != hipSuccess) return e;
    attr_done[a.epi] = true;
  }
  hipLaunchKernelGGL(fn, dim3(total_tiles), dim3(NT), LDS, s, a);
  return hipGetLastError();
}

}  // namespace

// tile_cfg: 0 = auto, 1 = 128x128 (4 waves), 2 = 256x128 (8 waves), 3 = 256x256 (8 waves)
int vc_gemm_launch(VcGemmArgs a, int tile_cfg, hipStream_t s, char* err, int errlen) {
  if (a.nprob < 1 || a.nprob > 2) { snprintf(err, errlen, "gemm: nprob must be 1 or 2"); return VC_ERR_ARG; }
  for (int i = 0; i < a.nprob; ++i) {
    const VcGemmProblem& p = a.p[i];
    if (p.M <= 0 || p.N <= 0 || p.K <= 0) { snprintf(err, errlen, "gemm: empty problem %d (M=%d N=%d K=%d)", i, p.M, p.N, p.K); return VC_ERR_ARG; }
    if (p.K % BK) { snprintf(err, errlen, "gemm: K=%d must be a multiple of %d", p.K, BK); return VC_ERR_ARG; }
    if (p.N % 4 || p.ldc % 4 || p.lda % 8) { snprintf(err, errlen, "gemm: need N%%4==0, ldc%%4==0, lda%%8==0 (N=%d ldc=%ld lda=%ld)", p.N, (long)p.ldc, (long)p.lda); return VC_ERR_ARG; }
    if (!p.A || !p.W || !p.C) { snprintf(err, errlen, "gemm: null operand"); return VC_ERR_ARG; }
    if ((uint64_t)p.M * (uint64_t)p.lda >= (1ull << 32) || (uint64_t)p.N * (uint64_t)p.K >= (1ull << 32)) {
      snprintf(err, errlen, "gemm: operand exceeds 32-bit element offsets"); return VC_ERR_ARG; }
    if (a.epi == VC_EPI_GATE_RES && (!p.res || !p.gate || p.rows_per_batch <= 0 || p.ldres % 4)) {
      snprintf(err, errlen, "gemm: gate/residual epilogue needs res, gate, rows_per_batch"); return VC_ERR_ARG; }
  }
  if (tile_cfg == 0) {
    // pick the tile that keeps >= ~2 block-waves of work on 256 CUs
    long t256 = 0, t2128 = 0;
    for (int i = 0; i < a.nprob; ++i) {
      t256 += (long)((a.p[i].M + 255) / 256) * ((a.p[i].N + 255) / 256);
      t2128 += (long)((a.p[i].M + 255) / 256) * ((a.p[i].N + 127) / 128);
    }
    tile_cfg = 1;
    (void)t256; (void)t2128;
  }
  int bm, bn;
  switch (tile_cfg) {
    case 1: bm = 128; bn = 128; break;
    case 2: bm = 256; bn = 128; break;
    case 3: bm = 256; bn = 256; break;
    default: snprintf(err, errlen, "gemm: bad tile_cfg %d", tile_cfg); return VC_ERR_ARG;
  }
  int total = 0;
  for (int i = 0; i < a.nprob; ++i) {
    a.p[i].tiles_m = (a.p[i].M + bm - 1) / bm;
    a.p[i].tiles_n = (a.p[i].N + bn - 1) / bn;
    a.p[i].tile_start = total;
    total += a.p[i].tiles_m * a.p[i].tiles_n;
  }
  hipError_t e;
  switch (tile_cfg) {
    case 1: e = launch_cfg<128, 128, 2, 2>(a, total, s); break;
    case 2: e = launch_cfg<256, 128, 4, 2>(a, total, s); break;
    default: e = launch_cfg<256, 256, 2, 4>(a, total, s); break;
  }
  if (e != hipSuccess) { snprintf(err, errlen, "gemm launch: %s", hipGetErrorString(e)); return VC_ERR_HIP; }
  return VC_OK;
}
