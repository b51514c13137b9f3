// bf16 MFMA GEMM for the FLUX linear layers on gfx950:  C[M,N] = epi( A[M,K] . W[N,K]^T + bias )
//
// Replaces every nn.Linear on the hot path (reference: models/modules/layers.py:55-57,93-95,118,
// 142-155,220-222,252-253; LoRA-merged weights, models/modules/lora.py:92-98) with fused epilogues:
//   EPI_BIAS      y = bf16(acc + b)                                   (qkv, linear1-qkv, img_in, ...)
//   EPI_GELU      y = bf16(gelu_tanh(bf16(acc + b)))                  (mlp.0 + nn.GELU("tanh"))
//   EPI_GATE_RES  y = bf16(res + bf16(gate * bf16(acc + b)))          (x + gate * proj(...), layers.py:190-195,245)
//   EPI_SILU      y = bf16(silu(bf16(acc + b)))                       (MLPEmbedder in_layer + SiLU)
//   EPI_QKV       EPI_BIAS, with the V column range written transposed to vt[b][h][d][l] (attention's B operand)
// The bf16() rounding points are the ones the reference materialises under torch.autocast(bf16).
//
// Structure: BMxBNx64 block tile, WMxWN waves, v_mfma_f32_16x16x32_bf16, both operands K-contiguous.
// HBM->LDS by global_load_lds (16 B/lane, LDS image lane-linear) with the 16-B-slot XOR swizzle applied
// on the SOURCE address and again on the ds_read_b128 address (conflict-free for 128-B rows).
// Double-buffered LDS, one barrier per K-tile. Operands are swapped in the MFMA (D = W_frag x A_frag)
// so each lane ends up with 4 consecutive n of one row m -> 8-byte epilogue stores.
// Up to two problems per launch ("grouped": img + txt streams of a DoubleStreamBlock share a grid).
#include <string.h>
#include <type_traits>
#include "common.h"
#include "vcloze_internal.h"

namespace {

constexpr int BK = 64;
// raster: ids walk GROUP_M m-tiles, then the n-tiles (an XCD's 32 resident workgroups = GROUP_M x 32/GROUP_M tiles)
#ifndef VC_GEMM_GROUP_M
#define VC_GEMM_GROUP_M 8
#endif
#ifndef VC_GEMM_STREAMK_MIN_K
#define VC_GEMM_STREAMK_MIN_K 6144      /* the stream remainder is not offered below this K (the partial traffic outweighs the short tiles) */
#endif

// CONV (loader-wave schedule only): the A operand is the im2col matrix of a 3x3 convolution over an NHWC map, gathered
// on the fly by the loader waves - K-tile kt lies inside tap kt*64 / C, row m is output pixel (m / W, m % W), out-of-
// image taps read a zero row appended to the map.  Geometry rides in the A-addressing fields (vc_conv3x3_launch).
//
// PERSIST (loader-wave schedule only): the grid is one workgroup per CU and every workgroup walks the tiles of its XCD's
// strip (tile index = slot, slot + W, ... with W = grid / 8 workgroups per XCD: the order the hardware dispatcher gives the
// one-block-per-tile launch).  What it buys is the seam between two tiles: while the 12 waves run the epilogue of tile i
// the loader waves already have W(0), W(1) of tile i+1 in flight - the weight operand, streamed from HBM, the long-latency
// one - into two ring slots that live ABOVE the epilogue's LDS staging area, and A(0) follows as soon as the staging area
// has been read back; a new workgroup would pay dispatch + offsets + the full HBM latency of its first tiles instead
// (6.0 k cycles of prologue + 1.2-3.8 k of dispatch gap per 90 k-cycle round at K = 3072).
template <int BM, int BN, int WM, int WN, int EPI, int PP, bool CONV = false, bool PERSIST = false, bool SPLITK = false, bool ZB = false>
__global__ __launch_bounds__((WM * WN + (PP == 2 ? 4 : 0)) * 64) void gemm_bf16_kernel(const VcGemmArgs args) {
  static_assert(!SPLITK || (PP == 2 && !CONV && !PERSIST), "split-K slices run on the plain loader-wave schedule");
  static_assert(!CONV || PP == 2, "the implicit-convolution A operand is gathered by loader waves");
  static_assert(!PERSIST || (PP == 2 && !CONV), "the persistent tile loop exists for the loader-wave schedule");
  constexpr int NCW = WM * WN;                       // compute waves
  constexpr int NT = (NCW + (PP == 2 ? 4 : 0)) * 64;  // PP == 2 adds 4 loader waves (one per SIMD)
  constexpr int NS = PP == 2 ? 256 : NT;              // threads that stage operand tiles
  constexpr int TM = BM / WM, TN = BN / WN;
  constexpr int MI = TM / 16, NI = TN / 16;
  constexpr int A_BYTES = BM * BK * 2, B_BYTES = BN * BK * 2;
  constexpr int STAGE_BYTES = A_BYTES + B_BYTES;
  constexpr int A_CH = BM * 8, B_CH = BN * 8;                       // 16-B chunks per operand tile
  constexpr int A_IT = (A_CH + NS - 1) / NS, B_IT = (B_CH + NS - 1) / NS;
  static_assert(A_CH % 64 == 0 && B_CH % 64 == 0, "a wave-instruction (64 chunks) must not straddle the tile end");
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
#ifdef VC_GEMM_TIMESTAMPS   // profiling builds only (tools/gemm_phases.py): entry / loop start / loop end / pass 1 / exit per block
  uint64_t* pts = (args.debug_ts && tid == 0) ? args.debug_ts + 8192 + (size_t)blockIdx.x * 8 : nullptr;
  if (pts) {
    pts[0] = __builtin_amdgcn_s_memtime();
    pts[5] = ((uint64_t)__builtin_amdgcn_s_getreg(20 | (31 << 11)) << 32) | (uint32_t)__builtin_amdgcn_s_getreg(4 | (31 << 11));   // XCC_ID, HW_ID
  }
#define VC_PHASE_STAMP(i) do { if (pts) pts[i] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define VC_PHASE_STAMP(i) do {} while (0)
#endif

  // ---- which tile(s) ----
  // one block per tile: logical id = xcd_remap(blockIdx) (XCD x works on one contiguous strip of ids).  PERSIST: the same
  // strips, walked by the W = grid / 8 resident workgroups of the XCD: ids strip0 + slot, strip0 + slot + W, ...
  int id_cur = xcd_remap(blockIdx.x, gridDim.x), id_end = 0, id_step = 0;
  // SPLITK instantiation (its own launch, behind the whole tiles of the call): block p is WORK ITEM p of the remainder - K-slice
  // p / sk_rem of tile sk_full + p % sk_rem (the items of one slice are neighbours: those that share an m-tile share their A
  // slab in one XCD's L2).  An item leaves its f32 accumulators in args.splitk_ws [tile][slice][BM][BN] instead of running an
  // epilogue; splitk_reduce_kernel finishes the tile.  Every other instantiation compiles exactly as without this.
  // STREAM form (args.sk_stream = n work items, for remainders of MORE than half a round, where no uniform S >= 2 fits): the
  // K-iterations of the sk_rem tiles, flattened tile-major, are dealt out evenly - item p owns iterations [p I / n, (p + 1) I / n)
  // of I = sk_rem * (K / 64), at most two segments (the end of one tile, the start of the next; n >= sk_rem), each a pass of this
  // loop with its own prologue, accumulators and partial tile (slot 2 p + segment).  Static assignment: bit-reproducible.
  int sk_slice = 0, sk_slot = -1;
  int sk_it = 0, sk_it_end = 0, sk_seg = 0;          // stream form: next iteration / end of this item's range, segment index
  bool sk_bias = true;                               // this pass starts its tile's K range: its accumulators start at the bias
  if constexpr (SPLITK) {
    if (args.sk_stream > 0) {
      const uint32_t I = (uint32_t)args.sk_rem * (uint32_t)(args.p[0].K / BK), n = (uint32_t)args.sk_stream;
      sk_it = (int)((uint64_t)blockIdx.x * I / n);
      sk_it_end = (int)((uint64_t)(blockIdx.x + 1) * I / n);
      if (sk_it >= sk_it_end) return;                // (more items than iterations: nothing to do, before any barrier)
    } else {
      sk_slice = blockIdx.x / args.sk_rem;
      const int r = blockIdx.x - sk_slice * args.sk_rem;
      id_cur = args.sk_full + r;
      sk_slot = r * args.sk_S + sk_slice;
      sk_bias = sk_slice == 0;
    }
  }
  if constexpr (PERSIST) {
    const int total = args.p[args.nprob - 1].tile_start + args.p[args.nprob - 1].tiles_m * args.p[args.nprob - 1].tiles_n;
    const int x = blockIdx.x & 7, q = total >> 3, r = total & 7;
    const int strip0 = x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q;
    id_cur = strip0 + (blockIdx.x >> 3);
    id_end = strip0 + q + (x < r ? 1 : 0);
    id_step = gridDim.x >> 3;
    if (id_cur >= id_end) return;
  }
  struct Tile { VcGemmProblem P; int m0, n0; };
  auto decode = [&](int id) {
    int pi = 0;
#pragma unroll
    for (int q = 1; q < VC_GEMM_MAX_PROBLEMS; ++q)
      if (q < args.nprob && id >= args.p[q].tile_start) pi = q;
    Tile t;
    t.P = pi == 3 ? args.p[3] : pi == 2 ? args.p[2] : pi == 1 ? args.p[1] : args.p[0];
    id -= t.P.tile_start;
    constexpr int GROUP_M = VC_GEMM_GROUP_M;
    const int in_group = GROUP_M * t.P.tiles_n;
    const int group = id / in_group;
    const int first_m = group * GROUP_M;
    const int gsz = min(t.P.tiles_m - first_m, GROUP_M);
    const int tm = first_m + (id % in_group) % gsz;
    const int tn = (id % in_group) / gsz;
    t.m0 = t.P.m_begin + tm * BM;     // m_begin: rows below it belong to the other launch of a split call
    t.n0 = tn * BN;
    if constexpr (ZB) {               // VcGemmArgs.batch: instance blockIdx.y of Z independent GEMMs of this shape
      t.P.A = (const bf16_t*)t.P.A + (long)blockIdx.y * t.P.a_zstride;
      t.P.W = (const bf16_t*)t.P.W + (long)blockIdx.y * t.P.w_zstride;
      t.P.C = (bf16_t*)t.P.C + (long)blockIdx.y * t.P.c_zstride;
    }
    return t;
  };
  // ---- staging source offsets (elements), one per 16-B chunk this thread copies ----
  const int stid = PP == 2 ? (tid - NCW * 64) & 255 : tid;   // staging thread / wave index (PP == 2: loader waves)
  const int swave = PP == 2 ? (wave - NCW) & 3 : wave;
  // loader-wave kernels run these inside the loader branch (and, PERSIST, at the seam between two tiles) only, each time into
  // registers that die with their last piece: nothing of them is live in a compute wave's K loop
  // BYTE_OFF (the loader-wave kernels): the offsets are 32-bit BYTE offsets against a scalar base - A against the operand, W
  // against the first row of the block's n-tile (the stacked modulation matrix alone is 6.5 GB) - so that every LDS-DMA piece is
  // `global_load_lds v_off, s[base]` with the K position folded into the scalar base: no vector instruction per piece in the
  // loader waves, which issue on the SIMDs the compute waves feed the matrix pipe from (element offsets against a bf16 pointer
  // made hipcc form a 64-bit address per piece: one v_lshl_add_u64 for each of the 14 pieces of a K-tile).
#ifndef VC_GEMM_NO_SADDR      // (A/B builds: round 3's per-piece 64-bit vector addresses)
  constexpr bool BYTE_OFF = PP == 2 && !CONV;
#else
  constexpr bool BYTE_OFF = false;
#endif
  auto staging_offsets_a = [&](uint32_t (&ao)[A_IT], const VcGemmProblem& Q, int m0q, int st) {
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
      const int c = i * NS + st;
      const int row = c >> 3, slot = (c & 7) ^ (row & 7);
      const int grow = min(m0q + row, Q.M - 1);
      ao[i] = (Q.a_rpb > 0 ? (uint32_t)(grow / Q.a_rpb) * (uint32_t)Q.a_bstride + (uint32_t)(grow % Q.a_rpb) * (uint32_t)Q.lda
                           : (uint32_t)grow * (uint32_t)Q.lda) + slot * 8;
      if constexpr (BYTE_OFF) ao[i] *= 2u;              // (validate_gemm: the A operand of a loader-wave launch stays below 4 GB)
    }
  };
  auto staging_offsets_b = [&](uint32_t (&bo)[B_IT], const VcGemmProblem& Q, int n0q, int st) {
#pragma unroll
    for (int i = 0; i < B_IT; ++i) {
      const int c = i * NS + st;
      const int row = c >> 3, slot = (c & 7) ^ (row & 7);
      const int grow = min(n0q + row, Q.N - 1);
      if constexpr (BYTE_OFF) bo[i] = ((uint32_t)(grow - n0q) * (uint32_t)Q.ldw + slot * 8) * 2u;     // relative to row n0q: < BN * ldw * 2
      else bo[i] = (uint32_t)grow * (uint32_t)Q.ldw + slot * 8;
    }
  };

  for (bool first_tile = true;; first_tile = false) {      // (one pass unless PERSIST, or a stream item with two segments)
  int sk_k0 = 0, sk_k1 = 0;
  if constexpr (SPLITK) {
    if (args.sk_stream > 0) {
      const int nk0 = args.p[0].K / BK, t = sk_it / nk0;
      sk_k0 = sk_it - t * nk0;
      sk_k1 = min(nk0, sk_it_end - t * nk0);
      id_cur = args.sk_full + t;
      sk_slot = blockIdx.x * 2 + sk_seg;
      sk_bias = sk_k0 == 0;
    }
  }
  const Tile tile_cur = decode(id_cur);
  const VcGemmProblem& P = tile_cur.P;
  const int m0 = tile_cur.m0, n0 = tile_cur.n0;
  const int M = P.M, N = P.N, K = P.K;
  const bool has_next = PERSIST && id_cur + id_step < id_end;

  // K-tiles [kt_lo, kt_lo + nk) of this work item (the whole K unless it is a split-K slice)
  int kt_lo = 0, nk = K / BK;
  if constexpr (SPLITK) {
    if (args.sk_stream > 0) {
      kt_lo = sk_k0;
      nk = sk_k1 - sk_k0;
    } else {
      kt_lo = (int)((long)nk * sk_slice / args.sk_S);
      nk = (int)((long)nk * (sk_slice + 1) / args.sk_S) - kt_lo;
    }
  }
  const bf16_t* __restrict__ Ab = (const bf16_t*)P.A + (long)kt_lo * BK;
  const bf16_t* __restrict__ Wb = (const bf16_t*)P.W + (long)kt_lo * BK;
  const char* __restrict__ Abytes = (const char*)Ab;                                   // BYTE_OFF bases (scalar)
  const char* __restrict__ Wbytes = (const char*)(Wb + (long)n0 * P.ldw);
  uint32_t a_off[A_IT], b_off[B_IT];
  auto staging_offsets = [&]() { staging_offsets_a(a_off, P, m0, stid); staging_offsets_b(b_off, P, n0, stid); };
  if constexpr (PP != 2) staging_offsets();
  // implicit 3x3 convolution: (y, x) of the output pixel each A piece of this thread belongs to, packed y << 16 | x
  const int cvC = (int)P.lda, cvW = P.a_rpb, cvH = (int)(P.a_bstride & 0xffffffff), cvMode = (int)(P.a_bstride >> 32);
  uint32_t cv_yx[CONV ? A_IT : 1];
  int cv_dy = 0, cv_dx = 0, cv_c0 = 0;               // tap and channel offset of the K-tile being staged
  if constexpr (CONV) {
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
      const int grow = min(m0 + ((i * NS + stid) >> 3), M - 1);
      cv_yx[i] = ((uint32_t)(grow / cvW) << 16) | (uint32_t)(grow % cvW);
    }
  }
  auto conv_set_k = [&](int k0) {                     // k0 is wave-uniform
    const int tap = k0 / cvC;
    cv_c0 = k0 - tap * cvC;
    cv_dy = tap / 3;
    cv_dx = tap - cv_dy * 3;
  };
  auto conv_src = [&](int i) -> const bf16_t* {
    const int y = (int)(cv_yx[CONV ? i : 0] >> 16), x = (int)(cv_yx[CONV ? i : 0] & 0xffff);
    long src;
    if (cvMode == 2) {                                // pad (0,1,0,1) + stride 2 over a [2H, 2W] map
      const int yy = 2 * y + cv_dy, xx = 2 * x + cv_dx;
      src = (yy < 2 * cvH && xx < 2 * cvW) ? (long)yy * (2 * cvW) + xx : 4L * cvH * cvW;
    } else {
      const int yy = y + cv_dy - 1, xx = x + cv_dx - 1;
      const bool ok = yy >= 0 && yy < cvH && xx >= 0 && xx < cvW;
      if (cvMode == 1) src = ok ? (long)(yy >> 1) * (cvW >> 1) + (xx >> 1) : (long)(cvH >> 1) * (cvW >> 1);   // nearest 2x
      else src = ok ? (long)yy * cvW + xx : (long)cvH * cvW;
    }
    return Ab + src * cvC + cv_c0 + (((stid & 7) ^ ((stid >> 3) & 7)) << 3);
  };

  auto stage = [&](int buf, int k0) {
#ifdef VC_GEMM_NO_DMA     // analysis builds only: core side of the loop alone (operands of K-tile 0 reused)
    if (k0 > 0) return;
#endif
    char* sa = smem + buf * STAGE_BYTES;
    char* sb = sa + A_BYTES;
#pragma unroll
    for (int i = 0; i < A_IT; ++i)
      if (A_CH % NS == 0 || i * NS + swave * 64 < A_CH) glds16(Ab + a_off[i] + k0, sa + (i * NS + swave * 64) * 16);
#pragma unroll
    for (int i = 0; i < B_IT; ++i)   // 256x288: the last sweep covers half a tile's worth -> waves 0-3 only (wave-uniform)
      if (B_CH % NS == 0 || i * NS + swave * 64 < B_CH) glds16(Wb + b_off[i] + k0, sb + (i * NS + swave * 64) * 16);
  };
  // PP == 2 (loader waves): A lives in a 2-deep ring, W in a 3-deep ring (2*A_BYTES + 3*B_BYTES = 136 KB for 256x192),
  // PERSIST: slot 2 stays behind the A ring, slots 0 and 1 move to the top of the 160 KB so that the epilogue's staging
  // area (which starts at 0 and covers the A ring and slot 2) never touches them: W(0), W(1) of the NEXT tile land there
  // while the current tile's epilogue runs.
  constexpr int W_RING0 = 2 * A_BYTES;
  constexpr int W_HI = 160 * 1024 - 2 * B_BYTES;
  auto wslot = [&](int slot) { return PERSIST ? (slot == 2 ? W_RING0 : W_HI + slot * B_BYTES) : W_RING0 + slot * B_BYTES; };
  auto stage_a_piece = [&](int slot, int k0, int i) {
    if constexpr (CONV) glds16(conv_src(i), smem + slot * A_BYTES + (i * NS + swave * 64) * 16);   // (conv_set_k(k0) done by the caller)
    else if constexpr (BYTE_OFF) {
      const char* ak = Abytes + (long)k0 * 2;
      // the instruction itself is asm: left to hipcc, (base + offset) is re-associated per piece into VGPR pairs + k, or the
      // zero-extension of the offset is hoisted out of the block where instruction selection would fold it - either way one
      // 64-bit VALU add per piece
      glds16_saddr(ak, a_off[i], smem + slot * A_BYTES + (i * NS + swave * 64) * 16);
    }
    else glds16(Ab + a_off[i] + k0, smem + slot * A_BYTES + (i * NS + swave * 64) * 16);
  };
  auto stage_w_piece = [&](int slot, int k0, int i) {
    if constexpr (BYTE_OFF) {
      const char* wk = Wbytes + (long)k0 * 2;
      glds16_saddr(wk, b_off[i], smem + wslot(slot) + (i * NS + swave * 64) * 16);
    }
    else glds16(Wb + b_off[i] + k0, smem + wslot(slot) + (i * NS + swave * 64) * 16);
  };

  // ---- fragment read offsets ----
  // (x + j * 2048) ^ 64 == (x ^ 64) + j * 2048 (bit 6 belongs to the 16-B-slot swizzle inside a 128-B row, j * 2048 is whole
  // rows): written the second way the K-slice-1 fragments of an operand share ONE base register and take their row offsets as
  // ds_read immediates; the first way hipcc kept one pre-XORed register per fragment and a v_add per fragment and K-tile - vector
  // instructions in the MEMORY segment, i.e. beside the partner wave's MFMAs on the same SIMD
#ifdef VC_GEMM_FRAG_XOR_PER_FRAGMENT      // (A/B builds: round 3's form)
#define FRAG_AT(x, off, kk) (((x) + (off)) ^ ((kk) * 64))
#else
#define FRAG_AT(x, off, kk) ((((x) ^ ((kk) * 64))) + (off))
#endif
  const int fr = lane & 15, fq = lane >> 4;
  const int sw0 = ((fq ^ (lane & 7)) << 4);  // kk=0 slot; kk=1 is sw0 ^ 64
  const int a_rd = (wm * TM + fr) * 128 + sw0;
  const int b_rd = A_BYTES + (wn * TN + fr) * 128 + sw0;

  f32x4 acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  if constexpr (PP == 0) {
    // ---- simple schedule: double-buffered LDS, one barrier per K-tile ----
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
  } else if constexpr (PP == 2) {
    // ---- ping-pong schedule with LOADER WAVES: the 8 compute waves alternate MEMORY (ds_read only) and COMPUTE
    // segments exactly as in the PP == 1 schedule below, but every LDS-DMA is issued by 4 extra waves (one per SIMD,
    // 3 waves per SIMD in all, <= 168 VGPRs).  An LDS-DMA wave-instruction holds its wave's issue slot for ~60-180
    // cycles; inside a compute wave's MEMORY segment that made the segment ~2x the partner's COMPUTE segment
    // (tools/gemm_l2hot.py: loop without DMA 1.7 PFLOP/s, loop without MFMA 1.6, both together 1.1-1.2).  The loaders
    // take part in every workgroup barrier (s_barrier counts all live waves), so their per-interval work is bounded:
    // during K-tile t they issue A(t+1) then W(t+2) over intervals 4t, 4t+1, 4t+2 (both slots' last readers, group 1's
    // M(t-1,1), retired before the barrier ending 4t-1) and wait with vmcnt(#W pieces) before the barrier ending
    // 4t+3: A(t+1) and W(t+1) have landed, W(t+2) - streamed from HBM, the long-latency operand - keeps flying.
    static_assert(NCW == 8 && A_CH % NS == 0 && B_CH % NS == 0, "loader-wave schedule: 8 compute waves, whole sweeps");
    auto bar = [&]() {
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
    };
    constexpr int NPIECE = A_IT + B_IT;                 // wave-instructions per loader wave per K-tile
    constexpr int P0 = (NPIECE + 2) / 3, P1 = (NPIECE - P0 + 1) / 2;   // pieces issued in intervals 4t / 4t+1 (rest: 4t+2)
    constexpr int WD = 2;                               // W is issued WD tiles ahead into a ring of WD+1 slots
    if (wave >= NCW) {
      staging_offsets();
      if (!PERSIST || first_tile) {
#pragma unroll
        for (int i = 0; i < A_IT; ++i) stage_a_piece(0, 0, i);
#pragma unroll
        for (int d = 0; d < WD; ++d)
          if (d < nk) {
#pragma unroll
            for (int i = 0; i < B_IT; ++i) stage_w_piece(d, d * BK, i);
          }
        // the first barrier needs A(0) and W(0) only: W(1) - the last B_IT pieces issued - keeps flying, exactly as W(t+2)
        // does in the steady state (it is waited for by the vmcnt(B_IT) that ends K-tile 0, one tile before its first reader)
        if (nk > 1) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(B_IT) : "memory");
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      } else {
        // this tile's W(0), W(1) (under the previous epilogue) and A(0) (after it) were issued at the end of the previous
        // pass: A(0) went out last, everything before it has had an epilogue's time to land
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      }
      bar();
      int wsd = WD;                                     // W slot of tile kt+WD
      for (int kt = 0; kt < nk; ++kt) {
        const int as1 = (kt & 1) ^ 1, k1 = (kt + 1) * BK, kd = (kt + WD) * BK;
        const bool more1 = kt + 1 < nk, mored = kt + WD < nk;
        if constexpr (CONV) conv_set_k(k1);
        // piece j of round t: j < A_IT -> A(t+1) piece j, else W(t+2) piece j - A_IT
        auto piece = [&](int j) {
#ifdef VC_GEMM_NO_DMA     // analysis builds only: core side of the loop alone
          return;
#endif
#pragma unroll
          for (int i = 0; i < A_IT; ++i)
            if (j == i && more1) stage_a_piece(as1, k1, i);
#pragma unroll
          for (int i = 0; i < B_IT; ++i)
            if (j == A_IT + i && mored) stage_w_piece(wsd, kd, i);
        };
#pragma unroll
        for (int j = 0; j < P0; ++j) piece(j);
        bar();
#pragma unroll
        for (int j = P0; j < P0 + P1; ++j) piece(j);
        bar();
#pragma unroll
        for (int j = P0 + P1; j < NPIECE; ++j) piece(j);
        bar();
        // A(t+1) and W(t+1) must have landed; W(t+2) - the last B_IT pieces issued - keeps flying
        if (mored) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(B_IT) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        bar();
        wsd = wsd == WD ? 0 : wsd + 1;
      }
      bar();
    } else {
      const int grp = wave >> 2;
      // static priority for the younger half (waves 4-7 lose every age-based arbitration otherwise) and no per-segment
      // s_setprio flips: +0.5...2.5 % over setprio(1) around each MFMA block (tools/gemm_ab.py)
      if (grp == 1) __builtin_amdgcn_s_setprio(1);
      // the accumulators start at the bias (6 8-B loads per lane, hidden under the wait for the first K-tile): epilogue
      // pass 1 is then convert + LDS write only (it was VALU-bound on the bias unpack/add: 6.9 k cycles per block)
      if (P.bias && (!SPLITK || sk_bias)) {             // (a split-K tile: the piece that starts its K range carries the bias)
#pragma unroll
        for (int j = 0; j < NI; ++j) {
          const int col = n0 + wn * TN + j * 16 + fq * 4;
          u32x2 bb = {0u, 0u};
          if (col < N) bb = *(const u32x2*)((const bf16_t*)P.bias + col);
          const f32x4 b4 = {lo_bf(bb[0]), hi_bf(bb[0]), lo_bf(bb[1]), hi_bf(bb[1])};
#pragma unroll
          for (int i = 0; i < MI; ++i) acc[i][j] = b4;
        }
      }
      bar();
      VC_PHASE_STAMP(1);
      if (grp == 1) bar();
      // A compute wave whose TM rows all lie past row M - 1 (the last m-tile of M = 3968 rows: two of its four wave rows) has
      // nothing to accumulate: it keeps the K loop's barriers - four per K-tile - and skips its fragment reads and MFMAs.  Its
      // SIMD's other waves run on; what comes back is power (the board runs at its cap), not time.  The epilogue is the same
      // for every wave (rows >= M are never stored).
#ifdef VC_GEMM_NO_DEAD_WAVES      // A/B builds
      constexpr bool wdead = false;
#else
      const bool wdead = __builtin_amdgcn_readfirstlane((int)(m0 + wm * TM >= M)) != 0;
#endif
      if (wdead) {
        for (int kt = 0; kt < nk; ++kt) { bar(); bar(); bar(); bar(); }
      } else
#if !defined(VC_GEMM_NO_LDSREAD) && !defined(VC_GEMM_NO_MFMA) && !defined(VC_GEMM_NO_SLOT_UNROLL)
      // The K loop unrolled over the ring period (A: 2 slots, W: 3 slots -> 6 K-tiles), so that every fragment address is ONE of
      // six lane-constant base registers + an immediate: no vector instruction at all in the MEMORY segment, which runs beside
      // the partner wave's MFMAs (with the slot offsets in scalar registers it was 5 v_add per K-tile, 11 before round 4).
      // b_hi: the W slots 1 and 2 lie beyond the 16-bit ds_read immediate of b_rd.
      if constexpr (!PERSIST) {
        uint32_t a_k[2] = {(uint32_t)a_rd, (uint32_t)a_rd ^ 64u}, b_lo[2] = {(uint32_t)b_rd, (uint32_t)b_rd ^ 64u}, b_hi[2];
        constexpr int HI = B_BYTES * 2;              // (slot 1 starts at W_RING0 + B_BYTES; b_rd already carries A_BYTES)
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
          b_hi[kk] = b_lo[kk] + HI;
          asm volatile("" : "+v"(a_k[kk]), "+v"(b_lo[kk]), "+v"(b_hi[kk]));      // opaque: not to be re-derived from one another in the loop
        }
        auto ktile = [&](auto AS, auto WS) {
          constexpr int as = decltype(AS)::value, wsl = decltype(WS)::value;
          constexpr int a_c = as * A_BYTES;
          constexpr int w_c = W_RING0 + wsl * B_BYTES - A_BYTES - (wsl > 0 ? HI : 0);
          static_assert(a_c + (MI - 1) * 2048 < 65536 && w_c >= 0 && w_c + (NI - 1) * 2048 < 65536, "fragment offsets must fit the ds_read immediate");
#pragma unroll
          for (int kk = 0; kk < 2; ++kk) {
            bf16x8 af[MI], bfr[NI];
            const uint32_t wb = wsl > 0 ? b_hi[kk] : b_lo[kk];
#pragma unroll
            for (int i = 0; i < MI; ++i) af[i] = *(const bf16x8*)(smem + a_k[kk] + (a_c + i * 16 * 128));
#pragma unroll
            for (int j = 0; j < NI; ++j) bfr[j] = *(const bf16x8*)(smem + wb + (w_c + j * 16 * 128));
            // lgkmcnt(0) as a builtin: the compiler's counter model sees the fragments landed (after an asm wait it re-inserted
            // lgkmcnt(5) ... (0) between the first MFMAs - trivially satisfied, measured +-0.00 % either way)
            __builtin_amdgcn_s_waitcnt(0xc07f);
            bar();
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
              for (int j = 0; j < NI; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
            bar();
          }
        };
        using S0 = std::integral_constant<int, 0>; using S1 = std::integral_constant<int, 1>; using S2 = std::integral_constant<int, 2>;
        for (int kt = 0;;) {
          ktile(S0{}, S0{}); if (++kt >= nk) break;
          ktile(S1{}, S1{}); if (++kt >= nk) break;
          ktile(S0{}, S2{}); if (++kt >= nk) break;
          ktile(S1{}, S0{}); if (++kt >= nk) break;
          ktile(S0{}, S1{}); if (++kt >= nk) break;
          ktile(S1{}, S2{}); if (++kt >= nk) break;
        }
      } else
#endif
      {
      int ws = 0;                                       // W slot of tile kt
#ifdef VC_GEMM_NO_LDSREAD   // analysis builds only (tools/gemm_power.py): the fragments of K-slice 0 feed every MFMA
        bf16x8 af[MI], bfr[NI];
#endif
        for (int kt = 0; kt < nk; ++kt) {
          const char* base_a = smem + (kt & 1) * A_BYTES;
          const char* base_b = smem + wslot(ws) - A_BYTES;                 // b_rd already carries +A_BYTES
#pragma unroll
          for (int kk = 0; kk < 2; ++kk) {
#ifdef VC_GEMM_NO_LDSREAD
            if (kt == 0 && kk == 0)
#else
            bf16x8 af[MI], bfr[NI];
#endif
            {
#pragma unroll
              for (int i = 0; i < MI; ++i) af[i] = *(const bf16x8*)(base_a + FRAG_AT(a_rd, i * 16 * 128, kk));
#pragma unroll
              for (int j = 0; j < NI; ++j) bfr[j] = *(const bf16x8*)(base_b + FRAG_AT(b_rd, j * 16 * 128, kk));
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            bar();
#ifdef VC_GEMM_NO_MFMA
#pragma unroll
            for (int i = 0; i < MI; ++i) asm volatile("" ::"v"(af[i]));
#pragma unroll
            for (int j = 0; j < NI; ++j) asm volatile("" ::"v"(bfr[j]));
#else
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
              for (int j = 0; j < NI; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
#endif
            bar();
          }
          ws = ws == WD ? 0 : ws + 1;
        }
      }
      if (grp == 0) bar();
    }
  } else {
    // ---- ping-pong schedule (8 waves = 2 per SIMD): the two waves of a SIMD alternate between a MEMORY segment
    // (ds_read the next K=32 slice of fragments) and a COMPUTE segment (MI*NI MFMAs), separated by workgroup
    // barriers; the younger half (waves 4-7, group 1) runs one barrier interval behind, so every SIMD always has
    // one wave feeding the matrix pipe while its partner loads.
    //   interval  4t       4t+1   4t+2   4t+3   4t+4        M(t,kk) = reads of tile t slice kk, C = its MFMAs
    //   group 0   M(t,0)   C(t,0) M(t,1) C(t,1) M(t+1,0)
    //   group 1   C(t-1,1) M(t,0) C(t,0) M(t,1) C(t,1)
    // Tile t+1's LDS-DMA is issued at the top of M(t,0) (interval 4t for group 0, 4t+1 for group 1): its buffer's
    // last readers, group 1's M(t-1,1), retired their ds_reads with lgkmcnt(0) before the barrier ending interval
    // 4t-1.  Every wave waits for its own pieces (vmcnt(0)) before the barrier ending interval 4t+3, one full
    // interval before the first read.  (Measured alternatives that did NOT help on MI355X, see DESIGN.md: spreading
    // the DMA issue over other segments; whole-K-tile segments; a 4-deep ring of K=32 stages - its 64-B row
    // pieces halve the bytes used per 128-B line and lost 20 %; the same loop on v_mfma_f32_32x32x16_bf16: -15 %.)
    static_assert(WM * WN == 8, "ping-pong schedule needs 8 waves");
    const int grp = wave >> 2;
    auto bar = [&]() {
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
    };
#ifdef VC_GEMM_TIMESTAMPS   // profiling builds only (tools/gemm_ts.py): s_memtime stamps of block 0, waves 0 and 4
    uint64_t* ts = (args.debug_ts && blockIdx.x == 0 && lane == 0 && (wave == 0 || wave == 4))
                       ? args.debug_ts + (wave >> 2) * 4096 : nullptr;
    int tsi = 0;
    auto stamp = [&]() { if (ts && tsi < 4096) ts[tsi++] = __builtin_amdgcn_s_memtime(); };
#else
    auto stamp = [&]() {};
#endif
    stage(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    bar();
    if (grp == 1) bar();
    for (int kt = 0; kt < nk; ++kt) {
      const int cur = kt & 1;
      const char* base = smem + cur * STAGE_BYTES;
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        // ---- memory segment
        stamp();
        if (kk == 0 && kt + 1 < nk) stage(cur ^ 1, (kt + 1) * BK);
        bf16x8 af[MI], bfr[NI];
#pragma unroll
        for (int i = 0; i < MI; ++i) af[i] = *(const bf16x8*)(base + ((a_rd + i * 16 * 128) ^ (kk * 64)));
#pragma unroll
        for (int j = 0; j < NI; ++j) bfr[j] = *(const bf16x8*)(base + ((b_rd + j * 16 * 128) ^ (kk * 64)));
        stamp();
        if (kk == 1 && grp == 1) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        stamp();
        bar();
        stamp();
        // ---- compute segment
        __builtin_amdgcn_s_setprio(1);
#ifdef VC_GEMM_NO_MFMA   // analysis builds only (tools/gemm_l2hot.py): memory side of the loop alone
#pragma unroll
        for (int i = 0; i < MI; ++i) asm volatile("" ::"v"(af[i]));
#pragma unroll
        for (int j = 0; j < NI; ++j) asm volatile("" ::"v"(bfr[j]));
#else
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
          for (int j = 0; j < NI; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
#endif
        __builtin_amdgcn_s_setprio(0);
        stamp();
        if (kk == 1 && grp == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        stamp();
        bar();
      }
    }
    if (grp == 0) bar();
  }

  VC_PHASE_STAMP(2);
  // PERSIST: the next tile (decoded once more at the top of the next pass; scalar work)
  const Tile tile_next = has_next ? decode(id_cur + id_step) : tile_cur;
  auto epilogue = [&]() {
  int etid = tid;                       // opaque copy: keeps the epilogue's address arithmetic below the K loop
  asm volatile("" : "+v"(etid));
  const int efr = PERSIST ? (etid & 15) : fr, efq = PERSIST ? ((etid & 63) >> 4) : fq;   // (PERSIST: nor above the tile loop)
  // ---- gate/residual epilogue of the loader-wave kernel: the residual chunks (and the gate chunk: the column of a
  // thread is the same in every pass-2 iteration) are requested NOW, so their HBM latency runs under pass 1 instead of
  // twice inside pass 2 (pass 2 was 10.6 k cycles per block against 2.7 k for the plain epilogue) ----
  constexpr int CPR = BN / 8;  // 16-B chunks per tile row
  constexpr int P2_IT = (BM * CPR + NT - 1) / NT;
  // pass 2 walks chunks c = etid + it * NT; where NT is a multiple of CPR (every product tile: 768 = 32 * 24 = 48 * 16) the
  // iterations of a thread are ROW_STEP rows apart in ONE 16-B column, so (row, column) need one division per thread, not one per
  // iteration, and C's row offset advances by a constant (the index arithmetic was ~25 of the ~150 instructions of an iteration
  // of this VALU-bound loop)
  constexpr bool STEP_ROWS = NT % CPR == 0;
  constexpr int ROW_STEP = NT / CPR;
  const int row_first = etid / CPR, cc_first = etid - row_first * CPR;
  constexpr bool PREF = PP == 2 && EPI == VC_EPI_GATE_RES && NT % CPR == 0;
  u32x4 rr_pre[PREF ? P2_IT : 1], gg_pre = {0u, 0u, 0u, 0u};
  int gg_batch = -1;
  long gate_step = 0;
  if (EPI == VC_EPI_GATE_RES && args.step_ptr) gate_step = (long)(*args.step_ptr) * args.gate_step_stride;
  // batch index of a tile row for the gate: rows_per_batch >= BM means at most one batch edge inside the tile
  int gb_lo = 0, gb_edge = 0x7fffffff;
  if (EPI == VC_EPI_GATE_RES) { gb_lo = m0 / P.rows_per_batch; gb_edge = (gb_lo + 1) * P.rows_per_batch; }
  auto gate_batch = [&](int m) { return P.rows_per_batch >= BM ? gb_lo + (m >= gb_edge ? 1 : 0) : m / P.rows_per_batch; };
  if constexpr (PREF) {
    const bf16_t* __restrict__ res = (const bf16_t*)P.res;
    const int n = n0 + cc_first * 8;
#pragma unroll
    for (int it = 0; it < P2_IT; ++it) {
      const int m = m0 + (STEP_ROWS ? row_first + it * ROW_STEP : (etid + it * NT) / CPR);
      rr_pre[it] = u32x4{0u, 0u, 0u, 0u};
      if (m < M && n < N && (etid + it * NT) < BM * CPR) {
        const long rrow = P.c_rpb > 0 ? (long)(m / P.c_rpb) * P.c_bstride + (long)(m % P.c_rpb) * P.ldc : (long)m * P.ldres;
        rr_pre[it] = *(const u32x4*)(res + rrow + n);
      }
    }
    const int mf = m0 + row_first;
    if (mf < M && n < N) {
      gg_batch = gate_batch(mf);
      gg_pre = *(const u32x4*)((const bf16_t*)P.gate + gate_step + (long)gg_batch * P.gate_bstride + n);
    }
  }
  // ---- epilogue, pass 1: lane holds C[m = ..+fr][n = ..+fq*4 .. +3]; t = bf16(acc + bias) -> LDS tile ----
  // (the K loop ended on a barrier, so the staging buffers are free).  Rows are padded by 16 B: the 16 rows a
  // ds_write_b64 lane group touches then land on distinct bank pairs (2-way at worst).
  constexpr int EP_LD = BN * 2 + 16;
  // EPI_QKV: a tile that lies wholly in the V column range goes through LDS TRANSPOSED ([BN][BM] + 16 B per row) and
  // leaves as 16-B runs of 8 consecutive tokens per (head, dim) row of vt; a tile that straddles vt_col0 (or unaligned
  // row geometry) takes the ordinary path and scatters its V elements one by one (test geometries only).
  constexpr int EPT_LD = BM * 2 + 16;
  // EPI_QKV with kn_heads = H > 0: the weight rows (output columns) arrive HEAD-PERMUTED (vcloze_hip.h): 2H blocks of
  // [head t (128): query head t for t < H, key head t - H after | V columns 64 t .. 64 t + 63] - every 192-wide tile holds ONE
  // whole query or key head, and with qn_scale / kn_scale the epilogue applies QKNorm + RoPE to it (layers.py:63-84,
  // math.py:112-117) before the row leaves: the "QKV + RoPE fused projection"; its 64 V columns leave transposed into vt.  C is
  // always written at the LOGICAL columns (q | k | v).  A 192- or 128-wide tile holds at most ONE run of 64 V columns, at tile
  // column v_lo; its staging image (vkind 4) is [BM] rows of 256 + 16 B for the other columns (in place), then the V run -
  // transposed, [64][BM * 2 + 16 B], when vt takes whole 16-B runs (vfast), else [BM] rows of 128 + 16 B (192-wide tiles only).
  // Every other case (vkind 5) maps each 16-B chunk of the generic pass 2 to its logical place.
  int vkind = 0, v_lo = -1;
  const int knH = EPI == VC_EPI_QKV ? P.kn_heads : 0;
  constexpr bool MIXED = EPI == VC_EPI_QKV && (BN == 192 || BN == 128);
  constexpr int EPH_LD = 128 * 2 + 16, EPV_LD = 64 * 2 + 16, EPV0 = BM * EPH_LD;
  bool vfast = false;
  auto qkv_col = [&](int n) {        // permuted column (multiple of 8) -> logical column
    const int t = n / 192, j = n - 192 * t;
    return j < 128 ? 128 * t + j : 256 * knH + 64 * t + j - 128;
  };
  if constexpr (EPI == VC_EPI_QKV) {
    const bool aligned = P.vt && ((P.vt_rpb | P.vt_row0 | P.vt_lpad | (int)(P.vt_bstride & 7)) & 7) == 0;
    if (knH > 0) {
      vkind = 5;
      if (BN == 192) { vkind = 4; v_lo = 128; vfast = aligned; }
      else if (BN == 128 && aligned && n0 % 384 != 0) { vkind = 4; v_lo = n0 % 384 == 128 ? 0 : 64; vfast = true; }
    } else if (P.vt) {
      vkind = n0 >= P.vt_col0 ? (aligned ? 1 : 2) : (n0 + BN > P.vt_col0 ? 2 : 0);
    }
    vkind = __builtin_amdgcn_readfirstlane(vkind);
    v_lo = __builtin_amdgcn_readfirstlane(v_lo);
  }
  const bf16_t* __restrict__ bias = (const bf16_t*)P.bias;
  // The staging form is decided ONCE, outside the unrolled (i, j) loops: KIND and the wave's first V block JV0 are compile-time
  // inside `stage`.  (Decided per accumulator block - vkind, vfast and the column test as run-time branches inside the loops -
  // pass 1 of the qkv tiles was a maze of ~6 scalar branches per block, 24 blocks per lane: 9.8 k ticks against 2.8 k for the
  // plain epilogue, tools/qkv_epilogue_phases.py.)
  //   KIND 0 row-major [BM][EP_LD]     1 transposed [BN][EPT_LD] (natural-order V tile)
  //        2 head-permuted tile, V run transposed (vfast)     3 head-permuted tile, V run row-major [BM][EPV_LD]
  //   blocks j >= JV0 of this wave are the tile's V columns (KIND 2 / 3; NI = none)
  auto stage = [&](auto KIND, auto JV0c) {
    constexpr int KD = decltype(KIND)::value, JV0 = decltype(JV0c)::value;
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      const int row = wm * TM + i * 16 + efr;
#pragma unroll
      for (int j = 0; j < NI; ++j) {
        const int col = wn * TN + j * 16 + efq * 4;
        float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
        if (PP == 2) {
          // bias already in the accumulators
        } else if (bias && n0 + col < N) {
          const u32x2 bb = *(const u32x2*)(bias + n0 + col);
          v[0] += lo_bf(bb[0]); v[1] += hi_bf(bb[0]); v[2] += lo_bf(bb[1]); v[3] += hi_bf(bb[1]);
        }
        u32x2 o;
        o[0] = pack2bf(v[0], v[1]);
        o[1] = pack2bf(v[2], v[3]);
        if constexpr (KD == 1) {
#pragma unroll
          for (int e = 0; e < 4; ++e) *(bf16_t*)(smem + (col + e) * EPT_LD + row * 2) = (bf16_t)(o[e >> 1] >> (16 * (e & 1)));
        } else if constexpr (KD >= 2) {
          if (j < JV0) {
            *(u32x2*)(smem + row * EPH_LD + col * 2) = o;
          } else if constexpr (KD == 2) {
#pragma unroll
            for (int e = 0; e < 4; ++e) *(bf16_t*)(smem + EPV0 + (col - v_lo + e) * EPT_LD + row * 2) = (bf16_t)(o[e >> 1] >> (16 * (e & 1)));
          } else {
            *(u32x2*)(smem + EPV0 + row * EPV_LD + (col - v_lo) * 2) = o;
          }
        } else {
          *(u32x2*)(smem + row * EP_LD + col * 2) = o;
        }
      }
    }
  };
  using K0 = std::integral_constant<int, 0>; using K1 = std::integral_constant<int, 1>;
  using K2 = std::integral_constant<int, 2>; using K3 = std::integral_constant<int, 3>;
  using JN = std::integral_constant<int, NI>; using J0 = std::integral_constant<int, 0>;
  using J2 = std::integral_constant<int, (NI > 2 ? 2 : 0)>;
  if (!(PP == 2 && wave >= NCW)) {     // loader waves hold no accumulators
    if constexpr (MIXED) {
      if (vkind == 4) {
        // a wave's TN columns are whole 16-column blocks on one side of the V run or the other: 192-wide tiles (TN = 96) put the
        // run [128, 192) into blocks j >= 2 of the wn = 1 waves; 128-wide tiles (TN = 64) give one wave column the whole run
        static_assert(TN == (BN == 192 ? 96 : 64), "the V run must start on a block boundary of one wave column");
        const bool v_wave = BN == 192 ? wn == 1 : wn * TN == v_lo;
        if (!v_wave) stage(K2{}, JN{});
        else if (BN == 192) { if (vfast) stage(K2{}, J2{}); else stage(K3{}, J2{}); }
        else stage(K2{}, J0{});           // (128-wide tiles take vkind 4 only with vfast)
      } else if (vkind == 1) {
        stage(K1{}, JN{});
      } else {
        stage(K0{}, JN{});
      }
    } else if constexpr (EPI == VC_EPI_QKV) {
      if (vkind == 1) stage(K1{}, JN{});
      else stage(K0{}, JN{});
    } else {
      stage(K0{}, JN{});
    }
  }
  if constexpr (PERSIST) {
    // the seam: the loaders compute the next tile's offsets (a_off / b_off are dead once the K loop has issued its last
    // pieces) and send W(0), W(1) of the next tile into ring slots 0 and 1 - above the staging area the other waves are
    // writing right now; every reader of those slots passed the barrier that ended the K loop
    if (wave >= NCW && has_next) {
      int st = stid;
      asm volatile("" : "+v"(st));     // (nothing of this is to be hoisted above the K loop)
      uint32_t bn[B_IT];
      staging_offsets_b(bn, tile_next.P, tile_next.n0, st);
      const char* Wn = BYTE_OFF ? (const char*)((const bf16_t*)tile_next.P.W + (long)tile_next.n0 * tile_next.P.ldw) : (const char*)tile_next.P.W;
      const int nkn = tile_next.P.K / BK;
#pragma unroll
      for (int d = 0; d < 2; ++d)
        if (d < nkn) {
#pragma unroll
          for (int i = 0; i < B_IT; ++i) glds16((Wn + d * BK * 2) + (BYTE_OFF ? (long)bn[i] : 2L * bn[i]), smem + wslot(d) + (i * NS + swave * 64) * 16);
        }
    }
  }
  if constexpr (PREF || PERSIST) {   // LDS writes done; the residual prefetch / the next tile's W pieces stay in flight
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // (__syncthreads would wait for them: vmcnt(0))
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  } else {
    __syncthreads();
  }
  VC_PHASE_STAMP(3);

  if constexpr (EPI == VC_EPI_QKV) {
    bf16_t* __restrict__ vt = (bf16_t*)P.vt;
    auto vt_at = [&](int m, int n) -> bf16_t* {
      const int b = m / P.vt_rpb;
      return vt + (long)b * P.vt_bstride + (long)(n - P.vt_col0) * P.vt_lpad + P.vt_row0 + (m - b * P.vt_rpb);
    };
    if (vkind == 1) {      // transposed tile: a lane moves 8 consecutive tokens of one V column = 16 B of one vt row
      constexpr int CPRT = BM / 8;
#pragma unroll 4
      for (int c = etid; c < BN * CPRT; c += NT) {
        const int trow = c / CPRT, cc = c % CPRT;
        const int n = n0 + trow, m = m0 + cc * 8;
        if (n >= N || m >= M) continue;
        const u32x4 tw = *(const u32x4*)(smem + trow * EPT_LD + cc * 16);
        if (m + 8 <= M) {
          *(u32x4*)vt_at(m, n) = tw;       // vt_rpb % 8 == 0 and m % 8 == 0: the 8 tokens share a batch element
        } else {
#pragma unroll
          for (int e = 0; e < 8; ++e)
            if (m + e < M) *vt_at(m + e, n) = (bf16_t)(tw[e >> 1] >> (16 * (e & 1)));
        }
      }
      VC_PHASE_STAMP(4);
      return;
    }
  }
  // ---- pass 2: row-major, 16 B per lane, whole rows per wave-instruction -> coalesced HBM traffic ----
  bf16_t* __restrict__ C = (bf16_t*)P.C;
  const bf16_t* __restrict__ res = (const bf16_t*)P.res;
  const bf16_t* __restrict__ gate = (const bf16_t*)P.gate;
  if constexpr (MIXED) {
    if (vkind == 4) {
      // 192-wide tile: columns 0..127 = head t: 16 lanes own one row (8 elements each) - RMS over the 128 by 4 xor-shuffles,
      // scale, RoPE on the interleaved pairs with the token's f32 (cos, sin) row; qknorm_rope8 (common.h) is the one definition
      // the pre-pass kernels of norm.hip use too: same bits as GEMM + pre-pass.  A head whose scale is NULL leaves as it is, and
      // so do the 64 columns of a 128-wide tile that are not its V run.
      const int t = (n0 + v_lo) / 192;                   // the 192-block the V run (and, BN == 192, the head) belongs to
      // Batch element of a tile row WITHOUT a division per row: with rows-per-batch >= BM a tile meets at most one batch edge, so
      // the index is the tile's first (one scalar division) + (row beyond the edge).  (m / vt_rpb, m / c_rpb and m % c_rpb per
      // 16-B chunk were ~70 of the ~170 vector instructions of a chunk in this VALU-bound pass.)
      struct RowSplit { int rpb, lo, edge; bool fast; };
      auto split_of = [&](int rpb) {
        RowSplit r;
        r.rpb = rpb; r.fast = rpb >= BM;
        r.lo = rpb > 0 ? __builtin_amdgcn_readfirstlane(m0 / rpb) : 0;
        r.edge = (r.lo + 1) * rpb;
        return r;
      };
      auto batch_of = [&](const RowSplit& r, int m) { return r.fast ? r.lo + (m >= r.edge ? 1 : 0) : m / r.rpb; };
      const RowSplit vsp = split_of(P.vt_rpb), csp = split_of(P.c_rpb);
      auto c_row = [&](int m) -> long {                  // element offset of C's row m (batch-strided rows: see VcGemmProblem)
        if (P.c_rpb <= 0) return (long)m * P.ldc;
        const int b = batch_of(csp, m);
        return (long)b * P.c_bstride + (long)(m - b * P.c_rpb) * P.ldc;
      };
      const bf16_t* __restrict__ hsc = BN == 192 ? (const bf16_t*)(t < knH ? P.qn_scale : P.kn_scale) : nullptr;
      const float post = (t < knH && P.qn_prescale) ? VC_QK_PRESCALE : 1.0f;
      float g[8] = {};
      if (hsc) {
        const u32x4 sw = *(const u32x4*)(hsc + (etid & 15) * 8);
#pragma unroll
        for (int e = 0; e < 4; ++e) { g[2 * e] = lo_bf(sw[e]); g[2 * e + 1] = hi_bf(sw[e]); }
      }
      for (int c = etid; c < BM * 16; c += NT) {          // NT % 16 == 0: a row's 16 lanes stay together
        const int row = c >> 4, sub = c & 15;
        if (BN == 128 && sub * 8 >= v_lo && sub * 8 < v_lo + 64) continue;      // (never with a head to normalise)
        const int m = min(m0 + row, M - 1);
        u32x4 o = *(const u32x4*)(smem + row * EPH_LD + sub * 16);
        if (hsc) {
          const int b = batch_of(vsp, m);
          const float* rp = P.kn_rope + (long)b * P.kn_rope_bstride + (long)(P.vt_row0 + (m - b * P.vt_rpb)) * 128 + sub * 8;
          const f32x4 c0 = *(const f32x4*)rp;
          const f32x4 c1 = *(const f32x4*)(rp + 4);
          const float cs[8] = {c0[0], c0[1], c0[2], c0[3], c1[0], c1[1], c1[2], c1[3]};
          o = qknorm_rope8(o, g, cs, post);
        }
        // (192-wide tile: the chunk is 8 columns of head t; a 128-wide one maps its chunk through the general formula)
        if (m0 + row < M) *(u32x4*)(C + c_row(m) + (BN == 192 ? 128 * t + sub * 8 : qkv_col(n0 + sub * 8))) = o;
      }
      // the V run: V columns 64 t .. 64 t + 63
      bf16_t* __restrict__ vtp = (bf16_t*)P.vt;
      if (vfast) {         // a lane moves 8 consecutive tokens of one V column = 16 B of one vt row
        constexpr int CPRT = BM / 8;
#pragma unroll 2
        for (int c = etid; c < 64 * CPRT; c += NT) {
          const int trow = c / CPRT, cc = c % CPRT;
          const int m = m0 + cc * 8;
          if (m >= M) continue;
          const u32x4 tw = *(const u32x4*)(smem + EPV0 + trow * EPT_LD + cc * 16);
          const int b = batch_of(vsp, m);
          bf16_t* d = vtp + (long)b * P.vt_bstride + (long)(64 * t + trow) * P.vt_lpad + P.vt_row0 + (m - b * P.vt_rpb);
          if (m + 8 <= M) {
            *(u32x4*)d = tw;               // vt_rpb % 8 == 0 and m % 8 == 0: the 8 tokens share a batch element
          } else {
#pragma unroll
            for (int e = 0; e < 8; ++e)
              if (m + e < M) d[e] = (bf16_t)(tw[e >> 1] >> (16 * (e & 1)));     // (same batch element: M ends it)
          }
        }
      } else {
        for (int c = etid; c < BM * 8; c += NT) {
          const int row = c >> 3, sub = c & 7;
          const int m = m0 + row;
          if (m >= M) continue;
          const u32x4 tw = *(const u32x4*)(smem + EPV0 + row * EPV_LD + sub * 16);
          if (vtp) {        // 8 V columns of one token: 8 rows of vt
            const int b = m / P.vt_rpb;
            bf16_t* d = vtp + (long)b * P.vt_bstride + (long)(64 * t + sub * 8) * P.vt_lpad + P.vt_row0 + (m - b * P.vt_rpb);
#pragma unroll
            for (int e = 0; e < 8; ++e) d[(long)e * P.vt_lpad] = (bf16_t)(tw[e >> 1] >> (16 * (e & 1)));
          } else {
            const long crow = P.c_rpb > 0 ? (long)(m / P.c_rpb) * P.c_bstride + (long)(m % P.c_rpb) * P.ldc : (long)m * P.ldc;
            *(u32x4*)(C + crow + 256 * knH + 64 * t + sub * 8) = tw;
          }
        }
      }
      VC_PHASE_STAMP(4);
      return;
    }
  }
  const long crow_first = (long)(m0 + row_first) * P.ldc, crow_step = (long)ROW_STEP * P.ldc;
#pragma unroll PREF ? P2_IT : 4
  for (int it = 0; it < (PREF ? P2_IT : BM * CPR); ++it) {
    const int c = etid + it * NT;
    if ((!PREF || BM * CPR % NT != 0) && c >= BM * CPR) break;
    const int row = STEP_ROWS ? row_first + it * ROW_STEP : c / CPR, cc = STEP_ROWS ? cc_first : c % CPR;
    const int m = m0 + row, n = n0 + cc * 8;
    if (m >= M || n >= N) continue;
    const long crow = P.c_rpb > 0 ? (long)(m / P.c_rpb) * P.c_bstride + (long)(m % P.c_rpb) * P.ldc
                                  : (STEP_ROWS ? crow_first + it * crow_step : (long)m * P.ldc);
    const u32x4 tw = *(const u32x4*)(smem + row * EP_LD + cc * 16);
    float v[8];
#pragma unroll
    for (int e = 0; e < 4; ++e) { v[2 * e] = lo_bf(tw[e]); v[2 * e + 1] = hi_bf(tw[e]); }
    u32x4 o = tw;
#ifdef VC_GEMM_NO_GELU      // analysis builds only (tools/step_ab.py): what ANY cheaper GELU could win at most - the math left out
    if (false) {
#else
    if (EPI == VC_EPI_GELU) {
#endif
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const f32x2 g = gelu_tanh2(f32x2{v[2 * e], v[2 * e + 1]});
        o[e] = pack2bf(g[0], g[1]);
      }
    } else if (EPI == VC_EPI_SILU) {
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = pack2bf(silu_f(v[2 * e]), silu_f(v[2 * e + 1]));
    } else if (EPI == VC_EPI_GATE_RES) {
      u32x4 gg, rr;
      if constexpr (PREF) {
        const int bt = gate_batch(m);
        gg = gg_pre;
        if (bt != gg_batch) gg = *(const u32x4*)(gate + gate_step + (long)bt * P.gate_bstride + n);   // tile spans a batch edge
        rr = rr_pre[it];
      } else {
        gg = *(const u32x4*)(gate + gate_step + (long)gate_batch(m) * P.gate_bstride + n);
        rr = *(const u32x4*)(res + (P.c_rpb > 0 ? crow : (long)m * P.ldres) + n);
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) {   // out = res + bf16(gate * t), two elements per VALU op
        const f32x2 gv = f32x2{lo_bf(gg[e]), hi_bf(gg[e])} * f32x2{v[2 * e], v[2 * e + 1]};
        const uint32_t gb = pack2bf(gv[0], gv[1]);
        const f32x2 sum = f32x2{lo_bf(rr[e]), hi_bf(rr[e])} + f32x2{lo_bf(gb), hi_bf(gb)};
        o[e] = pack2bf(sum[0], sum[1]);
      }
    }
    const int nl = EPI == VC_EPI_QKV && knH > 0 ? qkv_col(n) : n;      // logical column of this 16-B chunk
    if (EPI == VC_EPI_QKV && P.vt && (vkind == 2 || knH > 0) && nl >= P.vt_col0) {   // 8 V columns of one token: 8 rows of vt
      const int b = m / P.vt_rpb;
      bf16_t* d = (bf16_t*)P.vt + (long)b * P.vt_bstride + (long)(nl - P.vt_col0) * P.vt_lpad + P.vt_row0 + (m - b * P.vt_rpb);
#pragma unroll
      for (int e = 0; e < 8; ++e) d[(long)e * P.vt_lpad] = (bf16_t)(o[e >> 1] >> (16 * (e & 1)));
      continue;
    }
    *(u32x4*)(C + crow + nl) = o;
  }
  VC_PHASE_STAMP(4);
  };   // epilogue
  if constexpr (SPLITK) {
    {
      // a split-K slice: the f32 accumulators leave row-major, 16 B per lane (the 4 lanes of a row's fq group write 64 B runs;
      // L2 merges the halves of a line), rows / columns beyond M / N included - the scratch is tile-sized, the reducer masks
      if (wave < NCW) {
        int pl = lane;                    // opaque copy: keeps this address arithmetic below the K loop
        asm volatile("" : "+v"(pl));
        float* __restrict__ wsp = (float*)args.splitk_ws + (long)sk_slot * (BM * BN) + (wm * TM + (pl & 15)) * BN + wn * TN + (pl >> 4) * 4;
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
          for (int j = 0; j < NI; ++j) *(f32x4*)(wsp + i * 16 * BN + j * 16) = acc[i][j];
      }
      if (args.sk_stream > 0) {          // a second segment: the start of the next tile (every wave decides alike)
        sk_it += nk;
        ++sk_seg;
        if (sk_it < sk_it_end) continue;
      }
      break;
    }
  } else {
  epilogue();
  }
  if constexpr (!PERSIST) {
    break;
  } else {
    if (!has_next) break;
    // every wave has read its share of the staging area back: the A ring is free again
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (wave >= NCW) {
      int st = stid;
      asm volatile("" : "+v"(st));
      uint32_t an[A_IT];
      staging_offsets_a(an, tile_next.P, tile_next.m0, st);
      const char* An = (const char*)tile_next.P.A;
#pragma unroll
      for (int i = 0; i < A_IT; ++i) glds16(An + (BYTE_OFF ? (long)an[i] : 2L * an[i]), smem + (i * NS + swave * 64) * 16);
    }
    id_cur += id_step;
  }
  }    // tiles
}

// slice 0 + slice 1 + ... + slice S-1 of one thread's 8 columns, in that order
template <int S, int STRIDE>
VC_DEV void sum_slices(const float* __restrict__ src, f32x4& a0, f32x4& a1) {
  f32x4 lo[S], hi[S];
#pragma unroll
  for (int s = 0; s < S; ++s) { lo[s] = *(const f32x4*)(src + (long)s * STRIDE); hi[s] = *(const f32x4*)(src + (long)s * STRIDE + 4); }
  a0 = lo[0]; a1 = hi[0];
#pragma unroll
  for (int s = 1; s < S; ++s) { a0 += lo[s]; a1 += hi[s]; }
}

// Second launch of a split-K call: sums the S f32 partial tiles of every remainder tile in slice order (slice 0 carries the
// bias) and applies the epilogue exactly as pass 2 of gemm_bf16_kernel does - t = bf16(sum), then GELU / SiLU / res +
// bf16(gate * t) - one thread per 16-B chunk of C, whole partial rows (768 B) per wave-instruction.  HBM-bound:
// rem * S * BM * BN * 4 bytes read (they sit in the Infinity Cache: the slices were written microseconds ago).
template <int BM, int BN, int EPI>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const VcGemmArgs args) {
  constexpr int CPR = BN / 8, PER_TILE = BM * CPR, BPT = PER_TILE / 256;     // a block lies inside ONE tile: the problem is scalar
  static_assert(PER_TILE % 256 == 0, "whole blocks per tile");
  const int r = blockIdx.x / BPT;
  const int cc = (blockIdx.x - r * BPT) * 256 + threadIdx.x, row = cc / CPR, ch = cc - row * CPR;
  int id = __builtin_amdgcn_readfirstlane(args.sk_full + r), pi = 0;
#pragma unroll
  for (int q = 1; q < VC_GEMM_MAX_PROBLEMS; ++q)
    if (q < args.nprob && id >= args.p[q].tile_start) pi = q;
  const VcGemmProblem P = pi == 3 ? args.p[3] : pi == 2 ? args.p[2] : pi == 1 ? args.p[1] : args.p[0];   // scalar selects, as decode()
  id -= P.tile_start;
  constexpr int GROUP_M = VC_GEMM_GROUP_M;        // the raster of gemm_bf16_kernel's decode
  const int in_group = GROUP_M * P.tiles_n, group = id / in_group, first_m = group * GROUP_M;
  const int gsz = min(P.tiles_m - first_m, GROUP_M);
  const int m = P.m_begin + (first_m + (id % in_group) % gsz) * BM + row, n = ((id % in_group) / gsz) * BN + ch * 8;
  if (m >= P.M || n >= P.N) return;
  const float* __restrict__ src = (const float*)args.splitk_ws + (long)r * args.sk_S * (BM * BN) + row * BN + ch * 8;
  f32x4 a0, a1;
  if (args.sk_stream > 0) {
    // stream form: the pieces of tile r are the segments of the work items whose iteration ranges meet [r nk, (r + 1) nk), summed
    // in K order = ascending item (the first starts at K = 0 and carries the bias).  The piece list is scalar work; the first
    // four pieces' loads are in flight together (a remainder of more than half a round has at most three per tile).
    const uint32_t nk0 = (uint32_t)(args.p[0].K / BK), I = (uint32_t)args.sk_rem * nk0, n = (uint32_t)args.sk_stream;
    const uint32_t u0 = (uint32_t)r * nk0, u1 = u0 + nk0;
    auto it0 = [&](uint32_t q) { return (uint32_t)((uint64_t)q * I / n); };
    uint32_t q = (uint32_t)((uint64_t)u0 * n / I);
    while (q > 0 && it0(q) > u0) --q;
    while (q + 1 < n && it0(q + 1) <= u0) ++q;
    const float* __restrict__ base = (const float*)args.splitk_ws + row * BN + ch * 8;
    const float* pc[4] = {nullptr, nullptr, nullptr, nullptr};
    int np = 0;
    uint32_t q_more = n;                // first item beyond the fourth piece (n = none)
    for (; q < n && it0(q) < u1; ++q) {
      const uint32_t b = it0(q);
      if (b == it0(q + 1)) continue;
      if (np == 4) { q_more = q; break; }
      pc[np++] = base + (long)(q * 2 + (b / nk0 == (uint32_t)r ? 0 : 1)) * (BM * BN);
    }
    f32x4 lo[4], hi[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if (i < np) { lo[i] = *(const f32x4*)pc[i]; hi[i] = *(const f32x4*)(pc[i] + 4); }
    a0 = lo[0]; a1 = hi[0];
#pragma unroll
    for (int i = 1; i < 4; ++i)
      if (i < np) { a0 += lo[i]; a1 += hi[i]; }
    for (q = q_more; q < n && it0(q) < u1; ++q) {       // (forced test geometries: many short items per tile)
      const uint32_t b = it0(q);
      if (b == it0(q + 1)) continue;
      const float* pq = base + (long)(q * 2 + (b / nk0 == (uint32_t)r ? 0 : 1)) * (BM * BN);
      a0 += *(const f32x4*)pq; a1 += *(const f32x4*)(pq + 4);
    }
  } else
  switch (args.sk_S) {        // (wave-uniform; each case is fully unrolled: all 2 S loads of a thread are in flight together)
    case 2: sum_slices<2, BM * BN>(src, a0, a1); break;
    case 3: sum_slices<3, BM * BN>(src, a0, a1); break;
    case 4: sum_slices<4, BM * BN>(src, a0, a1); break;
    case 5: sum_slices<5, BM * BN>(src, a0, a1); break;
    case 6: sum_slices<6, BM * BN>(src, a0, a1); break;
    case 7: sum_slices<7, BM * BN>(src, a0, a1); break;
    default: sum_slices<8, BM * BN>(src, a0, a1); break;
  }
  float v[8];
  u32x4 o;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    o[e] = pack2bf(e < 2 ? a0[2 * e] : a1[2 * e - 4], e < 2 ? a0[2 * e + 1] : a1[2 * e - 3]);   // t = bf16(acc + bias)
    v[2 * e] = lo_bf(o[e]); v[2 * e + 1] = hi_bf(o[e]);
  }
  const long crow = P.c_rpb > 0 ? (long)(m / P.c_rpb) * P.c_bstride + (long)(m % P.c_rpb) * P.ldc : (long)m * P.ldc;
  if (EPI == VC_EPI_GELU) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const f32x2 g = gelu_tanh2(f32x2{v[2 * e], v[2 * e + 1]});
      o[e] = pack2bf(g[0], g[1]);
    }
  } else if (EPI == VC_EPI_SILU) {
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = pack2bf(silu_f(v[2 * e]), silu_f(v[2 * e + 1]));
  } else if (EPI == VC_EPI_GATE_RES) {
    const long gate_step = args.step_ptr ? (long)(*args.step_ptr) * args.gate_step_stride : 0;
    const u32x4 gg = *(const u32x4*)((const bf16_t*)P.gate + gate_step + (long)(m / P.rows_per_batch) * P.gate_bstride + n);
    const u32x4 rr = *(const u32x4*)((const bf16_t*)P.res + (P.c_rpb > 0 ? crow : (long)m * P.ldres) + n);
#pragma unroll
    for (int e = 0; e < 4; ++e) {   // out = res + bf16(gate * t), as pass 2 of the one-pass kernel
      const f32x2 gv = f32x2{lo_bf(gg[e]), hi_bf(gg[e])} * f32x2{v[2 * e], v[2 * e + 1]};
      const uint32_t gb = pack2bf(gv[0], gv[1]);
      const f32x2 sum = f32x2{lo_bf(rr[e]), hi_bf(rr[e])} + f32x2{lo_bf(gb), hi_bf(gb)};
      o[e] = pack2bf(sum[0], sum[1]);
    }
  }
  *(u32x4*)((bf16_t*)P.C + crow + n) = o;
}

// VcGemmArgs.batch = Z > 1: Z instances of every problem, grid (tiles, Z), the 128x128 tile with the plain bias epilogue
hipError_t launch_zbatch(const VcGemmArgs& a, int total_tiles, hipStream_t s) {
  constexpr int BM = 128, BN = 128, NT = 4 * 64, LDS_STAGES = 2 * (BM + BN) * BK * 2, LDS_EPI = BM * (BN * 2 + 16);
  constexpr int LDS = LDS_STAGES > LDS_EPI ? LDS_STAGES : LDS_EPI;
  void (*fn)(const VcGemmArgs) = gemm_bf16_kernel<BM, BN, 2, 2, VC_EPI_BIAS, 0, false, false, false, true>;
  static VcOncePerDevice attr_done;
  if (attr_done.need()) {
    hipError_t e = hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    if (e != hipSuccess) return e;
    attr_done.mark();
  }
  hipLaunchKernelGGL(fn, dim3(total_tiles, a.batch), dim3(NT), LDS, s, a);
  return hipGetLastError();
}

// First launch of the remainder: sk_rem * sk_S work items of the 256x192 loader-wave kernel, each over K / sk_S, no epilogue
hipError_t launch_splitk_slices(const VcGemmArgs& a, hipStream_t s) {
  constexpr int BM = 256, BN = 192, NT = 12 * 64, LDS = (2 * BM + 3 * BN) * BK * 2;
  void (*fn)(const VcGemmArgs) = gemm_bf16_kernel<BM, BN, 4, 2, VC_EPI_BIAS, 2, false, false, true>;
  static VcOncePerDevice attr_done;
  if (attr_done.need()) {
    hipError_t e = hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    if (e != hipSuccess) return e;
    attr_done.mark();
  }
  hipLaunchKernelGGL(fn, dim3(a.sk_stream > 0 ? a.sk_stream : a.sk_rem * a.sk_S), dim3(NT), LDS, s, a);
  return hipGetLastError();
}

hipError_t launch_splitk_reduce(const VcGemmArgs& a, hipStream_t s) {
  constexpr int BM = 256, BN = 192;
  const unsigned grid = (unsigned)(a.sk_rem * (BM * (BN / 8) / 256));
  switch (a.epi) {
    case VC_EPI_BIAS: hipLaunchKernelGGL((splitk_reduce_kernel<BM, BN, VC_EPI_BIAS>), dim3(grid), dim3(256), 0, s, a); break;
    case VC_EPI_GELU: hipLaunchKernelGGL((splitk_reduce_kernel<BM, BN, VC_EPI_GELU>), dim3(grid), dim3(256), 0, s, a); break;
    case VC_EPI_SILU: hipLaunchKernelGGL((splitk_reduce_kernel<BM, BN, VC_EPI_SILU>), dim3(grid), dim3(256), 0, s, a); break;
    case VC_EPI_GATE_RES: hipLaunchKernelGGL((splitk_reduce_kernel<BM, BN, VC_EPI_GATE_RES>), dim3(grid), dim3(256), 0, s, a); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

template <int BM, int BN, int WM, int WN, int PP, bool PERSIST = false>
hipError_t launch_cfg(const VcGemmArgs& a, int total_tiles, hipStream_t s) {
  constexpr int NT = (WM * WN + (PP == 2 ? 4 : 0)) * 64;
  constexpr int LDS_STAGES = (PP == 2 ? 2 * BM + 3 * BN : 2 * (BM + BN)) * BK * 2, LDS_EPI = BM * (BN * 2 + 16);
  constexpr int LDS_EPIT = BN * (BM * 2 + 16);      // EPI_QKV stages V tiles transposed
  // ... and a head-permuted 192- / 128-wide tile as [BM] rows of 128 columns + its 64 V columns, transposed or (unaligned vt) row-major
  constexpr int LDS_EPIM = (BN == 192 || BN == 128) ? BM * (128 * 2 + 16) + (64 * (BM * 2 + 16) > BM * (64 * 2 + 16) ? 64 * (BM * 2 + 16) : BM * (64 * 2 + 16)) : 0;
  constexpr int LDS0 = LDS_STAGES > LDS_EPI ? LDS_STAGES : LDS_EPI;
  constexpr int LDS1a = LDS0 > LDS_EPIT ? LDS0 : LDS_EPIT;
  constexpr int LDS1 = LDS1a > LDS_EPIM ? LDS1a : LDS_EPIM;
  // PERSIST: W ring slots 0 and 1 sit at the top of the 160 KB, above both staging images (and above slot 2)
  static_assert(!PERSIST || (160 * 1024 - 2 * BN * BK * 2 >= LDS_EPI && 160 * 1024 - 2 * BN * BK * 2 >= LDS_EPIT && 160 * 1024 - 2 * BN * BK * 2 >= LDS_EPIM &&
                             160 * 1024 - 2 * BN * BK * 2 >= (2 * BM + BN) * BK * 2), "no room for the next tile's W(0), W(1)");
  constexpr int LDS = PERSIST ? 160 * 1024 : LDS1;
  static_assert(LDS <= 160 * 1024, "tile does not fit the 160 KB LDS of a gfx950 CU");
  void (*fn)(const VcGemmArgs) = nullptr;
  switch (a.epi) {
    case VC_EPI_QKV: fn = gemm_bf16_kernel<BM, BN, WM, WN, VC_EPI_QKV, PP, false, PERSIST>; break;
    case VC_EPI_BIAS: fn = gemm_bf16_kernel<BM, BN, WM, WN, VC_EPI_BIAS, PP, false, PERSIST>; break;
    case VC_EPI_GELU: fn = gemm_bf16_kernel<BM, BN, WM, WN, VC_EPI_GELU, PP, false, PERSIST>; break;
    case VC_EPI_GATE_RES: if constexpr (PERSIST) return hipErrorInvalidValue; else fn = gemm_bf16_kernel<BM, BN, WM, WN, VC_EPI_GATE_RES, PP>; break;
    case VC_EPI_SILU: fn = gemm_bf16_kernel<BM, BN, WM, WN, VC_EPI_SILU, PP, false, PERSIST>; break;
    default: return hipErrorInvalidValue;
  }
  static VcOncePerDevice attr_done[5];
  if (attr_done[a.epi].need()) {
    hipError_t e = hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    if (e != hipSuccess) return e;
    attr_done[a.epi].mark();
  }
  hipLaunchKernelGGL(fn, dim3(PERSIST ? std::min(total_tiles, vc_cu_count() / 8 * 8) : total_tiles), dim3(NT), LDS, s, a);
  return hipGetLastError();
}

// 3x3 convolution over an NHWC map as ONE GEMM launch: the loader waves gather the im2col rows on the fly.
// BN = 128 for O <= 128 (the 128-channel levels of the VAE and its 3-channel conv_out), else 192.
template <int BN>
hipError_t launch_conv(const VcGemmArgs& a, int total_tiles, hipStream_t s) {
  constexpr int BM = 256, NT = 12 * 64;
  constexpr int LDS_STAGES = (2 * BM + 3 * BN) * BK * 2, LDS_EPI = BM * (BN * 2 + 16);
  constexpr int LDS = LDS_STAGES > LDS_EPI ? LDS_STAGES : LDS_EPI;
  void (*fn)(const VcGemmArgs) = a.epi == VC_EPI_GATE_RES ? gemm_bf16_kernel<BM, BN, 4, 2, VC_EPI_GATE_RES, 2, true>
                                                          : gemm_bf16_kernel<BM, BN, 4, 2, VC_EPI_BIAS, 2, true>;
  static VcOncePerDevice attr_done[2];
  const int k = a.epi == VC_EPI_GATE_RES ? 1 : 0;
  if (attr_done[k].need()) {
    hipError_t e = hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    if (e != hipSuccess) return e;
    attr_done[k].mark();
  }
  hipLaunchKernelGGL(fn, dim3(total_tiles), dim3(NT), LDS, s, a);
  return hipGetLastError();
}

}  // namespace

// y[H*W, O] = conv3x3(x) + bias (+ gate * . + res): x is an NHWC map [Hs*Ws + 1, C] whose LAST row is zero (the source of
// every out-of-image tap), w is [O, 9*C] with K ordered (dy, dx, c).  mode 0: same size; 1: x is the half-resolution map
// (nearest 2x upsampling folded in); 2: x is the double-resolution map, pad (0,1,0,1) + stride 2.
int vc_conv3x3_launch(const void* x, const void* w, const void* bias, void* out, int64_t ldc, const void* res, int64_t ldres,
                      const void* gate, int H, int W, int C, int O, int mode, hipStream_t s, char* err, int errlen) {
  if (!x || !w || !out) { snprintf(err, errlen, "conv3x3: null pointer"); return VC_ERR_ARG; }
  if (H <= 0 || W <= 0 || W >= 65536 || H >= 65536 || C <= 0 || C % 64 || O <= 0 || O % 8 || ldc % 8 || ldc < O || mode < 0 || mode > 2 ||
      (mode == 1 && ((H | W) & 1)) || (res && (!gate || ldres % 8)) || (int64_t)H * W >= (1LL << 31) / 4) {
    snprintf(err, errlen, "conv3x3: bad arguments H=%d W=%d C=%d O=%d mode=%d (C %% 64 == 0, O %% 8 == 0)", H, W, C, O, mode); return VC_ERR_ARG; }
  VcGemmArgs a;
  memset(&a, 0, sizeof(a));
  a.nprob = 1;
  a.epi = res ? VC_EPI_GATE_RES : VC_EPI_BIAS;
  VcGemmProblem& p = a.p[0];
  p.A = x; p.W = w; p.bias = bias; p.C = out; p.res = res; p.gate = gate;
  p.lda = C;                                   // channels per tap = row stride of the map
  p.a_rpb = W;                                 // output width
  p.a_bstride = (int64_t)H | ((int64_t)mode << 32);
  p.ldw = 9 * (int64_t)C; p.ldc = ldc; p.ldres = ldres;
  p.M = H * W; p.N = O; p.K = 9 * C; p.rows_per_batch = p.M;
  const int bn = O <= 128 ? 128 : 192;
  p.tiles_m = (p.M + 255) / 256; p.tiles_n = (p.N + bn - 1) / bn; p.tile_start = 0;
  hipError_t e = bn == 128 ? launch_conv<128>(a, p.tiles_m * p.tiles_n, s) : launch_conv<192>(a, p.tiles_m * p.tiles_n, s);
  if (e != hipSuccess) { snprintf(err, errlen, "conv3x3 launch: %s", hipGetErrorString(e)); return VC_ERR_HIP; }
  return VC_OK;
}

namespace {

constexpr int cfg_bm[6] = {0, 128, 256, 256, 256, 256}, cfg_bn[6] = {0, 128, 128, 256, 192, 288};

// Cost model fitted on MI355X (M=3968 FLUX shapes): time = block-rounds on 256 CUs x (tile area x (K + fixed
// prologue/epilogue charge) / streaming efficiency of that tile).  Candidates: 128x128 simple loop (2 blocks per
// CU; small or skinny problems), 256x192 with loader waves (1 block per CU), which beat the 256x256 / 256x192
// ping-pong and the 256x288 tiles on every FLUX shape in an interleaved A/B (tools/gemm_ab.py; those stay
// selectable by number), and its 256x128 sibling.  For M <= 4096, 256x192 gives N=3072 / 9216 / 12288 exactly
// 1 / 3 / 4 rounds.
struct TilePlan { int tile_cfg, pp; double cost; };
constexpr int N_CAND = 3;
constexpr int cand[N_CAND] = {1, 4, 2};
constexpr int cand_pp[N_CAND] = {0, 2, 2};        // 256x128 with loaders: more blocks when M is short (L = 1664: 168 vs 112)
constexpr double cand_eff[N_CAND] = {0.55, 0.94, 0.84}, cand_ovh[N_CAND] = {500.0, 350.0, 350.0};

inline long tiles_of(const VcGemmArgs& a, int c) {
  long tiles = 0;
  for (int i = 0; i < a.nprob; ++i) {
    const int rows = a.p[i].M - a.p[i].m_begin;
    if (rows > 0) tiles += (long)((rows + cfg_bm[c] - 1) / cfg_bm[c]) * ((a.p[i].N + cfg_bn[c] - 1) / cfg_bn[c]);
  }
  return tiles;
}
TilePlan best_tile(const VcGemmArgs& a) {
  TilePlan best{0, 0, 1e300};
  for (int ci = 0; ci < N_CAND; ++ci) {
    const int c = cand[ci];
    const int per_cu = (c == 1) ? 2 : 1;
    const long n_cu = vc_cu_count();
    const long rounds = (tiles_of(a, c) + n_cu * per_cu - 1) / (n_cu * per_cu);
    const double t = rounds * (per_cu * (double)cfg_bm[c] * cfg_bn[c] * ((double)a.p[0].K + cand_ovh[ci]) / cand_eff[ci]);
    if (t < best.cost) best = TilePlan{c, cand_pp[ci], t};
  }
  return best;
}

int launch_tiles(VcGemmArgs a, int tile_cfg, int pp, bool want_persist, hipStream_t s, char* err, int errlen, int splitk_S = 0, int stream_items = 0) {
  if (tile_cfg < 1 || tile_cfg > 5 || pp > 2 || (pp == 1 && tile_cfg < 3) || (pp == 2 && tile_cfg != 4 && tile_cfg != 2)) { snprintf(err, errlen, "gemm: bad tile_cfg %d", tile_cfg); return VC_ERR_ARG; }
  const int bm = cfg_bm[tile_cfg], bn = cfg_bn[tile_cfg];
  int total = 0, np = 0;
  for (int i = 0; i < a.nprob; ++i) {       // problems left without rows by a split are dropped from the grid
    if (a.p[i].M - a.p[i].m_begin <= 0) continue;
    a.p[np] = a.p[i];
    a.p[np].tiles_m = (a.p[np].M - a.p[np].m_begin + bm - 1) / bm;
    a.p[np].tiles_n = (a.p[np].N + bn - 1) / bn;
    a.p[np].tile_start = total;
    total += a.p[np].tiles_m * a.p[np].tiles_n;
    ++np;
  }
  if (np == 0) return VC_OK;
  a.nprob = np;
  a.sk_full = total; a.sk_rem = 0; a.sk_S = 1; a.sk_stream = 0;
  if (a.batch > 1) {
    if (tile_cfg != 1 || pp != 0 || a.epi != VC_EPI_BIAS) { snprintf(err, errlen, "gemm: batch > 1 runs on the 128x128 tile with VC_EPI_BIAS"); return VC_ERR_ARG; }
    const hipError_t ez = launch_zbatch(a, total, s);
    if (ez != hipSuccess) { snprintf(err, errlen, "gemm batch launch: %s", hipGetErrorString(ez)); return VC_ERR_HIP; }
    return VC_OK;
  }
  if (splitk_S > 1) {      // the tiles beyond the last whole round of the CUs run as splitk_S K-slices each (plan_gemm decided)
    const int n_cu = vc_cu_count();
    const int full = total / n_cu * n_cu, rem = total - full;
    if (tile_cfg != 4 || pp != 2 || a.epi == VC_EPI_QKV) { snprintf(err, errlen, "gemm: split-K runs on the 256x192 loader-wave tile, not with VC_EPI_QKV"); return VC_ERR_ARG; }
    if (rem > 0) {
      if (!a.splitk_ws || a.splitk_ws_bytes < (int64_t)rem * splitk_S * bm * bn * 4) {
        snprintf(err, errlen, "gemm: split-K of %d tiles x %d slices needs %lld bytes of splitk_ws (have %lld)", rem, splitk_S,
                 (long long)rem * splitk_S * bm * bn * 4, (long long)(a.splitk_ws ? a.splitk_ws_bytes : 0)); return VC_ERR_ARG; }
      for (int i = 0; i < np; ++i)
        if (a.p[i].K / BK < splitk_S) { snprintf(err, errlen, "gemm: K=%d is too short for %d slices", a.p[i].K, splitk_S); return VC_ERR_ARG; }
      a.sk_full = full; a.sk_rem = rem; a.sk_S = splitk_S;
    }
  } else if (stream_items > 0) {      // ... or as stream_items work items over their flattened K-iterations (stream form)
    const int n_cu = vc_cu_count();
    const int full = total / n_cu * n_cu, rem = total - full;
    if (tile_cfg != 4 || pp != 2 || a.epi == VC_EPI_QKV) { snprintf(err, errlen, "gemm: split-K runs on the 256x192 loader-wave tile, not with VC_EPI_QKV"); return VC_ERR_ARG; }
    if (rem > 0) {
      if (stream_items < rem || stream_items > 4096) { snprintf(err, errlen, "gemm: %d stream items for %d remainder tiles", stream_items, rem); return VC_ERR_ARG; }
      if (!a.splitk_ws || a.splitk_ws_bytes < (int64_t)stream_items * 2 * bm * bn * 4) {
        snprintf(err, errlen, "gemm: stream split-K with %d work items needs %lld bytes of splitk_ws (have %lld)", stream_items,
                 (long long)stream_items * 2 * bm * bn * 4, (long long)(a.splitk_ws ? a.splitk_ws_bytes : 0)); return VC_ERR_ARG; }
      for (int i = 1; i < np; ++i)
        if (a.p[i].K != a.p[0].K) { snprintf(err, errlen, "gemm: stream split-K needs one K for all problems"); return VC_ERR_ARG; }
      a.sk_full = full; a.sk_rem = rem; a.sk_stream = stream_items;
      a.sk_S = 2;                     // (marks "remainder split" for the code below; the stream form reads sk_stream, not sk_S)
    }
  }
  // VC_GEMM_PERSIST + more tiles than CUs on the loader-wave schedule: one persistent workgroup per CU walks them
  // (gemm_bf16_kernel PERSIST).  OPT-IN: bit-identical, and worth +0.05 % steps/s at cfg 2 (-0.4 ... -0.7 % per qkv / MLP-up
  // launch, profiles/r03d_*), +-0.2 % at the other geometries (profiles/r03e_persist_ab.log) - under the board's power limit
  // the dispatch gap and the first-tile latency a fresh workgroup pays are idle, low-power time that the clock gives back.
  // (Not the gated-residual epilogue: its residual prefetch registers + the loop state do not fit the 168-VGPR budget of the
  // 3-waves-per-SIMD kernel - 39 spilled registers.)
  const int n_cu8 = vc_cu_count() / 8 * 8;
  const bool persist = pp == 2 && want_persist && n_cu8 >= 8 && total > n_cu8 && a.epi != VC_EPI_GATE_RES && a.sk_S == 1;
  hipError_t e = hipSuccess;
  if (a.sk_S > 1) total = a.sk_full;        // the whole rounds run as ever (ids [0, sk_full)); the remainder follows below
  if (total > 0)
  switch (tile_cfg) {
    case 1: e = launch_cfg<128, 128, 2, 2, 0>(a, total, s); break;
    case 2: e = pp == 2 ? (persist ? launch_cfg<256, 128, 4, 2, 2, true>(a, total, s) : launch_cfg<256, 128, 4, 2, 2>(a, total, s))
                        : launch_cfg<256, 128, 4, 2, 0>(a, total, s); break;
    case 3: e = pp ? launch_cfg<256, 256, 2, 4, 1>(a, total, s) : launch_cfg<256, 256, 2, 4, 0>(a, total, s); break;
    case 4: e = pp == 2 ? (persist ? launch_cfg<256, 192, 4, 2, 2, true>(a, total, s) : launch_cfg<256, 192, 4, 2, 2>(a, total, s))
                        : pp ? launch_cfg<256, 192, 4, 2, 1>(a, total, s) : launch_cfg<256, 192, 4, 2, 0>(a, total, s); break;
    default: e = pp ? launch_cfg<256, 288, 4, 2, 1>(a, total, s) : launch_cfg<256, 288, 4, 2, 0>(a, total, s); break;
  }
  if (e != hipSuccess) { snprintf(err, errlen, "gemm launch: %s", hipGetErrorString(e)); return VC_ERR_HIP; }
  if (a.sk_S > 1) {
    e = launch_splitk_slices(a, s);
    if (e == hipSuccess) e = launch_splitk_reduce(a, s);
    if (e != hipSuccess) { snprintf(err, errlen, "gemm split-K launch: %s", hipGetErrorString(e)); return VC_ERR_HIP; }
  }
  return VC_OK;
}

}  // namespace

static int validate_gemm(VcGemmArgs& a, char* err, int errlen) {
  if (a.nprob < 1 || a.nprob > VC_GEMM_MAX_PROBLEMS) { snprintf(err, errlen, "gemm: nprob must be 1..%d", VC_GEMM_MAX_PROBLEMS); return VC_ERR_ARG; }
  for (int i = 0; i < a.nprob; ++i) {
    VcGemmProblem& p = a.p[i];
    p.m_begin = 0;
    if (p.M <= 0 || p.N <= 0 || p.K <= 0) { snprintf(err, errlen, "gemm: empty problem %d (M=%d N=%d K=%d)", i, p.M, p.N, p.K); return VC_ERR_ARG; }
    if (p.K % BK) { snprintf(err, errlen, "gemm: K=%d must be a multiple of %d", p.K, BK); return VC_ERR_ARG; }
    if (p.N % 8 || p.ldc % 8 || p.lda % 8) { snprintf(err, errlen, "gemm: need N, ldc, lda multiples of 8 (N=%d ldc=%ld lda=%ld)", p.N, (long)p.ldc, (long)p.lda); return VC_ERR_ARG; }
    if (!p.A || !p.W || !p.C) { snprintf(err, errlen, "gemm: null operand"); return VC_ERR_ARG; }
    if (p.a_rpb < 0 || p.c_rpb < 0 || p.a_bstride % 8 || p.c_bstride % 8 || (p.c_rpb > 0 && p.res && p.ldres != p.ldc)) {
      snprintf(err, errlen, "gemm: bad batch-strided row description"); return VC_ERR_ARG; }
    if ((p.a_rpb > 0 ? (uint64_t)((p.M + p.a_rpb - 1) / p.a_rpb) * (uint64_t)p.a_bstride : 0) >= (1ull << 32) ||
        (uint64_t)(p.a_rpb > 0 ? p.a_rpb : p.M) * (uint64_t)p.lda >= (1ull << 32) || (uint64_t)p.N * (uint64_t)p.ldw >= (1ull << 32) || (p.ldw != 0 && p.ldw < p.K) || p.ldw % 8) {
      snprintf(err, errlen, "gemm: operand exceeds 32-bit element offsets"); return VC_ERR_ARG; }
    // the loader-wave kernels address A with 32-bit BYTE offsets against the operand, W with byte offsets against its n-tile
    // (the largest element offset of A the kernel forms, + one K-tile: batch-strided rows may overlap or leave gaps, so both
    // terms count - advisor r04)
    const uint64_t a_last = p.a_rpb > 0 ? (uint64_t)((p.M - 1) / p.a_rpb) * (uint64_t)p.a_bstride + (uint64_t)(p.a_rpb - 1) * (uint64_t)p.lda
                                        : (uint64_t)(p.M - 1) * (uint64_t)p.lda;
    if (a_last + (uint64_t)p.K + 64 >= (1ull << 31) || (uint64_t)288 * (uint64_t)p.ldw >= (1ull << 31)) {
      snprintf(err, errlen, "gemm: A operand spanning 4 GB or more (2^31 bf16 elements; or a W row stride beyond 7 M elements) is not supported"); return VC_ERR_ARG; }
    if (a.epi == VC_EPI_GATE_RES && (!p.res || !p.gate || p.rows_per_batch <= 0 || p.ldres % 8 || p.gate_bstride % 8 || a.gate_step_stride % 8)) {
      snprintf(err, errlen, "gemm: gate/residual epilogue needs res, gate, rows_per_batch"); return VC_ERR_ARG; }
    if (a.epi == VC_EPI_QKV && p.kn_heads != 0 && (p.kn_heads < 0 || p.N != 384 * p.kn_heads || (p.vt && p.vt_col0 != 256 * p.kn_heads) || p.vt_rpb <= 0 ||
                                                  ((p.kn_scale || p.qn_scale) && (!p.kn_rope || p.vt_row0 < 0 || p.kn_rope_bstride < 0)))) {
      snprintf(err, errlen, "gemm: head-permuted qkv needs N = 3 * 128 * kn_heads (N=%d kn_heads=%d), vt_col0 = 2 * 128 * kn_heads, the row geometry "
                            "vt_rpb / vt_row0, and with kn_scale / qn_scale a rope table", p.N, p.kn_heads); return VC_ERR_ARG; }
    if ((a.epi != VC_EPI_QKV || p.kn_heads == 0) && (p.kn_heads != 0 || p.kn_scale || p.qn_scale || p.qn_prescale)) {
      snprintf(err, errlen, "gemm: kn_heads / kn_scale / qn_scale belong to VC_EPI_QKV with head-permuted weights"); return VC_ERR_ARG; }
    if (p.qn_prescale && !p.qn_scale) { snprintf(err, errlen, "gemm: qn_prescale without qn_scale"); return VC_ERR_ARG; }
    if (a.epi == VC_EPI_QKV && p.vt && (p.vt_rpb <= 0 || p.vt_col0 < 0 || p.vt_col0 % 8 || p.vt_col0 >= p.N || p.vt_row0 < 0 ||
                                        p.vt_lpad < p.vt_row0 + p.vt_rpb || p.vt_bstride < (int64_t)(p.N - p.vt_col0) * p.vt_lpad)) {
      snprintf(err, errlen, "gemm: bad V^T description (vt_col0=%d vt_rpb=%d vt_row0=%d vt_lpad=%d vt_bstride=%ld)", p.vt_col0, p.vt_rpb,
               p.vt_row0, p.vt_lpad, (long)p.vt_bstride); return VC_ERR_ARG; }
  }
  if (a.epi < 0 || a.epi > VC_EPI_QKV) { snprintf(err, errlen, "gemm: unknown epilogue %d", a.epi); return VC_ERR_ARG; }
  if (a.batch < 0 || a.batch > 65535) { snprintf(err, errlen, "gemm: batch must be 0..65535"); return VC_ERR_ARG; }
  if (a.batch > 1) {
    if (a.epi != VC_EPI_BIAS) { snprintf(err, errlen, "gemm: batch > 1 supports VC_EPI_BIAS only"); return VC_ERR_ARG; }
    for (int i = 0; i < a.nprob; ++i) {
      const VcGemmProblem& p = a.p[i];
      if (p.a_zstride % 8 || p.w_zstride % 8 || p.c_zstride % 8 || p.a_zstride < 0 || p.w_zstride < 0 || p.c_zstride < 0 || p.a_rpb || p.c_rpb ||
          (uint64_t)a.batch * (uint64_t)p.a_zstride >= (1ull << 40) || (uint64_t)a.batch * (uint64_t)p.w_zstride >= (1ull << 40)) {
        snprintf(err, errlen, "gemm: batch strides must be non-negative multiples of 8 elements (plain rows only)"); return VC_ERR_ARG; }
    }
  }
  return VC_OK;
}

// The launch plan of one vc_gemm call: cut = first row of the second launch (0 = one launch); tile / pp of the two launches.
struct GemmPlan { int cut, tile1, pp1, tile2, pp2, sk_S = 0, sk_tiles = 0, sk_stream = 0; };
static GemmPlan plan_gemm(const VcGemmArgs& a, int tile_cfg) {
  const int force_cut = (tile_cfg >> 8) & 255;         // tests: cut problem 0 at row force_cut * 256
  const int force_sk = (tile_cfg >> 16) & 15;          // tests / A-B: VC_GEMM_SPLITK(S)
  const bool no_split = (tile_cfg & VC_GEMM_NO_SPLIT) != 0, no_splitk = (tile_cfg & VC_GEMM_NO_SPLITK) != 0;
  const int tile_cfg_flags = tile_cfg;
  tile_cfg &= 63;
  if (a.batch > 1) return GemmPlan{0, 1, 0, 0, 0};          // Z instances per problem: the 128x128 tile (grid (tiles, Z))
  const long n_cus = vc_cu_count();
  auto sk_plan = [&](int S) {
    const long total = tiles_of(a, 4), rem = total % n_cus;
    GemmPlan pl{0, 4, 2, 0, 0};
    if (rem > 0) { pl.sk_S = S; pl.sk_tiles = (int)rem; }
    return pl;
  };
  // stream form: n work items share the remainder's K-iterations evenly (each >= ~12 iterations, at most one per CU)
  auto stream_plan = [&]() {
    const long total = tiles_of(a, 4), rem = total % n_cus;
    GemmPlan pl{0, 4, 2, 0, 0};
    if (rem > 0) {
      long n = rem * (a.p[0].K / BK) / 12;
      n = n < rem ? rem : n > n_cus ? n_cus : n;
      pl.sk_stream = (int)n; pl.sk_tiles = (int)rem;
    }
    return pl;
  };
  if (tile_cfg_flags & VC_GEMM_STREAMK) return stream_plan();
  if (force_sk >= 2) return sk_plan(force_sk > 8 ? 8 : force_sk);
  for (int i = 0; i < a.nprob; ++i)      // heads are normalised inside the epilogue: every head must lie in one 192-wide tile
    if (a.epi == VC_EPI_QKV && (a.p[i].kn_scale || a.p[i].qn_scale)) return GemmPlan{0, 4, tile_cfg != 0 && ((tile_cfg >> 4) & 3) != 2 ? (tile_cfg >> 4) & 3 : 2, 0, 0};
  if (tile_cfg != 0) return GemmPlan{0, tile_cfg & 15, (tile_cfg >> 4) & 3, 0, 0};
  const TilePlan whole = best_tile(a);
  // SPLIT-K REMAINDER: the 256x192 tiles are R whole rounds of the CUs plus r tiles - run those r as r * S slices of K / S
  // (S <= 8, r * S <= CUs: ONE short round) that leave f32 partial tiles for a small second launch, instead of a second round
  // at r / CUs fill or a narrower tile for everything.  Priced in the units of best_tile (one 256x192 tile of K = 15360 on its CU
  // = 8.2e8 units = 255 us: 3.2e6 units per us): a slice pays its own prologue and the partial store instead of an epilogue
  // (+150), the partials are written and read once at ~4 TB/s, the second launch costs a dependent kernel boundary (~3 us).
  // Taken at >= 7 % under the best one-launch plan: SDEdit stage (L = 4608: 288 tiles = 256 + 32 x 8 slices, K = 12288 /
  // 15360) and cfg 1 (L = 1664: 112 tiles x 2 slices); never at K = 3072 (the partial traffic outweighs 1 / S of a short tile).
  GemmPlan sk{0, 0, 0, 0, 0};
  double sk_cost = 1e300;
  if (!no_splitk && a.splitk_ws && a.epi != VC_EPI_QKV) {
    const long total = tiles_of(a, 4), R = total / n_cus, rem = total % n_cus;
    const int nk = a.p[0].K / BK;
    int S = rem > 0 ? (int)(n_cus / rem) : 0;
    if (S > 8) S = 8;
    if (S > nk / 8) S = nk / 8;
    bool same_k = true;
    for (int i = 1; i < a.nprob; ++i) same_k = same_k && a.p[i].K == a.p[0].K;
    const double bytes = (double)rem * S * cfg_bm[4] * cfg_bn[4] * 4;
    if (S >= 2 && same_k && bytes <= (double)a.splitk_ws_bytes) {
      const double area = (double)cfg_bm[4] * cfg_bn[4] / cand_eff[1];
      const double cost = R * area * (a.p[0].K + cand_ovh[1]) + area * ((double)a.p[0].K / S + cand_ovh[1] + 150.0) + (2.0 * bytes / 4e6 + 3.0) * 3.2e6;
      if (cost < 0.93 * whole.cost) { sk = sk_plan(S); sk_cost = cost; }
    }
  }
  // STREAM REMAINDER: more than half a round of tiles beyond the whole rounds (no uniform S >= 2 fits one round): every CU takes
  // f = rem / CUs of a tile's K-iterations, at most two segments, <= 3 partial tiles per remainder tile.  Decided by measurement,
  // not by the model (which prices a partly filled round at its full length; under the power cap it costs ~0.85 of one at 81 %
  // fill): interleaved whole steps, profiles/r05d_ab_*.log - cfg 3's N = 3072 launches (416 tiles = 256 + 160, f = 0.625: instead
  // of the row cut into 256 + 240 narrower tiles) +1.7 % per step, all of it from K >= 12288 (K = 3072 included: +0.0 %);
  // cfg 5's (464 = 256 + 208, f = 0.81: instead of a second round at 81 % fill) -1.2 %, with K = 3072 -2.1 %.  Taken for
  // 0.5 < f <= 0.7 and K >= VC_GEMM_STREAMK_MIN_K.
  if (!no_splitk && sk.sk_S == 0 && a.splitk_ws && a.epi != VC_EPI_QKV) {
    const long total = tiles_of(a, 4), R = total / n_cus, rem = total % n_cus;
    bool same_k = true;
    for (int i = 1; i < a.nprob; ++i) same_k = same_k && a.p[i].K == a.p[0].K;
    const bool prefer = (tile_cfg_flags & VC_GEMM_PREFER_STREAMK) != 0, any_k = (tile_cfg_flags & VC_GEMM_STREAMK_ANY_K) != 0;
    if (R >= 1 && 2 * rem > n_cus && same_k && (a.p[0].K >= VC_GEMM_STREAMK_MIN_K || any_k) && (double)n_cus * 2 * cfg_bm[4] * cfg_bn[4] * 4 <= (double)a.splitk_ws_bytes) {
      // (advisor r05: only where the one-launch plan would have chosen the 256x192 tile itself - at N = 256 or 4096 a 192-wide
      // tile wastes columns and another tile may cost far less than any remainder scheme on this one)
      if (prefer || any_k || (10 * rem <= 7 * n_cus && whole.tile_cfg == 4)) return stream_plan();
    }
  }
  // Block-round quantisation: cut problem 0's rows where the 256x192 tiles above the cut are (nearly) whole rounds of the 256
  // CUs and price the remainder with the tile that suits it.  The two launches follow each other on the stream (the first
  // has a flat tail by construction); a cut is taken when the model says it saves >= 10 % and both launches fill their rounds.
  // The model over-credits by an order of magnitude: under the board's power limit a partly filled round runs at a higher
  // clock, so quantisation costs far less than its fill factor.  Interleaved A/B, steps/s with / without cuts: L = 6656 (the
  // N = 3072 launches: 416 tiles -> 256 + 240, model -12.7 % per launch) 9.846 / 9.804 = +0.4 %; L = 7424 (N = 12288 launches
  // cut at 6144 rows, model -6.3 %) 8.683 / 8.693 = -0.1 % - hence the 10 % bar.
  int cut = 0;
  TilePlan rest_plan{0, 0, 0};
  double best_cut = 1e300;
  if (!no_split || force_cut > 0) {
    double best = force_cut > 0 ? 1e300 : 0.90 * whole.cost;
    const int tn = (a.p[0].N + cfg_bn[4] - 1) / cfg_bn[4];
    const int mt_all = (a.p[0].M + 255) / 256;
    for (int mt = 1; mt <= mt_all; ++mt) {
      if (force_cut > 0 && mt != force_cut) continue;
      const int rows = mt * 256 < a.p[0].M ? mt * 256 : a.p[0].M;
      if (rows == a.p[0].M && a.nprob == 1) break;            // nothing left for the second launch
      const long n_cu = vc_cu_count();
      const long tiles1 = (long)mt * tn, rounds1 = (tiles1 + n_cu - 1) / n_cu;
      if (force_cut == 0 && tiles1 < 0.97 * (double)n_cu * rounds1) continue;
      const double t1 = rounds1 * ((double)cfg_bm[4] * cfg_bn[4] * ((double)a.p[0].K + cand_ovh[1]) / cand_eff[1]);
      VcGemmArgs rest = a;
      rest.p[0].m_begin = rows;
      const TilePlan rp = best_tile(rest);
      // the remainder must fill its own rounds too: a half-empty second launch loses more than the model credits it with
      // (measured: L = 4608 cut into 4096 + 512 rows, 144 - 192 tiles in the second launch: -0.7 % steps/s)
      const int per_cu = rp.tile_cfg == 1 ? 2 : 1;
      const long tiles2 = tiles_of(rest, rp.tile_cfg), slots2 = (tiles2 + n_cu * per_cu - 1) / (n_cu * per_cu) * n_cu * per_cu;
      if (force_cut == 0 && tiles2 < 0.9 * slots2) continue;
      if (t1 + rp.cost < best) { best = t1 + rp.cost; cut = rows; rest_plan = rp; best_cut = best; }
    }
  }
  if (sk.sk_S > 1 && (cut == 0 || sk_cost <= best_cut)) return sk;
  if (cut == 0) return GemmPlan{0, whole.tile_cfg, whole.pp, 0, 0};
  return GemmPlan{cut, 4, 2, rest_plan.tile_cfg, rest_plan.pp};
}

// tile_cfg: 0 = auto, 1 = 128x128 (4 waves), 2 = 256x128, 3 = 256x256, 4 = 256x192, 5 = 256x288 (8 waves each);
// +16 = ping-pong main loop (3, 4, 5), +32 = ping-pong with loader waves (2, 4); VC_GEMM_NO_SPLIT / (k << 8): see the header
int vc_gemm_launch(VcGemmArgs a, int tile_cfg, hipStream_t s, char* err, int errlen) {
  int rc = validate_gemm(a, err, errlen);
  if (rc != VC_OK) return rc;
  const GemmPlan pl = plan_gemm(a, tile_cfg);
  const bool want_persist = (tile_cfg & VC_GEMM_PERSIST) != 0;
  if (pl.cut == 0) return launch_tiles(a, pl.tile1, pl.pp1, want_persist, s, err, errlen, pl.sk_S, pl.sk_stream);
  VcGemmArgs first = a;
  first.nprob = 1;
  first.p[0].M = pl.cut;                                       // rows [0, cut) of problem 0 on the 256x192 loader-wave tile
  rc = launch_tiles(first, pl.tile1, pl.pp1, want_persist, s, err, errlen);
  if (rc != VC_OK) return rc;
  a.p[0].m_begin = pl.cut;
  return launch_tiles(a, pl.tile2, pl.pp2, want_persist, s, err, errlen);
}

// the plan without the launch: out = {cut row, tile / loader mode of the first (or only) launch, of the second, tiles of both}
int vc_gemm_plan_impl(VcGemmArgs a, int tile_cfg, int32_t out[8], char* err, int errlen) {
  const int rc = validate_gemm(a, err, errlen);
  if (rc != VC_OK) return rc;
  const GemmPlan pl = plan_gemm(a, tile_cfg);
  if (pl.tile1 < 1 || pl.tile1 > 5) { snprintf(err, errlen, "gemm: bad tile_cfg %d", pl.tile1); return VC_ERR_ARG; }
  out[0] = pl.cut; out[1] = pl.tile1; out[2] = pl.pp1; out[3] = pl.tile2; out[4] = pl.pp2;
  VcGemmArgs first = a, rest = a;
  if (pl.cut) { first.nprob = 1; first.p[0].M = pl.cut; rest.p[0].m_begin = pl.cut; }
  out[5] = (int32_t)(tiles_of(first, pl.tile1) + (pl.cut ? tiles_of(rest, pl.tile2) : 0));
  out[6] = pl.sk_stream > 0 ? -pl.sk_stream : pl.sk_S; out[7] = pl.sk_tiles;
  return VC_OK;
}
