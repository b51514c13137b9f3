// Joint text+image flash attention for gfx950, ONE WAVE PER SIMD form (vc_attention variants 8 / 12).
//
// Same arithmetic and the same LDS images as attention.hip (models/math.py:63-99: softmax(q k^T / sqrt(128)) v over the
// stitched grid sequence, f32 online softmax in the log2 domain, bf16 P, padded keys masked, padded query rows zero),
// restructured around what bounds that kernel: at 32 queries per wave and two waves per SIMD its loop is issue-bound
// (49 % matrix-pipe busy; every K / V^T fragment read and every LDS-DMA piece serves only 32 queries, and the two
// waves of a SIMD contend for one VALU port).  Here a workgroup is 4 waves = one per SIMD, each wave owns 64 queries
// (two 32-query blocks) and the WHOLE 512-entry register file:
//
//   AGPR a[0:127]    O^T accumulators, 2 query blocks x 4 d-tiles x 16          (written only by the P.V MFMAs)
//   AGPR a[128:191]  Q fragments (MFMA B operands), 2 x 8 k-steps x 4           (loaded once per work item)
//   AGPR a[192:255]  K fragments of the NEXT tile (MFMA A operands), 2 x 8 x 4  (ds_read straight into AGPRs)
//   VGPR             S^T of two tiles (2 x 64), P (32), V^T fragment ring (32), softmax state, addresses
//
// Every K / V^T fragment feeds two MFMAs (both query blocks): half the LDS reads and half the LDS-DMA issues per FLOP.
// The tile loop is software-pipelined inside the wave, two phases of 32 MFMAs per 64-key tile:
//
//   phase A(t):  S(t+1) = K(t+1) . Q^T - m      ||  P(t) = 2^(S(t)), row sums, bf16 pack; V^T(t) fragments 0..7
//   phase B(t):  O += V^T(t) . P(t)^T            ||  row max of S(t+1), rescale decision; K(t+2) fragments -> AGPRs;
//                                                    V^T(t) fragments 8..15; LDS-DMA
//   one s_barrier per tile.
//
// No per-element scale / subtract in the loop: Q is multiplied by 128^-0.5 * log2(e) when it is loaded (once per work
// item), and the running row max m is subtracted BY THE MATRIX PIPE - every S chain runs a 9th k-step over two extra
// "dimensions", q_aug = (-m, -29952), k_aug = (1, key is masked ? 1 : 0), with m kept bf16-representable (any reference
// point works for a softmax), so the accumulators come out as c*q.k - m, masked keys at -29952, ready for v_exp.  With
// the deferred rescale (m moves only when a row grows by more than 2^8) the per-element VALU work is exp, add and half
// a convert and half a max3 - what fits beside 68 MFMAs per tile; the measured prices per 32-cycle MFMA gap (one
// wave per SIMD, tools/ubench/a64_gap.hip) are 7 issue slots, MFMA 1, plain VALU 1, v_exp 2, ds_read_b128 3.2.
//
// S^T lives in SIX 16-register blocks: phase A runs its four chains one after the other, and the blocks of S(t) whose
// probabilities are already exponentiated take the last two chains of S(t+1); the block roles rotate with period 3,
// as do the 3-deep K and V^T LDS rings, so the loop body is three tiles.
//
// The MFMAs, LDS reads and waits are inline asm with literal AGPR names (hipcc cannot be told which accumulators live
// in which half of the register file: with builtins it parks S in AGPRs and moves it through v_accvgpr_read by the
// hundred per tile - DESIGN.md 3.2); the softmax VALU code between them is ordinary C++ on compiler-allocated
// VGPRs, and the issue order is pinned slot by slot with sched_barrier(0) (<= 5-6 fillers per MFMA gap:
// MI355X_MICROARCH.md 'one wave per SIMD').  K and V^T tiles stream through 3-deep LDS rings by LDS-DMA; tile t issues
// V^T(t+2) and K(t+4) and waits with a COUNTED vmcnt(8), so every piece has a full tile of flight time.
//
// Audit after every change (Makefile target `audit64`): no spills, no scratch, no compiler-generated v_accvgpr_*.
#include <algorithm>
#include <type_traits>
#include "common.h"
#include "vcloze_internal.h"
#ifdef VC_A64_RING4
#include "attention64_sched.h"       // A/B builds: a ring of four V^T fragment registers (16 VGPRs less, -0.2 % per step: r06e)
#else
#include "attention64_sched8.h"      // a ring of eight (two 16-key steps)
#endif

namespace {

struct Attn64Args {
  const bf16_t* qkv;
  const bf16_t* vt;
  bf16_t* out;
  const int32_t* kv_len;
  const int32_t* kv_gap;      // optional [B][2]: keys / query rows lo <= i < hi masked too
  int64_t ld, bstride, ldo, out_bstride;
  int32_t B, L, Lpad, H, qblocks, items;
  int32_t full_rounds, tail_items, tail_units;   // full_rounds >= 0: tail split on (the schedule itself is derived per XCD, see Sched64)
  float* part;
  // optional in-kernel QKNorm + RoPE of the query rows (q_scale != nullptr): as vc_qknorm_rope_vt
  const bf16_t* q_scale; const bf16_t* q_scale2; const float* rope; int64_t rope_bstride; int32_t split;
  int32_t q_pre;        // the q columns hold normalised, rotated queries times 128^-0.5 * log2(e) (VcAttention.q_prescaled)
  int32_t inmerge;      // stream form only: tail pieces FIRST, combined at the end of the same launch (flags below), no merge kernel
  uint32_t* flags;      // [2 pieces per workgroup][2 query blocks]: 1 = piece complete and visible; zero before and after a launch
  uint64_t* debug_ts;   // profiling builds only (-DVC_ATTN_TIMESTAMPS): 32 words per workgroup (start, end, tiles, items; 8 per work item)
};
// BOUNDED (VcAttention.logit_bound): the caller guarantees |c q.k| <= bound (log2 domain) for every query / key pair - with
// QK-normed operands |q|, |k| <= sqrt(128) max|scale|, so the bound is a property of the model's norm scales.  A softmax
// needs its running max only to keep 2^x in range; with bounded logits the reference point stays 0 for the whole row:
// no row max, no rescale decision, no (-m) k-step (4 of 68 MFMAs per tile) - P = 2^(c q.k) directly, l and O accumulate
// in f32 (|x| <= 100: 2^100 * L fits), the result O / l is the same function.  Masked keys still take the extra k-step,
// in the tiles that hold any.

constexpr int KVB = 64;
constexpr int K_TILE = KVB * 256, V_TILE = 128 * KVB * 2;
constexpr int RING = 3;
constexpr int V_RING0 = RING * K_TILE;              // 48 KB of K ring, then 48 KB of V^T ring
constexpr int LDS64 = RING * (K_TILE + V_TILE);     // 96 KB
constexpr int QW = 64;                              // queries per wave
constexpr int QB = 4 * QW;                          // queries per work item

constexpr int A_O = 0, A_Q = 128, A_K = 192;        // AGPR map

// one partial result of the tail split: O^T fragments NORMALISED by the piece's own row sums, as f16 (11-bit mantissa:
// 8x finer than the bf16 output; values are convex combinations of V) [wave 4][qb 2][16 groups][lane 64][4 x f16],
// then [wave 4][qb 2][lane 64] (m, l) in f32.  Half the bytes of f32 accumulators: the pieces are written once and read
// once through the Infinity Cache, 17 MB each way at cfg 2.
constexpr int PART64_O_BYTES = 4 * 2 * 16 * 64 * 8;
constexpr int PART64_BYTES = PART64_O_BYTES + 4 * 2 * 64 * 8;
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
VC_DEV int chunk_begin64(int c, int units, int chunks) { return (int)(((long)c * units) / chunks); }
// one piece folded into the running combination of a tail item: acc = acc * keep + w * O_p with w = l_p 2^(m_p - m), keep =
// 2^(m_old - m).  ONE definition with explicit fused multiply-adds for the merge kernel and for the in-launch combine of the
// stream form: left to the optimiser, the two loops contract differently and the two routes differ in the last bit.
VC_DEV void merge_fold64(float (&acc)[16][4], float& wsum, float& m, const f16x4 (&v)[16], const f32x2 ml) {
  const float m_new = fmaxf(m, ml[0]);
  const float keep = __builtin_amdgcn_exp2f(m - m_new);          // 0 on the first piece (m = -inf), 1 while the maximum stands
  const float w = ml[1] * __builtin_amdgcn_exp2f(ml[0] - m_new);
  m = m_new;
  wsum = __builtin_fmaf(wsum, keep, w);
#pragma unroll
  for (int i = 0; i < 16; ++i)
#pragma unroll
    for (int e = 0; e < 4; ++e) acc[i][e] = __builtin_fmaf(w, (float)v[i][e], acc[i][e] * keep);
}

// Work schedule of the persistent grid, PER XCD (grid % 8 == 0; block b runs on XCD b % 8 - observed placement, used for
// speed only).  XCD x owns the contiguous logical items [start, start + n) that xcd_remap gives it: all query blocks of a
// head are neighbours there, so the K / V^T tiles of a head stream through ONE L2.  Its W = grid / 8 workgroups take
// `rounds` whole items each (item start + r * W + slot); the remaining `tail` items are cut along the keys into W equal
// chunks of (item, KV tile) units - inside the SAME XCD, so that the tail round re-reads K / V^T from the L2 that already
// holds them (round 2 cut the tail across the whole grid: every XCD streamed every tail head, 204 MB fetched per launch
// for 73 MB of operands).
struct Sched64 {
  int W, start, n, rounds, tail, units;
};
VC_DEV Sched64 sched64(int x, int G, int items, int nkt) {
  Sched64 s;
  s.W = G >> 3;
  const int q = items >> 3, r = items & 7;
  s.n = q + (x < r ? 1 : 0);
  s.start = x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q;
  s.rounds = s.n / s.W;
  s.tail = s.n - s.rounds * s.W;
  s.units = s.tail * nkt;
  return s;
}

VC_DEV int swap23(int i) { return (i & ~12) | ((i & 4) << 1) | ((i & 8) >> 1); }

template <int I, int N, class F>
VC_DEV void sfor(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    sfor<I + 1, N>(f);
  }
}
#define SB() __builtin_amdgcn_sched_barrier(0)
template <int PH, int I>
constexpr a64s::Tok tok_at() {      // token I of phase PH (0 = S = K.Q^T phase, 1 = P.V phase) of the generated filler schedule
  if constexpr (PH == 0) return a64s::A_TOK[I];
  else return a64s::B_TOK[I];
}

// ---- asm-owned instructions ----
template <int KA, int QA, bool ZERO>
VC_DEV void mfma_qk(f32x16& s) {      // S^T[u] (+)= K_frag . Q_frag   (A = K rows, B = Q rows: a lane owns ONE query column)
  if constexpr (ZERO)
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, a[%c1:%c2], a[%c3:%c4], 0" : "=v"(s) : "n"(KA), "n"(KA + 3), "n"(QA), "n"(QA + 3));
  else
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, a[%c1:%c2], a[%c3:%c4], %0" : "+v"(s) : "n"(KA), "n"(KA + 3), "n"(QA), "n"(QA + 3));
}
VC_DEV void mfma_aug(f32x16& s, const u32x4& k, const u32x4& q) {   // the 9th k-step: S^T += k_aug . q_aug  (= -m, or -29952 - m on masked keys)
  asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(s) : "v"(k), "v"(q));
}
template <int OA>
VC_DEV void mfma_pv(const u32x4& v, const u32x4& p) {   // O^T[dt] += Vt_frag . P_frag
  asm volatile("v_mfma_f32_32x32x16_bf16 a[%c2:%c3], %0, %1, a[%c2:%c3]" ::"v"(v), "v"(p), "n"(OA), "n"(OA + 15));
}
template <int A0, int OFF>
VC_DEV void lds_k(uint32_t addr) {    // K fragment -> AGPRs
  asm volatile("ds_read_b128 a[%c1:%c2], %0 offset:%c3" ::"v"(addr), "n"(A0), "n"(A0 + 3), "n"(OFF));
}
template <int OFF>
VC_DEV void lds_v(u32x4& f, uint32_t addr) {
  asm volatile("ds_read_b128 %0, %1 offset:%c2" : "=v"(f) : "v"(addr), "n"(OFF));
}
template <int A0, int OFF>
VC_DEV void load_q(const bf16_t* p) {   // 16 B of a query row -> AGPRs (waited for with vmcnt)
  asm volatile("global_load_dwordx4 a[%c1:%c2], %0, off offset:%c3" ::"v"(p), "n"(A0), "n"(A0 + 3), "n"(OFF) : "memory");
}
template <int N> VC_DEV void wait_lgkm() { asm volatile("s_waitcnt lgkmcnt(%c0)" ::"n"(N)); }
template <int N> VC_DEV void wait_vm() { asm volatile("s_waitcnt vmcnt(%c0)" ::"n"(N)); }
VC_DEV float v_max3(float a, float b, float c) {
  float r;
  asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}
VC_DEV float v_max(float a, float b) {      // (fmaxf would canonicalise both MFMA-written inputs first: two extra v_max per call)
  float r;
  asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
#define PIN(x) asm volatile("" : "+v"(x))     /* the value is materialised HERE: IR-level sinking cannot move its producers later */
VC_DEV uint32_t v_cvt_pk(float lo, float hi) {
  uint32_t r;
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
  return r;
}
// one LDS-DMA piece, `global_load_lds_dwordx4 v_off, s[base]`, with M0 = lds_wave + imm written in the same statement (one wait
// state before the DMA reads it)
VC_DEV void glds16_m0(const char* sbase, uint32_t voff, uint32_t lds_wave, int imm) {
  asm volatile("s_add_u32 m0, %2, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase), "s"(lds_wave), "i"(imm) : "memory", "m0", "scc");
}
template <int A> VC_DEV float agpr_read() {
  float r;
  asm volatile("v_accvgpr_read_b32 %0, a[%c1]" : "=v"(r) : "n"(A));
  return r;
}
template <int A> VC_DEV void agpr_write(float v) { asm volatile("v_accvgpr_write_b32 a[%c1], %0" ::"v"(v), "n"(A)); }
// Exchange with the other half-wave (lanes l and l ^ 32).  v_permlane32_swap swaps the upper half of its first operand
// with the lower half of its second: two copies of x become [x_lo, x_lo] and [x_hi, x_hi].  In asm, because hipcc
// (ROCm 7.2) folds the two results of the builtin into ONE register when both inputs are the same value; the s_nop is
// the VALU-write -> permlane-read hazard (2 wait states), which nothing pads inside an asm string.
VC_DEV void half_swap(float x, float& lo, float& hi) {
  asm volatile("v_mov_b32 %1, %2\n\tv_mov_b32 %0, %2\n\ts_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "=&v"(lo), "=&v"(hi) : "v"(x));
}
VC_DEV float xmax32(float x) {
  float lo, hi;
  half_swap(x, lo, hi);
  return v_max(lo, hi);
}
VC_DEV float xsum32(float x) {
  float lo, hi;
  half_swap(x, lo, hi);
  return lo + hi;
}

template <bool BOUNDED>
__global__ __launch_bounds__(256, 1) void attn64_kernel(const Attn64Args a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // the whole accumulator file belongs to the asm statements of this kernel
  asm volatile("" ::: "a0", "a15", "a31", "a47", "a63", "a79", "a95", "a111", "a127", "a143", "a159", "a175", "a191", "a207",
               "a223", "a239", "a255");

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lq = lane & 31, hh = lane >> 5;
  const float c_scale = 0.08838834764831845f * 1.4426950408889634f;   // 128^-0.5 * log2(e)
  const float MASKED = -29952.0f;                                      // bf16-exact; 2^(MASKED - m) == 0

  // ---- lane constants ----
  uint32_t k_rd[8], v_rd[4];
  const int krow = swap23(lq);          // key row (within a 32-key block) whose fragment this lane feeds to the MFMA
  {
#pragma unroll
    for (int t = 0; t < 8; ++t) k_rd[t] = krow * 256 + (((2 * t + hh) ^ (krow & 15)) << 4);
#pragma unroll
    for (int s = 0; s < 4; ++s) v_rd[s] = V_RING0 + lq * 128 + (((2 * s + hh) ^ ((lq >> 1) & 7)) << 4);
  }
  uint32_t k_off[4], v_off[4];      // LDS-DMA source byte offsets of this lane's 4 K and 4 V^T pieces (tile 0)
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = i * 256 + tid;
    const int row = c >> 4;
    k_off[i] = (uint32_t)row * (uint32_t)a.ld * 2u + (uint32_t)(((c & 15) ^ (row & 15)) << 4);
    const int d = c >> 3;
    v_off[i] = ((uint32_t)d * (uint32_t)a.Lpad + (uint32_t)((((c & 7) ^ ((d >> 1) & 7))) << 3)) * 2u;
  }
  const uint32_t k_step = (uint32_t)KVB * (uint32_t)a.ld * 2u;
  // keys past row L - 1 re-read row L - 1 (they are masked later); the 16-B column of a piece is the same for all 4
  const uint32_t k_max = (uint32_t)(a.L - 1) * (uint32_t)a.ld * 2u + (uint32_t)(((tid & 15) ^ ((tid >> 4) & 15)) << 4);
  // k_aug / q_aug fragments (d = 128 + hh*8 + e): only elements 0, 1 of the hh = 0 lanes are ever non-zero
  const uint32_t aug_on = hh == 0 ? 0xffffffffu : 0u;
  const uint32_t kaug_one = 0x3f80u & aug_on;                                   // k_aug[128] = 1
  const uint32_t qaug_mask = ((uint32_t)f2bf(MASKED) << 16) & aug_on;           // q_aug[129] = -29952

#ifdef VC_ATTN_TIMESTAMPS
  const uint64_t ts0 = __builtin_amdgcn_s_memtime();
  int ts_tiles = 0, ts_seg = 0;
  // per work item of the workgroup (up to 3): start, prologue done, first tile done, loop done, epilogue done, tiles
#define TS_SEG(k) do { if (a.debug_ts && tid == 0 && ts_seg < 3) a.debug_ts[blockIdx.x * 32 + 8 + ts_seg * 8 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define TS_SEG(k) do { } while (0)
#endif
  const int G = gridDim.x;
  const int nkt_all = (a.L + KVB - 1) / KVB;
  const bool split = a.full_rounds >= 0;
  int tu = 0, tu_end = 0, it_first = 0, rounds_left = 0x7fffffff, id_full = blockIdx.x, id_step = G, id_tail = 0;
  if (split) {      // few scalars survive into the loop: the next whole item and its stride, the tail's first item, the unit range
    const int slot = blockIdx.x >> 3;
    const Sched64 sc = sched64(blockIdx.x & 7, G, a.items, nkt_all);
    tu = chunk_begin64(slot, sc.units, sc.W);
    tu_end = chunk_begin64(slot + 1, sc.units, sc.W);
    it_first = tu / nkt_all;
    rounds_left = sc.rounds;
    id_full = sc.start + slot;
    id_step = sc.W;
    id_tail = sc.start + sc.rounds * sc.W;
  }

  for (;;) {
    int id, kt0 = 0, kt1 = -1, piece = -1;
    if (rounds_left > 0) {
      if (!split && id_full >= a.items) break;
      id = split ? id_full : xcd_remap(id_full, a.items);
      id_full += id_step;
      --rounds_left;
    } else {
      if (tu >= tu_end) break;
      const int it = tu / nkt_all;
      kt0 = tu - it * nkt_all;
      kt1 = min(nkt_all, kt0 + (tu_end - tu));
      tu += kt1 - kt0;
      id = id_tail + it;
      if (kt1 - kt0 != nkt_all) piece = blockIdx.x * 2 + (it - it_first);
    }
    const int qb_i = id % a.qblocks;
    const int bh = id / a.qblocks;
    const int h = bh % a.H, b = bh / a.H;
    const int L = a.L;
    const int kvlen = a.kv_len ? __builtin_amdgcn_readfirstlane(a.kv_len[b]) : L;
    const int gap_lo = a.kv_gap ? __builtin_amdgcn_readfirstlane(a.kv_gap[2 * b]) : 0;
    const int gap_hi = a.kv_gap ? __builtin_amdgcn_readfirstlane(a.kv_gap[2 * b + 1]) : 0;
    const int nkt = (kvlen + KVB - 1) / KVB;
    if (kt1 < 0) kt1 = nkt;
#ifdef VC_ATTN_TIMESTAMPS
    ts_tiles += kt1 - kt0;
    if (a.debug_ts && tid == 0 && ts_seg < 3) a.debug_ts[blockIdx.x * 32 + 8 + ts_seg * 8 + 5] = kt1 - kt0;
#endif
    TS_SEG(0);

    const bf16_t* __restrict__ qbase = a.qkv + (long)b * a.bstride + h * 128;
    const char* kbytes = (const char*)(qbase + a.H * 128);
    const char* vbytes = (const char*)(a.vt + ((long)(b * a.H + h) * 128) * a.Lpad);

    // LDS-DMA of source tile min(n, kt1 - 1) into ring slot SLOT (4 K + 4 V^T pieces per wave and tile)
    // (the wave's LDS destination base is re-derived from one SGPR per use: hoisted out of the loop body the distinct
    // M0 values - and their spills - cost more than one s_add each)
    // (VC_WL_PIN re-derives the base from one SGPR per use; the bounded tile, whose pieces sit in both phases, hits
    // "illegal VGPR to SGPR copy" in hipcc's backend with the pin and compiles to the same per-use s_add without it)
// (vector-register pins: with the per-wave `dead` branch around the tile bodies hipcc keeps these wave-uniform values in
// VECTOR registers - "illegal VGPR to SGPR copy" for a scalar pin; the DMA statements take lane offsets anyway)
#define VC_STREAM_PIN(a_, b_, c_) asm volatile("" : "+s"(a_), "+s"(b_), "+s"(c_))
#define VC_WL_PIN do { if constexpr (!BOUNDED) asm volatile("" : "+s"(wave_lds)); } while (0)
    int wave_lds = wave * 1024;
    auto dma_k = [&](auto SLOT, int n, int i) {
      const int kt = min(n, kt1 - 1);
      const uint32_t off = min(k_off[i] + (uint32_t)kt * k_step, k_max);
      VC_WL_PIN;
      glds16(kbytes + off, smem + wave_lds + (decltype(SLOT)::value * K_TILE + i * 4096));
    };
    auto dma_v = [&](auto SLOT, int n, int i) {
      const int kt = min(n, kt1 - 1);
      VC_WL_PIN;
      glds16(vbytes + (v_off[i] + (uint32_t)kt * (KVB * 2)), smem + wave_lds + (V_RING0 + decltype(SLOT)::value * V_TILE + i * 4096));
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>;

    // ---- prologue: K(0..2), V(0..1) in flight; Q * c -> AGPRs; O = 0 ----
#pragma unroll
    for (int i = 0; i < 4; ++i) { dma_k(I0{}, kt0, i); dma_v(I0{}, kt0, i); }
#pragma unroll
    for (int i = 0; i < 4; ++i) { dma_k(I1{}, kt0 + 1, i); dma_v(I1{}, kt0 + 1, i); }
#pragma unroll
    for (int i = 0; i < 4; ++i) dma_k(I2{}, kt0 + 2, i);
    const int q0 = qb_i * QB + wave * QW;
    if (a.q_pre) {
      // the projection's epilogue left QKNorm, RoPE and the softmax scale in the rows (VcGemmProblem.qn_prescale): the
      // fragments go from HBM straight into the MFMA operand registers - no arithmetic per work item
      sfor<0, 2>([&](auto QBc) {
        constexpr int qb = decltype(QBc)::value;
        const int tok = min(q0 + qb * 32 + lq, L - 1);
        const bf16_t* qp = qbase + (long)tok * a.ld + hh * 8;
        sfor<0, 8>([&](auto T) { constexpr int t = decltype(T)::value; load_q<A_Q + (qb * 8 + t) * 4, t * 32>(qp); });
      });
    } else
    sfor<0, 2>([&](auto QBc) {
      constexpr int qb = decltype(QBc)::value;
      const int tok = min(q0 + qb * 32 + lq, L - 1);
      const bf16_t* qp = qbase + (long)tok * a.ld + hh * 8;
      u32x4 raw[8];
#pragma unroll
      for (int t = 0; t < 8; ++t) raw[t] = *(const u32x4*)(qp + t * 16);
      if (a.q_scale) {
        // QKNorm (RMS over the 128 dims of the head: this lane's 64 + its partner's in the other half-wave; layers.py:63-84)
        // and RoPE (math.py:112-117) on the raw projection output, then the softmax scale; rounding points of the
        // reference: (x * rrms) -> bf16, * scale -> bf16; the rotated value is rounded ONCE, with the scale folded in
        float ss = 0.f;
#pragma unroll
        for (int t = 0; t < 8; ++t)
#pragma unroll
          for (int e = 0; e < 4; ++e) { const float x0 = lo_bf(raw[t][e]), x1 = hi_bf(raw[t][e]); ss += x0 * x0; ss += x1 * x1; }
        const float rrms = 1.0f / sqrtf(xsum32(ss) * (1.0f / 128.0f) + 1e-6f);
        const bf16_t* gsc = (tok < a.split ? a.q_scale : a.q_scale2) + hh * 8;
        const float* rp = a.rope + (long)b * a.rope_bstride + (long)tok * 128 + hh * 8;
        sfor<0, 8>([&](auto T) {
          constexpr int t = decltype(T)::value;
          const u32x4 gw = *(const u32x4*)(gsc + t * 16);
          const f32x4 c0 = *(const f32x4*)(rp + t * 16), c1 = *(const f32x4*)(rp + t * 16 + 4);
          const float cs[8] = {c0[0], c0[1], c0[2], c0[3], c1[0], c1[1], c1[2], c1[3]};
          sfor<0, 4>([&](auto E) {
            constexpr int e = decltype(E)::value;
            const float x0 = rbf(rbf(lo_bf(raw[t][e]) * rrms) * lo_bf(gw[e])), x1 = rbf(rbf(hi_bf(raw[t][e]) * rrms) * hi_bf(gw[e]));
            const float co = cs[2 * e], si = cs[2 * e + 1];
            agpr_write<A_Q + (qb * 8 + t) * 4 + e>(
                __builtin_bit_cast(float, pack2bf((co * x0 - si * x1) * c_scale, (si * x0 + co * x1) * c_scale)));
          });
        });
      } else {
        sfor<0, 8>([&](auto T) {
          constexpr int t = decltype(T)::value;
          sfor<0, 4>([&](auto E) {        // q * (128^-0.5 * log2 e), rounded to bf16 once more: S comes out in the log2 domain
            constexpr int e = decltype(E)::value;
            const uint32_t w = raw[t][e];
            agpr_write<A_Q + (qb * 8 + t) * 4 + e>(__builtin_bit_cast(float, pack2bf(lo_bf(w) * c_scale, hi_bf(w) * c_scale)));
          });
        });
      }
    });
    sfor<0, 128>([&](auto I) { agpr_write<A_O + decltype(I)::value>(0.f); });
    wait_vm<0>();
    __builtin_amdgcn_s_barrier();
    SB();
    TS_SEG(1);

    f32x16 SBk[6];                     // S^T blocks; roles rotate with the tile (see tile())
    u32x4 P[2][4];                     // [query block][16-key step]
    // (VGPRs from the start: the bounded form writes single words of P across the tile boundary, and an undefined
    // vector on the loop's back edge ends in "illegal VGPR to SGPR copy" in the backend)
#pragma unroll
    for (int i = 0; i < 8; ++i) asm volatile("" : "=v"(P[i >> 2][i & 3]));
    u32x4 vf[8];                       // V^T fragment ring
    float m_run[2] = {0.f, 0.f};       // running row max (log2 domain), always bf16-representable: -m_run sits in q_aug
    float l_acc[2] = {0.f, 0.f}, alpha[2] = {1.f, 1.f};
    u32x4 qaug[2] = {{qaug_mask, 0u, 0u, 0u}, {qaug_mask, 0u, 0u, 0u}};
    bool resc = false;

    // K fragments of ring slot SLOT -> a[192:255]
    auto read_k = [&](auto SLOT, auto UT) {
      constexpr int ut = decltype(UT)::value, u = ut >> 3, t = ut & 7;
      lds_k<A_K + ut * 4, decltype(SLOT)::value * K_TILE + u * 8192>(k_rd[t]);
    };
    // k_aug fragment of key block u of tile n: dimension 128 = 1, dimension 129 = (key >= kv_len)
    auto make_kaug = [&](int n, int u) -> u32x4 {
      const int key = n * KVB + u * 32 + krow;
      const uint32_t m = (key >= kvlen || (key >= gap_lo && key < gap_hi)) ? (0x3f800000u & aug_on) : 0u;
      return u32x4{kaug_one | m, 0u, 0u, 0u};
    };
    float mxp[4];
    auto max_step = [&](f32x16& Sx, auto Cc, auto Jc) {     // 8 steps per chain: 16 values -> one
      constexpr int c = decltype(Cc)::value, j = decltype(Jc)::value;
      if constexpr (j == 0) mxp[c] = v_max3(Sx[0], Sx[1], Sx[2]);
      else if constexpr (j < 7) mxp[c] = v_max3(mxp[c], Sx[2 * j + 1], Sx[2 * j + 2]);
      else mxp[c] = v_max(mxp[c], Sx[15]);
    };
    float mq[2];
    // S holds c*q.k - m_run: mq = how far a row's max lies ABOVE the running max
    auto decide0 = [&]() {
      mq[0] = v_max(mxp[0], mxp[1]);
      mq[1] = v_max(mxp[2], mxp[3]);
    };
    auto decide1 = [&](auto QBc) { mq[decltype(QBc)::value] = xmax32(mq[decltype(QBc)::value]); };
    // deferred rescale (as attention.hip): m moves only when some row of the wave grew by more than 2^8, or on the
    // first tile.  The new max is rounded to bf16 (it must sit in q_aug exactly); S(kt+1), already accumulated against
    // the old max, is corrected in place, O and l at the start of the next tile.
    auto decide2 = [&](auto B0, bool first) {       // B0: index of the first block of the tile's S in SBk
      constexpr int b0 = decltype(B0)::value;
      resc = first || !__all((mq[0] <= 8.0f) && (mq[1] <= 8.0f));
      if (resc) {
        sfor<0, 2>([&](auto QBc) {
          constexpr int qb = decltype(QBc)::value;
          const float want = first ? mq[qb] : m_run[qb] + fmaxf(mq[qb], 0.f);
          const bf16_t nb = f2bf(-want);
          const float m_new = -bf2f(nb);
          const float delta = m_new - m_run[qb];
          alpha[qb] = __builtin_amdgcn_exp2f(-delta);
          m_run[qb] = m_new;
          qaug[qb][0] = ((uint32_t)nb & aug_on) | qaug_mask;
          sfor<0, 2>([&](auto Uc) {
            constexpr int blk = (b0 + qb * 2 + decltype(Uc)::value) % 6;
#pragma unroll
            for (int r = 0; r < 16; ++r) SBk[blk][r] -= delta;
          });
        });
      }
    };
    // the rare path: O *= alpha, l *= alpha (runs between two P.V phases)
    auto rescale_o = [&]() {
      if (resc) {
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");          // the last P.V MFMAs retire before a[0:127] is read
        sfor<0, 128>([&](auto I) {
          constexpr int i = decltype(I)::value;
          agpr_write<A_O + i>(agpr_read<A_O + i>() * alpha[i >> 6]);
        });
        l_acc[0] *= alpha[0];
        l_acc[1] *= alpha[1];
        asm volatile("s_nop 7" ::: "memory");
        resc = false;
      }
    };
    // one S chain: 8 k-steps over the head dimension + the (-m, mask) step
    auto qk_step = [&](f32x16& Sx, auto Cc, auto Tc, const u32x4& kaug, bool masked_tile) {
      constexpr int c = decltype(Cc)::value, qb = c >> 1, u = c & 1, t = decltype(Tc)::value;
      if constexpr (t < 8) mfma_qk<A_K + (u * 8 + t) * 4, A_Q + (qb * 8 + t) * 4, t == 0>(Sx);
#ifndef VC_A64_NO_AUG       // analysis builds only (wrong results)
      else if (!BOUNDED || masked_tile) mfma_aug(Sx, kaug, qaug[qb]);     // BOUNDED: m = 0 for ever, only masks need the step
#endif
    };
    // does KV tile n hold a masked key (wave-uniform)?
    auto tile_masked = [&](int n) { return n * KVB + KVB > kvlen || (n * KVB < gap_hi && n * KVB + KVB > gap_lo); };

    // ---- first tile, not overlapped: S(kt0) = K(kt0) . Q^T (m = 0), its row max, K(kt0+1) fragments ----
    sfor<0, 16>([&](auto UT) { read_k(I0{}, UT); });
    wait_lgkm<0>();
    __builtin_amdgcn_s_barrier();      // every wave holds its K(kt0) fragments: slot 0 may take K(kt0+3)
#pragma unroll
    for (int i = 0; i < 4; ++i) dma_k(I0{}, kt0 + 3, i);
    SB();
    {
      u32x4 ka[2] = {make_kaug(kt0, 0), make_kaug(kt0, 1)};
      const bool msk0 = tile_masked(kt0);
      sfor<0, 36>([&](auto Gp) {
        constexpr int g = decltype(Gp)::value, c = g / 9, t = g % 9;
        qk_step(SBk[c], std::integral_constant<int, c>{}, std::integral_constant<int, t>{}, ka[c & 1], msk0);
      });
    }
    SB();
    sfor<0, 16>([&](auto UT) { read_k(I1{}, UT); });
    asm volatile("s_nop 15" ::: "memory");                          // S(kt0) complete before the VALU reads it
    SB();
    if constexpr (!BOUNDED) {
      sfor<0, 4>([&](auto Cc) { sfor<0, 8>([&](auto Jc) { max_step(SBk[decltype(Cc)::value], Cc, Jc); }); });
      decide0();
      decide1(I0{});
      decide1(I1{});
      decide2(I0{}, true);
    }
    resc = false;                                                   // O = 0, l = 0: nothing to rescale on the first tile
    if constexpr (BOUNDED) {                                        // the early pairs of the item's first tile (no phase B before it)
      sfor<0, a64s::N_EARLY>([&](auto Ic) {
        constexpr int k = a64s::EARLY_PAIR[decltype(Ic)::value], pq = k >> 4, idx = k & 15, pu = idx >> 3, r0 = (idx & 7) * 2;
        const float e0 = __builtin_amdgcn_exp2f(SBk[pq * 2 + pu][r0]), e1 = __builtin_amdgcn_exp2f(SBk[pq * 2 + pu][r0 + 1]);
        l_acc[pq] += e0;
        l_acc[pq] += e1;
        P[pq][pu * 2 + (r0 >> 3)][(r0 & 7) >> 1] = v_cvt_pk(e0, e1);
      });
    }
    wait_lgkm<0>();
    __builtin_amdgcn_s_barrier();      // every wave holds its K(kt0+1) fragments before tile kt0 sends K(kt0+4) into that slot
    SB();
    TS_SEG(2);

    // ---- one tile of the steady state; J = (tile - kt0) % 3 selects ring slots and the S block roles:
    //      S(kt) = blocks (4J + i) % 6, S(kt+1) = blocks (4J + 4 + i) % 6, i = chain = 2*qb + u  (the last two chains of
    //      S(kt+1) take the blocks of S(kt) that phase A has finished exponentiating by then) ----
    // (a) running-max form: the row max of S(kt+1) and the rescale decision fill the P.V phase
    auto tile_r = [&](auto Jc, int kt) {
      constexpr int J = decltype(Jc)::value;
      constexpr int BASE = (4 * J) % 6;
      using SLOT_V = std::integral_constant<int, J>;                 // V^T(kt)
      using SLOT_K2 = std::integral_constant<int, (J + 2) % 3>;      // K(kt+2) fragments / V^T(kt+2) DMA
      using SLOT_K4 = std::integral_constant<int, (J + 1) % 3>;      // K(kt+4) DMA
      rescale_o();
      u32x4 ka[2] = {make_kaug(kt + 1, 0), make_kaug(kt + 1, 1)};
      SB();
      // ---------------- phase A: S(kt+1) = K(kt+1) . Q^T - m  ||  P(kt), l  ||  V^T(kt) fragments 0..7 ----------------
      float pe0 = 0.f, pe1 = 0.f;
      sfor<0, 36>([&](auto Gp) {
        constexpr int g = decltype(Gp)::value, c = g / 9, t = g % 9;
        // every one of the 36 MFMA gaps takes a pair; the V^T fragments sit in the four thin gaps behind the (-m) steps
#ifndef VC_A64_NO_MFMA
        qk_step(SBk[(BASE + 4 + c) % 6], std::integral_constant<int, c>{}, std::integral_constant<int, t>{}, ka[c & 1], true);
#else
        if constexpr (t == 0) asm volatile("" : "=v"(SBk[(BASE + 4 + c) % 6]));
#endif
#ifndef VC_A64_NO_SOFTMAX
        if constexpr (g > 0 && g <= 32) {   // pair g-1: row sum and bf16 pack of the two probabilities exponentiated one gap earlier
          constexpr int k = g - 1, pq = k >> 4, idx = k & 15, pu = idx >> 3, r0 = (idx & 7) * 2;
          l_acc[pq] += pe0;
          l_acc[pq] += pe1;
          PIN(l_acc[pq]);
          P[pq][pu * 2 + (r0 >> 3)][(r0 & 7) >> 1] = v_cvt_pk(pe0, pe1);
        }
        if constexpr (g < 32) {
          constexpr int k = g, pq = k >> 4, idx = k & 15, pu = idx >> 3, r0 = (idx & 7) * 2;
          pe0 = __builtin_amdgcn_exp2f(SBk[(BASE + pq * 2 + pu) % 6][r0]);
          pe1 = __builtin_amdgcn_exp2f(SBk[(BASE + pq * 2 + pu) % 6][r0 + 1]);
        }
#endif
#ifndef VC_A64_NO_LDS
        if constexpr (g >= 32) {          // the four thin gaps at the end take the first eight V^T fragments
          constexpr int f = (g - 32) * 2;   // fragment (dt = f >> 2, s = f & 3)
          lds_v<SLOT_V::value * V_TILE + (f >> 2) * 4096>(vf[f], v_rd[f & 3]);
          lds_v<SLOT_V::value * V_TILE + ((f + 1) >> 2) * 4096>(vf[f + 1], v_rd[(f + 1) & 3]);
        }
#else
        if constexpr (g >= 32) { asm volatile("" : "=v"(vf[(g - 32) * 2])); asm volatile("" : "=v"(vf[(g - 32) * 2 + 1])); }
#endif
        if constexpr (g == 35) wait_lgkm<0>();
        SB();
      });
      // ---------------- phase B: O += V^T(kt) . P(kt)^T  ||  row max of S(kt+1), K(kt+2) -> AGPRs, DMA ----------------
      sfor<0, 32>([&](auto Gp) {
        constexpr int g = decltype(Gp)::value, dt = g >> 3, s = (g >> 1) & 3, qb = g & 1;
        if constexpr (g == 16) wait_lgkm<0>();      // V^T fragments 8..15 (and the K fragments issued so far)
#ifndef VC_A64_NO_MFMA
        mfma_pv<A_O + (qb * 4 + dt) * 16>(vf[(dt & 1) * 4 + s], P[qb][s]);
#else     // the asynchronous ds_read results stay live up to here (a dead output register would be reused while the read is in flight)
        asm volatile("" ::"v"(vf[(dt & 1) * 4 + s]), "v"(P[qb][s]));
#endif
#ifndef VC_A64_NO_SOFTMAX
        // S(kt+1) was completed by the last MFMAs of phase A: its first VALU read comes two MFMA gaps later
        if constexpr (g >= 2 && g <= 9)
          sfor<0, 4>([&](auto Cc) { max_step(SBk[(BASE + 4 + decltype(Cc)::value) % 6], Cc, std::integral_constant<int, g - 2>{}); });
        if constexpr (g == 10) decide0();
        if constexpr (g == 11) decide1(I0{});
        if constexpr (g == 12) decide1(I1{});
#ifndef VC_A64_NO_AUG
        if constexpr (g == 13) decide2(std::integral_constant<int, (BASE + 4) % 6>{}, false);
#endif
#endif
#ifndef VC_A64_NO_LDS
        if constexpr ((g & 1) && g < 16) {          // V^T fragment (dt + 2, s) into the register (dt, s) just retired
          lds_v<SLOT_V::value * V_TILE + (dt + 2) * 4096>(vf[(dt & 1) * 4 + s], v_rd[s]);
        }
        if constexpr (g >= 14 && g < 30) read_k(SLOT_K2{}, std::integral_constant<int, g - 14>{});
#endif
#ifndef VC_A64_NO_DMA     // analysis builds only (wrong results): the loop without one of its ingredients
        if constexpr (g < 2 || (g >= 18 && g < 30 && (g & 1) == 0)) {    // LDS-DMA pieces: gaps 0, 1, 18, 20, ... 28
          constexpr int i = g < 2 ? g : (g - 18) / 2 + 2;
          if constexpr (i < 4) dma_v(SLOT_K2{}, kt + 2, i);
          else dma_k(SLOT_K4{}, kt + 4, i - 4);
        }
#endif
        SB();
      });
#ifndef VC_A64_NO_DMA
      wait_vm<8>();
#endif
      wait_lgkm<0>();
#ifndef VC_A64_NO_BARRIER
      __builtin_amdgcn_s_barrier();
#endif
      SB();
    };

    // (b) bounded-logit form: no row max, so the fillers of a tile are exp / add / pack, the 32 fragment reads and the 8
    // LDS-DMA pieces - dealt out over the 64 MFMA gaps by tools/gen_a64_sched.py (attention64_sched.h) so that every gap
    // carries about the same issue price.  12 pairs of P(kt+1) are exponentiated in the P.V phase of tile kt, straight into
    // the P registers that the s-major order of that phase has already retired (and into l: nothing rescales it here).
    float pe0[32], pe1[32];              // probabilities between their v_exp and their row-sum add / pack (2-3 pairs live)
    float l_e[2] = {0.f, 0.f};           // row sums of the pairs exponentiated one tile early: they join l when their tile opens
                                         // (the last tile of an item exponentiates pairs of a tile that does not exist)
    auto run_tok = [&](auto PH, auto TI, auto SBASE, auto SLOT_V, auto SLOT_K2, auto SLOT_K4, int kt) {
      constexpr a64s::Tok t = tok_at<decltype(PH)::value, decltype(TI)::value>();
      constexpr int k = t.a, pq = k >> 4, idx = k & 15, pu = idx >> 3, r0 = (idx & 7) * 2;
      constexpr int blk = (decltype(SBASE)::value + pq * 2 + pu) % 6;
      if constexpr (t.kind == a64s::T_E0) { pe0[k & 31] = __builtin_amdgcn_exp2f(SBk[blk][r0 & 15]); PIN(pe0[k & 31]); }
      else if constexpr (t.kind == a64s::T_E1) { pe1[k & 31] = __builtin_amdgcn_exp2f(SBk[blk][(r0 + 1) & 15]); PIN(pe1[k & 31]); }
      else if constexpr (t.kind == a64s::T_A0) {
        if constexpr (decltype(PH)::value == 0) l_acc[pq & 1] += pe0[k & 31];
        else if constexpr (k != a64s::EARLY_FIRST[pq & 1]) l_e[pq & 1] += pe0[k & 31];
      } else if constexpr (t.kind == a64s::T_A1) {
        if constexpr (decltype(PH)::value == 0) { l_acc[pq & 1] += pe1[k & 31]; PIN(l_acc[pq & 1]); }
        else if constexpr (k != a64s::EARLY_FIRST[pq & 1]) { l_e[pq & 1] += pe1[k & 31]; PIN(l_e[pq & 1]); }
        else { l_e[pq & 1] = pe0[k & 31] + pe1[k & 31]; PIN(l_e[pq & 1]); }      // the first early pair of a query block opens the sum
      }
      else if constexpr (t.kind == a64s::T_CV) P[pq & 1][(pu * 2 + (r0 >> 3)) & 3][(r0 & 7) >> 1] = v_cvt_pk(pe0[k & 31], pe1[k & 31]);
#ifndef VC_A64_NO_LDS
      else if constexpr (t.kind == a64s::T_RV)       // V^T(kt) fragment (dt = a & 3, s = a >> 2) -> register b
        lds_v<decltype(SLOT_V)::value * V_TILE + (t.a & 3) * 4096>(vf[t.b % a64s::V_REGS], v_rd[(t.a >> 2) & 3]);
      else if constexpr (t.kind == a64s::T_RV2)      // the same, into a register the P.V phase has just retired
        lds_v<decltype(SLOT_V)::value * V_TILE + (t.a & 3) * 4096>(vf[t.b % a64s::V_REGS], v_rd[(t.a >> 2) & 3]);
      else if constexpr (t.kind == a64s::T_RK) read_k(SLOT_K2, std::integral_constant<int, t.a & 15>{});
      else if constexpr (t.kind == a64s::T_WAIT) wait_lgkm<t.a & 15>();
#endif
#ifndef VC_A64_NO_DMA
      else if constexpr (t.kind == a64s::T_DMA) {
        if constexpr (t.a < 4) dma_v(SLOT_K2, kt + 2, t.a & 3);
        else dma_k(SLOT_K4, kt + 4, t.a & 3);
      }
#endif
    };
    auto tile_b = [&](auto Jc, auto Mc, int kt) {
      constexpr int J = decltype(Jc)::value;
      constexpr bool MASKED = decltype(Mc)::value;    // S(kt+1) covers masked keys: its chains take the (0, -29952) k-step
      constexpr int BASE = (4 * J) % 6;
      using SLOT_V = std::integral_constant<int, J>;                 // V^T(kt)
      using SLOT_K2 = std::integral_constant<int, (J + 2) % 3>;      // K(kt+2) fragments / V^T(kt+2) DMA
      using SLOT_K4 = std::integral_constant<int, (J + 1) % 3>;      // K(kt+4) DMA
      using SA = std::integral_constant<int, BASE>;                  // S(kt)
      using SB1 = std::integral_constant<int, (BASE + 4) % 6>;       // S(kt+1)
      // (built at the head of the tile: an MFMA in an asm statement reads a register the VALU has just written without the
      // wait states hipcc would have inserted for its own instructions)
      u32x4 ka[2];
      if constexpr (MASKED) { ka[0] = make_kaug(kt + 1, 0); ka[1] = make_kaug(kt + 1, 1); }
      SB();
      // ---------------- phase A: S(kt+1) = K(kt+1) . Q^T  ||  20 pairs of P(kt), V^T(kt) fragments (s = 0, 1), V^T(kt+2) DMA ----
      sfor<0, 32>([&](auto Gp) {
        constexpr int g = decltype(Gp)::value, c = g >> 3, t = g & 7;
#ifndef VC_A64_NO_MFMA
        mfma_qk<A_K + ((c & 1) * 8 + t) * 4, A_Q + ((c >> 1) * 8 + t) * 4, t == 0>(SBk[(BASE + 4 + c) % 6]);
        if constexpr (t == 7 && MASKED) mfma_aug(SBk[(BASE + 4 + c) % 6], ka[c & 1], qaug[c >> 1]);     // masked keys (a tile that crosses kv_len or the gap)
#else
        if constexpr (t == 0) asm volatile("" : "=v"(SBk[(BASE + 4 + c) % 6]));
#endif
#ifndef VC_A64_NO_SOFTMAX
        if constexpr (g == 0) {            // the pairs of P(kt) that phase B of the previous tile exponentiated join the row sums
          l_acc[0] += l_e[0];
          l_acc[1] += l_e[1];
        }
        sfor<a64s::A_FIRST[g], a64s::A_FIRST[g + 1]>([&](auto Ti) {
          run_tok(I0{}, Ti, SA{}, SLOT_V{}, SLOT_K2{}, SLOT_K4{}, kt);
        });
#endif
        SB();
      });
      // ---------------- phase B: O += V^T(kt) . P(kt)^T, 16-key step major  ||  12 pairs of P(kt+1), V^T(kt) fragments
      //                  (s = 2, 3), K(kt+2) -> AGPRs, K(kt+4) DMA ----------------
      sfor<0, 32>([&](auto Gp) {
        constexpr int g = decltype(Gp)::value, s = g >> 3, dt = (g >> 1) & 3, qb = g & 1;
#ifndef VC_A64_NO_MFMA
        mfma_pv<A_O + (qb * 4 + dt) * 16>(vf[a64s::PV_REG[g]], P[qb][s]);
#else
        asm volatile("" ::"v"(vf[a64s::PV_REG[g]]), "v"(P[qb][s]));
#endif
#ifndef VC_A64_NO_SOFTMAX
        sfor<a64s::B_FIRST[g], a64s::B_FIRST[g + 1]>([&](auto Ti) {
          run_tok(I1{}, Ti, SB1{}, SLOT_V{}, SLOT_K2{}, SLOT_K4{}, kt);
        });
#endif
        SB();
      });
#ifndef VC_A64_NO_DMA
      wait_vm<8>();
#endif
      wait_lgkm<0>();
#ifndef VC_A64_NO_BARRIER
      __builtin_amdgcn_s_barrier();
#endif
      SB();
    };
    auto tile = [&](auto Jc, int kt) {
      if constexpr (BOUNDED) {           // two instruction streams: tiles with masked keys are the last of an item, or none
        if (tile_masked(min(kt + 1, kt1 - 1))) tile_b(Jc, std::true_type{}, kt);
        else tile_b(Jc, std::false_type{}, kt);
      } else tile_r(Jc, kt);
    };

    for (int kt = kt0;;) {
      tile(I0{}, kt); if (++kt >= kt1) break;
      tile(I1{}, kt); if (++kt >= kt1) break;
      tile(I2{}, kt); if (++kt >= kt1) break;
    }

    // ---- epilogue ----
    TS_SEG(3);
    asm volatile("s_waitcnt vmcnt(0)\n\ts_nop 15\n\ts_nop 15" ::: "memory");     // ring quiet, last P.V MFMAs retired
    if (piece >= 0) {          // part of an item's keys only: (O / l, m, l) of this key range, merged by attn64_merge_kernel
      char* pp = (char*)a.part + (long)piece * PART64_BYTES;
      sfor<0, 2>([&](auto QBc) {
        constexpr int qb = decltype(QBc)::value;
        const float l_tot = xsum32(l_acc[qb]);
        const float inv = 1.0f / l_tot;
        sfor<0, 16>([&](auto Gq) {
          constexpr int g = decltype(Gq)::value, A0 = A_O + (qb * 4 + (g >> 2)) * 16 + (g & 3) * 4;
          const f16x4 w = {(_Float16)(agpr_read<A0 + 0>() * inv), (_Float16)(agpr_read<A0 + 1>() * inv),
                           (_Float16)(agpr_read<A0 + 2>() * inv), (_Float16)(agpr_read<A0 + 3>() * inv)};
          *(f16x4*)(pp + (((wave * 2 + qb) * 16 + g) * 64 + lane) * 8) = w;
        });
        const f32x2 ml = {m_run[qb], l_tot};
        *(f32x2*)(pp + PART64_O_BYTES + ((wave * 2 + qb) * 64 + lane) * 8) = ml;
      });
    } else {
      sfor<0, 2>([&](auto QBc) {
        constexpr int qb = decltype(QBc)::value;
        const float l_tot = xsum32(l_acc[qb]);
        const int q = q0 + qb * 32 + lq;
        const float inv = (q < kvlen && !(q >= gap_lo && q < gap_hi)) ? 1.0f / l_tot : 0.0f;     // padded query rows -> 0 (pad_input, math.py:96)
        bf16_t* orow = a.out + (long)b * a.out_bstride + (long)min(q, L - 1) * a.ldo + h * 128;
        sfor<0, 16>([&](auto Gq) {
          constexpr int g = decltype(Gq)::value, dt = g >> 2, gg = g & 3, A0 = A_O + (qb * 4 + dt) * 16 + gg * 4;
          u32x2 w;
          w[0] = pack2bf(agpr_read<A0 + 0>() * inv, agpr_read<A0 + 1>() * inv);
          w[1] = pack2bf(agpr_read<A0 + 2>() * inv, agpr_read<A0 + 3>() * inv);
          if (q < L) *(u32x2*)(orow + dt * 32 + gg * 8 + hh * 4) = w;
        });
      });
    }
    // no barrier here: every wave passed the last tile's s_barrier after its final LDS reads, and a wave's own
    // vmcnt(0) above orders its in-flight pieces before the next item's prologue DMA into the same slots
#ifdef VC_ATTN_TIMESTAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (profiling build: the stamp sees the stores complete)
    TS_SEG(4);
    ++ts_seg;
#endif
  }  // work items
#ifdef VC_ATTN_TIMESTAMPS
  if (a.debug_ts && tid == 0) {
    a.debug_ts[blockIdx.x * 32 + 0] = ts0;
    a.debug_ts[blockIdx.x * 32 + 1] = __builtin_amdgcn_s_memtime();
    a.debug_ts[blockIdx.x * 32 + 2] = ts_tiles;
    a.debug_ts[blockIdx.x * 32 + 3] = ts_seg;
  }
#endif
}

// ---------------------------------------------------------------------------------------------------------------------------
// STREAM form of the bounded-logit kernel (round 6): the product's launches at cfg 2 / 3 / 5 - bounded logits AND queries that
// arrive normalised, rotated and scaled from the qkv GEMM's epilogue (VcAttention.q_prescaled), so an item has no arithmetic
// of its own.  Same tile (tile_b's generated schedule), but the K / V^T LDS-DMA stream of a workgroup does NOT stop at the
// end of a work item: the look-ahead pieces of an item's last tiles are the next item's first tiles, the next item's query
// fragments are requested at the start of the last tile (the Q registers are idle then: the last tile has no S(t+1) to
// compute), and the item boundary is  last tile (P.V only) -> O out (16-byte stores, no wait) -> S'(0) = K'(0) . Q'^T with the
// zeroing of O in its MFMA gaps -> first regular tile.  What round 5's timeline (profiles/r06a_attn64_timeline.log) charged
// per boundary - epilogue 10.8 k ticks, prologue 7.9 k (a full DMA round trip with the matrix pipe idle), first tile 2.5 k, and
// the 32 MFMAs of an S(t+1) that does not exist - becomes ~2 k + ~2 k.  An item of fewer than 4 tiles ends in a HARD boundary
// (drain, prologue as below): the look-ahead beyond it was issued before its successor was known.
template <bool BOUNDED>
__global__ __launch_bounds__(256, 1) void attn64s_kernel(const Attn64Args a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  asm volatile("" ::: "a0", "a15", "a31", "a47", "a63", "a79", "a95", "a111", "a127", "a143", "a159", "a175", "a191", "a207",
               "a223", "a239", "a255");
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lq = lane & 31, hh = lane >> 5;
  const float MASKED = -29952.0f;

  uint32_t k_rd[8], v_rd[4];
  const int krow = swap23(lq);
  {
#pragma unroll
    for (int t = 0; t < 8; ++t) k_rd[t] = krow * 256 + (((2 * t + hh) ^ (krow & 15)) << 4);
#pragma unroll
    for (int s = 0; s < 4; ++s) v_rd[s] = V_RING0 + lq * 128 + (((2 * s + hh) ^ ((lq >> 1) & 7)) << 4);
  }
  // LDS-DMA source offsets of this lane's piece 0; piece i of a tile is 16 K rows (32 V^T rows) further on - a SCALAR step (the
  // swizzled 16-byte column of a lane is the same in all four: row & 15 and (d >> 1) & 7 do not depend on i)
  const uint32_t k_off0 = (uint32_t)(tid >> 4) * (uint32_t)a.ld * 2u + (uint32_t)(((tid & 15) ^ ((tid >> 4) & 15)) << 4);
  const uint32_t v_off0 = ((uint32_t)(tid >> 3) * (uint32_t)a.Lpad + (uint32_t)((((tid & 7) ^ ((tid >> 4) & 7))) << 3)) * 2u;
  const uint32_t k_piece = 32u * (uint32_t)a.ld, v_piece = 64u * (uint32_t)a.Lpad;
  const uint32_t k_step = (uint32_t)KVB * (uint32_t)a.ld * 2u;
  // rows past L - 1 (the last tile of a sequence whose length is no multiple of 64) must not be fetched: their offsets are
  // clamped to the LAST 16 bytes of row L - 1's head - a SCALAR bound (row L - 1 itself never exceeds it, every later row does);
  // what such a lane then reads is some chunk of a real key row, and its key is masked whatever it holds
  const uint32_t k_bound = (uint32_t)(a.L - 1) * (uint32_t)a.ld * 2u + 240u;
  const uint32_t aug_on = hh == 0 ? 0xffffffffu : 0u;
  const uint32_t kaug_one = 0x3f80u & aug_on;
  const uint32_t qaug_mask = ((uint32_t)f2bf(MASKED) << 16) & aug_on;

#ifdef VC_ATTN_TIMESTAMPS
  const uint64_t ts0 = __builtin_amdgcn_s_memtime();
  int ts_tiles = 0, ts_seg = 0;
#endif
  // ---- work schedule: identical to attn64_kernel's (whole items per round, then this workgroup's chunk of the XCD's tail) ----
  const int G = gridDim.x;
  const int L = a.L;
  const int nkt_all = (L + KVB - 1) / KVB;
  const bool split = a.full_rounds >= 0;
  int tu = 0, tu_end = 0, it_first = 0, rounds_left = 0x7fffffff, id_full = blockIdx.x, id_step = G, id_tail = 0;
  if (split) {
    const int slot = blockIdx.x >> 3;
    const Sched64 sc = sched64(blockIdx.x & 7, G, a.items, nkt_all);
    tu = chunk_begin64(slot, sc.units, sc.W);
    tu_end = chunk_begin64(slot + 1, sc.units, sc.W);
    it_first = tu / nkt_all;
    rounds_left = sc.rounds;
    id_full = sc.start + slot;
    id_step = sc.W;
    id_tail = sc.start + sc.rounds * sc.W;
  }
  // one work item (or the part of a tail item) of this workgroup, as plain scalars (a struct handed to the lambdas by reference
  // ended up in scratch memory): id, tile range, partial index, the batch element's masks, and the byte offsets of the head's
  // K rows in qkv / V^T rows in vt (32 bits: checked by the launcher) - scalars added to the lane offset; the 64-bit bases stay
  // the kernel arguments (hipcc moves a selected 64-bit pointer into VGPRs, which an "s" asm operand cannot take)
#define VC_SEG(p) int p##id = 0, p##kt0 = 0, p##kt1 = 0, p##piece = -1, p##kvlen = 0, p##gap_lo = 0, p##gap_hi = 0; uint32_t p##ko = 0, p##vo = 0
  VC_SEG(c_);
  VC_SEG(n_);
  bool have_next = false;
  // fetch the next work item into the n_ set; false when the workgroup has none left
  auto next_seg = [&]() __attribute__((always_inline)) -> bool {
    int id, kt0 = 0, kt1 = -1, piece = -1;
    // a.inmerge: the workgroup's share of the tail comes FIRST, so that its pieces have long been published when the
    // workgroups that combine them (at the very end of their own work) ask for them
    if (rounds_left > 0 && !(a.inmerge && tu < tu_end)) {
      if (!split && id_full >= a.items) return false;
      id = split ? id_full : xcd_remap(id_full, a.items);
      id_full += id_step;
      --rounds_left;
    } else {
      if (tu >= tu_end) return false;
      const int it = tu / nkt_all;
      kt0 = tu - it * nkt_all;
      kt1 = min(nkt_all, kt0 + (tu_end - tu));
      tu += kt1 - kt0;
      id = id_tail + it;
      if (kt1 - kt0 != nkt_all) piece = blockIdx.x * 2 + (it - it_first);
    }
    const int bh = id / a.qblocks;
    const int h = bh % a.H, b = bh / a.H;
    n_id = id; n_kt0 = kt0; n_piece = piece;
    n_kvlen = a.kv_len ? __builtin_amdgcn_readfirstlane(a.kv_len[b]) : L;
    n_gap_lo = a.kv_gap ? __builtin_amdgcn_readfirstlane(a.kv_gap[2 * b]) : 0;
    n_gap_hi = a.kv_gap ? __builtin_amdgcn_readfirstlane(a.kv_gap[2 * b + 1]) : 0;
    n_kt1 = kt1 < 0 ? (n_kvlen + KVB - 1) / KVB : kt1;
    n_ko = ((uint32_t)b * (uint32_t)a.bstride + (uint32_t)(h * 128 + a.H * 128)) * 2u;
    n_vo = (uint32_t)(b * a.H + h) * 128u * (uint32_t)a.Lpad * 2u;
    return true;
  };
#define VC_SEG_ADVANCE do { c_id = n_id; c_kt0 = n_kt0; c_kt1 = n_kt1; c_piece = n_piece; c_kvlen = n_kvlen; c_gap_lo = n_gap_lo; \
                            c_gap_hi = n_gap_hi; c_ko = n_ko; c_vo = n_vo; have_next = next_seg(); } while (0)
  if (!next_seg()) return;
  VC_SEG_ADVANCE;

  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  using I2 = std::integral_constant<int, 2>;

  // LDS-DMA of stream tile n (counted in cur's tile indices: n >= c_kt1 is tile n - c_kt1 of the NEXT item) into ring slot
  // SLOT.  Beyond the end of what is known the last tile is fetched again (never read as a real tile).
  // M0 (the wave's LDS destination) is written INSIDE the asm statement, one s_add from ONE scalar: handed the builtin, hipcc
  // hoists the 24 distinct destinations of the three rotations out of the loop, spills them and reloads each with a v_readlane
  const uint32_t wave_lds = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(lptr_t)(smem + wave * 1024));
  // Source of stream tile n (counted in cur's tile indices).  An item occupies a whole number of rotations: its tiles, then
  // 0-2 EMPTY steps, so that every item starts at J = 0 (one instruction stream for the first step, a single-entry tile loop).
  // n in [kt1, pad1): an empty step, nothing to fetch (the item's last tile is fetched again: never read); n >= pad1: tile
  // n - pad1 of the NEXT item, if one is known.
  int pad1 = 0;          // c_kt0 + 3 * ceil((c_kt1 - c_kt0) / 3), set per item
  auto src_tile = [&](int n, bool& fwd) __attribute__((always_inline)) -> int {
    fwd = n >= pad1 && have_next;
    return fwd ? min(n_kt0 + (n - pad1), n_kt1 - 1) : min(n, c_kt1 - 1);
  };
  const char* const k_base = (const char*)a.qkv;
  const char* const v_base = (const char*)a.vt;
  // scalar part of a tile's source offsets (computed ONCE per tile, at its head: the ~15 scalar instructions of the look-ahead
  // mapping stay out of the MFMA gaps), then one piece = one v_add (+ v_min for K) and the asm statement
  auto k_src = [&](int n, uint32_t& bound) __attribute__((always_inline)) -> uint32_t {
    bool fwd;
    const int kt = src_tile(n, fwd);
    const uint32_t so = fwd ? n_ko : c_ko;
    bound = k_bound + so;
    return (uint32_t)kt * k_step + so;
  };
  auto v_src = [&](int n) __attribute__((always_inline)) -> uint32_t {
    bool fwd;
    const int kt = src_tile(n, fwd);
    return (fwd ? n_vo : c_vo) + (uint32_t)kt * (KVB * 2);
  };
  auto dma_k_at = [&](auto SLOT, uint32_t sc, uint32_t bound, int i) __attribute__((always_inline)) {
#ifndef VC_A64_NO_DMA      // analysis builds only (wrong results): the loop without one of its ingredients
    glds16_m0(k_base, min(k_off0 + (sc + (uint32_t)i * k_piece), bound), wave_lds, decltype(SLOT)::value * K_TILE + i * 4096);
#endif
  };
  auto dma_v_at = [&](auto SLOT, uint32_t sc, int i) __attribute__((always_inline)) {
#ifndef VC_A64_NO_DMA
    glds16_m0(v_base, v_off0 + (sc + (uint32_t)i * v_piece), wave_lds, V_RING0 + decltype(SLOT)::value * V_TILE + i * 4096);
#endif
  };
  auto dma_k = [&](auto SLOT, int n, int i) __attribute__((always_inline)) { uint32_t bd; const uint32_t sc = k_src(n, bd); dma_k_at(SLOT, sc, bd, i); };
  auto dma_v = [&](auto SLOT, int n, int i) __attribute__((always_inline)) { dma_v_at(SLOT, v_src(n), i); };
  // the 16 query fragments of item g -> a[128:191] (waited for with vmcnt by the caller)
  auto load_queries = [&](int g_id) __attribute__((always_inline)) {
    const int qb_i = g_id % a.qblocks;
    const int bh = g_id / a.qblocks;
    const int h = bh % a.H, b = bh / a.H;
    const bf16_t* qbase = a.qkv + (long)b * a.bstride + h * 128;
    const int q0 = qb_i * QB + wave * QW;
    sfor<0, 2>([&](auto QBc) {
      constexpr int qb = decltype(QBc)::value;
      const int tok = min(q0 + qb * 32 + lq, L - 1);
      const bf16_t* qp = qbase + (long)tok * a.ld + hh * 8;
      sfor<0, 8>([&](auto T) { constexpr int t = decltype(T)::value; load_q<A_Q + (qb * 8 + t) * 4, t * 32>(qp); });
    });
  };

  f32x16 SBk[6];
  u32x4 P[2][4];
#pragma unroll
  for (int i = 0; i < 8; ++i) asm volatile("" : "=v"(P[i >> 2][i & 3]));
  u32x4 vf[a64s::V_REGS];            // V^T fragment ring
  u32x4 vfr[8];                      // (running-max form: its own ring of eight, dt-major P.V order)
#pragma unroll
  for (int i = 0; i < 8; ++i) asm volatile("" : "=v"(vfr[i]));
#pragma unroll
  for (int i = 0; i < a64s::V_REGS; ++i) asm volatile("" : "=v"(vf[i]));
  float l_acc[2] = {0.f, 0.f}, l_e[2] = {0.f, 0.f};
  float pe0[32], pe1[32];
  int pub = -1;                       // a.inmerge: the piece whose flags are still to be set
  // A wave whose 64 queries all lie past row L - 1 (the last 256-query item of a sequence whose length is no multiple of 256: two
  // of four waves at L = 3968) has no row to write.  It keeps its part in the workgroup's K / V^T stream - its LDS-DMA pieces,
  // the counted waits, every barrier, the K'(0) fragment reads that belong to the NEXT item - and skips the item's MFMAs,
  // exponentials and fragment reads: 3 % of the launch's arithmetic at cfg 2, on a board that runs at its power cap.
  bool dead = false;
#ifdef VC_A64_NO_DEAD_WAVES      // A/B builds: every wave computes its padded rows (rounds 2-6)
#define VC_DEAD_OF(id) false
#else
#define VC_DEAD_OF(id) (BOUNDED && __builtin_amdgcn_readfirstlane((int)(((id) % a.qblocks) * QB + wave * QW >= L)) != 0)   /* (the running-max
     template has no registers left for the second item skeleton: 44 accumulator spills) */
#endif

  auto read_k = [&](auto SLOT, auto UT) __attribute__((always_inline)) {
    constexpr int ut = decltype(UT)::value, u = ut >> 3, t = ut & 7;
    lds_k<A_K + ut * 4, decltype(SLOT)::value * K_TILE + u * 8192>(k_rd[t]);
  };
  auto make_kaug = [&](int n, int u) __attribute__((always_inline)) -> u32x4 {
    const int key = n * KVB + u * 32 + krow;
    const uint32_t m = (key >= c_kvlen || (key >= c_gap_lo && key < c_gap_hi)) ? (0x3f800000u & aug_on) : 0u;
    return u32x4{kaug_one | m, 0u, 0u, 0u};
  };
  // the 9th k-step of an S chain over a tile with masked keys: S^T += k_aug . q_aug = -29952 on the masked keys (reference
  // point 0: bounded logits).  Cold path - both fragments are built here, with the wait states between a VALU write and an
  // MFMA read that hipcc inserts for its own instructions and an asm statement has to bring along
  auto masked_step = [&](f32x16& Sx, int n, int u) __attribute__((always_inline)) {
    u32x4 ka = make_kaug(n, u);
    u32x4 qa = {qaug_mask, 0u, 0u, 0u};
    asm volatile("s_nop 7\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(Sx) : "v"(ka), "v"(qa));
  };
  auto tile_masked = [&](int n) __attribute__((always_inline)) { return n * KVB + KVB > c_kvlen || (n * KVB < c_gap_hi && n * KVB + KVB > c_gap_lo); };

  // ---- running-max form (BOUNDED = false: weights whose QK-norm scales do not bound the logits): the row max m of a tile is
  // subtracted BY THE MATRIX PIPE (9th k-step: q_aug = (-m, -29952), k_aug = (1, key masked)), m moves only when a row of the
  // wave grows by more than 2^8 (deferred rescale), exactly as attn64_kernel<false> - same arithmetic, inside the stream
  // skeleton (stream DMA across items, soft boundaries, in-launch combine) ----
  // (scalars and two named vectors, not arrays: arrays captured by the nested lambdas ended up in scratch memory)
  float m_run0 = 0.f, m_run1 = 0.f, alpha0 = 1.f, alpha1 = 1.f;
  u32x4 qaug_r0 = {qaug_mask, 0u, 0u, 0u}, qaug_r1 = {qaug_mask, 0u, 0u, 0u};
  int resc = 0;
  float mxp0 = 0.f, mxp1 = 0.f, mxp2 = 0.f, mxp3 = 0.f, mq0 = 0.f, mq1 = 0.f;
  auto max_step = [&](f32x16& Sx, float& mx, auto Jc) __attribute__((always_inline)) {     // 8 steps per chain: 16 values -> one
    constexpr int j = decltype(Jc)::value;
    if constexpr (j == 0) mx = v_max3(Sx[0], Sx[1], Sx[2]);
    else if constexpr (j < 7) mx = v_max3(mx, Sx[2 * j + 1], Sx[2 * j + 2]);
    else mx = v_max(mx, Sx[15]);
  };
  auto max_steps = [&](auto B0, auto Jc) __attribute__((always_inline)) {                  // step j of all four chains of the S tile whose first block is B0
    constexpr int b0 = decltype(B0)::value;
    max_step(SBk[(b0 + 0) % 6], mxp0, Jc);
    max_step(SBk[(b0 + 1) % 6], mxp1, Jc);
    max_step(SBk[(b0 + 2) % 6], mxp2, Jc);
    max_step(SBk[(b0 + 3) % 6], mxp3, Jc);
  };
  auto decide0 = [&]() __attribute__((always_inline)) {
    mq0 = v_max(mxp0, mxp1);
    mq1 = v_max(mxp2, mxp3);
  };
  auto decide1 = [&](auto QBc) __attribute__((always_inline)) { if constexpr (decltype(QBc)::value == 0) mq0 = xmax32(mq0); else mq1 = xmax32(mq1); };
  auto new_max = [&](float mq, float& m_run, float& alpha, u32x4& qa, f32x16& S0, f32x16& S1, bool first) __attribute__((always_inline)) {
    const float want = first ? mq : m_run + fmaxf(mq, 0.f);
    const bf16_t nb = f2bf(-want);
    const float m_new = -bf2f(nb);
    const float delta = m_new - m_run;
    alpha = __builtin_amdgcn_exp2f(-delta);
    m_run = m_new;
    qa[0] = ((uint32_t)nb & aug_on) | qaug_mask;
#pragma unroll
    for (int r = 0; r < 16; ++r) { S0[r] -= delta; S1[r] -= delta; }
  };
  auto decide2 = [&](auto B0, bool first) __attribute__((always_inline)) {       // B0: index of the first block of the tile's S in SBk
    constexpr int b0 = decltype(B0)::value;
    resc = (first || !__all((mq0 <= 8.0f) && (mq1 <= 8.0f))) ? 1 : 0;
    if (resc) {
      new_max(mq0, m_run0, alpha0, qaug_r0, SBk[(b0 + 0) % 6], SBk[(b0 + 1) % 6], first);
      new_max(mq1, m_run1, alpha1, qaug_r1, SBk[(b0 + 2) % 6], SBk[(b0 + 3) % 6], first);
    }
  };
  // (always_inline: left to the inliner's size heuristics this body becomes a CALL, and everything it captures by reference
  // - l, alpha, the flag - then lives in scratch memory, loaded and stored in every tile)
  auto rescale_o = [&]() __attribute__((always_inline)) {      // the rare path: O *= alpha, l *= alpha (between two P.V phases)
    if (resc) {
      asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
      sfor<0, 64>([&](auto I) { constexpr int i = decltype(I)::value; agpr_write<A_O + i>(agpr_read<A_O + i>() * alpha0); });
      sfor<64, 128>([&](auto I) { constexpr int i = decltype(I)::value; agpr_write<A_O + i>(agpr_read<A_O + i>() * alpha1); });
      l_acc[0] *= alpha0;
      l_acc[1] *= alpha1;
      asm volatile("s_nop 7" ::: "memory");
      resc = 0;
    }
  };
  // one tile of the running-max form (attn64_kernel<false>'s tile_r: 36 + 32 MFMAs, every phase-A gap takes a pair, the row
  // max of S(kt+1) fills the P.V phase); LAST / RK as tile_s below
  auto tile_rs = [&](auto Jc, auto LASTc, int kt) __attribute__((always_inline)) {
    constexpr int J = decltype(Jc)::value;
    constexpr bool LAST = decltype(LASTc)::value, RK = !LAST || J == 1;
    constexpr int BASE = (4 * J) % 6;
    using SLOT_V = std::integral_constant<int, J>;
    using SLOT_K2 = std::integral_constant<int, (J + 2) % 3>;
    using SLOT_K4 = std::integral_constant<int, (J + 1) % 3>;
    rescale_o();
    const u32x4 ka0 = make_kaug(kt + 1, 0), ka1 = make_kaug(kt + 1, 1);
    uint32_t k_bd;
    uint32_t k_sc = k_src(kt + 4, k_bd), v_sc = v_src(kt + 2);
    VC_STREAM_PIN(k_sc, v_sc, k_bd);
    SB();
    float pe0r = 0.f, pe1r = 0.f;
    sfor<0, 36>([&](auto Gp) {
      constexpr int g = decltype(Gp)::value, c = g / 9, t = g % 9, qb = c >> 1, u = c & 1;
      if constexpr (!LAST) {
        if constexpr (t < 8) mfma_qk<A_K + (u * 8 + t) * 4, A_Q + (qb * 8 + t) * 4, t == 0>(SBk[(BASE + 4 + c) % 6]);
        else mfma_aug(SBk[(BASE + 4 + c) % 6], u == 0 ? ka0 : ka1, qb == 0 ? qaug_r0 : qaug_r1);
      }
      if constexpr (g > 0 && g <= 32) {   // pair g-1: row sum and bf16 pack of the two probabilities exponentiated one gap earlier
        constexpr int k = g - 1, pq = k >> 4, idx = k & 15, pu = idx >> 3, r0 = (idx & 7) * 2;
        l_acc[pq] += pe0r;
        l_acc[pq] += pe1r;
        PIN(l_acc[pq]);
        P[pq][pu * 2 + (r0 >> 3)][(r0 & 7) >> 1] = v_cvt_pk(pe0r, pe1r);
      }
      if constexpr (g < 32) {
        constexpr int k = g, pq = k >> 4, idx = k & 15, pu = idx >> 3, r0 = (idx & 7) * 2;
        pe0r = __builtin_amdgcn_exp2f(SBk[(BASE + pq * 2 + pu) % 6][r0]);
        pe1r = __builtin_amdgcn_exp2f(SBk[(BASE + pq * 2 + pu) % 6][r0 + 1]);
      }
      if constexpr (g >= 32) {            // the four thin gaps at the end take the first eight V^T fragments
        constexpr int f = (g - 32) * 2;   // fragment (dt = f >> 2, s = f & 3)
        lds_v<SLOT_V::value * V_TILE + (f >> 2) * 4096>(vfr[f], v_rd[f & 3]);
        lds_v<SLOT_V::value * V_TILE + ((f + 1) >> 2) * 4096>(vfr[f + 1], v_rd[(f + 1) & 3]);
      }
      if constexpr (g == 35) wait_lgkm<0>();
      SB();
    });
    sfor<0, 32>([&](auto Gp) {
      constexpr int g = decltype(Gp)::value, dt = g >> 3, s = (g >> 1) & 3, qb = g & 1;
      if constexpr (g == 16) wait_lgkm<0>();      // V^T fragments 8..15 (and the K fragments issued so far)
      mfma_pv<A_O + (qb * 4 + dt) * 16>(vfr[(dt & 1) * 4 + s], P[qb][s]);
      if constexpr (!LAST) {                      // S(kt+1) was completed by the last MFMAs of phase A: first VALU read two gaps later
        if constexpr (g >= 2 && g <= 9) max_steps(std::integral_constant<int, (BASE + 4) % 6>{}, std::integral_constant<int, g - 2>{});
        if constexpr (g == 10) decide0();
        if constexpr (g == 11) decide1(I0{});
        if constexpr (g == 12) decide1(I1{});
        if constexpr (g == 13) decide2(std::integral_constant<int, (BASE + 4) % 6>{}, false);
      }
      if constexpr ((g & 1) && g < 16) lds_v<SLOT_V::value * V_TILE + (dt + 2) * 4096>(vfr[(dt & 1) * 4 + s], v_rd[s]);
      if constexpr (RK && g >= 14 && g < 30) read_k(SLOT_K2{}, std::integral_constant<int, g - 14>{});
      if constexpr (g < 2 || (g >= 18 && g < 30 && (g & 1) == 0)) {    // LDS-DMA pieces: gaps 0, 1, 18, 20, ... 28
        constexpr int i = g < 2 ? g : (g - 18) / 2 + 2;
        if constexpr (i < 4) dma_v_at(SLOT_K2{}, v_sc, i);
        else dma_k_at(SLOT_K4{}, k_sc, k_bd, i - 4);
      }
      SB();
    });
    wait_vm<8>();
    wait_lgkm<0>();
    __builtin_amdgcn_s_barrier();
    SB();
  };
  // ---- one filler token of the generated schedule (attention64_sched.h); LAST: the item's last tile, which has no S(t+1) -
  // no exponentials of it, no K fragments of the tile after it (the K registers keep K'(0) of the next item), and its counted
  // waits, sized for the full read stream, become lgkmcnt(0) ----
  auto run_tok = [&](auto PH, auto TI, auto LASTc, auto RKc, auto SBASE, auto SLOT_V, auto SLOT_K2, auto SLOT_K4, uint32_t v_sc, uint32_t k_sc, uint32_t k_bd) __attribute__((always_inline)) {
    constexpr a64s::Tok t = tok_at<decltype(PH)::value, decltype(TI)::value>();
    constexpr bool LAST = decltype(LASTc)::value, PB = decltype(PH)::value == 1, RK = decltype(RKc)::value;
    constexpr int k = t.a, pq = k >> 4, idx = k & 15, pu = idx >> 3, r0 = (idx & 7) * 2;
    constexpr int blk = (decltype(SBASE)::value + pq * 2 + pu) % 6;
    constexpr bool is_pair = t.kind <= a64s::T_CV;
#ifdef VC_A64_NO_SOFTMAX      // analysis builds only (wrong results)
    if constexpr (is_pair) return;
#endif
#ifdef VC_A64_NO_LDS
    if constexpr (t.kind == a64s::T_RV || t.kind == a64s::T_RV2 || t.kind == a64s::T_RK || t.kind == a64s::T_WAIT) return;
#endif
    if constexpr (is_pair && PB && LAST) { }
    else if constexpr (t.kind == a64s::T_E0) { pe0[k & 31] = __builtin_amdgcn_exp2f(SBk[blk][r0 & 15]); PIN(pe0[k & 31]); }
    else if constexpr (t.kind == a64s::T_E1) { pe1[k & 31] = __builtin_amdgcn_exp2f(SBk[blk][(r0 + 1) & 15]); PIN(pe1[k & 31]); }
    else if constexpr (t.kind == a64s::T_A0) {
      if constexpr (!PB) l_acc[pq & 1] += pe0[k & 31];
      else if constexpr (k != a64s::EARLY_FIRST[pq & 1]) l_e[pq & 1] += pe0[k & 31];
    } else if constexpr (t.kind == a64s::T_A1) {
      if constexpr (!PB) { l_acc[pq & 1] += pe1[k & 31]; PIN(l_acc[pq & 1]); }
      else if constexpr (k != a64s::EARLY_FIRST[pq & 1]) { l_e[pq & 1] += pe1[k & 31]; PIN(l_e[pq & 1]); }
      else { l_e[pq & 1] = pe0[k & 31] + pe1[k & 31]; PIN(l_e[pq & 1]); }
    }
    else if constexpr (t.kind == a64s::T_CV) P[pq & 1][(pu * 2 + (r0 >> 3)) & 3][(r0 & 7) >> 1] = v_cvt_pk(pe0[k & 31], pe1[k & 31]);
    else if constexpr (t.kind == a64s::T_RV)
      lds_v<decltype(SLOT_V)::value * V_TILE + (t.a & 3) * 4096>(vf[t.b % a64s::V_REGS], v_rd[(t.a >> 2) & 3]);
    else if constexpr (t.kind == a64s::T_RV2)
      lds_v<decltype(SLOT_V)::value * V_TILE + (t.a & 3) * 4096>(vf[t.b % a64s::V_REGS], v_rd[(t.a >> 2) & 3]);
    else if constexpr (t.kind == a64s::T_RK) { if constexpr (RK) read_k(SLOT_K2, std::integral_constant<int, t.a & 15>{}); }
    else if constexpr (t.kind == a64s::T_WAIT) { if constexpr (LAST) wait_lgkm<0>(); else wait_lgkm<t.a & 15>(); }
    else if constexpr (t.kind == a64s::T_DMA) {
      if constexpr (t.a < 4) dma_v_at(SLOT_K2, v_sc, t.a & 3);
      else dma_k_at(SLOT_K4, k_sc, k_bd, t.a & 3);
    }
  };
  // ---- one tile: J = (stream tile index) % 3 selects ring slots and S block roles (attn64_kernel's tile_b) ----
  // LAST: the item's last tile - no S(t+1), no exponentials of it.  The K registers must hold K'(0) of the next item when its
  // first step runs: the step TWO positions before it reads them (a regular tile's K(t+2) read; with one empty step after the
  // last tile that is the last tile itself: J = 1), the step in between reads none.
  auto tile_s = [&](auto Jc, auto LASTc, int kt) __attribute__((always_inline)) {
    constexpr int J = decltype(Jc)::value;
    constexpr bool LAST = decltype(LASTc)::value;
    using RKc = std::integral_constant<bool, !LAST || J == 1>;
    constexpr int BASE = (4 * J) % 6;
    using SLOT_V = std::integral_constant<int, J>;
    using SLOT_K2 = std::integral_constant<int, (J + 2) % 3>;
    using SLOT_K4 = std::integral_constant<int, (J + 1) % 3>;
    using SA = std::integral_constant<int, BASE>;
    using SB1 = std::integral_constant<int, (BASE + 4) % 6>;
    // S(kt+1) covers masked keys (it crosses kv_len or the gap: an item's last tile, or none): its four chains take the
    // (0, -29952) k-step.  ONE scalar decides (a second instruction stream per J for the masked tile cost 48 registers of
    // allocation slack); the k_aug fragments are built at the head of the tile - an MFMA in an asm statement reads a register
    // the VALU has just written without the wait states hipcc inserts for its own instructions (masked_step pads them).
    const bool msk = !LAST && tile_masked(kt + 1);
    uint32_t k_bd;
    uint32_t k_sc = k_src(kt + 4, k_bd), v_sc = v_src(kt + 2);           // the stream's look-ahead: V^T(kt+2), K(kt+4)
    VC_STREAM_PIN(k_sc, v_sc, k_bd);                                      // (materialised HERE, not sunk into the gap of their first use)
    SB();
    sfor<0, 32>([&](auto Gp) {
      constexpr int g = decltype(Gp)::value, c = g >> 3, t = g & 7;
      if constexpr (!LAST) {
        mfma_qk<A_K + ((c & 1) * 8 + t) * 4, A_Q + ((c >> 1) * 8 + t) * 4, t == 0>(SBk[(BASE + 4 + c) % 6]);
        if constexpr (t == 7) { if (__builtin_expect(msk, 0)) masked_step(SBk[(BASE + 4 + c) % 6], kt + 1, c & 1); }
      }
      if constexpr (g == 0) {
        l_acc[0] += l_e[0];
        l_acc[1] += l_e[1];
      }
      sfor<a64s::A_FIRST[g], a64s::A_FIRST[g + 1]>([&](auto Ti) { run_tok(I0{}, Ti, LASTc, RKc{}, SA{}, SLOT_V{}, SLOT_K2{}, SLOT_K4{}, v_sc, k_sc, k_bd); });
      SB();
    });
    sfor<0, 32>([&](auto Gp) {
      constexpr int g = decltype(Gp)::value, s = g >> 3, dt = (g >> 1) & 3, qb = g & 1;
      mfma_pv<A_O + (qb * 4 + dt) * 16>(vf[a64s::PV_REG[g]], P[qb][s]);
      sfor<a64s::B_FIRST[g], a64s::B_FIRST[g + 1]>([&](auto Ti) { run_tok(I1{}, Ti, LASTc, RKc{}, SB1{}, SLOT_V{}, SLOT_K2{}, SLOT_K4{}, v_sc, k_sc, k_bd); });
      SB();
    });
#ifndef VC_A64_NO_DMA
    wait_vm<8>();
#endif
    wait_lgkm<0>();
#ifndef VC_A64_NO_BARRIER
    __builtin_amdgcn_s_barrier();
#endif
    SB();
  };
  // a dead wave's step: the wave's share of the stream's LDS-DMA (same pieces, same order as the tokens of a live tile), the K'(0)
  // fragments where a live step reads them (stream tile kt + 2 = the next item's first tile), the step's waits and its barrier
  auto dead_tile = [&](auto Jc, auto LASTc, int kt) __attribute__((always_inline)) {
    constexpr int J = decltype(Jc)::value;
    constexpr bool LAST = decltype(LASTc)::value;
    using SLOT_K2 = std::integral_constant<int, (J + 2) % 3>;
    using SLOT_K4 = std::integral_constant<int, (J + 1) % 3>;
    uint32_t k_bd;
    const uint32_t k_sc = k_src(kt + 4, k_bd), v_sc = v_src(kt + 2);
#pragma unroll
    for (int i = 0; i < 4; ++i) dma_v_at(SLOT_K2{}, v_sc, i);
    if (kt + 2 == pad1) sfor<0, 16>([&](auto UT) { read_k(SLOT_K2{}, UT); });      // the step two positions before the next item's first one
#pragma unroll
    for (int i = 0; i < 4; ++i) dma_k_at(SLOT_K4{}, k_sc, k_bd, i);
#ifndef VC_A64_NO_DMA
    wait_vm<8>();
#endif
    wait_lgkm<0>();
#ifndef VC_A64_NO_BARRIER
    __builtin_amdgcn_s_barrier();
#endif
    SB();
  };
  auto tile_any = [&](auto Jc, auto LASTc, int kt) __attribute__((always_inline)) {
    if constexpr (BOUNDED) tile_s(Jc, LASTc, kt);
    else tile_rs(Jc, LASTc, kt);
  };
  // ---- an EMPTY step of rotation J at stream position n (between an item's last tile and the rotation boundary): only the
  // stream's LDS-DMA (V^T(n+2), K(n+4): the next item's first tiles) and the step's waits; READ_K: the K'(0) fragments ----
  auto empty_s = [&](auto Jc, auto RKc, int n) __attribute__((always_inline)) {
    constexpr int J = decltype(Jc)::value;
    using SLOT_K2 = std::integral_constant<int, (J + 2) % 3>;
    using SLOT_K4 = std::integral_constant<int, (J + 1) % 3>;
#pragma unroll
    for (int i = 0; i < 4; ++i) dma_v(SLOT_K2{}, n + 2, i);
    if constexpr (decltype(RKc)::value) sfor<0, 16>([&](auto UT) { read_k(SLOT_K2{}, UT); });
#pragma unroll
    for (int i = 0; i < 4; ++i) dma_k(SLOT_K4{}, n + 4, i);
    wait_vm<8>();
    wait_lgkm<0>();
    __builtin_amdgcn_s_barrier();
    SB();
  };
  // ---- the first step of an item (always rotation J = 0): S(kt0) = K(kt0) . Q^T from the K fragments in a[192:255], O
  // cleared in the MFMA gaps; then the K fragments of the item's second tile (ring slot 1) and the pairs that the P.V phase
  // of a previous tile would have exponentiated ----
  auto first_s = [&](bool hard_start) __attribute__((always_inline)) {
    const bool msk0 = tile_masked(c_kt0);
    SB();
    sfor<0, 32>([&](auto Gp) {
      constexpr int g = decltype(Gp)::value, c = g >> 3, t = g & 7;
      mfma_qk<A_K + ((c & 1) * 8 + t) * 4, A_Q + ((c >> 1) * 8 + t) * 4, t == 0>(SBk[c]);
      if constexpr (t == 7) {
        if (__builtin_expect(msk0, 0)) masked_step(SBk[c], c_kt0, c & 1);      // (reference point 0 on an item's first tile in both forms)
      }
      sfor<0, 4>([&](auto Zc) { agpr_write<A_O + g * 4 + decltype(Zc)::value>(0.f); });
      SB();
    });
    if (hard_start) {                   // K(kt0+1) landed (V^T(0), K(2), V^T(1), K(3) behind it), in every wave
      wait_vm<16>();
      __builtin_amdgcn_s_barrier();
    }
    sfor<0, 16>([&](auto UT) { read_k(I1{}, UT); });
    asm volatile("s_nop 15" ::: "memory");                          // S complete before the VALU reads it
    SB();
    l_acc[0] = l_acc[1] = 0.f;
    l_e[0] = l_e[1] = 0.f;
    if constexpr (!BOUNDED) {             // the item's first reference point: the row max of its first tile
      m_run0 = m_run1 = 0.f;
      qaug_r0[0] = qaug_mask; qaug_r1[0] = qaug_mask;
      sfor<0, 8>([&](auto Jc) { max_steps(I0{}, Jc); });
      decide0();
      decide1(I0{});
      decide1(I1{});
      decide2(I0{}, true);
      resc = 0;                           // O = 0, l = 0: nothing to rescale
    }
    if constexpr (BOUNDED) sfor<0, a64s::N_EARLY>([&](auto Ic) {
      constexpr int k = a64s::EARLY_PAIR[decltype(Ic)::value], pq = k >> 4, idx = k & 15, pu = idx >> 3, r0 = (idx & 7) * 2;
      const float e0 = __builtin_amdgcn_exp2f(SBk[pq * 2 + pu][r0]), e1 = __builtin_amdgcn_exp2f(SBk[pq * 2 + pu][r0 + 1]);
      l_e[pq] += e0;
      l_e[pq] += e1;
      P[pq][pu * 2 + (r0 >> 3)][(r0 & 7) >> 1] = v_cvt_pk(e0, e1);
    });
    if (hard_start) wait_vm<8>();       // V^T(0) and K(2) landed: the first tile reads them
    wait_lgkm<0>();
    __builtin_amdgcn_s_barrier();      // every wave holds the fragments of the item's second K tile: the tile after next may overwrite the slot
    SB();
  };
  // ---- O of cur -> out (or the partial of a tail piece); no wait: the stores drain behind the next tiles ----
  auto store_out = [&]() __attribute__((always_inline)) {
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");               // the last P.V MFMAs retired before a[0:127] is read
    const int qb_i = c_id % a.qblocks;
    const int bh = c_id / a.qblocks;
    const int h = bh % a.H, b = bh / a.H;
    const int q0 = qb_i * QB + wave * QW;
    if (c_piece >= 0) {
      char* pp = (char*)a.part + (long)c_piece * PART64_BYTES;
      sfor<0, 2>([&](auto QBc) {
        constexpr int qb = decltype(QBc)::value;
        const float l_tot = xsum32(l_acc[qb]);
        const float inv = 1.0f / l_tot;
        // a.inmerge: another workgroup of THIS launch reads the piece - 8-byte agent-scope stores (write-through, `sc1`)
        // here, agent-scope loads there, a flag in between (cdna_hip_programming.md Guideline 16, form R1); else plain stores
        // for the merge kernel behind the launch boundary
        sfor<0, 16>([&](auto Gq) {
          constexpr int g = decltype(Gq)::value, A0 = A_O + (qb * 4 + (g >> 2)) * 16 + (g & 3) * 4;
          const f16x4 w = {(_Float16)(agpr_read<A0 + 0>() * inv), (_Float16)(agpr_read<A0 + 1>() * inv),
                           (_Float16)(agpr_read<A0 + 2>() * inv), (_Float16)(agpr_read<A0 + 3>() * inv)};
          char* dst = pp + (((wave * 2 + qb) * 16 + g) * 64 + lane) * 8;
          if (a.inmerge) __hip_atomic_store((uint64_t*)dst, __builtin_bit_cast(uint64_t, w), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          else *(f16x4*)dst = w;
        });
        const f32x2 ml = {BOUNDED ? 0.f : (qb == 0 ? m_run0 : m_run1), l_tot};          // (reference point 0 with bounded logits)
        char* dml = pp + PART64_O_BYTES + ((wave * 2 + qb) * 64 + lane) * 8;
        if (a.inmerge) __hip_atomic_store((uint64_t*)dml, __builtin_bit_cast(uint64_t, ml), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else *(f32x2*)dml = ml;
      });
      pub = c_piece;                                                  // (published one item later: see the item loop)
    } else {
      sfor<0, 2>([&](auto QBc) {
        constexpr int qb = decltype(QBc)::value;
        const float l_tot = xsum32(l_acc[qb]);
        const int q = q0 + qb * 32 + lq;
        const float inv = (q < c_kvlen && !(q >= c_gap_lo && q < c_gap_hi)) ? 1.0f / l_tot : 0.0f;     // padded query rows -> 0
        bf16_t* orow = a.out + (long)b * a.out_bstride + (long)min(q, L - 1) * a.ldo + h * 128 + hh * 8;
        // a lane holds d = 8 g + 4 hh + (0..3) of its query per 4 registers: the half-waves exchange so that the lower one owns
        // d = 16 m .. 16 m + 7 and the upper one 16 m + 8 .. 16 m + 15 - one 16-byte store per pair of groups
        sfor<0, 8>([&](auto Mq) {
          constexpr int m = decltype(Mq)::value, dt = m >> 1, A0 = A_O + (qb * 4 + dt) * 16 + (m & 1) * 8;
          // (v_cvt_pk_bf16_f32: the same round-to-nearest-even as the two conversions of pack2bf, in one instruction)
          uint32_t x0 = v_cvt_pk(agpr_read<A0 + 0>() * inv, agpr_read<A0 + 1>() * inv), x1 = v_cvt_pk(agpr_read<A0 + 2>() * inv, agpr_read<A0 + 3>() * inv);
          uint32_t y0 = v_cvt_pk(agpr_read<A0 + 4>() * inv, agpr_read<A0 + 5>() * inv), y1 = v_cvt_pk(agpr_read<A0 + 6>() * inv, agpr_read<A0 + 7>() * inv);
          asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(x0), "+v"(y0));
          asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(x1), "+v"(y1));
          const u32x4 w = {x0, x1, y0, y1};
          if (q < L) *(u32x4*)(orow + dt * 32 + (m & 1) * 16) = w;
        });
      });
    }
  };

  // ---- a.inmerge: a piece is PUBLISHED (flag = 1 for both query blocks) once its stores are complete in every wave: not by
  // draining the LDS-DMA stream behind store_out, but one item later - every step in between ended with a counted vmcnt
  // that covers the (older) stores and a barrier ----
  auto publish = [&](int piece) __attribute__((always_inline)) {
    if (tid == 0) {
      __hip_atomic_store(a.flags + piece * 2 + 0, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(a.flags + piece * 2 + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  };
  // =================================================== the item loop ===================================================
#ifdef VC_ATTN_TIMESTAMPS
#define TS_S(k) do { if (a.debug_ts && tid == 0 && ts_seg < 3) a.debug_ts[blockIdx.x * 32 + 8 + ts_seg * 8 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define TS_S(k) do { } while (0)
#endif
  bool hard = true;
  for (;;) {
    pad1 = c_kt0 + (c_kt1 - c_kt0 + 2) / 3 * 3;
    dead = VC_DEAD_OF(c_id);
#ifdef VC_ATTN_TIMESTAMPS
    ts_tiles += c_kt1 - c_kt0;
    if (a.debug_ts && tid == 0 && ts_seg < 3) a.debug_ts[blockIdx.x * 32 + 8 + ts_seg * 8 + 5] = c_kt1 - c_kt0;
#endif
    TS_S(0);
    if (hard) {
      // ---- HARD start: every wait stands in front of its first consumer.  Issue order K(0), Q, K(1), V^T(0), K(2), V^T(1)
      // [then K(3)]: S(kt0) needs K(0) and Q only; the rest lands under it (first_s waits for K(1) before it reads its fragments,
      // for V^T(0) and K(2) before the first tile; V^T(1) and K(3) are covered by that tile's own counted wait) ----
      const int kt0 = c_kt0;
#pragma unroll
      for (int i = 0; i < 4; ++i) dma_k(I0{}, kt0, i);
      if (!dead) load_queries(c_id);
#pragma unroll
      for (int i = 0; i < 4; ++i) dma_k(I1{}, kt0 + 1, i);
#pragma unroll
      for (int i = 0; i < 4; ++i) dma_v(I0{}, kt0, i);
#pragma unroll
      for (int i = 0; i < 4; ++i) dma_k(I2{}, kt0 + 2, i);
#pragma unroll
      for (int i = 0; i < 4; ++i) dma_v(I1{}, kt0 + 1, i);
      wait_vm<16>();                     // K(kt0) and the query fragments have landed
      __builtin_amdgcn_s_barrier();
      SB();
      sfor<0, 16>([&](auto UT) { read_k(I0{}, UT); });
      wait_lgkm<0>();
      __builtin_amdgcn_s_barrier();      // every wave holds its K(kt0) fragments: slot 0 may take K(kt0+3)
#pragma unroll
      for (int i = 0; i < 4; ++i) dma_k(I0{}, kt0 + 3, i);
      SB();
    }
    TS_S(1);
    const bool soft = have_next && (c_kt1 - c_kt0) >= 4;     // the look-ahead of a shorter item was issued before its successor was known
    const int kt_last = c_kt1 - 1;
    int kt = c_kt0, exit_j;
    if (__builtin_expect(dead, 0)) {
      // ---- the item of a DEAD wave: the skeleton of the live item below - every barrier, every wait, the wave's LDS-DMA, the
      // next item's query and K'(0) fragments - without its arithmetic ----
      if (hard) {
        wait_vm<16>();
        __builtin_amdgcn_s_barrier();
        wait_vm<8>();
      }
      wait_lgkm<0>();
      __builtin_amdgcn_s_barrier();
      SB();
      TS_S(2);
      for (;;) {
        if (kt >= kt_last) { exit_j = 0; break; }
        dead_tile(I0{}, std::false_type{}, kt); ++kt;
        if (kt >= kt_last) { exit_j = 1; break; }
        dead_tile(I1{}, std::false_type{}, kt); ++kt;
        if (kt >= kt_last) { exit_j = 2; break; }
        dead_tile(I2{}, std::false_type{}, kt); ++kt;
      }
      if (soft && !VC_DEAD_OF(n_id)) load_queries(n_id);
      if (exit_j == 0) {
        dead_tile(I0{}, std::true_type{}, kt);
        if (soft) { empty_s(I1{}, std::true_type{}, kt + 1); empty_s(I2{}, std::false_type{}, kt + 2); }
      } else if (exit_j == 1) {
        dead_tile(I1{}, std::true_type{}, kt);
        if (soft) empty_s(I2{}, std::false_type{}, kt + 1);
      } else {
        dead_tile(I2{}, std::true_type{}, kt);
      }
      TS_S(3);
      if (pub >= 0) { publish(pub); pub = -1; }
      if (c_piece >= 0) pub = c_piece;           // (no row of this wave exists; `pub` stays uniform across the workgroup)
    } else {
    first_s(hard);
    TS_S(2);
    // ---- the item's tiles but the last: a single-entry loop over the three rotations ----
    for (;;) {
      if (kt >= kt_last) { exit_j = 0; break; }
      tile_any(I0{}, std::false_type{}, kt); ++kt;
      if (kt >= kt_last) { exit_j = 1; break; }
      tile_any(I1{}, std::false_type{}, kt); ++kt;
      if (kt >= kt_last) { exit_j = 2; break; }
      tile_any(I2{}, std::false_type{}, kt); ++kt;
    }
    // ---- the last tile (P.V only), the empty steps up to the rotation boundary, O out ----
    if (soft && !VC_DEAD_OF(n_id)) load_queries(n_id);            // the Q registers are idle: the last tile computes no S(t+1)
    if (exit_j == 0) {
      tile_any(I0{}, std::true_type{}, kt);
      if (soft) { empty_s(I1{}, std::true_type{}, kt + 1); empty_s(I2{}, std::false_type{}, kt + 2); }
    } else if (exit_j == 1) {
      tile_any(I1{}, std::true_type{}, kt);
      if (soft) empty_s(I2{}, std::false_type{}, kt + 1);
    } else {
      tile_any(I2{}, std::true_type{}, kt);
    }
    TS_S(3);
    if (pub >= 0) { publish(pub); pub = -1; }
    store_out();
    }
    TS_S(4);
#ifdef VC_ATTN_TIMESTAMPS
    ++ts_seg;
#endif
    if (!have_next) break;
    VC_SEG_ADVANCE;
    hard = !soft;
    if (hard) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's look-ahead pieces land before the ring restarts
  }
  // =================================== a.inmerge: the tail items' pieces are combined here ===================================
  // task (tail item `it` of this XCD, query block qb) -> workgroup slot (2 it + qb) % W: out = sum_p l_p O_p / sum_p l_p over the
  // item's pieces in chunk order - the arithmetic and the order of attn64_merge_kernel (merge_fold64: bit-identical), the pieces
  // written by workgroups of this launch many tiles ago.  Every polled word is zero before a launch and zero again after it: a
  // flag is consumed by exactly one task, which clears it.  (Requesting the first task's pieces before the workgroup's last O
  // goes out - to run their round trips under those stores - was built: 136 more live registers across store_out, 924
  // compiler-generated accumulator moves; not kept.)
  if (a.inmerge) {
    if (pub >= 0) {                    // this workgroup's last work was a piece: drain its stores, then publish
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      publish(pub);
    }
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const Sched64 sc = sched64(xcd, G, a.items, nkt_all);
    for (int T = slot; T < 2 * sc.tail; T += sc.W) {
      const int it = T >> 1, qb = T & 1;
      const int u0 = it * nkt_all, u1 = u0 + nkt_all;
      int c = (int)(((long)u0 * sc.W) / sc.units);
      while (c > 0 && chunk_begin64(c, sc.units, sc.W) > u0) --c;
      while (c + 1 < sc.W && chunk_begin64(c + 1, sc.units, sc.W) <= u0) ++c;
      if (chunk_begin64(c + 1, sc.units, sc.W) >= u1) continue;      // the whole item ran inside one chunk: already written
      auto next_chunk = [&](int cc) __attribute__((always_inline)) {
        while (cc < sc.W && chunk_begin64(cc, sc.units, sc.W) < u1 && chunk_begin64(cc + 1, sc.units, sc.W) == chunk_begin64(cc, sc.units, sc.W)) ++cc;
        return (cc < sc.W && chunk_begin64(cc, sc.units, sc.W) < u1) ? cc : -1;
      };
      const long ml_off = PART64_O_BYTES + ((wave * 2 + qb) * 64 + lane) * 8;
      const long o_off = ((long)(wave * 2 + qb) * 16 * 64 + lane) * 8;
      float acc[16][4];
#pragma unroll
      for (int i = 0; i < 16; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[i][e] = 0.f;
      float wsum = 0.f, m = -INFINITY;
      // the item's pieces in chunk order, FOUR at a time (two or three in all at the product's geometries; up to W when an
      // XCD has a single tail item): all flags of a batch first, then ALL its loads in flight together (a chain of memory
      // round trips otherwise), then the folds in order
      constexpr int MAXP = 4;
      for (int cc = next_chunk(c); cc >= 0;) {
        int pcs[MAXP], np = 0;
        for (; cc >= 0 && np < MAXP; cc = next_chunk(cc + 1))
          pcs[np++] = (cc * 8 + xcd) * 2 + (it - chunk_begin64(cc, sc.units, sc.W) / nkt_all);
        // ONE relaxed agent-scope read of every flag of the batch, all in flight together (the pieces were published ~60 tiles
        // ago: nothing spins in practice); what is not there yet is polled, bounded - a piece that never arrives costs wrong
        // rows, not a hung GPU
        uint32_t fl[MAXP];
#pragma unroll
        for (int j = 0; j < MAXP; ++j) fl[j] = j < np ? __hip_atomic_load(a.flags + pcs[j] * 2 + qb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 1u;
#pragma unroll
        for (int j = 0; j < MAXP; ++j)
          if (j < np && fl[j] != 1u)
            for (unsigned spins = 0; __hip_atomic_load(a.flags + pcs[j] * 2 + qb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 1u; ++spins) {
              __builtin_amdgcn_s_sleep(16);
              if (spins > (1u << 20)) break;
            }
        f16x4 v[MAXP][16];
        f32x2 ml[MAXP];
#pragma unroll
        for (int j = 0; j < MAXP; ++j)
          if (j < np) {
            const char* pp = (const char*)a.part + (long)pcs[j] * PART64_BYTES;
#pragma unroll
            for (int i = 0; i < 16; ++i)
              v[j][i] = __builtin_bit_cast(f16x4, __hip_atomic_load((const uint64_t*)(pp + o_off + i * 512), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
            ml[j] = __builtin_bit_cast(f32x2, __hip_atomic_load((const uint64_t*)(pp + ml_off), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
          }
#pragma unroll
        for (int j = 0; j < MAXP; ++j)
          if (j < np) merge_fold64(acc, wsum, m, v[j], ml[j]);
        __syncthreads();               // every wave has seen the flags and holds its part of the pieces
        if (tid == 0)
          for (int j = 0; j < np; ++j) __hip_atomic_store(a.flags + pcs[j] * 2 + qb, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      const int id = sc.start + sc.rounds * sc.W + it;
      const int qb_i = id % a.qblocks, bh = id / a.qblocks;
      const int h = bh % a.H, b = bh / a.H;
      const int q = qb_i * QB + wave * QW + qb * 32 + lq;
      const float inv = 1.0f / wsum;
      bf16_t* orow = a.out + (long)b * a.out_bstride + (long)min(q, L - 1) * a.ldo + h * 128 + hh * 8;
#pragma unroll
      for (int mm = 0; mm < 8; ++mm) {
        uint32_t x0 = v_cvt_pk(acc[2 * mm][0] * inv, acc[2 * mm][1] * inv), x1 = v_cvt_pk(acc[2 * mm][2] * inv, acc[2 * mm][3] * inv);
        uint32_t y0 = v_cvt_pk(acc[2 * mm + 1][0] * inv, acc[2 * mm + 1][1] * inv), y1 = v_cvt_pk(acc[2 * mm + 1][2] * inv, acc[2 * mm + 1][3] * inv);
        asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(x0), "+v"(y0));
        asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(x1), "+v"(y1));
        const u32x4 wv = {x0, x1, y0, y1};
        if (q < L) *(u32x4*)(orow + mm * 16) = wv;
      }
    }
  }
#ifdef VC_ATTN_TIMESTAMPS
  if (a.debug_ts && tid == 0) {
    a.debug_ts[blockIdx.x * 32 + 0] = ts0;
    a.debug_ts[blockIdx.x * 32 + 1] = __builtin_amdgcn_s_memtime();
    a.debug_ts[blockIdx.x * 32 + 2] = ts_tiles;
    a.debug_ts[blockIdx.x * 32 + 3] = ts_seg;
  }
#endif
}

// Combines the pieces of the tail items: out = sum_p w_p (O_p / l_p) / sum_p w_p, w_p = l_p 2^(m_p - max m).  One
// workgroup per (XCD, tail item of that XCD, query block of the wave) on the XCD that wrote the pieces (block b -> XCD
// b % 8); thread layout = the writer's.  ONE pass over the pieces with a running maximum (the accumulator is rescaled when
// a piece raises it) and the next piece's loads issued before the current one is folded in: the kernel is a chain of
// L2 round trips, not bandwidth, and this keeps the chain at ~one trip (+ the store) whatever the number of pieces.
__global__ __launch_bounds__(256) void attn64_merge_kernel(const Attn64Args a, int G) {
  const int xcd = blockIdx.x & 7, it = blockIdx.x >> 4, qb = (blockIdx.x >> 3) & 1;
  const int nkt = (a.L + KVB - 1) / KVB;
  const Sched64 sc = sched64(xcd, G, a.items, nkt);
  if (it >= sc.tail) return;
  const int u0 = it * nkt, u1 = u0 + nkt;
  int c = (int)(((long)u0 * sc.W) / sc.units);
  while (c > 0 && chunk_begin64(c, sc.units, sc.W) > u0) --c;
  while (c + 1 < sc.W && chunk_begin64(c + 1, sc.units, sc.W) <= u0) ++c;
  if (chunk_begin64(c + 1, sc.units, sc.W) >= u1) return;        // the whole item ran inside one chunk: already written
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lq = lane & 31, hh = lane >> 5;
  const char* base = (const char*)a.part;
  const long ml_off = PART64_O_BYTES + ((wave * 2 + qb) * 64 + lane) * 8;
  const long o_off = ((long)(wave * 2 + qb) * 16 * 64 + lane) * 8;
  // chunk cc holds units of this item iff it is not empty (fewer units than blocks) and begins before u1
  auto next_chunk = [&](int cc) __attribute__((always_inline)) {
    while (cc < sc.W && chunk_begin64(cc, sc.units, sc.W) < u1 && chunk_begin64(cc + 1, sc.units, sc.W) == chunk_begin64(cc, sc.units, sc.W)) ++cc;
    return (cc < sc.W && chunk_begin64(cc, sc.units, sc.W) < u1) ? cc : -1;
  };
  auto piece_ptr = [&](int cc) __attribute__((always_inline)) {
    const int piece = (cc * 8 + xcd) * 2 + (it - chunk_begin64(cc, sc.units, sc.W) / nkt);
    return base + (long)piece * PART64_BYTES;
  };
  f16x4 v[16], vn[16];
  f32x2 ml, mln = {0.f, 0.f};
  int cc = next_chunk(c);
  {
    const char* pp = piece_ptr(cc);
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = *(const f16x4*)(pp + o_off + i * 512);
    ml = *(const f32x2*)(pp + ml_off);
  }
  float acc[16][4];
#pragma unroll
  for (int i = 0; i < 16; ++i)
#pragma unroll
    for (int e = 0; e < 4; ++e) acc[i][e] = 0.f;
  float wsum = 0.f, m = -INFINITY;
  for (;;) {
    const int nx = next_chunk(cc + 1);
    if (nx >= 0) {
      const char* pp = piece_ptr(nx);
#pragma unroll
      for (int i = 0; i < 16; ++i) vn[i] = *(const f16x4*)(pp + o_off + i * 512);
      mln = *(const f32x2*)(pp + ml_off);
    }
    merge_fold64(acc, wsum, m, v, ml);
    if (nx < 0) break;
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = vn[i];
    ml = mln;
    cc = nx;
  }
  const int id = sc.start + sc.rounds * sc.W + it;
  const int qb_i = id % a.qblocks, bh = id / a.qblocks;
  const int h = bh % a.H, b = bh / a.H;
  const int q = qb_i * QB + wave * QW + qb * 32 + lq;
  {
    // the writer's fragment layout: a lane holds d = 8 i + 4 hh + (0..3) of its query per group i.  The half-waves exchange
    // (v_permlane32_swap) so that the lower one owns d = 16 m .. 16 m + 7 and the upper one the next eight: 8 stores of
    // 16 bytes per lane instead of 16 of 8 (the tail of this kernel is store-issue bound, as every row-per-lane epilogue)
    const float inv = 1.0f / wsum;
    bf16_t* orow = a.out + (long)b * a.out_bstride + (long)min(q, a.L - 1) * a.ldo + h * 128 + hh * 8;
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      uint32_t x0 = v_cvt_pk(acc[2 * m][0] * inv, acc[2 * m][1] * inv), x1 = v_cvt_pk(acc[2 * m][2] * inv, acc[2 * m][3] * inv);
      uint32_t y0 = v_cvt_pk(acc[2 * m + 1][0] * inv, acc[2 * m + 1][1] * inv), y1 = v_cvt_pk(acc[2 * m + 1][2] * inv, acc[2 * m + 1][3] * inv);
      asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(x0), "+v"(y0));
      asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(x1), "+v"(y1));
      const u32x4 w = {x0, x1, y0, y1};
      if (q < a.L) *(u32x4*)(orow + m * 16) = w;
    }
  }
}

}  // namespace

// the pieces (two per workgroup); the flag words of the in-launch combine (VcAttention.variant bit 16) sit at the END of the
// whole attention scratch (vc_attention_flags_offset_impl, attention.hip), behind the partials of every other variant
int64_t vc_attention64_scratch_bytes_impl(int n_cu) { return (int64_t)n_cu * 2 * PART64_BYTES; }
int64_t vc_attention64_flags_bytes_impl(int n_cu) { return (int64_t)n_cu * 2 * 2 * 4; }

int vc_attention64_launch(const VcAttention& A, bool tail_split, int n_cu, uint64_t* debug_ts, hipStream_t s, char* err, int errlen) {
  Attn64Args a;
  a.debug_ts = debug_ts;
  const int32_t B = A.B, L = A.L, H = A.H;
  const int32_t* kv_len = A.kv_len;
  void* scratch = A.scratch; const int64_t scratch_bytes = A.scratch_bytes;
  a.qkv = (const bf16_t*)A.qkv; a.vt = (const bf16_t*)A.vt; a.out = (bf16_t*)A.out; a.kv_len = kv_len;
  a.kv_gap = kv_len ? A.kv_gap : nullptr;
  a.ld = A.ld; a.bstride = A.bstride; a.ldo = A.ldo; a.out_bstride = A.out_bstride;
  a.B = B; a.L = L; a.Lpad = A.Lpad; a.H = H;
  a.q_scale = (const bf16_t*)A.q_scale; a.q_scale2 = (const bf16_t*)(A.q_scale2 ? A.q_scale2 : A.q_scale);
  a.rope = A.rope; a.rope_bstride = A.rope_bstride; a.split = A.q_scale2 ? A.split : L;
  a.q_pre = A.q_prescaled != 0;
  a.inmerge = 0; a.flags = nullptr;
  a.qblocks = (L + QB - 1) / QB;
  a.items = a.qblocks * H * B;
  a.full_rounds = -1; a.tail_items = 0; a.tail_units = 0; a.part = (float*)scratch;
  static VcOncePerDevice done;
  hipError_t e;
  if (done.need()) {
    e = hipFuncSetAttribute((const void*)attn64_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS64);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)attn64_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS64);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)attn64s_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS64);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)attn64s_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS64);
    if (e != hipSuccess) { snprintf(err, errlen, "attention64 attribute: %s", hipGetErrorString(e)); return VC_ERR_HIP; }
    done.mark();
  }
  // logits bounded by the caller (|x| <= logit_bound in the log2 domain): 2^x, a row's sum over L keys and O stay far inside
  // f32 for bound <= 100, so the softmax needs no running max (kernel header)
  const bool bounded = A.logit_bound > 0.0f && A.logit_bound <= 100.0f;
  // bounded logits + prescaled queries (the product's launches wherever this kernel runs): the stream form
  // (its LDS-DMA addresses are kernel-argument base + 32-bit byte offset)
  const bool fits32 = ((uint64_t)B * (uint64_t)A.bstride + (uint64_t)L * (uint64_t)A.ld + 3u * (uint64_t)H * 128u) * 2u < (1ull << 32) &&
                      (uint64_t)B * (uint64_t)H * 128u * (uint64_t)A.Lpad * 2u < (1ull << 32) && L >= 16;
  const bool stream = a.q_pre && !A.q_scale && fits32;
  void (*kern)(const Attn64Args) = stream ? (bounded ? attn64s_kernel<true> : attn64s_kernel<false>) : bounded ? attn64_kernel<true> : attn64_kernel<false>;
  const int G = n_cu;
  const int nkt = (L + KVB - 1) / KVB;
  // the tail split is scheduled per XCD (Sched64): cut where some XCD has tail items and cutting shortens its critical
  // path by more than the merge costs (~4 tiles): plain = one more round of nkt tiles for the workgroups that draw a tail
  // item, split = ceil(tail * nkt / W) tiles for every workgroup of that XCD
  int any_tail = 0, worst_split = 0;
  if (G % 8 == 0)
    for (int x = 0; x < 8; ++x) {
      const int W = G / 8, q = a.items / 8, r = a.items % 8, n = q + (x < r ? 1 : 0), tail = n % W;
      any_tail |= tail;
      worst_split = std::max(worst_split, (int)(((long)tail * nkt + W - 1) / W));
    }
  if (tail_split && !kv_len && any_tail && scratch && scratch_bytes >= vc_attention64_scratch_bytes_impl(n_cu) && worst_split + 4 < nkt) {
    const bool has_flags = scratch_bytes >= vc_attention_scratch_bytes_impl();       // the whole buffer, flag words at its end
    a.full_rounds = a.items / G; a.tail_items = a.items - a.full_rounds * G; a.tail_units = a.tail_items * nkt;
    // variant bit 16 (stream form only): the pieces are combined inside the launch - the caller vouches that the flag words at
    // the end of the scratch were zero once and that nothing but these launches, one at a time, touches the scratch
    a.inmerge = stream && (A.variant & 16) && has_flags ? 1 : 0;
    a.flags = (uint32_t*)((char*)scratch + vc_attention_flags_offset_impl());
    hipLaunchKernelGGL(kern, dim3(G), dim3(256), LDS64, s, a);
    if (!a.inmerge) {
      // XCD x has (items / 8 [+ 1]) % (G / 8) tail items: 16 blocks (8 XCDs x 2 query blocks) per tail slot that any XCD fills
      const int W = G >> 3, qn = a.items >> 3, rn = a.items & 7;
      const int tail_slots = std::max(rn ? (qn + 1) % W : 0, qn % W);
      hipLaunchKernelGGL(attn64_merge_kernel, dim3(16 * tail_slots), dim3(256), 0, s, a, G);
    }
  } else {
    hipLaunchKernelGGL(kern, dim3(std::min(a.items, G)), dim3(256), LDS64, s, a);
  }
  e = hipGetLastError();
  if (e == hipSuccess) return VC_OK;
  snprintf(err, errlen, "attention64 launch: %s", hipGetErrorString(e));
  return VC_ERR_HIP;
}
