// Joint text+image flash attention for gfx950 (non-causal, head_dim 128, bf16 MFMA, f32 online softmax).
//
// Replaces models/math.py:63-99 (`attention` -> flash_attn_varlen_func, scale 128^-0.5, p_drop 0) for the
// stitched grid sequence cat(txt, img).  q,k are read straight out of the "B L (K H D)" qkv rows the QKV
// GEMM wrote (already QK-normed + RoPE'd in place by vc_qknorm_rope_vt); V comes pre-transposed as
// vt[b][h][d][Lpad] so that both MFMA operands are K-contiguous and no transpose happens in the loop.
//
// Work split: one workgroup = NW waves x 32 queries of one (batch, head); KV tiles of 64 keys stream through
// a double-buffered LDS ring by global_load_lds.  Per wave and KV tile:
//   S^T[u] = K_u . Q^T      2 x 8  v_mfma_f32_32x32x16_bf16   (swapped operands: a lane owns ONE query column,
//                                                              so row max / row sum are lane-local + one xor-32)
//   P = exp2(c*S - m)       f32, online max/sum, O rescaled by exp2(m_old - m_new)
//   O^T += Vt . P^T         4 x 4  v_mfma_f32_32x32x16_bf16
// K rows are fed to the MFMA with bits 2<->3 of the row index swapped, which makes the S^T accumulator
// registers of a lane line up with 8 CONTIGUOUS keys per 16-key step -> P feeds PV with no cross-lane moves.
// LDS images are lane-linear (DMA) with XOR slot swizzles applied on the source address and on the
// ds_read_b128 address (K: slot ^= row&15 on 256-B rows; Vt: slot ^= (row>>1)&7 on 128-B rows).
#include <algorithm>
#include "common.h"
#include "vcloze_internal.h"

namespace {

struct AttnArgs {
  const bf16_t* qkv;
  const bf16_t* vt;
  bf16_t* out;
  const int32_t* kv_len;
  const int32_t* kv_gap;      // optional [B][2]: keys / query rows lo <= i < hi masked too (padded tail of the text stream)
  int64_t ld, bstride, ldo, out_bstride;
  int32_t B, L, Lpad, H, qblocks, items;
  // tail split (variant +4): whole items for `full_rounds` rounds, then the remaining `tail_items` are cut into
  // gridDim.x equal chunks of (item, KV tile) units; a chunk that covers only part of an item leaves (O, m, l)
  // in `part` and attn_merge_kernel combines the pieces.  full_rounds < 0 = off.
  int32_t full_rounds, tail_items, tail_units;
  float* part;
  uint64_t* debug_ts;   // profiling builds only (-DVC_ATTN_TIMESTAMPS)
};

// one partial result: [wave 4][16 groups][lane 64][4] f32 accumulator fragments in register order, then [wave 4][lane 64]
// (m, l) pairs; every store / load is 16 B (8 B) per lane, lane-contiguous
constexpr int PART_O = 4 * 16 * 64 * 4;          // floats
constexpr int PART_FLOATS = PART_O + 4 * 64 * 2;
VC_DEV int chunk_begin(int c, int units, int chunks) { return (int)(((long)c * units) / chunks); }

constexpr int KVB = 64;               // keys per tile
constexpr int K_TILE = KVB * 256;     // bytes
constexpr int V_TILE = 128 * KVB * 2; // bytes
constexpr int STAGE = K_TILE + V_TILE;

VC_DEV int swap23(int i) { return (i & ~12) | ((i & 4) << 1) | ((i & 8) >> 1); }

template <int NW>
__global__ __launch_bounds__(NW * 64, 2) void attn_fwd_kernel(const AttnArgs a) {
  constexpr int NT = NW * 64;
  constexpr int K_IT = (K_TILE / 16) / NT, V_IT = (V_TILE / 16) / NT;
  constexpr int DMA_EVERY = 16 / (K_IT + V_IT);
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lq = lane & 31, hh = lane >> 5;

  // Persistent launch: gridDim.x = CUs x resident blocks; block p takes work items p, p + gridDim.x, ...  With 744
  // items on 512 slots (L=3968) the hardware dispatcher refills BOTH slots of the CUs that finish first (4 items on
  // some CUs, 2 on others); the static assignment gives every CU 3 (slot 0 runs a second item while slot 1 idles).
  const int G = gridDim.x;
  const int nkt_all = (a.L + KVB - 1) / KVB;                      // tail split runs without kv_len (uniform tile count)
  const bool split = a.full_rounds >= 0;
  const int chunk = split ? xcd_remap(blockIdx.x, G) : 0;         // neighbouring chunks (same head) share an XCD's L2
  int tu = split ? chunk_begin(chunk, a.tail_units, G) : 0;
  const int tu_end = split ? chunk_begin(chunk + 1, a.tail_units, G) : 0;
  const int it_first = tu / nkt_all;
  for (int seg = 0;; ++seg) {
  int item, kt0 = 0, kt1 = -1, piece = -1;
  if (!split) {
    item = blockIdx.x + seg * G;
    if (item >= a.items) break;
  } else if (seg < a.full_rounds) {
    item = blockIdx.x + seg * G;
  } else {
    if (tu >= tu_end) break;
    const int it = tu / nkt_all;
    kt0 = tu - it * nkt_all;
    kt1 = min(nkt_all, kt0 + (tu_end - tu));
    tu += kt1 - kt0;
    item = a.full_rounds * G + it;
    if (kt1 - kt0 != nkt_all) piece = chunk * 2 + (it - it_first);
  }
  int id = xcd_remap(item, a.items);
  const int qb = id % a.qblocks;
  const int bh = id / a.qblocks;
  const int h = bh % a.H, b = bh / a.H;
  const int L = a.L;
  const int kvlen = a.kv_len ? a.kv_len[b] : L;
  const int gap_lo = a.kv_gap ? a.kv_gap[2 * b] : 0, gap_hi = a.kv_gap ? a.kv_gap[2 * b + 1] : 0;

  const bf16_t* __restrict__ qbase = a.qkv + (long)b * a.bstride + h * 128;
  const bf16_t* __restrict__ kbase = qbase + a.H * 128;
  const bf16_t* __restrict__ vbase = a.vt + ((long)(b * a.H + h) * 128) * a.Lpad;

  // ---- staging offsets ----
  // 32-bit BYTE offsets from the wave-uniform K / Vt bases (scalar base + vector offset addressing): per tile a K
  // piece costs one add and one min (keys past L - 1 re-read row L - 1; they are masked later), a V piece one add -
  // the 64-bit multiply / add chains these replace were a quarter of the loop's VALU time.
  uint32_t k_off[K_IT], k_max[K_IT], v_off[V_IT];
  const uint32_t k_step = (uint32_t)KVB * (uint32_t)a.ld * 2u;
#pragma unroll
  for (int i = 0; i < K_IT; ++i) {
    const int c = i * NT + tid;
    const int row = c >> 4;
    const uint32_t col = (uint32_t)(((c & 15) ^ (row & 15)) << 4);
    k_off[i] = (uint32_t)row * (uint32_t)a.ld * 2u + col;
    k_max[i] = (uint32_t)(L - 1) * (uint32_t)a.ld * 2u + col;
  }
#pragma unroll
  for (int i = 0; i < V_IT; ++i) {
    const int c = i * NT + tid;
    const int d = c >> 3;
    v_off[i] = ((uint32_t)d * (uint32_t)a.Lpad + ((((c & 7) ^ ((d >> 1) & 7))) << 3)) * 2u;
  }
  const char* kbytes = (const char*)kbase;
  const char* vbytes = (const char*)vbase;
  // one LDS-DMA wave-instruction of tile kt: i < K_IT -> K piece i, else V piece i - K_IT
  auto stage_piece = [&](int buf, int kt, int i) {
    char* sk = smem + buf * STAGE;
    char* sv = sk + K_TILE;
#pragma unroll
    for (int j = 0; j < K_IT; ++j)
      if (i == j) glds16(kbytes + min(k_off[j] + (uint32_t)kt * k_step, k_max[j]), sk + (j * NT + wave * 64) * 16);
#pragma unroll
    for (int j = 0; j < V_IT; ++j)
      if (i == K_IT + j) glds16(vbytes + (v_off[j] + (uint32_t)kt * (KVB * 2)), sv + (j * NT + wave * 64) * 16);
  };
  auto stage = [&](int buf, int kt) {
#pragma unroll
    for (int i = 0; i < K_IT + V_IT; ++i) stage_piece(buf, kt, i);
  };

  const int nkt = (kvlen + KVB - 1) / KVB;
  if (kt1 < 0) kt1 = nkt;
  stage(0, kt0);

  // ---- Q fragments (B operand): lane = query lq, d = t*16 + hh*8 .. +7 ----
  const int q0 = qb * (NW * 32) + wave * 32;
  const int qrow = min(q0 + lq, L - 1);
  bf16x8 qf[8];
#pragma unroll
  for (int t = 0; t < 8; ++t) qf[t] = *(const bf16x8*)(qbase + (long)qrow * a.ld + t * 16 + hh * 8);

  // ---- LDS read offsets ----
  int k_rd[2];
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int row = u * 32 + swap23(lq);
    k_rd[u] = row * 256 + ((hh ^ (row & 15)) << 4);  // slot = 2t + hh  ->  ^ (t*32) per d-step
  }
  int v_rd[4];
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) {
    const int d = dt * 32 + lq;
    v_rd[dt] = K_TILE + d * 128 + ((hh ^ ((d >> 1) & 7)) << 4);  // slot = 2s + hh -> ^ (s*32)
  }

  f32x16 o[4];
#pragma unroll
  for (int dt = 0; dt < 4; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[dt][r] = 0.f;
  // the running max starts at a FINITE floor: a tile whose keys are all masked (a gap that begins at key 0, i.e. a sample
  // without text) then gives m_cand - m_run = 0 and p = 2^(-inf + 1e30) = 0 instead of the NaN of (-inf) - (-inf)
  float m_run = -1e30f, l_run = 0.f;
  const float c_scale = 0.08838834764831845f * 1.4426950408889634f;  // 128^-0.5 * log2(e)

#ifdef VC_ATTN_TIMESTAMPS
  uint64_t* ts = (a.debug_ts && blockIdx.x < 2 && lane == 0 && wave == 0) ? a.debug_ts + blockIdx.x * 4096 : nullptr;
  int tsi = 0;
  auto stamp = [&]() { if (ts && tsi < 4096) ts[tsi++] = __builtin_amdgcn_s_memtime(); };
#else
  auto stamp = [&]() {};
#endif
  __syncthreads();

  for (int kt = kt0; kt < kt1; ++kt) {
    const int cur = (kt - kt0) & 1;
    stamp();
    // Tile kt+1's LDS-DMA pieces (8 per wave with 4 waves, 4 with 8) go out one behind every DMA_EVERY-th QK^T MFMA: an
    // LDS-DMA wave-instruction holds the issue slot for ~75 cycles, and issued back to back at the top of the tile
    // (600 cycles without an MFMA from this wave) they cost 5 % of the kernel.  The last tile re-stages itself into the idle buffer so the body stays free
    // of branches (sched_group_barrier pins issue order only inside one basic block).
    const int ktn = min(kt + 1, kt1 - 1);
    const char* base = smem + cur * STAGE;

    // S^T = K . Q^T.  Fragment reads run one 8-deep batch AHEAD of the MFMAs that consume them (rotating register
    // set): a ds_read_b128 takes ~130+ cycles to return under load vs 32 cycles per MFMA, so reading just-in-time
    // (what the compiler schedules on its own) leaves the matrix pipe waiting on LDS latency 3/4 of the time.
    f32x16 s[2];
    bf16x8 fr8[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) fr8[t] = *(const bf16x8*)(base + (k_rd[0] ^ (t * 32)));
#pragma unroll
    for (int r = 0; r < 16; ++r) { s[0][r] = 0.f; s[1][r] = 0.f; }
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      s[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr8[t], qf[t], s[0], 0, 0, 0);
      fr8[t] = *(const bf16x8*)(base + (k_rd[1] ^ (t * 32)));
#ifndef VC_ATTN_NO_DMA      // analysis builds only (wrong results): the loop without its LDS-DMA
      if (t % DMA_EVERY == 0) stage_piece(cur ^ 1, ktn, t / DMA_EVERY);
#endif
    }
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      s[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr8[t], qf[t], s[1], 0, 0, 0);
      // V fragments of d-tiles 0,1 (k-steps 0..3 each) land while the softmax VALU work runs
      fr8[t] = *(const bf16x8*)(base + (v_rd[t >> 2] ^ ((t & 3) * 32)));
#ifndef VC_ATTN_NO_DMA
      if ((t + 8) % DMA_EVERY == 0) stage_piece(cur ^ 1, ktn, (t + 8) / DMA_EVERY);
#endif
    }
    // pin the issue order (hipcc otherwise sinks every read next to its MFMA): 8 reads, then MFMA/read pairs
    __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
#pragma unroll
    for (int t = 0; t < 16; ++t) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
#ifndef VC_ATTN_NO_DMA
      if (t % DMA_EVERY == 0) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
#endif
    }
    stamp();
    // mask keys beyond kv_len (only the last tile can hold any) and inside the gap (tiles that overlap it)
    if (kt * KVB + KVB > kvlen || (kt * KVB < gap_hi && kt * KVB + KVB > gap_lo)) {
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = kt * KVB + u * 32 + ((r >> 3) << 4) + hh * 8 + (r & 7);
          if (key >= kvlen || (key >= gap_lo && key < gap_hi)) s[u][r] = -INFINITY;
        }
    }
    // online softmax (log2 domain)
    float mx = s[0][0];
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[u][r]);
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    // Deferred rescale: keep the running max (and skip the 64-multiply rescale of O) while no row of this wave grows
    // its max by more than 2^8 - P then stays <= 256, which bf16 (relative precision) and the f32 sums absorb; the
    // branch is wave-uniform.  m_run = -1e30 until the first unmasked key forces the rescale path there.
    const float m_cand = fmaxf(m_run, mx * c_scale);
    float alpha = 1.0f;
    if (!__all(m_cand - m_run <= 8.0f)) {
      alpha = __builtin_amdgcn_exp2f(m_run - m_cand);
      m_run = m_cand;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[dt][r] *= alpha;
    }
    const float m_new = m_run;
    float psum = 0.f;
    bf16x8 pf[4];
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
#ifdef VC_ATTN_NO_EXP        // analysis builds only (wrong results): the loop without its transcendentals
        const float p = s[u][r] * c_scale - m_new;
#else
        const float p = __builtin_amdgcn_exp2f(s[u][r] * c_scale - m_new);
#endif
        psum += p;
        pf[u * 2 + (r >> 3)][r & 7] = (__bf16)p;
      }
    l_run = l_run * alpha + psum;

    stamp();
    // O^T += Vt . P^T  (fr8 holds d-tiles 0,1; each slot is refilled with d-tiles 2,3 right after its MFMA)
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      o[t >> 2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr8[t], pf[t & 3], o[t >> 2], 0, 0, 0);
      fr8[t] = *(const bf16x8*)(base + (v_rd[2 + (t >> 2)] ^ ((t & 3) * 32)));
    }
#pragma unroll
    for (int t = 0; t < 8; ++t)
      o[2 + (t >> 2)] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr8[t], pf[t & 3], o[2 + (t >> 2)], 0, 0, 0);
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 1);
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 1);
    }
    __builtin_amdgcn_sched_group_barrier(0x008, 8, 1);
    stamp();
    __syncthreads();
  }

  // ---- epilogue ----
  const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
  const int q = q0 + lq;
  if (piece >= 0) {          // part of an item's keys only: un-normalised O^T fragments + (m, l), merged by attn_merge_kernel
    float* pp = a.part + (long)piece * PART_FLOATS;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        f32x4 w = {o[dt][g * 4 + 0], o[dt][g * 4 + 1], o[dt][g * 4 + 2], o[dt][g * 4 + 3]};
        *(f32x4*)(pp + ((wave * 16 + dt * 4 + g) * 64 + lane) * 4) = w;
      }
    f32x2 ml = {m_run, l_tot};
    *(f32x2*)(pp + PART_O + (wave * 64 + lane) * 2) = ml;
  } else if (q < L) {
    const float inv = (q < kvlen && !(q >= gap_lo && q < gap_hi)) ? 1.0f / l_tot : 0.0f;  // padded query rows -> 0 (pad_input, math.py:96)
    bf16_t* orow = a.out + (long)b * a.out_bstride + (long)q * a.ldo + h * 128;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        u32x2 w;
        w[0] = pack2bf(o[dt][g * 4 + 0] * inv, o[dt][g * 4 + 1] * inv);
        w[1] = pack2bf(o[dt][g * 4 + 2] * inv, o[dt][g * 4 + 3] * inv);
        *(u32x2*)(orow + dt * 32 + g * 8 + hh * 4) = w;
      }
  }
  }  // work items
}

// Combines the pieces of the tail items that attn_fwd_kernel<4> cut along the keys: O = sum_p 2^(m_p - m) O_p,
// l = sum_p 2^(m_p - m) l_p, out = bf16(O / l).  One workgroup per tail item, thread layout = the writer's.
__global__ __launch_bounds__(256) void attn_merge_kernel(const AttnArgs a, int G) {
  const int it = blockIdx.x;
  const int nkt = (a.L + KVB - 1) / KVB;
  const int u0 = it * nkt, u1 = u0 + nkt;
  int c = (int)(((long)u0 * G) / a.tail_units);
  while (c > 0 && chunk_begin(c, a.tail_units, G) > u0) --c;
  while (c + 1 < G && chunk_begin(c + 1, a.tail_units, G) <= u0) ++c;
  if (chunk_begin(c + 1, a.tail_units, G) >= u1) return;        // the whole item ran inside one chunk: already written
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lq = lane & 31, hh = lane >> 5;
  float m = -INFINITY;
  for (int cc = c; cc < G && chunk_begin(cc, a.tail_units, G) < u1; ++cc) {
    if (chunk_begin(cc + 1, a.tail_units, G) == chunk_begin(cc, a.tail_units, G)) continue;   // empty chunk (fewer units than blocks)
    const int piece = cc * 2 + (it - chunk_begin(cc, a.tail_units, G) / nkt);
    m = fmaxf(m, a.part[(long)piece * PART_FLOATS + PART_O + (wave * 64 + lane) * 2]);
  }
  f32x4 acc[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  float l = 0.f;
  for (int cc = c; cc < G && chunk_begin(cc, a.tail_units, G) < u1; ++cc) {
    if (chunk_begin(cc + 1, a.tail_units, G) == chunk_begin(cc, a.tail_units, G)) continue;
    const int piece = cc * 2 + (it - chunk_begin(cc, a.tail_units, G) / nkt);
    const float* pp = a.part + (long)piece * PART_FLOATS;
    const f32x2 ml = *(const f32x2*)(pp + PART_O + (wave * 64 + lane) * 2);
    const float sc = __builtin_amdgcn_exp2f(ml[0] - m);
    l += ml[1] * sc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] += *(const f32x4*)(pp + ((wave * 16 + i) * 64 + lane) * 4) * sc;
  }
  const int item = a.full_rounds * G + it;
  const int id = xcd_remap(item, a.items);
  const int qb = id % a.qblocks, bh = id / a.qblocks;
  const int h = bh % a.H, b = bh / a.H;
  const int q = qb * 128 + wave * 32 + lq;
  if (q < a.L) {
    const float inv = 1.0f / l;
    bf16_t* orow = a.out + (long)b * a.out_bstride + (long)q * a.ldo + h * 128;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      u32x2 w;
      w[0] = pack2bf(acc[i][0] * inv, acc[i][1] * inv);
      w[1] = pack2bf(acc[i][2] * inv, acc[i][3] * inv);
      *(u32x2*)(orow + (i >> 2) * 32 + (i & 3) * 8 + hh * 4) = w;
    }
  }
}

}  // namespace


static uint64_t* g_attn_debug_ts = nullptr;
#ifdef VC_ATTN_TIMESTAMPS   // profiling builds only (`make debug`, tools/attn_ts.py): the product library exports the header's symbols and nothing else
extern "C" void vc_debug_set_attn_ts(void* p) { g_attn_debug_ts = (uint64_t*)p; }
#endif

static int attn_cu_count() { return vc_cu_count(); }

// [partials of whichever variant runs: the larger of the two layouts][flag words of attention64's in-launch combine, zero between
// launches (VcAttention.variant bit 16): behind everything any other variant writes]
int64_t vc_attention_flags_offset_impl() {
  const int64_t parts = std::max((int64_t)2 * attn_cu_count() * 2 * PART_FLOATS * (int64_t)sizeof(float), vc_attention64_scratch_bytes_impl(attn_cu_count()));
  return (parts + 255) & ~(int64_t)255;
}
int64_t vc_attention_scratch_bytes_impl() {
  return vc_attention_flags_offset_impl() + ((vc_attention64_flags_bytes_impl(attn_cu_count()) + 255) & ~(int64_t)255);
}

int vc_attention_launch(const VcAttention& A, hipStream_t s, char* err, int errlen) {
  const void* qkv = A.qkv; const void* vt = A.vt; void* out = A.out; const int32_t* kv_len = A.kv_len;
  const int64_t ld = A.ld, bstride = A.bstride, ldo = A.ldo, out_bstride = A.out_bstride;
  const int32_t B = A.B, L = A.L, Lpad = A.Lpad, H = A.H;
  int32_t variant = A.variant;
  void* scratch = A.scratch; const int64_t scratch_bytes = A.scratch_bytes;
  if (!qkv || !vt || !out) { snprintf(err, errlen, "attention: null pointer"); return VC_ERR_ARG; }
  if (B <= 0 || L <= 0 || H <= 0) { snprintf(err, errlen, "attention: empty problem B=%d L=%d H=%d", B, L, H); return VC_ERR_ARG; }
  if (Lpad < L || Lpad % KVB) { snprintf(err, errlen, "attention: Lpad=%d must be a multiple of %d and >= L=%d", Lpad, KVB, L); return VC_ERR_ARG; }
  if (ld % 8 || ldo % 4 || bstride % 8) { snprintf(err, errlen, "attention: strides must keep 16-B row alignment"); return VC_ERR_ARG; }
  if ((uint64_t)128 * (uint64_t)Lpad >= (1ull << 31)) { snprintf(err, errlen, "attention: Lpad too large"); return VC_ERR_ARG; }
  if ((uint64_t)(Lpad + KVB) * (uint64_t)ld * 2ull >= (1ull << 32)) { snprintf(err, errlen, "attention: one sample's K rows exceed 32-bit byte offsets (L=%d ld=%ld)", L, (long)ld); return VC_ERR_ARG; }
  if (A.q_scale && !(variant & 8)) { snprintf(err, errlen, "attention: in-kernel QKNorm + RoPE of the queries (q_scale) exists for variants 8 / 12 only"); return VC_ERR_ARG; }
  if (A.kv_gap && !kv_len) { snprintf(err, errlen, "attention: kv_gap needs kv_len"); return VC_ERR_ARG; }
  if (A.q_scale && !A.rope) { snprintf(err, errlen, "attention: q_scale given without a rope table"); return VC_ERR_ARG; }
  if (A.q_prescaled && (!(variant & 8) || A.q_scale)) { snprintf(err, errlen, "attention: q_prescaled exists for variants 8 / 12 and excludes q_scale"); return VC_ERR_ARG; }
  AttnArgs a;
  a.qkv = (const bf16_t*)qkv; a.vt = (const bf16_t*)vt; a.out = (bf16_t*)out; a.kv_len = kv_len;
  a.kv_gap = kv_len ? A.kv_gap : nullptr;
  a.ld = ld; a.bstride = bstride; a.ldo = ldo; a.out_bstride = out_bstride;
  a.B = B; a.L = L; a.Lpad = Lpad; a.H = H;
  if (variant & 8)    // one wave per SIMD, 64 queries per wave (attention64.hip); +4 = tail split
    return vc_attention64_launch(A, (variant & 4) != 0, attn_cu_count(), g_attn_debug_ts, s, err, errlen);
  a.debug_ts = g_attn_debug_ts;
  a.full_rounds = -1; a.tail_items = 0; a.tail_units = 0; a.part = (float*)scratch;
  const int lds = 2 * STAGE;
  hipError_t e;
  const bool persist = (variant & 2) != 0;   // +2: persistent grid with static item assignment (variants 2, 3)
  const bool tail_split = (variant & 4) != 0; // +4 (with 1 and 2 = variant 7): tail items cut along the keys
  if (tail_split && (variant & 3) != 3) { snprintf(err, errlen, "attention: the tail split (+4) exists for variant 3 only"); return VC_ERR_ARG; }
  variant &= 1;
  const int n_cu = attn_cu_count();
  if (variant == 1) {  // 4 waves x 32 queries
    a.qblocks = (L + 127) / 128;
    static VcOncePerDevice done;
    if (done.need()) { e = hipFuncSetAttribute((const void*)attn_fwd_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, lds); if (e != hipSuccess) goto fail; done.mark(); }
    a.items = a.qblocks * H * B;
    const int G = 2 * n_cu;
    const int nkt = (L + KVB - 1) / KVB;
    const int rounds = a.items / G, tail = a.items - rounds * G;
    // cut the tail only where it shortens the critical path by more than the merge costs (~3 tiles): plain = one more
    // round of nkt tiles for the blocks that draw a tail item, split = ceil(tail * nkt / G) tiles for every block
    const int split_tiles = (int)(((long)tail * nkt + G - 1) / G);
    if (tail_split && !kv_len && tail > 0 && scratch && scratch_bytes >= vc_attention_scratch_bytes_impl() && split_tiles + 3 < nkt) {
      a.full_rounds = rounds; a.tail_items = tail; a.tail_units = tail * nkt;
      hipLaunchKernelGGL(attn_fwd_kernel<4>, dim3(G), dim3(256), lds, s, a);
      hipLaunchKernelGGL(attn_merge_kernel, dim3(tail), dim3(256), 0, s, a, G);
    } else {
      hipLaunchKernelGGL(attn_fwd_kernel<4>, dim3(std::min(a.items, persist ? G : a.items)), dim3(256), lds, s, a);
    }
  } else {  // 8 waves x 32 queries
    a.qblocks = (L + 255) / 256;
    static VcOncePerDevice done8;
    if (done8.need()) { e = hipFuncSetAttribute((const void*)attn_fwd_kernel<8>, hipFuncAttributeMaxDynamicSharedMemorySize, lds); if (e != hipSuccess) goto fail; done8.mark(); }
    a.items = a.qblocks * H * B;
    hipLaunchKernelGGL(attn_fwd_kernel<8>, dim3(std::min(a.items, persist ? n_cu : a.items)), dim3(512), lds, s, a);
  }
  e = hipGetLastError();
  if (e == hipSuccess) return VC_OK;
fail:
  snprintf(err, errlen, "attention launch: %s", hipGetErrorString(e));
  return VC_ERR_HIP;
}
