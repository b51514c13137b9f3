/* vcloze_hip.h — C ABI of the MI355X (gfx950) denoising-path kernels for VisualCloze.
 *
 * The reference (lzyhha/VisualCloze) has no FFI layer: its hot path is Python calling torch /
 * flash-attn.  This header is the boundary a maintainer would bind instead (ctypes stub in
 * INTEGRATION.md).  Every entry point names the reference interface it replaces.
 *
 * Conventions: plain pointers are DEVICE pointers (HBM) unless marked host; bf16 = raw uint16;
 * `stream` is a hipStream_t passed as void* (NULL = default stream).  Return 0 on success,
 * negative VC_ERR_* otherwise; vc_last_error() returns the message for the calling thread.
 * Not thread-safe per stream/graph handle.  No ownership is ever transferred.
 */
#ifndef VCLOZE_HIP_H
#define VCLOZE_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define VC_OK 0
#define VC_ERR_ARG (-1)
#define VC_ERR_HIP (-2)
#define VC_ERR_STATE (-3)

#define VC_ABI_VERSION 10
int vc_abi_version(void);
const char* vc_last_error(void);
/* sizeof(VcGemmProblem), sizeof(VcGemmArgs), sizeof(VcLnStream), sizeof(VcAttention), sizeof(VcFluxConfig),
 * sizeof(VcFluxInputs), sizeof(VcFluxLaunchClass) (ABI 10: SEVEN entries) as this library was compiled: a binding checks them
 * against its own mirrors at load time, so a stale .so cannot silently disagree with the caller. */
void vc_struct_sizes(int32_t out[7]);
/* number of visible devices / name of device 0 ("" when none) — fails loudly, never falls back */
int vc_device_count(void);
int vc_device_info(int dev, char* name, int namelen, int* cu_count, int64_t* hbm_bytes);

/* ---- GEMM epilogues ---- */
#define VC_EPI_BIAS 0      /* y = bf16(acc + b)                         nn.Linear              */
#define VC_EPI_GELU 1      /* y = bf16(gelu_tanh(bf16(acc + b)))        layers.py:141-145,229  */
#define VC_EPI_GATE_RES 2  /* y = bf16(res + bf16(gate*bf16(acc + b)))  layers.py:190-195,245  */
#define VC_EPI_SILU 3      /* y = bf16(silu(bf16(acc + b)))             layers.py:55-60        */
#define VC_EPI_QKV 4       /* BIAS, and columns >= vt_col0 go TRANSPOSED to vt instead of C:        */
                           /* the "K H D" split of layers.py:166,236 with V laid out key-contiguous */
                           /* for the attention kernel (what vc_qknorm_rope_vt's VC_QKN_VT part does)*/

typedef struct VcGemmProblem {
  const void* A;    /* [M,K] bf16, row stride lda (elements) */
  const void* W;    /* [N,K] bf16, row stride ldw (nn.Linear.weight layout; rows may be padded) */
  const void* bias; /* [N] bf16 or NULL */
  void* C;          /* [M,N] bf16, row stride ldc */
  const void* res;  /* GATE_RES: residual [M,N] bf16 (may alias C), row stride ldres */
  const void* gate; /* GATE_RES: gate vector(s) bf16; row b of batch uses gate + b*gate_bstride */
  int64_t lda, ldw, ldc, ldres, gate_bstride;
  /* batch-strided rows (samples of a per-GPU batch interleave text and image rows): with a_rpb > 0 row m of A lives at
   * (m / a_rpb) * a_bstride + (m % a_rpb) * lda; likewise C and res with c_rpb / c_bstride (res must share C's layout).
   * 0 = plain rows at m * ld. */
  int64_t a_bstride, c_bstride;
  int32_t M, N, K;
  int32_t rows_per_batch; /* GATE_RES: batch index of row m is m / rows_per_batch */
  int32_t a_rpb, c_rpb;
  int32_t tiles_m, tiles_n, tile_start; /* filled by the launcher */
  int32_t m_begin;                      /* filled by the launcher (first row of a split launch, see vc_gemm); pass 0 */
  /* VC_EPI_QKV (vt may be NULL = plain BIAS): element (m, n >= vt_col0) is written to
   * vt[(m / vt_rpb) * vt_bstride + (n - vt_col0) * vt_lpad + vt_row0 + m % vt_rpb], i.e. vt is [batch][N - vt_col0][vt_lpad]
   * (= [B][H][128][Lpad] of vc_attention), vt_rpb the rows of this problem per batch element and vt_row0 where they start
   * in the joint sequence (text rows 0, image rows T).  Fast when vt_col0 is a multiple of the tile width the launcher
   * picks (2 * 3072 is, for every tile) and vt_rpb, vt_row0, vt_lpad are multiples of 8; correct otherwise. */
  void* vt;
  int64_t vt_bstride;
  int32_t vt_col0, vt_rpb, vt_row0, vt_lpad;
  /* VC_EPI_QKV, kn_heads = H > 0 (N = 3 * 128 * H): W's rows / bias arrive HEAD-PERMUTED so that every 192-column tile holds one
   * whole query or key head and half a value head - permuted column p is logical column (t = p / 192, j = p % 192)
   *     j < 128:   128 t + j             (t < H: query head t; t >= H: key head t - H)
   *     else:      256 H + 64 t + j - 128 (V columns 64 t .. 64 t + 63; vt_col0 = 256 H)
   * and C is written at the logical columns ("B L (K H D)", layers.py:166,236) by any tile shape (only the 256x192 tile writes V^T
   * in whole 16-B runs).  With kn_scale and / or qn_scale != NULL (forces the 256x192 tile) the epilogue ALSO applies QKNorm with
   * kn_scale / qn_scale [128] bf16 and RoPE from kn_rope ([B?][L][64][2] f32, row vt_row0 + m % vt_rpb of batch element
   * m / vt_rpb) to every key / query head before the row leaves (layers.py:63-84, math.py:112-117): bit-identical to
   * vc_qknorm_rope_vt(parts = VC_QKN_K / VC_QKN_Q) over the plain output, which is then not needed - the "QKV + RoPE fused
   * projection".  qn_prescale != 0: the query heads leave multiplied by 128^-0.5 * log2(e) before their one rounding
   * (= parts | VC_QKN_QPRE), the form VcAttention.q_prescaled reads. */
  const void* kn_scale;
  const float* kn_rope;
  int64_t kn_rope_bstride;
  int32_t kn_heads, qn_prescale;
  const void* qn_scale;
  /* VcGemmArgs.batch = Z > 1: Z independent GEMMs of this shape in ONE launch (blockIdx.y) - instance z reads A + z * a_zstride,
   * W + z * w_zstride and writes C + z * c_zstride (elements, multiples of 8): the per-head S = Q K^T and O = P V products of an
   * attention whose head_dim the fused kernel does not cover (T5: 64 heads x d_kv 64; CLIP) as two launches per layer. */
  int64_t a_zstride, w_zstride, c_zstride;
} VcGemmProblem;

#define VC_GEMM_MAX_PROBLEMS 4
typedef struct VcGemmArgs {
  VcGemmProblem p[VC_GEMM_MAX_PROBLEMS];
  int32_t nprob; /* 1..4 problems in one grid (img+txt streams of up to two samples) */
  int32_t epi;
  const int32_t* step_ptr;  /* optional device step counter: gate += *step_ptr * gate_step_stride */
  int64_t gate_step_stride;
  uint64_t* debug_ts;       /* NULL in production; profiling: per-segment s_memtime stamps of block 0 (tools/) */
  /* Split-K remainder (ABI 7).  splitk_ws: optional f32 scratch of splitk_ws_bytes >= VC_GEMM_SPLITK_WS_BYTES in device
   * memory, NULL = never split.  With it, an auto-tiled call (tile_cfg 0) whose 256x192 tiles are R whole rounds of the CUs
   * plus a remainder of r tiles may run the remainder as r * S work items of K / S each (S <= 8, r * S <= CUs), which leave
   * f32 partial tiles in the scratch; a second, HBM-bound launch sums the S partials of each tile in a fixed order and applies
   * the epilogue (bias, GELU / SiLU / gate + residual) - bit-reproducible (static assignment), equal to the one-pass kernel up
   * to f32 summation order.  Not for VC_EPI_QKV.  The scratch belongs to ONE stream at a time: launches that share it must be
   * ordered (one stream, one linear graph) - a reduce launch of one call and the slices of the next would race otherwise.
   * sk_*: filled by the launcher. */
  void* splitk_ws;
  int64_t splitk_ws_bytes;
  int32_t sk_full, sk_rem, sk_S;
  int32_t batch;            /* 0 / 1: one GEMM per problem; Z > 1: see VcGemmProblem.a_zstride (VC_EPI_BIAS, the 128x128 tile) */
  /* ABI 8, filled by the launcher: > 0 = the STREAM form of the remainder - where it is more than half a round of tiles (no
   * S >= 2 fits), sk_stream work items (one per CU) share the remainder's K-iterations evenly, tile-major: item p owns
   * iterations [p I / n, (p + 1) I / n) of I = sk_rem * K / 64, i.e. at most two segments (end of one tile, start of the next)
   * with one f32 partial tile each (slot 2 p + segment); the reduce launch sums a tile's pieces in K order.  Static: bit-reproducible. */
  int32_t sk_stream, sk_pad_;
} VcGemmArgs;
#define VC_GEMM_SPLITK_WS_BYTES (2LL * 256 * 256 * 192 * 4)   /* two 256x192 f32 partial tiles per work item of one round of 256 CUs */

/* Replaces torch.nn.functional.linear (+ fused neighbours) on the hot path.
 * tile_cfg: 0 = chosen by the launcher's cost model (what the product path passes); a fixed tile for tests and A/B runs:
 * 1 = 128x128, 2 = 256x128, 3 = 256x256, 4 = 256x192, 5 = 256x288, +16 = ping-pong main loop (3, 4, 5),
 * +32 = ping-pong with loader waves (2, 4).
 * With tile_cfg 0 the call may become TWO launches on `stream`: when the 256x192 tiles of the problems are far from a whole
 * number of rounds of the 256 CUs, the rows of problem 0 are cut at a multiple of 256 so that the first launch is exact
 * rounds and the remainder runs on the tile shape that suits it (block-round quantisation: e.g. M = 6656, N = 3072 is
 * 416 tiles = 2 rounds at 81 % fill, or 256 tiles + 240 narrower ones).  Results do not depend on the cut.
 * VC_GEMM_NO_SPLIT (64) as tile_cfg keeps it one launch; (k << 8), k <= 255, forces the cut at row k * 256 (tests).
 * VC_GEMM_SPLITK(S) (S << 16, 2 <= S <= 8; tests and A/B runs) forces the 256x192 loader-wave tile with the tiles beyond the
 * last whole round of the CUs (all tiles when there is no whole round) cut S ways along K (needs splitk_ws);
 * VC_GEMM_NO_SPLITK keeps an auto-tiled call from doing so. */
#define VC_GEMM_NO_SPLIT 64
#define VC_GEMM_SPLITK(S) ((S) << 16)
#define VC_GEMM_NO_SPLITK (1 << 20)
#define VC_GEMM_STREAMK (1 << 21)   /* tests / A-B: force the 256x192 loader-wave tile with the remainder tiles in the STREAM form */
#define VC_GEMM_PREFER_STREAMK (1 << 22)  /* A-B: an auto-tiled call takes the stream form wherever it is eligible, whatever the cost model says */
#define VC_GEMM_STREAMK_ANY_K (1 << 23)   /* A-B: ... also below the K the launcher offers it from */
/* VC_GEMM_PERSIST (128) added to tile_cfg: a launch with more tiles than CUs on a loader-wave tile runs as ONE persistent
 * workgroup per CU that walks the tiles of its XCD's strip and fetches the next tile's first operands during the current
 * tile's epilogue (same results bit for bit; not for VC_EPI_GATE_RES).  Opt-in: measured neutral on MI355X (+0.05 % steps/s;
 * the time it removes is idle time the power-limited clock already gives back), kept for boards where it is not. */
#define VC_GEMM_PERSIST 128
int vc_gemm(const VcGemmArgs* args, int tile_cfg, void* stream);
/* The plan vc_gemm would execute for these arguments, without launching anything (works without a GPU): out[0] = first row
 * of the second launch (0 = a single launch), out[1], out[2] = tile number (1..5 as above) and main-loop form (0 plain,
 * 1 ping-pong, 2 loader waves) of the first or only launch, out[3], out[4] = of the second, out[5] = tiles of both,
 * out[6] = split-K factor S of the remainder tiles (0 = none; -n = the stream form with n work items), out[7] = how many tiles
 * are split. */
int vc_gemm_plan(const VcGemmArgs* args, int tile_cfg, int32_t out[8]);

/* LayerNorm(eps=1e-6, no affine) + AdaLN modulate: y = bf16((1+scale)*LN(x) + shift).
 * Replaces layers.py:163-164,191,195,234 / 257.  x,y: [rows, D] bf16 (row strides ldx/ldy);
 * shift/scale: bf16 vectors of D, batch b at +b*mod_bstride, step s at +s*mod_step_stride. */
int vc_ln_modulate(const void* x, int64_t ldx, void* y, int64_t ldy, const void* shift, const void* scale,
                   int64_t mod_bstride, int32_t rows, int32_t D, int32_t rows_per_batch,
                   const int32_t* step_ptr, int64_t mod_step_stride, void* stream);

/* The same op over two row sets in ONE launch - the img and txt streams of a DoubleStreamBlock, which normalise
 * different tensors with different modulation rows (layers.py:163-164 / 175-176 and 191 / 195).  b may be NULL. */
typedef struct VcLnStream {
  const void* x; int64_t ldx;        /* [rows, D] bf16, row stride ldx */
  void* y; int64_t ldy;
  const void* shift; const void* scale;
  int32_t rows; int32_t rows_per_batch;
} VcLnStream;
int vc_ln_modulate2(const VcLnStream* a, const VcLnStream* b, int64_t mod_bstride, int32_t D, const int32_t* step_ptr,
                    int64_t mod_step_stride, void* stream);

/* QK-RMSNorm (layers.py:63-84) + RoPE (math.py:112-117) in place on q,k, and V transposed to
 * vt[b][h][d][Lpad] for the attention kernel.  qkv: token rows of stride ld (elements) holding
 * q | k | v at column offsets 0, H*128, 2*H*128 ("B L (K H D)", layers.py:166).
 * rope: [B?][L][64][2] f32 (cos, sin) per pair; rope_bstride 0 = shared by the batch.
 * Token rows < split use (q_scale, k_scale), rows >= split use (q_scale2, k_scale2): the text and image
 * streams of a DoubleStreamBlock own separate QKNorm scales (layers.py:167,174); NULL scale2 = one set.
 * parts: bit 0 = the q rows, bit 1 = the k rows, bit 2 = V -> vt; 7 = everything.  6 when the attention call
 * normalises the queries itself (VcAttention.q_scale).  Bit 3 (VC_QKN_QPRE, with bit 0): the softmax scale is folded into q. */
#define VC_QKN_Q 1
#define VC_QKN_K 2
#define VC_QKN_VT 4
#define VC_QKN_QPRE 8   /* with VC_QKN_Q: the query rows leave multiplied by 128^-0.5 * log2(e) (VcAttention.q_prescaled) */
int vc_qknorm_rope_vt(void* qkv, int64_t ld, int64_t bstride, const void* q_scale, const void* k_scale,
                      const void* q_scale2, const void* k_scale2, int32_t split, const float* rope,
                      int64_t rope_bstride, void* vt, int32_t B, int32_t L, int32_t Lpad, int32_t H, int32_t parts,
                      void* stream);

/* Joint text+image attention, non-causal, D=128, softmax scale 128^-0.5 (math.py:63-99 /
 * flash_attn_varlen_func).  q,k from the qkv rows above; vt from vc_qknorm_rope_vt.
 * kv_len[b] (host-visible semantics: keys >= kv_len masked, query rows >= kv_len written as 0,
 * = pad_input of math.py:96); NULL = all L.  kv_gap (optional, with kv_len): int32 [B][2] = (lo, hi): keys and query rows
 * lo <= i < hi are masked as well - the padded tail of the TEXT stream in the joint (txt, img) order; with a valid-first
 * permutation of each stream on the host this covers arbitrary masks (math.py:9-60).  out: [B, L, H*128] bf16, row stride ldo.
 * variant: 0 = 8 waves x 32 queries per workgroup, 1 = 4 waves x 32 queries (two workgroups per CU); +2 = the same
 * kernel on a persistent grid (one workgroup per resident slot, work items assigned statically); variants 0-3 produce
 * bit-identical results.  8 = ONE WAVE PER SIMD, 4 waves x 64 queries, software-pipelined inside the wave
 * (attention64.hip), persistent; q * 128^-0.5 log2(e) is rounded to bf16 when the queries are loaded.
 * +4 (7, 12) = TAIL SPLIT: the items beyond the last full round of resident workgroups are cut along the keys into one
 * equal chunk per workgroup; partial (O, m, l) go to `scratch` (>= vc_attention_scratch_bytes(), device memory,
 * contents undefined afterwards) and a second kernel on the same stream merges them.  One scratch serves one stream at a time.  Same softmax, different f32
 * summation order for those rows.  The launcher drops the split when scratch is NULL / too small, kv_len is given, or
 * it would not shorten the critical path.
 * +16 (28; ABI 9): NO second kernel - where the kernel's stream form runs (q_prescaled and logit_bound <= 100: the product's
 * launches) each workgroup does its share of the tail FIRST, publishes its pieces (agent-scope stores + one flag word per
 * piece and query block, at the end of `scratch`) and combines the pieces assigned to it at the very end of its own work -
 * same arithmetic and order as the merge kernel (bit-identical).  CONTRACT of bit 16: `scratch` is a whole
 * vc_attention_scratch_bytes() buffer whose LAST 16 * CUs bytes (rounded up to 256: the flag words, behind the partials of
 * every variant) were ZERO before the first launch; launches that use a scratch are ordered on one stream (every launch leaves
 * the flag words zero); all workgroups of the grid (one per CU) must be able to run concurrently.  A smaller scratch, or any
 * other kernel form, ignores the bit.  28 is the default of the host engines.
 * q_scale != NULL (variants 8, 12 only): the q columns of qkv hold the RAW projection output and QKNorm + RoPE
 * (layers.py:75-84, math.py:112-117) are applied to the 64 query rows a wave loads, with q_scale / q_scale2 / split /
 * rope / rope_bstride as in vc_qknorm_rope_vt (which is then called with parts = VC_QKN_K | VC_QKN_VT).
 * q_prescaled != 0 (variants 8, 12 only; excludes q_scale): the q columns already hold QK-normed, rotated queries TIMES
 * 128^-0.5 * log2(e), rounded once (VcGemmProblem.qn_prescale / VC_QKN_QPRE): the kernel loads them straight into its MFMA
 * operand registers - no per-item prologue arithmetic. */
typedef struct VcAttention {
  const void* qkv; int64_t ld, bstride;
  const void* vt; void* out; int64_t ldo, out_bstride;
  const int32_t* kv_len;
  int32_t B, L, Lpad, H, variant, split;
  void* scratch; int64_t scratch_bytes;
  const void* q_scale; const void* q_scale2; const float* rope; int64_t rope_bstride;
  const int32_t* kv_gap;
  /* > 0 (variants 8 / 12): the caller guarantees |q.k| * 128^-0.5 * log2(e) <= logit_bound for every query / key pair - with
   * QK-normed operands (layers.py:75-84) |q|, |k| <= sqrt(128) * max|scale|, i.e. logit_bound = 16.33 * max|query_norm.scale|
   * * max|key_norm.scale| is a property of the model's weights.  For logit_bound <= 100 the kernel then runs its softmax
   * with a fixed reference point 0 (no running max: same function, 4 of 68 MFMAs and the row-max VALU work per tile
   * saved); 0 = unknown: online softmax with running max. */
  float logit_bound; int32_t q_prescaled;
} VcAttention;
int vc_attention(const VcAttention* a, void* stream);
int64_t vc_attention_scratch_bytes(void);

/* timestep_embedding (layers.py:28-49): out[b, 0:half]=cos(1000*t*f), [half:]=sin, f host table. */
int vc_timestep_embedding(const float* t, const float* freqs, void* out_bf16, int32_t n, int32_t half,
                          int32_t round_t_bf16, void* stream);
/* elementwise helpers on bf16 vectors */
int vc_silu(const void* x, void* y, int64_t n, void* stream);
/* the two epilogues of vc_gemm as stand-alone passes, for the un-merged LoRA execution mode (LinearLora.forward,
 * models/modules/lora.py:92-98: base_out + lora_B(lora_A(x)) * scale is only complete after a second GEMM):
 * act2d: y[m,n] = bf16(act(x[m,n])) on row views, act 0 = GELU(tanh) (layers.py:141-145,229), 1 = SiLU (layers.py:55-60);
 * gate_residual: out[m,n] = bf16(res[m,n] + bf16(gate[n] * y[m,n])), gate advanced by *step_ptr * gate_step_stride
 * like VcGemmArgs (layers.py:190-195,245). */
int vc_act2d(const void* x, int64_t ldx, void* y, int64_t ldy, int32_t rows, int32_t cols, int32_t act, void* stream);
int vc_gate_residual(const void* y, int64_t ldy, const void* res, int64_t ldres, const void* gate, void* out, int64_t ldo,
                     int32_t rows, int32_t cols, const int32_t* step_ptr, int64_t gate_step_stride, void* stream);
/* y[i] = bf16(bf16(a[i] + b[i % bn]) + c[i % cn]); c may be NULL (model.py:102-107: vec = time + guidance + vector) */
int vc_add3(const void* a, const void* b, const void* c, void* y, int64_t n, int64_t bn, int64_t cn, void* stream);
/* device-to-device copy on `stream` (captured as a memcpy node): restores the step-invariant txt rows */
int vc_copy(void* dst, const void* src, int64_t bytes, void* stream);
/* x||cond -> [rows, cx+cc] (transport.py:195) */
int vc_concat_cols(const void* x, int32_t cx, const void* cond, int32_t cc, void* out, int64_t rows, void* stream);
/* Euler update of the fixed-grid solver: x = bf16(x + bf16(bf16(dt) * (-v))), dt = dts[*step_ptr] (f32 table;
 * torch casts the 0-dim f32 dt to the bf16 common dtype before the multiply, transport/integrators.py:119). */
int vc_euler_step(void* x, const void* v, const float* dts, const int32_t* step_ptr, int64_t n, void* stream);
/* The same update for an f32 ODE state (the solver keeps the caller's state dtype, integrators.py:119): x32 = x32 +
 * f32(bf16(bf16(dt) * (-v))) - torch's promotion of f32 + (0-dim f32 * bf16) - and shadow = bf16(x32), what img_in's Linear
 * reads under autocast.  v == NULL: refresh the shadow only. */
int vc_euler_step_f32(float* x32, void* shadow, const void* v, const float* dts, const int32_t* step_ptr, int64_t n, void* stream);
int vc_step_advance(int32_t* step_ptr, void* stream);
/* SDEdit start state x0 = noise*(1-s) + latent*s with the reference's bf16 roundings (visualcloze.py:221) */
int vc_sdedit_mix(const void* noise, const void* latent, float strength, void* out, int64_t n, void* stream);

/* ---- latent-grid packer / unpacker (the steps either side of the loop) ----
 * pack:   latent [C,h,w] bf16 -> tokens[(h/2)(w/2)][col0 .. col0+4C) of rows with stride ld
 *         ("c (h ph) (w pw) -> (h w) (c ph pw)", models/sampling.py:61, visualcloze.py:208-209,385-386)
 * mask:   pixel mask [H,W] bf16 -> [(H/16)(W/16)][col0 .. col0+256)  (8x8 unshuffle + 2x2 pack, visualcloze.py:381-382)
 * unpack: the inverse of pack (visualcloze.py:237,428) */
int vc_pack_latent(const void* latent, void* tokens, int32_t C, int32_t h, int32_t w, int64_t ld, int32_t col0, void* stream);
int vc_pack_mask(const void* mask, void* tokens, int32_t H, int32_t W, int64_t ld, int32_t col0, void* stream);
int vc_unpack_latent(const void* tokens, int64_t ld, int32_t col0, void* latent, int32_t C, int32_t h, int32_t w, void* stream);

/* ---- VAE decoder glue (reference twin models/modules/autoencoder.py): activations are NHWC bf16
 * [H*W, C], C % 8 == 0; the 3x3 convolutions and 1x1 projections themselves are vc_gemm over these buffers. ----
 * im2col3x3:   src [Hs*Ws, C] -> dst [H*W, 9*C], column (dy*3+dx)*C + c of row (y,x) = src(y+dy-1, x+dx-1), zero outside
 *              (nn.Conv2d(k=3, padding=1), autoencoder.py:64,66,101,126,157,209,235); mode 1 reads src at (y>>1, x>>1),
 *              i.e. F.interpolate(scale_factor=2, mode="nearest") folded into the gather (Upsample.forward, :103-106);
 *              mode 2 reads src [2H*2W, C] at (2y+dy, 2x+dx), zero beyond the edge = pad (0,1,0,1) + stride-2 conv
 *              (Downsample.forward, :91-95).
 * groupnorm:   nn.GroupNorm(G, C, eps, affine) over the whole [HW, C] map, f32 statistics (two-level, deterministic),
 *              y = bf16(.), then bf16(y*sigmoid(y)) if swish (autoencoder.py:21-22,30,63,65,234; :70-77,257-258).
 *              scratch: >= (ceil(HW/128) + 1) * 2 * G floats of device memory.
 * softmax_rows: x[r, 0:cols] <- bf16(softmax(v)) in place, v = bf16(scale * x[r, :]) (+ bias[r, :], rounded to bf16 again),
 *              f32 internal, cols <= 16384; bias NULL or bf16 [rows, cols] (T5 relative position bias); causal_period
 *              P > 0 masks columns j > r % P (CLIP text)   (scaled_dot_product_attention of AttnBlock.attention, :47;
 *              T5Attention / CLIPAttention of transformers).
 * transpose:   dst[c, r] = src[r, c] (V^T operand of the P.V GEMM).
 * nchw_to_nhwc: dst[p, c] = bf16(src[c, p] / div + add), zero for C <= c < Cp  (decode: z / scale_factor + shift_factor,
 *              :306-307); src f32 or bf16.   nhwc_to_nchw: dst[c, p] = src[p, c] for c < C; dst f32 or bf16.
 * gaussian_sample: moments [HW, Cp] NHWC (mean channels [0,Z), logvar [Z,2Z)) -> out [Z, HW] bf16 =
 *              scale * ((mean + exp(0.5*logvar) * noise) - shift); noise [Z, HW] bf16 or NULL for the mean
 *              (DiagonalGaussian.forward :268-275, AutoEncoder.encode :301-304). */
int vc_im2col3x3(const void* src, void* dst, int32_t H, int32_t W, int32_t C, int32_t mode, void* stream);
/* conv3x3: the same convolution as vc_im2col3x3 + vc_gemm in ONE launch, no [H*W, 9C] matrix in HBM: the GEMM's loader
 * waves gather the taps from the NHWC map x, which must carry ONE EXTRA ZERO ROW after its Hs*Ws pixel rows (the source
 * of every out-of-image tap).  w [O, 9*C] with K ordered (dy, dx, c), C % 64 == 0, O % 8 == 0; out [H*W, O] row stride
 * ldc; res/gate NULL -> y = bf16(acc + bias), else y = bf16(res + bf16(gate * bf16(acc + bias))) (gate [O]). */
int vc_conv3x3(const void* x, const void* w, const void* bias, void* out, int64_t ldc, const void* res, int64_t ldres,
               const void* gate, int32_t H, int32_t W, int32_t C, int32_t O, int32_t mode, void* stream);
int vc_groupnorm(const void* x, const void* gamma, const void* beta, void* y, void* scratch, int64_t scratch_bytes,
                 int64_t HW, int32_t C, int32_t G, float eps, int32_t swish, void* stream);
int vc_softmax_rows(void* x, int64_t ld, int32_t rows, int32_t cols, float scale, const void* bias, int64_t ld_bias,
                    int32_t causal_period, void* stream);
int vc_transpose(const void* src, int64_t ld_src, void* dst, int64_t ld_dst, int32_t rows, int32_t cols, void* stream);
int vc_nchw_to_nhwc(const void* src, int32_t src_is_f32, void* dst, int32_t C, int32_t Cp, int64_t HW, float div, float add, void* stream);
int vc_nhwc_to_nchw(const void* src, void* dst, int32_t dst_is_f32, int32_t C, int32_t Cp, int64_t HW, void* stream);
int vc_gaussian_sample(const void* moments, int32_t Cp, const void* noise, void* out, int32_t Z, int64_t HW, float scale, float shift, void* stream);

/* ---- text-encoder glue (reference call site models/modules/conditioner.py:5-37, arithmetic of
 * transformers' T5EncoderModel / CLIPTextModel run in bf16).  Projections and per-head attention products are vc_gemm. ----
 * embedding:  out[i, :] = table[clamp(ids[i]), :]                                       (nn.Embedding)
 * rmsnorm:    y = bf16(w * bf16(x * rsqrt(mean(x^2) + eps))), f32 statistics, D <= 4096   (T5LayerNorm)
 * layernorm:  y = bf16((x - mean) * rstd * w + b), f32 statistics, D <= 4096              (nn.LayerNorm, CLIP)
 * mul / add:  y = bf16(a * b) / bf16(a + b), n % 8 == 0          (T5DenseGatedActDense product; CLIP token + position)
 * quick_gelu: y = bf16(x * bf16(sigmoid(bf16(1.702 * x))))                                (CLIP hidden_act) */
int vc_embedding(const int32_t* ids, const void* table, int64_t ld_table, int32_t vocab, void* out, int32_t L, int32_t D, void* stream);
int vc_rmsnorm(const void* x, const void* weight, void* y, int32_t rows, int32_t D, float eps, void* stream);
int vc_layernorm(const void* x, const void* weight, const void* bias, void* y, int32_t rows, int32_t D, float eps, void* stream);
int vc_mul(const void* a, const void* b, void* y, int64_t n, void* stream);
int vc_add(const void* a, const void* b, void* y, int64_t n, void* stream);
int vc_quick_gelu(const void* x, void* y, int64_t n, void* stream);

/* ---- handle API: the whole of Flux.forward, and the whole fixed-grid Euler loop, behind one call each ----
 * vc_flux_forward       replaces Flux.forward                      (models/model.py:85-124)
 * vc_flux_sample_euler  replaces odeint(method="euler") over it    (transport/integrators.py:106-120, transport.py:361-410:
 *                       drift = -model(x || cond, 1 - t); x += dt * drift with the reference's bf16 roundings)
 * The handle holds no tensor: weights are zero-copy views bound by name, every activation lives in a caller-provided
 * workspace.  It does own small host-side state (a pinned staging buffer, the captured hipGraph of one solver step, events).
 * One handle per device per process; not thread-safe per handle.  bf16 everywhere, LoRA pairs already folded into the
 * bound weights (W + s*B@A, lora.py:92-98; the un-merged parity mode stays on the op-level API). */
typedef struct VcFluxConfig {   /* FluxParams, models/model.py:18-32 */
  int32_t in_channels, out_channels, vec_in_dim, context_in_dim, hidden_size, num_heads, depth, depth_single_blocks;
  int32_t mlp_hidden;           /* hidden_size * mlp_ratio */
  int32_t guidance_embed;
  int32_t axes_dim[3];          /* sum = 128 */
  int32_t theta;
} VcFluxConfig;
int vc_flux_create(const VcFluxConfig* cfg, void** handle);
int vc_flux_destroy(void* handle);
/* name = reference module path of an nn.Linear ("img_in", "time_in.in_layer", "double_blocks.3.img_attn.qkv",
 * "single_blocks.7.linear1", "final_layer.linear", ...): w [rows, cols] bf16 with row stride ldw, bias [rows] or NULL;
 * or of a QKNorm scale ("double_blocks.3.img_attn.norm.query_norm.scale": w [128], rows = 1);
 * or "modulation": EVERY Modulation / adaLN Linear (layers.py:120-126, 253) stacked along the rows in the order
 * double_blocks.i.img_mod.lin, double_blocks.i.txt_mod.lin (i ascending), single_blocks.i.modulation.lin,
 * final_layer.adaLN_modulation.1 - vc_flux_mod_offset(name) is each module's first row, vc_flux_mod_offset(NULL) the total -
 * so that one GEMM yields every shift / scale / gate of every solver step;
 * or, optionally, "timestep_freqs": the 128 F32 frequencies exp(-ln(10000) * k / 128) of timestep_embedding
 * (layers.py:41-43) as the caller's framework computes them (rows = 1, cols = 128; default: exp in f64, rounded to f32,
 * which is within 1 ulp of torch's f32 exp);
 * or, optionally, "splitk_ws": ONE device scratch of rows * cols F32 values >= VC_GEMM_SPLITK_WS_BYTES for the split-K / stream
 * remainders of every geometry this handle runs (rows = 1, cols = the float count) - bound BEFORE vc_flux_workspace_bytes is
 * asked, it keeps those 100 MB out of each workspace (default: carved into every workspace); a FIRST bind after
 * vc_flux_workspace_bytes or vc_flux_prepare has answered is refused (VC_ERR_STATE: it would change the layout of workspaces the
 * caller already holds); re-binding another buffer is allowed at any time.  Pointers must stay valid while bound. */
int vc_flux_bind_weight(void* handle, const char* name, const void* w, const void* bias, int32_t rows, int32_t cols, int64_t ldw);
int64_t vc_flux_mod_offset(void* handle, const char* module_name);
/* knobs for tests and A/B runs: "attn_variant" (-1 = by size, the default), "tile_cfg" (0; flag bits such as
 * VC_GEMM_PREFER_STREAMK ride along), "fuse_qnorm" (2, the default: query QKNorm + RoPE + the softmax scale in the qkv GEMM's
 * epilogue wherever the key heads are normalised there, VcGemmProblem.qn_scale / VcAttention.q_prescaled; 1: inside the attention
 * kernel, VcAttention.q_scale; 0: by the pre-pass), "fuse_vt" (1: V^T from the qkv GEMM's epilogue, VC_EPI_QKV);
 * "qkv_heads" (0; = num_heads when the caller bound HEAD-PERMUTED qkv weights - every `*_attn.qkv` and the first
 * 3 * hidden rows of every `linear1`, with their biases, in the row order VcGemmProblem.kn_heads describes) and, with it,
 * "fuse_knorm" (0; 1: QKNorm + RoPE of the key heads inside the qkv GEMM's epilogue - with fuse_qnorm and fuse_vt the
 * projection, the norms, RoPE and the V transpose are then ONE GEMM launch + the attention kernel: no pre-pass, no prologue);
 * "logit_bound_milli" (0 = every attention launch keeps a running max; > 0: the softmax without one where the logits are bounded -
 * the value, 16330 * max|query_norm.scale| * max|key_norm.scale| over all blocks rounded up, is a CAP: each attention launch gets
 * the smaller of it and its OWN block's bound, which the library computes from the bound QK-norm scales when it resolves the
 * weights (one 256-byte read-back per scale vector), so a checkpoint with outlier scales in a few blocks runs the running-max
 * template in those blocks only); "mlp_first", "splitk" (1: split-K / stream remainders where the launcher takes them). */
int vc_flux_set_option(void* handle, const char* name, int32_t value);
int64_t vc_flux_workspace_bytes(void* handle, int32_t B, int32_t T, int32_t N, int32_t max_steps);

typedef struct VcFluxInputs {   /* everything of model_kwargs that does not change along the trajectory */
  int32_t B, T, N, max_steps;   /* samples stacked in one launch sequence; text / image tokens; solver steps the workspace holds */
  const void* txt;              /* [B, T, context_in_dim] bf16, device */
  const void* y;                /* [B, vec_in_dim] bf16, device */
  const float* guidance;        /* HOST [B], NULL without guidance_embed */
  const float* img_ids;         /* HOST [B, N, 3] */
  const float* txt_ids;         /* HOST [B, T, 3] */
  const int32_t* kv_len;        /* HOST [B] or NULL: rows >= kv_len[b] of the joint (txt, img) sequence are masked */
  const int32_t* kv_gap;        /* HOST [B][2] or NULL: a second masked range (see VcAttention.kv_gap) */
  int32_t guidance_is_bf16;     /* the caller's guidance tensor is bf16: 1000 * g rounds to bf16 (layers.py:38) */
  int32_t _pad;
} VcFluxInputs;
/* txt_in, the guidance / vector embedders, the RoPE table (f64 on the host as math.py:102-109), masks; zeroes the V^T
 * padding.  workspace: device memory, >= vc_flux_workspace_bytes(B, T, N, max_steps), 256-B aligned, the caller's for as
 * long as forward / sample calls follow. */
int vc_flux_prepare(void* handle, const VcFluxInputs* in, void* workspace, int64_t workspace_bytes, void* stream);
/* ONE evaluation: out [B, N, out_channels] = Flux(img [B, N, in_channels]; timesteps HOST [B]) */
int vc_flux_forward(void* handle, const void* img, const float* timesteps, int32_t timesteps_is_bf16, void* out, void* stream);
/* The loop.  x [B, N, out_channels]: in = x(t_grid[0]), out = x(t_grid[n_points-1]); cond [B, N, in - out channels] bf16;
 * t_grid HOST f32 [n_points] (n_points - 1 <= max_steps evaluations); state_is_bf16 != 0: x, x_out and trajectory are bf16 -
 * the pipeline's state dtype, visualcloze.py:399 - and the model sees 1 - bf16(t_i) (torchdiffeq hands the drift
 * t.to(y.dtype)); state_is_bf16 == 0: x, x_out and trajectory are F32, the state is stepped in f32 (integrators.py:119 keeps
 * the caller's dtype; the velocity and dt * velocity stay bf16 as under autocast) and the model sees 1 - t_i; dt is
 * t[i+1] - t[i] in f32 either way; trajectory: NULL or [n_points - 1][B][N][out_channels] receiving the state after every step.  One captured hipGraph per step (re-captured only
 * when geometry, masks' kind, workspace or options change); with stream == NULL the steps are launched uncaptured.
 * begin / steps / end expose the same loop piecewise (bench.py times single steps): sample_euler = begin; steps(n-1); end. */
int vc_flux_sample_euler(void* handle, void* x, const void* cond, const float* t_grid, int32_t n_points, int32_t state_is_bf16,
                         void* trajectory, void* stream);
int vc_flux_sample_begin(void* handle, const void* x, const void* cond, const float* t_grid, int32_t n_points,
                         int32_t state_is_bf16, void* stream);
int vc_flux_sample_steps(void* handle, int32_t n_steps, void* trajectory, void* stream);
int vc_flux_sample_end(void* handle, void* x_out, void* stream);

/* ---- the plan's own stopwatch (ABI 10): HIP-event times of the launches of whole evaluations, class by class ----
 * What bench.py's `roofline` leg reports.  With a sample in flight (vc_flux_sample_begin), `evaluations` more evaluations of
 * Flux.forward at the trajectory's CURRENT step are issued on `stream` by the same code the step graph was captured from - not
 * captured, no Euler update: the state, the step counter and the graph stay as they are - with an event in front of and behind
 * every GEMM, attention and LayerNorm-modulate launch (one warm evaluation first).  Every launch therefore finds the caches as
 * its predecessor in the step leaves them.  Synchronises `stream`; entries come in order of first appearance.
 * Reference: there is none (the reference times whole pipelines, visualcloze.py); replaces the Python-ordered twin of the plan
 * (visualcloze_amd/engine.py) as the thing bench.py times. */
enum { VC_LAUNCH_GEMM = 1, VC_LAUNCH_ATTENTION = 2, VC_LAUNCH_LN_MODULATE = 3 };
typedef struct VcFluxLaunchClass {
  int32_t kind;        /* VC_LAUNCH_* */
  int32_t epi;         /* GEMM: the VC_EPI_* of the launch; attention: the VcAttention.variant that ran; else 0 */
  int32_t n, k;        /* GEMM: N and K of the launch's first problem (a grouped launch: its img stream); else 0 */
  int32_t launches;    /* launches of this class in the timed evaluations */
  int32_t reserved;
  double flops;        /* their arithmetic: 2 M N K over all problems (GEMM), 4 L^2 D B (attention: masked keys included), else 0 */
  double bytes;        /* LayerNorm-modulate: rows read + rows written; else 0 */
  float total_us;      /* sum of the launches' event times (an attention launch: the kernel and, where one runs, its merge kernel) */
  float min_us, max_us;
  float reserved2;
} VcFluxLaunchClass;
int vc_flux_profile(void* handle, int32_t evaluations, VcFluxLaunchClass* out, int32_t capacity, int32_t* count, void* stream);

/* ---- hipGraph helpers: capture the launches issued on `stream` between begin/end ---- */
int vc_stream_create(void** stream);
int vc_stream_destroy(void* stream);
int vc_stream_sync(void* stream);
int vc_graph_begin(void* stream);
int vc_graph_end(void* stream, void** graph_exec);
int vc_graph_launch(void* graph_exec, void* stream);
int vc_graph_destroy(void* graph_exec);

/* ---- timing on the launch stream (HIP events), for bench.py's roofline leg ---- */
int vc_event_create(void** ev);
int vc_event_record(void* ev, void* stream);
int vc_event_elapsed_ms(void* ev_start, void* ev_stop, float* ms); /* synchronises on ev_stop */
int vc_event_destroy(void* ev);

#ifdef __cplusplus
}
#endif
#endif
