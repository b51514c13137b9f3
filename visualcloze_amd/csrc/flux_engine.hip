// Handle API of include/vcloze_hip.h: Flux.forward (models/model.py:85-124) and the fixed-grid Euler loop around it
// (transport/integrators.py:106-120) as launch plans over the kernels of this library.  Host code only: it ORDERS launches
// (once per geometry, under stream capture, for the sampling loop); nothing here touches a tensor element except the RoPE
// angle table, which math.py:102-109 computes in float64 on the host as well.
//
// Workspace (caller's device memory), bf16 unless noted; B samples, T text / N image tokens, L = T + N, D hidden, S = max_steps:
//   XI [B*N, D] / XT [B*T, D]  residual streams of the DoubleStream blocks      X   [B*L, D]  joint stream of the SingleStream blocks
//   XH [B*L, D]   LayerNorm+modulate output (GEMM A operand)                     QKV [B*L, 3D] "B L (K H D)" rows, joint order
//   VT [B, H, 128, Lp]  V transposed per head (Lp = L rounded up to 64, padding zeroed once)
//   CAT [B*L, D+mlp]    attn | gelu(mlp) = linear2's input; CAT[:, :D] is also the DoubleStream attention output
//   HID [B*L, mlp]      MLP hidden of the DoubleStream blocks (image rows first)
//   MOD [S*B, n_mod]    every modulation vector of every block for every solver step (row s*B + b)
//   XS / COND / XIN / V the ODE state, the conditioning columns, x || cond, the velocity
#include "common.h"
#include "vcloze_internal.h"
#include <math.h>
#include <string.h>
#include <string>
#include <unordered_map>
#include <vector>

namespace {

struct Lin {
  const void* w = nullptr;
  const void* b = nullptr;
  int N = 0, K = 0;
  int64_t ldw = 0;
};

// bound: the block's own logit bound, 1.02 sqrt(128) log2(e) max|query_norm.scale| max|key_norm.scale| (both streams of a double block)
struct DoubleW { Lin qkv[2], proj[2], mlp0[2], mlp2[2]; const void* qs[2]; const void* ks[2]; int64_t mod[2]; float bound; };  // [0] = img, [1] = txt
struct SingleW { Lin qkv, mlp, lin2; const void* qs; const void* ks; int64_t mod; float bound; };

struct Err {
  char* buf; int len;
};

struct Buffers {   // the workspace carve-up
  bf16_t *XI, *XT, *X, *XH, *QKV, *VT, *CAT, *HID, *TXT0, *XIN, *V, *XS, *COND, *MOD, *TEMB, *H1, *TVEC, *GVEC, *YVEC, *VEC, *GE, *GH, *YH;
  float *ROPE, *TS, *DTS, *G32, *FREQS, *XS32;
  int32_t *STEP, *KVLEN, *KVGAP;
  void* ATT_SCRATCH = nullptr;
  int64_t att_scratch_bytes = 0;
  void* SK_WS = nullptr;               // f32 partial tiles of split-K GEMM remainders (VcGemmArgs.splitk_ws)
  int64_t sk_ws_bytes = 0;
};

struct Flux : Buffers {
  VcFluxConfig cfg{};
  int D = 0, H = 0, mlp = 0;
  int64_t n_mod = 0;
  std::unordered_map<std::string, Lin> bound;
  std::unordered_map<std::string, int64_t> mod_off;
  // resolved at prepare time
  bool resolved = false;
  Lin img_in, txt_in, time_in[2], vector_in[2], guidance_in[2], final_lin, modulation;
  std::vector<DoubleW> dbl;
  std::vector<SingleW> sgl;
  int64_t final_mod = 0;
  // options
  int attn_variant = -1, tile_cfg = 0, fuse_qnorm = 2, fuse_vt = 1, qkv_heads = 0, fuse_knorm = 0, logit_bound_milli = 0, mlp_first = 0, splitk = 1, n_cu = 256;
  // prepared geometry + workspace carve-up
  bool prepared = false;
  bool ws_sized = false;     // vc_flux_workspace_bytes / vc_flux_prepare have answered with the current carve-up
  int B = 0, T = 0, N = 0, L = 0, Lp = 0, S = 0;
  bool ragged = false, gapped = false;
  char* base = nullptr;
  // captured steps, most recently used first (a two-stage pipeline alternates between two geometries)
  hipGraphExec_t graph = nullptr;      // = graphs.front().second while a sample is in flight
  struct Key {
    char* base; int B, T, N, S, ragged, gapped, variant, tile, fuse, fuse_vt, state_f32, qkv_heads, fuse_knorm, bound, mlp_first, splitk; hipStream_t s;
    bool operator==(const Key& o) const {
      return base == o.base && B == o.B && T == o.T && N == o.N && S == o.S && ragged == o.ragged && gapped == o.gapped &&
             variant == o.variant && tile == o.tile && fuse == o.fuse && fuse_vt == o.fuse_vt && state_f32 == o.state_f32 &&
             qkv_heads == o.qkv_heads && fuse_knorm == o.fuse_knorm && bound == o.bound && mlp_first == o.mlp_first && splitk == o.splitk && s == o.s;
    }
  } key{};
  std::vector<std::pair<Key, hipGraphExec_t>> graphs;
  // sampling state
  int steps_total = 0, steps_done = 0;
  bool state_f32 = false;              // the sample in flight steps an f32 state (XS32; XS is its bf16 shadow)
  // host staging (pinned), reused once the copies that read it have completed
  char* pinned = nullptr;
  size_t pinned_bytes = 0, pinned_used = 0;
  hipEvent_t staged = nullptr;
  bool staged_pending = false;
  // vc_flux_profile: HIP-event pairs around the launches of one evaluation (off outside that call)
  struct ProfRec { int kind, epi, n, k; double flops, bytes; size_t e0; };
  bool prof_on = false;
  std::vector<hipEvent_t> prof_ev;
  size_t prof_used = 0;
  std::vector<ProfRec> prof_recs;
};

#define FAIL(code, ...)                         \
  do {                                          \
    snprintf(e.buf, e.len, __VA_ARGS__);        \
    return code;                                \
  } while (0)
#define TRY(x)                \
  do {                        \
    int rc_ = (x);            \
    if (rc_ != VC_OK) return rc_; \
  } while (0)
#define HIP(x, what)                                                        \
  do {                                                                      \
    hipError_t he_ = (x);                                                   \
    if (he_ != hipSuccess) FAIL(VC_ERR_HIP, what ": %s", hipGetErrorString(he_)); \
  } while (0)

inline int64_t align256(int64_t v) { return (v + 255) & ~(int64_t)255; }

// ---------------------------------------------------------------- workspace
struct Carver {
  char* base; int64_t off = 0;
  template <class T> T* take(int64_t count) {
    T* p = base ? (T*)(base + off) : nullptr;
    off += align256(count * (int64_t)sizeof(T));
    return p;
  }
};

int64_t carve(Buffers& f, const Flux& g, char* base, int B, int T, int N, int S) {
  const int64_t D = g.D, H = g.H, mlp = g.mlp, L = T + N, Lp = (L + 63) / 64 * 64;
  const int64_t in_ch = g.cfg.in_channels, out_ch = g.cfg.out_channels;
  Carver c{base};
  f.XI = c.take<bf16_t>(B * N * D);          f.XT = c.take<bf16_t>(B * T * D);
  f.X = c.take<bf16_t>(B * L * D);           f.XH = c.take<bf16_t>(B * L * D);
  f.QKV = c.take<bf16_t>(B * L * 3 * D);     f.VT = c.take<bf16_t>(B * H * 128 * Lp);
  f.CAT = c.take<bf16_t>(B * L * (D + mlp)); f.HID = c.take<bf16_t>(B * L * mlp);
  f.TXT0 = c.take<bf16_t>(B * T * D);        f.XIN = c.take<bf16_t>(B * N * in_ch);
  f.V = c.take<bf16_t>(B * N * out_ch);      f.XS = c.take<bf16_t>(B * N * out_ch);
  f.COND = c.take<bf16_t>(B * N * (in_ch - out_ch));
  f.MOD = c.take<bf16_t>((int64_t)S * B * g.n_mod);
  f.TEMB = c.take<bf16_t>((int64_t)S * B * 256);
  f.H1 = c.take<bf16_t>((int64_t)S * B * D); f.TVEC = c.take<bf16_t>((int64_t)S * B * D);
  f.VEC = c.take<bf16_t>((int64_t)S * B * D);
  f.GVEC = c.take<bf16_t>(B * D);            f.YVEC = c.take<bf16_t>(B * D);
  f.GE = c.take<bf16_t>(B * 256);            f.GH = c.take<bf16_t>(B * D);  f.YH = c.take<bf16_t>(B * D);
  f.ROPE = c.take<float>(B * L * 128);
  f.TS = c.take<float>((int64_t)S * B);      f.DTS = c.take<float>(S);
  f.G32 = c.take<float>(B);                  f.FREQS = c.take<float>(128);
  f.XS32 = c.take<float>(B * N * out_ch);    // f32 master copy of the ODE state (state_is_bf16 == 0)
  f.STEP = c.take<int32_t>(1);               f.KVLEN = c.take<int32_t>(B);  f.KVGAP = c.take<int32_t>(2 * B);
  f.att_scratch_bytes = vc_attention_scratch_bytes_impl();
  f.ATT_SCRATCH = c.take<char>(f.att_scratch_bytes);
  // one split-K scratch serves every geometry of a handle (all its launches are ordered on one stream): a caller-bound one
  // ("splitk_ws", vc_flux_bind_weight) keeps the 100 MB out of every cached workspace (advisor r04); without it, it is carved here
  auto sk = g.bound.find("splitk_ws");
  if (sk != g.bound.end()) { f.SK_WS = const_cast<void*>(sk->second.w); f.sk_ws_bytes = (int64_t)sk->second.N * sk->second.K * 4; }
  else { f.SK_WS = c.take<char>(VC_GEMM_SPLITK_WS_BYTES); f.sk_ws_bytes = VC_GEMM_SPLITK_WS_BYTES; }
  return c.off;
}

// ---------------------------------------------------------------- weights
int find(Flux& f, const std::string& name, Lin& out, int N, int K, bool need_bias, Err e) {
  auto it = f.bound.find(name);
  if (it == f.bound.end()) FAIL(VC_ERR_STATE, "flux: weight '%s' is not bound", name.c_str());
  out = it->second;
  if (out.N != N || out.K != K) FAIL(VC_ERR_ARG, "flux: weight '%s' is [%d, %d], expected [%d, %d]", name.c_str(), out.N, out.K, N, K);
  if (need_bias && !out.b) FAIL(VC_ERR_ARG, "flux: weight '%s' needs a bias", name.c_str());
  return VC_OK;
}
int find_scale(Flux& f, const std::string& name, const void*& out, Err e) {
  auto it = f.bound.find(name);
  if (it == f.bound.end()) FAIL(VC_ERR_STATE, "flux: scale '%s' is not bound", name.c_str());
  if (it->second.N * it->second.K != 128) FAIL(VC_ERR_ARG, "flux: scale '%s' must hold 128 values", name.c_str());
  out = it->second.w;
  return VC_OK;
}

void layout_modulation(Flux& f) {
  int64_t o = 0;
  for (int i = 0; i < f.cfg.depth; ++i) {
    f.mod_off["double_blocks." + std::to_string(i) + ".img_mod.lin"] = o; o += 6 * f.D;
    f.mod_off["double_blocks." + std::to_string(i) + ".txt_mod.lin"] = o; o += 6 * f.D;
  }
  for (int i = 0; i < f.cfg.depth_single_blocks; ++i) { f.mod_off["single_blocks." + std::to_string(i) + ".modulation.lin"] = o; o += 3 * f.D; }
  f.mod_off["final_layer.adaLN_modulation.1"] = o; o += 2 * f.D;
  f.n_mod = o;
}

// max |scale| of a bound QK-norm scale vector (128 bf16 on the device): read back ONCE, when the weights are resolved
int scale_absmax(const void* dev, double& out, Err e) {
  uint16_t h[128];
  HIP(hipMemcpy(h, dev, sizeof(h), hipMemcpyDeviceToHost), "hipMemcpy (QK-norm scale)");
  out = 0;
  bool nan = false;
  for (int i = 0; i < 128; ++i) {
    uint32_t u = (uint32_t)h[i] << 16;
    float v;
    memcpy(&v, &u, 4);
    const double a = v < 0 ? -(double)v : (double)v;
    if (a != a) nan = true;
    else if (a > out) out = a;
  }
  if (nan) out = 1e30;                 // (no bound: the running-max template)
  return VC_OK;
}
// |q|, |k| <= sqrt(128) max|scale| after QKNorm (RoPE is a rotation), so |128^-0.5 log2(e) q.k| <= this; + 2 % for the six bf16
// roundings on the way (model.py / engine.py compute the same number for the Python-ordered plan)
inline float logit_bound_of(double qmax, double kmax) { return (float)(1.02 * 11.313708498984761 * 1.4426950408889634 * qmax * kmax); }

int resolve(Flux& f, Err e) {
  if (f.resolved) return VC_OK;
  const int D = f.D, mlp = f.mlp;
  TRY(find(f, "img_in", f.img_in, D, f.cfg.in_channels, false, e));
  TRY(find(f, "txt_in", f.txt_in, D, f.cfg.context_in_dim, false, e));
  TRY(find(f, "time_in.in_layer", f.time_in[0], D, 256, false, e));
  TRY(find(f, "time_in.out_layer", f.time_in[1], D, D, false, e));
  TRY(find(f, "vector_in.in_layer", f.vector_in[0], D, f.cfg.vec_in_dim, false, e));
  TRY(find(f, "vector_in.out_layer", f.vector_in[1], D, D, false, e));
  if (f.cfg.guidance_embed) {
    TRY(find(f, "guidance_in.in_layer", f.guidance_in[0], D, 256, false, e));
    TRY(find(f, "guidance_in.out_layer", f.guidance_in[1], D, D, false, e));
  }
  TRY(find(f, "final_layer.linear", f.final_lin, f.cfg.out_channels, D, false, e));
  TRY(find(f, "modulation", f.modulation, (int)f.n_mod, D, false, e));
  f.dbl.assign(f.cfg.depth, DoubleW{});
  for (int i = 0; i < f.cfg.depth; ++i) {
    const std::string pf = "double_blocks." + std::to_string(i) + ".";
    const char* st[2] = {"img", "txt"};
    for (int k = 0; k < 2; ++k) {
      const std::string a = pf + st[k] + "_attn.", m = pf + st[k] + "_mlp.";
      TRY(find(f, a + "qkv", f.dbl[i].qkv[k], 3 * D, D, false, e));
      TRY(find(f, a + "proj", f.dbl[i].proj[k], D, D, false, e));
      TRY(find(f, m + "0", f.dbl[i].mlp0[k], mlp, D, false, e));
      TRY(find(f, m + "2", f.dbl[i].mlp2[k], D, mlp, false, e));
      TRY(find_scale(f, a + "norm.query_norm.scale", f.dbl[i].qs[k], e));
      TRY(find_scale(f, a + "norm.key_norm.scale", f.dbl[i].ks[k], e));
      f.dbl[i].mod[k] = f.mod_off[pf + st[k] + "_mod.lin"];
    }
    double q[2], k2[2];
    for (int k = 0; k < 2; ++k) { TRY(scale_absmax(f.dbl[i].qs[k], q[k], e)); TRY(scale_absmax(f.dbl[i].ks[k], k2[k], e)); }
    f.dbl[i].bound = logit_bound_of(q[0] > q[1] ? q[0] : q[1], k2[0] > k2[1] ? k2[0] : k2[1]);
  }
  f.sgl.assign(f.cfg.depth_single_blocks, SingleW{});
  for (int i = 0; i < f.cfg.depth_single_blocks; ++i) {
    const std::string pf = "single_blocks." + std::to_string(i) + ".";
    Lin l1;
    TRY(find(f, pf + "linear1", l1, 3 * D + mlp, D, false, e));
    SingleW& w = f.sgl[i];
    w.qkv = l1; w.qkv.N = 3 * D;                                  // linear1's rows [0, 3D) -> qkv, the rest -> mlp (layers.py:236)
    w.mlp = l1; w.mlp.N = mlp;
    w.mlp.w = (const bf16_t*)l1.w + (int64_t)3 * D * l1.ldw;
    w.mlp.b = l1.b ? (const void*)((const bf16_t*)l1.b + 3 * D) : nullptr;
    TRY(find(f, pf + "linear2", w.lin2, D, D + mlp, false, e));
    TRY(find_scale(f, pf + "norm.query_norm.scale", w.qs, e));
    TRY(find_scale(f, pf + "norm.key_norm.scale", w.ks, e));
    w.mod = f.mod_off[pf + "modulation.lin"];
    double q, k;
    TRY(scale_absmax(w.qs, q, e)); TRY(scale_absmax(w.ks, k, e));
    w.bound = logit_bound_of(q, k);
  }
  f.final_mod = f.mod_off["final_layer.adaLN_modulation.1"];
  f.resolved = true;
  return VC_OK;
}

// ---------------------------------------------------------------- launch helpers
// qkv projections: with fuse_vt the V third leaves the GEMM transposed into VT (EPI_QKV); with head-permuted weights (option
// qkv_heads) C receives the logical columns and, with fuse_knorm, the key heads leave QK-normed and rotated - and with fuse_qnorm
// the query heads too, times the softmax scale (VcGemmProblem.qn_prescale): no pre-pass, no prologue arithmetic in the attention
int attention_variant(const Flux& f);
// (where the one-wave-per-SIMD attention kernel runs: small geometries keep ONE pre-pass launch for q and k - the fused
// epilogue needs the 256x192 tile, which their short M does not fill)
bool kn_in_gemm(const Flux& f) { return f.fuse_knorm && f.qkv_heads > 0 && (attention_variant(f) & 8); }
bool qn_in_gemm(const Flux& f) { return kn_in_gemm(f) && f.fuse_qnorm >= 2; }
int qkv_epi(const Flux& f) { return f.fuse_vt || f.qkv_heads > 0 ? VC_EPI_QKV : VC_EPI_BIAS; }
void with_vt(Flux& f, VcGemmProblem& p, int rows, int row0, const void* q_scale, const void* k_scale) {
  if (f.fuse_vt) { p.vt = f.VT; p.vt_bstride = (int64_t)f.H * 128 * f.Lp; p.vt_col0 = 2 * f.D; p.vt_lpad = f.Lp; }
  if (f.qkv_heads > 0) {
    p.kn_heads = f.qkv_heads;
    if (kn_in_gemm(f)) { p.kn_scale = k_scale; p.kn_rope = f.ROPE; p.kn_rope_bstride = (int64_t)f.L * 128; }
    if (qn_in_gemm(f)) { p.qn_scale = q_scale; p.qn_prescale = 1; }
  }
  if (f.fuse_vt || f.qkv_heads > 0) { p.vt_rpb = rows; p.vt_row0 = row0; }
}

VcGemmProblem prob(const void* A, int64_t lda, const Lin& w, void* C, int64_t ldc, int M) {
  VcGemmProblem p;
  memset(&p, 0, sizeof(p));
  p.A = A; p.W = w.w; p.bias = w.b; p.C = C;
  p.lda = lda; p.ldw = w.ldw; p.ldc = ldc;
  p.M = M; p.N = w.N; p.K = w.K;
  p.rows_per_batch = M;
  return p;
}
// vc_flux_profile: an event in front of and behind a launch of the plan (one pair per launch, taken from a pool that grows on demand)
int prof_event(Flux& f, hipStream_t s, Err e) {
  if (f.prof_used == f.prof_ev.size()) {
    hipEvent_t ev;
    HIP(hipEventCreate(&ev), "hipEventCreate");
    f.prof_ev.push_back(ev);
  }
  HIP(hipEventRecord(f.prof_ev[f.prof_used++], s), "hipEventRecord");
  return VC_OK;
}
int prof_open(Flux& f, hipStream_t s, int kind, int epi, int n, int k, double flops, double bytes, Err e) {
  if (!f.prof_on) return VC_OK;
  f.prof_recs.push_back(Flux::ProfRec{kind, epi, n, k, flops, bytes, f.prof_used});
  return prof_event(f, s, e);
}
int prof_close(Flux& f, hipStream_t s, Err e) { return f.prof_on ? prof_event(f, s, e) : VC_OK; }

int gemm(Flux& f, const VcGemmProblem* ps, int n, int epi, const int32_t* step_ptr, int64_t gate_step_stride, hipStream_t s, Err e) {
  VcGemmArgs a;
  memset(&a, 0, sizeof(a));
  for (int i = 0; i < n; ++i) a.p[i] = ps[i];
  a.nprob = n; a.epi = epi; a.step_ptr = step_ptr; a.gate_step_stride = gate_step_stride;
  if (f.splitk) { a.splitk_ws = f.SK_WS; a.splitk_ws_bytes = f.sk_ws_bytes; }   // the launcher's cost model decides
  double flops = 0;
  for (int i = 0; i < n; ++i) flops += 2.0 * ps[i].M * ps[i].N * ps[i].K;
  TRY(prof_open(f, s, VC_LAUNCH_GEMM, epi, ps[0].N, ps[0].K, flops, 0, e));
  TRY(vc_gemm_launch(a, f.tile_cfg, s, e.buf, e.len));
  return prof_close(f, s, e);
}
int lin(Flux& f, const Lin& w, const void* A, int64_t lda, void* C, int64_t ldc, int M, int epi, hipStream_t s, Err e) {
  VcGemmProblem p = prob(A, lda, w, C, ldc, M);
  return gemm(f, &p, 1, epi, nullptr, 0, s, e);
}

struct Ctx {
  const int32_t* step_ptr;
  hipStream_t s;
  int64_t mss;     // MOD step stride (elements); the sample stride is n_mod
};
inline const bf16_t* modp(Flux& f, int64_t off, int idx) { return f.MOD + off + (int64_t)idx * f.D; }

int ln2(Flux& f, const Ctx& c, const DoubleW& w, int idx, Err e) {   // img + txt streams in one launch
  VcLnStream a{f.XI, f.D, f.XH, f.D, modp(f, w.mod[0], idx), modp(f, w.mod[0], idx + 1), f.B * f.N, f.N};
  VcLnStream b{f.XT, f.D, f.XH + (int64_t)f.B * f.N * f.D, f.D, modp(f, w.mod[1], idx), modp(f, w.mod[1], idx + 1), f.B * f.T, f.T};
  TRY(prof_open(f, c.s, VC_LAUNCH_LN_MODULATE, 0, 0, 0, 0, 4.0 * f.B * f.L * f.D, e));      // rows read + written, bf16
  TRY(vc_ln_modulate2_launch(&a, &b, f.n_mod, f.D, c.step_ptr, c.mss, c.s, e.buf, e.len));
  return prof_close(f, c.s, e);
}
int ln1(Flux& f, const Ctx& c, int64_t mod, Err e) {                  // the joint stream X -> XH
  TRY(prof_open(f, c.s, VC_LAUNCH_LN_MODULATE, 0, 0, 0, 0, 4.0 * f.B * f.L * f.D, e));
  TRY(vc_ln_modulate_launch(f.X, f.D, f.XH, f.D, modp(f, mod, 0), modp(f, mod, 1), f.n_mod, f.B * f.L, f.D, f.L, c.step_ptr, c.mss,
                            c.s, e.buf, e.len));
  return prof_close(f, c.s, e);
}

int attention_variant(const Flux& f) {
  if (f.attn_variant >= 0) return f.attn_variant;
  // 28 = 12 + 16: one wave per SIMD, tail split, and - where the stream form of the kernel runs - the tail pieces combined
  // inside the launch (the flag words of ATT_SCRATCH are zeroed by vc_flux_prepare, and only this handle's launches, ordered on
  // one stream, touch it).  Fewer 256-query items than CUs (cfg 1: 168): the same kernel WITHOUT a split (8) - one item per
  // workgroup beats the 32-queries-per-wave kernel down to half the CUs (round 6, cfg 1: 44.6-49.7 vs 56.3-59.4 us per launch
  // in situ; per step +1.7 % on one box, +-0.1 % on another - the q / k norm moves from the pre-pass into the qkv GEMM's
  // epilogue with it; cutting 168 items into 256 short pieces loses 1 %: profiles/r06o_ab_cfg1.log, r06q_ab_cfg1.log)
  const int items = ((f.L + 255) / 256) * f.H * f.B;
  return items >= f.n_cu ? 28 : 2 * items >= f.n_cu ? 8 : 3;
}

// QKNorm + RoPE (+ V^T) and the joint attention over QKV -> CAT[:, :D] (layers.py:165-185 / 236-241)
int attention(Flux& f, const Ctx& c, const void* q1, const void* k1, const void* q2, const void* k2, int split, float block_bound, Err e) {
  const int variant = attention_variant(f);
  const bool fused_q = (variant & 8) && f.fuse_qnorm, q_done = qn_in_gemm(f);     // q_done: by the projection's epilogue, prescaled
  const int64_t ld = 3 * f.D, ldo = f.D + f.mlp;
  const int parts = (kn_in_gemm(f) ? 0 : VC_QKN_K) | (fused_q ? 0 : VC_QKN_Q) | (f.fuse_vt ? 0 : VC_QKN_VT);
  if (parts)
    TRY(vc_qknorm_rope_vt_launch(f.QKV, ld, f.L * ld, q1, k1, q2, k2, split, f.ROPE, (int64_t)f.L * 128, f.VT, f.B, f.L, f.Lp, f.H,
                                 parts, c.s, e.buf, e.len));
  VcAttention a;
  memset(&a, 0, sizeof(a));
  a.qkv = f.QKV; a.ld = ld; a.bstride = f.L * ld;
  a.vt = f.VT; a.out = f.CAT; a.ldo = ldo; a.out_bstride = f.L * ldo;
  a.kv_len = f.ragged ? f.KVLEN : nullptr;
  a.kv_gap = f.gapped ? f.KVGAP : nullptr;
  a.B = f.B; a.L = f.L; a.Lpad = f.Lp; a.H = f.H; a.variant = variant;
  a.scratch = f.ATT_SCRATCH; a.scratch_bytes = f.att_scratch_bytes;
  if (q_done) a.q_prescaled = 1;
  else if (fused_q) { a.q_scale = q1; a.q_scale2 = q2; a.split = split; a.rope = f.ROPE; a.rope_bstride = (int64_t)f.L * 128; }
  // option logit_bound_milli > 0 switches the bounded softmax on; every block is then held to ITS OWN bound (never above the
  // caller's): a checkpoint with a few outlier QK-norm scales runs the running-max template in those blocks only
  const float cap = (float)f.logit_bound_milli * 1e-3f;
  a.logit_bound = f.logit_bound_milli > 0 ? (block_bound < cap ? block_bound : cap) : 0.0f;
  if (f.logit_bound_milli > 0 && !(block_bound < 1e30f)) a.logit_bound = 1e30f;       // (non-finite scales: no bound)
  TRY(prof_open(f, c.s, VC_LAUNCH_ATTENTION, variant, 0, 0, 4.0 * f.L * f.L * f.D * f.B, 0, e));     // the attention launch(es) alone
  TRY(vc_attention_launch(a, c.s, e.buf, e.len));
  return prof_close(f, c.s, e);
}

// DoubleStreamBlock (layers.py:158-196) on XI / XT, in place
int double_block(Flux& f, const Ctx& c, const DoubleW& w, Err e) {
  const int B = f.B, T = f.T, N = f.N, L = f.L, D = f.D, mlp = f.mlp;
  const int64_t ldq = 3 * D, ldc = D + mlp;
  bf16_t* XH_I = f.XH; bf16_t* XH_T = f.XH + (int64_t)B * N * D;
  bf16_t* HID_I = f.HID; bf16_t* HID_T = f.HID + (int64_t)B * N * mlp;
  bf16_t* streams[2] = {f.XI, f.XT};
  const int rows[2] = {N, T};
  TRY(ln2(f, c, w, 0, e));
  {  // qkv of both streams into the joint-order QKV rows (text first): C rows are batch-strided
    VcGemmProblem p[2] = {prob(XH_I, D, w.qkv[0], f.QKV + (int64_t)T * ldq, ldq, B * N), prob(XH_T, D, w.qkv[1], f.QKV, ldq, B * T)};
    p[0].c_rpb = N; p[1].c_rpb = T;
    p[0].c_bstride = p[1].c_bstride = (int64_t)L * ldq;
    with_vt(f, p[0], N, T, w.qs[0], w.ks[0]); with_vt(f, p[1], T, 0, w.qs[1], w.ks[1]);
    TRY(gemm(f, p, 2, qkv_epi(f), nullptr, 0, c.s, e));
  }
  TRY(attention(f, c, w.qs[1], w.ks[1], w.qs[0], w.ks[0], T, w.bound, e));   // rows < T: the text stream's scales
  {  // x += gate * proj(attn): A rows are batch-strided views of CAT[:, :D]
    VcGemmProblem p[2] = {prob(f.CAT + (int64_t)T * ldc, ldc, w.proj[0], f.XI, D, B * N), prob(f.CAT, ldc, w.proj[1], f.XT, D, B * T)};
    for (int k = 0; k < 2; ++k) {
      p[k].res = streams[k]; p[k].ldres = D; p[k].gate = modp(f, w.mod[k], 2); p[k].gate_bstride = f.n_mod;
      p[k].rows_per_batch = rows[k]; p[k].a_rpb = rows[k]; p[k].a_bstride = (int64_t)L * ldc;
    }
    TRY(gemm(f, p, 2, VC_EPI_GATE_RES, c.step_ptr, c.mss, c.s, e));
  }
  TRY(ln2(f, c, w, 3, e));
  {
    VcGemmProblem p[2] = {prob(XH_I, D, w.mlp0[0], HID_I, mlp, B * N), prob(XH_T, D, w.mlp0[1], HID_T, mlp, B * T)};
    TRY(gemm(f, p, 2, VC_EPI_GELU, nullptr, 0, c.s, e));
  }
  {
    VcGemmProblem p[2] = {prob(HID_I, mlp, w.mlp2[0], f.XI, D, B * N), prob(HID_T, mlp, w.mlp2[1], f.XT, D, B * T)};
    for (int k = 0; k < 2; ++k) {
      p[k].res = streams[k]; p[k].ldres = D; p[k].gate = modp(f, w.mod[k], 5); p[k].gate_bstride = f.n_mod; p[k].rows_per_batch = rows[k];
    }
    TRY(gemm(f, p, 2, VC_EPI_GATE_RES, c.step_ptr, c.mss, c.s, e));
  }
  return VC_OK;
}

// SingleStreamBlock (layers.py:232-245) on X, in place
int single_block(Flux& f, const Ctx& c, const SingleW& w, Err e) {
  const int M = f.B * f.L, D = f.D, mlp = f.mlp;
  const int64_t ldc = D + mlp;
  TRY(ln1(f, c, w.mod, e));
  {
    VcGemmProblem p = prob(f.XH, D, w.qkv, f.QKV, 3 * D, M);
    with_vt(f, p, f.L, 0, w.qs, w.ks);
    TRY(gemm(f, &p, 1, qkv_epi(f), nullptr, 0, c.s, e));
  }
  // the attention kernel runs right behind the projection that wrote its operands, and the MLP-up GEMM right in front of the
  // linear2 that reads its 97 MB (option mlp_first = the reference's textual order, layers.py:236-243)
  if (f.mlp_first) TRY(lin(f, w.mlp, f.XH, D, f.CAT + D, ldc, M, VC_EPI_GELU, c.s, e));
  TRY(attention(f, c, w.qs, w.ks, nullptr, nullptr, 0, w.bound, e));
  if (!f.mlp_first) TRY(lin(f, w.mlp, f.XH, D, f.CAT + D, ldc, M, VC_EPI_GELU, c.s, e));
  VcGemmProblem p = prob(f.CAT, ldc, w.lin2, f.X, D, M);
  p.res = f.X; p.ldres = D; p.gate = modp(f, w.mod, 2); p.gate_bstride = f.n_mod; p.rows_per_batch = f.L;
  return gemm(f, &p, 1, VC_EPI_GATE_RES, c.step_ptr, c.mss, c.s, e);
}

int d2d(void* dst, const void* src, int64_t bytes, hipStream_t s, Err e) {
  HIP(hipMemcpyAsync(dst, src, (size_t)bytes, hipMemcpyDeviceToDevice, s), "hipMemcpyAsync");
  return VC_OK;
}

// Flux.forward on `img_rows` (x || cond in XIN when NULL) -> `out` (V when NULL); the Euler update of XS when `euler`
int evaluate(Flux& f, const int32_t* step_ptr, bool concat, const void* img_rows, void* out, bool euler, hipStream_t s, Err e) {
  const int B = f.B, T = f.T, N = f.N, L = f.L, D = f.D;
  const int in_ch = f.cfg.in_channels, out_ch = f.cfg.out_channels;
  Ctx c{step_ptr, s, (int64_t)B * f.n_mod};
  if (concat) TRY(vc_concat_cols_launch(f.XS, out_ch, f.COND, in_ch - out_ch, f.XIN, (int64_t)B * N, s, e.buf, e.len));
  TRY(d2d(f.XT, f.TXT0, (int64_t)B * T * D * 2, s, e));
  TRY(lin(f, f.img_in, img_rows ? img_rows : f.XIN, in_ch, f.XI, D, B * N, VC_EPI_BIAS, s, e));
  for (auto& w : f.dbl) TRY(double_block(f, c, w, e));
  for (int b = 0; b < B; ++b) {   // cat((txt, img), 1) per sample (model.py:116)
    TRY(d2d(f.X + (int64_t)b * L * D, f.XT + (int64_t)b * T * D, (int64_t)T * D * 2, s, e));
    TRY(d2d(f.X + ((int64_t)b * L + T) * D, f.XI + (int64_t)b * N * D, (int64_t)N * D * 2, s, e));
  }
  for (auto& w : f.sgl) TRY(single_block(f, c, w, e));
  // LastLayer (layers.py:248-259) on the image rows
  TRY(ln1(f, c, f.final_mod, e));
  VcGemmProblem p = prob(f.XH + (int64_t)T * D, D, f.final_lin, out ? out : f.V, out_ch, B * N);
  p.a_rpb = N; p.a_bstride = (int64_t)L * D;
  TRY(gemm(f, &p, 1, VC_EPI_BIAS, nullptr, 0, s, e));
  if (euler) {
    if (f.state_f32) TRY(vc_euler_f32_launch(f.XS32, f.XS, f.V, f.DTS, step_ptr, (int64_t)B * N * out_ch, s, e.buf, e.len));
    else TRY(vc_euler_launch(f.XS, f.V, f.DTS, step_ptr, (int64_t)B * N * out_ch, s, e.buf, e.len));
    TRY(vc_step_advance_launch((int32_t*)step_ptr, s, e.buf, e.len));
  }
  return VC_OK;
}

// ---------------------------------------------------------------- host staging
int stage_begin(Flux& f, size_t bytes, Err e) {
  if (f.staged_pending) { HIP(hipEventSynchronize(f.staged), "hipEventSynchronize"); f.staged_pending = false; }
  if (bytes > f.pinned_bytes) {
    if (f.pinned) (void)hipHostFree(f.pinned);
    f.pinned = nullptr; f.pinned_bytes = 0;
    HIP(hipHostMalloc((void**)&f.pinned, bytes, hipHostMallocDefault), "hipHostMalloc");
    f.pinned_bytes = bytes;
  }
  f.pinned_used = 0;
  return VC_OK;
}
template <class T> T* stage_take(Flux& f, size_t count) {
  T* p = (T*)(f.pinned + f.pinned_used);
  f.pinned_used += (size_t)align256((int64_t)(count * sizeof(T)));
  return p;
}
int stage_send(Flux& f, void* dst, const void* staged, size_t bytes, hipStream_t s, Err e) {
  HIP(hipMemcpyAsync(dst, staged, bytes, hipMemcpyHostToDevice, s), "hipMemcpyAsync(H2D)");
  return VC_OK;
}
int stage_end(Flux& f, hipStream_t s, Err e) {
  HIP(hipEventRecord(f.staged, s), "hipEventRecord");
  f.staged_pending = true;
  return VC_OK;
}

inline float bf16_round(float v) {   // round-to-nearest-even through bf16 (finite inputs)
  uint32_t u;
  memcpy(&u, &v, 4);
  u += 0x7fffu + ((u >> 16) & 1u);
  u &= 0xffff0000u;
  memcpy(&v, &u, 4);
  return v;
}

// timestep embedding -> time_in MLP -> vec = time + guidance + vector -> every modulation of every (step, sample):
// model.py:102-108 + layers.py:120-126 for all modules in ONE GEMM.  times: rows s*B + b already in TS.
int time_precompute(Flux& f, int S, int timesteps_is_bf16, hipStream_t s, Err e) {
  const int B = f.B, D = f.D, M = S * B;
  TRY(vc_temb_launch(f.TS, f.FREQS, f.TEMB, M, 128, timesteps_is_bf16, s, e.buf, e.len));
  TRY(lin(f, f.time_in[0], f.TEMB, 256, f.H1, D, M, VC_EPI_SILU, s, e));
  TRY(lin(f, f.time_in[1], f.H1, D, f.TVEC, D, M, VC_EPI_BIAS, s, e));
  if (f.cfg.guidance_embed)
    TRY(vc_add3_launch(f.TVEC, f.GVEC, f.YVEC, f.VEC, (int64_t)M * D, (int64_t)B * D, (int64_t)B * D, s, e.buf, e.len));
  else
    TRY(vc_add3_launch(f.TVEC, f.YVEC, nullptr, f.VEC, (int64_t)M * D, (int64_t)B * D, 1, s, e.buf, e.len));
  TRY(vc_silu_launch(f.VEC, f.H1, (int64_t)M * D, s, e.buf, e.len));
  return lin(f, f.modulation, f.H1, D, f.MOD, f.n_mod, M, VC_EPI_BIAS, s, e);
}

constexpr size_t MAX_GRAPHS = 4;
void drop_prof(Flux& f) {
  for (auto ev : f.prof_ev) (void)hipEventDestroy(ev);
  f.prof_ev.clear();
}
void drop_graph(Flux& f) {
  for (auto& g : f.graphs) (void)hipGraphExecDestroy(g.second);
  f.graphs.clear();
  f.graph = nullptr;
}

// the hipGraph of ONE solver step: everything step-dependent (modulation rows, dt) is indexed on the device by STEP
int step_graph(Flux& f, hipStream_t s, Err e) {
  Flux::Key k{f.base, f.B, f.T, f.N, f.S, f.ragged, f.gapped, attention_variant(f), f.tile_cfg, f.fuse_qnorm, f.fuse_vt, f.state_f32,
              f.qkv_heads, f.fuse_knorm, f.logit_bound_milli, f.mlp_first, f.splitk, s};
  for (size_t i = 0; i < f.graphs.size(); ++i)
    if (f.graphs[i].first == k) {
      auto hit = f.graphs[i];
      f.graphs.erase(f.graphs.begin() + i);
      f.graphs.insert(f.graphs.begin(), hit);
      f.graph = hit.second; f.key = k;
      return VC_OK;
    }
  f.graph = nullptr;
  // warm-up outside capture (kernel attributes are set on first launch): one evaluation into V, the state is untouched
  TRY(evaluate(f, f.STEP, true, nullptr, nullptr, false, s, e));
  HIP(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal), "hipStreamBeginCapture");
  const int rc = evaluate(f, f.STEP, true, nullptr, nullptr, true, s, e);
  hipGraph_t g = nullptr;
  hipError_t he = hipStreamEndCapture(s, &g);
  if (rc != VC_OK) { if (g) (void)hipGraphDestroy(g); return rc; }
  HIP(he, "hipStreamEndCapture");
  hipGraphExec_t ge = nullptr;
  he = hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
  (void)hipGraphDestroy(g);
  HIP(he, "hipGraphInstantiate");
  if (f.graphs.size() >= MAX_GRAPHS) { (void)hipGraphExecDestroy(f.graphs.back().second); f.graphs.pop_back(); }
  f.graphs.insert(f.graphs.begin(), {k, ge});
  f.graph = ge; f.key = k;
  return VC_OK;
}

}  // namespace

#define H(handle)                                                         \
  Err e{err, errlen};                                                     \
  if (!(handle)) FAIL(VC_ERR_ARG, "flux: null handle");                   \
  Flux& f = *(Flux*)(handle)

int vc_flux_create_impl(const VcFluxConfig* cfg, void** handle, char* err, int errlen) {
  Err e{err, errlen};
  if (!cfg || !handle) FAIL(VC_ERR_ARG, "flux_create: null argument");
  if (cfg->hidden_size <= 0 || cfg->num_heads <= 0 || cfg->hidden_size != cfg->num_heads * 128)
    FAIL(VC_ERR_ARG, "flux_create: hidden_size must be num_heads * 128 (the gfx950 attention kernels are specialised for head_dim 128)");
  if (cfg->axes_dim[0] + cfg->axes_dim[1] + cfg->axes_dim[2] != 128 || (cfg->axes_dim[0] | cfg->axes_dim[1] | cfg->axes_dim[2]) & 1)
    FAIL(VC_ERR_ARG, "flux_create: axes_dim must be even and sum to 128");
  if (cfg->in_channels <= cfg->out_channels || cfg->out_channels <= 0 || cfg->depth < 0 || cfg->depth_single_blocks < 0 || cfg->mlp_hidden <= 0)
    FAIL(VC_ERR_ARG, "flux_create: bad channel / depth configuration");
  Flux* f = new Flux();
  f->cfg = *cfg;
  f->D = cfg->hidden_size; f->H = cfg->num_heads; f->mlp = cfg->mlp_hidden;
  layout_modulation(*f);
  int dev = 0;
  hipDeviceProp_t p;
  if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&p, dev) != hipSuccess) {
    delete f;
    FAIL(VC_ERR_HIP, "flux_create: no ROCm device");
  }
  f->n_cu = p.multiProcessorCount;
  if (hipEventCreateWithFlags(&f->staged, hipEventDisableTiming) != hipSuccess) { delete f; FAIL(VC_ERR_HIP, "flux_create: hipEventCreate failed"); }
  *handle = f;
  return VC_OK;
}

int vc_flux_destroy_impl(void* handle, char* err, int errlen) {
  H(handle);
  if (f.staged_pending) (void)hipEventSynchronize(f.staged);
  drop_graph(f);
  drop_prof(f);
  if (f.pinned) (void)hipHostFree(f.pinned);
  if (f.staged) (void)hipEventDestroy(f.staged);
  delete &f;
  return VC_OK;
}

int vc_flux_bind_weight_impl(void* handle, const char* name, const void* w, const void* bias, int32_t rows, int32_t cols, int64_t ldw,
                             char* err, int errlen) {
  H(handle);
  if (!name || !w || rows <= 0 || cols <= 0 || ldw < cols) FAIL(VC_ERR_ARG, "flux_bind_weight: bad arguments for '%s'", name ? name : "?");
  // the optional split-K scratch decides whether 100 MB are carved into EVERY workspace: binding it for the first time after a
  // workspace has been sized would silently change the carve-up of buffers the caller already holds (advisor r05)
  if (!strcmp(name, "splitk_ws") && f.ws_sized && f.bound.find("splitk_ws") == f.bound.end())
    FAIL(VC_ERR_STATE, "flux_bind_weight: bind 'splitk_ws' BEFORE the first vc_flux_workspace_bytes / vc_flux_prepare (it changes the size and "
                       "layout of every workspace)");
  Lin l;
  l.w = w; l.b = bias; l.N = rows; l.K = cols; l.ldw = ldw;
  f.bound[name] = l;
  f.resolved = false;
  f.prepared = false;   // the resolved per-block tables are rebuilt by the next vc_flux_prepare
  drop_graph(f);        // a captured step holds the old pointers
  return VC_OK;
}

int64_t vc_flux_mod_offset_impl(void* handle, const char* name) {
  if (!handle) return -1;
  Flux& f = *(Flux*)handle;
  if (!name) return f.n_mod;
  auto it = f.mod_off.find(name);
  return it == f.mod_off.end() ? -1 : it->second;
}

int vc_flux_set_option_impl(void* handle, const char* name, int32_t value, char* err, int errlen) {
  H(handle);
  if (!name) FAIL(VC_ERR_ARG, "flux_set_option: null name");
  if (!strcmp(name, "attn_variant")) f.attn_variant = value;
  else if (!strcmp(name, "tile_cfg")) f.tile_cfg = value;
  else if (!strcmp(name, "fuse_qnorm")) f.fuse_qnorm = value < 0 ? 0 : value > 2 ? 2 : value;
  else if (!strcmp(name, "fuse_vt")) f.fuse_vt = value != 0;
  else if (!strcmp(name, "qkv_heads")) {      // the bound qkv weights (linear1's first 3D rows) are head-permuted (vcloze_hip.h)
    if (value != 0 && value != f.H) FAIL(VC_ERR_ARG, "flux_set_option: qkv_heads must be 0 or num_heads = %d", f.H);
    f.qkv_heads = value;
  } else if (!strcmp(name, "fuse_knorm")) f.fuse_knorm = value != 0;
  else if (!strcmp(name, "logit_bound_milli")) f.logit_bound_milli = value > 0 ? value : 0;
  else if (!strcmp(name, "mlp_first")) f.mlp_first = value != 0;
  else if (!strcmp(name, "splitk")) f.splitk = value != 0;      // split-K remainders (VcGemmArgs.splitk_ws) where the launcher's model takes them
  else FAIL(VC_ERR_ARG, "flux_set_option: unknown option '%s'", name);
  return VC_OK;
}

int64_t vc_flux_workspace_bytes_impl(void* handle, int32_t B, int32_t T, int32_t N, int32_t max_steps) {
  if (!handle || B <= 0 || T <= 0 || N <= 0 || max_steps <= 0) return -1;
  Buffers tmp;                    // carve into a scratch copy: the pointers of the live handle stay as they are
  ((Flux*)handle)->ws_sized = true;
  return carve(tmp, *(Flux*)handle, nullptr, B, T, N, max_steps);
}

int vc_flux_prepare_impl(void* handle, const VcFluxInputs* in, void* workspace, int64_t workspace_bytes, hipStream_t s, char* err, int errlen) {
  H(handle);
  if (!in || !workspace) FAIL(VC_ERR_ARG, "flux_prepare: null argument");
  f.ws_sized = true;
  const int B = in->B, T = in->T, N = in->N, S = in->max_steps;
  if (B <= 0 || T <= 0 || N <= 0 || S <= 0) FAIL(VC_ERR_ARG, "flux_prepare: B, T, N, max_steps must be positive");
  if (!in->txt || !in->y || !in->img_ids || !in->txt_ids) FAIL(VC_ERR_ARG, "flux_prepare: txt, y, img_ids, txt_ids are required");
  if (f.cfg.guidance_embed && !in->guidance) FAIL(VC_ERR_ARG, "Didn't get guidance strength for guidance distilled model.");
  if (in->kv_gap && !in->kv_len) FAIL(VC_ERR_ARG, "flux_prepare: kv_gap needs kv_len");
  if ((uintptr_t)workspace & 255) FAIL(VC_ERR_ARG, "flux_prepare: workspace must be 256-byte aligned");
  TRY(resolve(f, e));
  f.prepared = false;
  const int64_t need = carve(f, f, (char*)workspace, B, T, N, S);
  if (workspace_bytes < need) FAIL(VC_ERR_ARG, "flux_prepare: workspace of %lld bytes, %lld needed", (long long)workspace_bytes, (long long)need);
  f.base = (char*)workspace;
  f.B = B; f.T = T; f.N = N; f.L = T + N; f.Lp = (f.L + 63) / 64 * 64; f.S = S;
  const int L = f.L, D = f.D;
  // masks
  f.ragged = f.gapped = false;
  for (int b = 0; b < B; ++b) {
    const int kv = in->kv_len ? in->kv_len[b] : L;
    if (kv < 0 || kv > L) FAIL(VC_ERR_ARG, "flux_prepare: kv_len[%d] = %d outside [0, %d]", b, kv, L);
    if (kv < L) f.ragged = true;
    if (in->kv_gap) {
      const int lo = in->kv_gap[2 * b], hi = in->kv_gap[2 * b + 1];
      if (lo < 0 || lo > hi || hi > kv) FAIL(VC_ERR_ARG, "flux_prepare: kv_gap[%d] = (%d, %d) must lie inside [0, kv_len = %d)", b, lo, hi, kv);
      if (hi > lo) f.gapped = true;
    }
  }
  if (f.gapped) f.ragged = true;
  // host tables -> pinned staging -> HBM
  const size_t n_rope = (size_t)B * L * 128;
  TRY(stage_begin(f, (n_rope + 128 + 4 * B + 64) * sizeof(float) + 8 * 256, e));
  float* rope = stage_take<float>(f, n_rope);
  {  // RoPE angles in float64 exactly as math.py:102-109, stored as (cos, sin) f32 per (token, pair).  Position ids take few
     // distinct values per axis (grid rows / columns), so cos / sin are evaluated once per (axis, value, frequency).
    int pair0 = 0;
    std::vector<float> table;
    std::vector<int> slot((size_t)B * L);
    for (int ax = 0; ax < 3; ++ax) {
      const int half = f.cfg.axes_dim[ax] / 2, d = f.cfg.axes_dim[ax];
      std::unordered_map<float, int> seen;
      std::vector<float> values;
      for (int b = 0; b < B; ++b)
        for (int r = 0; r < L; ++r) {
          const float id = r < T ? in->txt_ids[((size_t)b * T + r) * 3 + ax] : in->img_ids[((size_t)b * N + (r - T)) * 3 + ax];
          auto it = seen.find(id);
          if (it == seen.end()) { it = seen.emplace(id, (int)values.size()).first; values.push_back(id); }
          slot[(size_t)b * L + r] = it->second;
        }
      table.resize(values.size() * (size_t)half * 2);
      for (int j = 0; j < half; ++j) {
        const double omega = 1.0 / pow((double)f.cfg.theta, (double)(2 * j) / (double)d);
        for (size_t v = 0; v < values.size(); ++v) {
          const double ang = (double)values[v] * omega;
          table[(v * half + j) * 2] = (float)cos(ang);
          table[(v * half + j) * 2 + 1] = (float)sin(ang);
        }
      }
      for (size_t row = 0; row < (size_t)B * L; ++row)
        memcpy(rope + (row * 64 + pair0) * 2, table.data() + (size_t)slot[row] * half * 2, (size_t)half * 2 * sizeof(float));
      pair0 += half;
    }
  }
  float* freqs = stage_take<float>(f, 128);
  for (int k = 0; k < 128; ++k)   // layers.py:41-43: exp(-ln(10000) * k / 128), the argument formed in f32 as torch forms it
    freqs[k] = (float)exp((double)((float)(-log(10000.0)) * (float)k / 128.0f));
  int32_t* kvl = stage_take<int32_t>(f, B);
  int32_t* gap = stage_take<int32_t>(f, 2 * B);
  float* g32 = stage_take<float>(f, B);
  for (int b = 0; b < B; ++b) {
    kvl[b] = in->kv_len ? in->kv_len[b] : L;
    gap[2 * b] = in->kv_gap ? in->kv_gap[2 * b] : 0;
    gap[2 * b + 1] = in->kv_gap ? in->kv_gap[2 * b + 1] : 0;
    g32[b] = in->guidance ? in->guidance[b] : 0.0f;
  }
  TRY(stage_send(f, f.ROPE, rope, n_rope * sizeof(float), s, e));
  {
    auto it = f.bound.find("timestep_freqs");     // the caller's own table (torch's f32 exp), bit for bit
    if (it != f.bound.end()) {
      if (it->second.N * it->second.K != 128) FAIL(VC_ERR_ARG, "flux_prepare: 'timestep_freqs' must hold 128 f32 values");
      TRY(d2d(f.FREQS, it->second.w, 128 * sizeof(float), s, e));
    } else {
      TRY(stage_send(f, f.FREQS, freqs, 128 * sizeof(float), s, e));
    }
  }
  TRY(stage_send(f, f.KVLEN, kvl, B * sizeof(int32_t), s, e));
  TRY(stage_send(f, f.KVGAP, gap, 2 * B * sizeof(int32_t), s, e));
  TRY(stage_send(f, f.G32, g32, B * sizeof(float), s, e));
  TRY(stage_end(f, s, e));
  HIP(hipMemsetAsync(f.VT, 0, (size_t)B * f.H * 128 * f.Lp * 2, s), "hipMemsetAsync");
  // the flag words of the attention kernel's in-launch combine: zero before the first launch (every launch leaves them zero)
  HIP(hipMemsetAsync((char*)f.ATT_SCRATCH + vc_attention_flags_offset_impl(), 0, (size_t)vc_attention64_flags_bytes_impl(f.n_cu), s), "hipMemsetAsync");
  // step-invariant projections
  TRY(lin(f, f.txt_in, in->txt, f.cfg.context_in_dim, f.TXT0, D, B * T, VC_EPI_BIAS, s, e));
  if (f.cfg.guidance_embed) {
    TRY(vc_temb_launch(f.G32, f.FREQS, f.GE, B, 128, in->guidance_is_bf16, s, e.buf, e.len));
    TRY(lin(f, f.guidance_in[0], f.GE, 256, f.GH, D, B, VC_EPI_SILU, s, e));
    TRY(lin(f, f.guidance_in[1], f.GH, D, f.GVEC, D, B, VC_EPI_BIAS, s, e));
  }
  TRY(lin(f, f.vector_in[0], in->y, f.cfg.vec_in_dim, f.YH, D, B, VC_EPI_SILU, s, e));
  TRY(lin(f, f.vector_in[1], f.YH, D, f.YVEC, D, B, VC_EPI_BIAS, s, e));
  f.prepared = true;
  f.steps_total = f.steps_done = 0;
  return VC_OK;
}

int vc_flux_forward_impl(void* handle, const void* img, const float* timesteps, int32_t timesteps_is_bf16, void* out, hipStream_t s,
                         char* err, int errlen) {
  H(handle);
  if (!f.prepared) FAIL(VC_ERR_STATE, "flux_forward: call vc_flux_prepare first");
  if (!img || !timesteps || !out) FAIL(VC_ERR_ARG, "flux_forward: null argument");
  TRY(stage_begin(f, f.B * sizeof(float) + 256, e));
  float* ts = stage_take<float>(f, f.B);
  for (int b = 0; b < f.B; ++b) ts[b] = timesteps[b];
  TRY(stage_send(f, f.TS, ts, f.B * sizeof(float), s, e));
  TRY(stage_end(f, s, e));
  TRY(time_precompute(f, 1, timesteps_is_bf16, s, e));
  f.steps_total = f.steps_done = 0;   // MOD now holds this evaluation's rows, not a trajectory's
  return evaluate(f, nullptr, false, img, out, false, s, e);
}

int vc_flux_sample_begin_impl(void* handle, const void* x, const void* cond, const float* t_grid, int32_t n_points, int32_t state_is_bf16,
                              hipStream_t s, char* err, int errlen) {
  H(handle);
  if (!f.prepared) FAIL(VC_ERR_STATE, "flux_sample: call vc_flux_prepare first");
  if (!x || !cond || !t_grid) FAIL(VC_ERR_ARG, "flux_sample: null argument");
  const int S = n_points - 1, B = f.B;
  if (S < 1 || S > f.S) FAIL(VC_ERR_ARG, "flux_sample: %d steps, the prepared workspace holds 1..%d", S, f.S);
  TRY(stage_begin(f, ((size_t)S * B + S) * sizeof(float) + 512, e));
  float* ts = stage_take<float>(f, (size_t)S * B);
  float* dts = stage_take<float>(f, S);
  for (int i = 0; i < S; ++i) {
    // the drift sees t_i in the state's dtype (torchdiffeq _PerturbFunc); the model sees 1 - t (transport.py:384)
    const float tm = 1.0f - (state_is_bf16 ? bf16_round(t_grid[i]) : t_grid[i]);
    for (int b = 0; b < B; ++b) ts[(size_t)i * B + b] = tm;
    dts[i] = t_grid[i + 1] - t_grid[i];   // fixed grid: dt = t1 - t0 in f32
  }
  TRY(stage_send(f, f.TS, ts, (size_t)S * B * sizeof(float), s, e));
  TRY(stage_send(f, f.DTS, dts, S * sizeof(float), s, e));
  TRY(stage_end(f, s, e));
  TRY(time_precompute(f, S, 0, s, e));
  const int64_t n = (int64_t)B * f.N;
  f.state_f32 = !state_is_bf16;
  if (f.state_f32) {
    TRY(d2d(f.XS32, x, n * f.cfg.out_channels * 4, s, e));
    TRY(vc_euler_f32_launch(f.XS32, f.XS, nullptr, nullptr, nullptr, n * f.cfg.out_channels, s, e.buf, e.len));   // XS = bf16(XS32)
  } else {
    TRY(d2d(f.XS, x, n * f.cfg.out_channels * 2, s, e));
  }
  TRY(d2d(f.COND, cond, n * (f.cfg.in_channels - f.cfg.out_channels) * 2, s, e));
  HIP(hipMemsetAsync(f.STEP, 0, sizeof(int32_t), s), "hipMemsetAsync");
  if (s) TRY(step_graph(f, s, e));
  f.steps_total = S; f.steps_done = 0;
  return VC_OK;
}

int vc_flux_sample_steps_impl(void* handle, int32_t n_steps, void* trajectory, hipStream_t s, char* err, int errlen) {
  H(handle);
  if (!f.prepared || f.steps_total == 0) FAIL(VC_ERR_STATE, "flux_sample_steps: call vc_flux_sample_begin first");
  if (n_steps < 0 || f.steps_done + n_steps > f.steps_total)
    FAIL(VC_ERR_ARG, "flux_sample_steps: %d more steps after %d of %d", n_steps, f.steps_done, f.steps_total);
  if (s && (!f.graph || f.key.s != s)) FAIL(VC_ERR_STATE, "flux_sample_steps: the step was captured on another stream");
  const int64_t state_bytes = (int64_t)f.B * f.N * f.cfg.out_channels * (f.state_f32 ? 4 : 2);
  const void* state = f.state_f32 ? (const void*)f.XS32 : (const void*)f.XS;
  for (int i = 0; i < n_steps; ++i) {
    if (s) HIP(hipGraphLaunch(f.graph, s), "hipGraphLaunch");
    else TRY(evaluate(f, f.STEP, true, nullptr, nullptr, true, s, e));
    if (trajectory) TRY(d2d((char*)trajectory + (int64_t)i * state_bytes, state, state_bytes, s, e));
    ++f.steps_done;
  }
  return VC_OK;
}

int vc_flux_sample_end_impl(void* handle, void* x_out, hipStream_t s, char* err, int errlen) {
  H(handle);
  if (!f.prepared || f.steps_total == 0) FAIL(VC_ERR_STATE, "flux_sample_end: no sample in flight");
  if (!x_out) FAIL(VC_ERR_ARG, "flux_sample_end: null output");
  if (f.state_f32) return d2d(x_out, f.XS32, (int64_t)f.B * f.N * f.cfg.out_channels * 4, s, e);
  return d2d(x_out, f.XS, (int64_t)f.B * f.N * f.cfg.out_channels * 2, s, e);
}

// HIP-event times of the launches of the product's plan, class by class (vcloze_hip.h): `evaluations` evaluations of the sample in
// flight at its current step, issued un-captured on `s` by the same code that the step graph was captured from, no Euler update
// (the trajectory's state, its step counter and its graph are left as they are).
int vc_flux_profile_impl(void* handle, int32_t evaluations, VcFluxLaunchClass* out, int32_t capacity, int32_t* count, hipStream_t s,
                         char* err, int errlen) {
  H(handle);
  if (!f.prepared || f.steps_total == 0) FAIL(VC_ERR_STATE, "flux_profile: call vc_flux_sample_begin first");
  if (!out || !count || capacity <= 0 || evaluations <= 0) FAIL(VC_ERR_ARG, "flux_profile: bad argument");
  TRY(evaluate(f, f.STEP, true, nullptr, nullptr, false, s, e));       // warm: caches and clocks as inside a trajectory
  f.prof_recs.clear();
  f.prof_used = 0;
  f.prof_on = true;
  int rc = VC_OK;
  for (int i = 0; i < evaluations && rc == VC_OK; ++i) rc = evaluate(f, f.STEP, true, nullptr, nullptr, false, s, e);
  f.prof_on = false;
  if (rc != VC_OK) return rc;
  HIP(hipStreamSynchronize(s), "hipStreamSynchronize");
  int n = 0;
  for (const auto& r : f.prof_recs) {
    float ms = 0.f;
    HIP(hipEventElapsedTime(&ms, f.prof_ev[r.e0], f.prof_ev[r.e0 + 1]), "hipEventElapsedTime");
    const float us = ms * 1e3f;
    int j = 0;
    while (j < n && !(out[j].kind == r.kind && out[j].epi == r.epi && out[j].n == r.n && out[j].k == r.k)) ++j;
    if (j == n) {
      if (n == capacity) FAIL(VC_ERR_ARG, "flux_profile: more than %d launch classes", capacity);
      memset(&out[n], 0, sizeof(out[n]));
      out[n].kind = r.kind; out[n].epi = r.epi; out[n].n = r.n; out[n].k = r.k;
      out[n].min_us = us; out[n].max_us = us;
      ++n;
    }
    out[j].launches += 1;
    out[j].flops += r.flops;
    out[j].bytes += r.bytes;
    out[j].total_us += us;
    if (us < out[j].min_us) out[j].min_us = us;
    if (us > out[j].max_us) out[j].max_us = us;
  }
  *count = n;
  return VC_OK;
}
