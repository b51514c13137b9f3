// C ABI of libvcloze_hip.so (declared in include/vcloze_hip.h): argument checks, per-thread error string,
// stream / hipGraph / event helpers.  No torch types cross this boundary.
#include "vcloze_internal.h"
#include <string.h>

static thread_local char g_err[512] = "";
#define ERRBUF g_err, (int)sizeof(g_err)
static inline hipStream_t S(void* s) { return (hipStream_t)s; }
static int hip_fail(const char* what, hipError_t e) {
  snprintf(g_err, sizeof(g_err), "%s: %s", what, hipGetErrorString(e));
  return VC_ERR_HIP;
}

extern "C" {

int vc_abi_version(void) { return VC_ABI_VERSION; }
const char* vc_last_error(void) { return g_err; }
void vc_struct_sizes(int32_t out[7]) {
  out[0] = (int32_t)sizeof(VcGemmProblem); out[1] = (int32_t)sizeof(VcGemmArgs); out[2] = (int32_t)sizeof(VcLnStream);
  out[3] = (int32_t)sizeof(VcAttention); out[4] = (int32_t)sizeof(VcFluxConfig); out[5] = (int32_t)sizeof(VcFluxInputs);
  out[6] = (int32_t)sizeof(VcFluxLaunchClass);
}

int vc_device_count(void) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) { hip_fail("hipGetDeviceCount", e); return 0; }
  return n;
}
int vc_device_info(int dev, char* name, int namelen, int* cu_count, int64_t* hbm_bytes) {
  hipDeviceProp_t p;
  hipError_t e = hipGetDeviceProperties(&p, dev);
  if (e != hipSuccess) return hip_fail("hipGetDeviceProperties", e);
  if (name && namelen > 0) { strncpy(name, p.gcnArchName, namelen - 1); name[namelen - 1] = 0; }
  if (cu_count) *cu_count = p.multiProcessorCount;
  if (hbm_bytes) *hbm_bytes = (int64_t)p.totalGlobalMem;
  return VC_OK;
}

int vc_gemm(const VcGemmArgs* args, int tile_cfg, void* stream) {
  if (!args) { snprintf(g_err, sizeof(g_err), "gemm: null args"); return VC_ERR_ARG; }
  return vc_gemm_launch(*args, tile_cfg, S(stream), ERRBUF);
}
int vc_gemm_plan(const VcGemmArgs* args, int tile_cfg, int32_t out[8]) {
  if (!args || !out) { snprintf(g_err, sizeof(g_err), "gemm_plan: null argument"); return VC_ERR_ARG; }
  return vc_gemm_plan_impl(*args, tile_cfg, out, ERRBUF);
}
int vc_ln_modulate(const void* x, int64_t ldx, void* y, int64_t ldy, const void* shift, const void* scale,
                   int64_t mod_bstride, int32_t rows, int32_t D, int32_t rows_per_batch, const int32_t* step_ptr,
                   int64_t mod_step_stride, void* stream) {
  return vc_ln_modulate_launch(x, ldx, y, ldy, shift, scale, mod_bstride, rows, D, rows_per_batch, step_ptr,
                               mod_step_stride, S(stream), ERRBUF);
}
int vc_ln_modulate2(const VcLnStream* a, const VcLnStream* b, int64_t mod_bstride, int32_t D, const int32_t* step_ptr,
                    int64_t mod_step_stride, void* stream) {
  return vc_ln_modulate2_launch(a, b, mod_bstride, D, step_ptr, mod_step_stride, S(stream), ERRBUF);
}
int vc_qknorm_rope_vt(void* qkv, int64_t ld, int64_t bstride, const void* q_scale, const void* k_scale,
                      const void* q_scale2, const void* k_scale2, int32_t split, const float* rope,
                      int64_t rope_bstride, void* vt, int32_t B, int32_t L, int32_t Lpad, int32_t H, int32_t parts, void* stream) {
  return vc_qknorm_rope_vt_launch(qkv, ld, bstride, q_scale, k_scale, q_scale2, k_scale2, split, rope, rope_bstride, vt, B, L, Lpad, H,
                                  parts, S(stream), ERRBUF);
}
int vc_attention(const VcAttention* a, void* stream) {
  if (!a) { snprintf(g_err, sizeof(g_err), "attention: null args"); return VC_ERR_ARG; }
  return vc_attention_launch(*a, S(stream), ERRBUF);
}
int64_t vc_attention_scratch_bytes(void) { return vc_attention_scratch_bytes_impl(); }
int vc_timestep_embedding(const float* t, const float* freqs, void* out_bf16, int32_t n, int32_t half,
                          int32_t round_t_bf16, void* stream) {
  return vc_temb_launch(t, freqs, out_bf16, n, half, round_t_bf16, S(stream), ERRBUF);
}
int vc_silu(const void* x, void* y, int64_t n, void* stream) { return vc_silu_launch(x, y, n, S(stream), ERRBUF); }
int vc_act2d(const void* x, int64_t ldx, void* y, int64_t ldy, int32_t rows, int32_t cols, int32_t act, void* stream) {
  return vc_act2d_launch(x, ldx, y, ldy, rows, cols, act, S(stream), ERRBUF);
}
int vc_gate_residual(const void* y, int64_t ldy, const void* res, int64_t ldres, const void* gate, void* out, int64_t ldo,
                     int32_t rows, int32_t cols, const int32_t* step_ptr, int64_t gate_step_stride, void* stream) {
  return vc_gate_residual_launch(y, ldy, res, ldres, gate, out, ldo, rows, cols, step_ptr, gate_step_stride, S(stream), ERRBUF);
}
int vc_add3(const void* a, const void* b, const void* c, void* y, int64_t n, int64_t bn, int64_t cn, void* stream) {
  return vc_add3_launch(a, b, c, y, n, bn, cn, S(stream), ERRBUF);
}
int vc_copy(void* dst, const void* src, int64_t bytes, void* stream) {
  if (!dst || !src || bytes <= 0) { snprintf(g_err, sizeof(g_err), "copy: bad args"); return VC_ERR_ARG; }
  hipError_t e = hipMemcpyAsync(dst, src, (size_t)bytes, hipMemcpyDeviceToDevice, S(stream));
  return e == hipSuccess ? VC_OK : hip_fail("hipMemcpyAsync", e);
}
int vc_concat_cols(const void* x, int32_t cx, const void* cond, int32_t cc, void* out, int64_t rows, void* stream) {
  return vc_concat_cols_launch(x, cx, cond, cc, out, rows, S(stream), ERRBUF);
}
int vc_euler_step(void* x, const void* v, const float* dts, const int32_t* step_ptr, int64_t n, void* stream) {
  return vc_euler_launch(x, v, dts, step_ptr, n, S(stream), ERRBUF);
}
int vc_euler_step_f32(float* x32, void* shadow, const void* v, const float* dts, const int32_t* step_ptr, int64_t n, void* stream) {
  return vc_euler_f32_launch(x32, shadow, v, dts, step_ptr, n, S(stream), ERRBUF);
}
int vc_step_advance(int32_t* step_ptr, void* stream) { return vc_step_advance_launch(step_ptr, S(stream), ERRBUF); }

int vc_sdedit_mix(const void* noise, const void* latent, float strength, void* out, int64_t n, void* stream) {
  return vc_sdedit_mix_launch(noise, latent, strength, out, n, S(stream), ERRBUF);
}
int vc_im2col3x3(const void* src, void* dst, int32_t H, int32_t W, int32_t C, int32_t up, void* stream) {
  return vc_im2col3x3_launch(src, dst, H, W, C, up, S(stream), ERRBUF);
}
int vc_groupnorm(const void* x, const void* gamma, const void* beta, void* y, void* scratch, int64_t scratch_bytes,
                 int64_t HW, int32_t C, int32_t G, float eps, int32_t swish, void* stream) {
  return vc_groupnorm_launch(x, gamma, beta, y, scratch, scratch_bytes, HW, C, G, eps, swish, S(stream), ERRBUF);
}
int vc_softmax_rows(void* x, int64_t ld, int32_t rows, int32_t cols, float scale, const void* bias, int64_t ld_bias,
                    int32_t causal_period, void* stream) {
  return vc_softmax_rows_launch(x, ld, rows, cols, scale, bias, ld_bias, causal_period, S(stream), ERRBUF);
}
int vc_embedding(const int32_t* ids, const void* table, int64_t ld_table, int32_t vocab, void* out, int32_t L, int32_t D, void* stream) {
  return vc_embedding_launch(ids, table, ld_table, vocab, out, L, D, S(stream), ERRBUF);
}
int vc_rmsnorm(const void* x, const void* weight, void* y, int32_t rows, int32_t D, float eps, void* stream) {
  return vc_rownorm_launch(x, weight, nullptr, y, rows, D, eps, 0, S(stream), ERRBUF);
}
int vc_layernorm(const void* x, const void* weight, const void* bias, void* y, int32_t rows, int32_t D, float eps, void* stream) {
  return vc_rownorm_launch(x, weight, bias, y, rows, D, eps, 1, S(stream), ERRBUF);
}
int vc_mul(const void* a, const void* b, void* y, int64_t n, void* stream) { return vc_ewise_launch(a, b, y, n, 0, S(stream), ERRBUF); }
int vc_add(const void* a, const void* b, void* y, int64_t n, void* stream) { return vc_ewise_launch(a, b, y, n, 1, S(stream), ERRBUF); }
int vc_quick_gelu(const void* x, void* y, int64_t n, void* stream) { return vc_ewise_launch(x, nullptr, y, n, 2, S(stream), ERRBUF); }
int vc_transpose(const void* src, int64_t ld_src, void* dst, int64_t ld_dst, int32_t rows, int32_t cols, void* stream) {
  return vc_transpose_launch(src, ld_src, dst, ld_dst, rows, cols, S(stream), ERRBUF);
}
int vc_nchw_to_nhwc(const void* src, int32_t src_is_f32, void* dst, int32_t C, int32_t Cp, int64_t HW, float div, float add, void* stream) {
  return vc_nchw_to_nhwc_launch(src, src_is_f32, dst, C, Cp, HW, div, add, S(stream), ERRBUF);
}
int vc_nhwc_to_nchw(const void* src, void* dst, int32_t dst_is_f32, int32_t C, int32_t Cp, int64_t HW, void* stream) {
  return vc_nhwc_to_nchw_launch(src, dst, dst_is_f32, C, Cp, HW, S(stream), ERRBUF);
}
int vc_gaussian_sample(const void* moments, int32_t Cp, const void* noise, void* out, int32_t Z, int64_t HW, float scale, float shift, void* stream) {
  return vc_gaussian_sample_launch(moments, Cp, noise, out, Z, HW, scale, shift, S(stream), ERRBUF);
}
int vc_conv3x3(const void* x, const void* w, const void* bias, void* out, int64_t ldc, const void* res, int64_t ldres,
               const void* gate, int32_t H, int32_t W, int32_t C, int32_t O, int32_t mode, void* stream) {
  return vc_conv3x3_launch(x, w, bias, out, ldc, res, ldres, gate, H, W, C, O, mode, S(stream), ERRBUF);
}
int vc_pack_latent(const void* latent, void* tokens, int32_t C, int32_t h, int32_t w, int64_t ld, int32_t col0, void* stream) {
  return vc_pack_latent_launch(latent, tokens, C, h, w, ld, col0, S(stream), ERRBUF);
}
int vc_pack_mask(const void* mask, void* tokens, int32_t H, int32_t W, int64_t ld, int32_t col0, void* stream) {
  return vc_pack_mask_launch(mask, tokens, H, W, ld, col0, S(stream), ERRBUF);
}
int vc_unpack_latent(const void* tokens, int64_t ld, int32_t col0, void* latent, int32_t C, int32_t h, int32_t w, void* stream) {
  return vc_unpack_latent_launch(tokens, ld, col0, latent, C, h, w, S(stream), ERRBUF);
}

/* ---- handle API (flux_engine.hip) ---- */
int vc_flux_create(const VcFluxConfig* cfg, void** handle) { return vc_flux_create_impl(cfg, handle, ERRBUF); }
int vc_flux_destroy(void* handle) { return vc_flux_destroy_impl(handle, ERRBUF); }
int vc_flux_bind_weight(void* handle, const char* name, const void* w, const void* bias, int32_t rows, int32_t cols, int64_t ldw) {
  return vc_flux_bind_weight_impl(handle, name, w, bias, rows, cols, ldw, ERRBUF);
}
int64_t vc_flux_mod_offset(void* handle, const char* module_name) { return vc_flux_mod_offset_impl(handle, module_name); }
int vc_flux_set_option(void* handle, const char* name, int32_t value) { return vc_flux_set_option_impl(handle, name, value, ERRBUF); }
int64_t vc_flux_workspace_bytes(void* handle, int32_t B, int32_t T, int32_t N, int32_t max_steps) {
  return vc_flux_workspace_bytes_impl(handle, B, T, N, max_steps);
}
int vc_flux_prepare(void* handle, const VcFluxInputs* in, void* workspace, int64_t workspace_bytes, void* stream) {
  return vc_flux_prepare_impl(handle, in, workspace, workspace_bytes, S(stream), ERRBUF);
}
int vc_flux_forward(void* handle, const void* img, const float* timesteps, int32_t timesteps_is_bf16, void* out, void* stream) {
  return vc_flux_forward_impl(handle, img, timesteps, timesteps_is_bf16, out, S(stream), ERRBUF);
}
int vc_flux_sample_begin(void* handle, const void* x, const void* cond, const float* t_grid, int32_t n_points, int32_t state_is_bf16,
                         void* stream) {
  return vc_flux_sample_begin_impl(handle, x, cond, t_grid, n_points, state_is_bf16, S(stream), ERRBUF);
}
int vc_flux_sample_steps(void* handle, int32_t n_steps, void* trajectory, void* stream) {
  return vc_flux_sample_steps_impl(handle, n_steps, trajectory, S(stream), ERRBUF);
}
int vc_flux_profile(void* handle, int32_t evaluations, VcFluxLaunchClass* out, int32_t capacity, int32_t* count, void* stream) {
  return vc_flux_profile_impl(handle, evaluations, out, capacity, count, S(stream), ERRBUF);
}
int vc_flux_sample_end(void* handle, void* x_out, void* stream) { return vc_flux_sample_end_impl(handle, x_out, S(stream), ERRBUF); }
int vc_flux_sample_euler(void* handle, void* x, const void* cond, const float* t_grid, int32_t n_points, int32_t state_is_bf16,
                         void* trajectory, void* stream) {
  int rc = vc_flux_sample_begin_impl(handle, x, cond, t_grid, n_points, state_is_bf16, S(stream), ERRBUF);
  if (rc == VC_OK) rc = vc_flux_sample_steps_impl(handle, n_points - 1, trajectory, S(stream), ERRBUF);
  if (rc == VC_OK) rc = vc_flux_sample_end_impl(handle, x, S(stream), ERRBUF);
  return rc;
}

/* ---- streams / graphs / events ---- */
int vc_stream_create(void** stream) {
  if (!stream) { snprintf(g_err, sizeof(g_err), "stream_create: null"); return VC_ERR_ARG; }
  hipStream_t s;
  hipError_t e = hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  if (e != hipSuccess) return hip_fail("hipStreamCreate", e);
  *stream = (void*)s;
  return VC_OK;
}
int vc_stream_destroy(void* stream) {
  hipError_t e = hipStreamDestroy(S(stream));
  return e == hipSuccess ? VC_OK : hip_fail("hipStreamDestroy", e);
}
int vc_stream_sync(void* stream) {
  hipError_t e = hipStreamSynchronize(S(stream));
  return e == hipSuccess ? VC_OK : hip_fail("hipStreamSynchronize", e);
}
int vc_graph_begin(void* stream) {
  hipError_t e = hipStreamBeginCapture(S(stream), hipStreamCaptureModeThreadLocal);
  return e == hipSuccess ? VC_OK : hip_fail("hipStreamBeginCapture", e);
}
int vc_graph_end(void* stream, void** graph_exec) {
  if (!graph_exec) { snprintf(g_err, sizeof(g_err), "graph_end: null"); return VC_ERR_ARG; }
  hipGraph_t g = nullptr;
  hipError_t e = hipStreamEndCapture(S(stream), &g);
  if (e != hipSuccess) return hip_fail("hipStreamEndCapture", e);
  hipGraphExec_t ge = nullptr;
  e = hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
  (void)hipGraphDestroy(g);
  if (e != hipSuccess) return hip_fail("hipGraphInstantiate", e);
  *graph_exec = (void*)ge;
  return VC_OK;
}
int vc_graph_launch(void* graph_exec, void* stream) {
  hipError_t e = hipGraphLaunch((hipGraphExec_t)graph_exec, S(stream));
  return e == hipSuccess ? VC_OK : hip_fail("hipGraphLaunch", e);
}
int vc_graph_destroy(void* graph_exec) {
  hipError_t e = hipGraphExecDestroy((hipGraphExec_t)graph_exec);
  return e == hipSuccess ? VC_OK : hip_fail("hipGraphExecDestroy", e);
}
int vc_event_create(void** ev) {
  if (!ev) { snprintf(g_err, sizeof(g_err), "event_create: null"); return VC_ERR_ARG; }
  hipEvent_t e_;
  hipError_t e = hipEventCreate(&e_);
  if (e != hipSuccess) return hip_fail("hipEventCreate", e);
  *ev = (void*)e_;
  return VC_OK;
}
int vc_event_record(void* ev, void* stream) {
  hipError_t e = hipEventRecord((hipEvent_t)ev, S(stream));
  return e == hipSuccess ? VC_OK : hip_fail("hipEventRecord", e);
}
int vc_event_elapsed_ms(void* ev_start, void* ev_stop, float* ms) {
  hipError_t e = hipEventSynchronize((hipEvent_t)ev_stop);
  if (e != hipSuccess) return hip_fail("hipEventSynchronize", e);
  e = hipEventElapsedTime(ms, (hipEvent_t)ev_start, (hipEvent_t)ev_stop);
  return e == hipSuccess ? VC_OK : hip_fail("hipEventElapsedTime", e);
}
int vc_event_destroy(void* ev) {
  hipError_t e = hipEventDestroy((hipEvent_t)ev);
  return e == hipSuccess ? VC_OK : hip_fail("hipEventDestroy", e);
}

}  // extern "C"
