// internal launcher declarations shared by the .hip translation units
#pragma once
#include <hip/hip_runtime.h>
#include <stdio.h>
#include "../../include/vcloze_hip.h"

// One-time per-(kernel, device) setup (hipFuncSetAttribute) and per-device properties: the handle API is a public C ABI, a host
// may drive several GPUs from one process, so nothing of this may be remembered per process only.
constexpr int VC_MAX_DEVICES = 64;
inline int vc_device_index() {
  int d = 0;
  (void)hipGetDevice(&d);
  return d >= 0 && d < VC_MAX_DEVICES ? d : 0;
}
struct VcOncePerDevice {
  bool done[VC_MAX_DEVICES] = {};
  bool need() const { return !done[vc_device_index()]; }
  void mark() { done[vc_device_index()] = true; }
};
inline int vc_cu_count() {          // compute units of the CURRENT device
  static int n[VC_MAX_DEVICES] = {};
  const int d = vc_device_index();
  if (n[d] == 0) {
    hipDeviceProp_t p;
    n[d] = hipGetDeviceProperties(&p, d) == hipSuccess && p.multiProcessorCount > 0 ? p.multiProcessorCount : 256;
  }
  return n[d];
}
int vc_gemm_launch(VcGemmArgs a, int tile_cfg, hipStream_t s, char* err, int errlen);
int vc_gemm_plan_impl(VcGemmArgs a, int tile_cfg, int32_t out[8], char* err, int errlen);
int vc_attention_launch(const VcAttention& a, hipStream_t s, char* err, int errlen);
int64_t vc_attention_scratch_bytes_impl();
int64_t vc_attention_flags_offset_impl();
int64_t vc_attention64_flags_bytes_impl(int n_cu);
int vc_attention64_launch(const VcAttention& a, bool tail_split, int n_cu, uint64_t* debug_ts, hipStream_t s, char* err, int errlen);
int64_t vc_attention64_scratch_bytes_impl(int n_cu);
int vc_ln_modulate2_launch(const VcLnStream* a, const VcLnStream* b, int64_t mod_bstride, int32_t D, const int32_t* step_ptr,
                           int64_t mod_step_stride, hipStream_t s, char* err, int errlen);
int vc_ln_modulate_launch(const void* x, int64_t ldx, void* y, int64_t ldy, const void* shift, const void* scale,
                          int64_t mod_bstride, int32_t rows, int32_t D, int32_t rows_per_batch,
                          const int32_t* step_ptr, int64_t mod_step_stride, hipStream_t s, char* err, int errlen);
int vc_qknorm_rope_vt_launch(void* qkv, int64_t ld, int64_t bstride, const void* q_scale, const void* k_scale,
                             const void* q_scale2, const void* k_scale2, int32_t split, const float* rope, int64_t rope_bstride, void* vt, int32_t B, int32_t L, int32_t Lpad,
                             int32_t H, int32_t parts, hipStream_t s, char* err, int errlen);
int vc_temb_launch(const float* t, const float* freqs, void* out, int n, int half, int round_t, hipStream_t s, char* err, int errlen);
int vc_silu_launch(const void* x, void* y, int64_t n, hipStream_t s, char* err, int errlen);
int vc_act2d_launch(const void* x, int64_t ldx, void* y, int64_t ldy, int32_t rows, int32_t cols, int32_t act, hipStream_t s,
                    char* err, int errlen);
int vc_gate_residual_launch(const void* y, int64_t ldy, const void* res, int64_t ldres, const void* gate, void* out, int64_t ldo,
                            int32_t rows, int32_t cols, const int32_t* step_ptr, int64_t gate_step_stride, hipStream_t s,
                            char* err, int errlen);
int vc_add3_launch(const void* a, const void* b, const void* c, void* y, int64_t n, int64_t bn, int64_t cn, hipStream_t s, char* err, int errlen);
int vc_concat_cols_launch(const void* x, int cx, const void* cond, int cc, void* out, int64_t rows, hipStream_t s, char* err, int errlen);
int vc_euler_launch(void* x, const void* v, const float* dts, const int32_t* step_ptr, int64_t n, hipStream_t s, char* err, int errlen);
int vc_euler_f32_launch(float* x32, void* shadow, const void* v, const float* dts, const int32_t* step_ptr, int64_t n, hipStream_t s,
                        char* err, int errlen);
int vc_step_advance_launch(int32_t* step_ptr, hipStream_t s, char* err, int errlen);
int vc_pack_latent_launch(const void* in, void* out, int C, int h, int w, int64_t ld, int col0, hipStream_t s, char* err, int errlen);
int vc_pack_mask_launch(const void* in, void* out, int H, int W, int64_t ld, int col0, hipStream_t s, char* err, int errlen);
int vc_unpack_latent_launch(const void* in, int64_t ld, int col0, void* out, int C, int h, int w, hipStream_t s, char* err, int errlen);
int vc_sdedit_mix_launch(const void* noise, const void* latent, float strength, void* out, int64_t n, hipStream_t s, char* err, int errlen);
int vc_im2col3x3_launch(const void* src, void* dst, int H, int W, int C, int up, hipStream_t s, char* err, int errlen);
int vc_groupnorm_launch(const void* x, const void* gamma, const void* beta, void* y, void* scratch, int64_t scratch_bytes,
                        int64_t HW, int C, int G, float eps, int swish, hipStream_t s, char* err, int errlen);
int vc_softmax_rows_launch(void* x, int64_t ld, int rows, int cols, float scale, const void* bias, int64_t ldb, int causal_period,
                           hipStream_t s, char* err, int errlen);
int vc_transpose_launch(const void* src, int64_t lds_, void* dst, int64_t ldd, int R, int Cc, hipStream_t s, char* err, int errlen);
int vc_nchw_to_nhwc_launch(const void* src, int src_f32, void* dst, int C, int Cp, int64_t HW, float div, float add, hipStream_t s, char* err, int errlen);
int vc_nhwc_to_nchw_launch(const void* src, void* dst, int dst_f32, int C, int Cp, int64_t HW, hipStream_t s, char* err, int errlen);
int vc_gaussian_sample_launch(const void* moments, int Cp, const void* noise, void* out, int Z, int64_t HW, float scale, float shift,
                              hipStream_t s, char* err, int errlen);
int vc_embedding_launch(const int32_t* ids, const void* table, int64_t ldt, int V, void* out, int L, int D, hipStream_t s, char* err, int errlen);
int vc_rownorm_launch(const void* x, const void* w, const void* b, void* y, int rows, int D, float eps, int affine_ln, hipStream_t s, char* err, int errlen);
int vc_ewise_launch(const void* a, const void* b, void* y, int64_t n, int op, hipStream_t s, char* err, int errlen);
int vc_conv3x3_launch(const void* x, const void* w, const void* bias, void* out, int64_t ldc, const void* res, int64_t ldres,
                      const void* gate, int H, int W, int C, int O, int mode, hipStream_t s, char* err, int errlen);

// flux_engine.hip: the handle API
int vc_flux_create_impl(const VcFluxConfig* cfg, void** handle, char* err, int errlen);
int vc_flux_destroy_impl(void* handle, char* err, int errlen);
int vc_flux_bind_weight_impl(void* handle, const char* name, const void* w, const void* bias, int32_t rows, int32_t cols, int64_t ldw,
                             char* err, int errlen);
int64_t vc_flux_mod_offset_impl(void* handle, const char* name);
int vc_flux_set_option_impl(void* handle, const char* name, int32_t value, char* err, int errlen);
int64_t vc_flux_workspace_bytes_impl(void* handle, int32_t B, int32_t T, int32_t N, int32_t max_steps);
int vc_flux_prepare_impl(void* handle, const VcFluxInputs* in, void* workspace, int64_t workspace_bytes, hipStream_t s, char* err, int errlen);
int vc_flux_forward_impl(void* handle, const void* img, const float* timesteps, int32_t timesteps_is_bf16, void* out, hipStream_t s,
                         char* err, int errlen);
int vc_flux_sample_begin_impl(void* handle, const void* x, const void* cond, const float* t_grid, int32_t n_points, int32_t state_is_bf16,
                              hipStream_t s, char* err, int errlen);
int vc_flux_sample_steps_impl(void* handle, int32_t n_steps, void* trajectory, hipStream_t s, char* err, int errlen);
int vc_flux_profile_impl(void* handle, int32_t evaluations, VcFluxLaunchClass* out, int32_t capacity, int32_t* count, hipStream_t s,
                         char* err, int errlen);
int vc_flux_sample_end_impl(void* handle, void* x_out, hipStream_t s, char* err, int errlen);
