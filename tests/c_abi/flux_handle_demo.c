/* A host program in plain C99 that drives the WHOLE denoising path through the handle API of include/vcloze_hip.h -
 * no Python, no torch, no C++: what a maintainer binding the library from another language would write.
 *
 *   flux_handle_demo <out.bin>
 *
 * Builds a tiny Flux (FluxParams of tests/procedural.py::TINY) with procedural bf16 weights, binds them by reference
 * module path, prepares one sample (T = 16 text tokens, a 2-row grid of 2 x 6 latent tokens each), runs ONE evaluation
 * (vc_flux_forward = Flux.forward, models/model.py:85-124) and a 4-step Euler trajectory (vc_flux_sample_euler =
 * transport/integrators.py:106-120) and writes  forward [24 x 64] | final state [24 x 64] | trajectory [4 x 24 x 64]
 * as raw bf16.  tests/test_c_abi_gpu.py builds it with gcc, runs it on the GPU and compares the file bit for bit with the
 * same calls made from Python over the same procedural weights (value formula below = tests/test_c_abi_gpu.py::fill).
 */
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "vcloze_hip.h"

enum { IN_CH = 384, OUT_CH = 64, VEC = 64, CTX = 128, D = 256, HEADS = 2, DEPTH = 2, SINGLE = 2, MLP = 1024, T = 16, N = 24, STEPS = 4 };

#define CHECK_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)
#define CHECK_VC(x) do { int rc_ = (x); if (rc_ != 0) { fprintf(stderr, "%s -> %d: %s\n", #x, rc_, vc_last_error()); exit(3); } } while (0)

/* value k of tensor `seed`: an LCG word -> [-1, 1) -> * scale -> bf16 (round to nearest even) */
static uint16_t bf16_bits(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}
static uint16_t value(uint32_t seed, uint32_t k, float scale) {
  const uint32_t w = seed * 1664525u + k * 1013904223u + 12345u;
  const float f = ((float)(int32_t)(w >> 16) - 32768.0f) * (1.0f / 32768.0f);
  return bf16_bits(f * scale);
}
static uint32_t name_seed(const char* s) {   /* FNV-1a */
  uint32_t h = 2166136261u;
  for (; *s; ++s) { h ^= (uint8_t)*s; h *= 16777619u; }
  return h;
}
static void* dev_fill(const char* name, size_t count, float scale, float offset) {
  uint16_t* h = (uint16_t*)malloc(count * 2);
  const uint32_t seed = name_seed(name);
  for (size_t k = 0; k < count; ++k) {
    uint16_t b = value(seed, (uint32_t)k, scale);
    if (offset != 0.0f) { uint32_t u = (uint32_t)b << 16; float f; memcpy(&f, &u, 4); b = bf16_bits(f + offset); }
    h[k] = b;
  }
  void* d = NULL;
  CHECK_HIP(hipMalloc(&d, count * 2));
  CHECK_HIP(hipMemcpy(d, h, count * 2, hipMemcpyHostToDevice));
  free(h);
  return d;
}

static void* handle;
static void bind_linear(const char* name, int rows, int cols) {
  char nm[160];
  snprintf(nm, sizeof(nm), "%s.weight", name);
  void* w = dev_fill(nm, (size_t)rows * cols, 0.06f, 0.0f);
  snprintf(nm, sizeof(nm), "%s.bias", name);
  void* b = dev_fill(nm, (size_t)rows, 0.05f, 0.0f);
  CHECK_VC(vc_flux_bind_weight(handle, name, w, b, rows, cols, cols));
}
static void bind_scale(const char* name) {
  void* w = dev_fill(name, 128, 0.1f, 1.0f);            /* QKNorm scales around 1 */
  CHECK_VC(vc_flux_bind_weight(handle, name, w, NULL, 1, 128, 128));
}

int main(int argc, char** argv) {
  if (argc < 2) { fprintf(stderr, "usage: %s out.bin\n", argv[0]); return 1; }
  int32_t sizes[7];
  vc_struct_sizes(sizes);
  if (vc_abi_version() != VC_ABI_VERSION || sizes[4] != (int32_t)sizeof(VcFluxConfig) || sizes[5] != (int32_t)sizeof(VcFluxInputs)) {
    fprintf(stderr, "library / header mismatch\n");
    return 1;
  }
  VcFluxConfig cfg = {IN_CH, OUT_CH, VEC, CTX, D, HEADS, DEPTH, SINGLE, MLP, 1, {16, 56, 56}, 10000};
  CHECK_VC(vc_flux_create(&cfg, &handle));

  /* ---- weights, by reference module path ---- */
  bind_linear("img_in", D, IN_CH);
  bind_linear("txt_in", D, CTX);
  bind_linear("time_in.in_layer", D, 256);      bind_linear("time_in.out_layer", D, D);
  bind_linear("vector_in.in_layer", D, VEC);    bind_linear("vector_in.out_layer", D, D);
  bind_linear("guidance_in.in_layer", D, 256);  bind_linear("guidance_in.out_layer", D, D);
  bind_linear("final_layer.linear", OUT_CH, D);
  char nm[160];
  for (int i = 0; i < DEPTH; ++i) {
    const char* st[2] = {"img", "txt"};
    for (int k = 0; k < 2; ++k) {
      snprintf(nm, sizeof(nm), "double_blocks.%d.%s_attn.qkv", i, st[k]);  bind_linear(nm, 3 * D, D);
      snprintf(nm, sizeof(nm), "double_blocks.%d.%s_attn.proj", i, st[k]); bind_linear(nm, D, D);
      snprintf(nm, sizeof(nm), "double_blocks.%d.%s_mlp.0", i, st[k]);     bind_linear(nm, MLP, D);
      snprintf(nm, sizeof(nm), "double_blocks.%d.%s_mlp.2", i, st[k]);     bind_linear(nm, D, MLP);
      snprintf(nm, sizeof(nm), "double_blocks.%d.%s_attn.norm.query_norm.scale", i, st[k]); bind_scale(nm);
      snprintf(nm, sizeof(nm), "double_blocks.%d.%s_attn.norm.key_norm.scale", i, st[k]);   bind_scale(nm);
    }
  }
  for (int i = 0; i < SINGLE; ++i) {
    snprintf(nm, sizeof(nm), "single_blocks.%d.linear1", i); bind_linear(nm, 3 * D + MLP, D);
    snprintf(nm, sizeof(nm), "single_blocks.%d.linear2", i); bind_linear(nm, D, D + MLP);
    snprintf(nm, sizeof(nm), "single_blocks.%d.norm.query_norm.scale", i); bind_scale(nm);
    snprintf(nm, sizeof(nm), "single_blocks.%d.norm.key_norm.scale", i);   bind_scale(nm);
  }
  /* every Modulation / adaLN Linear stacked in the library's order: one [n_mod, D] matrix */
  const int64_t n_mod = vc_flux_mod_offset(handle, NULL);
  if (n_mod != (int64_t)DEPTH * 12 * D + (int64_t)SINGLE * 3 * D + 2 * D ||
      vc_flux_mod_offset(handle, "double_blocks.1.txt_mod.lin") != 18 * D || vc_flux_mod_offset(handle, "final_layer.adaLN_modulation.1") != n_mod - 2 * D) {
    fprintf(stderr, "unexpected modulation layout\n");
    return 1;
  }
  bind_linear("modulation", (int)n_mod, D);

  /* ---- one sample ---- */
  void* txt = dev_fill("input.txt", (size_t)T * CTX, 1.0f, 0.0f);
  void* y = dev_fill("input.y", VEC, 1.0f, 0.0f);
  void* x = dev_fill("input.x", (size_t)N * OUT_CH, 1.0f, 0.0f);
  void* cond = dev_fill("input.cond", (size_t)N * (IN_CH - OUT_CH), 1.0f, 0.0f);
  float img_ids[N * 3], txt_ids[T * 3], guidance[1] = {30.0f};
  memset(txt_ids, 0, sizeof(txt_ids));
  for (int r = 0; r < N; ++r) {            /* two grid rows of 2 x 6 tokens: (row index + 1, y, x), models/sampling.py:47-60 */
    img_ids[3 * r] = (float)(r / 12 + 1);
    img_ids[3 * r + 1] = (float)((r % 12) / 6);
    img_ids[3 * r + 2] = (float)(r % 6);
  }
  const int64_t ws_bytes = vc_flux_workspace_bytes(handle, 1, T, N, STEPS);
  void* ws = NULL;
  CHECK_HIP(hipMalloc(&ws, (size_t)ws_bytes));
  hipStream_t stream;
  CHECK_HIP(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
  VcFluxInputs in;
  memset(&in, 0, sizeof(in));
  in.B = 1; in.T = T; in.N = N; in.max_steps = STEPS;
  in.txt = txt; in.y = y; in.guidance = guidance; in.img_ids = img_ids; in.txt_ids = txt_ids;
  CHECK_VC(vc_flux_prepare(handle, &in, ws, ws_bytes, stream));

  /* ---- Flux.forward: one evaluation of x || cond at t = 0.7 ---- */
  void *img = NULL, *fwd = NULL, *traj = NULL;
  CHECK_HIP(hipMalloc(&img, (size_t)N * IN_CH * 2));
  CHECK_HIP(hipMalloc(&fwd, (size_t)N * OUT_CH * 2));
  CHECK_HIP(hipMalloc(&traj, (size_t)STEPS * N * OUT_CH * 2));
  CHECK_VC(vc_concat_cols(x, OUT_CH, cond, IN_CH - OUT_CH, img, N, stream));
  const float t07[1] = {0.7f};
  CHECK_VC(vc_flux_forward(handle, img, t07, 0, fwd, stream));

  /* ---- the Euler loop: 5 solver points from 0 to 1, bf16 state, x updated in place ---- */
  const float grid[STEPS + 1] = {0.0f, 0.25f, 0.5f, 0.75f, 1.0f};
  CHECK_VC(vc_flux_sample_euler(handle, x, cond, grid, STEPS + 1, 1, traj, stream));
  CHECK_HIP(hipStreamSynchronize(stream));

  uint16_t* host = (uint16_t*)malloc((size_t)(2 + STEPS) * N * OUT_CH * 2);
  CHECK_HIP(hipMemcpy(host, fwd, (size_t)N * OUT_CH * 2, hipMemcpyDeviceToHost));
  CHECK_HIP(hipMemcpy(host + N * OUT_CH, x, (size_t)N * OUT_CH * 2, hipMemcpyDeviceToHost));
  CHECK_HIP(hipMemcpy(host + 2 * N * OUT_CH, traj, (size_t)STEPS * N * OUT_CH * 2, hipMemcpyDeviceToHost));
  FILE* f = fopen(argv[1], "wb");
  if (!f || fwrite(host, 2, (size_t)(2 + STEPS) * N * OUT_CH, f) != (size_t)(2 + STEPS) * N * OUT_CH) { fprintf(stderr, "cannot write %s\n", argv[1]); return 1; }
  fclose(f);
  CHECK_VC(vc_flux_destroy(handle));
  printf("flux_handle_demo: wrote %d bf16 values\n", (2 + STEPS) * N * OUT_CH);
  return 0;
}
