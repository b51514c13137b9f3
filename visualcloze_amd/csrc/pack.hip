// Latent-grid packer / unpacker: the pure data movement either side of the sampling loop.
//   pack_latent    [C, h, w] -> tokens [(h/2)(w/2), C*4]   "c (h ph) (w pw) -> (h w) (c ph pw)", ph = pw = 2
//                  (models/sampling.py:61, visualcloze.py:208-209,385-386)
//   pack_mask      pixel mask [H, W] -> [(H/16)(W/16), 256]: 8x8 pixel-unshuffle then the same 2x2 packing
//                  (visualcloze.py:206-207,381-382)
//   unpack_latent  tokens -> [C, h, w]                      (visualcloze.py:237,428)
// HBM-bound, bit-exact (bf16 copies).  One thread produces one 16-B output chunk, so every wave-instruction
// writes whole token rows; the strided reads (runs of 2 / 8 elements) are absorbed by L2.
#include "common.h"
#include "vcloze_internal.h"

namespace {

// out[tok * ld + col0 + c*4 + ph*2 + pw] = in[c][2*hh+ph][2*ww+pw],  tok = hh*(w/2) + ww ; chunk = 2 channels
__global__ void pack_latent_kernel(const bf16_t* __restrict__ in, bf16_t* __restrict__ out, int C, int h, int w,
                                   long ld, int col0) {
  const int w2 = w >> 1, cpt = C >> 1;  // 16-B chunks per token
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)(h >> 1) * w2 * cpt) return;
  const int ck = (int)(i % cpt);
  const long tok = i / cpt;
  const int hh = (int)(tok / w2), ww = (int)(tok % w2);
  u32x4 o;
#pragma unroll
  for (int cc = 0; cc < 2; ++cc) {
    const int c = ck * 2 + cc;
    const bf16_t* p = in + ((long)c * h + 2 * hh) * w + 2 * ww;
    o[cc * 2 + 0] = *(const uint32_t*)p;        // (ph=0: pw=0,1)   2*ww even -> 4-B aligned when w is even
    o[cc * 2 + 1] = *(const uint32_t*)(p + w);  // (ph=1: pw=0,1)
  }
  *(u32x4*)(out + tok * ld + col0 + ck * 8) = o;
}

// out[tok * ld + col0 + (p8*8+q8)*4 + ph*2 + pw] = in[(2*hh+ph)*8 + p8][(2*ww+pw)*8 + q8]; chunk = 2 (p8,q8) pairs
__global__ void pack_mask_kernel(const bf16_t* __restrict__ in, bf16_t* __restrict__ out, int H, int W, long ld,
                                 int col0) {
  const int w2 = W >> 4;
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)(H >> 4) * w2 * 32) return;
  const int ck = (int)(i & 31);
  const long tok = i >> 5;
  const int hh = (int)(tok / w2), ww = (int)(tok % w2);
  bf16_t v[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int c8 = ck * 2 + (e >> 2), p8 = c8 >> 3, q8 = c8 & 7, ph = (e >> 1) & 1, pw = e & 1;
    v[e] = in[((long)(2 * hh + ph) * 8 + p8) * W + (2 * ww + pw) * 8 + q8];
  }
  u32x4 o;
#pragma unroll
  for (int e = 0; e < 4; ++e) o[e] = (uint32_t)v[2 * e] | ((uint32_t)v[2 * e + 1] << 16);
  *(u32x4*)(out + tok * ld + col0 + ck * 8) = o;
}

__global__ void unpack_latent_kernel(const bf16_t* __restrict__ in, long ld, int col0, bf16_t* __restrict__ out, int C,
                                     int h, int w) {
  const int w2 = w >> 1, cpt = C >> 1;
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)(h >> 1) * w2 * cpt) return;
  const int ck = (int)(i % cpt);
  const long tok = i / cpt;
  const int hh = (int)(tok / w2), ww = (int)(tok % w2);
  const u32x4 v = *(const u32x4*)(in + tok * ld + col0 + ck * 8);
#pragma unroll
  for (int cc = 0; cc < 2; ++cc) {
    bf16_t* p = out + ((long)(ck * 2 + cc) * h + 2 * hh) * w + 2 * ww;
    *(uint32_t*)p = v[cc * 2 + 0];
    *(uint32_t*)(p + w) = v[cc * 2 + 1];
  }
}

}  // namespace

#define PK_LAUNCH(name, kern, n, ...)                                                               \
  hipLaunchKernelGGL(kern, dim3((unsigned)(((n) + 255) / 256)), dim3(256), 0, s, __VA_ARGS__);     \
  {                                                                                                 \
    hipError_t e_ = hipGetLastError();                                                              \
    if (e_ != hipSuccess) { snprintf(err, errlen, name " launch: %s", hipGetErrorString(e_)); return VC_ERR_HIP; } \
  }                                                                                                 \
  return VC_OK;

int vc_pack_latent_launch(const void* in, void* out, int C, int h, int w, int64_t ld, int col0, hipStream_t s, char* err, int errlen) {
  if (!in || !out || C <= 0 || h <= 0 || w <= 0 || (C & 1) || (h & 1) || (w & 1) || ld % 8 || col0 % 8) {
    snprintf(err, errlen, "pack_latent: need even C, h, w and 16-B aligned ld / col0 (C=%d h=%d w=%d)", C, h, w); return VC_ERR_ARG; }
  const long n = (long)(h / 2) * (w / 2) * (C / 2);
  PK_LAUNCH("pack_latent", pack_latent_kernel, n, (const bf16_t*)in, (bf16_t*)out, C, h, w, (long)ld, col0)
}
int vc_pack_mask_launch(const void* in, void* out, int H, int W, int64_t ld, int col0, hipStream_t s, char* err, int errlen) {
  if (!in || !out || H <= 0 || W <= 0 || H % 16 || W % 16 || ld % 8 || col0 % 8) {
    snprintf(err, errlen, "pack_mask: H, W must be positive multiples of 16 (H=%d W=%d)", H, W); return VC_ERR_ARG; }
  const long n = (long)(H / 16) * (W / 16) * 32;
  PK_LAUNCH("pack_mask", pack_mask_kernel, n, (const bf16_t*)in, (bf16_t*)out, H, W, (long)ld, col0)
}
int vc_unpack_latent_launch(const void* in, int64_t ld, int col0, void* out, int C, int h, int w, hipStream_t s, char* err, int errlen) {
  if (!in || !out || C <= 0 || h <= 0 || w <= 0 || (C & 1) || (h & 1) || (w & 1) || ld % 8 || col0 % 8) {
    snprintf(err, errlen, "unpack_latent: need even C, h, w and 16-B aligned ld / col0"); return VC_ERR_ARG; }
  const long n = (long)(h / 2) * (w / 2) * (C / 2);
  PK_LAUNCH("unpack_latent", unpack_latent_kernel, n, (const bf16_t*)in, (long)ld, col0, (bf16_t*)out, C, h, w)
}
