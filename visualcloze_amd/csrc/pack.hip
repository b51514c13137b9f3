// Latent-grid packer / unpacker: the pure data movement either side of the sampling loop.
//   pack_latent    [C, h, w] -> tokens [(h/2)(w/2), C*4]   "c (h ph) (w pw) -> (h w) (c ph pw)", ph = pw = 2
//                  (models/sampling.py:61, visualcloze.py:208-209,385-386)
//   pack_mask      pixel mask [H, W] -> [(H/16)(W/16), 256]: 8x8 pixel-unshuffle then the same 2x2 packing
//                  (visualcloze.py:206-207,381-382)
//   unpack_latent  tokens -> [C, h, w]                      (visualcloze.py:237,428)
// HBM-bound, bit-exact (bf16 copies).  COALESCED ON BOTH SIDES: the token layout and the map layout disagree about what is
// contiguous (a token row holds all channels of one 2x2 patch, a map row one channel of many patches), so each workgroup moves a
// tile of TW tokens of one token row through LDS - consecutive lanes read consecutive w of the map (runs of 4 TW bytes per
// (channel, sub-row); 16 B per lane for the pixel mask) and write consecutive 16-B chunks of whole token rows, or the other way
// round for unpack.  (Round 3's kernels read 4-B words at a stride of h*w elements and left the gathering to L2.)
#include "common.h"
#include "vcloze_internal.h"

namespace {

constexpr int TW = 64;          // tokens of one token row per workgroup
constexpr int LDW = TW + 1;     // LDS row stride in 32-bit words (+1: the transposing pass walks rows)

// out[tok * ld + col0 + c*4 + ph*2 + pw] = in[c][2*hh+ph][2*ww+pw],  tok = hh*(w/2) + ww
__global__ __launch_bounds__(256) void pack_latent_kernel(const bf16_t* __restrict__ in, bf16_t* __restrict__ out, int C, int h, int w,
                                                         long ld, int col0) {
  extern __shared__ uint32_t lds[];                    // [C * 2 (c, ph)][LDW] words = (pw 0, pw 1) of one token
  const int w2 = w >> 1, hh = blockIdx.y, ww0 = blockIdx.x * TW, nt = min(TW, w2 - ww0);
  for (int i = threadIdx.x; i < C * 2 * TW; i += 256) {          // lanes walk w: one 4-B word per token, 256-B runs per wave
    const int row = i / TW, t = i - row * TW;
    if (t < nt) lds[row * LDW + t] = *(const uint32_t*)(in + ((long)(row >> 1) * h + 2 * hh + (row & 1)) * w + 2 * (ww0 + t));
  }
  __syncthreads();
  const int cpt = C >> 1;                                          // 16-B chunks (2 channels) per token
  for (int i = threadIdx.x; i < nt * cpt; i += 256) {              // lanes walk the chunks of whole token rows
    const int t = i / cpt, ck = i - t * cpt;
    u32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = lds[(ck * 4 + e) * LDW + t];      // rows (2ck, ph 0), (2ck, ph 1), (2ck+1, ph 0), (2ck+1, ph 1)
    *(u32x4*)(out + ((long)hh * w2 + ww0 + t) * ld + col0 + ck * 8) = o;
  }
}

// tokens -> [C, h, w]: the same tile the other way round
__global__ __launch_bounds__(256) void unpack_latent_kernel(const bf16_t* __restrict__ in, long ld, int col0, bf16_t* __restrict__ out, int C,
                                                           int h, int w) {
  extern __shared__ uint32_t lds[];
  const int w2 = w >> 1, hh = blockIdx.y, ww0 = blockIdx.x * TW, nt = min(TW, w2 - ww0), cpt = C >> 1;
  for (int i = threadIdx.x; i < nt * cpt; i += 256) {
    const int t = i / cpt, ck = i - t * cpt;
    const u32x4 v = *(const u32x4*)(in + ((long)hh * w2 + ww0 + t) * ld + col0 + ck * 8);
#pragma unroll
    for (int e = 0; e < 4; ++e) lds[(ck * 4 + e) * LDW + t] = v[e];
  }
  __syncthreads();
  for (int i = threadIdx.x; i < C * 2 * TW; i += 256) {
    const int row = i / TW, t = i - row * TW;
    if (t < nt) *(uint32_t*)(out + ((long)(row >> 1) * h + 2 * hh + (row & 1)) * w + 2 * (ww0 + t)) = lds[row * LDW + t];
  }
}

// out[tok * ld + col0 + (p8*8+q8)*4 + ph*2 + pw] = in[(2*hh+ph)*8 + p8][(2*ww+pw)*8 + q8]: a token is a 16 x 16 pixel block;
// a workgroup stages the 16 pixel rows of TWM tokens (lanes walk x, 16 B each: 512-B runs) and writes their 512-B token rows
constexpr int TWM = 16, MLD = TWM * 16 + 8;       // (+8 elements: the 16 pixel rows start on different banks)
__global__ __launch_bounds__(256) void pack_mask_kernel(const bf16_t* __restrict__ in, bf16_t* __restrict__ out, int H, int W, long ld,
                                                       int col0) {
  __shared__ __attribute__((aligned(16))) bf16_t px[16 * MLD];
  const int w2 = W >> 4, hh = blockIdx.y, ww0 = blockIdx.x * TWM, nt = min(TWM, w2 - ww0);
  for (int i = threadIdx.x; i < 16 * TWM * 2; i += 256) {          // 16 rows x (TWM tokens x 2 groups of 8 pixels)
    const int r = i / (TWM * 2), g = i - r * (TWM * 2);
    if (g < nt * 2) *(u32x4*)(px + r * MLD + g * 8) = *(const u32x4*)(in + ((long)hh * 16 + r) * W + (long)ww0 * 16 + g * 8);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < nt * 32; i += 256) {               // 32 chunks of 16 B per token row
    const int t = i >> 5, ck = i & 31;
    bf16_t v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int c8 = ck * 2 + (e >> 2), p8 = c8 >> 3, q8 = c8 & 7, ph = (e >> 1) & 1, pw = e & 1;
      v[e] = px[(ph * 8 + p8) * MLD + (2 * t + pw) * 8 + q8];
    }
    u32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = (uint32_t)v[2 * e] | ((uint32_t)v[2 * e + 1] << 16);
    *(u32x4*)(out + ((long)hh * w2 + ww0 + t) * ld + col0 + ck * 8) = o;
  }
}

}  // namespace

#define PK_LAUNCH(name, kern, grid, lds_bytes, ...)                                                 \
  hipLaunchKernelGGL(kern, grid, dim3(256), lds_bytes, s, __VA_ARGS__);                             \
  {                                                                                                 \
    hipError_t e_ = hipGetLastError();                                                              \
    if (e_ != hipSuccess) { snprintf(err, errlen, name " launch: %s", hipGetErrorString(e_)); return VC_ERR_HIP; } \
  }                                                                                                 \
  return VC_OK;

int vc_pack_latent_launch(const void* in, void* out, int C, int h, int w, int64_t ld, int col0, hipStream_t s, char* err, int errlen) {
  if (!in || !out || C <= 0 || h <= 0 || w <= 0 || (C & 1) || (h & 1) || (w & 1) || ld % 8 || col0 % 8) {
    snprintf(err, errlen, "pack_latent: need even C, h, w and 16-B aligned ld / col0 (C=%d h=%d w=%d)", C, h, w); return VC_ERR_ARG; }
  if (C > 64) { snprintf(err, errlen, "pack_latent: C=%d > 64 channels", C); return VC_ERR_ARG; }
  PK_LAUNCH("pack_latent", pack_latent_kernel, dim3((w / 2 + TW - 1) / TW, h / 2), (size_t)C * 2 * LDW * 4, (const bf16_t*)in, (bf16_t*)out, C, h, w, (long)ld, col0)
}
int vc_pack_mask_launch(const void* in, void* out, int H, int W, int64_t ld, int col0, hipStream_t s, char* err, int errlen) {
  if (!in || !out || H <= 0 || W <= 0 || H % 16 || W % 16 || ld % 8 || col0 % 8) {
    snprintf(err, errlen, "pack_mask: H, W must be positive multiples of 16 (H=%d W=%d)", H, W); return VC_ERR_ARG; }
  if (((uintptr_t)in | (uintptr_t)out) & 15) {     // the kernel reads the mask and writes the tokens in 16-B vectors (advisor r04)
    snprintf(err, errlen, "pack_mask: the mask and the token rows must be 16-byte aligned"); return VC_ERR_ARG; }
  PK_LAUNCH("pack_mask", pack_mask_kernel, dim3((W / 16 + TWM - 1) / TWM, H / 16), 0, (const bf16_t*)in, (bf16_t*)out, H, W, (long)ld, col0)
}
int vc_unpack_latent_launch(const void* in, int64_t ld, int col0, void* out, int C, int h, int w, hipStream_t s, char* err, int errlen) {
  if (!in || !out || C <= 0 || h <= 0 || w <= 0 || (C & 1) || (h & 1) || (w & 1) || ld % 8 || col0 % 8) {
    snprintf(err, errlen, "unpack_latent: need even C, h, w and 16-B aligned ld / col0"); return VC_ERR_ARG; }
  if (C > 64) { snprintf(err, errlen, "unpack_latent: C=%d > 64 channels", C); return VC_ERR_ARG; }
  PK_LAUNCH("unpack_latent", unpack_latent_kernel, dim3((w / 2 + TW - 1) / TW, h / 2), (size_t)C * 2 * LDW * 4, (const bf16_t*)in, (long)ld, col0, (bf16_t*)out, C, h, w)
}
