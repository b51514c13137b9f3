"""Generates tools/ubench/a64_gap.hip: issue-slot price list for the one-wave-per-SIMD attention kernel
(visualcloze_amd/csrc/attention64.hip).  Every variant is a loop of 32 v_mfma_f32_32x32x16_bf16 in that kernel's operand
placement with a fixed filler recipe in each MFMA gap, one wave per SIMD; reports s_memtime ticks per MFMA.

    python tools/ubench/gen_a64_gap.py && hipcc --offload-arch=gfx950 -O3 tools/ubench/a64_gap.hip -o tools/ubench/a64_gap.bin
"""
import os

def mfma(kind, g):
    if kind == "qk":    # D = VGPR (4 chains), A = AGPR K fragment, B = AGPR Q fragment
        t, u, qb = g >> 2, (g >> 1) & 1, g & 1
        d = 64 + 16 * (qb * 2 + u)
        a = 192 + (u * 8 + t) * 4
        b = 128 + (qb * 8 + t) * 4
        return f"v_mfma_f32_32x32x16_bf16 v[{d}:{d+15}], a[{a}:{a+3}], a[{b}:{b+3}], v[{d}:{d+15}]"
    if kind == "pv":    # D = AGPR (8 chains, revisited every 2nd MFMA), A, B = VGPR
        dt, s, qb = g >> 3, (g >> 1) & 3, g & 1
        d = (qb * 4 + dt) * 16
        a = 128 + ((dt & 1) * 4 + s) * 4
        b = 160 + (qb * 4 + s) * 4
        return f"v_mfma_f32_32x32x16_bf16 a[{d}:{d+15}], v[{a}:{a+3}], v[{b}:{b+3}], a[{d}:{d+15}]"
    if kind == "qkchain":   # chain-major: 9 consecutive MFMAs on ONE accumulator (8 k-steps + the row-max step), 4 chains
        c = (g // 9) & 3
        t = g % 9
        d = 64 + 16 * c
        a = 192 + ((c & 1) * 8 + min(t, 7)) * 4
        b = 128 + ((c >> 1) * 8 + min(t, 7)) * 4
        cin = f"v[{d}:{d+15}]" if t else "0"
        return f"v_mfma_f32_32x32x16_bf16 v[{d}:{d+15}], a[{a}:{a+3}], a[{b}:{b+3}], {cin}"
    if kind == "vv":    # everything in VGPRs, 4 chains
        d = 64 + 16 * (g & 3)
        a = 128 + (g & 7) * 4
        b = 160 + (g & 7) * 4
        return f"v_mfma_f32_32x32x16_bf16 v[{d}:{d+15}], v[{a}:{a+3}], v[{b}:{b+3}], v[{d}:{d+15}]"
    raise ValueError(kind)

def fillers(recipe, g):
    out = []
    r = 200 + (g % 8) * 4        # rotating scratch registers v200..v231
    rp = 200 + ((g - 1) % 8) * 4   # registers written one gap earlier
    for op in recipe:
        if op == "addp": out.append(f"v_add_f32 v234, v{rp}, v234"); rp += 1      # consumes the PREVIOUS gap's exp
        elif op == "cvtp": out.append(f"v_cvt_pk_bf16_f32 v235, v{rp-1}, v{rp-2}")
        elif op == "fma": out.append(f"v_fma_f32 v{r}, v{r}, v232, v233"); r += 1
        elif op == "exp": out.append(f"v_exp_f32 v{r}, v{r}"); r += 1
        elif op == "add": out.append(f"v_add_f32 v234, v{r-1}, v234")
        elif op == "cvt": out.append(f"v_cvt_pk_bf16_f32 v235, v{r-1}, v{r-2}")
        elif op == "max3": out.append(f"v_max3_f32 v{r}, v{r}, v232, v233"); r += 1
        elif op == "nop": out.append("s_nop 0")
        elif op == "ldsv": out.append(f"ds_read_b128 v[236:239], v240 offset:{(g % 16) * 1024}")
        elif op == "ldsa": out.append(f"ds_read_b128 a[{240 + (g % 4) * 4}:{243 + (g % 4) * 4}], v240 offset:{(g % 16) * 1024}")
        elif op == "salu": out.append("s_add_i32 s20, s20, 1")
        else: raise ValueError(op)
    return out

VARIANTS = [
    ("qk, no fillers", "qk", []),
    ("pv, no fillers", "pv", []),
    ("vv (all-VGPR MFMA), no fillers", "vv", []),
    ("qk + 1 fma", "qk", ["fma"]),
    ("qk + 2 fma", "qk", ["fma"] * 2),
    ("qk + 3 fma", "qk", ["fma"] * 3),
    ("qk + 4 fma", "qk", ["fma"] * 4),
    ("qk + 5 fma", "qk", ["fma"] * 5),
    ("qk + 6 fma", "qk", ["fma"] * 6),
    ("qk + 8 fma", "qk", ["fma"] * 8),
    ("vv + 4 fma", "vv", ["fma"] * 4),
    ("vv + 6 fma", "vv", ["fma"] * 6),
    ("pv + 4 fma", "pv", ["fma"] * 4),
    ("pv + 6 fma", "pv", ["fma"] * 6),
    ("qk + 2 exp", "qk", ["exp"] * 2),
    ("qk + 4 exp", "qk", ["exp"] * 4),
    ("qk + phase-A mix (2 exp, 2 add, cvt)", "qk", ["exp", "exp", "add", "add", "cvt"]),
    ("qk + phase-A mix + 1 V read", "qk", ["exp", "exp", "add", "add", "cvt", "ldsv"]),
    ("pv + 4 max3", "pv", ["max3"] * 4),
    ("pv + 4 max3 + nop", "pv", ["nop"] + ["max3"] * 4),
    ("pv + 4 fma + K read (AGPR) + nop", "pv", ["nop"] + ["fma"] * 4 + ["ldsa"]),
    ("pv + 1 K read (AGPR)", "pv", ["ldsa"]),
    ("pv + 1 V read (VGPR)", "pv", ["ldsv"]),
    ("qkchain (9 dependent MFMAs per chain), no fillers", "qkchain", []),
    ("qkchain + exp addp exp addp cvtp", "qkchain", ["exp", "addp", "exp", "addp", "cvtp"]),
    ("qk + exp addp exp addp cvtp", "qk", ["exp", "addp", "exp", "addp", "cvtp"]),
    ("qk + exp exp addp addp cvtp", "qk", ["exp", "exp", "addp", "addp", "cvtp"]),
    ("qk + exp addp exp addp cvtp + V read", "qk", ["exp", "addp", "exp", "addp", "cvtp", "ldsv"]),
    ("qk + exp addp exp addp cvtp + fma", "qk", ["exp", "addp", "exp", "addp", "cvtp", "fma"]),
    ("qk + exp addp exp addp cvtp + 2 fma", "qk", ["exp", "addp", "exp", "addp", "cvtp", "fma", "fma"]),
    ("pv + 5 fma + K read (AGPR)", "pv", ["fma"] * 5 + ["ldsa"]),
    ("pv + 4 fma + K read (AGPR)", "pv", ["fma"] * 4 + ["ldsa"]),
    ("pv + 3 fma + K read (AGPR)", "pv", ["fma"] * 3 + ["ldsa"]),
    ("pv + 2 fma + K read + V read", "pv", ["fma"] * 2 + ["ldsa", "ldsv"]),
    ("pv + 2 max3 + K read", "pv", ["max3"] * 2 + ["ldsa"]),
    ("qk + 4 s_nop", "qk", ["nop"] * 4),
    ("qk + 4 salu", "qk", ["salu"] * 4),
]

src = ['// GENERATED by tools/ubench/gen_a64_gap.py - do not edit.',
       '#include <hip/hip_runtime.h>', '#include <stdio.h>',
       'extern __shared__ char smem[];']
clob = ", ".join(f'"v{i}"' for i in range(64, 241)) + ', "s20", "a0", "a255"'
for vi, (name, kind, recipe) in enumerate(VARIANTS):
    body = []
    for g in range(36 if kind == "qkchain" else 32):
        body.append(mfma(kind, g))
        body += fillers(recipe, g)
    if any(op.startswith("lds") for op in recipe):
        body.append("s_waitcnt lgkmcnt(0)")
    asm = "\\n\\t".join(body)
    src.append(f'''
__global__ __launch_bounds__(256, 1) void k{vi}(unsigned long long* out, int iters) {{
  asm volatile("v_mov_b32 v240, 0\\n\\tv_mov_b32 v232, 1.0\\n\\tv_mov_b32 v233, 0\\n\\tv_mov_b32 v234, 0" ::: {clob});
  smem[threadIdx.x] = 0;
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) asm volatile("{asm}" ::: {clob}, "memory");
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (blockIdx.x == 0 && threadIdx.x == 0) out[0] = t1 - t0;
}}''')
src.append('int main() {\n  unsigned long long* out; hipMalloc(&out, 64);\n  const int iters = 2000;')
for vi, (name, kind, recipe) in enumerate(VARIANTS):
    src.append(f'''  {{ hipFuncSetAttribute((const void*)k{vi}, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k{vi}, dim3(256), dim3(256), 65536, 0, out, 10);
    hipEventRecord(e0); hipLaunchKernelGGL(k{vi}, dim3(256), dim3(256), 65536, 0, out, iters); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); unsigned long long cyc; hipMemcpy(&cyc, out, 8, hipMemcpyDeviceToHost);
    printf("%-44s %6.1f ticks/MFMA  %7.1f ns/MFMA (event)  %2d fillers/gap\\n", "{name}", (double)cyc / (iters * {36.0 if kind == "qkchain" else 32.0}), ms * 1e6 / (iters * {36.0 if kind == "qkchain" else 32.0}), {len(recipe)}); }}''')
src.append('  return 0;\n}')
open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "a64_gap.hip"), "w").write("\n".join(src) + "\n")
print("wrote a64_gap.hip with", len(VARIANTS), "variants")
