"""Where the GEMM's power goes: the product GEMM launches of cfg 2 looped for a few seconds per LIBRARY BUILD while
`rocm-smi --showpower --showclocks` is sampled, for the build of record and the elimination builds of the loader-wave loop

    make variant NAME=nomfma    DEFS=-DVC_GEMM_NO_MFMA       LDS-DMA + fragment reads, no MFMA
    make variant NAME=nodma     DEFS=-DVC_GEMM_NO_DMA        fragment reads + MFMA on the operands of K-tile 0, no LDS-DMA
    make variant NAME=noldsread DEFS=-DVC_GEMM_NO_LDSREAD    LDS-DMA + MFMA on the fragments of K-slice 0, no fragment reads

on random and on zero operands.  Energy per launch = average socket power x time per launch; the differences between the
builds price the three streams of the loop (matrix pipe, L2 -> LDS, LDS -> registers) in joules, which is the unit the
power-limited board trades for time.

    python tools/gemm_power.py [main nomfma nodma noldsread] [--seconds 4]

--attention does the same for the attention launch of cfg 2 (variant 12, bounded logits) and the elimination builds of
attention64.hip (VC_A64_NO_MFMA / NO_SOFTMAX / NO_LDS / NO_DMA -> `a64nomfma a64nosoftmax a64nolds a64nodma`)."""
import argparse
import ctypes as C
import os
import re
import subprocess
import sys
import threading
import time

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from visualcloze_amd import hip  # noqa: E402


def load(name):
    path = hip.LIB_PATH if name == "main" else os.path.join(os.path.dirname(hip.LIB_PATH), f"libvcloze_hip_{name}.so")
    l = C.CDLL(path)
    for sym, (res, args) in hip.SYMBOLS.items():
        fn = getattr(l, sym)
        fn.restype, fn.argtypes = res, args
    return l


def smi():
    try:
        out = subprocess.run(["rocm-smi", "--showpower", "--showclocks"], capture_output=True, text=True, timeout=20).stdout
    except Exception:
        return None
    w = re.search(r"Power \(W\): ([0-9.]+)", out)
    c = re.search(r"sclk clock level: \d+: \((\d+)Mhz\)", out)
    return (float(w.group(1)) if w else float("nan"), float(c.group(1)) if c else float("nan"))


def loop(fn, seconds):
    stop, samples = [False], []

    def sampler():
        time.sleep(1.2)
        while not stop[0]:
            s = smi()
            if s:
                samples.append(s)
            time.sleep(0.3)
    th = threading.Thread(target=sampler)
    th.start()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0, n = time.time(), 0
    e0.record()
    while time.time() - t0 < seconds:
        for _ in range(50):
            fn()
        n += 50
        torch.cuda.synchronize()
    e1.record()
    torch.cuda.synchronize()
    stop[0] = True
    th.join()
    us = e0.elapsed_time(e1) * 1e3 / n
    w = sum(s[0] for s in samples) / max(len(samples), 1)
    mhz = sum(s[1] for s in samples) / max(len(samples), 1)
    return us, w, mhz, len(samples)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("builds", nargs="*", default=None)
    ap.add_argument("--seconds", type=float, default=4.0)
    ap.add_argument("--attention", action="store_true")
    a = ap.parse_args()
    if a.builds is None or not a.builds:
        a.builds = ["main", "a64nomfma", "a64nosoftmax", "a64nolds", "a64nodma"] if a.attention else ["main", "nomfma", "nodma", "noldsread"]
    hip.require_gpu()
    dev = "cuda:0"
    L, D = 3968, 3072
    g = torch.Generator(device=dev).manual_seed(0)
    rnd = lambda *s, sc=1.0: (torch.randn(*s, device=dev, generator=g) * sc).to(torch.bfloat16)
    if a.attention:
        time.sleep(2.0)
        idle = smi()
        print(f"idle: {idle[0]:.0f} W, {idle[1]:.0f} MHz", flush=True)
        H = 24
        for data in ("random", "zero"):
            # q, k as the QKNorm leaves them (unit-RMS rows): |logit| <= sqrt(128) * log2(e) ... the bounded-softmax form applies
            qkv = rnd(L, 3 * D) if data == "random" else torch.zeros(L, 3 * D, dtype=torch.bfloat16, device=dev)
            vt = rnd(H, 128, L) if data == "random" else torch.zeros(H, 128, L, dtype=torch.bfloat16, device=dev)
            o = torch.empty(L, D, dtype=torch.bfloat16, device=dev)
            rows = {}
            for name in a.builds:
                hip._lib = load(name)
                us, w, mhz, ns = loop(lambda: hip.attention(qkv, vt, o, L, H, variant=12, logit_bound=60.0), a.seconds)
                rows[name] = (us, w, mhz)
                print(f"attention L=3968 H=24 {data:6s} {name:13s}: {us:7.1f} us/launch  {w:6.0f} W  {mhz:5.0f} MHz  "
                      f"{us * w * 1e-6:.4f} J/launch  ({us * (w - idle[0]) * 1e-6:.4f} J above idle; {ns} samples)", flush=True)
                time.sleep(1.0)
            if all(k in rows for k in ("main", "a64nomfma", "a64nosoftmax", "a64nolds", "a64nodma")):
                e = {k: v[0] * (v[1] - idle[0]) * 1e-6 for k, v in rows.items()}
                print(f"  -> above-idle energy per launch: all {e['main']:.4f} J; matrix pipe ~ {e['main'] - e['a64nomfma']:.4f}; softmax VALU ~ "
                      f"{e['main'] - e['a64nosoftmax']:.4f}; fragment reads ~ {e['main'] - e['a64nolds']:.4f}; LDS-DMA ~ {e['main'] - e['a64nodma']:.4f}", flush=True)
        return
    shapes = {   # name: (K, N, epilogue)
        "GATE_RES K=12288 N=3072": (4 * D, D, hip.EPI_GATE_RES),
        "BIAS     K=3072  N=9216": (D, 3 * D, hip.EPI_BIAS),
    }
    time.sleep(2.0)
    idle = smi()
    print(f"idle: {idle[0]:.0f} W, {idle[1]:.0f} MHz", flush=True)
    for sname, (K, N, epi) in shapes.items():
        for data in ("random", "zero"):
            A = rnd(L, K) if data == "random" else torch.zeros(L, K, dtype=torch.bfloat16, device=dev)
            W = rnd(N, K, sc=K ** -0.5) if data == "random" else torch.zeros(N, K, dtype=torch.bfloat16, device=dev)
            b = torch.zeros(N, dtype=torch.bfloat16, device=dev)
            x = rnd(L, N) if data == "random" else torch.zeros(L, N, dtype=torch.bfloat16, device=dev)
            gate = torch.full((N,), 0.0, dtype=torch.bfloat16, device=dev)     # x stays x: the loop does not drift
            prob = hip.make_problem(A, W, b, x, res=x, gate=gate) if epi == hip.EPI_GATE_RES else hip.make_problem(A, W, b, x)
            rows = {}
            for name in a.builds:
                hip._lib = load(name)
                us, w, mhz, ns = loop(lambda: hip.gemm(prob, epi=epi), a.seconds)
                rows[name] = (us, w, mhz)
                print(f"{sname} {data:6s} {name:10s}: {us:7.1f} us/launch  {w:6.0f} W  {mhz:5.0f} MHz  "
                      f"{us * w * 1e-6:.4f} J/launch  ({us * (w - idle[0]) * 1e-6:.4f} J above idle; {ns} samples)", flush=True)
                time.sleep(1.0)
            if all(k in rows for k in ("main", "nomfma", "nodma", "noldsread")):
                e = {k: v[0] * (v[1] - idle[0]) * 1e-6 for k, v in rows.items()}
                print(f"  -> above-idle energy per launch: all {e['main']:.4f} J; without MFMA {e['nomfma']:.4f} (matrix pipe ~ {e['main'] - e['nomfma']:.4f}); "
                      f"without LDS-DMA {e['nodma']:.4f} (L2->LDS ~ {e['main'] - e['nodma']:.4f}); without fragment reads {e['noldsread']:.4f} "
                      f"(LDS->registers ~ {e['main'] - e['noldsread']:.4f})", flush=True)


if __name__ == "__main__":
    main()
