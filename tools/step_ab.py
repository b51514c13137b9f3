"""A/B of whole solver steps across LIBRARY BUILDS, interleaved in one process: the 13 B-parameter model is built and its
weights prepared ONCE, every build (lib/libvcloze_hip_<name>.so from `make variant NAME=<name> DEFS=...`; `main` = the build
of record) gets its own C handle over the same weight tensors, and rounds of `--steps` graph replays alternate between the
builds.  Reports the median and the minimum ms / step per build and, with --attn, the in-situ attention time of each.

    python tools/step_ab.py main xcdtail [--workload 384-grid-2x3] [--rounds 5] [--steps 29] [--attn] [--opt name:key=val]

--opt sets engine options per build, e.g. `--opt b:attn_variant=8` (applied to the model's engine before the handle of
build `b` is created)."""
import argparse
import ctypes as C
import json
import os
import statistics
import sys
import time

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench  # noqa: E402
from visualcloze_amd import hip  # noqa: E402


def load(name):
    path = hip.LIB_PATH if name == "main" else os.path.join(os.path.dirname(hip.LIB_PATH), f"libvcloze_hip_{name}.so")
    l = C.CDLL(path)
    for sym, (res, args) in hip.SYMBOLS.items():
        if sym == "vc_flux_profile" and not hasattr(l, sym):      # an older library in the A/B
            continue
        fn = getattr(l, sym)
        fn.restype, fn.argtypes = res, args
    # ABI 9 added a bit of VcAttention.variant (28 = 12 + 16, ignored by an ABI-8 library): same struct layouts, so a round-5
    # library can still stand in an A/B
    # (ABI 10 added vc_flux_profile, which these A/Bs do not call: libraries back to ABI 8 can still stand in)
    assert l.vc_abi_version() in (hip.ABI_VERSION, 9, 8), (name, l.vc_abi_version())
    return l


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("builds", nargs="+", help="build[=alias]: `main` or the NAME of a `make variant` library; an alias lets one "
                                              "library appear twice with different --opt settings")
    ap.add_argument("--workload", default="384-grid-2x3")
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--steps", type=int, default=29)
    ap.add_argument("--attn", action="store_true", help="also time the attention launches in situ (Python-ordered plan)")
    ap.add_argument("--opt", action="append", default=[])
    ap.add_argument("--per-gpu-batch", type=int, default=1)
    a = ap.parse_args()
    hip.require_gpu()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    wl = bench.WORKLOADS[a.workload]
    model, _ = bench.build_model(dev, 0, 1)
    eng = model.prepare(free_parameters=True)
    x, kw = bench.make_inputs(dev, wl, seed=0, B=a.per_gpu_batch)
    builds = [(b.split("=")[0], b.split("=")[-1]) for b in a.builds]
    opts = {}
    for o in a.opt:
        who, kv = o.split(":")
        k, v = kv.split("=")
        opts.setdefault(who, {})[k] = int(v)
    base_opts = dict(attn_variant=eng.attn_variant, tile_cfg=eng.tile_cfg, fuse_qnorm=eng.fuse_qnorm, fuse_vt=eng.fuse_vt,
                     fuse_knorm=eng.fuse_knorm, bounded_softmax=eng.bounded_softmax, mlp_first=eng.mlp_first,
                     splitk=eng.splitk)
    jobs, libs = {}, {}
    for libname, alias in builds:
        libs[alias] = load(libname)
        hip._lib = libs[alias]
        for k, v in {**base_opts, **opts.get(alias, {})}.items():
            setattr(eng, k, v)
        model._handle = None
        job = bench.Job(model, x, kw, wl["steps"], t0=wl.get("t0", 0.0), do_shift=wl.get("do_shift", True))
        with torch.cuda.stream(eng.stream):
            for _ in range(3):
                job.step()
        torch.cuda.synchronize()
        jobs[alias] = (job, model._handle, {**base_opts, **opts.get(alias, {})})
    times = {alias: [] for _, alias in builds}
    finals = {}
    for r in range(a.rounds):
        for _, alias in builds:
            job, h, o = jobs[alias]
            hip._lib = libs[alias]
            model._handle = h
            for k, v in o.items():
                setattr(eng, k, v)
            with torch.cuda.stream(eng.stream):
                job.restart_sample()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(a.steps):
                    job.step()
                torch.cuda.synchronize()
                times[alias].append((time.perf_counter() - t0) / a.steps * 1e3)
                if r == 0:
                    finals[alias] = job.state().float().cpu()
    ref = finals[builds[0][1]]
    out = {}
    for _, alias in builds:
        ts = times[alias]
        d = ((finals[alias] - ref).norm() / ref.norm()).item()
        out[alias] = dict(median_ms=round(statistics.median(ts), 3), min_ms=round(min(ts), 3), steps_per_s=round(1e3 / statistics.median(ts), 3),
                          rel_l2_vs_first=float(f"{d:.3e}"), finite=bool(torch.isfinite(finals[alias]).all()))
    if a.attn:
        for _, alias in builds:
            job, h, o = jobs[alias]
            hip._lib = libs[alias]
            for k, v in o.items():
                setattr(eng, k, v)
            out[alias]["attention"] = {k: v for k, v in bench.roofline_attention(job, iters=2, via="python").items()
                                       if k in ("avg_launch_us", "median_launch_us", "isolated_us", "frac", "variant")}
    base = out[builds[0][1]]["median_ms"]
    for _, alias in builds:
        out[alias]["vs_first_pct"] = round((base / out[alias]["median_ms"] - 1) * 100, 2)
        print(alias, json.dumps(out[alias]), flush=True)


if __name__ == "__main__":
    main()
