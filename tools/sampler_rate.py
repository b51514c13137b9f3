"""Denoising-steps/s THROUGH THE PRODUCT SAMPLER (`Sampler.sample_ode(...)(x, model.forward, kwargs)`, the reference's call), next to
bench.py's figure for the same workload (bench.py drives the C handle directly, one step per call).  Advisor r04: the host-copy
cache of handle.py hit only in bench.py; with the cache keyed on memory the sampler route must not drain the stream per sample
either - visible at cfg 1, whose samples are 3 evaluations long.
    python tools/sampler_rate.py [--workload 384-grid-1x2] [--samples 40]"""
import argparse
import json
import os
import sys
import time

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench  # noqa: E402
from visualcloze_amd import hip  # noqa: E402
from visualcloze_amd.transport import Sampler, create_transport  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="384-grid-1x2")
    ap.add_argument("--samples", type=int, default=40)
    a = ap.parse_args()
    hip.require_gpu()
    dev = torch.device("cuda", 0)
    wl = bench.WORKLOADS[a.workload]
    model, _ = bench.build_model(dev, 0, 1)
    model.prepare(free_parameters=True)
    x, kw = bench.make_inputs(dev, wl, seed=0)
    fn = Sampler(create_transport()).sample_ode(sampling_method="euler", num_steps=wl["steps"], do_shift=wl.get("do_shift", True),
                                                time_shifting_factor=1)
    evals = wl["steps"] - 1
    for _ in range(3):
        fn(x, model.forward, kw)
    torch.cuda.synchronize()
    hc = model.handle()._host
    h0, m0 = hc.hits, hc.misses
    t0 = time.perf_counter()
    for _ in range(a.samples):
        out = fn(x, model.forward, kw)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    assert torch.isfinite(out.float()).all()
    print(json.dumps(dict(workload=a.workload, route="Sampler.sample_ode -> model.forward (fused: vc_flux_prepare + vc_flux_sample_euler per sample)",
                          samples=a.samples, evaluations_per_sample=evals, steps_per_s=round(a.samples * evals / dt, 3),
                          ms_per_sample=round(dt / a.samples * 1e3, 3), host_copy_cache=dict(hits=hc.hits - h0, misses=hc.misses - m0))))


if __name__ == "__main__":
    main()
