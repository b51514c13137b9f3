"""Full-DEPTH single evaluations at the two ENDS of cfg 2's time grid (test infrastructure): the CPU oracle's Flux.forward for
the whole 19 + 38-block model (13.1 B parameters, D = 3072, L = 512 + 3456, LoRA r256) with PROCEDURAL weights and inputs
(tests/procedural.py) at

    t = 1.0              the first evaluation of the 30-point shifted grid (timestep embedding of 1000)
    t = 1 - bf16(t_28)   the last one (~0.09: where the time embedding differs most from the middle of the grid)

in both oracle modes (bf16 / merged LoRA = the HIP path's rounding points; fp32 / un-merged = exact reference semantics)
-> tests/golden/fulldepth_times_oracle.npz.  tests/golden/fulldepth_cfg2_oracle.npz holds the same comparison at t = 0.62.
`--geom cfg3 | cfg5 | p34`: ONE evaluation at t = 0.62 of the same model on the 512-grid 2x3 (L = 6656) / 384-grid 3x4 (L = 7424) /
non-square 2x3 grid of 3:4 portraits (L = 3752: off every tile edge) geometry -> fulldepth_<geom>_oracle.npz (every second image token).  About 45 min on 8 cores (a large geometry: ~40 min); weights are kept as bf16 on the host (26 GB) and widened one tensor at a time.

    python tests/golden/make_fulldepth_times.py
"""
import importlib.util
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)

import oracle.flux_oracle as O  # noqa: E402
from tests.procedural import procedural_param  # noqa: E402


class LazyF32(dict):
    """bf16 storage, f32 on access (the values ARE bf16: exact)"""

    def __getitem__(self, k):
        return dict.__getitem__(self, k).float()

    def get(self, k, default=None):
        return self[k] if k in self else default


def flux_times():
    t = O.time_grid(30, 3456, True, 1)
    tm = 1.0 - t[:-1].to(torch.bfloat16).float()
    return [float(tm[0]), float(tm[-1])]


GEOMS = {      # the other full-depth fixtures: one evaluation at t = 0.62 on the two largest BASELINE geometries
    "cfg3": dict(rows=2, row_latent=(64, 192)),      # 512-grid 2x3, L = 512 + 6144
    "cfg5": dict(rows=3, row_latent=(48, 192)),      # 384-grid 3x4, L = 512 + 6912
    # a shape the pipeline really produces (visualcloze.py:28-60): 2x3 grid of 3:4 portraits, 320x432 px each, L = 512 + 3240
    "p34": dict(rows=2, row_latent=(54, 120)),
}
TOKEN_STRIDE = 2      # (the large geometries are stored for every second image token)


def geom_inputs(geom):
    """procedural inputs of a geometry (as make_fullwidth_traj.inputs, which holds cfg 2 and the SDEdit stage)"""
    from tests.procedural import ptensor
    c = GEOMS[geom]
    h, w = c["row_latent"]
    ids = O.grid_img_ids([(h, w)] * c["rows"])
    N = ids.shape[0]
    seed = 2000 + sorted(GEOMS).index(geom) * 10
    x = ptensor((1, N, 64), seed + 1, q=6)
    cond = torch.cat([ptensor((1, N, 64), seed + 2, q=6), (ptensor((1, N, 256), seed + 3, q=0, kmax=1).abs() > 0.5).float()], -1)
    return dict(x=x, cond=cond, img_ids=ids[None], txt=ptensor((1, 512, 4096), seed + 4, q=6), txt_ids=torch.zeros(1, 512, 3),
                y=ptensor((1, 768), seed + 5, q=6), txt_mask=torch.ones(1, 512, dtype=torch.int32),
                img_mask=torch.ones(1, N, dtype=torch.int32), guidance=torch.full((1,), 30.0))


def trajectory(K, sd, G, inp):
    """K consecutive full-depth evaluations with the state fed back (transport/integrators.py:99-120 through oracle.sample_euler):
    the one combination the other fixtures leave open - trajectories are 1 + 1 blocks, full depth is single evaluations."""
    t = O.time_grid(30, inp["x"].shape[1], True, 1)[:K + 1]
    out = {"x_sum": np.float64(inp["x"].double().sum().item()), "t": t.numpy(), "token_stride": np.int32(1)}
    res = {}
    for tag, P in (("bf16", O.Prec("bf16", "merged")), ("fp32", O.Prec("fp32", "ref"))):
        def model_fn(xin, tm, P=P):
            t1 = time.time()
            y = O.flux_forward(sd, G, xin, inp["img_ids"], inp["txt"], inp["txt_ids"], tm, inp["y"], inp["txt_mask"], inp["img_mask"],
                               inp["guidance"], P=P)
            print(f"  {tag} evaluation at t = {float(tm[0]):.6f}: {time.time() - t1:.0f} s", flush=True)
            return y
        with torch.no_grad():
            states, evals = O.sample_euler(model_fn, P.r(inp["x"]), P.r(inp["cond"]), t, P)
        res[tag] = states
        out[f"{tag}_model_t"] = np.asarray(evals, np.float64)
    for k in range(1, K + 1):
        b, f = res["bf16"][k], res["fp32"][k]
        assert torch.equal(b.to(torch.bfloat16).float(), b)
        out[f"bf16_{k}"] = b.to(torch.bfloat16).view(torch.int16).numpy()
        out[f"fp32_{k}"] = f.numpy().astype(np.float16)
        print(f"state {k}: oracle bf16-vs-fp32 rel-L2 {((b - f).norm() / f.norm()).item():.3e}", flush=True)
    np.savez_compressed(os.path.join(HERE, "fulldepth_traj_oracle.npz"), **out)
    print("wrote fulldepth_traj_oracle.npz", flush=True)


def main():
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("--geom", default=None, choices=sorted(GEOMS), help="instead of cfg 2's two grid ends: this geometry at t = 0.62")
    ap.add_argument("--traj", type=int, default=0, help="instead: the first K solver steps of cfg 2's 30-point grid through the full-depth "
                    "model with the state FED BACK (fixed-grid Euler, each mode steps its own state) -> fulldepth_traj_oracle.npz")
    a = ap.parse_args()
    spec = importlib.util.spec_from_file_location("ft", os.path.join(HERE, "make_fullwidth_traj.py"))
    FT = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(FT)
    from visualcloze_amd.model import FLUX_DEV_FILL, FluxLoraWrapper, FluxParams
    with torch.device("meta"):
        m = FluxLoraWrapper(lora_rank=256, lora_scale=1.0, params=FluxParams(**FLUX_DEV_FILL))
    t0 = time.time()
    sd = LazyF32({k: procedural_param(k, tuple(v.shape), device="cpu").to(torch.bfloat16) for k, v in m.state_dict().items()})
    print(f"procedural weights: {sum(dict.__getitem__(sd, k).numel() for k in sd) / 1e9:.2f} B parameters in {time.time() - t0:.0f} s", flush=True)
    inp = geom_inputs(a.geom) if a.geom else FT.inputs("cfg2")
    G = O.FluxGeometry()
    out = {"x_sum": np.float64(inp["x"].double().sum().item())}
    times = [0.62] if a.geom else flux_times()
    stride = TOKEN_STRIDE if a.geom else 1
    out["token_stride"] = np.int32(stride)
    name = f"fulldepth_{a.geom}_oracle.npz" if a.geom else "fulldepth_times_oracle.npz"
    out["times"] = np.asarray(times, np.float64)
    args = lambda t: (sd, G, torch.cat((inp["x"], inp["cond"]), -1), inp["img_ids"], inp["txt"], inp["txt_ids"],   # noqa: E731
                      torch.tensor([t]), inp["y"], inp["txt_mask"], inp["img_mask"], inp["guidance"])
    # both modes see the SAME guidance value (an f32 guidance tensor: 1000 * g = 30000 exactly; a bf16 one would round to
    # 29952 in the bf16 mode only, and the bf16-vs-fp32 floor is meant to hold arithmetic noise, not an input difference)
    orig = O.compute_vec
    O.compute_vec = lambda *a, **k: orig(*a, **{**k, "guidance_is_bf16": False})
    if a.traj:
        return trajectory(a.traj, sd, G, inp)
    for i, t in enumerate(times):
        for tag, P in (("bf16", O.Prec("bf16", "merged")), ("fp32", O.Prec("fp32", "ref"))):
            t1 = time.time()
            with torch.no_grad():
                y = O.flux_forward(*args(t), P=P)
            print(f"t = {t:.6f} {tag}: {time.time() - t1:.0f} s ({torch.get_num_threads()} threads)", flush=True)
            if tag == "bf16":
                assert torch.equal(y.to(torch.bfloat16).float(), y)
                out[f"bf16_{i}"] = y[:, ::stride].to(torch.bfloat16).view(torch.int16).numpy()
                yb = y
            else:
                out[f"fp32_{i}"] = y[:, ::stride].numpy().astype(np.float16)
                print(f"  oracle bf16-vs-fp32 rel-L2 {((yb - y).norm() / y.norm()).item():.3e}", flush=True)
        np.savez_compressed(os.path.join(HERE, name), **out)
    print("wrote", name, flush=True)


if __name__ == "__main__":
    main()
