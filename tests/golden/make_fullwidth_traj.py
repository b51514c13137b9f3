"""Full-WIDTH trajectory fixtures (test infrastructure): the CPU oracle's fixed-grid Euler trajectories at BASELINE sizes,
D = 3072 / H = 24 / T = 512 with 1 DoubleStreamBlock + 1 SingleStreamBlock and LoRA r256, PROCEDURAL weights and inputs
(tests/procedural.py: closed form, no RNG, bit-identical on every machine):

    cfg2    384-grid 2x3   N = 3456  L = 3968   30 solver points = 29 evaluations, shifted grid      (transport.py:361-410)
    sdedit  1024^2 target  N = 4096  L = 4608   10 points from strength 0.4, no shift = 9 evaluations (visualcloze.py:184-234)
    cfg5    384-grid 3x4   N = 6912  L = 7424   30 points, shifted grid; only with --only cfg5 -> fullwidth_traj_cfg5.npz
    cfg5_50 the same geometry and inputs, BASELINE's own 50 points = 49 evaluations; --only cfg5_50 -> fullwidth_traj_cfg5_50.npz
    p34     2x3 grid of 3:4 portraits (non-square: N = 3240, L = 3752), 30 points; --only p34 -> fullwidth_traj_p34.npz

For each: the bf16 / merged-LoRA oracle (same rounding points as the HIP path; bf16 state as visualcloze.py:399) and the
fp32 / un-merged oracle (exact reference semantics) -> tests/golden/fullwidth_traj.npz: final latents, a few intermediate
states, the oracle's own bf16-vs-fp32 deviation per saved state.  The oracle is pinned to the reference by
tests/test_oracle_golden.py; this script only RUNS it (about 22 min on 8 cores).

    python tests/golden/make_fullwidth_traj.py [--only cfg2|sdedit] [--evals K]
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)

import oracle.flux_oracle as O  # noqa: E402
from tests.procedural import procedural_param, ptensor  # noqa: E402

T = 512
TOKEN_STRIDE = 8
CASES = {      # seed: base of the procedural input draws (fixed per case, so that adding a case moves no other)
    "cfg2": dict(rows=2, row_latent=(48, 144), points=30, do_shift=True, strength=None, keep=(1, 10, 20, 29), seed=1000),
    "sdedit": dict(rows=1, row_latent=(128, 128), points=10, do_shift=False, strength=0.4, keep=(1, 5, 9), seed=1010),
    # the largest BASELINE geometry (384-grid 3x4, N = 6912, L = 7424): its own file, final state every 2nd token
    "cfg5": dict(rows=3, row_latent=(48, 192), points=30, do_shift=True, strength=None, keep=(1, 15, 29), seed=1020,
                 file="fullwidth_traj_cfg5.npz", final_stride=2),
    # BASELINE cfg 5 as it is quoted: 50 solver points = 49 evaluations (mu = 1.62667) on the same geometry and inputs
    "cfg5_50": dict(rows=3, row_latent=(48, 192), points=50, do_shift=True, strength=None, keep=(1, 25, 49), seed=1020,
                    file="fullwidth_traj_cfg5_50.npz", final_stride=2),
    # a shape the pipeline really produces (visualcloze.py:28-60): 2x3 grid of 3:4 portraits, N = 3240, L = 3752 - off every tile edge
    "p34": dict(rows=2, row_latent=(54, 120), points=30, do_shift=True, strength=None, keep=(1, 15, 29), seed=1040,
                file="fullwidth_traj_p34.npz"),
}


def key_shapes():
    """state-dict keys / shapes of FluxLoraWrapper(depth 1 + 1, r256) at FLUX width, from the module tree itself"""
    from visualcloze_amd.model import FLUX_DEV_FILL, FluxLoraWrapper, FluxParams
    with torch.device("meta"):
        m = FluxLoraWrapper(lora_rank=256, lora_scale=1.0,
                            params=FluxParams(**{**FLUX_DEV_FILL, "depth": 1, "depth_single_blocks": 1}))
    return [(k, tuple(v.shape)) for k, v in m.state_dict().items()]


def inputs(case):
    c = CASES[case]
    h, w = c["row_latent"]
    ids = O.grid_img_ids([(h, w)] * c["rows"])
    N = ids.shape[0]
    seed = c["seed"]
    x = ptensor((1, N, 64), seed + 1, q=6)
    cond = torch.cat([ptensor((1, N, 64), seed + 2, q=6), (ptensor((1, N, 256), seed + 3, q=0, kmax=1).abs() > 0.5).float()], -1)
    return dict(x=x, cond=cond, img_ids=ids[None], txt=ptensor((1, T, 4096), seed + 4, q=6), txt_ids=torch.zeros(1, T, 3),
                y=ptensor((1, 768), seed + 5, q=6), txt_mask=torch.ones(1, T, dtype=torch.int32),
                img_mask=torch.ones(1, N, dtype=torch.int32), guidance=torch.full((1,), 30.0))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default=None)
    ap.add_argument("--evals", type=int, default=None, help="stop after K evaluations (timing probe; nothing is written)")
    a = ap.parse_args()
    t0 = time.time()
    # the deployed model holds bf16 parameters (models/util.py:402 `.to(torch.bfloat16)`): every procedural value is
    # bf16-exact except the RMSNorm scales (1 + k/512), which are rounded here as loading them into the model rounds them
    sd = {k: procedural_param(k, s, device="cpu").to(torch.bfloat16).float() for k, s in key_shapes()}
    print(f"procedural weights: {sum(v.numel() for v in sd.values()) / 1e6:.0f} M parameters in {time.time() - t0:.0f} s", flush=True)
    G = O.FluxGeometry(depth=1, depth_single_blocks=1)
    default_cases = [k for k, c in CASES.items() if "file" not in c]        # fullwidth_traj.npz; the others on request
    # both modes see the SAME guidance value (an f32 guidance tensor: 1000 * g = 30000 exactly; a bf16 one would round to
    # 29952 in the bf16 mode only, and the bf16-vs-fp32 floor is meant to hold arithmetic noise, not an input difference)
    orig = O.compute_vec
    O.compute_vec = lambda *a, **k: orig(*a, **{**k, "guidance_is_bf16": False})
    for case in ([a.only] if a.only else default_cases):
        c, inp = CASES[case], inputs(case)
        path = os.path.join(HERE, c.get("file", "fullwidth_traj.npz"))
        out = dict(np.load(path)) if os.path.exists(path) else {}
        N = inp["x"].shape[1]
        t = O.time_grid(c["points"], N, c["do_shift"], 1 if c["do_shift"] else 1.0, c["strength"])
        if a.evals:
            t = t[:a.evals + 1]
        res = {}
        for tag, P in (("bf16", O.Prec("bf16", "merged")), ("fp32", O.Prec("fp32", "ref"))):
            def model_fn(xin, tm, P=P):
                return O.flux_forward(sd, G, xin, inp["img_ids"], inp["txt"], inp["txt_ids"], tm, inp["y"], inp["txt_mask"],
                                      inp["img_mask"], inp["guidance"], P=P)
            t1 = time.time()
            with torch.no_grad():
                states, evals = O.sample_euler(model_fn, P.r(inp["x"]), P.r(inp["cond"]), t, P)
            res[tag] = states
            print(f"{case} {tag}: {len(evals)} evaluations in {time.time() - t1:.0f} s ({torch.get_num_threads()} threads)", flush=True)
        if a.evals:
            continue
        keep = [k for k in c["keep"] if k < len(res["bf16"])]
        out[f"{case}_keep"] = np.asarray(keep, np.int32)
        out[f"{case}_t"] = t.numpy()
        out[f"{case}_x_sum"] = np.float64(inp["x"].double().sum().item())
        for k in keep:
            b, f = res["bf16"][k], res["fp32"][k]
            assert torch.equal(b.to(torch.bfloat16).float(), b)                    # bf16-mode states are bf16 values
            # kept small: the FINAL state whole, intermediate states every TOKEN_STRIDE-th token; the fp32 oracle's states
            # stored as float16 (2e-4 rel-L2 against floors >= 2.6e-3)
            sl = slice(None, None, c.get("final_stride", 1)) if k == keep[-1] else slice(None, None, TOKEN_STRIDE)
            out[f"{case}_bf16_{k}"] = b[:, sl].to(torch.bfloat16).view(torch.int16).numpy()
            out[f"{case}_fp32_{k}"] = f[:, sl].numpy().astype(np.float16)
            print(f"  state {k}: oracle bf16-vs-fp32 rel-L2 {((b - f).norm() / f.norm()).item():.3e}", flush=True)
        out["token_stride"] = np.int32(TOKEN_STRIDE)
        out[f"{case}_final_stride"] = np.int32(c.get("final_stride", 1))
        np.savez_compressed(path, **out)
        print("wrote", path, os.path.getsize(path), "bytes", flush=True)


if __name__ == "__main__":
    main()
