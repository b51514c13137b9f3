"""GPU-box diagnostic: per-stage deviation of the HIP path from the bf16 oracle on the tiny geometry."""
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
import oracle.flux_oracle as O  # noqa: E402
from tests.procedural import TINY, tiny_inputs  # noqa: E402
from tests.helpers import rel_l2, tiny_model  # noqa: E402


def main():
    model, sd = tiny_model()
    inp = tiny_inputs(B=1)
    dev = "cuda:0"
    eng = model.engine()
    T, N = inp["txt"].shape[1], inp["x"].shape[1]
    ws = eng.workspace(T, N, 1)
    bf = lambda t: t.to(dev, torch.bfloat16).contiguous()  # noqa: E731
    t = torch.tensor([0.7])
    eng.prepare_sample(ws, bf(inp["txt"]), bf(inp["y"]), inp["guidance"], False, inp["img_ids"], inp["txt_ids"], t, [T + N])
    ws.XIN.copy_(bf(torch.cat((inp["x"], inp["cond"]), -1)[0]))
    taps = {}
    eng.eval_once(ws, None, euler=False, concat=False, taps=taps)
    torch.cuda.synchronize()
    otaps = {}
    orig = O.compute_vec
    O.compute_vec = lambda *a, **k: orig(*a, **{**k, "guidance_is_bf16": False})
    want = O.flux_forward(sd, O.FluxGeometry(**TINY), torch.cat((inp["x"], inp["cond"]), -1), inp["img_ids"], inp["txt"],
                          inp["txt_ids"], t, inp["y"], inp["txt_mask"], inp["img_mask"], inp["guidance"],
                          P=O.Prec("bf16", "merged"), taps=otaps)
    O.compute_vec = orig
    print("vec     ", rel_l2(ws.VEC[0], otaps["vec"][0]))
    for k in taps:
        ref = otaps[k][0]
        print(f"{k:16s} rel-L2 {rel_l2(taps[k], ref):.3e}")
    print("final   ", rel_l2(ws.V, want[0]))


if __name__ == "__main__":
    main()
