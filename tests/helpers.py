"""Test helpers: the tiny procedural-weight model on the GPU and the smoke check of the HIP path against the CPU
oracle (used by tests/ and __graft_entry__.smoke(); lives outside the product package because it imports oracle/)."""
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)


def tiny_model(dev="cuda:0", dtype=torch.bfloat16):
    if REPO not in sys.path:
        sys.path.insert(0, REPO)
    from tests.procedural import TINY, TINY_RANK, procedural_param
    from visualcloze_amd.model import FluxLoraWrapper, FluxParams
    m = FluxLoraWrapper(lora_rank=TINY_RANK, lora_scale=1.0, params=FluxParams(**TINY))
    sd = {k: procedural_param(k, v.shape) for k, v in m.state_dict().items()}
    m.load_state_dict(sd, strict=True)
    return m.eval().to(dev, dtype), sd


def rel_l2(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return ((a - b).norm() / b.norm()).item()


def smoke() -> None:
    from visualcloze_amd import hip
    hip.require_gpu()
    if REPO not in sys.path:
        sys.path.insert(0, REPO)
    import oracle.flux_oracle as O  # test infrastructure: the checker, not the thing run
    from tests.procedural import TINY, tiny_inputs
    model, sd = tiny_model()
    inp = tiny_inputs(B=1)
    dev = "cuda:0"
    img = torch.cat((inp["x"], inp["cond"]), -1)
    t = torch.tensor([0.7])
    got = model(img.to(dev, torch.bfloat16), img_ids=inp["img_ids"].to(dev), txt=inp["txt"].to(dev, torch.bfloat16),
                txt_ids=inp["txt_ids"].to(dev), timesteps=t.to(dev), y=inp["y"].to(dev, torch.bfloat16),
                txt_mask=inp["txt_mask"].to(dev), img_mask=inp["img_mask"].to(dev), guidance=inp["guidance"].to(dev))
    torch.cuda.synchronize()
    G = O.FluxGeometry(**TINY)
    orig = O.compute_vec
    O.compute_vec = lambda *a, **k: orig(*a, **{**k, "guidance_is_bf16": False})
    try:
        want = O.flux_forward(sd, G, img, inp["img_ids"], inp["txt"], inp["txt_ids"], t, inp["y"], inp["txt_mask"],
                              inp["img_mask"], inp["guidance"], P=O.Prec("bf16", "merged"))
    finally:
        O.compute_vec = orig
    err = rel_l2(got, want)
    print(f"smoke: tiny Flux.forward on {torch.cuda.get_device_name(0)}: rel-L2 vs bf16 oracle = {err:.3e}")
    assert torch.isfinite(got.float()).all() and err < 2e-2, f"HIP path deviates from the oracle: {err}"


def parity_log(line: str) -> None:
    """Print a measured deviation and, when VC_PARITY_LOG names a file, append it there (the numbers DESIGN.md quotes
    come from these lines; pytest swallows stdout of passing tests)."""
    print("\n" + line)
    path = os.environ.get("VC_PARITY_LOG")
    if path:
        with open(path, "a") as f:
            f.write(line + "\n")
