"""Full-WIDTH vectors from the REFERENCE ITSELF (test infrastructure; runs only in the build container, where /root/reference is):
the reference's own `FluxLoraWrapper` (models/model.py:154-175) with ONE DoubleStreamBlock + ONE SingleStreamBlock at FLUX
width - hidden 3072, 24 heads, mlp_ratio 4, LoRA r256 - on cfg 2's geometry (T = 512, N = 3456, L = 3968), fp32 on the CPU,
under the three shims of make_golden.py (flash-attn -> dense SDPA per segment, torchdiffeq, torch.cuda.device).

Why: every other full-size fixture is an output of the ORACLE, and the oracle is pinned to the reference at the tiny geometry
(hidden 256, 2 heads).  This file holds the oracle to the reference once at D = 3072 / 24 heads / L = 3968
(tests/test_oracle_golden.py), and the HIP blocks are compared with it directly (tests/test_fullsize_gpu.py).

Weights and inputs are procedural (tests/procedural.py: closed form, identical on every machine) - the fixture stores OUTPUTS
only: `Flux.forward` [1, 3456, 64] whole, and strided samples of the DoubleStreamBlock's (img, txt) and the
SingleStreamBlock's outputs (layers.py:158-245), taken with forward hooks.

Round 6 adds the reference's own BF16 run of the same model and inputs - bf16 parameters (models/util.py:402), bf16 inputs
and guidance (visualcloze.py:399,413) under `torch.autocast("cpu", torch.bfloat16)` (visualcloze.py:363; SURVEY.md §8c names
this mode) - as `flux_bf16` and `double_img_bf16 / double_txt_bf16 / single_bf16` (bf16 bit patterns, uint16): the oracle's
bf16 mode (`Prec("bf16", "ref")`: WHERE it rounds) is held to it at D = 3072 / 24 heads / r = 256, not only at the tiny geometry.

    python tests/golden/make_fullwidth_reference.py        # -> tests/golden/fullwidth_reference.npz (a few minutes on 8 cores)
"""
import importlib.util
import os
import sys
import time

import numpy as np
import torch

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get("VC_REFERENCE", "/root/reference")
sys.path.insert(0, REPO)

T_MODEL = 0.62          # the Flux time of the evaluation (as the full-depth fixtures)
ROW_STRIDE, COL_STRIDE = 16, 16


def _load(name):
    spec = importlib.util.spec_from_file_location(name, os.path.join(HERE, name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def sample(t):
    """the stored subsample of a [1, rows, D] block output"""
    return t[0, ::ROW_STRIDE, ::COL_STRIDE].contiguous()


def main():
    MG, FT = _load("make_golden"), _load("make_fullwidth_traj")
    MG.install_shims()
    sys.path.insert(0, REF)
    from models.model import FluxLoraWrapper, FluxParams  # noqa: E402  (the reference's own classes)
    from visualcloze_amd.model import FLUX_DEV_FILL
    from tests.procedural import procedural_param

    t0 = time.time()
    params = FluxParams(**{**FLUX_DEV_FILL, "depth": 1, "depth_single_blocks": 1})
    model = FluxLoraWrapper(lora_rank=256, lora_scale=1.0, params=params).float().eval()
    key_shapes = [(k, tuple(v.shape)) for k, v in model.state_dict().items()]
    assert key_shapes == FT.key_shapes(), "the reference's state dict and this repo's differ"
    # the deployed model holds bf16 parameters (models/util.py:402): every procedural value is bf16-exact except the RMSNorm
    # scales, rounded here as loading them into the bf16 model rounds them - the same state dict the oracle fixtures use
    model.load_state_dict({k: procedural_param(k, s, device="cpu").to(torch.bfloat16).float() for k, s in key_shapes}, strict=True)
    inp = FT.inputs("cfg2")
    taps = {}
    model.double_blocks[0].register_forward_hook(lambda m, a, out: taps.update(double_img=out[0], double_txt=out[1]))
    model.single_blocks[0].register_forward_hook(lambda m, a, out: taps.update(single=out))
    with torch.no_grad():
        y = model(torch.cat((inp["x"], inp["cond"]), -1), timesteps=torch.tensor([T_MODEL]), txt=inp["txt"], txt_ids=inp["txt_ids"],
                  txt_mask=inp["txt_mask"], y=inp["y"], img_ids=inp["img_ids"], img_mask=inp["img_mask"], guidance=inp["guidance"])
    assert y.shape == (1, inp["x"].shape[1], 64) and torch.isfinite(y).all()
    out = dict(t=np.array([T_MODEL], dtype=np.float32), x_sum=np.array(inp["x"].double().sum().item()),
               flux=y.numpy().astype(np.float32), row_stride=np.array(ROW_STRIDE), col_stride=np.array(COL_STRIDE),
               double_img=sample(taps["double_img"]).numpy(), double_txt=sample(taps["double_txt"]).numpy(),
               single=sample(taps["single"]).numpy(),
               torch_version=np.array(torch.__version__))
    # ---- the reference's own bf16 run (autocast on the CPU), same weights and inputs ----
    t1 = time.time()
    mb = model.to(torch.bfloat16)
    tapsb = {}
    mb.double_blocks[0]._forward_hooks.clear()
    mb.single_blocks[0]._forward_hooks.clear()
    mb.double_blocks[0].register_forward_hook(lambda m, a, out: tapsb.update(double_img=out[0], double_txt=out[1]))
    mb.single_blocks[0].register_forward_hook(lambda m, a, out: tapsb.update(single=out))
    with torch.no_grad(), torch.autocast("cpu", torch.bfloat16):
        yb = mb(torch.cat((inp["x"], inp["cond"]), -1).bfloat16(), timesteps=torch.tensor([T_MODEL]), txt=inp["txt"].bfloat16(),
                txt_ids=inp["txt_ids"], txt_mask=inp["txt_mask"], y=inp["y"].bfloat16(), img_ids=inp["img_ids"],
                img_mask=inp["img_mask"], guidance=inp["guidance"].bfloat16())
    assert yb.shape == y.shape and torch.isfinite(yb.float()).all()
    bits = lambda t: t.to(torch.bfloat16).contiguous().view(torch.int16).numpy().view(np.uint16)  # noqa: E731
    out.update(flux_bf16=bits(yb), double_img_bf16=bits(sample(tapsb["double_img"])), double_txt_bf16=bits(sample(tapsb["double_txt"])),
               single_bf16=bits(sample(tapsb["single"])), flux_bf16_dtype=np.array(str(yb.dtype)))
    print(f"bf16 autocast run: {time.time() - t1:.0f} s, output dtype {yb.dtype}; reference bf16-vs-fp32 on Flux.forward: rel-L2 "
          f"{float((yb.float() - y).norm() / y.norm()):.3e}")
    path = os.path.join(HERE, "fullwidth_reference.npz")
    np.savez_compressed(path, **out)
    print(f"wrote {path} ({os.path.getsize(path)} bytes) in {time.time() - t0:.0f} s; |flux| rms {float(y.pow(2).mean().sqrt()):.4f}")


if __name__ == "__main__":
    main()
