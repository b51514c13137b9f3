"""VAE decode oracle vs golden vectors produced by the reference's own autoencoder.py (tests/golden/make_vae_golden.py),
and the state-dict contract of visualcloze_amd.vae against the reference's key list.  CPU only."""
import os

import numpy as np
import pytest
import torch

from oracle import vae_oracle as VO
from tests.procedural import TINY_AE, procedural_ae_param

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "vae_golden.npz"))


def tiny_sd():
    keys = [str(k) for k in G["keys"]]
    shapes = [tuple(int(x) for x in str(s).split(";")) for s in G["shapes"]]
    return {k: procedural_ae_param(k, s) for k, s in zip(keys, shapes)}


def rel_l2(a, b):
    a, b = torch.as_tensor(a).double(), torch.as_tensor(b).double()
    return float((a - b).norm() / b.norm())


@pytest.mark.parametrize("name", ["sq", "rect"])
def test_oracle_fp32_matches_reference(name):
    sd = tiny_sd()
    z = torch.tensor(G[f"{name}_z"])
    taps = {}
    out = VO.decode(sd, z, TINY_AE, "fp32")
    assert rel_l2(out, G[f"{name}_decode_fp32"]) < 2e-5
    dsd = {k[len("decoder."):]: v for k, v in sd.items() if k.startswith("decoder.")}
    out2 = VO.decoder_forward(dsd, z, TINY_AE, "fp32", taps)
    assert rel_l2(out2, G[f"{name}_decoder_fp32"]) < 2e-5
    for t in ("conv_in", "mid.block_1", "mid.attn_1"):
        assert rel_l2(taps[t], G[f"{name}_tap_{t.replace('.', '_')}"]) < 2e-5, t


def test_oracle_bf16_mode_tracks_reference_bf16_module():
    """The oracle's bf16 rounding points vs the reference module itself run in bfloat16 on CPU: both sit the same
    distance (bf16 noise) from the fp32 reference and within 2x of it from each other."""
    sd = tiny_sd()
    z = torch.tensor(G["sq_z"])
    o16 = VO.decode(sd, z, TINY_AE, "bf16")
    ref16, ref32 = G["sq_decode_refbf16"], G["sq_decode_fp32"]
    noise = rel_l2(ref16, ref32)
    assert rel_l2(o16, ref32) < 2.0 * noise + 1e-3
    assert rel_l2(o16, ref16) < 2.0 * noise + 1e-3


@pytest.mark.parametrize("name", ["sq", "rect"])
def test_oracle_encode_fp32_matches_reference(name):
    sd = tiny_sd()
    img, noise = torch.tensor(G[f"{name}_img"]), torch.tensor(G[f"{name}_noise"])
    esd = {k[len("encoder."):]: v for k, v in sd.items() if k.startswith("encoder.")}
    assert rel_l2(VO.encoder_forward(esd, img, TINY_AE, "fp32"), G[f"{name}_moments_fp32"]) < 2e-5
    assert rel_l2(VO.encode(sd, img, TINY_AE, noise, "fp32"), G[f"{name}_encode_fp32"]) < 2e-5


def test_state_dict_contract():
    from visualcloze_amd.vae import AutoEncoder, AutoEncoderParams
    ae = AutoEncoder(AutoEncoderParams(**TINY_AE))
    mine = {k: tuple(v.shape) for k, v in ae.state_dict().items()}
    keys = [str(k) for k in G["keys"]]
    shapes = [tuple(int(x) for x in str(s).split(";")) for s in G["shapes"]]
    assert list(mine) == keys                       # same names, same order as the reference's AutoEncoder
    assert [mine[k] for k in keys] == shapes


def test_decode_without_gpu_fails_loudly():
    from visualcloze_amd import hip
    from visualcloze_amd.vae import AutoEncoder, AutoEncoderParams
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    ae = AutoEncoder(AutoEncoderParams(**TINY_AE))
    with pytest.raises(hip.VclozeHipError):
        ae.decode(torch.zeros(1, TINY_AE["z_channels"], 4, 4))
    with pytest.raises(hip.VclozeHipError):
        ae.encode(torch.zeros(1, 3, 16, 16))
