"""Text-encoder oracle vs golden vectors produced by transformers itself (tests/golden/make_text_golden.py), and the
state-dict contracts of visualcloze_amd.text against the transformers key lists.  CPU only."""
import os

import numpy as np
import pytest
import torch

from oracle import text_oracle as TO
from tests.procedural import TINY_CLIP, TINY_T5, procedural_text_param

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "text_golden.npz"))


def sd_of(prefix):
    keys = [str(k) for k in G[prefix + "_keys"]]
    shapes = [tuple(int(x) for x in str(s).split(";")) for s in G[prefix + "_shapes"]]
    sd = {k: procedural_text_param(k, s) for k, s in zip(keys, shapes)}
    if "shared.weight" in sd:
        sd["encoder.embed_tokens.weight"] = sd["shared.weight"]
    return sd, keys, shapes


def rel_l2(a, b):
    a, b = torch.as_tensor(a).double(), torch.as_tensor(b).double()
    return float((a - b).norm() / b.norm())


@pytest.mark.parametrize("name", ["t5_a", "t5_b"])
def test_t5_oracle_fp32_matches_transformers(name):
    sd, _, _ = sd_of("t5")
    out = TO.t5_encode(sd, torch.tensor(G[name + "_ids"]), TINY_T5, "fp32")
    assert rel_l2(out, G[name + "_fp32"]) < 2e-5


def test_t5_oracle_bf16_tracks_transformers_bf16():
    sd, _, _ = sd_of("t5")
    o16 = TO.t5_encode(sd, torch.tensor(G["t5_a_ids"]), TINY_T5, "bf16")
    noise = rel_l2(G["t5_a_refbf16"], G["t5_a_fp32"])
    assert rel_l2(o16, G["t5_a_fp32"]) < 2.0 * noise + 1e-3
    assert rel_l2(o16, G["t5_a_refbf16"]) < 2.0 * noise + 1e-3


@pytest.mark.parametrize("name", ["clip_a", "clip_b"])
def test_clip_oracle_fp32_matches_transformers(name):
    sd, _, _ = sd_of("clip")
    pooled, hs = TO.clip_text(sd, torch.tensor(G[name + "_ids"]), TINY_CLIP, "fp32")
    assert rel_l2(hs, G[name + "_hidden_fp32"]) < 2e-5
    assert rel_l2(pooled, G[name + "_pooled_fp32"]) < 2e-5


def test_relative_position_buckets_match_product_code():
    from visualcloze_amd.text import t5_relative_buckets
    for L in (64, 512):
        assert torch.equal(t5_relative_buckets(L, 32, 128), TO.t5_relative_buckets(L, 32, 128))


def test_state_dict_contracts():
    from visualcloze_amd.text import CLIPTextConfig, CLIPTextModel, T5Config, T5EncoderModel
    _, keys, shapes = sd_of("t5")
    mine = {k: tuple(v.shape) for k, v in T5EncoderModel(T5Config(**TINY_T5)).state_dict().items()}
    assert list(mine) == keys and [mine[k] for k in keys] == shapes
    _, keys, shapes = sd_of("clip")
    mine = {k: tuple(v.shape) for k, v in CLIPTextModel(CLIPTextConfig(**TINY_CLIP)).state_dict().items()}
    assert list(mine) == keys and [mine[k] for k in keys] == shapes


def test_text_models_without_gpu_fail_loudly():
    from visualcloze_amd import hip
    from visualcloze_amd.text import CLIPTextConfig, CLIPTextModel, T5Config, T5EncoderModel
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(hip.VclozeHipError):
        T5EncoderModel(T5Config(**TINY_T5))(torch.zeros(1, 64, dtype=torch.long))
    with pytest.raises(hip.VclozeHipError):
        CLIPTextModel(CLIPTextConfig(**TINY_CLIP))(torch.zeros(1, 16, dtype=torch.long))
