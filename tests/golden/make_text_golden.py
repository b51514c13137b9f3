"""Golden vectors for the text encoders, produced by TRANSFORMERS ITSELF (the third-party package the
reference's HFEmbedder wraps, models/modules/conditioner.py:5-37).  Runs only in the build container.

    python tests/golden/make_text_golden.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
from tests.procedural import TINY_CLIP, TINY_T5, procedural_text_param, tiny_ids  # noqa: E402


def main():
    import transformers
    from transformers import CLIPTextConfig, CLIPTextModel, T5Config, T5EncoderModel
    out = {"transformers_version": np.array(transformers.__version__)}
    with torch.no_grad():
        t5 = T5EncoderModel(T5Config(feed_forward_proj="gated-gelu", dropout_rate=0.0, **TINY_T5)).eval()
        sd = {k: procedural_text_param(k, v.shape) for k, v in t5.state_dict().items()}
        sd["encoder.embed_tokens.weight"] = sd["shared.weight"]
        t5.load_state_dict(sd)
        out["t5_keys"] = np.array(list(sd))
        out["t5_shapes"] = np.array([";".join(map(str, sd[k].shape)) for k in sd])
        for name, L in {"t5_a": 64, "t5_b": 128}.items():
            ids = tiny_ids(L, TINY_T5["vocab_size"], seed=len(name) + L)
            out[name + "_ids"] = ids.numpy()
            out[name + "_fp32"] = t5(input_ids=ids[None], attention_mask=None).last_hidden_state[0].numpy()
        t516 = T5EncoderModel(T5Config(feed_forward_proj="gated-gelu", dropout_rate=0.0, **TINY_T5)).eval()
        t516.load_state_dict(sd)
        t516 = t516.to(torch.bfloat16)
        out["t5_a_refbf16"] = t516(input_ids=torch.tensor(out["t5_a_ids"])[None], attention_mask=None).last_hidden_state[0].float().numpy()

        clip = CLIPTextModel(CLIPTextConfig(hidden_act="quick_gelu", bos_token_id=1, pad_token_id=0, **TINY_CLIP)).eval()
        csd = {}
        for k, v in clip.state_dict().items():
            kk = k if k.startswith("text_model.") else "text_model." + k      # checkpoint layout (older transformers keep the prefix)
            if kk.endswith("position_ids"):
                continue
            csd[kk] = procedural_text_param(kk, v.shape)
        clip.load_state_dict({(k if k in clip.state_dict() else k[len("text_model."):]): v for k, v in csd.items()}, strict=False)
        out["clip_keys"] = np.array(list(csd))
        out["clip_shapes"] = np.array([";".join(map(str, csd[k].shape)) for k in csd])
        for name, (L, eos_at) in {"clip_a": (24, 9), "clip_b": (16, 15)}.items():
            ids = tiny_ids(L, TINY_CLIP["vocab_size"], seed=L, eos=TINY_CLIP["eos_token_id"], eos_at=eos_at)
            out[name + "_ids"] = ids.numpy()
            r = clip(input_ids=ids[None])
            out[name + "_pooled_fp32"] = r.pooler_output[0].numpy()
            out[name + "_hidden_fp32"] = r.last_hidden_state[0].numpy()
    path = os.path.join(HERE, "text_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes; transformers", transformers.__version__)


if __name__ == "__main__":
    main()
