"""Golden vectors for the VAE decode path, produced by the REFERENCE ITSELF.

Runs only in the build container (needs /root/reference): imports models/modules/autoencoder.py unchanged (pure
torch + einops), builds AutoEncoder(TINY_AE) with the procedural weights of tests/procedural.py and stores inputs and
fp32 outputs (plus the reference module run in bfloat16 on CPU) in tests/golden/vae_golden.npz.

    python tests/golden/make_vae_golden.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get("VC_REFERENCE", "/root/reference")
sys.path.insert(0, REPO)
from tests.procedural import TINY_AE, procedural_ae_param, tiny_ae_image, tiny_ae_latent, tiny_ae_noise  # noqa: E402


def main():
    sys.path.insert(0, REF)
    from models.modules.autoencoder import AutoEncoder, AutoEncoderParams   # the reference file, unmodified
    torch.manual_seed(0)
    ae = AutoEncoder(AutoEncoderParams(**TINY_AE)).eval()
    sd = {k: procedural_ae_param(k, v.shape) for k, v in ae.state_dict().items()}
    ae.load_state_dict(sd)
    out = {}
    keys = list(sd)
    out["keys"] = np.array(keys)
    out["shapes"] = np.array([";".join(map(str, sd[k].shape)) for k in keys])
    with torch.no_grad():
        for name, (h, w) in {"sq": (4, 4), "rect": (4, 6)}.items():
            z = tiny_ae_latent(h, w, seed=5 if name == "sq" else 6)
            out[f"{name}_z"] = z.numpy()
            out[f"{name}_decode_fp32"] = ae.decode(z).numpy()
            out[f"{name}_decoder_fp32"] = ae.decoder(z).numpy()
            # intermediate taps of the fp32 reference
            d = ae.decoder
            hcur = d.conv_in(z)
            out[f"{name}_tap_conv_in"] = hcur.numpy()
            hcur = d.mid.block_1(hcur)
            out[f"{name}_tap_mid_block_1"] = hcur.numpy()
            hcur = d.mid.attn_1(hcur)
            out[f"{name}_tap_mid_attn_1"] = hcur.numpy()
        # encode path: image -> moments -> z, with the DiagonalGaussian noise fixed (torch.randn_like is patched to
        # return the procedural tensor, so the reference's own encode() consumes it)
        for name, (H, W) in {"sq": (16, 16), "rect": (16, 24)}.items():
            img = tiny_ae_image(H, W)
            noise = tiny_ae_noise(H // 2, W // 2)
            out[f"{name}_img"] = img.numpy()
            out[f"{name}_noise"] = noise.numpy()
            out[f"{name}_moments_fp32"] = ae.encoder(img).numpy()
            real = torch.randn_like
            torch.randn_like = lambda t, **kw: noise.to(t.dtype)
            try:
                out[f"{name}_encode_fp32"] = ae.encode(img).numpy()
            finally:
                torch.randn_like = real
        ae16 = AutoEncoder(AutoEncoderParams(**TINY_AE)).eval()
        ae16.load_state_dict(sd)
        ae16 = ae16.to(torch.bfloat16)
        z = tiny_ae_latent(4, 4, seed=5)
        out["sq_decode_refbf16"] = ae16.decode(z.to(torch.bfloat16)).float().numpy()
    path = os.path.join(HERE, "vae_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, {k: v.shape for k, v in out.items() if k.endswith("fp32")}, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
