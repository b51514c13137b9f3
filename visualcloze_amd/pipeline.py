"""`VisualClozeModel.process_images` / `.upsampling` (visualcloze.py:147-245, 363-434) on the MI355X path.

    rows = denoise_grid(model, noise_rows, cond_latent_rows, mask_rows, txt, vec, cfg=30, steps=30)       # latent space
    up   = sdedit_upsample(model, noise, latent, blank_latent, txt, vec, cfg=30, steps=10, strength=0.4)  # latent space
    imgs = generate_grid(model, ae, t5, clip, row_images, row_masks, t5_ids, clip_ids, seed, ...)         # pixels in/out

`generate_grid` chains the whole tensor path of `process_images` (visualcloze.py:363-434): VAE-encode the grid rows,
pack latents + fill masks into `cond`, draw the per-row noise, encode the prompt with T5 / CLIP, run the fused sampler,
unpack, VAE-decode and map to [0, 1].  Image loading / resizing (PIL) and tokenisation are host work and stay with the
caller: inputs are row tensors in [-1, 1] and token ids.
"""
from __future__ import annotations

from typing import List, Optional, Sequence

import torch

from . import hip, packing
from .transport import Sampler, create_transport


def _kwargs(inp_ids, inp_mask, txt, vec, cond, cfg, dev):
    B, T = txt.shape[0], txt.shape[1]
    return dict(txt=txt, txt_ids=torch.zeros(B, T, 3, device=dev), txt_mask=torch.ones(B, T, dtype=torch.int32, device=dev),
                y=vec, img_ids=inp_ids, img_mask=inp_mask, cond=cond,
                guidance=torch.full((B,), cfg, device=dev, dtype=torch.bfloat16))   # visualcloze.py:413


@torch.no_grad()
def denoise_grid(model, noise_rows: List[torch.Tensor], cond_latent_rows: List[torch.Tensor],
                 mask_rows: List[torch.Tensor], txt: torch.Tensor, vec: torch.Tensor, cfg: float = 30.0,
                 steps: int = 30, solver: str = "euler", time_shifting_factor=1) -> List[torch.Tensor]:
    """One grid: per-row noise [1,16,h,w], VAE latents of the grid rows (already shifted/scaled), per-row PIXEL
    fill masks [1,1,8h,8w]; txt [1,T,4096], vec [1,768].  Returns the denoised row latents [1,16,h,w]."""
    dev = noise_rows[0].device
    img, img_ids, img_mask = packing.prepare_grid([noise_rows])                       # visualcloze.py:403
    cond = packing.pack_cond(cond_latent_rows, mask_rows)                              # :381-389
    fn = Sampler(create_transport("Linear", "velocity", do_shift=True)).sample_ode(
        sampling_method=solver, num_steps=steps, atol=1e-6, rtol=1e-3, reverse=False, do_shift=True,
        time_shifting_factor=time_shifting_factor)                                     # :284-292
    samples = fn(img, model.forward, _kwargs(img_ids, img_mask, txt, vec, cond, cfg, dev))[-1][:1]   # :415-420
    return packing.unpack_rows(samples, [tuple(r.shape[-2:]) for r in noise_rows])     # :425-429


@torch.no_grad()
def sdedit_upsample(model, noise: torch.Tensor, latent: torch.Tensor, blank_latent: torch.Tensor, txt: torch.Tensor,
                    vec: torch.Tensor, cfg: float = 30.0, steps: int = 10, strength: float = 0.4,
                    solver: str = "euler") -> torch.Tensor:
    """SDEdit refinement of one image (visualcloze.py:184-237): start from noise*(1-s) + latent*s, everything masked,
    un-shifted grid from t0 = strength.  noise/latent/blank_latent: [1,16,h,w]; returns [1,16,h,w]."""
    dev = latent.device
    h, w = latent.shape[-2:]
    img, img_ids, img_mask = packing.prepare_grid([[noise]])
    lat_tok, _, _ = packing.prepare_grid([[latent]])
    x0 = hip.sdedit_mix(img, lat_tok, strength)                                        # :221
    ones = torch.ones(1, 1, 8 * h, 8 * w, dtype=torch.bfloat16, device=dev)            # mask = 1 everywhere, :198
    cond = packing.pack_cond([blank_latent], [ones])                                   # :214
    fn = Sampler(create_transport("Linear", "velocity", do_shift=True)).sample_ode(
        sampling_method=solver, num_steps=steps, atol=1e-6, rtol=1e-3, reverse=False, do_shift=False,
        time_shifting_factor=1.0, strength=strength)                                   # :184-193
    sample = fn(x0, model.forward, _kwargs(img_ids, img_mask, txt, vec, cond, cfg, dev))[-1][:1]
    return packing.unpack_rows(sample, [(h, w)])[0]


@torch.no_grad()
def generate_grid(model, ae, t5, clip, row_images: List[torch.Tensor], row_masks: List[torch.Tensor],
                  t5_ids: torch.Tensor, clip_ids: torch.Tensor, seed: int, cfg: float = 30.0, steps: int = 30,
                  encode_noise: Optional[List[torch.Tensor]] = None, decode_rows: Optional[Sequence[int]] = None,
                  solver: str = "euler", time_shifting_factor=1) -> List[torch.Tensor]:
    """One grid, pixels in, pixels out (visualcloze.py:363-434).

    row_images[i]: [3, H, W_row] in [-1, 1] (the images of grid row i side by side); row_masks[i]: [1, 1, H, W_row]
    pixel fill mask (1 = generate); t5_ids [1, 512], clip_ids [1, 77]; `seed` drives the initial noise exactly like
    `torch.Generator(device).manual_seed(seed)` + one `torch.randn` per row (:394-399).  `encode_noise` are the VAE's
    DiagonalGaussian samples per row (None: drawn with torch.randn_like, as the reference's encode does).
    Returns the decoded rows `decode_rows` (default: all) as [3, H, W_row] tensors in [0, 1] (:430-432)."""
    dev = row_images[0].device
    lat = []
    for i, img in enumerate(row_images):                                               # :377-378
        n = None if encode_noise is None else encode_noise[i]
        lat.append(ae.encode(img[None].to(torch.bfloat16), noise=n))
    rng = torch.Generator(device=dev).manual_seed(int(seed))                           # :394
    noise = [torch.randn([1, 16, img.shape[-2] // 8, img.shape[-1] // 8], device=dev, generator=rng).to(torch.bfloat16)
             for img in row_images]                                                    # :395-399
    txt = t5(t5_ids)                                                                   # prepare_modified -> HFEmbedder
    vec, _ = clip(clip_ids)
    rows = denoise_grid(model, noise, lat, row_masks, txt, vec, cfg=cfg, steps=steps, solver=solver,
                        time_shifting_factor=time_shifting_factor)
    out = []
    for i in (range(len(rows)) if decode_rows is None else decode_rows):
        img = ae.decode(rows[i])[0]                                                    # :430
        out.append(((img.float() + 1.0) / 2.0).clamp_(0.0, 1.0))                       # :431-432
    return out
