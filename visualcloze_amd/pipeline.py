"""`VisualClozeModel.process_images` / `.upsampling` (visualcloze.py:147-245, 363-434) on the MI355X path.

    rows = denoise_grid(model, noise_rows, cond_latent_rows, mask_rows, txt, vec, cfg=30, steps=30)       # latent space
    up   = sdedit_upsample(model, noise, latent, blank_latent, txt, vec, cfg=30, steps=10, strength=0.4)  # latent space
    imgs = generate_grid(model, ae, t5, clip, row_images, row_masks, t5_ids, clip_ids, seed, ...)         # pixels in/out
    outs = generate_and_upsample(model, ae, t5, clip, ..., grid_w, mask_position, upsampling_size, ...)   # both stages chained

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
def sdedit_upsample_batch(model, noises: Sequence[torch.Tensor], latents: Sequence[torch.Tensor],
                          blank_latents: Sequence[torch.Tensor], txt: torch.Tensor, vec: torch.Tensor, cfg: float = 30.0,
                          steps: int = 10, strength: float = 0.4, solver: str = "euler") -> List[torch.Tensor]:
    """`sdedit_upsample` for K targets of ONE size and prompt advanced TOGETHER: the reference refines the masked cells of a
    grid one after the other (visualcloze.py:450-465), each an independent ODE solve over the same (T, N), so they stack into
    a per-GPU batch - one graph replay per solver step moves all of them (chunks of <= 4) and the GEMMs see M = K * L rows:
    at L = 4608 that is +14 % (K = 2) / +17 % (K = 4) evaluations per second over one target at a time
    (profiles/r04a_per_gpu_batch_probe.json).  Per target the result equals the one-at-a-time call up to the bf16 noise of
    another GEMM tile plan (tests/test_model_gpu.py).  txt [1,T,4096] / vec [1,768]: the shared content prompt."""
    K = len(noises)
    if K == 0:
        return []
    dev = latents[0].device
    h, w = latents[0].shape[-2:]
    if any(t.shape[-2:] != (h, w) for t in list(noises) + list(latents) + list(blank_latents)):
        raise ValueError("sdedit_upsample_batch: all targets must share one latent size")
    img, img_ids, img_mask = packing.prepare_grid([[n] for n in noises])
    lat_tok, _, _ = packing.prepare_grid([[l] for l in latents])
    x0 = hip.sdedit_mix(img, lat_tok, strength)                                        # :221, all targets at once
    ones = torch.ones(1, 1, 8 * h, 8 * w, dtype=torch.bfloat16, device=dev)
    cond = torch.cat([packing.pack_cond([b], [ones]) for b in blank_latents], dim=0)
    fn = Sampler(create_transport("Linear", "velocity", do_shift=True)).sample_ode(
        sampling_method=solver, num_steps=steps, atol=1e-6, rtol=1e-3, reverse=False, do_shift=False,
        time_shifting_factor=1.0, strength=strength)
    kw = _kwargs(img_ids, img_mask, txt.expand(K, -1, -1), vec.expand(K, -1), cond, cfg, dev)
    sample = fn(x0, model.forward, kw)[-1]
    return [packing.unpack_rows(sample[k:k + 1], [(h, w)])[0] for k in range(K)]


@torch.no_grad()
def generate_grid(model, ae, t5, clip, row_images: List[torch.Tensor], row_masks: List[torch.Tensor],
                  t5_ids: torch.Tensor, clip_ids: torch.Tensor, seed: int, cfg: float = 30.0, steps: int = 30,
                  encode_noise: Optional[List[torch.Tensor]] = None, decode_rows: Optional[Sequence[int]] = None,
                  solver: str = "euler", time_shifting_factor=1, rng: Optional[torch.Generator] = None) -> List[torch.Tensor]:
    """One grid, pixels in, pixels out (visualcloze.py:363-434).

    row_images[i]: [3, H, W_row] in [-1, 1] (the images of grid row i side by side); row_masks[i]: [1, 1, H, W_row]
    pixel fill mask (1 = generate); t5_ids [1, 512], clip_ids [1, 77]; `seed` drives the initial noise exactly like
    `torch.Generator(device).manual_seed(seed)` + one `torch.randn` per row (:394-399).  `encode_noise` are the VAE's
    DiagonalGaussian samples per row (None: drawn with torch.randn_like, as the reference's encode does).
    `rng`: the generator to draw from instead of a fresh one seeded with `seed` (the upsampling stage keeps drawing from
    the SAME generator, :450-458).  Returns the decoded rows `decode_rows` (default: all) as [3, H, W_row] tensors in
    [0, 1] (:430-432)."""
    dev = row_images[0].device
    lat = []
    for i, img in enumerate(row_images):                                               # :377-378
        n = None if encode_noise is None else encode_noise[i]
        lat.append(ae.encode(img[None].to(torch.bfloat16), noise=n))
    if rng is None:
        rng = torch.Generator(device=dev).manual_seed(int(seed))                       # :394
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


def to_uint8_image(img01: torch.Tensor):
    """`to_pil_image(row_sample.float())` (visualcloze.py:437-439): [3,H,W] in [0,1] -> 8-bit RGB PIL image; torchvision's
    conversion is `pic.mul(255).byte()`, i.e. it TRUNCATES."""
    from PIL import Image
    a = img01.detach().float().mul(255).to(torch.uint8).permute(1, 2, 0).contiguous().cpu().numpy()
    return Image.fromarray(a)


def upsampling_size(target_size):
    """The size rule of `upsampling` (visualcloze.py:165-179): default 1024x1024, at most 1024^2 pixels at the requested
    aspect ratio, both sides rounded down to multiples of 16.  (w, h) in, (w, h) out."""
    if target_size is None:
        target_size = (1024, 1024)
    if target_size[0] * target_size[1] > 1024 * 1024:
        aspect = target_size[0] / target_size[1]
        new_h = int((1024 * 1024 / aspect) ** 0.5)
        target_size = (int(new_h * aspect), new_h)
    return (target_size[0] // 16) * 16, (target_size[1] // 16) * 16


@torch.no_grad()
def upsample_image(model, ae, t5, clip, image, target_size, t5_ids: torch.Tensor, clip_ids: torch.Tensor,
                   rng: torch.Generator, cfg: float = 30.0, steps: int = 10, strength: float = 0.4,
                   encode_noise: Optional[Sequence[torch.Tensor]] = None, solver: str = "euler") -> torch.Tensor:
    """`VisualClozeModel.upsampling` (visualcloze.py:147-245) for one image: resize on the host (PIL, bicubic - the
    default of `Image.resize`), VAE-encode the image and a blank, SDEdit from `strength` with everything masked, decode.
    image: PIL image or [3,H,W] tensor in [0,1]; t5_ids / clip_ids: the tokenised CONTENT prompt (prefix stripping and
    tokenisation are host work, :149-163); `encode_noise`: the two DiagonalGaussian draws (image, blank) or None for
    torch.randn_like as the reference's `.sample()`.  Returns [3, h, w] in [0, 1]."""
    import numpy as np
    dev = next(ae.parameters()).device
    if torch.is_tensor(image):
        image = to_uint8_image(image)
    image = image.convert("RGB").resize(upsampling_size(target_size))                  # :179
    px = torch.from_numpy(np.asarray(image, dtype=np.uint8).copy()).permute(2, 0, 1).float().div(255.0)   # ToTensor, :133-137
    if strength >= 1.0:
        return px.to(dev)                                                              # :180-181
    px = ((px - 0.5) / 0.5).to(dev, torch.bfloat16)                                    # Normalize(0.5, 0.5)
    n_img, n_blank = (None, None) if encode_noise is None else encode_noise
    latent = ae.encode(px[None], noise=None if n_img is None else n_img[None] if n_img.dim() == 3 else n_img)      # :200,202
    blank = ae.encode(torch.zeros_like(px)[None], noise=None if n_blank is None else n_blank[None] if n_blank.dim() == 3 else n_blank)
    noise = torch.randn([1, 16, latent.shape[-2], latent.shape[-1]], device=dev, generator=rng).to(torch.bfloat16)    # :217
    txt = t5(t5_ids)
    vec, _ = clip(clip_ids)
    z = sdedit_upsample(model, noise, latent, blank, txt, vec, cfg=cfg, steps=steps, strength=strength, solver=solver)
    img = ae.decode(z)[0]                                                              # :238
    return ((img.float() + 1.0) / 2.0).clamp_(0.0, 1.0)                                # :239-240


@torch.no_grad()
def upsample_images(model, ae, t5, clip, images: Sequence, target_size, t5_ids: torch.Tensor, clip_ids: torch.Tensor,
                    rng: torch.Generator, cfg: float = 30.0, steps: int = 10, strength: float = 0.4,
                    encode_noise: Optional[Sequence] = None, solver: str = "euler") -> List[torch.Tensor]:
    """`upsample_image` for several images of one grid in ONE batched SDEdit solve (`sdedit_upsample_batch`).  Host work
    (resize), the VAE encodes and every random draw happen per image in the reference's order - image k: encode(image),
    encode(blank), noise from `rng` (visualcloze.py:200-217) - so each target starts from exactly the state the
    one-at-a-time loop gives it; the prompt is encoded once (it is the same for every target, :458)."""
    import numpy as np
    dev = next(ae.parameters()).device
    size = upsampling_size(target_size)
    pxs = []
    for image in images:
        if torch.is_tensor(image):
            image = to_uint8_image(image)
        image = image.convert("RGB").resize(size)                                      # :179
        pxs.append(torch.from_numpy(np.asarray(image, dtype=np.uint8).copy()).permute(2, 0, 1).float().div(255.0))
    if strength >= 1.0:
        return [px.to(dev) for px in pxs]                                              # :180-181
    lat, blank, noise = [], [], []
    for k, px in enumerate(pxs):
        px = ((px - 0.5) / 0.5).to(dev, torch.bfloat16)
        n_img, n_blank = (None, None) if encode_noise is None or encode_noise[k] is None else encode_noise[k]
        nz = lambda n: None if n is None else n[None] if n.dim() == 3 else n  # noqa: E731
        lat.append(ae.encode(px[None], noise=nz(n_img)))
        blank.append(ae.encode(torch.zeros_like(px)[None], noise=nz(n_blank)))
        noise.append(torch.randn([1, 16, lat[-1].shape[-2], lat[-1].shape[-1]], device=dev, generator=rng).to(torch.bfloat16))
    txt = t5(t5_ids)
    vec, _ = clip(clip_ids)
    zs = sdedit_upsample_batch(model, noise, lat, blank, txt, vec, cfg=cfg, steps=steps, strength=strength, solver=solver)
    return [((ae.decode(z)[0].float() + 1.0) / 2.0).clamp_(0.0, 1.0) for z in zs]     # :238-240


@torch.no_grad()
def generate_and_upsample(model, ae, t5, clip, row_images: List[torch.Tensor], row_masks: List[torch.Tensor],
                          t5_ids: torch.Tensor, clip_ids: torch.Tensor, seed: int, grid_w: int, mask_position: Sequence[bool],
                          target_size=None, content_t5_ids: Optional[torch.Tensor] = None,
                          content_clip_ids: Optional[torch.Tensor] = None, cfg: float = 30.0, steps: int = 30,
                          upsampling_steps: int = 10, upsampling_noise: float = 0.4, is_upsampling: bool = True,
                          encode_noise: Optional[List[torch.Tensor]] = None, upsample_encode_noise=None,
                          solver: str = "euler", time_shifting_factor=1, batch_targets: bool = True):
    """Both stages of `process_images` chained (visualcloze.py:363-465): the grid is generated, its LAST row is decoded and
    quantised to 8 bits as `to_pil_image` does, every cell of that row whose `mask_position` is set is cropped
    (:452,461) and - with `is_upsampling` - refined by `upsample_image` at `target_size`, all noise coming from ONE
    generator seeded with `seed` (:394,456).  `batch_targets` (default): the masked cells are refined TOGETHER, one graph replay
    per solver step for all of them (`upsample_images`; same draws in the same order, per-target results equal up to bf16
    noise); False = one after the other as the reference loops.  Returns the output images as [3, h, w] tensors in [0, 1]."""
    dev = row_images[0].device
    rng = torch.Generator(device=dev).manual_seed(int(seed))
    last = len(row_images) - 1
    row = generate_grid(model, ae, t5, clip, row_images, row_masks, t5_ids, clip_ids, seed, cfg=cfg, steps=steps,
                        encode_noise=encode_noise, decode_rows=[last], solver=solver, time_shifting_factor=time_shifting_factor,
                        rng=rng)[0]
    pil = to_uint8_image(row)                                                          # :437-439
    ret_w, ret_h = pil.width, pil.height
    cells = [pil.crop((i * ret_w // grid_w, 0, (i + 1) * ret_w // grid_w, ret_h)) for i, m in enumerate(mask_position) if m]   # :452,461
    ct5 = content_t5_ids if content_t5_ids is not None else t5_ids
    cclip = content_clip_ids if content_clip_ids is not None else clip_ids
    if is_upsampling and batch_targets and len(cells) > 1:
        return upsample_images(model, ae, t5, clip, cells, target_size, ct5, cclip, rng, cfg=cfg, steps=upsampling_steps,
                               strength=upsampling_noise, encode_noise=upsample_encode_noise, solver=solver)
    outs, k = [], 0
    for i, masked in enumerate(mask_position):
        if not masked:
            continue
        cell = pil.crop((i * ret_w // grid_w, 0, (i + 1) * ret_w // grid_w, ret_h))    # :452,461
        if is_upsampling:
            en = None if upsample_encode_noise is None else upsample_encode_noise[k]
            outs.append(upsample_image(model, ae, t5, clip, cell, target_size, content_t5_ids if content_t5_ids is not None else t5_ids,
                                       content_clip_ids if content_clip_ids is not None else clip_ids, rng, cfg=cfg,
                                       steps=upsampling_steps, strength=upsampling_noise, encode_noise=en, solver=solver))
        else:
            import numpy as np
            outs.append(torch.from_numpy(np.asarray(cell, dtype=np.uint8).copy()).permute(2, 0, 1).float().div(255.0).to(dev))
        k += 1
    return outs
