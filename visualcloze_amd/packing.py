"""Host-side mirror of the latent-grid packing either side of the loop, on the HIP kernels
`vc_pack_latent` / `vc_pack_mask` / `vc_unpack_latent`:

* `prepare_grid(samples)`      the tensor part of `prepare_modified` (models/sampling.py:37-100): row latents ->
                               `img [B,N,64]`, `img_ids [B,N,3]` (axis0 = row index + 1), `img_mask [B,N]`
* `pack_cond(latents, masks)`  `cat(latent tokens, packed fill mask)` = the `cond` tensor (visualcloze.py:381-389)
* `unpack_rows(samples, sizes)` per-row slices of the result back to `[1,16,h,w]` latents (visualcloze.py:420-429)
"""
from __future__ import annotations

from typing import List, Sequence, Tuple

import torch

from . import hip


def grid_img_ids(rows_hw: Sequence[Tuple[int, int]]) -> torch.Tensor:
    """models/sampling.py:55-60: per row j of LATENT size (h, w): ids[..., 0] = j + 1, [..., 1] = y, [..., 2] = x."""
    out = []
    for j, (h, w) in enumerate(rows_hw):
        ids = torch.zeros(h // 2, w // 2, 3)
        ids[..., 0] = j + 1
        ids[..., 1] = ids[..., 1] + torch.arange(h // 2)[:, None]
        ids[..., 2] = ids[..., 2] + torch.arange(w // 2)[None, :]
        out.append(ids.reshape(-1, 3))
    return torch.cat(out, dim=0)


def _dev_bf16(t: torch.Tensor) -> torch.Tensor:
    if not t.is_cuda:
        raise hip.VclozeHipError("packing runs on the GPU: move the latents to the device first (no CPU path)")
    return t.to(torch.bfloat16).contiguous()


def prepare_grid(samples: List[List[torch.Tensor]]):
    """samples[b] = list of row latents [1,C,h,w] (device tensors).  Returns img, img_ids, img_mask like
    `prepare_modified` (ids and mask as device tensors; padded rows are zero)."""
    assert isinstance(samples, list) and all(isinstance(s, list) for s in samples)
    dev = samples[0][0].device
    counts = [sum((r.shape[-2] // 2) * (r.shape[-1] // 2) for r in rows) for rows in samples]
    n, C4 = max(counts), samples[0][0].shape[-3] * 4
    img = torch.zeros(len(samples), n, C4, dtype=torch.bfloat16, device=dev)
    img_ids = torch.zeros(len(samples), n, 3)
    img_mask = torch.zeros(len(samples), n, dtype=torch.int32)
    for b, rows in enumerate(samples):
        o = 0
        for r in rows:
            lat = _dev_bf16(r.squeeze(0))
            k = (lat.shape[-2] // 2) * (lat.shape[-1] // 2)
            hip.pack_latent(lat, img[b, o:o + k])
            o += k
        img_ids[b, :o] = grid_img_ids([tuple(r.shape[-2:]) for r in rows])
        img_mask[b, :o] = 1
    return img, img_ids.to(dev), img_mask.to(dev)


def pack_cond(latent_rows: List[torch.Tensor], mask_rows: List[torch.Tensor]) -> torch.Tensor:
    """One sample: row latents [1,16,h,w] + per-row PIXEL masks [1,1,8h,8w] -> cond [1, N, 64 + 256]."""
    dev = latent_rows[0].device
    n = sum((r.shape[-2] // 2) * (r.shape[-1] // 2) for r in latent_rows)
    cond = torch.empty(1, n, 320, dtype=torch.bfloat16, device=dev)
    o = 0
    for lat, m in zip(latent_rows, mask_rows):
        lat, m = _dev_bf16(lat.squeeze(0)), _dev_bf16(m.reshape(m.shape[-2], m.shape[-1]))
        k = (lat.shape[-2] // 2) * (lat.shape[-1] // 2)
        hip.pack_latent(lat, cond[0, o:o + k], col0=0)
        hip.pack_mask(m, cond[0, o:o + k], col0=64)
        o += k
    return cond


def unpack_rows(sample: torch.Tensor, sizes_hw: Sequence[Tuple[int, int]]) -> List[torch.Tensor]:
    """sample [1, N, 64] -> list of [1,16,h,w] latents for rows of LATENT size (h, w)."""
    tok = _dev_bf16(sample[0])
    out, o = [], 0
    for h, w in sizes_hw:
        k = (h // 2) * (w // 2)
        lat = torch.empty(tok.shape[-1] // 4, h, w, dtype=torch.bfloat16, device=tok.device)
        hip.unpack_latent(tok[o:o + k], lat)
        out.append(lat[None])
        o += k
    return out
