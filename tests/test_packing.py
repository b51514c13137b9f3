"""Latent-grid packer: oracle vs the reference's own prepare_modified outputs (CPU), and the HIP
packer kernels vs the golden vectors (GPU, bit-exact — pure data movement)."""
import numpy as np
import pytest
import torch

import oracle.flux_oracle as O


def test_oracle_prepare_grid_matches_reference(golden):
    rows = [torch.tensor(golden[f"pack_row{i}"]) for i in range(4)]
    img, ids, mask = O.prepare_grid([rows[:2], rows[2:]])
    assert torch.equal(img, torch.tensor(golden["pack_img"]))
    assert torch.equal(ids, torch.tensor(golden["pack_img_ids"]))
    assert torch.equal(mask.float(), torch.tensor(golden["pack_img_mask"]))
    assert float(ids[0, -1, 0]) == 2.0 and int(mask[1].sum()) == 16          # row index + 1; ragged second sample


def test_oracle_mask_and_unpack_match_reference(golden):
    assert torch.equal(O.pack_mask(torch.tensor(golden["maskpack_in"])[0, 0]), torch.tensor(golden["maskpack_out"])[0])
    tok = torch.tensor(golden["unpack_in"])[0]
    assert torch.equal(O.unpack_latent(tok, 4, 24), torch.tensor(golden["unpack_out"])[0])
    lat = torch.tensor(golden["pack_row0"])[0]
    assert torch.equal(O.unpack_latent(O.pack_latent(lat), 4, 12), lat)       # round trip


@pytest.mark.gpu
def test_hip_packer_matches_reference_vectors(golden):
    from visualcloze_amd import packing
    rows = [torch.tensor(golden[f"pack_row{i}"]).cuda() for i in range(4)]     # values are bf16-exact (procedural)
    img, ids, mask = packing.prepare_grid([rows[:2], rows[2:]])
    torch.cuda.synchronize()
    assert torch.equal(img.float().cpu(), torch.tensor(golden["pack_img"]))
    assert torch.equal(ids.cpu(), torch.tensor(golden["pack_img_ids"]))
    assert torch.equal(mask.float().cpu(), torch.tensor(golden["pack_img_mask"]))
    # cond = latent tokens || packed fill mask
    pm = torch.tensor(golden["maskpack_in"]).cuda()
    lat = torch.randn(1, 16, 4, 12, generator=torch.Generator().manual_seed(0)).bfloat16().cuda()
    cond = packing.pack_cond([lat], [pm])
    torch.cuda.synchronize()
    assert torch.equal(cond[0, :, 64:].float().cpu(), torch.tensor(golden["maskpack_out"])[0])
    assert torch.equal(cond[0, :, :64].float().cpu(), O.pack_latent(lat[0].float().cpu()))
    # unpack
    tok = torch.tensor(golden["unpack_in"]).cuda()
    out = packing.unpack_rows(tok, [(4, 24)])
    torch.cuda.synchronize()
    assert torch.equal(out[0].float().cpu(), torch.tensor(golden["unpack_out"]))


@pytest.mark.gpu
def test_hip_packer_round_trip_full_size():
    """cfg-2 rows (16 x 48 x 144 latents): pack -> unpack is the identity; ids follow the row/y/x layout."""
    from visualcloze_amd import packing
    g = torch.Generator().manual_seed(1)
    rows = [torch.randn(1, 16, 48, 144, generator=g).bfloat16().cuda() for _ in range(2)]
    img, ids, mask = packing.prepare_grid([rows])
    assert img.shape == (1, 3456, 64) and int(mask.sum()) == 3456
    back = packing.unpack_rows(img, [(48, 144), (48, 144)])
    torch.cuda.synchronize()
    for a, b in zip(rows, back):
        assert torch.equal(a, b)
    assert ids[0, 1728].tolist() == [2.0, 0.0, 0.0] and ids[0, 1727].tolist() == [1.0, 23.0, 71.0]
    assert torch.equal(img[0].float().cpu(), torch.cat([O.pack_latent(r[0].float().cpu()) for r in rows]))
