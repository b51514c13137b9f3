"""VAE decoder on the GPU: glue kernels against torch references, the tiny decoder against the golden
vectors of the reference's autoencoder.py and the bf16 oracle, and a FLUX-width decoder against the oracle."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import vae_oracle as VO
from tests.procedural import TINY_AE, procedural_ae_param, ptensor, tiny_ae_latent

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
G = np.load(os.path.join(os.path.dirname(__file__), "golden", "vae_golden.npz"))


@pytest.fixture(scope="module")
def hip():
    from visualcloze_amd import hip as h
    h.require_gpu()
    return h


def rel_l2(a, b):
    a, b = torch.as_tensor(a).double().cpu(), torch.as_tensor(b).double().cpu()
    return float((a - b).norm() / b.norm())


def bf(t):
    return t.to(torch.bfloat16).to(DEV)


@pytest.mark.parametrize("H,W,C", [(4, 4, 64), (3, 5, 128), (1, 1, 8)])
def test_im2col3x3_stride2_is_exact(hip, H, W, C):
    """Downsample: pad (0,1,0,1) then 3x3 stride 2 (autoencoder.py:91-95)."""
    x = bf(ptensor((4 * H * W, C), 12, q=4))
    col = torch.empty(H * W, 9 * C, dtype=torch.bfloat16, device=DEV)
    hip.im2col3x3(x, col, H, W, down=True)
    img = F.pad(x.float().view(2 * H, 2 * W, C).permute(2, 0, 1)[None], (0, 1, 0, 1))
    ref = F.unfold(img, kernel_size=3, padding=0, stride=2)[0].view(C, 9, H * W).permute(2, 1, 0).reshape(H * W, 9 * C)
    assert torch.equal(col.float(), ref)


@pytest.mark.parametrize("H,W,C,up", [(4, 4, 64, False), (6, 10, 128, False), (8, 12, 64, True), (2, 2, 8, True)])
def test_im2col3x3_is_exact(hip, H, W, C, up):
    hs, ws = (H // 2, W // 2) if up else (H, W)
    x = bf(ptensor((hs * ws, C), 11, q=4))
    col = torch.empty(H * W, 9 * C, dtype=torch.bfloat16, device=DEV)
    hip.im2col3x3(x, col, H, W, up=up)
    img = x.float().view(hs, ws, C).permute(2, 0, 1)[None]
    if up:
        img = F.interpolate(img, scale_factor=2.0, mode="nearest")
    ref = F.unfold(img, kernel_size=3, padding=1)[0]                  # [(c, tap), H*W]
    ref = ref.view(C, 9, H * W).permute(2, 1, 0).reshape(H * W, 9 * C)  # -> [row, (tap, c)]
    assert torch.equal(col.float(), ref)


@pytest.mark.parametrize("HW,C,swish", [(16, 64, True), (300, 128, False), (1000, 256, True), (129, 512, True)])
def test_groupnorm_swish(hip, HW, C, swish):
    x = bf(ptensor((HW, C), 3, q=5) + 0.25)
    g, b = bf(ptensor((C,), 4, q=8, kmax=64, offset=1.0)), bf(ptensor((C,), 5, q=8, kmax=32))
    y = torch.empty_like(x)
    sc = torch.empty(hip.groupnorm_scratch_floats(HW), dtype=torch.float32, device=DEV)
    hip.groupnorm(x, g, b, y, sc, swish=swish)
    xr = x.float().t().reshape(1, C, HW, 1)
    ref = F.group_norm(xr, 32, g.float(), b.float(), eps=1e-6).to(torch.bfloat16).float()
    if swish:
        ref = (ref * torch.sigmoid(ref).to(torch.bfloat16).float()).to(torch.bfloat16).float()
    ref = ref.reshape(C, HW).t()
    err = (y.float() - ref).abs().max().item()
    assert err <= 2e-2 * ref.abs().max().item(), err
    y2 = torch.empty_like(x)
    hip.groupnorm(x, g, b, y2, sc, swish=swish)
    assert torch.equal(y, y2)                                            # deterministic reduction


@pytest.mark.parametrize("R,Cc", [(16, 16), (40, 24), (7, 2304), (3, 6912)])
def test_softmax_rows(hip, R, Cc):
    x = bf(ptensor((R, Cc), 9, q=3))
    ref = torch.softmax((x * 0.3).float(), dim=-1)          # scale * x is rounded to bf16 (a bf16 tensor op in the reference)
    hip.softmax_rows(x, 0.3)
    assert (x.float() - ref).abs().max().item() <= 8e-3 * ref.max().item() + 1e-6
    assert torch.allclose(x.float().sum(-1), torch.ones(R, device=DEV), atol=2e-2)


def test_transpose_and_layout_kernels(hip):
    x = bf(ptensor((37, 70), 13, q=4))
    t = torch.zeros(70, 40, dtype=torch.bfloat16, device=DEV)
    hip.transpose(x, t[:, :37])
    assert torch.equal(t[:, :37], x.t()) and float(t[:, 37:].abs().sum()) == 0.0
    z = ptensor((4, 3, 5), 17, q=5).to(DEV)
    nhwc = torch.empty(15, 64, dtype=torch.bfloat16, device=DEV)
    hip.nchw_to_nhwc(z.to(torch.bfloat16), nhwc, 0.3611, 0.1159)
    zb = z.to(torch.bfloat16)
    ref = ((zb / 0.3611) + 0.1159).float().reshape(4, 15).t()           # torch's own bf16 rounding sequence
    assert torch.equal(nhwc[:, :4].float(), ref) and float(nhwc[:, 4:].abs().sum()) == 0.0
    back = torch.empty(4, 3, 5, dtype=torch.float32, device=DEV)
    hip.nhwc_to_nchw(nhwc, back)
    assert torch.equal(back.reshape(4, 15).t(), nhwc[:, :4].float())


def tiny_model():
    from visualcloze_amd.vae import AutoEncoder, AutoEncoderParams
    ae = AutoEncoder(AutoEncoderParams(**TINY_AE))
    sd = {k: procedural_ae_param(k, v.shape) for k, v in ae.state_dict().items()}
    ae.load_state_dict(sd)
    return ae.to(DEV).to(torch.bfloat16), sd


@pytest.mark.parametrize("name", ["sq", "rect"])
def test_tiny_decode_matches_reference_golden_and_oracle(hip, name):
    ae, sd = tiny_model()
    z = torch.tensor(G[f"{name}_z"])
    out = ae.decode(z.to(DEV).to(torch.bfloat16)).float().cpu()
    ref32 = torch.tensor(G[f"{name}_decode_fp32"])
    o16 = VO.decode(sd, z, TINY_AE, "bf16")
    noise = rel_l2(o16, ref32)                      # what bf16 storage alone costs on these inputs
    assert out.shape == ref32.shape
    assert rel_l2(out, ref32) <= 3.0 * noise + 2e-3, (rel_l2(out, ref32), noise)
    assert rel_l2(out, o16) <= 2.0 * noise + 2e-3, (rel_l2(out, o16), noise)


def test_flux_width_decoder_matches_oracle(hip):
    """Full FLUX AutoEncoder geometry (ch 128, mult 1-2-4-4, 2 res blocks, z 16) on an 8x8 latent -> 64x64 image."""
    from visualcloze_amd.vae import FLUX_AE, AutoEncoder, AutoEncoderParams
    ae = AutoEncoder(AutoEncoderParams(**FLUX_AE))
    sd = {k: procedural_ae_param(k, v.shape) for k, v in ae.state_dict().items()}
    ae.load_state_dict(sd)
    ae = ae.to(DEV).to(torch.bfloat16)
    z = ptensor((1, 16, 8, 8), 21, q=5, kmax=96)
    out = ae.decode(z.to(DEV).to(torch.bfloat16)).float().cpu()
    assert out.shape == (1, 3, 64, 64) and torch.isfinite(out).all()
    o32 = VO.decode(sd, z, FLUX_AE, "fp32")
    o16 = VO.decode(sd, z, FLUX_AE, "bf16")
    noise = rel_l2(o16, o32)
    assert rel_l2(out, o32) <= 3.0 * noise + 2e-3, (rel_l2(out, o32), noise)
    out2 = ae.decode(z.to(DEV).to(torch.bfloat16)).float().cpu()
    assert torch.equal(out, out2)


@pytest.mark.parametrize("name", ["sq", "rect"])
def test_tiny_encode_matches_reference_golden_and_oracle(hip, name):
    ae, sd = tiny_model()
    img, noise = torch.tensor(G[f"{name}_img"]), torch.tensor(G[f"{name}_noise"])
    mom = ae.encoder(img.to(DEV).to(torch.bfloat16)).float().cpu()
    ref_m = torch.tensor(G[f"{name}_moments_fp32"])
    esd = {k[len("encoder."):]: v for k, v in sd.items() if k.startswith("encoder.")}
    noise_m = rel_l2(VO.encoder_forward(esd, img, TINY_AE, "bf16"), ref_m)
    assert mom.shape == ref_m.shape
    assert rel_l2(mom, ref_m) <= 3.0 * noise_m + 2e-3, (rel_l2(mom, ref_m), noise_m)
    z = ae.encode(img.to(DEV).to(torch.bfloat16), noise=noise.to(DEV)).float().cpu()
    ref_z = torch.tensor(G[f"{name}_encode_fp32"])
    o16 = VO.encode(sd, img, TINY_AE, noise, "bf16")
    nz = rel_l2(o16, ref_z)
    assert rel_l2(z, ref_z) <= 3.0 * nz + 2e-3, (rel_l2(z, ref_z), nz)
    zm = ae.encode(img.to(DEV).to(torch.bfloat16), sample=False).float().cpu()        # sample=False -> the mean
    assert rel_l2(zm, VO.encode(sd, img, TINY_AE, None, "bf16")) <= 2.0 * nz + 2e-3


def test_gaussian_sample_kernel_rounding(hip):
    """scale * ((mean + exp(0.5*logvar) * noise) - shift) with torch's own bf16 rounding sequence."""
    Z, h, w = 4, 3, 5
    mom = bf(ptensor((h * w, 8), 31, q=5))
    noise = bf(ptensor((Z, h, w), 32, q=5))
    out = torch.empty(Z, h, w, dtype=torch.bfloat16, device=DEV)
    hip.gaussian_sample(mom, noise, out, 0.3611, 0.1159)
    m = mom.t().reshape(8, h, w)
    mean, logvar = m[:Z], m[Z:]
    ref = 0.3611 * ((mean + torch.exp(0.5 * logvar) * noise) - 0.1159)       # bf16 tensor arithmetic
    assert (out.float() - ref.float()).abs().max().item() <= 2 ** -7 * ref.float().abs().max().item()


def test_flux_width_encoder_matches_oracle(hip):
    from visualcloze_amd.vae import FLUX_AE, AutoEncoder, AutoEncoderParams
    ae = AutoEncoder(AutoEncoderParams(**FLUX_AE))
    sd = {k: procedural_ae_param(k, v.shape) for k, v in ae.state_dict().items()}
    ae.load_state_dict(sd)
    ae = ae.to(DEV).to(torch.bfloat16)
    img = ptensor((1, 3, 64, 64), 41, q=7, kmax=127)
    mom = ae.encoder(img.to(DEV).to(torch.bfloat16)).float().cpu()
    esd = {k[len("encoder."):]: v for k, v in sd.items() if k.startswith("encoder.")}
    o32 = VO.encoder_forward(esd, img, FLUX_AE, "fp32")
    o16 = VO.encoder_forward(esd, img, FLUX_AE, "bf16")
    noise = rel_l2(o16, o32)
    assert mom.shape == (1, 32, 8, 8)
    assert rel_l2(mom, o32) <= 3.0 * noise + 2e-3, (rel_l2(mom, o32), noise)


def test_decoder_attention_with_odd_token_count(hip):
    """h*w not a multiple of 8 (a 400x400-pixel image has a 50x50 latent): padded key rows / score columns stay zero."""
    ae, sd = tiny_model()
    z = ptensor((1, TINY_AE["z_channels"], 5, 7), 51, q=5, kmax=96)
    out = ae.decode(z.to(DEV).to(torch.bfloat16)).float().cpu()
    o32 = VO.decode(sd, z, TINY_AE, "fp32")
    o16 = VO.decode(sd, z, TINY_AE, "bf16")
    noise = rel_l2(o16, o32)
    assert out.shape == o32.shape
    assert rel_l2(out, o32) <= 3.0 * noise + 2e-3, (rel_l2(out, o32), noise)


@pytest.mark.parametrize("H,W,C,O,mode", [(4, 4, 64, 64, 0), (9, 13, 128, 200, 0), (8, 12, 64, 72, 1), (3, 5, 64, 8, 2),
                                          (40, 56, 256, 128, 0), (34, 30, 64, 320, 1)])
@pytest.mark.parametrize("with_res", [False, True])
def test_conv3x3_implicit_gemm_equals_im2col_plus_gemm(hip, H, W, C, O, mode, with_res):
    """One-launch convolution (the loader waves gather the taps) vs the explicit im2col matrix + GEMM: bit-identical."""
    up, down = mode == 1, mode == 2
    hs, ws = (H // 2, W // 2) if up else (2 * H, 2 * W) if down else (H, W)
    x = torch.zeros(hs * ws + 1, C, dtype=torch.bfloat16, device=DEV)          # last row = the zero row
    x[:-1] = bf(ptensor((hs * ws, C), 61, q=5))
    w, b = bf(ptensor((O, 9 * C), 62, q=9)), bf(ptensor((O,), 63, q=6))
    res, gate = bf(ptensor((H * W, O), 64, q=5)), torch.ones(O, dtype=torch.bfloat16, device=DEV)
    col = torch.empty(H * W, 9 * C, dtype=torch.bfloat16, device=DEV)
    hip.im2col3x3(x[:-1], col, H, W, up=up, down=down)
    ref = torch.empty(H * W, O, dtype=torch.bfloat16, device=DEV)
    if with_res:
        hip.gemm(hip.make_problem(col, w, b, ref, res=res, gate=gate, rows_per_batch=H * W), epi=hip.EPI_GATE_RES)
    else:
        hip.gemm(hip.make_problem(col, w, b, ref), epi=hip.EPI_BIAS)
    out = torch.full((H * W, O), float("nan"), dtype=torch.bfloat16, device=DEV)
    hip.conv3x3(x, w, b, out, H, W, up=up, down=down, res=res if with_res else None, gate=gate if with_res else None)
    torch.cuda.synchronize()
    assert torch.equal(out, ref)
