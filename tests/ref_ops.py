"""Plain-PyTorch f32 references of the individual HIP ops (same rounding points as the kernels).
Used by the -m gpu op tests and tests/tools/gpu_probe.py; runs on whatever device the inputs live on."""
import torch


def rb(x):
    return x.to(torch.bfloat16).float()


def gemm_ref(a, w, bias, epi, res=None, gate=None):
    y = a.float() @ w.float().t()
    if bias is not None:
        y = y + bias.float()
    y = rb(y)
    if epi == 1:
        y = rb(torch.nn.functional.gelu(y, approximate="tanh"))
    elif epi == 3:
        y = rb(torch.nn.functional.silu(y))
    elif epi == 2:
        y = rb(res.float() + rb(gate.float()[None, :] * y))
    return y


def ln_modulate_ref(x, shift, scale):
    ln = torch.nn.functional.layer_norm(x.float(), (x.shape[-1],), eps=1e-6)
    return rb(rb(1 + scale.float()) * ln + shift.float())


def qknorm_rope_ref(qkv, q_scale, k_scale, rope, H):
    """qkv [L, 3*H*128] bf16 -> (q,k after norm+rope as [L,H,128] f32, vt [H,128,L] f32)."""
    L = qkv.shape[0]
    x = qkv[:, : 3 * H * 128].float().reshape(L, 3, H, 128)
    outs = []
    for i, sc in ((0, q_scale), (1, k_scale)):
        t = x[:, i]
        rr = torch.rsqrt((t * t).mean(-1, keepdim=True) + 1e-6)
        t = rb(rb(t * rr) * sc.float())
        tp = t.reshape(L, H, 64, 2)
        cos, sin = rope[:, None, :, 0], rope[:, None, :, 1]
        o0 = cos * tp[..., 0] - sin * tp[..., 1]
        o1 = sin * tp[..., 0] + cos * tp[..., 1]
        outs.append(rb(torch.stack([o0, o1], -1).reshape(L, H, 128)))
    vt = x[:, 2].permute(1, 2, 0).contiguous()
    return outs[0], outs[1], vt


def attention_ref(q, k, v, kv_len=None):
    """q,k,v [L,H,128] f32 -> [L, H*128] (f32 softmax, bf16-rounded output; padded query rows 0)."""
    L, H, D = q.shape
    n = L if kv_len is None else kv_len
    s = torch.einsum("qhd,khd->hqk", q[:n], k[:n]) * D ** -0.5
    p = torch.softmax(s, dim=-1)
    o = torch.einsum("hqk,khd->qhd", p, v[:n])
    out = torch.zeros(L, H * D, device=q.device)
    out[:n] = o.reshape(n, H * D)
    return rb(out)
