"""VAE encode / decode timing on one MI355X: FLUX AutoEncoder geometry, procedural weights, HIP events.

    python tools/vae_bench.py
"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from visualcloze_amd import hip
from visualcloze_amd.vae import FLUX_AE, AutoEncoder, AutoEncoderParams
from tests.procedural import procedural_ae_param, ptensor

dev = "cuda:0"


def conv_flops(ae, H, W, decode):
    """2*MACs of every convolution / projection / attention GEMM for an HxW image (decode) or input image (encode)."""
    p = AutoEncoderParams(**FLUX_AE)
    fl = 0.0
    def res(cin, cout, hw):
        f = 2.0 * hw * 9 * cin * cout + 2.0 * hw * 9 * cout * cout
        if cin != cout: f += 2.0 * hw * cin * cout
        return f
    def attn(c, hw):
        return 4 * 2.0 * hw * c * c + 2 * 2.0 * hw * hw * c
    n = len(p.ch_mult)
    if decode:
        h, w = H // 8, W // 8
        hw = h * w
        cin = p.ch * p.ch_mult[-1]
        fl += 2.0 * hw * 9 * p.z_channels * cin + 2 * res(cin, cin, hw) + attn(cin, hw)
        for lvl in reversed(range(n)):
            cout = p.ch * p.ch_mult[lvl]
            for _ in range(p.num_res_blocks + 1):
                fl += res(cin, cout, hw); cin = cout
            if lvl != 0:
                hw *= 4
                fl += 2.0 * hw * 9 * cin * cin
        fl += 2.0 * hw * 9 * cin * p.out_ch
    else:
        hw = H * W
        cin = p.ch
        fl += 2.0 * hw * 9 * p.in_channels * cin
        for lvl in range(n):
            cout = p.ch * p.ch_mult[lvl]
            for _ in range(p.num_res_blocks):
                fl += res(cin, cout, hw); cin = cout
            if lvl != n - 1:
                hw //= 4
                fl += 2.0 * hw * 9 * cin * cin
        fl += 2 * res(cin, cin, hw) + attn(cin, hw) + 2.0 * hw * 9 * cin * 2 * p.z_channels
    return fl


def timeit(fn, n=5):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def main():
    argparse.ArgumentParser().parse_args()
    hip.require_gpu()
    ae = AutoEncoder(AutoEncoderParams(**FLUX_AE))
    sd = {k: procedural_ae_param(k, v.shape) for k, v in ae.state_dict().items()}
    ae.load_state_dict(sd)
    ae = ae.to(dev).to(torch.bfloat16)
    rec = {"device": torch.cuda.get_device_name(0), "dtype": "bf16", "weights": "procedural", "cases": []}
    for (H, W) in [(384, 384), (384, 1152), (512, 512)]:
        z = ptensor((1, 16, H // 8, W // 8), 3, q=5, kmax=96).to(dev).to(torch.bfloat16)
        img = ptensor((1, 3, H, W), 4, q=7).to(dev).to(torch.bfloat16)
        noise = ptensor((1, 16, H // 8, W // 8), 5, q=5).to(dev).to(torch.bfloat16)
        md = timeit(lambda: ae.decode(z))
        me = timeit(lambda: ae.encode(img, noise=noise))
        fd, fe = conv_flops(ae, H, W, True), conv_flops(ae, H, W, False)
        rec["cases"].append({"image": f"{H}x{W}", "decode_ms": round(md, 3), "decode_tflops": round(fd / md / 1e9, 1), "decode_gflop": round(fd / 1e9, 1),
                             "encode_ms": round(me, 3), "encode_tflops": round(fe / me / 1e9, 1), "encode_gflop": round(fe / 1e9, 1)})
    print(json.dumps(rec))


if __name__ == "__main__":
    main()
