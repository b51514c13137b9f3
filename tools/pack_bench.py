"""f1 evidence: the latent-grid packer / unpacker (csrc/pack.hip) - bytes moved / kernel time at cfg 3's row size (one row of the
512-grid 2x3: latent 16 x 64 x 192, mask 512 x 1536) and on a map large enough to stream (16 x 1024 x 2048), HIP events around
back-to-back launches.  Run it under `rocprofv3 --kernel-trace --stats` for the per-kernel averages (tools/measure_round.sh style).

    python tools/pack_bench.py [--iters 200]
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from visualcloze_amd import hip  # noqa: E402


def timed(fn, iters):
    s = hip.cur_stream()
    for _ in range(5):
        fn()
    e0, e1 = hip.Event(), hip.Event()
    e0.record(s)
    for _ in range(iters):
        fn()
    e1.record(s)
    return e0.elapsed_ms(e1) / iters * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=200)
    ap.add_argument("--only", default=None, help="substring of the shape tag: run that shape alone (the PMC passes use 'streaming')")
    a = ap.parse_args()
    hip.require_gpu()
    dev = "cuda:0"
    out = []
    for tag, (C, h, w) in (("cfg3 row (16 x 64 x 192)", (16, 64, 192)), ("p34 row (16 x 54 x 120)", (16, 54, 120)),
                           ("streaming map (16 x 1024 x 2048)", (16, 1024, 2048))):
        if a.only and a.only not in tag:
            continue
        lat = torch.randn(C, h, w, device=dev).to(torch.bfloat16)
        n = (h // 2) * (w // 2)
        cond = torch.empty(n, 320, dtype=torch.bfloat16, device=dev)
        tok = torch.empty(n, 64, dtype=torch.bfloat16, device=dev)
        mask = (torch.rand(8 * h, 8 * w, device=dev) > 0.5).to(torch.bfloat16)
        back = torch.empty_like(lat)
        legs = {
            "pack_latent": (lambda: hip.pack_latent(lat, tok), 2 * lat.numel() * 2),
            "pack_latent into cond[:, :64] (ld 320)": (lambda: hip.pack_latent(lat, cond, col0=0), 2 * lat.numel() * 2),
            "pack_mask into cond[:, 64:]": (lambda: hip.pack_mask(mask, cond, col0=64), 2 * mask.numel() * 2),
            "unpack_latent": (lambda: hip.unpack_latent(tok, back), 2 * lat.numel() * 2),
        }
        for name, (fn, nbytes) in legs.items():
            us = timed(fn, a.iters)
            out.append(dict(shape=tag, kernel=name, bytes=nbytes, us=round(us, 2), gb_per_s=round(nbytes / us / 1e3, 1)))
            print(f"{tag:34s} {name:40s} {nbytes / 1e6:9.2f} MB  {us:8.2f} us  {nbytes / us / 1e3:8.1f} GB/s", flush=True)
        hip.pack_latent(lat, tok)
        hip.unpack_latent(tok, back)
        torch.cuda.synchronize()
        assert torch.equal(back, lat), "pack -> unpack round trip"
    print(json.dumps(out))


if __name__ == "__main__":
    main()
