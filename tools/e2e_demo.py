"""End-to-end tensor path of process_images at BASELINE cfg 2 on one MI355X: full FLUX.1-Fill-dev + LoRA geometry, the FLUX
AutoEncoder, T5-XXL and CLIP-L, all with random weights (timing / plumbing only; parity lives in tests/).  Prints one JSON
line with the per-stage times of `pipeline.generate_grid`'s pieces.     python tools/e2e_demo.py
`--cfg5`: BASELINE cfg 5 as a whole on one GPU - a 3x4 grid at 384 (L = 7424), 50 solver points, the last row's last TWO cells
masked, then the SDEdit upsampling of both targets to 1024x1024 (10 points from strength 0.4) - through
`pipeline.generate_and_upsample`, with the targets refined together (default) and one after the other."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench as B
from visualcloze_amd import hip, pipeline
from visualcloze_amd.text import CLIPTextConfig, CLIPTextModel, T5Config, T5EncoderModel
from visualcloze_amd.vae import FLUX_AE, AutoEncoder, AutoEncoderParams

dev = torch.device("cuda", 0)


def randomize(m, std=None):
    g = torch.Generator(device=dev).manual_seed(11)
    with torch.no_grad():
        for n, p in m.named_parameters():
            if ("norm" in n and n.endswith("weight")):
                p.fill_(1.0)
            elif n.endswith("bias"):
                p.zero_()
            else:
                fan_in = p[0].numel() if p.dim() > 1 else p.numel()
                p.normal_(0.0, fan_in ** -0.5, generator=g)


def timed(fn):
    torch.cuda.synchronize(); t0 = time.perf_counter(); r = fn(); torch.cuda.synchronize()
    return r, (time.perf_counter() - t0) * 1e3


def cfg5(model, ae, t5, clip):
    H, W, rows, cols = 384, 384, 3, 4
    Wrow = W * cols
    imgs = [torch.rand(3, H, Wrow, device=dev) * 2 - 1 for _ in range(rows)]
    masks = [torch.zeros(1, 1, H, Wrow, device=dev) for _ in range(rows)]
    masks[-1][..., 2 * W:] = 1                                          # the last two cells of the last row are generated
    t5_ids = torch.randint(0, 32000, (1, 512), device=dev)
    clip_ids = torch.randint(0, 49000, (1, 77), device=dev); clip_ids[0, 30] = 49407
    rec = {"grid": "3x4 @384 (L = 512 + 6912), 50 solver points; 2 masked targets upsampled to 1024x1024, 10 points from strength 0.4"}
    for tag, together in (("targets_together", True), ("targets_one_by_one", False)):
        for rep in range(2):                                            # second pass = warm
            out, ms = timed(lambda: pipeline.generate_and_upsample(model, ae, t5, clip, imgs, masks, t5_ids, clip_ids, 0, cols,
                                                                   [False, False, True, True], target_size=None, steps=50,
                                                                   upsampling_steps=10, upsampling_noise=0.4, batch_targets=together))
        assert len(out) == 2 and out[0].shape == (3, 1024, 1024) and all(torch.isfinite(o).all() for o in out)
        rec[tag + "_ms"] = round(ms, 1)
    _, ms1 = timed(lambda: pipeline.generate_grid(model, ae, t5, clip, imgs, masks, t5_ids, clip_ids, 0, steps=50, decode_rows=[rows - 1]))
    rec["stage1_alone_ms"] = round(ms1, 1)
    rec["upsampling_stage_ms"] = {k: round(rec[k + "_ms"] - ms1, 1) for k in ("targets_together", "targets_one_by_one")}
    rec["images_per_sec"] = round(2e3 / rec["targets_together_ms"], 4)
    # the SDEdit solves alone (latent space, 1024x1024 targets: N = 4096, 9 evaluations each), warm
    txt, vec = t5(t5_ids), clip(clip_ids)[0]
    r = lambda: torch.randn(1, 16, 128, 128, device=dev).to(torch.bfloat16)  # noqa: E731
    n, l, b = [r(), r()], [r(), r()], [r(), r()]
    for rep in range(2):
        _, t_b = timed(lambda: pipeline.sdedit_upsample_batch(model, n, l, b, txt, vec, steps=10, strength=0.4))
        _, t_s = timed(lambda: [pipeline.sdedit_upsample(model, n[k], l[k], b[k], txt, vec, steps=10, strength=0.4) for k in range(2)])
    rec["sdedit_solves_only_ms"] = {"two_targets_together": round(t_b, 1), "two_targets_one_by_one": round(t_s, 1)}
    print(json.dumps(rec))


def main():
    hip.require_gpu()
    torch.cuda.set_device(0)
    model, _ = B.build_model(dev, 0, 1)
    old = torch.get_default_dtype(); torch.set_default_dtype(torch.bfloat16)
    with torch.device(dev):
        ae, t5, clip = AutoEncoder(AutoEncoderParams(**FLUX_AE)), T5EncoderModel(T5Config()), CLIPTextModel(CLIPTextConfig())
    torch.set_default_dtype(old)
    for m in (ae, t5, clip):
        randomize(m)
    if "--cfg5" in sys.argv:
        return cfg5(model, ae, t5, clip)
    H, Wrow, rows = 384, 1152, 2                                        # 2x3 grid of 384x384 images
    imgs = [torch.rand(3, H, Wrow, device=dev) * 2 - 1 for _ in range(rows)]
    masks = [torch.zeros(1, 1, H, Wrow, device=dev), torch.cat((torch.zeros(1, 1, H, 768, device=dev), torch.ones(1, 1, H, 384, device=dev)), -1)]
    t5_ids = torch.randint(0, 32000, (1, 512), device=dev)
    clip_ids = torch.randint(0, 49000, (1, 77), device=dev); clip_ids[0, 30] = 49407
    rec = {"grid": "2x3 @384", "steps": 30}
    for rep in range(2):                                                # second pass = warm (graphs captured, buffers allocated)
        lat, rec["vae_encode_ms"] = timed(lambda: [ae.encode(i[None].to(torch.bfloat16)) for i in imgs])
        (txt, vec), rec["text_ms"] = timed(lambda: (t5(t5_ids), clip(clip_ids)[0]))
        rng = torch.Generator(device=dev).manual_seed(0)
        noise = [torch.randn([1, 16, H // 8, Wrow // 8], device=dev, generator=rng).to(torch.bfloat16) for _ in imgs]
        out, rec["sampling_loop_ms"] = timed(lambda: pipeline.denoise_grid(model, noise, lat, masks, txt, vec, cfg=30.0, steps=30))
        dec, rec["vae_decode_ms"] = timed(lambda: ae.decode(out[1])[0])
    img = ((dec.float() + 1) / 2).clamp(0, 1)
    assert img.shape == (3, H, Wrow) and torch.isfinite(img).all()
    tot = rec["vae_encode_ms"] + rec["text_ms"] + rec["sampling_loop_ms"] + rec["vae_decode_ms"]
    rec = {k: (round(v, 2) if isinstance(v, float) else v) for k, v in rec.items()}
    rec["total_ms"] = round(tot, 1)
    rec["grids_per_sec"] = round(1e3 / tot, 4)
    rec["loop_share"] = round(rec["sampling_loop_ms"] / tot, 4)
    print(json.dumps(rec))


if __name__ == "__main__":
    main()
