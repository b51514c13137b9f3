"""T5-XXL encoder (512 tokens) and CLIP-L text (77 tokens) timing on one MI355X: full geometry,
random weights drawn on the GPU (timing only; parity is tests/test_text_gpu.py), HIP events.   python tools/text_bench.py"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from visualcloze_amd import hip
from visualcloze_amd.text import CLIPTextConfig, CLIPTextModel, T5Config, T5EncoderModel

dev = "cuda:0"


def randomize(m):
    g = torch.Generator(device=dev).manual_seed(7)
    with torch.no_grad():
        for n, p in m.named_parameters():
            if "layer_norm" in n and n.endswith("weight"):
                p.fill_(1.0)
            elif n.endswith("bias"):
                p.zero_()
            else:
                p.normal_(0.0, p.shape[-1] ** -0.5, generator=g)


def timeit(fn, n=5):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def main():
    hip.require_gpu()
    rec = {"device": torch.cuda.get_device_name(0), "dtype": "bf16", "weights": "random (timing only)"}
    old = torch.get_default_dtype(); torch.set_default_dtype(torch.bfloat16)
    with torch.device(dev):
        t5 = T5EncoderModel(T5Config())
        clip = CLIPTextModel(CLIPTextConfig())
    torch.set_default_dtype(old)
    randomize(t5); randomize(clip)
    c = t5.cfg
    L = 512
    ids = torch.randint(0, c.vocab_size, (1, L), device=dev)
    ms = timeit(lambda: t5(ids))
    inner = c.num_heads * c.d_kv
    fl = c.num_layers * (2.0 * L * c.d_model * inner * 4 + 2.0 * L * c.d_model * c.d_ff * 3 + 4.0 * L * L * inner)
    rec["t5_xxl_512"] = {"ms": round(ms, 3), "gflop": round(fl / 1e9, 1), "tflops": round(fl / ms / 1e9, 1),
                         "params_b": round(sum(p.numel() for p in t5.parameters()) / 1e9, 2)}
    cc = clip.cfg
    cids = torch.randint(0, cc.vocab_size - 1, (1, 77), device=dev); cids[0, 20] = cc.eos_token_id
    ms = timeit(lambda: clip(cids))
    rec["clip_l_77"] = {"ms": round(ms, 3)}
    print(json.dumps(rec))


if __name__ == "__main__":
    main()
