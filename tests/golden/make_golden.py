"""Generate tests/golden/*.npz by running the REFERENCE ITSELF (/root/reference, read-only) on CPU.

Runs only in the build container (the reference never travels to the GPU box; only the vectors do).
The reference's hot path imports under three shims:
  * flash_attn / flash_attn.bert_padding — dense softmax(QK^T d^-1/2)V per cu_seqlens segment and the
    gather/scatter helpers with flash-attn >= 2.7's 5-tuple `unpad_input` (models/math.py:5-6,51);
  * torchdiffeq.odeint — fixed-grid explicit Euler on the given time points (transport/integrators.py:119);
  * torch.cuda.device — nullcontext (layers.py:185,241 enter it even on CPU).
Weights and inputs are procedural (tests/procedural.py), so fixtures hold inputs/outputs only.

    python tests/golden/make_golden.py         # rewrites tests/golden/*.npz
"""
import contextlib
import os
import sys
import types

import numpy as np
import torch

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get("VC_REFERENCE", "/root/reference")
sys.path.insert(0, REPO)


def install_shims():
    import importlib.machinery
    fa = types.ModuleType("flash_attn")
    bp = types.ModuleType("flash_attn.bert_padding")
    fa.__spec__ = importlib.machinery.ModuleSpec("flash_attn", None)      # transformers probes find_spec()
    bp.__spec__ = importlib.machinery.ModuleSpec("flash_attn.bert_padding", None)
    fa.__version__ = "0.0.0"

    def flash_attn_varlen_func(q, k, v, cu_seqlens_q, cu_seqlens_k, max_seqlen_q, max_seqlen_k, dropout_p=0.0,
                               softmax_scale=None, causal=False, **kw):
        assert dropout_p == 0.0 and not causal
        out = torch.empty_like(q)
        scale = softmax_scale if softmax_scale is not None else q.shape[-1] ** -0.5
        for i in range(len(cu_seqlens_q) - 1):
            qs, qe = int(cu_seqlens_q[i]), int(cu_seqlens_q[i + 1])
            ks, ke = int(cu_seqlens_k[i]), int(cu_seqlens_k[i + 1])
            qq, kk, vv = (t.float().transpose(0, 1) for t in (q[qs:qe], k[ks:ke], v[ks:ke]))  # H,L,D
            p = torch.softmax(qq @ kk.transpose(-1, -2) * scale, dim=-1)
            out[qs:qe] = (p @ vv).transpose(0, 1).to(q.dtype)
        return out

    def index_first_axis(x, indices):
        return x[indices]

    def unpad_input(hidden_states, attention_mask, unused_mask=None):
        seqlens = attention_mask.sum(dim=-1, dtype=torch.int32)
        indices = torch.nonzero(attention_mask.flatten(), as_tuple=False).flatten()
        cu = torch.nn.functional.pad(torch.cumsum(seqlens, dim=0, dtype=torch.int32), (1, 0))
        flat = hidden_states.reshape(-1, *hidden_states.shape[2:])
        return flat[indices], indices, cu, int(seqlens.max()), seqlens

    def pad_input(hidden_states, indices, batch, seqlen):
        out = torch.zeros(batch * seqlen, *hidden_states.shape[1:], dtype=hidden_states.dtype)
        out[indices] = hidden_states
        return out.reshape(batch, seqlen, *hidden_states.shape[1:])

    fa.flash_attn_varlen_func = flash_attn_varlen_func
    bp.index_first_axis, bp.unpad_input, bp.pad_input = index_first_axis, unpad_input, pad_input
    fa.bert_padding = bp
    sys.modules["flash_attn"], sys.modules["flash_attn.bert_padding"] = fa, bp

    td = types.ModuleType("torchdiffeq")
    calls = {"t": []}

    def odeint(func, y0, t, method="euler", **kw):
        assert method == "euler"
        ys = [y0]
        for i in range(len(t) - 1):
            dt = t[i + 1] - t[i]                 # 0-dim f32: `dt * f` takes f's dtype (bf16 state -> bf16 update)
            calls["t"].append(float(t[i]))
            # torchdiffeq hands the drift `t.to(y.dtype)` (_PerturbFunc.forward): with the bf16 state of
            # visualcloze.py:399 the model sees 1 - bf16(t_i) while dt still comes from the f32 grid
            ys.append(ys[-1] + dt * func(t[i].to(y0.dtype), ys[-1]))
        return torch.stack(ys)

    td.odeint, td._calls = odeint, calls
    sys.modules["torchdiffeq"] = td
    torch.cuda.device = lambda *a, **k: contextlib.nullcontext()


def main():
    install_shims()
    sys.path.insert(0, REF)
    from models.model import FluxLoraWrapper, FluxParams  # noqa: E402
    from models.modules import layers as RL  # noqa: E402
    from models import math as RM  # noqa: E402
    from models.modules.lora import LinearLora  # noqa: E402
    from transport import Sampler, create_transport  # noqa: E402
    import torchdiffeq  # noqa: E402

    from tests.procedural import TINY, TINY_RANK, procedural_param, ptensor, tiny_inputs

    torch.manual_seed(0)
    out = {}

    # ---------------- model with procedural weights ----------------
    model = FluxLoraWrapper(lora_rank=TINY_RANK, lora_scale=1.0, params=FluxParams(**TINY)).float().eval()
    key_shapes = [(k, tuple(v.shape)) for k, v in model.state_dict().items()]
    model.load_state_dict({k: procedural_param(k, s) for k, s in key_shapes}, strict=True)
    out["keys"] = np.array([k for k, _ in key_shapes])
    out["shapes"] = np.array([",".join(map(str, s)) for _, s in key_shapes])

    with torch.no_grad():
        # ---------------- leaf ops ----------------
        inp = tiny_inputs(B=1)
        ids = torch.cat((inp["txt_ids"], inp["img_ids"]), dim=1)
        pe = model.pe_embedder(ids)                       # [B,1,L,64,2,2]
        out["pe_ids"], out["pe"] = ids.numpy(), pe.numpy()
        q = ptensor((1, 2, ids.shape[1], 128), 11, q=6)
        k = ptensor((1, 2, ids.shape[1], 128), 12, q=6)
        rq, rk = RM.apply_rope(q, k, pe)
        out["rope_q_in"], out["rope_q_out"], out["rope_k_out"] = q.numpy(), rq.numpy(), rk.numpy()
        tt = torch.tensor([0.0, 0.348, 1.0])
        out["temb_t"], out["temb"] = tt.numpy(), RL.timestep_embedding(tt, 256).numpy()
        out["temb_g30"] = RL.timestep_embedding(torch.tensor([30.0]), 256).numpy()
        blk0 = model.double_blocks[0]
        v = ptensor((1, 2, ids.shape[1], 128), 13, q=6)
        nq, nk = blk0.img_attn.norm(q, k, v)
        out["qknorm_q"], out["qknorm_k"] = nq.numpy(), nk.numpy()
        vec = ptensor((1, 256), 14, q=6)
        m1, m2 = blk0.img_mod(vec)
        out["mod_vec"] = vec.numpy()
        out["mod_out"] = torch.cat([m1.shift, m1.scale, m1.gate, m2.shift, m2.scale, m2.gate], dim=-1).numpy()
        # attention: all-ones mask and a ragged (prefix) mask, B=2
        L = ids.shape[1]
        q2, k2, v2 = (ptensor((2, 2, L, 128), s, q=6) for s in (21, 22, 23))
        pe2 = pe.repeat(2, 1, 1, 1, 1, 1)
        full = torch.ones(2, L, dtype=torch.int32)
        ragged = full.clone()
        ragged[1, L - 9:] = 0
        out["attn_full"] = RM.attention(q2, k2, v2, pe=pe2, attn_mask=full).numpy()
        out["attn_ragged"] = RM.attention(q2, k2, v2, pe=pe2, attn_mask=ragged).numpy()
        out["attn_ragged_mask"] = ragged.numpy()
        general = full.clone()            # holes anywhere (math.py:9-60 gathers arbitrary masks)
        general[0, [3, 4, 17, 39]] = 0
        general[1, :2] = 0
        general[1, 20:29] = 0
        out["attn_general"] = RM.attention(q2, k2, v2, pe=pe2, attn_mask=general).numpy()
        out["attn_general_mask"] = general.numpy()
        # LinearLora with rank clipped to min(in, out) = 4
        ll = LinearLora(in_features=12, out_features=4, bias=torch.zeros(4), rank=8, dtype=torch.float32,
                        device=torch.device("cpu"), scale=0.5)
        ll.load_state_dict({kk: procedural_param("lltest." + kk, vv.shape) for kk, vv in ll.state_dict().items()})
        xin = ptensor((3, 12), 31, q=5)
        out["lora_clip_shapes"] = np.array([",".join(map(str, vv.shape)) for vv in ll.state_dict().values()])
        out["lora_clip_keys"] = np.array(list(ll.state_dict().keys()))
        out["lora_clip_in"], out["lora_clip_out"] = xin.numpy(), ll(xin).numpy()

        # ---------------- blocks ----------------
        img_h = ptensor((1, inp["x"].shape[1], 256), 41, q=6)
        txt_h = ptensor((1, inp["txt"].shape[1], 256), 42, q=6)
        im = torch.ones(1, img_h.shape[1], dtype=torch.int32)
        tm = torch.ones(1, txt_h.shape[1], dtype=torch.int32)
        di, dtx = blk0(img=img_h, txt=txt_h, vec=vec, pe=pe, img_mask=im, txt_mask=tm)
        out["blk_img_in"], out["blk_txt_in"] = img_h.numpy(), txt_h.numpy()
        out["double0_img"], out["double0_txt"] = di.numpy(), dtx.numpy()
        xs = torch.cat((txt_h, img_h), 1)
        out["single0"] = model.single_blocks[0](xs, vec=vec, pe=pe, attn_mask=torch.cat((tm, im), 1)).numpy()
        out["last"] = model.final_layer(img_h, vec).numpy()

        # ---------------- full forward, B=1 and ragged B=2 ----------------
        def fwd(i, t):
            return model(torch.cat((i["x"], i["cond"]), -1), timesteps=t, txt=i["txt"], txt_ids=i["txt_ids"],
                         txt_mask=i["txt_mask"], y=i["y"], img_ids=i["img_ids"], img_mask=i["img_mask"],
                         guidance=i["guidance"])
        out["flux_b1_t"] = np.array([0.7], dtype=np.float32)
        out["flux_b1"] = fwd(inp, torch.tensor([0.7])).numpy()
        inp2 = tiny_inputs(B=2, seed=7)
        inp2["img_mask"][1, -12:] = 0      # second sample is a shorter grid, padded (sampling.py:68-70)
        out["flux_b2_t"] = np.array([0.9, 0.25], dtype=np.float32)
        out["flux_b2"] = fwd(inp2, torch.tensor([0.9, 0.25])).numpy()
        inp3 = tiny_inputs(B=2, seed=7)   # holes in BOTH streams of sample 1, text padding on sample 0
        inp3["txt_mask"][0, -5:] = 0
        inp3["txt_mask"][1, [1, 7, 8]] = 0
        inp3["img_mask"][1, [0, 5, 6, 13, 23]] = 0
        out["flux_general_txt_mask"], out["flux_general_img_mask"] = inp3["txt_mask"].numpy(), inp3["img_mask"].numpy()
        out["flux_general"] = fwd(inp3, torch.tensor([0.9, 0.25])).numpy()

        inp4 = tiny_inputs(B=2, seed=13)  # sample 1 has NO text at all (txt_mask all zeros): keys 0 .. T-1 masked
        inp4["txt_mask"][1] = 0
        inp4["img_mask"][0, [0, 3]] = 0
        out["flux_notext_txt_mask"], out["flux_notext_img_mask"] = inp4["txt_mask"].numpy(), inp4["img_mask"].numpy()
        out["flux_notext"] = fwd(inp4, torch.tensor([0.8, 0.3])).numpy()

        # ---------------- sampler: time grids + trajectories ----------------
        sampler = Sampler(create_transport("Linear", "velocity", do_shift=True))
        for n_tok in (1152, 3456, 6144, 6912):
            for steps in (4, 30, 50):
                torchdiffeq._calls["t"].clear()
                fn = sampler.sample_ode(sampling_method="euler", num_steps=steps, atol=1e-6, rtol=1e-3, reverse=False,
                                        do_shift=True, time_shifting_factor=1)
                seen = []
                fn(torch.zeros(1, n_tok, 2), lambda x, timesteps, **kw: (seen.append(float(timesteps[0])), x * 0)[1], {})
                out[f"grid_{n_tok}_{steps}_solver_t"] = np.array(torchdiffeq._calls["t"], dtype=np.float64)
                out[f"grid_{n_tok}_{steps}_model_t"] = np.array(seen, dtype=np.float64)
        torchdiffeq._calls["t"].clear()
        fn = sampler.sample_ode(sampling_method="euler", num_steps=10, atol=1e-6, rtol=1e-3, reverse=False,
                                do_shift=False, time_shifting_factor=1.0, strength=0.4)
        seen = []
        fn(torch.zeros(1, 4096, 2), lambda x, timesteps, **kw: (seen.append(float(timesteps[0])), x * 0)[1], {})
        out["grid_upsample_model_t"] = np.array(seen, dtype=np.float64)

        # full tiny trajectory through the reference sampler + reference model (5 points -> 4 evals)
        fn = sampler.sample_ode(sampling_method="euler", num_steps=5, atol=1e-6, rtol=1e-3, reverse=False,
                                do_shift=True, time_shifting_factor=1)
        kw = dict(txt=inp["txt"], txt_ids=inp["txt_ids"], txt_mask=inp["txt_mask"], y=inp["y"], img_ids=inp["img_ids"],
                  img_mask=inp["img_mask"], cond=inp["cond"], guidance=inp["guidance"])
        traj = fn(inp["x"], model.forward, kw)
        out["traj_states"] = traj.numpy()
        # SDEdit-style: strength grid, no shift (visualcloze.py:184-193)
        fn = sampler.sample_ode(sampling_method="euler", num_steps=4, atol=1e-6, rtol=1e-3, reverse=False,
                                do_shift=False, time_shifting_factor=1.0, strength=0.4)
        out["traj_sdedit_last"] = fn(inp["x"], model.forward, kw)[-1].numpy()

        # bf16 noise floor of the reference itself (autocast on CPU), for tolerance statements
        mb = FluxLoraWrapper(lora_rank=TINY_RANK, lora_scale=1.0, params=FluxParams(**TINY)).eval()
        mb.load_state_dict({k: procedural_param(k, s) for k, s in key_shapes})
        mb = mb.to(torch.bfloat16)
        with torch.autocast("cpu", torch.bfloat16):
            yb = mb(torch.cat((inp["x"], inp["cond"]), -1).bfloat16(), timesteps=torch.tensor([0.7]),
                    txt=inp["txt"].bfloat16(), txt_ids=inp["txt_ids"], txt_mask=inp["txt_mask"], y=inp["y"].bfloat16(),
                    img_ids=inp["img_ids"], img_mask=inp["img_mask"], guidance=inp["guidance"].bfloat16())
        out["flux_b1_ref_bf16"] = yb.float().numpy()
        # the same sampler with the bf16 state / bf16 model of the pipeline (visualcloze.py:363,399): pins the
        # timestep the model sees (1 - bf16(t_i)) and the bf16 Euler update
        fn = sampler.sample_ode(sampling_method="euler", num_steps=5, atol=1e-6, rtol=1e-3, reverse=False,
                                do_shift=True, time_shifting_factor=1)
        seen = []
        kwb = dict(txt=inp["txt"].bfloat16(), txt_ids=inp["txt_ids"], txt_mask=inp["txt_mask"], y=inp["y"].bfloat16(),
                   img_ids=inp["img_ids"], img_mask=inp["img_mask"], cond=inp["cond"].bfloat16(),
                   guidance=inp["guidance"].bfloat16())

        def mb_fwd(x, timesteps, **k):
            seen.append(float(timesteps[0]))
            assert timesteps.dtype == torch.float32
            return mb.forward(x, timesteps=timesteps, **k)
        with torch.autocast("cpu", torch.bfloat16):
            trajb = fn(inp["x"].bfloat16(), mb_fwd, kwb)
        assert trajb.dtype == torch.bfloat16
        out["traj_bf16_model_t"] = np.array(seen, dtype=np.float64)
        out["traj_bf16_states"] = trajb.float().numpy()
        # an F32 state through the same bf16 model (a caller that does not cast its noise): the solver keeps the state's
        # dtype (integrators.py:119), the model sees 1 - t_i unrounded, only dt * f is bf16
        seen = []
        with torch.autocast("cpu", torch.bfloat16):
            trajf = fn(inp["x"].float(), mb_fwd, kwb)
        assert trajf.dtype == torch.float32
        out["traj_f32state_model_t"] = np.array(seen, dtype=np.float64)
        out["traj_f32state_states"] = trajf.numpy()

    # ---------------- latent-grid packer: the tensor part of prepare_modified (models/sampling.py:37-118) -------------
    # models.sampling imports cv2 / models.util (imwatermark) through image_embedders: stub those two modules.
    import importlib.machinery
    cv2 = types.ModuleType("cv2")
    cv2.__spec__ = importlib.machinery.ModuleSpec("cv2", None)
    sys.modules["cv2"] = cv2
    mu = types.ModuleType("models.util")
    mu.print_load_warning = lambda *a, **k: None
    sys.modules["models.util"] = mu
    from models.sampling import prepare_modified  # noqa: E402
    from einops import rearrange  # noqa: E402
    rows_a = [ptensor((1, 16, 4, 12), 61, q=5), ptensor((1, 16, 6, 8), 62, q=5)]     # two rows, different sizes
    rows_b = [ptensor((1, 16, 4, 12), 63, q=5), ptensor((1, 16, 2, 8), 64, q=5)]     # shorter second sample -> padded
    emb = [dict(txt=ptensor((16, 128), 65, q=6), vec=ptensor((64,), 66, q=6))] * 2
    pk = prepare_modified(t5=None, clip=None, img=[rows_a, rows_b], prompt=["a", "b"], proportion_empty_prompts=0.0,
                          text_emb=emb)
    for i, r in enumerate(rows_a + rows_b):
        out[f"pack_row{i}"] = r.numpy()
    for k in ("img", "img_ids", "img_mask", "txt_ids", "txt_mask"):
        out["pack_" + k] = pk[k].float().numpy()
    # fill-mask packing of the pipeline (einops patterns of visualcloze.py:381-382): pixel mask -> [N, 256]
    pm = (ptensor((1, 1, 32, 96), 67, q=0, kmax=1).abs() > 0.5).float()
    m8 = rearrange(pm, "b c (h ph) (w pw) -> b (c ph pw) h w", ph=8, pw=8)
    out["maskpack_in"] = pm.numpy()
    out["maskpack_out"] = rearrange(m8, "b c (h ph) (w pw) -> b (h w) (c ph pw)", ph=2, pw=2).numpy()
    # row-wise unpack of the result (visualcloze.py:425-429)
    tok = ptensor((1, 24, 64), 68, q=5)
    out["unpack_in"] = tok.numpy()
    out["unpack_out"] = rearrange(tok, "b (h w) (c ph pw) -> b c (h ph) (w pw)", ph=2, pw=2, h=2, w=12).numpy()

    path = os.path.join(HERE, "tiny_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes;", len(out), "arrays")
    f32, b16 = out["flux_b1"], out["flux_b1_ref_bf16"]
    print("reference bf16-vs-fp32 on flux_b1: rel-L2 %.3e max-abs %.3e" % (
        np.linalg.norm(f32 - b16) / np.linalg.norm(f32), np.abs(f32 - b16).max()))


if __name__ == "__main__":
    main()
