"""Host-side mirror of the reference's sampler interface (transport/__init__.py:4-62,
transport/transport.py:236-410, transport/integrators.py:79-120, transport/utils.py:33-44):

    sampler = Sampler(create_transport("Linear", "velocity", do_shift=True))
    sample_fn = sampler.sample_ode(sampling_method="euler", num_steps=30, ...)
    latents = sample_fn(x, model.forward, model_kwargs)[-1]

Only the configuration the inference pipeline uses is implemented (Linear path, velocity prediction,
fixed-grid Euler); anything else raises NotImplementedError.  When `model` is the bound `forward` of a
`visualcloze_amd.Flux`, the whole loop runs as hipGraph replays of one captured evaluation + Euler update
with zero host synchronisation inside the loop; a foreign callable is stepped eagerly with the same grid.
"""
from __future__ import annotations

import math
from typing import Callable, Optional

import torch

from . import hip


def time_shift(mu: float, sigma: float, t: torch.Tensor) -> torch.Tensor:
    """transport/utils.py:33-39 (endpoints 0 -> 0 and 1 -> 1 through inf arithmetic, as the reference)."""
    t = 1 - t
    t = math.exp(mu) / (math.exp(mu) + (1 / t - 1) ** sigma)
    return 1 - t


def get_lin_function(x1: float = 256, y1: float = 0.5, x2: float = 4096, y2: float = 1.15) -> Callable:
    m = (y2 - y1) / (x2 - x1)
    b = y1 - m * x1
    return lambda x: m * x + b


def solver_time_grid(num_steps: int, n_tokens: int, t0: float, t1: float, do_shift: bool,
                     time_shifting_factor: Optional[float]) -> torch.Tensor:
    """ode.__init__ + ode.sample (integrators.py:99-101,113-116): the solver's time POINTS (f32, CPU)."""
    assert t0 < t1, "ODE sampler has to be in forward time"
    t = torch.linspace(t0, t1, num_steps)
    if time_shifting_factor:
        t = t / (t + time_shifting_factor - time_shifting_factor * t)
    if do_shift:
        mu = get_lin_function(y1=0.5, y2=1.15)(n_tokens)
        t = time_shift(mu, 1.0, t)
    return t


class Transport:
    def __init__(self, path_type: str, prediction: str, do_shift: bool):
        self.path_type, self.prediction, self.do_shift = path_type, prediction, do_shift
        self.train_eps = self.sample_eps = 0  # velocity & Linear is stable everywhere (transport/__init__.py:46-48)

    def check_interval(self, reverse: bool = False):
        t0, t1 = 0, 1  # transport.py:70-96 for Linear + velocity, sde=False
        if reverse:
            t0, t1 = 1 - t0, 1 - t1
        return t0, t1


def create_transport(path_type="Linear", prediction="velocity", loss_weight=None, train_eps=None, sample_eps=None,
                     snr_type="uniform", loss_type="mse", do_shift=True) -> Transport:
    if path_type != "Linear" or prediction != "velocity":
        raise NotImplementedError("the MI355X denoising path implements Linear path + velocity prediction only")
    return Transport(path_type, prediction, do_shift)


class Sampler:
    def __init__(self, transport: Transport):
        self.transport = transport

    def sample_ode(self, *, sampling_method="dopri5", num_steps=50, atol=1e-6, rtol=1e-3, reverse=False,
                   do_shift=True, time_shifting_factor=None, strength=None, return_trajectory: bool = False):
        if sampling_method != "euler":
            raise NotImplementedError("only the fixed-grid 'euler' solver (the inference default) is implemented")
        t0, t1 = self.transport.check_interval(reverse=reverse)
        if strength is not None:
            t0 = (t1 - t0) * strength + t0
        assert t0 < t1, "ODE sampler has to be in forward time"

        def _sample(x: torch.Tensor, model: Callable, model_kwargs: dict) -> torch.Tensor:
            t = solver_time_grid(num_steps, x.shape[1], t0, t1, do_shift, time_shifting_factor)
            from .model import Flux
            owner = getattr(model, "__self__", None)
            if isinstance(owner, Flux) and getattr(model, "__name__", "") == "forward":
                if _fusable(owner, x):
                    return _sample_fused(owner, x, dict(model_kwargs), t, return_trajectory)
                # stepped eagerly; the velocity of this model is a bf16 tensor as the reference's is under autocast
                # (visualcloze.py:363) whatever dtype Flux.forward hands back to its caller: dt * f stays a bf16 product
                fwd = model
                model = lambda xin, **k: fwd(xin, **k).to(torch.bfloat16)  # noqa: E731
            return _sample_foreign(model, x, dict(model_kwargs), t, return_trajectory)

        return _sample


def _fusable(flux, x: torch.Tensor) -> bool:
    """The fused loop steps a bf16 state (the pipeline's, visualcloze.py:399) or - through the C handle - an f32 one IN f32
    (integrators.py:119 keeps the caller's state dtype).  Anything else (f16 / f64 states, an f32 state in the un-merged LoRA
    parity mode whose plan is ordered from Python) is stepped eagerly through Flux.forward with torch's own promotion rules:
    never a silent per-step rounding of the caller's state."""
    if x.dtype == torch.bfloat16:
        return True
    return x.dtype == torch.float32 and flux.handle() is not None


def _solver_t_as_state(t: torch.Tensor, x: torch.Tensor) -> torch.Tensor:
    """torchdiffeq hands the drift `t.to(y.dtype)` (`_PerturbFunc.forward`): with the bf16 state of
    visualcloze.py:399 the time a model sees is bf16(t_i), while dt = t_{i+1} - t_i comes from the f32 grid."""
    return t.to(x.dtype).to(torch.float32) if x.dtype.is_floating_point else t.to(torch.float32)


def model_times(t: torch.Tensor, x: torch.Tensor) -> torch.Tensor:
    """The `timesteps` of the S = len(t) - 1 Flux evaluations (f32): 1 - t_i with t_i rounded to the state dtype
    (integrators.py:108-109 builds `ones(B) * t` in f32 from it; transport.py:384 forms 1 - t)."""
    t32 = t.to(torch.float32)
    return torch.ones(len(t) - 1) * (1 - _solver_t_as_state(t32[:-1], x))


def _sample_foreign(model, x, kw, t, return_trajectory):
    """Any other callable: the same grid and update rule, one host-driven call per interval."""
    cond = kw.pop("cond", None)
    states = [x]
    for i in range(len(t) - 1):
        tt = torch.ones(x.size(0), device=x.device) * _solver_t_as_state(t[i], x).to(x.device)
        xin = torch.cat((x, cond), dim=-1) if cond is not None else x
        v = model(xin, timesteps=torch.ones_like(tt) * (1 - tt), **kw)
        assert v.shape == x.shape, "Output shape from ODE solver must match input shape"
        x = x + (t[i + 1] - t[i]).to(x.device) * (-v)
        if return_trajectory:
            states.append(x)
    return torch.stack(states) if return_trajectory else x[None]


@torch.no_grad()
def _sample_fused(flux, x, kw, t, return_trajectory):
    eng = flux.engine()
    dev = eng.dev
    B, N, C = x.shape
    cond = kw.get("cond")
    if cond is None:
        raise hip.VclozeHipError("fused sampler expects model_kwargs['cond'] (x || cond feeds img_in)")
    if C + cond.shape[-1] != flux.in_channels:
        raise hip.VclozeHipError(f"x ({C}) || cond ({cond.shape[-1]}) does not match in_channels {flux.in_channels}")
    txt, y, guidance = kw["txt"], kw["y"], kw.get("guidance")
    if flux.params.guidance_embed and guidance is None:
        raise ValueError("Didn't get guidance strength for guidance distilled model.")
    T = txt.shape[1]
    S = len(t) - 1
    t32 = t.to(torch.float32)
    eval_t = model_times(t32, x)                   # Flux sees 1 - t (transport.py:384), t in the state's dtype
    dts = (t32[1:] - t32[:-1]).contiguous()        # torchdiffeq fixed grid: dt = t1 - t0
    bf = lambda a: a.to(dev, torch.bfloat16).contiguous()  # noqa: E731
    sdt = x.dtype                                  # bf16, or f32 (handle path only): the state is stepped in ITS dtype
    out = torch.empty(B, N, C, dtype=sdt, device=dev)
    traj = []
    gbf16 = guidance is not None and guidance.dtype == torch.bfloat16
    from .model import MaskLayout, per_sample
    guidance = per_sample(guidance, B)
    lay = MaskLayout(kw.get("txt_mask"), kw.get("img_mask"), B, T, N)
    st = eng.stream
    st.wait_stream(torch.cuda.current_stream())
    # the C handle runs the whole trajectory of a chunk in ONE call (vc_flux_sample_euler); the un-merged LoRA mode uses the
    # Python-ordered plan (bf16 states only: _fusable)
    h = flux.handle()
    with torch.cuda.stream(st):
        s = st.cuda_stream
        for b0 in range(0, B, eng.MAX_BATCH):        # a chunk of samples advances together, one graph replay per step
            bs = min(eng.MAX_BATCH, B - b0)
            sl = slice(b0, b0 + bs)
            if h is not None:
                h.prepare(bf(lay.txt_rows(txt, sl)), bf(y[sl]), None if guidance is None else guidance[sl], gbf16,
                          lay.img_rows(kw["img_ids"], sl), lay.txt_rows(kw["txt_ids"], sl), S, lay.kv_len(sl), lay.kv_gap(sl), stream=s)
                xs = lay.img_rows(x, sl).to(dev, sdt, copy=True).contiguous()   # updated in place: never the caller's
                tj = torch.empty(S, bs, N, C, dtype=sdt, device=dev) if return_trajectory else None
                h.sample_euler(xs, bf(lay.img_rows(cond, sl)), t32, x.dtype == torch.bfloat16, s, trajectory=tj)
                if return_trajectory:
                    traj.append(torch.stack([lay.img_rows_back(tj[i], sl) for i in range(S)]))
                out[sl].copy_(lay.img_rows_back(xs, sl))
                continue
            ws = eng.workspace(T, N, S, bs)
            eng.prepare_sample(ws, bf(lay.txt_rows(txt, sl)), bf(y[sl]), None if guidance is None else guidance[sl], gbf16,
                               lay.img_rows(kw["img_ids"], sl), lay.txt_rows(kw["txt_ids"], sl), eval_t, lay.kv_len(sl), s=s,
                               kv_gap=lay.kv_gap(sl))
            ws.DTS.copy_(dts, non_blocking=True)
            ws.STEP.zero_()
            ws.XS.copy_(bf(lay.img_rows(x, sl)).reshape(bs * N, C))     # the state stays in kernel row order for all steps
            ws.COND.copy_(bf(lay.img_rows(cond, sl)).reshape(bs * N, -1))
            graph = eng.step_graph(ws, s)
            states = []
            for _ in range(S):
                graph.launch(s)          # one Flux evaluation + Euler update + step counter increment
                if return_trajectory:
                    states.append(lay.img_rows_back(ws.XS.reshape(bs, N, C), sl).clone())
            if return_trajectory:
                traj.append(torch.stack(states))                  # [S, bs, N, C]
            out[sl].copy_(lay.img_rows_back(ws.XS.reshape(bs, N, C), sl))
    torch.cuda.current_stream().wait_stream(st)
    if return_trajectory:
        return torch.cat((x.to(dev)[None], torch.cat(traj, dim=1).to(sdt)), dim=0)
    return out[None]
