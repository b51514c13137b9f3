#!/usr/bin/env python
"""bench.py — denoising-steps/s of the MI355X-native VisualCloze sampling loop (BASELINE.json metric).

A "step" is one solver step of the hot path for one grid: a full Flux evaluation (19 double + 38 single
blocks, L = 512 text + 3456 image tokens for the 384-grid 2x3 layout = BASELINE configs[1]) plus the Euler
update, replayed as one hipGraph.  Every rank (one process per GPU) runs its own independent grid: weak
scaling, no collective inside the step; the frozen weights are broadcast once over RCCL before timing.
The timed region OPENS on a sample boundary, so the per-sample precomputation (txt_in, vec path, all 29x1.06M
modulation rows, RoPE table; reported separately as `precompute_ms`) is inside it; inputs are resident in HBM.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` (dominant kernel = the bf16
MFMA GEMM with the gate/residual epilogue, timed live with HIP events on its launch stream) and
`cpu_baseline` (the CPU oracle — a "port" — timed on a bounded sample on the host cores).
"""
import argparse
import json
import os
import sys
import time

import torch

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

WORKLOADS = {
    # name: (resolution, rows, cols) -> per-row latent (h, w) = (res/8, cols*res/8)
    "384-grid-2x3": dict(rows=2, row_latent=(48, 144), steps=30),
    "512-grid-2x3": dict(rows=2, row_latent=(64, 192), steps=30),
    "384-grid-1x2": dict(rows=1, row_latent=(48, 96), steps=4),
    "384-grid-3x4": dict(rows=3, row_latent=(48, 192), steps=50),
    # the largest grid the reference's own UI offers (app.py:10-11: up to 5 in-context rows x 5 columns): L = 512 + 25 * 576 = 14912
    "384-grid-5x5": dict(rows=5, row_latent=(48, 240), steps=30),
    # cfg 5's SDEdit upsample stage of one 1024x1024 target (visualcloze.py:184-234): 10 points from strength 0.4, no shift
    "1024-sdedit-upsample": dict(rows=1, row_latent=(128, 128), steps=10, t0=0.4, do_shift=False),
    # shapes the pipeline really produces from non-square photographs (resize_with_aspect_ratio, visualcloze.py:28-60: area
    # ~384^2, both sides floored to multiples of 16, every row at the aspect of ITS first image, :312-323): 3:4 portraits are
    # 320x432 px = 54x40 latent = 540 tokens, a 2x3 grid of them N = 3240, L = 3752 - off every 64 / 128 / 256 tile edge;
    # "mixed": portrait first row, 4:3 landscape (432x320 px) second row - rows of different width in one sequence
    "384-grid-2x3-p34": dict(row_latents=[(54, 120), (54, 120)], steps=30),
    "384-grid-2x3-mixed": dict(row_latents=[(54, 120), (40, 162)], steps=30),
}
for _w in WORKLOADS.values():
    _w.setdefault("row_latents", [_w["row_latent"]] * _w["rows"] if "row_latent" in _w else None)
    _w.setdefault("rows", len(_w["row_latents"]))
MFMA_BF16_PEAK_TFLOPS = 2500.0   # dense, MI355X_MICROARCH.md
HBM_PEAK_GBS = 8000.0             # HBM3E, MI355X_MICROARCH.md


def grid_img_ids(rows, h=None, w=None):
    """models/sampling.py:56-59: axis0 = row index + 1, axis1 = y, axis2 = x (per concatenated row).  `rows`: a count of
    equal rows of latent size (h, w), or a list of per-row latent sizes."""
    out = []
    for j, (h, w) in enumerate([(h, w)] * rows if isinstance(rows, int) else rows):
        ids = torch.zeros(h // 2, w // 2, 3)
        ids[..., 0] = j + 1
        ids[..., 1] = torch.arange(h // 2)[:, None]
        ids[..., 2] = torch.arange(w // 2)[None, :]
        out.append(ids.reshape(-1, 3))
    return torch.cat(out, 0)


def build_model(dev, rank, world, lora_rank=256, params=None):
    """Random-init FLUX.1-Fill-dev + LoRA on `dev`.  The module is constructed on the META device and materialised with
    to_empty(): no rank runs nn.Linear's default initialisation of 13 B parameters only to overwrite it - rank 0 fills every
    parameter once (seeded), the other ranks receive them through the one-time broadcast."""
    from visualcloze_amd.model import FLUX_DEV_FILL, FluxLoraWrapper, FluxParams
    old = torch.get_default_dtype()
    torch.set_default_dtype(torch.bfloat16)
    try:
        with torch.device("meta"):
            model = FluxLoraWrapper(lora_rank=lora_rank, lora_scale=1.0, params=FluxParams(**(params or FLUX_DEV_FILL)))
    finally:
        torch.set_default_dtype(old)
    assert not list(model.buffers()), "to_empty() would leave buffers uninitialised"
    model.to_empty(device=dev)
    model.eval()
    g = torch.Generator(device=dev).manual_seed(1234)
    with torch.no_grad():
        for name, p in model.named_parameters():
            if rank != 0:
                p.zero_()                     # (defined contents; overwritten by the broadcast below)
                continue
            if name.endswith("norm.scale"):
                p.fill_(1.0)
            elif name.endswith(".bias"):
                p.zero_()
            else:                             # matrices incl. LoRA A/B: N(0, 0.02) so the LoRA path is live
                p.normal_(0.0, 0.02, generator=g)
    from visualcloze_amd import parallel as par
    torch.cuda.synchronize()
    bcast_s = par.broadcast_weights(model, src=0)   # one-time RCCL broadcast of the frozen weights over xGMI
    return model, bcast_s


def make_inputs(dev, wl, seed, B=1, ctx_dim=4096, vec_dim=768):
    ids = grid_img_ids(wl["row_latents"])
    N = ids.shape[0]
    g = torch.Generator(device="cpu").manual_seed(seed)
    x = torch.randn(B, N, 64, generator=g)
    cond = torch.randn(B, N, 320, generator=g)
    mask = torch.zeros(N)
    last = (wl["row_latents"][-1][0] // 2) * (wl["row_latents"][-1][1] // 2)
    mask[N - last + last * 2 // 3:] = 1                               # last cell(s) of the last row masked
    cond[..., 64:] = mask[None, :, None]
    kw = dict(txt=torch.randn(B, 512, ctx_dim, generator=g).to(dev, torch.bfloat16), txt_ids=torch.zeros(B, 512, 3, device=dev),
              txt_mask=torch.ones(B, 512, dtype=torch.int32, device=dev),
              y=torch.randn(B, vec_dim, generator=g).to(dev, torch.bfloat16), img_ids=ids[None].repeat(B, 1, 1).to(dev),
              img_mask=torch.ones(B, N, dtype=torch.int32, device=dev), cond=cond.to(dev, torch.bfloat16),
              guidance=torch.full((B,), 30.0, device=dev, dtype=torch.bfloat16))
    return x.to(dev, torch.bfloat16), kw


class Job:
    """Drives the product path exactly as transport._sample_fused does - the C handle API (vc_flux_prepare /
    vc_flux_sample_begin / vc_flux_sample_steps) - but one solver step per call, as the timing contract counts steps."""

    def __init__(self, model, x, kw, num_points, t0=0.0, do_shift=True):
        from visualcloze_amd.transport import solver_time_grid
        self.model, self.eng, self.h = model, model.engine(), model.handle()     # h None: --python-plan (A/B runs)
        self.x, self.kw = x.contiguous(), kw
        self.N, self.T = x.shape[1], kw["txt"].shape[1]
        self.t = solver_time_grid(num_points, self.N, t0, 1, do_shift, 1)
        self.S = num_points - 1
        self.s = self.eng.stream.cuda_stream
        self.step_in_sample = self.S   # forces a prepare on the first step
        self._ws = None

    def begin_sample(self):
        kw = self.kw
        if self.h is None:               # the same plan ordered from Python (engine.FluxEngine), for A/B runs
            from visualcloze_amd.transport import model_times
            eng, B = self.eng, self.x.shape[0]
            ws = self._pyws = eng.workspace(self.T, self.N, self.S, B)
            eng.prepare_sample(ws, kw["txt"], kw["y"], kw["guidance"], True, kw["img_ids"], kw["txt_ids"],
                               model_times(self.t, self.x), [ws.L] * B, s=self.s)
            ws.DTS.copy_((self.t[1:] - self.t[:-1]).contiguous(), non_blocking=True)
            ws.STEP.zero_()
            ws.XS.copy_(self.x.reshape(B * ws.N, -1))
            ws.COND.copy_(kw["cond"].reshape(B * ws.N, -1))
            self.graph = eng.step_graph(ws, self.s)
        else:
            self.h.prepare(kw["txt"], kw["y"], kw["guidance"], True, kw["img_ids"], kw["txt_ids"], self.S, stream=self.s)
            self.h.sample_begin(self.x, kw["cond"], self.t, True, self.s)
        self.step_in_sample = 0

    def restart_sample(self):
        self.step_in_sample = self.S

    def step(self):
        if self.step_in_sample >= self.S:
            self.begin_sample()
        if self.h is None:
            self.graph.launch(self.s)
        else:
            self.h.sample_steps(1, self.s)
        self.step_in_sample += 1

    def state(self):
        if self.h is None:
            return self._pyws.XS.reshape(self.x.shape).clone()
        out = torch.empty_like(self.x)
        self.h.sample_end(out, self.s)
        return out

    @property
    def ws(self):
        """A prepared workspace of the Python-ordered engine: operand memory (activations, gates) for the roofline legs,
        which launch single kernels through the op-level ABI."""
        if self._ws is None:
            from visualcloze_amd.transport import model_times
            kw, B = self.kw, self.x.shape[0]
            ws = self.eng.workspace(self.T, self.N, self.S, B)
            self.eng.prepare_sample(ws, kw["txt"], kw["y"], kw["guidance"], True, kw["img_ids"], kw["txt_ids"],
                                    model_times(self.t, self.x), [ws.L] * B, s=self.s)
            ws.XS.copy_(self.x.reshape(B * ws.N, -1))
            ws.COND.copy_(kw["cond"].reshape(B * ws.N, -1))
            self.eng.eval_once(ws, ws.STEP, euler=False, s=self.s)      # fills the activations the kernels are timed on
            self._ws = ws
        return self._ws

    def precompute_ms(self, iters=3):
        """wall time of one begin_sample() (host RoPE table + H2D, txt_in, vec path, the all-steps modulation GEMM)"""
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(iters):
            self.begin_sample()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / iters * 1e3


def flops_per_eval(T, N, D=3072, H=24, mlp=12288, depth=19, single=38, in_ch=384, out_ch=64):
    L = T + N
    lin = 2 * L * (depth * 12 * D * D + single * 12 * D * D) + 2 * N * in_ch * D + 2 * N * D * out_ch
    attn = (depth + single) * 4 * L * L * D
    return lin, attn


PMC_FILE = "profiles/r07z_pmc_summary.json"      # the committed `rocprofv3 --pmc` passes over bench.py on the final code


def pmc_traffic(kernel):
    """HBM-side bytes per launch of `kernel`, READ FROM THE COMMITTED rocprofv3 PMC passes of this same command
    (FETCH_SIZE and WRITE_SIZE in separate --pmc runs, KiB units; FETCH_SIZE x2 on gfx950 for wide coalesced reads,
    MI355X_MICROARCH.md §HBM) - hardware counters cannot be sampled from inside bench.py, so this field is a file
    lookup (`measured_in_this_run: false`), null when the file does not hold the kernel."""
    for rel in (PMC_FILE, "profiles/r04_pmc_summary.json", "profiles/r03_pmc_summary.json", "profiles/r02_pmc_summary.json"):
        try:
            d = json.load(open(os.path.join(REPO, rel)))
            name = kernel
            while name not in d and name.endswith(", false>"):       # (older files: fewer trailing template arguments)
                name = name[:-len(", false>")] + ">"
            k = d[name]
            fetch, write = 2.0 * k["FETCH_SIZE"] * 1024.0, k["WRITE_SIZE"] * 1024.0
            return dict(fetch_bytes=round(fetch), write_bytes=round(write), total_bytes=round(fetch + write),
                        unit="bytes/launch", measured_in_this_run=False,
                        source=f"{rel} (rocprofv3 --pmc passes of bench.py, FETCH_SIZE x2 gfx950 correction)")
        except Exception:
            continue
    return None


def pmc_mfma_busy(kernel):
    """Matrix-pipe UTILISATION of `kernel` from the committed PMC passes of this same command on the final code: the share of the
    kernel's shader cycles in which the MFMA pipe of a SIMD is busy, SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs over GRBM_GUI_ACTIVE / 8
    XCDs (MI355X_MICROARCH.md), and the clock the kernel held (cycles / duration).  A file lookup like pmc_traffic():
    hardware counters cannot be sampled from inside bench.py."""
    try:
        k = json.load(open(os.path.join(REPO, PMC_FILE)))[kernel]
        cyc = k["GRBM_GUI_ACTIVE"] / 8.0
        return dict(frac=round(k["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0 / cyc, 4), clock_ghz=round(cyc / k["_dur_ns"], 3),
                    duration_us=round(k["_dur_ns"] * 1e-3, 2), measured_in_this_run=False,
                    source=f"{PMC_FILE} (rocprofv3 --pmc pass of bench.py: SQ_VALU_MFMA_BUSY_CYCLES / 1024 over GRBM_GUI_ACTIVE / 8)")
    except Exception:
        return None


TRAFFIC_KERNEL = "gemm_bf16_kernel<256, 192, 4, 2, 2, 2"     # the GATE_RES instantiation of the loader-wave tile


def traffic_probe():
    """`python bench.py --traffic-probe`, run UNDER rocprofv3 by measure_traffic(): the launch mix `roofline_gemm` times -
    attn.proj (two streams grouped), mlp.2 (grouped) and linear2 in the ratio 19 : 19 : 38 - on cfg-2-sized random operands."""
    from visualcloze_amd import hip
    hip.require_gpu()
    dev, D, T, N, mlp = "cuda:0", 3072, 512, 3456, 12288
    L = T + N
    g = torch.Generator(device=dev).manual_seed(7)
    r = lambda *s, sc=1.0: (torch.randn(*s, device=dev, generator=g) * sc).to(torch.bfloat16)  # noqa: E731
    x = r(L, D)
    gate = r(D)
    att, hid, cat = r(L, D), r(L, mlp), r(L, D + mlp)
    wp, wm, wl = r(D, D, sc=D ** -0.5), r(D, mlp, sc=mlp ** -0.5), r(D, D + mlp, sc=(D + mlp) ** -0.5)
    b = torch.zeros(D, dtype=torch.bfloat16, device=dev)

    def grouped(a, w):
        return [hip.make_problem(a[T:], w, b, x[T:], res=x[T:], gate=gate), hip.make_problem(a[:T], w, b, x[:T], res=x[:T], gate=gate)]
    for _ in range(3):
        hip.gemm(grouped(att, wp), epi=hip.EPI_GATE_RES)
        hip.gemm(grouped(hid, wm), epi=hip.EPI_GATE_RES)
        for _ in range(2):
            hip.gemm(hip.make_problem(cat, wl, b, x, res=x, gate=gate), epi=hip.EPI_GATE_RES)
    # the product's attention launch at cfg 2 (24 heads, variant 28 = stream form with the tail combined in the launch, finished
    # prescaled query rows, bounded logits) on QK-normed random operands - for measure_mfma_busy()
    H = D // 128
    qkv = r(L, 3 * D)
    ones = torch.ones(128, dtype=torch.bfloat16, device=dev)
    pos = torch.arange(L, dtype=torch.float64)[:, None] * torch.linspace(0.01, 1.0, 64, dtype=torch.float64)[None]
    rope = torch.stack([torch.cos(pos), torch.sin(pos)], -1).float().to(dev).contiguous()
    vt = torch.zeros((1, H, 128, (L + 63) // 64 * 64), dtype=torch.bfloat16, device=dev)
    hip.qknorm_rope_vt(qkv, ones, ones, rope, vt, L, H, parts=hip.QKN_Q | hip.QKN_K | hip.QKN_VT | hip.QKN_QPRE)
    o = torch.empty((L, D), dtype=torch.bfloat16, device=dev)
    for _ in range(6):
        hip.attention(qkv, vt, o, L, H, variant=28, q_prescaled=True, logit_bound=16.65)
    torch.cuda.synchronize()


def _pmc_pass(counters, timeout=90):
    """one `rocprofv3 --kernel-trace --pmc <counters>` pass over `bench.py --traffic-probe` in a child process ->
    {(kernel name, counter): [values per dispatch]}, None when rocprofv3 is not here or the pass fails"""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None
    tmp = tempfile.mkdtemp(prefix="vc_pmc_", dir=os.environ.get("TMPDIR", "/tmp"))
    try:
        env = dict(os.environ, TMPDIR=os.environ.get("TMPDIR", "/tmp"))
        for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
            env.pop(k, None)
        r = subprocess.run([exe, "--kernel-trace", "--pmc"] + counters.split() + ["-d", tmp, "-o", "p", "--output-format", "csv", "--",
                            sys.executable, os.path.abspath(__file__), "--traffic-probe"], cwd=tmp, env=env,
                           capture_output=True, text=True, timeout=timeout)
        files = glob.glob(os.path.join(tmp, "**", "*counter_collection.csv"), recursive=True)
        if r.returncode != 0 or not files:
            return None
        out = {}
        for f in files:
            for row in csv.DictReader(open(f)):
                out.setdefault((row["Kernel_Name"].replace("(anonymous namespace)::", ""), row["Counter_Name"]), []).append(float(row["Counter_Value"]))
        return out
    except Exception:
        return None
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def measure_mfma_busy(timeout=90):
    """Matrix-pipe utilisation MEASURED IN THIS RUN, for the roofline GEMM and the product's attention launch: one `rocprofv3
    --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES` pass over `bench.py --traffic-probe` (isolated launches on cfg-2-sized
    QK-normed / random operands, child process): busy = SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs over GRBM_GUI_ACTIVE / 8 XCDs.
    -> {"gemm": frac, "attention": frac} (a key is missing when its kernel did not show up), None when the pass fails."""
    vals = _pmc_pass("GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES", timeout)
    if not vals:
        return None
    out = {}
    for key, sub_ in (("gemm", TRAFFIC_KERNEL), ("attention", "attn64s_kernel<true>")):
        busy = [v for (k, c), xs in vals.items() if sub_ in k and c == "SQ_VALU_MFMA_BUSY_CYCLES" for v in xs]
        act = [v for (k, c), xs in vals.items() if sub_ in k and c == "GRBM_GUI_ACTIVE" for v in xs]
        if len(busy) >= 4 and len(busy) == len(act) and sum(act) > 0:
            out[key] = dict(frac=round((sum(busy) / 1024.0) / (sum(act) / 8.0), 4), dispatches=len(busy), measured_in_this_run=True,
                            source="rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES over `bench.py --traffic-probe` (isolated launches, "
                                   "cfg-2-sized operands) in a child process: SQ_VALU_MFMA_BUSY_CYCLES / 1024 over GRBM_GUI_ACTIVE / 8")
    return out or None


def measure_traffic(timeout=90):
    """HBM-side bytes per launch of the roofline kernel, MEASURED IN THIS RUN: two separate `rocprofv3 --pmc` passes
    (FETCH_SIZE; WRITE_SIZE - they do not share a pass, MI355X_MICROARCH.md) over `bench.py --traffic-probe` in a child
    process, KiB units, FETCH_SIZE x2 (gfx950 counts 128-B requests as 64 B on wide coalesced reads).  None when rocprofv3
    is not on this machine or a pass fails (the committed PMC summary is then quoted instead, labelled as such)."""
    vals = {}
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        got = _pmc_pass(ctr, timeout)
        if not got:
            return None
        xs = [v for (k, c), vs in got.items() if TRAFFIC_KERNEL in k and c == ctr for v in vs]
        if len(xs) < 4:
            return None
        vals[ctr] = sum(xs) / len(xs)
    fetch, write = 2.0 * vals["FETCH_SIZE"] * 1024.0, vals["WRITE_SIZE"] * 1024.0
    return dict(fetch_bytes=round(fetch), write_bytes=round(write), total_bytes=round(fetch + write), unit="bytes/launch",
                measured_in_this_run=True,
                source="rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over `bench.py --traffic-probe` (the timed launch mix "
                       "on cfg-2-sized random operands) in a child process; FETCH_SIZE x2 gfx950 correction")


def roofline_gemm_pyplan(job, iters=3):
    """(A/B runs, `--python-plan`, tools/step_ab.py: through the Python-ordered twin of the plan.)  Time the dominant kernel — gemm_bf16_kernel<.., EPI_GATE_RES> (attn.proj, mlp.2, linear2: 76 launches and
    21.2 TFLOP per evaluation at cfg 2) — launch by launch on the engine stream with HIP events."""
    from visualcloze_amd import hip
    eng, ws = job.eng, job.ws
    D, T, N, L, B = eng.D, ws.T, ws.N, ws.L, ws.B
    nm = eng.W.n_mod
    mss = B * nm
    XI, XT, X, CAT, HID = ws.XI, ws.XT, ws.X, ws.CAT, ws.HID
    ATT = CAT[:, :D]
    HID_I, HID_T = HID[:B * N], HID[B * N:]
    att_i = dict(M=B * N, a_rpb=N, a_bstride=L * CAT.stride(0))
    att_t = dict(M=B * T, a_rpb=T, a_bstride=L * CAT.stride(0))
    launches, flops = [], 0.0

    def P(name, a_, o_, gate, rpb, **kw):
        return eng._prob(name, a_, o_, res=o_, gate=gate, rows_per_batch=rpb, gate_bstride=nm, **kw)
    for i in range(eng.g.depth):
        pf = f"double_blocks.{i}"
        im, tm = pf + ".img_mod.lin", pf + ".txt_mod.lin"
        launches.append([P(pf + ".img_attn.proj", ATT[T:], XI, eng._mod(ws, im, 2), N, **att_i),
                         P(pf + ".txt_attn.proj", ATT[:T], XT, eng._mod(ws, tm, 2), T, **att_t)])
        launches.append([P(pf + ".img_mlp.2", HID_I, XI, eng._mod(ws, im, 5), N),
                         P(pf + ".txt_mlp.2", HID_T, XT, eng._mod(ws, tm, 5), T)])
        flops += B * (2.0 * L * D * D + 2.0 * L * D * eng.mlp)
    for i in range(eng.g.depth_single_blocks):
        pf = f"single_blocks.{i}"
        launches.append([P(pf + ".linear2", CAT, X, eng._mod(ws, pf + ".modulation.lin", 2), L)])
        flops += B * 2.0 * L * D * (D + eng.mlp)
    s = job.s

    def run():
        for ps in launches:
            eng._gemm(ps, epi=hip.EPI_GATE_RES, step_ptr=ws.STEP, gate_step_stride=mss, s=s)
    ws.STEP.zero_()
    with torch.cuda.stream(job.eng.stream):
        run()
        e0, e1 = hip.Event(), hip.Event()
        e0.record(s)
        for _ in range(iters):
            run()
        e1.record(s)
        ms = e0.elapsed_ms(e1) / iters
    n = len(launches)
    achieved = flops / (ms * 1e-3) / 1e12
    return dict(bound="mfma", achieved=round(achieved, 1), peak=MFMA_BF16_PEAK_TFLOPS, unit="TFLOP/s",
                frac=round(achieved / MFMA_BF16_PEAK_TFLOPS, 4), traffic=pmc_traffic("gemm_bf16_kernel<256, 192, 4, 2, 2, 2, false, false, false, false>"),
                kernel="gemm_bf16_kernel<EPI_GATE_RES>", launches_per_eval=n,
                flops_per_launch=flops / n, avg_launch_us=round(ms * 1e3 / n, 2))


def roofline_attention_pyplan(job, iters=3):
    """(A/B runs, `--python-plan`, tools/step_ab.py: through the Python-ordered twin of the plan.)  The attention kernel AS THE PRODUCT RUNS IT (variant by size; with variant 12: finished, prescaled query rows from the
    qkv GEMM's epilogue, tail split), timed IN SITU: HIP events bracket each of the 57 attention launches inside
    whole evaluations of the product's launch plan, so every launch finds the caches as the step graph leaves them (its q / k
    rows and V^T just written by the qkv GEMM and the K pre-pass on other XCDs, the GEMMs' weights streaming through L2 / MALL
    before and after).  `isolated_us` = the same launch with the same arguments back to back on hot caches, for comparison
    only; `achieved` / `frac` are the in-situ numbers."""
    from visualcloze_amd import hip
    eng, ws = job.eng, job.ws
    s = job.s
    v = eng.attention_variant(ws)
    fused_q = bool(v & 8) and bool(eng.fuse_qnorm)
    q_done = eng._qn_in_gemm(ws)                # the qkv GEMM's epilogue left finished, prescaled query rows
    sc = eng.W.w["single_blocks.0.norm.query_norm.scale"]
    qn = (sc, None, 0, ws.ROPE) if fused_q and not q_done else None
    bound = eng.W.logit_bound if eng.bounded_softmax else 0.0
    # attention64.hip runs attn64_kernel<true> (no running max) when the weights' norm scales bound the logits by <= 100
    # and, when the queries arrive finished from the qkv GEMM's epilogue as well, its stream form attn64s_kernel (round 6)
    is_bounded = 0.0 < bound <= 100.0
    template = (("attn64s_kernel<true> (bounded logits, prescaled queries: K / V^T stream across work items)" if q_done else
                 "attn64_kernel<true> (bounded logits: no running max)") if is_bounded else
                ("attn64s_kernel<false> (running max, stream form)" if q_done else "attn64_kernel<false> (running max)")) if v & 8 else "attn_fwd_kernel"

    def in_situ():
        eng.eval_once(ws, ws.STEP, euler=False, s=s)              # warm
        eng.attn_events = []
        try:
            for _ in range(iters):
                eng.eval_once(ws, ws.STEP, euler=False, s=s)
        finally:
            ev, eng.attn_events = eng.attn_events, None
        torch.cuda.synchronize()
        return sorted(a.elapsed_ms(b) for a, b in ev)
    with torch.cuda.stream(eng.stream):
        ws.STEP.zero_()
        situ = in_situ()
        ms = sum(situ) / len(situ)
        runmax_ms = None
        if v & 8 and is_bounded:
            # the same launches through attn64_kernel<false>: what a checkpoint whose QK-norm scales break the logit bound
            # (16.33 max|q scale| max|k scale| > 100) would run - on record every round
            keep = eng.bounded_softmax
            eng.bounded_softmax = False
            try:
                rm = in_situ()
            finally:
                eng.bounded_softmax = keep
            runmax_ms = sum(rm) / len(rm)

        def iso():
            hip.attention(ws.QKV, ws.VT, ws.CAT[:, :eng.D], ws.L, eng.H, variant=v, stream=s, B=ws.B, scratch=eng.attn_scratch,
                          q_norm=qn, logit_bound=bound, q_prescaled=q_done)          # the SAME instantiation the product launches
        iso()
        e0, e1 = hip.Event(), hip.Event()
        e0.record(s)
        for _ in range(iters * 10):
            iso()
        e1.record(s)
        ms_iso = e0.elapsed_ms(e1) / (iters * 10)
    fl = 4.0 * ws.L * ws.L * eng.D * ws.B
    in_launch = bool(v & 16) and is_bounded and q_done          # VcAttention.variant bit 16: the tail pieces are combined inside the launch
    return dict(kernel=(template + (" (tail pieces combined in the launch)" if in_launch else " + attn64_merge_kernel")) if v & 8 else template, variant=v, logit_bound=round(bound, 3),
                query_norm="qkv GEMM epilogue (prescaled)" if q_done else ("attention prologue" if fused_q else "pre-pass"), timed="in situ: HIP events around each attention launch inside product-plan "
                "evaluations", launches_timed=len(situ), avg_launch_us=round(ms * 1e3, 2),
                median_launch_us=round(situ[len(situ) // 2] * 1e3, 2), isolated_us=round(ms_iso * 1e3, 2),
                achieved=round(fl / ms / 1e9, 1), unit="TFLOP/s", frac=round(fl / ms / 1e9 / MFMA_BF16_PEAK_TFLOPS, 4),
                runmax_us=round(runmax_ms * 1e3, 2) if runmax_ms else None,
                runmax_frac=round(fl / runmax_ms / 1e9 / MFMA_BF16_PEAK_TFLOPS, 4) if runmax_ms else None,
                runmax_note="the running-max template (attn64s_kernel<false> with prescaled queries) in situ on the same operands: what a "
                            "checkpoint whose QK-norm scales break the logit bound would run")


EPI_NAMES = {0: "BIAS", 1: "GELU", 2: "GATE_RES", 3: "SILU", 4: "QKV"}


def handle_profile(job, evaluations=3):
    """vc_flux_profile of the job's C handle: the launches of `evaluations` whole evaluations at the current step of the sample
    in flight, issued by the library's own plan (the code its step graph was captured from) with a HIP event in front of and
    behind every GEMM, attention and LayerNorm-modulate launch - the product times itself; no Python-ordered twin involved."""
    if job.step_in_sample >= job.S:
        job.begin_sample()
    with torch.cuda.stream(job.eng.stream):
        recs = job.h.profile(evaluations, job.s)
    for r in recs:
        r["evaluations"] = evaluations
    return recs


def launch_classes(recs):
    """the profile as a table for the bench record: per launch class, launches per evaluation, average / min / max event time,
    and the rate against the bound of its kind (dense bf16 MFMA peak for GEMM and attention, HBM peak for LayerNorm-modulate)"""
    from visualcloze_amd import hip
    out = []
    for r in recs:
        us = r["total_us"] / r["launches"]
        if r["kind"] == hip.LAUNCH_GEMM:
            name = f"gemm {EPI_NAMES.get(r['epi'], r['epi'])} N={r['n']} K={r['k']}"
        elif r["kind"] == hip.LAUNCH_ATTENTION:
            name = f"attention variant {r['epi']}"
        else:
            name = "ln_modulate"
        row = dict(launch=name, per_eval=r["launches"] // r["evaluations"], avg_us=round(us, 2), min_us=round(r["min_us"], 2), max_us=round(r["max_us"], 2))
        if r["flops"] > 0:
            tf = r["flops"] / (r["total_us"] * 1e-6) / 1e12
            row.update(tflops=round(tf, 1), frac_mfma=round(tf / MFMA_BF16_PEAK_TFLOPS, 4))
        if r["bytes"] > 0:
            gb = r["bytes"] / (r["total_us"] * 1e-6) / 1e9
            row.update(gb_per_s=round(gb, 1), frac_hbm=round(gb / HBM_PEAK_GBS, 4))
        out.append(row)
    return out


def roofline_gemm_handle(job, recs):
    """The dominant kernel - gemm_bf16_kernel<.., EPI_GATE_RES> (attn.proj, mlp.2, linear2: 76 launches and 21.2 TFLOP per
    evaluation at cfg 2) - from the handle's own profile: every launch timed where it stands in the step."""
    from visualcloze_amd import hip
    g = [r for r in recs if r["kind"] == hip.LAUNCH_GEMM and r["epi"] == hip.EPI_GATE_RES]
    n, flops, us = sum(r["launches"] for r in g), sum(r["flops"] for r in g), sum(r["total_us"] for r in g)
    achieved = flops / (us * 1e-6) / 1e12
    return dict(bound="mfma", achieved=round(achieved, 1), peak=MFMA_BF16_PEAK_TFLOPS, unit="TFLOP/s",
                frac=round(achieved / MFMA_BF16_PEAK_TFLOPS, 4), traffic=pmc_traffic("gemm_bf16_kernel<256, 192, 4, 2, 2, 2, false, false, false, false>"),
                kernel="gemm_bf16_kernel<EPI_GATE_RES>", launches_per_eval=n // g[0]["evaluations"],
                flops_per_launch=flops / n, avg_launch_us=round(us / n, 2),
                mfma_busy=pmc_mfma_busy("gemm_bf16_kernel<256, 192, 4, 2, 2, 2, false, false, false, false>"),
                timed="vc_flux_profile: HIP events around each launch inside whole evaluations issued by the C handle's own plan")


def roofline_attention_handle(job, recs):
    """The attention launch AS THE PRODUCT RUNS IT, from the handle's own profile (vc_flux_profile): HIP events bracket each of
    the 57 attention launches inside whole evaluations, so every launch finds the caches as the step leaves them (its q / k rows
    and V^T just written by the qkv GEMM on other XCDs, the GEMMs' weights streaming through L2 / MALL before and after).
    `runmax_us`: the same evaluations with the logit bound switched off - the running-max template a checkpoint whose QK-norm
    scales break the bound (16.33 max|q scale| max|k scale| > 100) would run - on record every round."""
    from visualcloze_amd import hip
    eng = job.eng
    a = [r for r in recs if r["kind"] == hip.LAUNCH_ATTENTION]
    n, flops, us = sum(r["launches"] for r in a), sum(r["flops"] for r in a), sum(r["total_us"] for r in a)
    v = a[0]["epi"]
    bound = eng.W.logit_bound if eng.bounded_softmax else 0.0
    is_bounded = 0.0 < bound <= 100.0
    o = job.h._opts                   # the handle's options decide where the query norm runs (flux_engine.hip: qn_in_gemm)
    q_done = bool(o.get("fuse_knorm")) and o.get("qkv_heads", 0) > 0 and bool(v & 8) and o.get("fuse_qnorm", 0) >= 2
    template = (("attn64s_kernel<true> (bounded logits, prescaled queries: K / V^T stream across work items)" if q_done else
                 "attn64_kernel<true> (bounded logits: no running max)") if is_bounded else
                ("attn64s_kernel<false> (running max, stream form)" if q_done else "attn64_kernel<false> (running max)")) if v & 8 else "attn_fwd_kernel"
    in_launch = bool(v & 16) and is_bounded and q_done
    runmax = None
    if v & 8 and is_bounded:
        opts = dict(job.h._opts)
        try:
            hip._check(hip.lib().vc_flux_set_option(job.h.h, b"logit_bound_milli", 0), "vc_flux_set_option")
            rm = [r for r in handle_profile(job, a[0]["evaluations"]) if r["kind"] == hip.LAUNCH_ATTENTION]
            runmax = sum(r["total_us"] for r in rm) / sum(r["launches"] for r in rm)
        finally:
            hip._check(hip.lib().vc_flux_set_option(job.h.h, b"logit_bound_milli", opts["logit_bound_milli"]), "vc_flux_set_option")
    tf = flops / (us * 1e-6) / 1e12
    fl1 = flops / n
    return dict(kernel=(template + (" (tail pieces combined in the launch)" if in_launch else " + attn64_merge_kernel")) if v & 8 else template,
                variant=v, logit_bound=round(bound, 3),
                query_norm="qkv GEMM epilogue (prescaled)" if q_done else "attention prologue / pre-pass",
                timed="in situ by vc_flux_profile: HIP events around each attention launch inside whole evaluations issued by the C handle's own plan",
                launches_timed=n, avg_launch_us=round(us / n, 2), min_launch_us=round(min(r["min_us"] for r in a), 2),
                max_launch_us=round(max(r["max_us"] for r in a), 2),
                achieved=round(tf, 1), unit="TFLOP/s", frac=round(tf / MFMA_BF16_PEAK_TFLOPS, 4),
                mfma_busy=pmc_mfma_busy("attn64s_kernel<true>") if (v & 8 and is_bounded and q_done) else None,
                runmax_us=round(runmax, 2) if runmax else None,
                runmax_frac=round(fl1 / (runmax * 1e-6) / 1e12 / MFMA_BF16_PEAK_TFLOPS, 4) if runmax else None,
                runmax_note="the running-max template (attn64s_kernel<false> with prescaled queries) in situ on the same operands: what a "
                            "checkpoint whose QK-norm scales break the logit bound would run")


def roofline_gemm(job, iters=3, via=None, recs=None):
    if (via or ("handle" if job.h is not None else "python")) == "handle":
        return roofline_gemm_handle(job, recs if recs is not None else handle_profile(job, iters))
    return roofline_gemm_pyplan(job, iters)


def roofline_attention(job, iters=3, via=None, recs=None):
    if (via or ("handle" if job.h is not None else "python")) == "handle":
        return roofline_attention_handle(job, recs if recs is not None else handle_profile(job, iters))
    return roofline_attention_pyplan(job, iters)



def cpu_baseline(T, N, wl):
    """The CPU oracle (a port of the reference path) on the host cores: one DoubleStreamBlock + one SingleStreamBlock
    at full width, extrapolated x(19, 38) to one evaluation - in fp32 (`value`: exact reference semantics, the faster
    of the two on a CPU) and in the bf16 mode that rounds where the reference's autocast run does."""
    import oracle.flux_oracle as O
    G = O.FluxGeometry()
    D, L = G.hidden_size, T + N
    cores = torch.get_num_threads()
    g = torch.Generator().manual_seed(0)
    r = lambda *s: torch.randn(*s, generator=g) * 0.02  # noqa: E731
    sd = {}
    for st in ("img", "txt"):
        sd.update({f"d.{st}_mod.lin.weight": r(6 * D, D), f"d.{st}_mod.lin.bias": torch.zeros(6 * D),
                   f"d.{st}_attn.qkv.weight": r(3 * D, D), f"d.{st}_attn.qkv.bias": torch.zeros(3 * D),
                   f"d.{st}_attn.norm.query_norm.scale": torch.ones(128), f"d.{st}_attn.norm.key_norm.scale": torch.ones(128),
                   f"d.{st}_attn.proj.weight": r(D, D), f"d.{st}_attn.proj.bias": torch.zeros(D),
                   f"d.{st}_mlp.0.weight": r(4 * D, D), f"d.{st}_mlp.0.bias": torch.zeros(4 * D),
                   f"d.{st}_mlp.2.weight": r(D, 4 * D), f"d.{st}_mlp.2.bias": torch.zeros(D)})
    sd.update({"s.modulation.lin.weight": r(3 * D, D), "s.modulation.lin.bias": torch.zeros(3 * D),
               "s.linear1.weight": r(7 * D, D), "s.linear1.bias": torch.zeros(7 * D),
               "s.linear2.weight": r(D, 5 * D), "s.linear2.bias": torch.zeros(D),
               "s.norm.query_norm.scale": torch.ones(128), "s.norm.key_norm.scale": torch.ones(128)})
    img, txt, vec = torch.randn(1, N, D, generator=g), torch.randn(1, T, D, generator=g), torch.randn(1, D, generator=g)
    ids = torch.cat((torch.zeros(T, 3), grid_img_ids(wl['row_latents'])))[None]
    cs = O.rope_cos_sin(ids, G.axes_dim, G.theta)
    x = torch.cat((txt, img), 1)

    def leg(P, reps):
        td = ts = 1e30
        with torch.no_grad():
            for _ in range(reps):
                t0 = time.time(); O.double_block(sd, "d", img, txt, vec, cs, G, P); td = min(td, time.time() - t0)
                t0 = time.time(); O.single_block(sd, "s", x, vec, cs, G, P); ts = min(ts, time.time() - t0)
        return td, ts, 1.0 / (G.depth * td + G.depth_single_blocks * ts)
    td, ts, v32 = leg(O.Prec("fp32"), 2)
    tdb, tsb, v16 = leg(O.Prec("bf16", "merged"), 1)
    return dict(value=round(v32, 5), unit="denoising-steps/sec", cores=cores, kind="port", bf16_value=round(v16, 5),
                sample=f"oracle: 1 DoubleStreamBlock + 1 SingleStreamBlock at L={L}, D={D}, extrapolated x(19,38) to one "
                       f"evaluation; fp32 {td:.2f}s + {ts:.2f}s (best of 2) -> value; bf16 rounding mode {tdb:.2f}s + {tsb:.2f}s "
                       f"-> bf16_value")


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=29)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="384-grid-2x3", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-traffic", action="store_true", help="skip the rocprofv3 --pmc child passes (roofline.traffic = committed file)")
    ap.add_argument("--traffic-probe", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--stub-engine", action="store_true", help=argparse.SUPPRESS)   # tests: the driver over gloo without a GPU
    # tests (tests/test_parallel_gpu.py): the REAL engine in a world of two on ONE GPU - RCCL refuses two ranks on one device,
    # so the ranks rendezvous over gloo, share --device-index, and run the tiny model (the record is marked, never a measurement)
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"], help=argparse.SUPPRESS)
    ap.add_argument("--device-index", type=int, default=None, help=argparse.SUPPRESS)
    ap.add_argument("--test-tiny", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--tile-cfg", type=int, default=None)
    ap.add_argument("--attn-variant", type=int, default=None)
    ap.add_argument("--no-fuse-vt", action="store_true", help="A/B: V^T by the pre-pass kernel instead of the qkv GEMM's epilogue")
    ap.add_argument("--no-fuse-knorm", action="store_true", help="A/B: key QKNorm + RoPE by the pre-pass kernel instead of the qkv GEMM's epilogue")
    ap.add_argument("--no-splitk", action="store_true", help="A/B: never cut GEMM remainder tiles along K (VcGemmArgs.splitk_ws)")
    ap.add_argument("--python-plan", action="store_true",
                    help="A/B: order the launches from Python (engine.FluxEngine) instead of the C handle API")
    ap.add_argument("--per-gpu-batch", type=int, default=1,
                    help="independent grids advanced together by one graph replay on each GPU (throughput mode; "
                         "BASELINE's cfg 2 is 1)")
    return ap.parse_args(argv)


def timed_region(job, steps, warmup, barrier=None, max_over_ranks=None, board=None):
    """The contract's timing: W untimed steps, barrier + synchronize, EXACTLY K steps, barrier + synchronize, max over
    ranks.  The timed region opens on a sample boundary, so the per-sample precompute of a new grid (txt_in, the vec
    path, all 29 x 1.06 M modulation rows in one GEMM, the RoPE table: `precompute_ms`) is inside it."""
    from visualcloze_amd import parallel as par
    barrier = barrier or par.barrier
    for _ in range(warmup):
        job.step()
    job.restart_sample()                            # the next step() starts a new grid
    barrier()
    if board is not None:
        board.__enter__()                           # a sampler thread reads the board's power / clock nodes while the K steps run
    t0 = time.perf_counter()
    for _ in range(steps):
        job.step()
    if torch.cuda.is_available():
        torch.cuda.synchronize()
    timed_region.own_elapsed = time.perf_counter() - t0     # this rank's own K steps, before it waits for the others (`per_rank`)
    barrier()                                       # synchronize + barrier + synchronize
    elapsed = time.perf_counter() - t0
    if board is not None:
        board.__exit__(None, None, None)
    return (max_over_ranks or par.max_over_ranks)(elapsed)


def result_record(a, wl, world, elapsed, T, N, bcast_s, weight_bytes, rccl_ranks=0):
    """The ONE JSON line of the contract (rank 0 adds roofline / cpu_baseline).  `rccl_ranks` is what RCCL itself counted
    (parallel.rccl_rank_count: an all-reduce of ones over the "nccl" communicator; 0 = no RCCL communicator in this run)."""
    PB = a.per_gpu_batch
    lin, attn = flops_per_eval(T, N)
    value = world * PB * a.steps / elapsed          # a replay advances PB grids by one solver step each
    rec = {
        "metric": "denoising-steps/sec", "value": round(value, 4), "unit": "denoising-steps/sec", "n_gpus": world,
        "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(elapsed / (a.steps * PB) * 1e3, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
        "config": {"workload": f"{a.workload} in-context, {wl['steps']} solver points = {wl['steps'] - 1} Flux evaluations "
                               f"per grid, L={T}+{N} tokens, FLUX.1-Fill-dev geometry + LoRA r256 (merged), random-init "
                               f"weights, {PB} independent grid(s) per GPU (data-parallel, no in-step collective)",
                   "global_batch": world * PB, "seq_len": T + N, "parallelism": f"dp{world}"},
        "img_per_sec": round(value / (wl["steps"] - 1), 5),
        "model_tflops_per_eval": round((lin + attn) / 1e12, 2),
        "achieved_model_tflops_per_gpu": round((lin + attn) / 1e12 * value / world, 1),
        "per_gpu_batch": PB,
        "rccl_ranks": rccl_ranks,
        "weight_broadcast_s": round(bcast_s, 3),
        "weight_broadcast_gbps": round(weight_bytes / bcast_s / 1e9, 1) if bcast_s > 0 else None,
    }
    return rec


def rank_record(rank, local, a, own_elapsed, board=None, numa=None, device=None):
    """What every rank contributes to the line's `per_rank` list: its own rate over its own wall time (the job's `value` uses
    the slowest rank's), its board's power / shader clock during the timed steps and where it ran - so a throttling GPU that
    sets max-over-ranks is visible in the one JSON line."""
    b = board.summary() if board is not None else {}
    return {"rank": rank, "local_rank": local, "device": device, "numa_node": numa,
            "steps_per_s": round(a.steps * a.per_gpu_batch / own_elapsed, 4), "elapsed_s": round(own_elapsed, 4),
            "power_w_avg": b.get("power_w_avg"), "power_cap_w": b.get("power_cap_w"), "sclk_mhz_avg": b.get("sclk_mhz_avg"),
            "board_source": b.get("source")}


def self_launch(a, argv):
    """`python bench.py --gpus N` started BARE (no WORLD_SIZE in the environment): re-run this command as N ranks under
    torch.distributed.run on this node - the launch line the task statement gives - and hand its exit code back.  Rank 0 of
    the child job prints the one JSON line on the inherited stdout."""
    import subprocess
    # --standalone: torchrun picks (and holds) its own rendezvous port - no probe socket closed before the bind (advisor r04)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--standalone", "--nnodes=1", f"--nproc-per-node={a.gpus}",
           "--local-addr", "127.0.0.1", os.path.abspath(__file__)] + list(sys.argv[1:] if argv is None else argv)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))   # dmabuf IPC for RCCL
    return subprocess.call(cmd, env=env)


class StubJob:
    """--stub-engine (tests only): the Job protocol without a GPU - what the multi-rank driver needs to be exercised over gloo"""

    def __init__(self, rank):
        self.rank = rank

    def restart_sample(self):
        pass

    def step(self):
        time.sleep(0.002 * (self.rank + 1))


def main(argv=None):
    a = parse_args(argv)
    if a.traffic_probe:
        return traffic_probe()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        rc = self_launch(a, argv)
        if rc != 0:
            raise SystemExit(rc)
        return
    rank = int(os.environ.get("RANK", 0))
    local = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {a.gpus}")
    if a.stub_engine:
        from visualcloze_amd import parallel as par
        par.init_distributed("gloo")
        wl = WORKLOADS[a.workload]
        elapsed = timed_region(StubJob(rank), a.steps, a.warmup)
        N = sum((h // 2) * (w // 2) for h, w in wl["row_latents"])
        rec = result_record(a, wl, world, elapsed, 512, N, bcast_s=0.0, weight_bytes=0, rccl_ranks=par.rccl_rank_count())
        rec["per_rank"] = par.gather_records(rank_record(rank, local, a, timed_region.own_elapsed))
        rec["data"] = "stub engine (driver test, no GPU)"
        rec["stub"] = True                          # NOT a measurement: the timings are time.sleep (advisor r04)
        rec["metric"] = "stub-driver-test"
        if rank == 0:
            print(json.dumps(rec), flush=True)
        par.barrier()
        if world > 1:
            torch.distributed.destroy_process_group()
        return
    from visualcloze_amd import hip
    hip.require_gpu()                       # fails loudly: there is no CPU path to fall back to
    if a.device_index is not None:
        local = a.device_index              # (tests: every rank on the same GPU)
    elif torch.cuda.device_count() < world or local >= torch.cuda.device_count():
        raise SystemExit(f"--gpus {a.gpus}: this node exposes {torch.cuda.device_count()} GPU(s) to rank {rank} (LOCAL_RANK {local}); "
                         f"one process per GPU needs {world} (ranks never share a device outside the --device-index tests)")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    from visualcloze_amd import parallel as par
    numa = par.pin_to_gpu_numa(local) if world > 1 else None     # one rank per GPU, each on its GPU's socket
    par.init_distributed(a.backend, dev)
    assert par.world() == world
    wl = WORKLOADS[a.workload]
    tiny = None
    if a.test_tiny:
        tiny = dict(in_channels=384, out_channels=64, vec_in_dim=64, context_in_dim=128, hidden_size=256, mlp_ratio=4.0, num_heads=2,
                    depth=2, depth_single_blocks=2, axes_dim=[16, 56, 56], theta=10_000, qkv_bias=True, guidance_embed=True)
    model, bcast_s = build_model(dev, rank, world, lora_rank=8 if tiny else 256, params=tiny)
    weight_bytes = sum(p.numel() * p.element_size() for p in model.parameters())
    eng = model.prepare(free_parameters=True)   # sampling-only process: keep the 23.8 GB of merged weights, drop the rest
    if a.tile_cfg is not None:
        eng.tile_cfg = a.tile_cfg
    if a.attn_variant is not None:
        eng.attn_variant = a.attn_variant
    model.use_handle = not a.python_plan
    eng.fuse_vt = not a.no_fuse_vt
    if a.no_fuse_knorm:
        eng.fuse_knorm = False
    eng.splitk = not a.no_splitk
    PB = a.per_gpu_batch
    x, kw = make_inputs(dev, wl, seed=par.sample_seed(0, rank * PB), B=PB,   # seed from the global sample index
                        **(dict(ctx_dim=tiny["context_in_dim"], vec_dim=tiny["vec_in_dim"]) if tiny else {}))
    job = Job(model, x, kw, wl["steps"], t0=wl.get("t0", 0.0), do_shift=wl.get("do_shift", True))

    from visualcloze_amd.board import BoardSampler, pci_bus_id_of
    # every rank samples ITS GPU (socket power, power cap, shader clock; sysfs hwmon) at 10 Hz - one small file read per
    # 100 ms beside a loop that issues one graph launch per ~50 ms; the rate is in the record (`board.hz`)
    board = BoardSampler(pci_bus_id_of(local), index=local, hz=10.0)
    with torch.cuda.stream(eng.stream):
        elapsed = timed_region(job, a.steps, a.warmup, max_over_ranks=lambda s: par.max_over_ranks(s, dev), board=board)
    with torch.cuda.stream(eng.stream):
        final = job.state().float()
    torch.cuda.synchronize()
    assert torch.isfinite(final).all(), "non-finite latent"

    T, N = 512, x.shape[1]
    rec = result_record(a, wl, world, elapsed, T, N, bcast_s, weight_bytes, rccl_ranks=par.rccl_rank_count(dev))
    rec["numa_node"] = numa
    rec["per_rank"] = par.gather_records(rank_record(rank, local, a, timed_region.own_elapsed, board, numa, pci_bus_id_of(local)))
    rec["hbm_resident_gb"] = round(torch.cuda.memory_allocated(dev) / 1e9, 1)        # merged weights + workspaces while sampling
    rec["hbm_peak_gb"] = round(torch.cuda.max_memory_allocated(dev) / 1e9, 1)        # during the one-time LoRA merge
    if tiny:
        rec.update(stub=True, metric="stub-driver-test", data="tiny test model (2+2 blocks, hidden 256): driver test, not a measurement",
                   final_latent_sum=float(final.double().sum()))
        if rank == 0:
            print(json.dumps(rec), flush=True)
    elif rank == 0:
        rec["board"] = board.summary()              # sampled DURING the timed steps: did this box sit at its power cap?
        rec["precompute_ms"] = round(job.precompute_ms(), 3)
        recs = handle_profile(job) if job.h is not None else None      # the C handle's own stopwatch (vc_flux_profile)
        rec["roofline"] = roofline_gemm(job, recs=recs)
        if recs is not None:
            rec["launch_classes"] = launch_classes(recs)
        profiled = any("rocprof" in os.environ.get(k, "").lower() for k in ("LD_PRELOAD", "ROCP_TOOL_LIBRARIES", "HSA_TOOLS_LIB"))
        if world == 1 and not a.no_traffic and not profiled:
            live = measure_traffic()                      # (not when this very run is being profiled already)
            if live is not None:
                rec["roofline"]["traffic"] = live
            busy_live = measure_mfma_busy()               # matrix-pipe utilisation, one more counter pass over the same probe
        rec["attention_kernel"] = roofline_attention(job, recs=recs)
        if world == 1 and not a.no_traffic and not profiled and busy_live:
            if "gemm" in busy_live:
                rec["roofline"]["mfma_busy"] = busy_live["gemm"]
            if "attention" in busy_live and rec["attention_kernel"].get("mfma_busy") is not None:
                rec["attention_kernel"]["mfma_busy"] = busy_live["attention"]
        if world == 1 and not a.no_cpu_baseline:
            rec["cpu_baseline"] = cpu_baseline(T, N, wl)
            rec["gpu_over_cpu"] = round(rec["value"] / rec["cpu_baseline"]["value"], 1)
        print(json.dumps(rec), flush=True)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
