"""-m gpu: the REAL engine in a world of two, on the one GPU a test box has.

RCCL refuses two ranks on one device, so the two processes rendezvous over gloo and share cuda:0; everything else is the
product's multi-rank path end to end: a rank whose weights are deliberately different receives rank 0's through
`broadcast_weights` (bucketed; the prepared engine of the receiver is invalidated), every rank prepares a sampling-only
engine (`prepare(free_parameters=True)`), samples ITS shard of the global sample list through the fused sampler (C handle,
hipGraph replays) with seeds from the global index, and `gather_latents` returns them to rank 0 in global order - where each
must equal the single-process result bit for bit.  Then `bench.py --gpus 2` itself, self-launched, over the same route."""
import json
import os
import socket
import subprocess
import sys
import textwrap

import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent("""
    import os, sys, torch
    sys.path.insert(0, %r)
    from visualcloze_amd import hip, parallel as par
    hip.require_gpu()
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    par.init_distributed("gloo")
    r, w = par.rank(), par.world()
    assert w == 2
    from tests.helpers import tiny_model
    from tests.procedural import tiny_inputs
    m, _ = tiny_model()
    if r == 1:                                   # a rank that must NOT keep its own weights
        with torch.no_grad():
            for p in m.parameters():
                p.mul_(0.5).add_(0.25)
        m.prepare()                              # ... and that already holds a prepared (merged) copy of the wrong ones
    def checksum():
        return float(sum(p.double().abs().sum() for p in m.parameters()))
    before = [None] * w
    torch.distributed.all_gather_object(before, checksum())
    assert before[0] != before[1], before
    secs = par.broadcast_weights(m, src=0, bucket_bytes=1 << 20)        # several buckets even at this size
    after = [None] * w
    torch.distributed.all_gather_object(after, checksum())
    assert after[0] == after[1] == before[0], (before, after)
    eng = m.prepare(free_parameters=True)        # sampling-only rank: the merged set stays, the parameters go
    assert sum(p.numel() for p in m.parameters()) == 0
    from visualcloze_amd.transport import Sampler, create_transport
    fn = Sampler(create_transport()).sample_ode(sampling_method="euler", num_steps=6, do_shift=True, time_shifting_factor=1)
    bf = torch.bfloat16
    def run(i):
        inp = tiny_inputs(B=1, seed=par.sample_seed(100, i))
        kw = dict(txt=inp["txt"].to(dev, bf), txt_ids=inp["txt_ids"].to(dev), txt_mask=inp["txt_mask"].to(dev), y=inp["y"].to(dev, bf),
                  img_ids=inp["img_ids"].to(dev), img_mask=inp["img_mask"].to(dev), cond=inp["cond"].to(dev, bf),
                  guidance=inp["guidance"].to(dev, bf))
        out = fn(inp["x"].to(dev, bf), m.forward, kw)[-1]
        torch.cuda.synchronize()
        return out
    n = 5                                        # rank 0: samples 0, 2, 4; rank 1: samples 1, 3
    mine = par.shard_indices(n)
    assert mine == list(range(r, n, 2))
    lat = par.gather_latents([run(i) for i in mine], n)
    if r == 0:
        assert len(lat) == n
        for i in range(n):
            want = run(i).cpu()                  # the single-process result of global sample i
            assert torch.isfinite(want.float()).all() and torch.equal(lat[i], want), i
        assert not torch.equal(lat[0], lat[1])
    else:
        assert lat is None
    t = par.max_over_ranks(0.5 + r)
    assert t == 1.5
    par.barrier()
    torch.distributed.destroy_process_group()
    print("rank", r, "ok", flush=True)
""") % REPO


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_real_engine_world2_on_one_gpu_over_gloo(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    port = _free_port()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   OMP_NUM_THREADS="4", HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=900)[0] for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, f"rank {r} failed:\n{o[-4000:]}"
        assert f"rank {r} ok" in o


def test_bench_gpus2_self_launch_with_the_real_engine():
    """`python bench.py --gpus 2` started bare: re-launches itself as two ranks (torch.distributed.run --standalone), each
    builds the model (meta construction, rank 0 initialises, broadcast into rank 1), prepares a sampling-only engine, runs the
    timed region through the C handle, max over ranks, rank 0 prints ONE line with n_gpus = 2.  (gloo + one shared device +
    the tiny model: the record says so and is marked as a driver test.)"""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "4"
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--backend", "gloo", "--device-index", "0",
                        "--test-tiny", "--workload", "384-grid-1x2", "--steps", "6", "--warmup", "2"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout + r.stderr)[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout + r.stderr
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["config"]["parallelism"] == "dp2" and rec["scaling"] == "weak"
    assert rec["rccl_ranks"] == 0                # the ranks met over gloo (RCCL refuses two ranks on one device): RCCL counted none
    assert [q["rank"] for q in rec["per_rank"]] == [0, 1] and all(q["steps_per_s"] > 0 for q in rec["per_rank"])
    assert rec["stub"] is True and rec["metric"] == "stub-driver-test"
    assert rec["value"] > 0 and rec["weight_broadcast_s"] > 0 and abs(rec["final_latent_sum"]) < float("inf")
