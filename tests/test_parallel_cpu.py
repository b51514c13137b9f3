"""CPU, world_size 2 over gloo: the data-parallel path (weight broadcast, sample sharding, per-sample seeds,
max-over-ranks timing, latent gather) that bench.py and multi-GPU sampling use with RCCL on the GPU box."""
import os
import socket
import subprocess
import sys
import textwrap

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent("""
    import os, sys, torch
    sys.path.insert(0, %r)
    from visualcloze_amd import parallel as par
    from visualcloze_amd.model import FluxLoraWrapper, FluxParams
    from tests.procedural import TINY
    par.init_distributed("gloo")
    r, w = par.rank(), par.world()
    assert w == 2
    torch.manual_seed(100 + r)                       # ranks start with DIFFERENT weights
    m = FluxLoraWrapper(lora_rank=4, lora_scale=1.0, params=FluxParams(**TINY))
    with torch.no_grad():
        for p in m.parameters():
            p.normal_(0, 0.02)
    before = torch.cat([p.reshape(-1) for p in m.parameters()]).clone()
    secs = par.broadcast_weights(m, src=0, bucket_bytes=1 << 20)   # small buckets -> several coalesced broadcasts
    after = torch.cat([p.reshape(-1) for p in m.parameters()])
    ref = [None, None]
    torch.distributed.all_gather_object(ref, after.double().sum().item())
    assert ref[0] == ref[1], ref                     # identical weights everywhere
    if r == 0:
        assert torch.equal(before, after)            # source unchanged
    else:
        assert not torch.equal(before, after)
    # sharding + seeds: 5 samples over 2 ranks; noise depends on the global index only
    mine = par.shard_indices(5)
    assert mine == ([0, 2, 4] if r == 0 else [1, 3])
    lat = [torch.randn(4, 3, generator=torch.Generator().manual_seed(par.sample_seed(7, i))) for i in mine]
    out = par.gather_latents(lat, 5)
    if r == 0:
        for i in range(5):
            exp = torch.randn(4, 3, generator=torch.Generator().manual_seed(7 + i))
            assert torch.equal(out[i], exp), i
    else:
        assert out is None
    # a job that mixes grid sizes cannot use the padded tensor collective: falls back to gather_object, same contract
    het = [torch.full((2 + i, 3), float(i)) for i in mine]
    out = par.gather_latents(het, 5)
    if r == 0:
        assert [tuple(o.shape) for o in out] == [(2 + i, 3) for i in range(5)] and all(float(out[i][0, 0]) == i for i in range(5))
    else:
        assert out is None
    # mixed dtypes on ONE rank only (rank 1 uniform) -> still the fallback everywhere
    mix = [torch.full((2,), float(i)).to(torch.bfloat16 if (r == 0 and i == 2) else torch.float32) for i in mine]
    out = par.gather_latents(mix, 5)
    if r == 0:
        assert out[2].dtype == torch.bfloat16 and out[1].dtype == torch.float32 and all(float(out[i][0]) == i for i in range(5))
    # bf16 latents (the pipeline's state dtype) through the tensor path
    lb = [torch.full((3,), float(i)).to(torch.bfloat16) for i in mine]
    out = par.gather_latents(lb, 5)
    if r == 0:
        assert all(o.dtype == torch.bfloat16 and float(o[0]) == i for i, o in enumerate(out))
    t = par.max_over_ranks(1.0 + r)
    assert t == 2.0
    par.barrier()
    torch.distributed.destroy_process_group()
    print("rank", r, "ok")
""") % REPO


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_data_parallel_path_world2_gloo(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    port = _free_port()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), OMP_NUM_THREADS="2")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=240)[0] for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, f"rank {r} failed:\n{o}"
        assert f"rank {r} ok" in o


BENCH_WORKER = textwrap.dedent("""
    import json, os, sys, time, torch
    sys.path.insert(0, %r)
    import bench
    from visualcloze_amd import parallel as par
    par.init_distributed("gloo")
    r, w = par.rank(), par.world()
    a = bench.parse_args(["--gpus", str(w), "--steps", "6", "--warmup", "2", "--workload", "384-grid-1x2"])
    wl = bench.WORKLOADS[a.workload]
    x, kw = bench.make_inputs("cpu", wl, seed=par.sample_seed(0, r * a.per_gpu_batch), B=a.per_gpu_batch)

    class StubJob:                       # stands in for the engine-backed Job: same driver protocol, no GPU
        def __init__(self): self.steps = self.restarts = 0; self.first_after_restart = None
        def restart_sample(self): self.restarts += 1; self.pending = True
        def step(self):
            if getattr(self, "pending", False): self.first_after_restart = self.steps; self.pending = False
            self.steps += 1; time.sleep(0.002 * (r + 1))      # rank w-1 is the slowest
    job = StubJob()
    elapsed = bench.timed_region(job, a.steps, a.warmup)
    assert job.steps == a.steps + a.warmup and job.restarts == 1 and job.first_after_restart == a.warmup
    assert elapsed >= 0.002 * w * a.steps                      # max over ranks: everyone reports the slowest rank's time
    rec = bench.result_record(a, wl, w, elapsed, 512, x.shape[1], bcast_s=0.5, weight_bytes=26.3e9, rccl_ranks=par.rccl_rank_count())
    assert rec["n_gpus"] == w and rec["scaling"] == "weak"
    assert rec["rccl_ranks"] == 0                              # a gloo world: RCCL saw no rank, and the record says so
    per_rank = par.gather_records(bench.rank_record(r, r, a, bench.timed_region.own_elapsed, numa=None))
    if r == 0:
        assert [q["rank"] for q in per_rank] == list(range(w))
        rates = [q["steps_per_s"] for q in per_rank]
        assert all(x > 0 for x in rates) and rates[0] > rates[-1]          # rank w-1 sleeps longest: the slow rank is visible
        assert abs(min(q["steps_per_s"] for q in per_rank) * w - rec["value"]) / rec["value"] < 0.25
        rec["per_rank"] = per_rank
    else:
        assert per_rank is None
    assert abs(rec["value"] - w * a.steps / elapsed) < 1e-3 and rec["config"]["parallelism"] == f"dp{w}"
    assert rec["weight_broadcast_gbps"] == 52.6
    # per-rank inputs come from the GLOBAL sample index: distinct across ranks, reproducible without the job
    sums = [None] * w
    torch.distributed.all_gather_object(sums, float(x.float().sum()))
    assert len(set(sums)) == w, sums
    x1, _ = bench.make_inputs("cpu", wl, seed=par.sample_seed(0, r), B=1)
    assert torch.equal(x, x1)
    lat = par.gather_latents([x[0, :4, :3].float()], w)
    if r == 0:
        assert len(lat) == w and all(abs(float(lat[i].sum()) - float(bench.make_inputs("cpu", wl, seed=i)[0][0, :4, :3].float().sum())) < 1e-6 for i in range(w))
        print(json.dumps(rec))
    par.barrier()
    torch.distributed.destroy_process_group()
    print("rank", r, "ok")
""") % REPO


def test_bench_driver_world8_gloo_stub_engine(tmp_path):
    """bench.py's distributed driver (rank/seed mapping, barrier-bracketed timing, max over ranks, whole-job value, the
    JSON record) with 8 ranks over gloo and a stub in place of the GPU engine: `torchrun --nproc-per-node 8 bench.py
    --gpus 8` on RCCL is then the only link this container cannot exercise."""
    import json
    script = tmp_path / "bench_worker.py"
    script.write_text(BENCH_WORKER)
    port = _free_port()
    procs = []
    for r in range(8):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="8", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=300)[0] for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, f"rank {r} failed:\n{o}"
        assert f"rank {r} ok" in o
    rec = json.loads([ln for ln in outs[0].splitlines() if ln.startswith("{")][0])
    assert rec["n_gpus"] == 8 and rec["metric"] == "denoising-steps/sec" and rec["higher_is_better"] is True
    assert rec["rccl_ranks"] == 0 and len(rec["per_rank"]) == 8


def test_bench_bare_gpus_flag_launches_its_own_ranks():
    """`python bench.py --gpus 2` WITHOUT WORLD_SIZE in the environment (the form the round-end driver used for --gpus 1) must not
    exit with an error: bench.main re-runs itself under `torch.distributed.run --standalone --nnodes=1 --nproc-per-node 2
    --local-addr 127.0.0.1`, both ranks rendezvous (gloo, stub engine: no GPU here), rank 0 prints the one JSON line of the contract."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "1"
    code = "import sys; sys.path.insert(0, %r); import bench; bench.main(['--gpus', '2', '--steps', '5', '--warmup', '1', '--stub-engine'])" % REPO
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout + r.stderr                      # ONE line, from rank 0 of the child job
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["steps"] == 5 and rec["config"]["parallelism"] == "dp2"
    assert rec["rccl_ranks"] == 0                                    # gloo ranks: no RCCL communicator existed
    assert [q["rank"] for q in rec["per_rank"]] == [0, 1] and rec["per_rank"][0]["steps_per_s"] > rec["per_rank"][1]["steps_per_s"]
    assert rec["stub"] is True and rec["metric"] == "stub-driver-test"      # a --stub-engine line can never be scraped as a measurement
    assert rec["value"] <= 2 * 5 / (5 * 0.004) * 1.01                # whole-job steps over the SLOWEST rank's time (rank 1: 4 ms / step)
    # a WORLD_SIZE that contradicts --gpus is still an error, not a silent single-rank run
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--stub-engine"], env=dict(env, WORLD_SIZE="1"),
                       capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "WORLD_SIZE=1" in r.stderr


def test_cpulist_parser():
    from visualcloze_amd import parallel as par
    assert par.parse_cpulist("0-3,8,10-11\n") == [0, 1, 2, 3, 8, 10, 11]
    assert par.parse_cpulist("") == []


def test_single_process_degenerates():
    from visualcloze_amd import parallel as par
    assert par.world() == 1 and par.rank() == 0
    assert par.shard_indices(3) == [0, 1, 2]
    assert par.max_over_ranks(1.5) == 1.5
    import torch
    assert par.broadcast_weights(torch.nn.Linear(2, 2)) == 0.0
    lat = par.gather_latents([torch.ones(2, 3), torch.zeros(2, 3)], 2)
    assert len(lat) == 2 and all(t.device.type == "cpu" for t in lat)
    src = [torch.ones(2, 3), torch.zeros(2, 3)]
    same = par.gather_latents(src, 2, to_host=False)            # world of one, to_host=False: the caller's own storage, no copy
    assert all(a.data_ptr() == b.data_ptr() for a, b in zip(same, src))


def test_forced_world1_collectives_gloo(tmp_path):
    """force=True runs the real collectives in a world of ONE process (what the one-GPU RCCL smoke test does with the
    "nccl" backend): bucketed flat-buffer broadcast, latent gather, max over ranks."""
    import subprocess
    code = textwrap.dedent("""
        import os, sys, torch
        sys.path.insert(0, %r)
        from visualcloze_amd import parallel as par
        par.init_distributed("gloo", force=True)
        assert torch.distributed.is_initialized() and par.world() == 1
        m = torch.nn.Sequential(*[torch.nn.Linear(64, 64) for _ in range(6)])
        before = torch.cat([p.reshape(-1) for p in m.parameters()]).clone()
        assert par.broadcast_weights(m, bucket_bytes=40000, force=True) > 0.0
        assert torch.equal(before, torch.cat([p.reshape(-1) for p in m.parameters()]))
        out = par.gather_latents([torch.arange(6.).reshape(2, 3)], 1, force=True)
        assert len(out) == 1 and torch.equal(out[0], torch.arange(6.).reshape(2, 3))
        assert par.max_over_ranks(2.5, force=True) == 2.5
        torch.distributed.destroy_process_group()
        print("ok")
    """) % REPO
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=240)
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout + r.stderr
