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


def test_single_process_degenerates():
    from visualcloze_amd import parallel as par
    assert par.world() == 1 and par.rank() == 0
    assert par.shard_indices(3) == [0, 1, 2]
    assert par.max_over_ranks(1.5) == 1.5
    import torch
    assert par.broadcast_weights(torch.nn.Linear(2, 2)) == 0.0
