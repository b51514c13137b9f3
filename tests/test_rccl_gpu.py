"""-m gpu: RCCL on ONE GPU.  A world-1 "nccl" (= RCCL on ROCm) process group is created on purpose and the data-parallel
helpers are forced through their collective code - communicator creation, the bucketed flat-buffer DEVICE broadcast of the
weights, the padded device-tensor latent gather, the max-over-ranks all-reduce - so that all of it has executed on an
MI355X before an 8-GPU run (no scaling number is claimed here).  Runs in a subprocess: a process group is
process-global state."""
import os
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
    par.init_distributed("nccl", dev, force=True)
    assert torch.distributed.is_initialized() and torch.distributed.get_backend() == "nccl" and par.world() == 1
    from tests.helpers import tiny_model
    from tests.procedural import tiny_inputs
    m, sd = tiny_model()
    before = torch.cat([p.detach().reshape(-1).float() for p in m.parameters()]).clone()
    n_bytes = sum(p.numel() * p.element_size() for p in m.parameters())
    secs = par.broadcast_weights(m, src=0, bucket_bytes=max(4096, n_bytes // 5), force=True)   # several coalesced buckets
    assert secs > 0.0
    after = torch.cat([p.detach().reshape(-1).float() for p in m.parameters()])
    assert torch.equal(before, after)
    # a 256 MiB bucket as the full-size job sends them (1 GiB there)
    big = torch.nn.Linear(8192, 8192, bias=False, device=dev, dtype=torch.bfloat16)
    big2 = torch.nn.Linear(8192, 8192, bias=False, device=dev, dtype=torch.bfloat16)
    s0 = float(big.weight.float().sum()) + float(big2.weight.float().sum())
    secs_big = par.broadcast_weights(torch.nn.Sequential(big, big2), bucket_bytes=1 << 30, force=True)
    assert float(big.weight.float().sum()) + float(big2.weight.float().sum()) == s0
    # the model still runs after the broadcast invalidated its prepared engine (rebuilt from the broadcast weights)
    import oracle.flux_oracle as O
    from tests.test_model_gpu import _fwd, _oracle, rel_l2
    inp = tiny_inputs(B=1)
    t = torch.tensor([0.7])
    assert rel_l2(_fwd(m, inp, t), _oracle(sd, inp, t)) < 2e-2
    lat = [torch.full((1, 24, 64), 3.0, device=dev, dtype=torch.bfloat16)]
    out = par.gather_latents(lat, 1, force=True)
    assert len(out) == 1 and out[0].device.type == "cpu" and out[0].dtype == torch.bfloat16 and float(out[0].float().mean()) == 3.0
    assert par.max_over_ranks(1.25, dev, force=True) == 1.25
    assert par.rccl_rank_count(dev) == 1         # what bench.py reports as `rccl_ranks`: RCCL's own count of the communicator
    par.barrier()
    torch.distributed.destroy_process_group()
    print("rccl world-1 ok: %%.1f MB in %%.4f s, 256 MiB bucket pair in %%.4f s" %% (n_bytes / 1e6, secs, secs_big))
""") % REPO


def test_rccl_world1_broadcast_gather_allreduce():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-c", WORKER], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "rccl world-1 ok" in r.stdout, r.stdout[-3000:] + r.stderr[-3000:]
    from tests.helpers import parity_log
    parity_log("[rccl] " + [ln for ln in r.stdout.splitlines() if "rccl world-1 ok" in ln][0])
