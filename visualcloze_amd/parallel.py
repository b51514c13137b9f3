"""Data-parallel sharding of independent grids over the GPUs of one node.

The reference has no multi-GPU inference (`sample.py:258` asserts one GPU); each grid is an independent ODE
solve, so the path shards by sample with NO collective inside a solver step: one process per GPU,
`torch.distributed` backend "nccl" (= RCCL over xGMI on ROCm; "gloo" for the CPU tests), rank 0 owns the
checkpoint and broadcasts the frozen weights once, every rank then samples its own grids with seeds that
depend on the GLOBAL sample index (results independent of world size), and only final latents are gathered.
"""
from __future__ import annotations

import os
import time
from typing import List, Optional, Sequence

import torch
import torch.distributed as dist


def env_rank_world():
    return int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))


def init_distributed(backend: Optional[str] = None, device: Optional[torch.device] = None, force: bool = False) -> None:
    """Rendezvous from MASTER_ADDR/MASTER_PORT (use 127.0.0.1 on one node).  A single-process job skips the process group
    unless `force` (the one-GPU RCCL smoke test creates a world-1 communicator on purpose)."""
    if dist.is_initialized() or (int(os.environ.get("WORLD_SIZE", 1)) == 1 and not force):
        return
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC only on this driver
    backend = backend or ("nccl" if torch.cuda.is_available() else "gloo")
    kw = {}
    if backend == "nccl" and device is not None:
        kw["device_id"] = device
    dist.init_process_group(backend, **kw)


def pin_to_gpu_numa(local_rank: int) -> Optional[int]:
    """Bind this process (and the threads it starts) to the CPUs of the NUMA node its GPU hangs off: eight ranks that
    launch one graph per step each must not migrate across sockets or share the cores next to another rank's GPU
    (SURVEY.md §8e hazard list).  PCI address from the device properties -> /sys/bus/pci/devices/<addr>/numa_node ->
    /sys/devices/system/node/node<N>/cpulist.  Returns the node, or None when the platform does not say (single-node
    hosts report -1) - then nothing is changed."""
    try:
        pr = torch.cuda.get_device_properties(local_rank)
        addr = f"{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}.0"
        node = int(open(f"/sys/bus/pci/devices/{addr}/numa_node").read().strip())
        if node < 0:
            return None
        cpus = parse_cpulist(open(f"/sys/devices/system/node/node{node}/cpulist").read())
        allowed = sorted(set(cpus) & set(os.sched_getaffinity(0)))
        if not allowed:
            return None
        os.sched_setaffinity(0, allowed)
        return node
    except Exception:
        return None


def parse_cpulist(text: str) -> List[int]:
    """"0-3,8,10-11" -> [0,1,2,3,8,10,11] (the kernel's cpulist format)."""
    out: List[int] = []
    for part in text.strip().split(","):
        if not part:
            continue
        lo, _, hi = part.partition("-")
        out.extend(range(int(lo), int(hi or lo) + 1))
    return out


def world() -> int:
    return dist.get_world_size() if dist.is_initialized() else 1


def rank() -> int:
    return dist.get_rank() if dist.is_initialized() else 0


def broadcast_weights(module: torch.nn.Module, src: int = 0, bucket_bytes: int = 1 << 30, force: bool = False) -> float:
    """One-time broadcast of every parameter from `src` (the frozen FLUX + LoRA weights, 26.3 GB bf16 at full
    size).  Parameters are coalesced into ~1 GiB flat buckets: xGMI is point-to-point, so a few large
    transfers per link beat a thousand small ones.  Returns the seconds spent.  `force`: run the collectives even in a
    world of one (exercises communicator creation and the bucket code on one GPU)."""
    if world() == 1 and not (force and dist.is_initialized()):
        return 0.0
    for m in module.modules():
        if getattr(m, "_params_freed", False):
            raise RuntimeError("broadcast_weights: this module's parameters were released by prepare(free_parameters=True) - "
                               "broadcast BEFORE preparing a sampling-only rank")
    t0 = time.time()
    params = [p.data for p in module.parameters()]
    i = 0
    while i < len(params):
        j, size = i, 0
        while j < len(params) and (j == i or size + params[j].numel() * params[j].element_size() <= bucket_bytes) \
                and params[j].dtype == params[i].dtype:
            size += params[j].numel() * params[j].element_size()
            j += 1
        if j - i == 1:
            dist.broadcast(params[i], src=src)
        else:
            flat = torch.cat([p.reshape(-1) for p in params[i:j]])
            dist.broadcast(flat, src=src)
            o = 0
            for p in params[i:j]:
                p.copy_(flat[o:o + p.numel()].view_as(p))
                o += p.numel()
            del flat
        i = j
    # dist.broadcast / copy_ through `.data` do not bump `Parameter._version`, which the engines' weight
    # fingerprints key on: drop every prepared (LoRA-merged) copy explicitly so it is rebuilt from the new weights.
    for m in module.modules():
        inv = getattr(m, "invalidate_engine", None)
        if callable(inv):
            inv()
    if torch.cuda.is_available():
        torch.cuda.synchronize()
    return time.time() - t0


def shard_indices(n_samples: int, r: Optional[int] = None, w: Optional[int] = None) -> List[int]:
    """Sample i runs on rank i mod world."""
    r = rank() if r is None else r
    w = world() if w is None else w
    return list(range(r, n_samples, w))


def sample_seed(base_seed: int, global_index: int) -> int:
    """Per-sample seed from the GLOBAL index, so a sample's noise does not depend on the world size."""
    return int(base_seed) + int(global_index)


def max_over_ranks(seconds: float, device: Optional[torch.device] = None, force: bool = False) -> float:
    if world() == 1 and not (force and dist.is_initialized()):
        return seconds
    t = torch.tensor([seconds], dtype=torch.float64, device=device or "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def rccl_rank_count(device: Optional[torch.device] = None) -> int:
    """How many ranks RCCL ITSELF has seen: the sum of a device-tensor all-reduce of ones over the "nccl" (= RCCL on ROCm)
    communicator - 0 when there is no process group or its backend is not RCCL (gloo test worlds, a bare single-GPU run).
    bench.py reports it as `rccl_ranks`: a record with n_gpus = 8 and rccl_ranks != 8 did not run over RCCL."""
    if not dist.is_initialized() or dist.get_backend() != "nccl":
        return 0
    dev = device or torch.device("cuda", torch.cuda.current_device())
    one = torch.ones(1, dtype=torch.int32, device=dev)
    dist.all_reduce(one, op=dist.ReduceOp.SUM)
    return int(one.item())


def gather_records(record: dict) -> Optional[List[dict]]:
    """Every rank's small dict on rank 0, in rank order (None elsewhere); a world of one returns [record]."""
    if world() == 1:
        return [record]
    objs = [None] * world() if rank() == 0 else None
    dist.gather_object(record, objs, dst=0)
    return objs


def gather_latents(local: Sequence[torch.Tensor], n_samples: int, force: bool = False,
                   to_host: bool = True) -> Optional[List[torch.Tensor]]:
    """Collect the final latents (0.44 MB per sample at cfg 2) on rank 0 in global sample order.  `to_host` (default):
    HOST tensors whatever the world size - one contract for the caller that saves or post-processes on the CPU; with
    `to_host=False` the tensors stay where the collective leaves them (the caller's own device tensors in a world of one -
    no D2H sync -, rank 0's GPU under RCCL, the host under gloo).  Every sample of the job normally has one shape and dtype - then ONE padded tensor collective moves
    them (no pickling; RCCL gathers device tensors, gloo host tensors); a job that mixes grid sizes or dtypes falls back to
    `gather_object`."""
    if any(t.shape != local[0].shape or t.dtype != local[0].dtype for t in local[1:]):
        uniform_here = 0
    else:
        uniform_here = 1
    if world() == 1 and not (force and dist.is_initialized()):
        return [t.detach().cpu() for t in local] if to_host else [t.detach() for t in local]
    w, r = world(), rank()
    slots = (n_samples + w - 1) // w
    on_gpu = dist.get_backend() == "nccl"
    dev = torch.device("cuda", torch.cuda.current_device()) if on_gpu else torch.device("cpu")
    ref = local[0] if local else None
    dtypes = [torch.float32, torch.bfloat16, torch.float16, torch.float64]
    # (ndim, shape[8], dtype) per rank, reduced with MAX and, negated, with MIN: equal on every rank that has a sample
    # <=> MAX == -MAX(-x) over those ranks; ranks without a sample contribute the neutral element to both
    BIG = 1 << 40
    hi = torch.full((11,), -BIG, dtype=torch.int64)
    lo = torch.full((11,), -BIG, dtype=torch.int64)
    if ref is not None and ref.dtype in dtypes and ref.dim() <= 8:
        v = torch.zeros(11, dtype=torch.int64)
        v[0] = ref.dim()
        v[1:1 + ref.dim()] = torch.tensor(ref.shape, dtype=torch.int64)
        v[9] = dtypes.index(ref.dtype)
        v[10] = uniform_here
        hi, lo = v.clone(), -v
    elif ref is not None:
        hi[10], lo[10] = 0, 0                            # a dtype / rank the tensor path does not cover
    meta = torch.stack((hi, lo)).to(dev)
    dist.all_reduce(meta, op=dist.ReduceOp.MAX)         # ranks without a sample learn the shape
    hi, lo = meta[0].cpu(), -meta[1].cpu()
    uniform = bool((hi[:10] == lo[:10]).all()) and int(lo[10]) == 1 and int(hi[0]) >= 0
    if not uniform:
        objs = [None] * w if r == 0 else None
        dist.gather_object([t.detach().cpu() for t in local], objs, dst=0)
        if r != 0:
            return None
        out: List[Optional[torch.Tensor]] = [None] * n_samples
        for rr in range(w):
            for i, k in enumerate(shard_indices(n_samples, rr, w)):
                out[k] = objs[rr][i]
        return out  # type: ignore[return-value]
    shp = [int(v) for v in hi[1:1 + int(hi[0])].tolist()]
    dtype = dtypes[int(hi[9])]
    buf = torch.zeros(slots, *shp, dtype=dtype, device=dev)
    for i, t in enumerate(local):
        buf[i].copy_(t)
    parts = [torch.empty_like(buf) for _ in range(w)] if r == 0 else None
    dist.gather(buf, parts, dst=0)
    if r != 0:
        return None
    out = [None] * n_samples
    for rr in range(w):
        for i, k in enumerate(shard_indices(n_samples, rr, w)):
            out[k] = parts[rr][i].cpu() if to_host else parts[rr][i]
    return out  # type: ignore[return-value]


def barrier() -> None:
    if torch.cuda.is_available():
        torch.cuda.synchronize()
    if world() > 1:
        dist.barrier()
    if torch.cuda.is_available():
        torch.cuda.synchronize()
