"""Multi-GPU sampling: one process per GPU, prompts sharded by rank, one all_gather of the finished latents.

The reference has no inference-time parallelism (scripts/txt2img.py is single process); each
prompt's trajectory is independent (no cross-sample op in the UNet), so the path shards with no data-path
collective.  The only exchange is the final gather (RCCL over xGMI on GPUs: backend "nccl"; "gloo" in CPU tests),
64 KiB per image -- latency bound, one call per batch.
"""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Join the process group described by RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT (torchrun contract)."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', str(rank)))
    # A launcher (torchrun) exports RANK / WORLD_SIZE also for ONE process: the group is then initialised as well (a
    # world of size 1 is legal), so that `torchrun --nproc-per-node 1 bench.py` exercises exactly the code path of N > 1 --
    # RCCL communicator setup, the all_gather of the latents, the all_reduce of the timing -- on a single GPU.
    launched = 'RANK' in os.environ and 'WORLD_SIZE' in os.environ
    if (world > 1 or launched) and not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29500')
        if backend is None:
            backend = 'nccl' if torch.cuda.is_available() else 'gloo'
        kw = {}
        if backend == 'nccl':
            # one process per GPU: bind the device BEFORE the communicator exists and tell the process group which one it is --
            # without device_id RCCL "guesses the device from the global rank" (its own warning), which is wrong as soon as ranks
            # and local devices are numbered differently
            n_dev = torch.cuda.device_count()
            if not 0 <= local_rank < n_dev:
                raise RuntimeError(f'LOCAL_RANK={local_rank} but this process sees {n_dev} GPU(s): launch one process per visible GPU')
            torch.cuda.set_device(local_rank)
            kw['device_id'] = torch.device('cuda', local_rank)
        if 'device_id' in kw:
            # decided up front from the signature: a TypeError raised INSIDE a torch that accepts the keyword must
            # propagate instead of silently restoring "RCCL guesses the device from the global rank"
            import inspect
            if 'device_id' not in inspect.signature(dist.init_process_group).parameters:
                kw.pop('device_id')
        dist.init_process_group(backend=backend, rank=rank, world_size=world, **kw)
        if backend == 'nccl' and torch.cuda.current_device() != local_rank:
            raise RuntimeError(f'rank {rank}: current device {torch.cuda.current_device()} != LOCAL_RANK {local_rank}')
        if dist.get_world_size() != world or dist.get_rank() != rank:
            raise RuntimeError(f'process group says rank {dist.get_rank()} / {dist.get_world_size()}, the environment {rank} / {world}')
    return rank, world, local_rank


def ranks_seen(device=None):
    """How many ranks actually took part: an all_reduce(SUM) of ones over the live communicator (what a SCALE record can be checked
    against, next to RCCL's own `nranks`); 1 without a process group."""
    if not dist.is_initialized():
        return 1
    t = torch.ones(1, dtype=torch.int64, device=device if device is not None else 'cpu')
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return int(t.item())


def shard(items, rank, world):
    """Rank r takes items r, r+W, r+2W, ... (global index kept so seeds do not depend on W)."""
    return [(i, it) for i, it in enumerate(items) if i % world == rank]


def gather_latents(local, n_total, rank=None, world=None):
    """local: [n_local, ...] latents of this rank's shard (global indices rank, rank+W, ...).
    Returns [n_total, ...] in global prompt order on every rank (one all_gather)."""
    if world is None:
        world = dist.get_world_size() if dist.is_initialized() else 1
    if rank is None:
        rank = dist.get_rank() if dist.is_initialized() else 0
    if world == 1 and not dist.is_initialized():
        return local
    n_max = (n_total + world - 1) // world
    pad = torch.zeros((n_max,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[:local.shape[0]] = local
    bufs = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(bufs, pad)
    out = torch.empty((n_total,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    for r in range(world):
        idx = list(range(r, n_total, world))
        if idx:
            out[idx] = bufs[r][:len(idx)]
    return out


def max_over_ranks(value, device):
    """Scalar max across ranks (used for the timed region of bench.py)."""
    if not dist.is_initialized():
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
