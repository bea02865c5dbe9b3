"""CPU, world_size 2 over gloo: prompt sharding and the single latent all_gather of the multi-GPU path."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_total, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from stable_diffusion_amd import dist as sd_dist
    r, w, _ = sd_dist.init_from_env(backend='gloo')
    assert (r, w) == (rank, world)
    mine = sd_dist.shard(list(range(n_total)), r, w)
    # "latent" of prompt i is a tensor filled with i (seeded by the GLOBAL index, independent of world size)
    local = torch.stack([torch.full((4, 8, 8), float(i)) for i, _ in mine]) if mine else torch.zeros((0, 4, 8, 8))
    full = sd_dist.gather_latents(local, n_total, r, w)
    ok = full.shape == (n_total, 4, 8, 8) and all(float(full[i].mean()) == float(i) for i in range(n_total))
    mx = sd_dist.max_over_ranks(10.0 + rank, torch.device('cpu'))
    ok = ok and sd_dist.ranks_seen(torch.device('cpu')) == world        # what bench.py reports as `ranks_seen`
    q.put((rank, bool(ok), mx))
    dist.destroy_process_group()


@pytest.mark.parametrize('n_total', [8, 5, 1])
def test_shard_and_gather_world2(n_total):
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_total, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in res)
    assert all(mx == 11.0 for _, _, mx in res)


def test_single_process_passthrough():
    from stable_diffusion_amd import dist as sd_dist
    x = torch.randn(3, 4, 8, 8)
    assert sd_dist.gather_latents(x, 3, 0, 1) is x
    assert sd_dist.shard(['a', 'b', 'c'], 0, 1) == [(0, 'a'), (1, 'b'), (2, 'c')]
    assert sd_dist.shard(['a', 'b', 'c'], 1, 2) == [(1, 'b')]


def _world1_worker(port, q):
    """A launcher exports RANK / WORLD_SIZE also for one process (torchrun --nproc-per-node 1): the group is initialised and
    the gather / timing reduce go through the collectives -- the code path the 1-GPU RCCL run of bench.py takes."""
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK='0', WORLD_SIZE='1', LOCAL_RANK='0')
    import bench
    from stable_diffusion_amd import dist as sd_dist
    r, w, _ = sd_dist.init_from_env(backend='gloo')
    inited = dist.is_initialized() and dist.get_world_size() == 1
    x = torch.arange(3 * 4 * 8 * 8, dtype=torch.float32).reshape(3, 4, 8, 8)
    full = sd_dist.gather_latents(x, 3, r, w)
    went_through_collective = full is not x and torch.equal(full, x)
    mx = sd_dist.max_over_ranks(7.5, torch.device('cpu'))
    elapsed, lat, img, allz = bench.timed_steps(lambda s: (torch.full((1, 4, 8, 8), float(s)), torch.zeros(1, 3, 8, 8)), 2, 1, w, r,
                                                torch.device('cpu'))
    ok = inited and went_through_collective and mx == 7.5 and allz.shape == (1, 4, 8, 8) and float(allz.mean()) == 1.0
    q.put(bool(ok))
    dist.destroy_process_group()


def test_launched_world_of_one_initialises_the_group():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    p = ctx.Process(target=_world1_worker, args=(_free_port(), q))
    p.start()
    assert q.get(timeout=180) is True
    p.join(timeout=60)
    assert p.exitcode == 0


def _bench_worker(rank, world, port, q):
    """bench.py's timed region (barrier-bracketed steps, one latent all_gather per step, max-over-ranks time) with a fake
    per-step workload: rank 1 is slower, so the reported time must be rank 1's on both ranks."""
    import time
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    import bench
    from stable_diffusion_amd import dist as sd_dist
    r, w, _ = sd_dist.init_from_env(backend='gloo')
    calls = []

    def step(s):
        calls.append(s)
        time.sleep(0.05 * (1 + 3 * rank))
        gidx = s * w + r                                   # global prompt index, as bench.py seeds its inputs
        return torch.full((1, 4, 8, 8), float(gidx)), torch.zeros(1, 3, 8, 8)
    elapsed, lat, img, allz = bench.timed_steps(step, 3, 2, w, r, torch.device('cpu'))
    ok = calls == [-1, -2, 0, 1, 2] and allz.shape == (w, 4, 8, 8) and [float(allz[i].mean()) for i in range(w)] == [4.0, 5.0]
    q.put((rank, bool(ok), elapsed))
    dist.destroy_process_group()


def test_bench_timed_region_world2():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_bench_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in res)
    assert res[0][2] == res[1][2] and 0.55 <= res[0][2] < 2.0         # 3 steps x 0.2 s on the slow rank, seen by both
