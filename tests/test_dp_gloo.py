"""world_size-2 data-parallel path on CPU (gloo): pair sharding + the single gradient all-reduce per step."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    from rel_pose_amd import parallel
    parallel.setup(rank, world, backend="gloo")
    torch.manual_seed(0)
    # a stand-in "pair -> pose" regressor with per-pair independent work (the HIP model needs a GPU)
    model = torch.nn.Sequential(torch.nn.Linear(32, 64), torch.nn.ReLU(), torch.nn.Linear(64, 14))
    ddp = parallel.wrap(model)
    g = torch.Generator().manual_seed(7)
    X, Y = torch.randn(8, 32, generator=g), torch.randn(8, 14, generator=g)     # 8 pairs, global batch
    idx = parallel.shard_pairs(8, rank, world)
    loss = (ddp(X[idx]) - Y[idx]).square().mean()
    loss.backward()
    grads = torch.cat([p.grad.reshape(-1) for p in model.parameters()])
    # explicit all-reduce helper gives the same mean
    local = [torch.full((3,), float(rank + 1)), torch.full((2, 2), float(10 * (rank + 1)))]
    parallel.allreduce_mean_(local)
    q.put((rank, idx, grads.numpy().copy(), [t.numpy().copy() for t in local]))   # by value, not shared memory
    dist.barrier()
    parallel.cleanup()


@pytest.mark.timeout(120)
def test_two_rank_data_parallel_matches_single_process():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get() for _ in range(world)], key=lambda t: t[0])
    res = [(r, i, torch.from_numpy(g_), [torch.from_numpy(t) for t in loc]) for r, i, g_, loc in res]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert res[0][1] == [0, 2, 4, 6] and res[1][1] == [1, 3, 5, 7]                # rank r takes pairs r::W
    assert torch.allclose(res[0][2], res[1][2], atol=0)                             # identical averaged grads
    torch.manual_seed(0)
    model = torch.nn.Sequential(torch.nn.Linear(32, 64), torch.nn.ReLU(), torch.nn.Linear(64, 14))
    g = torch.Generator().manual_seed(7)
    X, Y = torch.randn(8, 32, generator=g), torch.randn(8, 14, generator=g)
    (model(X) - Y).square().mean().backward()
    ref = torch.cat([p.grad.reshape(-1) for p in model.parameters()])
    assert torch.allclose(res[0][2], ref, atol=1e-6)                                # DP grad == full-batch grad
    assert torch.allclose(res[0][3][0], torch.full((3,), 1.5)) and torch.allclose(res[1][3][1], torch.full((2, 2), 15.0))


def _worker8(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    from rel_pose_amd import parallel
    parallel.setup(rank, world, backend="gloo")
    torch.set_num_threads(1)
    torch.manual_seed(0)
    model = torch.nn.Sequential(torch.nn.Linear(32, 64), torch.nn.ReLU(), torch.nn.Linear(64, 14))
    ddp = parallel.wrap(model)
    g = torch.Generator().manual_seed(7)
    X, Y = torch.randn(16, 32, generator=g), torch.randn(16, 14, generator=g)       # 16 pairs, global batch
    idx = parallel.shard_pairs(16, rank, world)
    (ddp(X[idx]) - Y[idx]).square().mean().backward()
    grads = torch.cat([p.grad.reshape(-1) for p in model.parameters()])
    local = [torch.full((3,), float(rank + 1)), torch.full((2, 2), float(10 * (rank + 1)))]
    parallel.allreduce_mean_(local)
    ragged = parallel.shard_pairs(12, rank, world)                                   # 12 pairs on 8 ranks: the tail wraps
    worst, per_rank = parallel.gather_step_times(0.010 * (rank + 1), 10)             # bench.py's `distributed` record
    q.put((rank, idx, grads.numpy().copy(), [t.numpy().copy() for t in local], ragged, worst, per_rank))
    dist.barrier()
    parallel.cleanup()


@pytest.mark.timeout(300)
def test_eight_rank_data_parallel_matches_single_process():
    """The shape the first 8-GPU lease will run (BASELINE configs[3], configs[4]): world size 8 -- sharding r::8, one gradient
    all-reduce, the explicit bucketed mean all-reduce, and the per-rank timing record of bench.py."""
    world, port = 8, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    procs = [ctx.Process(target=_worker8, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get() for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    for r, idx, g_, loc, ragged, worst, per_rank in res:
        assert idx == [r, r + 8]                                                    # rank r takes pairs r::8
        assert ragged == [r, (r + 8) % 12]                                          # same count on every rank, wrapped tail
        assert abs(worst - 0.080) < 1e-12 and per_rank == [float(i + 1) for i in range(8)]      # max over ranks, rank order
        assert torch.allclose(torch.from_numpy(loc[0]), torch.full((3,), 4.5)) and torch.allclose(torch.from_numpy(loc[1]), torch.full((2, 2), 45.0))
        assert torch.equal(torch.from_numpy(g_), torch.from_numpy(res[0][2]))        # identical averaged gradients on every rank
    covered = sorted(i for r in res for i in r[1])
    assert covered == list(range(16))                                               # every pair exactly once
    torch.manual_seed(0)
    model = torch.nn.Sequential(torch.nn.Linear(32, 64), torch.nn.ReLU(), torch.nn.Linear(64, 14))
    g = torch.Generator().manual_seed(7)
    X, Y = torch.randn(16, 32, generator=g), torch.randn(16, 14, generator=g)
    (model(X) - Y).square().mean().backward()
    ref = torch.cat([p.grad.reshape(-1) for p in model.parameters()])
    assert torch.allclose(torch.from_numpy(res[0][2]), ref, atol=1e-6)               # DP gradient == full-batch gradient


def _spawn_body(tmpdir):
    """what train.run does first: read the launcher environment, join the group, one collective"""
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    assert int(os.environ["LOCAL_RANK"]) == rank and os.environ["MASTER_ADDR"] == "127.0.0.1"
    from rel_pose_amd import parallel
    parallel.setup(rank, world, backend="gloo")
    t = torch.tensor([float(rank + 1)])
    dist.all_reduce(t)
    with open(os.path.join(tmpdir, "rank%d.txt" % rank), "w") as f:
        f.write("%d %d %.1f" % (rank, world, t.item()))
    parallel.cleanup()


@pytest.mark.timeout(120)
def test_spawn_starts_one_rank_per_gpu_like_the_reference(tmp_path):
    """`train.py --gpus N` without a launcher spawns N ranks itself (reference train.py:286-291)."""
    from rel_pose_amd import parallel
    parallel.spawn(_spawn_body, 2, (str(tmp_path),), master_port=_free_port())
    got = sorted(open(str(tmp_path / ("rank%d.txt" % r))).read() for r in range(2))
    assert got == ["0 2 3.0", "1 2 3.0"]


def test_loader_workers_do_not_oversubscribe_the_host():
    """Every rank of a node runs its own DataLoader worker pool (train.py): the per-rank worker count is capped so that all ranks
    together stay within the cores this process may use."""
    from rel_pose_amd import parallel
    try:
        cores = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        cores = os.cpu_count() or 1
    for world in (1, 2, 8):
        nw = parallel.loader_workers(64, world)
        assert 0 <= nw <= 64 and nw * world + world <= max(cores, world)          # workers + one main thread per rank
        assert parallel.loader_workers(0, world) == 0
    assert parallel.loader_workers(1, 1) == min(1, max(cores - 1, 0))
    assert parallel.loader_workers(4, 10 ** 6) == 0                                 # more ranks than cores: load in the main process


def test_ddp_wrapper_uses_bucket_views_and_the_documented_cap():
    from rel_pose_amd import parallel
    import inspect
    src = inspect.getsource(parallel.wrap)
    assert "gradient_as_bucket_view=True" in src and "bucket_cap_mb=BUCKET_CAP_MB" in src and parallel.BUCKET_CAP_MB == 8
