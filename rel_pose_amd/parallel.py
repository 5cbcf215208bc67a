"""Batch-of-pairs data parallelism (the only parallelism in rel_pose: reference train.py:28-36,66-67,128-130).

One process per GPU; pairs are independent units, so the forward needs no collective and the backward needs exactly one
exchange: the gradient all-reduce(mean) of the 19.26 M trainable fp32 values (77 MB), issued by
torch.nn.parallel.DistributedDataParallel in buckets that overlap the backward.  Backend "nccl" is RCCL on ROCm
(xGMI between the 8 GPUs of a node); "gloo" is used for the CPU tests.
"""
import os

import torch
import torch.distributed as dist


def setup(rank, world_size, backend=None, master_addr="127.0.0.1", master_port="12356"):
    os.environ.setdefault("MASTER_ADDR", master_addr)
    os.environ.setdefault("MASTER_PORT", str(master_port))
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    dist.init_process_group(backend=backend, init_method="env://", world_size=world_size, rank=rank)
    torch.manual_seed(0)                                            # reference train.py:35
    if backend == "nccl":
        torch.cuda.set_device(rank % torch.cuda.device_count())
    return backend


def _spawn_entry(local_rank, fn, world_size, args, master_addr, master_port):
    os.environ["MASTER_ADDR"] = master_addr
    os.environ["MASTER_PORT"] = str(master_port)
    os.environ["WORLD_SIZE"] = str(world_size)
    os.environ["RANK"] = str(local_rank)
    os.environ["LOCAL_RANK"] = str(local_rank)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # the host driver only supports dmabuf IPC (RCCL needs it)
    fn(*args)


def spawn(fn, nprocs, args=(), master_addr="127.0.0.1", master_port="12356"):
    """`python train.py --gpus N` without a launcher: start N ranks of `fn(*args)` on this node, one per GPU, exactly as the
    reference does with mp.spawn (train.py:286-291; rendezvous port 12356).  Each child gets the RANK / LOCAL_RANK /
    WORLD_SIZE / MASTER_* environment a launcher would have set, so `fn` is the same code path either way."""
    import torch.multiprocessing as mp
    mp.spawn(_spawn_entry, nprocs=nprocs, args=(fn, nprocs, tuple(args), master_addr, str(master_port)), join=True)


def shard_pairs(num_pairs, rank, world_size):
    """Indices of the pairs rank `rank` owns: r, r+W, r+2W, ... (what DistributedSampler(shuffle=False) yields;
    the tail is padded by wrapping so every rank gets the same count, as DistributedSampler does)."""
    per = -(-num_pairs // world_size)
    idx = [(rank + i * world_size) % num_pairs for i in range(per)]
    return idx


# Gradient buckets.  The backward of this model hands gradients over at 8 points only (one autograd Function per module: head, CrossBlock,
# 5 Blocks, then the CNN's convolutions one by one), in this order and size: pose_regressor 56.1 MB (ready first, ~1 ms into the
# backward), the ViT + EMM 11.1 MB (ready over the next ~12 ms), the CNN front-end 9.8 MB (ready last; resnet.conv1's 37 KB at the very
# end).  DDP fills buckets in reverse registration order ~ this order; the 56 MB tensor is a bucket of its own whatever the cap, so its
# all-reduce (the bulk of the 77 MB) overlaps ~20 ms of backward.  What is EXPOSED is the last bucket, which cannot start before the
# stem's weight gradient exists: with the default 25 MB cap that is ~21 MB (ViT + CNN together), with 8 MB it is the front-end's last
# ~6 MB -- per-link ring time 2 * 7/8 * 6 MB / ~45 GB/s effective per xGMI link ~ 0.25 ms instead of ~0.8 ms.  UNMEASURED on hardware
# (no multi-GPU node has been available to this build); the reasoning is the 7 x ~153 GB/s point-to-point links of
# MI355X_MICROARCH.md, not a measurement.  gradient_as_bucket_view: .grad tensors are views into the buckets, so the 77 MB
# copy-in / copy-out around the collective disappears.
BUCKET_CAP_MB = int(os.environ.get("RP_DDP_BUCKET_MB", "8"))


def wrap(model, device_ids=None):
    """DDP with the reference's settings (train.py:66-67: find_unused_parameters=False, buffers broadcast from rank 0 every forward)
    plus bucket views and the bucket cap reasoned above."""
    return torch.nn.parallel.DistributedDataParallel(model, device_ids=device_ids, find_unused_parameters=False,
                                                     gradient_as_bucket_view=True, bucket_cap_mb=BUCKET_CAP_MB)


def loader_workers(requested, local_world, reserve=1):
    """DataLoader workers PER RANK that do not oversubscribe the host: every rank of a node runs its own worker pool, and a decode
    worker keeps a core busy (profiles/r2_loader_bench.txt: 125 pairs/s per core, ~14 cores feed one GPU).  Returns
    min(requested, (usable cores // ranks on this node) - reserve) (>= 0), where usable cores = the affinity mask of this process (the
    cgroup / taskset view), not the machine's core count."""
    try:
        cores = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        cores = os.cpu_count() or 1
    per_rank = max(cores // max(int(local_world), 1) - reserve, 0)
    return max(min(int(requested), per_rank), 0)


def allreduce_mean_(tensors):
    """Explicit bucketed mean all-reduce (used when DDP is bypassed, e.g. under graph capture)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return
    flat = torch.cat([t.reshape(-1) for t in tensors])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    flat /= dist.get_world_size()
    o = 0
    for t in tensors:
        n = t.numel()
        t.copy_(flat[o:o + n].view_as(t))
        o += n


def gather_step_times(elapsed_s, steps, device="cpu"):
    """bench.py's max-over-ranks timing and its self-diagnosing record: (slowest rank's seconds for the timed steps, every rank's own
    ms per step in rank order).  One all_gather + one all-reduce(MAX) of a float64, on `device` (the GPU under RCCL, the CPU under gloo)."""
    own = torch.tensor([float(elapsed_s)], device=device, dtype=torch.float64)
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return float(elapsed_s), [round(1e3 * float(elapsed_s) / steps, 3)]
    every = [torch.zeros_like(own) for _ in range(dist.get_world_size())]
    dist.all_gather(every, own)
    worst = own.clone()
    dist.all_reduce(worst, op=dist.ReduceOp.MAX)
    return float(worst.item()), [round(1e3 * float(t.item()) / steps, 3) for t in every]


def cleanup():
    if dist.is_initialized():
        dist.destroy_process_group()


def clip_grad_norm_(parameters, max_norm):
    """torch.nn.utils.clip_grad_norm_(parameters, max_norm) (reference train.py:158) for single-device fp32 gradients, same
    arithmetic -- per-tensor 2-norms (one multi-tensor kernel), norm of the norms, coefficient clamp(max_norm / (total + 1e-6), max = 1),
    one multi-tensor scale -- without the per-tensor `.to(device)` calls of the stock helper (87 host-side calls per step here,
    visible once the step is launch-bound: --batch 6).  Returns the total norm."""
    import torch
    grads = [p.grad for p in parameters if p.grad is not None]
    if not grads:
        return torch.zeros(())
    total = torch.linalg.vector_norm(torch.stack(torch._foreach_norm(grads, 2.0)), 2.0)
    torch._foreach_mul_(grads, torch.clamp(max_norm / (total + 1e-6), max=1.0))
    return total
