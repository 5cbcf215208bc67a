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


def wrap(model, device_ids=None):
    """DDP with the reference's settings (train.py:66-67)."""
    return torch.nn.parallel.DistributedDataParallel(model, device_ids=device_ids, find_unused_parameters=False)


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
