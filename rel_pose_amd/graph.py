"""HIP-graph capture of the training step (MI355X-first replacement for "launch 300 kernels from Python per step").

The hot path is ~300 kernel launches per step, many of them 5-20 us (column sums, split-K reduces, LayerNorm partials);
issued eagerly the GPU idles ~15-20 % of the step waiting for the host.  `GraphedTrainStep` captures
  graph A : zero flat grad buffer -> forward -> loss -> backward          (static input buffers)
  [N > 1  : one all-reduce(SUM) of the flat gradient buffer over RCCL, eager, between the graphs; 77 MB over xGMI]
  graph B : grad / world, clip_grad_norm_, Adam(capturable) step
and replays them.  Gradients of all trainable parameters are views into ONE flat buffer, so the data-parallel exchange is a
single collective with no packing copies.  Semantics match the reference trainer (train.py:152-165): same loss weights,
clip value, Adam hyper-parameters; DDP's bucketed overlap is traded for launch-free replay (the exchange is ~1 ms).
BatchNorm running statistics: DDP broadcasts rank 0's buffers before every forward (broadcast_buffers=True, what the
reference runs with); this class bypasses DDP, so it does the same broadcast itself, as one flat collective next to the
gradient exchange -- otherwise the per-rank running means / variances drift apart and a checkpoint saved by rank 0 would not
describe the other replicas.
"""
import torch
import torch.distributed as dist

from . import ops
from .losses import geodesic_loss_tensors
from .se3 import SE3


class GraphedTrainStep:
    def __init__(self, model, optimizer, images, poses, intrinsics, w_tr=10.0, w_rot=10.0, clip=2.5, forward_fn=None):
        self.model, self.opt = model, optimizer
        self.w_tr, self.w_rot, self.clip = w_tr, w_rot, clip
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.images = images.clone()
        self.poses = poses.clone()
        self.intr = intrinsics.clone()
        self._intr_work = intrinsics.clone()
        self.forward_fn = forward_fn or (lambda imgs, Gs, intr: model(imgs, Gs, intrinsics=intr))
        self.params = [p for p in model.parameters() if p.requires_grad]
        n = sum(p.numel() for p in self.params)
        self.flat = torch.zeros(n, device=images.device, dtype=torch.float32)
        o = 0
        for p in self.params:
            # same memory format as the parameter (the CNN's convolution weights are channels-last): autograd then writes the
            # gradient in place instead of converting it ("gradient layout contract")
            chunk = self.flat[o:o + p.numel()]
            p.grad = chunk.as_strided(p.shape, p.stride()) if p.is_contiguous(memory_format=torch.channels_last) and p.dim() == 4 \
                else chunk.view_as(p)
            o += p.numel()
        self.loss = torch.zeros((), device=images.device)
        self.gA = self.gB = None
        self.buffers = [b for b in model.buffers() if b.is_floating_point()]        # BatchNorm running_mean / running_var
        self.bufflat = None
        if self.world > 1 and self.buffers:
            # the buffers become views into ONE flat tensor (like the gradients): rank 0's statistics reach every rank with a single
            # broadcast and no packing / unpacking copies between the two graph replays
            self.bufflat = torch.cat([b.detach().reshape(-1) for b in self.buffers])
            o = 0
            for b in self.buffers:
                b.data = self.bufflat[o:o + b.numel()].view_as(b)
                o += b.numel()

    # -- the two halves, written once and used both eagerly (warm-up) and under capture ---------------------------
    def _fwd_bwd(self):
        self.flat.zero_()
        self._intr_work.copy_(self.intr)                  # forward rescales intrinsics in place (src/model.py:100-109)
        Ps = SE3(self.poses)
        Gs = SE3.IdentityLike(Ps)
        est = self.forward_fn(self.images, Gs, self._intr_work)
        ltr, lrot = geodesic_loss_tensors(Ps, est)
        loss = self.w_tr * ltr + self.w_rot * lrot
        loss.backward()
        self.loss.copy_(loss.detach())

    def _update(self):
        if self.world > 1:
            self.flat.div_(self.world)
        torch.nn.utils.clip_grad_norm_(self.params, self.clip)
        self.opt.step()

    def _exchange(self):
        if self.world > 1:
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM)
            if self.bufflat is not None:                   # rank 0's BatchNorm statistics everywhere, like DDP's buffer broadcast
                dist.broadcast(self.bufflat, 0)

    def capture(self, warmup=3):
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(warmup):                      # MIOpen solver search, lazy inits, allocator warm-up
                self._fwd_bwd()
                self._exchange()
                self._update()
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        self.gA, self.gB = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        ops.invalidate_pad_cache()          # (under capture ops._padded caches nothing: the pads become graph nodes)
        with torch.cuda.graph(self.gA):
            self._fwd_bwd()
        with torch.cuda.graph(self.gB, pool=self.gA.pool()):
            self._update()
        return self

    def step(self, images=None, poses=None, intrinsics=None):
        """One training step; pass new batch tensors (same shapes) or nothing to reuse the resident batch."""
        if images is not None:
            self.images.copy_(images, non_blocking=True)
            self.poses.copy_(poses, non_blocking=True)
            self.intr.copy_(intrinsics, non_blocking=True)
        self.gA.replay()
        self._exchange()
        self.gB.replay()
        ops.invalidate_pad_cache()          # the replayed Adam step rewrote the parameters without bumping Tensor._version
        return self.loss
