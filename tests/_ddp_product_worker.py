"""Worker of tests/test_gpu_entrypoints.py::test_product_model_ddp_gradients_equal_full_batch (one process per rank)."""
import os
import sys
import types

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main(out_path):
    from oracle import relpose_oracle as O
    from rel_pose_amd import parallel
    from rel_pose_amd.model import ViTEss
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)                          # both ranks share the one GPU of the test box -> gloo, not RCCL
    parallel.setup(rank, world, backend="gloo")
    args = types.SimpleNamespace(noess="", pool_size=60, fc_hidden_size=512, fusion_transformer=True, transformer_depth=6,
                                 cross_features=False, use_single_softmax=False, no_pos_encoding=False, l1_pos_encoding=False)
    shapes = dict(O.vit_param_shapes())
    shapes.update(O.cnn_param_shapes())
    sd32 = O.make_state(shapes, torch.float32)

    def fresh():
        m = ViTEss(args)
        m.load_state_dict(sd32, strict=True)
        m = m.cuda().eval()              # eval-mode BatchNorm: batch statistics would differ between a shard and the full batch
        for p in list(m.resnet.layer3.parameters()) + list(m.resnet.layer4.parameters()):
            p.requires_grad = False      # reference train.py:60-64 (DDP needs every trainable parameter to receive a gradient)
        return m

    B = 8
    g = torch.Generator().manual_seed(5)
    imgs = torch.floor(torch.rand(B, 2, 3, 256, 320, generator=g) * 255.0)
    Gs = torch.tensor([0, 0, 0, 0, 0, 0, 1.0]).repeat(B, 2, 1)
    intr = torch.tensor([[300.0, 290.0, 160.0, 128.0]]).repeat(B, 2, 1).contiguous()
    cot = O.closed_form((B, 2, 7), 909, 1.0)
    idx = parallel.shard_pairs(B, rank, world)                  # rank r owns pairs r::W (DistributedSampler order)

    net = parallel.wrap(fresh(), [0])
    from rel_pose_amd.se3 import SE3
    est = net(imgs[idx].cuda(), SE3(Gs[idx].cuda()), intrinsics=intr[idx].clone().cuda())[0].data
    (est * cot[idx].cuda()).sum().div(len(idx)).backward()
    ddp_grads = {n: p.grad.detach().clone() for n, p in net.module.named_parameters() if p.grad is not None}

    if rank == 0:
        ref = fresh()
        est = ref(imgs.cuda(), SE3(Gs.cuda()), intrinsics=intr.clone().cuda())[0].data
        (est * cot.cuda()).sum().div(B).backward()
        worst, name_w, worst_hot = 0.0, "", 0.0
        for n, p in ref.named_parameters():
            if p.grad is None:
                continue
            assert n in ddp_grads, n
            e = float((ddp_grads[n] - p.grad).abs().max() / p.grad.abs().max().clamp_min(1e-30))
            if e > worst:
                worst, name_w = e, n
            if n.startswith("fusion_transformer") or n.startswith("pose_regressor"):
                worst_hot = max(worst_hot, e)
        with open(out_path, "w") as f:
            f.write("%d %.6e %s %.6e\n" % (len(ddp_grads), worst, name_w, worst_hot))
    dist.barrier()
    parallel.cleanup()


if __name__ == "__main__":
    main(sys.argv[1])
