import copy, types, torch, sys
sys.path.insert(0, "/root/repo")
from rel_pose_amd.graph import GraphedTrainStep
from rel_pose_amd.model import ViTEss
torch.manual_seed(0)
args = types.SimpleNamespace(fusion_transformer=True, transformer_depth=6, fc_hidden_size=512, cross_features=False, use_single_softmax=False, no_pos_encoding=False, l1_pos_encoding=False, noess=False, pool_size=60)
net0 = ViTEss(args).cuda().train()
B = 2
images = torch.floor(torch.rand(B, 2, 3, 384, 384, device="cuda") * 255.0)
q = torch.nn.functional.normalize(torch.randn(B, 4, device="cuda"), dim=1)
poses = torch.zeros(B, 2, 7, device="cuda"); poses[:, 0, 6] = 1.0; poses[:, 1, :3] = torch.rand(B, 3, device="cuda") - 0.5; poses[:, 1, 3:] = q * torch.sign(q[:, 3:4])
intr = torch.tensor([[192.0] * 4], device="cuda").repeat(B, 2, 1)
for fused in (False, True):
    res = {}
    for mode in ("eager", "graph"):
        net = copy.deepcopy(net0)
        opt = torch.optim.Adam([p for p in net.parameters() if p.requires_grad], lr=1e-4, capturable=True, fused=fused)
        gs = GraphedTrainStep(net, opt, images, poses, intr)
        out = []
        if mode == "graph":
            gs.capture(warmup=2)
            w_after_warm = [p.detach().clone() for p in gs.params]
            out.append(float(gs.step()))
        else:
            for i in range(3):
                gs._fwd_bwd(); gs._exchange()
                if i == 2: break
                gs._update()
            w_after_warm = [p.detach().clone() for p in gs.params]
            out.append(float(gs.loss))
        res[mode] = (out, w_after_warm, gs.flat.clone())
    dw = max(float((a - b).abs().max()) for a, b in zip(res["eager"][1], res["graph"][1]))
    dg = float((res["eager"][2] - res["graph"][2]).abs().max() / res["eager"][2].abs().max())
    print("fused=%s loss eager %.6f graph %.6f | max weight diff after 2 warm-up steps %.3e | grad rel diff of step 3 %.3e" % (fused, res["eager"][0][0], res["graph"][0][0], dw, dg))
