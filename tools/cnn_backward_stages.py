"""Stage-by-stage comparison of the CNN front-end backward (GPU product modules vs the fp64 oracle): forward error and the error of the
gradient ENTERING each block, then every resnet parameter gradient.  Shows where an end-to-end gradient discrepancy is born (round 3:
ReLU-mask flips inside layer1, not a kernel)."""
import sys, os; sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import rel_pose_amd._env
import torch, types
import torch.nn.functional as F
from oracle import relpose_oracle as O
from rel_pose_amd.model import ViTEss
from rel_pose_amd import ops
def rel(a,b):
    a=a.detach().double().cpu(); b=b.detach().double().cpu()
    return float((a-b).abs().max()/b.abs().max().clamp_min(1e-30))
args=types.SimpleNamespace(noess="", pool_size=60, fc_hidden_size=512, fusion_transformer=True, transformer_depth=6, cross_features=False, use_single_softmax=False, no_pos_encoding=False, l1_pos_encoding=False)
shapes=dict(O.vit_param_shapes()); shapes.update(O.cnn_param_shapes())
sd32, sd64 = O.make_state(shapes, torch.float32), O.make_state(shapes, torch.float64)
B,H,W=2,384,384
imgs=O.synthetic_images(B,H,W,key=77)
# oracle with retained intermediates
sd={k:(v.clone().requires_grad_(True) if v.is_floating_point() and 'running' not in k else v) for k,v in sd64.items()}
x=O.preprocess(imgs.double())
inter={}
def keep(n,t): t.retain_grad(); inter[n]=t; return t
x=F.conv2d(x, sd["resnet.conv1.weight"], None, 2, 3); keep('conv1',x)
x=F.relu(O._bn(sd,"resnet.bn1",x,True)); keep('bn1relu',x)
x=F.max_pool2d(x,3,2,1); keep('pool',x)
for li,stride in ((1,1),(2,2)):
    for b in range(2):
        x=O._basic_block(sd,"resnet.layer%d.%d."%(li,b),x,stride if b==0 else 1,True); keep('layer%d.%d'%(li,b),x)
cot=O.closed_form(tuple(x.shape), 99, 1.0, dtype=torch.float64)
(x*cot).sum().backward()
m=ViTEss(args); m.load_state_dict(sd32, strict=True); m=m.cuda().train()
r=m.resnet
g={}
def hk(n):
    def f(grad): g[n]=grad
    return f
xp=ops.preprocess(imgs.cuda(), pad=3)
c1,stats=ops.StemConvFn.apply(xp, r.conv1.weight, True); c1.register_hook(hk('conv1'))
y=ops.bn_relu_maxpool(r.bn1, r.maxpool, c1, stats); y.register_hook(hk('pool'))
outs={'conv1':c1,'pool':y}
for li,layer in ((1,r.layer1),(2,r.layer2)):
    for b in range(2):
        y=layer[b](y); y.register_hook(hk('layer%d.%d'%(li,b))); outs['layer%d.%d'%(li,b)]=y
(y*cot.float().cuda()).sum().backward()
for n in ['layer2.1','layer2.0','layer1.1','layer1.0','pool','conv1']:
    print("%-10s fwd %.2e   grad %.2e" % (n, rel(outs[n], inter[n]), rel(g[n], inter[n].grad)))
for n,p in m.named_parameters():
    if p.grad is not None and n.startswith('resnet') and sd[n].grad is not None:
        print("%-40s %.2e" % (n, rel(p.grad, sd[n].grad)))
