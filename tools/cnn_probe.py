import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from rel_pose_amd.model import ViTEss
dev = torch.device("cuda", 0)
def timeit(fn, n=5, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n
B = 64
images, poses, intr = bench.synthetic_batch(B, 384, dev, 1)
for cfg in sys.argv[1:]:
    bm, cl = cfg.split(",")
    torch.backends.cudnn.benchmark = bm == "1"
    model = ViTEss(bench.model_args()).to(dev).train()
    for p in list(model.resnet.layer3.parameters()) + list(model.resnet.layer4.parameters()): p.requires_grad = False
    if cl == "1":
        model.resnet = model.resnet.to(memory_format=torch.channels_last)
        model.extractor_final_conv = model.extractor_final_conv.to(memory_format=torch.channels_last)
    def step():
        fmap, _ = model.cnn_map(images, None)
        fmap.float().square().mean().backward()
    t = timeit(step)
    def fwd():
        with torch.no_grad(): model.cnn_map(images, None)
    print("benchmark=%s channels_last=%s: CNN fwd+bwd %.2f ms, fwd %.2f ms" % (bm, cl, t, timeit(fwd)), flush=True)
