import sys; sys.path.insert(0,'.')
import torch, bench
from rel_pose_amd import ops
dev=torch.device('cuda',0)
r=bench.supplementary_point(dev, tag='x', batch=8, hw=384, mode='fwd', precision='fp32', steps=3, warmup=1, timer_instance='mlp_fused_fwd', kernel_symbol='k', traffic_files=())
print(r['roofline'])
print(ops.FUSE_MLP, ops.GEMM_PRECISION)
