"""ViTEss -- drop-in for reference src/model.py (same constructor args, forward signature, attribute names and
state_dict keys), with the ViT + Essential-Matrix-Module hot path on hand-written gfx950 kernels.

forward(images, Gs, intrinsics=None, inference=False)      reference src/model.py:161-191
  images      [B,2,3,H,W] fp32 BGR 0..255
  Gs          SE3-like (.data [B,2,7]) or numpy [2,7]
  intrinsics  [B,2,4] (fx,fy,cx,cy) in pixels -- rescaled IN PLACE to the 24x24 grid like the reference (:100-109)
returns [SE3([B,2,7])]  (or numpy [2,7] of element 0 when inference=True)
"""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops
from .modules.extractor import ResidualBlock
from .modules.resnet import resnet18
from .modules.vision_transformer import _create_vision_transformer
from .se3 import SE3


class ViTEss(nn.Module):
    def __init__(self, args):
        super().__init__()
        noess = getattr(args, "noess", None) or None            # src/model.py:15-17: '' and absent both mean "off"
        if not getattr(args, "fusion_transformer", False):
            # the reference's non-transformer head cannot run: pool_transformer_output leaves [2B,60,24,24], reshape([B,-1])
            # gives 69120 columns for a Linear(34560,.) (src/model.py:63-71,180-189); with noess pool_attn gets 120 of 384
            raise NotImplementedError("fusion_transformer=False crashes in the reference itself (src/model.py:63-71,189); "
                                      "every scripts/*.sh passes --fusion_transformer")
        self.noess = noess
        self.total_num_features = 192
        self.feature_resolution = (24, 24)
        self.num_images = 2
        self.pose_size = 7
        self.num_patches = 24 * 24
        self.H2 = args.fc_hidden_size

        self.flatten = nn.Flatten(0, 1)
        self.resnet = resnet18(pretrained=True)
        self.resnet_pretrained = bool(getattr(self.resnet, "pretrained_loaded", False))
        self.resnet.fc = nn.Identity()
        self.extractor_final_conv = ResidualBlock(128, self.total_num_features, "batch", kernel_size=5)

        self.num_heads = 3
        self.transformer_depth = args.transformer_depth
        self.fusion_transformer = _create_vision_transformer(
            "vit_tiny_patch16_384", patch_size=16, embed_dim=self.total_num_features, depth=args.transformer_depth,
            num_heads=self.num_heads, cross_features=args.cross_features, use_single_softmax=args.use_single_softmax,
            no_pos_encoding=args.no_pos_encoding, noess=noess, l1_pos_encoding=args.l1_pos_encoding)
        nn.init.xavier_uniform_(self.fusion_transformer.pos_embed)     # src/model.py:54-56
        self.pos_encoding = None

        hd = self.total_num_features // self.num_heads
        self.H = int(self.num_heads * 2 * (hd + 6) * hd)               # 26880, src/model.py:61
        if self.noess:                                                 # src/model.py:73-82
            self.pool_feat1, self.pool_feat2 = 96, 43
            self.H = 24 * 24 * self.pool_feat2
            self.pool_attn = nn.Sequential(
                nn.Conv2d(self.total_num_features * 2, self.pool_feat1, kernel_size=1, bias=True),
                nn.BatchNorm2d(self.pool_feat1), nn.ReLU(),
                nn.Conv2d(self.pool_feat1, self.pool_feat2, kernel_size=1, bias=True), nn.BatchNorm2d(self.pool_feat2))
            self.pool_attn.to(memory_format=torch.channels_last)
        self.pose_regressor = nn.Sequential(
            nn.Linear(self.H, self.H2), nn.ReLU(), nn.Linear(self.H2, self.H2), nn.ReLU(),
            nn.Linear(self.H2, self.num_images * self.pose_size), nn.Unflatten(1, (self.num_images, self.pose_size)))
        # CNN front-end in channels-last: MIOpen's fp32 kernels are NHWC (saves its NCHW<->NHWC transposes, measured
        # 24.1 -> 20.1 ms fwd+bwd at 128 images) and the [2B,24,24,192] map IS the token layout (src/model.py:136-141)
        self.resnet.to(memory_format=torch.channels_last)
        self.extractor_final_conv.to(memory_format=torch.channels_last)
        # ImageNet statistics as (non-persistent) buffers: no per-call host-to-device copies, state_dict unchanged
        self.register_buffer("_mean", torch.tensor([0.485, 0.456, 0.406]).view(3, 1, 1), persistent=False)
        self.register_buffer("_std", torch.tensor([0.229, 0.224, 0.225]).view(3, 1, 1), persistent=False)
        self._intr_scale = {}

    # -- src/model.py:100-109 ---------------------------------------------------------------------
    def update_intrinsics(self, input_shape, intrinsics):
        sizey, sizex = self.feature_resolution
        key = (int(input_shape[-2]), int(input_shape[-1]), str(intrinsics.device), intrinsics.dtype)
        sc = self._intr_scale.get(key)
        if sc is None:          # (sx, sy, sx, sy): same fp32 products as the reference's two index_put_ lines, no index tensors
            scalex, scaley = sizex / input_shape[-1], sizey / input_shape[-2]
            sc = torch.tensor([scalex, scaley, scalex, scaley], dtype=intrinsics.dtype, device=intrinsics.device)
            self._intr_scale[key] = sc
        intrinsics.mul_(sc)     # in place on the CALLER's tensor, like the reference
        return intrinsics

    # -- src/model.py:111-143 ---------------------------------------------------------------------
    def cnn_map(self, images, intrinsics=None):
        """preprocessing + CNN front-end -> [2B,192,24,24] (PyTorch-ROCm / MIOpen: 'next' row 8f-1)."""
        if intrinsics is not None:
            intrinsics = self.update_intrinsics(images.shape, intrinsics)
        with ops.batches_tracked_batch(), ops.conv_params_bf16(self.resnet.layer1, self.resnet.layer2, self.extractor_final_conv):
            return self._cnn_layers(images), intrinsics

    def _cnn_layers(self, images):
        r = self.resnet
        if ops.stem_conv_ok(r.conv1, images):
            # BGR->RGB, /255, mean/std, nearest 224 (one bit-exact HIP kernel) written inside the stem's zero padding; hand-written conv1
            # (in training also bn1's batch statistics, from the convolution's epilogue)
            stats = None
            fn = ops.StemConvBf16Fn if ops.CNN_PRECISION == 1 else ops.StemConvFn      # (bf16 configuration: csrc/conv_stem_bf16.hip)
            if r.bn1.training and ops.STEM_STATS and ops.FUSE_STEM_POOL:
                c1, stats = fn.apply(ops.preprocess(images, pad=3), r.conv1.weight, True)
            else:
                c1 = fn.apply(ops.preprocess(images, pad=3), r.conv1.weight)
        else:
            c1, stats = ops.conv2d(r.conv1, ops.preprocess(images)), None
        x = ops.bn_relu_maxpool(r.bn1, r.maxpool, c1, stats)                    # stem: BatchNorm + ReLU + pool, one pass each way
        x = r.layer2(r.layer1(x))
        return self.extractor_final_conv(x)

    def extract_features(self, images, intrinsics=None):
        """tokens [2B,576,192] WITHOUT pos_embed (reference return value, src/model.py:136-143)."""
        fmap, intrinsics = self.cnn_map(images, intrinsics)
        if fmap.dtype != torch.float32:
            fmap = fmap.float()
        zero_pe = torch.zeros_like(self.fusion_transformer.pos_embed[0])
        return ops.TokensFn.apply(fmap, zero_pe), intrinsics

    def forward_tokens(self, fmap, Gs_data, intrinsics=None):
        """hot path: CNN map [2B,192,24,24] (or [2B,192,576]) -> normalised poses [B,2,7]."""
        ft = self.fusion_transformer
        if fmap.dtype != torch.float32:            # bf16 CNN front-end (the bf16 configuration): the token stream is fp32
            fmap = fmap.float()
        x = ops.TokensFn.apply(fmap, ft.pos_embed[0])
        for layer in range(self.transformer_depth):
            x = ft.blocks[layer](x, intrinsics=intrinsics)
        pr = self.pose_regressor
        if self.noess:
            # src/model.py:178,183-188.  features.reshape([B,24,24,-1]).permute(0,3,1,2) on the contiguous [2B,576,192] norm
            # output: "pixel" m of pair b holds tokens 2m, 2m+1 of the concatenated (image 0, image 1) token list -- in
            # channels-last memory that NCHW tensor IS the buffer, so the 1x1-conv/BN head (MIOpen, like the CNN front
            # end) runs on a view; only its [B,43,24,24] output is re-laid out (c-major) for the regressor.
            B = x.shape[0] // 2
            f = ops.LayerNormFn.apply(x, ft.norm.weight, ft.norm.bias)
            pooled = self.pool_attn(f.view(B, 24, 24, 2 * self.total_num_features).permute(0, 3, 1, 2))
            feats = pooled.contiguous(memory_format=torch.contiguous_format).reshape(B, -1)
            return ops.RegressFn.apply(feats, Gs_data, pr[0].weight, pr[0].bias, pr[2].weight, pr[2].bias, pr[4].weight,
                                       pr[4].bias)
        return ops.HeadFn.apply(x, Gs_data, ft.norm.weight, ft.norm.bias, pr[0].weight, pr[0].bias, pr[2].weight,
                                pr[2].bias, pr[4].weight, pr[4].bias)

    def forward(self, images, Gs, intrinsics=None, inference=False):
        if not hasattr(Gs, "data") or isinstance(Gs, np.ndarray):
            Gs = SE3(torch.from_numpy(np.asarray(Gs)).unsqueeze(0).to(images.device).float())
        fmap, intrinsics = self.cnn_map(images, intrinsics)
        out = self.forward_tokens(fmap, Gs.data.to(torch.float32), intrinsics)
        if inference:
            return out[0].detach().cpu().numpy()
        return [SE3(out)]
