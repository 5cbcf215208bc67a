"""Host-side mirror of the reference's ViT modules for the default rel_pose configuration.

Same class names, constructor arguments, attribute names (=> identical ``state_dict`` keys) and forward
signatures as reference src/modules/vision_transformer.py (Attention :307-333, Block :336-354,
CrossAttention :160-238, CrossBlock :265-296, VisionTransformer :357-443) and vit_layers/mlp.py:8-26,
but the modules hold parameters only: the arithmetic of a whole Block / CrossBlock runs as one autograd
Function over hand-written gfx950 kernels (rel_pose_amd/ops.py).  Ablation flags: cross_features,
use_single_softmax, l1_pos_encoding and noess are implemented (forward and backward, pinned against the reference);
no_pos_encoding, which crashes in the reference itself, is rejected loudly.
"""
import torch
import torch.nn as nn

from .. import ops


class Mlp(nn.Module):
    def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=nn.GELU, drop=0.):
        super().__init__()
        if drop != 0.:
            raise NotImplementedError("dropout is 0 in every rel_pose configuration")
        self.fc1 = nn.Linear(in_features, hidden_features or in_features)
        self.act = act_layer()
        self.fc2 = nn.Linear(hidden_features or in_features, out_features or in_features)
        self.drop = nn.Dropout(drop)


class Attention(nn.Module):
    def __init__(self, dim, num_heads=8, qkv_bias=False, attn_drop=0., proj_drop=0.):
        super().__init__()
        self.num_heads = num_heads
        self.scale = (dim // num_heads) ** -0.5
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.proj = nn.Linear(dim, dim)
        self.proj_drop = nn.Dropout(proj_drop)
        self.attn_drop = nn.Dropout(attn_drop)


def _check_dims(dim, num_heads):
    if dim != ops.DIM or num_heads != ops.HEADS:
        raise NotImplementedError("kernels are specialised for ViT-Tiny: dim 192, 3 heads x 64 (src/model.py:19-44)")


class Block(nn.Module):
    def __init__(self, dim, num_heads, mlp_ratio=4., qkv_bias=False, drop=0., attn_drop=0., drop_path=0.,
                 act_layer=nn.GELU, norm_layer=nn.LayerNorm):
        super().__init__()
        _check_dims(dim, num_heads)
        if drop or attn_drop or drop_path:
            raise NotImplementedError("all drop rates are 0 in rel_pose (vision_transformer.py:369,408)")
        self.norm1 = norm_layer(dim)
        self.attn = Attention(dim, num_heads=num_heads, qkv_bias=qkv_bias)
        self.drop_path = nn.Identity()
        self.norm2 = norm_layer(dim)
        self.mlp = Mlp(in_features=dim, hidden_features=int(dim * mlp_ratio), act_layer=act_layer)

    def forward(self, x, camera=None, intrinsics=None):
        a, m = self.attn, self.mlp
        return ops.BlockFn.apply(x, self.norm1.weight, self.norm1.bias, a.qkv.weight, a.qkv.bias, a.proj.weight,
                                 a.proj.bias, self.norm2.weight, self.norm2.bias, m.fc1.weight, m.fc1.bias,
                                 m.fc2.weight, m.fc2.bias)


class CrossAttention(nn.Module):
    def __init__(self, dim, num_heads=8, qkv_bias=False, attn_drop=0., proj_drop=0., cross_features=False,
                 use_single_softmax=False, no_pos_encoding=False, noess=False, l1_pos_encoding=False):
        super().__init__()
        if no_pos_encoding and not noess:
            raise NotImplementedError("--no_pos_encoding cannot run in the reference either: proj_fundamental is always "
                                      "Linear(210,192) (vision_transformer.py:179) but the flag feeds it 192 columns (:225-227)")
        self.cross_features, self.use_single_softmax, self.l1_pos_encoding = cross_features, use_single_softmax, l1_pos_encoding
        self.num_heads = num_heads
        self.scale = (dim // num_heads) ** -0.5
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.attn_drop = nn.Dropout(attn_drop)
        self.noess = noess
        if noess:       # plain cross attention: softmax(q_i k_partner^T) v_partner (vision_transformer.py:176-177,239-262)
            self.proj = nn.Linear(dim, dim)
        else:
            self.proj_fundamental = nn.Linear(dim + 6 * num_heads, dim)
        self.proj_drop = nn.Dropout(proj_drop)


class CrossBlock(nn.Module):
    def __init__(self, dim, num_heads, mlp_ratio=4., qkv_bias=False, drop=0., attn_drop=0., drop_path=0.,
                 act_layer=nn.GELU, norm_layer=nn.LayerNorm, cross_features=False, use_single_softmax=False,
                 no_pos_encoding=False, noess=False, l1_pos_encoding=False):
        super().__init__()
        _check_dims(dim, num_heads)
        self.norm1 = norm_layer(dim)
        self.cross_attn = CrossAttention(dim, num_heads=num_heads, qkv_bias=qkv_bias, cross_features=cross_features,
                                         use_single_softmax=use_single_softmax, no_pos_encoding=no_pos_encoding,
                                         noess=noess, l1_pos_encoding=l1_pos_encoding)
        self.drop_path = nn.Identity()
        self.norm2 = norm_layer(dim)
        self.mlp = Mlp(in_features=dim, hidden_features=int(dim * mlp_ratio), act_layer=act_layer)
        self.noess = noess
        self.strict_intrinsics = False

    def forward(self, x, camera=None, intrinsics=None):
        """x [2B,576,192] (images 2b, 2b+1 form pair b) -> [2B,70,192] ([2B,576,192] with noess).
        `intrinsics` [B,2,4] on the 24x24 grid."""
        B = x.shape[0] // 2
        if intrinsics is not None:
            intrinsics = intrinsics.to(device=x.device, dtype=torch.float32).contiguous()
            # unsupported intrinsics (unequal within a pair, principal point on an axis) are caught on the device without a
            # sync: rp_posenc turns the offending pair's encodings -- hence its pose and the loss -- into NaN.  strict_intrinsics
            # additionally raises on the host like the reference does (vision_transformer.py:117,124), at the price of a sync.
            if self.strict_intrinsics:
                assert bool(torch.all(intrinsics[:, 0] == intrinsics[:, 1])), "intrinsics differ within a pair"
                assert float(intrinsics[0, 0, 2] * intrinsics[0, 0, 3]) != 0.0, "principal point at the origin"
        a, m = self.cross_attn, self.mlp
        if self.noess:  # x + CrossAttn(LN1(x)); + Mlp(LN2(.)) (vision_transformer.py:297-304): a Block with partner keys/values
            return ops.BlockFn.apply(x, self.norm1.weight, self.norm1.bias, a.qkv.weight, a.qkv.bias, a.proj.weight,
                                     a.proj.bias, self.norm2.weight, self.norm2.bias, m.fc1.weight, m.fc1.bias,
                                     m.fc2.weight, m.fc2.bias, True)
        pos = ops.posenc(intrinsics, B, x.device, l1=a.l1_pos_encoding)
        return ops.CrossBlockFn.apply(x, pos, self.norm1.weight, self.norm1.bias, a.qkv.weight, a.qkv.bias,
                                      a.proj_fundamental.weight, a.proj_fundamental.bias, self.norm2.weight,
                                      self.norm2.bias, m.fc1.weight, m.fc1.bias, m.fc2.weight, m.fc2.bias,
                                      a.use_single_softmax, a.cross_features)


class VisionTransformer(nn.Module):
    """Parameter container with the reference's attribute names; like the reference's cut-down copy of timm it
    has no forward of its own -- ViTEss drives .blocks / .pos_embed / .norm by hand (src/model.py:169-178)."""

    def __init__(self, embed_dim=192, depth=6, num_heads=3, mlp_ratio=4., qkv_bias=True, cross_features=False,
                 use_single_softmax=False, no_pos_encoding=False, noess=False, l1_pos_encoding=False, **_unused):
        super().__init__()
        self.embed_dim = self.num_features = embed_dim
        norm = lambda d: nn.LayerNorm(d, eps=ops.LN_EPS)   # vision_transformer.py:396
        blocks = []
        for i in range(depth):
            if i == depth - 1:
                blocks.append(CrossBlock(embed_dim, num_heads, mlp_ratio, qkv_bias, norm_layer=norm,
                                         cross_features=cross_features, use_single_softmax=use_single_softmax,
                                         no_pos_encoding=no_pos_encoding, noess=noess,
                                         l1_pos_encoding=l1_pos_encoding))
            else:
                blocks.append(Block(embed_dim, num_heads, mlp_ratio, qkv_bias, norm_layer=norm))
        self.blocks = nn.Sequential(*blocks)
        self.norm = norm(embed_dim)
        self.pos_embed = nn.Parameter(torch.zeros(1, 576, embed_dim))
        self.pos_drop = nn.Dropout(p=0.)
        self.patch_embed = nn.Identity()      # src/model.py:48
        self.head = nn.Identity()             # src/model.py:49
        self.cls_token = None                 # src/model.py:50
        self.apply(_init_vit_weights)


def _init_vit_weights(m):
    # reference: trunc_normal_(std=.02) on Linear weights, zero biases, unit LayerNorm (vision_transformer.py:470-502)
    if isinstance(m, nn.Linear):
        nn.init.trunc_normal_(m.weight, std=.02)
        if m.bias is not None:
            nn.init.zeros_(m.bias)
    elif isinstance(m, nn.LayerNorm):
        nn.init.zeros_(m.bias)
        nn.init.ones_(m.weight)


def _create_vision_transformer(variant, default_cfg=None, **kwargs):
    """Factory with the reference's name (vision_transformer.py:544-557)."""
    if variant != "vit_tiny_patch16_384":
        raise NotImplementedError(variant)
    kwargs.pop("patch_size", None)
    return VisionTransformer(**kwargs)
