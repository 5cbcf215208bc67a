"""CPU restatement of rel_pose's hot path  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this module.  The product path (``rel_pose_amd``) never does; it fails loudly when the HIP
extension is missing.

What it restates (file:line are relative to the reference checkout, /root/reference):

  * preprocessing + index ops            src/model.py:100-125,136-141
  * CNN front-end (functional)           src/model.py:127-134, src/modules/extractor.py:5-65
  * LayerNorm eps=1e-6 / exact-erf GELU  src/modules/vision_transformer.py:396-397
  * Attention / Mlp / Block              src/modules/vision_transformer.py:321-333,349-354,
                                         src/modules/vit_layers/mlp.py:20-26
  * get_positional_encodings             src/modules/vision_transformer.py:90-158
  * CrossAttention (EMM) / CrossBlock    src/modules/vision_transformer.py:188-238,285-296
  * final norm, regressor, normalise     src/model.py:161-191,91-98,145-159

Pinning: ``tests/golden/make_fixtures.py`` imports the real reference in the build container
(it cannot travel to the GPU box) and commits its outputs on closed-form inputs/weights under
``tests/golden/``; ``tests/test_oracle_golden.py`` checks this restatement against them in fp32 and
fp64.  Everything is plain PyTorch CPU ops, any float dtype, differentiable (so autograd of the
restatement is the gradient oracle).  The SE(3) geodesic loss lives in un-vendored lietorch
(pinned lietorch==0.2, reference environment.yml:19) => its parity is UNPINNED.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

LN_EPS = 1e-6           # vision_transformer.py:396
N_TOK = 576             # 24x24 feature grid, src/model.py:20,23
GRID = 24
DIM = 192
HEADS = 3
HEAD_DIM = 64
POS_FEATS = 6


# --------------------------------------------------------------------------------------------
# closed-form (RNG- and version-independent) tensors: integer hash -> uniform(-1,1)
# --------------------------------------------------------------------------------------------
def hash_uniform(n, key):
    """n values in [-1,1): 64-bit integer mix of (index, key); pure integer arithmetic."""
    with np.errstate(over="ignore"):
        i = np.arange(n, dtype=np.uint64)
        x = i * np.uint64(0x9E3779B97F4A7C15) + np.uint64(key) * np.uint64(0xBF58476D1CE4E5B9) + np.uint64(0x94D049BB133111EB)
        x ^= x >> np.uint64(30)
        x *= np.uint64(0xBF58476D1CE4E5B9)
        x ^= x >> np.uint64(27)
        x *= np.uint64(0x94D049BB133111EB)
        x ^= x >> np.uint64(31)
    u = (x >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)   # [0,1)
    return 2.0 * u - 1.0


def closed_form(shape, key, scale=1.0, offset=0.0, dtype=torch.float32):
    n = int(np.prod(shape))
    v = hash_uniform(n, key) * scale + offset
    return torch.from_numpy(v.reshape(shape)).to(dtype)


def _key(name):
    h = 1469598103934665603
    for ch in name.encode():
        h = ((h ^ ch) * 1099511628211) % (1 << 64)
    return h % (1 << 31)


def vit_param_shapes(dim=DIM, heads=HEADS, depth=6, mlp_ratio=4, fc_hidden=512, noess=False):
    """state_dict keys/shapes of fusion_transformer.* and pose_regressor.* (SURVEY.md 8b); noess adds pool_attn.*
    and swaps cross_attn.proj_fundamental for cross_attn.proj (src/model.py:73-82, vision_transformer.py:176-179)."""
    hd = dim // heads
    hid = dim * mlp_ratio
    s = {"fusion_transformer.pos_embed": (1, N_TOK, dim)}
    for l in range(depth):
        p = "fusion_transformer.blocks.%d." % l
        for nm in ("norm1", "norm2"):
            s[p + nm + ".weight"] = (dim,)
            s[p + nm + ".bias"] = (dim,)
        a = "cross_attn." if l == depth - 1 else "attn."
        s[p + a + "qkv.weight"] = (3 * dim, dim)
        s[p + a + "qkv.bias"] = (3 * dim,)
        if l == depth - 1 and not noess:
            s[p + a + "proj_fundamental.weight"] = (dim, dim + POS_FEATS * heads)
            s[p + a + "proj_fundamental.bias"] = (dim,)
        else:
            s[p + a + "proj.weight"] = (dim, dim)
            s[p + a + "proj.bias"] = (dim,)
        s[p + "mlp.fc1.weight"] = (hid, dim)
        s[p + "mlp.fc1.bias"] = (hid,)
        s[p + "mlp.fc2.weight"] = (dim, hid)
        s[p + "mlp.fc2.bias"] = (dim,)
    s["fusion_transformer.norm.weight"] = (dim,)
    s["fusion_transformer.norm.bias"] = (dim,)
    H = heads * 2 * (hd + POS_FEATS) * hd          # src/model.py:61
    if noess:                                      # src/model.py:73-82
        H = N_TOK * 43
        s["pool_attn.0.weight"] = (96, 2 * dim, 1, 1)
        s["pool_attn.0.bias"] = (96,)
        s["pool_attn.3.weight"] = (43, 96, 1, 1)
        s["pool_attn.3.bias"] = (43,)
        for i_, c_ in ((1, 96), (4, 43)):
            s["pool_attn.%d.weight" % i_] = (c_,)
            s["pool_attn.%d.bias" % i_] = (c_,)
            s["pool_attn.%d.running_mean" % i_] = (c_,)
            s["pool_attn.%d.running_var" % i_] = (c_,)
            s["pool_attn.%d.num_batches_tracked" % i_] = ()
    s["pose_regressor.0.weight"] = (fc_hidden, H)
    s["pose_regressor.0.bias"] = (fc_hidden,)
    s["pose_regressor.2.weight"] = (fc_hidden, fc_hidden)
    s["pose_regressor.2.bias"] = (fc_hidden,)
    s["pose_regressor.4.weight"] = (14, fc_hidden)
    s["pose_regressor.4.bias"] = (14,)
    return s


def cnn_param_shapes():
    """resnet.* (torchvision resnet18 minus fc) + extractor_final_conv.* keys/shapes."""
    s = {}

    def bn(p, c):
        s[p + ".weight"] = (c,)
        s[p + ".bias"] = (c,)
        s[p + ".running_mean"] = (c,)
        s[p + ".running_var"] = (c,)
        s[p + ".num_batches_tracked"] = ()

    s["resnet.conv1.weight"] = (64, 3, 7, 7)
    bn("resnet.bn1", 64)
    inp = 64
    for li, (c, stride) in enumerate([(64, 1), (128, 2), (256, 2), (512, 2)], start=1):
        for b in range(2):
            p = "resnet.layer%d.%d." % (li, b)
            s[p + "conv1.weight"] = (c, inp if b == 0 else c, 3, 3)
            bn(p + "bn1", c)
            s[p + "conv2.weight"] = (c, c, 3, 3)
            bn(p + "bn2", c)
            if b == 0 and (stride != 1 or inp != c):
                s[p + "downsample.0.weight"] = (c, inp, 1, 1)
                bn(p + "downsample.1", c)
        inp = c
    e = "extractor_final_conv."
    s[e + "conv1.weight"] = (192, 128, 3, 3)
    s[e + "conv1.bias"] = (192,)
    s[e + "conv2.weight"] = (192, 192, 5, 5)
    s[e + "conv2.bias"] = (192,)
    for nm in ("norm1", "norm2", "norm3", "downsample.1"):
        bn(e + nm, 192)
    s[e + "downsample.0.weight"] = (192, 128, 5, 5)
    s[e + "downsample.0.bias"] = (192,)
    return s


def make_state(shapes, dtype=torch.float32):
    """Closed-form weights: linears/convs ~ U(-a,a) with a = sqrt(3/fan_in); LN/BN affine near identity."""
    sd = {}
    for name, shp in shapes.items():
        k = _key(name)
        if name.endswith("num_batches_tracked"):
            sd[name] = torch.zeros((), dtype=torch.long)
        elif name.endswith("running_mean"):
            sd[name] = closed_form(shp, k, 0.1, dtype=dtype)
        elif name.endswith("running_var"):
            sd[name] = closed_form(shp, k, 0.2, 1.0, dtype=dtype)
        elif ("norm" in name or ".bn" in name or name.endswith("downsample.1.weight") or name.endswith("downsample.1.bias")
              or name.startswith("pool_attn.1.") or name.startswith("pool_attn.4.")):
            if name.endswith("weight"):
                sd[name] = closed_form(shp, k, 0.2, 1.0, dtype=dtype)
            else:
                sd[name] = closed_form(shp, k, 0.1, dtype=dtype)
        elif name.endswith("pos_embed"):
            sd[name] = closed_form(shp, k, 0.5, dtype=dtype)
        elif name.endswith("bias"):
            sd[name] = closed_form(shp, k, 0.1, dtype=dtype)
        else:
            fan_in = int(np.prod(shp[1:]))
            gain = 1.6 if len(shp) == 4 else 1.0    # keep conv activations alive through ReLUs
            sd[name] = closed_form(shp, k, gain * math.sqrt(3.0 / fan_in), dtype=dtype)
    # downsample.1 aliases norm3 in the reference (extractor.py:26-49): one tensor, two names
    for suf in ("weight", "bias", "running_mean", "running_var", "num_batches_tracked"):
        a, b = "extractor_final_conv.norm3." + suf, "extractor_final_conv.downsample.1." + suf
        if a in sd:
            sd[b] = sd[a]
    return sd


def synthetic_images(B, H=384, W=384, key=7, dtype=torch.float32):
    """[B,2,3,H,W] 8-bit-like BGR images."""
    u = closed_form((B, 2, 3, H, W), key, 0.5, 0.5, dtype=torch.float64)
    return torch.floor(u * 255.0).to(dtype)


def synthetic_tokens(B2, key=11, dtype=torch.float32):
    """[2B,576,192] post-ReLU-like CNN features."""
    return closed_form((B2, N_TOK, DIM), key, 1.0, dtype=dtype).abs_()


# --------------------------------------------------------------------------------------------
# preprocessing + index ops  (bit-exact rows of SURVEY.md 8a: a2, a3)
# --------------------------------------------------------------------------------------------
def update_intrinsics(input_hw, intrinsics):
    """src/model.py:100-109: in-place rescale of the CALLER's tensor to the 24x24 grid."""
    H, W = input_hw
    intrinsics[:, :, [0, 2]] = (GRID / W) * intrinsics[:, :, [0, 2]]
    intrinsics[:, :, [1, 3]] = (GRID / H) * intrinsics[:, :, [1, 3]]
    return intrinsics


def preprocess(images):
    """src/model.py:115-118,124-125: BGR->RGB, /255, ImageNet mean/std, flatten pairs, nearest 224."""
    x = images[:, :, [2, 1, 0]] / 255.0
    mean = torch.as_tensor([0.485, 0.456, 0.406], dtype=x.dtype)
    std = torch.as_tensor([0.229, 0.224, 0.225], dtype=x.dtype)
    x = x.sub(mean[:, None, None]).div(std[:, None, None])
    x = x.flatten(0, 1)
    return nearest_resize(x, 224)


def nearest_src_index(out_size, in_size):
    """PyTorch 'nearest': src = min(floor(dst * (in/out) as fp32), in-1)."""
    scale = np.float32(in_size) / np.float32(out_size)
    idx = np.floor(np.arange(out_size, dtype=np.float32) * scale).astype(np.int64)
    return np.minimum(idx, in_size - 1)


def nearest_resize(x, size):
    iy = torch.from_numpy(nearest_src_index(size, x.shape[-2]))
    ix = torch.from_numpy(nearest_src_index(size, x.shape[-1]))
    return x[..., iy, :][..., ix]


def tokens_from_cnn(feat):
    """src/model.py:136-141: [2B,192,24,24] -> [2B,576,192], token n = row*24+col."""
    return feat.reshape(feat.shape[0], -1, N_TOK)[:, :DIM].permute(0, 2, 1)


# --------------------------------------------------------------------------------------------
# CNN front-end (functional; "next" row 8f-1, needed for the end-to-end check)
# --------------------------------------------------------------------------------------------
def _bn(sd, p, x, train):
    if train:
        return F.batch_norm(x, None, None, sd[p + ".weight"], sd[p + ".bias"], True, 0.1, 1e-5)
    return F.batch_norm(x, sd[p + ".running_mean"].to(x.dtype), sd[p + ".running_var"].to(x.dtype),
                        sd[p + ".weight"], sd[p + ".bias"], False, 0.1, 1e-5)


def _basic_block(sd, p, x, stride, train):
    idt = x
    if (p + "downsample.0.weight") in sd:
        idt = _bn(sd, p + "downsample.1", F.conv2d(x, sd[p + "downsample.0.weight"], None, stride), train)
    y = F.relu(_bn(sd, p + "bn1", F.conv2d(x, sd[p + "conv1.weight"], None, stride, 1), train))
    y = _bn(sd, p + "bn2", F.conv2d(y, sd[p + "conv2.weight"], None, 1, 1), train)
    return F.relu(y + idt)


def cnn_features(sd, x, train=False):
    """src/model.py:127-134 + extractor.py:51-65 -> [2B,192,24,24]."""
    x = F.conv2d(x, sd["resnet.conv1.weight"], None, 2, 3)
    x = F.relu(_bn(sd, "resnet.bn1", x, train))
    x = F.max_pool2d(x, 3, 2, 1)
    for li, stride in ((1, 1), (2, 2)):
        for b in range(2):
            x = _basic_block(sd, "resnet.layer%d.%d." % (li, b), x, stride if b == 0 else 1, train)
    e = "extractor_final_conv."
    y = F.relu(_bn(sd, e + "norm1", F.conv2d(x, sd[e + "conv1.weight"], sd[e + "conv1.bias"], 1, 1), train))
    y = F.relu(_bn(sd, e + "norm2", F.conv2d(y, sd[e + "conv2.weight"], sd[e + "conv2.bias"]), train))
    d = _bn(sd, e + "norm3", F.conv2d(x, sd[e + "downsample.0.weight"], sd[e + "downsample.0.bias"]), train)
    return F.relu(d + y)


# --------------------------------------------------------------------------------------------
# ViT pieces
# --------------------------------------------------------------------------------------------
def layernorm(x, w, b):
    return F.layer_norm(x, (x.shape[-1],), w, b, LN_EPS)


def gelu(x):
    return 0.5 * x * (1.0 + torch.erf(x * (1.0 / math.sqrt(2.0))))


def split_heads(qkv, heads=HEADS):
    """vision_transformer.py:323-324: [B',N,3C] -> q,k,v [B',H,N,d]; out col = s*C + h*d + e."""
    Bp, N, C3 = qkv.shape
    C = C3 // 3
    t = qkv.reshape(Bp, N, 3, heads, C // heads).permute(2, 0, 3, 1, 4)
    return t[0], t[1], t[2]


def attention(sd, p, x, heads=HEADS):
    """Attention.forward, vision_transformer.py:321-333."""
    Bp, N, C = x.shape
    q, k, v = split_heads(F.linear(x, sd[p + "qkv.weight"], sd[p + "qkv.bias"]), heads)
    scale = (C // heads) ** -0.5
    a = ((q @ k.transpose(-2, -1)) * scale).softmax(dim=-1)
    o = (a @ v).transpose(1, 2).reshape(Bp, N, C)
    return F.linear(o, sd[p + "proj.weight"], sd[p + "proj.bias"])


def mlp(sd, p, x):
    """Mlp.forward, vit_layers/mlp.py:20-26 (dropouts p=0)."""
    h = gelu(F.linear(x, sd[p + "fc1.weight"], sd[p + "fc1.bias"]))
    return F.linear(h, sd[p + "fc2.weight"], sd[p + "fc2.bias"])


def block(sd, p, x, heads=HEADS):
    """Block.forward, vision_transformer.py:349-354."""
    x = x + attention(sd, p + "attn.", layernorm(x, sd[p + "norm1.weight"], sd[p + "norm1.bias"]), heads)
    return x + mlp(sd, p + "mlp.", layernorm(x, sd[p + "norm2.weight"], sd[p + "norm2.bias"]))


def linspace24(dtype=torch.float32):
    return torch.linspace(-1, 1, steps=GRID, dtype=torch.float32).to(dtype)


def positional_encodings_loop(B, intrinsics=None):
    """Line-by-line restatement of get_positional_encodings (vision_transformer.py:90-158),
    including the 576-iteration host loop and torch.inverse.  fp32 like the reference.
    Slow; used to pin the closed form and as part of the CPU baseline's cost."""
    h = w = GRID
    positional = torch.ones([B, N_TOK, 6])
    ys = torch.linspace(-1, 1, steps=h)
    xs = torch.linspace(-1, 1, steps=w)
    p3 = ys.unsqueeze(0).repeat(B, w)
    p4 = xs.repeat_interleave(h).unsqueeze(0).repeat(B, 1)
    if intrinsics is not None:
        assert bool(torch.all(intrinsics[:, 0] == intrinsics[:, 1]))
        fx, fy, cx, cy = intrinsics[:, 0].float().unbind(dim=-1)
        assert float(cx[0] * cy[0]) != 0.0
        hpix, wpix = cy * 2, cx * 2
        K = torch.zeros([B, 3, 3])
        K[:, 0, 0] = (fx / wpix) * 2
        K[:, 1, 1] = (fy / hpix) * 2
        K[:, 0, 2] = (cx / wpix) * 2 - 1
        K[:, 1, 2] = (cy / hpix) * 2 - 1
        K[:, 2, 2] = 1
        Kinv = torch.inverse(K)
        for j in range(h):
            for k in range(w):
                w1, w2, w3 = torch.split(Kinv @ torch.tensor([xs[k], ys[j], 1]), 1, dim=1)
                p3[:, k * w + j] = w2.squeeze(1) / w3.squeeze(1)
                p4[:, k * w + j] = w1.squeeze(1) / w3.squeeze(1)
    positional[:, :, :5] = torch.stack([p3 * p3, p4 * p4, p3 * p4, p3, p4], dim=2)
    return positional


def positional_encodings(B, intrinsics=None, dtype=torch.float32, l1=False):
    """Closed form of the above: token n -> p3 = ys[n%24]*cy/fy,
    p4 = xs[n//24]*cx/fx (normalised principal point is identically 0).  The normalised focal
    lengths are formed in fp32 exactly as the reference does ((f/(2c))*2, then 1/x)."""
    n = torch.arange(N_TOK)
    ls = linspace24(torch.float32)
    p3 = ls[n % GRID].unsqueeze(0).repeat(B, 1)
    p4 = ls[n // GRID].unsqueeze(0).repeat(B, 1)
    if intrinsics is not None:
        fx, fy, cx, cy = intrinsics[:, 0].float().unbind(dim=-1)
        ifx = 1.0 / ((fx / (cx * 2)) * 2)
        ify = 1.0 / ((fy / (cy * 2)) * 2)
        p3 = p3 * ify[:, None]
        p4 = p4 * ifx[:, None]
    one = torch.ones_like(p3)
    if l1:      # get_l1_positional_encodings (vision_transformer.py:37-87): (1, 1, 1, p3, p4, 1)
        pos = torch.stack([one, one, one, p3, p4, one], dim=2)
    else:
        pos = torch.stack([p3 * p3, p4 * p4, p3 * p4, p3, p4, one], dim=2)
    return pos.to(dtype)


def cross_attention(sd, p, x1, x2, intrinsics=None, heads=HEADS, pos=None, return_parts=False, cross_features=False,
                    use_single_softmax=False, l1_pos_encoding=False):
    """CrossAttention.forward, ess branch (vision_transformer.py:188-238) incl. the runnable ablation flags
    (:201-203 single softmax, :208-209 L1 positional features, :218-220 cross features)."""
    B, N, C = x1.shape
    d = C // heads
    scale = d ** -0.5
    q1, k1, v1 = split_heads(F.linear(x1, sd[p + "qkv.weight"], sd[p + "qkv.bias"]), heads)
    q2, k2, v2 = split_heads(F.linear(x2, sd[p + "qkv.weight"], sd[p + "qkv.bias"]), heads)
    s1 = (q2 @ k1.transpose(-2, -1)) * scale
    s2 = (q1 @ k2.transpose(-2, -1)) * scale
    if use_single_softmax:
        a1, a2 = s1.softmax(dim=-1), s2.softmax(dim=-1)
    else:
        a1 = s1.softmax(dim=-1) * s1.softmax(dim=-2)
        a2 = s2.softmax(dim=-1) * s2.softmax(dim=-2)
    if pos is None:
        pos = positional_encodings(B, intrinsics, x1.dtype, l1=l1_pos_encoding)
    pe = pos.to(x1.dtype).unsqueeze(1).repeat(1, heads, 1, 1)
    v1 = torch.cat([v1, pe], dim=3)
    v2 = torch.cat([v2, pe], dim=3)
    if cross_features:
        f1 = (v2.transpose(-2, -1) @ a1) @ v1       # [B,H,70,70]
        f2 = (v1.transpose(-2, -1) @ a2) @ v2
    else:
        f1 = (v1.transpose(-2, -1) @ a1) @ v1
        f2 = (v2.transpose(-2, -1) @ a2) @ v2
    Ca = C + POS_FEATS * heads
    g1 = f1.reshape(B, Ca, Ca // heads).transpose(-2, -1)   # [B,70,210]: out[b,c,h*70+a] = F[b,h,a,c]
    g2 = f2.reshape(B, Ca, Ca // heads).transpose(-2, -1)
    o2 = F.linear(g2, sd[p + "proj_fundamental.weight"], sd[p + "proj_fundamental.bias"])
    o1 = F.linear(g1, sd[p + "proj_fundamental.weight"], sd[p + "proj_fundamental.bias"])
    if return_parts:
        return o2, o1, dict(f1=f1, f2=f2, s1=s1, s2=s2, pos=pos)
    return o2, o1                                    # flipped, vision_transformer.py:236-238


def cross_attention_noess(sd, p, x1, x2, heads=HEADS):
    """CrossAttention.forward, noess branch (vision_transformer.py:239-262): plain softmax attention whose keys/values
    come from the OTHER image; returns (image-1 slot, image-2 slot) = (attn(q1,k2,v2), attn(q2,k1,v1))."""
    B, N, C = x1.shape
    scale = (C // heads) ** -0.5
    q1, k1, v1 = split_heads(F.linear(x1, sd[p + "qkv.weight"], sd[p + "qkv.bias"]), heads)
    q2, k2, v2 = split_heads(F.linear(x2, sd[p + "qkv.weight"], sd[p + "qkv.bias"]), heads)
    o1 = (((q2 @ k1.transpose(-2, -1)) * scale).softmax(dim=-1) @ v1).transpose(1, 2).reshape(B, N, C)
    o2 = (((q1 @ k2.transpose(-2, -1)) * scale).softmax(dim=-1) @ v2).transpose(1, 2).reshape(B, N, C)
    o1 = F.linear(o1, sd[p + "proj.weight"], sd[p + "proj.bias"])
    o2 = F.linear(o2, sd[p + "proj.weight"], sd[p + "proj.bias"])
    return o2, o1                                    # flipped, vision_transformer.py:260-262


def cross_block(sd, p, x, intrinsics=None, heads=HEADS, pos=None, noess=False, **variant):
    """CrossBlock.forward (vision_transformer.py:285-304).  ess branch: no residual from x; noess branch: a Block."""
    b_s, h_w, nf = x.shape
    xp = x.reshape(-1, 2, h_w, nf)
    n1w, n1b = sd[p + "norm1.weight"], sd[p + "norm1.bias"]
    if noess:
        a, b = cross_attention_noess(sd, p + "cross_attn.", layernorm(xp[:, 0], n1w, n1b), layernorm(xp[:, 1], n1w, n1b),
                                     heads)
        x = x + torch.cat([a.unsqueeze(1), b.unsqueeze(1)], dim=1).reshape(b_s, h_w, nf)
        return x + mlp(sd, p + "mlp.", layernorm(x, sd[p + "norm2.weight"], sd[p + "norm2.bias"]))
    fa, fb = cross_attention(sd, p + "cross_attn.", layernorm(xp[:, 0], n1w, n1b), layernorm(xp[:, 1], n1w, n1b),
                             intrinsics, heads, pos, **variant)
    f = torch.cat([fa.unsqueeze(1), fb.unsqueeze(1)], dim=1).reshape(b_s, -1, nf)
    return f + mlp(sd, p + "mlp.", layernorm(f, sd[p + "norm2.weight"], sd[p + "norm2.bias"]))


def vit_features(sd, tokens, intrinsics=None, depth=6, heads=HEADS, pos=None, **variant):
    """src/model.py:169-178: +pos_embed, depth-1 Blocks, CrossBlock, final norm -> [2B,70,192]."""
    x = tokens + sd["fusion_transformer.pos_embed"]
    for l in range(depth - 1):
        x = block(sd, "fusion_transformer.blocks.%d." % l, x, heads)
    x = cross_block(sd, "fusion_transformer.blocks.%d." % (depth - 1), x, intrinsics, heads, pos, **variant)
    return layernorm(x, sd["fusion_transformer.norm.weight"], sd["fusion_transformer.norm.bias"])


def regress(sd, feats, B):
    """src/model.py:189,91-98."""
    x = feats.reshape(B, -1)
    x = F.relu(F.linear(x, sd["pose_regressor.0.weight"], sd["pose_regressor.0.bias"]))
    x = F.relu(F.linear(x, sd["pose_regressor.2.weight"], sd["pose_regressor.2.bias"]))
    return F.linear(x, sd["pose_regressor.4.weight"], sd["pose_regressor.4.bias"]).reshape(B, 2, 7)


def normalize_preds(Gs_data, pose_preds):
    """src/model.py:145-159: q / max(|q|, 0.01); slot 0 <- Gs[:, :1]; slot 1 <- prediction."""
    q = pose_preds[:, :, 3:]
    nrm = q.norm(dim=-1, keepdim=True)
    qn = q / torch.max(nrm, torch.full_like(nrm, 0.01))
    out = torch.cat([pose_preds[:, :, :3], qn], dim=-1)
    return torch.cat([Gs_data[:, :1], out[:, 1:]], dim=1)


def pool_attn(sd, feats, B, train=False):
    """--noess head, src/model.py:183-188,73-82: [2B,576,192] -reshape-> [B,24,24,384] -permute-> NCHW, 1x1 conv 384->96,
    BN, ReLU, 1x1 conv 96->43, BN, flattened c-major to [B,24768]."""
    f = feats.reshape(B, GRID, GRID, -1).permute(0, 3, 1, 2)
    f = F.relu(_bn(sd, "pool_attn.1", F.conv2d(f, sd["pool_attn.0.weight"], sd["pool_attn.0.bias"]), train))
    f = _bn(sd, "pool_attn.4", F.conv2d(f, sd["pool_attn.3.weight"], sd["pool_attn.3.bias"]), train)
    return f.reshape(B, -1)


def vit_ess_from_tokens(sd, tokens, Gs_data, intrinsics=None, train=False, **variant):
    """Hot path only: tokens [2B,576,192] -> poses [B,2,7] (intrinsics already on the 24-grid)."""
    B = tokens.shape[0] // 2
    feats = vit_features(sd, tokens, intrinsics, **variant)
    if variant.get("noess"):
        feats = pool_attn(sd, feats, B, train)
    return normalize_preds(Gs_data, regress(sd, feats, B))


def vit_ess_forward(sd, images, Gs_data, intrinsics=None, train=False, **variant):
    """ViTEss.forward end to end (src/model.py:161-191).  Mutates `intrinsics` like the reference."""
    B = images.shape[0]
    x = preprocess(images)
    if intrinsics is not None:
        intrinsics = update_intrinsics(images.shape[-2:], intrinsics)
    tokens = tokens_from_cnn(cnn_features(sd, x, train))
    return vit_ess_from_tokens(sd, tokens, Gs_data, intrinsics, train, **variant), tokens


# --------------------------------------------------------------------------------------------
# error metrics (SURVEY.md 8d "R,t error metric"; rotation angle as test_matterport.py:41)
# --------------------------------------------------------------------------------------------
def rel_err(a, b):
    a = a.double()
    b = b.double()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def pose_errors(pred, ref):
    """max|dt|/max|t_ref|, max|dq|/max|q_ref| and max rotation angle (rad) on slot 1."""
    t_err = rel_err(pred[:, 1, :3], ref[:, 1, :3])
    q_err = rel_err(pred[:, 1, 3:], ref[:, 1, 3:])
    qa = F.normalize(pred[:, 1, 3:].double(), dim=-1)
    qb = F.normalize(ref[:, 1, 3:].double(), dim=-1)
    ang = 2.0 * torch.acos((qa * qb).sum(-1).abs().clamp(max=1.0))
    return t_err, q_err, float(ang.max())
