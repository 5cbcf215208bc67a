"""RGBDAugmentor -- counterpart of reference src/data_readers/augmentation.py:7-37.

The reference jitters colour on the CPU, one sample at a time, through PIL (torchvision ColorJitter(brightness=.25, contrast=.25,
saturation=.25, hue=.4/3.14) + RandomGrayscale(p=.1)), which SURVEY.md 8f-3 names as the input-pipeline limiter at 8-GPU
rate.  Here the same transforms are plain tensor arithmetic on whatever device the images live on: a DataLoader worker can
call it per sample like the reference, or `augment_batch` can jitter a whole resident [B,2,3,H,W] batch on the GPU (one
parameter draw per PAIR, as in the reference, where the two images are glued side by side before the jitter).
torchvision is not installed here, so its exact random stream / 8-bit PIL rounding is not reproduced: parity of the
augmentation is statistical, not bitwise (it is random data augmentation)."""
import math

import numpy as np
import torch
import torch.nn.functional as F

_GRAY = (0.299, 0.587, 0.114)           # ITU-R 601-2 luma, what PIL / torchvision use


def _gray(rgb):
    """[...,3,H,W] RGB -> [...,1,H,W]"""
    r, g, b = rgb.unbind(dim=-3)
    return (_GRAY[0] * r + _GRAY[1] * g + _GRAY[2] * b).unsqueeze(-3)


def _blend(a, b, ratio):
    return (ratio * a + (1.0 - ratio) * b).clamp_(0.0, 1.0)


def adjust_brightness(rgb, f):
    return _blend(rgb, torch.zeros_like(rgb), f)


def adjust_contrast(rgb, f):
    mean = _gray(rgb).mean(dim=(-3, -2, -1), keepdim=True)
    return _blend(rgb, mean, f)


def adjust_saturation(rgb, f):
    return _blend(rgb, _gray(rgb), f)


def _rgb2hsv(rgb):
    r, g, b = rgb.unbind(dim=-3)
    maxc, minc = rgb.max(dim=-3).values, rgb.min(dim=-3).values
    eqc = maxc == minc
    cr = maxc - minc
    ones = torch.ones_like(maxc)
    s = cr / torch.where(eqc, ones, maxc)
    crd = torch.where(eqc, ones, cr)
    rc, gc, bc = (maxc - r) / crd, (maxc - g) / crd, (maxc - b) / crd
    hr = (maxc == r) * (bc - gc)
    hg = ((maxc == g) & (maxc != r)) * (2.0 + rc - bc)
    hb = ((maxc != g) & (maxc != r)) * (4.0 + gc - rc)
    h = torch.fmod((hr + hg + hb) / 6.0 + 1.0, 1.0)
    return torch.stack((h, s, maxc), dim=-3)


def _hsv2rgb(hsv):
    h, s, v = hsv.unbind(dim=-3)
    i = torch.floor(h * 6.0)
    f = h * 6.0 - i
    i = i.to(torch.int32) % 6
    p = (v * (1.0 - s)).clamp(0.0, 1.0)
    q = (v * (1.0 - f * s)).clamp(0.0, 1.0)
    t = (v * (1.0 - (1.0 - f) * s)).clamp(0.0, 1.0)
    sel = lambda c: torch.stack(c, dim=0).gather(0, i.long().unsqueeze(0)).squeeze(0)   # noqa: E731
    r = sel((v, q, p, p, t, v))
    g = sel((t, v, v, q, p, p))
    b = sel((p, p, t, v, v, q))
    return torch.stack((r, g, b), dim=-3)


def adjust_hue(rgb, shift):
    hsv = _rgb2hsv(rgb)
    h = torch.fmod(hsv[..., 0:1, :, :] + shift + 1.0, 1.0)
    return _hsv2rgb(torch.cat((h, hsv[..., 1:, :, :]), dim=-3))


class RGBDAugmentor:
    """perform augmentation on an image pair: colour jitter + resize to `reshape_size` with the intrinsics rescaled"""

    def __init__(self, reshape_size, datapath=None, generator=None):
        self.reshape_size = list(reshape_size)
        self.p_gray = 0.1
        self.brightness = self.contrast = self.saturation = 0.25
        self.hue = 0.4 / 3.14
        self.generator = generator            # torch.Generator (CPU) for reproducible draws; None = global RNG

    def _rand(self, n=1):
        return torch.rand(n, generator=self.generator)

    def draw(self):
        """one ColorJitter + RandomGrayscale parameter set: (order of the 4 ops, brightness, contrast, saturation, hue, gray?)"""
        order = torch.randperm(4, generator=self.generator).tolist()
        u = self._rand(5).tolist()
        return dict(order=order, b=1 - self.brightness + 2 * self.brightness * u[0],
                    c=1 - self.contrast + 2 * self.contrast * u[1], s=1 - self.saturation + 2 * self.saturation * u[2],
                    h=-self.hue + 2 * self.hue * u[3], gray=u[4] < self.p_gray)

    @staticmethod
    def apply(images, prm):
        """images [...,3,H,W] BGR 0..255 (every leading element gets the SAME parameters) -> same layout"""
        x = images.flip(-3) / 255.0                                   # BGR -> RGB, 0..1
        for op in prm["order"]:
            if op == 0:
                x = adjust_brightness(x, prm["b"])
            elif op == 1:
                x = adjust_contrast_pair(x, prm["c"])
            elif op == 2:
                x = adjust_saturation(x, prm["s"])
            else:
                x = adjust_hue(x, prm["h"])
        if prm["gray"]:
            x = _gray(x).expand_as(x)
        return (255.0 * x).flip(-3).contiguous()

    def color_transform(self, images):
        """images [num,3,H,W]: one parameter draw for all `num` images of the sample (reference :21-26)"""
        return self.apply(images, self.draw())

    def __call__(self, images, poses, intrinsics):
        images = self.color_transform(images)
        sizey, sizex = self.reshape_size
        scalex = sizex / images.shape[-1]
        scaley = sizey / images.shape[-2]
        intrinsics[:, [0, 2]] = scalex * intrinsics[:, [0, 2]]
        intrinsics[:, [1, 3]] = scaley * intrinsics[:, [1, 3]]
        images = F.interpolate(images, size=self.reshape_size)          # nearest, like the reference (:36)
        return images, poses, intrinsics

    def draw_batch(self, B):
        """B parameter rows for rp_augment_pairs: [order0..3, b, c, s, h, gray] (fp32, CPU)"""
        order = torch.argsort(torch.rand(B, 4, generator=self.generator), dim=1).float()      # a uniform random permutation per row
        u = torch.rand(B, 5, generator=self.generator)
        lo = torch.tensor([1 - self.brightness, 1 - self.contrast, 1 - self.saturation, -self.hue])
        span = 2 * torch.tensor([self.brightness, self.contrast, self.saturation, self.hue])
        return torch.cat((order, lo + span * u[:, :4], (u[:, 4:] < self.p_gray).float()), dim=1)

    @staticmethod
    def params_to_dict(row):
        """one draw_batch row as the dict `apply` takes (the checker of the HIP kernel in tests/)"""
        r = row.tolist()
        return dict(order=[int(v) for v in r[:4]], b=r[4], c=r[5], s=r[6], h=r[7], gray=bool(r[8]))

    def augment_batch_hip(self, images_u8, intrinsics, params=None):
        """the GPU-rate path: images_u8 [B,2,H,W,3] uint8 BGR as decoded (what the readers return with
        raw=True), resident on the GPU; one fused HIP pass does the BGR->RGB conversion, the four jitter ops in each pair's
        order, the greyscale draw and the nearest resize (csrc/augment.hip).  intrinsics [B,2,4] are rescaled in place."""
        from .. import ops
        B, _, H, W, _ = images_u8.shape
        sizey, sizex = self.reshape_size
        prm = self.draw_batch(B) if params is None else params
        out = ops.augment_pairs(images_u8, prm.to(images_u8.device, non_blocking=True), sizey, sizex)
        sc = torch.tensor([sizex / W, sizey / H] * 2, dtype=intrinsics.dtype, device=intrinsics.device)
        intrinsics.mul_(sc)
        return out, intrinsics

    def augment_batch(self, images, intrinsics):
        """device-side variant for a resident batch: images [B,2,3,H,W], intrinsics [B,2,4] (modified in place);
        per-pair parameter draws, all arithmetic on images.device"""
        out = torch.empty_like(images)
        for b in range(images.shape[0]):
            out[b] = self.apply(images[b], self.draw())
        sizey, sizex = self.reshape_size
        sc = torch.tensor([sizex / images.shape[-1], sizey / images.shape[-2]] * 2, dtype=intrinsics.dtype, device=intrinsics.device)
        intrinsics.mul_(sc)
        B = images.shape[0]
        out = F.interpolate(out.flatten(0, 1), size=self.reshape_size).view(B, 2, 3, sizey, sizex)
        return out, intrinsics


def adjust_contrast_pair(rgb, f):
    """contrast about the mean luma of the WHOLE sample: the reference glues the two images of a pair side by side into one
    PIL image before ColorJitter (augmentation.py:23-25), so the mean is taken over both"""
    mean = _gray(rgb).mean()
    return _blend(rgb, mean, f)
