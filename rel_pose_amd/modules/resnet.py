"""torchvision-free ResNet-18 trunk with torchvision's module/attribute names.

The reference builds ``torchvision.models.resnet18(pretrained=True)`` and drives
``conv1, bn1, relu, maxpool, layer1, layer2`` by hand (reference src/model.py:31,127-132);
``layer3``/``layer4`` exist only so that checkpoints load and ``train.py`` can freeze them
(reference train.py:60-64).  torchvision is not installed in this image, so the trunk is
restated here with identical ``state_dict`` keys.  Runs on
PyTorch-ROCm (MIOpen) for the convolutions; the BatchNorm + residual-add + ReLU chains around them
are fused HIP kernels (ops.bn_act) -- the CNN front-end is the first "next" row.
"""
import os

import torch
import torch.nn as nn

from ..ops import bn_act, conv2d, conv2d_shared


class BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 3, stride, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(planes, planes, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        # BatchNorm + residual add + ReLU as one fused pass pair on the GPU (rel_pose_amd/csrc/batchnorm.hip)
        # (the hand-written 3x3 convolutions hand the BatchNorm its batch statistics from their epilogue; without a downsample branch the
        # identity path goes through conv1's autograd node, whose input-gradient kernel adds the identity gradient in its epilogue)
        if self.downsample is None:
            y, st, idt = conv2d_shared(self.conv1, x)
        else:
            idt = bn_act(self.downsample[1], conv2d(self.downsample[0], x), relu=False)
            y, st = conv2d(self.conv1, x, want_stats=True)
        y = bn_act(self.bn1, y, stats=st)
        y, st = conv2d(self.conv2, y, want_stats=True)
        return bn_act(self.bn2, y, residual=idt, stats=st)


class ResNet18(nn.Module):
    """Same parameter/buffer names as torchvision.models.resnet18()."""

    def __init__(self, num_classes=1000):
        super().__init__()
        self.inplanes = 64
        self.conv1 = nn.Conv2d(3, 64, 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, 2, 1)
        self.layer1 = self._stage(64, 2, 1)
        self.layer2 = self._stage(128, 2, 2)
        self.layer3 = self._stage(256, 2, 2)
        self.layer4 = self._stage(512, 2, 2)
        self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
        self.fc = nn.Linear(512, num_classes)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.ones_(m.weight)
                nn.init.zeros_(m.bias)

    def _stage(self, planes, blocks, stride):
        down = None
        if stride != 1 or self.inplanes != planes:
            down = nn.Sequential(nn.Conv2d(self.inplanes, planes, 1, stride, bias=False),
                                 nn.BatchNorm2d(planes))
        layers = [BasicBlock(self.inplanes, planes, stride, down)]
        self.inplanes = planes
        for _ in range(1, blocks):
            layers.append(BasicBlock(planes, planes))
        return nn.Sequential(*layers)


WEIGHTS_ENV = "RELPOSE_RESNET18_WEIGHTS"


def resnet18(pretrained=False, **kw):
    """torchvision.models.resnet18(pretrained=...) (reference src/model.py:31).  ImageNet weights cannot be downloaded here:
    with pretrained=True a local torchvision resnet18 state_dict (same keys) is loaded from $RELPOSE_RESNET18_WEIGHTS when
    set; otherwise the trunk keeps its Kaiming init and `net.pretrained_loaded` stays False so that train.py can warn --
    every rel_pose checkpoint overrides these weights anyway, but a from-scratch run does not reproduce the reference's
    fine-tuning recipe without them."""
    net = ResNet18(**kw)
    net.pretrained_loaded = False
    path = os.environ.get(WEIGHTS_ENV) if pretrained else None
    if path:
        sd = torch.load(path, map_location="cpu", weights_only=True)
        sd = sd.get("state_dict", sd) if isinstance(sd, dict) else sd
        missing, unexpected = net.load_state_dict(sd, strict=False)
        bad = [k for k in missing if not k.startswith("fc.")] + [k for k in unexpected if not k.startswith("fc.")]
        if bad:
            raise RuntimeError("%s=%s is not a torchvision resnet18 state_dict (mismatched keys: %s)" % (WEIGHTS_ENV, path, bad[:5]))
        net.pretrained_loaded = True
    return net
