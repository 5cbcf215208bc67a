"""CNN tail: 128->192 residual unit with a valid 5x5 second conv (28x28 -> 24x24).

Mirrors the parameter names of reference src/modules/extractor.py:5-65 for the only
configuration rel_pose instantiates (norm_fn='batch', stride=1, kernel_size=5;
reference src/model.py:33): conv1, conv2, norm1..3, downsample.{0,1} with downsample.1
aliasing norm3.
"""
import torch.nn as nn

from ..ops import bn_act, conv2d


class ResidualBlock(nn.Module):
    def __init__(self, in_planes, planes, norm_fn="batch", stride=1, kernel_size=1):
        super().__init__()
        if norm_fn != "batch" or stride != 1:
            raise NotImplementedError("rel_pose only uses norm_fn='batch', stride=1")
        self.conv1 = nn.Conv2d(in_planes, planes, 3, padding=1)
        if kernel_size > 1:
            self.conv2 = nn.Conv2d(planes, planes, kernel_size)
        else:
            self.conv2 = nn.Conv2d(planes, planes, 3, padding=1)
        self.relu = nn.ReLU(inplace=True)
        self.norm1 = nn.BatchNorm2d(planes)
        self.norm2 = nn.BatchNorm2d(planes)
        self.downsample = None
        if kernel_size > 1:
            self.norm3 = nn.BatchNorm2d(planes)
            self.downsample = nn.Sequential(nn.Conv2d(in_planes, planes, kernel_size), self.norm3)

    def forward(self, x):
        y, st = conv2d(self.conv1, x, want_stats=True)          # (exact fp32: the own 128 -> 192 convolution hands norm1 its batch statistics)
        y = bn_act(self.norm1, y, stats=st)
        y = bn_act(self.norm2, conv2d(self.conv2, y))
        if self.downsample is not None:           # relu(norm3(conv(x)) + y): the add and the ReLU ride on norm3's pass
            return bn_act(self.downsample[1], conv2d(self.downsample[0], x), residual=y)
        return (x + y).relu_()
