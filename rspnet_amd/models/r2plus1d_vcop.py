"""R(2+1)D backbone (VCOP variant) as parameter container + layer plan.

Follows /root/reference/models/r2plus1d_vcop.py:13-72 (SpatioTemporalConv = (1,k,k) conv → BN → ReLU → (k,1,1) conv with
M = floor(k^3·Cin·Cout / (k^2·Cin + k·Cout)) mid channels, no bias), :75-123 (ResBlock; the down-sampling shortcut is
itself a factored 1x1x1 stride-2 pair followed by BN), :177-224 (net, get_feature).  Default PyTorch initialisation.
"""
import math

from torch import nn
from torch.nn.modules.utils import _triple

from ..engine import ConvBN, Plan


class SpatioTemporalConv(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, bias=False):
        super().__init__()
        k, s, p = _triple(kernel_size), _triple(stride), _triple(padding)
        mid = int(math.floor((k[0] * k[1] * k[2] * in_channels * out_channels) /
                             (k[1] * k[2] * in_channels + k[0] * out_channels)))
        self.geom = (((1, k[1], k[2]), (1, s[1], s[2]), (0, p[1], p[2])), ((k[0], 1, 1), (s[0], 1, 1), (p[0], 0, 0)))
        self.spatial_conv = nn.Conv3d(in_channels, mid, self.geom[0][0], stride=self.geom[0][1], padding=self.geom[0][2],
                                      bias=bias)
        self.bn = nn.BatchNorm3d(mid)
        self.temporal_conv = nn.Conv3d(mid, out_channels, self.geom[1][0], stride=self.geom[1][1],
                                       padding=self.geom[1][2], bias=bias)


class SpatioTemporalResBlock(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size, downsample=False):
        super().__init__()
        self.downsample = downsample
        padding = kernel_size // 2
        if downsample:
            self.downsampleconv = SpatioTemporalConv(in_channels, out_channels, 1, stride=2)
            self.downsamplebn = nn.BatchNorm3d(out_channels)
            self.conv1 = SpatioTemporalConv(in_channels, out_channels, kernel_size, padding=padding, stride=2)
        else:
            self.conv1 = SpatioTemporalConv(in_channels, out_channels, kernel_size, padding=padding)
        self.bn1 = nn.BatchNorm3d(out_channels)
        self.conv2 = SpatioTemporalConv(out_channels, out_channels, kernel_size, padding=padding)
        self.bn2 = nn.BatchNorm3d(out_channels)


class SpatioTemporalResLayer(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size, layer_size, downsample=False):
        super().__init__()
        self.block1 = SpatioTemporalResBlock(in_channels, out_channels, kernel_size, downsample)
        self.blocks = nn.ModuleList([SpatioTemporalResBlock(out_channels, out_channels, kernel_size)
                                     for _ in range(layer_size - 1)])


class R2Plus1DNet(nn.Module):
    classifier_names = ("linear",)

    def __init__(self, layer_sizes, with_classifier=False, return_conv=False, num_classes=101):
        super().__init__()
        if return_conv:
            raise NotImplementedError("return_conv is a VCOP fine-tune option, not on the pretext path")
        self.conv1 = SpatioTemporalConv(3, 64, (3, 7, 7), stride=(1, 2, 2), padding=(1, 3, 3))
        self.bn1 = nn.BatchNorm3d(64)
        self.conv2 = SpatioTemporalResLayer(64, 64, 3, layer_sizes[0])
        self.conv3 = SpatioTemporalResLayer(64, 128, 3, layer_sizes[1], downsample=True)
        self.conv4 = SpatioTemporalResLayer(128, 256, 3, layer_sizes[2], downsample=True)
        self.conv5 = SpatioTemporalResLayer(256, 512, 3, layer_sizes[3], downsample=True)
        if with_classifier:
            self.linear = nn.Linear(512, num_classes)

    def plan(self) -> Plan:
        nodes, counter = [], [1]

        def new():
            counter[0] += 1
            return counter[0] - 1

        def st(conv: SpatioTemporalConv, outer_bn, src, relu, residual=None):
            """factored conv followed by the BN that the caller applies to its output; returns the output slot"""
            (sk, ss, sp), (tk, ts, tp) = conv.geom
            mid, dst = new(), new()
            m = conv.spatial_conv.weight.shape[0]
            # (the stem's (1,7,7) stride-2 convolution stays on the 49-tap patch kernel, conv_stem.hip: run on virtual pixels
            #  (engine.VirtualStem) it measured the same step time, 76.51 vs 76.52 ms)
            nodes.append(ConvBN(conv.spatial_conv, conv.bn, src, mid, sk, ss, sp, relu=True, cout_pad=(m + 3) // 4 * 4))
            nodes.append(ConvBN(conv.temporal_conv, outer_bn, mid, dst, tk, ts, tp, relu=relu, residual=residual))
            return dst

        cur = st(self.conv1, self.bn1, 0, True)
        for layer in (self.conv2, self.conv3, self.conv4, self.conv5):
            for blk in [layer.block1] + list(layer.blocks):
                r = st(blk.conv1, blk.bn1, cur, True)
                short = cur
                if blk.downsample:
                    short = st(blk.downsampleconv, blk.downsamplebn, cur, False)
                cur = st(blk.conv2, blk.bn2, r, True, residual=short)      # relu(shortcut + bn2(conv2(res)))
        return Plan(nodes, input_slot=0, output_slot=cur)
