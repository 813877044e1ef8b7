"""S3D-G backbone as parameter container + layer plan.

Follows /root/reference/models/s3dg.py:6-34 (BasicConv3d = conv(no bias) + BN(eps 1e-3, momentum 1e-3) + ReLU), :36-72
(sep_conv = (1,k,k) BasicConv3d with the stride on ALL of T,H,W, then (k,1,1) BasicConv3d, then self-gating
x*sigmoid(Conv1x1x1_bias(mean(x)))), :74-99 (sep_inc: 4 branches, channel concat), :103-133 (stage list).  There is no
depth-wise convolution anywhere (groups=1).  Branch outputs are written straight into their channel slice of the block
output (no concat pass).
"""
from collections import OrderedDict

from torch import nn

from ..engine import ConvBN, ConvBNGroup, Gate, Plan, Pool


class BasicConv3d(nn.Module):
    def __init__(self, cin, cout, kernel_size=1, stride=1, padding=0):
        super().__init__()
        self.conv3d = nn.Conv3d(cin, cout, kernel_size=kernel_size, stride=stride, padding=padding, bias=False)
        self.bn = nn.BatchNorm3d(cout, eps=1e-3, momentum=0.001, affine=True)
        k = self.conv3d.kernel_size
        self.geom = (tuple(k), tuple(self.conv3d.stride), tuple(self.conv3d.padding))


class sep_conv(nn.Module):
    def __init__(self, cin, cout, kernel_size, stride=1, padding=0):
        super().__init__()
        down = BasicConv3d(cin, cout, (1, kernel_size, kernel_size), stride=stride, padding=(0, padding, padding))
        up = BasicConv3d(cout, cout, (kernel_size, 1, 1), stride=1, padding=(padding, 0, 0))
        self.sep_conv = nn.Sequential(down, up)
        self.excitation = nn.Conv3d(cout, cout, 1)


class sep_inc(nn.Module):
    def __init__(self, cin, o):
        super().__init__()
        self.widths = (o[0], o[2], o[4], o[5])
        self.branch0 = BasicConv3d(cin, o[0])
        self.branch1 = nn.Sequential(BasicConv3d(cin, o[1]), sep_conv(o[1], o[2], 3, 1, 1))
        self.branch2 = nn.Sequential(BasicConv3d(cin, o[3]), sep_conv(o[3], o[4], 3, 1, 1))
        self.branch3 = nn.Sequential(nn.MaxPool3d(kernel_size=3, stride=1, padding=1), BasicConv3d(cin, o[5]))


_POOLS = {"maxPool1": ((1, 3, 3), (1, 2, 2), (0, 1, 1)), "maxPool2": ((1, 3, 3), (1, 2, 2), (0, 1, 1)),
          "maxPool3": ((3, 3, 3), (2, 2, 2), (1, 1, 1)), "maxpool4": ((2, 2, 2), (2, 2, 2), (0, 0, 0))}


class S3D_G(nn.Module):
    classifier_names = ("fc",)

    def __init__(self, num_classes=400, drop_prob=0.5, in_channel=3, gate=True):
        super().__init__()
        if not gate:
            raise NotImplementedError("gate=False is not used by the pretext configs")
        inc = sep_inc
        self.feature = nn.Sequential(OrderedDict([
            ("sepConv1", sep_conv(in_channel, 64, 7, 2, 3)),
            ("maxPool1", nn.MaxPool3d((1, 3, 3), (1, 2, 2), (0, 1, 1))),
            ("basicConv3d", BasicConv3d(64, 64, 1, 1)),
            ("sep_conv2", sep_conv(64, 192, 3, 1, 1)),
            ("maxPool2", nn.MaxPool3d((1, 3, 3), (1, 2, 2), (0, 1, 1))),
            ("sepInc_3b", inc(192, [64, 96, 128, 16, 32, 32])),
            ("sepInc_3c", inc(256, [128, 128, 192, 32, 96, 64])),
            ("maxPool3", nn.MaxPool3d((3, 3, 3), (2, 2, 2), (1, 1, 1))),
            ("sepInc_4b", inc(480, [192, 96, 208, 16, 48, 64])),
            ("sepInc_4c", inc(512, [160, 112, 224, 24, 64, 64])),
            ("sepInc_4d", inc(512, [128, 128, 256, 24, 64, 64])),
            ("sepInc_4e", inc(512, [112, 144, 288, 32, 64, 64])),
            ("sepInc_4f", inc(528, [256, 160, 320, 32, 128, 128])),
            ("maxpool4", nn.MaxPool3d((2, 2, 2), (2, 2, 2), (0, 0, 0))),
            ("sepInc_5b", inc(832, [256, 160, 320, 32, 128, 128])),
            ("sepInc_5c", inc(832, [384, 192, 384, 48, 128, 128])),
        ]))
        self.fc = nn.Linear(1024, num_classes)      # unused by get_feature(); state-dict contract

    def plan(self) -> Plan:
        nodes, counter = [], [1]

        def new():
            counter[0] += 1
            return counter[0] - 1

        def basic(m: BasicConv3d, src, into=None, branch=0):
            dst = new()
            k, s, p = m.geom
            nodes.append(ConvBN(m.conv3d, m.bn, src, dst, k, s, p, relu=True, into=into, branch=branch))
            return dst

        def sep(m: sep_conv, src, into=None, branch=0):
            a = basic(m.sep_conv[0], src, branch=branch)
            b = basic(m.sep_conv[1], a, branch=branch)
            dst = new()
            nodes.append(Gate(m.excitation, b, dst, into=into, branch=branch))
            return dst

        cur = 0
        for name, m in self.feature.named_children():
            if isinstance(m, sep_conv):
                cur = sep(m, cur)
            elif isinstance(m, BasicConv3d):
                cur = basic(m, cur)
            elif isinstance(m, nn.MaxPool3d):
                dst = new()
                nodes.append(Pool(cur, dst, *_POOLS[name]))
                cur = dst
            else:
                total = sum(m.widths)
                cat = new()
                offs = [0, m.widths[0], m.widths[0] + m.widths[1], m.widths[0] + m.widths[1] + m.widths[2]]
                # the three pointwise convs that read the block input run as one GEMM when their filters are adjacent
                first = len(nodes)
                basic(m.branch0, cur, into=(cat, offs[0], total))
                b1 = basic(m.branch1[0], cur)
                b2 = basic(m.branch2[0], cur)
                nodes[first:] = [ConvBNGroup(nodes[first:])]
                # branches 1-3 are independent from here on: as nodes of a captured HIP graph they run side by side
                # (engine.BranchStreams)
                sep(m.branch1[1], b1, into=(cat, offs[1], total), branch=1)
                sep(m.branch2[1], b2, into=(cat, offs[2], total), branch=2)
                pooled = new()
                nodes.append(Pool(cur, pooled, (3, 3, 3), (1, 1, 1), (1, 1, 1), branch=3))
                basic(m.branch3[1], pooled, into=(cat, offs[3], total), branch=3)
                cur = cat
        return Plan(nodes, input_slot=0, output_slot=cur)

    def adjacent_parameters(self):
        """Parameter groups (names relative to this module) the flat layout should place back to back: the filters of each
        inception block's three pointwise convolutions on the block input (models/s3dg.py:80-88)."""
        return tuple((f"feature.{name}.branch0.conv3d.weight", f"feature.{name}.branch1.0.conv3d.weight",
                      f"feature.{name}.branch2.0.conv3d.weight")
                     for name, m in self.feature.named_children() if isinstance(m, sep_inc))
