"""3-D ResNet backbones (ResNet-18/34 with BasicBlock, shortcut type B) as parameter containers + layer plans.

Architecture and state-dict names follow /root/reference/models/resnet.py:48-77 (BasicBlock), :119-183 (ResNet,
_make_layer), :203-213 (get_feature): stem Conv3d 7x7x7 stride (1,2,2) pad 3 (no bias) → BN → ReLU → MaxPool3d(3, stride 2,
pad 1); four stages of BasicBlocks (3x3x3 convs without bias, BN, residual add, ReLU); down-sampling blocks use a 1x1x1
stride-2 conv + BN shortcut.  Initialisation as the reference: Kaiming-normal(fan_out) convs, BN weight 1 / bias 0.
"""
from torch import nn

from ..engine import ConvBN, Plan, Pool


def _conv(cin, cout, k, stride=1, padding=0):
    return nn.Conv3d(cin, cout, kernel_size=k, stride=stride, padding=padding, bias=False)


class BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = _conv(inplanes, planes, 3, stride, 1)
        self.bn1 = nn.BatchNorm3d(planes)
        self.conv2 = _conv(planes, planes, 3, 1, 1)
        self.bn2 = nn.BatchNorm3d(planes)
        self.downsample = downsample
        self.stride = stride


class Bottleneck(nn.Module):
    """1x1x1 -> 3x3x3 (stride) -> 1x1x1 (x4 channels), models/resnet.py:80-116."""
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = _conv(inplanes, planes, 1)
        self.bn1 = nn.BatchNorm3d(planes)
        self.conv2 = _conv(planes, planes, 3, stride, 1)
        self.bn2 = nn.BatchNorm3d(planes)
        self.conv3 = _conv(planes, planes * 4, 1)
        self.bn3 = nn.BatchNorm3d(planes * 4)
        self.downsample = downsample
        self.stride = stride


class ResNet(nn.Module):
    classifier_names = ("fc",)

    def __init__(self, block, layers, sample_size=112, sample_duration=16, shortcut_type="B", num_classes=400):
        super().__init__()
        if shortcut_type != "B":
            raise NotImplementedError("shortcut type A (zero-padded identity) is not used by the pretext configs")
        self.inplanes = 64
        self.conv1 = _conv(3, 64, 7, (1, 2, 2), (3, 3, 3))
        self.bn1 = nn.BatchNorm3d(64)
        self.layer1 = self._make_layer(block, 64, layers[0])
        self.layer2 = self._make_layer(block, 128, layers[1], stride=2)
        self.layer3 = self._make_layer(block, 256, layers[2], stride=2)
        self.layer4 = self._make_layer(block, 512, layers[3], stride=2)
        self.fc = nn.Linear(512 * block.expansion, num_classes)     # unused by get_feature(); state-dict contract
        for m in self.modules():
            if isinstance(m, nn.Conv3d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out")
            elif isinstance(m, nn.BatchNorm3d):
                m.weight.data.fill_(1)
                m.bias.data.zero_()

    def _make_layer(self, block, planes, blocks, stride=1):
        downsample = None
        if stride != 1 or self.inplanes != planes * block.expansion:
            downsample = nn.Sequential(_conv(self.inplanes, planes * block.expansion, 1, stride),
                                       nn.BatchNorm3d(planes * block.expansion))
        layers = [block(self.inplanes, planes, stride, downsample)]
        self.inplanes = planes * block.expansion
        layers += [block(self.inplanes, planes) for _ in range(1, blocks)]
        return nn.Sequential(*layers)

    def plan(self) -> Plan:
        nodes = [ConvBN(self.conv1, self.bn1, 0, 1, (7, 7, 7), (1, 2, 2), (3, 3, 3), relu=True, virtual_w=True),
                 Pool(1, 2, (3, 3, 3), (2, 2, 2), (1, 1, 1))]
        cur, nxt = 2, 3
        for layer in (self.layer1, self.layer2, self.layer3, self.layer4):
            for blk in layer:
                s = (blk.stride,) * 3
                one, zero = (1, 1, 1), (0, 0, 0)
                res = cur
                if blk.downsample is not None:
                    res = nxt
                    nxt += 1
                    nodes.append(ConvBN(blk.downsample[0], blk.downsample[1], cur, res, one, s, zero, relu=False))
                if isinstance(blk, Bottleneck):
                    a, b, out = nxt, nxt + 1, nxt + 2
                    nxt += 3
                    nodes.append(ConvBN(blk.conv1, blk.bn1, cur, a, one, one, zero, relu=True))
                    nodes.append(ConvBN(blk.conv2, blk.bn2, a, b, (3, 3, 3), s, one, relu=True))
                    nodes.append(ConvBN(blk.conv3, blk.bn3, b, out, one, one, zero, relu=True, residual=res))
                else:
                    mid, out = nxt, nxt + 1
                    nxt += 2
                    nodes.append(ConvBN(blk.conv1, blk.bn1, cur, mid, (3, 3, 3), s, one, relu=True))
                    nodes.append(ConvBN(blk.conv2, blk.bn2, mid, out, (3, 3, 3), one, one, relu=True, residual=res))
                cur = out
        return Plan(nodes, input_slot=0, output_slot=cur)


def resnet18(**kwargs):
    return ResNet(BasicBlock, [2, 2, 2, 2], **kwargs)


def resnet34(**kwargs):
    return ResNet(BasicBlock, [3, 4, 6, 3], **kwargs)


def resnet50(**kwargs):
    return ResNet(Bottleneck, [3, 4, 6, 3], **kwargs)
