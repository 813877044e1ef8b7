"""C3D backbone (parameter container + layer plan).

State-dict contract and architecture follow /root/reference/models/c3d.py:21-52,111-150: eight 3x3x3 convolutions
(stride 1, pad 1, WITH bias) each followed by BatchNorm3d(eps 1e-5, momentum 0.1) + ReLU; max-pools (1,2,2) after
block 1 and (2,2,2) after blocks 2, 3b, 4b; get_feature() stops after relu5b (no pool5, no linear).
The nn.Conv3d / nn.BatchNorm3d modules are used only as parameter holders (same names, shapes and default
initialisation as the reference); the arithmetic runs through rspnet_amd.engine.
"""
from torch import nn

from ..engine import ConvBN, Plan

# (suffix, Cin, Cout, pool after the block)
_BLOCKS = (
    ("1", 3, 64, ((1, 2, 2), (1, 2, 2))),
    ("2", 64, 128, ((2, 2, 2), (2, 2, 2))),
    ("3a", 128, 256, None),
    ("3b", 256, 256, ((2, 2, 2), (2, 2, 2))),
    ("4a", 256, 512, None),
    ("4b", 512, 512, ((2, 2, 2), (2, 2, 2))),
    ("5a", 512, 512, None),
    ("5b", 512, 512, None),
)


class C3D(nn.Module):
    feat_dim = 512
    classifier_names = ("linear",)

    def __init__(self, with_classifier=True, return_conv=False, num_classes=101):
        super().__init__()
        if return_conv:
            raise NotImplementedError("return_conv is a VCOP fine-tune option, not on the pretext path")
        self.with_classifier = with_classifier
        self.num_classes = num_classes
        for name, cin, cout, _ in _BLOCKS:
            setattr(self, "conv" + name, nn.Conv3d(cin, cout, kernel_size=3, padding=1))
            setattr(self, "bn" + name, nn.BatchNorm3d(cout))
        if with_classifier:
            self.linear = nn.Linear(512, num_classes)   # never used by get_feature(); kept for the state dict

    def plan(self) -> Plan:
        nodes = [ConvBN(getattr(self, "conv" + name), getattr(self, "bn" + name), src=i, dst=i + 1, k=(3, 3, 3),
                        p=(1, 1, 1), relu=True, pool=pool)
                 for i, (name, _, _, pool) in enumerate(_BLOCKS)]
        return Plan(nodes, input_slot=0, output_slot=len(nodes))
