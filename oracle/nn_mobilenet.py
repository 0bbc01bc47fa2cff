"""fp32 PyTorch restatement of torchvision==0.16.2 ``mobilenet_v3_large`` (eval
mode) as used for the face-attribute classifier.  TEST ORACLE -- parity unpinned.

Reference call sites: exp-1-debias-gender/1-main-debias.py:929-935 (build, last FC
replaced by Linear(1280, 80)), :1369-1371 (forward + gender slice);
exp-3: 6 logits, exp-4: 8 logits.  Module names follow torchvision
(``features.N``, ``features.N.block.M``, ``classifier.{0,3}``) so the
reference's classifier ``.pt`` state-dict loads unchanged.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

# (kernel, expanded, out, use_se, activation, stride) -- MobileNetV3-Large table
SETTINGS = [
    (3, 16, 16, False, "RE", 1), (3, 64, 24, False, "RE", 2), (3, 72, 24, False, "RE", 1),
    (5, 72, 40, True, "RE", 2), (5, 120, 40, True, "RE", 1), (5, 120, 40, True, "RE", 1),
    (3, 240, 80, False, "HS", 2), (3, 200, 80, False, "HS", 1), (3, 184, 80, False, "HS", 1),
    (3, 184, 80, False, "HS", 1), (3, 480, 112, True, "HS", 1), (3, 672, 112, True, "HS", 1),
    (5, 672, 160, True, "HS", 2), (5, 960, 160, True, "HS", 1), (5, 960, 160, True, "HS", 1),
]


def make_divisible(v, divisor=8):
    new_v = max(divisor, int(v + divisor / 2) // divisor * divisor)
    if new_v < 0.9 * v:
        new_v += divisor
    return new_v


class ConvBNAct(nn.Sequential):
    def __init__(self, cin, cout, k, stride=1, groups=1, act=None):
        layers = [nn.Conv2d(cin, cout, k, stride, (k - 1) // 2, groups=groups, bias=False),
                  nn.BatchNorm2d(cout, eps=0.001, momentum=0.01)]
        if act == "RE":
            layers.append(nn.ReLU())
        elif act == "HS":
            layers.append(nn.Hardswish())
        super().__init__(*layers)


class SqueezeExcitation(nn.Module):
    def __init__(self, c, squeeze):
        super().__init__()
        self.fc1 = nn.Conv2d(c, squeeze, 1)
        self.fc2 = nn.Conv2d(squeeze, c, 1)

    def forward(self, x):
        s = x.mean(dim=(2, 3), keepdim=True)
        s = F.hardsigmoid(self.fc2(F.relu(self.fc1(s))))
        return x * s


class InvertedResidual(nn.Module):
    def __init__(self, cin, k, exp, cout, se, act, stride):
        super().__init__()
        self.use_res = stride == 1 and cin == cout
        layers = []
        if exp != cin:
            layers.append(ConvBNAct(cin, exp, 1, act=act))
        layers.append(ConvBNAct(exp, exp, k, stride=stride, groups=exp, act=act))
        if se:
            layers.append(SqueezeExcitation(exp, make_divisible(exp // 4, 8)))
        layers.append(ConvBNAct(exp, cout, 1, act=None))
        self.block = nn.Sequential(*layers)

    def forward(self, x):
        y = self.block(x)
        return x + y if self.use_res else y


class MobileNetV3Large(nn.Module):
    def __init__(self, num_classes=80):
        super().__init__()
        feats = [ConvBNAct(3, 16, 3, stride=2, act="HS")]
        cin = 16
        for k, exp, cout, se, act, s in SETTINGS:
            feats.append(InvertedResidual(cin, k, exp, cout, se, act, s))
            cin = cout
        feats.append(ConvBNAct(cin, 960, 1, act="HS"))
        self.features = nn.Sequential(*feats)
        self.classifier = nn.Sequential(nn.Linear(960, 1280), nn.Hardswish(), nn.Dropout(0.2), nn.Linear(1280, num_classes))

    def forward(self, x):
        x = self.features(x).mean(dim=(2, 3))
        return self.classifier(x)


def randomize_bn(model: nn.Module, seed: int = 0):
    """Give BatchNorm non-trivial running stats/affine so synthetic tests exercise folding."""
    g = torch.Generator().manual_seed(seed)
    for m in model.modules():
        if isinstance(m, nn.BatchNorm2d):
            m.running_mean.copy_(torch.randn(m.num_features, generator=g) * 0.1)
            m.running_var.copy_(torch.rand(m.num_features, generator=g) * 0.5 + 0.75)
            m.weight.data.copy_(torch.rand(m.num_features, generator=g) * 0.5 + 0.75)
            m.bias.data.copy_(torch.randn(m.num_features, generator=g) * 0.1)
