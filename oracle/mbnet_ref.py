"""ORACLE (test infrastructure, never imported by the product path): torch-CPU fp32 restatements of the two remaining
classifiers the reference can build (src/tt100k/pipeline/e2e.py:324-329):

    build_classifier('mobilenetv2')  = torchvision ``mobilenet_v2(weights=None)``,  classifier[1] = Linear(1280, num_classes)
    build_classifier('efficientnet') = torchvision ``efficientnet_b0(weights=None)``, classifier[1] = Linear(1280, num_classes)

both in ``eval()`` mode.  torchvision is not installed in the build container, so the architectures
are restated from their published definitions (torchvision/models/mobilenetv2.py: ``InvertedResidual`` with ReLU6,
settings [t, c, n, s] = [1,16,1,1] [6,24,2,2] [6,32,3,2] [6,64,4,2] [6,96,3,1] [6,160,3,2] [6,320,1,1], features.18 =
1x1 to 1280; torchvision/models/efficientnet.py: ``MBConv`` with SiLU, ``SqueezeExcitation(expanded, max(1, in // 4))``,
B0 settings (expand, kernel, stride, in, out, layers) = (1,3,1,32,16,1) (6,3,2,16,24,2) (6,5,2,24,40,2) (6,3,2,40,80,3)
(6,5,1,80,112,3) (6,5,2,112,192,4) (6,3,1,192,320,1), features.8 = 1x1 to 1280; stochastic depth is the identity in eval
mode) with torchvision's module nesting, so the ``state_dict`` keys are torchvision's and a real checkpoint loads.
PARITY UNPINNED: the reference holds no weights or outputs for either architecture; the pre-processing in front of them
is the pinned one of e2e.py:366-370 (oracle/pil_resize_ref.py).
"""
from typing import List, Tuple

import numpy as np
import torch
import torch.nn as nn

from . import pil_resize_ref

MEAN, STD = 0.18, 0.34  # e2e.py:369


def _cna(inp, out, k=3, stride=1, groups=1, act=None):
    """torchvision.ops.Conv2dNormActivation: Sequential(conv (no bias), BatchNorm2d, activation)"""
    layers = [nn.Conv2d(inp, out, k, stride, (k - 1) // 2, groups=groups, bias=False), nn.BatchNorm2d(out)]
    if act is not None:
        layers.append(act())
    return nn.Sequential(*layers)


class InvertedResidual(nn.Module):  # mobilenetv2.py
    def __init__(self, inp, oup, stride, expand_ratio):
        super().__init__()
        hidden = int(round(inp * expand_ratio))
        self.use_res_connect = stride == 1 and inp == oup
        layers = []
        if expand_ratio != 1:
            layers.append(_cna(inp, hidden, 1, act=lambda: nn.ReLU6(inplace=True)))
        layers.extend([_cna(hidden, hidden, 3, stride, groups=hidden, act=lambda: nn.ReLU6(inplace=True)),
                       nn.Conv2d(hidden, oup, 1, 1, 0, bias=False), nn.BatchNorm2d(oup)])
        self.conv = nn.Sequential(*layers)

    def forward(self, x):
        return x + self.conv(x) if self.use_res_connect else self.conv(x)


class MobileNetV2(nn.Module):
    def __init__(self, num_classes: int):
        super().__init__()
        cfg = [[1, 16, 1, 1], [6, 24, 2, 2], [6, 32, 3, 2], [6, 64, 4, 2], [6, 96, 3, 1], [6, 160, 3, 2], [6, 320, 1, 1]]
        feats = [_cna(3, 32, 3, 2, act=lambda: nn.ReLU6(inplace=True))]
        inp = 32
        for t, c, n, s in cfg:
            for i in range(n):
                feats.append(InvertedResidual(inp, c, s if i == 0 else 1, t))
                inp = c
        feats.append(_cna(inp, 1280, 1, act=lambda: nn.ReLU6(inplace=True)))
        self.features = nn.Sequential(*feats)
        self.classifier = nn.Sequential(nn.Dropout(0.2), nn.Linear(1280, num_classes))

    def forward(self, x):
        x = self.features(x)
        x = torch.flatten(nn.functional.adaptive_avg_pool2d(x, (1, 1)), 1)
        return self.classifier(x)


class SqueezeExcitation(nn.Module):  # torchvision.ops.misc
    def __init__(self, inp, squeeze):
        super().__init__()
        self.avgpool = nn.AdaptiveAvgPool2d(1)
        self.fc1 = nn.Conv2d(inp, squeeze, 1)
        self.fc2 = nn.Conv2d(squeeze, inp, 1)
        self.activation = nn.SiLU(inplace=True)
        self.scale_activation = nn.Sigmoid()

    def forward(self, x):
        s = self.scale_activation(self.fc2(self.activation(self.fc1(self.avgpool(x)))))
        return s * x


class MBConv(nn.Module):  # efficientnet.py
    def __init__(self, expand, k, stride, inp, out):
        super().__init__()
        self.use_res_connect = stride == 1 and inp == out
        exp = inp * expand
        layers = []
        if exp != inp:
            layers.append(_cna(inp, exp, 1, act=lambda: nn.SiLU(inplace=True)))
        layers.append(_cna(exp, exp, k, stride, groups=exp, act=lambda: nn.SiLU(inplace=True)))
        layers.append(SqueezeExcitation(exp, max(1, inp // 4)))
        layers.append(_cna(exp, out, 1, act=None))
        self.block = nn.Sequential(*layers)

    def forward(self, x):
        y = self.block(x)
        return x + y if self.use_res_connect else y   # StochasticDepth is the identity in eval mode


class EfficientNetB0(nn.Module):
    def __init__(self, num_classes: int):
        super().__init__()
        cfg = [(1, 3, 1, 32, 16, 1), (6, 3, 2, 16, 24, 2), (6, 5, 2, 24, 40, 2), (6, 3, 2, 40, 80, 3), (6, 5, 1, 80, 112, 3),
               (6, 5, 2, 112, 192, 4), (6, 3, 1, 192, 320, 1)]
        feats = [_cna(3, 32, 3, 2, act=lambda: nn.SiLU(inplace=True))]
        for e, k, s, i, o, n in cfg:
            feats.append(nn.Sequential(*[MBConv(e, k, s if j == 0 else 1, i if j == 0 else o, o) for j in range(n)]))
        feats.append(_cna(320, 1280, 1, act=lambda: nn.SiLU(inplace=True)))
        self.features = nn.Sequential(*feats)
        self.avgpool = nn.AdaptiveAvgPool2d(1)
        self.classifier = nn.Sequential(nn.Dropout(0.2), nn.Linear(1280, num_classes))

    def forward(self, x):
        return self.classifier(torch.flatten(self.avgpool(self.features(x)), 1))


def build(arch: str, num_classes: int, state_dict=None) -> nn.Module:
    m = MobileNetV2(num_classes) if arch == "mobilenetv2" else EfficientNetB0(num_classes)
    if state_dict is not None:
        m.load_state_dict(state_dict, strict=True)
    return m.eval()


def seeded_state_dict(arch: str, num_classes: int, seed: int = 0, gain: float = None):
    """torchvision-named state_dict with seeded weights whose activations stay O(1) through the depth of the net (the
    squeeze-excitation gates of EfficientNet halve every block's signal: a larger conv gain compensates)."""
    g = torch.Generator().manual_seed(seed)
    gain = gain if gain is not None else (2.0 if arch == "mobilenetv2" else 2.33)
    m = build(arch, num_classes)
    sd = m.state_dict()
    out = {}
    for k, v in sd.items():
        if k.endswith("num_batches_tracked"):
            out[k] = v.clone()
        elif k.endswith("running_var"):
            out[k] = torch.empty_like(v).uniform_(0.75, 1.25, generator=g)
        elif k.endswith("running_mean"):
            out[k] = torch.randn(v.shape, generator=g) * 0.1
        elif v.dim() == 4:
            fan = v.shape[1] * v.shape[2] * v.shape[3]
            out[k] = torch.randn(v.shape, generator=g) * (gain / fan) ** 0.5
        elif v.dim() == 2:
            out[k] = torch.randn(v.shape, generator=g) * (1.0 / v.shape[1]) ** 0.5
        elif k.endswith(".weight"):   # BatchNorm gamma
            out[k] = torch.empty_like(v).uniform_(0.75, 1.25, generator=g)
        else:                         # biases
            out[k] = torch.randn(v.shape, generator=g) * 0.1
    return out


def preprocess(rois_bgr: List[np.ndarray], size: int = 64) -> torch.Tensor:
    """e2e.py:383-391: BGR->RGB, PIL Resize((64,64)) bilinear+antialias (uint8), ToTensor, Normalize."""
    return torch.from_numpy(np.stack([pil_resize_ref.classifier_input(r, size) for r in rois_bgr]))


def predict_batch(model: nn.Module, rois_bgr: List[np.ndarray], size: int = 64) -> Tuple[np.ndarray, np.ndarray]:
    """PyTorchClassifier.predict_batch (e2e.py:378-396): softmax probabilities and arg-max."""
    if len(rois_bgr) == 0:
        return np.array([]), np.array([])
    with torch.no_grad():
        probs = torch.softmax(model(preprocess(rois_bgr, size)), 1).numpy()
    return np.argmax(probs, 1), probs
