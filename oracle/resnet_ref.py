"""ORACLE (test infrastructure, never imported by the product path): torch-CPU fp32 restatement of the classifier the
reference builds with ``build_classifier('resnet18', num_classes)`` (src/tt100k/pipeline/e2e.py:320-323): torchvision's
``resnet18(weights=None)`` with ``fc = Linear(512, num_classes)``, ``eval()`` mode.

torchvision is not installed in the build container, so the architecture is restated from its published
definition (torchvision/models/resnet.py: ``BasicBlock`` = conv3x3(stride) + BN + ReLU, conv3x3 + BN, identity or
``downsample`` = conv1x1(stride) + BN, add, ReLU; stem conv7x7/s2 + BN + ReLU + MaxPool(3, 2, 1); layers [2, 2, 2, 2] of widths
64/128/256/512; adaptive average pool; fc) with torchvision's ``state_dict`` key names, so a real ``resnet18.pth`` loads.
PARITY UNPINNED: the reference holds no ResNet18 weights or outputs; the pre-processing in front of it is the pinned one of
e2e.py:366-370 (oracle/pil_resize_ref.py).
"""
from typing import List, Tuple

import numpy as np
import torch
import torch.nn as nn

from . import pil_resize_ref

MEAN, STD = 0.18, 0.34  # e2e.py:369


class BasicBlock(nn.Module):
    def __init__(self, inp: int, out: int, stride: int):
        super().__init__()
        self.conv1 = nn.Conv2d(inp, out, 3, stride, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(out)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(out, out, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(out)
        self.downsample = None
        if stride != 1 or inp != out:
            self.downsample = nn.Sequential(nn.Conv2d(inp, out, 1, stride, bias=False), nn.BatchNorm2d(out))

    def forward(self, x):
        idn = x if self.downsample is None else self.downsample(x)
        y = self.relu(self.bn1(self.conv1(x)))
        y = self.bn2(self.conv2(y))
        return self.relu(y + idn)


class ResNet18(nn.Module):
    def __init__(self, num_classes: int):
        super().__init__()
        self.conv1 = nn.Conv2d(3, 64, 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, 2, 1)
        self.layer1 = nn.Sequential(BasicBlock(64, 64, 1), BasicBlock(64, 64, 1))
        self.layer2 = nn.Sequential(BasicBlock(64, 128, 2), BasicBlock(128, 128, 1))
        self.layer3 = nn.Sequential(BasicBlock(128, 256, 2), BasicBlock(256, 256, 1))
        self.layer4 = nn.Sequential(BasicBlock(256, 512, 2), BasicBlock(512, 512, 1))
        self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
        self.fc = nn.Linear(512, num_classes)

    def forward(self, x):
        x = self.maxpool(self.relu(self.bn1(self.conv1(x))))
        x = self.layer4(self.layer3(self.layer2(self.layer1(x))))
        return self.fc(torch.flatten(self.avgpool(x), 1))


def seeded_state_dict(num_classes: int, seed: int = 4321, gain: float = 1.4) -> "dict[str, torch.Tensor]":
    """Synthetic weights with non-trivial BN statistics (identity BN would not exercise the folding); conv weights scaled so
    that activations keep an O(1) range through the 17 conv layers."""
    g = torch.Generator().manual_seed(seed)
    sd = ResNet18(num_classes).state_dict()
    for k, v in sd.items():
        if k.endswith("num_batches_tracked"):
            continue
        if v.ndim == 4:
            fan = v.shape[1] * v.shape[2] * v.shape[3]
            sd[k] = torch.randn(v.shape, generator=g) * (gain / fan) ** 0.5
        elif k.endswith("running_var"):
            sd[k] = torch.rand(v.shape, generator=g) * 0.5 + 0.75
        elif k.endswith("running_mean"):
            sd[k] = torch.randn(v.shape, generator=g) * 0.1
        elif k == "fc.weight":
            sd[k] = torch.randn(v.shape, generator=g) * (1.0 / 512) ** 0.5
        elif k.endswith("bias"):
            sd[k] = torch.randn(v.shape, generator=g) * 0.1
        else:  # BN weight
            sd[k] = torch.rand(v.shape, generator=g) * 0.5 + 0.75
    return sd


def build(num_classes: int, state_dict=None) -> ResNet18:
    m = ResNet18(num_classes)
    if state_dict is not None:
        m.load_state_dict(state_dict)
    return m.eval()


def preprocess(rois_bgr: List[np.ndarray], size: int = 64) -> torch.Tensor:
    """e2e.py:383-391: BGR->RGB, PIL Resize((64,64)) bilinear+antialias (uint8), ToTensor, Normalize."""
    return torch.from_numpy(np.stack([pil_resize_ref.classifier_input(r, size) for r in rois_bgr]))


def predict_batch(model: ResNet18, rois_bgr: List[np.ndarray], size: int = 64) -> Tuple[np.ndarray, np.ndarray]:
    if len(rois_bgr) == 0:
        return np.array([]), np.array([])
    with torch.no_grad():
        probs = torch.softmax(model(preprocess(rois_bgr, size)), 1).numpy()
    return np.argmax(probs, 1), probs
