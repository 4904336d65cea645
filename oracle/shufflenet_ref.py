"""Oracle: torch-CPU fp32 restatement of torchvision 0.16 ``shufflenet_v2_x1_0``
with the ``fc`` head the reference swaps in (``src/tt100k/pipeline/e2e.py:331-333``)
and of ``PyTorchClassifier.predict_batch`` (e2e.py:378-396).

TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED: torchvision is not installed in the
build container and the reference ships no classifier weights or outputs; the
architecture follows the published ShuffleNetV2 x1.0 definition (SURVEY
Appendix B).  Module/parameter names equal torchvision's state_dict keys, so a
real ``shufflenetv2.pth`` loads with ``load_state_dict``.
"""
from __future__ import annotations

from typing import List, Tuple

import numpy as np
import torch
import torch.nn as nn

from .pil_resize_ref import classifier_input

STAGE_REPEATS = (4, 8, 4)
STAGE_OUT = (24, 116, 232, 464, 1024)


def channel_shuffle(x: torch.Tensor, groups: int) -> torch.Tensor:
    b, c, h, w = x.shape
    return x.view(b, groups, c // groups, h, w).transpose(1, 2).reshape(b, c, h, w)


class InvertedResidual(nn.Module):
    def __init__(self, inp: int, oup: int, stride: int):
        super().__init__()
        self.stride = stride
        bf = oup // 2
        if stride > 1:
            self.branch1 = nn.Sequential(
                nn.Conv2d(inp, inp, 3, stride, 1, groups=inp, bias=False), nn.BatchNorm2d(inp),
                nn.Conv2d(inp, bf, 1, 1, 0, bias=False), nn.BatchNorm2d(bf), nn.ReLU(inplace=True))
        else:
            self.branch1 = nn.Sequential()
        self.branch2 = nn.Sequential(
            nn.Conv2d(inp if stride > 1 else bf, bf, 1, 1, 0, bias=False), nn.BatchNorm2d(bf), nn.ReLU(inplace=True),
            nn.Conv2d(bf, bf, 3, stride, 1, groups=bf, bias=False), nn.BatchNorm2d(bf),
            nn.Conv2d(bf, bf, 1, 1, 0, bias=False), nn.BatchNorm2d(bf), nn.ReLU(inplace=True))

    def forward(self, x):
        if self.stride == 1:
            x1, x2 = x.chunk(2, dim=1)
            out = torch.cat((x1, self.branch2(x2)), dim=1)
        else:
            out = torch.cat((self.branch1(x), self.branch2(x)), dim=1)
        return channel_shuffle(out, 2)


class ShuffleNetV2(nn.Module):
    def __init__(self, num_classes: int = 58):
        super().__init__()
        c = STAGE_OUT
        self.conv1 = nn.Sequential(nn.Conv2d(3, c[0], 3, 2, 1, bias=False), nn.BatchNorm2d(c[0]), nn.ReLU(inplace=True))
        self.maxpool = nn.MaxPool2d(3, 2, 1)
        inp = c[0]
        for name, rep, oup in zip(("stage2", "stage3", "stage4"), STAGE_REPEATS, c[1:4]):
            seq = [InvertedResidual(inp, oup, 2)] + [InvertedResidual(oup, oup, 1) for _ in range(rep - 1)]
            setattr(self, name, nn.Sequential(*seq))
            inp = oup
        self.conv5 = nn.Sequential(nn.Conv2d(inp, c[4], 1, 1, 0, bias=False), nn.BatchNorm2d(c[4]), nn.ReLU(inplace=True))
        self.fc = nn.Linear(c[4], num_classes)

    def forward(self, x):
        x = self.maxpool(self.conv1(x))
        x = self.stage4(self.stage3(self.stage2(x)))
        x = self.conv5(x).mean([2, 3])
        return self.fc(x)


def seeded_state_dict(num_classes: int, seed: int = 1234, gain: float = 1.7) -> "dict[str, torch.Tensor]":
    """Synthetic weights with non-trivial BN statistics (random-init torchvision
    weights have identity BN, which would not exercise BN folding)."""
    g = torch.Generator().manual_seed(seed)
    model = ShuffleNetV2(num_classes)
    sd = model.state_dict()
    for k, v in sd.items():
        if k.endswith("num_batches_tracked"):
            continue
        if k.endswith("running_var"):
            sd[k] = torch.rand(v.shape, generator=g) * 0.5 + 0.75
        elif k.endswith("running_mean"):
            sd[k] = torch.randn(v.shape, generator=g) * 0.1
        elif v.dim() == 1 and k.endswith("weight"):  # BN gamma
            sd[k] = torch.rand(v.shape, generator=g) * 0.5 + 0.75
        elif v.dim() == 1:  # BN beta / fc bias
            sd[k] = torch.randn(v.shape, generator=g) * 0.1
        else:
            fan_in = float(np.prod(v.shape[1:]))
            sd[k] = torch.randn(v.shape, generator=g) * (gain / fan_in) ** 0.5
    return sd


def build(num_classes: int, state_dict=None) -> ShuffleNetV2:
    m = ShuffleNetV2(num_classes)
    if state_dict is not None:
        m.load_state_dict(state_dict)
    return m.eval()


@torch.no_grad()
def predict_batch(model: ShuffleNetV2, rois_bgr: List[np.ndarray], size: int = 64) -> Tuple[np.ndarray, np.ndarray]:
    """e2e.py:378-396."""
    if len(rois_bgr) == 0:
        return np.array([]), np.array([])
    batch = torch.from_numpy(np.stack([classifier_input(r, size) for r in rois_bgr]))
    probs = torch.softmax(model(batch), dim=1).numpy()
    return np.argmax(probs, axis=1), probs
