import os
import sys

import pytest

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (_ROOT, os.path.join(_ROOT, "yolo-litepi_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(_ROOT, "tests", "golden")
REF_ROOT = "/root/reference"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "reference: needs /root/reference (build container only)")


def has_reference():
    return os.path.isdir(os.path.join(REF_ROOT, "src"))


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def synth_models(tmp_path_factory):
    """Seeded synthetic v1 / v2 detectors in NCNN format (same architecture as the reference's)."""
    from litepi import ncnn_export

    d = tmp_path_factory.mktemp("models")
    out = {}
    for preset in ("v1", "v2"):
        p, b = str(d / f"{preset}.param"), str(d / f"{preset}.bin")
        ncnn_export.export_detector(p, b, preset, seed=1234, cls_bias=-2.0)
        out[preset] = (p, b)
    return out
