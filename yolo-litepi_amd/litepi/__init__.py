"""litepi -- MI355X-native backend for YOLO-LitePi's two-stage inference path.

Importable as ``litepi`` once ``yolo-litepi_amd/`` is on ``sys.path`` (the repo-root
``conftest.py``, ``bench.py`` and ``__graft_entry__.py`` add it).  Importing the package does not
load the HIP library; constructing an ``Engine`` (or any of the drop-in classes) does, and fails
loudly when it is missing -- there is no CPU fallback.
"""
from .backend import Engine, HybridPipeline, NCNNDetector, PipelineMetrics, PyTorchClassifier  # noqa: F401

__all__ = ["Engine", "HybridPipeline", "NCNNDetector", "PipelineMetrics", "PyTorchClassifier"]
