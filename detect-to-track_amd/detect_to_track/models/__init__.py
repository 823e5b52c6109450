"""Op layer of detect_to_track, MI355X-native.

Re-exports the three custom ops under the names the reference's model graph imports
(reference models/__init__.py:3-5; callers correlation_tracker.py:29-30 and rfcn.py:23).
``CorrelationTracker`` (reference models/correlation_tracker.py) is the first caller rebuilt around them:
same interface, the view/permute/cat glue between the correlations and ROIPool fused into the kernels.
``resnet_backbone`` / ``RPN`` / ``RFCN`` / ``DetectTrackModule`` are the rest of the model graph,
plain torch modules with random weights: the surroundings bench_model.py times the ops in.
Importing this package loads libd2t_ops.so and raises ImportError if it has not been built.
"""
from . import _native  # noqa: F401  (loads the HIP library; fails loudly when absent)
from .ps_roipool.ps_roipool import PSROIPool, PSROIPoolFunction
from .pointwise_correlation.pointwise_correlation import PointwiseCorrelation, PointwiseCorrelationFunction
from .roipool.roipool import ROIPool, ROIPoolFunction
from .correlation_tracker import CorrelationTracker, TrackFeaturesFunction
from .resnet import resnet_backbone
from .rpn import RPN
from .rfcn import RFCN
from .detect_track import DetectTrackModule

__all__ = [
    "PSROIPool", "PSROIPoolFunction",
    "PointwiseCorrelation", "PointwiseCorrelationFunction",
    "ROIPool", "ROIPoolFunction",
    "CorrelationTracker", "TrackFeaturesFunction",
    "resnet_backbone", "RPN", "RFCN", "DetectTrackModule",
]
