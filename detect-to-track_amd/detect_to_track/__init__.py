"""detect_to_track -- MI355X-native custom-op hot path of detect-to-track.

Only the op layer lives here (``detect_to_track.models``): PointwiseCorrelation, ROIPool and
PSROIPool with the reference's module / autograd.Function surface, backed by hand-written
gfx950 HIP kernels in ``libd2t_ops.so``.  The rest of the reference's Python tree (trainer,
data, losses, model graph) is untouched by this repository and imports these names as before.
"""
