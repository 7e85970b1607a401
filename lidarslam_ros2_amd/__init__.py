"""MI355X-native scan-matching registration core for lidarslam_ros2's hot path.

Only the `pcl::Registration` path (NDT / GICP setInputTarget, setInputSource, align,
getFitnessScore) is implemented here — hand-written HIP kernels for gfx950 behind the C ABI in
include/lidarslam_reg.h.  See DESIGN.md.
"""
from .registration import (DIRECT1, DIRECT7, DIRECT26, KDTREE, GeneralizedIterativeClosestPoint,
                           NormalDistributionsTransform, Registration, align_batch)

__all__ = ["Registration", "NormalDistributionsTransform", "GeneralizedIterativeClosestPoint", "align_batch",
           "DIRECT1", "DIRECT7", "DIRECT26", "KDTREE"]
