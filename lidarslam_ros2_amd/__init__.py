"""MI355X-native scan-matching registration core for lidarslam_ros2's hot path.

Only the `pcl::Registration` path (NDT / GICP setInputTarget, setInputSource, align,
getFitnessScore) is implemented here — hand-written HIP kernels for gfx950 behind the C ABI in
include/lidarslam_reg.h.  See DESIGN.md.
"""
from .registration import (DIRECT1, DIRECT7, DIRECT26, KDTREE, GeneralizedIterativeClosestPoint,
                           NormalDistributionsTransform, Registration, align_batch)
from .loop_closure import LoopClosureParams, LoopEdge, SubMap, search_loop

__all__ = ["Registration", "NormalDistributionsTransform", "GeneralizedIterativeClosestPoint", "align_batch",
           "SubMap", "LoopClosureParams", "LoopEdge", "search_loop",
           "DIRECT1", "DIRECT7", "DIRECT26", "KDTREE"]
