import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lidarslam_ros2_amd import GeneralizedIterativeClosestPoint, synth
from lidarslam_ros2_amd.posemath import pose_delta
gc = synth.cfg_gicp_30k()
g = GeneralizedIterativeClosestPoint(0); g.setMaxCorrespondenceDistance(5.0); g.setTransformationEpsilon(1e-8)
t0 = time.perf_counter(); g.setInputTarget(gc.target); t1 = time.perf_counter()
g.setInputSource(gc.source); g.align(gc.guess); t2 = time.perf_counter()
print("target %d pts: setInputTarget %.2f ms; first source+align %.2f ms" % (len(gc.target), (t1-t0)*1e3, (t2-t1)*1e3))
ts, ta = [], []
for _ in range(8):
    t0 = time.perf_counter(); g.setInputSource(gc.source); t1 = time.perf_counter(); g.align(gc.guess); t2 = time.perf_counter()
    ts.append(t1-t0); ta.append(t2-t1)
print("setInputSource %.3f ms  align %.3f ms" % (np.median(ts)*1e3, np.median(ta)*1e3), g.last_result, pose_delta(g.getFinalTransformation(), gc.truth))
