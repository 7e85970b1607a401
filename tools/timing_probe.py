"""In-kernel timestamp probe (needs the -DLSR_TIMING build: LSR_LIB_NAME=liblidarslam_reg_timing.so).
Head stamps (0 entry, 1 state+rows in LDS, 6 row totals, 4 controller+request done) come from the
finalising launch of an align, main-loop stamps (7 start, 2 points done, 3 row written) from the last
derivative pass."""
import sys, os, ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lidarslam_ros2_amd import NormalDistributionsTransform, synth, _capi
lib = _capi.load()
buf = C.c_void_p()
lib.lsr_debug_timing_buffer.argtypes = [C.POINTER(C.c_void_p)]
assert lib.lsr_debug_timing_buffer(C.byref(buf)) == 0
case = synth.cfg_ndt_30k()
ndt = NormalDistributionsTransform(0); ndt.setResolution(5.0); ndt.setTransformationEpsilon(0.0); ndt.setMaximumIterations(30)
ndt.setInputTarget(case.target); ndt.setInputSource(case.source)
hip = C.CDLL("libamdhip64.so")
host = np.zeros((1024, 32), np.int64)
for rep in range(3):
    ndt.align(case.guess)
    hip.hipDeviceSynchronize(); hip.hipMemcpy(C.c_void_p(host.ctypes.data), buf, host.nbytes, 2)
    nb = 118
    w = host[:nb, 0::2].astype(np.float64) * 10.0
    h0 = w[:, 0].min()
    print("rep", rep, "HEAD: state+rows in LDS %.0f ns after entry | finalising launch: row totals +%.0f ns, controller+request +%.0f ns" % (
        np.median(w[:, 1]) - h0, np.median(w[:, 6] - w[:, 1]), np.median(w[:, 4] - w[:, 6])))
    if w[:, 8].max() > 0:
        print("        MAIN detail (last loop iteration of each block): loads issued at +%.0f, pair loop done +%.0f ns after main start" % (
            np.median(w[:, 8] - w[:, 7]), np.median(w[:, 9] - w[:, 7])))
    m0 = np.median(w[:, 7])
    print("        MAIN  (last pass): points done +%.0f  row written +%.0f ns (medians from main start); max row written +%.0f" % (
        np.median(w[:, 2]) - m0, np.median(w[:, 3]) - m0, w[:, 3].max() - m0))
