"""In-kernel timestamp probe (needs the -DLSR_TIMING build: LSR_LIB_NAME=liblidarslam_reg_timing.so)."""
import sys, os, ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lidarslam_ros2_amd import NormalDistributionsTransform, synth, _capi
from oracle import oracle as O
lib = _capi.load()
buf = C.c_void_p()
lib.lsr_debug_timing_buffer.argtypes = [C.POINTER(C.c_void_p)]
assert lib.lsr_debug_timing_buffer(C.byref(buf)) == 0
case = synth.cfg_ndt_30k()
ndt = NormalDistributionsTransform(0); ndt.setResolution(5.0); ndt.setTransformationEpsilon(0.0); ndt.setMaximumIterations(30)
ndt.setInputTarget(case.target); ndt.setInputSource(case.source)
p = O.matrix_to_pose(case.guess)
hip = C.CDLL("libamdhip64.so")
host = np.zeros((1024, 32), np.int64)
def fetch():
    hip.hipDeviceSynchronize()
    hip.hipMemcpy(C.c_void_p(host.ctypes.data), buf, host.nbytes, 2)
    return host.copy()
for hess in (True, False):
    for rep in range(3):
        for _ in range(5): ndt.derivatives(p, compute_hessian=hess)
        t = fetch()
        nb = 118
        w = t[:nb, 0::2].astype(np.float64) * 10.0  # ns (100 MHz)
        c = t[:nb, 1::2].astype(np.float64)
        t0 = w[:, 0].min()
        last = int(np.argmax(w[:, 7]))
        print("hess", hess, "rep", rep)
        names = ["entry", "state-read", "main-done", "lds-reduce", "stores-drained", "ticket", "partials-summed", "controller-done"]
        for k in range(6):
            print("  %-16s first %.0f  median %.0f  max %.0f ns" % (names[k], w[:, k].min() - t0, np.median(w[:, k]) - t0, w[:, k].max() - t0))
        for k in (6, 7):
            print("  %-16s last-block %.0f ns" % (names[k], w[last, k] - t0))
        dcy = c[last, 5] - c[last, 0]; dns = w[last, 5] - w[last, 0]
        print("  shader clock estimate: %.0f MHz" % (dcy / dns * 1e3))
