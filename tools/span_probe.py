"""Launch-to-launch timeline of the NDT chain (needs the -DLSR_TIMING build): for every launch the earliest
workgroup entry and the latest workgroup exit (s_memrealtime, 10 ns ticks)."""
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
init = np.zeros((1024, 32), np.int64); init[512:, 0] = np.iinfo(np.int64).max
for rep in range(3):
    hip.hipMemcpy(buf, C.c_void_p(init.ctypes.data), init.nbytes, 1)
    ndt.align(case.guess)
    hip.hipDeviceSynchronize(); hip.hipMemcpy(C.c_void_p(host.ctypes.data), buf, host.nbytes, 2)
    ok = (host[512:768, 1] > 0) & (host[512:768, 0] < np.iinfo(np.int64).max)
    sp = host[512:768, :2][ok].astype(np.float64) * 10.0   # ns; slot = seq & 255
    dur = sp[:, 1] - sp[:, 0]
    st = sp[np.argsort(sp[:, 0])]
    gaps = st[1:, 0] - st[:-1, 1]
    per = st[1:, 0] - st[:-1, 0]
    good = per < 40000
    print("rep %d: launches seen %d | first-entry -> last-exit %.2f us (median) | last-exit -> next first-entry %.2f us | entry->entry %.2f us" % (
        rep, ok.sum(), np.median(dur) / 1e3, np.median(gaps[good]) / 1e3, np.median(per[good]) / 1e3))
    w = host[:118, 0::2].astype(np.float64) * 10.0
    e, x = w[:, 0], w[:, 3]
    print("        last launch, per workgroup: entry spread %.2f us (p10..p90 %.2f) | exit spread %.2f us (p10..p90 %.2f) | per-WG entry->exit median %.2f max %.2f us" % (
        (e.max() - e.min()) / 1e3, (np.percentile(e, 90) - np.percentile(e, 10)) / 1e3, (x.max() - x.min()) / 1e3,
        (np.percentile(x, 90) - np.percentile(x, 10)) / 1e3, np.median(x - e) / 1e3, (x - e).max() / 1e3))
    o = np.argsort(e)
    print("        entry order (us after first):", np.round((e[o] - e.min())[::8] / 1e3, 2))
