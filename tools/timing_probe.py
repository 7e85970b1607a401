"""In-kernel timestamp probe (needs the -DLSR_TIMING build: `make -C lidarslam_ros2_amd/csrc timing`, then
LSR_LIB_NAME=liblidarslam_reg_timing.so).  For every launch variant (workgroup size x table mode) prints the medians
over workgroups of the stamps the derivative kernel leaves: 0 entry, 1 state+rows in LDS, 6 row totals, 5 controller
done, 4 request built, 7 main start, 2 points done, 3 row written.  Head stamps come from the finalising launch of an
align, main-loop stamps from the last derivative pass."""
import sys, os, ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lidarslam_ros2_amd import NormalDistributionsTransform, synth, _capi
lib = _capi.load()
buf = C.c_void_p()
lib.lsr_debug_timing_buffer.argtypes = [C.POINTER(C.c_void_p)]
assert lib.lsr_debug_timing_buffer(C.byref(buf)) == 0
import multiprocessing as mp
with mp.get_context("fork").Pool(min(32, len(os.sched_getaffinity(0)))) as pool:
    case = synth.cfg_ndt_30k(pool=pool)
hip = C.CDLL("libamdhip64.so")
host = np.zeros((1024, 32), np.int64)
PHASES = ["INIT", "MT_FIRST", "MT_TRIAL", "MT_HESS", "DIAG"]
for quad, wg, tab in ((1, 128, 2), (1, 64, 2)):
    ndt = NormalDistributionsTransform(0); ndt.setResolution(5.0); ndt.setTransformationEpsilon(0.0); ndt.setMaximumIterations(30)
    ndt.setTuning(workgroup=wg, table_mode=tab, quad=quad)
    ndt.setInputTarget(case.target); ndt.setInputSource(case.source)
    nb = (case.source.shape[0] + wg - 1) // wg
    for rep in range(3):
        if rep == 2:
            hip.hipMemset(C.c_void_p(buf.value + 800 * 32 * 8), 0, 16 * 32 * 8)
        ndt.align(case.guess)
        hip.hipDeviceSynchronize(); hip.hipMemcpy(C.c_void_p(host.ctypes.data), buf, host.nbytes, 2)
        w = host[:nb, 0::2].astype(np.float64) * 10.0   # ns
        if rep < 2:
            continue
        h = lambda a, b: np.median(w[:, a] - w[:, b])
        print(f"quad {quad} wg {wg} tab {tab}: HEAD state+rows in LDS +{h(1, 0):.0f} ns | totals +{h(6, 1):.0f} | controller +{h(5, 6):.0f} | "
              f"request +{h(4, 5):.0f} || MAIN (from main start 7): points done +{h(2, 7):.0f} | row written +{h(3, 2):.0f} "
              f"| {ndt.last_result['iterations']} it {ndt.last_result['n_evaluations']} passes", flush=True)
        print(f"      head detail (medians over workgroups, last launch): wave 0 entry -> its loads landed +{h(8, 0):.0f} ns -> barrier passed +{h(1, 8):.0f} ns | "
              f"last wave entry {h(9, 0):+.0f} ns after wave 0, its loads landed +{h(10, 9):.0f} ns", flush=True)
        if quad:
            print(f"      main detail (last pass, with Hessian): request read + table landed +{h(11, 7):.0f} ns | phase A +{h(12, 11):.0f} | barrier + phase B +{h(13, 12):.0f} | "
                  f"barrier +{h(14, 13):.0f} | phase C +{h(2, 14):.0f} | shuffles + split + atomics +{h(3, 2):.0f}", flush=True)
        for hs, name in ((0, "gradient-only"), (1, "with Hessian")):
            r = host[810 + hs].astype(np.float64)
            if r[0]:
                print(f"      {name:13s} passes: {int(r[0]):4d}; workgroup 0 means: head {10 * r[1] / r[0]:.0f} ns, main {10 * r[2] / r[0]:.0f} ns, tail {10 * r[3] / r[0]:.0f} ns, "
                      f"entry->exit {10 * r[5] / r[0]:.0f} ns at {r[4] / (10 * r[5]) :.3f} GHz shader clock", flush=True)
        for ph, name in enumerate(PHASES):
            tot, cnt, req = host[800 + ph, 0], host[800 + ph, 1], host[800 + ph, 2]
            if cnt:
                print(f"      controller after {name:8s}: {cnt:4d} calls, mean {10.0 * tot / cnt:.0f} ns; request build mean {10.0 * req / cnt:.0f} ns", flush=True)
