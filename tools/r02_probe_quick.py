"""cfg 2 / cfg 1 timing of the default path through the raw C ABI on one stream (no Python-side stream ordering)."""
import sys, os, time, ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import multiprocessing as mp
from lidarslam_ros2_amd import synth
with mp.get_context("fork").Pool(min(32, len(os.sched_getaffinity(0)))) as pool:
    case = synth.cfg_ndt_30k(pool=pool)
import torch
from lidarslam_ros2_amd import NormalDistributionsTransform, _capi
lib = _capi.load()
tgt = torch.from_numpy(synth.as_pointxyzi(case.target)).cuda(); src = torch.from_numpy(synth.as_pointxyzi(case.source)).cuda()
st = torch.cuda.current_stream().cuda_stream
fptr = C.POINTER(C.c_float)
g16 = np.ascontiguousarray(case.guess.T, np.float32).reshape(16); fin = np.zeros(16, np.float32)
for eps, mi, name, reps in ((0.0, 30, "cfg2", 15), (0.01, 35, "cfg1", 60)):
    ndt = NormalDistributionsTransform(0, stream=st); ndt.setResolution(5.0); ndt.setTransformationEpsilon(eps); ndt.setMaximumIterations(mi)
    ndt.setInputTarget(tgt)
    def step():
        _capi.check(lib.lsr_set_input_source_device(ndt._h, C.c_void_p(src.data_ptr()), 32, src.shape[0]), "src")
        _capi.check(lib.lsr_align(ndt._h, g16.ctypes.data_as(fptr), fin.ctypes.data_as(fptr), C.byref(ndt._last), None, 0), "align")
    for _ in range(5): step()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); step(); ts.append(time.perf_counter() - t0)
    r = ndt.last_result
    print(f"{os.environ.get('TAG','')} {name}: median {1e3*np.median(ts):.4f} ms ({1e6*np.median(ts)/max(1,r['n_evaluations']):.2f} us/pass) it {r['iterations']} passes {r['n_evaluations']}", flush=True)
