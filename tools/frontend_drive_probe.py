"""Per-scan breakdown of a frontend drive (different RAW scans, map updates as the reference does them): source time, align time, the
form the voxel filter took (LSR_VOXEL_FILTER_FORM), derivative passes — to see WHICH scans are the slow ones of bench.py frontend_stream."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import multiprocessing as mp
from lidarslam_ros2_amd import synth
from _cache import cached
def _d():
    with mp.get_context("fork").Pool(min(64, len(os.sched_getaffinity(0)))) as p: return synth.cfg_frontend_drive(24, pool=p)
drive = cached("probe_frontend_drive24", _d)
import torch
from lidarslam_ros2_amd import NormalDistributionsTransform, DIRECT7
from lidarslam_ros2_amd.frontend import FrontendParams, FrontendReplay, FrontendResult, as_pc2_payload
def make():
    r = NormalDistributionsTransform(0, stream=torch.cuda.current_stream().cuda_stream); r.setResolution(5.0); r.setTransformationEpsilon(0.01); r.setMaximumIterations(35); r.setNeighborhoodSearchMethod(DIRECT7)
    return r
reg, mapper = make(), make()
to_dev = lambda a: torch.from_numpy(np.ascontiguousarray(a, np.float32)).cuda()
devs = [torch.from_numpy(as_pc2_payload(s)).cuda() for s in drive["scans"]]
torch.cuda.synchronize()
for rep in range(2):
    fr = FrontendReplay(reg, FrontendParams(), to_device=to_dev, mapper=mapper)
    fr.initialise(drive["frames"], drive["frame_poses"], drive["guess0"])
    rows = []
    for j, d in enumerate(devs):
        n = int(d.shape[0])
        t0 = time.perf_counter()
        kept = reg.setInputSourcePointCloud2(d, n, 32, (0, 4, 8, 16), 0.1, 100.0, 0.2)
        t1 = time.perf_counter()
        reg.align(fr.pose.astype(np.float32))
        T = np.asarray(reg.getFinalTransformation(), np.float64)
        t2 = time.perf_counter()
        rows.append((j, 1e6 * (t1 - t0), 1e6 * (t2 - t1), reg.voxelFilterForm(), reg.last_result["n_evaluations"], reg.last_result["iterations"], kept))
        fr.pose = T
        if float(np.linalg.norm(T[:3, 3] - fr.key_position)) >= 1.5:
            fr.key_position = T[:3, 3].copy()
            fr._update_job(d, n, None, T)
            rows[-1] += ("update",)
for r in rows:
    print("scan %2d source %6.1f us align %6.1f us form %d passes %2d iterations %d kept %d %s" % (r[:7] + (r[7] if len(r) > 7 else "",)))
