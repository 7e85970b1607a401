"""Not a test: the experiment behind DESIGN.md §2's note on cfg-4 candidates 21 and 34 (run by hand, ~5 min on 8 cores).

bench.py holds all 64 cfg-4 candidates to the CPU oracle (tests/golden/cfg4_candidates_oracle.npz): same number of Newton
iterations on all 64, pose within 1e-3 m / 1e-4 rad on 62.  Candidates 21 and 34 take ~30 Newton iterations whose steps
are clamped to 0.1 m (the registration starts ~1 m off along a corridor) and whose direction H^-1 g is ill conditioned;
a perturbation of the derivative sums at fp32-ulp level — which is what separates the GPU's factorised per-pair maths
from the CPU's — moves the stopping point by millimetres, inside the eps = 0.01 m termination tolerance.  This script
shows the CPU path is just as sensitive: the SAME oracle sources compiled with FMA contraction (-mfma -ffp-contract=fast)
instead of -ffp-contract=off end 1.3e-3 m / 1.8e-4 rad apart on candidate 34 (6e-5 m on 21, 2e-7 m on well-conditioned
ones), with identical iteration and evaluation counts.

    python tests/sensitivity_cfg4.py
"""
import glob
import importlib.util
import multiprocessing as mp
import os
import shutil
import subprocess
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lidarslam_ros2_amd import synth  # noqa: E402
from lidarslam_ros2_amd.posemath import pose_delta  # noqa: E402
from oracle import oracle as O  # noqa: E402

if __name__ == "__main__":
    src = os.path.dirname(os.path.abspath(O.__file__))
    tmp = tempfile.mkdtemp(prefix="oracle_fma_")
    for f in glob.glob(os.path.join(src, "*.cpp")) + glob.glob(os.path.join(src, "*.h")) + [os.path.join(src, "oracle.py")]:
        shutil.copy(f, tmp)
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-fopenmp", "-mfma", "-mavx2", "-ffp-contract=fast", "-shared", "-o",
                           os.path.join(tmp, "liboracle.so")] + sorted(glob.glob(os.path.join(tmp, "*.cpp"))))
    spec = importlib.util.spec_from_file_location("oracle_fma", os.path.join(tmp, "oracle.py"))
    OF = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(OF)
    threads = len(os.sched_getaffinity(0))
    for c in (21, 34, 5, 12, 43):
        with mp.get_context("fork").Pool(threads) as pool:
            k = synth.cfg_loop_candidate(c, pool=pool)
        kw = dict(resolution=5.0, trans_eps=0.01, max_iterations=100, num_threads=threads)
        a = O.ndt_align(O.VoxelGridCovariance(k.target, 5.0), k.source, k.guess, **kw)
        b = OF.ndt_align(OF.VoxelGridCovariance(k.target, 5.0), k.source, k.guess, **kw)
        dt, ang = pose_delta(a["final"], b["final"])
        print(f"candidate {c}: CPU(-ffp-contract=off) vs CPU(-mfma, contraction) {dt:.2e} m {ang:.2e} rad | Newton iterations "
              f"{a['iterations']} / {b['iterations']}", flush=True)
