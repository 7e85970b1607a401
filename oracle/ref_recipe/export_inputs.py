"""TEST INFRASTRUCTURE (oracle/ref_recipe/README.md): writes the INPUTS of the committed golden fixtures where the reference dumper
reads them — binary PCD v0.7 files (x y z intensity, float32) and cases.json — under oracle/_ref/inputs/.  The clouds are the seeded
synthetic ones the fixtures were generated from (tests/golden/make_golden.py, make_golden_gicp.py, make_cfg4_fixture.py).

    python oracle/ref_recipe/export_inputs.py [--cfg4 N]     # N = number of cfg-4 candidates (default 64, 0 = skip: ~85 MB each way)
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from lidarslam_ros2_amd import synth  # noqa: E402


def write_pcd(path, xyz):
    xyz = np.asarray(xyz, np.float32)
    rec = np.zeros((xyz.shape[0], 4), np.float32)
    rec[:, :3] = xyz[:, :3]
    head = ("# .PCD v0.7 - Point Cloud Data file format\nVERSION 0.7\nFIELDS x y z intensity\nSIZE 4 4 4 4\nTYPE F F F F\nCOUNT 1 1 1 1\n"
            "WIDTH %d\nHEIGHT 1\nVIEWPOINT 0 0 0 1 0 0 0\nPOINTS %d\nDATA binary\n" % (rec.shape[0], rec.shape[0]))
    with open(path, "wb") as f:
        f.write(head.encode("ascii"))
        f.write(rec.tobytes())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cfg4", type=int, default=64)
    ap.add_argument("--stream-scans", dest="stream_scans", type=int, default=12, help="scans of the frontend drive (tests/test_frontend_stream_gpu.py: 12)")
    ap.add_argument("--out", default=os.path.join(ROOT, "oracle", "_ref", "inputs"))
    a = ap.parse_args()
    os.makedirs(a.out, exist_ok=True)
    cases = {}
    g = np.load(os.path.join(ROOT, "tests", "golden", "ndt_small_golden.npz"))
    c = synth.small_case(n_source=int(g["n_source"]), n_keyframes=int(g["n_keyframes"]))
    write_pcd(os.path.join(a.out, "ndt_small_target.pcd"), c.target)
    write_pcd(os.path.join(a.out, "ndt_small_source.pcd"), c.source)
    cases["ndt_small"] = {"target": "ndt_small_target.pcd", "source": "ndt_small_source.pcd", "resolution": float(g["res"]),
                          "guess_colmajor": np.asarray(c.guess, np.float32).T.reshape(-1).tolist(), "p": np.asarray(g["p"], np.float64).tolist(),
                          "schedules": [{"name": "eps001", "eps": 0.01, "max_iterations": 35}, {"name": "tight", "eps": 1e-6, "max_iterations": 30}]}
    gg = np.load(os.path.join(ROOT, "tests", "golden", "gicp_small_golden.npz"))
    cg = synth.small_case(n_source=int(gg["n_source"]), n_keyframes=int(gg["n_keyframes"]))
    write_pcd(os.path.join(a.out, "gicp_small_target_raw.pcd"), cg.target)
    write_pcd(os.path.join(a.out, "gicp_small_source.pcd"), cg.source)
    cases["gicp_small"] = {"target_raw": "gicp_small_target_raw.pcd", "source": "gicp_small_source.pcd", "leaf": float(gg["leaf"]),
                           "guess_colmajor": np.asarray(cg.guess, np.float32).T.reshape(-1).tolist(), "corr_dist": 5.0, "eps": 1e-8,
                           "max_iterations": 100, "head": 200}
    cfg4 = []
    for k in range(a.cfg4):
        cc = synth.cfg_loop_candidate(k)
        write_pcd(os.path.join(a.out, "cfg4_%02d_target.pcd" % k), cc.target)
        write_pcd(os.path.join(a.out, "cfg4_%02d_source.pcd" % k), cc.source)
        cfg4.append({"target": "cfg4_%02d_target.pcd" % k, "source": "cfg4_%02d_source.pcd" % k,
                     "guess_colmajor": np.asarray(cc.guess, np.float32).T.reshape(-1).tolist(),
                     "truth_rowmajor": np.asarray(cc.truth, np.float64).reshape(-1).tolist()})
    cases["cfg4"] = {"resolution": 5.0, "eps": 0.01, "max_iterations": 100, "candidates": cfg4}
    # ---- round 6: the two SEQUENCES the GPU tests hold against the oracle — the frontend loop over a drive (receiveCloud + updateMap,
    # tests/test_frontend_stream_gpu.py) and the loop gate over a route (searchLoop, tests/test_loop_closure_gpu.py)
    import multiprocessing as mp
    with mp.get_context("spawn").Pool(min(16, len(os.sched_getaffinity(0)))) as pool:
        drive = synth.cfg_frontend_drive(a.stream_scans, pool=pool)
    fs = {"frames": [], "scans": [], "guess0_colmajor": np.asarray(drive["guess0"], np.float64).T.reshape(-1).tolist(),
          "ndt_resolution": 5.0, "eps": 0.01, "max_iterations": 35, "vg_size_for_input": 0.2, "vg_size_for_map": 0.1, "trans_for_mapupdate": 1.5,
          "scan_min_range": 0.1, "scan_max_range": 100.0, "num_targeted_cloud": 10}
    for k, (fr, P) in enumerate(zip(drive["frames"], drive["frame_poses"])):
        write_pcd(os.path.join(a.out, "fs_frame_%02d.pcd" % k), fr)
        fs["frames"].append({"frame_cloud": "fs_frame_%02d.pcd" % k, "pose_colmajor": np.asarray(P, np.float64).T.reshape(-1).tolist()})
    for k, sc in enumerate(drive["scans"]):
        write_pcd(os.path.join(a.out, "fs_scan_%02d.pcd" % k), sc)
        fs["scans"].append({"scan_cloud": "fs_scan_%02d.pcd" % k})
    cases["frontend_stream"] = fs
    route = synth.make_loop_route()
    lg = {"submaps": [], "threshold_loop_closure_score": 1.0, "distance_loop_closure": 20.0, "range_of_searching_loop_closure": 10.0,
          "search_submap_num": 2, "voxel_leaf_size": 0.2, "ndt_resolution": 5.0, "eps": 0.01, "max_iterations": 100}
    for k, sm in enumerate(route):
        write_pcd(os.path.join(a.out, "lg_submap_%02d.pcd" % k), sm["cloud"])
        lg["submaps"].append({"submap_cloud": "lg_submap_%02d.pcd" % k, "position": [float(v) for v in sm["position"]],
                              "orientation_xyzw": [float(v) for v in sm["orientation"]], "distance": float(sm["distance"])})
    cases["loop_gate"] = lg
    with open(os.path.join(a.out, "cases.json"), "w") as f:
        json.dump(cases, f)
    print("wrote", a.out, "(%d cfg-4 candidates, %d drive scans, %d route submaps)" % (len(cfg4), len(fs["scans"]), len(lg["submaps"])))


if __name__ == "__main__":
    main()
