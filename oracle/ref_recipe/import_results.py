"""TEST INFRASTRUCTURE (oracle/ref_recipe/README.md): oracle/_ref/out/results.json (written by dump_fixtures, i.e. by the REFERENCE's own
pclomp code) -> tests/golden/ref_*.npz with the array names of the oracle fixtures, so that tests/golden_fixtures.py:load_golden() picks
them up.  Matrices arrive column-major (Eigen's layout) and are stored row-major 4x4 like the oracle fixtures."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def mat(v):
    return np.asarray(v, np.float64).reshape(4, 4).T.copy()


def main(path=None, out_dir=None):
    path = path or os.path.join(ROOT, "oracle", "_ref", "out", "results.json")
    out_dir = out_dir or os.path.join(ROOT, "tests", "golden")
    R = json.load(open(path))
    wrote = []
    if "ndt_small" in R:
        n = R["ndt_small"]
        kd = {}
        if "score_kdtree" in n:   # round 6: the KDTREE neighbourhood (dumps made by an older recipe do not hold it: the oracle's arrays stay)
            kd = dict(score_kdtree=float(n["score_kdtree"]), grad_kdtree=np.asarray(n["grad_kdtree"], np.float64),
                      hess_kdtree=np.asarray(n["hess_kdtree"], np.float64).reshape(6, 6), final_kdtree=mat(n["final_kdtree"]).astype(np.float32),
                      iters_kdtree=int(n["iters_kdtree"]),
                      leaf_centroid=np.where(np.asarray(n["leaf_centroid"], np.float64) > 1e299, np.nan,
                                             np.asarray(n["leaf_centroid"], np.float64)).astype(np.float32).reshape(-1, 3))
        np.savez_compressed(os.path.join(out_dir, "ref_ndt_small_golden.npz"), **kd,
                            score=float(n["score"]), grad=np.asarray(n["grad"], np.float64), hess=np.asarray(n["hess"], np.float64).reshape(6, 6),
                            final_eps001=mat(n["final_eps001"]).astype(np.float32), iters_eps001=int(n["iters_eps001"]),
                            final_tight=mat(n["final_tight"]).astype(np.float32), iters_tight=int(n["iters_tight"]),
                            leaf_idx=np.asarray(n["leaf_idx"], np.int32), leaf_n=np.asarray(n["leaf_n"], np.int32),
                            min_b=np.asarray(n["min_b"], np.int32), max_b=np.asarray(n["max_b"], np.int32))
        wrote.append("ref_ndt_small_golden.npz")
    if "gicp_small" in R:
        g = R["gicp_small"]
        np.savez_compressed(os.path.join(out_dir, "ref_gicp_small_golden.npz"),
                            n_target=int(g["n_target"]), target_head=np.asarray(g["target_head"], np.float32).reshape(-1, 3),
                            cov_src_head=np.asarray(g["cov_src_head"], np.float64).reshape(-1, 3, 3),
                            cov_tgt_head=np.asarray(g["cov_tgt_head"], np.float64).reshape(-1, 3, 3),
                            nn_idx=np.asarray(g["nn_idx"], np.int32), nn_d2=np.asarray(g["nn_d2"], np.float32),
                            final_bfgs=mat(g["final_bfgs"]).astype(np.float32), iters_bfgs=int(g["iters_bfgs"]), fitness=float(g["fitness"]))
        wrote.append("ref_gicp_small_golden.npz")
    if R.get("cfg4"):
        c = R["cfg4"]
        np.savez_compressed(os.path.join(out_dir, "ref_cfg4_candidates_oracle.npz"),
                            final=np.stack([mat(x["final"]) for x in c]), iterations=np.asarray([x["iterations"] for x in c], np.int32),
                            converged=np.asarray([bool(x["converged"]) for x in c]), fitness=np.asarray([x["fitness"] for x in c], np.float64),
                            truth=np.stack([np.asarray(x["truth_rowmajor"], np.float64).reshape(4, 4) for x in c]))
        wrote.append("ref_cfg4_candidates_oracle.npz")
    if R.get("frontend_stream"):
        fsr = R["frontend_stream"]
        np.savez_compressed(os.path.join(out_dir, "ref_frontend_stream.npz"),
                            poses=np.stack([mat(x["final"]) for x in fsr["scans"]]), iterations=np.asarray([x["iterations"] for x in fsr["scans"]], np.int32),
                            points_kept=np.asarray([x["points_kept"] for x in fsr["scans"]], np.int32),
                            update_at=np.asarray(fsr["update_at"], np.int32))
        wrote.append("ref_frontend_stream.npz")
    if R.get("loop_gate"):
        lgr = R["loop_gate"]
        np.savez_compressed(os.path.join(out_dir, "ref_loop_gate.npz"), pair_id=np.asarray(lgr["pair_id"], np.int32), final=mat(lgr["final"]),
                            fitness=float(lgr["fitness"]), accepted=bool(lgr["accepted"]), n_target_points=int(lgr["n_target_points"]),
                            iterations=int(lgr["iterations"]))
        wrote.append("ref_loop_gate.npz")
    print("wrote", wrote, "into", out_dir)
    return wrote


if __name__ == "__main__":
    main(*sys.argv[1:3])
