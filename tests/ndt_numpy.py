"""fp64 numpy restatement of the NDT score and gradient (Magnusson 2009 eqs 6.9, 6.12, 6.18-6.19) on a
voxel table dumped from the oracle — the "maths truth" the C++ oracle's analytic derivatives are
checked against by finite differences (SURVEY.md §8c KAT 3).  Test infrastructure only."""
import numpy as np


def rot_xyz(rx, ry, rz):
    cx, sx, cy, sy, cz, sz = np.cos(rx), np.sin(rx), np.cos(ry), np.sin(ry), np.cos(rz), np.sin(rz)
    Rx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]])
    Ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
    Rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
    return Rx @ Ry @ Rz


def jang_rows(p):
    cx, sx, cy, sy, cz, sz = np.cos(p[3]), np.sin(p[3]), np.cos(p[4]), np.sin(p[4]), np.cos(p[5]), np.sin(p[5])
    return np.array([
        [-sx * sz + cx * sy * cz, -sx * cz - cx * sy * sz, -cx * cy],
        [cx * sz + sx * sy * cz, cx * cz - sx * sy * sz, -sx * cy],
        [-sy * cz, sy * sz, cy],
        [sx * cy * cz, -sx * cy * sz, sx * sy],
        [-cx * cy * cz, cx * cy * sz, -cx * sy],
        [-cy * sz, -cy * cz, 0],
        [cx * cz - sx * sy * sz, -cx * sz - sx * sy * cz, 0],
        [sx * cz + cx * sy * sz, cx * sy * cz - sx * sz, 0]])


class NumpyNdt:
    def __init__(self, dump, min_b, max_b, leaf, d1, d2, search=7, centroids=None):
        self.leaf, self.d1, self.d2 = float(leaf), d1, d2
        self.min_b, self.max_b = np.asarray(min_b, np.int64), np.asarray(max_b, np.int64)
        div = self.max_b - self.min_b + 1
        self.mul = np.array([1, div[0], div[0] * div[1]], np.int64)
        ok = dump["n"] >= 6
        self.table = {int(k): (m, c) for k, m, c in zip(dump["idx"][ok], dump["mean"][ok], dump["icov"][ok])}
        self.kd = None
        if search == 0:      # KDTREE: a radius search over ALL leaf centroids (brute force: no cell structure is assumed here)
            assert centroids is not None
            self.kd = (np.asarray(centroids, np.float32)[ok], [self.table[int(k)] for k in dump["idx"][ok]],
                       np.float32(np.float64(np.float32(leaf)) * np.float64(np.float32(leaf))))
            self.off = np.zeros((0, 3), np.int64)
        elif search == 1:    # DIRECT1: the cell of the transformed point only
            self.off = np.zeros((1, 3), np.int64)
        elif search == 26:   # DIRECT26: the full 3x3x3 block, centre included (27 cells)
            self.off = np.array([[a, b, c] for a in (-1, 0, 1) for b in (-1, 0, 1) for c in (-1, 0, 1)], np.int64)
        else:                # DIRECT7: centre + the six face neighbours
            self.off = np.array([[0, 0, 0], [1, 0, 0], [-1, 0, 0], [0, 1, 0], [0, -1, 0], [0, 0, 1], [0, 0, -1]], np.int64)

    def score_grad(self, src, p):
        """fp64 score and gradient; cell lookup uses the same fp32 floor(x'/leaf) as the reference so the
        voxel assignment matches at cell faces."""
        p = np.asarray(p, np.float64)
        R = rot_xyz(*p[3:])
        xt = src.astype(np.float64) @ R.T + p[:3]
        ijk = np.floor(xt.astype(np.float32) / np.float32(self.leaf)).astype(np.int64)
        Jr = jang_rows(p)
        score, g = 0.0, np.zeros(6)
        for n in range(src.shape[0]):
            x = src[n].astype(np.float64)
            ja = Jr @ x
            J = np.array([[1, 0, 0, 0, ja[2], ja[5]], [0, 1, 0, ja[0], ja[3], ja[6]], [0, 0, 1, ja[1], ja[4], ja[7]]])
            leaves = []
            if self.kd is not None:
                # pcl::KdTreeFLANN::radiusSearch on the float point: L2_Simple<float>, strictly inside (float)(r * r)
                diff = np.float32(xt[n]).astype(np.float32) - self.kd[0]
                d = (diff[:, 0] * diff[:, 0] + diff[:, 1] * diff[:, 1]) + diff[:, 2] * diff[:, 2]
                leaves = [self.kd[1][k] for k in np.nonzero(d < self.kd[2])[0]]
            for o in self.off:
                c = ijk[n] + o
                if np.any(c < self.min_b) or np.any(c > self.max_b):
                    continue
                leaf = self.table.get(int(((c - self.min_b) * self.mul).sum()))
                if leaf is not None:
                    leaves.append(leaf)
            for leaf in leaves:
                q = xt[n] - leaf[0]
                Cq = leaf[1] @ q
                e = np.exp(-self.d2 * (q @ Cq) / 2)
                w = self.d2 * e
                if not (0 <= w <= 1):
                    continue
                score += -self.d1 * e
                g += self.d1 * w * (Cq @ J)
        return score, g
