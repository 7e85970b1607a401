import ctypes as C, numpy as np, sys, time
sys.path.insert(0, '/root/repo')
from scipy.spatial import cKDTree
lib = C.CDLL('/tmp/nnh/libnnh.so')
def build(pts, cell):
    inv = np.float32(1.0) / np.float32(cell)
    f = np.floor(pts * inv).astype(np.int64)
    f0 = f.min(0); f1 = f.max(0)
    org = np.where(f0 >= 0, f0 & ~7, -(((-f0) + 7) & ~7))
    cdim = ((f1 - org) >> 3) + 1
    rel = f - org
    cc = rel >> 3
    clin = cc[:, 0] + cdim[0] * (cc[:, 1] + cdim[1] * cc[:, 2])
    fine = (rel[:, 0] & 7) | ((rel[:, 1] & 7) << 3) | ((rel[:, 2] & 7) << 6)
    key = clin * 512 + fine
    order = np.argsort(key, kind='stable').astype(np.int32)
    ks = key[order]
    ublk = np.unique(clin)
    coarse_block = -np.ones(int(np.prod(cdim)), np.int32)
    coarse_block[ublk] = np.arange(len(ublk), dtype=np.int32)
    blk_of = coarse_block[clin[order]]
    fkey = blk_of.astype(np.int64) * 513 + fine[order]
    fine_start = np.searchsorted(fkey, np.arange(len(ublk) * 513), side='left').astype(np.int32)
    block_off = np.searchsorted(blk_of, np.arange(len(ublk) + 1), side='left').astype(np.int32)
    s = pts[order]
    return dict(cell=cell, org=org.astype(np.int32), cdim=cdim.astype(np.int32), coarse_block=coarse_block, block_off=block_off,
                fine_start=fine_start, sx=np.ascontiguousarray(s[:, 0]), sy=np.ascontiguousarray(s[:, 1]), sz=np.ascontiguousarray(s[:, 2]), order=order)
def P(a, t): return a.ctypes.data_as(C.POINTER(t))
def knn(g, q, k, fr):
    n = len(q); idx = np.zeros((n, k), np.int32); d2 = np.zeros((n, k), np.float32)
    qx, qy, qz = [np.ascontiguousarray(q[:, i]) for i in range(3)]
    lib.run_knn(C.c_float(g['cell']), P(g['org'], C.c_int), P(g['cdim'], C.c_int), P(g['coarse_block'], C.c_int), P(g['block_off'], C.c_int), P(g['fine_start'], C.c_int),
                P(g['sx'], C.c_float), P(g['sy'], C.c_float), P(g['sz'], C.c_float), P(g['order'], C.c_int), len(g['order']), P(qx, C.c_float), P(qy, C.c_float), P(qz, C.c_float), n, k, fr, P(idx, C.c_int), P(d2, C.c_float))
    return idx, d2
rng = np.random.default_rng(0)
from lidarslam_ros2_amd import synth
case = synth.small_case()
for name, pts in (("uniform", rng.uniform(-10, 10, (5000, 3)).astype(np.float32)), ("scan", case.source), ("target", case.target[:20000])):
    for cell, k, fr in ((1.0, 20, 2), (0.5, 1, 1), (0.5, 20, 2)):
        g = build(pts, cell)
        t = time.time(); idx, d2 = knn(g, pts[:3000], k, fr); dt = time.time() - t
        tr = cKDTree(pts.astype(np.float64)); dd, ii = tr.query(pts[:3000].astype(np.float64), k=k)
        if k == 1: dd = dd[:, None]; ii = ii[:, None]
        ok = np.allclose(np.sqrt(d2), dd, atol=1e-4)
        print(name, cell, k, fr, "ok" if ok else "MISMATCH", "max err", np.abs(np.sqrt(d2) - dd).max(), "%.2fs" % dt, flush=True)
# usage: g++ -O1 -std=c++17 -shared -fPIC -ffp-contract=off harness.cpp -o libnnh.so  (after generating nn_device_emu.hpp:
#   sed -e 's/#include "common.hpp"//' -e 's/^inline NNGridView make_view/static inline NNGridView make_view/' \
#       ../../lidarslam_ros2_amd/csrc/nn_device.hpp > nn_device_emu.hpp ; paths in this script assume /tmp/nnh)
