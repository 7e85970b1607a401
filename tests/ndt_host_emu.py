"""TEST INFRASTRUCTURE: the NDT derivative pass of the GPU kernels evaluated on the CPU (tools/ndt_host_emu/harness.cpp compiles
lidarslam_ros2_amd/csrc/ndt_point.hpp for the host) and plugged into the CPU oracle's Newton / More-Thuente loop
(oracle.ndt_align(deriv_cb=...)): a registration with the GPU's fp32 OPERATION ORDER (factorised pair terms, fp32 quad sums,
fp32-rounded voxel means, fmaf point transform) but the oracle's controller — no GPU needed."""
import ctypes as C
import os
import subprocess
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_lib = None


def build():
    """g++ -O2 -mfma -ffp-contract=off: explicit fmaf() becomes the hardware FMA, nothing else is contracted."""
    global _lib
    if _lib is not None:
        return _lib
    src = os.path.join(ROOT, "tools", "ndt_host_emu", "harness.cpp")
    hdr = os.path.join(ROOT, "lidarslam_ros2_amd", "csrc", "ndt_point.hpp")
    libdir = os.path.join(ROOT, "lidarslam_ros2_amd")
    import hashlib
    # one build directory per checkout: the library is linked with an rpath into THIS tree
    out = os.path.join(tempfile.gettempdir(), "lsr_ndt_host_emu_%d_%s" % (os.getuid(), hashlib.sha1(ROOT.encode()).hexdigest()[:10]))
    os.makedirs(out, exist_ok=True)
    so = os.path.join(out, "libndtemu.so")
    if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-mfma", "-ffp-contract=off", src, "-o", so + ".tmp", "-L" + libdir,
                               "-llidarslam_reg", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"])
        os.replace(so + ".tmp", so)
    L = C.CDLL(so)
    ip, dp, fp, vp = C.POINTER(C.c_int), C.POINTER(C.c_double), C.POINTER(C.c_float), C.c_void_p
    L.emu_create.restype = vp
    L.emu_create.argtypes = [ip, ip, C.c_float, C.c_int, ip, dp, dp, fp, C.c_int, C.c_double, C.c_double, C.c_int]
    L.emu_destroy.argtypes = [vp]
    L.emu_passes.restype = C.c_long
    L.emu_passes.argtypes = [vp]
    L.emu_derivatives.restype = C.c_double
    L.emu_derivatives.argtypes = [vp, dp, fp, C.c_int, dp, dp]
    _lib = L
    return L


class Emu:
    """The GPU-arithmetic derivative pass over (oracle voxel grid, source cloud)."""

    def __init__(self, O, grid, src, resolution, outlier_ratio=0.55, d1_sign=1):
        self.L = build()
        d = grid.dump()
        ok = d["n"] >= 6
        idx = np.ascontiguousarray(d["idx"][ok], np.int32)
        mean = np.ascontiguousarray(d["mean"][ok], np.float64)
        icov = np.ascontiguousarray(d["icov"][ok].reshape(-1, 9), np.float64)
        s = np.ascontiguousarray(np.asarray(src, np.float32)[:, :3])
        d1, d2, _ = O.gauss_constants(resolution, outlier_ratio)
        mn, mx = np.ascontiguousarray(grid.min_b, np.int32), np.ascontiguousarray(grid.max_b, np.int32)
        ip, dp, fp = C.POINTER(C.c_int), C.POINTER(C.c_double), C.POINTER(C.c_float)
        self.h = self.L.emu_create(mn.ctypes.data_as(ip), mx.ctypes.data_as(ip), C.c_float(grid.leaf), len(idx), idx.ctypes.data_as(ip),
                                   mean.ctypes.data_as(dp), icov.ctypes.data_as(dp), s.ctypes.data_as(fp), s.shape[0], d1, d2, d1_sign)
        self.cb = (C.cast(self.L.emu_deriv_cb, C.c_void_p).value, self.h)

    def derivatives(self, p, T=None, with_hessian=True):
        p = np.ascontiguousarray(p, np.float64)
        g, H = np.zeros(6), np.zeros((6, 6))
        Tp = None
        if T is not None:
            Tc = np.ascontiguousarray(np.asarray(T, np.float32).T).reshape(-1)
            Tp = Tc.ctypes.data_as(C.POINTER(C.c_float))
        sc = self.L.emu_derivatives(self.h, p.ctypes.data_as(C.POINTER(C.c_double)), Tp, int(with_hessian), g.ctypes.data_as(C.POINTER(C.c_double)),
                                    H.ctypes.data_as(C.POINTER(C.c_double)))
        return sc, g, H

    def passes(self):
        return int(self.L.emu_passes(self.h))

    def close(self):
        if self.h:
            self.L.emu_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def ndt_align(O, grid, src, guess, resolution, **kw):
    """oracle.ndt_align with the derivative evaluation replaced by the GPU-arithmetic emulation."""
    e = Emu(O, grid, src, resolution, kw.get("outlier_ratio", 0.55), kw.get("d1_sign", 1))
    try:
        return O.ndt_align(grid, src, guess, resolution=resolution, deriv_cb=e.cb, **kw)
    finally:
        e.close()
