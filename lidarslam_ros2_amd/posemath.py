"""Small SE(3) helpers shared by tests, smoke and bench (host side, numpy)."""
import numpy as np


def pose_delta(A, B):
    """(translation distance [m], rotation angle [rad]) between two 4x4 poses.

    The angle comes from the skew part of dR (sin theta), not arccos(trace): arccos near 1 turns the
    6e-8 rounding of fp32 matrix entries into ~3e-4 rad of fake rotation."""
    A, B = np.asarray(A, np.float64), np.asarray(B, np.float64)
    dt = float(np.linalg.norm(A[:3, 3] - B[:3, 3]))
    dR = A[:3, :3] @ B[:3, :3].T
    v = 0.5 * np.array([dR[2, 1] - dR[1, 2], dR[0, 2] - dR[2, 0], dR[1, 0] - dR[0, 1]])
    s = float(np.linalg.norm(v))
    c = (np.trace(dR) - 1.0) / 2.0
    return dt, float(np.arctan2(s, c))
