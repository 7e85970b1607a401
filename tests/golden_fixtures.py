"""TEST INFRASTRUCTURE: where a golden fixture comes from.  `tests/golden/<name>.npz` is the CPU ORACLE's output (this repository's own
restatement of ndt_omp / PCL: "parity unpinned").  `tests/golden/ref_<name>.npz` — same arrays, same names — is what
oracle/ref_recipe/ writes on a machine that can build the REFERENCE itself (PCL 1.12 + rsasaki0109/ndt_omp_ros2); when such a file is
present every test that reads the fixture prefers it, and says so: the day the reference's sources are reachable, pinning parity is
`make -C oracle ref` + `pytest`.  LSR_GOLDEN_DIR overrides the directory (tests of this loader use it)."""
import os

import numpy as np

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def golden_dir() -> str:
    return os.environ.get("LSR_GOLDEN_DIR", HERE)


def load_golden(name: str):
    """-> (npz, origin) with origin 'reference' (ref_<name>.npz present) or 'oracle'.  Arrays the reference dump does not hold (a
    quantity only the oracle defines, e.g. the Gauss-Newton GICP pose) are taken from the oracle fixture."""
    d = golden_dir()
    ref, orc = os.path.join(d, "ref_" + name + ".npz"), os.path.join(d, name + ".npz")
    if not os.path.exists(orc) and d != HERE:
        orc = os.path.join(HERE, name + ".npz")
    if os.path.exists(ref):
        r, o = np.load(ref), (np.load(orc) if os.path.exists(orc) else None)
        merged = {k: r[k] for k in r.files}
        if o is not None:
            for k in o.files:
                merged.setdefault(k, o[k])
        print("[golden] %s: REFERENCE fixture %s" % (name, ref))
        return merged, "reference"
    return np.load(orc), "oracle"


def load_reference(name: str):
    """-> dict of arrays from tests/golden/ref_<name>.npz (a dump of the REFERENCE's own code, oracle/ref_recipe) or None when no such
    dump has been made.  For the sequences whose oracle side is computed live by the tests (the frontend drive, the loop gate): the
    tests hold the device to the oracle always and to the reference as well once the file exists."""
    path = os.path.join(golden_dir(), "ref_" + name + ".npz")
    if not os.path.exists(path):
        return None
    r = np.load(path)
    print("[golden] %s: REFERENCE dump %s" % (name, path))
    return {k: r[k] for k in r.files}
