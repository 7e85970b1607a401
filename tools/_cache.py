"""Disk cache for the deterministic synthetic workloads of the probe scripts (LSR_BENCH_CACHE_DIR, set by
tools/round_profiles.sh): the rocprofv3 / PMC passes re-run one probe several times inside a GPU session."""
import os
import pickle


def cached(key, make):
    d = os.environ.get("LSR_BENCH_CACHE_DIR")
    if not d:
        return make()
    path = os.path.join(d, key + ".pkl")
    if os.path.exists(path):
        with open(path, "rb") as f:
            return pickle.load(f)
    v = make()
    os.makedirs(d, exist_ok=True)
    with open(path + ".tmp", "wb") as f:
        pickle.dump(v, f, protocol=4)
    os.replace(path + ".tmp", path)
    return v
