"""Build the HIP core (hipcc --offload-arch=gfx950) into lidarslam_ros2_amd/liblidarslam_reg.so."""
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))


def build(verbose: bool = False, jobs: int = 8) -> str:
    csrc = os.path.join(_HERE, "csrc")
    cmd = ["make", "-C", csrc, f"-j{jobs}"]
    if not verbose:
        cmd.insert(1, "-s")
    subprocess.check_call(cmd)
    out = os.path.join(_HERE, "liblidarslam_reg.so")
    if not os.path.exists(out):
        raise RuntimeError("hipcc build did not produce liblidarslam_reg.so")
    return out


if __name__ == "__main__":
    print(build(verbose=True))
