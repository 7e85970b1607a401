"""TEST INFRASTRUCTURE: builds tests/cpp/stub_ccl.cpp — the shared-memory stand-in for the collective library that lets two
PROCESSES on ONE device run csrc/comm.hip with world = 2 (LSR_RCCL_LIB) — once per checkout."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "cpp", "stub_ccl.cpp")
OUT = os.path.join(ROOT, "tests", "cpp", "_build", "libstub_ccl.so")


def build_stub() -> str:
    if not os.path.exists(OUT) or os.path.getmtime(OUT) < os.path.getmtime(SRC):
        os.makedirs(os.path.dirname(OUT), exist_ok=True)
        tmp = OUT + f".{os.getpid()}.tmp"
        subprocess.check_call(["g++", "-O1", "-std=c++17", "-Wall", "-shared", "-fPIC", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", SRC,
                               "-o", tmp, "-L/opt/rocm/lib", "-lamdhip64", "-lrt", "-pthread", "-Wl,-rpath,/opt/rocm/lib"])
        os.replace(tmp, OUT)
    return OUT


def two_rank_env(n_devices: int) -> dict:
    """Environment for a two-process run: the box's own RCCL with two devices, the stub (both ranks on device 0) with one."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    if n_devices < 2:
        env["LSR_RCCL_LIB"] = build_stub()
    return env
