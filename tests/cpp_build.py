"""Builds of the C++ test programs (tests/cpp/*.cpp): plain g++ host code over include/cufhe_amd.hpp and the C ABI.  One place for
the flags, so that every variant of tests/cpp/test_gate_api.cpp is compiled the same way on the CPU (tests/test_capi.py: does it
compile) and on the GPU box (tests/test_gpu_parity.py: does it pass)."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ROCM = os.environ.get("ROCM_PATH", "/opt/rocm")


def hip_flags():
    """the caller's side of Stream::st() (StreamOrdering in test_gate_api.cpp) calls the HIP runtime on the raw handle: host-only
    headers + libamdhip64, still plain g++"""
    if not os.path.exists(os.path.join(ROCM, "include", "hip", "hip_runtime_api.h")):
        return [], []
    return (["-DCUFHE_AMD_TEST_HIP", "-D__HIP_PLATFORM_AMD__", "-I" + os.path.join(ROCM, "include")],
            ["-L" + os.path.join(ROCM, "lib"), "-lamdhip64", "-Wl,-rpath," + os.path.join(ROCM, "lib")])


def build_gate_api(exe_name, defines=(), oracle="oracle", tfhepp=False, opt="-O2", extra=()):
    """tests/cpp/test_gate_api.cpp -> tests/cpp/<exe_name>; returns the path"""
    exe = os.path.join(ROOT, "tests", "cpp", exe_name)
    cdefs, libs = hip_flags()
    cmd = ["g++", opt, "-std=c++17"] + list(defines) + cdefs + list(extra)
    if tfhepp:
        cmd += ["-DCUFHE_AMD_USE_TFHEPP", "-I" + os.path.join(ROOT, "tests", "cpp", "tfhepp_stub")]
    cmd += ["-o", exe, os.path.join(ROOT, "tests", "cpp", "test_gate_api.cpp"),
            "-L" + os.path.join(ROOT, "cufhe_amd"), "-lcufhe_amd", "-L" + os.path.join(ROOT, "oracle"), "-l" + oracle,
            "-Wl,-rpath," + os.path.join(ROOT, "cufhe_amd"), "-Wl,-rpath," + os.path.join(ROOT, "oracle")] + libs
    subprocess.check_call(cmd)
    return exe
