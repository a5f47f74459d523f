"""Build libcufhe_amd.so in-tree with hipcc for gfx950 (no JIT cache: the .so travels to the GPU box)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "csrc", "capi.hip")
# the low-latency kernels are their own translation unit: the max-ilp machine-scheduling strategy makes them 3.5 % faster
# and the N = 512 parameter-set kernel 4 % slower, so it is not a flag for the whole library
SRC_LL = os.path.join(HERE, "csrc", "kernels_ll.hip")
LL_FLAGS = ["-mllvm", "-amdgpu-sched-strategy=max-ilp"]
# the main translation unit: the max-memory-clause strategy groups the LDS reads of a phase (same box:
# blind_rotate_kernel 35.69 -> 35.34 ms per 4096 rotations, the N = 512 parameter-set kernel 33.7 -> 32.8 ms, everything else within 0.2 %;
# max-ilp here: 36.26 ms)
MAIN_FLAGS = ["-mllvm", "-amdgpu-sched-strategy=max-memory-clause"]
DEPS = [os.path.join(HERE, "csrc", f) for f in ("capi.hip", "kernels_ll.hip", "kernels_common.hip.h", "kernels.hip.h", "kernels_lvl2.hip.h", "kernels_lvl2q.hip.h", "kernels_ks2.hip.h", "kernels_ll.hip.h", "kernels_ps.hip.h", "paramsets.inc.h", "ntt_wave512.h", "lvl2.inc.h", "sched_hip.inc.h", "sched_core.h", "ntt_wave.h", "ntt_r4.h", "fpfield.h")] + \
       [os.path.join(os.path.dirname(HERE), "include", "cufhe_amd.h")]
OUT = os.path.join(HERE, "libcufhe_amd.so")
# -ffp-contract=off: the field arithmetic spells out every fma; nothing may be re-fused
FLAGS = ["-O3", "--offload-arch=gfx950", "-ffp-contract=off", "-fPIC", "-std=c++17"]


def _compile_and_link(out, defines=(), extra=(), verbose=False):
    objs = []
    for src, more in ((SRC, MAIN_FLAGS), (SRC_LL, LL_FLAGS)):
        obj = out + "." + os.path.basename(src) + ".o"
        cmd = ["hipcc"] + FLAGS + list(defines) + list(extra) + more + ["-c", src, "-o", obj]
        if verbose:
            cmd.insert(1, "-Rpass-analysis=kernel-resource-usage")
        subprocess.check_call(cmd)
        objs.append(obj)
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-fPIC", "-shared", "-o", out] + objs)
    for o in objs:
        os.remove(o)
    return out


DIAG_OUT = os.path.join(HERE, "libcufhe_amd_diag.so")
# fault injection for tests/test_gpu_fault.py: one inverse wave of the paired low-latency kernel misses a rendezvous
FAULT_OUT = os.path.join(HERE, "libcufhe_amd_diag_ll2_timeout.so")


def build(force=False, verbose=False, diagnostic=None, extra=(), out=None):
    """diagnostic: list of ablation switch names (NO_TW, NO_XPOSE, NO_BK, BK0, PHASES, LL2_TIMEOUT, KS_NO_DMA, KS_NO_DIGITS): a build whose
    results are WRONG by design (timing experiments, fault injection); it goes to libcufhe_amd_diag.so (or `out`)
    and is never loaded by the package."""
    if diagnostic:
        return _compile_and_link(out or DIAG_OUT, ["-DCUFHE_AMD_DIAGNOSTIC_BUILD"] + [f"-DCUFHE_AMD_ABL_{d}" for d in diagnostic], extra)
    if not force and os.path.exists(OUT) and all(os.path.getmtime(OUT) >= os.path.getmtime(d) for d in DEPS):
        return OUT
    return _compile_and_link(OUT, (), extra, verbose)


def build_fault_injection(force=False):
    """libcufhe_amd_diag_ll2_timeout.so, rebuilt when any source is newer."""
    if not force and os.path.exists(FAULT_OUT) and all(os.path.getmtime(FAULT_OUT) >= os.path.getmtime(d) for d in DEPS):
        return FAULT_OUT
    return build(diagnostic=["LL2_TIMEOUT"], out=FAULT_OUT)


if __name__ == "__main__":
    diag = [a.split("=", 1)[1].split(",") for a in sys.argv if a.startswith("--diagnostic=")]
    print(build(force="--force" in sys.argv, verbose="-v" in sys.argv, diagnostic=diag[0] if diag else None))
