"""Build libcufhe_amd.so in-tree with hipcc for gfx950 (no JIT cache: the .so travels to the GPU box)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "csrc", "capi.hip")
DEPS = [os.path.join(HERE, "csrc", f) for f in ("capi.hip", "kernels.hip.h", "kernels_lvl2.hip.h", "kernels_ks2.hip.h", "kernels_ll.hip.h", "kernels_ps.hip.h", "paramsets.inc.h", "ntt_wave512.h", "lvl2.inc.h", "sched_hip.inc.h", "sched_core.h", "ntt_wave.h", "fpfield.h")] + \
       [os.path.join(os.path.dirname(HERE), "include", "cufhe_amd.h")]
OUT = os.path.join(HERE, "libcufhe_amd.so")
# -ffp-contract=off: the field arithmetic spells out every fma; nothing may be re-fused
FLAGS = ["-O3", "--offload-arch=gfx950", "-ffp-contract=off", "-fPIC", "-shared", "-std=c++17"]


DIAG_OUT = os.path.join(HERE, "libcufhe_amd_diag.so")


def build(force=False, verbose=False, diagnostic=None, extra=()):
    """diagnostic: list of ablation switch names (NO_TW, NO_XPOSE, NO_BK, BK0): a timing-only build whose
    results are WRONG; it goes to libcufhe_amd_diag.so and is never loaded by the package."""
    if diagnostic:
        cmd = ["hipcc"] + FLAGS + ["-DCUFHE_AMD_DIAGNOSTIC_BUILD"] + [f"-DCUFHE_AMD_ABL_{d}" for d in diagnostic] + \
              list(extra) + ["-o", DIAG_OUT, SRC]
        subprocess.check_call(cmd)
        return DIAG_OUT
    if not force and os.path.exists(OUT) and all(os.path.getmtime(OUT) >= os.path.getmtime(d) for d in DEPS):
        return OUT
    cmd = ["hipcc"] + FLAGS + list(extra) + ["-o", OUT, SRC]
    if verbose:
        cmd.insert(1, "-Rpass-analysis=kernel-resource-usage")
    subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    diag = [a.split("=", 1)[1].split(",") for a in sys.argv if a.startswith("--diagnostic=")]
    print(build(force="--force" in sys.argv, verbose="-v" in sys.argv, diagnostic=diag[0] if diag else None))
