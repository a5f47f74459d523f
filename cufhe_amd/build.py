"""Build libcufhe_amd.so in-tree with hipcc for gfx950 (no JIT cache: the .so travels to the GPU box)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "csrc", "capi.hip")
DEPS = [os.path.join(HERE, "csrc", f) for f in ("capi.hip", "kernels.hip.h", "kernels_lvl2.hip.h", "kernels_ll.hip.h", "ntt_wave512.h", "lvl2.inc.h", "sched.inc.h", "ntt_wave.h", "fpfield.h")] + \
       [os.path.join(os.path.dirname(HERE), "include", "cufhe_amd.h")]
OUT = os.path.join(HERE, "libcufhe_amd.so")
# -ffp-contract=off: the field arithmetic spells out every fma; nothing may be re-fused
FLAGS = ["-O3", "--offload-arch=gfx950", "-ffp-contract=off", "-fPIC", "-shared", "-std=c++17"]


def build(force=False, verbose=False):
    if not force and os.path.exists(OUT) and all(os.path.getmtime(OUT) >= os.path.getmtime(d) for d in DEPS):
        return OUT
    cmd = ["hipcc"] + FLAGS + ["-o", OUT, SRC]
    if verbose:
        cmd.insert(1, "-Rpass-analysis=kernel-resource-usage")
    subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose="-v" in sys.argv))
