#!/bin/bash
# tools/profile_round.sh TAG -- the rocprofv3 passes behind profiles/<TAG>_* and profiles/kernel_facts.json.
# Run on the GPU box from the repo root (gpurun -- 'bash tools/profile_round.sh r03').  The profiled command is the default
# bench run (every workload of the driver line: the headline batch, mux, mixed, the N = 2048 ring, a 512-gate launch on the
# paired low-latency kernel, the four parameter sets, the key switches): one --stats pass and five --pmc passes, each its own
# process with python3 directly behind `--`, --pmc never combined with --stats / sys traces (MI355X_MICROARCH.md, rocprofv3
# PMC slots: 8 SQ counters per pass; FETCH_SIZE and WRITE_SIZE do not fit one pass).
set -e
TAG=${1:-r03}
PART=${2:-all}     # stats | pmc | all: the passes take about 15 minutes together; a gpurun call is limited to 20
export TMPDIR=/tmp
OUT=gpurun_out/${TAG}_prof
mkdir -p $OUT
if [ "$PART" != "pmc" ]; then
rocprofv3 --kernel-trace --stats -d $OUT/stats -o st --output-format csv -- python3 bench.py --steps 5 --warmup 1 > $OUT/bench_under_rocprof.json
echo "stats pass done"
# one launch shape per kernel row: the headline workload alone (4096 NAND: every blind_rotate_kernel launch is 4096 rotations, every
# keyswitch_kernel launch 4096 ciphertexts), and the N = 2048 ring alone -- the AVERAGE of the row is then the launch the bench line prices
rocprofv3 --kernel-trace --stats -d $OUT/stats_nand -o st --output-format csv -- python3 bench.py --steps 10 --warmup 2 --no-extra --no-cpu-baseline > $OUT/nand_bench_under_rocprof.json
echo "nand-only stats pass done"
rocprofv3 --kernel-trace --stats -d $OUT/stats_lvl2 -o st --output-format csv -- python3 bench.py --workload nand_lvl2 --steps 3 --warmup 1 --no-extra --no-cpu-baseline > $OUT/nand_lvl2_bench_under_rocprof.json
echo "lvl2-only stats pass done"
fi
if [ "$PART" = "stats" ]; then exit 0; fi
rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_LDS GRBM_GUI_ACTIVE \
    --kernel-trace -d $OUT/pmc_sq -o pmc --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-api > /dev/null
echo "sq pass done"
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_BUSY_CYCLES \
    --kernel-trace -d $OUT/pmc_lds -o pmc --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-api > /dev/null
echo "lds pass done"
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/pmc_fetch -o pmc --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-api > /dev/null
echo "fetch pass done"
rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --kernel-trace -d $OUT/pmc_tcc -o pmc --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-api > /dev/null
echo "tcc pass done"
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_ACTIVE_INST_VALU \
    --kernel-trace -d $OUT/pmc_mix -o pmc --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-api > /dev/null
echo "instruction-mix pass done"
# the two blind-rotate kernels of the N = 2048 ring side by side (HIP events), their counters in one PMC pass, and -- when a
# diagnostic build with per-phase cycle counters was shipped (python cufhe_amd/build.py --diagnostic=PHASES) -- the cycles per phase
python3 tools/lvl2_ab.py 4096 2>&1 | grep -v amdgpu.ids > $OUT/lvl2_ab.txt
rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_LDS GRBM_GUI_ACTIVE \
    --kernel-trace -d $OUT/pmc_lvl2ab -o pmc --output-format csv -- python3 tools/lvl2_ab.py 4096 > /dev/null 2>&1
echo "lvl2 A/B pass done"
if [ -f cufhe_amd/libcufhe_amd_diag.so ]; then
    CUFHE_AMD_LIBRARY=cufhe_amd/libcufhe_amd_diag.so LVL2_KERNEL=0 python3 tools/lvl2_phases.py 2>&1 | grep -v amdgpu.ids > $OUT/lvl2_phases_half_waves.txt || true
    CUFHE_AMD_LIBRARY=cufhe_amd/libcufhe_amd_diag.so LVL2_KERNEL=1 python3 tools/lvl2_phases.py 2>&1 | grep -v amdgpu.ids > $OUT/lvl2_phases_quarter_waves.txt || true
fi
find $OUT -name "*.csv" | head -40
