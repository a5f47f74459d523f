#!/bin/bash
# tools/profile_round.sh TAG -- the rocprofv3 passes behind profiles/<TAG>_* and profiles/kernel_facts.json.
# Run on the GPU box from the repo root (gpurun -- 'bash tools/profile_round.sh r02').  One --stats pass and three
# --pmc passes per workload, each pass its own process, --pmc never combined with --stats / sys traces
# (MI355X_MICROARCH.md, rocprofv3 PMC slots: 8 SQ counters per pass; FETCH_SIZE and WRITE_SIZE do not fit one pass).
set -e
TAG=${1:-r02}
export TMPDIR=/tmp
OUT=gpurun_out/${TAG}_prof
mkdir -p $OUT
BENCH="python3 bench.py --no-extra --no-cpu-baseline"
for WL in ${WORKLOADS:-nand nand_lvl2}; do
  STEPS=5; [ $WL = nand_lvl2 ] && STEPS=2
  rocprofv3 --kernel-trace --stats -d $OUT/${WL}_stats -o st --output-format csv -- $BENCH --workload $WL --steps $STEPS > $OUT/${WL}_bench_under_rocprof.json
  rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_LDS GRBM_GUI_ACTIVE \
      --kernel-trace -d $OUT/${WL}_pmc_sq -o pmc --output-format csv -- $BENCH --workload $WL --steps 2 > /dev/null
  rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/${WL}_pmc_fetch -o pmc --output-format csv -- $BENCH --workload $WL --steps 2 > /dev/null
  rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --kernel-trace -d $OUT/${WL}_pmc_tcc -o pmc --output-format csv -- $BENCH --workload $WL --steps 2 > /dev/null
  echo "$WL passes done"
done
find $OUT -name "*.csv" | head -40
