export TMPDIR=/tmp
mkdir -p gpurun_out/r04_l2q_pmc
rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_LDS GRBM_GUI_ACTIVE --kernel-trace -d gpurun_out/r04_l2q_pmc/sq -o pmc --output-format csv -- python3 tools/lvl2_ab.py 4096 > gpurun_out/r04_l2q_pmc/run.txt 2>&1
