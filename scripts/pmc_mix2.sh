#!/bin/bash
# bash scripts/pmc_mix2.sh TAG  -> instruction mix + waits per kernel launch for the hot path (TOHC_LIB / TOHC_STREAMS from the caller)
TAG=$1
export TMPDIR=/tmp
rm -rf /tmp/pm_$TAG
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d /tmp/pm_$TAG -o r -- python3 bench.py --hot-path-only --no-cpu-baseline --steps 3 --warmup 1 --min-seconds 0.2 > gpurun_out/${TAG}_mix.json 2> gpurun_out/${TAG}_mix.err || { echo FAILED; tail -n 5 gpurun_out/${TAG}_mix.err; exit 1; }
python3 scripts/rocpd_summary.py pmc $(find /tmp/pm_$TAG -name '*.db' | head -1) gpurun_out/${TAG}_mix.csv
grep -E "kernel,|k_tail_epa|k_tail_pen|k_gjk<1" gpurun_out/${TAG}_mix.csv | cut -c1-230
