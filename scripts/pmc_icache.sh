#!/bin/bash
# bash scripts/pmc_icache.sh TAG  -> instruction-cache requests / hits / misses per kernel launch of the hot path
TAG=$1
export TMPDIR=/tmp
rm -rf /tmp/pi_$TAG
timeout -s KILL 120 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU -d /tmp/pi_$TAG -o r -- python3 bench.py --hot-path-only --no-cpu-baseline --steps 3 --warmup 1 --min-seconds 0.2 > gpurun_out/${TAG}_ic.json 2> gpurun_out/${TAG}_ic.err || { echo FAILED; tail -n 5 gpurun_out/${TAG}_ic.err; exit 1; }
python3 scripts/rocpd_summary.py pmc $(find /tmp/pi_$TAG -name '*.db' | head -1) gpurun_out/${TAG}_ic.csv
grep -E "kernel,|k_tail_epa|k_tail_pen|k_gjk|k_finish|k_broadphase|k_exact" gpurun_out/${TAG}_ic.csv | cut -c1-200
