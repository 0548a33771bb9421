#!/bin/bash
# bash scripts/pmc_mem.sh TAG  -> vector-memory path per kernel launch of the hot path: TA busy, L1 (TCP) accesses and misses to L2,
# stalls, and the mean latency of a vector memory instruction (SQ_INST_LEVEL_VMEM / SQ_INSTS_VMEM); TOHC_STREAMS=1 from the caller
# keeps the kernels apart
TAG=$1
export TMPDIR=/tmp
rm -rf /tmp/pm_$TAG
timeout -s KILL 120 rocprofv3 --pmc TA_BUSY_avr TA_FLAT_READ_WAVEFRONTS_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d /tmp/pm_$TAG -o r -- python3 bench.py --hot-path-only --no-cpu-baseline --steps 3 --warmup 1 --min-seconds 0.2 > gpurun_out/${TAG}_mem.json 2> gpurun_out/${TAG}_mem.err || { echo FAILED; tail -n 5 gpurun_out/${TAG}_mem.err; exit 1; }
python3 scripts/rocpd_summary.py pmc $(find /tmp/pm_$TAG -name '*.db' | head -1) gpurun_out/${TAG}_mem.csv
grep -E "kernel,|k_tail_epa|k_tail_pen|k_gjk|k_finish|k_broadphase_both|k_exact" gpurun_out/${TAG}_mem.csv | cut -c1-400
