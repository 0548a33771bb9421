#!/bin/bash
# bash scripts/kstats.sh TAG  -> per-kernel durations of the hot path, kernels one after the other (TOHC_STREAMS=1 must be
# set by the caller for that), gpurun_out/TAG_kstats.csv
TAG=${1:-k}
export TMPDIR=/tmp
rm -rf /tmp/prof_$TAG
timeout -s KILL 240 rocprofv3 --kernel-trace -d /tmp/prof_$TAG -o r -- python3 bench.py --hot-path-only --no-cpu-baseline --steps 5 --warmup 1 > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}.err || { echo FAILED; tail -n 5 gpurun_out/${TAG}.err; exit 1; }
python3 scripts/rocpd_summary.py stats $(find /tmp/prof_$TAG -name '*.db' | head -1) gpurun_out/${TAG}_kstats.csv
cut -c1-110 gpurun_out/${TAG}_kstats.csv | head -14
