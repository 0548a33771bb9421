#!/bin/bash
# Profiles of the bench workload (run on the GPU box from the repo root):  bash scripts/profile_bench.sh <tag>
# Every pass profiles the SAME command, `python3 bench.py --hot-path-only --no-cpu-baseline --steps 5 --warmup 1`
# (only the timed workload's launches), each in a run of its own as MI355X_MICROARCH.md prescribes for PMC counters.
# Summaries land in gpurun_out/<tag>_*.csv / .json; copy the ones to keep into profiles/.
set -e -o pipefail
TAG=${1:-prof}
OUT=$PWD/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
CMD="python3 bench.py --hot-path-only --no-cpu-baseline --steps 5 --warmup 1"
run() {   # name, rocprofv3 options...
  local name=$1; shift
  rm -rf /tmp/prof_$name
  timeout -s KILL 240 rocprofv3 "$@" -d /tmp/prof_$name -o r -- $CMD > $OUT/${TAG}_bench_under_${name}.json 2> $OUT/${TAG}_${name}.err
  find /tmp/prof_$name -name '*.db' | head -1
}
DB=$(run stats --kernel-trace)
python3 scripts/rocpd_summary.py stats $DB $OUT/${TAG}_bench_kernel_stats.csv
echo "stats done"
DBF=$(run fetch --pmc FETCH_SIZE)
echo "fetch done"
DBW=$(run write --pmc WRITE_SIZE)
python3 scripts/rocpd_summary.py hbm $DBF $DBW $OUT/${TAG}_hbm_traffic_pmc.csv
echo "hbm done"
DBS=$(run sq --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE)
python3 scripts/rocpd_summary.py pmc $DBS $OUT/${TAG}_sq_pmc.csv
echo "sq done"
