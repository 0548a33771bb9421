#!/bin/bash
# bash scripts/kstats_p.sh TAG P1 P2 ...  -> per-kernel durations of the hot path at several batch sizes (TOHC_STREAMS / TOHC_LIB from the caller)
TAG=$1; shift
export TMPDIR=/tmp
for P in "$@"; do
  rm -rf /tmp/prof_$TAG
  timeout -s KILL 240 rocprofv3 --kernel-trace -d /tmp/prof_$TAG -o r -- python3 bench.py --hot-path-only --no-cpu-baseline --steps 5 --warmup 1 --min-seconds 0.5 --problems-per-gpu $P > gpurun_out/${TAG}_$P.json 2> gpurun_out/${TAG}_$P.err || { echo FAILED; tail -n 5 gpurun_out/${TAG}_$P.err; exit 1; }
  python3 scripts/rocpd_summary.py stats $(find /tmp/prof_$TAG -name '*.db' | head -1) gpurun_out/${TAG}_${P}_kstats.csv
  echo "== P=$P"; grep -E "k_tail_epa|k_tail_pen|k_gjk<|k_finish|k_broadphase" gpurun_out/${TAG}_${P}_kstats.csv | cut -d, -f1-4,6,7 | cut -c1-110
done
