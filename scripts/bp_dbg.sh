#!/bin/bash
# broadphase time split (timing experiment): TOHC_BP_DBG 1 = scan only, 2 = + sphere stage, 0 = everything
for dbg in 0 1 2; do
  TOHC_BP_DBG=$dbg TOHC_LIB=$1 timeout -k 10 200 python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --hot-path-only > gpurun_out/bpdbg_$dbg.json 2> gpurun_out/bpdbg_$dbg.err || { echo FAIL $dbg; tail -3 gpurun_out/bpdbg_$dbg.err; }
  python3 -c "
import json;d=json.loads(open('gpurun_out/bpdbg_$dbg.json').read().strip().splitlines()[-1]);print('dbg', $dbg, round(d['ms_per_step'],3), {k:round(x,3) for k,x in d['kernel_ms_per_launch'].items()})"
done
