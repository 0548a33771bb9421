#!/bin/bash
for c in 1 2 4; do
  TOHC_CHUNKS=$c timeout -k 10 200 python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --hot-path-only > gpurun_out/ch_$c.json 2> gpurun_out/ch_$c.err || { echo FAIL $c; tail -3 gpurun_out/ch_$c.err; }
  python3 -c "
import json;d=json.loads(open('gpurun_out/ch_$c.json').read().strip().splitlines()[-1]);print('chunks', $c, round(d['value']/1e9,3), round(d['ms_per_step'],3), {k:round(x,3) for k,x in d['kernel_ms_per_launch'].items()})"
done
