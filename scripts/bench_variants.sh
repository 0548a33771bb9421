#!/bin/bash
# usage: bench_variants.sh "<variant names>" "<problems per gpu list>"  -- runs bench.py on experiment builds (scripts/native/libtohc_<name>.so)
for v in $1; do for P in $2; do
  L=""; [ "$v" != base ] && L=$PWD/scripts/native/libtohc_$v.so
  TOHC_LIB=$L timeout -k 10 200 python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --problems-per-gpu $P > gpurun_out/bv_${v}_$P.json 2> gpurun_out/bv_${v}_$P.err || { echo FAIL $v $P; tail -3 gpurun_out/bv_${v}_$P.err; exit 1; }
  python3 -c "
import json;d=json.loads(open('gpurun_out/bv_${v}_$P.json').read().strip().splitlines()[-1]);print('$v', $P, round(d['value']/1e9,3), round(d['ms_per_step'],3), {k:round(x,3) for k,x in d['kernel_ms_per_launch'].items()}, round(d['ms_per_sqp_linearization_single_problem_device'],3))"
done; done
