#!/bin/bash
# bash scripts/bench_libs.sh TAG name1 name2 ...  -> a short hot-path bench (default workload) per experiment build
# scripts/native/libtohc_<name>.so ("base" = the product library); BENCH_ARGS adds bench.py arguments (GPU box)
TAG=$1; shift
for v in "$@"; do
  L=""; [ "$v" != base ] && L=$PWD/scripts/native/libtohc_$v.so
  TOHC_LIB=$L timeout -k 10 200 python3 bench.py --no-cpu-baseline --hot-path-only --min-seconds 1 $BENCH_ARGS > gpurun_out/${TAG}_$v.json 2> gpurun_out/${TAG}_$v.err || { echo "FAILED $v"; tail -n 3 gpurun_out/${TAG}_$v.err; continue; }
  python3 - "$v" gpurun_out/${TAG}_$v.json <<'PY'
import json, sys
j = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
print("%-24s %.3f G  %.3f ms/step " % (sys.argv[1], j["value"] / 1e9, j["ms_per_step"]), {k: round(v, 3) for k, v in j["kernel_ms_per_pass"].items()}, "contacts", j["contacts_per_step_per_gpu"], flush=True)
PY
done
