#!/bin/bash
# bash scripts/bench_env.sh TAG "ENV=1 ENV2=2" ...  -> one short hot-path bench per environment set (GPU box);
# BENCH_ARGS adds arguments to bench.py (e.g. BENCH_ARGS="--config multistart")
TAG=$1; shift
for envs in "$@"; do
  name=$(echo "$envs" | tr ' =/' '___' | tail -c 60)
  env $envs timeout -k 10 200 python3 bench.py --no-cpu-baseline --hot-path-only --min-seconds 0.5 $BENCH_ARGS > gpurun_out/${TAG}_${name}.json 2> gpurun_out/${TAG}_${name}.err || echo "FAILED $envs"
  python3 - "$envs" gpurun_out/${TAG}_${name}.json <<'PY'
import json, sys
j = json.loads(open(sys.argv[2]).read())
print("%-40s %.3f G  %.3f ms/step " % (sys.argv[1], j["value"] / 1e9, j["ms_per_step"]), {k: round(v, 3) for k, v in j["kernel_ms_per_pass"].items()}, flush=True)
PY
done
