#!/bin/bash
# bash scripts/probe_single_env.sh "ENV=1 ENV2=2" ...  -> scripts/probe_single.py once per environment set (GPU box); PROBE_ARGS = its arguments
for envs in "$@"; do
  echo -n "$envs: "
  env $envs timeout -k 10 120 python3 scripts/probe_single.py $PROBE_ARGS 2>&1 | tail -n 1
done
