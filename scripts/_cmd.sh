set -o pipefail
timeout -k 10 400 python3 bench.py > gpurun_out/r03_zzz_fb42_bench.json 2> gpurun_out/r03_zzz_fb42_bench.err || { echo BENCH FAILED; tail -n 5 gpurun_out/r03_zzz_fb42_bench.err; exit 1; }
python3 -c "
import json; j=json.loads(open('gpurun_out/r03_zzz_fb42_bench.json').read().strip().splitlines()[-1])
print(j['value']/1e9, j['ms_per_step'], j['ms_per_sqp_linearization_single_problem_host_delivered'], j['ms_per_sqp_linearization_single_problem_device'], j['kernel_ms_per_pass'], j['side_workloads']['fullbody16']['value']/1e9, j['side_workloads']['fullbody16']['ms_per_step'], j['side_workloads']['fullbody16']['ms_per_sqp_linearization_single_problem_host_delivered'], j['value_host_delivered']/1e9, j['value_host_delivered_hinges']['value_sync']/1e9, j['value_host_delivered_hinges']['value_pipelined']/1e9, j['cpu_baseline']['value'], j['cpu_baseline']['all_cores']['value'], j['roofline']['frac'])"
bash scripts/profile_bench.sh r03_zzz_fb42 &&
TOHC_STREAMS=1 bash scripts/kstats.sh r03_zzz_fb42_serial &&
timeout -k 10 300 python3 bench.py --config multistart --no-cpu-baseline > gpurun_out/r03_zzz_ms_bench.json 2> gpurun_out/r03_zzz_ms_bench.err &&
timeout -k 10 300 python3 bench.py --shard timesteps --no-cpu-baseline > gpurun_out/r03_zzz_shard_timesteps_n1.json 2> gpurun_out/r03_zzz_shard.err
echo rc=$?
