export TMPDIR=/tmp
rm -rf /tmp/prof_p2
timeout -s KILL 240 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_ACTIVE_INST_SCA SQ_WAVES -d /tmp/prof_p2 -o r -- python3 bench.py --hot-path-only --no-cpu-baseline --steps 5 --warmup 1 > gpurun_out/p2.json 2> gpurun_out/p2.err
python3 scripts/rocpd_summary.py pmc $(find /tmp/prof_p2 -name '*.db' | head -1) gpurun_out/p2_pmc.csv
