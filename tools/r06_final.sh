#!/bin/bash
# round 6 final-tree evidence in one lease: profiles (kernel statistics + in-network PMC, f16x3), default bench with dispatch dumps, smoke, every BASELINE
# config in f16x3 and bf16x3 (cfg3), the full GPU suite.  usage: bash tools/r06_final.sh TAG   -> gpurun_out/r06/
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; T=${1:-r06k}; O=gpurun_out/r06; mkdir -p $O
bash tools/r06_profiles_f16x3.sh > $O/${T}_profiles.log 2>&1
cd $R
[ -s $O/r06_pmc_innetwork_f16x3.json ] && cp $O/r06_pmc_innetwork_f16x3.json profiles/r06_pmc_innetwork_f16x3.json      # the bench line below quotes this lease's counters
KDIP_PROFILE_DUMP=$O/${T}_bench_dispatches.csv timeout 900 python bench.py --steps 20 --warmup 5 2> $O/${T}_bench_default.err | tail -1 > $O/${T}_bench_default.json
python -c "import json; d=json.load(open('$O/${T}_bench_default.json')); print('bench', d['ms_per_step'], d['value'], d['dtype'], d['roofline']['avg_launch_us'], d['roofline']['frac'], d['bf16x3_parity_mode']['ms_per_step'], d['speedup_vs_cpu_baseline'])"
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke_${T}.log 2>&1; tail -1 $O/smoke_${T}.log
DTYPE=f16x3 bash tools/r06_workloads.sh 2>&1 | tail -6; mkdir -p $O/workloads_f16x3_$T; mv $O/bench_cfg*.json $O/configs_b16_tests.log $O/workloads_f16x3_$T/ 2>/dev/null
timeout 900 python bench.py --workload cfg3 --dtype bf16x3 --steps 10 --warmup 2 2>/dev/null | tail -1 > $O/workloads_f16x3_$T/bench_cfg3_bf16x3.json
python -c "import json; d=json.load(open('$O/workloads_f16x3_$T/bench_cfg3_bf16x3.json')); print('cfg3 bf16x3', d['ms_per_step'], d['value'])"
timeout 1800 python -m pytest tests/ -x -q -m gpu 2>&1 | tail -8 > $O/gputests_${T}_final.log; tail -2 $O/gputests_${T}_final.log
