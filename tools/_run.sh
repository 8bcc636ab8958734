mkdir -p gpurun_out/r04y
KDIP_PROFILE_DUMP=$GRAFT_REPO_ROOT/gpurun_out/r04y/bench_dispatches.csv python bench.py > gpurun_out/r04y/bench_default.json 2> gpurun_out/r04y/bench_default.err
