mkdir -p gpurun_out/r04m
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p1 -o b -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-f32-leg --no-graph-leg --no-large-batch > $R/gpurun_out/r04m/stats_bf16.log 2>&1
cp /tmp/p1/*kernel_stats.csv $R/gpurun_out/r04m/r04_kernel_stats.csv
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p2 -o b -- python $R/bench.py --dtype bf16x3 --steps 10 --warmup 3 --no-cpu-baseline --no-f32-leg --no-graph-leg --no-large-batch > $R/gpurun_out/r04m/stats_x3.log 2>&1
cp /tmp/p2/*kernel_stats.csv $R/gpurun_out/r04m/r04_kernel_stats_bf16x3.csv
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p3 -o b -- python $R/bench.py --batch 8 --streams 1 --steps 10 --warmup 3 --no-cpu-baseline --no-f32-leg --no-graph-leg --no-large-batch > $R/gpurun_out/r04m/stats_b8.log 2>&1
cp /tmp/p3/*kernel_stats.csv $R/gpurun_out/r04m/r04_kernel_stats_1stream_b8.csv
cd $R
PMC_TAG=r04 bash tools/pmc_innetwork.sh > gpurun_out/r04m/pmc_innetwork.log 2>&1
bash tools/pmc_x3.sh > gpurun_out/r04m/pmc_x3.log 2>&1
for d in a b c d e; do f=$(find gpurun_out/pmcnet/$d -name "*counter_collection.csv" | head -1); [ -n "$f" ] && cp $f gpurun_out/r04m/pmc_innetwork_${d}_counter_collection.csv; cp gpurun_out/pmcnet/dump_$d.csv gpurun_out/r04m/pmc_innetwork_${d}_launch_records.csv; done
ls -la gpurun_out/r04m gpurun_out/pmcnet | head -40
