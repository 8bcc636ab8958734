# round 6, headline arithmetic f16x3: rocprofv3 kernel statistics (two streams / one part-batch alone) + in-network PMC passes -> gpurun_out/r06/
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
stats() { n=$1; shift; rm -rf /tmp/prof_$n
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$n -o bench -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-f32-leg --no-graph-leg --no-large-batch --no-roofline "$@" > $O/${n}_bench.json 2> $O/${n}_bench.err
  f=$(find /tmp/prof_$n -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/r06_kernel_stats_$n.csv; }
stats f16x3 --dtype f16x3
stats f16x3_1stream_b8 --dtype f16x3 --batch 8 --streams 1
cd $R
PMC_TAG=r06 PMC_DTYPE=f16x3 bash tools/pmc_innetwork.sh > $O/pmc_innetwork_f16x3.log 2>&1
for d in a b c d e; do f=$(find gpurun_out/pmcnet/$d -name "*counter_collection.csv" | head -1); [ -n "$f" ] && cp $f $O/pmc_innetwork_f16x3_${d}_counter_collection.csv; done
cp gpurun_out/pmcnet/dump_a.csv $O/pmc_innetwork_f16x3_launch_records.csv 2>/dev/null
cp gpurun_out/pmcnet/r06_pmc_innetwork_f16x3.json $O/ 2>/dev/null
tail -4 $O/pmc_innetwork_f16x3.log | cut -c1-300
