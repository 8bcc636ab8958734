# FETCH_SIZE / WRITE_SIZE calibration passes (separate --pmc runs, kernel trace only) -> gpurun_out/r06/fetch_calibration.txt
# usage (GPU box): bash tools/calib/run_calib.sh
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06/calib; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
[ -x $R/tools/calib/fetch_calib ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o $R/tools/calib/fetch_calib $R/tools/calib/fetch_calib.hip
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/f -o pmc -- $R/tools/calib/fetch_calib > $O/f.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/w -o pmc -- $R/tools/calib/fetch_calib > $O/w.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum --output-format csv -d $O/r -o pmc -- $R/tools/calib/fetch_calib > $O/r.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum --output-format csv -d $O/q -o pmc -- $R/tools/calib/fetch_calib > $O/q.log 2>&1
cd $R && python tools/calib/fold_calib.py $O > gpurun_out/r06/fetch_calibration.txt; cat gpurun_out/r06/fetch_calibration.txt
