# rocprofv3 PMC passes over the IN-NETWORK launches of the conv3 kernels (one guided Heun step of the bench workload, 8 images
# per launch), one pass per counter group (never combined with tracing domains other than --kernel-trace); joined per
# (fusion mode, layer shape) into profiles/<PMC_TAG>_pmc_innetwork.json (default r03), which bench.py reports as roofline.traffic.
# usage: [PMC_TAG=r05] [PMC_DTYPE=bf16|bf16x3|f32] bash tools/pmc_innetwork.sh
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmcnet
rm -rf $O; mkdir -p $O
export PMC_DTYPE=${PMC_DTYPE:-bf16}
run() { timeout 600 rocprofv3 --kernel-trace --pmc $2 --output-format csv -d $O/$1 -o pmc -- python $R/tools/pmc_step.py $O/dump_$1.csv > $O/$1.log 2>&1; }
run a "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU"
run b "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_INSTS_MFMA"
run c "FETCH_SIZE"
run d "WRITE_SIZE"
run e "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum"
cd $R && python tools/pmc_join.py   # (also writes gpurun_out/pmcnet/<PMC_TAG>_pmc_innetwork.json, which gpurun merges back)
