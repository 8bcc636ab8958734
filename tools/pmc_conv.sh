# rocprofv3 PMC passes on the dominant conv shape (128->128 @ 256x256, bf16 storage-dtype epilogue); separate passes
# per counter group (never combined with tracing domains other than --kernel-trace).  usage: bash tools/pmc_conv.sh [batch]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
B=${1:-8}    # per-launch batch = bench batch 16 / 2 streams
run() { timeout 300 rocprofv3 --kernel-trace --pmc $2 --output-format csv -d $R/gpurun_out/pmc_$1 -o pmc -- python $R/tools/conv_micro.py $B 128 128 256 256 9 3 > $R/gpurun_out/pmc_$1.log 2>&1; }
run a "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU"
run b "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_MFMA"
run c "FETCH_SIZE"
run d "WRITE_SIZE"
run e "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum"
cd $R && python tools/pmc_to_json.py $B
