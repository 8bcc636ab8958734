# rocprofv3 PMC passes on the dominant conv shape of the split-precision (bf16x3) mode: 128->128 @ 256x256, 8 images, fp32 storage,
# conv_igemm_kernel<f32x3_t, 9, 2, 2, 2, 2, 1, 0> (1 bf16 + 2 f16 MFMAs per product).  Separate passes per counter group (never combined
# with tracing domains other than --kernel-trace).  usage: bash tools/pmc_x3.sh   -> gpurun_out/pmc_x3/pmc_x3_conv_micro.json
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_x3
rm -rf $O; mkdir -p $O
run() { timeout 300 rocprofv3 --kernel-trace --pmc $2 --output-format csv -d $O/$1 -o pmc -- python $R/tools/conv_micro.py 8 128 128 256 256 9 5 2 > $O/$1.log 2>&1; }
run a "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU"
run b "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_INSTS_MFMA"
run c "FETCH_SIZE"
run d "WRITE_SIZE"
run e "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum"
# algorithmic bytes: fp32 input + fp32 output + packed weights (2 planes x 2 bytes); algorithmic GFLOP = 2 * MAC (one product per MAC)
python $R/tools/pmc_fold.py $O "conv_igemm_kernel<kdip::f32x3_t, 9, 2, 2, 2, 2, 1, 0>" $((8*256*256*128*4*2 + 9*128*128*4)) 154.619 $O/pmc_x3_conv_micro.json \
  "tools/pmc_x3.sh: python tools/conv_micro.py 8 128 128 256 256 9 5 2 (split-precision conv, 3 MFMAs per product: MFMA busy counts all three)"
