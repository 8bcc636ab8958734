# SQ issue / wait / LDS counters of conv3 micro launches (tools/conv3_micro.py): usage: bash tools/pmc_conv3_micro.sh "B Cin Cout H W tf st res"
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmcmicro
rm -rf $O; mkdir -p $O
run() { timeout 300 rocprofv3 --kernel-trace --pmc $2 --output-format csv -d $O/$1 -o pmc -- python $R/tools/conv3_micro.py $ARGS 10 > $O/$1.log 2>&1; }
ARGS="$1"
run a "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU"
run b "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_INSTS_MFMA SQ_ACTIVE_INST_VMEM"
run c "GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_WAIT_INST_VMEM"
python - <<PY
import csv, glob, collections
tot = collections.defaultdict(lambda: [0, 0.0])
for f in glob.glob("$O/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "conv3_kernel" in r["Kernel_Name"]:
            k = r["Counter_Name"]; tot[k][0] += 1; tot[k][1] += float(r["Counter_Value"])
m = {k: v[1] / v[0] for k, v in tot.items()}
for k in sorted(m): print(f"{k:28s} {m[k]:16.1f}")
wc = m.get("SQ_WAVE_CYCLES", 1)
print("wait_any %.3f wait_inst_any %.3f active_inst_any %.3f (of wave cycles)" % (m.get("SQ_WAIT_ANY",0)/wc, m.get("SQ_WAIT_INST_ANY",0)/wc, m.get("SQ_ACTIVE_INST_ANY",0)/wc))
if "SQ_BUSY_CYCLES" in m: print("mfma busy / (4 simd * busy cycles...) raw ratio MFMA_BUSY/BUSY_CYCLES = %.3f" % (m["SQ_VALU_MFMA_BUSY_CYCLES"]/m["SQ_BUSY_CYCLES"]))
PY
