cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for cin in 32 128 256 512; do
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/scan_$cin -o t -- python $R/tools/conv_micro.py 16 $cin 128 256 256 9 3 > /dev/null 2>&1
  python - <<PY
import csv
rows=list(csv.DictReader(open("$R/gpurun_out/scan_$cin/t_kernel_trace.csv")))
d=[(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3 for r in rows if 'conv_igemm' in r['Kernel_Name']]
print("Cin=$cin conv us:", [round(x,1) for x in d])
PY
done
