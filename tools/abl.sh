cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for v in "" _nostage _nob _noepi _nosb _none; do for cin in 128 512; do
  KDIP_LIB_PATH=$R/k-diffusion-inverse-problems_amd/libkdip_hip$v.so timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/abl -o t -- python $R/tools/conv_micro.py 16 $cin 128 256 256 9 3 > /dev/null 2>&1
  python - <<PY
import csv
rows=list(csv.DictReader(open("$R/gpurun_out/abl/t_kernel_trace.csv")))
d=[(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3 for r in rows if 'conv_igemm' in r['Kernel_Name']]
print("variant [$v] Cin=$cin conv us:", [round(x,1) for x in d])
PY
done; done
