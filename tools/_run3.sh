mkdir -p gpurun_out/r04_run3
for cfg in "RES=0 ACC=0 ACT=1" "RES=1 ACC=0 ACT=0" "RES=1 ACC=1 ACT=1 K=3" "C=256 T=8192 B=8 RES=1 ACC=1 ACT=1 K=11 D=3"; do echo "=== $cfg"; env $cfg timeout 300 python tools/pipe_dbg.py 2>&1 | grep -v "amdgpu.ids\|Runtime\|return ufunc\|print("; done > gpurun_out/r04_run3/dbg.txt 2>&1
cut -c1-400 gpurun_out/r04_run3/dbg.txt
