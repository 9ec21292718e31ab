mkdir -p gpurun_out/r04_run4
./tools/ubench/glds_layout > gpurun_out/r04_run4/glds.txt 2>&1
for cfg in "RES=1 ACC=0 ACT=0" "RES=1 ACC=1 ACT=1 K=3" "C=256 T=8192 B=8 RES=1 ACC=1 ACT=1 K=11 D=3"; do echo "=== $cfg"; env $cfg timeout 300 python tools/pipe_dbg.py 2>&1 | grep -v "amdgpu.ids\|Runtime\|return ufunc\|print("; done > gpurun_out/r04_run4/dbg.txt 2>&1
VS_LIB=$PWD/visinger_amd/csrc/libvisinger_hip_perturb.so timeout 600 python tools/pipe_perturb.py > gpurun_out/r04_run4/perturb.txt 2>&1
cat gpurun_out/r04_run4/glds.txt; cut -c1-300 gpurun_out/r04_run4/dbg.txt | head -40; cat gpurun_out/r04_run4/perturb.txt
