mkdir -p gpurun_out/r04_run5
export VS_LIB=$PWD/visinger_amd/csrc/libvisinger_hip_perturb.so
(for cfg in "DBG=0" "DBG=1" "DBG=3" "DBG=31"; do echo "=== $cfg"; env $cfg timeout 300 python tools/pipe_stamps.py 2>&1 | grep -v "amdgpu.ids"; done) > gpurun_out/r04_run5/stamps.txt 2>&1
cat gpurun_out/r04_run5/stamps.txt
