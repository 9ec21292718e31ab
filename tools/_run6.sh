mkdir -p gpurun_out/r04_run6
VS_LIB=$PWD/visinger_amd/csrc/libvisinger_hip_perturb.so timeout 600 python tools/pipe_perturb.py > gpurun_out/r04_run6/perturb.txt 2>&1
cat gpurun_out/r04_run6/perturb.txt
