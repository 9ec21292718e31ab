set -x
mkdir -p gpurun_out/r04_run2
timeout 600 python -m pytest tests/test_conv_pipe_gpu.py -x -q > gpurun_out/r04_run2/pytest_pipe.txt 2>&1; echo "rc $?" >> gpurun_out/r04_run2/pytest_pipe.txt
tail -15 gpurun_out/r04_run2/pytest_pipe.txt
timeout 600 python tools/pipe_bench.py > gpurun_out/r04_run2/pipe_bench.txt 2>&1
cat gpurun_out/r04_run2/pipe_bench.txt
