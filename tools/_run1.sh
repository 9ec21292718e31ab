set -x
mkdir -p gpurun_out/r04_run1
python -m pytest tests -m gpu -x -q > gpurun_out/r04_run1/pytest.txt 2>&1; echo "pytest rc $?" >> gpurun_out/r04_run1/pytest.txt
python bench.py --steps 20 --warmup 5 > gpurun_out/r04_run1/bench_stdout.txt 2> gpurun_out/r04_run1/bench_err.txt; echo "bench rc $?"
cp bench_details.json gpurun_out/r04_run1/ 2>/dev/null
for K in 3 7 11; do C=128 K=$K D=1 python tools/conv_stamps.py > gpurun_out/r04_run1/stamps_c128_k$K.txt 2>&1; done
C=256 K=7 D=1 T=8192 python tools/conv_stamps.py > gpurun_out/r04_run1/stamps_c256_k7.txt 2>&1
tail -3 gpurun_out/r04_run1/pytest.txt; wc -c gpurun_out/r04_run1/bench_stdout.txt
