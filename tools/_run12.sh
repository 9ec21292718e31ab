mkdir -p gpurun_out/r04_run12
timeout 900 python -m pytest tests/test_train_gpu.py -x -q -k "config3_training_step_at_full_size" -s 2>&1 | grep -E "AssertionError|assert |config-3|passed|failed|e-0|median" | head -40 | cut -c1-400
