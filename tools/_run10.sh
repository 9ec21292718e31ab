mkdir -p gpurun_out/r04_run10
timeout 900 python -m pytest tests/test_train_gpu.py -x -q -k "config3_training_step_at_full_size" -s 2>&1 | tail -15 > gpurun_out/r04_run10/pytest.txt; cat gpurun_out/r04_run10/pytest.txt
timeout 600 python tools/train_launch_count.py > gpurun_out/r04_run10/launch_count.txt 2>&1; head -90 gpurun_out/r04_run10/launch_count.txt | cut -c1-180
