mkdir -p gpurun_out/r04_run11
timeout 900 python -m pytest tests/test_train_gpu.py -x -q -k "config3_training_step_at_full_size" -s 2>&1 | tail -6 > gpurun_out/r04_run11/pytest.txt; cat gpurun_out/r04_run11/pytest.txt
bash tools/profile_round.sh r04_b_config5 --config 5 > gpurun_out/r04_run11/prof5.txt 2>&1; tail -5 gpurun_out/r04_run11/prof5.txt
bash tools/profile_round.sh r04_b_config3 --config 3 > gpurun_out/r04_run11/prof3.txt 2>&1; tail -5 gpurun_out/r04_run11/prof3.txt
