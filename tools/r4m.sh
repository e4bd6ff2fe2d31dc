mkdir -p gpurun_out/r4m
timeout -k 10 900 python -m pytest tests/test_fullsize_gpu.py tests/test_build_gpu.py -x -q > gpurun_out/r4m/tests.log 2>&1
tail -5 gpurun_out/r4m/tests.log
