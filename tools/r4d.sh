mkdir -p gpurun_out/r4d
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r4d/gpu_tests.log 2>&1
tail -4 gpurun_out/r4d/gpu_tests.log
timeout -k 10 300 python bench.py --gpus 1 --config alpha-sweep --steps 50 --warmup 0 --burnin 100 > gpurun_out/r4d/sweep1.json 2> gpurun_out/r4d/sweep1.err
timeout -k 10 400 python bench.py > gpurun_out/r4d/bench.json 2> gpurun_out/r4d/bench.err
tail -c 400 gpurun_out/r4d/bench.json
