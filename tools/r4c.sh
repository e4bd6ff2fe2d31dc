mkdir -p gpurun_out/r4c
( cd build_variants/anomaly && for v in good bad badnoipra badnosv badinl; do MAGI_HIP_LIB=$PWD/var_$v.so timeout -k 10 120 python anom2.py; done; python anom_cmp.py bad badnoipra badnosv badinl ) > gpurun_out/r4c/anomaly2.txt 2>&1
for v in stamp_pair stamp_reg stamp_fe; do echo "=== $v"; MAGI_HIP_LIB=build_variants/$v.so timeout -k 10 200 python tools/exp_sep_stamps.py 8; done > gpurun_out/r4c/stamps.txt 2>&1
MAGI_BENCH_REHEARSE=1 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --config alpha-sweep --steps 3 --warmup 0 --burnin 5 > gpurun_out/r4c/sweep2.json 2> gpurun_out/r4c/sweep2.err
timeout -k 10 300 python bench.py --gpus 1 --config alpha-sweep --steps 50 --warmup 0 --burnin 100 > gpurun_out/r4c/sweep1.json 2> gpurun_out/r4c/sweep1.err
timeout -k 10 600 python -m pytest tests/test_theta_init_gpu.py tests/test_fit_gpu.py tests/test_api_gpu.py -m gpu -x -q -s > gpurun_out/r4c/tests.log 2>&1
tail -5 gpurun_out/r4c/tests.log
