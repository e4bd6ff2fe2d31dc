mkdir -p gpurun_out/r4q
timeout -k 10 900 python -m pytest tests/test_fullsize_gpu.py tests/test_sampler_gpu.py tests/test_distributed_gpu.py -x -q -k "config3 or family or placement or distributed or bench" > gpurun_out/r4q/tests.log 2>&1; echo "rc=$?" >> gpurun_out/r4q/tests.log
MAGI_BUILD_PROFILE=1 timeout -k 10 200 python3 tools/exp_build_profile.py 8192 > gpurun_out/r4q/build_profile_n8192.json 2> gpurun_out/r4q/build_profile.err
tail -6 gpurun_out/r4q/tests.log; tail -6 gpurun_out/r4q/build_profile_n8192.json
