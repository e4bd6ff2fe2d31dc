mkdir -p gpurun_out/r4b
( cd build_variants/anomaly && for v in good bad badO2 badO1 badzero badpat goodpat; do MAGI_HIP_LIB=$PWD/var_$v.so timeout -k 10 120 python anom.py; done ) > gpurun_out/r4b/anomaly.txt 2>&1
for v in wgtrace st1a st2a st1b st2b st3b; do echo "=== $v"; MAGI_HIP_LIB=build_variants/$v.so timeout -k 10 200 python tools/exp_wg_trace.py 8 1024 3 2>&1 | grep -v "^    FH\|^    FK\|^    FE\|last to end\|end histogram"; done > gpurun_out/r4b/stagger.txt 2>&1
timeout -k 10 400 python tools/exp_recovery_n1024.py > gpurun_out/r4b/recovery_n1024.json 2> gpurun_out/r4b/recovery_n1024.err
timeout -k 10 600 python -m pytest tests/test_sampler_gpu.py tests/test_fullsize_gpu.py tests/test_distributed_gpu.py -m gpu -x -q -k "checkpoint or n4096 or alpha_sweep or contract or slot_budget or even_and_odd or remap or inverse_properties" > gpurun_out/r4b/tests.log 2>&1
tail -5 gpurun_out/r4b/tests.log
