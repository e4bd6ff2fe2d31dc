mkdir -p gpurun_out/r4o
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > gpurun_out/r4o/tests.log 2>&1; echo "tests rc=$?" > gpurun_out/r4o/rc.txt
tail -4 gpurun_out/r4o/tests.log
