mkdir -p gpurun_out/r4n
timeout -k 10 300 python tools/exp_potrf_lookahead.py 8192 3 3 > gpurun_out/r4n/ab_p3.txt 2>&1 && \
timeout -k 10 300 python tools/exp_potrf_lookahead.py 8192 3 4 > gpurun_out/r4n/ab_p4.txt 2>&1 && \
timeout -k 10 300 python tools/exp_potrf_lookahead.py 4096 3 3 > gpurun_out/r4n/ab_n4096.txt 2>&1 && \
timeout -k 10 300 python tools/exp_potrf_lookahead.py 1024 3 3 > gpurun_out/r4n/ab_n1024.txt 2>&1 && \
timeout -k 10 300 python tools/exp_potrf_lookahead.py 2048 3 3 > gpurun_out/r4n/ab_n2048.txt 2>&1
cat gpurun_out/r4n/ab_*.txt
