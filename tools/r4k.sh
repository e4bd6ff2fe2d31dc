mkdir -p gpurun_out/r4k
for n in 1024 2048 4096; do for p in 3 4; do timeout -k 10 300 python tools/exp_potrf_lookahead.py $n 3 $p > gpurun_out/r4k/ab_n${n}_p$p.txt 2>&1 || exit 1; done; done
for f in gpurun_out/r4k/ab_n*; do head -3 $f; done
