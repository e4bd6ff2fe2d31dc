mkdir -p gpurun_out/r4p
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE OK')" > gpurun_out/r4p/smoke.log 2>&1 || { tail -5 gpurun_out/r4p/smoke.log; exit 1; }
timeout -k 10 600 python3 bench.py > gpurun_out/r4p/bench_n1.json 2> gpurun_out/r4p/bench_n1.err || { tail -5 gpurun_out/r4p/bench_n1.err; exit 1; }
MAGI_BUILD_PROFILE=1 timeout -k 10 200 python3 tools/exp_build_profile.py 8192 > gpurun_out/r4p/build_profile_n8192.json 2> gpurun_out/r4p/build_profile.err || exit 1
timeout -k 10 300 python3 tools/exp_potrf_lookahead.py 8192 3 > gpurun_out/r4p/potrf_lookahead_ab.txt 2>&1 || exit 1
tail -2 gpurun_out/r4p/smoke.log; cat gpurun_out/r4p/potrf_lookahead_ab.txt; python3 -c "
import json; d=json.load(open('gpurun_out/r4p/bench_n1.json')); r=d['roofline']; print(d['value'], d['speedup_vs_cpu_port'], r['frac'], r['us_per_launch'], r['n8192_potrf_frac'], r['n8192_build_s'], r['n8192_issued_over_algorithmic'], r['mc8_us_per_slot'], r['mc8_leapfrogs_per_s'])"
