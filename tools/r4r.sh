mkdir -p gpurun_out/r4r
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > gpurun_out/r4r/tests.log 2>&1; echo "tests rc=$?" > gpurun_out/r4r/rc.txt
tail -3 gpurun_out/r4r/tests.log; cat gpurun_out/r4r/rc.txt
timeout -k 10 300 python3 bench.py --no-extra-configs --no-cpu-baseline > gpurun_out/r4r/bench_short.json 2> gpurun_out/r4r/bench_short.err
timeout -k 10 300 python3 bench.py --chains-per-gpu 8 --steps 200 --no-cpu-baseline --no-extra-configs > gpurun_out/r4r/bench_8.json 2> gpurun_out/r4r/bench_8.err
python3 -c "
import json
for f in ('bench_short','bench_8'):
    d=json.load(open('gpurun_out/r4r/%s.json'%f)); r=d['roofline']; print(f, d['value'], d['ms_per_step'], r['us_per_launch'], r.get('us_per_launch_point'), r['frac'])"
