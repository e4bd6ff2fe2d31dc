mkdir -p gpurun_out/r4s
for rep in 1 2 3; do
  for v in new old; do
    if [ $v = old ]; then export MAGI_HIP_LIB=$GRAFT_REPO_ROOT/build_variants/ab/libmagi_hip_scratch.so; else unset MAGI_HIP_LIB; fi
    timeout -k 10 200 python3 bench.py --no-extra-configs --no-cpu-baseline > gpurun_out/r4s/b1_${v}_$rep.json 2> gpurun_out/r4s/err.txt || exit 1
    timeout -k 10 200 python3 bench.py --chains-per-gpu 8 --steps 200 --no-cpu-baseline --no-extra-configs > gpurun_out/r4s/b8_${v}_$rep.json 2>> gpurun_out/r4s/err.txt || exit 1
  done
done
python3 - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r4s/b*_*.json')):
    d=json.load(open(f)); r=d['roofline']
    print(f.split('/')[-1], 'value %.2f ms/step %.3f stream %.2f point %.2f standalone %.2f slot_frac_of_load_only %.4f' % (d['value'], d['ms_per_step'], r['us_per_launch'], r['us_per_launch_point'], r['standalone_us_per_launch'], r['slot_frac_of_load_only']))
PY
