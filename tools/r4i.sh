mkdir -p gpurun_out/r4i
for rep in 1 2; do
for v in 1 0; do
  MAGI_SEP_XCD_ORDER=$v timeout -k 10 200 python bench.py --chains-per-gpu 8 --steps 100 --no-cpu-baseline --no-extra-configs > gpurun_out/r4i/b8_x${v}_$rep.json 2> gpurun_out/r4i/b8_x${v}_$rep.err
  python - <<PY
import json
d=json.load(open("gpurun_out/r4i/b8_x${v}_$rep.json")); r=d["roofline"]
print("xcd_order=$v rep $rep", d["leapfrogs_per_s"], d["us_per_slot_issued"], r["us_per_launch"], r["us_per_launch_point"], r["standalone_us_per_launch"])
PY
done; done > gpurun_out/r4i/ab.txt 2>&1
cat gpurun_out/r4i/ab.txt
timeout -k 10 300 python -m pytest tests/test_sampler_gpu.py tests/test_fused_gpu.py -m gpu -x -q > gpurun_out/r4i/tests.log 2>&1; tail -2 gpurun_out/r4i/tests.log
