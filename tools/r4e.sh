mkdir -p gpurun_out/r4e
for rep in 1 2; do
for v in cur pt_old; do
  if [ $v = cur ]; then L=magi_v2_amd/libmagi_hip.so; else L=build_variants/$v.so; fi
  MAGI_HIP_LIB=$L timeout -k 10 200 python bench.py --chains-per-gpu 8 --steps 100 --no-cpu-baseline --no-extra-configs > gpurun_out/r4e/b8_${v}_$rep.json 2> gpurun_out/r4e/b8_${v}_$rep.err
  python - <<PY
import json
d=json.load(open("gpurun_out/r4e/b8_${v}_$rep.json")); r=d["roofline"]
print("$v $rep", d["leapfrogs_per_s"], d["us_per_slot_issued"], r["us_per_launch"], r["us_per_launch_point"], r["standalone_point_us"])
PY
done; done > gpurun_out/r4e/ab.txt 2>&1
timeout -k 5 100 tools/micro/mfma_rate > gpurun_out/r4e/micro_mfma_rate.txt 2>&1
cat gpurun_out/r4e/ab.txt
