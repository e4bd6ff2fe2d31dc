mkdir -p gpurun_out/r4f
timeout -k 10 600 python -m pytest tests/test_fused_gpu.py tests/test_sampler_gpu.py tests/test_logpost_gpu.py tests/test_fullsize_gpu.py -m gpu -x -q -k "not config5 and not 8192 and not inverse_properties and not remapped" > gpurun_out/r4f/tests.log 2>&1
tail -3 gpurun_out/r4f/tests.log
for rep in 1 2; do
for v in cur pt_old; do
  if [ $v = cur ]; then L=magi_v2_amd/libmagi_hip.so; else L=build_variants/$v.so; fi
  MAGI_HIP_LIB=$L timeout -k 10 200 python bench.py --chains-per-gpu 8 --steps 100 --no-cpu-baseline --no-extra-configs > gpurun_out/r4f/b8_${v}_$rep.json 2> gpurun_out/r4f/b8_${v}_$rep.err
  python - <<PY
import json
d=json.load(open("gpurun_out/r4f/b8_${v}_$rep.json")); r=d["roofline"]
print("$v $rep", d["leapfrogs_per_s"], d["us_per_slot_issued"], r["us_per_launch"], r["us_per_launch_point"], r["standalone_us_per_launch"])
PY
done; done > gpurun_out/r4f/ab.txt 2>&1
cat gpurun_out/r4f/ab.txt
MAGI_HIP_LIB=build_variants/stamp_reg2.so timeout -k 10 200 python tools/exp_sep_stamps.py 8 > gpurun_out/r4f/stamps.txt 2>&1
