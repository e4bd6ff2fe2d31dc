mkdir -p gpurun_out/r4h
timeout -k 10 600 python -m pytest tests/test_fused_gpu.py tests/test_sampler_gpu.py tests/test_logpost_gpu.py tests/test_fullsize_gpu.py tests/test_user_drift_gpu.py -m gpu -x -q -k "not config5 and not 8192 and not inverse_properties and not remapped" > gpurun_out/r4h/tests.log 2>&1
tail -3 gpurun_out/r4h/tests.log
for rep in 1 2; do
for v in cur mf16; do
  if [ $v = cur ]; then L=magi_v2_amd/libmagi_hip.so; else L=build_variants/$v.so; fi
  MAGI_HIP_LIB=$L timeout -k 10 200 python bench.py --chains-per-gpu 8 --steps 100 --no-cpu-baseline --no-extra-configs > gpurun_out/r4h/b8_${v}_$rep.json 2> gpurun_out/r4h/b8_${v}_$rep.err
  python - <<PY
import json
d=json.load(open("gpurun_out/r4h/b8_${v}_$rep.json")); r=d["roofline"]
print("$v $rep", d["leapfrogs_per_s"], d["us_per_slot_issued"], r["us_per_launch"], r["us_per_launch_point"], r["standalone_us_per_launch"])
PY
done; done > gpurun_out/r4h/ab.txt 2>&1
cat gpurun_out/r4h/ab.txt
MAGI_HIP_LIB=build_variants/wgtrace_mf4.so timeout -k 10 200 python tools/exp_wg_trace.py 8 1024 3 > gpurun_out/r4h/wg8_mf4.txt 2>&1
MAGI_HIP_LIB=magi_v2_amd/libmagi_hip.so timeout -k 10 200 python tools/exp_mc.py 1024 3 4 8 16 > gpurun_out/r4h/mc_kernel.txt 2>&1
