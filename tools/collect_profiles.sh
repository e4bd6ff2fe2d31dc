#!/bin/bash
# Collects the round's profiles on a GPU box (run through gpurun from the repo root):  bash tools/collect_profiles.sh r02
# Every rocprofv3 command has the program itself after `--`; counter passes are separate runs (no tracing flags next to --pmc).
set -o pipefail
R=${1:-r04}
O=gpurun_out/$R
export TMPDIR=/tmp
mkdir -p $O
# 1. the bench line (N = 1 GPU, defaults of the driver) and the config-3 per-GPU operating point (8 chains)
timeout -k 10 600 python3 bench.py > $O/bench_n1.json 2> $O/bench_n1.err || exit 1
timeout -k 10 300 python3 bench.py --chains-per-gpu 8 --steps 200 --no-cpu-baseline --no-extra-configs > $O/bench_8chains.json 2> $O/bench_8chains.err || exit 1
# 2. kernel trace of the bench on the GRAPH path (2-slot graphs keep rocprofv3's node bookkeeping small), and with the default 64-slot graphs
MAGI_GRAPH_SLOTS=2 timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_graph2 -- python3 bench.py --steps 5 --warmup 1 --burnin 30 --no-cpu-baseline --no-extra-configs --profile-slots 2 > $O/kt_graph2.log 2>&1 || exit 1
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_graph64 -- python3 bench.py --steps 2 --warmup 0 --burnin 4 --no-cpu-baseline --no-extra-configs --profile-slots 2 > $O/kt_graph64.log 2>&1; echo "graph64 rc=$?" > $O/kt_graph64.rc
# 3. memory-side traffic of k_stream: FETCH_SIZE and WRITE_SIZE in separate passes
MAGI_GRAPH_SLOTS=2 timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 bench.py --steps 1 --warmup 0 --burnin 12 --no-cpu-baseline --no-extra-configs --profile-slots 2 > $O/pmc_fetch.log 2>&1 || exit 1
MAGI_GRAPH_SLOTS=2 timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 bench.py --steps 1 --warmup 0 --burnin 12 --no-cpu-baseline --no-extra-configs --profile-slots 2 > $O/pmc_write.log 2>&1 || exit 1
MAGI_GRAPH_SLOTS=2 timeout -k 10 200 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/pmc_l2 -- python3 bench.py --steps 1 --warmup 0 --burnin 12 --no-cpu-baseline --no-extra-configs --profile-slots 2 > $O/pmc_l2.log 2>&1; echo "l2 rc=$?" >> $O/kt_graph64.rc
# 4. BASELINE config 5: N = 8192 x 4 matrix build -- kernel trace (per-class kernel names), per-class profile, GEMM counters
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_build8k -- python3 tools/exp_build_once.py 8192 3 nola > $O/kt_build8k.log 2>&1 || exit 1
MAGI_BUILD_PROFILE=1 timeout -k 10 200 python3 tools/exp_build_profile.py 8192 > $O/build_profile_n8192.json 2> $O/build_profile.err || exit 1
python3 tools/potrf_trace_summary.py $(ls $O/kt_build8k/*/*kernel_trace.csv | head -1) 8192 4 3 > $O/potrf_rank_k_by_launch.txt 2>&1
python3 tools/trtri_trace_summary.py $(ls $O/kt_build8k/*/*kernel_trace.csv | head -1) 8192 4 > $O/trtri_by_level.txt 2>&1
timeout -k 10 300 python3 tools/exp_remap_min.py 8192 10 24 > $O/gemm_remap_min_8192.txt 2>&1
timeout -k 10 300 python3 tools/exp_remap_min.py 4096 10 24 > $O/gemm_remap_min_4096.txt 2>&1
# the factorisation with look-ahead: A/B on the device clock, and who overlaps whom (kernel trace of a build with look-ahead on)
timeout -k 10 300 python3 tools/exp_potrf_lookahead.py 8192 3 > $O/potrf_lookahead_ab.txt 2>&1 || exit 1
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $O/kt_build8k_la -- python3 tools/exp_build_once.py 8192 2 > $O/kt_build8k_la.log 2>&1 || exit 1
python3 tools/potrf_timeline.py $(ls $O/kt_build8k_la/*/*kernel_trace.csv | head -1) 120 > $O/potrf_lookahead_timeline.txt 2>&1
rm -rf $O/kt_build8k_la
timeout -k 10 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES --output-format csv -d $O/pmc_gemm -- python3 tools/exp_build_once.py 8192 > $O/pmc_gemm.log 2>&1 || exit 1
# 5. micro; the matrix-core streaming kernel by chain count; the 8-chain sampler's kernel durations on the graph path
timeout -k 5 60 tools/micro/mfma_rate > $O/micro_mfma_rate.txt 2>&1
timeout -k 10 200 python3 tools/exp_mc.py 1024 3 8 16 > $O/mc_kernel.txt 2>&1
MAGI_GRAPH_SLOTS=2 timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $O/kt8_graph2 -- python3 bench.py --steps 4 --warmup 1 --burnin 30 --chains-per-gpu 8 --no-cpu-baseline --no-extra-configs --profile-slots 2 > $O/kt8_graph2.log 2>&1
# 5b. memory-side traffic at config 3's operating point (8 chains): FETCH_SIZE and WRITE_SIZE in separate passes
MAGI_GRAPH_SLOTS=2 timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc8_fetch -- python3 bench.py --steps 1 --warmup 0 --burnin 12 --chains-per-gpu 8 --no-cpu-baseline --no-extra-configs --profile-slots 2 > $O/pmc8_fetch.log 2>&1
MAGI_GRAPH_SLOTS=2 timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc8_write -- python3 bench.py --steps 1 --warmup 0 --burnin 12 --chains-per-gpu 8 --no-cpu-baseline --no-extra-configs --profile-slots 2 > $O/pmc8_write.log 2>&1
python3 tools/pmc_summary.py $O/pmc8_traffic_summary.csv $O/pmc8_fetch $O/pmc8_write > /dev/null
# 5c. config 4 across "GPUs" on this one box: the one-rank line of bench.py --config alpha-sweep
timeout -k 10 300 python3 bench.py --config alpha-sweep --steps 100 --warmup 0 --burnin 200 > $O/bench_alpha_sweep_n1.json 2> $O/bench_alpha_sweep.err
# 6. summaries on the box; the raw per-launch csv files (tens of MB) stay behind: gpurun returns at most 64 MiB
python3 tools/pmc_summary.py $O/pmc_traffic_summary.csv $O/pmc_fetch $O/pmc_write > /dev/null
python3 tools/pmc_summary.py $O/pmc_l2_summary.csv $O/pmc_l2 > /dev/null
python3 tools/pmc_summary.py $O/pmc_gemm_summary.csv $O/pmc_gemm > /dev/null
python3 tools/trace_summary.py $(ls $O/kt_graph2/*/*kernel_trace.csv | head -1) > $O/kt_graph2_summary.txt
python3 tools/trace_summary.py $(ls $O/kt8_graph2/*/*kernel_trace.csv | head -1) > $O/kt8_graph2_summary.txt
find $O -name "*kernel_trace.csv" -delete
find $O -name "*counter_collection.csv" -delete
du -sh $O
echo COLLECTED
