mkdir -p gpurun_out/r4l
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r4l/kt -- python3 tools/exp_build_once.py 8192 2 > gpurun_out/r4l/kt.log 2>&1
f=$(ls gpurun_out/r4l/kt/*/*kernel_trace.csv | head -1)
python tools/potrf_timeline.py $f 140 > gpurun_out/r4l/timeline.txt 2>&1
head -5 $f > gpurun_out/r4l/trace_head.txt
rm -rf gpurun_out/r4l/kt
tail -3 gpurun_out/r4l/timeline.txt
