cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/gitl -- python3 $R/bench.py --cpu-seconds 0 --no-sweep --steps 5 --warmup 2 --frames-in-flight 1 --no-two-slots > /dev/null 2>&1
f=$(find $R/gpurun_out/gitl -name "*kernel_trace.csv" | head -1)
python3 $R/tools/gi_timeline.py $f
