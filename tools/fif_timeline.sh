#!/bin/bash
# ON THE GPU BOX: kernel trace of the default bench line (two frames in flight) -> gpurun_out/fif_timeline.txt
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/fiftl -- python3 $R/bench.py --cpu-seconds 0 --no-sweep --steps 40 --warmup 5 > /dev/null 2>&1
f=$(find $R/gpurun_out/fiftl -name "*kernel_trace.csv" | head -1)
python3 $R/tools/fif_timeline.py $f 40 | tee $R/gpurun_out/fif_timeline.txt
python3 $R/tools/fif_timeline.py $f 40 > /dev/null
