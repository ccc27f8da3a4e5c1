#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
for mode in separate same separate same; do
  VCT_COMM_STREAM=$mode VCT_BENCH_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29652 timeout 600 python bench.py --width 1920 --height 136 \
    --steps 200 --warmup 20 --cpu-seconds 0 --no-sweep --slabs equal 2>/dev/null | grep "^{" | \
    python -c "import sys,json; d=json.loads(sys.stdin.read()); print('comm stream $mode: ms_per_step', d['ms_per_step'], 'kernel_ms', d['trace_kernel_ms'], 'frame ok', d['gathered_frame_equals_single_gpu_frame'])"
done
