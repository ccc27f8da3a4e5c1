#!/bin/bash
# ON THE GPU BOX: multi-GPU logic that one GPU can run -- tests, the slab probe (equal / balanced / interleaved / feedback),
# and the gather-starvation probe: the native 1-rank step loop on a frame the size of an 8-way slab (1920 x 136), with and
# without compute units reserved for the communication stream
cd ${GRAFT_REPO_ROOT:-/root/repo}
OUT=gpurun_out/r04_multi; mkdir -p $OUT
(timeout 1800 python -m pytest tests/test_gpu_multi.py -m gpu -x -q 2>&1 | grep -E "passed|failed|error|Error|assert" | tail -8) | tee $OUT/pytest.log
timeout 900 python tools/slab_probe.py 2>/dev/null | tee $OUT/slab_probe.txt
for k in 0 8 0 8 16; do
  VCT_COMM_RESERVED_CUS=$k VCT_BENCH_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29650 timeout 600 python bench.py --width 1920 --height 136 \
    --steps 200 --warmup 20 --cpu-seconds 0 --no-sweep --slabs equal 2>/dev/null | grep "^{" | \
    python -c "import sys,json; d=json.loads(sys.stdin.read()); print('reserved_cus', d['config']['comm_reserved_cus'], 'ms_per_step', d['ms_per_step'], 'kernel_ms', d['trace_kernel_ms'], 'issue_us', d['host_issue_us_per_step'])" | tee -a $OUT/reserved_probe.txt
done
