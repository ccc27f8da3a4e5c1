#!/bin/bash
# Run ON THE GPU BOX: parity tests, then bench.py per trace variant (compact line each).
# Usage: tools/gpu_check.sh [variants...]   (default "0 1")
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
VARS=${*:-0 1}
(timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -12) > gpurun_out/pytest_gpu.log
cat gpurun_out/pytest_gpu.log
: > gpurun_out/variants.txt
for v in $VARS; do
  timeout 300 python bench.py --steps 30 --warmup 5 --cpu-seconds 0 --variant $v 2>&1 | grep "^{" | \
    python -c "import sys,json; d=json.loads(sys.stdin.read()); print('variant',d['config']['trace_variant'],'kernel_ms',d['trace_kernel_ms'],'Mcones/s',d['value'],'frac',d['roofline']['frac'],'gi',d['gi_pass_ms'])" | tee -a gpurun_out/variants.txt
done
