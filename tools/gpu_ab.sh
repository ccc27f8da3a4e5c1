#!/bin/bash
# ON THE GPU BOX: bench each build/ab/*.so (interleaved, 2 rounds) and print kernel ms.
cd ${GRAFT_REPO_ROOT:-/root/repo}
: > gpurun_out/ab.txt
for round in 1 2; do
for lib in build/ab/*.so; do
  VCT_AMD_LIB=$PWD/$lib timeout 300 python bench.py --steps 30 --warmup 5 --cpu-seconds 0 $* 2>&1 | grep "^{" | \
    python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$lib','kernel_ms',d['trace_kernel_ms'],'Mcones/s',d['value'])" | tee -a gpurun_out/ab.txt
done; done
