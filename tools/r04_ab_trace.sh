#!/bin/bash
# ON THE GPU BOX: interleaved A/B of build/ab/*.so on the trace kernel: atrium (configs[1]) and the street at 1024^3 / 4K
# (configs[4]); parity of every variant first (trace + golden tests).  Usage: tools/r04_ab_trace.sh
cd ${GRAFT_REPO_ROOT:-/root/repo}
for lib in build/ab/*.so; do
  echo "== $lib parity: $(VCT_AMD_LIB=$PWD/$lib timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_golden.py -m gpu -x -q -k 'trace or golden or bounce or aniso' 2>&1 | grep -E 'passed|failed' | tail -1)"
done
for round in 1 2; do for lib in build/ab/*.so; do
  for args in "--scene atrium" "--scene bistro --voxel-dim 1024 --width 3840 --height 2160"; do
    VCT_AMD_LIB=$PWD/$lib timeout 300 python bench.py $args --steps 20 --warmup 5 --cpu-seconds 0 --no-sweep 2>/dev/null | grep "^{" | \
      python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$lib', '$args'.split()[1], 'kernel_ms', d['trace_kernel_ms'])"
  done
done; done
