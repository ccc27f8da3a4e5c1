cd ${GRAFT_REPO_ROOT:-/root/repo}
for round in 1 2; do for lib in build/ab/*.so; do
  VCT_AMD_LIB=$PWD/$lib timeout 300 python bench.py --steps 30 --warmup 5 --cpu-seconds 0 --no-sweep 2>&1 | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$lib','frame',d['trace_kernel_ms'])"
  VCT_AMD_LIB=$PWD/$lib timeout 300 python tools/slab_probe.py 2>&1 | grep "^8 slabs" | cut -c1-90 | sed "s|^|$lib |"
done; done
