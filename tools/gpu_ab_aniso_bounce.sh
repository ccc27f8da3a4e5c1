cd ${GRAFT_REPO_ROOT:-/root/repo}
for round in 1 2; do for lib in build/ab/*.so; do
  VCT_AMD_LIB=$PWD/$lib timeout 300 python bench.py --steps 20 --warmup 3 --cpu-seconds 0 --no-sweep --anisotropic 2>&1 | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$lib','aniso trace',d['trace_kernel_ms'])"
  VCT_AMD_LIB=$PWD/$lib timeout 300 python bench.py --steps 5 --warmup 1 --cpu-seconds 0 --no-sweep --voxel-dim 512 --width 1920 --height 1080 --bounces 2 2>&1 | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$lib','bounce',d['gi_pass_ms']['bounce_and_mips'])"
done; done
