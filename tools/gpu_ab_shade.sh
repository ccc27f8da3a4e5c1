cd ${GRAFT_REPO_ROOT:-/root/repo}
for round in 1 2; do for lib in build/ab/*.so; do for sc in atrium atrium-textured; do
VCT_AMD_LIB=$PWD/$lib python bench.py --scene $sc --steps 5 --warmup 2 --cpu-seconds 0 --no-sweep 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$lib','$sc','gbuffer',d['gi_pass_ms']['gbuffer_raster'],'one_call',d['gi_pass_one_call_ms'])"
done; done; done
