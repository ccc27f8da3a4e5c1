cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout 900 python -m pytest tests/test_gpu_pipeline.py -x -q 2>&1 | tail -15
for r in 1 2; do
for f in 1 2; do
python bench.py --frames-in-flight $f --cpu-seconds 0 --no-sweep 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('fif',d['frames_in_flight'],'value',d['value'],'ms/step',d['ms_per_step'],'kernel',d['trace_kernel_ms'])"
done; done
./voxel-cone-tracing_amd/vct_demo --voxels 256 --size 1920x1080 --frames 60 | head -3
./voxel-cone-tracing_amd/vct_demo --voxels 256 --size 1920x1080 --frames 60 --frames-in-flight 2 | head -3
