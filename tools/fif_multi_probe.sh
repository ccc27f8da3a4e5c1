#!/bin/bash
# ON THE GPU BOX: two frames in flight through the native multi-GPU step loop -- (a) 1-rank RCCL communicator on a frame the
# size of an 8-way slab, (b) two real ranks on the one GPU in direct-slab mode (functional).
cd ${GRAFT_REPO_ROOT:-/root/repo}
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$1', 'ms_per_step', d['ms_per_step'], 'one stream', d['frames_in_flight']['ms_per_step_one_stream'], 'kernel', d['trace_kernel_ms'], 'gathered ok', d['gathered_frame_equals_single_gpu_frame'], d['frames_in_flight']['slot_streams_overlap'])"; }
for r in 1 2; do
for f in 1 2; do
for sl in equal interleaved; do
VCT_BENCH_FORCE_DIST=1 python bench.py --height 136 --slabs $sl --cpu-seconds 0 --no-sweep --steps 200 --frames-in-flight $f 2>/dev/null | line "1-rank rccl slab-size $sl fif=$f"
done
VCT_BENCH_FORCE_DIST=1 python bench.py --slabs balanced --cpu-seconds 0 --no-sweep --steps 50 --frames-in-flight $f 2>/dev/null | line "1-rank rccl whole frame fif=$f"
done; done
VCT_COMM_MODE=direct VCT_BENCH_BACKEND=gloo python bench.py --gpus 2 --steps 20 --warmup 2 --cpu-seconds 0 --no-sweep --slabs balanced --frames-in-flight 2 2>/dev/null | line "2 ranks direct fif=2"
VCT_COMM_MODE=direct VCT_BENCH_BACKEND=gloo python bench.py --gpus 2 --steps 20 --warmup 2 --cpu-seconds 0 --no-sweep --slabs interleaved --frames-in-flight 2 2>/dev/null | line "2 ranks direct interleaved fif=2"
