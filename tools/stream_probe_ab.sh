#!/bin/bash
# ON THE GPU BOX: does probing for streams that really overlap (VCT_STREAM_PROBE, vct_capi.hip create_overlapping_stream) change
# (a) the native multi-GPU step loop with a 1-rank communicator on a frame the size of an 8-way slab, (b) vct_gi_pass?
cd ${GRAFT_REPO_ROOT:-/root/repo}
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$1', 'ms_per_step', d['ms_per_step'], 'kernel', d['trace_kernel_ms'], 'gi one call', d.get('gi_pass_one_call_ms'), 'gi sum', d.get('gi_pass_total_ms'))"; }
for r in 1 2; do
for p in 0 1; do
VCT_STREAM_PROBE=$p VCT_BENCH_FORCE_DIST=1 python bench.py --height 136 --slabs equal --cpu-seconds 0 --no-sweep --steps 200 2>/dev/null | line "slab-loop probe=$p"
VCT_STREAM_PROBE=$p python bench.py --cpu-seconds 0 --no-sweep --frames-in-flight 1 2>/dev/null | line "whole-frame probe=$p"
done; done
