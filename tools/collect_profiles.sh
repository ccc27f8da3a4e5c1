#!/bin/bash
# HERE (after tools/final_run.sh ran on the GPU box and gpurun merged gpurun_out/): condense a round's profile set into
# profiles/<tag>f_* and the sha-gated replay files bench.py reads (trace_traffic*.json, stage_traffic*.json, valu_model.json).
# Then run tools/bench_lines.sh on the GPU box once more: the clean bench lines replay the counters only when the sha of
# the sources matches the files just written.      Usage: tools/collect_profiles.sh r06
set -e
TAG=${1:-r06}
cd "$(dirname "$0")/.."
G=gpurun_out; P=profiles
sum() { python3 tools/summarize_prof.py "$@" > /dev/null; }
sum $G/prof_$TAG            $P/${TAG}f_final.txt          $P/trace_traffic.json               stage_traffic.json
sum $G/prof_${TAG}_c5        $P/${TAG}f_c5.txt             $P/trace_traffic_c5.json            stage_traffic_c5.json
sum $G/prof_${TAG}_noise     $P/${TAG}f_noise.txt          $P/trace_traffic_noise.json         stage_traffic_noise.json
sum $G/prof_${TAG}_noise_rec $P/${TAG}f_noise_records.txt  $P/trace_traffic_noise_records.json stage_traffic_noise_records.json
for c in c3 bistro1080 tex; do sum $G/prof_${TAG}_$c $P/${TAG}f_$c.txt; done
for c in c3 c5 bistro1080 tex noise noise_rec; do
  [ -s $G/prof_${TAG}_$c/bench.json ] && cp $G/prof_${TAG}_$c/bench.json $P/${TAG}f_${c}_profiled_bench.json
done
cp $G/${TAG}_lines/raster_binned.txt $P/${TAG}f_raster_binned.txt
cp $G/${TAG}_lines/raster_direct.txt $P/${TAG}f_raster_direct.txt
cp $G/${TAG}_lines/vox_atrium.txt $P/${TAG}f_voxelize_pmc_atrium.txt
cp $G/${TAG}_lines/vox_c5.txt $P/${TAG}f_voxelize_pmc_c5.txt
cp $G/${TAG}_pytest_gpu.txt $P/${TAG}_pytest_gpu.txt
[ -s $G/${TAG}_fuzz.txt ] && cp $G/${TAG}_fuzz.txt $P/${TAG}_fuzz.txt
python3 tools/valu_model.py --write > /dev/null
for f in bench c5_bench c3_bench bistro1080_bench tex_bench; do
  [ -s $G/${TAG}f_$f.json ] && grep "^{" $G/${TAG}f_$f.json | tail -1 > $P/${TAG}f_$f.json
done
ls -la $P | grep "${TAG}" | head -40
