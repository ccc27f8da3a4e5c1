#!/usr/bin/env python3
"""Issue-cycle model of the trace kernel's march loop (no GPU needed).

    tools/valu_model.py  [--write]

Compiles csrc/vct_trace.hip to gfx950 assembly, takes the specular march loop of
k_trace_tile_split<true,1,false,false,false,true> (the whole-frame instantiation; the diffuse loop has the same body), splits it at its labels into
  head      position, three constant divisions, level-1 coordinates, anchor, coverage test
  coop      cooperative block: Morton offset, load, decode, LDS slab, 8-texel gather, interpolation
  fallback  per-lane gather
  l2head    level-2 coordinates, anchor, coverage test
  tail      level blend + front-to-back composite
and prices every VALU instruction with the issue cost MEASURED for its class on this GPU
(tools/valu_bench.hip at 8 waves per SIMD, gpurun_out/valu_bench.txt; classes it does not cover are priced as
4-cycle ops).  The segments are weighted with the measured path frequencies (profiles/r02i_trace_stats.json:
cooperative gather / empty block / per-lane = 75.5 / 10.9 / 13.7 % of the level samples).

Output: VALU instructions and issue cycles per wave-step and the mean issue cycles per instruction -- the factor
bench.py uses for `roofline.valu_pipe_busy_model` (profiles/valu_model.json, keyed by the kernel-source sha).
The segmentation relies on the block layout the compiler currently emits; the script checks the instruction
counts it finds against the per-wave-step count of the PMC profile and refuses to write on a mismatch > 8 %.
"""
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

# tools/valu_bench.hip at 8 waves per SIMD (the march runs at 7), x 0.96: the bench prints cycles for an assumed
# 2.4 GHz and its scalar-ALU lines (one instruction per 4 cycles per SIMD by construction) read 4.17
COST = {"fma": 2.42, "mul": 2.20, "add": 2.18, "logic2": 2.25, "cvt": 3.95, "floor": 3.95, "int3": 4.0, "max": 3.96,
        "cndmask": 2.4, "readlane": 4.0, "cmp": 2.4, "mov": 2.3, "mul_lo": 4.0, "other": 4.0}
SALU_CYCLES = 4.0       # s_add / s_and / s_mul / s_lshl: 4.17 "cycles @2.4 GHz" at 1, 4 and 8 waves per SIMD
P_HIT, P_ZERO, P_FALLBACK = 0.7546, 0.1087, 0.1367


def classify(op):
    if op.startswith(("v_fma", "v_fmac")): return "fma"
    if op.startswith("v_mul_f32"): return "mul"
    if op.startswith(("v_mul_lo", "v_mad")): return "mul_lo"
    if op.startswith(("v_add_f32", "v_sub_f32", "v_subrev_f32")): return "add"
    if op.startswith(("v_add_u32", "v_sub_u32", "v_subrev_u32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_lshlrev_b32",
                      "v_lshrrev_b32", "v_not")): return "logic2"
    if op.startswith("v_cvt"): return "cvt"
    if op.startswith(("v_floor", "v_fract")): return "floor"
    if op.startswith(("v_or3", "v_add3", "v_lshl_add", "v_lshl_or", "v_and_or", "v_bitop3", "v_bfe", "v_add_lshl", "v_perm")): return "int3"
    if op.startswith(("v_max", "v_min", "v_med3")): return "max"
    if op.startswith("v_cndmask"): return "cndmask"
    if op.startswith(("v_readlane", "v_readfirstlane")): return "readlane"
    if op.startswith("v_cmp"): return "cmp"
    if op.startswith("v_mov"): return "mov"
    return "other"


def price(lines):
    ops = [ln.split()[0] for ln in lines if re.match(r"\s+v_", ln)]
    return len(ops), sum(COST[classify(o)] for o in ops)


def main():
    src = os.path.join(ROOT, "voxel-cone-tracing_amd", "csrc", "vct_trace.hip")
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "t.s")
        subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off",
                        "-fno-slp-vectorize", "-fPIC", "-Wno-unused-function", "-S", "--cuda-device-only", "-o", out, src],
                       check=True, capture_output=True)
        text = open(out).read()
    m = re.search(r"^_ZN12_GLOBAL__N_118k_trace_tile_splitILb1ELi1ELb0ELb0ELb0ELb1EEEv14VctTraceParams:.*?\.end_amdhsa_kernel", text, re.S | re.M)
    body = m.group(0).split("\n")
    # the specular march = the largest depth-1 inner loop (the diffuse march, inside the cone loop, is depth 2 and has the
    # same body; small depth-1 loops, if any, are prologue code)
    loops = [i for i, ln in enumerate(body) if "Inner Loop Header: Depth=1" in ln]

    def loop_at(start):
        end = next(i for i in range(start + 1, len(body)) if re.match(r"\.LBB\d+_\d+:\s*$", body[i]))   # first label outside the loop
        return body[start:end]
    loop = max((loop_at(i) for i in loops), key=len)
    # blocks of the loop, split at labels
    blocks, cur = [], []
    for ln in loop:
        if (re.match(r"\.LBB\d+_\d+:", ln) or re.match(r";\s*%bb\.\d+:", ln)) and cur:
            blocks.append(cur); cur = []
        cur.append(ln)
    blocks.append(cur)
    sized = [(price(b), b) for b in blocks]
    # by construction: two per-lane blocks (eight texel loads each) and two cooperative blocks (eight ds_read_b128 each), one
    # pair per level.  (Round 6: texels arrive decoded through typed-buffer loads -- ~76 and ~52 VALU; before: ~190 and ~70
    # with global_load_dword + cvt / mul / fma.)
    def texel_loads(b):
        return sum(1 for ln in b if "buffer_load_format_xyzw" in ln or re.match(r"\s+global_load_dword\s", ln))
    fb = sorted((((n, c), i) for i, ((n, c), b) in enumerate(sized) if texel_loads(b) >= 6 and n >= 40), key=lambda x: x[1])
    coop = sorted((((n, c), i) for i, ((n, c), b) in enumerate(sized) if sum("ds_read_b128" in ln for ln in b) >= 8), key=lambda x: x[1])
    big = sorted(fb + coop)
    # the march is unrolled by two (VCT_UNROLL2): two identical steps per loop body -- model the first
    assert len(fb) in (2, 4) and len(coop) == len(fb), [x[0] for x in big]
    unrolled = len(fb) == 4
    fb, coop = fb[:2], coop[:2]
    big = sorted(fb + coop)
    n_fb, c_fb = fb[0][0]
    n_coop, c_coop = coop[0][0]
    # the cooperative path starts in the block before (Morton offset, load, all-zero test): that prefix alone is the
    # cost of an empty block
    (n_zero, c_zero), _ = sized[coop[0][1] - 1]
    assert n_zero < 20, n_zero
    n_coop += n_zero; c_coop += c_zero
    # head = everything before the first big block; l2head = between the level-1 and level-2 sample code
    first = min(i for _, i in big)
    n_head, c_head = price([ln for _, b in sized[:first] for ln in b])
    l1 = sorted(i for _, i in big)[:2]
    l2 = sorted(i for _, i in big)[2:]
    mid = [ln for (_, b) in sized[max(l1) + 1:min(l2)] for ln in b]
    n_mid_all, c_mid_all = price(mid)
    rest = sized[max(l2) + 1:]
    if unrolled:        # the first step ends where the second one's head (position + coordinates, >= 30 VALU) begins
        cut = next(i for i, ((n, _), _) in enumerate(rest) if n >= 30)
        rest = rest[:cut]
    tail = [ln for (_, b) in rest for ln in b]
    n_tail_all, c_tail_all = price(tail)
    # blocks between / after the samples also hold the zero-block paths (4 v_mov each): they are priced in full,
    # which overstates l2head / tail by a few instructions
    def sample(k):
        return (P_HIT * (n_coop, c_coop)[k] + P_ZERO * (n_zero, c_zero)[k] + P_FALLBACK * (n_fb, c_fb)[k])
    n_step = n_head + n_mid_all + n_tail_all + 2 * sample(0)
    c_step = c_head + c_mid_all + c_tail_all + 2 * sample(1)
    res = {"segments": {"head": [n_head, round(c_head, 1)], "coop": [n_coop, round(c_coop, 1)],
                        "fallback": [n_fb, round(c_fb, 1)], "between_levels": [n_mid_all, round(c_mid_all, 1)],
                        "tail": [n_tail_all, round(c_tail_all, 1)]},
           "valu_per_wave_step_model": round(n_step, 1), "issue_cycles_per_wave_step_model": round(c_step, 1),
           "model_issue_cycles_per_valu_instr": round(c_step / n_step, 3),
           "path_frequencies": {"coop_gather": P_HIT, "empty_block": P_ZERO, "per_lane": P_FALLBACK},
           "cost_table_cycles": COST}
    import bench
    res["kernel_source_sha16"] = bench.kernel_source_sha()
    tt = os.path.join(ROOT, "profiles", "trace_traffic.json")
    if os.path.exists(tt):
        t = json.load(open(tt))
        stats = json.load(open(os.path.join(ROOT, "profiles", "r02i_trace_stats.json")))["launches"][0]
        measured = t["wave_instructions_per_launch"]["valu"] / stats["wave_steps"]
        res["valu_per_wave_step_pmc"] = round(measured, 1)
        res["pipe_busy_model"] = round(t["wave_instructions_per_launch"]["valu"] * res["model_issue_cycles_per_valu_instr"]
                                       / 1024.0 / t["gpu_cycles_per_launch"], 3)
        # the scalar pipe: one instruction per 4 cycles per SIMD (measured), SALU + SMEM instructions of the launch
        res["salu_issue_cycles_per_instr"] = SALU_CYCLES
        res["salu_pipe_busy_model"] = round(t["wave_instructions_per_launch"]["salu"] * SALU_CYCLES
                                            / 1024.0 / t["gpu_cycles_per_launch"], 3)
        ok = abs(measured - n_step) / measured < 0.08
    else:
        ok = False
    print(json.dumps(res, indent=1))
    if "--write" in sys.argv:
        if not ok:
            sys.exit("model and PMC instruction counts per wave-step differ by more than 8 %: not written")
        with open(os.path.join(ROOT, "profiles", "valu_model.json"), "w") as fh:
            json.dump(res, fh, indent=1)


if __name__ == "__main__":
    main()
