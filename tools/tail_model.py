"""ON THE GPU BOX: how much of the trace launch is tail, and what would a heaviest-first tile order buy?
List-scheduling model of the launch from the per-pixel per-cone step counts of the frame (debug outputs): a tile's
workgroup = 3 waves (cones 0-2, 3-5, specular), a wave's time ~ sum over its cones of the longest lane's steps, the
workgroup ends with its slowest wave; S workgroup slots; tiles start in dispatch order as slots free up.
Usage: python tools/tail_model.py [atrium|bistro] [V] [W H]"""
import heapq
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402,F401
import vctpkg  # noqa: E402

vct = vctpkg.load()
from voxel_cone_tracing_amd import scene as sc  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "atrium"
V = int(sys.argv[2]) if len(sys.argv) > 2 else 256
w = int(sys.argv[3]) if len(sys.argv) > 4 else 1920
h = int(sys.argv[4]) if len(sys.argv) > 4 else 1080
if name == "bistro":
    s = sc.Scene(sc.BISTRO, 1.0, 1234); cam = sc.default_camera(position=(-58.0, -19.0, 1.5), yaw=0.0, pitch=12.0)
else:
    s = sc.Scene(sc.ATRIUM, 1.0, 1234); cam = sc.default_camera(position=(-56.0, -9.0, 2.0), yaw=0.0, pitch=8.0)
ctx = vct.Context(vct.default_config(voxel_dim=V, width=w, height=h, shadow_map_size=4096, debug_outputs=1))
ctx.upload_scene(s)
light = (0.0, 1.0, 0.25)
ctx.set_camera_position(tuple(cam.position)); ctx.set_light_direction(light)
ctx.render_shadow_map(sc.light_view_proj(light))
ctx.voxelize(); ctx.inject_light(); ctx.build_mips()
ctx.render_gbuffer(sc.camera_view_proj(cam, w, h))
ctx.trace_resident(); ctx.synchronize()
st = ctx.steps().reshape(h, w, 7).astype(np.int64)
ty, tx = (h + 7) // 8, (w + 7) // 8
pad = np.zeros((ty * 8, tx * 8, 7), np.int64); pad[:h, :w] = st
t = pad.reshape(ty, 8, tx, 8, 7).transpose(0, 2, 1, 3, 4).reshape(ty * tx, 64, 7)
mx = t.max(1)                                                    # longest lane per cone
waves = np.stack([mx[:, 0:3].sum(1), mx[:, 3:6].sum(1), mx[:, 6]], 1)
dur = waves.max(1).astype(np.float64) + 2.0                      # + a constant for set-up / composite (in step units)
total_lane_steps = int(st.sum())
print(f"{name} V={V} {w}x{h}: {ty * tx} tiles, {total_lane_steps} steps; wave-steps per tile: mean {waves.sum(1).mean():.1f}, "
      f"workgroup duration (slowest wave) mean {dur.mean():.1f} p50 {np.median(dur):.0f} p99 {np.percentile(dur, 99):.0f} max {dur.max():.0f}")


def xcd_order(n, run=16):
    b = np.arange(n); group = 8 * run
    g = b // group; local = b - g * group
    m = g * group + (local & 7) * run + (local >> 3)
    return np.where((g + 1) * group <= n, m, b)


def makespan(order, slots):
    heap = [0.0] * slots
    heapq.heapify(heap)
    end = 0.0
    for i in order:
        t0 = heapq.heappop(heap)
        t1 = t0 + dur[i]
        end = max(end, t1)
        heapq.heappush(heap, t1)
    return end


n = ty * tx
cur = xcd_order(n)
blocks = (np.arange(n) // 16)
bcost = np.bincount(blocks, weights=dur)
border = np.argsort(-bcost, kind="stable")
heavy_blocks = np.concatenate([np.arange(b * 16, min((b + 1) * 16, n)) for b in border])
heavy_tiles = np.argsort(-dur, kind="stable")
for slots in (2304, 2048):
    ideal = dur.sum() / slots
    a, b_, c = makespan(cur, slots), makespan(heavy_blocks, slots), makespan(heavy_tiles, slots)
    print(f"slots {slots}: ideal {ideal:.1f}; current order {a:.1f} (+{(a / ideal - 1) * 100:.1f} %), heaviest 16-tile runs first {b_:.1f} "
          f"(+{(b_ / ideal - 1) * 100:.1f} %), heaviest tiles first {c:.1f} (+{(c / ideal - 1) * 100:.1f} %)")
# an 8-way slab (17 tile rows)
rows = 17
sub = np.arange(rows * tx)
for nm, od in (("current", xcd_order(len(sub))), ("heaviest tiles first", sub[np.argsort(-dur[sub], kind="stable")])):
    e = makespan(od, 2304)
    print(f"first 8-way slab ({len(sub)} tiles), slots 2304: {nm}: {e:.1f} against ideal {dur[sub].sum() / 2304:.1f}")
