"""One rank of the direct-slab test (tests/test_gpu_multi.py): `python direct_rank.py <rank> <world> <idfile> <out.npz> [kill]`.
Every rank runs on device 0 (one GPU per box: RCCL refuses that, the direct mode does not use RCCL).  Rank 0 writes the
frames it assembled: equal slabs, load-aware slabs, interleaved rows -- and the frame one context traces alone."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
os.environ["VCT_COMM_MODE"] = "direct"
rank, world, idfile, out = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
die = len(sys.argv) > 5 and sys.argv[5] == "kill"      # failure path: the last rank exits before its first frame step
late = len(sys.argv) > 5 and sys.argv[5] == "late"     # teardown path: the host deadline fires while flag waits are still QUEUED

import vctpkg
vct = vctpkg.load()
from voxel_cone_tracing_amd import scene as sc

w, h, V = (1280, 720, 64) if late else (200, 123, 64)    # 16 tile rows, the last one ragged (late: launches worth queueing)
scene = sc.Scene(sc.ATRIUM, 0.15, 1234)
cam = sc.default_camera(position=(-56.0, -9.0, 2.0), yaw=0.0, pitch=8.0)
ctx = vct.Context(vct.default_config(voxel_dim=V, width=w, height=h, shadow_map_size=512, device=0))
ctx.set_camera_position(tuple(cam.position)); ctx.set_light_direction((0.0, 1.0, 0.25))
ctx.upload_triangles(scene.pos, scene.material, scene.albedo)
ctx.upload_mesh_attributes(*scene.frames(), scene.specular)
ctx.render_shadow_map(sc.light_view_proj((0.0, 1.0, 0.25)))
ctx.voxelize(); ctx.inject_light(); ctx.build_mips()
vp = sc.camera_view_proj(cam, w, h)
ctx.render_gbuffer(vp)
alone = ctx.trace_current() if rank == 0 else None
steps_alone = ctx.last_step_count() if rank == 0 else 0

if rank == 0:
    ident = vct.comm_unique_id()
    with open(idfile + ".tmp", "wb") as fh:
        fh.write(ident)
    os.replace(idfile + ".tmp", idfile)
else:
    t0 = time.time()
    while not os.path.exists(idfile):
        if time.time() - t0 > 60:
            raise SystemExit("no id file")
        time.sleep(0.01)
    ident = open(idfile, "rb").read()
ctx.comm_init(ident, rank, world)
info = ctx.comm_info()
assert info["nranks"] == world and info["rank"] == rank
if die and rank == world - 1:
    os._exit(0)
if late:
    # The root never steps.  Rank 1 queues far more compute than the communicator's deadline, then three frame steps: the
    # third one's flag wait (release[0] >= 1, which only a stepping root publishes) is still queued behind that compute
    # when vct_comm_sync's HOST deadline fires.  vct_comm_destroy must then drain the queued wait and the trace kernel
    # that stores into the root's mapped frame BEFORE it closes the IPC mappings (a GPU fault otherwise), and the context
    # must stay usable.
    done_marker = idfile + ".rank1_done"
    if rank == 0:
        t0 = time.time()
        while not os.path.exists(done_marker) and time.time() - t0 < 120:
            time.sleep(0.05)
        ctx.comm_destroy(); ctx.close()
        sys.exit(0)
    for _ in range(1500):
        ctx.trace_resident()
    for _ in range(3):
        ctx.frame_step()
    t0 = time.time()
    try:
        ctx.comm_sync()
        err = ""
    except vct.VctError as e:
        err = str(e)
    waited = time.time() - t0
    ctx.comm_destroy()                        # queued waits + peer stores drained first, then the mappings go
    again = ctx.trace_current()               # the context survives
    np.savez(out, err=np.array(err), waited=np.float64(waited), usable=np.array(bool(np.isfinite(vct.half_to_float(again)).all())))
    open(done_marker, "w").close()
    ctx.close()
    sys.exit(3 if err else 0)
frames = {}
ty = (h + 7) // 8
try:
    # 1. equal contiguous slabs, five frames through both buffers
    for _ in range(5):
        ctx.frame_step()
    ctx.comm_sync()
    if rank == 0:
        frames["equal"] = ctx.comm_download_frame()
    # 2. load-aware boundaries (deliberately uneven)
    starts = [0] + [min(ty, max(0, round(ty * (r + 1) / world) + (1 if r % 2 == 0 else -1))) for r in range(world - 1)] + [ty]
    ctx.comm_set_slab_rows(starts)
    for _ in range(3):
        ctx.frame_step()
    ctx.comm_sync()
    if rank == 0:
        frames["uneven"] = ctx.comm_download_frame()
    # 3. interleaved tile rows
    ctx.comm_set_interleaved(True)
    for _ in range(3):
        ctx.frame_step()
    ctx.comm_sync()
    if rank == 0:
        frames["interleaved"] = ctx.comm_download_frame()
        frames["gather_ms"] = np.float32(ctx.comm_last_gather_ms())
    ok = True
except vct.VctError as e:
    ok = False
    frames["error"] = np.array(str(e))
if rank == 0:
    np.savez(out, alone=alone, ok=np.array(ok), **frames)
ctx.comm_destroy()
ctx.close()
sys.exit(0 if ok else 3)
