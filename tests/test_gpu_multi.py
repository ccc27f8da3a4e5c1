"""Multi-GPU path (BASELINE.json config 4) on the ONE GPU the test box has.

  * the native step (vct_comm_init / vct_frame_step: slab trace -> ncclGather called from C++) with a 1-rank
    RCCL communicator: the gathered frame equals the frame the plain single-GPU calls produce, frame after
    frame through both gather buffers;
  * slab-restricted G-buffer raster (vct_render_gbuffer_rows) == the same rows of the full raster;
  * two RANKS sharing the GPU (RCCL refuses two ranks on one device, so the gather is staged through the
    host over gloo): vct_trace_resident_rows + vct_set_frame_target across processes, gathered frame
    bit-equal to the single-context frame -- the whole bench.py N-rank flow;
  * the C++ caller: vct_demo --gpus 1 re-launches itself as a rank process and gathers through RCCL.
There is no reference counterpart (R/main.cpp:77-94 drives a single GL context).
"""
import json
import os
import subprocess
import time
import sys

import numpy as np
import pytest

import vctpkg

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def vct():
    import torch
    assert torch.cuda.is_available()
    return vctpkg.load()


def small_pipeline(vct, w=200, h=120, V=64):
    from voxel_cone_tracing_amd import scene as sc
    scene = sc.Scene(sc.ATRIUM, 0.15, 1234)
    cam = sc.default_camera(position=(-56.0, -9.0, 2.0), yaw=0.0, pitch=8.0)
    ctx = vct.Context(vct.default_config(voxel_dim=V, width=w, height=h, shadow_map_size=512))
    ctx.set_camera_position(tuple(cam.position))
    ctx.set_light_direction((0.0, 1.0, 0.25))
    ctx.upload_triangles(scene.pos, scene.material, scene.albedo)
    ctx.upload_mesh_attributes(*scene.frames(), scene.specular)
    ctx.render_shadow_map(sc.light_view_proj((0.0, 1.0, 0.25)))
    ctx.voxelize(); ctx.inject_light(); ctx.build_mips()
    return ctx, sc.camera_view_proj(cam, w, h)


def test_slab_partition_matches_survey(vct):
    rows = [vct.slab_partition(1080, 8, r) for r in range(8)]          # 135 tile rows -> 17 x 7 + 16
    assert [r1 - r0 for r0, r1, _ in rows] == [17] * 7 + [16] and all(p == 17 for _, _, p in rows)
    assert rows[0][0] == 0 and rows[-1][1] == 135
    assert all(rows[i][1] == rows[i + 1][0] for i in range(7))
    assert vct.slab_partition(16, 8, 7)[:2] == (2, 2)                  # more ranks than tile rows: empty slabs


def test_native_frame_step_one_rank_communicator(vct):
    ctx, vp = small_pipeline(vct)
    ctx.render_gbuffer(vp)
    want = ctx.trace_current()
    ctx.comm_init(vct.comm_unique_id(), 0, 1)
    assert ctx.comm_slab() == (0, 15)
    for _ in range(3):                          # both gather buffers, reuse of the first
        ctx.frame_step()
    ctx.comm_sync()
    assert np.array_equal(ctx.comm_download_frame(), want)
    with pytest.raises(vct.VctError):
        ctx.comm_init(vct.comm_unique_id(), 0, 1)       # already initialised
    ctx.comm_destroy()
    assert np.array_equal(ctx.trace_current(), want)     # the context-owned frame target is back
    with pytest.raises(vct.VctError):
        ctx.frame_step()
    ctx.close()


def test_frame_steps_of_changing_frames_gather_each_frame(vct):
    """Three different 1080p frames in a cycle through the communicator's TWO gather buffers: the gather of a step must
    wait for that step's trace (0.4 ms here), or it picks up what the buffer held two steps earlier -- another camera's
    frame.  (The test above steps the same frame three times: a gather that ran too early would copy identical bytes.
    On ONE rank the root's slab is traced in place and the read-back waits for the context's stream as well, so even a
    build whose communication stream does not wait for the trace passes here -- that wait can only be observed with a
    second GPU; this test pins the buffer cycle and the changing content.)"""
    from voxel_cone_tracing_amd import scene as sc
    w, h = 1920, 1080
    ctx, _ = small_pipeline(vct, w, h, V=128)
    cams = [sc.default_camera(position=(-56.0 + 4.0 * k, -9.0 + k, 2.0 - k), yaw=6.0 * k, pitch=8.0 - 2.0 * k) for k in range(3)]
    vps = [sc.camera_view_proj(c, w, h) for c in cams]
    want = []
    for c, vp in zip(cams, vps):
        ctx.set_camera_position(tuple(c.position)); ctx.render_gbuffer(vp)
        want.append(ctx.trace_current())
    assert not np.array_equal(want[0], want[1]) and not np.array_equal(want[1], want[2])
    ctx.comm_init(vct.comm_unique_id(), 0, 1)
    for i in range(9):
        k = i % 3
        ctx.set_camera_position(tuple(cams[k].position)); ctx.render_gbuffer(vps[k])
        ctx.frame_step()
        assert np.array_equal(ctx.comm_download_frame(), want[k]), f"step {i}"
    # ... and two steps back to back before the read-back (the second one's buffer is the other one)
    for i in range(6):
        k = i % 3
        ctx.set_camera_position(tuple(cams[k].position)); ctx.render_gbuffer(vps[k])
        ctx.frame_step()
        if i & 1:
            assert np.array_equal(ctx.comm_download_frame(), want[k]), f"pair {i}"
    ctx.comm_destroy()
    ctx.close()


def test_slab_raster_equals_rows_of_full_raster(vct):
    ctx, vp = small_pipeline(vct, w=203, h=117)          # ragged size
    ctx.render_gbuffer(vp)
    full = ctx.download_gbuffer()
    frame = ctx.trace_current()
    h, w = 117, 203
    got = np.zeros_like(full)
    parts = np.zeros_like(frame)
    for rank in range(4):
        r0, r1, _ = vct.slab_partition(h, 4, rank)
        ctx.render_gbuffer_rows(vp, r0, r1)
        g = ctx.download_gbuffer().reshape(23, h, w)
        got.reshape(23, h, w)[:, r0 * 8:min(r1 * 8, h)] = g[:, r0 * 8:min(r1 * 8, h)]
        ctx.trace_gbuffer_rows(r0, r1)
        parts[r0 * 8:min(r1 * 8, h)] = ctx.download_frame()[r0 * 8:min(r1 * 8, h)]
    assert np.array_equal(got.view(np.uint32), full.view(np.uint32))
    assert np.array_equal(parts, frame)
    ctx.close()


def test_two_ranks_sharing_the_gpu_gather_the_single_gpu_frame():
    env = dict(os.environ, VCT_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", "29611", os.path.join(ROOT, "bench.py"),
           "--gpus", "2", "--steps", "3", "--warmup", "1", "--width", "320", "--height", "180",
           "--voxel-dim", "64", "--scene-detail", "0.15", "--shadow-size", "512", "--cpu-seconds", "0", "--no-sweep"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["gathered_frame_equals_single_gpu_frame"] is True
    assert d["cone_steps_per_frame"] > 0


def test_interleaved_rows_data_path_for_emulated_ranks(vct):
    """Interleaved slabs (tile row r -> rank r % world; vct_comm_set_interleaved).  No multi-GPU box: the data path of
    `world` ranks is walked on one GPU -- (1) a strided trace puts exactly the rank's rows at their places, (2) the
    library's self-test traces every emulated rank's rows strided + packed, runs the root's de-interleave kernel and
    compares with the frame of one launch, (3) the native step with a 1-rank communicator in interleaved mode."""
    w, h = 200, 123                                 # 16 tile rows, the last one ragged
    ctx, vp = small_pipeline(vct, w=w, h=h)
    ctx.render_gbuffer(vp)
    want = ctx.trace_current()
    total = ctx.last_step_count()
    ty = (h + 7) // 8
    for world in (2, 3, 8):
        steps = 0
        for rank in range(world):
            ctx.trace_gbuffer_strided(rank, ty, world)
            steps += ctx.last_step_count()
        assert steps == total                                   # every tile row traced exactly once over the ranks
        assert ctx.selftest_interleaved(world) == 0
    ctx.comm_init(vct.comm_unique_id(), 0, 1)
    ctx.comm_set_interleaved(True)
    assert ctx.comm_slab() == (0, ty)
    for _ in range(3):
        ctx.frame_step()
    ctx.comm_sync()
    assert np.array_equal(ctx.comm_download_frame(), want)
    ctx.comm_set_interleaved(False)
    ctx.frame_step()
    ctx.comm_sync()
    assert np.array_equal(ctx.comm_download_frame(), want)
    # contiguous boundaries after interleaved mode: the last call wins (interleaved off), the frame stays right
    for starts in (None, [0, ty]):
        ctx.comm_set_interleaved(True)
        ctx.comm_set_slab_rows(starts)
        assert ctx.comm_slab() == (0, ty)
        ctx.frame_step()
        ctx.comm_sync()
        assert np.array_equal(ctx.comm_download_frame(), want)
    ctx.comm_destroy()
    ctx.close()


def test_native_bench_loop_with_forced_one_rank_group():
    """The pipelined native step loop of bench.py (what N > 1 runs) with a 1-rank RCCL communicator."""
    env = dict(os.environ, VCT_BENCH_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29613")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "4", "--warmup", "1", "--width", "320",
           "--height", "180", "--voxel-dim", "64", "--scene-detail", "0.15", "--shadow-size", "512",
           "--cpu-seconds", "0", "--no-sweep"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    assert d["gathered_frame_equals_single_gpu_frame"] is True and d["host_issue_us_per_step"] > 0


def test_native_bench_loop_interleaved_with_forced_one_rank_group():
    env = dict(os.environ, VCT_BENCH_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29614")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "4", "--warmup", "1", "--width", "320",
           "--height", "180", "--voxel-dim", "64", "--scene-detail", "0.15", "--shadow-size", "512",
           "--cpu-seconds", "0", "--no-sweep", "--slabs", "interleaved"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    assert d["gathered_frame_equals_single_gpu_frame"] is True and d["config"]["slabs"] == "interleaved"
    # the line explains itself: RCCL's own rank count, per-rank slab kernel / step / exchange figures
    mg = d["multi_gpu"]
    assert mg["rccl_nranks"] == 1 and len(mg["per_rank"]) == 1 and mg["per_rank"][0]["rccl"]["rank"] == 0
    assert mg["per_rank"][0]["slab_kernel_ms"] > 0 and mg["root_exchange_ms"] >= 0 and mg["per_rank"][0]["slab_cone_steps"] > 0


def test_cpp_caller_multi_gpu_launcher(vct):
    """vct_demo --gpus 1: the C++ caller re-launches itself as rank 0 of a 1-rank RCCL communicator; its frame
    checksum equals the plain single-GPU run's."""
    demo = os.path.join(ROOT, "voxel-cone-tracing_amd", "vct_demo")
    args = ["--scene", "procedural:cornell", "--voxels", "32", "--size", "96x64", "--shadow", "256", "--frames", "2"]
    one = subprocess.run([demo] + args, capture_output=True, text=True, timeout=300)
    multi = subprocess.run([demo] + args + ["--gpus", "1"], capture_output=True, text=True, timeout=300)
    assert one.returncode == 0, one.stdout + one.stderr
    assert multi.returncode == 0, multi.stdout + multi.stderr

    def fnv(txt):
        return [t for t in txt.split() if t.startswith("fnv1a=")][-1]
    assert "gpus=1" in multi.stdout and fnv(one.stdout) == fnv(multi.stdout)


def test_bench_native_step_prints_exactly_one_json_line():
    """bench.py through the N-rank code path (native vct_frame_step with a 1-rank RCCL communicator, gloo control
    plane): stdout carries the one JSON line of the contract and nothing else -- gloo and RCCL both announce
    themselves on stdout when their groups form."""
    import json
    import subprocess
    import sys
    env = dict(os.environ, VCT_BENCH_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29541")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1",
                          "--cpu-seconds", "0", "--no-sweep", "--voxel-dim", "64", "--width", "320", "--height", "200",
                          "--shadow-size", "512", "--scene-detail", "0.2"],
                         capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.split("\n") if ln.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["value"] > 0
    assert "native vct_frame_step" in d["config"]["parallelism"] or d["config"]["parallelism"].startswith("single GPU")


def test_bench_self_launches_its_ranks_without_a_launcher():
    """`python3 bench.py --gpus 2 ...` exactly as the driver types it for N > 1 when no torch.distributed.run wraps it:
    the parent starts the two ranks itself (fresh processes, before it touched a GPU), relays ONE JSON line and exits 0.
    Over gloo here (two ranks share the one GPU; RCCL refuses that) -- the launcher is what is under test."""
    env = dict(os.environ, VCT_BENCH_BACKEND="gloo")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
           "--width", "320", "--height", "180", "--voxel-dim", "64", "--scene-detail", "0.15", "--shadow-size", "512",
           "--cpu-seconds", "0", "--no-sweep"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.split("\n") if ln.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["gathered_frame_equals_single_gpu_frame"] is True and d["value"] > 0


def test_self_launcher_kills_all_ranks_when_one_fails():
    """A rank that fails (here: an impossible grid size) must not leave its peers waiting in a collective: the parent
    kills every rank and exits non-zero without a JSON line."""
    env = dict(os.environ, VCT_BENCH_BACKEND="gloo")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                          "--voxel-dim", "100", "--width", "64", "--height", "64", "--cpu-seconds", "0", "--no-sweep",
                          "--timeout", "300"], env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode != 0
    assert not [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert "all ranks killed" in out.stderr


def test_frame_step_leaves_the_frame_target_alone(vct):
    """vct_frame_step passes its gather buffer to the launch instead of re-pointing the context's frame target
    (ADVICE round 2): full-frame calls after a step still use the context-owned frame or the caller's target."""
    import torch
    ctx, vp = small_pipeline(vct)
    ctx.render_gbuffer(vp)
    want = ctx.trace_current()
    mine = torch.zeros((120, 200, 4), dtype=torch.float16, device="cuda")
    ctx.set_frame_target(mine.data_ptr())
    ctx.comm_init(vct.comm_unique_id(), 0, 1)
    ctx.frame_step(); ctx.frame_step()
    ctx.comm_sync()
    assert ctx.frame_device()[0] == mine.data_ptr()          # the caller's target survived the steps
    assert np.array_equal(ctx.comm_download_frame(), want)
    assert float(mine.abs().sum()) == 0.0                    # and the steps did not write through it
    got = ctx.trace_current()                                # a full-frame call on the rank context
    torch.cuda.synchronize()
    assert np.array_equal(got, want) and np.array_equal(mine.cpu().numpy().view(np.uint16), want)
    ctx.comm_destroy()
    assert ctx.frame_device()[0] == mine.data_ptr()
    ctx.set_frame_target(None)
    ctx.close()


def test_gi_pass_on_a_rank_context_equals_the_single_gpu_pass(vct):
    """vct_gi_pass after vct_comm_init: the rank's share of the moving-light frame as ONE call (slab-scissored G-buffer
    stream, vct_frame_step at the join).  With a 1-rank communicator the gathered frame is the single-GPU pass's."""
    from voxel_cone_tracing_amd import scene as sc
    ctx, vp = small_pipeline(vct)
    lights = [(0.0, 1.0, 0.25), (0.3, 1.0, -0.2)]
    want = []
    for L in lights:
        ctx.set_light_direction(L)
        ctx.gi_pass(sc.light_view_proj(L), vp)
        ctx.synchronize()
        want.append(ctx.download_frame())
    ctx.comm_init(vct.comm_unique_id(), 0, 1)
    for L, w in zip(lights, want):
        ctx.set_light_direction(L)
        ctx.gi_pass(sc.light_view_proj(L), vp)
        ctx.comm_sync()
        assert np.array_equal(ctx.comm_download_frame(), w)
    ctx.comm_destroy()
    ctx.close()


def test_row_step_histogram_and_load_aware_slabs(vct):
    """The step counters double as the per-tile-row cost histogram; boundaries cut from it give slabs whose union is
    the single-GPU frame and whose step counts are closer to equal than equal-row slabs."""
    ctx, vp = small_pipeline(vct, w=320, h=200, V=64)
    ctx.render_gbuffer(vp)
    frame = ctx.trace_current()
    total = ctx.last_step_count()
    rows = ctx.last_row_steps()
    assert rows.shape == (25,) and int(rows.sum()) == total and total > 0
    world = 4
    starts = vct.slab_partition_weighted(rows, world)
    parts = np.zeros_like(frame)
    steps = []
    for r in range(world):
        r0, r1 = int(starts[r]), int(starts[r + 1])
        ctx.trace_gbuffer_rows(r0, r1)
        steps.append(ctx.last_step_count())
        assert steps[-1] == int(rows[r0:r1].sum())
        got = ctx.last_row_steps()
        assert np.array_equal(got[r0:r1], rows[r0:r1]) and int(got.sum()) == steps[-1]
        parts[r0 * 8:min(r1 * 8, 200)] = ctx.download_frame()[r0 * 8:min(r1 * 8, 200)]
    assert np.array_equal(parts, frame)
    equal = [int(rows[a:b].sum()) for a, b in ((0, 7), (7, 14), (14, 21), (21, 25))]
    assert max(steps) <= max(equal)
    # installing boundaries on a (1-rank) communicator: validated, and the frame is still the frame
    ctx.comm_init(vct.comm_unique_id(), 0, 1)
    with pytest.raises(vct.VctError):
        ctx.comm_set_slab_rows([0, 24])                       # must end at the frame's 25 tile rows
    ctx.comm_set_slab_rows([0, 25])
    assert ctx.comm_slab() == (0, 25)
    ctx.frame_step(); ctx.frame_step()
    ctx.comm_sync()
    assert np.array_equal(ctx.comm_download_frame(), frame)
    ctx.comm_set_slab_rows(None)
    ctx.frame_step()
    assert np.array_equal(ctx.comm_download_frame(), frame)
    ctx.comm_destroy()
    ctx.close()


def test_bench_falls_back_when_the_native_communicator_cannot_form():
    """If vct_comm_init fails on any rank (here: forced), every rank drops to the Python-paced step with
    torch.distributed's gather, the frame is still the single-GPU frame, and the line says what happened -- an N-GPU
    run that cannot use the native step still produces a number instead of nothing."""
    env = dict(os.environ, VCT_BENCH_FORCE_DIST="1", VCT_BENCH_FAIL_NATIVE="1", MASTER_ADDR="127.0.0.1",
               MASTER_PORT="29617")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--width", "320",
           "--height", "180", "--voxel-dim", "64", "--scene-detail", "0.15", "--shadow-size", "512",
           "--cpu-seconds", "0", "--no-sweep"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.split("\n") if ln.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    assert d["gathered_frame_equals_single_gpu_frame"] is True and d["value"] > 0
    assert "falling back" in out.stderr


@pytest.mark.parametrize("world", [2, 3])
def test_direct_slabs_two_processes_on_one_gpu(tmp_path, world):
    """VCT_COMM_MODE=direct (experimental): every rank's trace stores its slab straight into the root's frame buffers
    (hipIpc-mapped), flags in shared host memory replace the collective, no RCCL.  Ranks as separate processes on the
    one GPU: equal slabs, uneven boundaries, interleaved rows -- every assembled frame equals the single-context one."""
    idfile, out = str(tmp_path / "id"), str(tmp_path / "out.npz")
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "direct_rank.py"), str(r), str(world), idfile, out],
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(world)]
    logs = [p.communicate(timeout=600)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(logs)[-3000:]
    z = np.load(out)
    assert bool(z["ok"])
    for k in ("equal", "uneven", "interleaved"):
        assert np.array_equal(z[k], z["alone"]), k
    assert float(z["gather_ms"]) >= 0.0


def test_direct_slabs_dead_peer_is_an_error_not_a_hang(tmp_path):
    """A rank that disappears: the root's flag wait gives up at the communicator's deadline and vct_comm_sync reports it."""
    idfile, out = str(tmp_path / "id"), str(tmp_path / "out.npz")
    env = dict(os.environ, VCT_COMM_TIMEOUT_MS="3000")
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "direct_rank.py"), str(r), "2", idfile, out, "kill"],
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=env) for r in range(2)]
    t0 = time.time()
    logs = [p.communicate(timeout=300)[0] for p in procs]
    assert procs[0].returncode == 3, logs[0][-2000:]
    assert time.time() - t0 < 120
    z = np.load(out)
    assert not bool(z["ok"]) and ("gave up" in str(z["error"]) or "did not complete" in str(z["error"]))


def test_direct_slabs_teardown_with_waits_still_queued(tmp_path):
    """The host deadline of vct_comm_sync fires while a rank's flag wait and the trace that stores into the root's mapped
    frame are still QUEUED behind long compute: vct_comm_destroy raises the abort word, drains both streams and only then
    closes the IPC mappings (round-5 advisor: a GPU memory fault otherwise); the context stays usable."""
    idfile, out = str(tmp_path / "id"), str(tmp_path / "out.npz")
    env = dict(os.environ, VCT_COMM_TIMEOUT_MS="150")
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "direct_rank.py"), str(r), "2", idfile, out, "late"],
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=env) for r in range(2)]
    logs = [p.communicate(timeout=300)[0] for p in procs]
    assert procs[1].returncode == 3, logs[1][-2000:]          # the expected error, not a fault / abort
    assert procs[0].returncode == 0, logs[0][-2000:]
    z = np.load(out)
    assert "did not complete" in str(z["err"]) or "gave up" in str(z["err"])
    assert float(z["waited"]) < 30.0 and bool(z["usable"])


def test_cpp_caller_direct_slabs_two_ranks_on_one_gpu(vct):
    """vct_demo --gpus 2 with VCT_COMM_MODE=direct and both ranks on device 0: the facade's multi-GPU sequence through the
    direct-slab mode; rank 0's frame checksum equals the single-GPU run's."""
    demo = os.path.join(ROOT, "voxel-cone-tracing_amd", "vct_demo")
    args = ["--scene", "procedural:cornell", "--voxels", "32", "--size", "96x64", "--shadow", "256", "--frames", "4"]
    one = subprocess.run([demo] + args, capture_output=True, text=True, timeout=300)
    env = dict(os.environ, VCT_COMM_MODE="direct", VCT_DEMO_SINGLE_DEVICE="1")
    two = subprocess.run([demo] + args + ["--gpus", "2"], capture_output=True, text=True, timeout=300, env=env)
    assert one.returncode == 0, one.stdout + one.stderr
    assert two.returncode == 0, two.stdout + two.stderr

    def fnv(txt):
        return [t for t in txt.split() if t.startswith("fnv1a=")][-1]
    assert "gpus=2" in two.stdout and fnv(one.stdout) == fnv(two.stdout)


@pytest.mark.parametrize("slabs", ["balanced", "interleaved"])
def test_bench_native_loop_two_ranks_direct_slabs_on_one_gpu(slabs):
    """bench.py --gpus 2 through its NATIVE N-rank step loop (vct_frame_step per frame, load-aware slabs with two
    feedback rounds or interleaved rows, the acceptance check, the self-describing multi_gpu block) with two real ranks:
    possible on a one-GPU box only in the direct-slab mode (RCCL refuses two ranks on one device).  A functional run,
    labelled as such in the line."""
    env = dict(os.environ, VCT_COMM_MODE="direct", VCT_BENCH_BACKEND="gloo")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1", "--width", "320",
           "--height", "180", "--voxel-dim", "64", "--scene-detail", "0.15", "--shadow-size", "512", "--cpu-seconds", "0",
           "--no-sweep", "--slabs", slabs]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["gathered_frame_equals_single_gpu_frame"] is True
    assert "direct slabs" in d["config"]["comm_mode"] and "FUNCTIONAL TEST" in d["config"]["parallelism"]
    mg = d["multi_gpu"]
    assert mg["rccl_nranks"] == 2 and [p["rank"] for p in mg["per_rank"]] == [0, 1]
    assert all(p["slab_kernel_ms"] > 0 and p["slab_cone_steps"] > 0 for p in mg["per_rank"])
    assert sum(p["slab_cone_steps"] for p in mg["per_rank"]) == d["cone_steps_per_frame"]
