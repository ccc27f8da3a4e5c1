#!/usr/bin/env python3
"""bench.py -- Mcones/s and ms per GI pass of the voxel-cone-tracing hot path on MI355X.

Workload (BASELINE.json configs[1]): Sponza-class scene (procedural atrium, ~257k triangles,
seed 1234 -- "synthetic": the reference ships no assets), 256^3 voxel grid, 1920x1080 frame,
6 diffuse + 1 specular cone per pixel.

A step = one trace of the whole frame over the G-buffer and the brick mip chain already resident in
HBM (what the reference does per frame: Render(), VCT.h:146-190; its voxelization runs once at
init, VCT.h:138-139).  The once-per-scene GPU stages (voxelize, inject/resolve, mip build) are
timed in the same run and reported in `gi_pass_ms`.

N > 1: the frame's 8-pixel tile rows are split into N slabs, each rank traces its slab against its own
replica of the volume, and ONE gather (RCCL) assembles the RGBA16F frame on rank 0 -- strong scaling of the
same frame.  Either launched by torch.distributed.run (RANK / WORLD_SIZE in the environment), or plainly as
`python3 bench.py --gpus N`: the parent then starts the N rank processes itself -- before it has imported
torch or loaded the HIP library, so nothing that touched a GPU is ever re-executed -- relays rank 0's JSON
line, and kills every rank and exits non-zero if one fails or the run exceeds its deadline.

Prints one JSON line (rank 0).
"""
import argparse
import ctypes
import json
import os
import signal
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "tests")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
PREROLL_STEPS = 64
VALU_PEAK_GINSTR = 1228.8      # 256 CUs x 4 SIMDs x 2.4 GHz / 2 cycles per wave64 VALU instruction (same guide:
                               # "v_fma_f32 (wave64) 2 cyc"); conversions / 3-operand integer ops take 4
USEFUL_FMA_PER_64_STEPS = 72   # 2 levels x 8 texels x 4 channels + 8 (level blend, composite): the algorithm's own FMAs
BYTES_PER_STEP = 64            # 2 levels x 8 texels x 4 B
BYTES_PER_PIXEL = 100          # 92 B G-buffer in + 8 B RGBA16F out


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--voxel-dim", type=int, default=256)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--scene", default="atrium", choices=["atrium", "atrium-textured", "bistro", "cornell", "noise"],
                    help="atrium-textured: the same atrium with procedural diffuse / specular / height maps "
                         "(bump-mapped normals decohere the specular cones); bistro: Bistro-exterior-class street "
                         "(BASELINE.json configs[4]: 2.8 M triangles at --scene-detail 1, 43 %% alpha-tested foliage "
                         "cards, clutter, every surface textured with noisy height maps)")
    ap.add_argument("--noise-dense", action="store_true",
                    help="--scene noise: every voxel gets random RGBA bytes (no empty space) -- with --gbuffer random "
                         "at 1024^3 this is the HBM-bound stress: per-lane gathers over a 4.6 GiB chain")
    ap.add_argument("--gbuffer", default="coherent", choices=["coherent", "random"],
                    help="--scene noise: screen-coherent floor patch, or independent random pixels (incoherent cones)")
    ap.add_argument("--obj", default=None, metavar="PATH",
                    help="Wavefront .obj (+ .mtl) to use instead of a procedural scene, e.g. the real Sponza; model "
                         "units like the reference's (world = 0.05 * model, VCT.h:183)")
    ap.add_argument("--cam", type=float, nargs=5, default=None, metavar=("X", "Y", "Z", "YAW", "PITCH"),
                    help="camera position (world units) and yaw/pitch in degrees (default: the scene's preset)")
    ap.add_argument("--scene-detail", type=float, default=1.0,
                    help="tessellation scale of the procedural atrium (1.0 = 257k triangles, ~3.3 = Bistro-class 2.8M)")
    ap.add_argument("--shadow-size", type=int, default=4096)
    ap.add_argument("--variant", type=int, default=0)
    ap.add_argument("--bounces", type=int, default=1, choices=[1, 2],
                    help="2 = re-inject lit voxels once before the screen trace (BASELINE config 3)")
    ap.add_argument("--anisotropic", action="store_true",
                    help="six directional mip chains, direction-weighted sampling (north-star option)")
    ap.add_argument("--no-sweep", action="store_true",
                    help="skip the 3-aperture roughness sweep (profiling runs: keeps every trace launch identical)")
    ap.add_argument("--footprint-records", action="store_true",
                    help="vct_set_footprint_records(1): 32-byte footprint records beside the chain (for volumes that do "
                         "not fit the caches; include/vct.h)")
    ap.add_argument("--no-hbm-stress", action="store_true",
                    help="skip the HBM-bound stress line (dense random 1024^3 chain, random G-buffer) of the default run")
    ap.add_argument("--cpu-seconds", type=float, default=12.0,
                    help="target CPU time of the oracle baseline sample (0 disables)")
    ap.add_argument("--slabs", default="balanced", choices=["balanced", "equal", "interleaved"],
                    help="N > 1: balanced = slab boundaries of equal cone-step cost from the first frame's per-row step "
                         "histogram (vct_slab_partition_weighted), equal = ceil(tile_rows / N) rows per rank")
    ap.add_argument("--frames-in-flight", type=int, default=int(os.environ.get("VCT_BENCH_FRAMES_IN_FLIGHT", "1")), choices=[1, 2],
                    help="1 (default) = every step on one stream, each launch behind the one before; 2 = consecutive steps "
                         "alternate between two frame slots (vct_set_frames_in_flight: own stream, G-buffer and frame each), "
                         "so that step k + 1 starts while step k drains.  With 1 at N = 1 the two-slot form of the same K "
                         "steps is timed afterwards and reported beside (frames_in_flight.ms_per_step_two_slots)")
    ap.add_argument("--no-two-slots", action="store_true",
                    help="do not time the two-slot form of the K steps beside the one-stream steps (profiled runs: every "
                         "trace launch of the kernel trace then has the GPU to itself)")
    ap.add_argument("--timeout", type=float, default=float(os.environ.get("VCT_BENCH_TIMEOUT_S", "900")),
                    help="self-launched N > 1 run: seconds before the parent kills every rank and exits non-zero")
    return ap.parse_args()


def self_launch(args):
    """`python3 bench.py --gpus N` without a launcher: start N fresh rank processes (own sessions), relay rank 0's
    stdout, enforce a deadline.  Runs BEFORE torch or libvct_amd.so are imported: this process never touches a GPU."""
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), VCT_BENCH_SELF_LAUNCHED="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr, stderr=sys.stderr,
                                      text=True, start_new_session=True))

    def kill_all():
        for q in procs:
            if q.poll() is None:
                try:
                    os.killpg(q.pid, signal.SIGKILL)       # exactly the sessions started above
                except (ProcessLookupError, PermissionError):
                    pass
        for q in procs:
            try:
                q.wait(timeout=10)
            except subprocess.TimeoutExpired:
                pass

    deadline = time.monotonic() + args.timeout
    rc, why = 0, ""
    try:
        while True:
            codes = [q.poll() for q in procs]
            bad = [(i, c) for i, c in enumerate(codes) if c not in (None, 0)]
            if bad:
                rc, why = (bad[0][1] if bad[0][1] > 0 else 1), f"rank {bad[0][0]} exited with {bad[0][1]}"
                break
            if all(c == 0 for c in codes):
                break
            if time.monotonic() > deadline:
                rc, why = 124, f"no result within {args.timeout:.0f} s (a rank hangs?)"
                break
            time.sleep(0.05)
    finally:
        kill_all()
    out = procs[0].stdout.read() if procs[0].stdout else ""
    lines = [ln for ln in out.splitlines() if ln.startswith("{")]
    if rc == 0 and len(lines) != 1:
        rc, why = 1, f"rank 0 printed {len(lines)} JSON lines"
    if rc:
        sys.stderr.write(f"bench.py --gpus {args.gpus}: {why}; all ranks killed\n")
        sys.stderr.write(out[-2000:])
        raise SystemExit(rc)
    print(lines[0], flush=True)


def build_inputs(args, vct, sc):
    """Scene + camera.  The raster input stages (shadow map, G-buffer) run on the GPU in main();
    the noise scene has no triangles and uploads a synthetic volume + G-buffer instead."""
    w, h, V = args.width, args.height, args.voxel_dim
    if args.scene == "noise":
        import synth
        if args.noise_dense:
            # content of period 256 above 256^3 (caches key on addresses: same traffic as fully random bytes, a
            # fifth of the host time) -- the volume of hbm_stress()
            B = min(V, 256)
            vol = np.tile(np.random.default_rng(7).integers(0, 256, (B, B, B, 4), dtype=np.uint8), (V // B,) * 3 + (1,))
        else:
            vol = synth.noise_volume(V)
        planes = synth.coherent_gbuffer(w, h) if args.gbuffer == "coherent" else synth.random_gbuffer(w * h, seed=42)
        return dict(volume=vol, planes=planes, cam=(0.0, 4.0, 0.0), light=(0.0, 1.0, 0.25), scene=None,
                    label=("dense random volume" if args.noise_dense else "thin-shell noise volume") +
                          f" (seed 7) + {args.gbuffer} G-buffer")
    light = (0.0, 1.0, 0.25)                                     # VCT.h:14
    if args.obj:
        scene = sc.Scene(args.obj)
        cam = sc.default_camera()                                # VCT.h:8: (0,4,0), yaw -90
        label = f"{os.path.basename(args.obj)} ({scene.ntri} tris)"
    elif args.scene in ("atrium", "atrium-textured"):
        tex = args.scene == "atrium-textured"
        scene = sc.Scene(sc.ATRIUM_TEXTURED if tex else sc.ATRIUM, args.scene_detail, 1234)
        cam = sc.default_camera(position=(-56.0, -9.0, 2.0), yaw=0.0, pitch=8.0)
        label = (f"procedural atrium (Sponza-class, {scene.ntri} tris, seed 1234" +
                 (f", {len(scene.textures)} procedural texture maps)" if tex else ")"))
    elif args.scene == "bistro":
        scene = sc.Scene(sc.BISTRO, args.scene_detail, 1234)
        cam = sc.default_camera(position=(-58.0, -19.0, 1.5), yaw=0.0, pitch=12.0)
        label = (f"procedural Bistro-exterior-class street ({scene.ntri} tris, "
                 f"{100.0 * float((scene.material == 5).mean()):.0f} % alpha-tested foliage cards, "
                 f"{len(scene.textures)} texture maps, seed 1234)")
    else:
        scene = sc.Scene(sc.CORNELL)
        cam = sc.default_camera(position=(0.0, 0.0, 58.0), yaw=-90.0)
        label = f"procedural Cornell box ({scene.ntri} tris)"
    if args.cam:
        cam = sc.default_camera(position=tuple(args.cam[:3]), yaw=args.cam[3], pitch=args.cam[4])
    return dict(scene=scene, camera=cam, light_vp=sc.light_view_proj(light),
                view_proj=sc.camera_view_proj(cam, w, h), planes=None,
                cam=tuple(cam.position), light=light, label=label)


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return self_launch(args)
    import torch
    import torch.distributed as dist
    import vctpkg
    vct = vctpkg.load()
    from voxel_cone_tracing_amd import scene as sc
    from voxel_cone_tracing_amd import slabs

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} != WORLD_SIZE {world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the voxel-cone-tracing path has no CPU fallback")
    # Data path of N ranks: the library's native step (vct_frame_step: slab trace -> ONE ncclGather, RCCL called
    # directly from C++).  torch.distributed only carries the control plane (unique id, barrier, timing
    # reductions) over gloo.  VCT_BENCH_BACKEND=gloo: functional test of the N-rank flow on a box with fewer
    # GPUs than ranks (ranks share devices, Python-paced step, gather staged on the host) -- never a measurement.
    backend = os.environ.get("VCT_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    # VCT_BENCH_FORCE_DIST=1: run the N-rank step loop (double-buffered slabs, comm stream, RCCL gather)
    # with a 1-rank communicator -- exercises that code path on a single-GPU box.
    force_dist = world == 1 and os.environ.get("VCT_BENCH_FORCE_DIST") == "1"
    if force_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
    if world > 1 or force_dist:
        # gloo announces its connections on STDOUT ("[Gloo] Rank r is connected to ..."): the contract is ONE line on
        # stdout, so file descriptor 1 points at stderr while the process group forms
        sys.stdout.flush()
        saved_fd = os.dup(1)
        os.dup2(2, 1)
        try:
            dist.init_process_group("gloo", rank=rank, world_size=world)
            dist.barrier()
        finally:
            sys.stdout.flush()
            ctypes.CDLL(None).fflush(None)
            os.dup2(saved_fd, 1)
            os.close(saved_fd)
    # VCT_COMM_MODE=direct (experimental "direct slabs": no RCCL, the ranks' kernels store into the root's frame through
    # hipIpc mappings, flags instead of a collective) does not mind ranks sharing a device: with it the native step loop
    # runs under the gloo functional backend too -- the one way to drive N > 1 native ranks on a one-GPU box
    direct = os.environ.get("VCT_COMM_MODE", "")[:1] == "d"
    native = (world > 1 or force_dist) and (backend == "nccl" or direct)

    w, h, V = args.width, args.height, args.voxel_dim
    inp = build_inputs(args, vct, sc)
    cfg = vct.default_config(voxel_dim=V, width=w, height=h, device=local_rank,
                             trace_variant=args.variant, shadow_map_size=args.shadow_size,
                             voxel_attributes=1 if args.bounces == 2 else 0,
                             anisotropic_mips=1 if args.anisotropic else 0)
    ctx = vct.Context(cfg)
    if args.footprint_records:
        ctx.set_footprint_records(True)
    ctx.set_camera_position(inp["cam"])
    ctx.set_light_direction(inp["light"])
    ext_stream = torch.cuda.ExternalStream(ctx.stream(), device=local_rank)

    def ev():
        return torch.cuda.Event(enable_timing=True)

    # ---- once-per-scene GPU stages (timed with events on the context's stream) ----
    gi = {}
    gi_fused = None
    stage_counts = None
    with torch.cuda.stream(ext_stream):
        if inp["scene"] is not None:
            s = inp["scene"]
            ctx.upload_scene(s)         # triangles, frames, and (textured scenes) texture coordinates + maps
            # 12 passes, every stage bracketed by events; reported = median of the last 5 (the first passes warm
            # allocations and caches, bring the clocks up after the host-side scene set-up, and -- scenes with alpha-tested
            # textures -- are the six passes in which the context samples both visibility forms before it keeps one)
            passes = []
            for _ in range(12):
                ei = [ev() for _ in range(3)]
                ei[0].record(); ctx.render_shadow_map(inp["light_vp"])        # DrawDepthTexture
                ei[1].record(); ctx.render_gbuffer(inp["view_proj"])          # raster part of Render
                ei[2].record()
                e = [ev() for _ in range(4)]
                e[0].record(); ctx.voxelize()
                e[1].record(); ctx.inject_light()
                e[2].record(); ctx.build_mips()
                e[3].record()
                if args.bounces == 2:
                    e.append(ev())
                    ctx.bounce()                 # bounce kernel + mips of the bounce-1 chain
                    e[4].record()
                ctx.synchronize()
                one = {"voxelize": e[0].elapsed_time(e[1]), "inject_resolve": e[1].elapsed_time(e[2]),
                       "build_mips": e[2].elapsed_time(e[3]), "shadow_map_raster": ei[0].elapsed_time(ei[1]),
                       "gbuffer_raster": ei[1].elapsed_time(ei[2])}
                if args.bounces == 2:
                    one["bounce_and_mips"] = e[3].elapsed_time(e[4])
                passes.append(one)
            gi = {k: float(np.median([q[k] for q in passes[-5:]])) for k in passes[-1]}
            if args.bounces == 2:
                gi["bounce_cone_steps"] = float(ctx.last_step_count())
            if args.bounces != 2 and world == 1:
                # the same six stages as ONE call (vct_gi_pass: G-buffer raster on a second stream beside the
                # voxel stages) -- wall time of the whole pass, events around the calls, mean of 20 after 5 warm-ups
                ctx.set_trace_timing(False)       # a frame loop: no timing events around the trace (include/vct.h)
                for _ in range(5):
                    ctx.gi_pass(inp["light_vp"], inp["view_proj"])
                ef = [ev(), ev()]
                ef[0].record()
                for _ in range(20):
                    ctx.gi_pass(inp["light_vp"], inp["view_proj"])
                ef[1].record()
                ctx.synchronize()
                ctx.set_trace_timing(True)
                gi_fused = ef[0].elapsed_time(ef[1]) / 20.0
            stage_counts = ctx.stage_counts()
            inp["planes"] = ctx.download_gbuffer()     # host copy only for the CPU baseline / checks
        else:
            ctx.upload_volume(inp["volume"])
            e = [ev(), ev()]
            e[0].record(); ctx.build_mips(); e[1].record()
            ctx.synchronize()
            gi = {"voxelize": None, "inject_resolve": None, "build_mips": e[0].elapsed_time(e[1])}

    # ---- slab of this rank ----
    # native (default for N > 1): vct_comm_init allocates two gather buffers per rank; vct_frame_step makes the
    # kernel write this rank's slab straight into one of them (full-frame addressing, no copies) and issues
    # the ONE gather of the frame on a communication stream; the next frame's trace may fill the other buffer meanwhile
    # (measured on one GPU: the two do not actually overlap -- profiles/experiments/r04_gather_timeline.txt -- so a
    # frame costs slab trace + dispatch gap + wire time; the `multi_gpu` block of the line reports the parts).
    use_dist = world > 1 or force_dist
    native_fallback, fallback_group = None, None
    r0, r1, _per = vct.slab_partition(h, world, rank)
    y0, y1 = r0 * 8, min(r1 * 8, h)
    slab_px = max(0, y1 - y0) * w
    fgs = []
    if native:
        idt = torch.zeros(128, dtype=torch.uint8)
        if rank == 0:
            idt = torch.frombuffer(bytearray(vct.comm_unique_id()), dtype=torch.uint8).clone()
        dist.broadcast(idt, src=0)
        sys.stdout.flush()                   # RCCL prints a version banner on stdout when a communicator forms
        saved_fd = os.dup(1)
        os.dup2(2, 1)
        native_err = None
        try:
            if os.environ.get("VCT_BENCH_FAIL_NATIVE") == "1":       # tests: walk the fallback below
                raise RuntimeError("VCT_BENCH_FAIL_NATIVE=1")
            ctx.comm_init(bytes(idt.numpy().tobytes()), rank, world)
        except Exception as e:               # noqa: BLE001 -- whatever went wrong, the ranks must agree on what to do next
            native_err = f"rank {rank}: {e}"
        finally:
            ctypes.CDLL(None).fflush(None)   # the banner sits in the C library's stdout buffer
            os.dup2(saved_fd, 1)
            os.close(saved_fd)
        # all-or-nothing across the ranks: vct_comm_init is all-or-nothing per rank (a failed init leaves no
        # communicator behind); if any rank could not form the communicator, every rank drops to the Python-paced
        # step below with torch.distributed's own RCCL gather, and the line says so.  A run that cannot use the native
        # step is worth more than no run.
        errs = [None] * world
        dist.all_gather_object(errs, native_err)
        errs = [e for e in errs if e]
        if errs:
            print(f"[bench] native communicator unavailable ({errs[0]}); falling back to torch.distributed gather",
                  file=sys.stderr)
            if native_err is None:
                ctx.comm_destroy()
            native = False
            native_fallback = errs[0]
            if backend == "nccl" and world > 1:
                fallback_group = dist.new_group(backend="nccl")
    if native:
        assert ctx.comm_slab() == (r0, r1)
        if inp["scene"] is not None and world > 1:
            ctx.render_gbuffer_rows(inp["view_proj"], r0, r1)   # each rank rasterises only its slab from now on
    else:
        nbuf = 2 if use_dist else 1
        fgs = [slabs.FrameGather(h, w, world, rank, f"cuda:{local_rank}", group=fallback_group) for _ in range(nbuf)]
        bases = [f.slab.data_ptr() - y0 * w * 8 for f in fgs]
        ctx.set_frame_target(bases[0])
    if inp["scene"] is None:
        ctx.trace(inp["planes"], rows=(r0, r1))             # uploads the synthetic G-buffer
    else:                                   # G-buffer already resident (vct_render_gbuffer)
        ctx.trace_gbuffer_rows(r0, r1)
    steps_slab = ctx.last_step_count()
    slab_rows = [vct.slab_partition(h, world, r)[:2] for r in range(world)]
    if native and args.slabs == "balanced":          # (also with the forced 1-rank group: the tests walk this code on one GPU)
        # load-aware slabs: equal rows are not equal work.  The trace counted its executed steps per
        # tile row; the ranks' histograms are summed over the control plane, every rank cuts the same boundaries of
        # equal step cost from them and installs them on its communicator (uneven slabs travel as one fused
        # ncclSend / ncclRecv group, each straight to its rows of the root's frame).  Untimed set-up, like the
        # first frame it is derived from; the gathered frame is checked against the single-GPU frame below.
        def install(cost_rows):
            """Every rank cuts the same boundaries from the same all-reduced per-row cost and re-rasterises its slab."""
            starts = vct.slab_partition_weighted(np.maximum(cost_rows, 1.0).astype(np.uint64), world)
            ctx.comm_set_slab_rows(starts)
            rows = [(int(starts[r]), int(starts[r + 1])) for r in range(world)]
            a, b = rows[rank]
            assert ctx.comm_slab() == (a, b)
            if inp["scene"] is not None:
                ctx.render_gbuffer_rows(inp["view_proj"], a, b)
                ctx.trace_gbuffer_rows(a, b)
            else:
                ctx.trace(inp["planes"], rows=(a, b))
            return rows

        hist = torch.from_numpy(ctx.last_row_steps().astype(np.float64))
        dist.all_reduce(hist, op=dist.ReduceOp.SUM)
        slab_rows = install(hist.numpy())
        # two rounds of feedback: equal steps are not quite equal time (the cost of a step varies with the coherence
        # of the cones), so each rank prices its rows by the kernel time per executed step it measures on its own slab
        # (tools/slab_probe.py: slowest 8-way slab 0.1135 -> 0.107 ms on one GPU)
        for _ in range(2):
            r0, r1 = slab_rows[rank]
            for _ in range(8):
                ctx.trace_gbuffer_rows(r0, r1)
            ms = []
            for _ in range(5):
                ctx.trace_gbuffer_rows(r0, r1)
                ms.append(ctx.last_trace_ms())
            mine = ctx.last_row_steps().astype(np.float64)
            cost = torch.from_numpy(mine * (float(np.median(ms)) * 1e6 / max(float(mine.sum()), 1.0)))
            dist.all_reduce(cost, op=dist.ReduceOp.SUM)
            slab_rows = install(cost.numpy())
        r0, r1 = slab_rows[rank]
        y0, y1 = r0 * 8, min(r1 * 8, h)
        slab_px = max(0, y1 - y0) * w
        steps_slab = ctx.last_step_count()

    if native and args.slabs == "interleaved":
        # interleaved slabs: tile row r belongs to rank r % world -- every rank samples the whole frame,
        # so the slabs cost the same by construction (no histogram, no feedback rounds); one equal-count ncclGather,
        # the root de-interleaves behind it.  Every rank keeps the whole G-buffer resident.
        ty_all = (h + 7) // 8
        ctx.comm_set_interleaved(True)
        if inp["scene"] is not None:
            ctx.render_gbuffer(inp["view_proj"])
        ctx.trace_gbuffer_strided(min(rank, ty_all), ty_all, world)
        steps_slab = ctx.last_step_count()
        slab_rows = [(r, r + len(range(r, ty_all, world))) for r in range(world)]       # (first row, first row + rows)
        slab_px = sum(min(8, h - 8 * r) for r in range(rank, ty_all, world)) * w

    # Two frames in flight (N = 1): a second frame slot with its own resident G-buffer of the same view; the timed steps
    # alternate between the slots, so a step's ramp overlaps the drain of the step before it.
    # N > 1 (native step): the same per rank -- vct_frame_step traces on the selected slot's stream, so slab k + 1 starts
    # while slab k drains (a slab launch pays the same ~20 us of ramp + drain as a whole frame: a third of an 8-way slab).
    fif = args.frames_in_flight if (not use_dist or native) and args.variant != 4 else 1
    if fif == 2:
        try:
            ctx.set_frames_in_flight(2)
        except vct.VctError as e:                 # (a second G-buffer did not fit, ...): one frame at a time, and the line says so
            print(f"[bench] two frames in flight unavailable ({e}); one frame at a time", file=sys.stderr)
            fif = 1
    def setup_slot1():
        ctx.select_frame_slot(1)
        if native and args.slabs == "interleaved":
            if inp["scene"] is not None:
                ctx.render_gbuffer(inp["view_proj"])
            else:
                ctx.trace(inp["planes"])
            ty_all = (h + 7) // 8
            ctx.trace_gbuffer_strided(min(rank, ty_all), ty_all, world)
        elif inp["scene"] is None:
            ctx.trace(inp["planes"], rows=(r0, r1))
        else:
            if native and world > 1:
                ctx.render_gbuffer_rows(inp["view_proj"], r0, r1)       # this rank's slab only, like slot 0
            else:
                ctx.render_gbuffer(inp["view_proj"])
            ctx.trace_gbuffer_rows(r0, r1)
        assert ctx.last_step_count() == steps_slab
        ctx.select_frame_slot(0)

    if fif == 2:
        setup_slot1()
    comm_stream = torch.cuda.Stream(device=local_rank) if (use_dist and not native) else None
    traced = [torch.cuda.Event() for _ in range(len(fgs))]
    gathered = [torch.cuda.Event() for _ in range(len(fgs))]
    step_no = [0]

    def one_step():
        if not use_dist:
            if fif == 2:                          # frame k in slot k & 1: its own stream, G-buffer and frame
                ctx.select_frame_slot(step_no[0] & 1)
                step_no[0] += 1
            ctx.trace_resident()                  # the trace kernel, on the slot's stream
            return
        if native:
            if fif == 2:
                ctx.select_frame_slot(step_no[0] & 1)
                step_no[0] += 1
            ctx.frame_step()                      # slab trace + ONE ncclGather, issued from C++
            return
        k = step_no[0] % len(fgs)
        step_no[0] += 1
        ext_stream.wait_event(gathered[k])        # buffer k is free once its previous gather is done
        ctx.set_frame_target(bases[k])
        ctx.trace_resident()
        traced[k].record(ext_stream)
        comm_stream.wait_event(traced[k])
        with torch.cuda.stream(comm_stream):      # functional path: gather staged through the host
            fgs[k].gather()
            gathered[k].record(comm_stream)

    def fence():
        if native:
            ctx.comm_sync()
        if world > 1:
            dist.barrier()
        ctx.synchronize()                         # every frame slot's stream
        torch.cuda.synchronize()

    for ev_ in gathered:                          # "previous gather" of the first use of each buffer
        ev_.record(ext_stream)

    kernel_ms = []
    # Untimed pre-roll: the set-up above ends with ~40 ms of host work (G-buffer download for the CPU baseline, step
    # counts) during which the GPU idles and drops its clocks; the first ~25 launches after that run up to 18 % slow
    # (rocprof kernel trace: 808 -> 685 us over 20 ms).  A renderer runs frame after frame, so the steady state is what
    # is measured: PREROLL launches of the same step bring the clocks back before the W warm-up steps.
    # The timed steps are issued the way a frame loop issues them: without the two timing events that vct_last_trace_ms
    # reads around every launch (vct_set_trace_timing: they cost a launch ~7 us of dispatch gaps); the kernel time is
    # measured afterwards, in its own untimed loop, with the events back on.
    ctx.set_trace_timing(False)
    for _ in range(PREROLL_STEPS):
        one_step()
    for _ in range(args.warmup):
        one_step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        one_step()
    t_issue = time.perf_counter() - t0            # host time to ISSUE the steps (pacing overhead)
    fence()
    dt = time.perf_counter() - t0
    # per-launch device time of the trace kernel: HIP events on the context's stream, collected
    # in a second, untimed loop so the event reads do not serialise the timed region
    fence()
    # (one launch in flight at a time here, whatever --frames-in-flight: the duration of a launch that has the GPU to
    # itself -- what rocprofv3 reports for the same launches and what the roofline block prices)
    dt_one = None
    if fif == 2:
        ctx.select_frame_slot(0)
        t1 = time.perf_counter()
        for _ in range(args.steps):
            if native:
                ctx.frame_step()
            else:
                ctx.trace_resident()
        fence()
        dt_one = time.perf_counter() - t1     # the same K steps on ONE stream: what frames_in_flight buys, same run
    ctx.set_trace_timing(True)
    for _ in range(min(args.steps, 20)):
        ctx.trace_resident()
        kernel_ms.append(ctx.last_trace_ms())
    fence()
    ctx.set_trace_timing(False)
    dt_two = None
    if fif == 1 and not use_dist and args.variant != 4 and args.frames_in_flight == 1 and not args.no_two_slots:
        # ... and the other way round: the same K steps alternating between two frame slots, timed beside the headline
        try:
            ctx.set_frames_in_flight(2)
            setup_slot1()
            for k in range(PREROLL_STEPS + args.warmup):
                ctx.select_frame_slot(k & 1); ctx.trace_resident()
            fence()
            t1 = time.perf_counter()
            for k in range(args.steps):
                ctx.select_frame_slot(k & 1); ctx.trace_resident()
            fence()
            dt_two = time.perf_counter() - t1
            ctx.select_frame_slot(0)
            ctx.set_frames_in_flight(1)
        except vct.VctError as e:
            print(f"[bench] two frames in flight unavailable ({e}); not measured", file=sys.stderr)
    ctx.set_trace_timing(True)

    # N > 1 lines explain themselves: what RCCL says the communicator is, every rank's slab kernel time, step count and
    # exchange-step time (untimed reads, after the timed region)
    multi = None
    if use_dist and native:
        info = ctx.comm_info()
        ctx.frame_step()
        mine = {"rank": rank, "rccl": info, "slab_kernel_ms": round(float(np.mean(kernel_ms)), 4),
                "slab_cone_steps": int(steps_slab), "slab_tile_rows": list(slab_rows[rank]),
                "exchange_ms": round(ctx.comm_last_gather_ms(), 4)}
        per_rank = [None] * world
        if world > 1:
            dist.all_gather_object(per_rank, mine)
        else:
            per_rank = [mine]
        if rank == 0:
            multi = {"rccl_nranks": info["nranks"], "rccl_version": info["rccl_version"],
                     "per_rank": per_rank,
                     "slowest_slab_kernel_ms": max(p["slab_kernel_ms"] for p in per_rank),
                     "root_exchange_ms": per_rank[0]["exchange_ms"],
                     "note": "exchange_ms = device time of the frame's one ncclGather (or send/recv group) on that rank, "
                             "between two events on the stream it ran on; the root's includes waiting for the slowest peer"}
    red_dev = "cpu"                               # control plane over gloo
    tmax = torch.tensor([dt], dtype=torch.float64, device=red_dev)
    steps_all = torch.tensor([steps_slab], dtype=torch.float64, device=red_dev)
    kmax = torch.tensor([float(np.mean(kernel_ms))], dtype=torch.float64, device=red_dev)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(steps_all, op=dist.ReduceOp.SUM)
        dist.all_reduce(kmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())
    total_steps = int(steps_all.item())
    kernel_ms_avg = float(kmax.item())

    gather_ok = None
    if use_dist:
        # acceptance check of SURVEY.md 8e, untimed: the gathered frame is bit-identical to the frame
        # one GPU traces alone
        if native:
            ctx.frame_step()
            ctx.comm_sync()
            full = ctx.comm_download_frame() if rank == 0 else None
            ctx.comm_destroy()                    # restores the context-owned frame target
        else:
            ctx.set_frame_target(bases[0])
            ctx.trace_resident()
            ctx.synchronize()
            full = fgs[0].gather()
            torch.cuda.synchronize()
            full = full.cpu().numpy().view(np.uint16) if rank == 0 else None
            ctx.set_frame_target(None)
        if rank == 0:
            if inp["scene"] is not None:
                ctx.render_gbuffer(inp["view_proj"])          # the whole G-buffer again (ranks kept only slabs)
                alone = ctx.trace_current()
            else:
                alone = ctx.trace(inp["planes"])
            gather_ok = bool(np.array_equal(full, alone))

    if rank == 0:
        npix = w * h
        cones = npix * 7
        ms_per_step = dt / args.steps * 1e3
        value = cones / (dt / args.steps) / 1e6
        # roofline of the dominant kernel (trace) on this rank's launch
        alg_bytes = steps_slab * BYTES_PER_STEP + slab_px * BYTES_PER_PIXEL
        k_ms = float(np.mean(kernel_ms))
        alg_gbs = alg_bytes / (k_ms * 1e-3) / 1e9
        prof = pmc_profile(args, world)
        result = {
            "metric": "Mcones/s (+ ms per GI pass), Sponza-class 256^3 @1080p" if (V, w, h) == (256, 1920, 1080)
            else f"Mcones/s (+ ms per GI pass), Sponza-class {V}^3 @{w}x{h}",
            "value": round(value, 1),
            "unit": "Mcones/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "preroll_steps": PREROLL_STEPS,      # untimed, before the warm-up steps (clock ramp after the host-side set-up)
            "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic" if not args.obj else "user-supplied mesh",
            "config": {"workload": f"{inp['label']}, {V}^3 RGBA8 brick chain, {w}x{h}, 6 diffuse + 1 "
                                   f"specular cone/px, trace of a resident (GPU-rasterised) G-buffer",
                       "voxel_dim": V, "width": w, "height": h, "cones_per_pixel": 7, "bounces": args.bounces,
                       "anisotropic_mips": bool(args.anisotropic),
                       "parallelism": ("single GPU" + (", 2 frames in flight (consecutive steps on alternate frame slots)" if fif == 2 else "")) if world == 1 else
                       f"{world} screen-tile slabs ({args.slabs if native else 'equal'}) + 1 RCCL gather " +
                       ("(native vct_frame_step" + (", direct slabs: no collective)" if direct else ")") if native
                        else "(Python-paced step, torch.distributed gather)") +
                       ("" if backend == "nccl" else f" [FUNCTIONAL TEST over {backend}, not a measurement]") +
                       ("" if not native_fallback else f" [native communicator unavailable: {native_fallback}]"),
                       "slabs": args.slabs if (native and use_dist) else ("equal" if use_dist else None),
                       "trace_variant": args.variant, "footprint_records": bool(args.footprint_records),
                       "comm_mode": None if world == 1 and not force_dist else ("direct slabs (experimental)" if direct and native else "rccl"),
                       # compute units kept away from the trace for the gather's stream (VCT_COMM_RESERVED_CUS; 0: none)
                       "comm_reserved_cus": ctx.stage_counts().get("comm_reserved_cus", 0),
                       # visibility form of the G-buffer pass that was timed (chosen per context by timing both)
                       "raster_form": {0: None, 1: "direct", 2: "tile-binned"}.get(ctx.stage_counts().get("raster_form", 0)),
                       "slab_tile_rows": [b - a for a, b in slab_rows]},
            "frames_in_flight": {"n": fif, "slot_streams_overlap": ctx.frame_slot_streams_overlap() if fif == 2 else None,
                                 "ms_per_step_one_stream": None if dt_one is None else round(dt_one / args.steps * 1e3, 4),
                                 "ms_per_step_two_slots": None if dt_two is None else round(dt_two / args.steps * 1e3, 4),
                                 "note": "n = the form `ms_per_step` / `value` were timed in (1: every step on one stream; 2: consecutive "
                                         "steps alternate between two frame slots -- own stream, own resident G-buffer of the same view, "
                                         "own frame -- so a step starts while the one before it drains); the other form of the same K "
                                         "steps is timed right after it and reported here.  Since the timing events around a launch "
                                         "can be switched off (trace_timing_events) one stream leaves no gap for a second slot to hide "
                                         "on trace-only steps; the frame loop (raster + trace) keeps a small gain: DESIGN.md 3.6"},
            "trace_timing_events": "off in the timed steps and in gi_pass_one_call_ms (vct_set_trace_timing(0): the way a frame loop "
                                   "issues launches -- the two events vct_last_trace_ms reads cost a launch ~7 us of dispatch gaps); "
                                   "on for trace_kernel_ms, measured in its own loop",
            "cone_steps_per_frame": total_steps,
            "gathered_frame_equals_single_gpu_frame": gather_ok,
            "host_issue_us_per_step": round(t_issue / args.steps * 1e6, 2),
            "trace_kernel_ms": round(kernel_ms_avg, 4),
            "gi_pass_ms": {k: (None if v is None else round(v, 4)) for k, v in gi.items()}
            | {"trace": round(kernel_ms_avg, 4)},
            # one full GI pass = every stage once (moving light + moving camera): raster inputs,
            # voxelize, inject, mips, (bounce,) trace
            "gi_pass_total_ms": round(sum(v for k, v in gi.items() if v is not None and k != "bounce_cone_steps")
                                      + kernel_ms_avg, 4),
            # the same pass issued as one vct_gi_pass call (raster and voxel stages overlapped on two streams)
            "gi_pass_one_call_ms": None if gi_fused is None else round(gi_fused, 4),
            "roofline": roofline_block(prof, k_ms, steps_slab, alg_bytes, alg_gbs),
        }
        if fif == 2 and result["roofline"].get("valu_wave_instructions_per_launch"):
            # the same instruction count against the STEP time of the timed region (two launches in flight: a launch's own
            # duration says nothing there) -- what the overlapped drain / ramp adds to the pipes' utilisation
            rate = result["roofline"]["valu_wave_instructions_per_launch"] / (ms_per_step * 1e-3) / 1e9
            result["roofline"]["frac_at_step_rate"] = round(rate / VALU_PEAK_GINSTR, 4)
            result["roofline"]["frac_at_step_rate_note"] = ("VALU wave-instructions per launch / ms_per_step of the timed region "
                                                            "(2 frames in flight); `frac` = the same count / one launch's own duration")
        if multi is not None:
            result["multi_gpu"] = multi
        if stage_counts is not None:
            result["stage_roofline"] = stage_roofline(args, gi, stage_counts, k_ms, alg_bytes, w * h)
        if world == 1 and not args.no_sweep:
            # BASELINE.json config 5's "glossy cones at 3 roughness levels": the specular aperture is a
            # runtime parameter (trace.fs:218 uses 0.07 and mentions 0.105); same frame, same chain
            sweep = []
            for ts in (0.07, 0.105, 0.2):
                ctx.set_cone_apertures(0.577, ts)
                ms = []
                for _ in range(3):                    # six launches back to back, the last one timed: steady state
                    for _ in range(6):
                        ctx.trace_resident()
                    ms.append(ctx.last_trace_ms())
                sweep.append({"tan_specular": ts, "trace_kernel_ms": round(float(np.mean(ms)), 4),
                              "cone_steps": ctx.last_step_count(),
                              "Mcones_per_s": round(cones / (float(np.mean(ms)) * 1e-3) / 1e6, 1)})
            ctx.set_cone_apertures(0.577, 0.07)
            result["roughness_sweep"] = sweep
        if world == 1 and not args.no_sweep and args.variant == 0 and not args.anisotropic:
            # What bit-exactness with the oracle costs: the same kernel with a one-multiply unorm8 decode and
            # reciprocal-multiply divisions (config.trace_variant 3 -- opt-in, never the default, not `value`)
            exact = ctx.trace_current().copy()
            exact_steps = ctx.last_step_count()
            ms_e, ms_l = [], []
            # blocks exact / loose / exact / loose / ..., each six launches issued back to back of which the LAST is timed
            # (a launch on an idle GPU -- what a timed launch per call would measure -- runs 2-5 % slower than the steady
            # state `value` is made of, and not by the same factor for both variants); medians over three blocks each
            for variant, acc in ((0, ms_e), (3, ms_l)) * 3:
                ctx.set_trace_variant(variant)
                for _ in range(6):
                    ctx.trace_resident()
                acc.append(ctx.last_trace_ms())
            loose = ctx.trace_current().copy()
            loose_steps = ctx.last_step_count()
            ctx.set_trace_variant(0)
            a = vct.half_to_float(exact.reshape(-1, 4)).astype(np.float64)
            b = vct.half_to_float(loose.reshape(-1, 4)).astype(np.float64)
            result["exactness_tax"] = {
                "exact_trace_kernel_ms": round(float(np.median(ms_e)), 4),
                "loose_trace_kernel_ms": round(float(np.median(ms_l)), 4),
                "exact_over_loose": round(float(np.median(ms_e)) / float(np.median(ms_l)), 4),
                "rel_l2_loose_vs_exact_frame": float(np.sqrt(((a - b) ** 2).sum() / max((a ** 2).sum(), 1e-30))),
                "halves_that_differ": int((exact != loose).sum()),
                "cone_steps_exact": int(exact_steps), "cone_steps_loose": int(loose_steps),
                "note": "loose = config.trace_variant 3: x * RN(1/d) for the constant divisions (rounds 4-5 it also decoded UNORM8 "
                        "as c * RN(1/255), wrong in the last bit for 126 bytes; since round 6 texels arrive decoded -- exactly -- "
                        "from the texture path in both variants); opt-in, never `value`",
            }
        default_workload = (args.scene == "atrium" and (V, w, h) == (256, 1920, 1080) and not args.obj
                            and args.variant == 0 and not args.anisotropic and args.bounces == 1)
        if world == 1 and not args.no_sweep and not args.no_hbm_stress and default_workload:
            result["hbm_stress"] = hbm_stress(vct, local_rank)
        if world == 1 and args.cpu_seconds > 0:
            result["cpu_baseline"] = cpu_baseline(args, inp, ctx, vct)
        elif world > 1:
            # timed on rank 0 at N = 1 only (the host cores are shared by the N ranks here)
            result["cpu_baseline"] = {"value": None, "unit": "Mcones/s", "cores": None, "kind": "port",
                                      "sample": "not timed at N > 1: the figure is the N = 1 line's cpu_baseline "
                                                "(same workload, same host)"}
        print(json.dumps(result), flush=True)
    ctx.close()
    if world > 1 or force_dist:
        dist.barrier()
        dist.destroy_process_group()


def hbm_stress(vct, device):
    """BASELINE.json configs[4] is labelled "HBM-bound stress", but a real street touches a thin shell of its 1024^3
    grid and stays cache-resident.  This is the case that IS HBM-bound, every round, in the driver's
    own bench line: a dense random 1024^3 chain (4.57 GiB: 18x the Infinity Cache) traced from a random 1080p G-buffer --
    incoherent cones, every level sample a per-lane gather.  Texel content repeats with period 256 (a 64 MiB random
    block tiled 4 x 4 x 4: generating 4 GiB of random bytes would take the host 20 s); caches key on addresses, so the
    traffic is that of a fully random volume.  Reported: kernel time, SURVEY 8d algorithmic GB/s, and the counter
    traffic of the committed rocprofv3 pass of the same workload (profiles/trace_traffic_noise.json) when present."""
    import synth
    import torch
    V, w, h = 1024, 1920, 1080
    base = np.random.default_rng(7).integers(0, 256, (256, 256, 256, 4), dtype=np.uint8)
    vol = np.tile(base, (4, 4, 4, 1))
    planes = synth.random_gbuffer(w * h, seed=42)
    with vct.Context(vct.default_config(voxel_dim=V, width=w, height=h, device=device)) as c2:
        c2.set_camera_position((0.0, 4.0, 0.0))
        c2.set_light_direction((0.0, 1.0, 0.25))
        c2.upload_volume(vol)
        del vol
        c2.build_mips()
        c2.trace(planes)

        def timed():
            ms = []
            for _ in range(3):
                for _ in range(4):
                    c2.trace_resident()
                ms.append(c2.last_trace_ms())
            return float(np.median(ms))
        k_ms = timed()
        steps = c2.last_step_count()
        frame = c2.download_frame()
        # the same trace through footprint records (vct_set_footprint_records: one 32-byte fetch per per-lane level sample
        # of the levels >= 1; +4.9 GB beside the 4.57 GiB chain) -- the layout for volumes that do not fit the caches
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        c2.set_footprint_records(True)            # (allocates and builds once, untimed)
        c2.synchronize()
        st = torch.cuda.ExternalStream(c2.stream(), device=device)
        e0.record(st); c2.build_mips(); e1.record(st)
        c2.synchronize()
        build_ms = e0.elapsed_time(e1)
        c2.trace_resident()
        r_ms = timed()
        same = bool(np.array_equal(c2.download_frame(), frame)) and c2.last_step_count() == steps
    alg = steps * BYTES_PER_STEP + w * h * BYTES_PER_PIXEL
    out = {"workload": "dense random RGBA8 chain 1024^3 (4.57 GiB, period-256 content), random G-buffer 1920x1080",
           "trace_kernel_ms": round(k_ms, 4), "cone_steps": int(steps),
           "Mcones_per_s": round(w * h * 7 / (k_ms * 1e-3) / 1e6, 1),
           "algorithmic_bytes": int(alg), "algorithmic_GBps": round(alg / (k_ms * 1e-3) / 1e9, 1),
           "algorithmic_frac_of_8TBps": round(alg / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
           "counter_GBps": None, "counter_over_algorithmic": None,
           "footprint_records": {"trace_kernel_ms": round(r_ms, 4), "Mcones_per_s": round(w * h * 7 / (r_ms * 1e-3) / 1e6, 1),
                                 "algorithmic_GBps": round(alg / (r_ms * 1e-3) / 1e9, 1),
                                 "algorithmic_frac_of_8TBps": round(alg / (r_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                 "speedup": round(k_ms / r_ms, 3), "same_frame_and_steps": same,
                                 "mip_build_with_records_ms": round(build_ms, 3),
                                 "extra_bytes": int((vct.chain_texels(V) - V ** 3) * 32)}}
    # counter traffic of the committed rocprofv3 passes of the same workload (tools/profile_configs.sh: noise / noise_rec),
    # replayed only for the kernel sources they were recorded on; corrected as profiles/r01_fetch_write_calibration.txt
    # prescribes (2 x FETCH_SIZE + WRITE_SIZE); the x2 was re-calibrated on scattered 32-byte records, the pattern of the
    # footprint records (profiles/r05_fetch_calibration_scattered.txt: one 128-byte request per access, tallied as 64 B)
    for key, fname, ms, tgt in (("", "trace_traffic_noise.json", k_ms, out),
                                ("", "trace_traffic_noise_records.json", r_ms, out["footprint_records"])):
        path = os.path.join(ROOT, "profiles", fname)
        if not os.path.exists(path):
            continue
        with open(path) as fh:
            t = json.load(fh)
        if t.get("kernel_source_sha16") == kernel_source_sha() and t.get("hbm_bytes_per_launch"):
            tgt["counter_GBps"] = round(t["hbm_bytes_per_launch"] / (ms * 1e-3) / 1e9, 1)
            tgt["counter_frac_of_8TBps"] = round(tgt["counter_GBps"] / HBM_PEAK_GBS, 4)
            tgt["counter_over_algorithmic"] = round(t["hbm_bytes_per_launch"] / alg, 2)
            tgt["counter_source"] = t.get("source")
    return out


def kernel_source_sha():
    """sha256 over the sources the trace kernel is compiled from: ties a committed PMC profile to the
    binary that is being benchmarked (the .so itself is git-ignored and rebuilt per box)."""
    import hashlib
    h = hashlib.sha256()
    csrc = os.path.join(ROOT, "voxel-cone-tracing_amd", "csrc")
    for f in ("vct_trace.hip", "vct_internal.h", "vct_layout.h"):
        with open(os.path.join(csrc, f), "rb") as fh:
            h.update(fh.read())
    with open(os.path.join(ROOT, "Makefile"), "rb") as fh:
        h.update(fh.read())
    return h.hexdigest()[:16]


def pmc_profile(args, world):
    """Per-launch PMC figures of the trace kernel (profiles/trace_traffic.json, written by
    tools/summarize_prof.py from separate rocprofv3 --pmc passes of this same command).  Counters cannot be
    read from inside the timed process, so they are REPLAYED from that file -- and only when the file was
    recorded for the default workload with exactly the kernel sources being benchmarked (sha match)."""
    common = (args.variant == 0 and not args.obj and args.bounces == 1 and not args.anisotropic
              and args.scene_detail == 1.0)
    default = (common and args.voxel_dim == 256 and args.width == 1920 and args.height == 1080 and args.scene == "atrium")
    # BASELINE.json configs[4] (the Bistro-class street, 1024^3, 4K) has its own profile
    c5 = (common and args.voxel_dim == 1024 and args.width == 3840 and args.height == 2160 and args.scene == "bistro")
    if not (default or c5):
        return {"ok": False, "why": "no PMC profile for this configuration"}
    path = os.path.join(ROOT, "profiles", "trace_traffic.json" if default else "trace_traffic_c5.json")
    if not os.path.exists(path):
        return {"ok": False, "why": f"profiles/{os.path.basename(path)} missing"}
    with open(path) as fh:
        t = json.load(fh)
    sha = kernel_source_sha()
    if t.get("kernel_source_sha16") != sha:
        return {"ok": False, "why": f"profiles/{os.path.basename(path)} was recorded for kernel sources "
                                    f"{t.get('kernel_source_sha16')}, this build is {sha}: re-run tools/profile_gpu.sh"}
    t["ok"] = True
    if world != 1:
        # N > 1: the counters were recorded for the whole frame on one GPU.  The kernel is the same and its instruction
        # count follows the cone steps (profiles/r04f_*: 267 VALU per 64 steps on the atrium whatever the rows), so
        # rank 0's slab gets the whole-frame counts scaled by its share of the steps; HBM bytes are not scaled (the
        # chain's cold misses do not split by rows) and are left out.
        if not t.get("cone_steps_per_launch"):
            return {"ok": False, "why": "PMC profile without a step count: cannot be scaled to a slab"}
        t["scale_by_steps"] = True
        t["hbm_bytes_per_launch"] = None
    mp = os.path.join(ROOT, "profiles", "valu_model.json")      # tools/valu_model.py: issue-cycle model of the same sources
    if os.path.exists(mp):
        with open(mp) as fh:
            m = json.load(fh)
        if m.get("kernel_source_sha16") == sha:
            t["model_issue_cycles_per_valu_instr"] = m.get("model_issue_cycles_per_valu_instr")
            t["salu_issue_cycles_per_instr"] = m.get("salu_issue_cycles_per_instr")
    return t


def roofline_block(prof, kernel_ms, cone_steps, alg_bytes, alg_gbs):
    """The bound that actually binds the trace kernel is VALU issue (at every BASELINE size
    the chain's working set is L2 / Infinity-Cache resident and a cooperative 4x4x4 block fetch moves 4 B per
    lane-step where the per-texel figure of SURVEY.md 8d counts 64 B), so `frac` is the VALU issue rate
    against 1024 SIMDs x 2.4 GHz / 2 cycles -- a LOWER bound of pipe utilisation, since a third of the
    kernel's instructions are 4-cycle ops.  The HBM-side numbers are kept next to it."""
    # (round 6: with the texels decoded by the texture path the kernel issues 18 % fewer instructions and is no longer
    # bound by issue alone -- pipes 83 % busy, 38 % of the wave-cycles parked on memory; the VALU figure stays the
    # reported roofline because it is still the resource closest to its limit)
    r = {"bound": "valu_issue", "achieved": None, "peak": VALU_PEAK_GINSTR, "unit": "G wave-instr/s",
         "frac": None, "traffic": None,
         "hbm_algorithmic": {"bytes_per_launch": int(alg_bytes), "GBps": round(alg_gbs, 1),
                             "frac_of_8TBps": round(alg_gbs / HBM_PEAK_GBS, 4),
                             "definition": "SURVEY.md 8d: cone steps * 64 B + px * 100 B per launch / HIP-event "
                                           "kernel time; NOT a bound of this kernel (can exceed 1)"}}
    # implementation-independent floor: the algorithm's own arithmetic per 64 cone steps (one wave instruction
    # serves 64 lanes) = 2 levels x 8 texels x 4 channels of trilinear FMAs + 8 for the level blend and the
    # front-to-back composite; everything else the kernel issues (addresses, decode, weights, floor / cvt) is
    # overhead of THIS implementation.  useful_frac = those instructions per second / the 2-cycle issue peak.
    r["useful_fma_per_64_steps"] = USEFUL_FMA_PER_64_STEPS
    r["useful_frac"] = round(cone_steps / 64.0 * USEFUL_FMA_PER_64_STEPS / (kernel_ms * 1e-3) / 1e9 / VALU_PEAK_GINSTR, 4)
    if not prof.get("ok"):
        r["note"] = "VALU instruction count / HBM traffic not reported: " + prof["why"]
        return r
    wi = dict(prof.get("wave_instructions_per_launch") or {})
    gpu_cyc_scale = None
    if prof.get("scale_by_steps"):
        f = cone_steps / float(prof["cone_steps_per_launch"])
        wi = {k: v * f for k, v in wi.items()}
        r["scaled"] = (f"whole-frame counters x {f:.4f} = this rank's share of the frame's cone steps "
                       f"(N > 1: no per-slab PMC pass exists)")
        gpu_cyc_scale = True
    valu = wi.get("valu")
    if valu:
        rate = valu / (kernel_ms * 1e-3) / 1e9
        r["achieved"] = round(rate, 1)
        r["frac"] = round(rate / VALU_PEAK_GINSTR, 4)
        r["frac_reading"] = ("an ISSUE RATE, not a speed: round 6 took 18 % of the kernel's instructions away (the UNORM8 decode is done by "
                             "the texture path: 543 -> 447 M wave-instructions per launch), the kernel got 10 % faster (0.612 -> 0.55 ms) "
                             "and this fraction FELL from 0.72 to 0.66; see valu_pipe_busy_model, wave_cycles_waiting_on_memory and useful_frac")
        r["valu_wave_instructions_per_launch"] = int(valu)
        r["valu_wave_instructions_per_64_cone_steps"] = round(valu * 64 / max(cone_steps, 1), 1)
        # `frac` counts every instruction as a 2-cycle issue; a third of this kernel's mix are 4-cycle ops.
        # Pipe occupancy by the measured per-op issue costs (tools/valu_model.py -> profiles/valu_model.json):
        # instruction count x mean issue cycles of the march loop's mix / (1024 SIMDs x GPU cycles of a
        # launch, GRBM_GUI_ACTIVE of the PMC pass).
        cyc, gpu_cyc = prof.get("model_issue_cycles_per_valu_instr"), prof.get("gpu_cycles_per_launch")
        if gpu_cyc_scale:
            gpu_cyc = None            # the slab's launch is shorter than the profiled one: no cycle count for it
        if cyc and gpu_cyc:
            r["valu_pipe_busy_model"] = round(valu * cyc / 1024.0 / gpu_cyc, 3)
        # the scalar pipe issues one instruction per 4 cycles per SIMD (tools/valu_bench.hip); half as many scalar as
        # vector instructions made it the hidden second bound of this kernel until the anchor spread became a table
        scyc, salu = prof.get("salu_issue_cycles_per_instr"), wi.get("salu")
        if scyc and salu and gpu_cyc:
            r["salu_pipe_busy_model"] = round(salu * scyc / 1024.0 / gpu_cyc, 3)
    for k in ("wave_cycles_waiting_on_memory", "wave_cycles_waiting_to_issue"):      # SQ_WAIT_ANY / SQ_WAIT_INST_ANY over SQ_WAVE_CYCLES
        if prof.get(k) is not None and not prof.get("scale_by_steps"):
            r[k] = round(prof[k], 3)
    hbm = prof.get("hbm_bytes_per_launch")
    if hbm:
        r["traffic"] = round(hbm / (kernel_ms * 1e-3) / 1e9, 1)
        r["traffic_unit"] = "GB/s of HBM (PMC bytes per launch / this run's kernel time)"
        r["traffic_bytes_per_launch"] = int(hbm)
    r["note"] = (f"instruction and byte counts per launch replayed from {prof.get('source')} (rocprofv3 --pmc "
                 f"passes of this command, kernel sources {prof.get('kernel_source_sha16')} = this build); "
                 f"kernel time measured live with HIP events on the context stream")
    return r


def stage_roofline(args, gi, counts, trace_ms, trace_bytes, npix):
    """Per stage of one GI pass: ALGORITHMIC bytes (what the stage must read and write once, cache-served
    re-reads such as the per-pixel triangle fetch and the PCF taps not counted), measured time, and the fraction of the
    8 TB/s HBM peak that corresponds to.  Every stage but the trace streams its data once, so HBM is their roofline;
    where a stage sits far below it the `bound` field says what it waits on instead.  PMC
    FETCH_SIZE / WRITE_SIZE per kernel: profiles/stage_traffic.json (tools/summarize_prof.py), replayed when present."""
    S, V = args.shadow_size, args.voxel_dim
    ntri, cand, bricks = counts["triangles"], counts["vox_candidates"], counts["touched_bricks"]
    textured = args.scene in ("bistro", "atrium-textured") or bool(args.obj)
    upper = sum((V >> l) ** 3 for l in range(3, V.bit_length()))          # levels >= 3: dense (tiny)
    stage_bytes = {
        # triangles in, one depth word per shadow-map texel out
        "shadow_map_raster": (ntri * 36 + S * S * 4, "latency: per-triangle fp64 set-up, then dependent loads and L2 atomics of the coverage loops"),
        # triangles in, per pixel 8 B visibility word + 92 B G-buffer out
        "gbuffer_raster": (ntri * 36 + npix * 100, "L2 atomics (visibility), then ALU + dependent fetches (shade)"),
        # per fragment a 4 B list entry + 8 B stored barycentrics (+ 12 B stored albedo in a scene with textures) + its
        # triangle's 36 B (re-read per fragment, cache-served: counted once per triangle), per touched brick 2 KiB of
        # staged texels out; accumulation happens in LDS
        "voxelize": (ntri * 36 + cand * (24 if textured else 12) + bricks * 2048,
                     "the 25-tap PCF: latency of its 6x6 window fetch at 4 waves per SIMD, and ~420 of the 520-630 VALU per 64 "
                     "fragments (DESIGN.md 3.2); set-up, barycentrics and albedo are per-mesh precomputations since round 4"),
        # per voxel of a touched brick: staged texel read, level-0 texel written
        "inject_resolve": (bricks * 512 * 8, "hbm"),
        # per touched brick 2 KiB read, 1/8 + 1/64 + 1/512 of it written; dense above level 2
        "build_mips": (int(bricks * 2048 * (1 + 0.142)) + upper * 36, "launch latency (3 dependent dispatches of microseconds)"),
        "trace": (int(trace_bytes), "valu_issue (roofline block above; the SURVEY 8d byte figure is not a bound of it)"),
    }
    out = {}
    for k, (b, bound) in stage_bytes.items():
        ms = trace_ms if k == "trace" else gi.get(k)
        if not ms:
            continue
        gbs = b / (ms * 1e-3) / 1e9
        out[k] = {"algorithmic_bytes": int(b), "ms": round(ms, 4), "GBps": round(gbs, 1),
                  "frac_of_hbm_peak": round(gbs / HBM_PEAK_GBS, 4), "bound": bound}
    # per-kernel PMC bytes exist for the two profiled workloads only (tools/summarize_prof.py: the default one and configs[4])
    common = (args.variant == 0 and not args.obj and args.bounces == 1 and not args.anisotropic and args.scene_detail == 1.0)
    name = None
    if common and args.voxel_dim == 256 and args.width == 1920 and args.height == 1080 and args.scene == "atrium":
        name = "stage_traffic.json"
    elif common and args.voxel_dim == 1024 and args.width == 3840 and args.height == 2160 and args.scene == "bistro":
        name = "stage_traffic_c5.json"
    path = os.path.join(ROOT, "profiles", name or "-")
    if name and os.path.exists(path):
        with open(path) as fh:
            t = json.load(fh)
        if t.get("source_sha16") == all_sources_sha():
            out["pmc_hbm_bytes_per_dispatch"] = t.get("kernels")
            out["pmc_source"] = t.get("source")
    return out


def all_sources_sha():
    """sha256 over every kernel source + the Makefile (gate of profiles/stage_traffic.json)."""
    import glob
    import hashlib
    hs = hashlib.sha256()
    csrc = os.path.join(ROOT, "voxel-cone-tracing_amd", "csrc")
    for f in sorted(glob.glob(os.path.join(csrc, "*.hip")) + glob.glob(os.path.join(csrc, "*.h"))) + [os.path.join(ROOT, "Makefile")]:
        with open(f, "rb") as fh:
            hs.update(fh.read())
    return hs.hexdigest()[:16]


def usable_cpus():
    """Host threads this process can really keep busy: the affinity mask, capped by the cgroup CPU
    quota (the GPU box shows 256 hardware threads but grants its container 16 CPUs of time)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    quota = None
    try:
        with open("/sys/fs/cgroup/cpu.max") as fh:                      # cgroup v2
            q, per = fh.read().split()
            if q != "max":
                quota = float(q) / float(per)
    except (OSError, ValueError):
        try:                                                            # cgroup v1
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as fq, open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as fp:
                q, per = float(fq.read()), float(fp.read())
                if q > 0:
                    quota = q / per
        except (OSError, ValueError):
            pass
    if quota is not None:
        n = max(1, min(n, int(np.ceil(quota))))
    return n, quota


def cpu_baseline(args, inp, ctx, vct):
    """The scalar oracle (a port of the reference shader, oracle/vct_oracle.cpp) timed on the host
    cores on a bounded tile sample of the same frame, against the same volume the GPU traced."""
    from oracle import pyoracle
    w, h, V = args.width, args.height, args.voxel_dim
    chain = ctx.download_chain()                       # the GPU-built chain, linear layout
    aniso = ctx.download_aniso() if args.anisotropic else None
    p = pyoracle.default_params(V, camera_pos=inp["cam"], light_dir=inp["light"])
    otrace = (lambda pp, ch, pl, nthreads: pyoracle.trace_aniso(pp, ch, aniso, pl, nthreads=nthreads)) \
        if args.anisotropic else pyoracle.trace
    planes = inp["planes"]
    tiles_x, tiles_y = (w + 7) // 8, (h + 7) // 8
    cores, quota = usable_cpus()

    def sample(every):
        ys, xs = np.divmod(np.arange(w * h), w)
        tile = (ys // 8) * tiles_x + xs // 8
        return np.nonzero(tile % every == 0)[0]

    # probe: 1/256 of the tiles on one thread
    idx = sample(256)
    t = time.perf_counter()
    r = otrace(p, chain, planes[:, idx], nthreads=1)
    t1 = time.perf_counter() - t
    rate_1t = len(idx) * 7 / t1 / 1e6
    # all-thread sample sized for ~cpu_seconds
    est_full = (w * h) / len(idx) * t1 / cores
    every = int(min(256, max(1, 2 ** int(np.ceil(np.log2(max(est_full / args.cpu_seconds, 1.0)))))))
    idx = sample(every)
    sub = np.ascontiguousarray(planes[:, idx])
    reps, tn = 0, 0.0
    while tn < min(args.cpu_seconds, 6.0) and reps < 8:      # a few seconds of wall time on all threads
        t = time.perf_counter()
        r = otrace(p, chain, sub, nthreads=cores)
        tn += time.perf_counter() - t
        reps += 1
    tn /= reps
    rate = len(idx) * 7 / tn / 1e6
    # parity of the GPU frame on the sampled pixels
    frame = ctx.trace(planes).reshape(-1, 4)[idx]
    err = float(np.linalg.norm(vct.half_to_float(frame).astype(np.float64) - r["rgba32f"]) /
                max(np.linalg.norm(r["rgba32f"].astype(np.float64)), 1e-30))
    return {"value": round(rate, 2), "unit": "Mcones/s", "cores": cores, "kind": "port",
            "sample": (f"the whole {w}x{h} frame" if every == 1 else
                       f"every {every}th 8x8 tile of the same {w}x{h} frame") +
                      f" ({len(idx)} px, {r['total_steps']} cone steps), {reps} passes of {tn:.2f} s on {cores} threads "
                      f"({reps * tn * cores:.0f} core-seconds)",
            "host_threads_visible": os.cpu_count(), "cgroup_cpu_quota": quota,
            "value_1thread": round(rate_1t, 3), "gpu_vs_oracle_rel_l2": err,
            "ms_per_frame_extrapolated": round(w * h * 7 / rate / 1e3, 1)}


if __name__ == "__main__":
    main()
