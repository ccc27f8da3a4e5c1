"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle on the same seeded
inputs.  Bars: bit-exact for the integer/byte stages (layout, mips, voxelization) and for the
march (per-cone step counts and raw cone vec4s); the RGBA16F frame within 1e-3 relative L2 of the
oracle's fp32 frame (BASELINE.json north_star tolerance) -- in practice it matches the oracle's
own fp16 rounding except where powf differs by an ulp."""
import numpy as np
import pytest

import synth
import vctpkg

pytestmark = pytest.mark.gpu

REL_L2_TOL = 1e-3   # north_star: "within 1e-3 relative L2 (fp32)"


@pytest.fixture(scope="module")
def vct():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a real MI355X"
    return vctpkg.load()


def make_ctx(vct, V, w, h, **kw):
    return vct.Context(vct.default_config(voxel_dim=V, width=w, height=h, debug_outputs=1, **kw))


def check_frame(vct, oracle, ctx, chain, planes, w, h, p=None, rows=None):
    p = p or oracle.default_params(ctx.cfg.voxel_dim)
    ref = oracle.trace(p, chain, planes, nthreads=8, want_cones=True)
    out = ctx.trace(planes, rows=rows)
    steps, cones = ctx.steps(), ctx.cones()
    sel = np.ones(h * w, bool)
    if rows is not None:
        sel[:] = False
        sel[rows[0] * 8 * w: min(rows[1] * 8, h) * w] = True
    assert np.array_equal(steps[sel], ref["steps"][sel]), "per-cone step counts differ"
    assert np.array_equal(cones[sel].view(np.uint32), ref["cones"][sel].view(np.uint32)), \
        "raw cone results are not bit-identical"
    got = vct.half_to_float(out.reshape(-1, 4))[sel]
    err = synth.rel_l2(got, ref["rgba32f"][sel])
    assert err <= REL_L2_TOL, err
    same16 = (out.reshape(-1, 4)[sel] == ref["rgba16f"][sel]).mean()
    assert same16 > 0.999, same16
    if rows is None:
        assert ctx.last_step_count() == ref["total_steps"]
    return ref, out


@pytest.mark.parametrize("V", [16, 64])
def test_layout_roundtrip_and_mips(vct, oracle, V):
    l0 = synth.noise_volume(V, seed=V, occupancy=0.2)
    chain = oracle.build_mips(l0)
    with make_ctx(vct, V, 8, 8) as ctx:
        ctx.upload_chain(chain)
        assert np.array_equal(ctx.download_chain(), chain)          # linear <-> Morton
        junk = np.random.default_rng(0).integers(0, 256, chain.shape, dtype=np.uint8)
        junk[: V ** 3] = l0.reshape(-1, 4)
        ctx.upload_chain(junk)
        ctx.build_mips()
        assert np.array_equal(ctx.download_chain(), chain)          # HIP mips == oracle mips


def test_mips_random_bytes_rounding(vct, oracle):
    V = 32
    l0 = np.random.default_rng(5).integers(0, 256, (V, V, V, 4), dtype=np.uint8)
    chain = oracle.build_mips(l0)
    with make_ctx(vct, V, 8, 8) as ctx:
        ctx.upload_volume(l0)
        ctx.build_mips()
        assert np.array_equal(ctx.download_chain(), chain)


@pytest.mark.parametrize("variant", [0, 1, 2, 4])
def test_trace_random_gbuffer(vct, oracle, variant):
    V, w, h = 64, 128, 128
    chain = oracle.build_mips(synth.noise_volume(V))
    planes = synth.random_gbuffer(w * h, seed=42, discard_frac=0.05)
    with make_ctx(vct, V, w, h, trace_variant=variant) as ctx:
        ctx.upload_chain(chain)
        check_frame(vct, oracle, ctx, chain, planes, w, h)


@pytest.mark.parametrize("V,variant", [(64, 0), (256, 0), (32, 1)])
def test_footprint_records_give_the_same_bits(vct, oracle, V, variant):
    """vct_set_footprint_records: per-lane level samples fetch one 32-byte record (the footprint's 8 texels) instead of
    eight texels.  Random G-buffer = every sample per-lane.  Frame, per-cone results and step counts must not move;
    the records follow every way the levels >= 1 can change (upload of a chain, upload + mip build, switching off / on)."""
    w, h = 64, 48
    l0 = synth.noise_volume(V, seed=5, occupancy=0.3)
    chain = oracle.build_mips(l0)
    planes = synth.random_gbuffer(w * h, seed=11, discard_frac=0.05)
    with make_ctx(vct, V, w, h, trace_variant=variant) as ctx:
        ctx.upload_chain(chain)
        want = ctx.trace(planes).copy()
        want_steps, want_cones = ctx.steps().copy(), ctx.cones().copy()
        total = ctx.last_step_count()
        ctx.set_footprint_records(True)                      # valid chain: records built now
        got = ctx.trace(planes)
        assert np.array_equal(got, want) and np.array_equal(ctx.steps(), want_steps)
        assert np.array_equal(ctx.cones().view(np.uint32), want_cones.view(np.uint32)) and ctx.last_step_count() == total
        # another volume through upload + mip build: the records must follow
        l0b = synth.noise_volume(V, seed=6, occupancy=0.2)
        ctx.upload_volume(l0b)
        ctx.build_mips()
        with_records = ctx.trace(planes).copy()
        ctx.set_footprint_records(False)
        assert np.array_equal(ctx.trace(planes), with_records)
        ref = oracle.trace(oracle.default_params(V), oracle.build_mips(l0b), planes, nthreads=8)
        assert np.array_equal(ctx.steps(), ref["steps"])
        ctx.set_footprint_records(True)
        ctx.upload_chain(chain)                               # back to the first chain
        assert np.array_equal(ctx.trace(planes), want)


def test_footprint_records_follow_a_voxelized_scene(vct, oracle):
    """Records on while the chain comes from the voxelizer (sparse mip builds in between): same frame as without."""
    from voxel_cone_tracing_amd import scene as sc
    V, w, h = 64, 96, 64
    scene = sc.Scene(sc.CORNELL)
    planes = synth.random_gbuffer(w * h, seed=2, extent=55.0)
    frames = []
    for on in (False, True):
        with make_ctx(vct, V, w, h) as ctx:
            ctx.set_footprint_records(on)
            ctx.upload_triangles(scene.pos, scene.material, scene.albedo)
            for _ in range(2):                               # second pass: the sparse mip build
                ctx.voxelize(); ctx.inject_light(); ctx.build_mips()
            frames.append((ctx.trace(planes).copy(), ctx.steps().copy()))
    assert np.array_equal(frames[0][0], frames[1][0]) and np.array_equal(frames[0][1], frames[1][1])
    assert frames[0][1].sum() > 0


def test_footprint_records_with_the_second_bounce_and_the_whole_pass(vct, oracle):
    """Records on through the paths that swap chains: the second bounce (its chain has no records: per-texel gathers),
    a new voxelization after it, and vct_gi_pass -- frames equal to the same sequence without records."""
    from voxel_cone_tracing_amd import scene as sc
    V, w, h, S = 32, 64, 40, 128
    scene = sc.Scene(sc.CORNELL)
    cam = sc.default_camera(position=(0.0, 0.0, 58.0))
    light = (0.0, 1.0, 0.25)
    planes = synth.random_gbuffer(w * h, seed=8, extent=55.0)
    outs = []
    for on in (False, True):
        with vct.Context(vct.default_config(voxel_dim=V, width=w, height=h, shadow_map_size=S, voxel_attributes=1,
                                            debug_outputs=1)) as ctx:
            ctx.set_footprint_records(on)
            ctx.set_camera_position(tuple(cam.position)); ctx.set_light_direction(light)
            ctx.upload_triangles(scene.pos, scene.material, scene.albedo)
            ctx.upload_mesh_attributes(*scene.frames(), scene.specular)
            ctx.render_shadow_map(sc.light_view_proj(light))
            ctx.voxelize(); ctx.inject_light(); ctx.build_mips()
            a = ctx.trace(planes).copy()
            ctx.bounce()
            b = ctx.trace(planes).copy()                     # through the bounce chain
            sb = ctx.steps().copy()
            ctx.voxelize(); ctx.inject_light(); ctx.build_mips()
            c = ctx.trace(planes).copy()                     # back on the first chain, records rebuilt
            ctx.gi_pass(sc.light_view_proj(light), sc.camera_view_proj(cam, w, h))
            ctx.synchronize()
            d = ctx.download_frame().copy()
            outs.append((a, b, sb, c, d))
    for x, y in zip(outs[0], outs[1]):
        assert np.array_equal(x, y)
    assert np.array_equal(outs[0][0], outs[0][3]) and (outs[0][0] != outs[0][1]).any()


def test_loose_variant_stays_inside_the_frame_tolerance(vct, oracle):
    """config.trace_variant = 3 (one-multiply unorm8 decode, reciprocal-multiply divisions -- the opt-in kernel that
    prices the exactness, bench.py exactness_tax) is NOT bit-exact: it must stay within the north-star's 1e-3 relative
    L2 of the oracle's frame, and only a small fraction of the cones may take a different number of steps."""
    V, w, h = 64, 128, 128
    chain = oracle.build_mips(synth.noise_volume(V))
    planes = synth.random_gbuffer(w * h, seed=42, discard_frac=0.05)
    ref = oracle.trace(oracle.default_params(V), chain, planes, nthreads=8)
    with make_ctx(vct, V, w, h) as ctx:
        ctx.upload_chain(chain)
        exact = ctx.trace(planes).copy()
        assert np.array_equal(ctx.steps(), ref["steps"])
        ctx.set_trace_variant(3)
        loose = ctx.trace(planes).copy()
        changed = int((ctx.steps() != ref["steps"]).sum())
        ctx.set_trace_variant(0)
        again = ctx.trace(planes)
    assert np.array_equal(again, exact)                          # the switch goes back
    err = synth.rel_l2(vct.half_to_float(loose.reshape(-1, 4)), ref["rgba32f"])
    assert err <= 1e-3, err
    assert changed <= 0.001 * ref["steps"].size, changed          # a step count flips only where alpha lands on MAX_ALPHA
    print(f"loose variant: rel-L2 {err:.2e}, {changed} of {ref['steps'].size} cones with another step count, "
          f"{int((loose != exact).sum())} of {exact.size} halves differ")


def test_compacted_trace_equals_the_tile_trace_on_sparse_frames(vct, oracle):
    """config.trace_variant = 4 (live pixels of 16x16 super-tiles packed into whole waves) gives the frame, the per-cone
    step counts and the total of variant 0 -- on a frame with many discarded pixels, a ragged size, and a slab."""
    V, w, h = 32, 75, 53
    chain = oracle.build_mips(synth.noise_volume(V, seed=9, occupancy=0.1))
    planes = synth.random_gbuffer(w * h, seed=3, discard_frac=0.55)
    with make_ctx(vct, V, w, h) as ctx:
        ctx.upload_chain(chain)
        ref, a = check_frame(vct, oracle, ctx, chain, planes, w, h)
        na = ctx.last_step_count()
        ctx.set_trace_variant(4)
        ref2, b = check_frame(vct, oracle, ctx, chain, planes, w, h)
        assert ctx.last_step_count() == na
        assert np.array_equal(a, b)
        slab = ctx.trace(planes, rows=(2, 5))
        with pytest.raises(vct.VctError):        # its step counts are per virtual tile: no per-row histogram to hand out
            ctx.last_row_steps()
        ctx.set_trace_variant(0)
        assert np.array_equal(slab, ctx.trace(planes, rows=(2, 5)))
        assert int(ctx.last_row_steps().sum()) == ctx.last_step_count()


def test_trace_coherent_gbuffer_dense_volume(vct, oracle):
    V, w, h = 64, 96, 64
    chain = oracle.build_mips(synth.noise_volume(V, seed=3, occupancy=0.25))
    planes = synth.coherent_gbuffer(w, h)
    with make_ctx(vct, V, w, h) as ctx:
        ctx.upload_chain(chain)
        check_frame(vct, oracle, ctx, chain, planes, w, h)


def test_trace_ragged_frame_and_uniforms(vct, oracle):
    V, w, h = 32, 37, 21          # not multiples of the 8x8 tile
    chain = oracle.build_mips(synth.noise_volume(V, seed=9, occupancy=0.1))
    planes = synth.random_gbuffer(w * h, seed=1, discard_frac=0.2)
    cam, light = (10.0, 20.0, -5.0), (0.3, 0.8, -0.4)
    p = oracle.default_params(V, camera_pos=cam, light_dir=light, ambient_factor=0.6,
                              tan_specular=0.105)
    with make_ctx(vct, V, w, h, ambient_factor=0.6, tan_specular=0.105) as ctx:
        ctx.upload_chain(chain)
        ctx.set_camera_position(cam)
        ctx.set_light_direction(light)
        ref, out = check_frame(vct, oracle, ctx, chain, planes, w, h, p=p)
        dead = planes[18] < 0.5
        assert dead.any()
        white = np.float16(1.0).view(np.uint16)
        assert np.all(out.reshape(-1, 4)[dead] == white)       # VCT.h:156-159 (ambient >= 0.5)


def test_trace_empty_and_opaque_volumes(vct, oracle):
    V, w, h = 16, 16, 16
    planes = synth.random_gbuffer(w * h, seed=2)
    for rgba in ((0, 0, 0, 0), (51, 102, 204, 255), (128, 128, 128, 64)):
        l0 = np.zeros((V, V, V, 4), np.uint8)
        l0[...] = rgba
        chain = oracle.build_mips(l0)
        with make_ctx(vct, V, w, h) as ctx:
            ctx.upload_chain(chain)
            ref, _ = check_frame(vct, oracle, ctx, chain, planes, w, h)
            if rgba[3] == 255:
                assert np.all(ref["steps"] == 1)


def test_trace_clamp_mode(vct, oracle):
    V, w, h = 32, 32, 32
    chain = oracle.build_mips(synth.noise_volume(V, seed=4, occupancy=0.3))
    planes = synth.random_gbuffer(w * h, seed=6)
    p = oracle.default_params(V, wrap_repeat=0)
    with make_ctx(vct, V, w, h, wrap_repeat=0) as ctx:
        ctx.upload_chain(chain)
        check_frame(vct, oracle, ctx, chain, planes, w, h, p=p)


def test_slab_equals_full_frame(vct, oracle):
    V, w, h = 32, 64, 72          # 9 tile rows
    chain = oracle.build_mips(synth.noise_volume(V, seed=8, occupancy=0.1))
    planes = synth.random_gbuffer(w * h, seed=12)
    with make_ctx(vct, V, w, h) as ctx:
        ctx.upload_chain(chain)
        full = ctx.trace(planes)
        parts = np.zeros_like(full)
        for r0, r1 in ((0, 3), (3, 4), (4, 9)):
            slab = ctx.trace(planes, rows=(r0, r1))
            parts[r0 * 8:r1 * 8] = slab[r0 * 8:r1 * 8]
            assert not slab[:r0 * 8].any() and not slab[r1 * 8:].any()
        assert np.array_equal(parts, full)      # bit-identical to the single-GPU frame
        check_frame(vct, oracle, ctx, chain, planes, w, h, rows=(3, 7))


def random_scene(ntri, seed, big=2):
    r = np.random.default_rng(seed)
    c = r.uniform(-1300, 1300, (ntri, 1, 3))
    pos = c + r.normal(scale=25.0, size=(ntri, 3, 3))
    pos[:big] = c[:big] + r.normal(scale=400.0, size=(big, 3, 3))     # a few large triangles
    pos[big] = pos[big, 0][None, :]                                   # degenerate (point)
    # axis-aligned wall: exercises exact-on-boundary overlap decisions
    pos[big + 1] = [[-1000, -1000, 200], [1000, -1000, 200], [1000, 1000, 200]]
    mat = r.integers(0, 5, ntri).astype(np.int32)
    alb = r.uniform(0.1, 1.0, (5, 4)).astype(np.float32)
    return pos.astype(np.float32), mat, alb


def light_setup(S, seed):
    r = np.random.default_rng(seed)
    yy, xx = np.meshgrid(np.arange(S), np.arange(S), indexing="ij")
    depth = 0.5 + 0.2 * np.sin(xx * 0.11) * np.cos(yy * 0.07) + r.uniform(-0.01, 0.01, (S, S))
    depth = np.round(depth * (2 ** 24 - 1)) / (2 ** 24 - 1)
    vp = np.array([[1 / 120.0, 0, 0, 0], [0, 0, -1 / 120.0, 0], [0, -1 / 100.0, 0, 0],
                   [0, 0, 0, 1]], np.float32)      # row-major light ortho looking down -Y
    return depth.astype(np.float32), vp


@pytest.mark.parametrize("with_shadow", [False, True])
def test_voxelize_conservative_matches_oracle(vct, oracle, with_shadow):
    V = 64
    pos, mat, alb = random_scene(600, seed=21)
    depth, vp = light_setup(256, 3) if with_shadow else (None, None)
    p = oracle.default_params(V)
    sc = oracle.make_scene(pos, mat, alb, shadow_depth=depth, light_vp=vp)
    want_l0 = oracle.voxelize_conservative(p, sc)
    assert 0.001 < (want_l0[..., 3] > 0).mean() < 0.5
    want = oracle.build_mips(want_l0)
    with make_ctx(vct, V, 8, 8) as ctx:
        ctx.upload_triangles(pos, mat, alb)
        if with_shadow:
            ctx.upload_shadow_map(depth, vp)
        ctx.voxelize()
        ctx.inject_light()
        ctx.build_mips()
        got = ctx.download_chain()
        assert np.array_equal(got, want)
        # re-voxelizing is idempotent (accumulators are cleared, result is order-independent)
        ctx.voxelize()
        ctx.inject_light()
        ctx.build_mips()
        assert np.array_equal(ctx.download_chain(), want)


def test_call_order_errors(vct):
    with make_ctx(vct, 16, 8, 8) as ctx:
        with pytest.raises(vct.VctError):
            ctx.voxelize()                      # no triangles yet
        with pytest.raises(vct.VctError):
            ctx.inject_light()                  # no accumulators yet
        with pytest.raises(vct.VctError):
            ctx.trace_resident()                # nothing resident
        with pytest.raises(vct.VctError):
            ctx.voxelize(7)                     # unknown mode
        bad = np.zeros((23, 4 * 4), np.float32)
        gb = vct.GBuffer()
        gb.planes, gb.width, gb.height, gb.layout, gb.location = bad.ctypes.data, 4, 4, 0, 0
        import ctypes as C
        assert vct.lib().vct_trace(ctx._h, C.byref(gb), None, 0) != 0
        assert b"size differs" in vct.lib().vct_last_error(ctx._h)


def test_stale_mips_are_refused_and_upload_then_sparse_builds_stay_correct(vct, oracle):
    """Call orders around vct_build_mips: a trace over a chain whose level 0 changed since the last mip
    build is an error (the reference rebuilds its mips right after every voxelization, VCT.h:248); after
    an uploaded volume, voxelize + inject twice and ONE mip build must leave no ancestor of the uploaded
    content behind (the sparse mip build may only skip bricks whose ancestors are known to be zero)."""
    V, w, h = 32, 16, 8
    pos, mat, alb = random_scene(200, seed=12)
    p = oracle.default_params(V)
    want = oracle.build_mips(oracle.voxelize_conservative(p, oracle.make_scene(pos, mat, alb)))
    planes = synth.random_gbuffer(w * h, seed=2)
    with make_ctx(vct, V, w, h) as ctx:
        junk = np.random.default_rng(5).integers(0, 256, (V, V, V, 4), dtype=np.uint8)
        ctx.upload_volume(junk)
        with pytest.raises(vct.VctError):
            ctx.trace(planes)                       # level 0 uploaded, mips not rebuilt
        ctx.upload_chain(oracle.build_mips(junk))   # a full chain is consistent by definition
        ctx.trace(planes)
        ctx.upload_triangles(pos, mat, alb)
        ctx.voxelize(); ctx.inject_light()
        with pytest.raises(vct.VctError):
            ctx.trace(planes)                       # injected, mips stale
        ctx.voxelize(); ctx.inject_light()          # second pass before any mip build
        ctx.build_mips()
        assert np.array_equal(ctx.download_chain(), want)
        ctx.voxelize(); ctx.inject_light(); ctx.build_mips()      # and the sparse form afterwards
        assert np.array_equal(ctx.download_chain(), want)
        check_frame(vct, oracle, ctx, want, planes, w, h)
        steps = ctx.last_step_count()
        assert ctx.selftest_const_divide(75.0) == 0
        assert ctx.last_step_count() == steps       # the self-test does not disturb the step counters


def test_texel_buffer_conversion_is_the_exact_decode(vct):
    """Round 6: the trace kernels take the four floats the texture path returns for an RGBA8 UNORM texel (typed-buffer
    load) instead of decoding the bytes -- [GL] value = byte / 255.  Every byte value in every channel
    position through that load equals the exact decode bit for bit; vct_create refuses a device where it does not."""
    with make_ctx(vct, 16, 8, 8) as ctx:
        assert ctx.selftest_texel_buffer() == 0


def test_const_divide_exhaustive(vct):
    """The trace kernel's x/d (fma(x, r_hi, x * r_lo): two instructions) equals the IEEE divide for EVERY finite fp32 x,
    for every divisor the BASELINE grids and apertures use: half_G = 75 and the per-step occlusion denominators
    1 + 0.03*diameter (trace.fs:61,101).  The form is not exact for arbitrary divisors -- the library verifies
    the divisors of each step table on the device before using them and otherwise runs the IEEE-divide kernel;
    `vct_selftest_const_divide` is that check."""
    with make_ctx(vct, 16, 8, 8) as ctx:
        divisors = [75.0]
        for V in (64, 256, 512, 1024):
            vs = np.float32(150.0) / np.float32(V)
            for t in (0.577, 0.07, 0.105, 0.2):
                dist = vs
                while dist < 75.0:
                    dia = max(vs, np.float32(2.0) * np.float32(t) * dist)
                    divisors.append(float(np.float32(1.0) + np.float32(0.03) * dia))
                    dist = np.float32(dist + dia)
        for d in sorted(set(divisors)):
            assert ctx.selftest_const_divide(d) == 0, d
        # arbitrary divisors may or may not pass; the call reports, it does not fail
        others = [ctx.selftest_const_divide(d) for d in (37.5, 3.0, 1.0175781, 2.7341, 0.3333333, 1e-3, 977.0)]
        assert all(n >= 0 for n in others)


def test_facade_demo_matches_binding(vct):
    """The C++ caller written against host/Voxel_Cone_Tracing.h (the reference application's call
    sequence) produces the same RGBA16F frame as the same stages driven through the binding."""
    import os
    import subprocess
    from voxel_cone_tracing_amd import scene as sc
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "voxel-cone-tracing_amd", "vct_demo")
    assert os.path.exists(exe), "build it with `make demo`"
    V, w, h, S = 64, 128, 128, 512
    out = subprocess.run([exe, "--scene", "procedural:cornell", "--voxels", str(V), "--size", f"{w}x{h}",
                          "--shadow", str(S), "--frames", "1"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    fields = dict(kv.split("=") for kv in out.stdout.strip().split("\n")[-1].split())
    scene = sc.Scene(sc.CORNELL)
    light = (0.0, 1.0, 0.25)
    import raster_oracle
    depth, light_vp = raster_oracle.shadow_map(sc, scene, light, S)
    cam = sc.default_camera(position=(0.0, 0.0, 58.0))
    planes = raster_oracle.gbuffer(sc, scene, cam, w, h, depth, light_vp)
    with vct.Context(vct.default_config(voxel_dim=V, width=w, height=h, shadow_map_size=S)) as ctx:
        ctx.set_camera_position((0.0, 0.0, 58.0))
        ctx.set_light_direction(light)
        ctx.upload_triangles(scene.pos, scene.material, scene.albedo)
        ctx.upload_shadow_map(depth, light_vp)
        ctx.voxelize()
        ctx.inject_light()
        ctx.build_mips()
        frame = ctx.trace(planes)
        assert int(fields["cone_steps"]) == ctx.last_step_count()
    hsh = 1469598103934665603
    for v in frame.reshape(-1).tolist():
        hsh = ((hsh ^ v) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    assert fields["fnv1a"] == f"{hsh:016x}"
    assert (planes[18] >= 0.5).mean() > 0.5          # the camera actually sees the box
    # DynamicLight: every Render() is one vct_gi_pass; with an unchanged light it is the same frame
    dyn = subprocess.run([exe, "--scene", "procedural:cornell", "--voxels", str(V), "--size", f"{w}x{h}",
                          "--shadow", str(S), "--frames", "2", "--dynamic-light"], capture_output=True, text=True, timeout=300)
    assert dyn.returncode == 0, dyn.stdout + dyn.stderr
    two = subprocess.run([exe, "--scene", "procedural:cornell", "--voxels", str(V), "--size", f"{w}x{h}",
                          "--shadow", str(S), "--frames", "2"], capture_output=True, text=True, timeout=300)
    fd = dict(kv.split("=") for kv in dyn.stdout.strip().split("\n")[-1].split())
    f2 = dict(kv.split("=") for kv in two.stdout.strip().split("\n")[-1].split())
    assert fd["fnv1a"] == f2["fnv1a"] and fd["cone_steps"] == f2["cone_steps"]


def test_frame_gather_single_rank_device_path(vct, oracle):
    """slabs.FrameGather on the GPU with world = 1 (the N-GPU path is the same code over RCCL):
    vct_trace_slab writes the slab straight into the gather buffer in HBM."""
    import torch
    from voxel_cone_tracing_amd import slabs
    V, w, h = 32, 40, 21
    chain = oracle.build_mips(synth.noise_volume(V, seed=8, occupancy=0.1))
    planes = synth.random_gbuffer(w * h, seed=12)
    with make_ctx(vct, V, w, h) as ctx:
        ctx.upload_chain(chain)
        want = ctx.trace(planes)
        # a middle slab of a 3-way split, written by the kernel straight into its gather buffer
        world, rank = 3, 1
        fg = slabs.FrameGather(h, w, world, rank, "cuda:0", root=rank)
        r0, r1 = slabs.partition(h, world)[rank]
        y0, y1 = fg.my_rows()
        ctx.set_frame_target(fg.slab.data_ptr() - y0 * w * 8)
        ctx.trace(planes, rows=(r0, r1), out_device_ptr=fg.slab.data_ptr() - y0 * w * 8)
        ctx.trace_resident()
        ctx.synchronize()
        torch.cuda.synchronize()
        assert np.array_equal(fg.slab[: y1 - y0].cpu().numpy().view(np.uint16), want[y0:y1])
        ctx.set_frame_target(None)
        assert np.array_equal(ctx.trace(planes), want)


@pytest.mark.parametrize("with_shadow", [False, True])
def test_second_bounce_matches_oracle(vct, oracle, with_shadow):
    """BASELINE.json config 3 (2-bounce), at test size: voxel attributes, the bounce-1 level 0 and
    its mip chain are bit-identical to the oracle's, and so is the screen trace through it."""
    V, w, h = 32, 24, 16
    pos, mat, alb = random_scene(300, seed=33)
    depth, vp = light_setup(128, 5) if with_shadow else (None, None)
    p = oracle.default_params(V)
    sc = oracle.make_scene(pos, mat, alb, shadow_depth=depth, light_vp=vp)
    l0, want_alb, want_nrm = oracle.voxelize_conservative_attr(p, sc)
    chain0 = oracle.build_mips(l0)
    want_l1, want_steps = oracle.bounce(p, chain0, want_alb, want_nrm, nthreads=8)
    want_chain1 = oracle.build_mips(want_l1)
    assert (want_l1 != l0).any() and want_steps > 0
    planes = synth.random_gbuffer(w * h, seed=4)
    with make_ctx(vct, V, w, h, voxel_attributes=1) as ctx:
        with pytest.raises(vct.VctError):
            ctx.bounce()                          # nothing voxelized yet
        ctx.upload_triangles(pos, mat, alb)
        if with_shadow:
            ctx.upload_shadow_map(depth, vp)
        ctx.voxelize()
        ctx.inject_light()
        with pytest.raises(vct.VctError):
            ctx.bounce()                          # mips of bounce 0 not built yet
        ctx.build_mips()
        assert np.array_equal(ctx.download_chain(), chain0)
        got_alb, got_nrm = ctx.voxel_attributes()
        assert np.array_equal(got_alb, want_alb)
        assert np.array_equal(got_nrm, want_nrm)
        ctx.bounce()
        assert ctx.last_step_count() == want_steps
        assert np.array_equal(ctx.download_chain(), want_chain1)
        check_frame(vct, oracle, ctx, want_chain1, planes, w, h)      # the trace reads the bounce-1 chain
        ctx.voxelize()                                                # a new inject returns to bounce 0
        ctx.inject_light()
        ctx.build_mips()
        assert np.array_equal(ctx.download_chain(), chain0)
        check_frame(vct, oracle, ctx, chain0, planes, w, h)


def test_second_bounce_visits_brick_zero_once(vct, oracle):
    """Found by tools/fuzz_gpu.py (seeds 2857, 1268 of round 3): the second bounce walks the voxelizer's brick slots, and a
    slot that only the mark-only run of the reference-mode voxelizer created (rare: its pixel-centre raster + depth
    quantisation can name a voxel in a brick the conservative mode does not touch) had no brick recorded -- it read as
    brick 0, so a scene with content in brick 0 marched that brick's voxels once more per such slot: same chain, too many
    cone steps.  A soup of the fuzzer's kind that has such a slot (searched on the CPU) + a triangle in the grid's corner."""
    V = 64
    r = np.random.default_rng(48)
    ntri = int(r.integers(50, 400))
    c = r.uniform(-1300, 1300, (ntri, 1, 3))
    pos = (c + r.normal(scale=r.choice([10.0, 40.0, 150.0]), size=(ntri, 3, 3))).astype(np.float32)
    if r.random() < 0.5:
        pos[: ntri // 8] = np.round(pos[: ntri // 8] / 58.59375) * 58.59375      # vertices on voxel corners
    mat = r.integers(0, 4, ntri).astype(np.int32)
    alb = r.uniform(0.05, 1.0, (4, 4)).astype(np.float32)
    pos[5] = [[-1450.0, -1430.0, -1400.0], [-1250.0, -1440.0, -1350.0], [-1400.0, -1200.0, -1420.0]]
    p = oracle.default_params(V)
    sc = oracle.make_scene(pos, mat, alb)
    l0, want_alb, want_nrm = oracle.voxelize_conservative_attr(p, sc)
    nb = V // 8

    def bricks(level0):
        return level0[..., 3].reshape(nb, 8, nb, 8, nb, 8).any(axis=(1, 3, 5))
    assert l0[:8, :8, :8, 3].any()                               # brick 0 (voxels 0..7 on every axis) holds content
    assert (bricks(oracle.voxelize_reference(p, sc)) & ~bricks(l0)).any()      # a brick only the reference mode touches
    chain0 = oracle.build_mips(l0)
    want_l1, want_steps = oracle.bounce(p, chain0, want_alb, want_nrm, nthreads=8)
    with make_ctx(vct, V, 8, 8, voxel_attributes=1) as ctx:
        ctx.upload_triangles(pos, mat, alb)
        ctx.voxelize(); ctx.inject_light(); ctx.build_mips()
        assert ctx.stage_counts()["accumulator_bricks"] > int(bricks(l0).sum())     # slots without a conservative fragment
        ctx.bounce()
        assert ctx.last_step_count() == want_steps
        assert np.array_equal(ctx.download_chain(), oracle.build_mips(want_l1))


@pytest.mark.parametrize("with_shadow", [False, True])
def test_heavy_bricks_are_cut_into_chunks(vct, oracle, with_shadow):
    """The voxelize pass takes a brick slot's fragments heaviest-first and cuts a slot above 4096 fragments into chunks
    whose LDS sums meet in HBM accumulators, resolved by a second kernel.  8,000 small triangles crowded
    into a few bricks: slots far above the limit, with voxel attributes, twice in a row (the accumulators must
    come back to zero), then the second bounce on top."""
    V = 32
    r = np.random.default_rng(5)
    ntri = 8000
    c = np.array([[[-300.0, 150.0, 420.0]]]) + r.normal(scale=60.0, size=(ntri, 1, 3))
    pos = (c + r.normal(scale=70.0, size=(ntri, 3, 3))).astype(np.float32)
    mat = r.integers(0, 5, ntri).astype(np.int32)
    alb = r.uniform(0.1, 1.0, (5, 4)).astype(np.float32)
    depth, vp = light_setup(128, 9) if with_shadow else (None, None)
    p = oracle.default_params(V)
    sc = oracle.make_scene(pos, mat, alb, shadow_depth=depth, light_vp=vp)
    l0, want_alb, want_nrm = oracle.voxelize_conservative_attr(p, sc)
    chain0 = oracle.build_mips(l0)
    with make_ctx(vct, V, 8, 8, voxel_attributes=1) as ctx:
        ctx.upload_triangles(pos, mat, alb)
        if with_shadow:
            ctx.upload_shadow_map(depth, vp)
        counts = ctx.stage_counts()
        assert counts["vox_candidates"] > 4096 * counts["accumulator_bricks"]      # pigeonhole: a slot above the limit
        for _ in range(2):
            ctx.voxelize(); ctx.inject_light(); ctx.build_mips()
            assert np.array_equal(ctx.download_chain(), chain0)
            got_alb, got_nrm = ctx.voxel_attributes()
            assert np.array_equal(got_alb, want_alb) and np.array_equal(got_nrm, want_nrm)
        want_l1, want_steps = oracle.bounce(p, chain0, want_alb, want_nrm, nthreads=8)
        ctx.bounce()
        assert ctx.last_step_count() == want_steps
        assert np.array_equal(ctx.download_chain(), oracle.build_mips(want_l1))
    with make_ctx(vct, V, 8, 8) as ctx:                          # and without attributes
        ctx.upload_triangles(pos, mat, alb)
        if with_shadow:
            ctx.upload_shadow_map(depth, vp)
        ctx.voxelize(); ctx.inject_light(); ctx.build_mips()
        assert np.array_equal(ctx.download_chain(), chain0)


def test_second_bounce_dense_scene_overflows_the_voxel_list(vct, oracle):
    """More than 1/8 of the grid occupied: the compacted list of occupied voxels (capacity V^3 / 8) overflows and the
    bricks that did not fit are marched brick by brick (k_bounce_bricks).  Same level 0, chain and step count."""
    V = 16
    r = np.random.default_rng(77)
    ntri = 1500
    c = r.uniform(-1400, 1400, (ntri, 1, 3))
    pos = (c + r.normal(scale=260.0, size=(ntri, 3, 3))).astype(np.float32)        # big triangles: a dense grid
    mat = r.integers(0, 3, ntri).astype(np.int32)
    alb = r.uniform(0.2, 0.9, (3, 4)).astype(np.float32)
    p = oracle.default_params(V)
    sc = oracle.make_scene(pos, mat, alb)
    l0, want_alb, want_nrm = oracle.voxelize_conservative_attr(p, sc)
    occupied = int((l0[..., 3] > 0).sum())
    assert occupied > V ** 3 // 8, occupied                                        # the list cannot hold them all
    chain0 = oracle.build_mips(l0)
    want_l1, want_steps = oracle.bounce(p, chain0, want_alb, want_nrm, nthreads=8)
    with make_ctx(vct, V, 8, 8, voxel_attributes=1) as ctx:
        ctx.upload_triangles(pos, mat, alb)
        ctx.voxelize(); ctx.inject_light(); ctx.build_mips()
        assert np.array_equal(ctx.download_chain(), chain0)
        for _ in range(2):
            ctx.bounce()
            assert ctx.last_step_count() == want_steps
            assert np.array_equal(ctx.download_chain(), oracle.build_mips(want_l1))
            ctx.voxelize(); ctx.inject_light(); ctx.build_mips()


@pytest.mark.parametrize("w,h", [(1, 1), (9, 3)])
def test_smallest_grid_and_frames(vct, oracle, w, h):
    """voxel_dim = 8 (the minimum: one brick, 4 levels, the coarse levels of every cone step are the one-texel level)
    and frames smaller than a tile: mips, both voxelizer modes and the trace against the oracle."""
    V = 8
    l0 = synth.noise_volume(V, seed=5, occupancy=0.3)
    chain = oracle.build_mips(l0)
    pos, mat, alb = random_scene(40, seed=8)
    p = oracle.default_params(V)
    scn = oracle.make_scene(pos, mat, alb)
    with make_ctx(vct, V, w, h) as ctx:
        ctx.upload_volume(l0); ctx.build_mips()
        assert np.array_equal(ctx.download_chain(), chain)
        check_frame(vct, oracle, ctx, chain, synth.random_gbuffer(w * h, seed=w + h), w, h)
        ctx.upload_triangles(pos, mat, alb)
        ctx.voxelize(vct.VOX_REFERENCE); ctx.inject_light(); ctx.build_mips()
        assert np.array_equal(ctx.download_chain(), oracle.build_mips(oracle.voxelize_reference(p, scn)))
        ctx.voxelize(); ctx.inject_light(); ctx.build_mips()
        cons = oracle.build_mips(oracle.voxelize_conservative(p, scn))
        assert np.array_equal(ctx.download_chain(), cons)
        check_frame(vct, oracle, ctx, cons, synth.coherent_gbuffer(w, h, seed=3), w, h)


def test_bounce_needs_attributes(vct):
    pos, mat, alb = random_scene(50, seed=1)
    with make_ctx(vct, 16, 8, 8) as ctx:
        ctx.upload_triangles(pos, mat, alb)
        ctx.voxelize(); ctx.inject_light(); ctx.build_mips()
        with pytest.raises(vct.VctError):
            ctx.bounce()


def test_sparse_mips_and_bounce_track_changing_scenes(vct, oracle):
    """The sparse resolve / mip build / bounce-chain updates keep no residue when the geometry under
    them changes: voxelize scene A, then B (different bricks), then A again, with and without a
    bounce, and compare every chain with a from-scratch oracle build."""
    V = 32
    scenes = [random_scene(120, seed=s) for s in (41, 42)]
    p = oracle.default_params(V)
    want = []
    for pos, mat, alb in scenes:
        sc = oracle.make_scene(pos, mat, alb)
        l0, a, n = oracle.voxelize_conservative_attr(p, sc)
        c0 = oracle.build_mips(l0)
        l1, _ = oracle.bounce(p, c0, a, n, nthreads=8)
        want.append((c0, oracle.build_mips(l1)))
    assert not np.array_equal(want[0][0], want[1][0])
    with make_ctx(vct, V, 8, 8, voxel_attributes=1) as ctx:
        for which, do_bounce in ((0, True), (1, False), (0, False), (1, True), (1, True), (0, True)):
            pos, mat, alb = scenes[which]
            ctx.upload_triangles(pos, mat, alb)
            ctx.voxelize(); ctx.inject_light(); ctx.build_mips()
            assert np.array_equal(ctx.download_chain(), want[which][0]), (which, do_bounce, "bounce 0")
            if do_bounce:
                ctx.bounce()
                assert np.array_equal(ctx.download_chain(), want[which][1]), (which, "bounce 1")
        # an uploaded volume in between forces (and survives) the dense paths
        junk = synth.noise_volume(V, seed=3, occupancy=0.3)
        ctx.upload_volume(junk)
        ctx.build_mips()
        assert np.array_equal(ctx.download_chain(), oracle.build_mips(junk))
        with pytest.raises(vct.VctError):
            ctx.bounce()
        pos, mat, alb = scenes[1]
        ctx.upload_triangles(pos, mat, alb)
        ctx.voxelize(); ctx.inject_light(); ctx.build_mips()
        assert np.array_equal(ctx.download_chain(), want[1][0])
        ctx.bounce()
        assert np.array_equal(ctx.download_chain(), want[1][1])


@pytest.mark.parametrize("with_shadow", [False, True])
def test_voxelize_reference_mode_matches_oracle(vct, oracle, with_shadow):
    """VCT_VOX_REFERENCE: the reference's own voxelization (S/Voxelization.vs/.gs/.fs: dominant-axis
    V x V raster at pixel centres, vox.fs:58-86 index, last triangle in submission order wins) is
    bit-identical to the oracle's restatement, and switching modes on one context leaves no residue."""
    V = 64
    pos, mat, alb = random_scene(500, seed=77)
    depth, vp = light_setup(256, 9) if with_shadow else (None, None)
    p = oracle.default_params(V)
    sc = oracle.make_scene(pos, mat, alb, shadow_depth=depth, light_vp=vp)
    want_ref = oracle.build_mips(oracle.voxelize_reference(p, sc))
    want_cons = oracle.build_mips(oracle.voxelize_conservative(p, sc))
    occ_ref = (want_ref[: V ** 3, 3] > 0).mean()
    assert 0.001 < occ_ref < (want_cons[: V ** 3, 3] > 0).mean()      # centre sampling covers less
    with make_ctx(vct, V, 8, 8) as ctx:
        ctx.upload_triangles(pos, mat, alb)
        if with_shadow:
            ctx.upload_shadow_map(depth, vp)
        for mode, want in ((vct.VOX_REFERENCE, want_ref), (vct.VOX_CONSERVATIVE_AVG, want_cons),
                           (vct.VOX_REFERENCE, want_ref), (vct.VOX_REFERENCE, want_ref)):
            ctx.voxelize(mode)
            ctx.inject_light()
            ctx.build_mips()
            assert np.array_equal(ctx.download_chain(), want), mode


def test_anisotropic_mips_option_matches_oracle(vct, oracle):
    """config.anisotropic_mips = 1 (north-star option, no reference code): the six directional
    chains and the direction-weighted trace through them are bit-identical to the oracle's; the
    default isotropic path is untouched."""
    V, w, h = 32, 40, 24
    l0 = synth.noise_volume(V, seed=5, occupancy=0.2)
    chain = oracle.build_mips(l0)
    want_aniso = oracle.build_mips_aniso(l0)
    p = oracle.default_params(V)
    for planes in (synth.coherent_gbuffer(w, h), synth.random_gbuffer(w * h, seed=8, discard_frac=0.1)):
        ref = oracle.trace_aniso(p, chain, want_aniso, planes, nthreads=8, want_cones=True)
        iso = oracle.trace(p, chain, planes, nthreads=8)
        assert not np.array_equal(ref["rgba16f"], iso["rgba16f"])
        with make_ctx(vct, V, w, h, anisotropic_mips=1) as ctx:
            ctx.upload_volume(l0)
            ctx.build_mips()
            assert np.array_equal(ctx.download_chain(), chain)
            assert np.array_equal(ctx.download_aniso(), want_aniso)
            out = ctx.trace(planes)
            assert np.array_equal(ctx.steps(), ref["steps"])
            assert np.array_equal(ctx.cones().view(np.uint32), ref["cones"].view(np.uint32))
            assert ctx.last_step_count() == ref["total_steps"]
            assert (out.reshape(-1, 4) == ref["rgba16f"]).mean() > 0.999
            ctx.upload_chain(chain)                      # full-chain upload rebuilds the directional chains too
            assert np.array_equal(ctx.download_aniso(), want_aniso)
        with make_ctx(vct, V, w, h) as ctx:
            ctx.upload_chain(chain)
            check_frame(vct, oracle, ctx, chain, planes, w, h)
            with pytest.raises(vct.VctError):
                ctx.download_aniso()


def test_device_resident_tiled_gbuffer_is_traced_in_place(vct, oracle):
    """vct_gbuffer with layout = TILED, location = DEVICE: the kernel reads the caller's HBM buffer
    (no copy) and produces the frame of the host-linear path."""
    import torch
    V, w, h = 32, 37, 21
    chain = oracle.build_mips(synth.noise_volume(V, seed=9, occupancy=0.1))
    planes = synth.random_gbuffer(w * h, seed=1, discard_frac=0.2)
    tx, ty = (w + 7) // 8, (h + 7) // 8
    tiled = np.zeros((ty, tx, 23, 64), np.float32)
    img = planes.reshape(23, h, w)
    for y in range(h):
        for x in range(w):
            tiled[y // 8, x // 8, :, (y % 8) * 8 + (x % 8)] = img[:, y, x]
    dev = torch.from_numpy(tiled).cuda()
    with make_ctx(vct, V, w, h) as ctx:
        ctx.upload_chain(chain)
        want = ctx.trace(planes)
        got = ctx.trace(dev.data_ptr(), layout=vct.GB_TILED)
        assert np.array_equal(got, want)
        dev.zero_()                                     # the context holds no copy of it
        torch.cuda.synchronize()
        blank = ctx.trace(dev.data_ptr(), layout=vct.GB_TILED)
        assert not np.array_equal(blank, want)


def test_accumulators_are_allocated_per_touched_brick(vct):
    """A 1024^3 context with voxel attributes: the voxelizer's accumulators (16 + 24 B per voxel) and the
    resolved attribute volumes (8 B per voxel) are pooled per brick the mesh can touch, so the context costs
    the 4.57 GiB chain plus megabytes -- not the 52 GiB of dense V^3 buffers."""
    import torch
    from voxel_cone_tracing_amd import scene as sc
    scene = sc.Scene(sc.ATRIUM, 0.5, 1234)
    torch.cuda.synchronize()
    free0, _ = torch.cuda.mem_get_info()
    with vct.Context(vct.default_config(voxel_dim=1024, width=64, height=64, shadow_map_size=512,
                                        voxel_attributes=1)) as ctx:
        ctx.upload_triangles(scene.pos, scene.material, scene.albedo)
        ctx.upload_mesh_attributes(*scene.frames(), scene.specular)
        ctx.render_shadow_map(sc.light_view_proj((0.0, 1.0, 0.25)))
        ctx.voxelize(); ctx.inject_light(); ctx.build_mips()
        ctx.voxelize(vct.VOX_REFERENCE); ctx.inject_light(); ctx.build_mips()
        ctx.synchronize()
        free1, _ = torch.cuda.mem_get_info()
        used_gib = (free0 - free1) / 2 ** 30
        assert 4.5 < used_gib < 8.0, used_gib


def test_contexts_release_all_hbm(vct):
    """vct_destroy frees every buffer a context acquired along the way (accumulators, attribute and
    directional chains, work lists, raster buffers, bounce chain, ...)."""
    import torch
    from voxel_cone_tracing_amd import scene as sc
    scene = sc.Scene(sc.CORNELL)

    def exercise():
        with vct.Context(vct.default_config(voxel_dim=64, width=64, height=48, shadow_map_size=256,
                                            voxel_attributes=1, anisotropic_mips=1, debug_outputs=1)) as ctx:
            ctx.upload_triangles(scene.pos, scene.material, scene.albedo)
            ctx.upload_mesh_attributes(*scene.frames(), scene.specular)
            ctx.render_shadow_map(sc.light_view_proj((0.0, 1.0, 0.25)))
            ctx.voxelize(vct.VOX_REFERENCE); ctx.inject_light(); ctx.build_mips()
            ctx.voxelize(); ctx.inject_light(); ctx.build_mips()
            ctx.bounce()
            ctx.render_gbuffer(sc.camera_view_proj(sc.default_camera(position=(0.0, 0.0, 58.0)), 64, 48))
            ctx.trace_current()
            ctx.download_chain(); ctx.download_gbuffer(); ctx.download_aniso()

    exercise()                                   # first use may grow runtime-internal pools
    torch.cuda.synchronize()
    free0, _ = torch.cuda.mem_get_info()
    for _ in range(5):
        exercise()
    torch.cuda.synchronize()
    free1, _ = torch.cuda.mem_get_info()
    assert free0 - free1 < 8 << 20, (free0, free1)     # nothing accumulates (a context here holds > 100 MB)


def test_extreme_sizes(vct, oracle):
    """Smallest grid (8^3: one brick, 4 levels), 1x1 and 1xN frames, all pixels discarded."""
    V = 8
    l0 = synth.noise_volume(V, seed=2, occupancy=0.3)
    chain = oracle.build_mips(l0)
    for w, h in ((1, 1), (1, 9), (9, 1), (8, 8)):
        planes = synth.random_gbuffer(w * h, seed=w * 31 + h)
        with make_ctx(vct, V, w, h) as ctx:
            ctx.upload_volume(l0)
            ctx.build_mips()
            assert np.array_equal(ctx.download_chain(), chain)
            check_frame(vct, oracle, ctx, chain, planes, w, h)
            dead = planes.copy()
            dead[18] = 0.0
            out = ctx.trace(dead)
            assert ctx.last_step_count() == 0
            assert np.all(out == np.array([0x3800, 0x3800, 0x3800, 0x3c00], np.uint16))   # clear colour 0.5,0.5,0.5,1
    pos, mat, alb = random_scene(40, seed=3)
    with make_ctx(vct, V, 8, 8, voxel_attributes=1) as ctx:
        ctx.upload_triangles(pos, mat, alb)
        ctx.voxelize(); ctx.inject_light(); ctx.build_mips()
        p = oracle.default_params(V)
        l0v, a, n = oracle.voxelize_conservative_attr(p, oracle.make_scene(pos, mat, alb))
        c0 = oracle.build_mips(l0v)
        assert np.array_equal(ctx.download_chain(), c0)
        ctx.bounce()
        l1, _ = oracle.bounce(p, c0, a, n)
        assert np.array_equal(ctx.download_chain(), oracle.build_mips(l1))


def test_bounce_is_refused_after_a_new_mesh_until_it_is_voxelized(vct, oracle):
    """vct_upload_triangles replaces the per-brick attribute pools but level 0 / the touched-brick
    flags still describe the OLD mesh; a bounce in that state would index the new pools with the old bricks."""
    V = 32
    pos, mat, alb = random_scene(300, seed=33)
    pos2, mat2, alb2 = random_scene(40, seed=91)
    pos2 = (pos2 * 0.2 + 900.0).astype(np.float32)          # a small mesh in a corner: most old bricks get no slot
    p = oracle.default_params(V)
    with make_ctx(vct, V, 8, 8, voxel_attributes=1) as ctx:
        ctx.upload_triangles(pos, mat, alb)
        ctx.voxelize(); ctx.inject_light(); ctx.build_mips()
        ctx.bounce()
        ctx.upload_triangles(pos2, mat2, alb2)
        with pytest.raises(vct.VctError) as e:
            ctx.bounce()
        assert "voxel attributes" in str(e.value)
        ctx.voxelize(); ctx.inject_light(); ctx.build_mips()
        sc = oracle.make_scene(pos2, mat2, alb2)
        l0, want_alb, want_nrm = oracle.voxelize_conservative_attr(p, sc)
        chain0 = oracle.build_mips(l0)
        assert np.array_equal(ctx.download_chain(), chain0)
        want_l1, want_steps = oracle.bounce(p, chain0, want_alb, want_nrm, nthreads=8)
        ctx.bounce()
        assert ctx.last_step_count() == want_steps
        assert np.array_equal(ctx.download_chain(), oracle.build_mips(want_l1))


def test_empty_row_range_counts_zero_steps(vct, oracle):
    """a trace over an empty tile-row range (a rank whose slab is empty) launches nothing -- the step
    count it reports is 0 and the next launch does not inherit stale counters."""
    V, w, h = 16, 24, 16
    chain = oracle.build_mips(synth.noise_volume(V, seed=3, occupancy=0.2))
    planes = synth.random_gbuffer(w * h, seed=9)
    with make_ctx(vct, V, w, h) as ctx:
        ctx.upload_chain(chain)
        ctx.trace(planes)
        full = ctx.last_step_count()
        assert full > 0
        for _ in range(3):
            ctx.trace_gbuffer_rows(1, 1)
            assert ctx.last_step_count() == 0
            assert int(ctx.last_row_steps().sum()) == 0
        ctx.trace_gbuffer_rows(0, 2)
        assert ctx.last_step_count() == full
        ctx.trace_gbuffer_rows(2, 2)
        ctx.trace_gbuffer_rows(0, 2)
        assert ctx.last_step_count() == full


def test_shipped_divisor_table_is_verified_entry_by_entry(vct):
    """csrc/vct_divisors.h lets a fresh process skip the 2 ms device check per divisor; EVERY entry must pass that
    check here, and the table must cover the BASELINE grids and apertures (tools/gen_divisor_table.py)."""
    import os
    import re
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    txt = open(os.path.join(root, "voxel-cone-tracing_amd", "csrc", "vct_divisors.h")).read()
    bits = [int(x, 16) for x in re.findall(r"0x([0-9a-f]{8})u", txt)]
    assert bits == sorted(bits) and len(bits) == len(set(bits)) >= 250
    sys.path.insert(0, os.path.join(root, "tools"))
    import gen_divisor_table
    want = {int(np.float32(d).view(np.uint32)) for d in gen_divisor_table.divisors()}
    assert want == set(bits), "regenerate the table: python3 tools/gen_divisor_table.py"
    with make_ctx(vct, 16, 8, 8) as ctx:
        for b in bits:
            d = float(np.uint32(b).view(np.float32))
            assert ctx.selftest_const_divide(d) == 0, hex(b)
