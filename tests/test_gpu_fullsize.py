"""Full BASELINE size (256^3 chain, 1920x1080 frame) on the GPU, checked through size-independent
properties instead of a full scalar-oracle pass: closed forms on empty / opaque volumes (SURVEY.md 4
KATs), idempotence, slab union == full frame, step-count consistency, and the oracle on a strided
sample of tiles."""
import numpy as np
import pytest

import synth
import vctpkg

pytestmark = pytest.mark.gpu

V, W, H = 256, 1920, 1080


@pytest.fixture(scope="module")
def vct():
    import torch
    assert torch.cuda.is_available()
    return vctpkg.load()


@pytest.fixture(scope="module")
def planes():
    return synth.coherent_gbuffer(W, H)


def test_empty_and_opaque_volume_closed_forms(vct, planes):
    with vct.Context(vct.default_config(voxel_dim=V, width=W, height=H, debug_outputs=1)) as ctx:
        # empty volume: every cone runs to MAX_DISTANCE (7 diffuse / 29 specular steps at 256^3) and
        # returns 0, so colour = ambient*albedo + shadow*cos*albedo + spec*shadow*specColor (trace.fs:201-227)
        out = vct.half_to_float(ctx.trace(planes).reshape(-1, 4))
        steps = ctx.steps()
        assert (steps[:, :6] == 7).all() and (steps[:, 6] == 29).all()
        assert ctx.last_step_count() == W * H * (6 * 7 + 29)
        g = planes.astype(np.float64)
        L = np.array([0.0, 1.0, 0.25]); L /= np.linalg.norm(L)
        N = g[12:15]
        cos_t = np.maximum((N * L[:, None]).sum(0), 0.0)
        E = np.array([0.0, 4.0, 0.0])[:, None] - g[0:3]; E /= np.linalg.norm(E, axis=0)
        R = -L[:, None] - 2.0 * (N * -L[:, None]).sum(0) * N; R /= np.linalg.norm(R, axis=0)
        spec = np.maximum((E * R).sum(0), 0.0) ** 20.0
        want = 0.1 * g[15:18] + g[22] * cos_t * g[15:18] + spec * g[22] * g[19:22]
        assert np.abs(out[:, :3] - want.T).max() < 2e-3
        assert (out[:, 3] == 1.0).all()
        # uniform opaque volume: every cone stops after one step with alpha 1
        l0 = np.empty((V, V, V, 4), np.uint8)
        l0[...] = (40, 80, 160, 255)
        ctx.upload_volume(l0)
        ctx.build_mips()
        ctx.trace(planes)
        assert (ctx.steps() == 1).all()
        cones = ctx.cones()
        assert np.allclose(cones[:, :6, 3], 0.980118, atol=1e-6) and np.allclose(cones[:, 6, 3], 0.982726, atol=1e-6)
        assert np.allclose(cones[..., 0], 40 / 255, atol=1e-6)


def test_idempotence_slabs_and_sampled_oracle(vct, oracle, planes):
    chain = oracle.build_mips(synth.noise_volume(V))
    with vct.Context(vct.default_config(voxel_dim=V, width=W, height=H, debug_outputs=1)) as ctx:
        ctx.upload_chain(chain)
        full = ctx.trace(planes)
        steps = ctx.steps().copy()
        total = ctx.last_step_count()
        assert total == int(steps.astype(np.int64).sum())           # counter bank == per-cone counts
        assert np.array_equal(ctx.trace(planes), full)              # idempotent, bit for bit
        ctx.trace_resident(); ctx.synchronize()
        assert ctx.last_step_count() == total
        parts = np.zeros_like(full)                                  # 8 slabs like an 8-GPU node
        from voxel_cone_tracing_amd import slabs
        seen = 0
        for r0, r1 in slabs.partition(H, 8):
            slab = ctx.trace(planes, rows=(r0, r1))
            parts[r0 * 8:r1 * 8] = slab[r0 * 8:r1 * 8]
            seen += ctx.last_step_count()
        assert np.array_equal(parts, full) and seen == total
        # the scalar oracle on every 97th tile
        tiles = np.arange((H // 8) * (W // 8))[::97]
        ys, xs = np.divmod(np.arange(W * H), W)
        sel = np.isin((ys // 8) * (W // 8) + xs // 8, tiles)
        ref = oracle.trace(oracle.default_params(V), chain, planes[:, sel], nthreads=8)
        assert np.array_equal(steps[sel], ref["steps"])
        assert (full.reshape(-1, 4)[sel] == ref["rgba16f"]).mean() > 0.999
        assert synth.rel_l2(vct.half_to_float(full.reshape(-1, 4)[sel]), ref["rgba32f"]) <= 1e-3
