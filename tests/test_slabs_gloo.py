"""Multi-GPU host logic on CPU: screen-tile slab partition + the single gather, world_size 2 and 3
over gloo.  The per-rank tracer here is the oracle (tests may use it as a stand-in; on a GPU box
bench.py / the gpu test drive vct_trace_slab instead) -- what is under test is the sharding code in
voxel-cone-tracing_amd/slabs.py: rank r's rows land bit-identically where a single-process frame
has them."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _load_slabs():
    import importlib.util
    spec = importlib.util.spec_from_file_location(
        "vct_slabs", os.path.join(ROOT, "voxel-cone-tracing_amd", "slabs.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_partition_covers_frame_without_overlap():
    slabs = _load_slabs()
    for h in (8, 21, 72, 1080, 2160):
        for world in (1, 2, 3, 4, 8):
            parts = slabs.partition(h, world)
            assert len(parts) == world
            assert parts[0][0] == 0 and parts[-1][1] == slabs.tile_rows(h)
            for (a0, a1), (b0, b1) in zip(parts, parts[1:]):
                assert a1 == b0 and a0 <= a1
            sizes = {r1 - r0 for r0, r1 in parts if r1 - r0 > 0}
            assert max(sizes) == (slabs.tile_rows(h) + world - 1) // world
    # 1080p over 8 GPUs: 135 tile rows -> 17,17,17,17,17,17,17,16
    assert [r1 - r0 for r0, r1 in slabs.partition(1080, 8)] == [17] * 7 + [16]


def _worker(rank, world, port, w, h, V, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle import pyoracle
    slabs = _load_slabs()
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        chain = pyoracle.build_mips(synth.noise_volume(V, seed=8, occupancy=0.1))
        planes = synth.random_gbuffer(w * h, seed=12, discard_frac=0.1)
        p = pyoracle.default_params(V)
        fg = slabs.FrameGather(h, w, world, rank, "cpu")
        y0, y1 = fg.my_rows()
        if y1 > y0:
            part = pyoracle.trace(p, chain, planes[:, y0 * w:y1 * w])["rgba16f"]
            fg.slab[: y1 - y0] = torch.from_numpy(part.view(np.float16).reshape(y1 - y0, w, 4))
        frame = fg.gather()
        if rank == 0:
            full = pyoracle.trace(p, chain, planes)["rgba16f"].reshape(h, w, 4)
            got = frame.numpy().view(np.uint16)
            q.put(bool(np.array_equal(got, full)))
        else:
            assert frame is None
        dist.barrier()
    finally:
        dist.destroy_process_group()


def _worker_interleaved(rank, world, port, w, h, V, q):
    """Interleaved assignment (tile row r -> rank r % world): every rank traces its rows, packs them back to back, ONE
    gather, the root de-interleaves."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle import pyoracle
    slabs = _load_slabs()
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        chain = pyoracle.build_mips(synth.noise_volume(V, seed=8, occupancy=0.1))
        planes = synth.random_gbuffer(w * h, seed=12, discard_frac=0.1)
        p = pyoracle.default_params(V)
        fg = slabs.FrameGather(h, w, world, rank, "cpu")
        mine = torch.zeros((h, w, 4), dtype=torch.float16)               # this rank's rows at their own places
        for r in slabs.interleaved_rows(h, world, rank):
            y0, y1 = r * 8, min(r * 8 + 8, h)
            part = pyoracle.trace(p, chain, planes[:, y0 * w:y1 * w])["rgba16f"]
            mine[y0:y1] = torch.from_numpy(part.view(np.float16).reshape(y1 - y0, w, 4))
        fg.pack_interleaved(mine)
        frame = fg.gather_interleaved()
        if rank == 0:
            full = pyoracle.trace(p, chain, planes)["rgba16f"].reshape(h, w, 4)
            q.put(bool(np.array_equal(frame.numpy().view(np.uint16), full)))
        else:
            assert frame is None
        dist.barrier()
    finally:
        dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("world,h", [(2, 72), (3, 40), (2, 8)])
def test_slab_gather_equals_single_process_frame(world, h):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, 24, h, 16, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert q.get(timeout=5) is True


@pytest.mark.parametrize("world,h", [(2, 72), (3, 43), (4, 20)])
def test_interleaved_gather_equals_single_process_frame(world, h):
    slabs = _load_slabs()
    rows = [slabs.interleaved_rows(h, world, r) for r in range(world)]
    assert sorted(sum(rows, [])) == list(range(slabs.tile_rows(h)))           # every tile row exactly once
    assert max(len(x) for x in rows) - min(len(x) for x in rows) <= 1          # balanced to one row
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_interleaved, args=(r, world, port, 24, h, 16, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert q.get(timeout=5) is True


def test_weighted_partition_balances_cost_not_rows():
    """vct_slab_partition_weighted (pure host arithmetic in libvct_amd.so): contiguous slabs of near-equal COST."""
    import vctpkg
    vct = vctpkg.load()
    rng = np.random.default_rng(3)
    for rows, world in ((135, 8), (270, 8), (135, 2), (23, 4), (5, 8), (1, 3)):
        cost = rng.integers(0, 10 ** 6, rows).astype(np.uint64)
        cost[: rows // 3] //= 50                      # a cheap "sky" band at the top of the frame
        starts = vct.slab_partition_weighted(cost, world)
        assert starts[0] == 0 and starts[-1] == rows and np.all(np.diff(starts) >= 0)
        per = np.array([cost[a:b].sum() + (b - a) for a, b in zip(starts[:-1], starts[1:])], np.float64)
        target = (cost.sum() + rows) / world
        biggest_row = float(cost.max() + 1)
        assert per.max() <= target + biggest_row        # no slab exceeds its share by more than one row
        if rows >= 8 * world:
            equal = [cost[a:b].sum() for a, b in slabs_equal(rows, world)]
            assert per.max() <= max(equal) + world      # never worse than the equal-rows cut
    flat = vct.slab_partition_weighted(np.full(136, 7, np.uint64), 8)
    assert list(np.diff(flat)) == [17] * 8            # uniform cost -> equal rows


def slabs_equal(rows, world):
    per = (rows + world - 1) // world
    return [(min(r * per, rows), min((r + 1) * per, rows)) for r in range(world)]


def test_bench_self_launcher_fails_loudly_when_a_rank_fails():
    """`python3 bench.py --gpus 2` with no launcher and no GPU: the parent starts two rank processes (before touching
    torch.cuda or the HIP library), every rank exits with "needs a GPU", and the parent kills the rest, prints no
    JSON line and exits non-zero -- the failure path of the self-launcher the driver's SCALE run relies on."""
    import subprocess
    if torch.cuda.is_available():
        pytest.skip("GPU present: the success path is covered by tests/test_gpu_multi.py")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                          "--timeout", "240"], capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert out.returncode != 0
    assert not [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert "all ranks killed" in out.stderr
