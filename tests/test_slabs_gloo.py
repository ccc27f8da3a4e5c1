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
