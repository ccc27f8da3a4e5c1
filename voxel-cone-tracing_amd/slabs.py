"""Screen-tile slab sharding of one frame across the GPUs of a node (BASELINE.json config 4).

Every pixel's 7 cones read only that pixel's G-buffer entry and the (replicated, read-only) voxel
chain (S/VoxelConeTracing.fs:165-228), so the trace needs no exchange: the frame's 8-pixel tile rows
are cut into `world` contiguous slabs, rank r traces slab r with vct_trace_slab, and ONE gather
assembles the RGBA16F frame on the root.  On ROCm the "nccl" backend is RCCL; the root receives the
other slabs over its direct xGMI links.  The same code runs over gloo with CPU tensors (tests).
"""
import torch
import torch.distributed as dist

TILE = 8


def tile_rows(height):
    return (height + TILE - 1) // TILE


def partition(height, world):
    """[(tile_row0, tile_row1)] per rank: equal-sized slabs of ceil(tile_rows / world) rows (the
    last ranks may get a short or empty slab) -- equal sizes keep the gather a single collective."""
    ty = tile_rows(height)
    per = (ty + world - 1) // world
    out = []
    for r in range(world):
        r0 = min(r * per, ty)
        out.append((r0, min(r0 + per, ty)))
    return out


def slab_pixel_rows(height, world, rank):
    r0, r1 = partition(height, world)[rank]
    return r0 * TILE, min(r1 * TILE, height)


def interleaved_rows(height, world, rank):
    """Tile rows of rank `rank` under the interleaved assignment (row r -> rank r % world)."""
    return list(range(rank, tile_rows(height), world))


class FrameGather:
    """Owns the padded gather buffers.  `slab` tensors are [rows_per_rank*8, width, 4] float16."""

    def __init__(self, height, width, world, rank, device, group=None, root=0):
        self.height, self.width, self.world, self.rank, self.root = height, width, world, rank, root
        self.group = group
        self.rows = ((tile_rows(height) + world - 1) // world) * TILE     # padded pixel rows per rank
        self.slab = torch.zeros((self.rows, width, 4), dtype=torch.float16, device=device)
        self.frame = None
        if rank == root:
            self.frame = torch.zeros((world * self.rows, width, 4), dtype=torch.float16, device=device)

    def my_rows(self):
        """Pixel rows [y0, y1) of the frame this rank produces."""
        return slab_pixel_rows(self.height, self.world, self.rank)

    def pack_interleaved(self, frame_rows):
        """Interleaved assignment: copy this rank's tile rows out of a full-height [height, width, 4] tensor (what a
        strided trace wrote at the rows' own places) into self.slab, back to back."""
        for j, r in enumerate(interleaved_rows(self.height, self.world, self.rank)):
            y0, y1 = r * TILE, min(r * TILE + TILE, self.height)
            self.slab[j * TILE: j * TILE + (y1 - y0)].copy_(frame_rows[y0:y1])

    def gather_interleaved(self, force_collective=False):
        """The same ONE collective as gather(); the root then de-interleaves: tile row j of rank r's slab is tile row
        j * world + r of the frame.  Root returns the [height, width, 4] frame, the others None."""
        if self.gather(force_collective) is None:
            return None
        out = torch.empty((self.height, self.width, 4), dtype=self.frame.dtype, device=self.frame.device)
        for r in range(self.world):
            for j, row in enumerate(interleaved_rows(self.height, self.world, r)):
                y0, y1 = row * TILE, min(row * TILE + TILE, self.height)
                out[y0:y1].copy_(self.frame[r * self.rows + j * TILE: r * self.rows + j * TILE + (y1 - y0)])
        return out

    def gather(self, force_collective=False):
        """One collective: every rank contributes self.slab; root returns the [height,width,4] frame
        (a view of its buffer), the others None.  A single rank only copies unless
        `force_collective` (testing the collective path with a 1-rank group)."""
        if self.world == 1 and not force_collective:
            self.frame[: self.rows].copy_(self.slab)
            return self.frame[: self.height]
        if self.slab.is_cuda and dist.get_backend(self.group) == "gloo":
            # functional testing only (several ranks sharing one GPU cannot use RCCL): stage on the host
            send = self.slab.cpu()
            recv = [torch.empty_like(send) for _ in range(self.world)] if self.rank == self.root else None
            dist.gather(send, recv, dst=self.root, group=self.group)
            if self.rank == self.root:
                for r in range(self.world):
                    self.frame[r * self.rows:(r + 1) * self.rows].copy_(recv[r])
            return self.frame[: self.height] if self.rank == self.root else None
        recv = None
        if self.rank == self.root:
            recv = [self.frame[r * self.rows:(r + 1) * self.rows] for r in range(self.world)]
        dist.gather(self.slab, recv, dst=self.root, group=self.group)
        return self.frame[: self.height] if self.rank == self.root else None
