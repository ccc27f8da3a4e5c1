"""Screen-space statistics of a scene's triangles for a camera / the light: bounding-box sizes, tile overlaps.
CPU only (numpy); guided the tile-binned raster of round 4 (csrc/vct_raster.hip)."""
import argparse
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import vctpkg  # noqa: E402


def stats(pos, vp, W, H, tile, label, mat=None, alpha_mat=None):
    n = pos.shape[0]
    p = pos.reshape(n, 3, 3).astype(np.float32) * np.float32(0.05)
    m = vp.reshape(4, 4).T.astype(np.float32)          # column-major -> row-major
    hom = np.concatenate([p, np.ones((n, 3, 1), np.float32)], axis=2) @ m.T
    w = hom[..., 3]
    near_ok = (hom[..., 2] + w >= 0).all(axis=1) & (w > 1e-20).all(axis=1)
    clipped = ~near_ok & ((hom[..., 2] + w >= 0).any(axis=1))
    sx = (hom[..., 0] / w * 0.5 + 0.5) * W
    sy = (hom[..., 1] / w * 0.5 + 0.5) * H
    area = (sx[:, 1] - sx[:, 0]) * (sy[:, 2] - sy[:, 0]) - (sx[:, 2] - sx[:, 0]) * (sy[:, 1] - sy[:, 0])
    x0 = np.maximum(0, np.floor(sx.min(axis=1))); x1 = np.minimum(W - 1, np.floor(sx.max(axis=1)))
    y0 = np.maximum(0, np.floor(sy.min(axis=1))); y1 = np.minimum(H - 1, np.floor(sy.max(axis=1)))
    ok = near_ok & (area > 0) & (x1 >= x0) & (y1 >= y0)
    bw = (x1 - x0 + 1)[ok]; bh = (y1 - y0 + 1)[ok]
    box = bw * bh
    tw = (x1[ok] // tile - x0[ok] // tile + 1); th = (y1[ok] // tile - y0[ok] // tile + 1)
    nt = tw * th
    print(f"== {label}: {n} tris, {W}x{H}, tile {tile}")
    print(f"   front-facing on screen: {ok.sum()}  near-clipped (not analysed): {clipped.sum()}")
    if mat is not None and alpha_mat is not None:
        am = (mat == alpha_mat)[ok]
        print(f"   of them alpha-tested material: {am.sum()}, box px {box[am].sum():.3e} (all: {box.sum():.3e})")
    for lim in (1, 4, 16, 64, 256, 1024, 4096, 1 << 30):
        sel = box <= lim
        print(f"   box <= {lim:>10}: {sel.sum():>9} tris, {box[sel].sum():.3e} box px, tile entries {nt[sel].sum():.3e}")
    print(f"   tile entries total {nt.sum():.3e}; tris in 1 tile {np.sum(nt == 1)}, 2: {np.sum(nt == 2)}, 3-4: {np.sum((nt > 2) & (nt <= 4))}, >4: {np.sum(nt > 4)}, >64: {np.sum(nt > 64)}")
    # per-tile counts (bbox binning)
    tx = (W + tile - 1) // tile; ty = (H + tile - 1) // tile
    cnt = np.zeros(tx * ty, np.int64)
    small = nt <= 64
    X0 = (x0[ok] // tile).astype(np.int64); Y0 = (y0[ok] // tile).astype(np.int64)
    TW = tw.astype(np.int64); TH = th.astype(np.int64)
    for dy in range(8):
        for dx in range(8):
            s = small & (TW > dx) & (TH > dy)
            np.add.at(cnt, (Y0[s] + dy) * tx + X0[s] + dx, 1)
    print(f"   tiles {tx*ty}: non-empty {np.sum(cnt > 0)}, mean entries {cnt.mean():.1f}, p50 {np.percentile(cnt, 50):.0f}, "
          f"p90 {np.percentile(cnt, 90):.0f}, p99 {np.percentile(cnt, 99):.0f}, max {cnt.max()} (tris spanning <= 8x8 tiles only)")
    return cnt


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scene", default="bistro")
    ap.add_argument("--width", type=int, default=3840)
    ap.add_argument("--height", type=int, default=2160)
    ap.add_argument("--tile", type=int, default=16)
    a = ap.parse_args()
    sc = importlib.import_module("voxel_cone_tracing_amd.scene") if False else None
    vctpkg.load()
    sc = sys.modules["voxel_cone_tracing_amd.scene"] if "voxel_cone_tracing_amd.scene" in sys.modules else importlib.import_module("voxel_cone_tracing_amd.scene")
    if a.scene == "bistro":
        s = sc.Scene(sc.BISTRO, 1.0, 1234); cam = sc.default_camera(position=(-58.0, -19.0, 1.5), yaw=0.0, pitch=12.0); am = 5
    else:
        s = sc.Scene(sc.ATRIUM, 1.0, 1234); cam = sc.default_camera(position=(-56.0, -9.0, 2.0), yaw=0.0, pitch=8.0); am = None
    vp = sc.camera_view_proj(cam, a.width, a.height)
    stats(s.pos, vp, a.width, a.height, a.tile, f"{a.scene} main draw", s.material, am)
    lvp = sc.light_view_proj((0.0, 1.0, 0.25))
    stats(s.pos, lvp, 4096, 4096, a.tile, f"{a.scene} shadow pass", s.material, am)


if __name__ == "__main__":
    main()
