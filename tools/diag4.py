import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import synth, vctpkg
from oracle import pyoracle as o
vct = vctpkg.load()
V, w, h = 64, 128, 128
chain = o.build_mips(synth.noise_volume(V))
for df in (0.0, 0.05):
    planes = synth.random_gbuffer(w * h, seed=42, discard_frac=df)
    p = o.default_params(V)
    ref = o.trace(p, chain, planes, nthreads=8, want_cones=True)
    with vct.Context(vct.default_config(voxel_dim=V, width=w, height=h, debug_outputs=1)) as ctx:
        ctx.upload_chain(chain)
        ctx.trace(planes)
        c = ctx.cones()
    bad = (c.view(np.uint32) != ref["cones"].view(np.uint32)).any(axis=2)   # [npix,7]
    dead = planes[18] < 0.5
    ys, xs = np.divmod(np.arange(w*h), w)
    tile = (ys//8)*(w//8) + xs//8
    dead_tiles = np.unique(tile[dead])
    badpix = bad.any(axis=1)
    print("discard", df, "bad cones by idx", bad.sum(axis=0), "bad pixels", badpix.sum(),
          "of which in tiles with a dead pixel", np.isin(tile[badpix], dead_tiles).sum(),
          "dead tiles", len(dead_tiles), "of", (w//8)*(h//8))
    # lanes: is bad lane index > dead lane index?
    lane = (ys%8)*8 + xs%8
    if badpix.any():
        for t in np.unique(tile[badpix])[:5]:
            print(" tile", t, "dead lanes", lane[(tile==t)&dead], "bad lanes", lane[(tile==t)&badpix])
