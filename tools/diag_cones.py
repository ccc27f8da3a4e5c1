import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import synth, vctpkg
from oracle import pyoracle as o
vct = vctpkg.load()
V, w, h = 64, 128, 128
chain = o.build_mips(synth.noise_volume(V))
planes = synth.random_gbuffer(w * h, seed=42, discard_frac=0.05)
p = o.default_params(V)
ref = o.trace(p, chain, planes, nthreads=8, want_cones=True)
with vct.Context(vct.default_config(voxel_dim=V, width=w, height=h, debug_outputs=1)) as ctx:
    ctx.upload_chain(chain)
    out = ctx.trace(planes)
    cones = ctx.cones()
d = cones.view(np.uint32).astype(np.int64) - ref["cones"].view(np.uint32).astype(np.int64)
bad = np.argwhere(d != 0)
print("mismatching elements", len(bad), "of", d.size)
print("by cone index", np.bincount(bad[:, 1], minlength=7))
print("by channel", np.bincount(bad[:, 2], minlength=4))
print("abs ulp pct", np.percentile(np.abs(d[d!=0]),[50,90,99,100]))
for px, c, ch in bad[:10]:
    print(px, c, ch, cones[px, c], ref["cones"][px, c], ref["steps"][px])

# isolate: world-aligned exact frames (dirs need no rounding in TBN) 
g2 = planes.copy()
n = w*h
g2[3:6] = np.array([[0],[0.05],[0]],np.float32); g2[6:9]=np.array([[0.05],[0],[0]],np.float32); g2[9:12]=np.array([[0],[0],[-0.05]],np.float32)
g2[12:15] = np.array([[0],[1],[0]],np.float32)
ref2 = o.trace(p, chain, g2, nthreads=8, want_cones=True)
with vct.Context(vct.default_config(voxel_dim=V, width=w, height=h, debug_outputs=1)) as ctx:
    ctx.upload_chain(chain)
    ctx.trace(g2)
    c2 = ctx.cones()
d2 = c2.view(np.uint32).astype(np.int64) - ref2["cones"].view(np.uint32).astype(np.int64)
bad2 = np.argwhere(d2 != 0)
print("aligned frames: mismatching", len(bad2), "by cone", np.bincount(bad2[:,1], minlength=7) if len(bad2) else None)
