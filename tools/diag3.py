import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import synth, vctpkg
from oracle import pyoracle as o
vct = vctpkg.load()
V, w, h = 64, 128, 128
chain = o.build_mips(synth.noise_volume(V))
planes = synth.random_gbuffer(w * h, seed=42)
vs = 150.0/64
for nsteps in (1,2,3,4,5,6,7,8,9,10,12,18):
    md = vs*(nsteps+0.5) if nsteps<8 else None
    # spec dist sequence: vs*(k+1) for k<7 then grows; choose max_distance between
    p = o.default_params(V)
    tab=[]; dist=vs
    while dist<75: 
        tab.append(dist); dist += max(vs, 0.14*dist)
    md = 75.0 if nsteps>=len(tab) else (tab[nsteps-1]+tab[nsteps])/2
    p.max_distance = md
    ref = o.trace(p, chain, planes, nthreads=8, want_cones=True)
    with vct.Context(vct.default_config(voxel_dim=V, width=w, height=h, debug_outputs=1, max_distance=md)) as ctx:
        ctx.upload_chain(chain)
        ctx.trace(planes)
        c = ctx.cones(); st = ctx.steps()
    bad = (c.view(np.uint32) != ref["cones"].view(np.uint32))
    print("spec steps<=", nsteps, "md", round(md,3), "steps eq", np.array_equal(st, ref["steps"]), "max spec steps", ref["steps"][:,6].max(),
          "mismatch by cone", bad.any(axis=2).sum(axis=0))
