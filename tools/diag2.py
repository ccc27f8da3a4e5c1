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
f32=np.float32
with vct.Context(vct.default_config(voxel_dim=V, width=w, height=h, debug_outputs=1)) as ctx:
    ctx.upload_chain(chain)
    ctx.trace(planes)
    c = ctx.cones().reshape(w*h, 28)
P = planes[0:3].T; Nw = planes[3:6].T; N = planes[12:15].T
cam = np.array([0,4,0],f32)
def dot(a,b): return (a[:,0]*b[:,0] + a[:,1]*b[:,1]) + a[:,2]*b[:,2]
def norm(a):
    l = np.sqrt(dot(a,a)); return a / l[:,None]
E = norm(cam[None,:] - P)
I = E * f32(-1)
dd = f32(2) * dot(N, I)
R = I - dd[:,None]*N
Rd = norm(R)
vs = f32(150)/f32(64)
start = P + Nw*vs
for name, ref, got in (("E",E,c[:,0:3]),("Rd",Rd,c[:,3:6]),("start",start,c[:,6:9])):
    bad = (ref.astype(f32).view(np.uint32) != got.view(np.uint32))
    print(name, "mismatch", bad.sum(), "of", bad.size)
    idx = np.argwhere(bad)[:3]
    for i,j in idx: print("  ", i, j, ref[i], got[i], P[i], N[i])
