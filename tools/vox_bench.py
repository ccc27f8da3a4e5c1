"""ON THE GPU BOX: time the once-per-scene stages (voxelize / resolve / mips) with and without the
shadow map, HIP events on the context stream."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, vctpkg
vct = vctpkg.load()
from voxel_cone_tracing_amd import scene as sc
V = int(sys.argv[1]) if len(sys.argv) > 1 else 256
S = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
s = sc.Scene(sc.ATRIUM, 1.0, 1234)
depth, vp = s.shadow_map((0, 1, 0.25), S)
ctx = vct.Context(vct.default_config(voxel_dim=V, width=64, height=64, shadow_map_size=S))
ctx.upload_triangles(s.pos, s.material, s.albedo)
st = torch.cuda.ExternalStream(ctx.stream())
def ev(): return torch.cuda.Event(enable_timing=True)
for shadow in (False, True):
    ctx.upload_shadow_map(depth if shadow else None, vp)
    best = None
    with torch.cuda.stream(st):
        for _ in range(5):
            e = [ev() for _ in range(4)]
            e[0].record(); ctx.voxelize(); e[1].record(); ctx.inject_light(); e[2].record(); ctx.build_mips(); e[3].record()
            ctx.synchronize()
            t = [e[i].elapsed_time(e[i + 1]) for i in range(3)]
            best = t if best is None else [min(a, b) for a, b in zip(best, t)]
    ch = ctx.download_chain()
    print(f"V={V} shadow={shadow}: voxelize {best[0]:.3f} ms  resolve {best[1]:.3f} ms  mips {best[2]:.3f} ms  occupied L0 {(ch[:V**3,3]>0).mean():.4f}  ntri {s.ntri}")
