import sys, os, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, vctpkg, bench
vct = vctpkg.load()
from voxel_cone_tracing_amd import scene as sc
args = bench.parse()
w, h, V = 1920, 1080, 256
inp = bench.build_inputs(args, vct, sc)
ctx = vct.Context(vct.default_config(voxel_dim=V, width=w, height=h, shadow_map_size=4096))
ctx.set_camera_position(inp["cam"]); ctx.set_light_direction(inp["light"])
ctx.upload_scene(inp["scene"])
ctx.render_shadow_map(inp["light_vp"]); ctx.render_gbuffer(inp["view_proj"])
ctx.voxelize(); ctx.inject_light(); ctx.build_mips()
ctx.trace_gbuffer_rows(0, (h + 7) // 8); ctx.synchronize()
st = torch.cuda.ExternalStream(ctx.stream())
for trial in range(3):
    for K in (20, 50, 200):
        for _ in range(5): ctx.trace_resident()
        torch.cuda.synchronize()
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(K + 1)]
        t0 = time.perf_counter()
        with torch.cuda.stream(st):
            for i in range(K):
                evs[i].record(); ctx.trace_resident()
            evs[K].record()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) * 1e3
        iv = [evs[i].elapsed_time(evs[i + 1]) for i in range(K)]
        print(f"K={K} wall {dt:.3f} ms  = {dt/K*1e3:.1f} us/step; event sum {sum(iv):.3f}; first 6 intervals {[round(x,3) for x in iv[:6]]} last {round(iv[-1],3)} median {sorted(iv)[K//2]:.4f}")
        # without events
        for _ in range(5): ctx.trace_resident()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(K): ctx.trace_resident()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) * 1e3
        print(f"   no events: wall {dt:.3f} ms = {dt/K*1e3:.1f} us/step, issue {(t1-t0)*1e3:.3f} ms")
