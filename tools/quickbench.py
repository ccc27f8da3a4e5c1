import sys, os, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import synth, vctpkg
vct = vctpkg.load()
V = int(sys.argv[1]) if len(sys.argv) > 1 else 256
w, h = 1920, 1080
vol = synth.noise_volume(V)
gbs = {"coherent": synth.coherent_gbuffer(w, h), "random": synth.random_gbuffer(w*h)}
for variant in (0, 1, 2):
    with vct.Context(vct.default_config(voxel_dim=V, width=w, height=h, trace_variant=variant)) as ctx:
        ctx.upload_volume(vol); ctx.build_mips()
        for name, gb in gbs.items():
            ctx.trace(gb)
            steps = ctx.last_step_count()
            ts = []
            for _ in range(5):
                ctx.trace_resident(); ts.append(ctx.last_trace_ms())
            ms = min(ts)
            gbps = (steps*64 + w*h*100)/ (ms*1e-3)/1e9
            print(f"V={V} variant={variant} {name}: {ms:.3f} ms  steps={steps} ({steps/(w*h):.1f}/px)  {w*h*7/ms/1e3:.1f} Mcones/s  algGB/s={gbps:.0f} frac={gbps/8000:.3f}")
