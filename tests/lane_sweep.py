"""Render lanes against sample share on the Cornell frame (not collected by pytest): wall milliseconds per (spp, lanes)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracing_amd as ra
ctx = ra.Context(0); s = ra.HostScene(sys.argv[1] if len(sys.argv) > 1 else "cornell"); ds = ra.DeviceScene(ctx, s.desc_ptr)
w = int(sys.argv[2]) if len(sys.argv) > 2 else 800
cam = s.camera(w, w)
for spp in (1024, 512, 256, 128):
    row = []
    for lanes in (1, 2, 3, 4):
        os.environ["RT_AMD_RENDER_LANES"] = str(lanes)
        ds.render_device(cam, ra.make_opts(32, 32, begin=0, end=spp))
        best = 1e9
        for rep in range(3):
            t0 = time.perf_counter(); ds.render_device(cam, ra.make_opts(32, 32, begin=0, end=spp)); best = min(best, time.perf_counter() - t0)
        row.append("%d lanes %.2f ms" % (lanes, best * 1e3))
    print("%4d spp: " % spp + ", ".join(row), flush=True)
