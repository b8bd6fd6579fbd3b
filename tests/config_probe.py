"""The BASELINE configurations that fit one GPU, at their full sizes (not collected by pytest): wall clock and Msamples/s."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracing_amd as ra
ctx = ra.Context(0)
for name, w, h, sx, sy in (("spheres2", 1440, 810, 32, 16), ("cornell", 800, 800, 32, 32), ("helmet", 1440, 1440, 32, 32)):
    s = ra.HostScene(name); ds = ra.DeviceScene(ctx, s.desc_ptr); cam = s.camera(w, h)
    ds.render_device(cam, ra.make_opts(sx, sy, begin=0, end=32))            # warm-up: kernel upload, workspace
    best = None
    for _ in range(2):
        t0 = time.perf_counter(); _, st = ds.render_device(cam, ra.make_opts(sx, sy)); dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    print("%-9s %4dx%-4d %4d spp: %.3f s wall, %.1f Msamples/s, %.2f G rays/s" % (name, w, h, sx * sy, best, w * h * sx * sy / best / 1e6,
          (st.rays_closest + st.rays_shadow) / best / 1e9), flush=True)
    ds.close(); s.close()
