"""Strong-scaling rehearsal on one GPU: time the share of the 1024-spp Cornell frame that one of N ranks renders (not collected by pytest)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracing_amd as ra
ctx = ra.Context(0); s = ra.HostScene(sys.argv[1] if len(sys.argv) > 1 else "cornell"); ds = ra.DeviceScene(ctx, s.desc_ptr)
w = int(sys.argv[2]) if len(sys.argv) > 2 else 800
cam = s.camera(w, w)
ds.render_device(cam, ra.make_opts(32, 32, begin=0, end=128))
t1 = None
for n in (1, 2, 4, 8):
    spp = 1024 // n
    best = 1e9
    for rep in range(3):
        t0 = time.perf_counter(); _, st = ds.render_device(cam, ra.make_opts(32, 32, begin=0, end=spp)); dt = time.perf_counter() - t0
        best = min(best, dt)
    t1 = t1 or best
    print("ranks %d: %4d spp per rank, %.2f ms wall (device %.2f ms) -> %.0f Msamples/s whole job if every rank does this, efficiency %.3f" %
          (n, spp, best * 1e3, st.seconds_total * 1e3, w * w * 1024 / best / 1e6, t1 / (n * best)), flush=True)
