"""Small driver for profiling: render one scene through the C ABI.  usage: prof_frame.py scene width sx sy [pool] [repeats]
(PROF_HEIGHT in the environment: a frame that is not square)
One warm-up render at 1/4 of the samples (kernel code upload, workspace allocation), then `repeats` timed renders; prints the best."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracing_amd as ra
name, w, sx, sy = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
pool = int(sys.argv[5]) if len(sys.argv) > 5 else 0
reps = int(sys.argv[6]) if len(sys.argv) > 6 else 1
h = int(os.environ.get("PROF_HEIGHT", w))
if name == "mrspheres":
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import mrspheres_bin; mrspheres_bin.ensure(ra.default_assets_dir())
ctx = ra.Context(0); s = ra.HostScene(name); ds = ra.DeviceScene(ctx, s.desc_ptr)
cam = s.camera(w, h)
if reps > 1:
    ds.render_device(cam, ra.make_opts(sx, sy, pool_paths=pool))
best = None
flags = ra.capi.RT_RENDER_TIME_KERNELS if os.environ.get("PROF_TIME") else 0
for _ in range(reps):
    _, st = ds.render_device(cam, ra.make_opts(sx, sy, pool_paths=pool, flags=flags))
    if best is None or st.seconds_total < best.seconds_total:
        best = st
st = best
print("samples", st.samples, "closest", st.rays_closest, "shadow", st.rays_shadow, "dev_s", st.seconds_total, "Msamples/s", st.samples / st.seconds_total / 1e6)
if flags:
    print("  per class [s]: trace %.4f shade %.4f shadow %.4f other %.4f" % (st.seconds_trace, st.seconds_shade, st.seconds_shadow, st.seconds_other))
