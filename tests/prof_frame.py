"""Small driver for profiling: render one scene once through the C ABI.  usage: prof_frame.py scene width sx sy [pool]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracing_amd as ra
name, w, sx, sy = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
pool = int(sys.argv[5]) if len(sys.argv) > 5 else 0
ctx = ra.Context(0); s = ra.HostScene(name); ds = ra.DeviceScene(ctx, s.desc_ptr)
cam = s.camera(w, w)
img, st = ds.render(cam, ra.make_opts(sx, sy, pool_paths=pool))
print("samples", st.samples, "closest", st.rays_closest, "shadow", st.rays_shadow, "dev_s", st.seconds_total, "Msamples/s", st.samples / st.seconds_total / 1e6)
