import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracing_amd as ra
ctx = ra.Context(0); s = ra.HostScene("cornell"); ds = ra.DeviceScene(ctx, s.desc_ptr)
cam = s.camera(800, 800)
for _ in range(3):
    _, st = ds.render_device(cam, ra.make_opts(32, 32, begin=0, end=128))
print(st.seconds_total)
