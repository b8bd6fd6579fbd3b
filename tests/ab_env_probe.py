"""A/B of one environment knob on a frame (not collected by pytest): identity at 16 spp, then time at sx*sx spp.
usage: ab_env_probe.py SCENE WIDTH SX KNOB [HEIGHT]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import raytracing_amd as ra
name, w, sx, knob = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
h = int(sys.argv[5]) if len(sys.argv) > 5 else w
if name == "mrspheres":
    import mrspheres_bin; mrspheres_bin.ensure(ra.default_assets_dir())
ctx = ra.Context(0); s = ra.HostScene(name); ds = ra.DeviceScene(ctx, s.desc_ptr); cam = s.camera(w, h)
def run(val, opts, reps):
    os.environ[knob] = val
    best = None
    for _ in range(reps):
        img, st = ds.render(cam, opts)
        best = st.seconds_total if best is None else min(best, st.seconds_total)
    del os.environ[knob]
    return img, st, best
a, sa, _ = run("1", ra.make_opts(4, 4), 1)
b, sb, _ = run("0", ra.make_opts(4, 4), 1)
print("identity 16 spp:", np.array_equal(a, b), "rays", (sa.rays_closest, sa.rays_shadow) == (sb.rays_closest, sb.rays_shadow), "replayed %d of %d closest-hit rays" % (sa.rays_replayed, sa.rays_closest), flush=True)
n = w * h * sx * sx
for val in ("0", "1", "0", "1"):
    img, st, best = run(val, ra.make_opts(sx, sx), 2)
    print("%s=%s  %.4f s  %.1f Msamples/s  replayed %.5f" % (knob, val, best, n / best / 1e6, st.rays_replayed / max(1, st.rays_closest)), flush=True)
