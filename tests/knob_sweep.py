"""Times a frame at several values of one environment knob (not collected by pytest).
usage: knob_sweep.py SCENE WIDTH SX KNOB V1,V2,..."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import raytracing_amd as ra
name, w, sx, knob, vals = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], sys.argv[5].split(",")
if name == "mrspheres":
    import mrspheres_bin; mrspheres_bin.ensure(ra.default_assets_dir())
ctx = ra.Context(0); s = ra.HostScene(name); ds = ra.DeviceScene(ctx, s.desc_ptr); cam = s.camera(w, w)
ref = None
for rep in range(2):
    for v in vals:
        os.environ[knob] = v
        best = None
        for _ in range(2):
            img, st = ds.render(cam, ra.make_opts(sx, sx))
            best = st.seconds_total if best is None else min(best, st.seconds_total)
        if ref is None: ref = img
        print("%s=%s  %.4f s  %.1f Msamples/s  same image %s" % (knob, v, best, w * w * sx * sx / best / 1e6, np.array_equal(ref, img)), flush=True)
