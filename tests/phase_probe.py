"""Rounds and busy lanes per phase of the persistent-lane frame kernels (diagnostic build with -DRT_PHASE_STATS in RT_AMD_LIB_DIR; not collected by pytest).
usage: phase_probe.py SCENE WIDTH SX"""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import raytracing_amd as ra
from raytracing_amd import capi
name, w, sx = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
if name == "mrspheres":
    import mrspheres_bin; mrspheres_bin.ensure(ra.default_assets_dir())
ctx = ra.Context(0); s = ra.HostScene(name); ds = ra.DeviceScene(ctx, s.desc_ptr); cam = s.camera(w, w)
lib = ctypes.CDLL(os.path.join(capi.LIB_DIR, "librt_amd.so"))
out = (ctypes.c_ulonglong * 12)()
lib.rt_debug_phase(out, 1)
img, st = ds.render(cam, ra.make_opts(sx, sx))
lib.rt_debug_phase(out, 1)
for k, nm in ((0, "closest"), (6, "any")):
    r = [out[k + j] for j in range(6)]
    tot = r[0] + r[2] + r[4]
    print(nm, "rounds walk/leaf/refill %d %d %d (%.3f %.3f %.3f)" % (r[0], r[2], r[4], r[0] / tot, r[2] / tot, r[4] / tot),
          "lanes per round %.1f %.1f %.1f" % (r[1] / max(1, r[0]), r[3] / max(1, r[2]), r[5] / max(1, r[4])), flush=True)
print("rays closest %d shadow %d" % (st.rays_closest, st.rays_shadow))
