"""Order-free closest hits (k_extend_flat) against the ordered walk on the Cornell frame (not collected by pytest): identity, replayed share, time."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import raytracing_amd as ra
ctx = ra.Context(0)
name = sys.argv[1] if len(sys.argv) > 1 else "cornell"
w = int(sys.argv[2]) if len(sys.argv) > 2 else 800
sx = int(sys.argv[3]) if len(sys.argv) > 3 else 16
s = ra.HostScene(name); ds = ra.DeviceScene(ctx, s.desc_ptr); cam = s.camera(w, w)
def run(env, opts, reps=2):
    for k, v in env.items(): os.environ[k] = v
    best = None
    for _ in range(reps):
        t0 = time.perf_counter(); img, st = ds.render(cam, opts); dt = time.perf_counter() - t0
        best = st.seconds_total if best is None else min(best, st.seconds_total)
    for k in env: del os.environ[k]
    return img, st, best
o16 = ra.make_opts(4, 4)
a, sa, _ = run({"RT_AMD_FLAT": "1"}, o16, 1)
b, sb, _ = run({"RT_AMD_FLAT": "0"}, o16, 1)
print("identity 16 spp:", np.array_equal(a, b), "replayed", sa.rays_replayed, "of", sa.rays_closest, "closest-hit rays = %.4f" % (sa.rays_replayed / max(1, sa.rays_closest)), flush=True)
of = ra.make_opts(sx, sx)
n = w * w * sx * sx
for env in ({"RT_AMD_FLAT": "0"}, {"RT_AMD_FLAT": "1"}, {"RT_AMD_FLAT": "0", "RT_AMD_RENDER_LANES": "1"}, {"RT_AMD_FLAT": "1", "RT_AMD_RENDER_LANES": "1"},
            {"RT_AMD_FLAT": "1", "RT_AMD_FLAT_FROM": "0"}):
    img, st, best = run(env, of)
    print(env, "%.4f s  %.1f Msamples/s  replayed %.4f" % (best, n / best / 1e6, st.rays_replayed / max(1, st.rays_closest)), flush=True)
