"""Scratch GPU check (not collected by pytest): traversal + tiny render against the oracle."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import raytracing_amd as ra
import oracle_lib as ol

def rand_rays(n, lo, hi, seed):
    rng = np.random.default_rng(seed)
    o = rng.uniform(lo, hi, (n, 3)).astype(np.float32)
    d = rng.normal(size=(n, 3)).astype(np.float32); d /= np.linalg.norm(d, axis=1, keepdims=True)
    return o, d.astype(np.float32), np.full(n, np.inf, np.float32)

ctx = ra.Context(0)
for name, lo, hi in (("cornell", -1, 6), ("sphere1", -12, 12)):
    s = ra.HostScene(name); osc = ol.OracleScene(s.desc_ptr); ds = ra.DeviceScene(ctx, s.desc_ptr)
    o, d, tm = rand_rays(200000, lo, hi, 1)
    t0 = time.time(); p1, t1, b1 = ds.trace_closest(o, d, tm); tg = time.time() - t0
    p2, t2, b2, nn, nt = osc.trace_closest(o, d, tm)
    print(name, "closest prim equal:", (p1 == p2).all(), "mismatch", int((p1 != p2).sum()), "t bit-equal:", (t1.view(np.uint32) == t2.view(np.uint32)).all(),
          "b max diff", float(np.abs(b1 - b2)[p1 >= 0].max()), "hit frac", float((p1 >= 0).mean()), "gpu s", tg)
    tm2 = np.full(o.shape[0], 3.0, np.float32)
    h1 = ds.trace_any(o, d, tm2); h2 = osc.trace_any(o, d, tm2)
    print(name, "any equal:", (h1 == h2).all(), "occluded frac", float(h1.mean()))
    cam = s.camera(128, 128)
    opts = ra.make_opts(4, 4)
    t0 = time.time(); img, st = ds.render(cam, opts); tg = time.time() - t0
    ref, cnt = osc.render(cam, opts, mode=1)
    print(name, "render gpu s", tg, "dev s", st.seconds_total, "samples", st.samples, "closest", st.rays_closest, cnt["closest"], "shadow", st.rays_shadow, cnt["shadow"])
    print(name, "means gpu", img.mean(axis=(0, 1)) / 16, "oracle", ref.mean(axis=(0, 1)) / 16)
    diff = np.abs(img - ref)
    print(name, "pixels equal", float((diff.max(axis=2) == 0).mean()), "pixels within 1e-3 rel", float((diff.max(axis=2) <= 1e-3 * np.maximum(1, np.abs(ref).max(axis=2))).mean()))
    ds.close(); osc.close()
ctx.close()
