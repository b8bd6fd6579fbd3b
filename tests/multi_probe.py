"""Several contexts on ONE GPU sharing the samples of a frame (rt_multi with the device listed k times): wall time of the whole frame.
usage: multi_probe.py scene width sx sy k [k ...]   (not collected by pytest)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracing_amd as ra
name, w, sx, sy = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
s = ra.HostScene(name); cam = s.camera(w, w)
for k in [int(a) for a in sys.argv[5:]]:
    ms = ra.MultiScene([0] * k, s.desc_ptr)
    best = 1e9
    for rep in range(4):
        t0 = time.perf_counter(); img, per, tot = ms.render(cam, ra.make_opts(sx, sy)); dt = time.perf_counter() - t0
        if rep: best = min(best, dt)
    print("contexts %d: %.1f ms wall (slowest device time %.1f ms) -> %.0f Msamples/s incl. copy to host; mean %s" %
          (k, best * 1e3, tot.seconds_total * 1e3, w * w * sx * sy / best / 1e6, img.mean(axis=(0, 1)) / (sx * sy)), flush=True)
    ms.close()
