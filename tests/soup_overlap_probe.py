"""Does the soup traversal gain from two launches at once?  Two contexts on one GPU trace half of the ray batch each from two host threads
(ctypes releases the GIL); compared with one context tracing the whole batch.  (not collected by pytest)"""
import sys, os, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import raytracing_amd as ra
import bench
n_tris, n_rays = 10_000_000, 1 << 24
scene = ra.HostScene("soup", n_tris)
rays, g = bench.soup_rays(n_rays)
ctxs = [ra.Context(0), ra.Context(0)]
dss = [ra.DeviceScene(c, scene.desc_ptr) for c in ctxs]
hits = torch.empty((4, n_rays), dtype=torch.float32, device="cuda")
torch.cuda.synchronize()
def one():
    st = dss[0].trace_closest_device(n_rays, rays.data_ptr(), hits.data_ptr(), repeats=3)
    return st.seconds
half = n_rays // 2
ra_ = rays[:, :half].contiguous(); rb_ = rays[:, half:].contiguous()
ha = torch.empty((4, half), dtype=torch.float32, device="cuda"); hb = torch.empty((4, half), dtype=torch.float32, device="cuda")
torch.cuda.synchronize()
def two():
    def work(k, r, h):
        dss[k].trace_closest_device(half, r.data_ptr(), h.data_ptr(), repeats=3)
    t0 = time.perf_counter()
    th = [threading.Thread(target=work, args=(0, ra_, ha)), threading.Thread(target=work, args=(1, rb_, hb))]
    [t.start() for t in th]; [t.join() for t in th]
    return (time.perf_counter() - t0) / 3
one(); two()
t0 = time.perf_counter(); s1 = one(); w1 = (time.perf_counter() - t0) / 3
print("one launch of %d rays: %.2f ms (events), %.2f ms wall per repeat" % (n_rays, s1 * 1e3, w1 * 1e3))
print("two launches of %d rays at once: %.2f ms wall per repeat" % (half, two() * 1e3))
