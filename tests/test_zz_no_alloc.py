"""rt_trace_closest_device_noalloc, the entry point for callers that cannot allocate device memory: same bits as
rt_trace_closest_device on the same device-resident rays, on two calls in a row (the second with fewer rays: the static scratch
keeps the first call's contents), the same work counters, and no change in free device memory across the calls."""
import ctypes as C

import numpy as np
import pytest

import raytracing_amd as ra

pytestmark = pytest.mark.gpu


def _rays(rng, n, lo, hi):
    o = rng.uniform(lo, hi, (3, n)).astype(np.float32)
    d = rng.normal(size=(3, n)).astype(np.float32); d /= np.linalg.norm(d, axis=0, keepdims=True)
    d[0, : n // 50] = 0                                           # some axis-parallel rays (inf / NaN slabs)
    tmax = np.full((1, n), np.inf, np.float32); tmax[0, n // 2: n // 2 + n // 10] = rng.uniform(0, 20, n // 10)
    return np.concatenate([o, d, tmax], axis=0)


@pytest.mark.parametrize("name,param,lo,hi", [("soup", 200000, -5, 105), ("cornell", 0, -1, 6), ("sphere1", 0, -12, 12)])
def test_noalloc_entry_matches_the_allocating_one(name, param, lo, hi, device_scene):
    torch = pytest.importorskip("torch")
    ds = device_scene(name, param)
    rng = np.random.default_rng(11)
    for n in (300000, 70001):                                     # second call smaller than the first
        rays = torch.from_numpy(_rays(rng, n, lo, hi)).cuda().contiguous()
        want = torch.empty((4, n), dtype=torch.float32, device="cuda")
        got = torch.full((4, n), -7.0, dtype=torch.float32, device="cuda")
        st_ref = ds.trace_closest_device(n, rays.data_ptr(), want.data_ptr(), repeats=1, count_work=True)
        torch.cuda.synchronize()
        free_before = torch.cuda.mem_get_info()[0]
        st = ds.trace_closest_device_noalloc(n, rays.data_ptr(), got.data_ptr(), count_work=True)
        torch.cuda.synchronize()
        assert torch.cuda.mem_get_info()[0] == free_before        # nothing was allocated or freed on the device
        assert torch.equal(got.view(torch.int32), want.view(torch.int32)), "hits differ in %d words" % int((got.view(torch.int32) != want.view(torch.int32)).sum())
        assert st.nodes_visited == st_ref.nodes_visited and st.prims_tested == st_ref.prims_tested and st.nodes_visited > n
        assert st.seconds > 0
        # without counters and without a stats record
        got.fill_(-7.0)
        torch.cuda.synchronize()
        free_before = torch.cuda.mem_get_info()[0]
        ds.trace_closest_device_noalloc(n, rays.data_ptr(), got.data_ptr(), want_stats=False)
        torch.cuda.synchronize()
        assert torch.cuda.mem_get_info()[0] == free_before
        assert torch.equal(got.view(torch.int32), want.view(torch.int32))
        del rays, want, got
        torch.cuda.synchronize()


def test_noalloc_entry_on_a_deep_tree_with_a_reduced_grid(host_scene, gpu_ctx):
    """A chain-shaped tree of 60 interior levels: the traversal stacks of the full grid no longer fit the 32 MiB of static scratch
    (101 entries per lane), so the entry point launches fewer blocks (162 instead of 1024); same bits as the allocating entry point."""
    torch = pytest.importorskip("torch")
    hs = host_scene("soup", 200)
    pb = ra.prim_bounds(hs.desc_ptr); tri = hs.tri_p()
    n_prims, levels = len(pb), 60
    order = np.argsort(0.5 * (pb[:, 0] + pb[:, 3]), kind="stable").astype(np.int32)
    nodes = np.zeros(2 * levels + 1, ra.BVH_NODE_DTYPE)

    def union(lo, hi):
        b = pb[order[lo:hi]]
        return b[:, :3].min(axis=0), b[:, 3:].max(axis=0)
    for k in range(levels):                                       # interior node 2k: left child = leaf of slot k, right child = the rest
        nodes[2 * k]["bmin"], nodes[2 * k]["bmax"] = union(k, n_prims)
        nodes[2 * k]["offset"] = 2 * k + 2; nodes[2 * k]["nprim"] = 0; nodes[2 * k]["axis"] = 0
        nodes[2 * k + 1]["bmin"], nodes[2 * k + 1]["bmax"] = union(k, k + 1)
        nodes[2 * k + 1]["offset"] = k; nodes[2 * k + 1]["nprim"] = 1
    nodes[2 * levels]["bmin"], nodes[2 * levels]["bmax"] = union(levels, n_prims)
    nodes[2 * levels]["offset"] = levels; nodes[2 * levels]["nprim"] = n_prims - levels
    d = type(hs.desc).from_buffer_copy(hs.desc)
    d.n_nodes = len(nodes); d.bvh_nodes = nodes.ctypes.data_as(C.POINTER(ra.capi.BvhNode)); d.prim_order = order.ctypes.data_as(C.POINTER(C.c_int32))
    ds = ra.DeviceScene(gpu_ctx, C.pointer(d))
    rng = np.random.default_rng(3)
    for n in (300000, 50000):
        r = _rays(rng, n, -5, 105)
        aim = tri[rng.integers(0, n_prims, n // 2)].mean(axis=1).T          # half of the rays point at a triangle's centroid
        r[3:6, : n // 2] = aim - r[0:3, : n // 2]
        rays = torch.from_numpy(r).cuda().contiguous()
        want = torch.empty((4, n), dtype=torch.float32, device="cuda")
        got = torch.full((4, n), -7.0, dtype=torch.float32, device="cuda")
        st_ref = ds.trace_closest_device(n, rays.data_ptr(), want.data_ptr(), repeats=1, count_work=True)
        torch.cuda.synchronize()
        free_before = torch.cuda.mem_get_info()[0]
        st = ds.trace_closest_device_noalloc(n, rays.data_ptr(), got.data_ptr(), count_work=True)
        torch.cuda.synchronize()
        assert torch.cuda.mem_get_info()[0] == free_before
        assert torch.equal(got.view(torch.int32), want.view(torch.int32))
        assert st.nodes_visited == st_ref.nodes_visited and st.prims_tested == st_ref.prims_tested
        assert int((want[0].view(torch.int32) >= 0).sum()) > n // 8
        del rays, want, got
    ds.close()


def test_noalloc_entry_refuses_what_it_cannot_hold(device_scene):
    torch = pytest.importorskip("torch")
    ds = device_scene("cornell")
    buf = torch.zeros(64, device="cuda")
    with pytest.raises(RuntimeError, match="2\\^28"):
        ds.trace_closest_device_noalloc((1 << 28) + 1, buf.data_ptr(), buf.data_ptr())
    with pytest.raises(RuntimeError, match="count_work"):
        ds.trace_closest_device_noalloc(4, buf.data_ptr(), buf.data_ptr(), count_work=True, want_stats=False)
