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


def test_noalloc_entry_refuses_what_it_cannot_hold(device_scene):
    torch = pytest.importorskip("torch")
    ds = device_scene("cornell")
    buf = torch.zeros(64, device="cuda")
    with pytest.raises(RuntimeError, match="2\\^28"):
        ds.trace_closest_device_noalloc((1 << 28) + 1, buf.data_ptr(), buf.data_ptr())
    with pytest.raises(RuntimeError, match="count_work"):
        ds.trace_closest_device_noalloc(4, buf.data_ptr(), buf.data_ptr(), count_work=True, want_stats=False)
