"""Sample-split data parallelism: every rank renders a contiguous range of the per-pixel sample indices of the whole
frame, then the float sum buffers are combined with ONE reduce (RCCL over xGMI when the backend is nccl).

The render itself is injected (`scene.render_device` on a GPU rank), so the split / reduce logic can be exercised with
torch.distributed's gloo backend on CPU ranks in tests."""
import numpy as np

from . import make_opts


def split_samples(total, rank, world):
    """Contiguous, exhaustive, near-equal partition of sample indices [0, total) — rank r gets [begin, end).  The rule lives in
    the library (rt_split_samples), so that the torchrun path and rt_multi_render share it."""
    from . import split_samples as _split
    return _split(total, rank, world)


def reduce_frame(frame, dist, dst=0):
    """One collective per frame: sum of the per-rank radiance sums lands on rank `dst`."""
    if dist is not None and dist.is_initialized() and dist.get_world_size() > 1:
        if getattr(frame, "is_cuda", False) and dist.get_backend() == "gloo":
            dist.all_reduce(frame, op=dist.ReduceOp.SUM)      # gloo has no reduce for device tensors (one-GPU rehearsals only)
        else:
            dist.reduce(frame, dst=dst, op=dist.ReduceOp.SUM)
    return frame


def render_frame_split(scene, cam, sx, sy, max_depth, rank, world, frame, dist, seed=(0x243F6A88, 0x85A308D3), timings=None):
    """Render this rank's share of the sx*sy samples into `frame` (torch tensor H x W x 3 on the rank's device) and reduce.
    Returns the rank's rt_render_stats.  timings (a dict of lists, optional) receives this step's wall-clock milliseconds of the
    rank's render ("render_ms": rt_render_device returns when the share is rendered) and of the copy + reduce ("reduce_ms",
    which includes waiting for slower ranks); measuring the second costs one device synchronisation per step."""
    import time
    import torch
    begin, end = split_samples(sx * sy, rank, world)
    t0 = time.perf_counter()
    ptr, stats = scene.render_device(cam, make_opts(sx, sy, begin=begin, end=end, max_depth=max_depth, seed=seed))
    t1 = time.perf_counter()
    n = cam.width * cam.height * 3
    # wrap the library-owned device buffer without copying through the host
    src = _as_tensor(ptr, n, frame.device)
    frame.view(-1).copy_(src)
    reduce_frame(frame, dist)
    if timings is not None:
        if getattr(frame, "is_cuda", False):
            torch.cuda.synchronize()
        timings.setdefault("render_ms", []).append((t1 - t0) * 1e3)
        timings.setdefault("reduce_ms", []).append((time.perf_counter() - t1) * 1e3)
    return stats


class _DevArray:
    """__cuda_array_interface__ view of a raw device pointer (float32, contiguous)."""

    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": "<f4", "data": (int(ptr), False), "version": 2}


def _as_tensor(ptr, n, device):
    import torch
    return torch.as_tensor(_DevArray(ptr, n), device=device)
