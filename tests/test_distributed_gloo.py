"""The N>1 path on CPU ranks: sample-split + one reduce (raytracing_amd/distributed.py) with torch.distributed's gloo backend,
world_size 2.  The per-rank render is injected; here it is the oracle (the checker), because the product has no CPU path.
The union of the ranks' sample ranges must reproduce the single-process frame."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_split_samples_is_a_partition():
    from raytracing_amd.distributed import split_samples
    for total in (1, 7, 16, 1024, 1000):
        for world in (1, 2, 3, 4, 8):
            ranges = [split_samples(total, r, world) for r in range(world)]
            assert ranges[0][0] == 0 and ranges[-1][1] == total
            assert all(ranges[i][1] == ranges[i + 1][0] for i in range(world - 1))
            sizes = [e - b for b, e in ranges]
            assert max(sizes) - min(sizes) <= 1
    assert [split_samples(1024, r, 8) for r in range(8)] == [(128 * r, 128 * r + 128) for r in range(8)]
    with pytest.raises(ValueError):
        split_samples(16, 2, 2)
    # the rule, spelled out (the library's rt_split_samples is what both the torchrun path and rt_multi_render call)
    for total, world in ((1000, 8), (5, 8), (0, 3)):
        base, rem = divmod(total, world)
        for r in range(world):
            b = r * base + min(r, rem)
            assert split_samples(total, r, world) == (b, b + base + (1 if r < rem else 0))


def test_multi_device_entry_fails_loudly_without_a_gpu(host_scene):
    import ctypes as C
    import raytracing_amd as ra
    if ra.rt().rt_device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(RuntimeError, match="HIP"):
        ra.MultiScene([0, 1], host_scene("cornell").desc_ptr)


def _worker(rank, world, port, out_path):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import raytracing_amd as ra
    import oracle_lib as ol
    from raytracing_amd.distributed import reduce_frame, split_samples
    scene = ra.HostScene("cornell"); osc = ol.OracleScene(scene.desc_ptr)
    cam = scene.camera(48, 48)
    begin, end = split_samples(16, rank, world)
    part, _ = osc.render(cam, ra.make_opts(4, 4, begin=begin, end=end), mode=1, threads=2)
    frame = torch.from_numpy(part.copy())
    reduce_frame(frame, dist, dst=0)
    if rank == 0:
        np.save(out_path, frame.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sample_split_reproduces_single_process_frame(tmp_path, host_scene, oracle_scene):
    import raytracing_amd as ra
    out = str(tmp_path / "frame.npy")
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    got = np.load(out)
    want, _ = oracle_scene("cornell").render(host_scene("cornell").camera(48, 48), ra.make_opts(4, 4), mode=1)
    assert np.allclose(got, want, rtol=1e-5, atol=1e-6)
    assert got.sum() > 0
