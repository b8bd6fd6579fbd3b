#!/usr/bin/env python3
"""Headline benchmark: Msamples/s, Cornell box 800x800 @ 1024 spp, sample-split across N MI355X with one RCCL reduce.

  python bench.py --gpus N --steps K --warmup W            (N=1)
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

One step = one full 800x800 frame of 1024 samples per pixel: rank r renders samples [r*1024/N, (r+1)*1024/N) of every
pixel through the C ABI (rt_render_device), then the float sum buffers are reduced to rank 0 with one RCCL reduce.
Rank 0 prints ONE JSON line.  Besides the frame throughput it carries
  roofline     — the closest-hit BVH traversal kernel on the 10M-triangle soup (the HBM-bound configuration), algorithmic
                 bytes per launch / HIP-event launch time, measured live here;
  cpu_baseline — the CPU oracle (oracle/rt_oracle.cpp, the restatement of the reference's recursive renderer that is
                 bit-exact with it) timed on this host's cores on a bounded sample of the same workload.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402
import torch  # noqa: E402

import raytracing_amd as ra  # noqa: E402

W = H = 800
SX = SY = 32          # StratifiedSampler(32, 32) = 1024 spp (RayTracing.cpp:428-429 passes spp for both)
MAX_DEPTH = 50        # RayTracing.cpp:431
HBM_PEAK_GBS = 8000.0


def _next_bounce_rays(rays, hits, tri_p, gen):
    """Cosine-hemisphere rays off the geometric normal of every hit (the soup workload's bounces): origin offset as
    OffsetRayOrigin (geometry.h:916-923), normal = normalize(cross(p0-p2, p1-p2)) (triangle.cpp:146) turned towards the
    incoming ray.  Misses drop out.  Pure torch: this only PRODUCES the next batch, it is not part of the timed kernel."""
    prim = hits[0].view(torch.int32)
    keep = (prim >= 0).nonzero(as_tuple=True)[0]
    if keep.numel() == 0:
        return None
    o = rays[0:3, keep]; d = rays[3:6, keep]; t = hits[1, keep]
    P = tri_p[prim[keep].long()]                                   # (m, 3, 3)
    n = torch.cross(P[:, 0] - P[:, 2], P[:, 1] - P[:, 2], dim=1)
    n = (n / n.norm(dim=1, keepdim=True)).t()                      # (3, m)
    flip = (n * d).sum(0) > 0
    nf = torch.where(flip, -n, n)
    p = o + t * d
    u1 = torch.rand(keep.numel(), generator=gen, device=rays.device); u2 = torch.rand(keep.numel(), generator=gen, device=rays.device)
    r = u1.sqrt(); phi = 2 * np.pi * u2
    lx, ly, lz = r * phi.cos(), r * phi.sin(), (1 - u1).clamp_min(0).sqrt()
    a = torch.where(nf[0].abs() > 0.9, torch.tensor([0.0, 1.0, 0.0], device=rays.device)[:, None], torch.tensor([1.0, 0.0, 0.0], device=rays.device)[:, None])
    s_ = torch.cross(nf.t(), a.t(), dim=1).t(); s_ = s_ / s_.norm(dim=0, keepdim=True)
    tt = torch.cross(nf.t(), s_.t(), dim=1).t()
    w = lx * s_ + ly * tt + lz * nf
    off = nf * (1e-4 * nf.abs().sum(0))
    no = p + off                                                   # w lies on nf's side by construction
    tmax = torch.full((1, keep.numel()), float("inf"), device=rays.device)
    return torch.cat([no, w, tmax], dim=0).contiguous()


def traversal_roofline(ctx, n_tris, n_rays, repeats, bounces=4):
    """Closest-hit traversal of incoherent rays through the N-triangle soup; rays resident in HBM (SoA).  The headline
    figure is the first batch (uniform origins and directions); `bounce_batches` follows the rays through up to four
    cosine-hemisphere bounces."""
    scene = ra.HostScene("soup", n_tris)
    ds = ra.DeviceScene(ctx, scene.desc_ptr)
    g = torch.Generator(device="cuda"); g.manual_seed(1234)
    o = torch.rand((3, n_rays), generator=g, device="cuda") * 100.0
    d = torch.randn((3, n_rays), generator=g, device="cuda"); d = d / d.norm(dim=0, keepdim=True)
    tmax = torch.full((1, n_rays), float("inf"), device="cuda")
    rays = torch.cat([o, d, tmax], dim=0).contiguous()
    del o, d, tmax

    def measure(rays, reps):
        n = rays.shape[1]
        hits = torch.empty((4, n), dtype=torch.float32, device="cuda")
        torch.cuda.synchronize()
        st = ds.trace_closest_device(n, rays.data_ptr(), hits.data_ptr(), repeats=1, count_work=True)     # counts + warm-up
        nodes, tests = st.nodes_visited, st.prims_tested
        st = ds.trace_closest_device(n, rays.data_ptr(), hits.data_ptr(), repeats=reps)
        # algorithmic bytes: 32 B per node slab-tested + 36 B per triangle tested + 32 B ray in + 16 B hit out
        nbytes = 32.0 * nodes + 36.0 * tests + 48.0 * n
        return hits, nodes, tests, nbytes, st.seconds

    hits, nodes, tests, bytes_per_launch, seconds = measure(rays, repeats)
    achieved = bytes_per_launch / seconds / 1e9
    prim = hits[0].view(torch.int32)
    # HBM traffic per launch comes from PMC passes (rocprofv3 cannot wrap its own caller): profiles/*_traverse_traffic.json holds
    # the counters of the same kernel on the same workload; reported only when the workload matches, else null.
    traffic = None; traffic_src = None
    try:
        import glob
        for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_traverse_traffic.json"))):
            rec = json.load(open(f))
            if rec["workload"] == {"n_tris": n_tris, "n_rays": n_rays}:
                traffic = float(rec["traffic_bytes_per_launch"]); traffic_src = os.path.relpath(f, ROOT)
    except Exception:
        traffic = None
    out = {"bound": "hbm", "kernel": "k_trace_pairs<closest, quad records>", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
           "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_unit": "bytes per launch (PMC, %s)" % traffic_src if traffic else None,
           "algorithmic_bytes_per_launch": bytes_per_launch,
           "workload": "%d-triangle soup, %d incoherent rays" % (n_tris, n_rays),
           "bytes_per_ray": round(bytes_per_launch / n_rays, 1), "nodes_per_ray": round(nodes / n_rays, 2), "tris_per_ray": round(tests / n_rays, 2),
           "launch_ms": round(seconds * 1e3, 3), "mrays_per_s": round(n_rays / seconds / 1e6, 1),
           "hit_fraction": round(float((prim >= 0).float().mean().item()), 4)}
    if bounces > 0:
        tri_p = torch.from_numpy(scene.tri_p()).to("cuda")
        batches = [{"bounce": 0, "rays": n_rays, "achieved": round(achieved, 1), "launch_ms": round(seconds * 1e3, 3)}]
        tot_b, tot_s = bytes_per_launch, seconds
        for b in range(1, bounces + 1):
            rays = _next_bounce_rays(rays, hits, tri_p, g)
            if rays is None:
                break
            hits, nodes, tests, nb, sec = measure(rays, max(1, repeats // 2))
            batches.append({"bounce": b, "rays": int(rays.shape[1]), "achieved": round(nb / sec / 1e9, 1), "launch_ms": round(sec * 1e3, 3),
                            "nodes_per_ray": round(nodes / rays.shape[1], 1)})
            tot_b += nb; tot_s += sec
        out["bounce_batches"] = batches
        out["all_bounces"] = {"achieved": round(tot_b / tot_s / 1e9, 1), "frac": round(tot_b / tot_s / 1e9 / HBM_PEAK_GBS, 4)}
        del tri_p
    ds.close(); scene.close()
    return out


def _time_reference(scene, cores):
    """The reference's own code (its core/*.cpp compiled from source into oracle/_ref/ref_driver, see oracle/Makefile) rendering
    the same Cornell scene on the host cores: processes instead of threads, because its integrator keeps state in globals."""
    import subprocess
    import tempfile
    import ref_io
    driver = os.path.join(ROOT, "oracle", "_ref", "ref_driver")
    if not os.path.exists(driver):
        return None
    tmp = tempfile.mkdtemp(prefix="rt_bench_")
    sbin, cbin, out = (os.path.join(tmp, f) for f in ("cornell.scene", "cam.bin", "img.f32"))
    ref_io.dump_scene(scene, sbin, tmp)
    open(cbin, "wb").write(ref_io.cam_spec("cornell", W, H))

    def run(sx, sy):
        t0 = time.time()
        subprocess.run([driver, "render", sbin, cbin, str(sx), str(sy), str(MAX_DEPTH), "0", str(cores), out], check=True,
                       stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        return time.time() - t0

    try:
        dt = run(2, 2); spp = 4
        for sx, sy in ((16, 16), (16, 8), (8, 8), (8, 4), (4, 4)):
            if dt / 4 * sx * sy <= 25.0:
                spp = sx * sy; dt = run(sx, sy)
                break
    except Exception:
        return None
    return {"value": round(W * H * spp / dt / 1e6, 4), "unit": "Msamples/s", "cores": cores, "kind": "reference",
            "sample": "Cornell %dx%d @ %d spp (%.1f s) by the reference's own sources compiled here (oracle/_ref/ref_driver), %d processes" % (W, H, spp, dt, cores)}


def cpu_baseline(scene):
    """CPU side of the comparison on this host's cores, Cornell 800x800 at a few spp (cost is linear in spp): the reference
    itself when its build is present, else the oracle's restatement of its recursive per-pixel renderer (which is also timed,
    on a smaller sample, as a cross-check of the two)."""
    import oracle_lib as ol
    cores = min(os.cpu_count() or 1, 16)
    ref = _time_reference(scene, cores)
    osc = ol.OracleScene(scene.desc_ptr)
    cam = scene.camera(W, H)
    t0 = time.time()
    osc.render(cam, ra.make_opts(2, 2), mode=0, threads=cores)
    dt = time.time() - t0
    spp = 4
    # bound the sample: the largest strata grid whose projected time stays under 25 s (8 s when it is only the cross-check)
    budget = 8.0 if ref else 25.0
    for sx, sy in ((16, 16), (16, 8), (8, 8), (8, 4), (4, 4)):
        if dt / 4 * sx * sy <= budget:
            spp = sx * sy
            t0 = time.time()
            osc.render(cam, ra.make_opts(sx, sy), mode=0, threads=cores)
            dt = time.time() - t0
            break
    osc.close()
    port = {"value": round(W * H * spp / dt / 1e6, 4), "unit": "Msamples/s", "cores": cores, "kind": "port",
            "sample": "Cornell %dx%d @ %d spp (%.1f s); CPU restatement of the reference's recursive renderer, bit-exact with it" % (W, H, spp, dt)}
    if ref:
        ref["port_cross_check"] = port
        return ref
    return port


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--soup-tris", type=int, default=10_000_000)
    ap.add_argument("--soup-rays", type=int, default=1 << 24)
    ap.add_argument("--soup-bounces", type=int, default=4)
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-cpu", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU: the path has no CPU implementation")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    from raytracing_amd.distributed import render_frame_split

    ctx = ra.Context(local_rank)
    scene = ra.HostScene("cornell")
    ds = ra.DeviceScene(ctx, scene.desc_ptr)
    cam = scene.camera(W, H)
    frame = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda")

    def step():
        return render_frame_split(ds, cam, SX, SY, MAX_DEPTH, rank, world, frame, dist)

    def fence():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    stats = None
    for _ in range(args.steps):
        stats = step()
    fence()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    if rank == 0:
        samples = W * H * SX * SY
        img = frame.cpu().numpy() / (SX * SY)
        out = {
            "metric": "Msamples/s, Cornell box 800x800 @1024spp", "value": round(samples * args.steps / dt / 1e6, 2), "unit": "Msamples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
            "wall_clock_s_per_frame": round(dt / args.steps, 4), "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic (the reference's Cornellbox() scene, built by the host surface)",
            "config": {"workload": "Cornell box 800x800, 1024 spp (32x32 strata), maxDepth 50, samples split across ranks, one RCCL reduce",
                       "parallelism": "spp-split x%d" % world},
            "mrays_per_s": round((stats.rays_closest + stats.rays_shadow) * world * args.steps / dt / 1e6, 1) if stats else None,
            "image_mean_rgb": [round(float(v), 5) for v in img.mean(axis=(0, 1))],
        }
        if not args.no_roofline:
            try:
                out["roofline"] = traversal_roofline(ctx, args.soup_tris, args.soup_rays, 5, args.soup_bounces)
            except Exception as e:  # the frame number above is still valid
                out["roofline"] = {"error": str(e)}
        if world == 1 and not args.no_cpu:
            out["cpu_baseline"] = cpu_baseline(scene)
            out["speedup_vs_cpu_baseline"] = round(out["value"] / out["cpu_baseline"]["value"], 1)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    ds.close(); ctx.close()


if __name__ == "__main__":
    main()
