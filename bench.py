#!/usr/bin/env python3
"""Headline benchmark: Msamples/s, Cornell box 800x800 @ 1024 spp, sample-split across N MI355X with one RCCL reduce.

  python bench.py --gpus N --steps K --warmup W            (N=1)
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

One step = one full frame: rank r renders samples rt_split_samples(r, N) of every pixel through the C ABI (rt_render_device),
then the float sum buffers are reduced to rank 0 with one RCCL reduce.  Rank 0 prints ONE JSON line.  Besides the frame
throughput it carries
  roofline        the closest-hit BVH traversal kernel on the 10M-triangle soup (the HBM-bound configuration): algorithmic bytes
                  per launch / HIP-event launch time, measured live here (N = 1, or --roofline);
  roofline_frame  the frame's own bound, which is VALU issue, not HBM (the Cornell scene lives in LDS): per-class kernel times
                  measured live (HIP events around every launch of one extra, untimed frame) and the VALU counters of the same
                  frame from the committed PMC summary under profiles/;
  cpu_baseline    the reference renderer itself (its sources compiled into oracle/_ref/ref_driver) timed on this host's cores on a
                  bounded sample of the same workload, on all cores and on 8 (the reference's own thread_count).
Other workloads: --scene helmet | spheres2 renders that configuration's frame instead (same split and reduce);
--workload soup shards the soup's ray batch across the ranks (no collective on the data path) and reports Mrays/s.
"""
import argparse
import glob
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

MAX_DEPTH = 50        # RayTracing.cpp:431
HBM_PEAK_GBS = 8000.0
# scene -> (width, height, strata x, strata y): the configurations of BASELINE.json.  StratifiedSampler(x, y) gives x*y samples per
# pixel (RayTracing.cpp:428-429 passes spp for both; 512 spp is 32 x 16)
FRAMES = {"cornell": (800, 800, 32, 32), "helmet": (1440, 1440, 32, 32), "spheres2": (1440, 810, 32, 16)}
FRAME_TEXT = {"cornell": "Cornell box 800x800, 1024 spp (32x32 strata)", "helmet": "DamagedHelmet glTF 1440x1440, 1024 spp (32x32 strata)",
              "spheres2": "analytic spheres (matte / mirror / glass + earth texture) 1440x810, 512 spp (32x16 strata)"}


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


def soup_rays(n_rays, seed=1234):
    g = torch.Generator(device="cuda"); g.manual_seed(seed)
    o = torch.rand((3, n_rays), generator=g, device="cuda") * 100.0
    d = torch.randn((3, n_rays), generator=g, device="cuda"); d = d / d.norm(dim=0, keepdim=True)
    tmax = torch.full((1, n_rays), float("inf"), device="cuda")
    return torch.cat([o, d, tmax], dim=0).contiguous(), g


def traversal_roofline(ctx, n_tris, n_rays, repeats, bounces=4, ray_slice=None):
    """Closest-hit traversal of incoherent rays through the N-triangle soup; rays resident in HBM (SoA).  The headline
    figure is the first batch (uniform origins and directions); `bounce_batches` follows the rays through up to four
    cosine-hemisphere bounces.  ray_slice = (begin, end): this rank's part of the batch (the soup workload across ranks)."""
    scene = ra.HostScene("soup", n_tris)
    ds = ra.DeviceScene(ctx, scene.desc_ptr)
    rays, g = soup_rays(n_rays)
    if ray_slice is not None:
        rays = rays[:, ray_slice[0]:ray_slice[1]].contiguous()
        n_rays = rays.shape[1]

    def measure(rays, reps):
        n = rays.shape[1]
        hits = torch.empty((4, n), dtype=torch.float32, device="cuda")
        torch.cuda.synchronize()
        st = ds.trace_closest_device(n, rays.data_ptr(), hits.data_ptr(), repeats=1, count_work=True)     # counts + warm-up
        nodes, tests = st.nodes_visited, st.prims_tested
        st = ds.trace_closest_device(n, rays.data_ptr(), hits.data_ptr(), repeats=reps)
        # algorithmic bytes: 32 B per node slab-tested + 36 B per triangle tested + 32 B ray in + 16 B hit out
        nbytes = 32.0 * nodes + 36.0 * tests + 48.0 * n
        return hits, nodes, tests, nbytes, st.seconds, int(st.rays_replayed)

    hits, nodes, tests, bytes_per_launch, seconds, replayed = measure(rays, repeats)
    achieved = bytes_per_launch / seconds / 1e9
    prim = hits[0].view(torch.int32)
    # HBM traffic per launch comes from PMC passes (rocprofv3 cannot wrap its own caller): profiles/*_traverse_traffic.json holds
    # the counters of the same kernel on the same workload; reported only when the workload matches, else null.
    traffic = None; traffic_src = None; traffic_kernel = None; stale = None
    try:
        sha = lib_sha16()
        for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_traverse_traffic.json"))):
            rec = json.load(open(f))
            if rec["workload"] == {"n_tris": n_tris, "n_rays": n_rays}:
                if rec.get("lib_sha16") == sha:      # counters of the very library that is running
                    traffic = float(rec["traffic_bytes_per_launch"]); traffic_src = os.path.relpath(f, ROOT); traffic_kernel = rec.get("kernel")
                else:
                    stale = os.path.relpath(f, ROOT)
    except Exception:
        traffic = None
    # The timed launch is the order-free walk over eight-wide records (three tree levels per 128-byte line) including the second, ordered walk of the few
    # near-ties; the ALGORITHMIC bytes are those of the reference's own walk (32 B per node it slab-tests, counted by the ordered counting
    # kernel), so achieved = algorithmic bytes / time can exceed what the memory system moves: `traffic` (PMC) is the real figure.
    out = {"bound": "hbm", "kernel": traffic_kernel or "k_trace_pairs (closest hit; order-free over eight-wide quantised records where the scene has them, near-ties walked again in the reference's order inside the launch)",
           "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
           "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_unit": "bytes per launch (PMC, %s)" % traffic_src if traffic else None,
           "traffic_note": None if traffic else (("%s was measured on another build of librt_amd.so" % stale) if stale else "no PMC summary for this workload"),
           "real_hbm_gbs": round(traffic / seconds / 1e9, 1) if traffic else None,
           "real_hbm_frac": round(traffic / seconds / 1e9 / HBM_PEAK_GBS, 4) if traffic else None,
           "traffic_over_algorithmic": round(traffic / bytes_per_launch, 4) if traffic else None,
           "frac_note": "achieved counts the bytes of the reference's own walk (32 B per node it slab-tests, 36 B per triangle it tests); the timed kernel reaches the same "
                        "answers through three tree levels per 128-byte line and fetches traffic_over_algorithmic of those bytes, so frac may pass 1 while real_hbm_frac stays below it",
           "algorithmic_bytes_per_launch": bytes_per_launch,
           "workload": "%d-triangle soup, %d incoherent rays" % (n_tris, n_rays),
           "bytes_per_ray": round(bytes_per_launch / n_rays, 1), "nodes_per_ray": round(nodes / n_rays, 2), "tris_per_ray": round(tests / n_rays, 2),
           "launch_ms": round(seconds * 1e3, 3), "mrays_per_s": round(n_rays / seconds / 1e6, 1),
           "hit_fraction": round(float((prim >= 0).float().mean().item()), 4),
           "rays_replayed_in_reference_order": replayed}
    tot_rays, tot_b, tot_s = n_rays, bytes_per_launch, seconds
    if bounces > 0:
        tri_p = torch.from_numpy(scene.tri_p()).to("cuda")
        batches = [{"bounce": 0, "rays": n_rays, "achieved": round(achieved, 1), "launch_ms": round(seconds * 1e3, 3)}]
        for b in range(1, bounces + 1):
            rays = _next_bounce_rays(rays, hits, tri_p, g)
            if rays is None:
                break
            hits, nodes, tests, nb, sec, _ = measure(rays, max(1, repeats // 2))
            batches.append({"bounce": b, "rays": int(rays.shape[1]), "achieved": round(nb / sec / 1e9, 1), "launch_ms": round(sec * 1e3, 3),
                            "nodes_per_ray": round(nodes / rays.shape[1], 1)})
            tot_b += nb; tot_s += sec; tot_rays += int(rays.shape[1])
        out["bounce_batches"] = batches
        out["all_bounces"] = {"achieved": round(tot_b / tot_s / 1e9, 1), "frac": round(tot_b / tot_s / 1e9 / HBM_PEAK_GBS, 4)}
        del tri_p
    out["_totals"] = (tot_rays, tot_b, tot_s)
    ds.close(); scene.close()
    return out


def host_cores():
    """Cores this process may use: the scheduler affinity, cut down to the cgroup's CPU quota when there is one."""
    info = {"host_cpu_count": os.cpu_count() or 1}
    try:
        info["affinity"] = len(os.sched_getaffinity(0))
    except Exception:
        info["affinity"] = info["host_cpu_count"]
    cores = info["affinity"]
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            info["cgroup_cpu_quota"] = round(int(quota) / int(period), 2)
            cores = max(1, min(cores, int(int(quota) / int(period) + 0.5)))
    except Exception:
        pass
    info["cores_used"] = cores
    return cores, info


def _time_reference(scene, name, W, H, procs, budget_s):
    """The reference's own code (its core/*.cpp compiled from source into oracle/_ref/ref_driver, see oracle/Makefile) rendering
    the same scene on the host cores: processes instead of threads, because its integrator keeps state in globals."""
    import subprocess
    import tempfile
    import ref_io
    driver = os.path.join(ROOT, "oracle", "_ref", "ref_driver")
    if not os.path.exists(driver):
        return None
    tmp = tempfile.mkdtemp(prefix="rt_bench_")
    sbin, cbin, out = (os.path.join(tmp, f) for f in ("scene.bin", "cam.bin", "img.f32"))
    ref_io.dump_scene(scene, sbin, tmp)
    open(cbin, "wb").write(ref_io.cam_spec(name, W, H))

    def run(sx, sy):
        t0 = time.time()
        subprocess.run([driver, "render", sbin, cbin, str(sx), str(sy), str(MAX_DEPTH), "0", str(procs), out], check=True,
                       stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        return time.time() - t0

    try:
        dt = run(1, 1); spp = 1
        for sx, sy in ((16, 16), (16, 8), (8, 8), (8, 4), (4, 4), (4, 2), (2, 2)):
            if dt * sx * sy <= budget_s:
                spp = sx * sy; dt = run(sx, sy)
                break
    except Exception:
        return None
    return {"value": round(W * H * spp / dt / 1e6, 4), "unit": "Msamples/s", "cores": procs, "kind": "reference",
            "sample": "%s %dx%d @ %d spp (%.1f s) by the reference's own sources compiled here (oracle/_ref/ref_driver), %d processes" % (name, W, H, spp, dt, procs)}


def cpu_baseline(scene, name, W, H):
    """CPU side of the comparison on this host's cores, the same frame at a few spp (cost is linear in spp): the reference itself
    when its build is present — on every core this process may use and on 8 (its own thread_count, integrator.cpp:311) — else the
    oracle's restatement of its recursive per-pixel renderer (which is also timed, on a smaller sample, as a cross-check)."""
    import oracle_lib as ol
    cores, info = host_cores()
    ref = _time_reference(scene, name, W, H, cores, 18.0)
    ref8 = _time_reference(scene, name, W, H, 8, 10.0) if ref and cores != 8 else None
    osc = ol.OracleScene(scene.desc_ptr)
    cam = scene.camera(W, H)
    t0 = time.time()
    osc.render(cam, ra.make_opts(1, 1), mode=0, threads=cores)
    dt = time.time() - t0
    spp = 1
    budget = 6.0 if ref else 25.0           # only the cross-check when the reference itself was timed
    for sx, sy in ((16, 16), (16, 8), (8, 8), (8, 4), (4, 4), (4, 2), (2, 2)):
        if dt * sx * sy <= budget:
            spp = sx * sy
            t0 = time.time()
            osc.render(cam, ra.make_opts(sx, sy), mode=0, threads=cores)
            dt = time.time() - t0
            break
    osc.close()
    port = {"value": round(W * H * spp / dt / 1e6, 4), "unit": "Msamples/s", "cores": cores, "kind": "port",
            "sample": "%s %dx%d @ %d spp (%.1f s); CPU restatement of the reference's recursive renderer, bit-exact with it" % (name, W, H, spp, dt)}
    out = ref if ref else port
    if ref:
        out["port_cross_check"] = port
        if ref8:
            out["eight_processes"] = {"value": ref8["value"], "unit": "Msamples/s", "sample": ref8["sample"]}
    out["host"] = info
    return out


def lib_sha16():
    """First 16 hex digits of the sha-256 of the HIP library that is loaded (what tests/pmc_frame_json.py stamps its summaries with)."""
    import hashlib
    p = os.path.join(ROOT, "raytracing_amd", "lib", "librt_amd.so")
    return hashlib.sha256(open(p, "rb").read()).hexdigest()[:16] if os.path.exists(p) else None


def frame_roofline(ds, cam, sx, sy, name, W, H, n_paths_per_s, stats):
    """What bounds the frame.  Scenes that fit LDS (Cornell) are bound by VALU issue: the fraction reported is the share of the
    chip's VALU lane-slots that did useful work = VALU busy x lane utilisation, both from the PMC summary of the same frame
    committed under profiles/ (rocprofv3 cannot wrap its own caller); the per-class kernel times are measured here, on one lane (the events
    that bracket every launch serialise them), while the timed frames run two or three lanes at once."""
    _, st = ds.render_device(cam, ra.make_opts(sx, sy, max_depth=MAX_DEPTH, flags=ra.capi.RT_RENDER_TIME_KERNELS))
    parts = {"trace": st.seconds_trace, "shade": st.seconds_shade, "shadow": st.seconds_shadow, "other": st.seconds_other}
    tot = sum(parts.values()) or 1.0
    out = {"bound": "valu", "workload": "%s %dx%d, %d spp" % (name, W, H, sx * sy),
           "kernel_class_share": {k: round(v / tot, 4) for k, v in parts.items()},
           "kernel_class_seconds": {k: round(v, 5) for k, v in parts.items()},
           "what": {"trace": "closest-hit kernels (k_extend*)", "shade": "shading kernels (on LDS scenes they trace the shadow ray too)",
                    "shadow": "any-hit kernels", "other": "ray generation, late-bounce kernel, accumulation"},
           "grays_per_s": round((stats["rays_closest"] + stats["rays_shadow"]) / stats["seconds"] / 1e9, 3),
           "rays_per_sample": round((stats["rays_closest"] + stats["rays_shadow"]) / max(1, stats["samples"]), 3)}
    pmc = None; stale = None
    sha = lib_sha16()
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_frame_pmc_%s.json" % name))):
        try:
            rec = json.load(open(f))
            if rec.get("scene") == name and rec.get("width") == W and rec.get("height") == H:
                if rec.get("lib_sha16") == sha:
                    pmc = rec; pmc_src = os.path.relpath(f, ROOT)
                else:
                    stale = os.path.relpath(f, ROOT)
        except Exception:
            pass
    if pmc:
        # DERIVED, not counted in this run.  The counters come from a run in which the launches are serialised (rocprofv3 counts one dispatch at
        # a time) over a sample of the frame's spp; the timed frame overlaps two or three lanes, each on its own stream: it issues the same
        # instructions in less time, so its share of useful lane-slots is the serialised share times (serialised kernel time / timed frame
        # time).  The serialised time is the sum of the dispatch durations of the PMC run itself, scaled from its spp to the frame's; where
        # the counter file has no timestamps, the event-bracketed sum measured above stands in (slightly larger: it includes launch gaps).
        # Quoted only when the PMC summary was made with the very library that is running (sha-256 of librt_amd.so).
        serial = pmc["frame"]["useful_lane_fraction"]
        ser_s = pmc.get("serial_kernel_seconds")
        if ser_s and pmc.get("sample_spp"):
            ser_s = ser_s * (sx * sy) / pmc["sample_spp"]; ser_src = "sum of the PMC run's dispatch durations, scaled to the frame's spp"
        else:
            ser_s = tot; ser_src = "HIP events around every launch of one extra frame (includes launch gaps)"
        overlap = ser_s / stats["seconds"] if stats["seconds"] > 0 else 1.0
        frac = min(1.0, serial * overlap)
        out.update({"achieved": round(frac, 4), "peak": 1.0, "unit": "fraction of VALU lane-slots doing useful work", "frac": round(frac, 4), "derived": True,
                    "serialised": {"useful_lane_fraction": serial, "valu_busy": pmc["frame"]["valu_busy"],
                                   "lane_utilisation": pmc["frame"]["lane_utilisation"], "frame_seconds": round(ser_s, 5), "frame_seconds_from": ser_src},
                    "timed_frame_seconds": round(stats["seconds"], 5), "overlap_factor": round(overlap, 4),
                    "derivation": "frac = serialised.useful_lane_fraction (PMC, one dispatch at a time) x serialised.frame_seconds / timed_frame_seconds (render lanes overlapped), capped at 1",
                    "per_kernel": pmc["kernels"], "pmc_source": pmc_src, "pmc_sample": pmc.get("sample"), "lib_sha16": sha})
    else:
        out.update({"achieved": None, "peak": 1.0, "unit": "fraction of VALU lane-slots doing useful work", "frac": None,
                    "why_null": ("the PMC summary %s was made with another build of librt_amd.so" % stale) if stale else "no PMC summary of this frame under profiles/"})
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--scene", choices=sorted(FRAMES), default="cornell", help="frame configuration (default: the headline metric's)")
    ap.add_argument("--workload", choices=["frame", "soup"], default="frame")
    ap.add_argument("--soup-tris", type=int, default=10_000_000)
    ap.add_argument("--soup-rays", type=int, default=1 << 24)
    ap.add_argument("--soup-bounces", type=int, default=4)
    ap.add_argument("--roofline", action="store_true", help="run the soup roofline leg on rank 0 even when there are several ranks")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-cpu", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU: the path has no CPU implementation")
    # Rehearsal of the multi-rank path on a box with one GPU: RT_BENCH_ONE_DEVICE=1 puts every rank on device 0 and RT_BENCH_BACKEND=gloo
    # replaces RCCL (which refuses two ranks on one device); the numbers of such a run mean nothing, the code path is the same.
    if os.environ.get("RT_BENCH_ONE_DEVICE") == "1":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        backend = os.environ.get("RT_BENCH_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=backend)
    from raytracing_amd.distributed import render_frame_split, split_samples

    def fence():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def reduce_max(x):
        if dist is None:
            return x
        t = torch.tensor([x], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def reduce_sum(xs):
        if dist is None:
            return [float(x) for x in xs]
        t = torch.tensor(list(xs), dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return [float(v) for v in t.tolist()]

    ctx = ra.Context(local_rank)

    if args.workload == "soup":
        # the ray batch is sharded across the ranks (every rank holds the whole tree); no collective on the data path
        b, e = split_samples(args.soup_rays, rank, world)
        fence()
        t0 = time.perf_counter()
        r = traversal_roofline(ctx, args.soup_tris, args.soup_rays, max(1, args.steps), args.soup_bounces, ray_slice=(b, e))
        rays, nbytes, secs = r.pop("_totals")
        fence()
        secs = reduce_max(secs)
        rays, nbytes = reduce_sum([rays, nbytes])
        if rank == 0:
            print(json.dumps({"metric": "Mrays/s, closest-hit traversal, %d-triangle soup, incoherent %d-bounce diffuse rays" % (args.soup_tris, args.soup_bounces),
                              "value": round(rays / secs / 1e6, 1), "unit": "Mrays/s", "n_gpus": world, "steps": max(1, args.steps), "warmup": 1,
                              "ms_per_step": round(secs * 1e3, 3), "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32",
                              "data": "synthetic", "config": {"workload": "soup: %d rays of bounce 0 sharded across ranks, tree replicated" % args.soup_rays,
                                                              "parallelism": "ray-split x%d" % world},
                              "aggregate_algorithmic_gbs": round(nbytes / secs / 1e9, 1), "roofline": r}), flush=True)
        if dist is not None:
            dist.barrier(); dist.destroy_process_group()
        ctx.close()
        return

    W, H, SX, SY = FRAMES[args.scene]
    scene = ra.HostScene(args.scene)
    ds = ra.DeviceScene(ctx, scene.desc_ptr)
    cam = scene.camera(W, H)
    frame = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda")

    timings = {}

    def step():
        return render_frame_split(ds, cam, SX, SY, MAX_DEPTH, rank, world, frame, dist, timings=timings if world > 1 else None)

    for _ in range(args.warmup):
        step()
    fence()
    timings.clear()
    t0 = time.perf_counter()
    stats = None
    for _ in range(args.steps):
        stats = step()
    fence()
    dt = reduce_max(time.perf_counter() - t0)
    # several ranks: where a frame's time goes — every rank's own render (mean over the steps) and the copy + reduce that follows it
    per_rank = None
    if world > 1:
        slots = [0.0] * (2 * world)      # (an all-reduce of a vector with this rank's slots filled in: works on every backend)
        slots[2 * rank] = sum(timings["render_ms"]) / len(timings["render_ms"]); slots[2 * rank + 1] = sum(timings["reduce_ms"]) / len(timings["reduce_ms"])
        slots = reduce_sum(slots)
        per_rank = [[round(slots[2 * r], 3), round(slots[2 * r + 1], 3)] for r in range(world)]
    # per-frame counters of the last step, summed over the ranks (each rank traced its own share of the samples)
    s_samples, s_closest, s_shadow = reduce_sum([stats.samples, stats.rays_closest, stats.rays_shadow]) if stats else (0, 0, 0)

    if rank == 0:
        samples = W * H * SX * SY
        img = frame.cpu().numpy() / (SX * SY)
        headline = args.scene == "cornell"
        out = {
            "metric": "Msamples/s, %s" % ("Cornell box 800x800 @1024spp" if headline else FRAME_TEXT[args.scene]),
            "value": round(samples * args.steps / dt / 1e6, 2), "unit": "Msamples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
            "wall_clock_s_per_frame": round(dt / args.steps, 4), "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic (the reference's scene function, built by the host surface)",
            "config": {"workload": "%s, maxDepth 50, samples split across ranks, one RCCL reduce" % FRAME_TEXT[args.scene],
                       "parallelism": "spp-split x%d" % world},
            "mrays_per_s": round((s_closest + s_shadow) * args.steps / dt / 1e6, 1),
            "image_mean_rgb": [round(float(v), 5) for v in img.mean(axis=(0, 1))],
        }
        if per_rank:
            r_ms = [p[0] for p in per_rank]
            out["render_ms_per_rank"] = {"max": max(r_ms), "min": min(r_ms), "all": r_ms,
                                         "what": "wall clock of each rank's own rt_render_device call (its share of the samples), mean over the timed steps"}
            out["reduce_ms"] = {"rank0": per_rank[0][1], "min_over_ranks": min(p[1] for p in per_rank),
                                "what": "copy of the rank's sums into the frame tensor + the one reduce, until the device is idle again; a rank that "
                                        "finished its render early waits here for the others, so the smallest value is the cost of the reduce itself"}
        if world == 1:
            try:
                out["roofline_frame"] = frame_roofline(ds, cam, SX, SY, args.scene, W, H, out["value"] * 1e6,
                                                       {"rays_closest": s_closest, "rays_shadow": s_shadow, "samples": s_samples, "seconds": dt / args.steps})
            except Exception as e:
                out["roofline_frame"] = {"error": str(e)}
        if (world == 1 or args.roofline) and not args.no_roofline:
            try:
                r = traversal_roofline(ctx, args.soup_tris, args.soup_rays, 5, args.soup_bounces)
                r.pop("_totals", None)
                out["roofline"] = r
            except Exception as e:  # the frame number above is still valid
                out["roofline"] = {"error": str(e)}
        if world == 1 and not args.no_cpu:
            out["cpu_baseline"] = cpu_baseline(scene, args.scene, W, H)
            out["speedup_vs_cpu_baseline"] = round(out["value"] / out["cpu_baseline"]["value"], 1)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    ds.close(); ctx.close()


if __name__ == "__main__":
    main()
