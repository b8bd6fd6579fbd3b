"""Order-free batch trace against the ordered one on the soup (not collected by pytest): bit equality, replayed share, launch time."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import raytracing_amd as ra
n_tris = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
n_rays = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 24
ctx = ra.Context(0); s = ra.HostScene("soup", n_tris); ds = ra.DeviceScene(ctx, s.desc_ptr)
g = torch.Generator(device="cuda"); g.manual_seed(1234)
o = torch.rand((3, n_rays), generator=g, device="cuda") * 100.0
d = torch.randn((3, n_rays), generator=g, device="cuda"); d = d / d.norm(dim=0, keepdim=True)
rays = torch.cat([o, d, torch.full((1, n_rays), float("inf"), device="cuda")], dim=0).contiguous()
res = {}
for mode in ("0", "1"):
    os.environ["RT_AMD_FREE_WALK"] = mode
    hits = torch.empty((4, n_rays), dtype=torch.float32, device="cuda")
    ds.trace_closest_device(n_rays, rays.data_ptr(), hits.data_ptr(), repeats=1)
    st = ds.trace_closest_device(n_rays, rays.data_ptr(), hits.data_ptr(), repeats=5)
    res[mode] = hits.clone()
    print("free walk", mode, "launch %.3f ms" % (st.seconds * 1e3), "replayed", st.rays_replayed, flush=True)
a, b = res["0"].view(torch.int32), res["1"].view(torch.int32)
print("bit-identical:", bool(torch.equal(a, b)), "differing words:", int((a != b).sum()), "hits", int((a[0] >= 0).sum()))
