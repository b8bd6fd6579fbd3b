"""Knob sweep of the order-free batch trace on the soup (not collected by pytest)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import raytracing_amd as ra
n_tris, n_rays = 10_000_000, 1 << 24
ctx = ra.Context(0); s = ra.HostScene("soup", n_tris); ds = ra.DeviceScene(ctx, s.desc_ptr)
g = torch.Generator(device="cuda"); g.manual_seed(1234)
o = torch.rand((3, n_rays), generator=g, device="cuda") * 100.0
d = torch.randn((3, n_rays), generator=g, device="cuda"); d = d / d.norm(dim=0, keepdim=True)
rays = torch.cat([o, d, torch.full((1, n_rays), float("inf"), device="cuda")], dim=0).contiguous()
hits = torch.empty((4, n_rays), dtype=torch.float32, device="cuda")
for cfg in sys.argv[1:]:
    env = dict(kv.split("=") for kv in cfg.split(",") if kv)
    for k, v in env.items(): os.environ[k] = v
    ds.trace_closest_device(n_rays, rays.data_ptr(), hits.data_ptr(), repeats=1)
    st = ds.trace_closest_device(n_rays, rays.data_ptr(), hits.data_ptr(), repeats=5)
    print("%-60s %.3f ms  replayed %d" % (cfg, st.seconds * 1e3, st.rays_replayed), flush=True)
    for k in env: del os.environ[k]
