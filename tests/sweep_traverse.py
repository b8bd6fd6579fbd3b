"""Tuning sweep for the batch trace kernel (run on the GPU box): one scene build, many settings."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import raytracing_amd as ra
n_tris = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
n_rays = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 24
ctx = ra.Context(0)
scene = ra.HostScene("soup", n_tris); ds = ra.DeviceScene(ctx, scene.desc_ptr)
g = torch.Generator(device="cuda"); g.manual_seed(1234)
o = torch.rand((3, n_rays), generator=g, device="cuda") * 100.0
d = torch.randn((3, n_rays), generator=g, device="cuda"); d = d / d.norm(dim=0, keepdim=True)
rays = torch.cat([o, d, torch.full((1, n_rays), float("inf"), device="cuda")], dim=0).contiguous()
hits = torch.empty((4, n_rays), dtype=torch.float32, device="cuda")
def run(tag):
    ds.trace_closest_device(n_rays, rays.data_ptr(), hits.data_ptr(), repeats=1)
    st = ds.trace_closest_device(n_rays, rays.data_ptr(), hits.data_ptr(), repeats=3)
    print("%-28s %8.2f ms  %7.1f Mrays/s" % (tag, st.seconds * 1e3, n_rays / st.seconds / 1e6), flush=True)
for lm in (20,):
    os.environ["RT_AMD_LEAF_MIN"] = str(lm); run("leaf_min=%d" % lm)
os.environ["RT_AMD_LEAF_MIN"] = "20"
for b in (4, 5, 4, 5):
    os.environ["RT_AMD_TRACE_BLOCKS_PER_CU"] = str(b); run("blocks_per_cu=%d" % b)
os.environ["RT_AMD_EXACT_SLABS"] = "1"; run("bpc4 exact slabs")
os.environ["RT_AMD_EXACT_SLABS"] = "0"
