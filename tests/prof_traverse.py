"""Profiling driver: the roofline workload of bench.py alone (closest-hit traversal of the triangle soup).
usage: prof_traverse.py [n_tris] [n_rays] [repeats] [bounce batches]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracing_amd as ra
import bench
n_tris = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
n_rays = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 24
rep = int(sys.argv[3]) if len(sys.argv) > 3 else 5
bounces = int(sys.argv[4]) if len(sys.argv) > 4 else 4
ctx = ra.Context(0)
print(json.dumps(bench.traversal_roofline(ctx, n_tris, n_rays, rep, bounces)))
