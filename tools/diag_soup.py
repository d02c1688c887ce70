"""Phase split of k_trace<closest> on a triangle soup (diagnostic, GPU; counting build).  Usage: diag_soup.py [triangles]"""
import sys, os, argparse
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from pathtracer_amd import PathTracer
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20_000_000
w, h = 1920, 1080
scene, _ = bench.build_scene(argparse.Namespace(workload="c5", grid=[205, 204], tris=n, width=w, height=h, seed=0x5EED))
pt = PathTracer(scene, max_bounces=2)
pt.Init(w, h)
pt.render(1)
pt.reset_stats(); pt.set_profiling(True, False); pt.render(1); t = pt.stats()
pt.reset_stats(); pt.set_profiling(False, True); pt.render(1); c = pt.stats()
d = c["debug"]
tot = max(d[0] + d[1] + d[2] + d[3], 1)
print(f"extend {t['extend_ms']:.2f} ms, {c['closest_rays']} rays, nodes/ray {c['bvh_node_visits'] / c['closest_rays']:.1f} tris/ray {c['bvh_tri_tests'] / c['closest_rays']:.1f}")
print(f"wave cycles: setup {100 * d[0] / tot:.1f} % node {100 * d[1] / tot:.1f} % drain {100 * d[2] / tot:.1f} % rest {100 * d[3] / tot:.1f} % | node iters {d[4]} lanes/iter {d[5] / max(d[4], 1):.1f} "
      f"cycles/iter {d[1] / max(d[4], 1):.0f} | drain iters {d[6] >> 32} lanes/iter {(d[6] & 0xFFFFFFFF) / max(d[6] >> 32, 1):.1f} cycles/iter {d[2] / max(d[6] >> 32, 1):.0f}")
