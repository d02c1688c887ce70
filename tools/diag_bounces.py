"""Per-bounce traversal work and kernel time on the C4 workload (diagnostic, GPU)."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pathtracer_amd import PathTracer, DeviceScene, scenes

w, h = 3840, 2160
nx, ny = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (205, 204)
ds = DeviceScene(scenes.grid_c4(nx, ny, w, h))
prev = None
for mb in (1, 2, 3, 4, 6):
    pt = PathTracer(ds, max_bounces=mb)
    pt.Init(w, h)
    pt.render(1)
    pt.reset_stats(); pt.set_profiling(True, False); pt.render(1); t = pt.stats()
    pt.reset_stats(); pt.set_profiling(False, True); pt.render(1); c = pt.stats()
    cur = dict(rays=c["closest_rays"], nodes=c["bvh_node_visits"], tris=c["bvh_tri_tests"], ext=t["extend_ms"], sh=t["shade_ms"], co=t["connect_ms"],
               srays=c["shadow_rays"], snodes=c["shadow_node_visits"], stris=c["shadow_tri_tests"])
    d = cur if prev is None else {k: cur[k] - prev[k] for k in cur}
    print(f"bounces<{mb}: closest rays {d['rays']:>9d} nodes/ray {d['nodes']/max(d['rays'],1):6.1f} tris/ray {d['tris']/max(d['rays'],1):5.1f} extend {d['ext']:7.2f} ms "
          f"({d['rays']/max(d['ext'],1e-9)/1e3:7.1f} Mrays/s) | shadow rays {d['srays']:>9d} nodes/ray {d['snodes']/max(d['srays'],1):6.1f} tris/ray {d['stris']/max(d['srays'],1):5.1f} "
          f"connect {d['co']:6.2f} ms shade {d['sh']:6.2f} ms", flush=True)
    prev = cur
    pt.close_render()
