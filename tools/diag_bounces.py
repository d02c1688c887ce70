"""Per-bounce traversal work and kernel time on the C4 workload (diagnostic, GPU)."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pathtracer_amd import PathTracer, DeviceScene, scenes

w, h = 3840, 2160
nx, ny = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (205, 204)
ds = DeviceScene(scenes.grid_c4(nx, ny, w, h))
prev = None
spp = int(os.environ.get("DIAG_SPP", "1"))      # samples per batch (1 = latency-bound launches; 16+ = the bench's regime)
for mb in (1, 2, 3, 4, 6, 30):
    pt = PathTracer(ds, max_bounces=mb)
    pt.Init(w, h)
    pt.render(spp)
    pt.reset_stats(); pt.set_overlap(1 << 30); pt.set_profiling(2, False); pt.render(spp); t = pt.stats(); pt.set_overlap(0)
    pt.reset_stats(); pt.set_profiling(False, True); pt.render(spp); c = pt.stats()
    dbg = c["debug"]
    cur = dict(cs=dbg[0], cn=dbg[1], cd=dbg[2], cr=dbg[3], nit=dbg[4], nact=dbg[5], dit=dbg[6] >> 32, dact=dbg[6] & 0xFFFFFFFF, rays=c["closest_rays"], nodes=c["bvh_node_visits"], tris=c["bvh_tri_tests"], ext=t["extend_ms"], sh=t["shade_ms"], co=t["connect_ms"],
               srays=c["shadow_rays"], snodes=c["shadow_node_visits"], stris=c["shadow_tri_tests"])
    d = cur if prev is None else {k: cur[k] - prev[k] for k in cur}
    print(f"bounces<{mb}: closest rays {d['rays']:>9d} nodes/ray {d['nodes']/max(d['rays'],1):6.1f} tris/ray {d['tris']/max(d['rays'],1):5.1f} extend {d['ext']:7.2f} ms "
          f"({d['rays']/max(d['ext'],1e-9)/1e3:7.1f} Mrays/s) | shadow rays {d['srays']:>9d} nodes/ray {d['snodes']/max(d['srays'],1):6.1f} tris/ray {d['stris']/max(d['srays'],1):5.1f} "
          f"connect {d['co']:6.2f} ms shade {d['sh']:6.2f} ms", flush=True)
    tot = max(d['cs'] + d['cn'] + d['cd'] + d['cr'], 1)
    print(f"      extend wave cycles: setup {100*d['cs']/tot:4.1f} %  node {100*d['cn']/tot:4.1f} %  drain {100*d['cd']/tot:4.1f} %  rest {100*d['cr']/tot:4.1f} % | node iters {d['nit']} lanes/iter {d['nact']/max(d['nit'],1):4.1f} cycles/iter {d['cn']/max(d['nit'],1):5.0f} | drain iters {d['dit']} lanes/iter {d['dact']/max(d['dit'],1):4.1f} cycles/iter {d['cd']/max(d['dit'],1):5.0f}", flush=True)
    prev = cur
    pt.close_render()
