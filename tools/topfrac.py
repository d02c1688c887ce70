import sys, os
sys.path.insert(0, os.getcwd())
from pathtracer_amd import PathTracer, DeviceScene, scenes
w, h = 3840, 2160
ds = DeviceScene(scenes.grid_c4(205, 204, w, h))
for mb in (1, 2, 30):
    pt = PathTracer(ds, max_bounces=mb); pt.Init(w, h)
    pt.set_profiling(False, True); pt.render(1); c = pt.stats()
    print(mb, 'nodes', c['bvh_node_visits'], 'top', c['debug'][7], 'frac', c['debug'][7] / max(c['bvh_node_visits'], 1), 'rays', c['closest_rays'])
    pt.close_render()
