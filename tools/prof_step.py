"""One short run of the C4 workload for rocprofv3 (kernel trace or --pmc passes)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pathtracer_amd import PathTracer, DeviceScene, scenes
mb = int(sys.argv[1]) if len(sys.argv) > 1 else 2
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1
w, h = 3840, 2160
pt = PathTracer(DeviceScene(scenes.grid_c4(205, 204, w, h)), max_bounces=mb)
pt.Init(w, h)
for _ in range(steps):
    pt.render(1)
print(pt.stats())
