"""Where k_shade's waves spend their time (needs the PT_SHADE_PROBE build: make variant NAME=sprobe DEFS=-DPT_SHADE_PROBE,
MI355PT_LIBRARY=.../libmi355pt_sprobe.so).  Usage: shade_probe.py [workload c1..c4] [samples per step]"""
import sys, os, argparse
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from pathtracer_amd import PathTracer, DeviceScene
wl = sys.argv[1] if len(sys.argv) > 1 else "c4"
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 16
w, h = {"c1": (512, 512), "c2": (1920, 1080)}.get(wl, (3840, 2160))
scene, _ = bench.build_scene(argparse.Namespace(workload=wl, grid=[205, 204], tris=0, width=w, height=h, seed=0x5EED))
pt = PathTracer(DeviceScene(scene) if not isinstance(scene, DeviceScene) else scene)
pt.Init(w, h)
pt.render(spp)
pt.reset_stats()
pt.set_profiling(2, False)
pt.render(spp)
st = pt.stats()
g, s, r, l = st["debug"][0:4]
tot = g + s + r + l
print(f"{wl}: shade {st['shade_ms']:.2f} ms per step; wave time: gather {g / tot:.3f}, path-state loads {l / tot:.3f}, surface+material {s / tot:.3f}, sampling/eval/outputs {r / tot:.3f}; "
      f"vertices {st['shaded_vertices']}")
