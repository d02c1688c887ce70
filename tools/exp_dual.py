"""Experiment: two renderers on one GPU, each owning every other 64x64 tile, stepped from two host threads.
Usage: python tools/exp_dual.py [steps] [samples per step]"""
import os, sys, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import argparse
import bench
from pathtracer_amd import PathTracer

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 1
args = argparse.Namespace(workload="c4", grid=[205, 204], tris=0, width=3840, height=2160, seed=0x5EED)
scene, workload = bench.build_scene(args)
def run(n_renderers):
    pts = []
    for k in range(n_renderers):
        kw = dict(tile_size=64, tile_offset=k, tile_stride=n_renderers) if n_renderers > 1 else {}
        pt = PathTracer(scene, device=0, seed=args.seed, **kw)
        pt.Init(args.width, args.height)
        pts.append(pt)
    for pt in pts: pt.step(spp)
    for pt in pts: pt.sync()
    for pt in pts: pt.reset_stats()
    def work(pt):
        for _ in range(steps): pt.step(spp)
        pt.sync()
    t0 = time.perf_counter()
    th = [threading.Thread(target=work, args=(pt,)) for pt in pts]
    for t in th: t.start()
    for t in th: t.join()
    dt = time.perf_counter() - t0
    rays = sum(pt.stats()["closest_rays"] + pt.stats()["shadow_rays"] for pt in pts)
    print(f"{n_renderers} renderer(s): {rays / dt / 1e6:.1f} Mrays/s, {dt / steps / spp * 1e3:.2f} ms per frame sample", flush=True)
    for pt in pts: pt.close_render()
run(1); run(2); run(3); run(1)
