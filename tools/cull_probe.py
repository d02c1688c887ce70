"""Sensitivity probe (run by hand: python -m pytest tools/cull_probe.py -s -q): how many of the adversarial rays of
tests/test_gpu_soup_cull.py break round 1 s plain distance cull (MI355PT_CULL_PLAIN) and the default one, under two drain schedules."""
import numpy as np
import pytest
from pathtracer_amd import PathTracer
from tests.test_gpu_soup_cull import _shingles, _grazing_rays, CENTERS
pytestmark = pytest.mark.gpu

@pytest.mark.parametrize("where", list(CENTERS))
@pytest.mark.parametrize("mode", ["MI355PT_CULL_PLAIN", "MI355PT_DEFAULT"])
@pytest.mark.parametrize("lanes", ["64", "1"])
def test_plain(where, mode, lanes, monkeypatch):
    from oracle.binding import OracleScene
    center, size = CENTERS[where]
    s, stack, light = _shingles(center, size)
    monkeypatch.setenv(mode, "1")
    monkeypatch.setenv("MI355PT_NODE_LANES", lanes)
    monkeypatch.setenv("MI355PT_OWNER_LANES", "8")
    pt = PathTracer(s); pt.Init(32, 32)
    orc = OracleScene(s)
    o, d, L, c = _grazing_rays(stack, size, 40000, 9)
    node, mesh, face, t = pt.trace_rays(o, d)
    onode, omesh, oface, ot = orc.trace_rays(o, d)
    same = (node == onode) & (face == oface) & (t.view(np.uint32) == ot.view(np.uint32))
    bad = np.flatnonzero(~same)
    print(f"PLAIN {mode} lanes {lanes} closest {where}: {bad.size} of {same.size} differ; |cos| quantiles of the bad ones {np.quantile(c[bad],[0,0.5,1]) if bad.size else None}")
    o, d, L, c = _grazing_rays(stack, size, 40000, 13, towards_x=True)
    rng = np.random.default_rng(2)
    tmax = np.where(rng.random(L.size) < 0.3, np.float32(1.0e30), L * rng.uniform(0.9, 1.1, L.size).astype(np.float32)).astype(np.float32)
    got = pt.trace_visibility(o, d, tmax, light); want = orc.light_vis(o, d, tmax, light)
    print(f"PLAIN {mode} lanes {lanes} visibility {where}: {(got != want).sum()} differ")
    pt.close_render()
