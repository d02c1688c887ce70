"""Converged-image parity as SURVEY section 8(d)(3) words it, and the one workload statistic the survey recorded from the
REAL reference binary.

* `test_converged_image_independent_seeds`: C1 at 128x128, 4096 spp on the GPU with seed A against 4096 spp of the
  oracle with a DIFFERENT seed B: relMSE = mean (a - b)^2 / (b^2 + 1e-2) on the tone-mapped mean image <= 1e-3
  (north_star's bar).  Comparing the same samples (test_progressive_image) cannot see a bias that an RNG-stream or
  sample-index mistake would put into both sides; two independent estimates can: their difference is Monte-Carlo
  noise (about 7e-5 here, measured oracle-vs-oracle) plus any bias, so the test also holds the result to 2e-4.
* `test_converged_views_c2_c3`: the same for the textured Disney scene (C2: four 1024^2 maps, 32-tap filter at bounce 0, two
  meshes) and the Cook-Torrance Cornell with depth of field (C3), whole views at a resolution the oracle finishes in half a
  minute; the 10 M-triangle C4 is held to the same bar on a crop of its full-size frame in
  tests/test_gpu_parity.py::test_full_size_10m_triangle_scene (it shares that test's scene build).  Each result is reported
  against the oracle-vs-oracle noise floor of the same view (two more seeds).
* `test_c1_ray_mix_matches_reference_probe`: SURVEY section 6 [probe]: the reference itself, on assets/cornell.xml at
  512x512, traced 4.45 rays per sample, about 57 % of them closest-hit.  Second-hand (a number written down by the
  survey, not a fixture), but it is the only output of the real reference this repo can compare with.
"""
import numpy as np
import pytest

from pathtracer_amd import PathTracer, capi, scenes
from tests import common

pytestmark = pytest.mark.gpu


def test_converged_image_independent_seeds():
    from oracle.binding import OracleScene
    w = h = 128
    spp = 4096
    s = scenes.cornell_c1(w, h)
    pt = PathTracer(s, seed=0xA11CE)
    pt.Init(w, h)
    for _ in range(spp // 256):
        pt.step(256)
    pt.sync()
    mean, cnt = pt.read_accum()
    assert (cnt == spp).all()
    orc = OracleScene(s)
    prm = capi.make_params(w, h, seed=0xB0B)
    import os
    omean, ocnt, _, _ = orc.render(prm, spp, threads=max(1, min(16, (os.cpu_count() or 2) - 1)))
    assert (ocnt == spp).all()
    r = common.rel_mse(mean, omean)
    assert r <= 1e-3, f"relMSE {r}"
    assert r <= 2e-4, f"relMSE {r}: above the Monte-Carlo noise of two independent 4096-spp estimates (~7e-5)"
    pt.close_render()


@pytest.mark.parametrize("name,w,h,spp", [("c3_cook_torrance_dof", 96, 64, 4096), ("c2_textured_1024_maps", 64, 36, 2048)])
def test_converged_views_c2_c3(name, w, h, spp):
    """relMSE <= 1e-3 (north_star) between the GPU's converged view (seed A) and the oracle's (seed B), and no worse than twice
    the oracle-vs-oracle noise floor (seeds B, C): include/render.h:107-246 end to end -- Cook-Torrance sampling and the
    thin lens for C3, the filtered texture path and the BVHs of teapot.obj + cat.obj for C2 -- with nothing shared but the scene."""
    from oracle.binding import OracleScene
    s = scenes.cornell_c3(w, h, dof=0.5) if name.startswith("c3") else scenes.textured_c2(w, h, tex_size=1024)
    pt = PathTracer(s, seed=0xA11CE)
    pt.Init(w, h)
    orc = OracleScene(s)
    r, floor, gpu, ob = common.converged_crop(pt, orc, (0, 0, w, h), spp)
    print(f"{name}: relMSE gpu-vs-oracle {r:.3e}, oracle-vs-oracle noise floor {floor:.3e} at {spp} spp")
    assert gpu.max() > 0.2 and ob.max() > 0.2
    assert r <= 1e-3, f"relMSE {r}"
    assert r <= 2.0 * floor + 2e-5, f"relMSE {r} against a noise floor of {floor}: a bias, not Monte-Carlo noise"
    pt.close_render()


def test_c1_ray_mix_matches_reference_probe():
    w = h = 512
    pt = PathTracer(scenes.cornell_c1(w, h))
    pt.Init(w, h)
    pt.render(4)
    st = pt.stats()
    rays = st["closest_rays"] + st["shadow_rays"]
    per_sample = rays / st["samples"]
    assert st["samples"] == w * h * 4
    assert abs(per_sample - 4.45) <= 0.01 * 4.45, per_sample
    assert abs(st["closest_rays"] / rays - 0.57) <= 0.02, st["closest_rays"] / rays
    pt.close_render()
