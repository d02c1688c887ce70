"""Converged-image parity as SURVEY section 8(d)(3) words it, and the one workload statistic the survey recorded from the
REAL reference binary.

* `test_converged_image_independent_seeds`: C1 at 128x128, 4096 spp on the GPU with seed A against 4096 spp of the
  oracle with a DIFFERENT seed B: relMSE = mean (a - b)^2 / (b^2 + 1e-2) on the tone-mapped mean image <= 1e-3
  (north_star's bar).  Comparing the same samples (test_progressive_image) cannot see a bias that an RNG-stream or
  sample-index mistake would put into both sides; two independent estimates can: their difference is Monte-Carlo
  noise (about 7e-5 here, measured oracle-vs-oracle) plus any bias, so the test also holds the result to 2e-4.
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
