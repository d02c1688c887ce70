"""Generates tests/golden/oracle_vectors.npz from the CPU oracle (oracle/pt_oracle.cpp).

PROVENANCE: these vectors are outputs of THIS REPO'S ORACLE, not of the reference binary (the
reference does not build here and ships no fixtures for this path: DESIGN.md "Oracle pin status").
They freeze the oracle's results so that (a) a change to the oracle that alters results is
caught, and (b) the HIP path can be checked on the GPU box against committed data as well as
against the live oracle.  Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle.binding import OracleScene          # noqa: E402
from pathtracer_amd import capi                  # noqa: E402
from tests import common                         # noqa: E402

SEED = 0x5EED
GW, GH = 48, 32          # golden resolution (small: the file is committed)
NPAIRS = 2048


def vectors(name):
    mk, _, _ = common.small_scenes()[name]
    s = mk(GW, GH)
    orc = OracleScene(s)
    prm = capi.make_params(GW, GH, seed=SEED)
    node, mesh, face, t, _ = orc.trace_primary(prm, 0, 0, GW, GH)
    rng = np.random.default_rng(123)
    pixel = rng.integers(0, GW * GH, NPAIRS).astype(np.uint32)
    sample = rng.integers(0, 64, NPAIRS).astype(np.uint32)
    lin, mp = orc.sample_radiance(prm, pixel, sample)
    mean, cnt, rgb8, st = orc.render(prm, 4, threads=4)
    return {f"{name}/node": node, f"{name}/mesh": mesh, f"{name}/face": face, f"{name}/t": t,
            f"{name}/pixel": pixel, f"{name}/sample": sample, f"{name}/linear": lin, f"{name}/mapped": mp,
            f"{name}/mean4": mean, f"{name}/rgb8_4": rgb8,
            f"{name}/rays": np.array([st["closest"], st["shadow"], st["samples"]], dtype=np.uint64)}


def main():
    out = {}
    for name in common.small_scenes():
        out.update(vectors(name))
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "oracle_vectors.npz"), **out)
    print("wrote", len(out), "arrays")


if __name__ == "__main__":
    main()
