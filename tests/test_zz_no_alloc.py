"""mi355pt_trace_rays_static -- the entry point that allocates nothing (include/mi355pt.h) -- against
mi355pt_trace_rays, the existing entry point it is a variant of: same rays, same production closest-hit kernel,
results bit-identical (integer ids and the bits of t).  Two calls in a row on the same GPU, the second with a
smaller input, because every call shares that GPU's static scratch array and finds the last call's contents in it.
Device buffers for the inputs and outputs are the caller's job (hipMalloc through ctypes here); the call itself allocates nothing.
"""
import ctypes as C

import numpy as np
import pytest

from oracle.binding import OracleScene
from pathtracer_amd import PathTracer, capi
from tests import common

pytestmark = pytest.mark.gpu


def _surface_rays(mesh, n, seed):
    rng = np.random.default_rng(seed)
    f = rng.integers(0, mesh.faces.shape[0], n)
    bary = rng.random((n, 3)).astype(np.float32)
    bary /= bary.sum(1, keepdims=True)
    o = (mesh.vertices[mesh.faces[f, :3]] * bary[:, :, None]).sum(1).astype(np.float32)
    d = rng.normal(size=(n, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    return (o + np.float32(1e-4) * d).astype(np.float32), d.astype(np.float32)


def _first_mesh(node):
    if node.obj is not None and getattr(node.obj, "meshes", None):
        return node.obj.meshes[0]
    for c in node.children:
        m = _first_mesh(c)
        if m is not None:
            return m
    return None


class _Hip:
    """The HIP runtime libmi355pt.so itself is linked against, through ctypes: device buffers for the caller's inputs and
    outputs (torch's bundled runtime cannot be initialised once another copy of the runtime is live in the process)."""

    def __init__(self):
        path = None
        with open("/proc/self/maps") as f:
            for line in f:
                if "libamdhip64" in line:
                    path = line.split()[-1]
                    break
        assert path, "libmi355pt.so should have pulled in libamdhip64"
        self.lib = C.CDLL(path)
        self.lib.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
        self.lib.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        self.lib.hipMemset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
        self.lib.hipFree.argtypes = [C.c_void_p]

    def alloc(self, nbytes):
        p = C.c_void_p()
        assert self.lib.hipMalloc(C.byref(p), max(nbytes, 16)) == 0
        return p

    def upload(self, arr):
        p = self.alloc(arr.nbytes)
        assert self.lib.hipMemcpy(p, arr.ctypes.data, arr.nbytes, 1) == 0
        return p

    def download(self, p, n, dtype):
        out = np.zeros(n, dtype=dtype)
        assert self.lib.hipMemcpy(out.ctypes.data, p, out.nbytes, 2) == 0
        return out


def _static_trace(lib, hip, dscene, o, d):
    n = o.shape[0]
    po, pd = hip.upload(np.ascontiguousarray(o)), hip.upload(np.ascontiguousarray(d))
    outs = [hip.alloc(4 * n) for _ in range(4)]
    for p in outs:
        assert hip.lib.hipMemset(p, 0xAB, 4 * n) == 0
    assert hip.lib.hipDeviceSynchronize() == 0
    capi.check(lib, lib.mi355pt_trace_rays_static(dscene.h, po, pd, n, outs[0], outs[1], outs[2], outs[3], None), "trace_rays_static")
    assert hip.lib.hipDeviceSynchronize() == 0       # the call only enqueues (null stream here)
    res = (hip.download(outs[0], n, np.int32), hip.download(outs[1], n, np.int32), hip.download(outs[2], n, np.int32),
           hip.download(outs[3], n, np.float32))
    for p in [po, pd] + outs:
        hip.lib.hipFree(p)
    return res


@pytest.mark.parametrize("name", ["teapot_cornell", "stress_layout"])
def test_static_entry_matches_trace_rays(name):
    mk, w, h = common.small_scenes()[name]
    s = mk(w, h)
    pt = PathTracer(s)
    pt.Init(w, h)
    lib = pt.lib
    hip = _Hip()
    m = _first_mesh(s.rootNode)
    # first call: a large batch (several chunks per wave, deep stacks); second call: a smaller one, other rays
    for n, seed in ((100000, 1), (7001, 2), (64, 3), (1, 4)):
        o, d = _surface_rays(m, n, seed)
        want = pt.trace_rays(o, d)
        got = _static_trace(lib, hip, pt.dscene, o, d)
        assert (want[0] >= 0).mean() > 0.5
        for a, b in zip(got[:3], want[:3]):
            assert np.array_equal(a, b)
        assert np.array_equal(got[3].view(np.uint32), want[3].view(np.uint32)), "t must be bit-identical"
        if n <= 7001:              # ... and against the oracle's TraceNode (exhaustive walk): the static path's own parity evidence
            on, om, of, ot = OracleScene(s).trace_rays(o, d)
            assert np.array_equal(got[0], on) and np.array_equal(got[1], om) and np.array_equal(got[2], of)
            assert np.array_equal(got[3].view(np.uint32), ot.view(np.uint32))
    pt.close_render()


def test_static_entry_takes_the_exact_walk_switch(monkeypatch):
    """MI355PT_NO_CULL -- the exhaustive walk of the reference, the switch for callers that need its answer on grazing rays
     -- reaches the static entry point too (it is read when the scene is created): on the
    adversarial shingle scene of tests/test_gpu_soup_cull.py the static path then agrees with the oracle on every ray."""
    from tests.test_gpu_soup_cull import CENTERS, _grazing_rays, _shingles
    center, size = CENTERS["1e4"]
    s, stack, _ = _shingles(center, size)
    o, d, L, c = _grazing_rays(stack, size, 40000, 9)
    monkeypatch.setenv("MI355PT_NO_CULL", "1")
    pt = PathTracer(s)                          # the scene reads the switch when it is created
    pt.Init(32, 32)
    got = _static_trace(pt.lib, _Hip(), pt.dscene, o, d)
    on, om, of, ot = OracleScene(s).trace_rays(o, d)
    assert (on >= 0).mean() > 0.5
    assert np.array_equal(got[0], on) and np.array_equal(got[2], of) and np.array_equal(got[3].view(np.uint32), ot.view(np.uint32))
    pt.close_render()
