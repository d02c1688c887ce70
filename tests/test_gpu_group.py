"""Multi-GPU behind the C ABI (mi355pt_group_*, mi355pt_comm_*, mi355pt_scene_save / _load) on the one GPU a test box
has: a group over device 0 alone, logical shards that share device 0, and the packed reduce forced through RCCL with a
one-rank communicator.  What is compared:
  * one shard: the group IS the single renderer, bit for bit;
  * tile split, any number of shards: every pixel has one owner and the packed sum adds zeros to it -> bit-identical
    to the single renderer, counts exact;
  * sample split: the shards render the sample indices 0 .. G*spp-1 between them -> counts exact, mean equal to the
    single renderer's up to the fp32 rounding of (n1 m1 + n2 m2) / (n1 + n2) against a sequential running mean, and
    within the progressive-image bar of the ORACLE's render of the same samples.
The N > 1 DEVICE case cannot run on this box; it is the same code with distinct ordinals and ncclCommInitAll over them.
"""
import ctypes as C
import os

import numpy as np
import pytest

from pathtracer_amd import DeviceScene, PathTracer, PathTracerGroup, capi
from tests import common

pytestmark = pytest.mark.gpu

SCENES = common.small_scenes()


def _single(s, w, h, spp):
    pt = PathTracer(s)
    pt.Init(w, h)
    img = pt.render(spp)
    mean, cnt = pt.read_accum()
    st = pt.stats()
    pt.close_render()
    return img, mean, cnt, st


@pytest.mark.parametrize("force_rccl", [False, True])
def test_group_of_one_device_is_the_single_renderer(force_rccl):
    mk, w, h = SCENES["teapot_cornell"]
    s = mk(w, h)
    img, mean, cnt, st = _single(s, w, h, 3)
    g = PathTracerGroup(s, devices=[0], split="tile", force_rccl=force_rccl)      # forced: pack -> ncclReduce (1 rank) -> unpack
    g.Init(w, h)
    g.step(2); g.step(1)
    gm, gc = g.read_accum()
    assert np.array_equal(gc, cnt) and int(gc[0]) == 3
    assert np.array_equal(gm.view(np.uint32), mean.view(np.uint32))
    assert np.array_equal(g.GetPixels(), img)
    gs = g.stats()
    assert gs["samples"] == st["samples"] and gs["closest_rays"] == st["closest_rays"] and gs["shadow_rays"] == st["shadow_rays"]
    if force_rccl:
        assert g.readout_ms() > 0.0
    g.close()


@pytest.mark.parametrize("shards", [2, 3])
def test_tile_split_over_logical_shards_is_bit_identical(shards):
    mk, w, h = SCENES["teapot_cornell"]
    s = mk(w, h)
    img, mean, cnt, st = _single(s, w, h, 2)
    g = PathTracerGroup(s, devices=[0] * shards, split="tile")
    g.Init(w, h)
    g.step(2)
    gm, gc = g.read_accum()
    assert np.array_equal(gc, cnt)
    assert np.array_equal(gm.view(np.uint32), mean.view(np.uint32))
    assert np.array_equal(g.GetPixels(), img)
    assert g.stats()["samples"] == w * h * 2
    # a second step after a read-out: the shards' own accumulators were not disturbed by the combine
    g.step(1)
    img3, mean3, cnt3, _ = _single(s, w, h, 3)
    gm, gc = g.read_accum()
    assert np.array_equal(gc, cnt3) and np.array_equal(gm.view(np.uint32), mean3.view(np.uint32))
    g.close()


def test_sample_split_over_logical_shards_matches_single_and_oracle():
    from oracle.binding import OracleScene
    mk, w, h = SCENES["cornell_c1"]
    s = mk(w, h)
    spp, shards = 4, 2
    img, mean, cnt, st = _single(s, w, h, spp * shards)
    g = PathTracerGroup(s, devices=[0] * shards, split="sample")
    g.Init(w, h)
    g.step(spp)
    gm, gc = g.read_accum()
    assert np.array_equal(gc, cnt) and int(gc[0]) == spp * shards
    assert np.allclose(gm, mean, rtol=2e-6, atol=1e-7)
    d = np.abs(g.GetPixels().astype(np.int32) - img.astype(np.int32))
    assert d.max() <= 1
    gs = g.stats()
    assert gs["samples"] == st["samples"] and gs["closest_rays"] == st["closest_rays"]
    # ... and against the oracle's render of the same sample indices
    prm = capi.make_params(w, h)
    omean, ocnt, orgb, _ = OracleScene(s).render(prm, spp * shards, threads=8)
    assert np.array_equal(gc, ocnt)
    assert common.rel_mse(gm, omean) <= 1e-3
    g.close()


def test_comm_reduce_with_one_rank_matches_read_accum():
    """The process-per-GPU read-out (mi355pt_comm_* + mi355pt_render_reduce) with a one-rank communicator: the packed
    ncclReduce is the whole collective path short of a second GPU."""
    mk, w, h = SCENES["cornell_c1"]
    s = mk(w, h)
    pt = PathTracer(s)
    pt.Init(w, h)
    pt.render(3)
    mean, cnt = pt.read_accum()
    lib = pt.lib
    uid = (C.c_uint8 * capi.COMM_ID_BYTES)()
    capi.check(lib, lib.mi355pt_comm_unique_id(uid), "comm_unique_id")
    comm = C.c_void_p()
    capi.check(lib, lib.mi355pt_comm_create(uid, 0, 1, 0, C.byref(comm)), "comm_create")
    for split, exact in ((capi.SPLIT_TILE, True), (capi.SPLIT_SAMPLE, False)):
        m = np.zeros((w * h, 3), dtype=np.float32); c = np.zeros(w * h, dtype=np.uint32); rgb = np.zeros((h, w, 3), dtype=np.uint8)
        ms = C.c_double(0)
        capi.check(lib, lib.mi355pt_render_reduce(pt.h, comm, 0, split, m.ctypes.data, c.ctypes.data, rgb.ctypes.data, C.byref(ms)), "render_reduce")
        assert np.array_equal(c, cnt) and ms.value > 0
        if exact:
            assert np.array_equal(m.view(np.uint32), mean.view(np.uint32)) and np.array_equal(rgb, pt.GetPixels())
        else:
            assert np.allclose(m, mean, rtol=1e-6, atol=1e-7)
    assert lib.mi355pt_render_reduce(pt.h, comm, 1, capi.SPLIT_TILE, None, None, None, None) == -1      # root out of range
    lib.mi355pt_comm_destroy(comm)
    pt.close_render()


def test_scene_file_round_trip(tmp_path):
    """mi355pt_scene_save -> mi355pt_scene_load: the loaded scene renders what the created scene renders, bit for bit,
    and answers get_bvh when the file carries the trees."""
    for name in ("teapot_cornell", "textured_c2", "mesh_light"):
        mk, w, h = SCENES[name]
        s = mk(w, h)
        img, mean, cnt, _ = _single(s, w, h, 2)
        path = str(tmp_path / f"{name}.mi355")
        DeviceScene.save(s, path, with_trees=(name == "teapot_cornell"))
        ds = DeviceScene.load(path, 0)
        pt = PathTracer(ds)
        pt.Init(w, h)
        assert np.array_equal(pt.render(2), img)
        m2, c2 = pt.read_accum()
        assert np.array_equal(c2, cnt) and np.array_equal(m2.view(np.uint32), mean.view(np.uint32))
        ref = DeviceScene(s)
        if name == "teapot_cornell":
            a, fa, na = ds.get_bvh(0)
            b, fb, nb = ref.get_bvh(0)
            assert na == nb and common.bvh_equal(a, fa, b, fb, na)
        else:
            with pytest.raises(capi.BackendError):
                ds.get_bvh(0)
        pt.close_render(); ds.close(); ref.close()


def test_create_multi_uploads_one_build_twice():
    mk, w, h = SCENES["grid_small"]
    s = mk(w, h)
    lib = capi.load_library()
    desc, keep = s.to_desc()
    devs = (capi.i32 * 2)(0, 0)
    out = (C.c_void_p * 2)()
    capi.check(lib, lib.mi355pt_scene_create_multi(C.byref(desc), devs, 2, out), "scene_create_multi")
    hits = []
    for k in range(2):
        ds = DeviceScene.__new__(DeviceScene)
        ds.lib, ds.desc, ds._keep, ds.device, ds.h = lib, desc, keep, 0, C.c_void_p(out[k])
        pt = PathTracer(ds)
        pt.Init(w, h)
        hits.append(pt.trace_primary())
        pt.close_render(); ds.close()
    for a, b in zip(*hits):
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    bad = (capi.i32 * 2)(0, 99)
    assert lib.mi355pt_scene_create_multi(C.byref(desc), bad, 2, out) == -2 and not out[0]


def test_path_pool_is_sized_by_the_batch_and_survives_a_small_limit(monkeypatch):
    """The pool is allocated for the batch a step asks for (one sample per step -> one frame of slots), grows when a later
    step asks for more, and a limit below one frame only makes a step run in more chunks: same image."""
    mk, w, h = SCENES["cornell_c1"]
    s = mk(w, h)
    img, mean, cnt, _ = _single(s, w, h, 5)
    pt = PathTracer(s)
    pt.Init(w, h)
    pt.step(1); pt.step(4)                 # 1-sample pool, then regrown for 4
    pt.sync()
    m, c = pt.read_accum()
    assert np.array_equal(c, cnt) and np.array_equal(m.view(np.uint32), mean.view(np.uint32))
    pt.close_render()
    img8, mean8, cnt8, _ = _single(s, w, h, 8)
    monkeypatch.setenv("MI355PT_MAX_PATHS", "65536")      # 9 216 pixels x 8 spp = 73 728 paths > 65 536 slots: chunks of 7 + 1 samples
    pt = PathTracer(s)
    pt.Init(w, h)
    pt.render(8)
    m, c = pt.read_accum()
    assert np.array_equal(c, cnt8) and np.array_equal(m.view(np.uint32), mean8.view(np.uint32))
    pt.close_render()


def test_group_reset_keeps_the_scenes_and_forced_rccl_adds_shards_first():
    """mi355pt_group_reset = PathTracer::Init on a live group: empty accumulators (also of another size), the uploaded scenes kept;
    and two shards of one device under MI355PT_GROUP_FORCE_RCCL: packed, added on the device, then through ncclReduce."""
    mk, w, h = SCENES["grid_small"]
    s = mk(w, h)
    g = PathTracerGroup(s, devices=[0, 0], split="sample", force_rccl=True)
    g.Init(w, h)
    g.step(2)
    m1, c1 = g.read_accum()
    assert (c1 == 4).all() and g.readout_ms() > 0.0
    g.Init(w, h)                                   # reset: same size
    m0, c0 = g.read_accum()
    assert not c0.any() and not m0.any()
    g.step(2)
    m2, c2 = g.read_accum()
    assert np.array_equal(c1, c2) and np.array_equal(m1.view(np.uint32), m2.view(np.uint32)), "a reset group renders what a fresh one renders"
    img, mean, cnt, _ = _single(s, w, h, 4)
    assert np.allclose(m2, mean, rtol=2e-6, atol=1e-7)
    g.Init(w // 2, h // 2)                         # reset: another size
    g.step(1)
    m3, c3 = g.read_accum()
    assert m3.shape == ((w // 2) * (h // 2), 3) and (c3 == 2).all()
    g.close()
