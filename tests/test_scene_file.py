"""mi355pt_scene_save / _load and the multi-device entry points on a machine WITHOUT a GPU: flattening and saving are host
work and must succeed; everything that needs a device must fail with an error code and a message, never fall back
(the GPU side of these calls is tests/test_gpu_group.py)."""
import ctypes as C
import os
import struct

import pytest

from pathtracer_amd import DeviceScene, capi
from tests import common


def _no_gpu():
    return capi.load_library().mi355pt_device_count() == 0


def test_save_is_host_only_and_the_file_is_self_describing(tmp_path):
    mk, w, h = common.small_scenes()["teapot_cornell"]
    s = mk(w, h)
    a, b = str(tmp_path / "a.mi355"), str(tmp_path / "b.mi355")
    DeviceScene.save(s, a)
    DeviceScene.save(s, b, with_trees=True)
    assert os.path.getsize(b) > os.path.getsize(a) > 6320 * (48 + 80)          # slots + attribute records of the teapot at least
    assert not os.path.exists(a + ".part")                                      # written under a temporary name, then renamed
    raw = open(a, "rb").read()
    magic, version = struct.unpack_from("<II", raw, 0)
    assert magic == 0x3553494D and version == 3
    # deterministic: the same description flattens to the same bytes (one build can stand for N)
    DeviceScene.save(s, str(tmp_path / "c.mi355"))
    assert open(str(tmp_path / "c.mi355"), "rb").read() == raw


def test_bad_files_are_rejected_with_a_message(tmp_path):
    lib = capi.load_library()
    mk, w, h = common.small_scenes()["cornell_c1"]
    good = str(tmp_path / "g.mi355")
    DeviceScene.save(mk(w, h), good)
    raw = open(good, "rb").read()
    cases = {"missing": None, "empty": b"", "magic": b"XXXX" + raw[4:], "version": raw[:4] + struct.pack("<I", 99) + raw[8:],
             "layout": raw[:8] + struct.pack("<I", 1) + raw[12:], "truncated": raw[:len(raw) // 2]}
    for name, data in cases.items():
        p = str(tmp_path / f"{name}.mi355")
        if data is not None:
            open(p, "wb").write(data)
        h_ = C.c_void_p()
        rc = lib.mi355pt_scene_load(os.fsencode(p), 0, C.byref(h_))
        assert rc == -1 and not h_.value, name
        assert lib.mi355pt_last_error(), name
    assert lib.mi355pt_scene_save(None, b"/tmp/x", 0) == -1
    desc, keep = mk(w, h).to_desc()
    assert lib.mi355pt_scene_save(C.byref(desc), b"/nonexistent-dir/x.mi355", 0) == -1


@pytest.mark.skipif(not _no_gpu(), reason="a GPU is present")
def test_without_a_gpu_multi_device_calls_fail_loudly(tmp_path):
    lib = capi.load_library()
    mk, w, h = common.small_scenes()["cornell_c1"]
    s = mk(w, h)
    p = str(tmp_path / "s.mi355")
    DeviceScene.save(s, p)
    hnd = C.c_void_p()
    assert lib.mi355pt_scene_load(os.fsencode(p), 0, C.byref(hnd)) == -2 and b"no CPU fallback" in lib.mi355pt_last_error()
    desc, keep = s.to_desc()
    devs = (capi.i32 * 2)(0, 1)
    out = (C.c_void_p * 2)()
    assert lib.mi355pt_scene_create_multi(C.byref(desc), devs, 2, out) == -2 and not out[0] and not out[1]
    prm = capi.make_params(w, h)
    g = C.c_void_p()
    assert lib.mi355pt_group_create(C.byref(desc), devs, 2, C.byref(prm), capi.SPLIT_SAMPLE, 0, C.byref(g)) in (-2, -4) and not g.value
    assert lib.mi355pt_group_create(C.byref(desc), devs, 2, C.byref(prm), 7, 0, C.byref(g)) == -1        # unknown split
    assert lib.mi355pt_group_step(None, 1) == -1 and lib.mi355pt_group_size(None) == 0
    assert lib.mi355pt_abi_version() == capi.ABI_VERSION
