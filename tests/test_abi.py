"""The C-ABI library loads without a GPU, exports every symbol include/mi355pt.h declares, the
ctypes mirror matches the header's struct layout, and GPU entry points fail loudly (no CPU
fallback) when no gfx950 device is present.  No compute call is made here."""
import ctypes as C
import os
import re
import subprocess
import sys

import pytest

from pathtracer_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "mi355pt.h")


def declared_functions():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    text = re.sub(r"//[^\n]*", "", text)
    return sorted(set(re.findall(r"\b(mi355pt_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_what_capi_binds():
    decl = declared_functions()
    assert len(decl) >= 20
    assert sorted(capi.SYMBOLS) == decl


def test_library_exports_every_declared_symbol():
    lib = capi.load_library()
    for name in declared_functions():
        assert hasattr(lib, name), f"libmi355pt.so does not export {name}"
    assert b"mi355pt" in lib.mi355pt_version()


def test_every_entry_point_cites_the_reference():
    """include/*.h: each entry point's comment names the reference code it replaces."""
    text = open(HEADER).read()
    assert text.count("src/") + text.count("include/") >= 15


def test_struct_layout_matches_header(tmp_path):
    pairs = {"mi355pt_node": capi.Node, "mi355pt_bvh_node": capi.BvhNode, "mi355pt_mesh": capi.Mesh, "mi355pt_model": capi.Model,
             "mi355pt_texmap": capi.TexMap, "mi355pt_texture": capi.Texture, "mi355pt_material": capi.Material,
             "mi355pt_light": capi.Light, "mi355pt_camera": capi.Camera, "mi355pt_scene_desc": capi.SceneDesc,
             "mi355pt_render_params": capi.RenderParams, "mi355pt_stats": capi.Stats}
    src = tmp_path / "sz.c"
    body = "".join(f'printf("{n} %zu\\n", sizeof({n}));' for n in pairs)
    src.write_text(f'#include <stdio.h>\n#include "mi355pt.h"\nint main(void){{{body}return 0;}}\n')
    exe = tmp_path / "sz"
    subprocess.check_call(["gcc", "-std=c99", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])   # the header is plain C
    out = dict(l.split() for l in subprocess.check_output([str(exe)]).decode().splitlines())
    for n, t in pairs.items():
        assert int(out[n]) == C.sizeof(t), f"{n}: header {out[n]} B, ctypes {C.sizeof(t)} B"


def test_missing_library_is_a_loud_error(tmp_path):
    with pytest.raises(Exception):
        capi.load_library(str(tmp_path / "nope.so"))


def test_product_never_touches_the_oracle():
    """The product path must not import, link or call anything under oracle/."""
    pkg = os.path.join(ROOT, "pathtracer_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")) or f == "Makefile":
                text = open(os.path.join(dp, f), errors="ignore").read()
                assert not re.search(r"^\s*(from|import)\s+oracle|liboracle|#include\s+\"[^\"]*oracle", text, flags=re.M), f
    code = "import sys; import pathtracer_amd; sys.exit(1 if any(m == 'oracle' or m.startswith('oracle.') for m in sys.modules) else 0)"
    assert subprocess.call([sys.executable, "-c", code], cwd=ROOT) == 0


@pytest.mark.skipif(capi.load_library().mi355pt_device_count() > 0, reason="a GPU is present")
def test_no_gpu_means_errors_not_fallback():
    from pathtracer_amd import PathTracer, scenes
    s = scenes.cornell_c1(16, 16)
    with pytest.raises(capi.BackendError) as e:
        PathTracer(s)
    assert "device" in str(e.value).lower() or "hip" in str(e.value).lower()
