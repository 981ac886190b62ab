"""CPU-side checks of the drop-in boundary: the library loads, exports every symbol include/re_hip.h
declares, the ctypes structs match the header layout, and the product fails loudly without a GPU.
No compute calls are made here."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    src = open(os.path.join(ROOT, "include", "re_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(re_[a-z_0-9]+)\s*\(", src)))


def test_every_declared_symbol_is_exported():
    import render_engine_amd as R
    L = R._capi.load()
    names = header_functions()
    assert len(names) >= 20
    for n in names:
        assert hasattr(L, n), f"{n} declared in include/re_hip.h but not exported"
    assert set(R._capi.EXPORTS) <= set(names)
    assert L.re_abi_version() == 3


def test_struct_layouts_match_header():
    from render_engine_amd import _capi
    assert C.sizeof(_capi.Config) == 20
    assert C.sizeof(_capi.CameraC) == 16 * 4 + 3 * 4 + 3 * 4 + 4 + 4 + 8 * 4 + 8 * 4
    assert C.sizeof(_capi.InstanceRange) == 20
    assert C.sizeof(_capi.TickResult) == 12
    assert C.sizeof(_capi.Entities) == 8 + 13 * 8
    assert C.sizeof(_capi.Visible) == 5 * 4 + 4 + 3 * 8
    assert C.sizeof(_capi.Stats) == 5 * 4 + 4 + 8 + 10 * 4           # 5 counts, padding, device_bytes, 10 counters


def test_fails_loudly_without_a_device():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    import render_engine_amd as R
    with pytest.raises(R.RenderEngineError) as e:
        R.Pipeline()
    assert "no CPU path" in str(e.value) or "HIP" in str(e.value)


def test_no_product_import_of_the_oracle():
    """the oracle is test infrastructure: nothing under render_engine_amd/ or include/ may reference it"""
    for base, _, files in os.walk(os.path.join(ROOT, "render_engine_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(base, f), errors="ignore").read()
                assert "import oracle" not in txt and "re_oracle" not in txt and "libre_oracle" not in txt, f


def test_host_camera_matches_oracle_camera_helpers():
    """host-side perspective/look_at/mat4_mul (render_engine_amd/pipeline.py) against the oracle's restatement"""
    import oracle as ro
    from render_engine_amd import pipeline as P
    for pos, d, far in [((1000, 1000, 1150), (0, 0, -1), 1000.0), ((8192.5, 100.25, 77), (0.6, 0.0, -0.8), 8192.0)]:
        proj = P.perspective(np.float32(1280) / np.float32(720), np.radians(np.float32(45.0)), 0.1, far)
        np.testing.assert_allclose(proj, ro.perspective(np.float32(1280) / np.float32(720), np.radians(np.float32(45.0)), 0.1, far), rtol=1e-6)
        view = P.look_at(np.asarray(pos, np.float32), np.asarray(pos, np.float32) + np.asarray(d, np.float32))
        np.testing.assert_array_equal(view, ro.look_at(pos, np.asarray(pos, np.float32) + np.asarray(d, np.float32)))
        np.testing.assert_array_equal(P.mat4_mul(proj, view), ro.mat4_mul(proj, view))
    lo, hi = P.create_level_of_views(1000.0)
    olo, ohi = ro.default_lod(1000.0)
    np.testing.assert_array_equal(lo, olo); np.testing.assert_array_equal(hi, ohi)


def test_synthetic_world_is_deterministic_and_in_bounds():
    from render_engine_amd import synthetic
    a = synthetic.lattice_world(cells_per_axis=10, first_cell=100, spinner_every=7)
    b = synthetic.box_world((10, 10, 10), first_cell=100, spinner_every=7, index_range=(500, 1000))
    assert a[500:1000].tobytes() == b.tobytes()          # any rank can generate its own shard
    lo = a["pos"] + a["original"][:, 0::2] * a["scale"]; hi = a["pos"] + a["original"][:, 1::2] * a["scale"]
    assert np.all(np.floor(lo / 64) == np.floor(hi / 64))   # every box inside one level-0 section



def test_no_exception_crosses_the_abi():
    """include/re_hip.h: "no exceptions/aborts cross the ABI".  Every extern "C" entry point is a function-try-block (csrc/re_guard.h); a host-only path
    that throws is used as the witness: re_history_set_state with a blob length no vector can hold makes std::vector::assign throw std::length_error
    before a byte is read -- without the guard that is std::terminate -> abort() inside the caller's process."""
    import render_engine_amd as R
    from render_engine_amd import _capi
    L = _capi.load()
    ids = _capi.TypeIds(*range(1, 10)); h = C.c_void_p()
    assert L.re_history_create(C.byref(ids), 0, C.byref(h)) == 0
    blob = (C.c_uint8 * 16)()
    rc = L.re_history_set_state(h, blob, 1 << 62, blob, 0)
    assert rc in (-5, -3), rc                                   # RE_E_STATE (std::exception) or RE_E_CAPACITY (std::bad_alloc)
    msg = L.re_history_last_error(h).decode()
    assert "re_history_set_state" in msg, msg
    assert L.re_history_set_state(h, blob, 16, blob, 8) == 0     # the object is still usable
    L.re_history_destroy(h)


def test_every_entry_point_is_guarded():
    """each extern "C" definition with a body of more than one line is a function-try-block closed by RE_ABI_GUARD*; the one-liners cannot throw"""
    n = 0
    for f in ("re_api.hip", "re_lighting.hip", "re_history.cpp"):
        lines = open(os.path.join(ROOT, "render_engine_amd", "csrc", f)).read().split("\n")
        for i, ln in enumerate(lines):
            if not ln.startswith('extern "C"'):
                continue
            if ln.rstrip().endswith("{"):
                assert ln.rstrip().endswith("try {"), (f, i + 1, ln[:80])
                j = i + 1
                while not lines[j].startswith("}"):
                    j += 1
                assert "RE_ABI_GUARD" in lines[j] or "catch (...)" in lines[j], (f, j + 1, lines[j][:80])
                n += 1
            else:
                assert not re.search(r"\b(new|vector|push_back|resize|assign|std::string\()", ln), (f, i + 1, ln[:120])
    assert n >= 45


def test_rust_shim_declares_every_symbol_of_the_header():
    """integration/rust (the extern "C" block a maintainer adds to the reference crate) is source only -- no Rust toolchain in this image --, so what
    can be checked here is that it names every function include/re_hip.h declares, that its ABI-version comment is current, and that the record sizes it relies on
    are the library's (the #[repr(C)] structs are field-for-field transcriptions of the header's)"""
    import render_engine_amd as R
    from render_engine_amd import _capi
    ffi = open(os.path.join(ROOT, "integration", "rust", "src", "gpu_visible_set", "ffi.rs")).read()
    declared = set(re.findall(r"pub fn (re_[a-z_0-9]+)", ffi))
    assert set(header_functions()) <= declared, sorted(set(header_functions()) - declared)
    assert "re_abi_version() -> u32;" in ffi and "// %d" % _capi.load().re_abi_version() in ffi.split("re_abi_version() -> u32;")[1].split("\n")[0]
    assert R.ENTITY_DT.itemsize == 140 and "// 140 bytes" in ffi                 # re_entity_state == ENTITY_DT == ReEntityState
    mod = open(os.path.join(ROOT, "integration", "rust", "src", "gpu_visible_set", "mod.rs")).read()
    for wrapper in ("upload_instance_data_to_render_system", "cull_result", "add_entities", "apply_changes_with_added", "take_migrants", "allgather_visible", "EntityIds"):
        assert wrapper in mod, wrapper
