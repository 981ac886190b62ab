"""INTEGRATION.md shows the Rust binding a maintainer adds (integration/rust/src/gpu_visible_set/ffi.rs) verbatim: the two must not drift apart."""
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_integration_md_carries_the_current_ffi_block():
    ffi = open(os.path.join(ROOT, "integration", "rust", "src", "gpu_visible_set", "ffi.rs")).read().rstrip()
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    assert ffi in doc, "INTEGRATION.md section 2 is not integration/rust/src/gpu_visible_set/ffi.rs verbatim (regenerate the block)"
