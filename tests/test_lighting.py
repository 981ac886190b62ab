"""configs[4]: deferred lighting.  GPU kernel (through the C ABI) vs the CPU evaluation of second_pass_frag.glsl.
Tolerance 1e-4 abs on colours in [0,1] (BASELINE.json)."""
import numpy as np
import pytest

import oracle as ro

TOL = 1e-4


def oracle_lights(L):
    from render_engine_amd import lighting
    keep = []
    S = lighting.fill_lights_struct(ro.LightsC(), L, keep)
    return S, keep


def test_oracle_lighting_properties():
    """CPU only: the restated GLSL -- spot term counted twice, default-diffuse floor, clamp, ambient-only branch"""
    from render_engine_amd import lighting
    pos, nrm, alb = lighting.synthetic_gbuffer(32, 32, patch=64.0)
    L = lighting.synthetic_lights(n_spot=8, n_point=0, patch=64.0, radius=40.0)
    S, keep = oracle_lights(L)
    full = ro.deferred_lighting(pos, nrm, alb, S)
    assert full.shape == (1024, 4) and np.all(full[:, 3] == 1.0) and np.all((full[:, :3] >= 0) & (full[:, :3] <= 1))
    L0 = dict(L); L0["n_spot"] = 0
    S0, keep0 = oracle_lights(L0)
    none = ro.deferred_lighting(pos, nrm, alb, S0)
    np.testing.assert_allclose(none[:, :3], (alb[:, :3] / np.float32(255.0)) * np.float32(0.2), atol=1e-7)   # 0 < cutoff: floor only
    L1 = dict(L); L1["any_light_source_visible"] = 0
    S1, keep1 = oracle_lights(L1)
    amb = ro.deferred_lighting(pos, nrm, alb, S1)
    np.testing.assert_allclose(amb[:, :3], (alb[:, :3] / np.float32(255.0)) * np.float32(0.2), atol=1e-7)
    sub = ro.deferred_lighting(pos, nrm, alb, S, idx=np.array([5, 77, 1000], np.uint32))
    np.testing.assert_array_equal(sub, full[[5, 77, 1000]])
    assert float(np.abs(full[:, :3] - none[:, :3]).max()) > 0.05            # the lights do contribute


@pytest.mark.gpu
@pytest.mark.parametrize("w,h,ns,npt", [(96, 64, 40, 0), (200, 130, 300, 5), (64, 64, 1500, 0)])
def test_lighting_matches_oracle_small(w, h, ns, npt):
    from render_engine_amd import lighting
    pos, nrm, alb = lighting.synthetic_gbuffer(w, h, patch=160.0)
    L = lighting.synthetic_lights(n_spot=ns, n_point=npt, patch=160.0, radius=40.0)
    dl = lighting.DeferredLighting(w, h, max_spot_lights=2048, max_point_lights=16)
    dl.upload_gbuffer(pos, nrm, alb); dl.set_lights(L)
    dl.run()
    got = dl.read()
    S, keep = oracle_lights(L)
    exp = ro.deferred_lighting(pos, nrm, alb, S)
    assert np.abs(got - exp).max() <= TOL, np.abs(got - exp).max()
    L1 = dict(L); L1["any_light_source_visible"] = 0
    dl.set_lights(L1); dl.run()
    S1, keep1 = oracle_lights(L1)
    assert np.abs(dl.read() - ro.deferred_lighting(pos, nrm, alb, S1)).max() <= TOL
    dl.close()


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["tall", "one_point", "mixed_radii", "far_apart"])
def test_lighting_odd_light_sets(case):
    """light sets that stress the slab order of the records (re_lighting_set_lights): largest extent along y, no extent at all, radii from tiny to
    larger than the scene and negative (a distance is never below a negative radius: the light never shines), a few lights far outside the scene"""
    from render_engine_amd import lighting
    w, h = 160, 96
    pos, nrm, alb = lighting.synthetic_gbuffer(w, h, patch=200.0)
    L = lighting.synthetic_lights(n_spot=700, n_point=3, patch=200.0, radius=30.0)
    rng = np.random.default_rng(11)
    sp = L["spot_pos"]
    if case == "tall":
        sp[:, 1] = 1000.0 + rng.random(700).astype(np.float32) * 900.0 - 450.0; sp[:, 0] = 1100.0 + rng.random(700).astype(np.float32) * 5.0
    elif case == "one_point":
        sp[:] = np.array([1090.0, 1012.0, 1060.0], np.float32); L["spot_radius"] = (rng.random(700) * 90.0).astype(np.float32)
    elif case == "mixed_radii":
        r = (1.0 + rng.random(700) * 120.0).astype(np.float32); r[5] = 5000.0; r[17] = -40.0; r[300:310] = -1.0; r[400] = 0.0
        L["spot_radius"] = r
    else:
        sp[::50] += np.float32(1.0e6); sp[7] -= np.float32(3.0e7)
    dl = lighting.DeferredLighting(w, h, max_spot_lights=1024, max_point_lights=16)
    dl.upload_gbuffer(pos, nrm, alb); dl.set_lights(L)
    dl.run()
    got = dl.read()
    S, keep = oracle_lights(L)
    exp = ro.deferred_lighting(pos, nrm, alb, S)
    assert np.abs(got - exp).max() <= TOL, (case, np.abs(got - exp).max())
    dl.close()


@pytest.mark.gpu
def test_lighting_full_size_sampled():
    """4096x4096 G-buffer, 4096 radius-40 lights (configs[4]): 4096 random pixels against the brute-force CPU evaluation,
    plus size-independent properties (alpha, range, determinism)."""
    from render_engine_amd import lighting
    w = h = 4096
    pos, nrm, alb = lighting.synthetic_gbuffer(w, h)
    L = lighting.synthetic_lights(n_spot=4096, n_point=0)
    dl = lighting.DeferredLighting(w, h, max_spot_lights=4096, max_point_lights=64)
    dl.upload_gbuffer(pos, nrm, alb); dl.set_lights(L)
    us = dl.run()
    idx = np.random.default_rng(3).integers(0, w * h, 4096).astype(np.uint32)
    got = dl.read_pixels(idx)
    S, keep = oracle_lights(L)
    exp = ro.deferred_lighting(pos, nrm, alb, S, idx=idx)
    assert np.abs(got - exp).max() <= TOL, np.abs(got - exp).max()
    assert np.all(got[:, 3] == 1.0) and got[:, :3].min() >= 0 and got[:, :3].max() <= 1
    lit = np.abs(got[:, :3] - (alb[idx, :3] / np.float32(255.0)) * np.float32(0.2)).max(axis=1) > 1e-3
    assert lit.mean() > 0.5                                            # most pixels are within reach of some light
    dl.run()
    np.testing.assert_array_equal(dl.read_pixels(idx), got)            # bitwise reproducible (ordered light lists)
    print(f"deferred lighting 4096x4096x4096 lights: {us:.1f} us")
    dl.close()
