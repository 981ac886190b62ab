"""Deferred lighting (BASELINE.json configs[4]): host wrapper of re_lighting_* and the synthetic G-buffer / light set
of SURVEY.md section 8d (height-field patch, radius-40 'spot' lights as the sample's stars, space_logic/solar_system/sun.rs:98-117)."""
import ctypes as C

import numpy as np

from . import _capi
from .pipeline import RenderEngineError
from .synthetic import uniform

SEED_LIGHTS, SEED_GBUF = 0x5EED0005, 0x5EED0006

LIGHT_FIELDS = ["spot_pos", "spot_diffuse", "spot_specular", "spot_ambient", "spot_linear", "spot_quadratic", "spot_radius",
                "point_pos", "point_dir", "point_diffuse", "point_specular", "point_ambient", "point_linear", "point_quadratic",
                "point_cutoff", "point_outer_cutoff"]


def synthetic_gbuffer(width, height, patch=2048.0, origin=(1000.0, 1000.0, 1000.0)):
    """gPosition = (x, 1000 + 8*noise, z) over a patch, gNormal = normalised gradient, gAlbedoSpec = hashed RGB in [0.2, 0.9] (RGBA8)."""
    xs = (np.arange(width, dtype=np.float32) + np.float32(0.5)) * np.float32(patch / width) + np.float32(origin[0])
    zs = (np.arange(height, dtype=np.float32) + np.float32(0.5)) * np.float32(patch / height) + np.float32(origin[2])
    X, Z = np.meshgrid(xs, zs)
    fx, fz = np.float32(2 * np.pi / 97.0), np.float32(2 * np.pi / 131.0)
    h = np.float32(8.0) * (np.float32(0.5) * np.sin(X * fx) * np.cos(Z * fz) + np.float32(0.5) * np.sin((X + Z) * np.float32(0.031)))
    dhdx = np.float32(8.0) * (np.float32(0.5) * fx * np.cos(X * fx) * np.cos(Z * fz) + np.float32(0.5 * 0.031) * np.cos((X + Z) * np.float32(0.031)))
    dhdz = np.float32(8.0) * (-np.float32(0.5) * fz * np.sin(X * fx) * np.sin(Z * fz) + np.float32(0.5 * 0.031) * np.cos((X + Z) * np.float32(0.031)))
    pos = np.zeros((height, width, 4), np.float32)
    pos[..., 0] = X; pos[..., 1] = np.float32(origin[1]) + h; pos[..., 2] = Z; pos[..., 3] = 1.0
    n = np.stack([-dhdx, np.ones_like(h), -dhdz], axis=-1).astype(np.float32)
    n /= np.sqrt((n * n).sum(-1, keepdims=True), dtype=np.float32)
    nrm = np.zeros((height, width, 4), np.float32); nrm[..., :3] = n
    idx = np.arange(width * height, dtype=np.uint64)
    alb = np.zeros((height * width, 4), np.uint8)
    for k in range(3):
        alb[:, k] = (255.0 * (0.2 + 0.7 * uniform(SEED_GBUF, idx, k))).astype(np.uint8)
    alb[:, 3] = 255
    return pos.reshape(-1, 4), nrm.reshape(-1, 4), alb


def synthetic_lights(n_spot=4096, n_point=0, patch=2048.0, origin=(1000.0, 1000.0, 1000.0), radius=40.0):
    i = np.arange(n_spot, dtype=np.uint64); j = np.arange(n_point, dtype=np.uint64) + np.uint64(1 << 20)
    u = lambda ix, s: uniform(SEED_LIGHTS, ix, s)
    L = dict(n_spot=n_spot, n_point=n_point)
    L["spot_pos"] = np.stack([origin[0] + patch * u(i, 0), origin[1] + 40.0 * u(i, 1), origin[2] + patch * u(i, 2)], axis=1).astype(np.float32)
    col = np.stack([0.2 + 0.8 * u(i, 3), 0.2 + 0.8 * u(i, 4), 0.2 + 0.8 * u(i, 5)], axis=1).astype(np.float32)
    L["spot_diffuse"] = col; L["spot_specular"] = col.copy()
    L["spot_ambient"] = np.concatenate([col, np.full((n_spot, 1), 0.25, np.float32)], axis=1).astype(np.float32)
    L["spot_linear"] = np.full(n_spot, 0.007, np.float32); L["spot_quadratic"] = np.full(n_spot, 0.0002, np.float32)
    L["spot_radius"] = np.full(n_spot, radius, np.float32)
    L["point_pos"] = np.stack([0.4 * u(j, 0) - 0.2, 0.6 + 0.4 * u(j, 1), 0.4 * u(j, 2) - 0.2], axis=1).astype(np.float32)   # near the unit sphere: the cone term uses normalize(frag) - pos
    L["point_dir"] = np.stack([u(j, 3) - 0.5, -0.5 - u(j, 4), u(j, 5) - 0.5], axis=1).astype(np.float32)
    pc = np.stack([0.2 + 0.8 * u(j, 6), 0.2 + 0.8 * u(j, 7), 0.2 + 0.8 * u(j, 8)], axis=1).astype(np.float32)
    L["point_diffuse"] = pc; L["point_specular"] = pc.copy()
    L["point_ambient"] = np.concatenate([pc, np.full((n_point, 1), 0.05, np.float32)], axis=1).astype(np.float32)
    L["point_linear"] = np.full(n_point, 0.0007, np.float32); L["point_quadratic"] = np.full(n_point, 0.000002, np.float32)
    L["point_cutoff"] = np.full(n_point, 0.3, np.float32); L["point_outer_cutoff"] = np.full(n_point, -0.2, np.float32)
    L["camera_pos"] = np.array([origin[0] + patch / 2, origin[1] + 300.0, origin[2] + patch / 2], np.float32)
    L["no_light_source_cutoff"] = 0.2; L["default_diffuse_factor"] = 0.2; L["any_light_source_visible"] = 1
    return L


def fill_lights_struct(S, L, keep):
    """fills a ctypes struct with the re_lights / ro_lights field layout from the dict of arrays"""
    S.n_spot, S.n_point = L["n_spot"], L["n_point"]
    for f in LIGHT_FIELDS:
        a = np.ascontiguousarray(L[f], np.float32); keep.append(a)
        setattr(S, f, a.ctypes.data_as(C.POINTER(C.c_float)))
    S.camera_pos[:] = [float(x) for x in L["camera_pos"]]
    S.no_light_source_cutoff = float(L["no_light_source_cutoff"]); S.default_diffuse_factor = float(L["default_diffuse_factor"])
    S.any_light_source_visible = int(L["any_light_source_visible"])
    return S


class DeferredLighting:
    def __init__(self, width, height, max_spot_lights=4096, max_point_lights=64, device=0):
        self._L = _capi.load()
        cfg = _capi.LightingConfig(device, width, height, max_spot_lights, max_point_lights)
        h = C.c_void_p()
        rc = self._L.re_lighting_create(C.byref(cfg), C.byref(h))
        if rc != 0:
            raise RenderEngineError(f"re_lighting_create failed ({rc}): {self._L.re_lighting_last_error(None).decode()}")
        self._h, self.width, self.height = h, width, height

    def _check(self, rc, what):
        if rc != 0:
            raise RenderEngineError(f"{what} failed ({rc}): {self._L.re_lighting_last_error(self._h).decode()}")

    def upload_gbuffer(self, pos, nrm, alb):
        pos = np.ascontiguousarray(pos, np.float32); nrm = np.ascontiguousarray(nrm, np.float32); alb = np.ascontiguousarray(alb, np.uint8)
        self._check(self._L.re_lighting_upload_gbuffer(self._h, pos.ctypes.data, nrm.ctypes.data, alb.ctypes.data), "re_lighting_upload_gbuffer")

    def set_lights(self, L):
        keep = []; S = fill_lights_struct(_capi.Lights(), L, keep)
        self._check(self._L.re_lighting_set_lights(self._h, C.byref(S)), "re_lighting_set_lights")

    def run(self):
        us = C.c_float()
        self._check(self._L.re_lighting_run(self._h, C.byref(us)), "re_lighting_run")
        return us.value

    def read(self):
        out = np.zeros((self.width * self.height, 4), np.float32)
        self._check(self._L.re_lighting_read(self._h, out.ctypes.data), "re_lighting_read")
        return out

    def read_pixels(self, idx):
        idx = np.ascontiguousarray(idx, np.uint32); out = np.zeros((len(idx), 4), np.float32)
        self._check(self._L.re_lighting_read_pixels(self._h, idx.ctypes.data, len(idx), out.ctypes.data), "re_lighting_read_pixels")
        return out

    def close(self):
        if getattr(self, "_h", None):
            self._L.re_lighting_destroy(self._h); self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
