"""ctypes binding of the CPU oracle (oracle/re_oracle.c).

TEST INFRASTRUCTURE ONLY.  Importable from tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py -- never from render_engine_amd/ (the product path).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libre_oracle.so")

# flags (mirror re_oracle.h)
F_STATIC, F_HAS_VEL, F_HAS_ACC, F_HAS_ROT = 0x001, 0x002, 0x004, 0x008
F_HAS_ROTVEL, F_HAS_ROTACC, F_HAS_SCALE, F_ALWAYS_EXEC = 0x010, 0x020, 0x040, 0x080
F_OOB_LOGIC, F_HAS_MOVED, F_HAS_ROTATED, F_USER = 0x100, 0x200, 0x400, 0x800
F_CAN_COLLIDE = 0x1000
F_LIGHT_DIRECTIONAL, F_LIGHT_POINT, F_LIGHT_SPOT = 0x2000, 0x4000, 0x8000      # FindLightType of the entity (light sets of its world section)

AABB_DT = np.dtype([("xmin", "f4"), ("xmax", "f4"), ("ymin", "f4"), ("ymax", "f4"), ("zmin", "f4"), ("zmax", "f4")])
ENTITY_DT = np.dtype([
    ("id", "u4"), ("model_index", "u4"), ("render_system", "u4"), ("sortable", "u4"), ("flags", "u4"),
    ("original", AABB_DT),
    ("pos", "f4", 3), ("rot_axis", "f4", 3), ("rot_angle", "f4"), ("scale", "f4", 3),
    ("vel", "f4", 3), ("acc", "f4", 3), ("rotvel_axis", "f4", 3), ("rotvel", "f4"),
    ("rotacc_axis", "f4", 3), ("rotacc", "f4"),
])
GROUP_DT = np.dtype([("model_index", "u4"), ("render_system", "u4"), ("sortable", "u4"), ("begin", "u4"), ("count", "u4")])


CHANGE_DT = np.dtype([("kind", "u4"), ("entity_id", "u4"), ("component", "u4"), ("pad", "u4"), ("value", "f4", (4,))])
CHANGE_MODIFY, CHANGE_DELETE, CHANGE_MAKE_STATIC, CHANGE_WAKE_UP = 0, 1, 2, 3


class Aabb(C.Structure):
    _fields_ = [(n, C.c_float) for n in ("xmin", "xmax", "ymin", "ymax", "zmin", "zmax")]

    def tup(self):
        return (self.xmin, self.xmax, self.ymin, self.ymax, self.zmin, self.zmax)


class Camera(C.Structure):
    _fields_ = [("pv", C.c_float * 16), ("pos", C.c_float * 3), ("dir", C.c_float * 3), ("far_draw", C.c_float),
                ("n_lod", C.c_uint32), ("lod_min", C.c_float * 8), ("lod_max", C.c_float * 8)]


class LightsC(C.Structure):
    _fields_ = [("n_spot", C.c_uint32), ("n_point", C.c_uint32)] + \
               [(n, C.POINTER(C.c_float)) for n in ("spot_pos", "spot_diffuse", "spot_specular", "spot_ambient", "spot_linear", "spot_quadratic", "spot_radius",
                                                     "point_pos", "point_dir", "point_diffuse", "point_specular", "point_ambient", "point_linear", "point_quadratic",
                                                     "point_cutoff", "point_outer_cutoff")] + \
               [("camera_pos", C.c_float * 3), ("no_light_source_cutoff", C.c_float), ("default_diffuse_factor", C.c_float), ("any_light_source_visible", C.c_uint32)]


def deferred_lighting(pos, nrm, alb, lights_struct, idx=None):
    """CPU evaluation of second_pass_frag.glsl for all pixels, or only the pixel indices in idx."""
    pos = np.ascontiguousarray(pos, np.float32); nrm = np.ascontiguousarray(nrm, np.float32); alb = np.ascontiguousarray(alb, np.uint8)
    npix = pos.shape[0]
    L = lib()
    L.ro_deferred_lighting.argtypes = [C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(LightsC), C.c_void_p, C.c_uint32, C.c_void_p]
    if idx is None:
        out = np.zeros((npix, 4), np.float32)
        L.ro_deferred_lighting(npix, pos.ctypes.data, nrm.ctypes.data, alb.ctypes.data, C.byref(lights_struct), None, 0, out.ctypes.data)
    else:
        idx = np.ascontiguousarray(idx, np.uint32); out = np.zeros((len(idx), 4), np.float32)
        L.ro_deferred_lighting(npix, pos.ctypes.data, nrm.ctypes.data, alb.ctypes.data, C.byref(lights_struct), idx.ctypes.data, len(idx), out.ctypes.data)
    return out


def lighting_spot_pairs(pos, lights_struct):
    """number of (pixel, spot light) pairs within the light radius: the pairs whose lighting terms the shader evaluates (exact)"""
    L = lib()
    pos = np.ascontiguousarray(pos, np.float32)
    L.ro_lighting_spot_pairs.restype = C.c_uint64
    L.ro_lighting_spot_pairs.argtypes = [C.c_uint32, C.c_void_p, C.POINTER(LightsC)]
    return int(L.ro_lighting_spot_pairs(len(pos), pos.ctypes.data, C.byref(lights_struct)))


def build(force=False):
    if force or not os.path.exists(_LIB_PATH) or \
            os.path.getmtime(_LIB_PATH) < max(os.path.getmtime(os.path.join(_HERE, f)) for f in ("re_oracle.c", "re_cpu_soa.c", "re_oracle.h")):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    L = C.CDLL(build())
    fp, u32p, u64p = C.POINTER(C.c_float), C.POINTER(C.c_uint32), C.POINTER(C.c_uint64)
    L.ro_sincosf.argtypes = [C.c_float, fp, fp]
    L.ro_norm3.restype = C.c_float; L.ro_norm3.argtypes = [C.c_float] * 3
    L.ro_mat4_mul.argtypes = [fp, fp, fp]
    L.ro_mat4_vec4.argtypes = [fp, fp, fp]
    L.ro_trs_matrix.argtypes = [fp, C.c_int, fp, C.c_float, C.c_int, fp, fp]
    L.ro_apply_transformation.restype = Aabb; L.ro_apply_transformation.argtypes = [Aabb, fp]
    L.ro_combine_aabb.restype = Aabb; L.ro_combine_aabb.argtypes = [Aabb, Aabb]
    L.ro_distance_to_aabb.restype = C.c_float; L.ro_distance_to_aabb.argtypes = [Aabb, fp]
    L.ro_make_planes.argtypes = [fp, fp]
    L.ro_frustum_aabb_visible.restype = C.c_int; L.ro_frustum_aabb_visible.argtypes = [fp, Aabb]
    L.ro_logic_aabb_in_view.restype = C.c_int; L.ro_logic_aabb_in_view.argtypes = [C.c_float, fp, Aabb]
    L.ro_lod_adjusted_model_index.restype = C.c_uint32
    L.ro_lod_adjusted_model_index.argtypes = [C.c_uint32, C.c_float, C.c_uint32, fp, fp]
    L.ro_default_lod.argtypes = [C.c_float, fp, fp]
    L.ro_max_level.restype = C.c_uint32; L.ro_max_level.argtypes = [C.c_uint32, C.c_uint32]
    L.ro_pack_key.restype = C.c_uint64; L.ro_pack_key.argtypes = [C.c_uint32] * 4
    L.ro_key_to_aabb.restype = Aabb; L.ro_key_to_aabb.argtypes = [C.c_uint64, C.c_uint32]
    L.ro_assign_cells.restype = C.c_int
    L.ro_assign_cells.argtypes = [Aabb, C.c_uint32, C.c_uint32, u64p, C.POINTER(C.c_int)]
    L.ro_perspective.argtypes = [C.c_float] * 4 + [fp]
    L.ro_look_at.argtypes = [fp, fp, fp, fp]
    L.ro_world_new.restype = C.c_void_p; L.ro_world_new.argtypes = [C.c_uint32, C.c_uint32]
    L.ro_world_free.argtypes = [C.c_void_p]
    L.ro_set_threads.argtypes = [C.c_void_p, C.c_int]
    L.ro_register_entities.restype = C.c_int; L.ro_register_entities.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p]
    L.ro_tree_add.restype = C.c_int; L.ro_tree_add.argtypes = [C.c_void_p, C.c_uint32, Aabb, C.c_int, C.c_int]
    L.ro_tree_remove.argtypes = [C.c_void_p, C.c_uint32]
    L.ro_end_of_changes.argtypes = [C.c_void_p]
    L.ro_num_cells.restype = C.c_uint32; L.ro_num_cells.argtypes = [C.c_void_p]
    L.ro_num_shared.restype = C.c_uint32; L.ro_num_shared.argtypes = [C.c_void_p]
    L.ro_get_cells.restype = C.c_uint32
    L.ro_get_cells.argtypes = [C.c_void_p, C.c_uint32] + [C.c_void_p] * 6
    L.ro_get_cell_entities.restype = C.c_uint32
    L.ro_get_cell_entities.argtypes = [C.c_void_p, C.c_uint64, C.c_uint32, C.c_void_p, u32p]
    L.ro_entity_lookup.restype = C.c_int; L.ro_entity_lookup.argtypes = [C.c_void_p, C.c_uint32, u64p, C.POINTER(C.c_int)]
    L.ro_get_entity.restype = C.c_int
    L.ro_get_entity.argtypes = [C.c_void_p, C.c_uint32, fp, C.POINTER(Aabb), fp, fp, fp, fp, u32p]
    L.ro_get_shared.restype = C.c_int
    L.ro_get_shared.argtypes = [C.c_void_p, C.c_uint32, u64p, C.POINTER(C.c_int), C.POINTER(Aabb), C.c_uint32, C.c_void_p, u32p, u32p]
    L.ro_frame_cull.restype = C.c_uint32; L.ro_frame_cull.argtypes = [C.c_void_p, C.POINTER(Camera), C.c_uint32, C.c_void_p]
    L.ro_frame_render.restype = C.c_uint32
    L.ro_frame_render.argtypes = [C.c_void_p, C.POINTER(Camera), C.c_int, C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, u32p]
    L.ro_frame_tick.restype = C.c_uint32
    L.ro_frame_tick.argtypes = [C.c_void_p, C.POINTER(Camera), C.c_float, C.c_uint32, C.c_void_p, u32p]
    L.ro_apply_changes.restype = C.c_uint32
    L.ro_apply_changes.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_int, C.c_uint32, C.c_void_p, u32p]
    L.ro_apply_changes_ex.restype = C.c_uint32
    L.ro_apply_changes_ex.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_int, C.c_uint32, C.c_void_p, u32p]
    L.ro_frame_collide.restype = C.c_uint32; L.ro_frame_collide.argtypes = [C.c_void_p, C.POINTER(Camera), C.c_uint32, C.c_void_p]
    L.ro_related_sections.restype = C.c_uint32; L.ro_related_sections.argtypes = [C.c_void_p, C.c_uint64, C.c_uint32, C.c_void_p]
    L.ro_find_related.restype = C.c_uint32
    L.ro_find_related.argtypes = [C.c_void_p, C.c_uint64, C.c_uint32, C.c_void_p, u32p, C.c_uint32, C.c_void_p, u32p]
    L.soa_build.restype = C.c_void_p
    L.soa_build.argtypes = [C.c_uint32] + [C.c_void_p] * 7 + [C.c_uint32, C.c_uint32, C.c_int]
    L.soa_free.argtypes = [C.c_void_p]
    L.soa_frame.restype = C.c_uint32
    L.soa_frame.argtypes = [C.c_void_p, C.POINTER(Camera), C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, u32p, u32p, C.c_void_p]
    _lib = L
    return L


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def aabb(t):
    return Aabb(*[float(x) for x in t])


def combine_aabb(a, b):
    """StaticAABB::combine_aabb (world/bounding_volumes/aabb.rs + world/dimension/range.rs:38-61), epsilon-biased union."""
    return np.array(lib().ro_combine_aabb(aabb(a), aabb(b)).tup(), dtype=np.float32)


def unpack_key(k):
    k = int(k)
    return ((k >> 48) & 0xFFFF, (k >> 32) & 0xFFFF, (k >> 16) & 0xFFFF, k & 0xFFFF)  # level, x, z, y


def pack_key(level, x, z, y):
    return (level << 48) | (x << 32) | (z << 16) | y


def sincos(x):
    s, c = C.c_float(), C.c_float()
    lib().ro_sincosf(np.float32(x), C.byref(s), C.byref(c))
    return np.float32(s.value), np.float32(c.value)


def trs_matrix(pos, axis=None, angle=0.0, scale=None):
    out = np.zeros(16, np.float32)
    p = np.asarray(pos, np.float32)
    a = np.asarray(axis if axis is not None else (1, 0, 0), np.float32)
    s = np.asarray(scale if scale is not None else (1, 1, 1), np.float32)
    lib().ro_trs_matrix(_fp(p), int(axis is not None), _fp(a), np.float32(angle), int(scale is not None), _fp(s), _fp(out))
    return out


def make_planes(pv):
    pv = np.ascontiguousarray(pv, np.float32).reshape(16)
    out = np.zeros(24, np.float32)
    lib().ro_make_planes(_fp(pv), _fp(out))
    return out.reshape(6, 4)


def perspective(aspect, fovy, near, far):
    out = np.zeros(16, np.float32)
    lib().ro_perspective(np.float32(aspect), np.float32(fovy), np.float32(near), np.float32(far), _fp(out))
    return out


def look_at(eye, target, up=(0, 1, 0)):
    out = np.zeros(16, np.float32)
    e, t, u = (np.asarray(v, np.float32) for v in (eye, target, up))
    lib().ro_look_at(_fp(e), _fp(t), _fp(u), _fp(out))
    return out


def mat4_mul(a, b):
    a = np.ascontiguousarray(a, np.float32).reshape(16); b = np.ascontiguousarray(b, np.float32).reshape(16)
    out = np.zeros(16, np.float32)
    lib().ro_mat4_mul(_fp(a), _fp(b), _fp(out))
    return out


def default_lod(render_distance):
    lo, hi = np.zeros(5, np.float32), np.zeros(5, np.float32)
    lib().ro_default_lod(np.float32(render_distance), _fp(lo), _fp(hi))
    return lo, hi


def make_camera(pos, direction, far, fov_deg=45.0, window=(1280, 720), near=0.1, lod=None, pv=None):
    """Camera as main.rs:25-30 / CameraBuilder::build (exports/camera_object.rs:341-386) would produce it."""
    cam = Camera()
    pos = np.asarray(pos, np.float32); d = np.asarray(direction, np.float32)
    if pv is None:
        proj = perspective(np.float32(window[0]) / np.float32(window[1]), np.float32(np.radians(np.float32(fov_deg))), near, far)
        view = look_at(pos, pos + d)
        pv = mat4_mul(proj, view)
    cam.pv[:] = [float(x) for x in np.asarray(pv, np.float32).reshape(16)]
    cam.pos[:] = [float(x) for x in pos]; cam.dir[:] = [float(x) for x in d]
    cam.far_draw = float(far)
    lo, hi = lod if lod is not None else default_lod(far)
    cam.n_lod = len(lo)
    for i in range(len(lo)):
        cam.lod_min[i] = float(lo[i]); cam.lod_max[i] = float(hi[i])
    return cam


class World:
    """BoundingBoxTree + ECS columns of the reference, CPU side."""

    def __init__(self, outline=16384, atomic=64, threads=1):
        self.L = lib()
        self.h = self.L.ro_world_new(outline, atomic)
        self.outline, self.atomic = outline, atomic
        self.L.ro_set_threads(self.h, threads)

    def close(self):
        if self.h:
            self.L.ro_world_free(self.h); self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def register(self, descs):
        descs = np.ascontiguousarray(descs, ENTITY_DT)
        return self.L.ro_register_entities(self.h, len(descs), descs.ctypes.data)

    def set_model_lod(self, model_index, render_system, lod_min, lod_max):
        """custom level-of-view bands of one model (register_model_with_render_system, render_flow.rs:1069-1076)"""
        lo = np.ascontiguousarray(lod_min, np.float32); hi = np.ascontiguousarray(lod_max, np.float32)
        self.L.ro_set_model_lod.restype = None
        self.L.ro_set_model_lod.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p]
        self.L.ro_set_model_lod(self.h, model_index, render_system, len(lo), lo.ctypes.data, hi.ctypes.data)

    def tree_add(self, eid, box, add_if_oob=False, is_static=False):
        return self.L.ro_tree_add(self.h, eid, aabb(box), int(add_if_oob), int(is_static))

    def tree_remove(self, eid):
        self.L.ro_tree_remove(self.h, eid)

    def end_of_changes(self):
        self.L.ro_end_of_changes(self.h)

    def cells(self):
        n = self.L.ro_num_cells(self.h)
        keys = np.zeros(n, np.uint64); tight = np.zeros(n, AABB_DT)
        nl, ns, nsh = (np.zeros(n, np.uint32) for _ in range(3)); st = np.zeros(n, np.uint8)
        self.L.ro_get_cells(self.h, n, keys.ctypes.data, tight.ctypes.data, nl.ctypes.data, ns.ctypes.data, nsh.ctypes.data, st.ctypes.data)
        return dict(keys=keys, tight=tight, n_local=nl, n_static=ns, n_shared=nsh, is_static_section=st)

    def cell_entities(self, key, cap=4096):
        ids = np.zeros(cap, np.uint32); nl = C.c_uint32()
        n = self.L.ro_get_cell_entities(self.h, int(key), cap, ids.ctypes.data, C.byref(nl))
        return ids[:nl.value].copy(), ids[nl.value:n].copy()

    def lookup(self, eid):
        keys = (C.c_uint64 * 8)(); nk = C.c_int()
        kind = self.L.ro_entity_lookup(self.h, eid, keys, C.byref(nk))
        return kind, [int(keys[i]) for i in range(nk.value)]

    def entity(self, eid):
        mat = np.zeros(16, np.float32); box = Aabb(); pos = np.zeros(3, np.float32)
        rot = np.zeros(4, np.float32); rv = np.zeros(4, np.float32); vel = np.zeros(3, np.float32); fl = C.c_uint32()
        ok = self.L.ro_get_entity(self.h, eid, _fp(mat), C.byref(box), _fp(pos), _fp(rot), _fp(rv), _fp(vel), C.byref(fl))
        if not ok:
            return None
        return dict(mat=mat, aabb=np.array(box.tup(), np.float32), pos=pos, rot=rot, rotvel=rv, vel=vel, flags=fl.value)

    def shared_sections(self, cap=4096):
        out = []
        for i in range(self.L.ro_num_shared(self.h)):
            keys = (C.c_uint64 * 8)(); nk = C.c_int(); box = Aabb(); ids = np.zeros(cap, np.uint32)
            na, ns = C.c_uint32(), C.c_uint32()
            self.L.ro_get_shared(self.h, i, keys, C.byref(nk), C.byref(box), cap, ids.ctypes.data, C.byref(na), C.byref(ns))
            out.append(dict(keys=[int(keys[k]) for k in range(nk.value)], aabb=box.tup(),
                            active=ids[:na.value].copy(), static=ids[na.value:na.value + ns.value].copy()))
        return out

    def cull(self, cam):
        n = self.L.ro_frame_cull(self.h, C.byref(cam), 0, None)
        keys = np.zeros(n, np.uint64)
        self.L.ro_frame_cull(self.h, C.byref(cam), n, keys.ctypes.data)
        return keys  # visible_sections_vec sorted, duplicates included

    def visible_lights(self, cam, type_flag, cap=65536):
        """ids of the lights of one type RenderFlow::render finds near the camera (flows/shadow_flow.rs:455-513), ascending"""
        ids = np.zeros(cap, np.uint32)
        self.L.ro_visible_lights.restype = C.c_uint32; self.L.ro_visible_lights.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p]
        n = self.L.ro_visible_lights(self.h, C.byref(cam), type_flag, cap, ids.ctypes.data)
        return ids[:min(n, cap)].copy()

    def render(self, cam, emit_duplicates=False, cap=None, gcap=4096):
        ng = C.c_uint32()
        groups = np.zeros(gcap, GROUP_DT)
        if cap is None:
            cap = self.L.ro_frame_render(self.h, C.byref(cam), int(emit_duplicates), 0, None, None, gcap, groups.ctypes.data, C.byref(ng))
        ids = np.zeros(cap, np.uint32); mats = np.zeros((cap, 16), np.float32)
        total = self.L.ro_frame_render(self.h, C.byref(cam), int(emit_duplicates), cap, ids.ctypes.data, mats.ctypes.data, gcap, groups.ctypes.data, C.byref(ng))
        return dict(total=total, ids=ids[:min(total, cap)], mats=mats[:min(total, cap)], groups=groups[:ng.value].copy())

    def apply_changes(self, changes, end_of_frame=True, cap=4096, added=None):
        """changes: structured array CHANGE_DT (kind, entity_id, component, pad, value[4]); added: ENTITY_DT array the ADD_ENTITY changes index (pad);
        returns (n entities re-placed, oob ids)"""
        ch = np.ascontiguousarray(changes, dtype=CHANGE_DT)
        oob = np.zeros(cap, np.uint32); noob = C.c_uint32()
        ad = np.ascontiguousarray(added, ENTITY_DT) if added is not None else np.zeros(0, ENTITY_DT)
        n = self.L.ro_apply_changes_ex(self.h, ch.ctypes.data, len(ch), ad.ctypes.data if len(ad) else None, len(ad), 1 if end_of_frame else 0, cap, oob.ctypes.data, C.byref(noob))
        return n, oob[:min(noob.value, cap)].copy()

    def collide(self, cam, cap=None):
        """handle_collisions of this frame (after cull, before tick): array [n, 2] of (this_entity, other_entity) invocations"""
        if cap is None:
            cap = self.L.ro_frame_collide(self.h, C.byref(cam), 0, None)
        pairs = np.zeros((max(cap, 1), 2), np.uint32)
        n = self.L.ro_frame_collide(self.h, C.byref(cam), cap, pairs.ctypes.data)
        return pairs[:min(n, cap)].copy()

    def related_sections(self, key):
        n = self.L.ro_related_sections(self.h, int(key), 0, None)
        out = np.zeros(max(n, 1), np.uint64)
        self.L.ro_related_sections(self.h, int(key), n, out.ctypes.data)
        return sorted(int(k) for k in out[:n])

    def find_related(self, key, cap=4096):
        """(unique section keys, [shared section key tuples]) of find_related_entities([key])"""
        uk = np.zeros(cap, np.uint64); sk = np.zeros(cap * 9, np.uint64); nu, ns = C.c_uint32(), C.c_uint32()
        self.L.ro_find_related(self.h, int(key), cap, uk.ctypes.data, C.byref(nu), cap * 9, sk.ctypes.data, C.byref(ns))
        shared, o = [], 0
        for _ in range(ns.value):
            nk = int(sk[o]); shared.append(tuple(int(k) for k in sk[o + 1:o + 1 + nk])); o += 1 + nk
        return [int(k) for k in uk[:nu.value]], shared

    def tick(self, cam, dt, cap=4096):
        oob = np.zeros(cap, np.uint32); noob = C.c_uint32()
        n = self.L.ro_frame_tick(self.h, C.byref(cam), np.float32(dt), cap, oob.ctypes.data, C.byref(noob))
        return n, oob[:min(noob.value, cap)].copy()


class SoaWorld:
    """the "optimised CPU" row (re_cpu_soa.c): sorted keys + SoA + OpenMP for worlds of static entities in unique sections.
    Built from an entity array and the oracle World that holds the same entities (matrices, AABBs and section keys are taken
    from it, so both agree by construction on everything except the data structures and the traversal)."""

    def __init__(self, world, ents, threads=1):
        self.L = lib()
        n = len(ents)
        keys = np.zeros(n, np.uint64); boxes = np.zeros(n, AABB_DT); mats = np.zeros((n, 16), np.float32)
        _soa_fill(world, ents, keys, boxes, mats)
        ids = np.ascontiguousarray(ents["id"], np.uint32); model = np.ascontiguousarray(ents["model_index"], np.uint32)
        rs = np.ascontiguousarray(ents["render_system"], np.uint32); srt = np.ascontiguousarray(ents["sortable"], np.uint32)
        self.h = self.L.soa_build(n, keys.ctypes.data, boxes.ctypes.data, ids.ctypes.data, model.ctypes.data, rs.ctypes.data, srt.ctypes.data, mats.ctypes.data,
                                  world.outline, world.atomic, threads)
        self.nsec = len(np.unique(keys)); self.mark = np.zeros(self.nsec + 1, np.uint8)

    def frame(self, cam, cap=0, gcap=4096):
        ids = np.zeros(max(cap, 1), np.uint32); mats = np.zeros((max(cap, 1), 16), np.float32); groups = np.zeros(gcap, GROUP_DT)
        ng, nvec = C.c_uint32(), C.c_uint32()
        total = self.L.soa_frame(self.h, C.byref(cam), cap, ids.ctypes.data if cap else None, mats.ctypes.data if cap else None, gcap, groups.ctypes.data,
                                 C.byref(ng), C.byref(nvec), self.mark.ctypes.data)
        return dict(total=total, ids=ids[:min(total, cap)], mats=mats[:min(total, cap)], groups=groups[:ng.value].copy(), n_visible_vec=nvec.value)

    def close(self):
        if self.h:
            self.L.soa_free(self.h); self.h = None


def _soa_fill(world, ents, keys, boxes, mats):
    """section key, StaticAABB and TransformationMatrix of every entity.  Static, translation-only entities inside one section
    (the lattice configs): the matrix is the translation, the AABB is OriginalAABB + position (apply_transformation of the two
    corners), the key follows from the AABB minimum -- checked against the oracle on a sample."""
    a = world.atomic
    pos = np.ascontiguousarray(ents["pos"], np.float32)
    org = np.ascontiguousarray(ents["original"]).view(np.float32).reshape(len(ents), 6)
    mats[:] = 0; mats[:, 0] = mats[:, 5] = mats[:, 10] = mats[:, 15] = 1; mats[:, 12:15] = pos
    for k, (lo, hi) in enumerate(((0, 1), (2, 3), (4, 5))):
        boxes[("xmin", "ymin", "zmin")[k]] = org[:, lo] * np.float32(1.0) + pos[:, k]
        boxes[("xmax", "ymax", "zmax")[k]] = org[:, hi] * np.float32(1.0) + pos[:, k]
    cx = np.floor(boxes["xmin"] / a).astype(np.uint64); cy = np.floor(boxes["ymin"] / a).astype(np.uint64); cz = np.floor(boxes["zmin"] / a).astype(np.uint64)
    assert (np.floor(boxes["xmax"] / a) == cx).all() and (np.floor(boxes["ymax"] / a) == cy).all() and (np.floor(boxes["zmax"] / a) == cz).all(), "SoaWorld: entities must lie inside one level-0 section"
    keys[:] = (cx << np.uint64(32)) | (cz << np.uint64(16)) | cy
    step = max(1, len(ents) // 64)
    for e in ents[::step]:
        kind, ks = world.lookup(int(e["id"])); st = world.entity(int(e["id"]))
        i = int(np.nonzero(ents["id"] == e["id"])[0][0])
        assert kind == 1 and ks[0] == int(keys[i]) and (st["mat"] == mats[i]).all() and (st["aabb"] == np.array([boxes[i][f] for f in ("xmin", "xmax", "ymin", "ymax", "zmin", "zmax")], np.float32)).all()
