"""Host-side mirror of the reference interface for the visible-set path, over the C ABI.

Names follow the reference: `Pipeline.register_model_instances` (flows/pipeline.rs:186-208),
`Pipeline.execute` = cull + render gather/pack + logic tick (flows/pipeline.rs:212-276), `Camera`
built like `CameraBuilder::build` (exports/camera_object.rs:341-386), `CullResult`,
`InstanceRange`.  All compute happens in librender_engine_hip.so; this file only marshals.
"""
import ctypes as C

import numpy as np

from . import _capi

# One entity as EntityTransformationBuilder fills it (exports/entity_transformer.rs:12-29)
ENTITY_DT = np.dtype([
    ("id", "u4"), ("model_index", "u4"), ("render_system", "u4"), ("sortable", "u4"), ("flags", "u4"),
    ("original", "f4", 6),           # xmin,xmax,ymin,ymax,zmin,zmax
    ("pos", "f4", 3), ("rot_axis", "f4", 3), ("rot_angle", "f4"), ("scale", "f4", 3),
    ("vel", "f4", 3), ("acc", "f4", 3), ("rotvel_axis", "f4", 3), ("rotvel", "f4"),
    ("rotacc_axis", "f4", 3), ("rotacc", "f4"),
])


class RenderEngineError(RuntimeError):
    pass


def _f32(x):
    return np.float32(x)


def perspective(aspect, fovy, near, far):
    """nalgebra Perspective3::new(aspect, fovy, znear, zfar) (column-major 16 floats)."""
    aspect, fovy, near, far = map(_f32, (aspect, fovy, near, far))
    m = np.zeros(16, np.float32)
    m11 = _f32(1.0) / _f32(np.tan(fovy / _f32(2.0)))
    m[5] = m11; m[0] = m11 / aspect
    m[10] = (far + near) / (near - far)
    m[14] = far * near * _f32(2.0) / (near - far)
    m[11] = -1.0
    return m


def look_at(eye, target, up=(0.0, 1.0, 0.0)):
    """Right-handed look-at view matrix (glm::look_at convention), column-major."""
    eye, target, up = (np.asarray(v, np.float32) for v in (eye, target, up))

    def nrm(v):
        return v / np.sqrt((v[0] * v[0] + v[1] * v[1]) + v[2] * v[2], dtype=np.float32)
    f = nrm(target - eye)
    s = nrm(np.array([f[1] * up[2] - f[2] * up[1], f[2] * up[0] - f[0] * up[2], f[0] * up[1] - f[1] * up[0]], np.float32))
    u = np.array([s[1] * f[2] - s[2] * f[1], s[2] * f[0] - s[0] * f[2], s[0] * f[1] - s[1] * f[0]], np.float32)
    m = np.zeros(16, np.float32); m[15] = 1.0
    m[0], m[4], m[8] = s; m[1], m[5], m[9] = u; m[2], m[6], m[10] = -f
    m[12] = -((s[0] * eye[0] + s[1] * eye[1]) + s[2] * eye[2])
    m[13] = -((u[0] * eye[0] + u[1] * eye[1]) + u[2] * eye[2])
    m[14] = ((f[0] * eye[0] + f[1] * eye[1]) + f[2] * eye[2])
    return m


def mat4_mul(a, b):
    """Column-major product with nalgebra's accumulation order (k = 0..3, separately rounded)."""
    a = np.asarray(a, np.float32).reshape(4, 4); b = np.asarray(b, np.float32).reshape(4, 4)   # [col][row]
    out = np.zeros((4, 4), np.float32)
    for j in range(4):
        for i in range(4):
            y = a[0, i] * b[j, 0]
            y = a[1, i] * b[j, 1] + y
            y = a[2, i] * b[j, 2] + y
            y = a[3, i] * b[j, 3] + y
            out[j, i] = y
    return out.reshape(16)


def create_level_of_views(render_distance):
    """prelude/default_render_system.rs:240-256"""
    rd = _f32(render_distance)
    v1 = rd * _f32(0.10); v2 = rd * _f32(0.15) + v1; v3 = rd * _f32(0.20) + v2; v4 = rd * _f32(0.25) + v3; v5 = rd * _f32(0.30) + v4
    return np.array([0.0, v1, v2, v3, v4], np.float32), np.array([v1, v2, v3, v4, v5], np.float32)


CHANGE_DT = np.dtype([("kind", "u4"), ("entity_id", "u4"), ("component", "u4"), ("reserved", "u4"), ("value", "f4", (4,))])   # re_change


class Camera:
    """CameraBuilder defaults of the sample game: fov 45 deg, near 0.1 (main.rs:25-30)."""

    def __init__(self, position, direction, far_draw_distance, fov_degrees=45.0, window_dimensions=(1280, 720),
                 near_draw_distance=0.1, level_of_views=None, projection_view=None):
        self.position = np.asarray(position, np.float32)
        self.direction = np.asarray(direction, np.float32)
        self.far_draw_distance = float(far_draw_distance)
        if projection_view is None:
            proj = perspective(_f32(window_dimensions[0]) / _f32(window_dimensions[1]), np.radians(_f32(fov_degrees)),
                               near_draw_distance, far_draw_distance)
            view = look_at(self.position, self.position + self.direction)
            projection_view = mat4_mul(proj, view)
        self.projection_view = np.asarray(projection_view, np.float32).reshape(16)
        self.level_of_views = level_of_views if level_of_views is not None else create_level_of_views(far_draw_distance)

    def to_c(self):
        c = _capi.CameraC()
        c.projection_view[:] = [float(x) for x in self.projection_view]
        c.position[:] = [float(x) for x in self.position]; c.direction[:] = [float(x) for x in self.direction]
        c.far_draw = self.far_draw_distance
        lo, hi = self.level_of_views
        c.n_lod = len(lo)
        for i in range(len(lo)):
            c.lod_min[i] = float(lo[i]); c.lod_max[i] = float(hi[i])
        return c


def section_keys(ents, tree_outline_length=16384, tree_atomic_length=64):
    """re_section_keys: (keys [n, 8] uint64, n_keys [n] uint8) -- the world sections every entity (ENTITY_DT array) is registered in: one key (its
    unique section) or the 2..8 sections its shared section links; n_keys 0 = rejected as out of bounds.  Host arithmetic; no GPU needed."""
    L = _capi.load()
    e = np.ascontiguousarray(ents, ENTITY_DT); n = len(e)
    keep = [np.ascontiguousarray(e["flags"]), np.ascontiguousarray(e["original"], np.float32), np.ascontiguousarray(e["pos"], np.float32),
            np.ascontiguousarray(np.concatenate([e["rot_axis"], e["rot_angle"][:, None]], axis=1), np.float32) if n else np.zeros((0, 4), np.float32),
            np.ascontiguousarray(e["scale"], np.float32)]
    E = _capi.Entities(); E.n = n
    E.flags = keep[0].ctypes.data_as(C.POINTER(C.c_uint32)); E.original_aabb = keep[1].ctypes.data_as(C.POINTER(C.c_float)); E.position = keep[2].ctypes.data_as(C.POINTER(C.c_float))
    E.rotation = keep[3].ctypes.data_as(C.POINTER(C.c_float)); E.scale = keep[4].ctypes.data_as(C.POINTER(C.c_float))
    cfg = _capi.Config(0, tree_outline_length, tree_atomic_length, 0, 0)
    keys = np.zeros((max(n, 1), 8), np.uint64); nk = np.zeros(max(n, 1), np.uint8)
    rc = L.re_section_keys(C.byref(cfg), C.byref(E), keys.ctypes.data, nk.ctypes.data)
    if rc != _capi.RE_OK:
        raise RenderEngineError(f"re_section_keys failed ({rc})")
    return keys[:n], nk[:n]


def first_section_keys(ents, tree_outline_length=16384, tree_atomic_length=64):
    """the smallest section key of every entity (0 = rejected): the key that decides the owning shard"""
    keys, nk = section_keys(ents, tree_outline_length, tree_atomic_length)
    k = np.where(np.arange(8)[None, :] < nk[:, None], keys, np.uint64(0xFFFFFFFFFFFFFFFF)).min(axis=1) if len(keys) else np.zeros(0, np.uint64)
    return np.where(nk > 0, k, np.uint64(0)).astype(np.uint64)


def shard_by_first_section(ents, n_shards, tree_outline_length=16384, tree_atomic_length=64, halo=False):
    """Shards of a world for `n_shards` GPUs: contiguous ranges of the entities' smallest section key with near-equal entity counts; entities that
    share a first section stay together (a section's tight AABB folds all of its entities: SURVEY 8e).  Returns one index array per shard -- or,
    with halo=True, (own, halo) pairs: `halo` are the entities another shard owns that this shard uploads as replicas with F_PHANTOM, because the
    static-section flag of a unique section depends on all of its entities and on every shared section linking it: for each shared section the
    shard owns, the members of the unique sections it links and the members of the other shared sections linking those."""
    keys, nk = section_keys(ents, tree_outline_length, tree_atomic_length)
    first = np.where(np.arange(8)[None, :] < nk[:, None], keys, np.uint64(0xFFFFFFFFFFFFFFFF)).min(axis=1)
    first = np.where(nk > 0, first, np.uint64(0))
    order = np.argsort(first, kind="stable")
    ks = first[order]; n = len(ks)
    cuts = [0]
    for r in range(1, n_shards):
        c = max((n * r) // n_shards, cuts[-1])
        while 0 < c < n and ks[c] == ks[c - 1]:
            c += 1                                              # never split the entities of one section
        cuts.append(min(c, n))
    cuts.append(n)
    own = [np.sort(order[cuts[r]:cuts[r + 1]]) for r in range(n_shards)]
    if not halo:
        return own
    owner = np.zeros(n, np.int64)
    for r, idx in enumerate(own):
        owner[idx] = r
    unique_members, linking = {}, {}                            # section key -> entities registered in it alone / shared-section members linking it
    for i in range(n):
        if nk[i] == 1:
            unique_members.setdefault(int(keys[i, 0]), []).append(i)
        elif nk[i] > 1:
            for k in keys[i, :nk[i]]:
                linking.setdefault(int(k), []).append(i)
    out = []
    for r, idx in enumerate(own):
        linked = {int(k) for i in idx if nk[i] > 1 for k in keys[i, :nk[i]]}          # unique sections the shard's shared sections link
        h = set()
        for k in linked:
            h.update(i for i in unique_members.get(k, ()) if owner[i] != r)
            h.update(i for i in linking.get(k, ()) if owner[i] != r)
        out.append((idx, np.array(sorted(h), np.int64)))
    return out


class Pipeline:
    """One GPU's share of the world: BoundingBoxTree + ECS columns resident in HBM."""

    def __init__(self, tree_outline_length=16384, tree_atomic_length=64, device=0, max_instances=0, flags=0):
        self._L = _capi.load()
        cfg = _capi.Config(device, tree_outline_length, tree_atomic_length, max_instances, flags)
        h = C.c_void_p()
        rc = self._L.re_create(C.byref(cfg), C.byref(h))
        if rc != _capi.RE_OK:
            raise RenderEngineError(f"re_create failed ({rc}): {self._L.re_last_error(None).decode()}")
        self._h = h
        self.outline_length, self.atomic_length = tree_outline_length, tree_atomic_length
        self._keep = None

    def close(self):
        if getattr(self, "_h", None):
            self._L.re_destroy(self._h); self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc, what):
        if rc != _capi.RE_OK:
            raise RenderEngineError(f"{what} failed ({rc}): {self._L.re_last_error(self._h).decode()}")

    # -- Pipeline::register_model_instances ------------------------------------------------------
    @staticmethod
    def _columns(ents):
        e = np.ascontiguousarray(ents, ENTITY_DT)
        n = len(e)
        return n, dict(
            entity_id=np.ascontiguousarray(e["id"]), model_index=np.ascontiguousarray(e["model_index"]),
            render_system=np.ascontiguousarray(e["render_system"]), sortable=np.ascontiguousarray(e["sortable"]),
            flags=np.ascontiguousarray(e["flags"]), original_aabb=np.ascontiguousarray(e["original"]),
            position=np.ascontiguousarray(e["pos"]),
            rotation=np.ascontiguousarray(np.concatenate([e["rot_axis"], e["rot_angle"][:, None]], axis=1)) if n else np.zeros((0, 4), np.float32),
            scale=np.ascontiguousarray(e["scale"]), velocity=np.ascontiguousarray(e["vel"]), acceleration=np.ascontiguousarray(e["acc"]),
            rotation_velocity=np.ascontiguousarray(np.concatenate([e["rotvel_axis"], e["rotvel"][:, None]], axis=1)) if n else np.zeros((0, 4), np.float32),
            rotation_acceleration=np.ascontiguousarray(np.concatenate([e["rotacc_axis"], e["rotacc"][:, None]], axis=1)) if n else np.zeros((0, 4), np.float32),
        )

    @staticmethod
    def _entities_struct(n, entity_id, model_index, flags, original_aabb, position, render_system=None, sortable=None,
                         rotation=None, scale=None, velocity=None, acceleration=None, rotation_velocity=None, rotation_acceleration=None):
        E = _capi.Entities(); E.n = n
        keep = []

        def u32(a):
            if a is None:
                return None
            a = np.ascontiguousarray(a, np.uint32); keep.append(a); return a.ctypes.data_as(C.POINTER(C.c_uint32))

        def f32(a):
            if a is None:
                return None
            a = np.ascontiguousarray(a, np.float32); keep.append(a); return a.ctypes.data_as(C.POINTER(C.c_float))
        E.entity_id = u32(entity_id); E.model_index = u32(model_index); E.render_system = u32(render_system); E.sortable = u32(sortable)
        E.flags = u32(flags); E.original_aabb = f32(original_aabb); E.position = f32(position); E.rotation = f32(rotation)
        E.scale = f32(scale); E.velocity = f32(velocity); E.acceleration = f32(acceleration)
        E.rotation_velocity = f32(rotation_velocity); E.rotation_acceleration = f32(rotation_acceleration)
        return E, keep

    def register_model_instances(self, ents):
        """Pipeline::register_model_instances (flows/pipeline.rs:186-208): create_entity + apply_choices per instance, then end_of_changes.  ents: numpy
        structured array of ENTITY_DT.  The first call of a pipeline uploads the world; later calls -- at any time, as in the reference -- APPEND
        (re_add_entities).  Returns the number of instances rejected as out of bounds."""
        n, cols = self._columns(ents)
        E, keep = self._entities_struct(n, **cols)
        rej = C.c_uint32()
        self._check(self._L.re_add_entities(self._h, C.byref(E), C.byref(rej)), "re_add_entities")
        return rej.value

    def replace_world(self, ents):
        """re_upload_entities: the world of this context is REPLACED by these entities (a fresh tree, a fresh static render cache)"""
        n, cols = self._columns(ents)
        return self.upload_columns(n, **cols)

    def upload_columns(self, n, entity_id, model_index, flags, original_aabb, position, render_system=None, sortable=None,
                       rotation=None, scale=None, velocity=None, acceleration=None, rotation_velocity=None, rotation_acceleration=None):
        """SoA upload straight into re_upload_entities (no structured-array staging)."""
        E, keep = self._entities_struct(n, entity_id, model_index, flags, original_aabb, position, render_system, sortable,
                                        rotation, scale, velocity, acceleration, rotation_velocity, rotation_acceleration)
        rej = C.c_uint32()
        self._check(self._L.re_upload_entities(self._h, C.byref(E), C.byref(rej)), "re_upload_entities")
        return rej.value

    def set_model_lod(self, model_index, render_system, lod_min, lod_max):
        """custom level-of-view bands of one model (register_model_with_render_system(.., custom_level_of_view, ..), render_flow.rs:1069-1076)"""
        lo = np.ascontiguousarray(lod_min, np.float32); hi = np.ascontiguousarray(lod_max, np.float32)
        self._check(self._L.re_set_model_lod(self._h, model_index, render_system, len(lo), lo.ctypes.data_as(C.POINTER(C.c_float)), hi.ctypes.data_as(C.POINTER(C.c_float))), "re_set_model_lod")

    # -- Pipeline::execute, split at the reference's own seams -----------------------------------
    def cull_and_pack(self, camera, emit_duplicates=False, asynchronous=False, copy=True, force_large_pack=False, defer_pack=False, two_lanes=False, one_launch=False):
        """pipeline.rs:216-229 + render_flow.rs:401-410.  Returns dict(total, ids, mats, groups, ...)."""
        cam = camera if isinstance(camera, _capi.CameraC) else camera.to_c()
        vis = _capi.Visible()
        flags = (_capi.CULL_EMIT_DUPLICATES if emit_duplicates else 0) | (_capi.CULL_ASYNC if asynchronous else 0) | \
            (_capi.CULL_FORCE_LARGE_PACK if force_large_pack else 0) | (_capi.CULL_DEFER_PACK if defer_pack else 0) | (_capi.CULL_TWO_LANES if two_lanes else 0) | (_capi.CULL_ONE_LAUNCH if one_launch else 0)
        self._check(self._L.re_cull_pack(self._h, C.byref(cam), flags, C.byref(vis)), "re_cull_pack")
        if asynchronous:
            return None
        return self._visible_to_py(vis, copy)

    def _visible_to_py(self, vis, copy=True):
        groups = np.zeros(vis.n_groups, dtype=[("model_index", "u4"), ("render_system", "u4"), ("sortable", "u4"), ("begin", "u4"), ("count", "u4")])
        for g in range(vis.n_groups):
            r = vis.groups[g]
            groups[g] = (r.model_index, r.render_system, r.sortable, r.begin_instance, r.count)
        out = dict(n_visible_sections=vis.n_visible_sections, n_visible_vec=vis.n_visible_vec, total=vis.n_instances,
                   n_written=vis.n_written, groups=groups, d_entity_ids=vis.d_entity_ids, d_matrices=vis.d_matrices)
        if copy:
            ids = np.zeros(vis.n_written, np.uint32); mats = np.zeros((vis.n_written, 16), np.float32); nw = C.c_uint32()
            self._check(self._L.re_copy_visible(self._h, ids.ctypes.data, mats.ctypes.data, vis.n_written, C.byref(nw)), "re_copy_visible")
            out["ids"], out["mats"] = ids[:nw.value], mats[:nw.value]
        return out

    def tick(self, delta_time, all_dynamic=False, asynchronous=False):
        """logic_flow.rs:230 update_positions + :255 apply_change."""
        tr = _capi.TickResult()
        flags = (_capi.TICK_ALL_DYNAMIC if all_dynamic else 0) | (_capi.TICK_ASYNC if asynchronous else 0)
        self._check(self._L.re_tick(self._h, np.float32(delta_time), flags, C.byref(tr)), "re_tick")
        return None if asynchronous else dict(n_changed=tr.n_changed, n_rebucket=tr.n_rebucket, n_out_of_bounds=tr.n_out_of_bounds)

    def run_frames(self, camera, n, delta_time=0.016, cull_flags=0, tick_flags=0):
        """n frames of cull_and_pack + tick driven from native code (re_run_frames): per-frame wall times in microseconds, and the last
        frame's results when the calls were synchronous"""
        cam = camera if isinstance(camera, _capi.CameraC) else camera.to_c()
        us = np.zeros(max(n, 1), np.float32); vis = _capi.Visible(); tr = _capi.TickResult()
        self._check(self._L.re_run_frames(self._h, C.byref(cam), np.float32(delta_time), cull_flags, tick_flags, n, us.ctypes.data, C.byref(vis), C.byref(tr)), "re_run_frames")
        sync = not (cull_flags & _capi.CULL_ASYNC)
        return us[:n], (self._visible_to_py(vis, copy=False) if sync and n else None), dict(n_changed=tr.n_changed, n_rebucket=tr.n_rebucket, n_out_of_bounds=tr.n_out_of_bounds)

    def apply_changes(self, changes, added=None):
        """helper_things/entity_change_helpers.rs:32-189 for the change requests of user logic.
        `changes`: structured array CHANGE_DT (kind, entity_id, component, reserved, value[4]); `added`: ENTITY_DT array of the entities that
        CHANGE_ADD_ENTITY changes add (reserved = index into it)."""
        ch = np.ascontiguousarray(changes, dtype=CHANGE_DT)
        tr = _capi.TickResult()
        if added is not None:
            n, cols = self._columns(added)
            E, keep = self._entities_struct(n, **cols)
            self._check(self._L.re_apply_changes_ex(self._h, ch.ctypes.data, len(ch), C.byref(E), 0, C.byref(tr)), "re_apply_changes_ex")
        else:
            self._check(self._L.re_apply_changes(self._h, ch.ctypes.data, len(ch), 0, C.byref(tr)), "re_apply_changes")
        return dict(n_changed=tr.n_changed, n_rebucket=tr.n_rebucket, n_out_of_bounds=tr.n_out_of_bounds)

    def collide(self, capacity=None):
        """LogicFlow::handle_collisions (flows/logic_flow.rs:452-651) of the frame, between cull_and_pack and tick: array [n, 2] of
        the (this_entity, other_entity) arguments of every collision-logic invocation, in no particular order"""
        n = C.c_uint32()
        if capacity is None:
            self._check(self._L.re_collide(self._h, 0, None, 0, C.byref(n)), "re_collide")
            capacity = n.value
        pairs = np.zeros((max(capacity, 1), 2), np.uint32)
        self._check(self._L.re_collide(self._h, 0, pairs.ctypes.data, capacity, C.byref(n)), "re_collide")
        return pairs[:min(n.value, capacity)].copy(), n.value

    # -- multi-GPU exchange (RCCL behind the C ABI) ---------------------------------------------------
    @staticmethod
    def comm_unique_id():
        """rank 0: the 128-byte id every rank passes to comm_init (the host distributes it over its own channel)"""
        L = _capi.load()
        buf = (C.c_uint8 * _capi.COMM_ID_BYTES)()
        rc = L.re_comm_unique_id(buf)
        if rc != _capi.RE_OK:
            raise RenderEngineError(f"re_comm_unique_id failed ({rc}): {L.re_last_error(None).decode()}")
        return bytes(buf)

    def comm_init(self, unique_id, rank, n_ranks, slab_instances):
        buf = (C.c_uint8 * _capi.COMM_ID_BYTES).from_buffer_copy(unique_id)
        self._check(self._L.re_comm_init(self._h, buf, rank, n_ranks, slab_instances), "re_comm_init")

    def allgather_visible(self, asynchronous=False, copy=True):
        g = _capi.Gathered()
        self._check(self._L.re_allgather_visible(self._h, _capi.GATHER_ASYNC if asynchronous else 0, C.byref(g)), "re_allgather_visible")
        return None if asynchronous else self._gathered_to_py(g, copy)

    def gather_wait(self, copy=True):
        g = _capi.Gathered()
        self._check(self._L.re_gather_wait(self._h, C.byref(g)), "re_gather_wait")
        return self._gathered_to_py(g, copy)

    def _gathered_to_py(self, g, copy):
        counts = [int(g.counts[r]) for r in range(g.n_ranks)]
        out = dict(counts=counts, overflowed=bool(g.overflowed), d_entity_ids=g.d_entity_ids, d_matrices=g.d_matrices,
                   ids_rank_stride=g.ids_rank_stride, matrices_rank_stride=g.matrices_rank_stride)
        if copy:                                           # rank-ordered concatenation on the host
            ids, mats = [], []
            for r, n in enumerate(counts):
                ids.append(_capi.device_to_host(self._h, g.d_entity_ids + 4 * r * g.ids_rank_stride, 4 * n).view(np.uint32))
                mats.append(_capi.device_to_host(self._h, g.d_matrices + 4 * r * g.matrices_rank_stride, 64 * n).view(np.float32).reshape(n, 16))
            out["ids"] = np.concatenate(ids) if ids else np.zeros(0, np.uint32)
            out["mats"] = np.concatenate(mats) if mats else np.zeros((0, 16), np.float32)
        return out

    # -- entities that change GPU (the second, sparse exchange) -----------------------------
    def set_shard_range(self, key_lo, key_hi):
        """this pipeline owns the world sections with key_lo <= key < key_hi (the smallest key of an entity's sections decides)"""
        self._check(self._L.re_set_shard_range(self._h, int(key_lo), int(key_hi)), "re_set_shard_range")

    def list_migrants(self, capacity=65536):
        """ids of the entities re-bucketed since the last call whose section now belongs to another pipeline's key range"""
        ids = np.zeros(max(capacity, 1), np.uint32); n = C.c_uint32()
        self._check(self._L.re_list_migrants(self._h, ids.ctypes.data, capacity, C.byref(n)), "re_list_migrants")
        if n.value > capacity:
            return self.list_migrants(n.value)
        return ids[:n.value].copy()

    def export_entities(self, ids):
        """the complete current state of these entities as an ENTITY_DT array (re_entity_state has the same 140-byte layout): what register_model_instances
        of another pipeline needs to take them over"""
        ids = np.ascontiguousarray(ids, np.uint32)
        out = np.zeros(len(ids), ENTITY_DT)
        assert ENTITY_DT.itemsize == 140
        if len(ids):
            self._check(self._L.re_export_entities(self._h, ids.ctypes.data, len(ids), out.ctypes.data), "re_export_entities")
        return out

    def take_migrants(self):
        """list + export + remove: the entities this pipeline hands over, as ENTITY_DT records"""
        ids = self.list_migrants()
        if not len(ids):
            return np.zeros(0, ENTITY_DT)
        states = self.export_entities(ids)
        ch = np.zeros(len(ids), CHANGE_DT); ch["kind"] = _capi.CHANGE_DELETE; ch["entity_id"] = ids
        self.apply_changes(ch)
        return states

    def wait(self, copy=False):
        vis = _capi.Visible(); tr = _capi.TickResult()
        self._check(self._L.re_wait(self._h, C.byref(vis), C.byref(tr)), "re_wait")
        return self._visible_to_py(vis, copy), dict(n_changed=tr.n_changed, n_rebucket=tr.n_rebucket, n_out_of_bounds=tr.n_out_of_bounds)

    def set_output_buffers(self, ids_ptr, mats_ptr, capacity):
        self._check(self._L.re_set_output_buffers(self._h, ids_ptr, mats_ptr, capacity), "re_set_output_buffers")

    def set_output_count(self, count_ptr):
        self._check(self._L.re_set_output_count(self._h, count_ptr), "re_set_output_count")

    # -- ECS read-back -----------------------------------------------------------------------------
    def read_component(self, entity_id, component):
        if component == _capi.C_FLAGS:
            v = np.zeros(1, np.uint32)
        else:
            v = np.zeros(_capi.COMPONENT_FLOATS[component], np.float32)
        self._check(self._L.re_read_component(self._h, entity_id, component, v.ctypes.data), "re_read_component")
        return v

    def has_component(self, entity_id, component):
        """ECS::check_component_written (objects/ecs.rs:348-367)"""
        bit = {_capi.C_POSITION: "POSITION", _capi.C_ROTATION: "ROTATION", _capi.C_SCALE: "SCALE", _capi.C_VELOCITY: "VELOCITY", _capi.C_ACCELERATION: "ACCELERATION",
               _capi.C_ROTATION_VEL: "VELOCITY_ROTATION", _capi.C_ROTATION_ACC: "ACCELERATION_ROTATION", _capi.C_TRANSFORMATION: "TRANSFORMATION",
               _capi.C_STATIC_AABB: "STATIC_AABB", _capi.C_ORIGINAL_AABB: "ORIGINAL_AABB"}[component]
        return bool(self.ecs_bitset(entity_id) >> _capi.ECS_BIT[bit] & 1)

    def ecs_bitset(self, entity_id):
        """ecs.bitsets[entity] (objects/ecs.rs:61-72) in the reference's registration order"""
        b = C.c_uint32()
        self._check(self._L.re_ecs_bitset(self._h, entity_id, C.byref(b)), "re_ecs_bitset")
        return b.value

    def visible_lights(self, cam, light_type, capacity=65536):
        """ids (ascending) of the lights of one type (F_LIGHT_*) RenderFlow::render finds near the camera: flows/shadow_flow.rs:455-513"""
        camc = cam.to_c() if hasattr(cam, "to_c") else cam
        ids = np.zeros(max(capacity, 1), np.uint32); n = C.c_uint32()
        self._check(self._L.re_visible_lights(self._h, C.byref(camc), light_type, ids.ctypes.data, capacity, C.byref(n)), "re_visible_lights")
        return ids[:min(n.value, capacity)].copy()

    def get_indexes_for_components(self, components):
        """ECS::get_indexes_for_components (objects/ecs.rs:238-285): ascending entity ids carrying all the components"""
        comps = (C.c_int * max(len(components), 1))(*components)
        n = C.c_uint32()
        self._check(self._L.re_ecs_query(self._h, comps, len(components), None, 0, C.byref(n)), "re_ecs_query")
        ids = np.zeros(max(n.value, 1), np.uint32)
        self._check(self._L.re_ecs_query(self._h, comps, len(components), ids.ctypes.data, n.value, C.byref(n)), "re_ecs_query")
        return ids[:n.value]

    def out_of_bounds(self, cap=4096):
        ids = np.zeros(cap, np.uint32); n = C.c_uint32()
        self._check(self._L.re_get_out_of_bounds(self._h, ids.ctypes.data, cap, C.byref(n)), "re_get_out_of_bounds")
        return ids[:min(n.value, cap)]

    # -- introspection -----------------------------------------------------------------------------
    def stats(self):
        s = _capi.Stats(); self._check(self._L.re_get_stats(self._h, C.byref(s)), "re_get_stats")
        return {k: getattr(s, k) for k, _ in s._fields_}

    def sections(self):
        n = self.stats()["n_sections"]
        keys = np.zeros(n, np.uint64); tight = np.zeros((n, 6), np.float32); nl = np.zeros(n, np.uint32); ns = np.zeros(n, np.uint32)
        st = np.zeros(n, np.uint8); cnt = C.c_uint32()
        self._check(self._L.re_debug_get_sections(self._h, n, keys.ctypes.data, tight.ctypes.data, nl.ctypes.data, ns.ctypes.data, st.ctypes.data, C.byref(cnt)), "re_debug_get_sections")
        return dict(keys=keys, tight=tight, n_local=nl, n_static=ns, is_static_section=st)

    def shared_sections(self):
        """the shared world sections in canonical id order: [dict(keys, aabb, active, static)] (EntityIds)"""
        n = self.stats()["n_shared_sections"]; cap = max(n, 1); mcap = max(self.stats()["n_entities"], 1)
        keys = np.zeros((cap, 8), np.uint64); nk = np.zeros(cap, np.uint8); box = np.zeros((cap, 6), np.float32); na = np.zeros(cap, np.uint32); ns = np.zeros(cap, np.uint32)
        ids = np.zeros(mcap, np.uint32); offs = np.zeros(cap + 1, np.uint32); cnt = C.c_uint32()
        self._check(self._L.re_debug_get_shared_sections(self._h, cap, keys.ctypes.data, nk.ctypes.data, box.ctypes.data, na.ctypes.data, ns.ctypes.data, mcap, ids.ctypes.data,
                                                         offs.ctypes.data, C.byref(cnt)), "re_debug_get_shared_sections")
        out = []
        for i in range(min(cnt.value, cap)):
            o = int(offs[i])
            out.append(dict(keys=[int(k) for k in keys[i, :nk[i]]], aabb=tuple(float(v) for v in box[i]), active=ids[o:o + int(na[i])].copy(), static=ids[o + int(na[i]):o + int(na[i]) + int(ns[i])].copy()))
        return out

    def visible_sections(self):
        n = self.stats()["n_sections"]
        keys = np.zeros(n, np.uint64); mult = np.zeros(n, np.uint8); cnt = C.c_uint32()
        self._check(self._L.re_debug_get_visible_sections(self._h, n, keys.ctypes.data, mult.ctypes.data, C.byref(cnt)), "re_debug_get_visible_sections")
        return keys[:cnt.value], mult[:cnt.value]

    def timings_us(self):
        a, b, c = C.c_float(), C.c_float(), C.c_float()
        self._check(self._L.re_get_timings(self._h, C.byref(a), C.byref(b), C.byref(c)), "re_get_timings")
        return dict(cull=a.value, pack=b.value, tick=c.value)

    def timings_off(self):
        self._check(self._L.re_get_timings(self._h, None, None, None), "re_get_timings")

    def timing_begin(self, max_launches, every=1, kernel="scan"):
        """per-launch HIP-event timing of one kernel: "scan" (k_scan_cull), "tick" (k_tick) or "pack_large" (k_pack_large)"""
        kind = {"scan": 0, "tick": 1, "pack_large": 2}[kernel]
        self._check(self._L.re_timing_begin(self._h, max_launches, (every & 0xFFFF) | (kind << 16)), "re_timing_begin")

    def timing_collect(self, cap=65536):
        us = np.zeros(cap, np.float32); n = C.c_uint32()
        self._check(self._L.re_timing_collect(self._h, us.ctypes.data, cap, C.byref(n)), "re_timing_collect")
        return us[:min(n.value, cap)]

    def last_candidates(self):
        n = C.c_uint32(); self._check(self._L.re_get_last_candidates(self._h, C.byref(n)), "re_get_last_candidates")
        return n.value

    def stream(self):
        return self._L.re_get_stream(self._h)
