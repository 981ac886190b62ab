"""History / replay of a play session: thin mirror of re_history_* (csrc/re_history.cpp) plus the replay loop of
Pipeline::debug_execute (flows/pipeline.rs:279-421) over this package's Pipeline.

The wire format is the reference's (threads/history_thread.rs:150-205, helper_things/game_loader.rs:32-71): bincode 1.3 records in
gameplay_history.txt, their byte lengths in gameplay_byte_lookup.txt."""
import ctypes as C

import numpy as np

from . import _capi
from .pipeline import CHANGE_DT, Camera, RenderEngineError

FC = _capi.FC
# TypeIdentifier values are TypeId bits of one reference build; any nine distinct non-zero numbers make a self-consistent file
DEFAULT_TYPE_IDS = dict(position=0x1001, rotation=0x1002, scale=0x1003, velocity=0x1004, acceleration=0x1005, rotation_velocity=0x1006,
                        rotation_acceleration=0x1007, has_moved=0x1008, has_rotated=0x1009)


def _ids(type_ids):
    t = _capi.TypeIds()
    for k, v in (type_ids or DEFAULT_TYPE_IDS).items():
        setattr(t, k, int(v))
    return t


class History:
    """the recorded FrameChange stream of a session (StoredHistoryState.game_history_changes_to_apply, flattened as write_to_disk does)"""

    def __init__(self, type_ids=None, vec3_as_array=False, _handle=None):
        self._L = _capi.load()
        self._flags = _capi.HISTORY_VEC3_AS_ARRAY if vec3_as_array else 0
        self._type_ids = _ids(type_ids)
        if _handle is None:
            h = C.c_void_p()
            rc = self._L.re_history_create(C.byref(self._type_ids), self._flags, C.byref(h))
            if rc != _capi.RE_OK:
                raise RenderEngineError(f"re_history_create failed ({rc})")
            _handle = h
        self._h = _handle

    def close(self):
        if self._h:
            self._L.re_history_destroy(self._h); self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc, what):
        if rc != _capi.RE_OK:
            raise RenderEngineError(f"{what} failed ({rc}): {self._L.re_history_last_error(self._h).decode()}")

    # -- recording (what HistoryThread stores per frame, threads/history_thread.rs) --------------------
    def _record(self, kind, f=(), i=(), changes=None):
        fc = _capi.FrameChange(); fc.kind = kind
        for k, v in enumerate(f):
            fc.f[k] = float(v)
        for k, v in enumerate(i):
            fc.i[k] = int(v)
        keep = None
        if changes is not None:
            keep = np.ascontiguousarray(changes, dtype=CHANGE_DT)
            fc.n_changes = len(keep); fc.changes = keep.ctypes.data
        self._check(self._L.re_history_record(self._h, C.byref(fc)), "re_history_record")

    def camera_view_change(self, position, direction): self._record(FC["CAMERA_VIEW_CHANGE"], list(position) + list(direction))
    def camera_stationary(self): self._record(FC["CAMERA_STATIONARY"])
    def delta_time(self, dt): self._record(FC["DELTA_TIME"], [dt])
    def draw_distances_change(self, near, far, fov): self._record(FC["DRAW_DISTANCES_CHANGE"], [near, far, fov])
    def window_dimensions_change(self, width, height): self._record(FC["WINDOW_DIMENSIONS_CHANGE"], i=[width, height])
    def entity_change(self, changes): self._record(FC["ENTITY_CHANGE"], changes=changes)
    def end_frame(self): self._record(FC["END_FRAME_CHANGE"])

    def set_state(self, ecs_blob=b"", tree_blob=b""):
        """the two leading blobs of the history file (bincode of the reference's ECS / BoundingBoxTree): opaque here"""
        self._check(self._L.re_history_set_state(self._h, ecs_blob, len(ecs_blob), tree_blob, len(tree_blob)), "re_history_set_state")

    def state(self):
        pe, pt, ne, nt = C.c_void_p(), C.c_void_p(), C.c_uint64(), C.c_uint64()
        self._check(self._L.re_history_get_state(self._h, C.byref(pe), C.byref(ne), C.byref(pt), C.byref(nt)), "re_history_get_state")
        return (C.string_at(pe.value, ne.value) if ne.value else b""), (C.string_at(pt.value, nt.value) if nt.value else b"")

    # -- reading ----------------------------------------------------------------------------------
    def __len__(self):
        n = C.c_uint32(); self._check(self._L.re_history_count(self._h, C.byref(n)), "re_history_count"); return n.value

    def get(self, index):
        """(kind, floats[6], ints[2], changes as CHANGE_DT array)"""
        fc = _capi.FrameChange()
        self._check(self._L.re_history_get(self._h, index, C.byref(fc)), "re_history_get")
        ch = np.zeros(fc.n_changes, CHANGE_DT)
        if fc.n_changes:
            C.memmove(ch.ctypes.data, fc.changes, fc.n_changes * CHANGE_DT.itemsize)
        return fc.kind, np.array(fc.f[:], np.float32), (fc.i[0], fc.i[1]), ch

    def encode(self, index):
        """the bincode bytes of one FrameChange record"""
        n = C.c_uint64(); self._check(self._L.re_history_encode(self._h, index, None, 0, C.byref(n)), "re_history_encode")
        buf = (C.c_uint8 * max(n.value, 1))()
        self._check(self._L.re_history_encode(self._h, index, buf, n.value, C.byref(n)), "re_history_encode")
        return bytes(buf[:n.value])

    def write(self, history_path, lookup_path):
        self._check(self._L.re_history_write(self._h, str(history_path).encode(), str(lookup_path).encode()), "re_history_write")

    @classmethod
    def load(cls, history_path, lookup_path, type_ids=None, vec3_as_array=False):
        L = _capi.load(); h = C.c_void_p(); t = _ids(type_ids)
        rc = L.re_history_load(C.byref(t), _capi.HISTORY_VEC3_AS_ARRAY if vec3_as_array else 0, str(history_path).encode(), str(lookup_path).encode(), C.byref(h))
        if rc != _capi.RE_OK:
            raise RenderEngineError(f"re_history_load failed ({rc}): {L.re_history_last_error(None).decode()}")
        return cls(type_ids, vec3_as_array, _handle=h)

    def frame_indexes(self):
        """frame_indexes of a debug session: the index of every EndFrameChange (flows/pipeline.rs:85-95); frame k replays the records
        [frame_indexes[k - 1], frame_indexes[k]), so an EndFrameChange is consumed (as a no-op) by the frame after it (:317-330)"""
        return [k for k in range(len(self)) if self.get(k)[0] == FC["END_FRAME_CHANGE"]]


class ReplayCamera:
    """the camera state debug_execute mutates (exports/camera_object.rs:70-113)"""

    def __init__(self, position, direction, far, fov_degrees=45.0, window=(1280, 720), near=0.1):
        self.position, self.direction = np.asarray(position, np.float32), np.asarray(direction, np.float32)
        self.near, self.far, self.fov, self.window = float(near), float(far), float(fov_degrees), (int(window[0]), int(window[1]))

    def camera(self):
        return Camera(self.position, self.direction, self.far, fov_degrees=self.fov, window_dimensions=self.window, near_draw_distance=self.near)


def replay(world, history, cam, on_frame=None):
    """Pipeline::debug_execute with play == true, custom_movement == false, no user logic, over `world` -- anything with the frame calls of
    render_engine_amd.Pipeline (cull_and_pack / apply_changes / tick), so the oracle's World can be driven by the same loop in the tests.
    Per recorded frame: the visibility query with the camera as the frame found it (:283-294), then the frame's records in order --
    EntityChange -> apply_change (:330-347), CameraViewChange -> camera, then the logic phase with the last recorded DeltaTime (:348-368),
    CameraStationary -> the logic phase (:369-385), DrawDistancesChange / WindowDimensionsChange -> camera (:390-397).
    The reference renders with the visible sections of the frame's FIRST camera and the frame's LAST camera matrices; this boundary takes
    one camera per re_cull_pack, so the replay renders each frame with the camera the frame started with (the next frame sees the moved one).
    Returns the list of per-frame results (or of on_frame's return values)."""
    out = []
    begin = 0
    for end in history.frame_indexes():
        res = world.cull_and_pack(cam.camera())
        dt = 0.0
        for k in range(begin, end):
            kind, f, i, ch = history.get(k)
            if kind == FC["ENTITY_CHANGE"]:
                if len(ch):
                    world.apply_changes(ch)
            elif kind == FC["CAMERA_VIEW_CHANGE"]:
                cam.position, cam.direction = f[0:3].copy(), f[3:6].copy()
                world.tick(float(dt))
            elif kind == FC["CAMERA_STATIONARY"]:
                world.tick(float(dt))
            elif kind == FC["DELTA_TIME"]:
                dt = f[0]
            elif kind == FC["DRAW_DISTANCES_CHANGE"]:
                cam.near, cam.far, cam.fov = float(f[0]), float(f[1]), float(f[2])
            elif kind == FC["WINDOW_DIMENSIONS_CHANGE"]:
                cam.window = (int(i[0]), int(i[1]))
        out.append(on_frame(res) if on_frame else res)
        begin = end
    return out
