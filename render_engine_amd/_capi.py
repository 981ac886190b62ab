"""ctypes binding of librender_engine_hip.so (include/re_hip.h).  No fallback: if the library is
missing or cannot be loaded the import of the product path fails loudly."""
import ctypes as C
import os

from . import build as _build

RE_OK = 0
F_STATIC, F_HAS_VEL, F_HAS_ACC, F_HAS_ROT = 0x001, 0x002, 0x004, 0x008
F_HAS_ROTVEL, F_HAS_ROTACC, F_HAS_SCALE, F_ALWAYS_EXEC = 0x010, 0x020, 0x040, 0x080
F_OOB_LOGIC, F_HAS_MOVED, F_HAS_ROTATED, F_USER = 0x100, 0x200, 0x400, 0x800
F_CAN_COLLIDE = 0x1000
F_PHANTOM = 0x10000            # halo replica of an entity another shard owns: in the tree, never drawn or ticked
F_LIGHT_DIRECTIONAL, F_LIGHT_POINT, F_LIGHT_SPOT = 0x2000, 0x4000, 0x8000      # FindLightType of the entity (light sets of its world section)
CULL_EMIT_DUPLICATES, CULL_ASYNC, CULL_FORCE_LARGE_PACK, CULL_FORCE_STREAM, CULL_DEFER_PACK, CULL_TWO_LANES, CULL_ONE_LAUNCH = 0x1, 0x2, 0x4, 0x8, 0x10, 0x20, 0x40
TICK_ALL_DYNAMIC, TICK_ASYNC = 0x1, 0x2
(C_POSITION, C_ROTATION, C_SCALE, C_VELOCITY, C_ACCELERATION, C_ROTATION_VEL, C_ROTATION_ACC,
 C_TRANSFORMATION, C_STATIC_AABB, C_ORIGINAL_AABB, C_FLAGS) = range(11)
COMPONENT_FLOATS = {C_POSITION: 3, C_ROTATION: 4, C_SCALE: 3, C_VELOCITY: 3, C_ACCELERATION: 3, C_ROTATION_VEL: 4,
                    C_ROTATION_ACC: 4, C_TRANSFORMATION: 16, C_STATIC_AABB: 6, C_ORIGINAL_AABB: 6}

_u32p, _fp = C.POINTER(C.c_uint32), C.POINTER(C.c_float)


class Config(C.Structure):
    _fields_ = [("device", C.c_int32), ("outline_length", C.c_uint32), ("atomic_length", C.c_uint32),
                ("max_instances", C.c_uint32), ("flags", C.c_uint32)]


class Entities(C.Structure):
    _fields_ = [("n", C.c_uint32), ("entity_id", _u32p), ("model_index", _u32p), ("render_system", _u32p), ("sortable", _u32p),
                ("flags", _u32p), ("original_aabb", _fp), ("position", _fp), ("rotation", _fp), ("scale", _fp), ("velocity", _fp),
                ("acceleration", _fp), ("rotation_velocity", _fp), ("rotation_acceleration", _fp)]


class CameraC(C.Structure):
    _fields_ = [("projection_view", C.c_float * 16), ("position", C.c_float * 3), ("direction", C.c_float * 3),
                ("far_draw", C.c_float), ("n_lod", C.c_uint32), ("lod_min", C.c_float * 8), ("lod_max", C.c_float * 8)]


class InstanceRange(C.Structure):
    _fields_ = [("model_index", C.c_uint32), ("render_system", C.c_uint32), ("sortable", C.c_uint32),
                ("begin_instance", C.c_uint32), ("count", C.c_uint32)]


class Visible(C.Structure):
    _fields_ = [("n_visible_sections", C.c_uint32), ("n_visible_vec", C.c_uint32), ("n_instances", C.c_uint32),
                ("n_written", C.c_uint32), ("n_groups", C.c_uint32), ("groups", C.POINTER(InstanceRange)),
                ("d_entity_ids", C.c_void_p), ("d_matrices", C.c_void_p)]


CFG_FULL_REBUILD = 0x1
CFG_TIGHT_SLACK = 0x2
CFG_PROBE = 0x4
CFG_PROBE_ALWAYS = 0x8
CHANGE_MODIFY, CHANGE_DELETE, CHANGE_MAKE_STATIC, CHANGE_WAKE_UP, CHANGE_REMOVE_COMPONENT, CHANGE_ADD_ENTITY, CHANGE_ADD_SORTABLE, CHANGE_REMOVE_SORTABLE = 0, 1, 2, 3, 4, 5, 6, 7
# bit positions of re_ecs_bitset == registration order of the reference (ECS::new + LogicFlow::new)
ECS_BIT = dict(TYPE_IDENTIFIER=0, CAN_CAUSE_COLLISIONS=2, HAS_MOVED=3, POSITION=4, VELOCITY=5, ACCELERATION=6, HAS_ROTATED=7, ROTATION=8, VELOCITY_ROTATION=9,
               ACCELERATION_ROTATION=10, SCALE=11, TRANSFORMATION=12, MODEL_ID=13, STATIC_AABB=15, ORIGINAL_AABB=16, ALWAYS_EXECUTE_LOGIC=20)


class TypeIds(C.Structure):
    _fields_ = [(k, C.c_uint64) for k in ("position", "rotation", "scale", "velocity", "acceleration", "rotation_velocity", "rotation_acceleration", "has_moved", "has_rotated")]


class FrameChange(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("f", C.c_float * 6), ("i", C.c_int32 * 2), ("n_changes", C.c_uint32), ("changes", C.c_void_p)]


FC = dict(CAMERA_VIEW_CHANGE=0, CAMERA_STATIONARY=1, DELTA_TIME=2, DRAW_DISTANCES_CHANGE=3, WINDOW_DIMENSIONS_CHANGE=4, ENTITY_CHANGE=5, END_FRAME_CHANGE=6)
HISTORY_VEC3_AS_ARRAY = 1


class TickResult(C.Structure):
    _fields_ = [("n_changed", C.c_uint32), ("n_rebucket", C.c_uint32), ("n_out_of_bounds", C.c_uint32)]


class Stats(C.Structure):
    _fields_ = [("n_entities", C.c_uint32), ("n_dynamic", C.c_uint32), ("n_sections", C.c_uint32),
                ("n_shared_sections", C.c_uint32), ("max_level", C.c_uint32), ("device_bytes", C.c_uint64),
                ("n_probe_frames", C.c_uint32), ("n_table_rebuilds", C.c_uint32), ("n_fused_frames", C.c_uint32), ("reserved", C.c_uint32),
                ("n_seal_waits", C.c_uint32), ("n_sync_fallbacks", C.c_uint32), ("n_section_slots", C.c_uint32), ("n_device_rebuckets", C.c_uint32),
                ("n_segment_redos", C.c_uint32), ("n_host_rebuckets", C.c_uint32)]


class Gathered(C.Structure):
    _fields_ = [("n_ranks", C.c_uint32), ("overflowed", C.c_uint32), ("counts", _u32p), ("d_entity_ids", C.c_void_p), ("ids_rank_stride", C.c_uint32),
                ("d_matrices", C.c_void_p), ("matrices_rank_stride", C.c_uint32)]


COMM_ID_BYTES, GATHER_ASYNC = 128, 0x1


class LightingConfig(C.Structure):
    _fields_ = [("device", C.c_int32), ("width", C.c_uint32), ("height", C.c_uint32), ("max_spot_lights", C.c_uint32), ("max_point_lights", C.c_uint32)]


class Lights(C.Structure):
    _fields_ = [("n_spot", C.c_uint32), ("n_point", C.c_uint32)] + \
               [(n, _fp) for n in ("spot_pos", "spot_diffuse", "spot_specular", "spot_ambient", "spot_linear", "spot_quadratic", "spot_radius",
                                   "point_pos", "point_dir", "point_diffuse", "point_specular", "point_ambient", "point_linear", "point_quadratic",
                                   "point_cutoff", "point_outer_cutoff")] + \
               [("camera_pos", C.c_float * 3), ("no_light_source_cutoff", C.c_float), ("default_diffuse_factor", C.c_float), ("any_light_source_visible", C.c_uint32)]


# every symbol include/re_hip.h declares
EXPORTS = ["re_create", "re_destroy", "re_last_error", "re_abi_version", "re_upload_entities", "re_set_model_lod", "re_cull_pack", "re_tick",
           "re_apply_changes", "re_apply_changes_ex", "re_add_entities", "re_set_shard_range", "re_list_migrants", "re_export_entities", "re_collide", "re_wait", "re_run_frames", "re_comm_unique_id", "re_comm_init", "re_comm_adopt", "re_comm_destroy", "re_allgather_visible", "re_gather_wait", "re_copy_visible", "re_set_output_buffers", "re_set_output_count", "re_read_component", "re_ecs_bitset", "re_ecs_query", "re_visible_lights", "re_section_keys", "re_get_out_of_bounds", "re_get_stats",
           "re_debug_get_sections", "re_debug_get_shared_sections", "re_debug_get_visible_sections", "re_debug_copy_to_host", "re_get_timings", "re_get_stream",
           "re_timing_begin", "re_timing_collect", "re_get_last_candidates",
           "re_lighting_create", "re_lighting_destroy", "re_lighting_last_error", "re_lighting_upload_gbuffer", "re_lighting_set_lights",
           "re_lighting_run", "re_lighting_read", "re_lighting_read_pixels",
           "re_history_create", "re_history_destroy", "re_history_last_error", "re_history_set_state", "re_history_get_state", "re_history_record",
           "re_history_count", "re_history_get", "re_history_encode", "re_history_write", "re_history_load"]

_lib = None


def device_to_host(handle, ptr, nbytes):
    """bytes of device memory through the library itself (re_debug_copy_to_host): the process may hold a second HIP runtime (a PyTorch wheel brings its
    own), and a pointer of this library means nothing to that one"""
    import numpy as np
    out = np.zeros(max(nbytes, 1), np.uint8)
    rc = load().re_debug_copy_to_host(handle, ptr, out.ctypes.data, nbytes)
    if rc != 0:
        raise RuntimeError(f"re_debug_copy_to_host failed ({rc})")
    return out[:nbytes]


def library_path():
    return os.environ.get("RE_HIP_LIBRARY", _build.LIB_PATH)     # override only for A/B experiments with alternative builds


def load():
    """Loads the HIP library; raises (never falls back) when it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    path = library_path()
    if not os.path.exists(path):
        raise RuntimeError(f"{path} is not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                           "(render_engine_amd has no CPU fallback)")
    L = C.CDLL(path)
    vp = C.c_void_p
    L.re_create.restype = C.c_int; L.re_create.argtypes = [C.POINTER(Config), C.POINTER(vp)]
    L.re_destroy.restype = None; L.re_destroy.argtypes = [vp]
    L.re_last_error.restype = C.c_char_p; L.re_last_error.argtypes = [vp]
    L.re_abi_version.restype = C.c_uint32; L.re_abi_version.argtypes = []
    L.re_upload_entities.restype = C.c_int; L.re_upload_entities.argtypes = [vp, C.POINTER(Entities), _u32p]
    L.re_set_model_lod.restype = C.c_int; L.re_set_model_lod.argtypes = [vp, C.c_uint32, C.c_uint32, C.c_uint32, _fp, _fp]
    L.re_cull_pack.restype = C.c_int; L.re_cull_pack.argtypes = [vp, C.POINTER(CameraC), C.c_uint32, C.POINTER(Visible)]
    L.re_tick.restype = C.c_int; L.re_tick.argtypes = [vp, C.c_float, C.c_uint32, C.POINTER(TickResult)]
    L.re_wait.restype = C.c_int; L.re_wait.argtypes = [vp, C.POINTER(Visible), C.POINTER(TickResult)]
    L.re_run_frames.restype = C.c_int; L.re_run_frames.argtypes = [vp, C.POINTER(CameraC), C.c_float, C.c_uint32, C.c_uint32, C.c_uint32, vp, C.POINTER(Visible), C.POINTER(TickResult)]
    L.re_comm_unique_id.restype = C.c_int; L.re_comm_unique_id.argtypes = [vp]
    L.re_comm_init.restype = C.c_int; L.re_comm_init.argtypes = [vp, vp, C.c_int, C.c_int, C.c_uint32]
    L.re_comm_adopt.restype = C.c_int; L.re_comm_adopt.argtypes = [vp, vp, C.c_int, C.c_int, C.c_uint32]
    L.re_comm_destroy.restype = C.c_int; L.re_comm_destroy.argtypes = [vp]
    L.re_allgather_visible.restype = C.c_int; L.re_allgather_visible.argtypes = [vp, C.c_uint32, C.POINTER(Gathered)]
    L.re_gather_wait.restype = C.c_int; L.re_gather_wait.argtypes = [vp, C.POINTER(Gathered)]
    L.re_collide.restype = C.c_int; L.re_collide.argtypes = [vp, C.c_uint32, vp, C.c_uint32, C.POINTER(C.c_uint32)]
    L.re_apply_changes.restype = C.c_int; L.re_apply_changes.argtypes = [vp, vp, C.c_uint32, C.c_uint32, C.POINTER(TickResult)]
    L.re_apply_changes_ex.restype = C.c_int; L.re_apply_changes_ex.argtypes = [vp, vp, C.c_uint32, C.POINTER(Entities), C.c_uint32, C.POINTER(TickResult)]
    L.re_add_entities.restype = C.c_int; L.re_add_entities.argtypes = [vp, C.POINTER(Entities), _u32p]
    L.re_set_shard_range.restype = C.c_int; L.re_set_shard_range.argtypes = [vp, C.c_uint64, C.c_uint64]
    L.re_list_migrants.restype = C.c_int; L.re_list_migrants.argtypes = [vp, vp, C.c_uint32, _u32p]
    L.re_export_entities.restype = C.c_int; L.re_export_entities.argtypes = [vp, vp, C.c_uint32, vp]
    L.re_copy_visible.restype = C.c_int; L.re_copy_visible.argtypes = [vp, vp, vp, C.c_uint32, _u32p]
    L.re_set_output_buffers.restype = C.c_int; L.re_set_output_buffers.argtypes = [vp, vp, vp, C.c_uint32]
    L.re_read_component.restype = C.c_int; L.re_read_component.argtypes = [vp, C.c_uint32, C.c_int, vp]
    L.re_ecs_bitset.restype = C.c_int; L.re_ecs_bitset.argtypes = [vp, C.c_uint32, _u32p]
    L.re_section_keys.restype = C.c_int; L.re_section_keys.argtypes = [C.POINTER(Config), C.POINTER(Entities), vp, vp]
    L.re_visible_lights.restype = C.c_int; L.re_visible_lights.argtypes = [vp, C.POINTER(CameraC), C.c_uint32, vp, C.c_uint32, _u32p]
    L.re_ecs_query.restype = C.c_int; L.re_ecs_query.argtypes = [vp, C.POINTER(C.c_int), C.c_uint32, vp, C.c_uint32, _u32p]
    L.re_get_out_of_bounds.restype = C.c_int; L.re_get_out_of_bounds.argtypes = [vp, vp, C.c_uint32, _u32p]
    L.re_get_stats.restype = C.c_int; L.re_get_stats.argtypes = [vp, C.POINTER(Stats)]
    L.re_debug_get_sections.restype = C.c_int; L.re_debug_get_sections.argtypes = [vp, C.c_uint32, vp, vp, vp, vp, vp, _u32p]
    L.re_debug_get_shared_sections.restype = C.c_int; L.re_debug_get_shared_sections.argtypes = [vp, C.c_uint32, vp, vp, vp, vp, vp, C.c_uint32, vp, vp, _u32p]
    L.re_debug_get_visible_sections.restype = C.c_int; L.re_debug_get_visible_sections.argtypes = [vp, C.c_uint32, vp, vp, _u32p]
    L.re_get_timings.restype = C.c_int; L.re_get_timings.argtypes = [vp, _fp, _fp, _fp]
    L.re_debug_copy_to_host.restype = C.c_int; L.re_debug_copy_to_host.argtypes = [vp, vp, vp, C.c_uint64]
    L.re_get_stream.restype = vp; L.re_get_stream.argtypes = [vp]
    L.re_set_output_count.restype = C.c_int; L.re_set_output_count.argtypes = [vp, vp]
    L.re_timing_begin.restype = C.c_int; L.re_timing_begin.argtypes = [vp, C.c_uint32, C.c_uint32]
    L.re_timing_collect.restype = C.c_int; L.re_timing_collect.argtypes = [vp, vp, C.c_uint32, _u32p]
    L.re_get_last_candidates.restype = C.c_int; L.re_get_last_candidates.argtypes = [vp, _u32p]
    L.re_lighting_create.restype = C.c_int; L.re_lighting_create.argtypes = [C.POINTER(LightingConfig), C.POINTER(vp)]
    L.re_lighting_destroy.restype = None; L.re_lighting_destroy.argtypes = [vp]
    L.re_lighting_last_error.restype = C.c_char_p; L.re_lighting_last_error.argtypes = [vp]
    L.re_lighting_upload_gbuffer.restype = C.c_int; L.re_lighting_upload_gbuffer.argtypes = [vp, vp, vp, vp]
    L.re_lighting_set_lights.restype = C.c_int; L.re_lighting_set_lights.argtypes = [vp, C.POINTER(Lights)]
    L.re_lighting_run.restype = C.c_int; L.re_lighting_run.argtypes = [vp, _fp]
    L.re_lighting_read.restype = C.c_int; L.re_lighting_read.argtypes = [vp, vp]
    L.re_lighting_read_pixels.restype = C.c_int; L.re_lighting_read_pixels.argtypes = [vp, vp, C.c_uint32, vp]
    L.re_history_create.restype = C.c_int; L.re_history_create.argtypes = [C.POINTER(TypeIds), C.c_uint32, C.POINTER(vp)]
    L.re_history_destroy.restype = None; L.re_history_destroy.argtypes = [vp]
    L.re_history_last_error.restype = C.c_char_p; L.re_history_last_error.argtypes = [vp]
    L.re_history_set_state.restype = C.c_int; L.re_history_set_state.argtypes = [vp, vp, C.c_uint64, vp, C.c_uint64]
    L.re_history_get_state.restype = C.c_int; L.re_history_get_state.argtypes = [vp, C.POINTER(vp), C.POINTER(C.c_uint64), C.POINTER(vp), C.POINTER(C.c_uint64)]
    L.re_history_record.restype = C.c_int; L.re_history_record.argtypes = [vp, C.POINTER(FrameChange)]
    L.re_history_count.restype = C.c_int; L.re_history_count.argtypes = [vp, _u32p]
    L.re_history_get.restype = C.c_int; L.re_history_get.argtypes = [vp, C.c_uint32, C.POINTER(FrameChange)]
    L.re_history_encode.restype = C.c_int; L.re_history_encode.argtypes = [vp, C.c_uint32, vp, C.c_uint64, C.POINTER(C.c_uint64)]
    L.re_history_write.restype = C.c_int; L.re_history_write.argtypes = [vp, C.c_char_p, C.c_char_p]
    L.re_history_load.restype = C.c_int; L.re_history_load.argtypes = [C.POINTER(TypeIds), C.c_uint32, C.c_char_p, C.c_char_p, C.POINTER(vp)]
    _lib = L
    return L
