//! Raw binding of `librender_engine_hip.so` (`include/re_hip.h`), 1:1 with the C ABI.  Add as `src/gpu_visible_set/ffi.rs`.
//! Not compiled in the build image of this repository (no Rust toolchain there): every symbol below is exported by the
//! library, which `tests/test_c_abi.py` checks against the header.
#![allow(non_camel_case_types, dead_code)]
use std::os::raw::{c_char, c_int, c_void};

#[repr(C)] pub struct ReCtx { _private: [u8; 0] }
#[repr(C)] pub struct ReConfig { pub device: i32, pub outline_length: u32, pub atomic_length: u32, pub max_instances: u32, pub flags: u32 }
#[repr(C)] pub struct ReEntities {
    pub n: u32,
    pub entity_id: *const u32, pub model_index: *const u32, pub render_system: *const u32, pub sortable: *const u32, pub flags: *const u32,
    pub original_aabb: *const f32, pub position: *const f32, pub rotation: *const f32, pub scale: *const f32,
    pub velocity: *const f32, pub acceleration: *const f32, pub rotation_velocity: *const f32, pub rotation_acceleration: *const f32,
}
#[repr(C)] pub struct ReCamera { pub projection_view: [f32; 16], pub position: [f32; 3], pub direction: [f32; 3], pub far_draw: f32,
                                 pub n_lod: u32, pub lod_min: [f32; 8], pub lod_max: [f32; 8] }
#[repr(C)] #[derive(Copy, Clone)] pub struct ReInstanceRange { pub model_index: u32, pub render_system: u32, pub sortable: u32, pub begin_instance: u32, pub count: u32 }
#[repr(C)] pub struct ReVisible { pub n_visible_sections: u32, pub n_visible_vec: u32, pub n_instances: u32, pub n_written: u32, pub n_groups: u32,
                                  pub groups: *const ReInstanceRange, pub d_entity_ids: *const u32, pub d_matrices: *const f32 }
#[repr(C)] pub struct ReTickResult { pub n_changed: u32, pub n_rebucket: u32, pub n_out_of_bounds: u32 }
#[repr(C)] pub struct ReChange { pub kind: u32, pub entity_id: u32, pub component: u32, pub reserved: u32, pub value: [f32; 4] }
#[repr(C)] #[derive(Copy, Clone)] pub struct ReCollision { pub this_entity: u32, pub other_entity: u32 }
#[repr(C)] pub struct ReGathered { pub n_ranks: u32, pub overflowed: u32, pub counts: *const u32, pub d_entity_ids: *const u32, pub ids_rank_stride: u32,
                                   pub d_matrices: *const f32, pub matrices_rank_stride: u32 }
#[repr(C)] #[derive(Copy, Clone, Default)] pub struct ReEntityState { pub entity_id: u32, pub model_index: u32, pub render_system: u32, pub sortable: u32, pub flags: u32,
    pub original_aabb: [f32; 6], pub position: [f32; 3], pub rotation: [f32; 4], pub scale: [f32; 3], pub velocity: [f32; 3], pub acceleration: [f32; 3],
    pub rotation_velocity: [f32; 4], pub rotation_acceleration: [f32; 4] }                                   // 140 bytes
#[repr(C)] #[derive(Copy, Clone, Default)] pub struct ReStats { pub n_entities: u32, pub n_dynamic: u32, pub n_sections: u32, pub n_shared_sections: u32, pub max_level: u32,
    pub device_bytes: u64, pub n_probe_frames: u32, pub n_table_rebuilds: u32, pub n_fused_frames: u32, pub reserved: u32, pub n_seal_waits: u32, pub n_sync_fallbacks: u32,
    pub n_section_slots: u32, pub n_device_rebuckets: u32, pub n_segment_redos: u32, pub n_host_rebuckets: u32 }
#[repr(C)] pub struct ReLighting { _private: [u8; 0] }
#[repr(C)] pub struct ReLightingConfig { pub device: i32, pub width: u32, pub height: u32, pub max_spot_lights: u32, pub max_point_lights: u32 }
#[repr(C)] pub struct ReLights { pub n_spot: u32, pub n_point: u32,
    pub spot_pos: *const f32, pub spot_diffuse: *const f32, pub spot_specular: *const f32, pub spot_ambient: *const f32, pub spot_linear: *const f32, pub spot_quadratic: *const f32, pub spot_radius: *const f32,
    pub point_pos: *const f32, pub point_dir: *const f32, pub point_diffuse: *const f32, pub point_specular: *const f32, pub point_ambient: *const f32, pub point_linear: *const f32,
    pub point_quadratic: *const f32, pub point_cutoff: *const f32, pub point_outer_cutoff: *const f32,
    pub camera_pos: [f32; 3], pub no_light_source_cutoff: f32, pub default_diffuse_factor: f32, pub any_light_source_visible: u32 }
#[repr(C)] pub struct ReHistory { _private: [u8; 0] }
#[repr(C)] pub struct ReTypeIds { pub position: u64, pub rotation: u64, pub scale: u64, pub velocity: u64, pub acceleration: u64, pub rotation_velocity: u64,
                                  pub rotation_acceleration: u64, pub has_moved: u64, pub has_rotated: u64 }
#[repr(C)] pub struct ReFrameChange { pub kind: u32, pub f: [f32; 6], pub i: [i32; 2], pub n_changes: u32, pub changes: *const ReChange }

extern "C" {
    pub fn re_create(cfg: *const ReConfig, out: *mut *mut ReCtx) -> c_int;
    pub fn re_destroy(ctx: *mut ReCtx);
    pub fn re_last_error(ctx: *const ReCtx) -> *const c_char;
    pub fn re_upload_entities(ctx: *mut ReCtx, ents: *const ReEntities, n_rejected: *mut u32) -> c_int;
    pub fn re_cull_pack(ctx: *mut ReCtx, cam: *const ReCamera, flags: u32, out: *mut ReVisible) -> c_int;
    pub fn re_tick(ctx: *mut ReCtx, delta_time: f32, flags: u32, out: *mut ReTickResult) -> c_int;
    pub fn re_apply_changes(ctx: *mut ReCtx, changes: *const ReChange, n: u32, flags: u32, out: *mut ReTickResult) -> c_int;
    pub fn re_collide(ctx: *mut ReCtx, flags: u32, pairs: *mut ReCollision, capacity: u32, n_total: *mut u32) -> c_int;
    pub fn re_wait(ctx: *mut ReCtx, vis: *mut ReVisible, tick: *mut ReTickResult) -> c_int;
    pub fn re_copy_visible(ctx: *mut ReCtx, ids: *mut u32, mats: *mut f32, capacity: u32, n_written: *mut u32) -> c_int;
    pub fn re_set_output_buffers(ctx: *mut ReCtx, d_ids: *mut u32, d_mats: *mut f32, capacity: u32) -> c_int;
    pub fn re_set_output_count(ctx: *mut ReCtx, d_count: *mut u32) -> c_int;   // 4 device words {written, total, frame, 0} written by the pack (an all-gather slab header of the host's own; re_comm_* does this itself)
    pub fn re_read_component(ctx: *mut ReCtx, entity_id: u32, component: c_int, dst: *mut c_void) -> c_int;
    pub fn re_get_out_of_bounds(ctx: *mut ReCtx, ids: *mut u32, capacity: u32, n: *mut u32) -> c_int;
    // round 2
    pub fn re_abi_version() -> u32;                                                                            // 3
    pub fn re_set_model_lod(ctx: *mut ReCtx, model_index: u32, render_system: u32, n_lod: u32, lod_min: *const f32, lod_max: *const f32) -> c_int;
    pub fn re_run_frames(ctx: *mut ReCtx, cam: *const ReCamera, delta_time: f32, cull_flags: u32, tick_flags: u32, n: u32,
                         wall_us: *mut f32, last_visible: *mut ReVisible, last_tick: *mut ReTickResult) -> c_int;
    pub fn re_ecs_bitset(ctx: *mut ReCtx, entity_id: u32, bits: *mut u32) -> c_int;
    pub fn re_ecs_query(ctx: *mut ReCtx, components: *const c_int, n_components: u32, ids: *mut u32, capacity: u32, n: *mut u32) -> c_int;
    pub fn re_visible_lights(ctx: *mut ReCtx, cam: *const ReCamera, light_type: u32, ids: *mut u32, capacity: u32, n: *mut u32) -> c_int;
    pub fn re_section_keys(cfg: *const ReConfig, ents: *const ReEntities, keys: *mut u64 /* [n * 8] */, n_keys: *mut u8 /* [n] */) -> c_int;   // host arithmetic: the sharding key of a multi-GPU loader
    pub fn re_comm_unique_id(id: *mut u8) -> c_int;                                                            // RE_COMM_ID_BYTES = 128
    pub fn re_comm_init(ctx: *mut ReCtx, id: *const u8, rank: c_int, n_ranks: c_int, slab_instances: u32) -> c_int;
    pub fn re_comm_adopt(ctx: *mut ReCtx, nccl_comm: *mut c_void, rank: c_int, n_ranks: c_int, slab_instances: u32) -> c_int;
    pub fn re_comm_destroy(ctx: *mut ReCtx) -> c_int;
    pub fn re_allgather_visible(ctx: *mut ReCtx, flags: u32, out: *mut ReGathered) -> c_int;
    pub fn re_gather_wait(ctx: *mut ReCtx, out: *mut ReGathered) -> c_int;
    // round 3 (ABI version 3)
    pub fn re_add_entities(ctx: *mut ReCtx, ents: *const ReEntities, n_rejected: *mut u32) -> c_int;          // Pipeline::register_model_instances at any time: APPENDS
    pub fn re_apply_changes_ex(ctx: *mut ReCtx, changes: *const ReChange, n: u32, added: *const ReEntities, flags: u32, out: *mut ReTickResult) -> c_int;   // + RE_CHANGE_ADD_ENTITY
    pub fn re_set_shard_range(ctx: *mut ReCtx, key_lo: u64, key_hi: u64) -> c_int;
    pub fn re_list_migrants(ctx: *mut ReCtx, entity_ids: *mut u32, capacity: u32, n: *mut u32) -> c_int;
    pub fn re_export_entities(ctx: *mut ReCtx, entity_ids: *const u32, n: u32, out: *mut ReEntityState) -> c_int;
    pub fn re_get_stats(ctx: *mut ReCtx, out: *mut ReStats) -> c_int;
    pub fn re_debug_copy_to_host(ctx: *mut ReCtx, d_src: *const c_void, dst: *mut c_void, bytes: u64) -> c_int;
    // introspection / profiling
    pub fn re_debug_get_sections(ctx: *mut ReCtx, capacity: u32, keys: *mut u64, tight_aabb6: *mut f32, n_local: *mut u32, n_static: *mut u32, is_static_section: *mut u8, n: *mut u32) -> c_int;
    pub fn re_debug_get_shared_sections(ctx: *mut ReCtx, capacity: u32, keys: *mut u64, n_keys: *mut u8, aabb6: *mut f32, n_active: *mut u32, n_static: *mut u32,
                                        member_capacity: u32, member_ids: *mut u32, member_offsets: *mut u32, n: *mut u32) -> c_int;
    pub fn re_debug_get_visible_sections(ctx: *mut ReCtx, capacity: u32, keys: *mut u64, multiplicity: *mut u8, n: *mut u32) -> c_int;
    pub fn re_get_timings(ctx: *mut ReCtx, cull_us: *mut f32, pack_us: *mut f32, tick_us: *mut f32) -> c_int;
    pub fn re_timing_begin(ctx: *mut ReCtx, max_launches: u32, every: u32) -> c_int;
    pub fn re_timing_collect(ctx: *mut ReCtx, microseconds: *mut f32, capacity: u32, n: *mut u32) -> c_int;
    pub fn re_get_last_candidates(ctx: *mut ReCtx, n_candidates: *mut u32) -> c_int;
    pub fn re_get_stream(ctx: *mut ReCtx) -> *mut c_void;
    // deferred lighting (BASELINE configs[4]; second pass of RenderSystem::draw)
    pub fn re_lighting_create(cfg: *const ReLightingConfig, out: *mut *mut ReLighting) -> c_int;
    pub fn re_lighting_destroy(l: *mut ReLighting);
    pub fn re_lighting_last_error(l: *const ReLighting) -> *const c_char;
    pub fn re_lighting_upload_gbuffer(l: *mut ReLighting, g_position: *const f32, g_normal: *const f32, g_albedo_spec: *const u8) -> c_int;
    pub fn re_lighting_set_lights(l: *mut ReLighting, lights: *const ReLights) -> c_int;
    pub fn re_lighting_run(l: *mut ReLighting, kernel_microseconds: *mut f32) -> c_int;
    pub fn re_lighting_read(l: *mut ReLighting, out_rgba: *mut f32) -> c_int;
    pub fn re_lighting_read_pixels(l: *mut ReLighting, pixel_index: *const u32, n: u32, out_rgba: *mut f32) -> c_int;
    pub fn re_history_create(ids: *const ReTypeIds, flags: u32, out: *mut *mut ReHistory) -> c_int;
    pub fn re_history_destroy(h: *mut ReHistory);
    pub fn re_history_last_error(h: *const ReHistory) -> *const c_char;
    pub fn re_history_set_state(h: *mut ReHistory, ecs_blob: *const c_void, ecs_bytes: u64, tree_blob: *const c_void, tree_bytes: u64) -> c_int;
    pub fn re_history_get_state(h: *mut ReHistory, ecs_blob: *mut *const c_void, ecs_bytes: *mut u64, tree_blob: *mut *const c_void, tree_bytes: *mut u64) -> c_int;
    pub fn re_history_record(h: *mut ReHistory, fc: *const ReFrameChange) -> c_int;
    pub fn re_history_count(h: *mut ReHistory, n: *mut u32) -> c_int;
    pub fn re_history_get(h: *mut ReHistory, index: u32, out: *mut ReFrameChange) -> c_int;
    pub fn re_history_encode(h: *mut ReHistory, index: u32, dst: *mut u8, capacity: u64, n_bytes: *mut u64) -> c_int;
    pub fn re_history_write(h: *mut ReHistory, history_path: *const c_char, lookup_path: *const c_char) -> c_int;
    pub fn re_history_load(ids: *const ReTypeIds, flags: u32, history_path: *const c_char, lookup_path: *const c_char, out: *mut *mut ReHistory) -> c_int;
}

pub const RE_OK: c_int = 0;
pub const RE_F_STATIC: u32 = 0x001; pub const RE_F_HAS_VEL: u32 = 0x002; pub const RE_F_HAS_ACC: u32 = 0x004; pub const RE_F_HAS_ROT: u32 = 0x008;
pub const RE_F_HAS_ROTVEL: u32 = 0x010; pub const RE_F_HAS_ROTACC: u32 = 0x020; pub const RE_F_HAS_SCALE: u32 = 0x040; pub const RE_F_ALWAYS_EXEC: u32 = 0x080;
pub const RE_F_OOB_LOGIC: u32 = 0x100; pub const RE_F_USER: u32 = 0x800; pub const RE_F_CAN_COLLIDE: u32 = 0x1000;
pub const RE_CULL_EMIT_DUPLICATES: u32 = 0x1; pub const RE_CULL_ASYNC: u32 = 0x2; pub const RE_CULL_DEFER_PACK: u32 = 0x10; pub const RE_CULL_TWO_LANES: u32 = 0x20; pub const RE_CULL_ONE_LAUNCH: u32 = 0x40;
pub const RE_CHANGE_MODIFY: u32 = 0; pub const RE_CHANGE_DELETE: u32 = 1; pub const RE_CHANGE_MAKE_STATIC: u32 = 2; pub const RE_CHANGE_WAKE_UP: u32 = 3;
pub const RE_C_POSITION: u32 = 0; pub const RE_C_ROTATION: u32 = 1; pub const RE_C_SCALE: u32 = 2; pub const RE_C_VELOCITY: u32 = 3; pub const RE_C_ACCELERATION: u32 = 4;
pub const RE_C_ROTATION_VEL: u32 = 5; pub const RE_C_ROTATION_ACC: u32 = 6;
pub const RE_CHANGE_REMOVE_COMPONENT: u32 = 4; pub const RE_CHANGE_ADD_ENTITY: u32 = 5; pub const RE_CHANGE_ADD_SORTABLE: u32 = 6; pub const RE_CHANGE_REMOVE_SORTABLE: u32 = 7;
pub const RE_C_TRANSFORMATION: u32 = 7; pub const RE_C_STATIC_AABB: u32 = 8; pub const RE_C_ORIGINAL_AABB: u32 = 9; pub const RE_C_FLAGS: u32 = 10;
pub const RE_GATHER_ASYNC: u32 = 0x1; pub const RE_COMM_ID_BYTES: usize = 128;
pub const RE_FC_CAMERA_VIEW_CHANGE: u32 = 0; pub const RE_FC_CAMERA_STATIONARY: u32 = 1; pub const RE_FC_DELTA_TIME: u32 = 2; pub const RE_FC_DRAW_DISTANCES_CHANGE: u32 = 3;
pub const RE_FC_WINDOW_DIMENSIONS_CHANGE: u32 = 4; pub const RE_FC_ENTITY_CHANGE: u32 = 5; pub const RE_FC_END_FRAME_CHANGE: u32 = 6;
pub const RE_F_LIGHT_DIRECTIONAL: u32 = 0x2000; pub const RE_F_LIGHT_POINT: u32 = 0x4000; pub const RE_F_LIGHT_SPOT: u32 = 0x8000;
pub const RE_F_PHANTOM: u32 = 0x10000;
