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
    pub fn re_set_output_count(ctx: *mut ReCtx, d_count: *mut u32) -> c_int;   // device word receiving the instance count (all-gather slab header)
    pub fn re_read_component(ctx: *mut ReCtx, entity_id: u32, component: c_int, dst: *mut c_void) -> c_int;
    pub fn re_get_out_of_bounds(ctx: *mut ReCtx, ids: *mut u32, capacity: u32, n: *mut u32) -> c_int;
}

pub const RE_OK: c_int = 0;
pub const RE_F_STATIC: u32 = 0x001; pub const RE_F_HAS_VEL: u32 = 0x002; pub const RE_F_HAS_ACC: u32 = 0x004; pub const RE_F_HAS_ROT: u32 = 0x008;
pub const RE_F_HAS_ROTVEL: u32 = 0x010; pub const RE_F_HAS_ROTACC: u32 = 0x020; pub const RE_F_HAS_SCALE: u32 = 0x040; pub const RE_F_ALWAYS_EXEC: u32 = 0x080;
pub const RE_F_OOB_LOGIC: u32 = 0x100; pub const RE_F_USER: u32 = 0x800; pub const RE_F_CAN_COLLIDE: u32 = 0x1000;
pub const RE_CULL_EMIT_DUPLICATES: u32 = 0x1; pub const RE_CULL_ASYNC: u32 = 0x2; pub const RE_CULL_DEFER_PACK: u32 = 0x10; pub const RE_CULL_TWO_LANES: u32 = 0x20;
pub const RE_CHANGE_MODIFY: u32 = 0; pub const RE_CHANGE_DELETE: u32 = 1; pub const RE_CHANGE_MAKE_STATIC: u32 = 2; pub const RE_CHANGE_WAKE_UP: u32 = 3;
pub const RE_C_POSITION: u32 = 0; pub const RE_C_ROTATION: u32 = 1; pub const RE_C_SCALE: u32 = 2; pub const RE_C_VELOCITY: u32 = 3; pub const RE_C_ACCELERATION: u32 = 4;
pub const RE_C_ROTATION_VEL: u32 = 5; pub const RE_C_ROTATION_ACC: u32 = 6;
