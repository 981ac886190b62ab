//! Safe wrapper around `librender_engine_hip.so` for render_engine: add as `src/gpu_visible_set/mod.rs` next to `ffi.rs`
//! and declare `mod gpu_visible_set;` in `src/lib.rs`.  The call sites it replaces are listed in `INTEGRATION.md` section 3.
//!
//! Source only: the build image of the library has no Rust toolchain, so this file has not been compiled there.  It uses
//! nothing beyond `std` and the crate's own types (`EntityId`, `ModelId`, `InstanceRange`, `Camera`, the movement components).
pub mod ffi;

use ffi::*;
use std::ffi::CStr;
use std::mem::MaybeUninit;

/// One resident copy of the world on one GPU.
pub struct GpuVisibleSet { ctx: *mut ReCtx, collision_pairs: Vec<ReCollision> }

// the context is driven from the render thread only (Pipeline::execute), like the flows it replaces
unsafe impl Send for GpuVisibleSet {}

#[derive(Debug)]
pub struct GpuError { pub code: i32, pub message: String }

/// Column view of the ECS the registration step fills once (threads/render_thread.rs:186-206), one row per entity.
#[derive(Default)]
pub struct EntityColumns {
    pub entity_id: Vec<u32>, pub model_index: Vec<u32>, pub render_system: Vec<u32>, pub sortable: Vec<u32>, pub flags: Vec<u32>,
    pub original_aabb: Vec<[f32; 6]>,              // x_min, x_max, y_min, y_max, z_min, z_max (StaticAABB field order)
    pub position: Vec<[f32; 3]>, pub rotation: Vec<[f32; 4]>, pub scale: Vec<[f32; 3]>,
    pub velocity: Vec<[f32; 3]>, pub acceleration: Vec<[f32; 3]>, pub rotation_velocity: Vec<[f32; 4]>, pub rotation_acceleration: Vec<[f32; 4]>,
}

pub struct Frame { pub visible_sections: u32, pub instances: u32, pub written: u32, pub groups: Vec<ReInstanceRange> }

impl GpuVisibleSet {
    /// `outline_length` / `atomic_length`: the arguments of `BoundingBoxTree::new` (render_thread.rs:127).
    pub fn new(device: i32, outline_length: u32, atomic_length: u32, max_instances: u32) -> Result<GpuVisibleSet, GpuError> {
        let cfg = ReConfig { device, outline_length, atomic_length, max_instances, flags: 0 };
        let mut ctx: *mut ReCtx = std::ptr::null_mut();
        let rc = unsafe { re_create(&cfg, &mut ctx) };
        if rc != RE_OK { return Err(GpuError { code: rc, message: last_error(std::ptr::null()) }); }
        Ok(GpuVisibleSet { ctx, collision_pairs: Vec::with_capacity(4096) })
    }

    fn check(&self, rc: i32) -> Result<(), GpuError> {
        if rc == RE_OK { Ok(()) } else { Err(GpuError { code: rc, message: last_error(self.ctx) }) }
    }

    /// Pipeline::register_model_instances for everything created so far; returns the number of entities rejected as out of bounds
    /// (the `eprintln!` of EntityTransformationBuilder::apply_choices, exports/entity_transformer.rs:71-74).
    pub fn upload(&mut self, c: &EntityColumns) -> Result<u32, GpuError> {
        let n = c.entity_id.len();
        assert!([c.model_index.len(), c.render_system.len(), c.sortable.len(), c.flags.len(), c.original_aabb.len(), c.position.len(), c.rotation.len(), c.scale.len(),
                 c.velocity.len(), c.acceleration.len(), c.rotation_velocity.len(), c.rotation_acceleration.len()].iter().all(|&l| l == n));
        let e = ReEntities {
            n: n as u32, entity_id: c.entity_id.as_ptr(), model_index: c.model_index.as_ptr(), render_system: c.render_system.as_ptr(), sortable: c.sortable.as_ptr(),
            flags: c.flags.as_ptr(), original_aabb: c.original_aabb.as_ptr() as *const f32, position: c.position.as_ptr() as *const f32,
            rotation: c.rotation.as_ptr() as *const f32, scale: c.scale.as_ptr() as *const f32, velocity: c.velocity.as_ptr() as *const f32,
            acceleration: c.acceleration.as_ptr() as *const f32, rotation_velocity: c.rotation_velocity.as_ptr() as *const f32,
            rotation_acceleration: c.rotation_acceleration.as_ptr() as *const f32,
        };
        let mut rejected = 0u32;
        self.check(unsafe { re_upload_entities(self.ctx, &e, &mut rejected) })?;
        Ok(rejected)
    }

    /// flows/pipeline.rs:216-240: both visibility queries and the render gather.  `projection_view` = projection * view, column-major
    /// (`nalgebra_glm::Mat4::as_slice`); `lod` = `create_level_of_views(far)` as (min, max) pairs (prelude/default_render_system.rs:240-256).
    pub fn cull_pack(&mut self, projection_view: &[f32; 16], position: [f32; 3], direction: [f32; 3], far_draw: f32, lod: &[(f32, f32)], emit_duplicates: bool) -> Result<Frame, GpuError> {
        let mut cam = ReCamera { projection_view: *projection_view, position, direction, far_draw, n_lod: lod.len().min(8) as u32, lod_min: [0.0; 8], lod_max: [0.0; 8] };
        for (i, (lo, hi)) in lod.iter().take(8).enumerate() { cam.lod_min[i] = *lo; cam.lod_max[i] = *hi; }
        let mut vis = MaybeUninit::<ReVisible>::zeroed();
        self.check(unsafe { re_cull_pack(self.ctx, &cam, if emit_duplicates { RE_CULL_EMIT_DUPLICATES } else { 0 }, vis.as_mut_ptr()) })?;
        let vis = unsafe { vis.assume_init() };
        let groups = if vis.n_groups == 0 { Vec::new() } else { unsafe { std::slice::from_raw_parts(vis.groups, vis.n_groups as usize) }.to_vec() };
        Ok(Frame { visible_sections: vis.n_visible_sections, instances: vis.n_instances, written: vis.n_written, groups })
    }

    /// flows/render_flow.rs:939-992: the instance bytes into the persistent-mapped GL buffer of render system 0
    /// (`MappedBuffer::write_data_serialized` semantics: truncates at the buffer's capacity).  Returns instances written.
    pub fn copy_instances(&mut self, ids: &mut [u32], matrices: *mut f32, capacity_instances: u32) -> Result<u32, GpuError> {
        let mut n = 0u32;
        let cap = capacity_instances.min(ids.len() as u32);
        self.check(unsafe { re_copy_visible(self.ctx, ids.as_mut_ptr(), matrices, cap, &mut n) })?;
        Ok(n)
    }

    /// flows/logic_flow.rs:243: `handle_collisions` up to the collision-logic callbacks; the slice holds the
    /// (this_entity, other_entity) arguments of every invocation, in no particular order.
    pub fn collide(&mut self) -> Result<&[ReCollision], GpuError> {
        loop {
            let mut n = 0u32;
            let cap = self.collision_pairs.capacity() as u32;
            let rc = unsafe { re_collide(self.ctx, 0, self.collision_pairs.as_mut_ptr(), cap, &mut n) };
            self.check(rc)?;
            if n <= cap { unsafe { self.collision_pairs.set_len(n as usize) }; return Ok(&self.collision_pairs); }
            self.collision_pairs = Vec::with_capacity(n as usize + n as usize / 4);      // grew past the buffer: once more with room
        }
    }

    /// flows/logic_flow.rs:230 + the kinematic part of :255 (update_positions, apply_change of the kinematic requests).
    pub fn tick(&mut self, delta_time: f32) -> Result<ReTickResult, GpuError> {
        let mut t = MaybeUninit::<ReTickResult>::zeroed();
        self.check(unsafe { re_tick(self.ctx, delta_time, 0, t.as_mut_ptr()) })?;
        Ok(unsafe { t.assume_init() })
    }

    /// helper_things/entity_change_helpers.rs:32-189 for the requests user logic returned this frame (list order kept).
    pub fn apply_changes(&mut self, changes: &[ReChange]) -> Result<ReTickResult, GpuError> {
        let mut t = MaybeUninit::<ReTickResult>::zeroed();
        self.check(unsafe { re_apply_changes(self.ctx, changes.as_ptr(), changes.len() as u32, 0, t.as_mut_ptr()) })?;
        Ok(unsafe { t.assume_init() })
    }

    /// entities that left the world without OutOfBoundsLogic since the last call (entity_change_helpers.rs:331-349)
    pub fn out_of_bounds(&mut self) -> Result<Vec<u32>, GpuError> {
        let mut ids = vec![0u32; 4096]; let mut n = 0u32;
        self.check(unsafe { re_get_out_of_bounds(self.ctx, ids.as_mut_ptr(), ids.len() as u32, &mut n) })?;
        ids.truncate(n.min(4096) as usize);
        Ok(ids)
    }

    /// ECS::get_copy::<Position / Rotation / ...> for user logic (objects/ecs.rs:653-664); `dst` sized for the component (re_hip.h RE_C_*)
    pub fn read_component(&mut self, entity_id: u32, component: u32, dst: &mut [f32]) -> Result<(), GpuError> {
        self.check(unsafe { re_read_component(self.ctx, entity_id, component as i32, dst.as_mut_ptr() as *mut std::ffi::c_void) })
    }
}

impl Drop for GpuVisibleSet {
    fn drop(&mut self) { if !self.ctx.is_null() { unsafe { re_destroy(self.ctx) }; self.ctx = std::ptr::null_mut(); } }
}

fn last_error(ctx: *const ReCtx) -> String {
    let p = unsafe { re_last_error(ctx) };
    if p.is_null() { String::new() } else { unsafe { CStr::from_ptr(p) }.to_string_lossy().into_owned() }
}

/// `EntityChangeRequest` of one kinematic component -> ReChange (exports/logic_components.rs; value = xyz / axis + angle)
pub fn modify(entity_id: u32, component: u32, value: [f32; 4]) -> ReChange { ReChange { kind: RE_CHANGE_MODIFY, entity_id, component, reserved: 0, value } }
pub fn delete(entity_id: u32) -> ReChange { ReChange { kind: RE_CHANGE_DELETE, entity_id, component: 0, reserved: 0, value: [0.0; 4] } }
pub fn make_static(entity_id: u32) -> ReChange { ReChange { kind: RE_CHANGE_MAKE_STATIC, entity_id, component: 0, reserved: 0, value: [0.0; 4] } }
pub fn wake_up(entity_id: u32) -> ReChange { ReChange { kind: RE_CHANGE_WAKE_UP, entity_id, component: 0, reserved: 0, value: [0.0; 4] } }
