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

pub struct Frame { pub visible_sections: u32, pub visible_sections_vec: u32, pub instances: u32, pub written: u32, pub groups: Vec<ReInstanceRange> }

/// What replaces `CullResult` (flows/visible_world_flow.rs:17-36) at the call sites of this path: the render and logic flows of the GPU path never look
/// at the section ids themselves -- the gather (render_flow.rs:401-410), the tick gate (logic_flow.rs:216-223) and the collision phase read the visible
/// set on the device --, so the adapter carries its SIZES (what the reference prints / asserts on) and the handle.
pub struct GpuCullResult { pub visible_sections_map_len: u32, pub visible_sections_vec_len: u32 }

/// ECS::create_entity's id allocator (objects/ecs.rs:384-402): the last freed id, else the next one.  The library takes entity ids from its host
/// (RE_CHANGE_ADD_ENTITY / re_add_entities); the shim keeps the counter and the free list exactly as the reference's ECS does.
#[derive(Default)]
pub struct EntityIds { next: u32, free: Vec<u32> }
impl EntityIds {
    pub fn with_existing(n_entities: u32) -> EntityIds { EntityIds { next: n_entities, free: Vec::new() } }
    pub fn create_entity(&mut self) -> u32 { match self.free.pop() { Some(i) => i, None => { let i = self.next; self.next += 1; i } } }
    pub fn remove_entity(&mut self, id: u32) { self.free.push(id); }
}

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
        Ok(Frame { visible_sections: vis.n_visible_sections, visible_sections_vec: vis.n_visible_vec, instances: vis.n_instances, written: vis.n_written, groups })
    }

    /// `CullResult` adapter of flows/pipeline.rs:222-229 (the union of the two visibility queries, duplicates kept in the vec)
    pub fn cull_result(frame: &Frame) -> GpuCullResult { GpuCullResult { visible_sections_map_len: frame.visible_sections, visible_sections_vec_len: frame.visible_sections_vec } }

    /// flows/render_flow.rs:401-410 + upload_instance_data_to_render_system (:939-992) for one render system: the frame's (ModelId, sortable) groups become
    /// `ModelRenderingInformation::instance_location` entries (`InstanceRange { begin_instance, count }`), every other entry's count is zeroed first (:952-962),
    /// the TransformationMatrix bytes go into the persistent-mapped instance buffer (truncated at its capacity like MappedBuffer::write_data_serialized,
    /// render_components/mapped_buffer.rs:166-189), and the buffer is flushed for the bytes written.
    ///
    /// `set_range(model_index_with_lod, render_system_index, sortable_index, begin_instance, count)` is the crate-side closure that does
    ///     `render_system.model_rendering_information.entry(ModelId { model_index, render_system_index }).or_insert_with(ModelRenderingInformation::new)
    ///          .instance_location.insert(sortable_index, InstanceRange { begin_instance, count })`
    /// and `zero_counts()` the loop of :952-962; they are closures because `ModelRenderingInformation::new` and `BufferWriteInfo`'s fields are private to the crate's modules.
    pub fn upload_instance_data_to_render_system(&mut self, frame: &Frame, render_system_index: u32, mapped_matrix_buffer: *mut f32, buffer_bytes: isize,
                                                 mut zero_counts: impl FnMut(), mut set_range: impl FnMut(u32, u32, usize, u32, u32)) -> Result<isize, GpuError> {
        zero_counts();
        for g in frame.groups.iter().filter(|g| g.render_system == render_system_index) { set_range(g.model_index, g.render_system, g.sortable as usize, g.begin_instance, g.count); }
        let capacity = (buffer_bytes.max(0) as usize / 64) as u32;             // 64 bytes per instance: one column-major Matrix4<f32>
        let mut ids = vec![0u32; frame.written.min(capacity) as usize];
        let n = self.copy_instances(&mut ids, mapped_matrix_buffer, capacity)?;
        Ok(n as isize * 64)                                                    // bytes to hand to RenderSystem::flush_per_instance_buffers
    }

    /// Pipeline::register_model_instances after the first frame (flows/pipeline.rs:186-208): the new instances are APPENDED (`upload` replaces the world)
    pub fn add_entities(&mut self, c: &EntityColumns) -> Result<u32, GpuError> {
        let e = Self::columns(c);
        let mut rejected = 0u32;
        self.check(unsafe { re_add_entities(self.ctx, &e, &mut rejected) })?;
        Ok(rejected)
    }

    /// apply_change with AddEntity arms (helper_things/entity_change_helpers.rs:48-107): `changes` in list order, `added` the entities the
    /// RE_CHANGE_ADD_ENTITY changes refer to (`reserved` = index; `entity_id` from `EntityIds::create_entity`)
    pub fn apply_changes_with_added(&mut self, changes: &[ReChange], added: &EntityColumns) -> Result<ReTickResult, GpuError> {
        let e = Self::columns(added);
        let mut t = MaybeUninit::<ReTickResult>::zeroed();
        self.check(unsafe { re_apply_changes_ex(self.ctx, changes.as_ptr(), changes.len() as u32, &e, 0, t.as_mut_ptr()) })?;
        Ok(unsafe { t.assume_init() })
    }

    /// level_views.custom of one model (register_model_with_render_system(.., custom_level_of_view, ..), flows/render_flow.rs:1069-1076)
    pub fn set_model_lod(&mut self, model_index: u32, render_system: u32, bands: &[(f32, f32)]) -> Result<(), GpuError> {
        let lo: Vec<f32> = bands.iter().map(|b| b.0).collect(); let hi: Vec<f32> = bands.iter().map(|b| b.1).collect();
        self.check(unsafe { re_set_model_lod(self.ctx, model_index, render_system, bands.len().min(8) as u32, lo.as_ptr(), hi.as_ptr()) })
    }

    /// ECS::check_component_written / the entity's bitset (objects/ecs.rs:61-72, 348-367) in the reference's registration order
    pub fn ecs_bitset(&mut self, entity_id: u32) -> Result<u32, GpuError> { let mut b = 0u32; self.check(unsafe { re_ecs_bitset(self.ctx, entity_id, &mut b) })?; Ok(b) }

    /// ECS::get_indexes_for_components (objects/ecs.rs:238-285): ascending ids of the entities that carry all of `components` (RE_C_*)
    pub fn get_indexes_for_components(&mut self, components: &[i32]) -> Result<Vec<u32>, GpuError> {
        let mut n = 0u32;
        self.check(unsafe { re_ecs_query(self.ctx, components.as_ptr(), components.len() as u32, std::ptr::null_mut(), 0, &mut n) })?;
        let mut ids = vec![0u32; n as usize];
        self.check(unsafe { re_ecs_query(self.ctx, components.as_ptr(), components.len() as u32, ids.as_mut_ptr(), n, &mut n) })?;
        ids.truncate(n as usize);
        Ok(ids)
    }

    /// find_nearby_lights (flows/shadow_flow.rs:455-513) for one FindLightType (RE_F_LIGHT_*): ascending entity ids
    pub fn visible_lights(&mut self, projection_view: &[f32; 16], position: [f32; 3], direction: [f32; 3], far_draw: f32, light_type: u32) -> Result<Vec<u32>, GpuError> {
        let cam = ReCamera { projection_view: *projection_view, position, direction, far_draw, n_lod: 0, lod_min: [0.0; 8], lod_max: [0.0; 8] };
        let mut n = 0u32; let mut ids = vec![0u32; 4096];
        loop {
            self.check(unsafe { re_visible_lights(self.ctx, &cam, light_type, ids.as_mut_ptr(), ids.len() as u32, &mut n) })?;
            if n as usize <= ids.len() { ids.truncate(n as usize); return Ok(ids); }
            ids = vec![0u32; n as usize];
        }
    }

    // ---- several GPUs (one GpuVisibleSet per GPU, one process per GPU): the frame's exchange steps -------------------------------------------------
    /// rank 0 creates the id and hands it to the other ranks over the host's own channel
    pub fn comm_unique_id() -> Result<[u8; RE_COMM_ID_BYTES], GpuError> {
        let mut id = [0u8; RE_COMM_ID_BYTES];
        let rc = unsafe { re_comm_unique_id(id.as_mut_ptr()) };
        if rc != RE_OK { return Err(GpuError { code: rc, message: last_error(std::ptr::null()) }); }
        Ok(id)
    }
    pub fn comm_init(&mut self, id: &[u8; RE_COMM_ID_BYTES], rank: i32, n_ranks: i32, slab_instances: u32) -> Result<(), GpuError> {
        self.check(unsafe { re_comm_init(self.ctx, id.as_ptr(), rank, n_ranks, slab_instances) })
    }
    /// between cull_pack and tick: every GPU ends with every GPU's packed visible instances, in rank order; returns the per-rank counts
    pub fn allgather_visible(&mut self) -> Result<Vec<u32>, GpuError> {
        let mut g = MaybeUninit::<ReGathered>::zeroed();
        self.check(unsafe { re_allgather_visible(self.ctx, 0, g.as_mut_ptr()) })?;
        let g = unsafe { g.assume_init() };
        Ok(unsafe { std::slice::from_raw_parts(g.counts, g.n_ranks as usize) }.to_vec())
    }
    /// the world sections this GPU owns (the smallest key of an entity's sections decides): switches the migrant bookkeeping on
    pub fn set_shard_range(&mut self, key_lo: u64, key_hi: u64) -> Result<(), GpuError> { self.check(unsafe { re_set_shard_range(self.ctx, key_lo, key_hi) }) }
    /// after the tick: the entities whose section left this GPU's key range, with their complete state, removed here; the host routes each record to the
    /// GPU that owns its new section (re_section_keys is the same host arithmetic on every rank) and calls `add_entities` there
    pub fn take_migrants(&mut self) -> Result<Vec<ReEntityState>, GpuError> {
        let mut n = 0u32; let mut ids = vec![0u32; 4096];
        loop {
            self.check(unsafe { re_list_migrants(self.ctx, ids.as_mut_ptr(), ids.len() as u32, &mut n) })?;
            if n as usize <= ids.len() { ids.truncate(n as usize); break; }
            ids = vec![0u32; n as usize];
        }
        let mut out = vec![ReEntityState::default(); ids.len()];
        if ids.is_empty() { return Ok(out); }
        self.check(unsafe { re_export_entities(self.ctx, ids.as_ptr(), ids.len() as u32, out.as_mut_ptr()) })?;
        let del: Vec<ReChange> = ids.iter().map(|&i| delete(i)).collect();
        self.apply_changes(&del)?;
        Ok(out)
    }

    fn columns(c: &EntityColumns) -> ReEntities {
        let n = c.entity_id.len();
        assert!([c.model_index.len(), c.render_system.len(), c.sortable.len(), c.flags.len(), c.original_aabb.len(), c.position.len(), c.rotation.len(), c.scale.len(),
                 c.velocity.len(), c.acceleration.len(), c.rotation_velocity.len(), c.rotation_acceleration.len()].iter().all(|&l| l == n));
        ReEntities {
            n: n as u32, entity_id: c.entity_id.as_ptr(), model_index: c.model_index.as_ptr(), render_system: c.render_system.as_ptr(), sortable: c.sortable.as_ptr(),
            flags: c.flags.as_ptr(), original_aabb: c.original_aabb.as_ptr() as *const f32, position: c.position.as_ptr() as *const f32,
            rotation: c.rotation.as_ptr() as *const f32, scale: c.scale.as_ptr() as *const f32, velocity: c.velocity.as_ptr() as *const f32,
            acceleration: c.acceleration.as_ptr() as *const f32, rotation_velocity: c.rotation_velocity.as_ptr() as *const f32,
            rotation_acceleration: c.rotation_acceleration.as_ptr() as *const f32,
        }
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

/// `UniqueWorldSectionId` (world/bounding_box_tree_v2.rs:21-26) <-> the library's packed key `level:16 | x:16 | z:16 | y:16`
pub fn section_key_parts(key: u64) -> (u16, u16, u16, u16) { ((key >> 48) as u16, (key >> 32) as u16, (key >> 16) as u16, key as u16) }      // (level, x, z, y)

impl GpuVisibleSet {
    /// The CONTENT of `CullResult` (flows/visible_world_flow.rs:17-36) for callers that do want the ids -- debug overlays, `debug_execute`, tests:
    /// `.0` = `visible_sections_map` (every visible section once, ascending key), `.1` = `visible_sections_vec` (a section found by both the logic and the render
    /// culler appears twice, as in the reference's concatenation, pipeline.rs:222-229).  Build the crate's types with `section_key_parts`:
    /// `UniqueWorldSectionId::new(level, SectionOffsets{ x, z, y })`.  A device read-back: not for the per-frame path (use `cull_result` there).
    pub fn cull_result_sections(&mut self, frame: &Frame) -> Result<(Vec<u64>, Vec<u64>), GpuError> {
        let cap = frame.visible_sections.max(1);
        let mut keys = vec![0u64; cap as usize]; let mut mult = vec![0u8; cap as usize]; let mut n = 0u32;
        self.check(unsafe { re_debug_get_visible_sections(self.ctx, cap, keys.as_mut_ptr(), mult.as_mut_ptr(), &mut n) })?;
        keys.truncate(n.min(cap) as usize); mult.truncate(n.min(cap) as usize);
        let mut vec_ids = Vec::with_capacity(frame.visible_sections_vec as usize);
        for (k, m) in keys.iter().zip(mult.iter()) { for _ in 0..*m { vec_ids.push(*k); } }
        Ok((keys, vec_ids))
    }
}

/// History / replay files of a session (threads/history_thread.rs:150-205, helper_things/game_loader.rs:32-71): the `FrameChange` stream in the reference's
/// bincode layout; the ECS / tree blobs in front of it pass through as opaque bytes.  Host code only -- no GPU context involved.
pub struct History { h: *mut ReHistory }
impl History {
    /// `ids`: the `TypeIdentifier` bits of the running build, e.g. `TypeIdentifier::from(TypeId::of::<Position>())` (objects/ecs.rs:97-110)
    pub fn new(ids: &ReTypeIds) -> Result<History, GpuError> {
        let mut h: *mut ReHistory = std::ptr::null_mut();
        let rc = unsafe { re_history_create(ids, 0, &mut h) };
        if rc != RE_OK { return Err(GpuError { code: rc, message: String::from("re_history_create failed") }); }
        Ok(History { h })
    }
    /// GameLoadResult::load: reads the pair `gameplay_history.txt` / `gameplay_byte_lookup.txt`
    pub fn load(ids: &ReTypeIds, history_path: &str, lookup_path: &str) -> Result<History, GpuError> {
        let (hp, lp) = (std::ffi::CString::new(history_path).unwrap(), std::ffi::CString::new(lookup_path).unwrap());
        let mut h: *mut ReHistory = std::ptr::null_mut();
        let rc = unsafe { re_history_load(ids, 0, hp.as_ptr(), lp.as_ptr(), &mut h) };
        if rc != RE_OK { return Err(GpuError { code: rc, message: format!("re_history_load({}) failed", history_path) }); }
        Ok(History { h })
    }
    fn check(&self, rc: i32) -> Result<(), GpuError> {
        if rc == RE_OK { return Ok(()); }
        let p = unsafe { re_history_last_error(self.h) };
        Err(GpuError { code: rc, message: if p.is_null() { String::new() } else { unsafe { CStr::from_ptr(p) }.to_string_lossy().into_owned() } })
    }
    /// the serialized ECS and BoundingBoxTree the reference writes in front of the records (opaque here)
    pub fn set_state(&mut self, ecs_blob: &[u8], tree_blob: &[u8]) -> Result<(), GpuError> {
        self.check(unsafe { re_history_set_state(self.h, ecs_blob.as_ptr() as *const std::ffi::c_void, ecs_blob.len() as u64, tree_blob.as_ptr() as *const std::ffi::c_void, tree_blob.len() as u64) })
    }
    /// HistoryThread's per-frame record (one `FrameChange`: a camera / window change or the frame's `EntityChange` list as `ReChange`s)
    pub fn record(&mut self, fc: &ReFrameChange) -> Result<(), GpuError> { self.check(unsafe { re_history_record(self.h, fc) }) }
    pub fn len(&mut self) -> Result<u32, GpuError> { let mut n = 0u32; self.check(unsafe { re_history_count(self.h, &mut n) })?; Ok(n) }
    /// record `index`; its `changes` pointer stays valid until the next call on this object (feed it to `GpuVisibleSet::apply_changes` in a replay loop, pipeline.rs:279-421)
    pub fn get(&mut self, index: u32) -> Result<ReFrameChange, GpuError> {
        let mut fc = MaybeUninit::<ReFrameChange>::uninit();
        self.check(unsafe { re_history_get(self.h, index, fc.as_mut_ptr()) })?;
        Ok(unsafe { fc.assume_init() })
    }
    /// HistoryThread::write_to_disk: the history file and its byte-offset lookup file
    pub fn write(&mut self, history_path: &str, lookup_path: &str) -> Result<(), GpuError> {
        let (hp, lp) = (std::ffi::CString::new(history_path).unwrap(), std::ffi::CString::new(lookup_path).unwrap());
        self.check(unsafe { re_history_write(self.h, hp.as_ptr(), lp.as_ptr()) })
    }
}
impl Drop for History { fn drop(&mut self) { if !self.h.is_null() { unsafe { re_history_destroy(self.h) }; self.h = std::ptr::null_mut(); } } }

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
pub fn remove_component(entity_id: u32, component: u32) -> ReChange { ReChange { kind: RE_CHANGE_REMOVE_COMPONENT, entity_id, component, reserved: 0, value: [0.0; 4] } }
/// EntityChangeInformation::AddEntity: `index` into the `added` columns handed to apply_changes_with_added; `entity_id` from EntityIds::create_entity
pub fn add_entity(entity_id: u32, index: u32) -> ReChange { ReChange { kind: RE_CHANGE_ADD_ENTITY, entity_id, component: 0, reserved: index, value: [0.0; 4] } }
/// EntityChangeInformation::AddSortableComponent / RemoveSortableComponent: `sortable_index` = position of the sortable component in registration order
pub fn add_sortable(entity_id: u32, sortable_index: u32) -> ReChange { ReChange { kind: RE_CHANGE_ADD_SORTABLE, entity_id, component: sortable_index, reserved: 0, value: [0.0; 4] } }
pub fn remove_sortable(entity_id: u32) -> ReChange { ReChange { kind: RE_CHANGE_REMOVE_SORTABLE, entity_id, component: 0, reserved: 0, value: [0.0; 4] } }
/// ReEntityState records (migrants) as the columns add_entities takes
pub fn columns_of(states: &[ReEntityState]) -> EntityColumns {
    let mut c = EntityColumns::default();
    for s in states {
        c.entity_id.push(s.entity_id); c.model_index.push(s.model_index); c.render_system.push(s.render_system); c.sortable.push(s.sortable); c.flags.push(s.flags);
        c.original_aabb.push(s.original_aabb); c.position.push(s.position); c.rotation.push(s.rotation); c.scale.push(s.scale); c.velocity.push(s.velocity);
        c.acceleration.push(s.acceleration); c.rotation_velocity.push(s.rotation_velocity); c.rotation_acceleration.push(s.rotation_acceleration);
    }
    c
}
