/*
 * re_hip.h -- C ABI of librender_engine_hip.so: the MI355X-native per-frame visible-set
 * pipeline of render_engine (spatial-hash visibility query -> frustum/logic cull -> instance
 * model-matrix pack, and the ECS kinematic / TRS->mat4 tick).
 *
 * Plain pointers and sizes only; no C++/torch types.  A Rust `extern "C"` block binds these 1:1
 * (INTEGRATION.md shows the shim).  Citations are reference file:line under /root/reference/src.
 *
 * The reference has no FFI/plugin layer (one Rust crate, generics + fn pointers), so a per-AABB
 * `TraversalDecider::aabb_in_view` callback (culling/trait.rs:4-7) is the wrong granularity for
 * a GPU.  The library replaces three *call sites* of Pipeline::execute wholesale:
 *
 *   re_cull_pack  <->  flows/pipeline.rs:216-229   (RenderFrustumCuller/LogicFrustumCuller +
 *                      VisibleWorldFlow::find_visible_world_ids_{entire_world,frustum_aabb} -> CullResult)
 *                 +    flows/render_flow.rs:401-410 (extract_static_data, sort_world_section_active_entities,
 *                      append_written_information, upload_instance_data_to_render_system -> InstanceRange table
 *                      + the bytes MappedBuffer::write_data_serialized would receive, render_components/mapped_buffer.rs:166-189)
 *   re_tick       <->  flows/logic_flow.rs:230 (update_positions/apply_kinematics :308-448)
 *                 +    flows/logic_flow.rs:255 -> helper_things/entity_change_helpers.rs:32-189,217-262
 *                      (apply_change / update_aabb_after_kinematic_change)
 *   re_upload_entities <-> flows/pipeline.rs:186-208 register_model_instances with an AddInstanceFunction
 *                      that runs EntityTransformationBuilder::apply_choices (exports/entity_transformer.rs:55-75)
 *                      + BoundingBoxTree::add_entity/end_of_changes (world/bounding_box_tree_v2.rs:563-762,1055-1130)
 *
 * Threading: one ctx = one caller thread at a time (Pipeline::execute runs on the render thread,
 * threads/render_thread.rs:475-481); different ctxs (one per GPU) may be driven concurrently.
 * Errors: no exceptions/aborts cross the ABI.  Every call returns RE_OK (0) or a negative code;
 * re_last_error(ctx) holds the message (the reference panics/unwraps -> lib.rs:45-61 panic hook).
 */
#ifndef RE_HIP_H
#define RE_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RE_OK            0
#define RE_E_ARG        -1   /* bad argument (null pointer, dt == 0 with rotating entities, ...) */
#define RE_E_HIP        -2   /* HIP runtime error; message carries hipGetErrorString */
#define RE_E_CAPACITY   -3   /* a configured capacity was exceeded */
#define RE_E_OOB        -4   /* entity AABB out of world bounds (add_entity -> Err(()), bounding_box_tree_v2.rs:569-572) */
#define RE_E_STATE      -5   /* call sequence error (e.g. tick before any cull) */
#define RE_E_UNSUPPORTED -6

/* component-presence / behaviour bits of an entity (re_entities.flags).  They mirror which
 * components EntityTransformationBuilder writes (exports/entity_transformer.rs:99-142). */
#define RE_F_STATIC      0x001u  /* is_entity_static: goes to static_entities of its world section */
#define RE_F_HAS_VEL     0x002u  /* Velocity */
#define RE_F_HAS_ACC     0x004u  /* Acceleration */
#define RE_F_HAS_ROT     0x008u  /* Rotation */
#define RE_F_HAS_ROTVEL  0x010u  /* VelocityRotation */
#define RE_F_HAS_ROTACC  0x020u  /* AccelerationRotation */
#define RE_F_HAS_SCALE   0x040u  /* Scale */
#define RE_F_ALWAYS_EXEC 0x080u  /* AlwaysExecuteLogic (logic_flow.rs:803-836) */
#define RE_F_OOB_LOGIC   0x100u  /* entity type has OutOfBoundsLogic => add_if_out_bounds (entity_change_helpers.rs:264-274) */
#define RE_F_HAS_MOVED   0x200u  /* HasMoved marker, maintained by re_tick */
#define RE_F_HAS_ROTATED 0x400u  /* HasRotated marker, maintained by re_tick */
#define RE_F_USER        0x800u  /* the user entity (flows/pipeline.rs:125-151): TransformationMatrix stays identity, StaticAABB = OriginalAABB
                                   * translated by Position; added to the tree as a non-static entity */

#define RE_F_LIGHT_DIRECTIONAL 0x2000u /* EntityTransformationBuilder::new(.., light_type = Some(FindLightType::Directional), ..): the entity joins the light set of its world
                                       * section (world/bounding_box_tree_v2.rs:157-228, add_entity :601-627, 690-730); see re_visible_lights */
#define RE_F_LIGHT_POINT  0x4000u
#define RE_F_LIGHT_SPOT   0x8000u
#define RE_F_PHANTOM     0x10000u /* halo replica of an entity another GPU's shard owns (re_section_keys): it takes part in the tree -- section membership,
                                   * shared-section links, static-section flags, tight AABBs, the "other" side of collision pairs -- and is never drawn, ticked or listed as a light here */
#define RE_F_CAN_COLLIDE 0x1000u /* CanCauseCollisions (EntityTransformationBuilder.can_cause_collision, exports/entity_transformer.rs:66-69) */

typedef struct re_ctx re_ctx;

typedef struct {
    int32_t  device;            /* HIP device ordinal */
    uint32_t outline_length;    /* BoundingBoxTree::new(outline, atomic): 16384 in render_thread.rs:127 */
    uint32_t atomic_length;     /* world_section_length: 64 (load_models.rs:52) */
    uint32_t max_instances;     /* capacity of the packed instance buffer, in instances (0 = number of entities) */
    uint32_t flags;             /* RE_CFG_* */
} re_config;

#define RE_CFG_DEFAULT 0u
#define RE_CFG_FULL_REBUILD 0x1u  /* testing: after section changes rebuild the whole section table instead of patching it in place */
#define RE_CFG_PROBE        0x4u  /* opt-in: visibility query by hash probes of the candidate cells (O(candidates), like the reference's contains_key probes)
                                   * instead of the key stream whenever the candidate boxes hold < 1/4 of the table's sections; needs 32 B per section more */
#define RE_CFG_PROBE_ALWAYS 0x8u  /* testing, with RE_CFG_PROBE: probe whenever the candidate cells can be enumerated, however small the table */
#define RE_CFG_TIGHT_SLACK  0x2u  /* testing: almost no spare slots / row-pool slack, so patches and full rebuilds alternate */

/* Entities, struct-of-arrays, host pointers; copied during the call.  Optional arrays may be
 * NULL when no entity carries the corresponding flag. */
typedef struct {
    uint32_t n;
    const uint32_t *entity_id;       /* EntityId(u32), objects/entity_id.rs:6 */
    const uint32_t *model_index;     /* ModelId.model_index, models/model_definitions.rs:10-14 */
    const uint32_t *render_system;   /* ModelId.render_system_index (NULL = 0) */
    const uint32_t *sortable;        /* sortable bucket 0..3, objects/sorted_entities.rs (NULL = 0) */
    const uint32_t *flags;           /* RE_F_* */
    const float *original_aabb;      /* n*6: xmin,xmax,ymin,ymax,zmin,zmax (OriginalAABB, model space) */
    const float *position;           /* n*3 */
    const float *rotation;           /* n*4 axis xyz + angle (radians); read when RE_F_HAS_ROT */
    const float *scale;              /* n*3; read when RE_F_HAS_SCALE */
    const float *velocity;           /* n*3; RE_F_HAS_VEL */
    const float *acceleration;       /* n*3; RE_F_HAS_ACC */
    const float *rotation_velocity;  /* n*4 axis + rate; RE_F_HAS_ROTVEL */
    const float *rotation_acceleration; /* n*4; RE_F_HAS_ROTACC */
} re_entities;

/* Camera inputs Pipeline::execute reads (flows/pipeline.rs:216-229, render_flow.rs:389-399). */
typedef struct {
    float projection_view[16];  /* camera.get_projection_matrix() * get_view_matrix(), column-major */
    float position[3];
    float direction[3];
    float far_draw;             /* get_far_draw_distance() */
    uint32_t n_lod;             /* default LevelOfView bands (<= 8), prelude/default_render_system.rs:240-256 */
    float lod_min[8];
    float lod_max[8];
} re_camera;

/* re_cull_pack flags */
#define RE_CULL_EMIT_DUPLICATES 0x1u  /* reproduce the reference's double emission for world sections present twice in
                                         visible_sections_vec (CullResult::extend, visible_world_flow.rs:31-35, pipeline.rs:228);
                                         default 0 = each visible instance once (the visible-ID *set*) */
#define RE_CULL_ASYNC           0x2u  /* enqueue only; results are valid after re_wait() */
#define RE_CULL_DEFER_PACK      0x10u /* with RE_CULL_ASYNC, in a world without dynamic entities: leave the instance pack of this frame to the launch of the
                                       * next re_cull_pack (one launch per frame; the pack's dependent round trips overlap the next key stream) or to the
                                       * next call that needs the result (re_wait, re_copy_visible, ...).  The packed instances are complete only then. */
#define RE_CULL_TWO_LANES       0x20u /* with RE_CULL_ASYNC | RE_CULL_DEFER_PACK: alternate such frames between two sets of per-frame resources on two HIP
                                       * streams, so that the launches of consecutive frames overlap on the GPU (the pack of frame f then rides in launch
                                       * f + 2).  Needs a second copy of the per-frame buffers (~70 B per entity).  re_get_stream returns the stream of the
                                       * frame issued last. */
#define RE_CULL_FORCE_STREAM    0x8u  /* with RE_CFG_PROBE: take the key stream for this frame anyway */
#define RE_CULL_ONE_LAUNCH      0x40u /* opt-in, synchronous frames with a small visible set and at most 256 group slots: ONE launch between the call and its answer -- the
                                      * scan's last workgroup to finish publishes the InstanceRange table and the counts itself (k_scan_cull_sync), the pack launch behind it only
                                      * moves the instances.  About 1 us less per frame than the two dependent launches; the scan launch then contains the publication chain
                                      * (DESIGN.md section 4), which is why it is not the default */
#define RE_CULL_FORCE_LARGE_PACK 0x4u /* always use the multi-kernel pack (count/scan/scatter) instead of k_pack_small */

/* One (ModelId, sortable) group of the packed buffer == ModelRenderingInformation.instance_location
 * entry (render_flow.rs:964-983). */
typedef struct {
    uint32_t model_index;       /* LOD-adjusted: model_index | min(lod,7) << 25 (model_definitions.rs:31-59) */
    uint32_t render_system;
    uint32_t sortable;
    uint32_t begin_instance;    /* InstanceRange.begin_instance */
    uint32_t count;             /* InstanceRange.count */
} re_instance_range;

/* Result of re_cull_pack.  Pointers are library-owned and stay valid until the next
 * re_cull_pack / re_destroy on the ctx. */
typedef struct {
    uint32_t n_visible_sections;      /* |CullResult.visible_sections_map| */
    uint32_t n_visible_vec;           /* visible_sections_vec.len() (duplicates counted) */
    uint32_t n_instances;             /* instances the reference would write (sum of group counts) */
    uint32_t n_written;               /* instances actually stored (min(n_instances, capacity)); the rest is
                                         truncated like MappedBuffer::write_data_serialized, mapped_buffer.rs:171-186 */
    uint32_t n_groups;
    const re_instance_range *groups;  /* host memory, n_groups entries */
    const uint32_t *d_entity_ids;     /* DEVICE memory, n_written entity ids, group after group */
    const float    *d_matrices;       /* DEVICE memory, n_written * 16 floats: TransformationMatrix, column-major,
                                         64 B/instance, byte-identical to what specify_type_ids! appends
                                         (prelude/layout_update_macros.rs:15-21) */
} re_visible;

/* re_tick flags */
#define RE_TICK_ALL_DYNAMIC 0x1u  /* tick every dynamic entity regardless of visibility (default: reference semantics --
                                     only entities in active visible world sections + AlwaysExecuteLogic entities) */
#define RE_TICK_ASYNC       0x2u

typedef struct {
    uint32_t n_changed;        /* entities whose change requests were applied */
    uint32_t n_rebucket;       /* of those, entities whose world section changed (re-bucketed) */
    uint32_t n_out_of_bounds;  /* entities add_entity rejected (see re_get_out_of_bounds) */
} re_tick_result;

/* components for re_read_component (ECS::get_copy<T>, objects/ecs.rs:653-664) */
#define RE_C_POSITION        0   /* 3 floats */
#define RE_C_ROTATION        1   /* 4 floats axis+angle */
#define RE_C_SCALE           2   /* 3 floats */
#define RE_C_VELOCITY        3   /* 3 floats */
#define RE_C_ACCELERATION    4   /* 3 floats */
#define RE_C_ROTATION_VEL    5   /* 4 floats */
#define RE_C_ROTATION_ACC    6   /* 4 floats */
#define RE_C_TRANSFORMATION  7   /* 16 floats */
#define RE_C_STATIC_AABB     8   /* 6 floats */
#define RE_C_ORIGINAL_AABB   9   /* 6 floats */
#define RE_C_FLAGS          10   /* 1 uint32: RE_F_* incl. HasMoved/HasRotated */

/* ---- lifecycle ---- */
int         re_create(const re_config *cfg, re_ctx **out);
void        re_destroy(re_ctx *ctx);
const char *re_last_error(const re_ctx *ctx);       /* ctx may be NULL: last error of re_create */
uint32_t    re_abi_version(void);

/* ---- world ---- */
/* Replaces the world with these entities: per entity TRS -> TransformationMatrix + StaticAABB on the
 * GPU, spatial-hash cell assignment, one end_of_changes.  n_rejected (optional) receives the number
 * of entities whose AABB is out of bounds (they are not inserted; apply_choices prints an error). */
int re_upload_entities(re_ctx *ctx, const re_entities *ents, uint32_t *n_rejected);

/* Custom level-of-view bands of one model == register_model_with_render_system(.., custom_level_of_view, ..) (flows/render_flow.rs:1069-1076): instances of
 * that model take their LOD from these bands instead of the camera's default ones (level_views.custom, render_flow.rs:495-499, 889-893;
 * ModelId::level_of_view_adjusted_model_index, models/model_definitions.rs:31-59: first band with min <= d <= max, else 7).  At most 8 bands; n_lod == 0
 * removes the model's bands.  Kept across re_upload_entities, like the model registration itself. */
int re_set_model_lod(re_ctx *ctx, uint32_t model_index, uint32_t render_system, uint32_t n_lod, const float *lod_min, const float *lod_max);

/* ---- frame ---- */
int re_cull_pack(re_ctx *ctx, const re_camera *cam, uint32_t flags, re_visible *out);
int re_tick(re_ctx *ctx, float delta_time, uint32_t flags, re_tick_result *out);
int re_wait(re_ctx *ctx, re_visible *out_visible /*nullable*/, re_tick_result *out_tick /*nullable*/);

/* n frames of the reference's frame loop (threads/render_thread.rs:217-250 -> Pipeline::execute) from native code: re_cull_pack(cam, cull_flags)
 * followed by re_tick(dt, tick_flags), n times -- nothing but the two calls above in a loop (with re_allgather_visible between them when the context has a
 * communicator), so that a measurement does not pay for an interpreter between them.  wall_us (nullable, n floats) receives the wall time of each frame; last_visible / last_tick (nullable) the results of the last one
 * when its call was synchronous. */
int re_run_frames(re_ctx *ctx, const re_camera *cam, float delta_time, uint32_t cull_flags, uint32_t tick_flags, uint32_t n,
                  float *wall_us, re_visible *last_visible, re_tick_result *last_tick);

/* Copy the packed instance buffer to host memory (the persistent-mapped GL buffer of
 * RenderSystem::get_instanced_mapped_buffers, render_system/render_system.rs:210-214).  At most
 * capacity_instances are written; *n_written receives the count (truncate-and-report). */
int re_copy_visible(re_ctx *ctx, uint32_t *entity_ids_host, float *matrices_host,
                    uint32_t capacity_instances, uint32_t *n_written);
/* Direct the packed output into caller-owned DEVICE buffers (e.g. the all-gather send slab);
 * NULLs restore the internal buffers. */
int re_set_output_buffers(re_ctx *ctx, uint32_t *d_entity_ids, float *d_matrices, uint32_t capacity_instances);
/* Optional header in DEVICE memory, FOUR consecutive words, filled by every later re_cull_pack (the header of an all-gather slab, so that the
 * exchange needs no host round trip): {instances written to the output buffers, instances of the frame before truncation to the buffers'
 * capacity, frame number, 0}.  A frame that cross-frame speculation cancelled (a tick found entities changing section; re_wait replays it) writes
 * 0xFFFFFFFF into the first word instead.  NULL switches it off.
 * The packed ids / matrices themselves are only STREAM-ORDERED: a synchronous re_cull_pack returns when the InstanceRange table and the counts are on the
 * host, which is before the last id / matrix store has landed -- whatever reads d_entity_ids / d_matrices (a copy, a collective, a draw) must be ordered
 * behind the pack on re_get_stream(ctx), as re_copy_visible and re_allgather_visible are. */
int re_set_output_count(re_ctx *ctx, uint32_t *d_count);

/* ---- multi-GPU exchange (BASELINE configs[3]): one process and one context per GPU, world sections sharded by contiguous key range, and ONE
 * exchange step per frame -- the all-gather of every GPU's packed visible-instance buffer over RCCL / xGMI.  Nothing in the reference corresponds to it (it
 * is a single-process program); it takes the place of RenderFlow handing its SortResult to the one render system (flows/render_flow.rs:401-410) when the
 * world is spread over several GPUs.  RCCL is loaded at run time (librccl.so; RE_RCCL_LIBRARY overrides the path).
 *   re_comm_unique_id  rank 0 creates the id (ncclGetUniqueId) and hands the 128 bytes to the other ranks over the host's own channel;
 *   re_comm_init       every rank: ncclCommInitRank on the context's device + the send / receive slabs of `slab_instances` instances each
 *                      ([4-word header | pad to 16 words | ids[cap] | matrices[cap * 16]]; size it for the EXPECTED visible set of a rank: xGMI is point to
 *                      point, every rank's slab crosses every link once per frame);
 *   re_comm_adopt      the same with a communicator (ncclComm_t) the host created itself.
 * With a communicator, re_cull_pack packs every frame straight into the frame's send slab (two alternate; RE_CULL_DEFER_PACK / RE_CULL_TWO_LANES are
 * ignored), and re_allgather_visible -- between re_cull_pack and re_tick -- enqueues the all-gather behind the pack on the context's stream and, unless
 * RE_GATHER_ASYNC, waits for it.  Every rank ends with the same buffer in rank order.  If a rank's visible set outgrew its slab (every rank reads that
 * from the gathered headers) a second, variable-length round follows: every rank packs its frame again, untruncated, and the full buffers are gathered
 * padded to the largest count (out->overflowed = 1; the rank strides then differ).  A frame that cross-frame speculation had cancelled is replayed and
 * gathered again (RE_GATHER_ASYNC + re_gather_wait: every rank must then call re_gather_wait for every frame). */
#define RE_COMM_ID_BYTES 128
#define RE_GATHER_ASYNC  0x1u
typedef struct {
    uint32_t n_ranks, overflowed;
    const uint32_t *counts;            /* host memory, n_ranks entries: instances of each rank's frame */
    const uint32_t *d_entity_ids;      /* DEVICE: rank r's ids start at d_entity_ids + r * ids_rank_stride */
    uint32_t ids_rank_stride;          /* in uint32 elements */
    const float *d_matrices;           /* DEVICE: rank r's 4x4 matrices (64 B each, column-major) start at d_matrices + r * matrices_rank_stride */
    uint32_t matrices_rank_stride;     /* in float elements */
} re_gathered;
int re_comm_unique_id(uint8_t id[RE_COMM_ID_BYTES]);
int re_comm_init(re_ctx *ctx, const uint8_t id[RE_COMM_ID_BYTES], int rank, int n_ranks, uint32_t slab_instances);
int re_comm_adopt(re_ctx *ctx, void *nccl_comm, int rank, int n_ranks, uint32_t slab_instances);
int re_comm_destroy(re_ctx *ctx);      /* also done by re_destroy */
int re_allgather_visible(re_ctx *ctx, uint32_t flags, re_gathered *out /* nullable */);
int re_gather_wait(re_ctx *ctx, re_gathered *out /* nullable */);

/* ---- entities that move from one GPU's share of the world to another's ("movers that cross a shard boundary need a second, sparse exchange") ----
 * A context owns the world sections whose key lies in [key_lo, key_hi) (re_set_shard_range; the smallest key of an entity's section(s) decides, as in
 * re_section_keys).  After a tick, re_list_migrants names the entities of this context whose section has left that range; re_export_entities returns
 * their complete state -- the record layout of EntityTransformationBuilder's choices, one re_entity_state per entity, the components as they are NOW --;
 * the host removes them here (RE_CHANGE_DELETE), routes each record to the owner of its new section over its own channel (a few records per frame) and
 * registers them there with re_add_entities.  Entities of a unique section live on exactly one GPU, so a mover between unique sections needs nothing
 * else; halo replicas (RE_F_PHANTOM) of shared sections that straddle the boundary are not migrated. */
typedef struct re_entity_state {
    uint32_t entity_id, model_index, render_system, sortable, flags;
    float original_aabb[6], position[3], rotation[4], scale[3], velocity[3], acceleration[3], rotation_velocity[4], rotation_acceleration[4];
} re_entity_state;
int re_set_shard_range(re_ctx *ctx, uint64_t key_lo, uint64_t key_hi);      /* key_lo == key_hi == 0: the context owns everything (default) */
int re_list_migrants(re_ctx *ctx, uint32_t *entity_ids, uint32_t capacity, uint32_t *n);   /* entities re-bucketed since the last call whose smallest section key is outside the range; drains the list */
int re_export_entities(re_ctx *ctx, const uint32_t *entity_ids, uint32_t n, re_entity_state *out);

/* Change requests returned by user logic (LogicFunction / CollisionFunction -> Vec<EntityChangeInformation>,
 * objects/entity_change_request.rs) == apply_change (helper_things/entity_change_helpers.rs:32-189) for the kinds that touch
 * this path.  Processed in list order with the reference's rules: the last write of a component wins; Position alone takes
 * the translation-only path (matrix column 3 + translated OriginalAABB), Rotation / Scale force the full TRS recompute; an
 * entity whose section is unchanged is left where it is; a deleted entity ignores later requests; an entity that leaves the
 * world is kept when RE_F_OOB_LOGIC, otherwise removed and reported (re_get_out_of_bounds).  Velocity-type components can be
 * written for any entity (one registered without Velocity / VelocityRotation gets its slot of the dynamic table on the first write).  The call belongs to the frame's logic phase (after
 * re_cull_pack): like Pipeline::execute it clears the changed-static-section set afterwards (pipeline.rs:271).
 * The reference's static render cache is a snapshot the logic phase never refreshes: an entity made static by a change is
 * therefore not drawn until it wakes up again (modelled), and a static entity of the snapshot that is woken, deleted or
 * rewritten keeps being drawn with its old bytes.  The latter is a ghost instance: a copy of {id, matrix} taken at the first
 * such change and parked behind the static rows of the section whose cache entry holds the entity (its own section, or the
 * linking section that cached its shared section); it is drawn whenever that section's cached static data is, also after the
 * section was emptied and re-created.  Ghosts live beyond the entity columns, max(2048, n/8) of them per upload
 * (RE_E_CAPACITY beyond that; the batch is then refused whole). */
#define RE_CHANGE_MODIFY       0u  /* ModifyRequest: component = RE_C_POSITION .. RE_C_ROTATION_ACC, value = the new component */
#define RE_CHANGE_DELETE       1u  /* DeleteRequest (:156-172) */
#define RE_CHANGE_MAKE_STATIC  2u  /* MakeObjectStatic (:112-122) */
#define RE_CHANGE_WAKE_UP      3u  /* WakeUpRequest (:123-133) */
#define RE_CHANGE_REMOVE_COMPONENT 4u /* RemoveComponent((EntityId, TypeIdentifier)) (:151-154) -> ECS::remove_component_type_id_internal (objects/ecs.rs:523-556):
                                       * component = RE_C_ROTATION, RE_C_SCALE, RE_C_VELOCITY, RE_C_ACCELERATION, RE_C_ROTATION_VEL or RE_C_ROTATION_ACC.  The
                                       * presence bit is cleared and nothing else happens (no matrix recompute); later reads of Rotation / Scale see the default
                                       * (movement_components.rs:41-55), kinematics skip an absent velocity component (logic_flow.rs:366-448). */
#define RE_CHANGE_ADD_ENTITY       5u  /* AddEntity (:48-107): create_entity + EntityTransformationBuilder::apply_choices, inline in list order.  reserved = index into the `added`
                                       * entities of re_apply_changes_ex; entity_id = the id ECS::create_entity hands out (objects/ecs.rs:384-402: the last freed id, else the next
                                       * one) -- the host's shim keeps that counter and free list -- and must equal added->entity_id[reserved].  Later changes of the same list may
                                       * name the new entity (AddEntity applies its own change request right away, :79).  An entity whose AABB lies out of bounds is created but not
                                       * inserted into the tree (apply_choices prints an error).  A static entity added inside a frame is in the tree's static set but not drawn until
                                       * its section is re-cached (the frozen static render cache, see above). */
#define RE_CHANGE_ADD_SORTABLE     6u  /* AddSortableComponent((EntityId, TypeIdentifier)) (:138-141) -> ECS::write_sortable_component (objects/ecs.rs:202-205): component = index of
                                       * the sortable component in registration order == re_entities.sortable; the entity's instances move to that (ModelId, sortable) group */
#define RE_CHANGE_REMOVE_SORTABLE  7u  /* RemoveSortableComponent (:142-146) -> back to the default sortable component (index 0) */
typedef struct re_change { uint32_t kind, entity_id, component, reserved; float value[4]; } re_change;
/* out: n_changed = entities whose matrix/AABB were recomputed, n_rebucket = of those, entities that changed section,
 * n_out_of_bounds = entities removed because they left the world */
int re_apply_changes(re_ctx *ctx, const re_change *changes, uint32_t n, uint32_t flags, re_tick_result *out /*nullable*/);
/* the same with the entities RE_CHANGE_ADD_ENTITY changes refer to (`added` may be NULL when the list adds none) */
int re_apply_changes_ex(re_ctx *ctx, const re_change *changes, uint32_t n, const re_entities *added, uint32_t flags, re_tick_result *out /*nullable*/);
/* Pipeline::register_model_instances at any time (flows/pipeline.rs:186-208): create_entity + apply_choices for every instance, then end_of_changes -- the
 * entities are APPENDED to the world (re_upload_entities replaces it; on a context without a world this call is the upload).  entity_id: ids not in use
 * (ids of removed entities may be reused, as ECS::create_entity does).  Between frames: the changed-static set of the tree is still there at the next
 * re_cull_pack, so a section that received a static entity is re-cached by that render from its live static set (render_flow.rs:549-594) -- unlike an
 * entity added inside a frame.  A velocity-type component written later to an entity registered without one gets a slot of the dynamic table then
 * (objects/entity_change_request.rs:29-30: the component is registered on write).  n_rejected (optional): entities out of bounds, created but not inserted. */
int re_add_entities(re_ctx *ctx, const re_entities *ents, uint32_t *n_rejected);

/* Collision broad phase of the frame == LogicFlow::handle_collisions (flows/logic_flow.rs:452-651) up to the collision-logic
 * callbacks: which (this_entity, other_entity) pairs the reference would hand to the CollisionFunction of this_entity's type.
 * Call order of a frame as in LogicFlow::execute (:230-244): re_cull_pack, re_collide, re_tick -- the tests read this frame's
 * StaticAABBs (update_positions only queues the kinematic changes).
 *   moved entities  = the entities the tick of this frame processes that carry Velocity or VelocityRotation and
 *                     RE_F_CAN_COLLIDE (once per listing of their world section in visible_sections_vec), plus the RE_F_USER entity;
 *   per world section holding one: the sections related to it (BoundingBoxTree::find_related_entities,
 *                     bounding_box_tree_v2.rs:950-1048: ancestors and descendants, transitively) whose AABB is within 200 units
 *                     of the camera, and their shared sections within 200 units;
 *   AABB overlap (closed intervals) of the moved entity with the non-static entities of those sections / all entities of those
 *                     shared sections: (moved, other), and (other, moved) too when other is not itself a moved entity.
 * The pairs come in no particular order (the reference's order depends on thread timing); as a multiset they equal the reference
 * with moved_entities taken in ascending EntityId (that order decides which entity the Shared arm of :488-498 drops).
 * *n_total = number of invocations; the first `capacity` are written to `pairs` (host memory). */
typedef struct { uint32_t this_entity, other_entity; } re_collision;
int re_collide(re_ctx *ctx, uint32_t flags /*0*/, re_collision *pairs, uint32_t capacity, uint32_t *n_total);

/* ECS read-back for user logic (LogicFunction reads components through &ECS, exports/logic_components.rs:15-18).  A component the entity does not
 * carry (never written, or removed by RE_CHANGE_REMOVE_COMPONENT) yields RE_E_ARG == ECS::get_copy -> None (objects/ecs.rs:653-664). */
int re_read_component(re_ctx *ctx, uint32_t entity_id, int component, void *dst);

/* ---- ECS presence semantics (objects/ecs.rs:61-72, 348-367, 457-556) ----
 * The reference keeps one bitset per entity: bit i is set once the i-th REGISTERED component type has been written for the entity and cleared by
 * remove_component / remove_entity.  Bit positions are registration order: ECS::new registers TypeIdentifier first (ecs.rs:141), LogicFlow::new the
 * engine's components after it (flows/logic_flow.rs:83-110).  re_ecs_bitset returns that bitset (bitsets[entity][0..4] as one little-endian word). */
#define RE_ECS_BIT_TYPE_IDENTIFIER      0   /* the entity-type marker (write_entity_type); not tracked here: always 0 */
#define RE_ECS_BIT_CAN_CAUSE_COLLISIONS 2
#define RE_ECS_BIT_HAS_MOVED            3
#define RE_ECS_BIT_POSITION             4
#define RE_ECS_BIT_VELOCITY             5
#define RE_ECS_BIT_ACCELERATION         6
#define RE_ECS_BIT_HAS_ROTATED          7
#define RE_ECS_BIT_ROTATION             8
#define RE_ECS_BIT_VELOCITY_ROTATION    9
#define RE_ECS_BIT_ACCELERATION_ROTATION 10
#define RE_ECS_BIT_SCALE                11
#define RE_ECS_BIT_TRANSFORMATION       12
#define RE_ECS_BIT_MODEL_ID             13
#define RE_ECS_BIT_STATIC_AABB          15
#define RE_ECS_BIT_ORIGINAL_AABB        16
#define RE_ECS_BIT_ALWAYS_EXECUTE_LOGIC 20
int re_ecs_bitset(re_ctx *ctx, uint32_t entity_id, uint32_t *bits);   /* 0 for an entity that was removed (remove_entity clears every bit, ecs.rs:557-600) */
/* Sharding helper: the world sections entity i is registered in -- n_keys[i] = 1 (its unique section) or 2..8 (the sections its shared
 * section links), 0 = rejected as out of bounds; keys[i * 8 ..].  A world is spread over several GPUs by contiguous ranges of an entity's SMALLEST key:
 * every unique section's entities, and every shared section together with the section that caches its static entities, then live on one shard.  What a
 * shard's static-section flags still depend on -- the entities of the sections its shared sections link, and the other shared sections linking those --
 * it uploads as halo replicas (RE_F_PHANTOM).  Host arithmetic (no device, no context); the functions re_upload_entities runs on the GPU. */
int re_section_keys(const re_config *cfg, const re_entities *entities, uint64_t *keys /* [n * 8] */, uint8_t *n_keys /* [n] */);

/* The lights of one type that RenderFlow::render hands to the deferred pass (upload_*_lights, render_system/render_system.rs:676-800) and to the shadow
 * flow: find_nearby_world_sections_maps (flows/render_flow.rs:249-254, flows/shadow_flow.rs:494-513: the whole-world visibility query with an AABB culler
 * of radius cam->far_draw around cam->position) followed by find_nearby_lights (shadow_flow.rs:455-487: the light sets of those unique world sections and
 * of the shared sections linked to them).  light_type: one of RE_F_LIGHT_DIRECTIONAL / RE_F_LIGHT_POINT / RE_F_LIGHT_SPOT.  *n = number found (may exceed
 * capacity); ids in ascending order.  Independent of re_cull_pack (it runs its own visibility test), after the movers of the last tick are in. */
int re_visible_lights(re_ctx *ctx, const re_camera *cam, uint32_t light_type, uint32_t *ids, uint32_t capacity, uint32_t *n);

/* ECS::get_indexes_for_components (objects/ecs.rs:238-285): the entities that carry ALL the given components (RE_C_*), in ascending EntityId (the
 * reference returns a BTreeSet).  *n = their number; the first `capacity` ids are written.  Runs on the GPU over the presence column. */
int re_ecs_query(re_ctx *ctx, const int *components, uint32_t n_components, uint32_t *entity_ids, uint32_t capacity, uint32_t *n);
/* entity ids removed because they left the world without OutOfBoundsLogic (update_entity_in_tree, entity_change_helpers.rs:
 * 325-351) by ticks and change batches since the previous call; the call drains the list */
int re_get_out_of_bounds(re_ctx *ctx, uint32_t *entity_ids, uint32_t capacity, uint32_t *n);

/* ---- introspection (parity tests, profiling) ---- */
typedef struct {
    uint32_t n_entities, n_dynamic, n_sections, n_shared_sections, max_level;
    uint64_t device_bytes;
    uint32_t n_probe_frames, n_table_rebuilds, n_fused_frames, reserved /* lane switches */;   /* frames served by the probe path (RE_CFG_PROBE); full section-table rebuilds so far */
    uint32_t n_seal_waits;      /* result blocks in mapped host memory (frame result + InstanceRange table, tick counters, collision header) whose
                                 * seal did not agree when the polled "done" word became visible.  0 while the publication protocol holds. */
    uint32_t n_sync_fallbacks;  /* of those, blocks that only agreed after a stream synchronise */
    uint32_t n_section_slots;   /* slots of the resident section table == keys the visibility scan streams per frame (world sections + padding / spare slots) */
    uint32_t n_device_rebuckets; /* re-bucket batches (movers of a tick that changed world section) whose bookkeeping ran on the device; the others took the host path */
    uint32_t n_segment_redos;    /* frames issued a second time because clustered world sections overflowed one cursor segment of the instance list: the frame is redone
                                  * with the list as one segment (which holds every instance of the world twice), and the context keeps that layout until the next upload */
    uint32_t n_host_rebuckets;  /* batches of section changes (or the rest of one the device took in part) whose bookkeeping ran on the host: change-request batches, static movers,
                                 * batches that touch a section parking ghost instances of the frozen render cache, a table without slack */
} re_stats;
int re_get_stats(re_ctx *ctx, re_stats *out);
/* world sections in ascending key order: key = level<<48 | x<<32 | z<<16 | y (UniqueWorldSectionId field order,
 * bounding_box_tree_v2.rs:21-26); tight = UniqueWorldSectionEntities.aabb (6 floats each).  Any pointer may be NULL. */
int re_debug_get_sections(re_ctx *ctx, uint32_t capacity, uint64_t *keys, float *tight_aabb6,
                          uint32_t *n_local, uint32_t *n_static, uint8_t *is_static_section, uint32_t *n);
/* The shared world sections in canonical id order (keys lexicographic, then count): the 2..8 section keys of each, its AABB (end_of_changes, shared branch:
 * world/bounding_box_tree_v2.rs:1104-1125), the number of active / static members and their EntityIds (member_offsets[i] .. member_offsets[i + 1]; active first,
 * each part in ascending EntityId).  Read from the device table; RE_E_STATE when the host mirrors of the library are out of step with it. */
int re_debug_get_shared_sections(re_ctx *ctx, uint32_t capacity, uint64_t *keys /* [capacity * 8] */, uint8_t *n_keys, float *aabb6, uint32_t *n_active, uint32_t *n_static,
                                 uint32_t member_capacity, uint32_t *member_ids, uint32_t *member_offsets /* [capacity + 1] */, uint32_t *n);
/* visible_sections_map of the last cull, ascending; multiplicity[i] = 2 when the section is in both
 * the logic and the render result (appears twice in visible_sections_vec). */
int re_debug_get_visible_sections(re_ctx *ctx, uint32_t capacity, uint64_t *keys, uint8_t *multiplicity, uint32_t *n);
/* device time of the kernels of the last synchronous cull_pack / tick, microseconds (hipEvent on the ctx stream).  The first call
 * switches the event recording on (it costs ~12 us per synchronous frame, so it is off until asked for) and returns zeros; call again
 * after the next frame.  A call with three NULL pointers switches the recording off again. */
int re_get_timings(re_ctx *ctx, float *cull_us, float *pack_us, float *tick_us);
/* per-launch HIP-event timing of one kernel over a timed region (the events are tied to the dispatch itself, so the figure holds in synchronous and in
 * asynchronous frame loops alike), sampling every `every`-th launch (0 or 1 = all): re_timing_begin(ctx, max_launches, every | RE_TIME_x << 16);
 * ...frames...; re_timing_collect(ctx, us, cap, &n) (synchronises).  RE_TIME_SCAN (0) = the dominant kernel, the section-key scan + cull. */
#define RE_TIME_SCAN        0u
#define RE_TIME_TICK        1u
#define RE_TIME_PACK_LARGE  2u
int re_timing_begin(re_ctx *ctx, uint32_t max_launches, uint32_t every);
int re_timing_collect(re_ctx *ctx, float *microseconds, uint32_t capacity, uint32_t *n);
/* number of world sections inside a candidate box in the last cull (== hash probes the reference would make) */
int re_get_last_candidates(re_ctx *ctx, uint32_t *n_candidates);
/* bytes of DEVICE memory of the ctx (a packed or gathered buffer) into host memory, ordered behind the work on the ctx's stream -- for hosts that hold no HIP
 * runtime of their own, or a different one (a process may carry a second ROCm stack: a pointer of this library means nothing to that one) */
int re_debug_copy_to_host(re_ctx *ctx, const void *d_src, void *dst, uint64_t bytes);
/* the HIP stream of the ctx (hipStream_t as void*) so callers can order their own work after it */
void *re_get_stream(re_ctx *ctx);

/* ------------------------------------------------------------------------------------------------------------
 * Deferred lighting (BASELINE.json configs[4]): the second pass of RenderSystem::draw
 * (render_system/render_system.rs:507-585) as a compute kernel -- the math of
 * render_engine_assets/shaders/second_pass_frag.glsl:20-139, light uniform arrays as uploaded by
 * render_system.rs:752-766 (cone "point" lights) and :814-830 (radius "spot" lights).
 * ------------------------------------------------------------------------------------------------------------ */
typedef struct re_lighting re_lighting;
typedef struct { int32_t device; uint32_t width, height, max_spot_lights, max_point_lights; } re_lighting_config;
typedef struct {
    uint32_t n_spot, n_point;
    const float *spot_pos, *spot_diffuse, *spot_specular;                 /* n_spot*3 */
    const float *spot_ambient;                                            /* n_spot*4 (rgb, a) */
    const float *spot_linear, *spot_quadratic, *spot_radius;              /* n_spot */
    const float *point_pos, *point_dir, *point_diffuse, *point_specular;  /* n_point*3 */
    const float *point_ambient;                                           /* n_point*4 */
    const float *point_linear, *point_quadratic, *point_cutoff, *point_outer_cutoff;   /* n_point */
    float camera_pos[3];
    float no_light_source_cutoff, default_diffuse_factor;                 /* render_system_setup.rs:24-25 */
    uint32_t any_light_source_visible;
} re_lights;
int         re_lighting_create(const re_lighting_config *cfg, re_lighting **out);
void        re_lighting_destroy(re_lighting *l);
const char *re_lighting_last_error(const re_lighting *l);
/* G-buffer of the first pass (prelude/default_render_system.rs:104-107): gPosition / gNormal RGBA32F, gAlbedoSpec RGBA8; host pointers, copied.
 * gLightPosition is not needed: the shadow value it feeds is discarded by the shader (second_pass_frag.glsl:105). */
int re_lighting_upload_gbuffer(re_lighting *l, const float *g_position, const float *g_normal, const uint8_t *g_albedo_spec);
int re_lighting_set_lights(re_lighting *l, const re_lights *lights);
int re_lighting_run(re_lighting *l, float *kernel_microseconds /* nullable */);      /* FragColor RGBA32F stays in HBM */
int re_lighting_read(re_lighting *l, float *out_rgba);                                 /* width*height*4 floats */
int re_lighting_read_pixels(re_lighting *l, const uint32_t *pixel_index, uint32_t n, float *out_rgba);

/* ---- history / replay wire format ----
 * The per-frame FrameChange records of the history thread (threads/public_common_structures.rs:7-16), written with bincode 1.3 as
 * threads/history_thread.rs:150-205 does: gameplay_history.txt = bincode(ECS) | bincode(BoundingBoxTree) | bincode(FrameChange)*, and
 * gameplay_byte_lookup.txt = the byte length of every blob, one decimal number per line; read back as GameLoadResult::load does
 * (helper_things/game_loader.rs:32-71: the LAST line of the lookup file is the empty string after the final newline and is skipped).
 * Host code only.  The first two blobs are carried as opaque bytes (this library's state is SoA columns, not the reference's hash maps).
 * TypeIdentifier values are std::any::TypeId bits of one build of the reference (objects/ecs.rs:92-110): the host passes the ids of the
 * components this path carries. */
typedef struct re_history re_history;
typedef struct re_type_ids { uint64_t position, rotation, scale, velocity, acceleration, rotation_velocity, rotation_acceleration, has_moved, has_rotated; } re_type_ids;
#define RE_FC_CAMERA_VIEW_CHANGE        0u  /* f[0..2] = position, f[3..5] = direction (SerializableCameraInfo, exports/camera_object.rs:47-53) */
#define RE_FC_CAMERA_STATIONARY         1u
#define RE_FC_DELTA_TIME                2u  /* f[0] */
#define RE_FC_DRAW_DISTANCES_CHANGE     3u  /* f[0..2] = near, far, fov */
#define RE_FC_WINDOW_DIMENSIONS_CHANGE  4u  /* i[0..1] = width, height */
#define RE_FC_ENTITY_CHANGE             5u  /* changes[0..n_changes): Vec<EntityChangeInformation>, the variants re_apply_changes knows */
#define RE_FC_END_FRAME_CHANGE          6u
typedef struct re_frame_change { uint32_t kind; float f[6]; int32_t i[2]; uint32_t n_changes; const re_change *changes; } re_frame_change;
#define RE_HISTORY_VEC3_AS_ARRAY 1u         /* TVec3 as 12 bytes instead of a serde sequence (u64 count + 12 bytes); see re_history.cpp */
int         re_history_create(const re_type_ids *ids, uint32_t flags, re_history **out);
void        re_history_destroy(re_history *h);
const char *re_history_last_error(const re_history *h /* NULL: the error of a failed re_history_load */);
int re_history_set_state(re_history *h, const void *ecs_blob, uint64_t ecs_bytes, const void *tree_blob, uint64_t tree_bytes);
int re_history_get_state(re_history *h, const void **ecs_blob, uint64_t *ecs_bytes, const void **tree_blob, uint64_t *tree_bytes);
int re_history_record(re_history *h, const re_frame_change *fc);                     /* appends a copy (incl. the change list) */
int re_history_count(re_history *h, uint32_t *n);
int re_history_get(re_history *h, uint32_t index, re_frame_change *out);             /* out->changes points into the history object */
int re_history_encode(re_history *h, uint32_t index, uint8_t *dst /*nullable*/, uint64_t capacity, uint64_t *n_bytes);   /* one record's bincode bytes */
int re_history_write(re_history *h, const char *history_path, const char *lookup_path);
int re_history_load(const re_type_ids *ids, uint32_t flags, const char *history_path, const char *lookup_path, re_history **out);

#ifdef __cplusplus
}
#endif
#endif
