/*
 * re_oracle.h -- CPU ORACLE for the render_engine visible-set pipeline.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load this library.  The product path
 * (render_engine_amd/, include/re_hip.h) never links, imports or calls it.
 *
 * It is a plain-C restatement of the reference's algorithm (Rust, /root/reference/src);
 * every function cites the reference file:line it follows.  The reference cannot be
 * compiled here (no cargo/rustc, crates un-vendored), so:
 *
 *   PARITY PINNING
 *   - pinned by the reference's own known-answer tests: spatial-hash cell assignment
 *     (world/bounding_box_tree_v2.rs:1602-2319, transcribed to tests/golden/tree_cells.json),
 *     UniqueWorldSectionId::to_aabb (:2306-2319), removal sequences (:1837-2217).
 *   - PARITY UNPINNED for everything that goes through nalgebra/nalgebra-glm 0.25.4/0.11.0
 *     (not vendored under /root/reference): frustum planes, corner test, kinematics,
 *     TRS->mat4, AABB transform, distance_to_aabb, LOD.  The reference holds no tests for
 *     those; the formulas below restate nalgebra's published algorithms (operation order
 *     documented per function) and are cross-checked by an independent numpy-float32 mirror
 *     in tests/.
 *
 * Deterministic choices where the reference depends on hashbrown iteration order
 * (documented in DESIGN.md "hash-order quirks"): sets iterate in ascending EntityId,
 * maps in ascending key order.
 *
 * sin/cos: the reference calls f32::sin_cos (platform libm, <=1 ulp, platform dependent).
 * The oracle and the HIP kernels share one deterministic evaluation (ro_sincosf: f64
 * Cody-Waite reduction + fdlibm kernel polynomials, rounded once to f32), so the GPU can be
 * compared bit-for-bit with the oracle while both stay within 1 ulp of any libm.
 */
#ifndef RE_ORACLE_H
#define RE_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* StaticAABB: x_range, y_range, z_range (world/bounding_volumes/aabb.rs:7-12) */
typedef struct { float xmin, xmax, ymin, ymax, zmin, zmax; } ro_aabb;

/* component-presence / behaviour flags of an entity description */
#define RO_F_STATIC      0x001u  /* EntityTransformationBuilder.is_entity_static */
#define RO_F_HAS_VEL     0x002u  /* Velocity written */
#define RO_F_HAS_ACC     0x004u  /* Acceleration written */
#define RO_F_HAS_ROT     0x008u  /* Rotation written */
#define RO_F_HAS_ROTVEL  0x010u  /* VelocityRotation written */
#define RO_F_HAS_ROTACC  0x020u  /* AccelerationRotation written */
#define RO_F_HAS_SCALE   0x040u  /* Scale written */
#define RO_F_ALWAYS_EXEC 0x080u  /* AlwaysExecuteLogic written */
#define RO_F_OOB_LOGIC   0x100u  /* entity type has OutOfBoundsLogic (add_if_out_bounds) */
#define RO_F_HAS_MOVED   0x200u  /* HasMoved marker (output) */
#define RO_F_USER        0x800u  /* the user entity (flows/pipeline.rs:125-144): identity TransformationMatrix, StaticAABB = OriginalAABB translated */
#define RO_F_HAS_ROTATED 0x400u  /* HasRotated marker (output) */
#define RO_F_LIGHT_DIRECTIONAL 0x2000u /* EntityTransformationBuilder::new(.., Some(FindLightType::Directional), ..): member of its section's light set */
#define RO_F_LIGHT_POINT  0x4000u
#define RO_F_LIGHT_SPOT   0x8000u
#define RO_F_CAN_COLLIDE 0x1000u /* CanCauseCollisions (EntityTransformationBuilder.can_cause_collision, entity_transformer.rs:66-69) */

/* One entity as EntityTransformationBuilder would be filled (exports/entity_transformer.rs:12-29) */
typedef struct {
    uint32_t id;             /* EntityId */
    uint32_t model_index;    /* ModelId.model_index (bits 25..31 reserved for LOD) */
    uint32_t render_system;  /* ModelId.render_system_index */
    uint32_t sortable;       /* sortable-component bucket: 0 default,1 Directional,2 Point,3 Spot */
    uint32_t flags;          /* RO_F_* */
    ro_aabb  original;       /* OriginalAABB (model space) */
    float pos[3];
    float rot_axis[3]; float rot_angle;
    float scale[3];
    float vel[3];
    float acc[3];
    float rotvel_axis[3]; float rotvel;
    float rotacc_axis[3]; float rotacc;
} ro_entity_desc;

/* Camera inputs of Pipeline::execute (flows/pipeline.rs:216-229, render_flow.rs:389-399) */
typedef struct {
    float pv[16];          /* projection*view, column-major (nalgebra storage order) */
    float pos[3];
    float dir[3];
    float far_draw;        /* Camera::get_far_draw_distance */
    uint32_t n_lod;        /* default LevelOfView bands (<=8) */
    float lod_min[8];
    float lod_max[8];
} ro_camera;

/* one (ModelId, sortable) group of the packed instance buffer (render_flow.rs:964-983) */
typedef struct {
    uint32_t model_index;   /* LOD-adjusted */
    uint32_t render_system;
    uint32_t sortable;
    uint32_t begin;         /* InstanceRange.begin_instance */
    uint32_t count;         /* InstanceRange.count */
} ro_group;

typedef struct ro_world ro_world;

/* ---- stateless math (exported for unit tests and the numpy mirror) ---- */
void     ro_sincosf(float x, float *s, float *c);
float    ro_norm3(float x, float y, float z);
void     ro_mat4_mul(const float *a, const float *b, float *out);            /* nalgebra gemm order */
void     ro_mat4_vec4(const float *m, const float *v, float *out);
void     ro_trs_matrix(const float pos[3], int has_rot, const float axis[3], float angle,
                       int has_scale, const float scale[3], float out[16]);   /* entity_transformer.rs:99-142 */
ro_aabb  ro_apply_transformation(ro_aabb a, const float m[16]);              /* aabb.rs:95-114 */
ro_aabb  ro_combine_aabb(ro_aabb a, ro_aabb b);                              /* range.rs:38-61 */
float    ro_distance_to_aabb(ro_aabb a, const float cam[3]);                 /* aabb_helper_functions.rs:58-72 */
void     ro_make_planes(const float pv[16], float planes[24]);               /* render_frustum_culler.rs:59-78 */
int      ro_frustum_aabb_visible(const float planes[24], ro_aabb a);         /* render_frustum_culler.rs:83-118 */
int      ro_logic_aabb_in_view(float lookahead, const float cam[3], ro_aabb a); /* logic_frustum_culler.rs:32-46 */
uint32_t ro_lod_adjusted_model_index(uint32_t model_index, float d, uint32_t n,
                                     const float *lmin, const float *lmax);   /* model_definitions.rs:31-59 */
void     ro_default_lod(float render_distance, float lmin[5], float lmax[5]);/* default_render_system.rs:240-256 */
uint32_t ro_max_level(uint32_t outline, uint32_t atomic);                    /* bounding_box_tree_v2.rs:1356-1359 */
uint64_t ro_pack_key(uint32_t level, uint32_t x, uint32_t z, uint32_t y);
ro_aabb  ro_key_to_aabb(uint64_t key, uint32_t atomic);                      /* bounding_box_tree_v2.rs:95-109 */
/* cell assignment: returns number of unique world sections (1 => keys[0] is the Unique id,
 * >1 => Shared section made of keys[0..n)); *oob = aabb_out_of_bounds before clamping */
int      ro_assign_cells(ro_aabb a, uint32_t outline, uint32_t atomic, uint64_t keys[8], int *oob); /* :451-551,1298-1397 */
/* camera helpers (restated nalgebra Perspective3::new / look_at_rh via glm conventions) */
void     ro_perspective(float aspect, float fovy, float znear, float zfar, float out[16]);
void     ro_look_at(const float eye[3], const float target[3], const float up[3], float out[16]);

/* ---- world (BoundingBoxTree + the ECS columns the hot path reads) ---- */
ro_world *ro_world_new(uint32_t outline, uint32_t atomic);
void      ro_world_free(ro_world *w);
void      ro_set_model_lod(ro_world *w, uint32_t model_index, uint32_t render_system, uint32_t n, const float *lmin, const float *lmax); /* level_views.custom, render_flow.rs:889-893 */
void      ro_set_threads(ro_world *w, int nthreads);        /* rayon pool size for the par_chunks sites */

/* Pipeline::register_model_instances (pipeline.rs:186-208): create n entities, apply_choices,
 * end_of_changes.  Returns the number rejected as out of bounds. */
int  ro_register_entities(ro_world *w, uint32_t n, const ro_entity_desc *descs);
/* raw BoundingBoxTree::add_entity / remove_entity on a bare AABB (tree known-answer tests);
 * returns 0 Ok, -1 Err(()) */
int  ro_tree_add(ro_world *w, uint32_t id, ro_aabb a, int add_if_out_bounds, int is_static);
void ro_tree_remove(ro_world *w, uint32_t id);
void ro_end_of_changes(ro_world *w);

/* introspection */
uint32_t ro_num_cells(const ro_world *w);
uint32_t ro_num_shared(const ro_world *w);
/* fills up to cap cells in ascending key order; returns number of cells */
uint32_t ro_get_cells(const ro_world *w, uint32_t cap, uint64_t *keys, ro_aabb *tight,
                      uint32_t *n_local, uint32_t *n_static, uint32_t *n_shared, uint8_t *is_static_section);
/* entity ids of one cell: active (ascending id) then static (ascending id); returns total */
uint32_t ro_get_cell_entities(const ro_world *w, uint64_t key, uint32_t cap, uint32_t *ids, uint32_t *n_local);
/* entity lookup: returns 0 none, 1 unique (keys[0]), 2 shared (keys[0..*nkeys)) */
int  ro_entity_lookup(const ro_world *w, uint32_t id, uint64_t keys[8], int *nkeys);
/* copies entity state; returns 0 if the entity does not exist */
int  ro_get_entity(const ro_world *w, uint32_t id, float mat[16], ro_aabb *aabb, float pos[3],
                   float rot[4], float rotvel[4], float vel[3], uint32_t *flags);
/* shared section i (ascending canonical order): keys, entity ids (active then static) */
int  ro_get_shared(ro_world *w, uint32_t i, uint64_t keys[8], int *nkeys, ro_aabb *aabb,
                   uint32_t cap, uint32_t *ids, uint32_t *n_active, uint32_t *n_static);

/* ---- one frame, in the order of Pipeline::execute (pipeline.rs:212-276) ---- */
/* cull: fills the world's CullResult.  Returns visible_sections_vec.len() (duplicates
 * included, pipeline.rs:228).  keys_out (cap entries) receives the vec in sorted order. */
uint32_t ro_frame_cull(ro_world *w, const ro_camera *cam, uint32_t cap, uint64_t *keys_out);
uint32_t ro_visible_lights(ro_world *w, const ro_camera *cam, uint32_t type_flag, uint32_t cap, uint32_t *ids_out);
/* render gather + pack (render_flow.rs:401-410).  emit_duplicates!=0 reproduces the reference's
 * double emission for cells present twice in visible_sections_vec; 0 emits each instance once
 * (the ID *set*).  ids/mats (cap instances) are written group after group; returns total
 * instances (may exceed cap; nothing beyond cap is written).  groups: up to gcap entries. */
uint32_t ro_frame_render(ro_world *w, const ro_camera *cam, int emit_duplicates,
                         uint32_t cap, uint32_t *ids, float *mats,
                         uint32_t gcap, ro_group *groups, uint32_t *n_groups);
/* logic tick (logic_flow.rs:230 update_positions + :255 update_bounding_box_tree/apply_change).
 * Uses the CullResult of the last ro_frame_cull.  Returns number of entities whose change
 * requests were applied; oob_ids (cap) receives entities rejected by add_entity. */
uint32_t ro_frame_tick(ro_world *w, const ro_camera *cam, float dt,
                       uint32_t cap, uint32_t *oob_ids, uint32_t *n_oob);
/* collision broad phase (logic_flow.rs:452-651): call after ro_frame_cull and before ro_frame_tick of the same frame.
 * Returns the number of collision-logic invocations; the first cap (this_entity, other_entity) pairs go to pairs. */
uint32_t ro_frame_collide(ro_world *w, const ro_camera *cam, uint32_t cap, uint32_t *pairs);
/* related_world_sections[key] (bounding_box_tree_v2.rs:334): the existing ancestor and descendant sections */
uint32_t ro_related_sections(const ro_world *w, uint64_t key, uint32_t cap, uint64_t *out);
/* find_related_entities of one section (:950-1048): unique section keys; shared sections as {nkeys, keys...} records */
uint32_t ro_find_related(const ro_world *w, uint64_t key, uint32_t cap, uint64_t *unique_keys, uint32_t *n_unique,
                         uint32_t shared_cap, uint64_t *shared_keys, uint32_t *n_shared);

/* FrameChange::EntityChange entries that touch this path (objects/entity_change_request.rs; applied by
 * helper_things/entity_change_helpers.rs:32-189).  component: 0 Position, 1 Rotation(axis,angle), 2 Scale, 3 Velocity,
 * 4 Acceleration, 5 VelocityRotation(axis,rate), 6 AccelerationRotation(axis,rate). */
#define RO_CHANGE_MODIFY      0u
#define RO_CHANGE_DELETE      1u
#define RO_CHANGE_MAKE_STATIC 2u
#define RO_CHANGE_WAKE_UP     3u
#define RO_CHANGE_REMOVE_COMPONENT 4u   /* component = 1..6 (Rotation, Scale, Velocity, Acceleration, VelocityRotation, AccelerationRotation) */
typedef struct { uint32_t kind, entity_id, component, pad; float value[4]; } ro_change;
#define RO_CHANGE_ADD_ENTITY      5u   /* pad = index into `added` (ro_apply_changes_ex) */
#define RO_CHANGE_ADD_SORTABLE    6u   /* component = sortable index */
#define RO_CHANGE_REMOVE_SORTABLE 7u
uint32_t ro_apply_changes(ro_world *w, const ro_change *changes, uint32_t n, int end_of_frame, uint32_t cap, uint32_t *oob_ids, uint32_t *n_oob);
uint32_t ro_apply_changes_ex(ro_world *w, const ro_change *changes, uint32_t n, const ro_entity_desc *added, uint32_t n_added, int end_of_frame, uint32_t cap, uint32_t *oob_ids, uint32_t *n_oob);

/* ---- config 5: deferred lighting, CPU evaluation of render_engine_assets/shaders/second_pass_frag.glsl:20-139 ----
 * Light uniform arrays as RenderSystem uploads them (render_system/render_system.rs:752-766, 814-830). */
typedef struct {
    uint32_t n_spot, n_point;
    const float *spot_pos;      /* n_spot*3 */
    const float *spot_diffuse, *spot_specular;   /* n_spot*3 each */
    const float *spot_ambient;  /* n_spot*4 (rgb, a) */
    const float *spot_linear, *spot_quadratic, *spot_radius;   /* n_spot each */
    const float *point_pos, *point_dir, *point_diffuse, *point_specular;   /* n_point*3 each */
    const float *point_ambient; /* n_point*4 */
    const float *point_linear, *point_quadratic, *point_cutoff, *point_outer_cutoff;   /* n_point each */
    float camera_pos[3];
    float no_light_source_cutoff, default_diffuse_factor;
    uint32_t any_light_source_visible;
} ro_lights;
/* gpos/gnormal: RGBA32F (4 floats per pixel), galbedo: RGBA8; out: RGBA32F.  Only pixels listed in idx (n of them) when idx != NULL. */
void ro_deferred_lighting(uint32_t npix, const float *gpos, const float *gnormal, const uint8_t *galbedo, const ro_lights *L,
                          const uint32_t *idx, uint32_t n, float *out);
/* work count of config 5: (pixel, spot light) pairs within the light radius (exact; lights binned on an x-z grid) */
uint64_t ro_lighting_spot_pairs(uint32_t npix, const float *gpos, const ro_lights *L);

#ifdef __cplusplus
}
#endif
#endif
