// render_engine_hip.hpp -- C++ host-side mirror of the reference's Rust interface for the visible-set path, over the C ABI
// of re_hip.h (librender_engine_hip.so).  Header-only, C++17, no HIP or torch types.
//
// The reference is one Rust crate without an FFI layer and this image has no Rust toolchain, so the host side above the C
// ABI is written in C++ with the reference's own names, argument meaning and error behaviour (the Rust shim a maintainer
// adds is in INTEGRATION.md).  Mirrors, file:line under /root/reference/src:
//   StaticAABB, XRange/YRange/ZRange        world/bounding_volumes/aabb.rs:7-141, world/dimension/range.rs:7-99
//   Position .. AccelerationRotation         exports/movement_components.rs:13-164 (axis constructors normalise)
//   EntityTransformationBuilder              exports/entity_transformer.rs:14-142 (check_invariants asserts -> exceptions)
//   CameraBuilder / Camera                   exports/camera_object.rs (perspective + look_at of nalgebra-glm)
//   RenderFrustumCuller / LogicFrustumCuller culling/render_frustum_culler.rs:36-118, culling/logic_frustum_culler.rs:22-46
//   Pipeline::{register_model_instances, execute}, ECS::{create_entity, get_copy, write_sortable_component}
//                                            flows/pipeline.rs:186-276, objects/ecs.rs:199-205,384-402,653-664
//   EntityChangeRequest / EntityChangeInformation  objects/entity_change_request.rs, applied by helper_things/entity_change_helpers.rs:32-189
#pragma once
#include <array>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <functional>
#include <optional>
#include <stdexcept>
#include <string>
#include <vector>

#include "re_hip.h"

namespace render_engine {

struct Error : std::runtime_error { int code; Error(int c, const std::string &m) : std::runtime_error(m), code(c) {} };

using EntityId = uint32_t;
struct TVec3 { float x = 0, y = 0, z = 0; };
inline TVec3 vec3(float x, float y, float z) { return TVec3{ x, y, z }; }
using Mat4 = std::array<float, 16>;                      // column-major, like nalgebra

struct XRange { float min, max; }; struct YRange { float min, max; }; struct ZRange { float min, max; };
struct StaticAABB {
    XRange x_range; YRange y_range; ZRange z_range;
    static StaticAABB new_(XRange x, YRange y, ZRange z) { return StaticAABB{ x, y, z }; }
    void translate(TVec3 t) { x_range.min += t.x; x_range.max += t.x; y_range.min += t.y; y_range.max += t.y; z_range.min += t.z; z_range.max += t.z; }   // aabb.rs translate
};
struct ModelId { uint32_t model_index = 0; uint32_t render_system_index = 0; };

namespace detail {
inline TVec3 normalize(TVec3 v) { float n = std::sqrt((v.x * v.x + v.y * v.y) + v.z * v.z); return TVec3{ v.x / n, v.y / n, v.z / n }; }
}
// movement components (exports/movement_components.rs); the rotation types normalise their axis in ::new
struct Position { TVec3 v; static Position new_(TVec3 p) { return Position{ p }; } TVec3 get_position() const { return v; } };
struct Velocity { TVec3 v; static Velocity new_(TVec3 p) { return Velocity{ p }; } TVec3 get_velocity() const { return v; } };
struct Acceleration { TVec3 v; static Acceleration new_(TVec3 p) { return Acceleration{ p }; } TVec3 get_acceleration() const { return v; } };
struct Scale { TVec3 v{ 1, 1, 1 }; static Scale new_(TVec3 p) { return Scale{ p }; } TVec3 get_scale() const { return v; } };
struct Rotation {
    TVec3 axis{ 1, 0, 0 }; float angle = 0;                                   // Rotation::default (:41-47)
    static Rotation new_(TVec3 a, float radians) { return Rotation{ detail::normalize(a), radians }; }
    TVec3 get_rotation_axis() const { return axis; } float get_rotation() const { return angle; }
};
struct VelocityRotation { TVec3 axis{ 1, 0, 0 }; float rate = 0; static VelocityRotation new_(TVec3 a, float r) { return VelocityRotation{ detail::normalize(a), r }; } };
struct AccelerationRotation { TVec3 axis{ 1, 0, 0 }; float rate = 0; static AccelerationRotation new_(TVec3 a, float r) { return AccelerationRotation{ detail::normalize(a), r }; } };
struct TransformationMatrix { Mat4 m; const Mat4 &get_matrix() const { return m; } };

enum class FindLightType { Directional, Point, Spot };

// ---- camera (exports/camera_object.rs): right-handed perspective * look_at, nalgebra operation order ----
namespace detail {
inline Mat4 perspective(float aspect, float fovy, float znear, float zfar) {     // nalgebra Perspective3::new
    Mat4 m{}; float m11 = 1.0f / std::tan(fovy / 2.0f);
    m[5] = m11; m[0] = m11 / aspect; m[10] = (zfar + znear) / (znear - zfar); m[14] = zfar * znear * 2.0f / (znear - zfar); m[11] = -1.0f;
    return m;
}
inline Mat4 look_at(TVec3 eye, TVec3 target, TVec3 up) {
    TVec3 f = normalize(TVec3{ target.x - eye.x, target.y - eye.y, target.z - eye.z });
    TVec3 s = normalize(TVec3{ f.y * up.z - f.z * up.y, f.z * up.x - f.x * up.z, f.x * up.y - f.y * up.x });
    TVec3 u{ s.y * f.z - s.z * f.y, s.z * f.x - s.x * f.z, s.x * f.y - s.y * f.x };
    Mat4 m{}; m[15] = 1.0f;
    m[0] = s.x; m[4] = s.y; m[8] = s.z; m[1] = u.x; m[5] = u.y; m[9] = u.z; m[2] = -f.x; m[6] = -f.y; m[10] = -f.z;
    m[12] = -((s.x * eye.x + s.y * eye.y) + s.z * eye.z); m[13] = -((u.x * eye.x + u.y * eye.y) + u.z * eye.z); m[14] = ((f.x * eye.x + f.y * eye.y) + f.z * eye.z);
    return m;
}
inline Mat4 mul(const Mat4 &a, const Mat4 &b) {                                   // nalgebra gemm accumulation order, k = 0..3
    Mat4 o{};
    for (int j = 0; j < 4; j++) for (int i = 0; i < 4; i++) {
        float y = a[0 * 4 + i] * b[j * 4 + 0]; y = a[1 * 4 + i] * b[j * 4 + 1] + y; y = a[2 * 4 + i] * b[j * 4 + 2] + y; y = a[3 * 4 + i] * b[j * 4 + 3] + y;
        o[j * 4 + i] = y;
    }
    return o;
}
}  // namespace detail

struct LevelOfView { float min_distance, max_distance; };
inline std::vector<LevelOfView> create_level_of_views(float render_distance) {    // prelude/default_render_system.rs:240-256
    float v1 = render_distance * 0.10f, v2 = render_distance * 0.15f + v1, v3 = render_distance * 0.20f + v2, v4 = render_distance * 0.25f + v3, v5 = render_distance * 0.30f + v4;
    return { { 0.0f, v1 }, { v1, v2 }, { v2, v3 }, { v3, v4 }, { v4, v5 } };
}

class Camera {
  public:
    Mat4 get_projection_matrix() const { return detail::perspective((float)window_.first / (float)window_.second, fov_, near_, far_); }
    Mat4 get_view_matrix() const { return detail::look_at(position_, TVec3{ position_.x + direction_.x, position_.y + direction_.y, position_.z + direction_.z }, TVec3{ 0, 1, 0 }); }
    TVec3 get_position() const { return position_; } TVec3 get_direction() const { return direction_; }
    float get_far_draw_distance() const { return far_; }
    void force_hard_position(TVec3 p) { position_ = p; }
  private:
    friend class CameraBuilder;
    std::pair<int, int> window_{ 1280, 720 }; TVec3 position_{}, direction_{ 0, 0, -1 }; float fov_ = 45.0f * 3.14159265358979323846f / 180.0f, near_ = 0.1f, far_ = 1000.0f;
};
class CameraBuilder {
  public:
    explicit CameraBuilder(std::pair<int, int> window_dimensions) { c_.window_ = window_dimensions; }
    CameraBuilder &with_position(TVec3 p) { c_.position_ = p; return *this; }
    CameraBuilder &with_direction(TVec3 d) { c_.direction_ = d; return *this; }
    CameraBuilder &with_fov(float radians) { c_.fov_ = radians; return *this; }
    CameraBuilder &with_near_draw_distance(float d) { c_.near_ = d; return *this; }
    CameraBuilder &with_far_draw_distance(float d) { c_.far_ = d; return *this; }
    Camera build() const { return c_; }
  private:
    Camera c_;
};

// ---- TraversalDecider implementations (culling/trait.rs:4-7), host versions of what the kernels evaluate per candidate section ----
class RenderFrustumCuller {                                 // culling/render_frustum_culler.rs:36-118
  public:
    explicit RenderFrustumCuller(const Mat4 &projection_view) { update_plane_coefficients(projection_view); }
    void update_plane_coefficients(const Mat4 &pv) {
        // rows of P*V == columns of its transpose (:59-66): Left = r3 + r0, Right = r3 - r0, Bottom = r3 + r1, Top = r3 - r1, Near = r3, Far = r3 - r2
        auto row = [&](int r, int k) { return pv[k * 4 + r]; };
        for (int k = 0; k < 4; k++) {
            planes_[0][k] = row(3, k) + row(0, k); planes_[1][k] = row(3, k) - row(0, k); planes_[2][k] = row(3, k) + row(1, k);
            planes_[3][k] = row(3, k) - row(1, k); planes_[4][k] = row(3, k);             planes_[5][k] = row(3, k) - row(2, k);
        }
        for (auto &p : planes_) { float n = std::sqrt((p[0] * p[0] + p[1] * p[1]) + p[2] * p[2]); for (float &v : p) v = v / n; }   // true division (:68-77)
    }
    bool aabb_in_view(const StaticAABB &a) const {           // :83-118: every plane must have a corner that is not strictly behind it
        const float xs[2] = { a.x_range.min, a.x_range.max }, ys[2] = { a.y_range.min, a.y_range.max }, zs[2] = { a.z_range.min, a.z_range.max };
        for (const auto &p : planes_) {
            bool any = false;
            for (int i = 0; i < 8 && !any; i++) { float d = ((p[0] * xs[i & 1] + p[1] * ys[(i >> 1) & 1]) + p[2] * zs[(i >> 2) & 1]) + p[3]; any = !(d < 0.0f); }
            if (!any) return false;
        }
        return true;
    }
    const std::array<std::array<float, 4>, 6> &planes() const { return planes_; }
  private:
    std::array<std::array<float, 4>, 6> planes_{};
};
class LogicFrustumCuller {                                  // culling/logic_frustum_culler.rs:22-46
  public:
    LogicFrustumCuller(float lookahead_distance, TVec3 camera_position) : lookahead_(lookahead_distance), cam_(camera_position) {}
    bool aabb_in_view(const StaticAABB &a) const {
        const float xs[2] = { a.x_range.min, a.x_range.max }, ys[2] = { a.y_range.min, a.y_range.max }, zs[2] = { a.z_range.min, a.z_range.max };
        for (int i = 0; i < 8; i++) {
            float dx = cam_.x - xs[i & 1], dy = cam_.y - ys[(i >> 1) & 1], dz = cam_.z - zs[(i >> 2) & 1];
            if (std::sqrt((dx * dx + dy * dy) + dz * dz) <= lookahead_) return true;
        }
        return false;
    }
  private:
    float lookahead_; TVec3 cam_;
};

// ---- change requests of user logic (objects/entity_change_request.rs) ----
class EntityChangeRequest {
  public:
    explicit EntityChangeRequest(EntityId id) : entity_id(id) {}
    EntityId entity_id;
    void add_new_change(Position p) { push(RE_C_POSITION, { p.v.x, p.v.y, p.v.z, 0 }); }
    void add_new_change(Rotation r) { push(RE_C_ROTATION, { r.axis.x, r.axis.y, r.axis.z, r.angle }); }
    void add_new_change(Scale s) { push(RE_C_SCALE, { s.v.x, s.v.y, s.v.z, 0 }); }
    void add_new_change(Velocity v) { push(RE_C_VELOCITY, { v.v.x, v.v.y, v.v.z, 0 }); }
    void add_new_change(Acceleration a) { push(RE_C_ACCELERATION, { a.v.x, a.v.y, a.v.z, 0 }); }
    void add_new_change(VelocityRotation r) { push(RE_C_ROTATION_VEL, { r.axis.x, r.axis.y, r.axis.z, r.rate }); }
    void add_new_change(AccelerationRotation r) { push(RE_C_ROTATION_ACC, { r.axis.x, r.axis.y, r.axis.z, r.rate }); }
    size_t number_changes() const { return changes.size(); }
    std::vector<re_change> changes;
  private:
    void push(uint32_t comp, std::array<float, 4> v) { re_change c{}; c.kind = RE_CHANGE_MODIFY; c.entity_id = entity_id; c.component = comp; std::memcpy(c.value, v.data(), 16); changes.push_back(c); }
};
struct EntityChangeInformation {
    enum Kind { ModifyRequest, DeleteRequest, MakeObjectStatic, WakeUpRequest } kind;
    EntityId entity_id = 0; std::vector<re_change> modify;
    static EntityChangeInformation Modify(const EntityChangeRequest &r) { return { ModifyRequest, r.entity_id, r.changes }; }
    static EntityChangeInformation Delete(EntityId id) { return { DeleteRequest, id, {} }; }
    static EntityChangeInformation MakeStatic(EntityId id) { return { MakeObjectStatic, id, {} }; }
    static EntityChangeInformation WakeUp(EntityId id) { return { WakeUpRequest, id, {} }; }
};

// ---- what a frame produces (RenderFlow: ModelRenderingInformation.instance_location + the mapped instance buffer) ----
struct InstanceRange { uint32_t begin_instance, count; };
struct FrameResult {
    uint32_t visible_sections = 0, visible_sections_vec = 0, instances = 0;
    std::vector<re_instance_range> groups;                  // (LOD-adjusted ModelId, render system, sortable) -> InstanceRange
    std::vector<EntityId> entity_ids; std::vector<float> matrices;   // host copies (filled when execute(.., copy = true))
    std::vector<re_collision> collisions;                            // (this_entity, other_entity) of every collision-logic invocation (execute(.., collide = true))
    re_tick_result tick{};
};

class Pipeline;
using AddInstanceFunction = std::function<void(Pipeline &, const std::vector<EntityId> &, StaticAABB)>;

// EntityTransformationBuilder (exports/entity_transformer.rs): records the choices, apply_choices hands the entity to the pipeline
class EntityTransformationBuilder {
  public:
    EntityTransformationBuilder(EntityId entity_id, bool is_initially_static, std::optional<FindLightType> light_type, bool can_cause_collision)
        : entity_id_(entity_id), is_static_(is_initially_static), light_(light_type), can_collide_(can_cause_collision) {}
    EntityTransformationBuilder &with_translation(Position p) { translation_ = p; return *this; }
    EntityTransformationBuilder &with_velocity(Velocity v) { velocity_ = v; return *this; }
    EntityTransformationBuilder &with_acceleration(Acceleration a) { acceleration_ = a; return *this; }
    EntityTransformationBuilder &with_rotation(Rotation r) { rotation_ = r; return *this; }
    EntityTransformationBuilder &with_rotation_velocity(VelocityRotation r) { rotation_velocity_ = r; return *this; }
    EntityTransformationBuilder &with_rotation_acceleration(AccelerationRotation r) { rotation_acceleration_ = r; return *this; }
    EntityTransformationBuilder &with_scale(Scale s) { scale_ = s; return *this; }
    inline void apply_choices(StaticAABB original_aabb, Pipeline &pipeline);
    void check_invariants() const {                              // entity_transformer.rs:77-97 (assert! -> exception)
        if (!translation_) throw Error(RE_E_ARG, "A translation is required to be provided");
        if (acceleration_ && !velocity_) throw Error(RE_E_ARG, "Providing acceleration requires providing velocity");
        if (rotation_acceleration_ && !rotation_velocity_) throw Error(RE_E_ARG, "Providing rotation acceleration requires providing rotation velocity");
        if (rotation_acceleration_ && !rotation_) throw Error(RE_E_ARG, "Providing rotation acceleration requires providing a rotation");
        if (rotation_velocity_ && !rotation_) throw Error(RE_E_ARG, "Providing rotation velocity requires providing a rotation");
    }
  private:
    EntityId entity_id_; bool is_static_; std::optional<FindLightType> light_; bool can_collide_;
    std::optional<Position> translation_; std::optional<Velocity> velocity_; std::optional<Acceleration> acceleration_;
    std::optional<Rotation> rotation_; std::optional<VelocityRotation> rotation_velocity_; std::optional<AccelerationRotation> rotation_acceleration_;
    std::optional<Scale> scale_;
};

// Pipeline (flows/pipeline.rs): owns the ECS columns the path needs and the GPU context.  Registration is collected on the host
// (create_entity / write_component / add_entity of the reference) and handed to the GPU before the next frame: the whole world before the
// first one, the instances registered since afterwards (re_add_entities).
class Pipeline {
  public:
    Pipeline(uint32_t tree_outline_length, uint32_t tree_atomic_length, int device = 0, uint32_t max_instances = 0) {
        re_config cfg{ device, tree_outline_length, tree_atomic_length, max_instances, RE_CFG_DEFAULT };
        int rc = re_create(&cfg, &ctx_);
        if (rc != RE_OK) throw Error(rc, std::string("re_create: ") + re_last_error(nullptr));
    }
    ~Pipeline() { if (ctx_) re_destroy(ctx_); }
    Pipeline(const Pipeline &) = delete; Pipeline &operator=(const Pipeline &) = delete;

    // ECS::create_entity (objects/ecs.rs:384-402): ids are dense, in creation order
    EntityId create_entity() { rows_.emplace_back(); rows_.back().id = (EntityId)rows_.size() - 1; return rows_.back().id; }
    // Pipeline::register_model_instances (flows/pipeline.rs:186-208)
    void register_model_instances(ModelId model_id, size_t number_instances_to_add, StaticAABB original_aabb, const AddInstanceFunction &add_function) {
        std::vector<EntityId> created; created.reserve(number_instances_to_add);
        for (size_t i = 0; i < number_instances_to_add; i++) { EntityId e = create_entity(); rows_[e].model = model_id; created.push_back(e); }
        add_function(*this, created, original_aabb);
        uploaded_ = false;
    }
    // Pipeline::register_user_entity + create_user_entity_instance (flows/pipeline.rs:125-174)
    EntityId register_user_entity(TVec3 camera_pos, StaticAABB original_aabb, ModelId model_id) {
        EntityId e = create_entity(); Row &r = rows_[e];
        r.model = model_id; r.original = original_aabb; r.pos = camera_pos; r.flags = RE_F_USER | RE_F_HAS_VEL | RE_F_HAS_ACC | RE_F_CAN_COLLIDE; r.placed = true;   // CanCauseCollisions + UserAlwaysCausesCollisions (pipeline.rs:135-136)
        uploaded_ = false; return e;
    }
    void write_sortable_component(EntityId e, uint32_t sortable_index) {                                                           // ecs.rs:202-205
        row(e).sortable = sortable_index;
        if (e < n_sent_) { re_change ch{}; ch.kind = RE_CHANGE_ADD_SORTABLE; ch.entity_id = e; ch.component = sortable_index; check(re_apply_changes(ctx_, &ch, 1, 0, nullptr), "re_apply_changes"); }
        else uploaded_ = false;
    }
    void write_always_execute_logic(EntityId e) { row(e).flags |= RE_F_ALWAYS_EXEC; uploaded_ = false; }
    void write_out_of_bounds_logic(EntityId e) { row(e).flags |= RE_F_OOB_LOGIC; uploaded_ = false; }                           // the entity type has OutOfBoundsLogic

    // Pipeline::execute (flows/pipeline.rs:212-276) for this path: both visibility queries + render gather, then the kinematic tick
    // collide: also run the collision phase of LogicFlow::execute (logic_flow.rs:243) between the visibility queries and the tick; the
    // caller dispatches FrameResult::collisions to the CollisionFunction of each this_entity's type
    FrameResult execute(const Camera &camera, float delta_time, bool copy = false, bool emit_duplicates = false, bool collide = false) {
        upload_if_needed();
        executed_ = true;
        re_camera cam{};
        Mat4 pv = detail::mul(camera.get_projection_matrix(), camera.get_view_matrix());
        std::memcpy(cam.projection_view, pv.data(), 64);
        TVec3 p = camera.get_position(), d = camera.get_direction();
        cam.position[0] = p.x; cam.position[1] = p.y; cam.position[2] = p.z; cam.direction[0] = d.x; cam.direction[1] = d.y; cam.direction[2] = d.z;
        cam.far_draw = camera.get_far_draw_distance();
        auto lov = create_level_of_views(camera.get_far_draw_distance());         // level_views.default (render_flow.rs:398)
        cam.n_lod = (uint32_t)lov.size(); for (size_t i = 0; i < lov.size(); i++) { cam.lod_min[i] = lov[i].min_distance; cam.lod_max[i] = lov[i].max_distance; }
        re_visible vis{}; FrameResult out;
        check(re_cull_pack(ctx_, &cam, emit_duplicates ? RE_CULL_EMIT_DUPLICATES : 0u, &vis), "re_cull_pack");
        out.visible_sections = vis.n_visible_sections; out.visible_sections_vec = vis.n_visible_vec; out.instances = vis.n_instances;
        out.groups.assign(vis.groups, vis.groups + vis.n_groups);
        if (copy && vis.n_written) {
            out.entity_ids.resize(vis.n_written); out.matrices.resize((size_t)vis.n_written * 16); uint32_t n = 0;
            check(re_copy_visible(ctx_, out.entity_ids.data(), out.matrices.data(), vis.n_written, &n), "re_copy_visible");
        }
        if (collide) {
            uint32_t n = 0; check(re_collide(ctx_, 0u, nullptr, 0u, &n), "re_collide");
            out.collisions.resize(n);
            if (n) check(re_collide(ctx_, 0u, out.collisions.data(), n, &n), "re_collide");
        }
        check(re_tick(ctx_, delta_time, 0u, &out.tick), "re_tick");
        return out;
    }
    // apply_change (helper_things/entity_change_helpers.rs:32-189) for the requests user logic returned this frame
    re_tick_result apply_change(const std::vector<EntityChangeInformation> &changes) {
        std::vector<re_change> list;
        for (const auto &c : changes) {
            if (c.kind == EntityChangeInformation::ModifyRequest) list.insert(list.end(), c.modify.begin(), c.modify.end());
            else { re_change r{}; r.entity_id = c.entity_id; r.kind = c.kind == EntityChangeInformation::DeleteRequest ? RE_CHANGE_DELETE : c.kind == EntityChangeInformation::MakeObjectStatic ? RE_CHANGE_MAKE_STATIC : RE_CHANGE_WAKE_UP; list.push_back(r); }
        }
        re_tick_result t{}; check(re_apply_changes(ctx_, list.data(), (uint32_t)list.size(), 0u, &t), "re_apply_changes");
        return t;
    }
    // ECS::get_copy::<T> (objects/ecs.rs:653-664)
    Position get_copy_position(EntityId e) { float v[3]; read(e, RE_C_POSITION, v); return Position{ { v[0], v[1], v[2] } }; }
    Rotation get_copy_rotation(EntityId e) { float v[4]; read(e, RE_C_ROTATION, v); return Rotation{ { v[0], v[1], v[2] }, v[3] }; }
    TransformationMatrix get_copy_transformation_matrix(EntityId e) { TransformationMatrix t; read(e, RE_C_TRANSFORMATION, t.m.data()); return t; }
    StaticAABB get_copy_static_aabb(EntityId e) { float v[6]; read(e, RE_C_STATIC_AABB, v); return StaticAABB{ { v[0], v[1] }, { v[2], v[3] }, { v[4], v[5] } }; }
    std::vector<EntityId> out_of_bounds_entities() { uint32_t n = 0; std::vector<EntityId> ids(4096); check(re_get_out_of_bounds(ctx_, ids.data(), (uint32_t)ids.size(), &n), "re_get_out_of_bounds"); ids.resize(std::min<uint32_t>(n, 4096)); return ids; }
    uint32_t rejected_at_registration() const { return n_rejected_; }
    // shadow_flow::find_nearby_lights (flows/shadow_flow.rs:455-513) for the camera of the frame: the light entities of one type near it, ascending ids
    std::vector<EntityId> find_nearby_lights(const Camera &camera, FindLightType light_type) {
        upload_if_needed();
        re_camera cam{}; const TVec3 cp = camera.get_position();
        cam.position[0] = cp.x; cam.position[1] = cp.y; cam.position[2] = cp.z; cam.far_draw = camera.get_far_draw_distance();      // (the light query reads the position and the far distance only)
        const uint32_t flag = light_type == FindLightType::Directional ? RE_F_LIGHT_DIRECTIONAL : light_type == FindLightType::Point ? RE_F_LIGHT_POINT : RE_F_LIGHT_SPOT;
        uint32_t n = 0; check(re_visible_lights(ctx_, &cam, flag, nullptr, 0u, &n), "re_visible_lights");
        std::vector<EntityId> ids(n);
        if (n) check(re_visible_lights(ctx_, &cam, flag, ids.data(), n, &n), "re_visible_lights");
        return ids;
    }
    // RenderFlow::register_model_with_render_system with custom_level_of_view (flows/render_flow.rs:1069-1076)
    void register_custom_level_of_view(ModelId model_id, const std::vector<float> &min_distance, const std::vector<float> &max_distance) {
        check(re_set_model_lod(ctx_, model_id.model_index, model_id.render_system_index, (uint32_t)std::min(min_distance.size(), max_distance.size()), min_distance.data(), max_distance.data()), "re_set_model_lod");
    }
    // ECS::get_indexes_for_components over the movement components (objects/ecs.rs:238-285); components: RE_C_*
    std::vector<EntityId> get_indexes_for_components(const std::vector<int> &components) {
        upload_if_needed();
        uint32_t n = 0; check(re_ecs_query(ctx_, components.data(), (uint32_t)components.size(), nullptr, 0u, &n), "re_ecs_query");
        std::vector<EntityId> ids(n);
        if (n) check(re_ecs_query(ctx_, components.data(), (uint32_t)components.size(), ids.data(), n, &n), "re_ecs_query");
        return ids;
    }
    bool has_component(EntityId e, uint32_t ecs_bit) { upload_if_needed(); uint32_t bits = 0; check(re_ecs_bitset(ctx_, e, &bits), "re_ecs_bitset"); return (bits >> ecs_bit) & 1u; }   // ecs_bit: RE_ECS_BIT_*
    re_ctx *context() { upload_if_needed(); return ctx_; }

  private:
    friend class EntityTransformationBuilder;
    struct Row {
        EntityId id = 0; ModelId model; uint32_t sortable = 0, flags = 0; StaticAABB original{};
        TVec3 pos{}, scale{ 1, 1, 1 }, vel{}, acc{}; Rotation rot{}; VelocityRotation rotvel{}; AccelerationRotation rotacc{}; bool placed = false;
    };
    Row &row(EntityId e) { if (e >= rows_.size()) throw Error(RE_E_ARG, "unknown entity"); return rows_[e]; }
    void check(int rc, const char *what) { if (rc != RE_OK) throw Error(rc, std::string(what) + ": " + re_last_error(ctx_)); }
    void read(EntityId e, int component, void *dst) { upload_if_needed(); check(re_read_component(ctx_, e, component, dst), "re_read_component"); }
    void upload_if_needed() {
        if (uploaded_) return;
        // Before the first executed frame every registration is collected and uploaded once (re_upload_entities REPLACES the world).  Afterwards the
        // instances registered since are APPENDED (re_add_entities == Pipeline::register_model_instances at any time, flows/pipeline.rs:186-208).
        const size_t first = executed_ ? n_sent_ : 0, n = rows_.size() - first;
        std::vector<uint32_t> id(n), model(n), rs(n), sortable(n), flags(n);
        std::vector<float> aabb(n * 6), pos(n * 3), rot(n * 4), scl(n * 3), vel(n * 3), acc(n * 3), rv(n * 4), ra(n * 4);
        size_t m = 0;
        for (size_t ri = first; ri < rows_.size(); ri++) {
            const Row &r = rows_[ri];
            if (!r.placed) continue;                                                   // created but never given a transformation: not in the tree
            id[m] = r.id; model[m] = r.model.model_index; rs[m] = r.model.render_system_index; sortable[m] = r.sortable; flags[m] = r.flags;
            const float a6[6] = { r.original.x_range.min, r.original.x_range.max, r.original.y_range.min, r.original.y_range.max, r.original.z_range.min, r.original.z_range.max };
            std::memcpy(&aabb[m * 6], a6, 24);
            const float p3[3] = { r.pos.x, r.pos.y, r.pos.z }, s3[3] = { r.scale.x, r.scale.y, r.scale.z }, v3[3] = { r.vel.x, r.vel.y, r.vel.z }, c3[3] = { r.acc.x, r.acc.y, r.acc.z };
            std::memcpy(&pos[m * 3], p3, 12); std::memcpy(&scl[m * 3], s3, 12); std::memcpy(&vel[m * 3], v3, 12); std::memcpy(&acc[m * 3], c3, 12);
            const float r4[4] = { r.rot.axis.x, r.rot.axis.y, r.rot.axis.z, r.rot.angle }, v4[4] = { r.rotvel.axis.x, r.rotvel.axis.y, r.rotvel.axis.z, r.rotvel.rate },
                        c4[4] = { r.rotacc.axis.x, r.rotacc.axis.y, r.rotacc.axis.z, r.rotacc.rate };
            std::memcpy(&rot[m * 4], r4, 16); std::memcpy(&rv[m * 4], v4, 16); std::memcpy(&ra[m * 4], c4, 16);
            m++;
        }
        re_entities E{}; E.n = (uint32_t)m; E.entity_id = id.data(); E.model_index = model.data(); E.render_system = rs.data(); E.sortable = sortable.data(); E.flags = flags.data();
        E.original_aabb = aabb.data(); E.position = pos.data(); E.rotation = rot.data(); E.scale = scl.data(); E.velocity = vel.data(); E.acceleration = acc.data();
        E.rotation_velocity = rv.data(); E.rotation_acceleration = ra.data();
        if (executed_) { uint32_t rej = 0; check(re_add_entities(ctx_, &E, &rej), "re_add_entities"); n_rejected_ += rej; }
        else check(re_upload_entities(ctx_, &E, &n_rejected_), "re_upload_entities");
        n_sent_ = rows_.size(); uploaded_ = true;
    }
    re_ctx *ctx_ = nullptr; std::vector<Row> rows_; bool uploaded_ = false, executed_ = false; uint32_t n_rejected_ = 0; size_t n_sent_ = 0;
};

inline void EntityTransformationBuilder::apply_choices(StaticAABB original_aabb, Pipeline &pipeline) {     // entity_transformer.rs:55-75, 99-142
    check_invariants();
    Pipeline::Row &r = pipeline.row(entity_id_);
    r.original = original_aabb; r.placed = true; r.flags &= (RE_F_ALWAYS_EXEC | RE_F_OOB_LOGIC);
    if (can_collide_) r.flags |= RE_F_CAN_COLLIDE;                  // entity_transformer.rs:66-69
    if (is_static_) r.flags |= RE_F_STATIC;
    r.pos = translation_->v;
    if (velocity_) { r.vel = velocity_->v; r.flags |= RE_F_HAS_VEL; }
    if (acceleration_) { r.acc = acceleration_->v; r.flags |= RE_F_HAS_ACC; }
    if (rotation_) { r.rot = *rotation_; r.flags |= RE_F_HAS_ROT; }
    if (rotation_velocity_) { r.rotvel = *rotation_velocity_; r.flags |= RE_F_HAS_ROTVEL; }
    if (rotation_acceleration_) { r.rotacc = *rotation_acceleration_; r.flags |= RE_F_HAS_ROTACC; }
    if (scale_) { r.scale = scale_->v; r.flags |= RE_F_HAS_SCALE; }
    if (light_) r.flags |= *light_ == FindLightType::Directional ? RE_F_LIGHT_DIRECTIONAL : *light_ == FindLightType::Point ? RE_F_LIGHT_POINT : RE_F_LIGHT_SPOT;   // the entity joins its section's light set (add_entity, bounding_box_tree_v2.rs:601-627)
    pipeline.uploaded_ = false;
}

}  // namespace render_engine
