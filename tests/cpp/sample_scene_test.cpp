// C++ host-mirror test: the reference's sample scene built through render_engine::Pipeline / EntityTransformationBuilder /
// CameraBuilder (include/render_engine_hip.hpp), run frame by frame on the GPU and compared with the CPU oracle
// (oracle/re_oracle.h) -- ids per group, matrices, visible-section counts, tick counts: bit-exact.
// usage: sample_scene_test <scene.txt> [--host-only]
//   scene.txt: one entity per line "model_index sortable kind x y z sx  ax ay az angle  rax ray raz rate  aabb*6"
//   kind: 0 = user entity, 1 = rotating body (Rotation + VelocityRotation + Scale), 2 = scaled body without rotation
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>

#include "render_engine_hip.hpp"
extern "C" {
#include "re_oracle.h"
}

using namespace render_engine;

struct Ent { uint32_t model, sortable, kind; float p[3], s, rot[4], rv[4], box[6]; };

static int fail(const char *what) { std::fprintf(stderr, "FAIL: %s\n", what); return 1; }

int main(int argc, char **argv) {
    if (argc < 2) return fail("usage: sample_scene_test scene.txt [--host-only]");
    const bool host_only = argc > 2 && std::string(argv[2]) == "--host-only";
    std::vector<Ent> ents;
    if (FILE *f = std::fopen(argv[1], "r")) {
        Ent e;
        while (std::fscanf(f, "%u %u %u %f %f %f %f %f %f %f %f %f %f %f %f %f %f %f %f %f %f", &e.model, &e.sortable, &e.kind, &e.p[0], &e.p[1], &e.p[2], &e.s, &e.rot[0], &e.rot[1], &e.rot[2],
                           &e.rot[3], &e.rv[0], &e.rv[1], &e.rv[2], &e.rv[3], &e.box[0], &e.box[1], &e.box[2], &e.box[3], &e.box[4], &e.box[5]) == 21) ents.push_back(e);
        std::fclose(f);
    } else return fail("cannot open scene file");
    if (ents.empty()) return fail("empty scene");

    // ---- camera: main.rs:24-33 ----
    Camera camera = CameraBuilder({ 1280, 720 }).with_position(vec3(1000.0f, 1000.0f, 1150.0f)).with_direction(vec3(0.0f, 0.0f, -1.0f)).with_far_draw_distance(1000.0f).build();
    const Mat4 pv = detail::mul(camera.get_projection_matrix(), camera.get_view_matrix());
    // host pieces against the oracle: projection * view, frustum planes, both cullers on a few boxes
    {
        float P[16], V[16], PV[16], planes[24];
        const float eye[3] = { 1000.0f, 1000.0f, 1150.0f }, tgt[3] = { 1000.0f, 1000.0f, 1149.0f }, up[3] = { 0, 1, 0 };
        ro_perspective(1280.0f / 720.0f, 45.0f * 3.14159265358979323846f / 180.0f, 0.1f, 1000.0f, P); ro_look_at(eye, tgt, up, V); ro_mat4_mul(P, V, PV);
        if (std::memcmp(PV, pv.data(), 64) != 0) return fail("projection*view differs from the oracle");
        ro_make_planes(PV, planes);
        RenderFrustumCuller rc(pv); LogicFrustumCuller lc(64.0f, camera.get_position());
        if (std::memcmp(planes, rc.planes().data(), 96) != 0) return fail("frustum planes differ from the oracle");
        for (int i = 0; i < 200; i++) {
            float cx = 600.0f + 9.0f * i, cy = 900.0f + 1.5f * i, cz = 1300.0f - 7.0f * i, h = 3.0f + 0.7f * i;
            StaticAABB a{ { cx - h, cx + h }, { cy - h, cy + h }, { cz - h, cz + h } }; ro_aabb b{ cx - h, cx + h, cy - h, cy + h, cz - h, cz + h };
            if (rc.aabb_in_view(a) != (ro_frustum_aabb_visible(planes, b) != 0)) return fail("RenderFrustumCuller::aabb_in_view differs");
            const float cp[3] = { 1000.0f, 1000.0f, 1150.0f };
            if (lc.aabb_in_view(a) != (ro_logic_aabb_in_view(64.0f, cp, b) != 0)) return fail("LogicFrustumCuller::aabb_in_view differs");
        }
    }
    // EntityTransformationBuilder::check_invariants
    {
        bool threw = false;
        try { EntityTransformationBuilder(0, false, std::nullopt, false).with_rotation_velocity(VelocityRotation::new_(vec3(0, 1, 0), 1.0f)).check_invariants(); }
        catch (const Error &) { threw = true; }
        if (!threw) return fail("check_invariants did not reject a builder without translation");
    }
    if (host_only) { std::printf("OK host-only checks\n"); return 0; }

    // ---- registration through the mirror, in the reference's order: the user entity is created by ECS::new, then the instances ----
    Pipeline pipeline(16384, 64);
    ro_world *w = ro_world_new(16384, 64);
    std::vector<ro_entity_desc> descs;
    for (size_t i = 0; i < ents.size(); i++) {
        const Ent &e = ents[i];
        StaticAABB box{ { e.box[0], e.box[1] }, { e.box[2], e.box[3] }, { e.box[4], e.box[5] } };
        ro_entity_desc d{}; d.id = (uint32_t)i; d.model_index = e.model; d.sortable = e.sortable; d.original = ro_aabb{ e.box[0], e.box[1], e.box[2], e.box[3], e.box[4], e.box[5] };
        d.pos[0] = e.p[0]; d.pos[1] = e.p[1]; d.pos[2] = e.p[2]; d.scale[0] = d.scale[1] = d.scale[2] = 1.0f; d.rot_axis[0] = 1.0f; d.rotvel_axis[0] = 1.0f; d.rotacc_axis[0] = 1.0f;
        if (e.kind == 0) {
            EntityId id = pipeline.register_user_entity(vec3(e.p[0], e.p[1], e.p[2]), box, ModelId{ e.model, 0 });
            if (id != i) return fail("user entity id");
            d.flags = RO_F_USER | RO_F_HAS_VEL | RO_F_HAS_ACC | RO_F_CAN_COLLIDE;
        } else {
            pipeline.register_model_instances(ModelId{ e.model, 0 }, 1, box, [&](Pipeline &p, const std::vector<EntityId> &created, StaticAABB aabb) {
                EntityTransformationBuilder b(created[0], false, std::nullopt, /*can_cause_collision=*/e.kind == 1);   // the asteroids collide (asteroid.rs)
                b.with_translation(Position::new_(vec3(e.p[0], e.p[1], e.p[2]))).with_scale(Scale::new_(vec3(e.s, e.s, e.s)));
                if (e.kind == 1) b.with_rotation(Rotation::new_(vec3(e.rot[0], e.rot[1], e.rot[2]), e.rot[3])).with_rotation_velocity(VelocityRotation::new_(vec3(e.rv[0], e.rv[1], e.rv[2]), e.rv[3]));
                b.apply_choices(aabb, p);
                if (e.sortable) p.write_sortable_component(created[0], e.sortable);
            });
            d.flags = RO_F_HAS_SCALE | (e.kind == 1 ? (RO_F_HAS_ROT | RO_F_HAS_ROTVEL | RO_F_CAN_COLLIDE) : 0u);
            d.scale[0] = d.scale[1] = d.scale[2] = e.s;
            if (e.kind == 1) { d.rot_axis[0] = e.rot[0]; d.rot_axis[1] = e.rot[1]; d.rot_axis[2] = e.rot[2]; d.rot_angle = e.rot[3]; d.rotvel_axis[0] = e.rv[0]; d.rotvel_axis[1] = e.rv[1]; d.rotvel_axis[2] = e.rv[2]; d.rotvel = e.rv[3]; }
        }
        descs.push_back(d);
    }
    if (ro_register_entities(w, (uint32_t)descs.size(), descs.data()) != 0) return fail("oracle rejected entities");

    ro_camera oc{}; std::memcpy(oc.pv, pv.data(), 64);
    oc.pos[0] = 1000.0f; oc.pos[1] = 1000.0f; oc.pos[2] = 1150.0f; oc.dir[2] = -1.0f; oc.far_draw = 1000.0f;
    auto lov = create_level_of_views(1000.0f); oc.n_lod = (uint32_t)lov.size();
    for (size_t i = 0; i < lov.size(); i++) { oc.lod_min[i] = lov[i].min_distance; oc.lod_max[i] = lov[i].max_distance; }

    const uint32_t cap = 4096;
    std::vector<uint32_t> oids(cap); std::vector<float> omats((size_t)cap * 16); std::vector<ro_group> ogroups(256); std::vector<uint64_t> okeys(cap);
    size_t n_collisions = 0;
    for (int frame = 0; frame < 8; frame++) {
        FrameResult fr = pipeline.execute(camera, 1.0f / 60.0f, /*copy=*/true, /*emit_duplicates=*/false, /*collide=*/true);
        uint32_t nvec = ro_frame_cull(w, &oc, cap, okeys.data());
        {   // the collision phase, between the visibility queries and the tick (logic_flow.rs:230-244)
            std::vector<uint32_t> op((size_t)cap * 2);
            uint32_t nc = ro_frame_collide(w, &oc, cap, op.data());
            if (nc != fr.collisions.size()) return fail("collision count");
            std::vector<std::pair<uint32_t, uint32_t>> a, b;
            for (uint32_t i = 0; i < nc; i++) { a.push_back({ op[2 * i], op[2 * i + 1] }); b.push_back({ fr.collisions[i].this_entity, fr.collisions[i].other_entity }); }
            std::sort(a.begin(), a.end()); std::sort(b.begin(), b.end());
            if (a != b) return fail("collision pairs differ");
            n_collisions += nc;
        }
        uint32_t ng = 0, total = ro_frame_render(w, &oc, 0, cap, oids.data(), omats.data(), (uint32_t)ogroups.size(), ogroups.data(), &ng);
        uint32_t noob = 0, ochanged = ro_frame_tick(w, &oc, 1.0f / 60.0f, 0, nullptr, &noob);
        if (fr.visible_sections_vec != nvec) return fail("visible_sections_vec length");
        if (fr.instances != total || fr.groups.size() != ng) return fail("instance / group count");
        if (fr.tick.n_changed != ochanged || fr.tick.n_out_of_bounds != noob) return fail("tick counts");
        // groups as {(model, render system, sortable) -> sorted ids}; matrices per id
        std::map<std::array<uint32_t, 3>, std::vector<uint32_t>> gg, og; std::map<uint32_t, const float *> gm, om;
        for (const auto &g : fr.groups) { auto &v = gg[{ g.model_index, g.render_system, g.sortable }]; for (uint32_t k = 0; k < g.count; k++) { v.push_back(fr.entity_ids[g.begin_instance + k]); gm[fr.entity_ids[g.begin_instance + k]] = &fr.matrices[(size_t)(g.begin_instance + k) * 16]; } std::sort(v.begin(), v.end()); }
        for (uint32_t i = 0; i < ng; i++) { auto &v = og[{ ogroups[i].model_index, ogroups[i].render_system, ogroups[i].sortable }]; for (uint32_t k = 0; k < ogroups[i].count; k++) { v.push_back(oids[ogroups[i].begin + k]); om[oids[ogroups[i].begin + k]] = &omats[(size_t)(ogroups[i].begin + k) * 16]; } std::sort(v.begin(), v.end()); }
        if (gg != og) return fail("instance groups differ");
        for (auto &kv : om) if (std::memcmp(kv.second, gm[kv.first], 64) != 0) return fail("instance matrix differs");
    }
    // a change request of user logic through the mirror: move the wormhole, spin up the mine producer (entity ids from the scene order)
    {
        const EntityId wormhole = (EntityId)ents.size() - 2, mine = (EntityId)ents.size() - 1;
        EntityChangeRequest a(wormhole); a.add_new_change(Position::new_(vec3(1010.0f, 1003.0f, 990.0f)));
        EntityChangeRequest b(mine); b.add_new_change(VelocityRotation::new_(vec3(1.0f, 0.0f, 0.0f), 1.5f)); b.add_new_change(Scale::new_(vec3(6.0f, 6.0f, 6.0f)));
        re_tick_result t = pipeline.apply_change({ EntityChangeInformation::Modify(a), EntityChangeInformation::Modify(b) });
        std::vector<ro_change> oc2;
        for (const auto *rq : { &a, &b }) for (const re_change &c : rq->changes) { ro_change o{}; o.kind = c.kind; o.entity_id = c.entity_id; o.component = c.component; std::memcpy(o.value, c.value, 16); oc2.push_back(o); }
        uint32_t noob = 0, napplied = ro_apply_changes(w, oc2.data(), (uint32_t)oc2.size(), 1, 0, nullptr, &noob);
        if (t.n_changed != napplied) return fail("apply_change count");
        FrameResult fr = pipeline.execute(camera, 1.0f / 60.0f, true);
        ro_frame_cull(w, &oc, cap, okeys.data());
        uint32_t ng = 0, total = ro_frame_render(w, &oc, 0, cap, oids.data(), omats.data(), (uint32_t)ogroups.size(), ogroups.data(), &ng);
        if (fr.instances != total) return fail("instances after apply_change");
        TransformationMatrix tm = pipeline.get_copy_transformation_matrix(wormhole);
        for (uint32_t i = 0; i < total; i++) if (oids[i] == wormhole && std::memcmp(&omats[(size_t)i * 16], tm.m.data(), 64) != 0) return fail("wormhole matrix after apply_change");
    }
    // an instance registered AFTER frames have run (Pipeline::register_model_instances at any time, flows/pipeline.rs:186-208): appended, not a new world
    {
        size_t src = 0; for (size_t i = 0; i < ents.size(); i++) if (ents[i].kind == 1) { src = i; break; }          // a copy of the first asteroid, elsewhere
        const Ent e = ents[src];
        StaticAABB box{ { e.box[0], e.box[1] }, { e.box[2], e.box[3] }, { e.box[4], e.box[5] } };
        const float px = 1004.0f, py = 1001.0f, pz = 1040.0f;
        EntityId created_id = 0;
        pipeline.register_model_instances(ModelId{ e.model, 0 }, 1, box, [&](Pipeline &p, const std::vector<EntityId> &created, StaticAABB aabb) {
            created_id = created[0];
            EntityTransformationBuilder b(created[0], false, std::nullopt, true);
            b.with_translation(Position::new_(vec3(px, py, pz))).with_scale(Scale::new_(vec3(e.s, e.s, e.s)))
             .with_rotation(Rotation::new_(vec3(e.rot[0], e.rot[1], e.rot[2]), e.rot[3])).with_rotation_velocity(VelocityRotation::new_(vec3(e.rv[0], e.rv[1], e.rv[2]), e.rv[3]));
            b.apply_choices(aabb, p);
        });
        ro_entity_desc d{}; d.id = created_id; d.model_index = e.model; d.original = ro_aabb{ e.box[0], e.box[1], e.box[2], e.box[3], e.box[4], e.box[5] };
        d.pos[0] = px; d.pos[1] = py; d.pos[2] = pz; d.scale[0] = d.scale[1] = d.scale[2] = e.s; d.rotacc_axis[0] = 1.0f;
        d.flags = RO_F_HAS_SCALE | RO_F_HAS_ROT | RO_F_HAS_ROTVEL | RO_F_CAN_COLLIDE;
        d.rot_axis[0] = e.rot[0]; d.rot_axis[1] = e.rot[1]; d.rot_axis[2] = e.rot[2]; d.rot_angle = e.rot[3]; d.rotvel_axis[0] = e.rv[0]; d.rotvel_axis[1] = e.rv[1]; d.rotvel_axis[2] = e.rv[2]; d.rotvel = e.rv[3];
        if (ro_register_entities(w, 1, &d) != 0) return fail("oracle rejected the late instance");
        for (int frame = 0; frame < 2; frame++) {
            FrameResult fr = pipeline.execute(camera, 1.0f / 60.0f, true);
            ro_frame_cull(w, &oc, cap, okeys.data());
            uint32_t ng = 0, total = ro_frame_render(w, &oc, 0, cap, oids.data(), omats.data(), (uint32_t)ogroups.size(), ogroups.data(), &ng);
            uint32_t noob = 0, ochanged = ro_frame_tick(w, &oc, 1.0f / 60.0f, 0, nullptr, &noob);
            if (fr.instances != total || fr.tick.n_changed != ochanged) return fail("frame after a late registration");
            bool seen = false;
            for (uint32_t i = 0; i < total; i++) if (oids[i] == created_id) { seen = true; TransformationMatrix tm = pipeline.get_copy_transformation_matrix(created_id); (void)tm; }
            if (!seen) return fail("the late instance is not drawn");
        }
    }
    ro_world_free(w);
    std::printf("OK %zu entities, 11 frames bit-exact, %zu collision invocations, 1 instance registered after the first frame\n", ents.size(), n_collisions);
    return 0;
}
