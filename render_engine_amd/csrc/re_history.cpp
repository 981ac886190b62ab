// re_history.cpp -- the history / replay wire format of the reference: the per-frame FrameChange records that the history thread
// writes with bincode 1.3 (threads/history_thread.rs:150-205: gameplay_history.txt = bincode(ECS) | bincode(BoundingBoxTree) | bincode(FrameChange)*,
// gameplay_byte_lookup.txt = the byte length of every blob, one decimal number per line) and that GameLoadResult::load reads back
// (helper_things/game_loader.rs:32-71) for Pipeline::debug_execute (flows/pipeline.rs:279-421).  Host code only (no device work).
//
// bincode 1.3 default configuration: little endian, fixed-width integers; enum variant = u32 index; Vec<T> / String = u64 length + elements;
// tuples, structs and fixed arrays = their fields back to back; f32 = 4 bytes; unit variants = the index alone.
//   FrameChange (threads/public_common_structures.rs:7-16)             variant indices 0..6 in declaration order
//   EntityChangeInformation (objects/entity_change_request.rs:10-27)   variant indices 0..11 in declaration order
//   EntityChangeRequest { entity_id: EntityId(u32), type_id: Vec<(TypeIdentifier { t: [u64; 1] }, Vec<u8>)> } (:31-36); the Vec<u8> is the component's
//   in-memory bytes (add_new_change, :64-79): Position / Velocity / Acceleration / Scale = 3 f32, Rotation / VelocityRotation / AccelerationRotation =
//   3 f32 axis + f32 (exports/movement_components.rs:6-39)
//   SerializableCameraInfo { position: TVec3<f32>, direction: TVec3<f32> } (exports/camera_object.rs:47-53).  nalgebra 0.25 serialises a statically
//   sized matrix through ArrayStorage's serde impl as a SEQUENCE (u64 element count + elements), not as a fixed array: 8 + 12 bytes per TVec3.  That is
//   read from nalgebra's published source, not verifiable here (no Rust toolchain, crate not vendored): RE_HISTORY_VEC3_AS_ARRAY writes 12 bytes instead.
// TypeIdentifier is the raw std::any::TypeId of the reference BINARY (objects/ecs.rs:92-110): a file is only valid for the build that wrote it, so the
// host supplies the ids of the components this path knows (re_type_ids).  The first two blobs (ECS, BoundingBoxTree) are written / kept as opaque bytes:
// this library's state is the SoA columns of DESIGN.md section 3, not the reference's hash maps.
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "re_hip.h"
#include "re_guard.h"

namespace {
struct Frame { re_frame_change fc; std::vector<re_change> changes; };
void put_u32(std::vector<uint8_t> &b, uint32_t v) { for (int k = 0; k < 4; k++) b.push_back((uint8_t)(v >> (8 * k))); }
void put_u64(std::vector<uint8_t> &b, uint64_t v) { for (int k = 0; k < 8; k++) b.push_back((uint8_t)(v >> (8 * k))); }
void put_f32(std::vector<uint8_t> &b, float f) { uint32_t v; memcpy(&v, &f, 4); put_u32(b, v); }
struct Reader {
    const uint8_t *p; size_t n, o = 0; bool ok = true;
    uint32_t u32() { if (o + 4 > n) { ok = false; return 0; } uint32_t v = 0; for (int k = 0; k < 4; k++) v |= (uint32_t)p[o + k] << (8 * k); o += 4; return v; }
    uint64_t u64() { if (o + 8 > n) { ok = false; return 0; } uint64_t v = 0; for (int k = 0; k < 8; k++) v |= (uint64_t)p[o + k] << (8 * k); o += 8; return v; }
    float f32() { uint32_t v = u32(); float f; memcpy(&f, &v, 4); return f; }
};
// EntityChangeInformation variant indices (objects/entity_change_request.rs:10-27)
enum : uint32_t { ECI_ADD_ENTITY = 0, ECI_ADD_OWNED = 1, ECI_ADD_REFERENCED = 2, ECI_ADD_SORTABLE = 3, ECI_REMOVE_SORTABLE = 4, ECI_MODIFY = 5, ECI_REMOVE_COMPONENT = 6,
                  ECI_REMOVE_OWNED = 7, ECI_REMOVE_REFERENCED = 8, ECI_DELETE = 9, ECI_MAKE_STATIC = 10, ECI_WAKE_UP = 11 };
}  // namespace

struct re_history {
    re_type_ids ids{}; uint32_t flags = 0;
    std::vector<uint8_t> ecs_blob, tree_blob;
    std::vector<Frame> frames;
    std::string err;
    uint64_t type_of(uint32_t comp) const {
        switch (comp) {
            case RE_C_POSITION: return ids.position; case RE_C_ROTATION: return ids.rotation; case RE_C_SCALE: return ids.scale; case RE_C_VELOCITY: return ids.velocity;
            case RE_C_ACCELERATION: return ids.acceleration; case RE_C_ROTATION_VEL: return ids.rotation_velocity; case RE_C_ROTATION_ACC: return ids.rotation_acceleration;
            default: return 0;
        }
    }
    int comp_of(uint64_t t) const {
        if (t == ids.position) return RE_C_POSITION; if (t == ids.rotation) return RE_C_ROTATION; if (t == ids.scale) return RE_C_SCALE; if (t == ids.velocity) return RE_C_VELOCITY;
        if (t == ids.acceleration) return RE_C_ACCELERATION; if (t == ids.rotation_velocity) return RE_C_ROTATION_VEL; if (t == ids.rotation_acceleration) return RE_C_ROTATION_ACC;
        return -1;
    }
    static uint32_t comp_floats(uint32_t comp) { return (comp == RE_C_ROTATION || comp == RE_C_ROTATION_VEL || comp == RE_C_ROTATION_ACC) ? 4u : 3u; }
    void put_vec3(std::vector<uint8_t> &b, const float *v) const { if (!(flags & RE_HISTORY_VEC3_AS_ARRAY)) put_u64(b, 3); for (int k = 0; k < 3; k++) put_f32(b, v[k]); }
    bool get_vec3(Reader &r, float *v) const { if (!(flags & RE_HISTORY_VEC3_AS_ARRAY) && r.u64() != 3) return false; for (int k = 0; k < 3; k++) v[k] = r.f32(); return r.ok; }

    int encode(const Frame &f, std::vector<uint8_t> &b) {
        const re_frame_change &fc = f.fc;
        put_u32(b, fc.kind);
        switch (fc.kind) {
            case RE_FC_CAMERA_VIEW_CHANGE: put_vec3(b, fc.f); put_vec3(b, fc.f + 3); break;
            case RE_FC_CAMERA_STATIONARY: case RE_FC_END_FRAME_CHANGE: break;
            case RE_FC_DELTA_TIME: put_f32(b, fc.f[0]); break;
            case RE_FC_DRAW_DISTANCES_CHANGE: put_f32(b, fc.f[0]); put_f32(b, fc.f[1]); put_f32(b, fc.f[2]); break;
            case RE_FC_WINDOW_DIMENSIONS_CHANGE: put_u32(b, (uint32_t)fc.i[0]); put_u32(b, (uint32_t)fc.i[1]); break;
            case RE_FC_ENTITY_CHANGE:
                put_u64(b, f.changes.size());
                for (const re_change &c : f.changes) {
                    switch (c.kind) {
                        case RE_CHANGE_MODIFY: {
                            const uint64_t t = type_of(c.component);
                            if (!t) { err = "re_history: component without a TypeIdentifier"; return RE_E_ARG; }
                            put_u32(b, ECI_MODIFY); put_u32(b, c.entity_id); put_u64(b, 1); put_u64(b, t);
                            const uint32_t nf = comp_floats(c.component); put_u64(b, 4u * nf); for (uint32_t k = 0; k < nf; k++) put_f32(b, c.value[k]);
                            break;
                        }
                        case RE_CHANGE_REMOVE_COMPONENT: { const uint64_t t = type_of(c.component); if (!t) { err = "re_history: component without a TypeIdentifier"; return RE_E_ARG; }
                                                           put_u32(b, ECI_REMOVE_COMPONENT); put_u32(b, c.entity_id); put_u64(b, t); break; }
                        case RE_CHANGE_DELETE: put_u32(b, ECI_DELETE); put_u32(b, c.entity_id); break;
                        case RE_CHANGE_MAKE_STATIC: put_u32(b, ECI_MAKE_STATIC); put_u32(b, c.entity_id); break;
                        case RE_CHANGE_WAKE_UP: put_u32(b, ECI_WAKE_UP); put_u32(b, c.entity_id); break;
                        default: err = "re_history: unknown change kind"; return RE_E_ARG;
                    }
                }
                break;
            default: err = "re_history: unknown FrameChange kind"; return RE_E_ARG;
        }
        return RE_OK;
    }
    int decode(const uint8_t *p, size_t n, Frame &f) {
        Reader r{ p, n };
        f.fc = re_frame_change{}; f.changes.clear();
        f.fc.kind = r.u32();
        switch (f.fc.kind) {
            case RE_FC_CAMERA_VIEW_CHANGE: if (!get_vec3(r, f.fc.f) || !get_vec3(r, f.fc.f + 3)) { err = "re_history: malformed CameraViewChange"; return RE_E_ARG; } break;
            case RE_FC_CAMERA_STATIONARY: case RE_FC_END_FRAME_CHANGE: break;
            case RE_FC_DELTA_TIME: f.fc.f[0] = r.f32(); break;
            case RE_FC_DRAW_DISTANCES_CHANGE: f.fc.f[0] = r.f32(); f.fc.f[1] = r.f32(); f.fc.f[2] = r.f32(); break;
            case RE_FC_WINDOW_DIMENSIONS_CHANGE: f.fc.i[0] = (int32_t)r.u32(); f.fc.i[1] = (int32_t)r.u32(); break;
            case RE_FC_ENTITY_CHANGE: {
                const uint64_t cnt = r.u64();
                for (uint64_t i = 0; i < cnt && r.ok; i++) {
                    const uint32_t v = r.u32();
                    re_change c{};
                    switch (v) {
                        case ECI_MODIFY: {
                            c.entity_id = r.u32();
                            const uint64_t ncomp = r.u64();
                            for (uint64_t k = 0; k < ncomp && r.ok; k++) {            // a request of several components == its components one after another
                                const uint64_t t = r.u64(), nb = r.u64();
                                const int comp = comp_of(t);
                                if (comp < 0 || nb != 4u * comp_floats((uint32_t)comp)) {
                                    if (t == ids.has_moved || t == ids.has_rotated) { r.o += nb; continue; }   // marker components of the kinematics: maintained by re_tick itself
                                    err = "re_history: ModifyRequest of a component this path does not carry"; return RE_E_UNSUPPORTED;
                                }
                                re_change m{}; m.kind = RE_CHANGE_MODIFY; m.entity_id = c.entity_id; m.component = (uint32_t)comp;
                                for (uint32_t q = 0; q < nb / 4u; q++) m.value[q] = r.f32();
                                f.changes.push_back(m);
                            }
                            continue;
                        }
                        case ECI_REMOVE_COMPONENT: { c.kind = RE_CHANGE_REMOVE_COMPONENT; c.entity_id = r.u32(); const int comp = comp_of(r.u64());
                                                     if (comp < 0) { err = "re_history: RemoveComponent of a component this path does not carry"; return RE_E_UNSUPPORTED; }
                                                     c.component = (uint32_t)comp; break; }
                        case ECI_DELETE: c.kind = RE_CHANGE_DELETE; c.entity_id = r.u32(); break;
                        case ECI_MAKE_STATIC: c.kind = RE_CHANGE_MAKE_STATIC; c.entity_id = r.u32(); break;
                        case ECI_WAKE_UP: c.kind = RE_CHANGE_WAKE_UP; c.entity_id = r.u32(); break;
                        default: err = "re_history: EntityChangeInformation variant outside this path (AddEntity, owned / referenced entities, sortable components)"; return RE_E_UNSUPPORTED;
                    }
                    f.changes.push_back(c);
                }
                break;
            }
            default: err = "re_history: unknown FrameChange variant"; return RE_E_ARG;
        }
        if (!r.ok || r.o != n) { err = "re_history: FrameChange record has the wrong length"; return RE_E_ARG; }
        f.fc.changes = f.changes.data(); f.fc.n_changes = (uint32_t)f.changes.size();
        return RE_OK;
    }
};

static std::string g_history_error;

extern "C" int re_history_create(const re_type_ids *ids, uint32_t flags, re_history **out) try {
    if (!ids || !out) return RE_E_ARG;
    re_history *h = new re_history(); h->ids = *ids; h->flags = flags; *out = h;
    return RE_OK;
} RE_ABI_GUARD_NOCTX(g_history_error, "re_history_create")
extern "C" void re_history_destroy(re_history *h) { delete h; }
extern "C" const char *re_history_last_error(const re_history *h) { return h ? h->err.c_str() : g_history_error.c_str(); }
extern "C" int re_history_set_state(re_history *h, const void *ecs_blob, uint64_t ecs_bytes, const void *tree_blob, uint64_t tree_bytes) try {
    if (!h || (ecs_bytes && !ecs_blob) || (tree_bytes && !tree_blob)) return RE_E_ARG;
    h->ecs_blob.assign((const uint8_t *)ecs_blob, (const uint8_t *)ecs_blob + ecs_bytes); h->tree_blob.assign((const uint8_t *)tree_blob, (const uint8_t *)tree_blob + tree_bytes);
    return RE_OK;
} RE_ABI_GUARD(h, "re_history_set_state")
extern "C" int re_history_get_state(re_history *h, const void **ecs_blob, uint64_t *ecs_bytes, const void **tree_blob, uint64_t *tree_bytes) try {
    if (!h) return RE_E_ARG;
    if (ecs_blob) *ecs_blob = h->ecs_blob.data(); if (ecs_bytes) *ecs_bytes = h->ecs_blob.size();
    if (tree_blob) *tree_blob = h->tree_blob.data(); if (tree_bytes) *tree_bytes = h->tree_blob.size();
    return RE_OK;
} RE_ABI_GUARD(h, "re_history_get_state")
extern "C" int re_history_record(re_history *h, const re_frame_change *fc) try {
    if (!h || !fc || (fc->n_changes && !fc->changes)) return RE_E_ARG;
    Frame f; f.fc = *fc; f.changes.assign(fc->changes, fc->changes + fc->n_changes);
    std::vector<uint8_t> probe; int rc = h->encode(f, probe); if (rc != RE_OK) return rc;      // refuse what could not be written later
    h->frames.push_back(std::move(f));
    return RE_OK;
} RE_ABI_GUARD(h, "re_history_record")
extern "C" int re_history_count(re_history *h, uint32_t *n) { if (!h || !n) return RE_E_ARG; *n = (uint32_t)h->frames.size(); return RE_OK; }
extern "C" int re_history_get(re_history *h, uint32_t index, re_frame_change *out) try {
    if (!h || !out || index >= h->frames.size()) return RE_E_ARG;
    Frame &f = h->frames[index]; f.fc.changes = f.changes.data(); f.fc.n_changes = (uint32_t)f.changes.size();
    *out = f.fc;
    return RE_OK;
} RE_ABI_GUARD(h, "re_history_get")
extern "C" int re_history_encode(re_history *h, uint32_t index, uint8_t *dst, uint64_t capacity, uint64_t *n) try {
    if (!h || index >= h->frames.size()) return RE_E_ARG;
    std::vector<uint8_t> b; int rc = h->encode(h->frames[index], b); if (rc != RE_OK) return rc;
    if (n) *n = b.size();
    if (dst) memcpy(dst, b.data(), b.size() < capacity ? b.size() : capacity);
    return RE_OK;
} RE_ABI_GUARD(h, "re_history_encode")
extern "C" int re_history_write(re_history *h, const char *history_path, const char *lookup_path) try {
    if (!h || !history_path || !lookup_path) return RE_E_ARG;
    FILE *fh = fopen(history_path, "wb"), *fl = fopen(lookup_path, "wb");
    if (!fh || !fl) { if (fh) fclose(fh); if (fl) fclose(fl); h->err = "re_history_write: cannot open the output files"; return RE_E_ARG; }
    std::vector<uint64_t> lens;
    auto blob = [&](const std::vector<uint8_t> &b) { if (!b.empty()) fwrite(b.data(), 1, b.size(), fh); lens.push_back(b.size()); };
    blob(h->ecs_blob); blob(h->tree_blob);                                  // history_thread.rs:178-185
    int rc = RE_OK;
    for (const Frame &f : h->frames) { std::vector<uint8_t> b; rc = h->encode(f, b); if (rc != RE_OK) break; blob(b); }   // :187-198
    for (uint64_t l : lens) fprintf(fl, "%llu\n", (unsigned long long)l);  // :202-206: one length per line
    fclose(fh); fclose(fl);
    return rc;
} RE_ABI_GUARD(h, "re_history_write")
extern "C" int re_history_load(const re_type_ids *ids, uint32_t flags, const char *history_path, const char *lookup_path, re_history **out) try {
    if (!ids || !history_path || !lookup_path || !out) return RE_E_ARG;
    FILE *fh = fopen(history_path, "rb"), *fl = fopen(lookup_path, "rb");
    if (!fh || !fl) { if (fh) fclose(fh); if (fl) fclose(fl); g_history_error = "re_history_load: cannot open the input files"; return RE_E_ARG; }
    std::vector<uint8_t> all; uint8_t buf[65536]; size_t got;
    while ((got = fread(buf, 1, sizeof buf, fh)) > 0) all.insert(all.end(), buf, buf + got);
    std::vector<uint64_t> lens; unsigned long long v;
    while (fscanf(fl, "%llu", &v) == 1) lens.push_back(v);                  // game_loader.rs:41-44: the lookup file splits the history file
    fclose(fh); fclose(fl);
    re_history *h = new re_history(); h->ids = *ids; h->flags = flags;
    size_t off = 0;
    auto take = [&](uint64_t n, const uint8_t **p) { if (off + n > all.size()) return false; *p = all.data() + off; off += n; return true; };
    const uint8_t *p = nullptr;
    if (lens.size() < 2 || !take(lens[0], &p)) { g_history_error = "re_history_load: lookup file does not match the history file"; delete h; return RE_E_ARG; }
    h->ecs_blob.assign(p, p + lens[0]);
    if (!take(lens[1], &p)) { g_history_error = "re_history_load: lookup file does not match the history file"; delete h; return RE_E_ARG; }
    h->tree_blob.assign(p, p + lens[1]);
    for (size_t i = 2; i < lens.size(); i++) {
        if (!take(lens[i], &p)) { g_history_error = "re_history_load: lookup file does not match the history file"; delete h; return RE_E_ARG; }
        Frame f; int rc = h->decode(p, lens[i], f);
        if (rc != RE_OK) { g_history_error = h->err; delete h; return rc; }
        h->frames.push_back(std::move(f));
    }
    *out = h;
    return RE_OK;
} RE_ABI_GUARD_NOCTX(g_history_error, "re_history_load")
