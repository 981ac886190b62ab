// Per-frame latency of the synchronous C ABI calls (what a frame loop that draws every frame sees), without any Python in between:
// configs[1]-style lattice (one static entity per level-0 section), re_cull_pack + re_tick per frame, both synchronous.
// build: g++ -O2 -std=c++17 -I include tools/cpp/sync_latency.cpp -L render_engine_amd/lib -lrender_engine_hip -Wl,-rpath,$PWD/render_engine_amd/lib -o tools/cpp/sync_latency
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "re_hip.h"

static void mul(const float *a, const float *b, float *o) { for (int c = 0; c < 4; c++) for (int r = 0; r < 4; r++) { float s = 0; for (int k = 0; k < 4; k++) s += a[k * 4 + r] * b[c * 4 + k]; o[c * 4 + r] = s; } }

int main(int argc, char **argv) {
    const uint32_t axis = argc > 1 ? (uint32_t)atoi(argv[1]) : 216u, frames = argc > 2 ? (uint32_t)atoi(argv[2]) : 400u, first = (256 - axis) / 2;
    const size_t n = (size_t)axis * axis * axis;
    std::vector<uint32_t> id(n), model(n), flags(n, RE_F_STATIC); std::vector<float> box(n * 6), pos(n * 3);
    for (size_t i = 0; i < n; i++) {
        const uint32_t cx = (uint32_t)(i / ((size_t)axis * axis)) + first, cz = (uint32_t)((i / axis) % axis) + first, cy = (uint32_t)(i % axis) + first;
        id[i] = (uint32_t)i; model[i] = (uint32_t)(i % 8);
        const float h = 1.0f; float *b = &box[i * 6]; b[0] = -h; b[1] = h; b[2] = -h; b[3] = h; b[4] = -h; b[5] = h;
        pos[i * 3 + 0] = 64.0f * cx + 32.0f; pos[i * 3 + 1] = 64.0f * cy + 32.0f; pos[i * 3 + 2] = 64.0f * cz + 32.0f;
    }
    re_config cfg{}; cfg.device = 0; cfg.outline_length = 16384; cfg.atomic_length = 64; cfg.max_instances = 1u << 16;
    re_ctx *ctx = nullptr;
    if (re_create(&cfg, &ctx) != RE_OK) { std::fprintf(stderr, "re_create: %s\n", re_last_error(nullptr)); return 1; }
    re_entities E{}; E.n = (uint32_t)n; E.entity_id = id.data(); E.model_index = model.data(); E.flags = flags.data(); E.original_aabb = box.data(); E.position = pos.data();
    uint32_t rej = 0;
    if (re_upload_entities(ctx, &E, &rej) != RE_OK) { std::fprintf(stderr, "upload: %s\n", re_last_error(ctx)); return 1; }
    // camera at the centre looking down -z, far 1000, 45 degrees, 1280x720 (main.rs:23-33)
    const float c = (first + axis / 2.0f) * 64.0f, fovy = 45.0f * 3.14159265358979f / 180.0f, aspect = 1280.0f / 720.0f, zn = 0.1f, zf = 1000.0f, t = 1.0f / std::tan(fovy / 2.0f);
    float P[16] = { t / aspect, 0, 0, 0, 0, t, 0, 0, 0, 0, (zf + zn) / (zn - zf), -1, 0, 0, 2 * zf * zn / (zn - zf), 0 };
    float V[16] = { 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, -c, -c, -c, 1 };
    re_camera cam{}; mul(P, V, cam.projection_view);
    cam.position[0] = cam.position[1] = cam.position[2] = c; cam.direction[2] = -1.0f; cam.far_draw = 1000.0f;
    const float lo[5] = { 0, 100, 250, 500, 750 }, hi[5] = { 100, 250, 500, 750, 1000 };        // five equal-ish bands are enough for a latency run
    cam.n_lod = 5; for (int i = 0; i < 5; i++) { cam.lod_min[i] = lo[i]; cam.lod_max[i] = hi[i]; }
    re_visible vis{}; re_tick_result tr{};
    std::vector<double> us;
    for (uint32_t f = 0; f < frames + 20; f++) {
        auto t0 = std::chrono::steady_clock::now();
        if (re_cull_pack(ctx, &cam, 0, &vis) != RE_OK || re_tick(ctx, 0.016f, 0, &tr) != RE_OK) { std::fprintf(stderr, "frame: %s\n", re_last_error(ctx)); return 1; }
        auto t1 = std::chrono::steady_clock::now();
        if (f >= 20) us.push_back(std::chrono::duration<double, std::micro>(t1 - t0).count());
    }
    std::sort(us.begin(), us.end());
    std::printf("entities %zu visible sections %u instances %u | synchronous re_cull_pack + re_tick per frame: median %.1f us, p10 %.1f, p90 %.1f\n",
                n, vis.n_visible_sections, vis.n_instances, us[us.size() / 2], us[us.size() / 10], us[us.size() * 9 / 10]);
    // host cost of issuing a frame (asynchronous calls return once the launches are enqueued) and the time until its result is there
    std::vector<double> issue, wait;
    for (uint32_t f = 0; f < frames; f++) {
        auto t0 = std::chrono::steady_clock::now();
        if (re_cull_pack(ctx, &cam, RE_CULL_ASYNC, nullptr) != RE_OK) { std::fprintf(stderr, "frame: %s\n", re_last_error(ctx)); return 1; }
        auto t1 = std::chrono::steady_clock::now();
        if (re_wait(ctx, &vis, &tr) != RE_OK) { std::fprintf(stderr, "wait: %s\n", re_last_error(ctx)); return 1; }
        auto t2 = std::chrono::steady_clock::now();
        issue.push_back(std::chrono::duration<double, std::micro>(t1 - t0).count()); wait.push_back(std::chrono::duration<double, std::micro>(t2 - t1).count());
    }
    std::sort(issue.begin(), issue.end()); std::sort(wait.begin(), wait.end());
    std::printf("asynchronous re_cull_pack returns after %.1f us (host side: frame parameters, candidate spans, two launches); re_wait then takes %.1f us\n", issue[issue.size() / 2], wait[wait.size() / 2]);
    re_destroy(ctx);
    return 0;
}
