/* "Optimised CPU" row of the measurement: the same frame as the reference's cull + render gather, but with the
 * data structures a CPU programmer would pick when free of the reference's hash maps -- sorted section keys (binary search per
 * candidate row instead of one hash probe per candidate), SoA section and entity columns, OpenMP over rows / visible sections,
 * counting sort into the instance buffer.  Test infrastructure like the rest of oracle/: bench.py reports it next to the port
 * so that the GPU speed-up is not flattered by the reference's overheads; tests check it against the oracle.
 * Scope: worlds of static entities in unique sections (BASELINE configs[1]); arithmetic identical to re_oracle.c (same helper
 * functions), so the visible set and the packed instances are bit-equal to the oracle's. */
#include "re_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef struct {
    uint32_t atomic, outline, nthreads;
    uint32_t nsec, nent, nclass;
    uint64_t *key;            /* sorted */
    ro_aabb *tight;
    uint32_t *begin;          /* nsec + 1: entity range of the section */
    uint8_t *cached;          /* static cache entry non-empty (decided at the first frame, render_flow.rs:549-594) */
    uint32_t *ent_id, *ent_class; float *ent_mat;     /* section-major */
    uint32_t *class_model, *class_rs, *class_sort;
    int frozen;
    /* per-frame scratch */
    uint32_t *vis, vis_cap; uint8_t *vis_mult;
    uint32_t *hist;           /* nthreads x nslots */
} soa_world;

static inline uint32_t f2u32s(float f) { if (!(f > 0.0f)) return 0u; if (f >= 4294967296.0f) return 0xFFFFFFFFu; return (uint32_t)f; }
static inline float fmaxr(float a, float b) { return fmaxf(a, b); }

typedef struct { uint64_t key; uint32_t ent; } keyent;
static int cmp_keyent(const void *a, const void *b) {
    const keyent *x = (const keyent *)a, *y = (const keyent *)b;
    if (x->key != y->key) return x->key < y->key ? -1 : 1;
    return x->ent < y->ent ? -1 : x->ent > y->ent;
}

void *soa_build(uint32_t n, const uint64_t *ent_key, const ro_aabb *ent_aabb, const uint32_t *id, const uint32_t *model, const uint32_t *rs,
                const uint32_t *sortable, const float *mats, uint32_t outline, uint32_t atomic, int nthreads) {
    soa_world *w = (soa_world *)calloc(1, sizeof *w);
    w->atomic = atomic; w->outline = outline; w->nthreads = nthreads < 1 ? 1u : (uint32_t)nthreads; w->nent = n;
    keyent *ke = (keyent *)malloc(sizeof(keyent) * ((size_t)n + 1));
    for (uint32_t i = 0; i < n; i++) { ke[i].key = ent_key[i]; ke[i].ent = i; }
    qsort(ke, n, sizeof(keyent), cmp_keyent);                       /* entities of one section in ascending upload (= id) order */
    uint32_t nsec = 0;
    for (uint32_t i = 0; i < n; i++) if (i == 0 || ke[i].key != ke[i - 1].key) nsec++;
    w->nsec = nsec;
    w->key = (uint64_t *)malloc(sizeof(uint64_t) * ((size_t)nsec + 1)); w->tight = (ro_aabb *)malloc(sizeof(ro_aabb) * ((size_t)nsec + 1));
    w->begin = (uint32_t *)malloc(sizeof(uint32_t) * ((size_t)nsec + 2)); w->cached = (uint8_t *)calloc((size_t)nsec + 1, 1);
    w->ent_id = (uint32_t *)malloc(sizeof(uint32_t) * ((size_t)n + 1)); w->ent_class = (uint32_t *)malloc(sizeof(uint32_t) * ((size_t)n + 1));
    w->ent_mat = (float *)malloc(sizeof(float) * 16 * ((size_t)n + 1));
    /* dense (model, render system, sortable) classes */
    uint32_t ccap = 64; w->class_model = (uint32_t *)malloc(4 * ccap); w->class_rs = (uint32_t *)malloc(4 * ccap); w->class_sort = (uint32_t *)malloc(4 * ccap);
    uint32_t s = 0;
    for (uint32_t i = 0; i < n; i++) {
        const uint32_t e = ke[i].ent;
        if (i == 0 || ke[i].key != ke[i - 1].key) { w->key[s] = ke[i].key; w->begin[s] = i; w->tight[s] = ent_aabb[e]; s++; }
        else w->tight[s - 1] = ro_combine_aabb(w->tight[s - 1], ent_aabb[e]);          /* end_of_changes fold (bounding_box_tree_v2.rs:1084-1100) */
        w->ent_id[i] = id[e]; memcpy(w->ent_mat + (size_t)i * 16, mats + (size_t)e * 16, 64);
        uint32_t c = 0;
        for (; c < w->nclass; c++) if (w->class_model[c] == model[e] && w->class_rs[c] == rs[e] && w->class_sort[c] == sortable[e]) break;
        if (c == w->nclass) {
            if (w->nclass == ccap) { ccap *= 2; w->class_model = (uint32_t *)realloc(w->class_model, 4 * ccap); w->class_rs = (uint32_t *)realloc(w->class_rs, 4 * ccap); w->class_sort = (uint32_t *)realloc(w->class_sort, 4 * ccap); }
            w->class_model[c] = model[e]; w->class_rs[c] = rs[e]; w->class_sort[c] = sortable[e]; w->nclass++;
        }
        w->ent_class[i] = c;
    }
    w->begin[nsec] = n;
    free(ke);
    w->vis_cap = 1u << 16; w->vis = (uint32_t *)malloc(4 * (size_t)w->vis_cap); w->vis_mult = (uint8_t *)malloc(w->vis_cap);
    w->hist = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)w->nthreads * (w->nclass * 8u + 1u));
    return w;
}
void soa_free(void *p) {
    soa_world *w = (soa_world *)p; if (!w) return;
    free(w->key); free(w->tight); free(w->begin); free(w->cached); free(w->ent_id); free(w->ent_class); free(w->ent_mat);
    free(w->class_model); free(w->class_rs); free(w->class_sort); free(w->vis); free(w->vis_mult); free(w->hist); free(w);
}

static uint32_t lower_bound_key(const uint64_t *k, uint32_t n, uint64_t x) {
    uint32_t lo = 0, hi = n;
    while (lo < hi) { uint32_t mid = lo + ((hi - lo) >> 1); if (k[mid] < x) lo = mid + 1; else hi = mid; }
    return lo;
}

typedef struct { uint32_t *v; uint32_t n, cap; } hitvec;
static void hit_push(hitvec *h, uint32_t x) { if (h->n == h->cap) { h->cap = h->cap ? h->cap * 2 : 1024; h->v = (uint32_t *)realloc(h->v, 4 * (size_t)h->cap); } h->v[h->n++] = x; }
/* one visibility query (visible_world_flow.rs:40-146): candidate rows of every level in parallel, one binary search per (x, z) row
 * instead of one hash probe per candidate id; hits = indices of the sections that exist and pass the predicate */
static void query(const soa_world *w, int which, ro_aabb box, const float planes[24], float lookahead, const float cam[3], hitvec *hits) {
    const uint32_t maxl = ro_max_level(w->outline, w->atomic);
    const float wsl = (float)w->atomic;
    for (uint32_t level = 0; level < maxl; level++) {
        const float ll = wsl * ldexpf(1.0f, (int)level);
        const uint32_t nx = f2u32s(ceilf((box.xmax - box.xmin) / ll)), ny = f2u32s(ceilf((box.ymax - box.ymin) / ll)), nz = f2u32s(ceilf((box.zmax - box.zmin) / ll));
        const uint32_t bx = f2u32s(box.xmin / ll), by = f2u32s(box.ymin / ll), bz = f2u32s(box.zmin / ll);
        if (!nx || !ny || !nz) continue;
        const long rows = (long)nx * (long)nz;
#ifdef _OPENMP
#pragma omp parallel num_threads(w->nthreads) if (w->nthreads > 1 && rows >= 64)
#endif
        {
            hitvec local = { 0 };
#ifdef _OPENMP
#pragma omp for schedule(static) nowait
#endif
            for (long r = 0; r < rows; r++) {
                const uint32_t x = bx + (uint32_t)(r / nz), z = bz + (uint32_t)(r % nz);
                if (x > 0xFFFFu || z > 0xFFFFu || by > 0xFFFFu) continue;
                const uint32_t y1 = by + ny - 1u > 0xFFFFu ? 0xFFFFu : by + ny - 1u;
                const uint64_t k0 = ro_pack_key(level, x, z, by), k1 = ro_pack_key(level, x, z, y1);
                for (uint32_t i = lower_bound_key(w->key, w->nsec, k0); i < w->nsec && w->key[i] <= k1; i++) {
                    const uint32_t y = (uint32_t)(w->key[i] & 0xFFFFu);
                    const float fx = (float)x * ll, fy = (float)y * ll, fz = (float)z * ll;
                    const ro_aabb g = { fx, fx + ll, fy, fy + ll, fz, fz + ll };
                    if (which ? ro_frustum_aabb_visible(planes, g) : ro_logic_aabb_in_view(lookahead, cam, g)) hit_push(&local, i);
                }
            }
#ifdef _OPENMP
#pragma omp critical
#endif
            { for (uint32_t k = 0; k < local.n; k++) hit_push(hits, local.v[k]); }
            free(local.v);
        }
    }
}

/* one frame: both visibility queries and the render gather of static data (pipeline.rs:216-229, render_flow.rs:401-410).
 * Returns the instance count; ids / matrices grouped like ro_frame_render (groups in ascending slot order). */
uint32_t soa_frame(void *p, const ro_camera *cam, uint32_t cap, uint32_t *out_ids, float *out_mats, uint32_t gcap, ro_group *groups, uint32_t *n_groups,
                   uint32_t *n_vis_vec, uint8_t *mark /* nsec bytes of scratch, zero on entry and on return */) {
    soa_world *w = (soa_world *)p;
    float planes[24]; ro_make_planes(cam->pv, planes);
    const float wsl = (float)w->atomic, draw = wsl * 2.0f, half = cam->far_draw / 2.0f;
    ro_aabb lb = { fmaxr(cam->pos[0] - draw, 0.0f), cam->pos[0] + draw, fmaxr(cam->pos[1] - draw, 0.0f), cam->pos[1] + draw, fmaxr(cam->pos[2] - draw, 0.0f), cam->pos[2] + draw };
    const float cx = cam->dir[0] * half + cam->pos[0], cy = cam->dir[1] * half + cam->pos[1], cz = cam->dir[2] * half + cam->pos[2];
    ro_aabb rb = { fmaxr(cx - half, 0.0f), cx + half, fmaxr(cy - half, 0.0f), cy + half, fmaxr(cz - half, 0.0f), cz + half };
    hitvec hits = { 0 };
    query(w, 0, lb, planes, wsl, cam->pos, &hits);
    query(w, 1, rb, planes, wsl, cam->pos, &hits);
    /* the static cache freezes at the first frame for every section (all of them are in changed_static_unique after registration) */
    if (!w->frozen) {
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(w->nthreads) if (w->nthreads > 1)
#endif
        for (long i = 0; i < (long)w->nsec; i++) w->cached[i] = ro_distance_to_aabb(w->tight[i], cam->pos) < cam->far_draw;
        w->frozen = 1;
    }
    /* visible_sections_map (set) + the multiplicity each section has in visible_sections_vec */
    uint32_t nv = 0, vec = hits.n;
    for (uint32_t k = 0; k < hits.n; k++) {
        const uint32_t i = hits.v[k];
        if (!mark[i]) {
            if (nv == w->vis_cap) { w->vis_cap *= 2; w->vis = (uint32_t *)realloc(w->vis, 4 * (size_t)w->vis_cap); w->vis_mult = (uint8_t *)realloc(w->vis_mult, w->vis_cap); }
            w->vis[nv++] = i;
        }
        mark[i]++;
    }
    for (uint32_t v = 0; v < nv; v++) { w->vis_mult[v] = mark[w->vis[v]]; mark[w->vis[v]] = 0; }
    free(hits.v);
    if (n_vis_vec) *n_vis_vec = vec;
    /* extract_static_data (render_flow.rs:458-542): count per (class, LOD) slot, prefix, scatter */
    const uint32_t nslots = w->nclass * 8u;
    uint32_t *tot = w->hist;
    memset(tot, 0, sizeof(uint32_t) * (nslots + 1u));
    uint32_t *lod_of = (uint32_t *)malloc(4 * ((size_t)nv + 1));
    for (uint32_t v = 0; v < nv; v++) {
        const uint32_t i = w->vis[v];
        lod_of[v] = 0xFFFFFFFFu;
        if (!w->cached[i]) continue;
        const float d = ro_distance_to_aabb(w->tight[i], cam->pos);
        if (d > cam->far_draw) continue;                                   /* :489 */
        const uint32_t lod = ro_lod_adjusted_model_index(0u, d, cam->n_lod, cam->lod_min, cam->lod_max) >> 25;
        lod_of[v] = lod;
        for (uint32_t e = w->begin[i]; e < w->begin[i + 1]; e++) tot[w->ent_class[e] * 8u + lod]++;
    }
    uint32_t total = 0, ng = 0;
    for (uint32_t sidx = 0; sidx < nslots; sidx++) {
        const uint32_t c = tot[sidx]; tot[sidx] = total;
        if (c && ng < gcap && groups) { groups[ng].model_index = w->class_model[sidx >> 3] | ((sidx & 7u) << 25); groups[ng].render_system = w->class_rs[sidx >> 3]; groups[ng].sortable = w->class_sort[sidx >> 3]; groups[ng].begin = total; groups[ng].count = c; }
        if (c) ng++;
        total += c;
    }
    if (n_groups) *n_groups = ng;
    if (out_ids && out_mats)
        for (uint32_t v = 0; v < nv; v++) {
            if (lod_of[v] == 0xFFFFFFFFu) continue;
            const uint32_t i = w->vis[v];
            for (uint32_t e = w->begin[i]; e < w->begin[i + 1]; e++) {
                const uint32_t at = tot[w->ent_class[e] * 8u + lod_of[v]]++;
                if (at < cap) { out_ids[at] = w->ent_id[e]; memcpy(out_mats + (size_t)at * 16, w->ent_mat + (size_t)e * 16, 64); }   /* the 64-byte append (mapped_buffer.rs:166-189) */
            }
        }
    free(lod_of);
    return total;
}
