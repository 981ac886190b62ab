/*
 * re_oracle.c -- CPU ORACLE (test infrastructure only; see re_oracle.h header comment).
 *
 * Plain-C restatement of the reference's per-frame visible-set pipeline.  Citations are
 * relative to /root/reference/src.  Compile with -ffp-contract=off -fno-fast-math: Rust never
 * contracts a*b+c into an FMA, and every product/sum below is rounded separately on purpose.
 */
#define _GNU_SOURCE
#include "re_oracle.h"

#include <math.h>
#include <time.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------------------------------
 * small containers
 * ---------------------------------------------------------------------------------------- */
typedef struct { uint32_t *v; uint32_t n, cap; } u32set;   /* ascending, unique */
typedef struct { uint64_t *v; uint32_t n, cap; } u64set;   /* ascending, unique */

static uint32_t u32set_lb(const u32set *s, uint32_t x) {
    uint32_t lo = 0, hi = s->n;
    while (lo < hi) { uint32_t m = (lo + hi) >> 1; if (s->v[m] < x) lo = m + 1; else hi = m; }
    return lo;
}
static int u32set_has(const u32set *s, uint32_t x) { uint32_t i = u32set_lb(s, x); return i < s->n && s->v[i] == x; }
static int u32set_add(u32set *s, uint32_t x) {
    uint32_t i = u32set_lb(s, x);
    if (i < s->n && s->v[i] == x) return 0;
    if (s->n == s->cap) { s->cap = s->cap ? s->cap * 2 : 4; s->v = (uint32_t *)realloc(s->v, s->cap * sizeof(uint32_t)); }
    memmove(s->v + i + 1, s->v + i, (s->n - i) * sizeof(uint32_t));
    s->v[i] = x; s->n++; return 1;
}
static int u32set_del(u32set *s, uint32_t x) {
    uint32_t i = u32set_lb(s, x);
    if (!(i < s->n && s->v[i] == x)) return 0;
    memmove(s->v + i, s->v + i + 1, (s->n - i - 1) * sizeof(uint32_t));
    s->n--; return 1;
}
static void u32set_free(u32set *s) { free(s->v); s->v = NULL; s->n = s->cap = 0; }

static uint32_t u64set_lb(const u64set *s, uint64_t x) {
    uint32_t lo = 0, hi = s->n;
    while (lo < hi) { uint32_t m = (lo + hi) >> 1; if (s->v[m] < x) lo = m + 1; else hi = m; }
    return lo;
}
static int u64set_has(const u64set *s, uint64_t x) { uint32_t i = u64set_lb(s, x); return i < s->n && s->v[i] == x; }
static int u64set_add(u64set *s, uint64_t x) {
    uint32_t i = u64set_lb(s, x);
    if (i < s->n && s->v[i] == x) return 0;
    if (s->n == s->cap) { s->cap = s->cap ? s->cap * 2 : 8; s->v = (uint64_t *)realloc(s->v, s->cap * sizeof(uint64_t)); }
    memmove(s->v + i + 1, s->v + i, (s->n - i) * sizeof(uint64_t));
    s->v[i] = x; s->n++; return 1;
}
static void u64set_clear(u64set *s) { s->n = 0; }
static void u64set_free(u64set *s) { free(s->v); s->v = NULL; s->n = s->cap = 0; }
/* bulk-append then sort/unique: used where the reference inserts millions of keys into a HashSet */
static int cmp_u64(const void *a, const void *b) { uint64_t x = *(const uint64_t *)a, y = *(const uint64_t *)b; return x < y ? -1 : x > y; }
static void u64set_push_unsorted(u64set *s, uint64_t x) {
    if (s->n == s->cap) { s->cap = s->cap ? s->cap * 2 : 8; s->v = (uint64_t *)realloc(s->v, s->cap * sizeof(uint64_t)); }
    s->v[s->n++] = x;
}
static void u64set_normalize(u64set *s) {
    if (s->n < 2) return;
    qsort(s->v, s->n, sizeof(uint64_t), cmp_u64);
    uint32_t o = 1;
    for (uint32_t i = 1; i < s->n; i++) if (s->v[i] != s->v[o - 1]) s->v[o++] = s->v[i];
    s->n = o;
}

/* open-addressing u64 -> int32 map (stands in for hashbrown::HashMap<UniqueWorldSectionId, ..>) */
typedef struct { uint64_t *k; int32_t *v; uint32_t cap, n, tomb; } kmap;
#define KM_EMPTY 0xFFFFFFFFFFFFFFFFull
#define KM_TOMB  0xFFFFFFFFFFFFFFFEull
static uint64_t km_hash(uint64_t x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33; return x; }
static void km_init(kmap *m, uint32_t cap) {
    m->cap = cap; m->n = 0; m->tomb = 0;
    m->k = (uint64_t *)malloc(sizeof(uint64_t) * cap); m->v = (int32_t *)malloc(sizeof(int32_t) * cap);
    for (uint32_t i = 0; i < cap; i++) m->k[i] = KM_EMPTY;
}
static void km_free(kmap *m) { free(m->k); free(m->v); m->k = NULL; m->v = NULL; }
static int32_t km_get(const kmap *m, uint64_t key) {
    uint32_t mask = m->cap - 1, i = (uint32_t)km_hash(key) & mask;
    for (;;) {
        uint64_t k = m->k[i];
        if (k == key) return m->v[i];
        if (k == KM_EMPTY) return -1;
        i = (i + 1) & mask;
    }
}
static void km_put_nogrow(kmap *m, uint64_t key, int32_t val) {
    uint32_t mask = m->cap - 1, i = (uint32_t)km_hash(key) & mask;
    int64_t firsttomb = -1;
    for (;;) {
        uint64_t k = m->k[i];
        if (k == key) { m->v[i] = val; return; }
        if (k == KM_TOMB && firsttomb < 0) firsttomb = i;
        if (k == KM_EMPTY) {
            if (firsttomb >= 0) { i = (uint32_t)firsttomb; m->tomb--; }
            m->k[i] = key; m->v[i] = val; m->n++; return;
        }
        i = (i + 1) & mask;
    }
}
static void km_put(kmap *m, uint64_t key, int32_t val) {
    if ((uint64_t)(m->n + m->tomb + 1) * 10 > (uint64_t)m->cap * 6) {
        kmap nm; km_init(&nm, (m->n * 4 > m->cap) ? m->cap * 2 : m->cap);
        for (uint32_t i = 0; i < m->cap; i++) if (m->k[i] < KM_TOMB) km_put_nogrow(&nm, m->k[i], m->v[i]);
        km_free(m); *m = nm;
    }
    km_put_nogrow(m, key, val);
}
static void km_del(kmap *m, uint64_t key) {
    uint32_t mask = m->cap - 1, i = (uint32_t)km_hash(key) & mask;
    for (;;) {
        uint64_t k = m->k[i];
        if (k == key) { m->k[i] = KM_TOMB; m->n--; m->tomb++; return; }
        if (k == KM_EMPTY) return;
        i = (i + 1) & mask;
    }
}

/* ------------------------------------------------------------------------------------------
 * scalar helpers with Rust semantics
 * ---------------------------------------------------------------------------------------- */
/* Rust `f as u32`: truncate toward zero, saturate, NaN -> 0 */
static inline uint32_t f2u32(float f) {
    if (!(f > 0.0f)) return 0u;
    if (f >= 4294967296.0f) return 0xFFFFFFFFu;
    return (uint32_t)f;
}
/* Rust f32::max / f32::min: if one operand is NaN the other is returned (== fmaxf/fminf) */
static inline float rmax(float a, float b) { return fmaxf(a, b); }
static inline float rmin(float a, float b) { return fminf(a, b); }

/* nalgebra Matrix::norm for a 3-vector: dotc special-cases dimension 3 as (a + b) + c, then sqrt */
float ro_norm3(float x, float y, float z) { return sqrtf((x * x + y * y) + z * z); }

/* ------------------------------------------------------------------------------------------
 * deterministic sin/cos shared with the HIP kernels (see header).  All arithmetic in f64 with
 * separately rounded operations; identical source is compiled into the device code.
 * ---------------------------------------------------------------------------------------- */
void ro_sincosf(float xf, float *s, float *c) {
    const double INV_PIO2 = 6.36619772367581382433e-01;
    const double PIO2_1   = 1.57079632673412561417e+00; /* first 33 bits of pi/2 */
    const double PIO2_1T  = 6.07710050650619224932e-11; /* pi/2 - PIO2_1 */
    const double TWO_PI   = 6.28318530717958623200e+00;
    const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03,
                 S3 = -1.98412698298579493134e-04, S4 = 2.75573137070700676789e-06,
                 S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
    const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03,
                 C3 = 2.48015872894767294178e-05, C4 = -2.75573143513906633035e-07,
                 C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
    double x = (double)xf;
    if (!(fabs(x) < 1.0e6)) {
        if (!(fabs(x) <= 3.5e38)) { *s = (float)(x - x); *c = (float)(x - x); return; } /* inf/nan -> nan */
        x = fmod(x, TWO_PI);                                                           /* exact operation */
    }
    double fn = rint(x * INV_PIO2);
    double r = (x - fn * PIO2_1) - fn * PIO2_1T;
    long long n = (long long)fn;
    double z = r * r;
    double ps = S1 + z * (S2 + z * (S3 + z * (S4 + z * (S5 + z * S6))));
    double sn = r + r * (z * ps);
    double pc = C1 + z * (C2 + z * (C3 + z * (C4 + z * (C5 + z * C6))));
    double cs = (1.0 - 0.5 * z) + (z * z) * pc;
    double so, co;
    switch ((int)(n & 3)) {
        case 0: so = sn;  co = cs;  break;
        case 1: so = cs;  co = -sn; break;
        case 2: so = -sn; co = -cs; break;
        default: so = -cs; co = sn; break;
    }
    *s = (float)so; *c = (float)co;
}

/* ------------------------------------------------------------------------------------------
 * matrices (column-major 4x4, m[col*4+row], nalgebra storage)
 * ---------------------------------------------------------------------------------------- */
/* nalgebra 0.25 Mul -> gemm(1, a, b, 0): per output column j, gemv: y = a[:,0]*b[0,j], then
 * y = a[:,k]*b[k,j] + y for k = 1..3 (axcpy, each product and sum rounded separately). */
void ro_mat4_mul(const float *a, const float *b, float *out) {
    float r[16];
    for (int j = 0; j < 4; j++)
        for (int i = 0; i < 4; i++) {
            float y = a[0 * 4 + i] * b[j * 4 + 0];
            y = a[1 * 4 + i] * b[j * 4 + 1] + y;
            y = a[2 * 4 + i] * b[j * 4 + 2] + y;
            y = a[3 * 4 + i] * b[j * 4 + 3] + y;
            r[j * 4 + i] = y;
        }
    memcpy(out, r, sizeof r);
}
void ro_mat4_vec4(const float *m, const float *v, float *out) {
    float r[4];
    for (int i = 0; i < 4; i++) {
        float y = m[0 * 4 + i] * v[0];
        y = m[1 * 4 + i] * v[1] + y;
        y = m[2 * 4 + i] * v[2] + y;
        y = m[3 * 4 + i] * v[3] + y;
        r[i] = y;
    }
    memcpy(out, r, sizeof r);
}
static void mat4_identity(float *m) { memset(m, 0, 16 * sizeof(float)); m[0] = m[5] = m[10] = m[15] = 1.0f; }

/* nalgebra_glm::translate(m, v) = m.prepend_translation(v):
 *   scale = m[3,0..3] . v (tr_dot, 3-vector: (a+b)+c);  post = m[0..3,0..3] * v (gemv order);
 *   m[3,3] += scale;  m[0..3,3] += post */
static void mat4_translate(float *m, const float v[3]) {
    float scale = (m[0 * 4 + 3] * v[0] + m[1 * 4 + 3] * v[1]) + m[2 * 4 + 3] * v[2];
    float post[3];
    for (int i = 0; i < 3; i++) {
        float y = m[0 * 4 + i] * v[0];
        y = m[1 * 4 + i] * v[1] + y;
        y = m[2 * 4 + i] * v[2] + y;
        post[i] = y;
    }
    m[3 * 4 + 3] += scale;
    for (int i = 0; i < 3; i++) m[3 * 4 + i] += post[i];
}
/* nalgebra_glm::rotate(m, angle, axis) = m * Rotation3::from_axis_angle(Unit::new_normalize(axis), angle).to_homogeneous()
 * Rotation3::from_axis_angle: identity when angle == 0, else the Rodrigues entries below. */
static void mat4_rotate(float *m, float angle, const float axis[3]) {
    float n = ro_norm3(axis[0], axis[1], axis[2]);
    float ux = axis[0] / n, uy = axis[1] / n, uz = axis[2] / n;
    float r[16]; mat4_identity(r);
    if (angle != 0.0f) {
        float sqx = ux * ux, sqy = uy * uy, sqz = uz * uz;
        float sn, cs; ro_sincosf(angle, &sn, &cs);
        float omc = 1.0f - cs;
        /* row-major listing of Matrix3::new(...) */
        float m11 = sqx + (1.0f - sqx) * cs;
        float m12 = ux * uy * omc - uz * sn;
        float m13 = ux * uz * omc + uy * sn;
        float m21 = ux * uy * omc + uz * sn;
        float m22 = sqy + (1.0f - sqy) * cs;
        float m23 = uy * uz * omc - ux * sn;
        float m31 = ux * uz * omc - uy * sn;
        float m32 = uy * uz * omc + ux * sn;
        float m33 = sqz + (1.0f - sqz) * cs;
        r[0] = m11; r[1] = m21; r[2] = m31;
        r[4] = m12; r[5] = m22; r[6] = m32;
        r[8] = m13; r[9] = m23; r[10] = m33;
    }
    ro_mat4_mul(m, r, m);
}
/* nalgebra_glm::scale(m, v) = m.prepend_nonuniform_scaling(v): column i (all 4 rows) *= v[i] */
static void mat4_scale(float *m, const float v[3]) {
    for (int c = 0; c < 3; c++) for (int r = 0; r < 4; r++) m[c * 4 + r] *= v[c];
}

/* EntityTransformationBuilder::write_components (exports/entity_transformer.rs:99-142) and
 * update_aabb_after_kinematic_change (helper_things/entity_change_helpers.rs:248-250) */
void ro_trs_matrix(const float pos[3], int has_rot, const float axis[3], float angle,
                   int has_scale, const float scale[3], float out[16]) {
    mat4_identity(out);
    mat4_translate(out, pos);
    if (has_rot) mat4_rotate(out, angle, axis);
    if (has_scale) mat4_scale(out, scale);
}

/* StaticAABB::apply_transformation (world/bounding_volumes/aabb.rs:95-114): only the min and
 * max corners are transformed */
ro_aabb ro_apply_transformation(ro_aabb a, const float m[16]) {
    float c0[4] = { a.xmin, a.ymin, a.zmin, 1.0f }, c1[4] = { a.xmax, a.ymax, a.zmax, 1.0f };
    float f[4], s[4];
    ro_mat4_vec4(m, c0, f); ro_mat4_vec4(m, c1, s);
    ro_aabb o;
    o.xmin = rmin(f[0], s[0]); o.ymin = rmin(f[1], s[1]); o.zmin = rmin(f[2], s[2]);
    o.xmax = rmax(f[0], s[0]); o.ymax = rmax(f[1], s[1]); o.zmax = rmax(f[2], s[2]);
    return o;
}

/* Range::combine (world/dimension/range.rs:38-61), epsilon-biased union */
static void range_combine(float amin, float amax, float bmin, float bmax, float *omin, float *omax) {
    const float epsilon = 0.01f;
    *omin = ((amin - epsilon) < bmin) ? amin : bmin;
    *omax = ((amax + epsilon) > bmax) ? amax : bmax;
}
ro_aabb ro_combine_aabb(ro_aabb a, ro_aabb b) {
    ro_aabb o;
    range_combine(a.xmin, a.xmax, b.xmin, b.xmax, &o.xmin, &o.xmax);
    range_combine(a.ymin, a.ymax, b.ymin, b.ymax, &o.ymin, &o.ymax);
    range_combine(a.zmin, a.zmax, b.zmin, b.zmax, &o.zmin, &o.zmax);
    return o;
}

/* distance_to_aabb (helper_things/aabb_helper_functions.rs:58-72) */
float ro_distance_to_aabb(ro_aabb a, const float cam[3]) {
    float lx = a.xmax - a.xmin, ly = a.ymax - a.ymin, lz = a.zmax - a.zmin;
    float largest = rmax(rmax(lx, ly), lz);
    float h = largest / 2.0f;
    float radius = sqrtf((h * h) * 3.0f);                    /* powi(2) == h*h */
    float cx = (a.xmin + a.xmax) / 2.0f, cy = (a.ymin + a.ymax) / 2.0f, cz = (a.zmin + a.zmax) / 2.0f;
    float d = ro_norm3(cam[0] - cx, cam[1] - cy, cam[2] - cz);
    return rmax(d - radius, 0.0f);
}

/* RenderFrustumCuller::update_plane_coefficients (culling/render_frustum_culler.rs:59-78).
 * column(k) of transpose(PV) == row k of PV. */
void ro_make_planes(const float pv[16], float planes[24]) {
    float row[4][4];
    for (int r = 0; r < 4; r++) for (int c = 0; c < 4; c++) row[r][c] = pv[c * 4 + r];
    float p[6][4];
    for (int i = 0; i < 4; i++) {
        p[0][i] = row[3][i] + row[0][i];   /* Left   */
        p[1][i] = row[3][i] - row[0][i];   /* Right  */
        p[2][i] = row[3][i] + row[1][i];   /* Bottom */
        p[3][i] = row[3][i] - row[1][i];   /* Top    */
        p[4][i] = row[3][i] - 0.0f;        /* Near: column(3) - vec4(0,0,0,0) */
        p[5][i] = row[3][i] - row[2][i];   /* Far    */
    }
    for (int k = 0; k < 6; k++) {
        float len = ro_norm3(p[k][0], p[k][1], p[k][2]);
        for (int i = 0; i < 4; i++) planes[k * 4 + i] = p[k][i] / len;
    }
}

/* StaticAABB::get_aabb_points (aabb.rs:128-140) */
static void aabb_points(ro_aabb a, float pts[8][3]) {
    const float xs[2] = { a.xmin, a.xmax }, ys[2] = { a.ymin, a.ymax }, zs[2] = { a.zmin, a.zmax };
    int k = 0;
    for (int ix = 0; ix < 2; ix++) for (int iy = 0; iy < 2; iy++) for (int iz = 0; iz < 2; iz++) {
        pts[k][0] = xs[ix]; pts[k][1] = ys[iy]; pts[k][2] = zs[iz]; k++;
    }
}

/* RenderFrustumCuller::aabb_visible (render_frustum_culler.rs:83-118): per plane, OR over the 8
 * corners of !(dist < 0); AND over planes.  NaN distances count as inside. */
int ro_frustum_aabb_visible(const float planes[24], ro_aabb a) {
    float pts[8][3]; aabb_points(a, pts);
    for (int k = 0; k < 6; k++) {
        const float *pl = planes + k * 4;
        int any = 0;
        for (int i = 0; i < 8; i++) {
            float d = pl[0] * pts[i][0] + pl[1] * pts[i][1] + pl[2] * pts[i][2] + pl[3];
            any |= !(d < 0.0f);
        }
        if (!any) return 0;
    }
    return 1;
}

/* LogicFrustumCuller::aabb_in_view (culling/logic_frustum_culler.rs:32-46) */
int ro_logic_aabb_in_view(float lookahead, const float cam[3], ro_aabb a) {
    float pts[8][3]; aabb_points(a, pts);
    float best = 3.40282347e+38f; /* f32::MAX */
    for (int i = 0; i < 8; i++) {
        float d = ro_norm3(pts[i][0] - cam[0], pts[i][1] - cam[1], pts[i][2] - cam[2]);
        best = rmin(best, d);
    }
    return best <= lookahead;
}

/* ModelId::level_of_view_adjusted_model_index (models/model_definitions.rs:31-59) */
uint32_t ro_lod_adjusted_model_index(uint32_t model_index, float d, uint32_t n, const float *lmin, const float *lmax) {
    uint32_t lod = 7u;
    for (uint32_t i = 0; i < n; i++) if (lmin[i] <= d && d <= lmax[i]) { lod = i < 7u ? i : 7u; break; }
    return model_index | (lod << 25);
}

/* level_views.custom (flows/render_flow.rs:495-499, 889-893): a model registered with custom_level_of_view uses its own bands, any other the
 * render system's default bands (the camera's).  Linear table: the sample registers a handful of models. */
typedef struct { uint32_t model_index, render_system, n; float lmin[8], lmax[8]; } custom_lod_t;
typedef struct { float one_thread[5]; float total_us; } tth_t;   /* TimeTakeHistory (helper_things/cpu_usage_reducer.rs:30-35) */
static void tth_init(tth_t *h);
static const custom_lod_t *custom_lod_find(const custom_lod_t *t, uint32_t nt, uint32_t model_index, uint32_t rs) {
    for (uint32_t i = 0; i < nt; i++) if (t[i].model_index == model_index && t[i].render_system == rs) return &t[i];
    return NULL;
}
static uint32_t lod_model_index(const custom_lod_t *t, uint32_t nt, const ro_camera *cam, uint32_t model_index, uint32_t rs, float d) {
    const custom_lod_t *c = nt ? custom_lod_find(t, nt, model_index, rs) : NULL;
    return c ? ro_lod_adjusted_model_index(model_index, d, c->n, c->lmin, c->lmax) : ro_lod_adjusted_model_index(model_index, d, cam->n_lod, cam->lod_min, cam->lod_max);
}

/* create_level_of_views (prelude/default_render_system.rs:240-256) */
void ro_default_lod(float rd, float lmin[5], float lmax[5]) {
    float v1 = rd * 0.10f;
    float v2 = rd * 0.15f + v1;
    float v3 = rd * 0.20f + v2;
    float v4 = rd * 0.25f + v3;
    float v5 = rd * 0.30f + v4;
    lmin[0] = 0.0f; lmax[0] = v1; lmin[1] = v1; lmax[1] = v2; lmin[2] = v2; lmax[2] = v3;
    lmin[3] = v3; lmax[3] = v4; lmin[4] = v4; lmax[4] = v5;
}

/* BoundingBoxTree::max_level (world/bounding_box_tree_v2.rs:1356-1359) */
uint32_t ro_max_level(uint32_t outline, uint32_t atomic) {
    float v = ceilf(log2f((float)outline / (float)atomic));
    return f2u32(v) & 0xFFFFu;
}

/* key layout: level:16 | x:16 | z:16 | y:16  (field order of UniqueWorldSectionId, :21-26) */
uint64_t ro_pack_key(uint32_t level, uint32_t x, uint32_t z, uint32_t y) {
    return ((uint64_t)(level & 0xFFFFu) << 48) | ((uint64_t)(x & 0xFFFFu) << 32) | ((uint64_t)(z & 0xFFFFu) << 16) | (uint64_t)(y & 0xFFFFu);
}
#define KEY_LEVEL(k) ((uint32_t)((k) >> 48) & 0xFFFFu)
#define KEY_X(k) ((uint32_t)((k) >> 32) & 0xFFFFu)
#define KEY_Z(k) ((uint32_t)((k) >> 16) & 0xFFFFu)
#define KEY_Y(k) ((uint32_t)(k) & 0xFFFFu)

/* UniqueWorldSectionId::to_aabb (:95-109) */
ro_aabb ro_key_to_aabb(uint64_t key, uint32_t atomic) {
    uint32_t level = KEY_LEVEL(key);
    float side = (float)((level < 32 ? (1u << level) : 0u) * atomic);
    float mx = side * (float)KEY_X(key), my = side * (float)KEY_Y(key), mz = side * (float)KEY_Z(key);
    ro_aabb a = { mx, mx + side, my, my + side, mz, mz + side };
    return a;
}

/* calculate_number_world_sections_each_dimension closure (:1315-1346) */
static uint32_t num_sections_1d(float min, float max, uint32_t level_length) {
    float ll = (float)level_length;
    if (truncf(min / ll) == truncf(max / ll)) return 1;
    uint32_t n;
    if (ceilf(min / ll) > (min / ll)) { min = ceilf(min / ll) * ll; n = 1; } else n = 0;
    while (min < max) { n += 1; min += ll; }
    return n;
}
static uint32_t num_sections_total(uint32_t ll, ro_aabb a) {
    return num_sections_1d(a.xmin, a.xmax, ll) * num_sections_1d(a.ymin, a.ymax, ll) * num_sections_1d(a.zmin, a.zmax, ll);
}
/* find_aabb_level_from_length_and_origin (:532-551) */
static void level_from_origin(ro_aabb a, uint32_t atomic, uint32_t *level, uint32_t *ll) {
    uint32_t len = atomic, lv = 0;
    uint32_t n = num_sections_total(len, a);
    while (n > 1) { len *= 2u; lv += 1; n = num_sections_total(len, a); }
    *level = lv; *ll = len;
}
/* aabb_out_of_bounds (helper_things/aabb_helper_functions.rs:43-52) */
static int aabb_oob(ro_aabb a, float L) {
    return a.xmin < 0.0f || a.ymin < 0.0f || a.zmin < 0.0f || a.xmax > L || a.ymax > L || a.zmax > L;
}
/* normalize_aabb (:1384-1397) */
static int normalize_aabb(ro_aabb *a, float L) {
    int oob = aabb_oob(*a, L);
    a->xmin = rmin(rmax(a->xmin, 0.0f), L); a->ymin = rmin(rmax(a->ymin, 0.0f), L); a->zmin = rmin(rmax(a->zmin, 0.0f), L);
    a->xmax = rmin(rmax(a->xmax, 0.0f), L); a->ymax = rmin(rmax(a->ymax, 0.0f), L); a->zmax = rmin(rmax(a->zmax, 0.0f), L);
    return oob;
}

/* add_entity's cell decision: normalize_aabb (:567), find_all_unique_world_section_ids (:466-506),
 * and for a single section find_unique_world_section_id (:451-460) */
static int assign_cells_norm(ro_aabb bv, uint32_t atomic, uint64_t keys[8]) {
    /* find_aabb_level_from_length (:513-530): level of the origin-shifted box */
    ro_aabb shifted = { 0.0f, bv.xmax - bv.xmin, 0.0f, bv.ymax - bv.ymin, 0.0f, bv.zmax - bv.zmin };
    uint32_t level, ll; level_from_origin(shifted, atomic, &level, &ll);
    uint32_t nx = num_sections_1d(bv.xmin, bv.xmax, ll), ny = num_sections_1d(bv.ymin, bv.ymax, ll), nz = num_sections_1d(bv.zmin, bv.zmax, ll);
    uint64_t total = (uint64_t)nx * ny * nz;
    if (total == 0 || total > 8) return (int)(total > 8 ? -2 : 0);      /* reference asserts len <= 8 (:502) */
    int n = 0;
    for (uint32_t x = 0; x < nx; x++) for (uint32_t y = 0; y < ny; y++) for (uint32_t z = 0; z < nz; z++) {
        /* calculate_aabb_section_indexes (:1367-1378) */
        uint32_t ix = (f2u32(bv.xmin) + ll * x) / ll, iy = (f2u32(bv.ymin) + ll * y) / ll, iz = (f2u32(bv.zmin) + ll * z) / ll;
        keys[n++] = ro_pack_key(level, ix, iz, iy);
    }
    if (n == 1) {
        uint32_t l2, len2; level_from_origin(bv, atomic, &l2, &len2);
        keys[0] = ro_pack_key(l2, f2u32(bv.xmin) / len2, f2u32(bv.zmin) / len2, f2u32(bv.ymin) / len2);
    }
    return n;
}
int ro_assign_cells(ro_aabb a, uint32_t outline, uint32_t atomic, uint64_t keys[8], int *oob) {
    int o = normalize_aabb(&a, (float)outline);
    if (oob) *oob = o;
    return assign_cells_norm(a, atomic, keys);
}

/* camera helpers.  nalgebra Perspective3::new(aspect, fovy, znear, zfar) */
void ro_perspective(float aspect, float fovy, float znear, float zfar, float out[16]) {
    memset(out, 0, 16 * sizeof(float));
    float m11 = 1.0f / tanf(fovy / 2.0f);
    out[5] = m11;
    out[0] = m11 / aspect;
    out[10] = (zfar + znear) / (znear - zfar);
    out[14] = zfar * znear * 2.0f / (znear - zfar);
    out[11] = -1.0f;
}
/* right-handed look-at in the conventional form (f = normalize(target-eye), s = normalize(f x up),
 * u = s x f).  nalgebra routes this through a unit quaternion; the result agrees to rounding
 * but is NOT claimed bit-identical -- the view matrix is an *input* of the hot path. */
void ro_look_at(const float eye[3], const float target[3], const float up[3], float out[16]) {
    float f[3] = { target[0] - eye[0], target[1] - eye[1], target[2] - eye[2] };
    float fl = ro_norm3(f[0], f[1], f[2]); f[0] /= fl; f[1] /= fl; f[2] /= fl;
    float s[3] = { f[1] * up[2] - f[2] * up[1], f[2] * up[0] - f[0] * up[2], f[0] * up[1] - f[1] * up[0] };
    float sl = ro_norm3(s[0], s[1], s[2]); s[0] /= sl; s[1] /= sl; s[2] /= sl;
    float u[3] = { s[1] * f[2] - s[2] * f[1], s[2] * f[0] - s[0] * f[2], s[0] * f[1] - s[1] * f[0] };
    mat4_identity(out);
    out[0] = s[0]; out[4] = s[1]; out[8] = s[2];
    out[1] = u[0]; out[5] = u[1]; out[9] = u[2];
    out[2] = -f[0]; out[6] = -f[1]; out[10] = -f[2];
    out[12] = -((s[0] * eye[0] + s[1] * eye[1]) + s[2] * eye[2]);
    out[13] = -((u[0] * eye[0] + u[1] * eye[1]) + u[2] * eye[2]);
    out[14] = ((f[0] * eye[0] + f[1] * eye[1]) + f[2] * eye[2]);
}

/* ------------------------------------------------------------------------------------------
 * world state
 * ---------------------------------------------------------------------------------------- */
typedef struct {
    uint8_t alive;          /* exists in the ECS */
    uint8_t lookup;         /* entities_index_lookup: 0 none, 1 Unique, 2 Shared */
    uint32_t model_index, render_system, sortable, flags;
    ro_aabb original, aabb;
    float pos[3], rot[4], scale[3], vel[3], acc[3], rotvel[4], rotacc[4];
    float mat[16];
    uint64_t ukey;          /* lookup == 1 */
    int32_t  shared;        /* lookup == 2: index into shared table */
} ent_t;

typedef struct {
    uint8_t used;
    uint8_t is_static_section;     /* member of static_world_sections */
    uint64_t key;
    ro_aabb aabb, backup;          /* UniqueWorldSectionEntities.aabb / back_up_aabb */
    u32set local, stat;            /* local_entities / static_entities */
    u32set shared;                 /* shared_sections_ids (indices into shared table) */
} cell_t;

typedef struct {
    uint8_t used;
    int nkeys; uint64_t keys[8];   /* SharedWorldSectionId (level + ordered offsets) */
    u32set ents, stat;
    ro_aabb aabb;
    uint32_t stamp;                /* "processed_world_sections" membership for the current pass */
} shared_t;

/* cached static rendering data of one unique world section (render_flow.rs:549-594) */
typedef struct { uint32_t id, model_index, render_system, sortable; float mat[16]; } cache_ent_t;   /* the WrittenInformation bytes: a snapshot */
typedef struct { uint64_t key; cache_ent_t *e; uint32_t n, cap; } cache_t;

typedef struct { uint32_t *v; uint32_t n, cap; } u32vec;
static void u32vec_push(u32vec *s, uint32_t x) {
    if (s->n == s->cap) { s->cap = s->cap ? s->cap * 2 : 16; s->v = (uint32_t *)realloc(s->v, s->cap * sizeof(uint32_t)); }
    s->v[s->n++] = x;
}

struct ro_world {
    uint32_t outline, atomic;
    int nthreads;
    custom_lod_t *custom_lod; uint32_t n_custom_lod;         /* level_views.custom: see lod_model_index */
    tth_t tth_render, tth_positions;                         /* VISIBLE_WORLD_SECTIONS_HISTORY (render_flow.rs:649), POSITION_TIME_HISTORY (logic_flow.rs:356) */
    ent_t *ents; uint32_t ents_cap;
    cell_t *cells; uint32_t ncells_alloc, cells_cap; u32vec cell_free; uint32_t ncells_live;
    kmap cellmap;
    shared_t *shared; uint32_t nshared_alloc, shared_cap, nshared_live; u32vec shared_free, shared_pending_free;
    kmap sharedmap;
    uint32_t *shared_order; uint32_t shared_order_n; int shared_order_dirty;
    /* changed_world_sections: hash set (membership, :710/:911) + list (iteration) */
    kmap changed_cells_map; u64set changed_cells;
    u64set changed_static_unique;          /* appended unsorted, normalised before iteration */
    u32vec changed_shared;                 /* may hold duplicates; normalised before iteration */
    uint32_t total_combining;
    uint32_t pass_id;
    /* ECS-side index lists (stand in for get_indexes_for_components, objects/ecs.rs:238-285) */
    u32vec always_exec, marked;
    /* static render cache: key -> cache index */
    kmap cachemap; cache_t *caches; uint32_t ncaches, caches_cap;
    /* last CullResult */
    uint64_t *vis_vec; uint32_t vis_n, vis_cap;   /* visible_sections_vec, duplicates kept */
    u64set vis_map;                                /* visible_sections_map */
    u64set light_vis; ro_aabb light_box;           /* the light query's visible_sections_map and culler (ro_visible_lights) */
    float planes[24]; float lookahead; float campos[3];
};

ro_world *ro_world_new(uint32_t outline, uint32_t atomic) {
    ro_world *w = (ro_world *)calloc(1, sizeof(ro_world));
    w->outline = outline; w->atomic = atomic; w->nthreads = 1;
    tth_init(&w->tth_render); tth_init(&w->tth_positions);
    km_init(&w->cellmap, 1024); km_init(&w->cachemap, 1024); km_init(&w->sharedmap, 256); km_init(&w->changed_cells_map, 256);
    w->shared_order_dirty = 1;
    return w;
}
void ro_set_threads(ro_world *w, int n) { w->nthreads = n < 1 ? 1 : n; }
void ro_world_free(ro_world *w) {
    if (!w) return;
    for (uint32_t i = 0; i < w->ncells_alloc; i++) { u32set_free(&w->cells[i].local); u32set_free(&w->cells[i].stat); u32set_free(&w->cells[i].shared); }
    for (uint32_t i = 0; i < w->nshared_alloc; i++) { u32set_free(&w->shared[i].ents); u32set_free(&w->shared[i].stat); }
    for (uint32_t i = 0; i < w->ncaches; i++) free(w->caches[i].e);
    free(w->ents); free(w->cells); free(w->cell_free.v); free(w->shared); free(w->shared_free.v); free(w->shared_pending_free.v);
    free(w->shared_order); free(w->caches); free(w->vis_vec); free(w->changed_shared.v); free(w->always_exec.v); free(w->marked.v);
    km_free(&w->cellmap); km_free(&w->cachemap); km_free(&w->sharedmap); km_free(&w->changed_cells_map);
    u64set_free(&w->changed_cells); u64set_free(&w->changed_static_unique); u64set_free(&w->vis_map); u64set_free(&w->light_vis);
    free(w->custom_lod);
    free(w);
}
/* register_model_with_render_system(.., custom_level_of_view, ..) (flows/render_flow.rs:1069-1076): custom bands of one model; n == 0 removes them */
void ro_set_model_lod(ro_world *w, uint32_t model_index, uint32_t render_system, uint32_t n, const float *lmin, const float *lmax) {
    custom_lod_t *c = NULL;
    for (uint32_t i = 0; i < w->n_custom_lod; i++) if (w->custom_lod[i].model_index == model_index && w->custom_lod[i].render_system == render_system) c = &w->custom_lod[i];
    if (!n) { if (c) { *c = w->custom_lod[--w->n_custom_lod]; } return; }
    if (!c) { w->custom_lod = (custom_lod_t *)realloc(w->custom_lod, (size_t)(w->n_custom_lod + 1) * sizeof(custom_lod_t)); c = &w->custom_lod[w->n_custom_lod++]; }
    c->model_index = model_index; c->render_system = render_system; c->n = n > 8 ? 8 : n;
    for (uint32_t k = 0; k < c->n; k++) { c->lmin[k] = lmin[k]; c->lmax[k] = lmax[k]; }
}
static ent_t *ent_slot(ro_world *w, uint32_t id) {
    if (id >= w->ents_cap) {
        uint32_t nc = w->ents_cap ? w->ents_cap : 64; while (nc <= id) nc *= 2;
        w->ents = (ent_t *)realloc(w->ents, (size_t)nc * sizeof(ent_t));
        memset(w->ents + w->ents_cap, 0, (size_t)(nc - w->ents_cap) * sizeof(ent_t));
        w->ents_cap = nc;
    }
    return &w->ents[id];
}
static const ent_t *ent_get(const ro_world *w, uint32_t id) { return id < w->ents_cap ? &w->ents[id] : NULL; }

static int  chg_has(const ro_world *w, uint64_t key) { return km_get(&w->changed_cells_map, key) >= 0; }
static void chg_add(ro_world *w, uint64_t key) {
    if (chg_has(w, key)) return;
    km_put(&w->changed_cells_map, key, 1);
    u64set_push_unsorted(&w->changed_cells, key);
}

static int32_t cell_find(const ro_world *w, uint64_t key) { return km_get(&w->cellmap, key); }
static int32_t cell_create(ro_world *w, uint64_t key) {
    int32_t idx;
    if (w->cell_free.n) idx = (int32_t)w->cell_free.v[--w->cell_free.n];
    else {
        if (w->ncells_alloc == w->cells_cap) { w->cells_cap = w->cells_cap ? w->cells_cap * 2 : 256; w->cells = (cell_t *)realloc(w->cells, (size_t)w->cells_cap * sizeof(cell_t)); }
        idx = (int32_t)w->ncells_alloc++;
        memset(&w->cells[idx], 0, sizeof(cell_t));
    }
    cell_t *c = &w->cells[idx];
    c->used = 1; c->is_static_section = 0; c->key = key;
    memset(&c->aabb, 0, sizeof(ro_aabb));                 /* StaticAABB::point_aabb() */
    c->backup = ro_key_to_aabb(key, w->atomic);
    c->local.n = c->stat.n = c->shared.n = 0;
    km_put(&w->cellmap, key, idx);
    w->ncells_live++;
    return idx;
}
static void cell_destroy(ro_world *w, int32_t idx) {
    cell_t *c = &w->cells[idx];
    km_del(&w->cellmap, c->key);
    c->used = 0; c->local.n = c->stat.n = c->shared.n = 0;
    u32vec_push(&w->cell_free, (uint32_t)idx);
    w->ncells_live--;
}
static uint64_t shared_hash(int nkeys, const uint64_t *keys) {
    uint64_t h = 0x9E3779B97F4A7C15ull ^ (uint64_t)nkeys;
    for (int i = 0; i < nkeys; i++) h = km_hash(h ^ keys[i]);
    return h & 0x7FFFFFFFFFFFFFFFull;
}
static int shared_same(const shared_t *s, int nkeys, const uint64_t *keys) {
    return s->used && s->nkeys == nkeys && memcmp(s->keys, keys, sizeof(uint64_t) * (size_t)nkeys) == 0;
}
static int32_t shared_lookup(const ro_world *w, int nkeys, const uint64_t *keys) {
    int32_t idx = km_get(&w->sharedmap, shared_hash(nkeys, keys));
    if (idx < 0) return -1;
    if (shared_same(&w->shared[idx], nkeys, keys)) return idx;
    for (uint32_t i = 0; i < w->nshared_alloc; i++) if (shared_same(&w->shared[i], nkeys, keys)) return (int32_t)i;  /* hash collision */
    return -1;
}
static int32_t shared_create(ro_world *w, int nkeys, const uint64_t *keys) {
    int32_t idx;
    if (w->shared_free.n) idx = (int32_t)w->shared_free.v[--w->shared_free.n];
    else {
        if (w->nshared_alloc == w->shared_cap) { w->shared_cap = w->shared_cap ? w->shared_cap * 2 : 64; w->shared = (shared_t *)realloc(w->shared, (size_t)w->shared_cap * sizeof(shared_t)); }
        idx = (int32_t)w->nshared_alloc++;
        memset(&w->shared[idx], 0, sizeof(shared_t));
    }
    shared_t *s = &w->shared[idx];
    s->used = 1; s->nkeys = nkeys; memcpy(s->keys, keys, sizeof(uint64_t) * (size_t)nkeys);
    s->ents.n = s->stat.n = 0; memset(&s->aabb, 0, sizeof(ro_aabb));   /* SharedWorldSectionEntities::new: point_aabb */
    s->stamp = 0;
    if (km_get(&w->sharedmap, shared_hash(nkeys, keys)) < 0) km_put(&w->sharedmap, shared_hash(nkeys, keys), idx);
    w->nshared_live++; w->shared_order_dirty = 1;
    return idx;
}

/* ------------------------------------------------------------------------------------------
 * BoundingBoxTree::remove_entity (:787-942).  Lights and related_world_sections are not
 * modelled (light queries and the collision broad phase are outside the hot path).
 * ---------------------------------------------------------------------------------------- */
void ro_tree_remove(ro_world *w, uint32_t id) {
    ent_t *e = ent_slot(w, id);
    if (e->lookup == 0) return;
    int kind = e->lookup; e->lookup = 0;
    uint64_t to_remove[8]; int n_remove = 0;
    if (kind == 2) {
        int32_t si = e->shared; shared_t *s = &w->shared[si];
        /* :808-828 with no lights: removing a static member marks every linked unique section */
        if (u32set_has(&s->stat, id)) for (int k = 0; k < s->nkeys; k++) u64set_push_unsorted(&w->changed_static_unique, s->keys[k]);
        /* SharedWorldSectionEntities::remove_entity (:307-315) */
        if (!u32set_del(&s->ents, id)) u32set_del(&s->stat, id);
        if (s->ents.n == 0 && s->stat.n == 0) {                  /* :838-870 */
            for (int k = 0; k < s->nkeys; k++) {
                int32_t ci = cell_find(w, s->keys[k]);
                if (ci < 0) continue;                            /* reference: unreachable!() */
                cell_t *c = &w->cells[ci];
                u32set_del(&c->shared, (uint32_t)si);
                if (c->local.n == 0 && c->stat.n == 0 && c->shared.n == 0) to_remove[n_remove++] = s->keys[k];
            }
            if (km_get(&w->sharedmap, shared_hash(s->nkeys, s->keys)) == si) km_del(&w->sharedmap, shared_hash(s->nkeys, s->keys));
            s->used = 0; w->nshared_live--; w->shared_order_dirty = 1;
            u32vec_push(&w->shared_pending_free, (uint32_t)si);
        }
        u32vec_push(&w->changed_shared, (uint32_t)si);           /* :872 */
    } else {
        uint64_t key = e->ukey;
        int32_t ci = cell_find(w, key);
        if (ci >= 0) {
            cell_t *c = &w->cells[ci];
            if (!u32set_del(&c->local, id)) {                    /* :895-901 */
                u32set_del(&c->stat, id);
                u64set_push_unsorted(&w->changed_static_unique, key);
            }
            if (c->local.n == 0 && c->stat.n == 0 && c->shared.n == 0) to_remove[n_remove++] = key;
            else {
                if (chg_has(w, key)) w->total_combining += 1;   /* :911-915 */
                else w->total_combining += c->local.n + c->stat.n;
            }
        }
        chg_add(w, key);                                         /* :921 */
    }
    for (int i = 0; i < n_remove; i++) {                         /* :925-937 */
        int32_t ci = cell_find(w, to_remove[i]);
        if (ci >= 0) cell_destroy(w, ci);
    }
}

/* ------------------------------------------------------------------------------------------
 * BoundingBoxTree::add_entity (:563-762)
 * ---------------------------------------------------------------------------------------- */
int ro_tree_add(ro_world *w, uint32_t id, ro_aabb bv, int add_if_oob, int is_static) {
    ent_t *e = ent_slot(w, id);
    int oob = normalize_aabb(&bv, (float)w->outline);
    if (oob && !add_if_oob) return -1;                              /* :569-572 */
    uint64_t keys[8];
    int n = assign_cells_norm(bv, w->atomic, keys);
    if (n <= 0) return -1;                                          /* reference would panic (:502 / empty id) */
    if (n != 1) {
        int32_t si = shared_lookup(w, n, keys);
        if (e->lookup == 2 && si >= 0 && e->shared == si) return 0; /* entity_exists_in_section (:765-782) */
        ro_tree_remove(w, id);
        si = shared_lookup(w, n, keys);                             /* the removal may have deleted the section */
        if (is_static) for (int k = 0; k < n; k++) u64set_push_unsorted(&w->changed_static_unique, keys[k]);   /* :590-596 */
        if (si < 0) {
            si = shared_create(w, n, keys);
            for (int k = 0; k < n; k++) {                           /* :634-674 */
                int32_t ci = cell_find(w, keys[k]);
                if (ci < 0) ci = cell_create(w, keys[k]);
                u32set_add(&w->cells[ci].shared, (uint32_t)si);
            }
        }
        shared_t *s = &w->shared[si];
        if (is_static) u32set_add(&s->stat, id); else u32set_add(&s->ents, id);
        e = ent_slot(w, id);
        e->lookup = 2; e->shared = si;
        u32vec_push(&w->changed_shared, (uint32_t)si);              /* :678 */
    } else {
        uint64_t key = keys[0];
        if (e->lookup == 1 && e->ukey == key) return 0;             /* entity_exists_in_section */
        ro_tree_remove(w, id);
        int32_t ci = cell_find(w, key);
        if (ci >= 0) {
            cell_t *c = &w->cells[ci];
            if (is_static) { u32set_add(&c->stat, id); u64set_push_unsorted(&w->changed_static_unique, key); }
            else u32set_add(&c->local, id);
            if (chg_has(w, key)) w->total_combining += 1;          /* :710-714 */
            else w->total_combining += c->local.n + c->stat.n;
        } else {
            ci = cell_create(w, key);
            cell_t *c = &w->cells[ci];
            if (is_static) { u32set_add(&c->stat, id); u64set_push_unsorted(&w->changed_static_unique, key); }
            else u32set_add(&c->local, id);
            w->total_combining += 1;                                /* :744 */
        }
        e = ent_slot(w, id);
        e->lookup = 1; e->ukey = key;
        chg_add(w, key);                                            /* :759 */
    }
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * update_static_world_sections (:1133-1213) and end_of_changes (:1055-1130).
 * The changed sets iterate in ascending key / ascending shared index (deterministic stand-in
 * for hashbrown order; the result depends on that order only when one frame changes two shared
 * sections, one with and one without active entities, that link the same unique section).
 * ---------------------------------------------------------------------------------------- */
static int cmp_u32(const void *a, const void *b) { uint32_t x = *(const uint32_t *)a, y = *(const uint32_t *)b; return x < y ? -1 : x > y; }
static void u32vec_normalize(u32vec *s) {
    if (s->n < 2) return;
    qsort(s->v, s->n, sizeof(uint32_t), cmp_u32);
    uint32_t o = 1;
    for (uint32_t i = 1; i < s->n; i++) if (s->v[i] != s->v[o - 1]) s->v[o++] = s->v[i];
    s->n = o;
}

static void update_static_world_sections(ro_world *w) {
    for (uint32_t i = 0; i < w->changed_cells.n; i++) {
        int32_t ci = cell_find(w, w->changed_cells.v[i]);
        if (ci < 0) continue;                         /* removed section: leaves the static set */
        cell_t *c = &w->cells[ci];
        int add_static = 0;
        if (c->local.n == 0) {
            if (c->shared.n == 0) add_static = 1;
            else for (uint32_t k = 0; k < c->shared.n; k++) {
                shared_t *s = &w->shared[c->shared.v[k]];
                if (s->used && s->ents.n == 0) add_static = 1;
            }
        }
        c->is_static_section = (uint8_t)add_static;
    }
    for (uint32_t i = 0; i < w->changed_shared.n; i++) {
        shared_t *s = &w->shared[w->changed_shared.v[i]];
        if (!s->used) continue;
        for (int k = 0; k < s->nkeys; k++) {
            int32_t ci = cell_find(w, s->keys[k]);
            if (ci < 0) continue;
            if (s->ents.n == 0) { if (w->cells[ci].local.n == 0) w->cells[ci].is_static_section = 1; }
            else w->cells[ci].is_static_section = 0;
        }
    }
}

static int cmp_shared_idx_world(const void *a, const void *b, void *arg);

void ro_end_of_changes(ro_world *w) {
    u64set_normalize(&w->changed_cells);
    u32vec_normalize(&w->changed_shared);
    /* deterministic stand-in for hash order: canonical id order (keys lexicographic, then count), independent of table slots */
    if (w->changed_shared.n > 1) qsort_r(w->changed_shared.v, w->changed_shared.n, sizeof(uint32_t), cmp_shared_idx_world, (void *)w);
    update_static_world_sections(w);
    int too_many = w->total_combining > 500;
    for (uint32_t i = 0; i < w->changed_cells.n; i++) {
        uint64_t key = w->changed_cells.v[i];
        int32_t ci = cell_find(w, key);
        if (ci < 0) continue;
        cell_t *c = &w->cells[ci];
        uint32_t adj = 20u + KEY_LEVEL(key) * 5u; if (adj > 50u) adj = 50u;         /* :1066,1074 */
        if (too_many && (c->local.n + c->stat.n) > adj) c->aabb = c->backup;
        else {
            ro_aabb u; memset(&u, 0, sizeof u);
            int first = 1;
            for (int pass = 0; pass < 2; pass++) {                                  /* local_entities.chain(static_entities) */
                const u32set *s = pass == 0 ? &c->local : &c->stat;
                for (uint32_t k = 0; k < s->n; k++) {
                    const ent_t *e = &w->ents[s->v[k]];
                    if (first) { u = e->aabb; first = 0; continue; }
                    u = ro_combine_aabb(u, e->aabb);
                }
            }
            c->aabb = u;
        }
    }
    /* shared branch (:1104-1125): first_entity is never cleared, so the AABB of the section is
     * the AABB of the LAST entity iterated (entities, then static_entities) */
    for (uint32_t i = 0; i < w->changed_shared.n; i++) {
        shared_t *s = &w->shared[w->changed_shared.v[i]];
        if (!s->used) continue;
        ro_aabb u; memset(&u, 0, sizeof u);
        for (uint32_t k = 0; k < s->ents.n; k++) u = w->ents[s->ents.v[k]].aabb;
        for (uint32_t k = 0; k < s->stat.n; k++) u = w->ents[s->stat.v[k]].aabb;
        s->aabb = u;
    }
    w->changed_shared.n = 0; u64set_clear(&w->changed_cells); w->total_combining = 0;
    km_free(&w->changed_cells_map); km_init(&w->changed_cells_map, 256);
    for (uint32_t i = 0; i < w->shared_pending_free.n; i++) u32vec_push(&w->shared_free, w->shared_pending_free.v[i]);
    w->shared_pending_free.n = 0;
}

/* ------------------------------------------------------------------------------------------
 * entity registration: Pipeline::register_model_instances (flows/pipeline.rs:186-208) with an
 * AddInstanceFunction that fills an EntityTransformationBuilder and calls apply_choices
 * (exports/entity_transformer.rs:55-75)
 * ---------------------------------------------------------------------------------------- */
static void normalize3(const float v[3], float out[3]) {
    float n = ro_norm3(v[0], v[1], v[2]);
    out[0] = v[0] / n; out[1] = v[1] / n; out[2] = v[2] / n;
}

/* One instance: ECS::create_entity (the id comes with the description) + EntityTransformationBuilder::apply_choices (exports/entity_transformer.rs:55-75,
 * 99-142): components, TransformationMatrix, StaticAABB, add_entity(id, aabb, false, is_static, light).  Returns 1 when the tree rejected it (out of bounds).
 * Called by register_model_instances (flows/pipeline.rs:186-208) and by the AddEntity arm of apply_change (helper_things/entity_change_helpers.rs:48-107). */
static int register_one(ro_world *w, const ro_entity_desc *d) {
    int rejected = 0;
    for (uint32_t i = 0; i < 1; i++) {
        ent_t *e = ent_slot(w, d[i].id);
        if (e->lookup) ro_tree_remove(w, d[i].id);
        memset(e, 0, sizeof *e);
        e->alive = 1;
        e->model_index = d[i].model_index; e->render_system = d[i].render_system; e->sortable = d[i].sortable;
        e->flags = d[i].flags & ~(RO_F_HAS_MOVED | RO_F_HAS_ROTATED);
        e->original = d[i].original;
        memcpy(e->pos, d[i].pos, sizeof e->pos);
        /* Rotation::new / VelocityRotation::new / AccelerationRotation::new normalise the axis
         * (exports/movement_components.rs:108-118,131-141,154-164) */
        if (e->flags & RO_F_HAS_ROT) { normalize3(d[i].rot_axis, e->rot); e->rot[3] = d[i].rot_angle; }
        else { e->rot[0] = 1.0f; e->rot[1] = 0.0f; e->rot[2] = 0.0f; e->rot[3] = 0.0f; }              /* Rotation::default (:41-47) */
        if (e->flags & RO_F_HAS_SCALE) memcpy(e->scale, d[i].scale, sizeof e->scale);
        else { e->scale[0] = e->scale[1] = e->scale[2] = 1.0f; }                                       /* Scale::default (:49-55) */
        memcpy(e->vel, d[i].vel, sizeof e->vel); memcpy(e->acc, d[i].acc, sizeof e->acc);
        if (e->flags & RO_F_HAS_ROTVEL) { normalize3(d[i].rotvel_axis, e->rotvel); e->rotvel[3] = d[i].rotvel; }
        if (e->flags & RO_F_HAS_ROTACC) { normalize3(d[i].rotacc_axis, e->rotacc); e->rotacc[3] = d[i].rotacc; }
        if (e->flags & RO_F_ALWAYS_EXEC) u32vec_push(&w->always_exec, d[i].id);
        if (e->flags & RO_F_USER) {
            /* Pipeline::register_user_entity (flows/pipeline.rs:125-144): OriginalAABB.translate(camera_pos) (aabb.rs translate: each
             * bound += offset) and an identity TransformationMatrix that nothing recomputes; added non-static (pipeline.rs:151) */
            memset(e->mat, 0, sizeof e->mat); e->mat[0] = e->mat[5] = e->mat[10] = e->mat[15] = 1.0f;
            e->aabb = e->original;
            e->aabb.xmin += e->pos[0]; e->aabb.xmax += e->pos[0]; e->aabb.ymin += e->pos[1]; e->aabb.ymax += e->pos[1];
            e->aabb.zmin += e->pos[2]; e->aabb.zmax += e->pos[2];
        } else {
        ro_trs_matrix(e->pos, (e->flags & RO_F_HAS_ROT) != 0, e->rot, e->rot[3], (e->flags & RO_F_HAS_SCALE) != 0, e->scale, e->mat);
        e->aabb = ro_apply_transformation(e->original, e->mat);
        }
        /* apply_choices: add_entity(id, &transformed, false, is_static, light) (:71) */
        if (ro_tree_add(w, d[i].id, e->aabb, 0, (e->flags & RO_F_STATIC) != 0) != 0) rejected++;
    }
    return rejected;
}
int ro_register_entities(ro_world *w, uint32_t n, const ro_entity_desc *d) {
    int rejected = 0;
    for (uint32_t i = 0; i < n; i++) rejected += register_one(w, &d[i]);
    ro_end_of_changes(w);
    return rejected;
}

/* ------------------------------------------------------------------------------------------
 * introspection
 * ---------------------------------------------------------------------------------------- */
uint32_t ro_num_cells(const ro_world *w) { return w->ncells_live; }
uint32_t ro_num_shared(const ro_world *w) { return w->nshared_live; }

typedef struct { uint64_t key; int32_t idx; } keyidx;
static int cmp_keyidx(const void *a, const void *b) { const keyidx *x = (const keyidx *)a, *y = (const keyidx *)b; return x->key < y->key ? -1 : x->key > y->key; }

uint32_t ro_get_cells(const ro_world *w, uint32_t cap, uint64_t *keys, ro_aabb *tight, uint32_t *n_local, uint32_t *n_static, uint32_t *n_shared, uint8_t *is_static_section) {
    keyidx *ki = (keyidx *)malloc(sizeof(keyidx) * ((size_t)w->ncells_live + 1));
    uint32_t n = 0;
    for (uint32_t i = 0; i < w->ncells_alloc; i++) if (w->cells[i].used) { ki[n].key = w->cells[i].key; ki[n].idx = (int32_t)i; n++; }
    qsort(ki, n, sizeof(keyidx), cmp_keyidx);
    for (uint32_t i = 0; i < n && i < cap; i++) {
        const cell_t *c = &w->cells[ki[i].idx];
        if (keys) keys[i] = c->key;
        if (tight) tight[i] = c->aabb;
        if (n_local) n_local[i] = c->local.n;
        if (n_static) n_static[i] = c->stat.n;
        if (n_shared) n_shared[i] = c->shared.n;
        if (is_static_section) is_static_section[i] = c->is_static_section;
    }
    free(ki);
    return n;
}
/* entity ids of one cell: active (ascending) then static (ascending) */
uint32_t ro_get_cell_entities(const ro_world *w, uint64_t key, uint32_t cap, uint32_t *ids, uint32_t *n_local) {
    int32_t ci = cell_find(w, key);
    if (ci < 0) { if (n_local) *n_local = 0; return 0; }
    const cell_t *c = &w->cells[ci]; uint32_t o = 0;
    for (uint32_t k = 0; k < c->local.n; k++, o++) if (o < cap) ids[o] = c->local.v[k];
    for (uint32_t k = 0; k < c->stat.n; k++, o++) if (o < cap) ids[o] = c->stat.v[k];
    if (n_local) *n_local = c->local.n;
    return o;
}
int ro_entity_lookup(const ro_world *w, uint32_t id, uint64_t keys[8], int *nkeys) {
    const ent_t *e = ent_get(w, id);
    if (!e || e->lookup == 0) { if (nkeys) *nkeys = 0; return 0; }
    if (e->lookup == 1) { keys[0] = e->ukey; if (nkeys) *nkeys = 1; return 1; }
    const shared_t *s = &w->shared[e->shared];
    memcpy(keys, s->keys, sizeof(uint64_t) * (size_t)s->nkeys); if (nkeys) *nkeys = s->nkeys;
    return 2;
}
int ro_get_entity(const ro_world *w, uint32_t id, float mat[16], ro_aabb *aabb, float pos[3], float rot[4], float rotvel[4], float vel[3], uint32_t *flags) {
    const ent_t *e = ent_get(w, id);
    if (!e || !e->alive) return 0;
    if (mat) memcpy(mat, e->mat, sizeof e->mat);
    if (aabb) *aabb = e->aabb;
    if (pos) memcpy(pos, e->pos, sizeof e->pos);
    if (rot) memcpy(rot, e->rot, sizeof e->rot);
    if (rotvel) memcpy(rotvel, e->rotvel, sizeof e->rotvel);
    if (vel) memcpy(vel, e->vel, sizeof e->vel);
    if (flags) *flags = e->flags;
    return 1;
}
static int cmp_shared_idx_world(const void *a, const void *b, void *arg) {
    const ro_world *w = (const ro_world *)arg;
    const shared_t *x = &w->shared[*(const uint32_t *)a], *y = &w->shared[*(const uint32_t *)b];
    int n = x->nkeys < y->nkeys ? x->nkeys : y->nkeys;
    for (int i = 0; i < n; i++) if (x->keys[i] != y->keys[i]) return x->keys[i] < y->keys[i] ? -1 : 1;
    return x->nkeys - y->nkeys;
}
static void shared_order_refresh(ro_world *w) {
    if (!w->shared_order_dirty) return;
    free(w->shared_order);
    w->shared_order = (uint32_t *)malloc(sizeof(uint32_t) * ((size_t)w->nshared_live + 1)); uint32_t n = 0;
    for (uint32_t i = 0; i < w->nshared_alloc; i++) if (w->shared[i].used) w->shared_order[n++] = i;
    qsort_r(w->shared_order, n, sizeof(uint32_t), cmp_shared_idx_world, (void *)w);
    w->shared_order_n = n; w->shared_order_dirty = 0;
}
int ro_get_shared(ro_world *w, uint32_t i, uint64_t keys[8], int *nkeys, ro_aabb *aabb, uint32_t cap, uint32_t *ids, uint32_t *n_active, uint32_t *n_static) {
    shared_order_refresh(w);
    if (i >= w->shared_order_n) return 0;
    const shared_t *s = &w->shared[w->shared_order[i]];
    memcpy(keys, s->keys, sizeof(uint64_t) * (size_t)s->nkeys); *nkeys = s->nkeys; if (aabb) *aabb = s->aabb;
    uint32_t o = 0;
    for (uint32_t k = 0; k < s->ents.n && o < cap; k++) ids[o++] = s->ents.v[k];
    for (uint32_t k = 0; k < s->stat.n && o < cap; k++) ids[o++] = s->stat.v[k];
    if (n_active) *n_active = s->ents.n;
    if (n_static) *n_static = s->stat.n;
    return 1;
}

/* ------------------------------------------------------------------------------------------
 * VisibleWorldFlow::find_visible_world_ids (flows/visible_world_flow.rs:40-115)
 * ---------------------------------------------------------------------------------------- */
typedef struct { uint64_t key; ro_aabb aabb; } cand_t;

static void vis_push(ro_world *w, uint64_t key) {
    if (w->vis_n == w->vis_cap) { w->vis_cap = w->vis_cap ? w->vis_cap * 2 : 256; w->vis_vec = (uint64_t *)realloc(w->vis_vec, sizeof(uint64_t) * w->vis_cap); }
    w->vis_vec[w->vis_n++] = key;
}

static int aabb_intersect(ro_aabb a, ro_aabb b);
/* which: 0 = LogicFrustumCuller, 1 = RenderFrustumCuller, 2 = the AABB culler of the light query (results in light_vis) */
static void find_visible_world_ids(ro_world *w, int which, ro_aabb box) {
    uint32_t maxl = ro_max_level(w->outline, w->atomic);
    float wsl = (float)w->atomic;
    /* serial enumeration (:47-90) */
    size_t ncand = 0, cap = 4096; cand_t *cands = (cand_t *)malloc(cap * sizeof(cand_t));
    for (uint32_t level = 0; level < maxl; level++) {
        float ll = wsl * ldexpf(1.0f, (int)level);                 /* 2.0_f32.powf(level as f32) */
        uint32_t nx = f2u32(ceilf((box.xmax - box.xmin) / ll)), ny = f2u32(ceilf((box.ymax - box.ymin) / ll)), nz = f2u32(ceilf((box.zmax - box.zmin) / ll));
        uint32_t bx = f2u32(box.xmin / ll), by = f2u32(box.ymin / ll), bz = f2u32(box.zmin / ll);
        for (uint32_t x = 0; x < nx; x++) for (uint32_t y = 0; y < ny; y++) for (uint32_t z = 0; z < nz; z++) {
            if (ncand == cap) { cap *= 2; cands = (cand_t *)realloc(cands, cap * sizeof(cand_t)); }
            cand_t *c = &cands[ncand++];
            c->key = ro_pack_key(level, (bx + x) & 0xFFFFu, (bz + z) & 0xFFFFu, (by + y) & 0xFFFFu);
            float fx = (float)(bx + x) * ll, fy = (float)(by + y) * ll, fz = (float)(bz + z) * ll;
            c->aabb.xmin = fx; c->aabb.xmax = fx + ll; c->aabb.ymin = fy; c->aabb.ymax = fy + ll; c->aabb.zmin = fz; c->aabb.zmax = fz + ll;
        }
    }
    /* par_chunks(25) probe + predicate (:94-108); results are merged in candidate order */
    long nchunks = (long)((ncand + 24) / 25);
    uint8_t *hit = (uint8_t *)calloc(ncand ? ncand : 1, 1);
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 1) num_threads(w->nthreads) if (w->nthreads > 1)
#endif
    for (long ch = 0; ch < nchunks; ch++) {
        size_t b = (size_t)ch * 25, e = b + 25 < ncand ? b + 25 : ncand;
        for (size_t i = b; i < e; i++) {
            if (cell_find(w, cands[i].key) < 0) continue;          /* is_section_in_existence (:389-392) */
            hit[i] = (uint8_t)(which == 2 ? aabb_intersect(w->light_box, cands[i].aabb)      /* shadow_flow.rs:80-86: Culler::aabb_in_view */
                               : which ? ro_frustum_aabb_visible(w->planes, cands[i].aabb)
                                       : ro_logic_aabb_in_view(w->lookahead, w->campos, cands[i].aabb));
        }
    }
    if (which == 2) { for (size_t i = 0; i < ncand; i++) if (hit[i]) u64set_push_unsorted(&w->light_vis, cands[i].key); }
    else for (size_t i = 0; i < ncand; i++) if (hit[i]) { vis_push(w, cands[i].key); u64set_push_unsorted(&w->vis_map, cands[i].key); }
    free(hit); free(cands);
}

/* Pipeline::execute cull section (flows/pipeline.rs:216-229) */
uint32_t ro_frame_cull(ro_world *w, const ro_camera *cam, uint32_t cap, uint64_t *keys_out) {
    float wsl = (float)w->atomic;
    ro_make_planes(cam->pv, w->planes);
    w->lookahead = wsl; memcpy(w->campos, cam->pos, sizeof w->campos);      /* LogicFrustumCuller::new(wsl, pos) */
    w->vis_n = 0; u64set_clear(&w->vis_map);
    /* find_visible_world_ids_entire_world -> generate_original_culling_aabb (:131-145), draw = 2*wsl */
    float draw = wsl * 2.0f;
    ro_aabb lb = { rmax(cam->pos[0] - draw, 0.0f), cam->pos[0] + draw, rmax(cam->pos[1] - draw, 0.0f), cam->pos[1] + draw, rmax(cam->pos[2] - draw, 0.0f), cam->pos[2] + draw };
    find_visible_world_ids(w, 0, lb);
    /* find_visible_world_ids_frustum_aabb (:117-129) */
    float half = cam->far_draw / 2.0f;
    float cx = cam->dir[0] * half + cam->pos[0], cy = cam->dir[1] * half + cam->pos[1], cz = cam->dir[2] * half + cam->pos[2];
    ro_aabb rb = { rmax(cx - half, 0.0f), cx + half, rmax(cy - half, 0.0f), cy + half, rmax(cz - half, 0.0f), cz + half };
    find_visible_world_ids(w, 1, rb);
    u64set_normalize(&w->vis_map);
    if (keys_out) {
        uint64_t *tmp = (uint64_t *)malloc(sizeof(uint64_t) * ((size_t)w->vis_n + 1));
        memcpy(tmp, w->vis_vec, sizeof(uint64_t) * w->vis_n);
        qsort(tmp, w->vis_n, sizeof(uint64_t), cmp_u64);
        memcpy(keys_out, tmp, sizeof(uint64_t) * (w->vis_n < cap ? w->vis_n : cap));
        free(tmp);
    }
    return w->vis_n;
}

/* The lights RenderFlow::render hands to the deferred pass and the shadow flow: find_nearby_world_sections_maps (flows/shadow_flow.rs:494-513: the
 * whole-world visibility query with an AABB culler of radius far_draw around the camera) -> find_nearby_lights (:455-487: the light sets of those
 * unique sections and of the shared sections linked to them; light sets: world/bounding_box_tree_v2.rs:157-228, maintained by add_entity :601-627,
 * 690-730 and remove_entity :806-831, 886-893).  unique_sections_with_lights only ever holds stale extra members (sections without lights add
 * nothing), so the query is restated on the section membership itself.  type_flag: RO_F_LIGHT_*.  Ids in ascending order; returns their number. */
uint32_t ro_visible_lights(ro_world *w, const ro_camera *cam, uint32_t type_flag, uint32_t cap, uint32_t *ids_out) {
    const float r = cam->far_draw;
    w->light_box = (ro_aabb){ cam->pos[0] - r, cam->pos[0] + r, cam->pos[1] - r, cam->pos[1] + r, cam->pos[2] - r, cam->pos[2] + r };
    ro_aabb lb = { rmax(cam->pos[0] - r, 0.0f), cam->pos[0] + r, rmax(cam->pos[1] - r, 0.0f), cam->pos[1] + r, rmax(cam->pos[2] - r, 0.0f), cam->pos[2] + r };   /* generate_original_culling_aabb */
    u64set_clear(&w->light_vis);
    find_visible_world_ids(w, 2, lb);
    u64set_normalize(&w->light_vis);
    u32vec out = { 0 };
    uint32_t pass = ++w->pass_id;
    for (uint32_t i = 0; i < w->light_vis.n; i++) {
        int32_t ci = cell_find(w, w->light_vis.v[i]);
        if (ci < 0) continue;
        const cell_t *c = &w->cells[ci];
        for (int part = 0; part < 2; part++) { const u32set *set = part ? &c->stat : &c->local;
            for (uint32_t k = 0; k < set->n; k++) { const ent_t *e = &w->ents[set->v[k]]; if (e->alive && (e->flags & type_flag)) u32vec_push(&out, set->v[k]); } }
        for (uint32_t k = 0; k < c->shared.n; k++) {
            shared_t *sh = &w->shared[c->shared.v[k]];
            if (!sh->used || sh->stamp == pass) continue;                     /* processed_shared_sections (:460, 471) */
            sh->stamp = pass;
            for (int part = 0; part < 2; part++) { const u32set *set = part ? &sh->stat : &sh->ents;
                for (uint32_t q = 0; q < set->n; q++) { const ent_t *e = &w->ents[set->v[q]]; if (e->alive && (e->flags & type_flag)) u32vec_push(&out, set->v[q]); } }
        }
    }
    qsort(out.v, out.n, sizeof(uint32_t), cmp_u32);
    uint32_t n = 0;
    for (uint32_t i = 0; i < out.n; i++) { if (i && out.v[i] == out.v[i - 1]) continue; if (ids_out && n < cap) ids_out[n] = out.v[i]; n++; }
    free(out.v);
    return n;
}

/* ------------------------------------------------------------------------------------------
 * RenderFlow: static cache, active sort, append, upload (flows/render_flow.rs:401-410)
 * ---------------------------------------------------------------------------------------- */
typedef struct { uint32_t model_index, render_system, sortable, id; float mat[16]; } inst_t;
typedef struct { inst_t *v; size_t n, cap; } instvec;
static void iv_push(instvec *iv, uint32_t m, uint32_t rs, uint32_t so, uint32_t id, const float *mat) {
    if (iv->n == iv->cap) { iv->cap = iv->cap ? iv->cap * 2 : 1024; iv->v = (inst_t *)realloc(iv->v, iv->cap * sizeof(inst_t)); }
    inst_t *t = &iv->v[iv->n++]; t->model_index = m; t->render_system = rs; t->sortable = so; t->id = id; memcpy(t->mat, mat, sizeof t->mat);
}
static int cmp_inst(const void *a, const void *b) {
    const inst_t *x = (const inst_t *)a, *y = (const inst_t *)b;
    if (x->model_index != y->model_index) return x->model_index < y->model_index ? -1 : 1;
    if (x->render_system != y->render_system) return x->render_system < y->render_system ? -1 : 1;
    if (x->sortable != y->sortable) return x->sortable < y->sortable ? -1 : 1;
    if (x->id != y->id) return x->id < y->id ? -1 : 1;
    return 0;
}

static cache_t *cache_for(ro_world *w, uint64_t key, int create) {
    int32_t idx = km_get(&w->cachemap, key);
    if (idx >= 0) return &w->caches[idx];
    if (!create) return NULL;
    if (w->ncaches == w->caches_cap) { w->caches_cap = w->caches_cap ? w->caches_cap * 2 : 256; w->caches = (cache_t *)realloc(w->caches, (size_t)w->caches_cap * sizeof(cache_t)); }
    cache_t *c = &w->caches[w->ncaches]; memset(c, 0, sizeof *c); c->key = key;
    km_put(&w->cachemap, key, (int32_t)w->ncaches++);
    return c;
}
static void cache_push(cache_t *c, const ent_t *e, uint32_t id) {
    if (c->n == c->cap) { c->cap = c->cap ? c->cap * 2 : 2; c->e = (cache_ent_t *)realloc(c->e, c->cap * sizeof(cache_ent_t)); }
    cache_ent_t *t = &c->e[c->n++]; t->id = id; t->model_index = e->model_index; t->render_system = e->render_system; t->sortable = e->sortable; memcpy(t->mat, e->mat, sizeof t->mat);
}

/* sort_world_section_static_entities (render_flow.rs:549-594): rebuild the cached static data of
 * every unique section whose static membership changed.  The distance test of
 * sort_unique_world_sections (:749-754) runs with the camera of THIS frame; a section that fails
 * it is cached empty until its static set changes again.  The cache is a SNAPSHOT (ids, model ids and the 64
 * matrix bytes): changes the logic phase makes to static entities never reach it, because changed_static_unique is
 * cleared before the next render (pipeline.rs:271). */
static void rebuild_static_cache(ro_world *w, const ro_camera *cam) {
    if (w->changed_static_unique.n == 0) return;
    u64set_normalize(&w->changed_static_unique);
    uint32_t pass = ++w->pass_id;
    for (uint32_t i = 0; i < w->changed_static_unique.n; i++) {
        uint64_t key = w->changed_static_unique.v[i];
        cache_t *c = cache_for(w, key, 1);
        c->n = 0;
        int32_t ci = cell_find(w, key);
        if (ci < 0) continue;                                        /* sort_unique returns None: empty data cached */
        const cell_t *cell = &w->cells[ci];
        float d = ro_distance_to_aabb(cell->aabb, cam->pos);
        if (d < cam->far_draw)
            for (uint32_t k = 0; k < cell->stat.n; k++) { const ent_t *e = &w->ents[cell->stat.v[k]]; if (e->alive) cache_push(c, e, cell->stat.v[k]); }
        for (uint32_t s = 0; s < cell->shared.n; s++) {             /* sort_shared_world_sections(is_static = true) (:808-866) */
            shared_t *sh = &w->shared[cell->shared.v[s]];
            if (sh->stamp == pass) continue;
            sh->stamp = pass;
            float d2 = ro_distance_to_aabb(sh->aabb, cam->pos);
            if (d2 < cam->far_draw)
                for (uint32_t k = 0; k < sh->stat.n; k++) { const ent_t *e = &w->ents[sh->stat.v[k]]; if (e->alive) cache_push(c, e, sh->stat.v[k]); }
        }
    }
}

/* ------------------------------------------------------------------------------------------
 * TimeTakeHistory::apply_to_function (helper_things/cpu_usage_reducer.rs:59-93): the reference's adaptive "one thread for a time budget, then
 * every thread" schedule of its per-section loops (render_flow.rs:649, logic_flow.rs:356).  The budget is an exponentially weighted mean of
 * the single-thread times of the last five calls (alpha = 0.6, :5-21), capped at 10 % of the last total (:24, 104-114) -- and end_frame
 * stores 1000 us for any call that took measurable time (:117-130), so from the second call on the budget is at most 100 us.  The parallel
 * part is par_chunks(1): one section per task; results are merged under a lock per section, as the reference's Mutex-guarded appends.
 * ---------------------------------------------------------------------------------------- */
static double wall_us(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return (double)t.tv_sec * 1e6 + (double)t.tv_nsec * 1e-3; }
static void tth_init(tth_t *h) { for (int k = 0; k < 5; k++) h->one_thread[k] = 16000.0f * 0.10f; h->total_us = 16000.0f; }
static float tth_allowed(const tth_t *h) {
    static const float coef[5] = { 0.6f, 0.6f * 0.4f, 0.6f * 0.4f * 0.4f, 0.6f * 0.4f * 0.4f * 0.4f, 0.6f * 0.4f * 0.4f * 0.4f * 0.4f };
    float s = 0.0f; for (int k = 0; k < 5; k++) s += h->one_thread[k] * coef[k];
    const float cap = h->total_us * 0.10f;
    return s < cap ? s : cap;
}
typedef void (*tth_body)(void *ctx, uint32_t i, int parallel_phase);
static void tth_apply(tth_t *h, int nthreads, uint32_t n, tth_body f, void *ctx) {
    const double frame0 = wall_us();                                  /* start_frame */
    uint32_t done = 0;
    const double t0 = wall_us(); const float maxt = tth_allowed(h);
    while ((float)floor(wall_us() - t0) < maxt) {                     /* elapsed().as_micros() as f32 */
        if (done >= n) return;                                        /* (returns without touching the history, like the reference) */
        f(ctx, done, 0); done++;
    }
    for (int k = 4; k > 0; k--) h->one_thread[k] = h->one_thread[k - 1];   /* rotate_right(1) + overwrite [0] */
    h->one_thread[0] = (float)floor(wall_us() - t0);
    const long rest = (long)n - (long)done;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 1) num_threads(nthreads) if (nthreads > 1 && rest > 1)
#endif
    for (long i = 0; i < rest; i++) f(ctx, done + (uint32_t)i, nthreads > 1);
    h->total_us = floor(wall_us() - frame0) == 0.0 ? 0.0f : 1000.0f;    /* end_frame (:117-130) */
}

/* one active visible section of sort_world_section_active_entities (sort_world_chunk -> sort_unique_world_sections + sort_shared_world_sections,
 * render_flow.rs:718-866) */
typedef struct { ro_world *w; const ro_camera *cam; const uint64_t *vec; instvec *iv; uint32_t pass; } render_ctx;
static void render_active_section(void *vctx, uint32_t i, int parallel_phase) {
    render_ctx *R = (render_ctx *)vctx; ro_world *w = R->w; const ro_camera *cam = R->cam;
    int32_t ci = cell_find(w, R->vec[i]);
    if (ci < 0 || w->cells[ci].is_static_section) return;         /* is_section_active (:397-401) */
    const cell_t *cell = &w->cells[ci];
    instvec local = { 0 }, *out = parallel_phase ? &local : R->iv;
    float d = ro_distance_to_aabb(cell->aabb, cam->pos);
    if (d < cam->far_draw)                                           /* :754 */
        for (uint32_t k = 0; k < cell->local.n; k++) {
            const ent_t *e = &w->ents[cell->local.v[k]];
            if (!e->alive) continue;
            iv_push(out, lod_model_index(w->custom_lod, w->n_custom_lod, cam, e->model_index, e->render_system, d), e->render_system, e->sortable, cell->local.v[k], e->mat);
        }
    for (uint32_t s = 0; s < cell->shared.n; s++) {
        shared_t *sh = &w->shared[cell->shared.v[s]];
        int seen;
#ifdef _OPENMP
#pragma omp critical(ro_processed_world_sections)
#endif
        { seen = sh->stamp == R->pass; sh->stamp = R->pass; }         /* processed_world_sections (:811), a Mutex<HashSet> in the reference */
        if (seen) continue;
        float d2 = ro_distance_to_aabb(sh->aabb, cam->pos);
        if (d2 < cam->far_draw)                                      /* :822 */
            for (uint32_t k = 0; k < sh->ents.n; k++) {
                const ent_t *e = &w->ents[sh->ents.v[k]];
                if (!e->alive) continue;
                iv_push(out, lod_model_index(w->custom_lod, w->n_custom_lod, cam, e->model_index, e->render_system, d2), e->render_system, e->sortable, sh->ents.v[k], e->mat);
            }
    }
    if (parallel_phase) {                                            /* append_written_information under sorted_data.lock() (:621-623) */
#ifdef _OPENMP
#pragma omp critical(ro_sorted_data)
#endif
        for (size_t k = 0; k < local.n; k++) iv_push(R->iv, local.v[k].model_index, local.v[k].render_system, local.v[k].sortable, local.v[k].id, local.v[k].mat);
        free(local.v);
    }
}

uint32_t ro_frame_render(ro_world *w, const ro_camera *cam, int emit_duplicates, uint32_t cap, uint32_t *ids, float *mats,
                         uint32_t gcap, ro_group *groups, uint32_t *n_groups) {
    instvec iv = { 0 };
    rebuild_static_cache(w, cam);
    /* the vec the flows iterate: visible_sections_vec (duplicates kept) or its set */
    const uint64_t *vec = emit_duplicates ? w->vis_vec : w->vis_map.v;
    uint32_t nvec = emit_duplicates ? w->vis_n : w->vis_map.n;
    /* extract_static_data (:458-542) */
    for (uint32_t i = 0; i < nvec; i++) {
        cache_t *c = cache_for(w, vec[i], 0);
        if (!c) continue;
        int32_t ci = cell_find(w, vec[i]);
        float d = ci >= 0 ? ro_distance_to_aabb(w->cells[ci].aabb, cam->pos) : 0.0f;
        if (d > cam->far_draw) continue;                             /* :489 */
        for (uint32_t k = 0; k < c->n; k++)
            iv_push(&iv, lod_model_index(w->custom_lod, w->n_custom_lod, cam, c->e[k].model_index, c->e[k].render_system, d), c->e[k].render_system, c->e[k].sortable, c->e[k].id, c->e[k].mat);
    }
    /* sort_world_section_active_entities (:603-653): the sections through TimeTakeHistory::apply_to_function (:649) */
    {
        render_ctx rc = { w, cam, vec, &iv, ++w->pass_id };
        tth_apply(&w->tth_render, w->nthreads, nvec, render_active_section, &rc);
    }
    /* append + upload (:661-713, :939-992): groups back to back; group order here = ascending
     * (model, render system, sortable) as a deterministic stand-in for HashMap iteration */
    qsort(iv.v, iv.n, sizeof(inst_t), cmp_inst);
    uint32_t ng = 0, total = 0;
    for (size_t i = 0; i < iv.n; i++) {
        const inst_t *t = &iv.v[i];
        int newg = (i == 0) || t->model_index != iv.v[i - 1].model_index || t->render_system != iv.v[i - 1].render_system || t->sortable != iv.v[i - 1].sortable;
        if (newg) {
            if (ng < gcap) { groups[ng].model_index = t->model_index; groups[ng].render_system = t->render_system; groups[ng].sortable = t->sortable; groups[ng].begin = total; groups[ng].count = 0; }
            ng++;
        }
        if (ng - 1 < gcap) groups[ng - 1].count++;
        if (total < cap) {
            if (ids) ids[total] = t->id;
            if (mats) memcpy(mats + (size_t)total * 16, t->mat, 16 * sizeof(float));   /* specify_type_ids!: the 64 raw bytes of TransformationMatrix */
        }
        total++;
    }
    if (n_groups) *n_groups = ng;
    free(iv.v);
    return total;
}

/* ------------------------------------------------------------------------------------------
 * LogicFlow::update_positions / apply_kinematics (flows/logic_flow.rs:308-448) and
 * apply_change / update_aabb_after_kinematic_change (helper_things/entity_change_helpers.rs)
 * ---------------------------------------------------------------------------------------- */
typedef struct {
    uint32_t id;
    uint8_t vel_set, pos_set, rotvel_set, rot_set;
    float vel[3], pos[3], rotvel[4], rot[4];
} change_t;
typedef struct { change_t *v; size_t n, cap; } changevec;

static void apply_kinematics_one(const ro_world *w, uint32_t id, float dt, changevec *cv) {
    const ent_t *e = &w->ents[id];
    if (!e->alive) return;
    change_t ch; memset(&ch, 0, sizeof ch); ch.id = id;
    if (e->flags & RO_F_HAS_VEL) {
        if (e->flags & RO_F_HAS_ACC) {
            if (ro_norm3(e->acc[0], e->acc[1], e->acc[2]) != 0.0f) {          /* :384 */
                for (int k = 0; k < 3; k++) ch.vel[k] = e->vel[k] + e->acc[k] * dt;   /* velocity += acceleration * dt */
                ch.vel_set = 1;
            }
        }
        /* position uses the velocity still stored in the ECS: the change request is deferred (:392-399) */
        if (ro_norm3(e->vel[0], e->vel[1], e->vel[2]) != 0.0f) {
            for (int k = 0; k < 3; k++) ch.pos[k] = e->pos[k] + e->vel[k] * dt;
            ch.pos_set = 1;
        }
    }
    if (e->flags & RO_F_HAS_ROTVEL) {
        if (e->flags & RO_F_HAS_ROTACC) {
            if (e->rotacc[3] != 0.0f) {                                         /* :418 */
                /* AccelerationRotation * dt -> new(axis*dt, a*dt) normalises; += adds, then normalises
                 * (exports/movement_components.rs:231-250, 277-298) */
                float sc[3] = { e->rotacc[0] * dt, e->rotacc[1] * dt, e->rotacc[2] * dt }, nrm[3];
                normalize3(sc, nrm);
                float sum[3] = { e->rotvel[0] + nrm[0], e->rotvel[1] + nrm[1], e->rotvel[2] + nrm[2] };
                float ang = e->rotvel[3] + e->rotacc[3] * dt;
                normalize3(sum, ch.rotvel); ch.rotvel[3] = ang; ch.rotvel_set = 1;
            }
        }
        if (e->rotvel[3] != 0.0f) {                                             /* :429 */
            float sc[3] = { e->rotvel[0] * dt, e->rotvel[1] * dt, e->rotvel[2] * dt }, nrm[3];
            normalize3(sc, nrm);
            float sum[3] = { e->rot[0] + nrm[0], e->rot[1] + nrm[1], e->rot[2] + nrm[2] };
            float ang = e->rot[3] + e->rotvel[3] * dt;
            normalize3(sum, ch.rot); ch.rot[3] = ang; ch.rot_set = 1;
        }
    }
    if (ch.vel_set || ch.pos_set || ch.rotvel_set || ch.rot_set) {
        if (cv->n == cv->cap) { cv->cap = cv->cap ? cv->cap * 2 : 256; cv->v = (change_t *)realloc(cv->v, cv->cap * sizeof(change_t)); }
        cv->v[cv->n++] = ch;
    }
}

/* update_aabb_after_kinematic_change (entity_change_helpers.rs:217-262) + update_entity_in_tree (:325-351): translation-only
 * entities first, then the kinematic ones, each set in ascending EntityId (stand-in for hash order).  Returns the number of
 * entities that left the world without OutOfBoundsLogic (deleted, :336-349); their ids go to oob_ids[have..]. */
static uint32_t update_aabb_after_kinematic_change(ro_world *w, const u32set *only_translation_p, const u32set *kinematics_p, uint32_t *oob_ids, uint32_t cap, uint32_t have) {
    const u32set only_translation = *only_translation_p, kinematics = *kinematics_p;
    uint32_t noob = have;
    for (uint32_t i = 0; i < only_translation.n; i++) {
        uint32_t id = only_translation.v[i]; ent_t *e = &w->ents[id];
        ro_aabb a = e->original;                                      /* OriginalAABB translated: rotation and scale ignored (:223-224) */
        a.xmin += e->pos[0]; a.xmax += e->pos[0]; a.ymin += e->pos[1]; a.ymax += e->pos[1]; a.zmin += e->pos[2]; a.zmax += e->pos[2];
        e->mat[12] = e->pos[0]; e->mat[13] = e->pos[1]; e->mat[14] = e->pos[2];   /* column 3 xyz overwritten (:228-233) */
        e->aabb = a;
        /* update_entity_in_tree (:325-351) */
        if (ro_tree_add(w, id, a, (e->flags & RO_F_OOB_LOGIC) != 0, 0) != 0) { if (noob < cap && oob_ids) oob_ids[noob] = id; noob++; w->ents[id].alive = 0; }
    }
    for (uint32_t i = 0; i < kinematics.n; i++) {
        uint32_t id = kinematics.v[i]; ent_t *e = &w->ents[id];
        /* Rotation/Scale default when absent (:245-246); all three factors are always applied (:248-250) */
        ro_trs_matrix(e->pos, 1, e->rot, e->rot[3], 1, e->scale, e->mat);
        e->aabb = ro_apply_transformation(e->original, e->mat);
        if (ro_tree_add(w, id, e->aabb, (e->flags & RO_F_OOB_LOGIC) != 0, 0) != 0) { if (noob < cap && oob_ids) oob_ids[noob] = id; noob++; w->ents[id].alive = 0; }
    }
    return noob - have;
}

/* one active visible section of update_positions (updated_kinematics_fn, logic_flow.rs:321-354); the change requests go to one list under a lock
 * (expected_frame_changes.lock().push, :403, :437) */
typedef struct { ro_world *w; float dt; changevec *cv; uint32_t pass; } tick_ctx;
static void tick_active_section(void *vctx, uint32_t i, int parallel_phase) {
    tick_ctx *T = (tick_ctx *)vctx; ro_world *w = T->w;
    int32_t ci = cell_find(w, w->vis_map.v[i]);
    if (ci < 0 || w->cells[ci].is_static_section) return;         /* :216-223 */
    const cell_t *cell = &w->cells[ci];
    changevec local = { 0 }, *out = parallel_phase ? &local : T->cv;
    for (uint32_t k = 0; k < cell->local.n; k++) apply_kinematics_one(w, cell->local.v[k], T->dt, out);
    for (uint32_t s = 0; s < cell->shared.n; s++) {
        shared_t *sh = &w->shared[cell->shared.v[s]];
        int seen;
#ifdef _OPENMP
#pragma omp critical(ro_processed_world_sections)
#endif
        { seen = sh->stamp == T->pass; sh->stamp = T->pass; }
        if (seen) continue;
        if (ro_logic_aabb_in_view(w->lookahead, w->campos, sh->aabb) || ro_frustum_aabb_visible(w->planes, sh->aabb))   /* :338-339 */
            for (uint32_t k = 0; k < sh->ents.n; k++) apply_kinematics_one(w, sh->ents.v[k], T->dt, out);
    }
    if (parallel_phase) {
#ifdef _OPENMP
#pragma omp critical(ro_expected_frame_changes)
#endif
        for (size_t k = 0; k < local.n; k++) { if (T->cv->n == T->cv->cap) { T->cv->cap = T->cv->cap ? T->cv->cap * 2 : 64; T->cv->v = (change_t *)realloc(T->cv->v, T->cv->cap * sizeof(change_t)); } T->cv->v[T->cv->n++] = local.v[k]; }
        free(local.v);
    }
}

uint32_t ro_frame_tick(ro_world *w, const ro_camera *cam, float dt, uint32_t cap, uint32_t *oob_ids, uint32_t *n_oob) {
    (void)cam;
    changevec cv = { 0 };
    uint32_t noob = 0;
    /* reset_has_changed_component (:776-801) */
    for (uint32_t i = 0; i < w->marked.n; i++) w->ents[w->marked.v[i]].flags &= ~(RO_F_HAS_MOVED | RO_F_HAS_ROTATED);
    w->marked.n = 0;
    /* update_positions over the active visible sections.  Set semantics: a section visited twice
     * (duplicate in visible_sections_vec) yields an identical, idempotent change request. */
    {   /* the sections through TimeTakeHistory::apply_to_function (logic_flow.rs:356) */
        tick_ctx tc = { w, dt, &cv, ++w->pass_id };
        tth_apply(&w->tth_positions, w->nthreads, w->vis_map.n, tick_active_section, &tc);
    }
    /* find_always_execute_entities (:803-836) + apply_kinematics(always_execute_entities) (:357) */
    for (uint32_t i = 0; i < w->always_exec.n; i++) {
        uint32_t id = w->always_exec.v[i];
        const ent_t *e = &w->ents[id];
        if (!e->alive || !(e->flags & RO_F_ALWAYS_EXEC) || e->lookup == 0) continue;
        int seen = 0;
        if (e->lookup == 1) seen = u64set_has(&w->vis_map, e->ukey);
        else { const shared_t *sh = &w->shared[e->shared]; for (int k = 0; k < sh->nkeys && !seen; k++) seen = u64set_has(&w->vis_map, sh->keys[k]); }
        if (!seen) apply_kinematics_one(w, id, dt, &cv);
    }
    /* apply_change: apply_entity_change_requests (:276-323).  One entity issues up to two
     * ModifyRequests: {Velocity, Position, HasMoved} then {VelocityRotation, Rotation, HasRotated}. */
    u32set only_translation = { 0 }, kinematics = { 0 };
    for (size_t i = 0; i < cv.n; i++) {
        change_t *c = &cv.v[i]; ent_t *e = &w->ents[c->id];
        int was_marked = (e->flags & (RO_F_HAS_MOVED | RO_F_HAS_ROTATED)) != 0;
        if (c->vel_set) memcpy(e->vel, c->vel, sizeof e->vel);
        if (c->pos_set) { memcpy(e->pos, c->pos, sizeof e->pos); e->flags |= RO_F_HAS_MOVED; }
        if (c->rotvel_set) memcpy(e->rotvel, c->rotvel, sizeof e->rotvel);
        if (c->rot_set) { memcpy(e->rot, c->rot, sizeof e->rot); e->flags |= RO_F_HAS_ROTATED; }
        if (!was_marked && (e->flags & (RO_F_HAS_MOVED | RO_F_HAS_ROTATED))) u32vec_push(&w->marked, c->id);
        if (c->pos_set) { if (!u32set_has(&kinematics, c->id)) u32set_add(&only_translation, c->id); }
        if (c->rot_set) { u32set_add(&kinematics, c->id); u32set_del(&only_translation, c->id); }
    }
    noob += update_aabb_after_kinematic_change(w, &only_translation, &kinematics, oob_ids, cap, noob);
    uint32_t napplied = only_translation.n + kinematics.n;
    u32set_free(&only_translation); u32set_free(&kinematics); free(cv.v);
    ro_end_of_changes(w);                                            /* entity_change_helpers.rs:188 */
    u64set_clear(&w->changed_static_unique);                         /* Pipeline::execute -> clear_changed_static_unique (pipeline.rs:271) */
    if (n_oob) *n_oob = noob;
    return napplied;
}

/* =======================================================================================
 * Collision broad phase: LogicFlow::handle_collisions (flows/logic_flow.rs:452-651) over
 * BoundingBoxTree::find_related_entities (bounding_box_tree_v2.rs:950-1048)
 * ======================================================================================= */
/* a strictly above d: parent = level + 1, index / 2 (UniqueWorldSectionId::higher_level_world_section :46-64) */
static int key_is_ancestor(uint64_t a, uint64_t d) {
    uint32_t la = KEY_LEVEL(a), ld = KEY_LEVEL(d);
    if (la <= ld || la - ld > 16) return 0;
    uint32_t s = la - ld;
    return (KEY_X(d) >> s) == KEY_X(a) && (KEY_Z(d) >> s) == KEY_Z(a) && (KEY_Y(d) >> s) == KEY_Y(a);
}
/* related_world_sections[key] (:334): register_created_section_with_others (:1219-1296) links a created section with every
 * existing child section (all levels below) and every existing parent section (all levels above), both ways; removing a
 * section unlinks both ways (:925-936).  So the list of a section is: the existing sections that are its ancestors or
 * descendants.  Stated by that definition (the reference walks 8^level child ids per creation). */
uint32_t ro_related_sections(const ro_world *w, uint64_t key, uint32_t cap, uint64_t *out) {
    uint32_t n = 0;
    for (uint32_t i = 0; i < w->ncells_alloc; i++) {
        const cell_t *c = &w->cells[i];
        if (!c->used) continue;
        if (key_is_ancestor(c->key, key) || key_is_ancestor(key, c->key)) { if (n < cap && out) out[n] = c->key; n++; }
    }
    return n;
}
typedef struct { int32_t *cells; uint32_t ncells; int32_t *shared; uint32_t nshared; } related_t;
/* find_related_entities_internal (:966-1048): worklist over related_world_sections with a processed set; every section popped
 * contributes its local_entities, and each of its shared sections once (both branches of the in-view test push the same result) */
static void find_related(const ro_world *w, uint64_t start, related_t *r, uint8_t *cell_seen, uint8_t *shared_seen) {
    uint64_t *stack = (uint64_t *)malloc(sizeof(uint64_t) * 64); uint32_t sn = 0, scap = 64;
    r->ncells = r->nshared = 0;
    stack[sn++] = start;
    uint64_t *rel = (uint64_t *)malloc(sizeof(uint64_t) * ((size_t)w->ncells_live + 1));
    while (sn) {
        uint64_t x = stack[--sn];
        int32_t ci = cell_find(w, x);
        if (ci < 0 || cell_seen[ci]) continue;                            /* processed_world_sections.insert (:980-983) */
        cell_seen[ci] = 1; r->cells[r->ncells++] = ci;
        const cell_t *cell = &w->cells[ci];
        for (uint32_t k = 0; k < cell->shared.n; k++) {
            int32_t si = (int32_t)cell->shared.v[k];
            if (shared_seen[si]) continue;                                /* processed_shared_sections (:1003) */
            shared_seen[si] = 1; r->shared[r->nshared++] = si;
        }
        uint32_t nr = ro_related_sections(w, x, w->ncells_live, rel);     /* :1044 */
        for (uint32_t k = 0; k < nr; k++) { if (sn == scap) { scap *= 2; stack = (uint64_t *)realloc(stack, sizeof(uint64_t) * scap); } stack[sn++] = rel[k]; }
    }
    for (uint32_t i = 0; i < r->ncells; i++) cell_seen[r->cells[i]] = 0;
    for (uint32_t i = 0; i < r->nshared; i++) shared_seen[r->shared[i]] = 0;
    free(stack); free(rel);
}
/* the find_related_entities of one section, for the known-answer test of the reference (:2220-2303): unique section keys, then
 * for every shared section its number of keys followed by the keys */
uint32_t ro_find_related(const ro_world *w, uint64_t key, uint32_t cap, uint64_t *unique_keys, uint32_t *n_unique, uint32_t shared_cap, uint64_t *shared_keys, uint32_t *n_shared) {
    related_t r; r.cells = (int32_t *)malloc(sizeof(int32_t) * (w->ncells_alloc + 1)); r.shared = (int32_t *)malloc(sizeof(int32_t) * (w->nshared_alloc + 1));
    uint8_t *cs = (uint8_t *)calloc(w->ncells_alloc + 1, 1), *ss = (uint8_t *)calloc(w->nshared_alloc + 1, 1);
    find_related(w, key, &r, cs, ss);
    for (uint32_t i = 0; i < r.ncells && i < cap; i++) unique_keys[i] = w->cells[r.cells[i]].key;
    uint32_t o = 0;
    for (uint32_t i = 0; i < r.nshared; i++) {
        const shared_t *sh = &w->shared[r.shared[i]];
        if (o + 9 > shared_cap) break;
        shared_keys[o++] = (uint64_t)sh->nkeys;
        for (int k = 0; k < sh->nkeys; k++) shared_keys[o++] = sh->keys[k];
    }
    *n_unique = r.ncells; *n_shared = r.nshared;
    free(r.cells); free(r.shared); free(cs); free(ss);
    return r.ncells + r.nshared;
}
static int aabb_intersect(ro_aabb a, ro_aabb b) {                         /* StaticAABB::intersect (aabb.rs:68-73), closed intervals (range.rs:71) */
    return a.xmin <= b.xmax && a.xmax >= b.xmin && a.ymin <= b.ymax && a.ymax >= b.ymin && a.zmin <= b.zmax && a.zmax >= b.zmin;
}
typedef struct { uint64_t key; u32vec moved; } relevant_t;
/* One handle_collisions pass with the state of this frame (after ro_frame_cull, BEFORE ro_frame_tick applies the kinematic
 * changes: update_positions only queues them, so the collision tests read this frame's StaticAABBs, logic_flow.rs:230,243).
 * Output: the (this_entity, other_entity) arguments of every collision-logic invocation (apply_collision_only_to_self), as a
 * multiset -- the reference's order depends on thread timing.  The one place where that order changes the RESULT is the
 * Shared arm of the section map (:488-498): the moved entity that creates a section's entry through a Shared lookup is not
 * pushed into it.  moved_entities is taken in ascending EntityId (a section listed twice in visible_sections_vec pushes its
 * entities twice, in a row), the user entity last (:236-240), which fixes that choice.
 * Returns the total number of invocations; the first cap are written to pairs (2 ids each). */
uint32_t ro_frame_collide(ro_world *w, const ro_camera *cam, uint32_t cap, uint32_t *pairs) {
    u32vec moved = { 0 };
    int32_t user = -1;
    /* update_positions (:308-358): apply_kinematics pushes every processed entity that carries Velocity or VelocityRotation
     * (entity_moved is set by the component's presence, :380-441) and CanCauseCollisions (:443-446) */
#define MOVES(e) ((e)->alive && ((e)->flags & (RO_F_HAS_VEL | RO_F_HAS_ROTVEL)) && ((e)->flags & RO_F_CAN_COLLIDE))
    uint32_t pass = ++w->pass_id;
    for (uint32_t i = 0; i < w->vis_n; i++) {                              /* active_world_sections keeps the duplicates of the vec (:216-223) */
        int32_t ci = cell_find(w, w->vis_vec[i]);
        if (ci < 0 || w->cells[ci].is_static_section) continue;
        const cell_t *cell = &w->cells[ci];
        for (uint32_t k = 0; k < cell->local.n; k++) if (MOVES(&w->ents[cell->local.v[k]])) u32vec_push(&moved, cell->local.v[k]);
        for (uint32_t s = 0; s < cell->shared.n; s++) {
            shared_t *sh = &w->shared[cell->shared.v[s]];
            if (sh->stamp == pass) continue;
            sh->stamp = pass;
            if (ro_logic_aabb_in_view(w->lookahead, w->campos, sh->aabb) || ro_frustum_aabb_visible(w->planes, sh->aabb))
                for (uint32_t k = 0; k < sh->ents.n; k++) if (MOVES(&w->ents[sh->ents.v[k]])) u32vec_push(&moved, sh->ents.v[k]);
        }
    }
    for (uint32_t i = 0; i < w->always_exec.n; i++) {                      /* always_execute_entities (:357, 803-836) */
        uint32_t id = w->always_exec.v[i];
        const ent_t *e = &w->ents[id];
        if (!e->alive || !(e->flags & RO_F_ALWAYS_EXEC) || e->lookup == 0) continue;
        int seen = 0;
        if (e->lookup == 1) seen = u64set_has(&w->vis_map, e->ukey);
        else { const shared_t *sh = &w->shared[e->shared]; for (int k = 0; k < sh->nkeys && !seen; k++) seen = u64set_has(&w->vis_map, sh->keys[k]); }
        if (!seen && MOVES(e)) u32vec_push(&moved, id);
    }
    if (moved.n) qsort(moved.v, moved.n, sizeof(uint32_t), cmp_u32);
    for (uint32_t id = 0; id < w->ents_cap; id++) if (w->ents[id].alive && (w->ents[id].flags & RO_F_USER)) { user = (int32_t)id; break; }
    if (user >= 0) u32vec_push(&moved, (uint32_t)user);                    /* UserAlwaysCausesCollisions (pipeline.rs:136, logic_flow.rs:236-240) */
    uint8_t *is_moved = (uint8_t *)calloc(w->ents_cap + 1, 1);            /* moved_entities_map (:476) */
    for (uint32_t i = 0; i < moved.n; i++) is_moved[moved.v[i]] = 1;
    /* relevant_world_sections (:479-514), entries in creation order */
    relevant_t *rel = NULL; uint32_t nrel = 0, rel_cap = 0; kmap relmap; km_init(&relmap, 256);
    for (uint32_t i = 0; i < moved.n; i++) {
        const ent_t *e = &w->ents[moved.v[i]];
        uint64_t keys[8]; int nk = 0, is_shared = e->lookup == 2;
        if (e->lookup == 1) { keys[0] = e->ukey; nk = 1; }
        else if (e->lookup == 2) { const shared_t *sh = &w->shared[e->shared]; nk = sh->nkeys; memcpy(keys, sh->keys, sizeof(uint64_t) * (size_t)nk); }
        for (int k = 0; k < nk; k++) {
            int32_t ri = km_get(&relmap, keys[k]);
            if (ri >= 0) { u32vec_push(&rel[ri].moved, moved.v[i]); continue; }
            if (nrel == rel_cap) { rel_cap = rel_cap ? rel_cap * 2 : 64; rel = (relevant_t *)realloc(rel, rel_cap * sizeof(relevant_t)); }
            memset(&rel[nrel], 0, sizeof(relevant_t)); rel[nrel].key = keys[k];
            if (!is_shared) u32vec_push(&rel[nrel].moved, moved.v[i]);    /* Unique: vec![*entity] (:509); Shared: Vec::new() (:494) */
            km_put(&relmap, keys[k], (int32_t)nrel++);
        }
    }
    related_t r; r.cells = (int32_t *)malloc(sizeof(int32_t) * (w->ncells_alloc + 1)); r.shared = (int32_t *)malloc(sizeof(int32_t) * (w->nshared_alloc + 1));
    uint8_t *cs = (uint8_t *)calloc(w->ncells_alloc + 1, 1), *ss = (uint8_t *)calloc(w->nshared_alloc + 1, 1);
    u32vec self = { 0 }, both = { 0 };
    uint32_t total = 0;
#define EMIT(a, b) do { if (total < cap && pairs) { pairs[2 * total] = (a); pairs[2 * total + 1] = (b); } total++; } while (0)
    for (uint32_t ri = 0; ri < nrel; ri++) {
        find_related(w, rel[ri].key, &r, cs, ss);                          /* :546 */
        self.n = both.n = 0;
        for (uint32_t i = 0; i < r.ncells; i++) {
            const cell_t *c = &w->cells[r.cells[i]];
            if (ro_distance_to_aabb(c->aabb, cam->pos) > 200.0f) continue;  /* :553-558 */
            for (uint32_t k = 0; k < c->local.n; k++) { uint32_t o = c->local.v[k]; if (!w->ents[o].alive) continue; u32vec_push(is_moved[o] ? &self : &both, o); }
        }
        for (uint32_t i = 0; i < r.nshared; i++) {
            const shared_t *sh = &w->shared[r.shared[i]];
            if (ro_distance_to_aabb(sh->aabb, cam->pos) > 200.0f) continue; /* :561-566 */
            for (uint32_t k = 0; k < sh->ents.n; k++) { uint32_t o = sh->ents.v[k]; if (!w->ents[o].alive) continue; u32vec_push(is_moved[o] ? &self : &both, o); }
        }
        for (uint32_t m = 0; m < rel[ri].moved.n; m++) {                   /* collision_fn (:619-650) */
            uint32_t me = rel[ri].moved.v[m];
            ro_aabb a = w->ents[me].aabb;
            for (uint32_t k = 0; k < self.n; k++) { if (self.v[k] == me) continue; if (aabb_intersect(a, w->ents[self.v[k]].aabb)) EMIT(me, self.v[k]); }
            for (uint32_t k = 0; k < both.n; k++) if (aabb_intersect(a, w->ents[both.v[k]].aabb)) { EMIT(me, both.v[k]); EMIT(both.v[k], me); }
        }
    }
#undef EMIT
#undef MOVES
    for (uint32_t ri = 0; ri < nrel; ri++) free(rel[ri].moved.v);
    free(rel); km_free(&relmap); free(r.cells); free(r.shared); free(cs); free(ss); free(self.v); free(both.v); free(moved.v); free(is_moved);
    return total;
}

/* apply_change (helper_things/entity_change_helpers.rs:32-189) for the change kinds that touch this path, in list order:
 * ModifyRequest of one component (apply_entity_change_requests :276-323; a request of several components is the same as its
 * components one after another: the classification only accumulates), DeleteRequest (:156-172), MakeObjectStatic (:112-122),
 * WakeUpRequest (:123-133); then update_aabb_after_kinematic_change and end_of_changes (:186-188).
 * end_of_frame != 0 additionally clears changed_static_unique like Pipeline::execute does after the logic flow (pipeline.rs:271).
 * Deviation: a DeleteRequest also drops the entity from the translation-only set (the reference would unwrap a removed
 * component there and panic). */
uint32_t ro_apply_changes(ro_world *w, const ro_change *ch, uint32_t n, int end_of_frame, uint32_t cap, uint32_t *oob_ids, uint32_t *n_oob) {
    return ro_apply_changes_ex(w, ch, n, NULL, 0, end_of_frame, cap, oob_ids, n_oob);
}
uint32_t ro_apply_changes_ex(ro_world *w, const ro_change *ch, uint32_t n, const ro_entity_desc *added, uint32_t n_added, int end_of_frame, uint32_t cap, uint32_t *oob_ids, uint32_t *n_oob) {
    u32set only_translation = { 0 }, kinematics = { 0 }, deleted = { 0 };
    for (uint32_t i = 0; i < n; i++) {
        const ro_change *c = &ch[i];
        if (c->kind == RO_CHANGE_ADD_ENTITY) {                  /* AddEntity (entity_change_helpers.rs:48-107): inline, in list order; the entity leaves the three sets (:54-56) */
            if (!added || c->pad >= n_added) continue;
            const uint32_t nid = added[c->pad].id;
            u32set_del(&kinematics, nid); u32set_del(&only_translation, nid); u32set_del(&deleted, nid);
            (void)register_one(w, &added[c->pad]);
            continue;
        }
        if (c->entity_id >= w->ents_cap || !w->ents[c->entity_id].alive) continue;
        ent_t *e = &w->ents[c->entity_id];
        const uint32_t id = c->entity_id;
        switch (c->kind) {
        case RO_CHANGE_MODIFY: {
            if (u32set_has(&deleted, id)) break;
            int pos = 0, rot = 0, scl = 0;
            switch (c->component) {
            case 0: memcpy(e->pos, c->value, 12); pos = 1; break;
            case 1: normalize3(c->value, e->rot); e->rot[3] = c->value[3]; e->flags |= RO_F_HAS_ROT; rot = 1; break;
            case 2: memcpy(e->scale, c->value, 12); e->flags |= RO_F_HAS_SCALE; scl = 1; break;
            case 3: memcpy(e->vel, c->value, 12); e->flags |= RO_F_HAS_VEL; break;
            case 4: memcpy(e->acc, c->value, 12); e->flags |= RO_F_HAS_ACC; break;
            case 5: normalize3(c->value, e->rotvel); e->rotvel[3] = c->value[3]; e->flags |= RO_F_HAS_ROTVEL; break;
            case 6: normalize3(c->value, e->rotacc); e->rotacc[3] = c->value[3]; e->flags |= RO_F_HAS_ROTACC; break;
            default: break;
            }
            if (pos && !rot && !scl) { if (!u32set_has(&kinematics, id)) u32set_add(&only_translation, id); }
            else if (pos || rot || scl) { u32set_add(&kinematics, id); u32set_del(&only_translation, id); }
            break;
        }
        case RO_CHANGE_REMOVE_COMPONENT:                         /* RemoveComponent((entity, type)) -> ecs.remove_component_type_id_internal (entity_change_helpers.rs:151-154,
                                                                   objects/ecs.rs:523-556): the presence bit is cleared, nothing else happens; later reads see the default */
            if (!e->alive) break;
            switch (c->component) {
            case 1: e->flags &= ~RO_F_HAS_ROT; e->rot[0] = 1.0f; e->rot[1] = e->rot[2] = e->rot[3] = 0.0f; break;     /* Rotation::default (movement_components.rs:41-47) */
            case 2: e->flags &= ~RO_F_HAS_SCALE; e->scale[0] = e->scale[1] = e->scale[2] = 1.0f; break;                /* Scale::default (:49-55) */
            case 3: e->flags &= ~RO_F_HAS_VEL; break;
            case 4: e->flags &= ~RO_F_HAS_ACC; break;
            case 5: e->flags &= ~RO_F_HAS_ROTVEL; break;
            case 6: e->flags &= ~RO_F_HAS_ROTACC; break;
            default: break;
            }
            break;
        case RO_CHANGE_ADD_SORTABLE:                            /* AddSortableComponent -> ECS::write_sortable_component (objects/ecs.rs:202-205): the entity's sortable bucket */
            if (u32set_has(&deleted, id)) break;
            e->sortable = c->component; break;
        case RO_CHANGE_REMOVE_SORTABLE:                         /* RemoveSortableComponent -> the default sortable component (objects/ecs.rs:210-213) */
            if (u32set_has(&deleted, id)) break;
            e->sortable = 0; break;
        case RO_CHANGE_DELETE:
            ro_tree_remove(w, id);
            u32set_del(&kinematics, id); u32set_del(&only_translation, id); u32set_add(&deleted, id);
            e->alive = 0;
            break;
        case RO_CHANGE_MAKE_STATIC: case RO_CHANGE_WAKE_UP: {
            const int st = c->kind == RO_CHANGE_MAKE_STATIC;
            ro_tree_remove(w, id);
            e = &w->ents[id];
            if (ro_tree_add(w, id, e->aabb, (e->flags & RO_F_OOB_LOGIC) != 0, st) == 0) { e = &w->ents[id]; if (st) e->flags |= RO_F_STATIC; else e->flags &= ~RO_F_STATIC; }
            break;
        }
        default: break;
        }
    }
    uint32_t noob = update_aabb_after_kinematic_change(w, &only_translation, &kinematics, oob_ids, cap, 0);
    uint32_t napplied = only_translation.n + kinematics.n;
    u32set_free(&only_translation); u32set_free(&kinematics); u32set_free(&deleted);
    ro_end_of_changes(w);
    if (end_of_frame) u64set_clear(&w->changed_static_unique);
    if (n_oob) *n_oob = noob;
    return napplied;
}

/* ------------------------------------------------------------------------------------------
 * config 5: deferred lighting -- second_pass_frag.glsl main() and helpers, evaluated per pixel
 * in f32 over ALL lights in index order (no culling).
 * ---------------------------------------------------------------------------------------- */
static void v3_normalize(const float v[3], float o[3]) { float n = sqrtf((v[0] * v[0] + v[1] * v[1]) + v[2] * v[2]); o[0] = v[0] / n; o[1] = v[1] / n; o[2] = v[2] / n; }
static float v3_dot(const float a[3], const float b[3]) { return (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2]; }
static float v3_len(const float a[3]) { return sqrtf(v3_dot(a, a)); }
static float clampf(float x, float lo, float hi) { return fminf(fmaxf(x, lo), hi); }

/* calculateSpecular (:124-130) */
static void gl_specular(const float frag[3], const float ldir[3], const float spec[3], const float n[3], const float cam[3], float out[3]) {
    float cd[3] = { cam[0] - frag[0], cam[1] - frag[1], cam[2] - frag[2] }, cdn[3], h[3], hn[3];
    v3_normalize(cd, cdn);
    h[0] = ldir[0] + cdn[0]; h[1] = ldir[1] + cdn[1]; h[2] = ldir[2] + cdn[2];
    v3_normalize(h, hn);
    float f = powf(fmaxf(v3_dot(n, hn), 0.0f), 64.0f);
    out[0] = spec[0] * f; out[1] = spec[1] * f; out[2] = spec[2] * f;
}
/* calculateAttenuation (:132-136) */
static float gl_attenuation(const float frag[3], float lin, float quad, const float lp[3]) {
    float d[3] = { lp[0] - frag[0], lp[1] - frag[1], lp[2] - frag[2] };
    float dist = v3_len(d);
    return 1.0f / (1.0f + lin * dist + quad * dist * dist);
}
/* calculateSpotLights (:93-114); the shadow value is computed and discarded in the shader */
static void gl_spot(const ro_lights *L, const float frag[3], const float n[3], const float od[3], float acc[3]) {
    acc[0] = acc[1] = acc[2] = 0.0f;
    for (uint32_t i = 0; i < L->n_spot; i++) {
        const float *lp = L->spot_pos + 3 * i;
        float d[3] = { lp[0] - frag[0], lp[1] - frag[1], lp[2] - frag[2] };
        if (v3_len(d) > L->spot_radius[i]) continue;
        float nd[3]; v3_normalize(d, nd);
        float att = gl_attenuation(frag, L->spot_linear[i], L->spot_quadratic[i], lp);
        const float *am = L->spot_ambient + 4 * i, *df = L->spot_diffuse + 3 * i, *sp = L->spot_specular + 3 * i;
        float dc = fmaxf(v3_dot(n, nd), 0.0f), spec[3];
        gl_specular(frag, nd, sp, n, L->camera_pos, spec);
        for (int k = 0; k < 3; k++) {
            acc[k] += (od[k] * am[k] * am[3]) * att;            /* calculateAmbient * attenuation */
            acc[k] += (df[k] * od[k] * dc) * att;               /* calculateDiffuse * attenuation */
            acc[k] += spec[k] * att;
        }
    }
}
/* calculatePointLights (:72-91): the cone term uses normalize(fragPosition) - lightPosition */
static void gl_point(const ro_lights *L, const float frag[3], const float n[3], const float od[3], float acc[3]) {
    acc[0] = acc[1] = acc[2] = 0.0f;
    float fn[3]; v3_normalize(frag, fn);
    for (uint32_t i = 0; i < L->n_point; i++) {
        const float *lp = L->point_pos + 3 * i;
        float dirn[3]; v3_normalize(L->point_dir + 3 * i, dirn);
        float a[3] = { fn[0] - lp[0], fn[1] - lp[1], fn[2] - lp[2] };
        float angle = v3_dot(a, dirn);
        float eps = L->point_cutoff[i] - L->point_outer_cutoff[i];
        float intensity = clampf((angle - L->point_outer_cutoff[i]) / eps, 0.0f, 1.0f);
        float d[3] = { lp[0] - frag[0], lp[1] - frag[1], lp[2] - frag[2] }, nd[3];
        v3_normalize(d, nd);
        float att = gl_attenuation(frag, L->point_linear[i], L->point_quadratic[i], lp);
        const float *am = L->point_ambient + 4 * i, *df = L->point_diffuse + 3 * i, *sp = L->point_specular + 3 * i;
        float dc = fmaxf(v3_dot(n, nd), 0.0f), spec[3];
        gl_specular(frag, nd, sp, n, L->camera_pos, spec);
        for (int k = 0; k < 3; k++) {
            acc[k] += (od[k] * am[k] * am[3]) * att;
            acc[k] += (df[k] * od[k] * dc) * att * intensity;
            acc[k] += spec[k] * att;
        }
    }
}
void ro_deferred_lighting(uint32_t npix, const float *gpos, const float *gnormal, const uint8_t *galbedo, const ro_lights *L,
                          const uint32_t *idx, uint32_t n, float *out) {
    uint32_t count = idx ? n : npix;
#ifdef _OPENMP
#pragma omp parallel for schedule(static)
#endif
    for (long j = 0; j < (long)count; j++) {
        uint32_t p = idx ? idx[j] : (uint32_t)j;
        float od[3] = { galbedo[4 * p] / 255.0f, galbedo[4 * p + 1] / 255.0f, galbedo[4 * p + 2] / 255.0f };
        float c[3];
        if (!L->any_light_source_visible) {                      /* :30-34: ambient with vec4(1,1,1,defaultDiffuseFactor) */
            for (int k = 0; k < 3; k++) c[k] = od[k] * 1.0f * L->default_diffuse_factor;
        } else {
            const float *frag = gpos + 4 * p, *nrm = gnormal + 4 * p;
            float s1[3], pt[3], s2[3];
            gl_spot(L, frag, nrm, od, s1); gl_point(L, frag, nrm, od, pt); gl_spot(L, frag, nrm, od, s2);   /* :42-44: the spot term twice */
            for (int k = 0; k < 3; k++) {
                float v = s1[k]; v += pt[k]; v += s2[k];
                v += (float)(v < L->no_light_source_cutoff) * od[k] * L->default_diffuse_factor;
                c[k] = clampf(v, 0.0f, 1.0f);
            }
        }
        out[4 * j] = c[0]; out[4 * j + 1] = c[1]; out[4 * j + 2] = c[2]; out[4 * j + 3] = 1.0f;
    }
}

/* Work count of config 5: the number of (pixel, spot light) pairs with |light - frag| <= radius, i.e. the pairs whose lighting
 * terms calculateSpotLights evaluates (second_pass_frag.glsl:93-114; the shader calls it twice per pixel).  Exact: the lights are binned on an
 * x-z grid with cells of the largest radius, and every pixel tests the lights of the 3 x 3 cells around it with the shader's own predicate. */
uint64_t ro_lighting_spot_pairs(uint32_t npix, const float *gpos, const ro_lights *L) {
    const uint32_t ns = L->n_spot;
    if (!ns || !npix) return 0;
    float rmax = 0.0f, x0 = INFINITY, x1 = -INFINITY, z0 = INFINITY, z1 = -INFINITY;
    for (uint32_t i = 0; i < ns; i++) {
        if (L->spot_radius[i] > rmax) rmax = L->spot_radius[i];
        const float *lp = L->spot_pos + 3 * i;
        if (lp[0] < x0) x0 = lp[0]; if (lp[0] > x1) x1 = lp[0]; if (lp[2] < z0) z0 = lp[2]; if (lp[2] > z1) z1 = lp[2];
    }
    if (!(rmax > 0.0f)) rmax = 1.0f;
    const float cell = rmax * 1.0001f;
    long gx = (long)floorf((x1 - x0) / cell) + 1, gz = (long)floorf((z1 - z0) / cell) + 1;
    if (gx < 1) gx = 1; if (gz < 1) gz = 1;
    if (gx * gz > (1L << 24)) { gx = gx > 4096 ? 4096 : gx; gz = gz > 4096 ? 4096 : gz; }      /* (cells then cover more than one radius: still exact, only slower) */
    const float cx = (x1 - x0) / (float)gx > cell ? (x1 - x0) / (float)gx : cell, cz = (z1 - z0) / (float)gz > cell ? (z1 - z0) / (float)gz : cell;
    uint32_t *start = (uint32_t *)calloc((size_t)(gx * gz) + 1, 4), *items = (uint32_t *)malloc((size_t)ns * 4), *cell_of = (uint32_t *)malloc((size_t)ns * 4);
    for (uint32_t i = 0; i < ns; i++) {
        const float *lp = L->spot_pos + 3 * i;
        long ix = (long)floorf((lp[0] - x0) / cx), iz = (long)floorf((lp[2] - z0) / cz);
        if (ix >= gx) ix = gx - 1; if (iz >= gz) iz = gz - 1; if (ix < 0) ix = 0; if (iz < 0) iz = 0;
        cell_of[i] = (uint32_t)(iz * gx + ix); start[cell_of[i] + 1]++;
    }
    for (long c = 0; c < gx * gz; c++) start[c + 1] += start[c];
    uint32_t *fill = (uint32_t *)calloc((size_t)(gx * gz), 4);
    for (uint32_t i = 0; i < ns; i++) items[start[cell_of[i]] + fill[cell_of[i]]++] = i;
    uint64_t total = 0;
#ifdef _OPENMP
#pragma omp parallel for schedule(static) reduction(+ : total)
#endif
    for (long p = 0; p < (long)npix; p++) {
        const float *frag = gpos + 4 * p;
        long ix = (long)floorf((frag[0] - x0) / cx), iz = (long)floorf((frag[2] - z0) / cz);
        for (long dz = -1; dz <= 1; dz++) for (long dx = -1; dx <= 1; dx++) {
            const long jx = ix + dx, jz = iz + dz;
            if (jx < 0 || jz < 0 || jx >= gx || jz >= gz) continue;
            const long c = jz * gx + jx;
            for (uint32_t k = start[c]; k < start[c + 1]; k++) {
                const uint32_t i = items[k]; const float *lp = L->spot_pos + 3 * i;
                float d[3] = { lp[0] - frag[0], lp[1] - frag[1], lp[2] - frag[2] };
                if (!(v3_len(d) > L->spot_radius[i])) total++;
            }
        }
    }
    free(start); free(items); free(cell_of); free(fill);
    return total;
}
