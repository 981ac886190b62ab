// re_math.h -- exact-arithmetic building blocks shared by the HIP kernels and the host side of
// librender_engine_hip.so.  Everything here must round exactly like the reference's Rust/nalgebra
// code: one IEEE-754 operation per source operation, no FMA contraction (the library is built
// with -ffp-contract=off), correctly rounded sqrt and division (hipcc default).
// Citations: reference file:line under /root/reference/src.
#pragma once
#include <stdint.h>
#include <math.h>

#if defined(__HIPCC__)
#define RE_HD __host__ __device__ __forceinline__
#else
#define RE_HD inline
#endif

namespace re {

struct Aabb { float xmin, xmax, ymin, ymax, zmin, zmax; };   // StaticAABB: x_range, y_range, z_range

// Rust `f as u32`: truncate toward zero, saturate, NaN -> 0
RE_HD uint32_t f2u32(float f) {
    if (!(f > 0.0f)) return 0u;
    if (f >= 4294967296.0f) return 0xFFFFFFFFu;
    return (uint32_t)f;
}
// f32::max / f32::min: a NaN operand yields the other operand
RE_HD float rmax(float a, float b) { return fmaxf(a, b); }
RE_HD float rmin(float a, float b) { return fminf(a, b); }

// nalgebra norm of a 3-vector: (x*x + y*y) + z*z, then sqrt
RE_HD float norm3(float x, float y, float z) { return sqrtf((x * x + y * y) + z * z); }

// UniqueWorldSectionId packed as level:16 | x:16 | z:16 | y:16 (bounding_box_tree_v2.rs:21-26)
RE_HD uint64_t pack_key(uint32_t level, uint32_t x, uint32_t z, uint32_t y) {
    return ((uint64_t)(level & 0xFFFFu) << 48) | ((uint64_t)(x & 0xFFFFu) << 32) | ((uint64_t)(z & 0xFFFFu) << 16) | (uint64_t)(y & 0xFFFFu);
}
RE_HD uint32_t key_level(uint64_t k) { return (uint32_t)(k >> 48) & 0xFFFFu; }
RE_HD uint32_t key_x(uint64_t k) { return (uint32_t)(k >> 32) & 0xFFFFu; }
RE_HD uint32_t key_z(uint64_t k) { return (uint32_t)(k >> 16) & 0xFFFFu; }
RE_HD uint32_t key_y(uint64_t k) { return (uint32_t)k & 0xFFFFu; }

// UniqueWorldSectionId::to_aabb (bounding_box_tree_v2.rs:95-109)
RE_HD Aabb key_to_aabb(uint64_t key, uint32_t atomic) {
    uint32_t level = key_level(key);
    float side = (float)((level < 32u ? (1u << level) : 0u) * atomic);
    float mx = side * (float)key_x(key), my = side * (float)key_y(key), mz = side * (float)key_z(key);
    Aabb a = { mx, mx + side, my, my + side, mz, mz + side };
    return a;
}

// Deterministic sin/cos: the reference calls f32::sin_cos (platform libm).  Device and CPU oracle
// evaluate the same f64 sequence (Cody-Waite reduction + fdlibm kernel polynomials, one final
// rounding to f32), so results are reproducible bit for bit and within 1 ulp of any libm.
RE_HD void sincos_det(float xf, float *s, float *c) {
    const double INV_PIO2 = 6.36619772367581382433e-01;
    const double PIO2_1   = 1.57079632673412561417e+00;
    const double PIO2_1T  = 6.07710050650619224932e-11;
    const double TWO_PI   = 6.28318530717958623200e+00;
    const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03,
                 S3 = -1.98412698298579493134e-04, S4 = 2.75573137070700676789e-06,
                 S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
    const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03,
                 C3 = 2.48015872894767294178e-05, C4 = -2.75573143513906633035e-07,
                 C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
    double x = (double)xf;
    if (!(fabs(x) < 1.0e6)) {
        if (!(fabs(x) <= 3.5e38)) { *s = (float)(x - x); *c = (float)(x - x); return; }
        x = fmod(x, TWO_PI);
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

// 4x4 column-major: m[col*4+row].  nalgebra Mul -> gemm: per output element the products are
// accumulated k = 0..3, each product and each sum rounded separately.
RE_HD void mat4_mul(const float *a, const float *b, float *out) {
    float r[16];
    for (int j = 0; j < 4; j++)
        for (int i = 0; i < 4; i++) {
            float y = a[0 * 4 + i] * b[j * 4 + 0];
            y = a[1 * 4 + i] * b[j * 4 + 1] + y;
            y = a[2 * 4 + i] * b[j * 4 + 2] + y;
            y = a[3 * 4 + i] * b[j * 4 + 3] + y;
            r[j * 4 + i] = y;
        }
    for (int k = 0; k < 16; k++) out[k] = r[k];
}
RE_HD void mat4_vec4(const float *m, float v0, float v1, float v2, float v3, float *out) {
    for (int i = 0; i < 4; i++) {
        float y = m[0 * 4 + i] * v0;
        y = m[1 * 4 + i] * v1 + y;
        y = m[2 * 4 + i] * v2 + y;
        y = m[3 * 4 + i] * v3 + y;
        out[i] = y;
    }
}

// translate(identity, p) ; rotate(.., angle, axis) ; scale(.., s) in nalgebra-glm order
// (exports/entity_transformer.rs:99-142; helper_things/entity_change_helpers.rs:248-250).
// apply_rot / apply_scale say whether the factor is applied at all (registration applies only
// components that were supplied; the change path applies all three with defaults).
RE_HD void trs_matrix(const float pos[3], bool apply_rot, const float axis[3], float angle,
                      bool apply_scale, const float scl[3], float m[16]) {
    for (int k = 0; k < 16; k++) m[k] = 0.0f;
    m[0] = m[5] = m[10] = m[15] = 1.0f;
    {   // prepend_translation on identity: bottom row . p, 3x3 block * p, then added to column 3
        float sc = (m[0 * 4 + 3] * pos[0] + m[1 * 4 + 3] * pos[1]) + m[2 * 4 + 3] * pos[2];
        float post[3];
        for (int i = 0; i < 3; i++) {
            float y = m[0 * 4 + i] * pos[0];
            y = m[1 * 4 + i] * pos[1] + y;
            y = m[2 * 4 + i] * pos[2] + y;
            post[i] = y;
        }
        m[15] += sc;
        for (int i = 0; i < 3; i++) m[12 + i] += post[i];
    }
    if (apply_rot) {
        // Rotation3::from_axis_angle(Unit::new_normalize(axis), angle).to_homogeneous(); identity when angle == 0
        float n = norm3(axis[0], axis[1], axis[2]);
        float ux = axis[0] / n, uy = axis[1] / n, uz = axis[2] / n;
        float r[16];
        for (int k = 0; k < 16; k++) r[k] = 0.0f;
        r[0] = r[5] = r[10] = r[15] = 1.0f;
        if (angle != 0.0f) {
            float sqx = ux * ux, sqy = uy * uy, sqz = uz * uz;
            float sn, cs; sincos_det(angle, &sn, &cs);
            float omc = 1.0f - cs;
            r[0] = sqx + (1.0f - sqx) * cs;   r[4] = ux * uy * omc - uz * sn; r[8]  = ux * uz * omc + uy * sn;
            r[1] = ux * uy * omc + uz * sn;   r[5] = sqy + (1.0f - sqy) * cs; r[9]  = uy * uz * omc - ux * sn;
            r[2] = ux * uz * omc - uy * sn;   r[6] = uy * uz * omc + ux * sn; r[10] = sqz + (1.0f - sqz) * cs;
        }
        mat4_mul(m, r, m);
    }
    if (apply_scale)
        for (int c = 0; c < 3; c++) for (int r = 0; r < 4; r++) m[c * 4 + r] *= scl[c];
}

// StaticAABB::apply_transformation (world/bounding_volumes/aabb.rs:95-114): min and max corners only
RE_HD Aabb apply_transformation(const Aabb &a, const float *m) {
    float f[4], s[4];
    mat4_vec4(m, a.xmin, a.ymin, a.zmin, 1.0f, f);
    mat4_vec4(m, a.xmax, a.ymax, a.zmax, 1.0f, s);
    Aabb o;
    o.xmin = rmin(f[0], s[0]); o.ymin = rmin(f[1], s[1]); o.zmin = rmin(f[2], s[2]);
    o.xmax = rmax(f[0], s[0]); o.ymax = rmax(f[1], s[1]); o.zmax = rmax(f[2], s[2]);
    return o;
}

// Range::combine, epsilon-biased union (world/dimension/range.rs:38-61)
RE_HD Aabb combine_aabb(const Aabb &a, const Aabb &b) {
    const float eps = 0.01f;
    Aabb o;
    o.xmin = ((a.xmin - eps) < b.xmin) ? a.xmin : b.xmin; o.xmax = ((a.xmax + eps) > b.xmax) ? a.xmax : b.xmax;
    o.ymin = ((a.ymin - eps) < b.ymin) ? a.ymin : b.ymin; o.ymax = ((a.ymax + eps) > b.ymax) ? a.ymax : b.ymax;
    o.zmin = ((a.zmin - eps) < b.zmin) ? a.zmin : b.zmin; o.zmax = ((a.zmax + eps) > b.zmax) ? a.zmax : b.zmax;
    return o;
}

// distance_to_aabb (helper_things/aabb_helper_functions.rs:58-72)
RE_HD float distance_to_aabb(const Aabb &a, float cx, float cy, float cz) {
    float lx = a.xmax - a.xmin, ly = a.ymax - a.ymin, lz = a.zmax - a.zmin;
    float h = rmax(rmax(lx, ly), lz) / 2.0f;
    float radius = sqrtf((h * h) * 3.0f);
    float mx = (a.xmin + a.xmax) / 2.0f, my = (a.ymin + a.ymax) / 2.0f, mz = (a.zmin + a.zmax) / 2.0f;
    float d = norm3(cx - mx, cy - my, cz - mz);
    return rmax(d - radius, 0.0f);
}

// RenderFrustumCuller::aabb_visible (culling/render_frustum_culler.rs:83-118).  planes: 6 x (nx,ny,nz,w).
// The reference ORs !(dist < 0) over the 8 corners of the box per plane and ANDs over the planes, with
// dist = ((nx*px + ny*py) + nz*pz) + w evaluated left to right.  Evaluated here ONCE per plane, with the larger of the two
// products per axis: IEEE multiplication and addition are monotone in each operand, so this distance is >= the rounded distance
// of every corner, and it IS one of the reference's eight evaluations (the corner that realises the three maxima) -- the OR over
// the corners equals the test of this one value, bit for bit, whatever the order of min and max in the box (the 2-corner
// apply_transformation can swap them).  A NaN plane coefficient makes both products of its axis NaN, so the distance is NaN and
// counts as inside exactly as in the reference.  (Not reproduced: an infinite coefficient against a zero coordinate, where one
// product is NaN and the other infinite.)  6x fewer operations on the candidate path.
RE_HD bool frustum_aabb_visible(const float *planes, const Aabb &a) {
    for (int k = 0; k < 6; k++) {
        const float nx = planes[k * 4 + 0], ny = planes[k * 4 + 1], nz = planes[k * 4 + 2], w = planes[k * 4 + 3];
        const float xm = fmaxf(nx * a.xmin, nx * a.xmax), ym = fmaxf(ny * a.ymin, ny * a.ymax), zm = fmaxf(nz * a.zmin, nz * a.zmax);
        const float d = ((xm + ym) + zm) + w;
        if (d < 0.0f) return false;
    }
    return true;
}

// LogicFrustumCuller::aabb_in_view (culling/logic_frustum_culler.rs:32-46): min over the 8 corners of |corner - camera| <= lookahead,
// each distance as norm3 = sqrt((dx*dx + dy*dy) + dz*dz).  The nearest corner is nearest per axis, and squares, sums and sqrt are
// monotone under rounding, so the minimum over the corners is the one evaluation with the smaller square per axis (f32::min
// ignores NaN: a NaN camera leaves f32::MAX and the test fails, as here).
RE_HD bool logic_aabb_in_view(float lookahead, float cx, float cy, float cz, const Aabb &a) {
    const float x0 = a.xmin - cx, x1 = a.xmax - cx, y0 = a.ymin - cy, y1 = a.ymax - cy, z0 = a.zmin - cz, z1 = a.zmax - cz;
    const float sx = rmin(x0 * x0, x1 * x1), sy = rmin(y0 * y0, y1 * y1), sz = rmin(z0 * z0, z1 * z1);
    const float best = sqrtf((sx + sy) + sz);
    return best <= lookahead;                              // NaN compares false, like f32::MAX <= lookahead
}

// ModelId::level_of_view_adjusted_model_index: LOD index only (models/model_definitions.rs:31-59)
RE_HD uint32_t lod_index(float d, uint32_t n, const float *lmin, const float *lmax) {
    for (uint32_t i = 0; i < n; i++) if (lmin[i] <= d && d <= lmax[i]) return i < 7u ? i : 7u;
    return 7u;
}

// ---- spatial-hash cell assignment (world/bounding_box_tree_v2.rs) ----
// calculate_number_world_sections_each_dimension closure (:1315-1346)
// Division by a section length.  Section lengths are atomic * 2^level; when the atomic length is a power of two (64 and 32 in every config) so is
// every level length, and x / len == x * (1 / len) bit for bit (both are the correctly rounded value of the same real number: scaling by a power of
// two is exact), likewise u / len == u >> log2(len).  A correctly rounded f32 division costs ~10 instructions and a u32 division ~40 on gfx950, and
// the section decision makes a couple of dozen of them per entity.
RE_HD float div_len(float x, uint32_t len) {
    if ((len & (len - 1u)) == 0u) return x * (1.0f / (float)len);          // (1 / 2^k is exact; len >= 1)
    return x / (float)len;
}
RE_HD uint32_t udiv_len(uint32_t u, uint32_t len) {
#if defined(__HIP_DEVICE_COMPILE__)
    if ((len & (len - 1u)) == 0u) return u >> (31 - __clz((int)len));
#else
    if ((len & (len - 1u)) == 0u) return u >> __builtin_ctz(len);
#endif
    return u / len;
}
RE_HD uint32_t num_sections_1d(float mn, float mx, uint32_t level_length) {
    float ll = (float)level_length;
    const float qmn = div_len(mn, level_length);
    if (truncf(qmn) == truncf(div_len(mx, level_length))) return 1u;
    uint32_t n;
    if (ceilf(qmn) > qmn) { mn = ceilf(qmn) * ll; n = 1u; } else n = 0u;
    while (mn < mx) { n += 1u; mn += ll; }
    return n;
}
RE_HD uint32_t num_sections_total(uint32_t ll, const Aabb &a) {
    return num_sections_1d(a.xmin, a.xmax, ll) * num_sections_1d(a.ymin, a.ymax, ll) * num_sections_1d(a.zmin, a.zmax, ll);
}
// find_aabb_level_from_length_and_origin (:532-551)
RE_HD void level_from_origin(const Aabb &a, uint32_t atomic, uint32_t *level, uint32_t *ll) {
    uint32_t len = atomic, lv = 0;
    uint32_t n = num_sections_total(len, a);
    while (n > 1u && lv < 31u) { len *= 2u; lv += 1u; n = num_sections_total(len, a); }
    *level = lv; *ll = len;
}
// aabb_out_of_bounds (helper_things/aabb_helper_functions.rs:43-52) + normalize_aabb (:1384-1397)
RE_HD bool normalize_aabb(Aabb *a, float L) {
    bool oob = a->xmin < 0.0f || a->ymin < 0.0f || a->zmin < 0.0f || a->xmax > L || a->ymax > L || a->zmax > L;
    a->xmin = rmin(rmax(a->xmin, 0.0f), L); a->ymin = rmin(rmax(a->ymin, 0.0f), L); a->zmin = rmin(rmax(a->zmin, 0.0f), L);
    a->xmax = rmin(rmax(a->xmax, 0.0f), L); a->ymax = rmin(rmax(a->ymax, 0.0f), L); a->zmax = rmin(rmax(a->zmax, 0.0f), L);
    return oob;
}
// add_entity's section decision on a normalised box: find_all_unique_world_section_ids (:466-506),
// and for a single section find_unique_world_section_id (:451-460).  Returns the number of unique
// world sections (0 or >8: the reference would panic).
RE_HD int assign_sections(const Aabb &bv, uint32_t atomic, uint64_t keys[8]) {
    Aabb shifted = { 0.0f, bv.xmax - bv.xmin, 0.0f, bv.ymax - bv.ymin, 0.0f, bv.zmax - bv.zmin };
    uint32_t level, ll; level_from_origin(shifted, atomic, &level, &ll);
    uint32_t nx = num_sections_1d(bv.xmin, bv.xmax, ll), ny = num_sections_1d(bv.ymin, bv.ymax, ll), nz = num_sections_1d(bv.zmin, bv.zmax, ll);
    uint64_t total = (uint64_t)nx * ny * nz;
    if (total == 0 || total > 8) return total > 8 ? -2 : 0;
    int n = 0;
    for (uint32_t x = 0; x < nx; x++) for (uint32_t y = 0; y < ny; y++) for (uint32_t z = 0; z < nz; z++) {
        uint32_t ix = udiv_len(f2u32(bv.xmin) + ll * x, ll), iy = udiv_len(f2u32(bv.ymin) + ll * y, ll), iz = udiv_len(f2u32(bv.zmin) + ll * z, ll);   // :1367-1378
        keys[n++] = pack_key(level, ix, iz, iy);
    }
    if (n == 1) {
        uint32_t l2, len2; level_from_origin(bv, atomic, &l2, &len2);
        keys[0] = pack_key(l2, udiv_len(f2u32(bv.xmin), len2), udiv_len(f2u32(bv.zmin), len2), udiv_len(f2u32(bv.ymin), len2));
    }
    return n;
}

// BoundingBoxTree::max_level (:1356-1359)
inline uint32_t max_level(uint32_t outline, uint32_t atomic) {
    return f2u32(ceilf(log2f((float)outline / (float)atomic))) & 0xFFFFu;
}

// RenderFrustumCuller::update_plane_coefficients (culling/render_frustum_culler.rs:59-78); host side, once per frame
inline void make_planes(const float pv[16], float planes[24]) {
    float row[4][4];
    for (int r = 0; r < 4; r++) for (int c = 0; c < 4; c++) row[r][c] = pv[c * 4 + r];
    float p[6][4];
    for (int i = 0; i < 4; i++) {
        p[0][i] = row[3][i] + row[0][i];
        p[1][i] = row[3][i] - row[0][i];
        p[2][i] = row[3][i] + row[1][i];
        p[3][i] = row[3][i] - row[1][i];
        p[4][i] = row[3][i] - 0.0f;
        p[5][i] = row[3][i] - row[2][i];
    }
    for (int k = 0; k < 6; k++) {
        float len = norm3(p[k][0], p[k][1], p[k][2]);
        for (int i = 0; i < 4; i++) planes[k * 4 + i] = p[k][i] / len;
    }
}

}  // namespace re
