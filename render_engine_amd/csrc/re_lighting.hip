// re_lighting.hip -- BASELINE.json configs[4]: the deferred-lighting second pass of render_engine
// (render_engine_assets/shaders/second_pass_frag.glsl:20-139) as a HIP compute kernel for gfx950, plus its C ABI.
//
// Tiled deferred shading.  One 256-thread workgroup owns a 32x16-pixel tile (2 pixels per lane; each wave a compact 16x8 block of it):
//   1. tile AABB of the G-buffer positions (wave shuffles + LDS),
//   2. exact-conservative culling of the radius ("spot") lights against the tile AABB, 256 lights per round, compacted
//      into an LDS list (ballot + prefix: the per-pixel summation order is deterministic).  Only the lights that can reach the
//      tile along one axis are tested: re_lighting_set_lights orders the records by LIGHT_BUCKETS slabs along the axis of the
//      lights' largest extent (ascending light index inside a slab), and a tile takes the slabs its AABB, grown by the largest
//      radius, overlaps -- 192 of the 4096 lights of configs[4] instead of all of them,
//   3. every lane shades its pixels over the list, a pair at a time as 2-vectors; the listed lights' records (64 B) are staged in LDS by the
//      whole workgroup and read as broadcasts, so the inner loop is pure f32 VALU -- the kernel is VALU-bound, not HBM-bound; a light that
//      reaches no pixel of the wave's block costs the wave its distance test only,
//   4. cone ("point") lights cannot be culled (no radius in the shader) and are evaluated for every pixel,
//   5. epilogue exactly as main(): (spot + point) + spot -- the spot term is added twice in the shader --, the
//      default-diffuse floor, clamp.
// gLightPosition is not read: the shadow value it feeds is computed and discarded by the shader (:105).
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>
#include "re_hip.h"
#include "re_guard.h"

namespace {

constexpr int TILE = 32, LT_THREADS = 256, LIST_CAP = 384;
constexpr int NPX = 2, NPAIR = NPX / 2, TILE_H = 8 * NPX;   // pixels per lane (pairs of them shaded as 2-vectors); a workgroup's tile is TILE x TILE_H pixels, a wave's share of it 16 x 4 NPX
constexpr uint32_t CULL_CHUNK = 1;      // rounds of 256 lights whose positions are in flight together (a tile of configs[4] tests 192 lights; more rounds in flight cost registers: 8 -> 152 VGPRs, 2 -> 128)
constexpr uint32_t LIGHT_BUCKETS = 4096; // slabs along the sort axis of the radius lights

struct LightParams {
    uint32_t width, height, n_spot, n_point;
    float cam[3]; float cutoff, default_diffuse; uint32_t any_visible;
    uint32_t axis; float kmin, inv_w, rmax;                                   // slab of a light: light_bucket(position[axis])
};
// the slab a coordinate falls into: monotone in `key` (subtraction, multiplication by a positive constant, clamp and floor all are), so every light whose
// coordinate lies in [a, b] has its slab in [light_bucket(a), light_bucket(b)] -- the host files the lights with this same function
__host__ __device__ inline uint32_t light_bucket(float key, float kmin, float inv_w) {
    float b = (key - kmin) * inv_w;
    b = b > 0.0f ? b : 0.0f;                                                    // (also NaN -> 0)
    b = b < (float)(LIGHT_BUCKETS - 1u) ? b : (float)(LIGHT_BUCKETS - 1u);
    return (uint32_t)b;
}

__device__ __forceinline__ float3 f3(float x, float y, float z) { return make_float3(x, y, z); }
__device__ __forceinline__ float dot3(float3 a, float3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
__device__ __forceinline__ float3 sub3(float3 a, float3 b) { return f3(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ float3 norm3v(float3 v) { float n = sqrtf(dot3(v, v)); return f3(v.x / n, v.y / n, v.z / n); }
__device__ __forceinline__ float uniform(float v) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v))); }   // a value every lane holds alike -> a scalar register
__device__ __forceinline__ float pow64(float x) { x *= x; x *= x; x *= x; x *= x; x *= x; x *= x; return x; }

// one light's contribution to one pixel; A = (pos, radius|unused), B = (diffuse, linear), C = (specular, quadratic), D = ambient rgba
__device__ __forceinline__ void shade(float3 frag, float3 nrm, float3 od, float3 camdir, float4 A, float4 B, float4 C, float4 D, bool radius_cut, float intensity, float3 &acc) {
#pragma clang fp contract(fast)          // this file's arithmetic is tolerance-bound (1e-4), not bit-exact like the cull path: multiply-adds may fuse here (the library is built with -ffp-contract=off)
    // The shader's divisions and square roots are GPU-precision operations in the reference too (GLSL gives no IEEE guarantee); here they are the
    // hardware's reciprocal / reciprocal square root (1 ulp) instead of the ~10-instruction IEEE sequences: the kernel is bound by the VALU work of
    // this function, and the result stays within 1e-6 of the f32 GLSL restatement (tolerance of the path: 1e-4).
    float3 d = sub3(f3(A.x, A.y, A.z), frag);
    const float d2 = dot3(d, d);
    if (radius_cut) {                                                          // :97-100: dist > radius, decided exactly (the cut is a discontinuity) -- the square root only near the boundary
        const float r2 = A.w * A.w;
        if (A.w < 0.0f || d2 > r2 * 1.000001f) return;                         // (a distance is never below a negative radius)
        const bool shell = d2 > r2 * 0.999999f;
        if (__builtin_amdgcn_ballot_w64(shell)) {                              // a wave-uniform branch the compiler cannot flatten (the asm pins the operand inside it): flattened,
            float t = d2; asm volatile("" : "+v"(t));                           // the IEEE square root (22 instructions) ran for every pixel inside the radius
            if (shell && sqrtf(t) > A.w) return;
        }
    }
    const float inv = __builtin_amdgcn_rsqf(d2), dist = d2 * inv;
    float3 nd = f3(d.x * inv, d.y * inv, d.z * inv);
    float att = __builtin_amdgcn_rcpf(1.0f + B.w * dist + C.w * dist * dist);  // calculateAttenuation :132-136
    float dc = fmaxf(dot3(nrm, nd), 0.0f);                                     // calculateDiffuse :118-122
    float3 hv = f3(nd.x + camdir.x, nd.y + camdir.y, nd.z + camdir.z);         // calculateSpecular :124-130
    const float invh = __builtin_amdgcn_rsqf(dot3(hv, hv));
    float sf = pow64(fmaxf(dot3(nrm, hv) * invh, 0.0f));
    acc.x += (od.x * D.x * D.w) * att; acc.x += (B.x * od.x * dc) * att * intensity; acc.x += (C.x * sf) * att;
    acc.y += (od.y * D.y * D.w) * att; acc.y += (B.y * od.y * dc) * att * intensity; acc.y += (C.y * sf) * att;
    acc.z += (od.z * D.z * D.w) * att; acc.z += (B.z * od.z * dc) * att * intensity; acc.z += (C.z * sf) * att;
}

// The radius lights' inner loop works on TWO pixels of a lane at a time, as 2-vectors: gfx950 issues v_pk_{add,mul,fma}_f32 on a register pair at the rate of
// the scalar forms, so everything but the reciprocals / reciprocal square roots costs half the instructions per pixel.  A pixel of the pair that is outside the
// radius (or outside the image) is computed along and its contribution dropped at the end (selected, not multiplied: its intermediate values may be inf / NaN).
typedef float f2 __attribute__((ext_vector_type(2)));
struct PixelPair { f2 fx, fy, fz, nx, ny, nz, ox, oy, oz, cx, cy, cz; };        // position, normal, albedo, direction to the camera
__device__ __forceinline__ f2 rsq2(f2 v) { f2 r; r.x = __builtin_amdgcn_rsqf(v.x); r.y = __builtin_amdgcn_rsqf(v.y); return r; }
__device__ __forceinline__ f2 rcp2(f2 v) { f2 r; r.x = __builtin_amdgcn_rcpf(v.x); r.y = __builtin_amdgcn_rcpf(v.y); return r; }
__device__ __forceinline__ f2 max0(f2 v) { f2 r; r.x = fmaxf(v.x, 0.0f); r.y = fmaxf(v.y, 0.0f); return r; }
__device__ __forceinline__ void shade_pair(const PixelPair &X, bool live0, bool live1, float4 A, float4 B, float4 C, float4 D, f2 &ax, f2 &ay, f2 &az) {
#pragma clang fp contract(fast)
    const f2 dx = A.x - X.fx, dy = A.y - X.fy, dz = A.z - X.fz;
    const f2 d2 = (dx * dx + dy * dy) + dz * dz;
    // :97-100: dist > radius, decided exactly (the cut is a discontinuity) -- the square root only near the boundary
    const float r2 = A.w * A.w, r2_out = r2 * 1.000001f, r2_in = r2 * 0.999999f;
    bool in0 = live0 && !(A.w < 0.0f) && !(d2.x > r2_out), in1 = live1 && !(A.w < 0.0f) && !(d2.y > r2_out);      // (a distance is never below a negative radius)
    const bool shell0 = in0 && d2.x > r2_in, shell1 = in1 && d2.y > r2_in;
    if (__builtin_amdgcn_ballot_w64(shell0 || shell1)) {                      // a wave-uniform branch the compiler cannot flatten (the asm pins the operands inside it): flattened,
        float t0 = d2.x, t1 = d2.y; asm volatile("" : "+v"(t0), "+v"(t1));     // the IEEE square root (22 instructions) ran for every pixel inside the radius
        if (shell0 && sqrtf(t0) > A.w) in0 = false;
        if (shell1 && sqrtf(t1) > A.w) in1 = false;
    }
    if (!(in0 || in1)) return;
    const f2 inv = rsq2(d2), dist = d2 * inv;
    const f2 ndx = dx * inv, ndy = dy * inv, ndz = dz * inv;
    const f2 att = rcp2(1.0f + B.w * dist + C.w * dist * dist);                // calculateAttenuation :132-136
    const f2 dc = max0((X.nx * ndx + X.ny * ndy) + X.nz * ndz);                // calculateDiffuse :118-122
    const f2 hx = ndx + X.cx, hy = ndy + X.cy, hz = ndz + X.cz;                // calculateSpecular :124-130
    const f2 invh = rsq2((hx * hx + hy * hy) + hz * hz);
    f2 sf = max0(((X.nx * hx + X.ny * hy) + X.nz * hz) * invh);
    sf *= sf; sf *= sf; sf *= sf; sf *= sf; sf *= sf; sf *= sf;                 // pow(., 64)
    // ambient * albedo + diffuse * albedo * dc + specular * sf, all times the attenuation
    const f2 tx = (X.ox * (B.x * dc + D.x * D.w) + C.x * sf) * att, ty = (X.oy * (B.y * dc + D.y * D.w) + C.y * sf) * att, tz = (X.oz * (B.z * dc + D.z * D.w) + C.z * sf) * att;
    f2 cx, cy, cz;
    cx.x = in0 ? tx.x : 0.0f; cx.y = in1 ? tx.y : 0.0f; cy.x = in0 ? ty.x : 0.0f; cy.y = in1 ? ty.y : 0.0f; cz.x = in0 ? tz.x : 0.0f; cz.y = in1 ? tz.y : 0.0f;
    ax += cx; ay += cy; az += cz;
}

__global__ __launch_bounds__(LT_THREADS) void k_deferred_lighting(LightParams P, const float4 *__restrict__ gpos, const float4 *__restrict__ gnormal, const uchar4 *__restrict__ galbedo,
                                                                   const float4 *__restrict__ spot,     // 4 float4 per light: A, B, C, D (slab order)
                                                                   const uint32_t *__restrict__ slab_start,   // LIGHT_BUCKETS + 1: first record of each slab
                                                                   const float4 *__restrict__ point,    // 5 float4 per light: A(pos), B, C, D, E(dir.xyz normalised, -) + F(cutoff, outer, -, -) packed as 6
                                                                   float4 *__restrict__ out) {
    __shared__ float s_red[4][6];
    __shared__ float4 s_rec[LIST_CAP * 4];                                    // the listed lights' records (A, B, C, D)
    __shared__ uint32_t s_wcnt[4];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wid = tid >> 6;
    // A wave owns a 16x16 QUADRANT of the tile (lane -> column lane & 15, rows (lane >> 4) + 4k): a listed light that misses the quadrant costs the wave its distance test only (shade_pair
    // returns with no lane inside).  With the rows of a wave spread over the whole tile (round 2: row (tid >> 5) + 8k) every wave went through the full evaluation of nearly every listed light.
    static_assert(TILE == 32 && LT_THREADS == 256 && NPX % 2 == 0, "four waves, one quadrant each; pixels in pairs");
    const uint32_t tx = blockIdx.x * TILE + 16u * (wid & 1u) + (lane & 15u), ty0 = blockIdx.y * TILE_H + (uint32_t)(4 * NPX) * (wid >> 1) + (lane >> 4);
    float3 frag[NPX], nrm[NPX], od[NPX], camdir[NPX], acc[NPX]; bool live[NPX];
    float lo[3] = { 3.4e38f, 3.4e38f, 3.4e38f }, hi[3] = { -3.4e38f, -3.4e38f, -3.4e38f };
#pragma unroll
    for (int k = 0; k < NPX; k++) {
        uint32_t y = ty0 + 4u * k;
        live[k] = tx < P.width && y < P.height;
        frag[k] = nrm[k] = od[k] = camdir[k] = f3(0.f, 0.f, 0.f); acc[k] = f3(0.f, 0.f, 0.f);
        if (live[k]) {
            size_t p = (size_t)y * P.width + tx;
            // (streamed once: non-temporal, so the G-buffer does not push the light records and slab bounds out of the L2)
            typedef float v4f __attribute__((ext_vector_type(4)));
            const v4f gpv = __builtin_nontemporal_load(reinterpret_cast<const v4f *>(gpos) + p), gnv = __builtin_nontemporal_load(reinterpret_cast<const v4f *>(gnormal) + p);
            const uint32_t gaw = __builtin_nontemporal_load(reinterpret_cast<const uint32_t *>(galbedo) + p);
            const float4 gp = make_float4(gpv.x, gpv.y, gpv.z, gpv.w), gn = make_float4(gnv.x, gnv.y, gnv.z, gnv.w);
            uchar4 ga; ga.x = (unsigned char)(gaw & 0xFFu); ga.y = (unsigned char)((gaw >> 8) & 0xFFu); ga.z = (unsigned char)((gaw >> 16) & 0xFFu); ga.w = (unsigned char)(gaw >> 24);
            frag[k] = f3(gp.x, gp.y, gp.z); nrm[k] = f3(gn.x, gn.y, gn.z);
            // (per-pixel setup with the hardware's 1-ulp operations, like the shading itself: the IEEE divisions and square root here were ~300 of a tile's ~1,900 instructions per wave)
            constexpr float k255 = 1.0f / 255.0f;
            od[k] = f3(ga.x * k255, ga.y * k255, ga.z * k255);
            const float3 tc = sub3(f3(P.cam[0], P.cam[1], P.cam[2]), frag[k]); const float itc = __builtin_amdgcn_rsqf(dot3(tc, tc));
            camdir[k] = f3(tc.x * itc, tc.y * itc, tc.z * itc);
            lo[0] = fminf(lo[0], gp.x); lo[1] = fminf(lo[1], gp.y); lo[2] = fminf(lo[2], gp.z);
            hi[0] = fmaxf(hi[0], gp.x); hi[1] = fmaxf(hi[1], gp.y); hi[2] = fmaxf(hi[2], gp.z);
        }
    }
    if (P.any_visible) {
        // ---- tile AABB ----
#pragma unroll
        for (int a = 0; a < 3; a++)
            for (int d = 32; d >= 1; d >>= 1) { lo[a] = fminf(lo[a], __shfl_xor(lo[a], d, 64)); hi[a] = fmaxf(hi[a], __shfl_xor(hi[a], d, 64)); }
        if (lane == 0) for (int a = 0; a < 3; a++) { s_red[wid][a] = lo[a]; s_red[wid][3 + a] = hi[a]; }
        __syncthreads();
        for (int a = 0; a < 3; a++) { lo[a] = uniform(fminf(fminf(s_red[0][a], s_red[1][a]), fminf(s_red[2][a], s_red[3][a]))); hi[a] = uniform(fmaxf(fmaxf(s_red[0][3 + a], s_red[1][3 + a]), fmaxf(s_red[2][3 + a], s_red[3][3 + a]))); }
        // ---- radius lights: cull 256 per round into the ordered LDS list, shade whenever the list could overflow ----
        // Two things keep a tile from waiting on one memory round trip after the other: the positions of CULL_CHUNK rounds of lights are requested
        // together before their tests, and the 64-byte records of the listed lights are fetched into LDS by the whole workgroup in one go, so the
        // per-pixel loop reads them as LDS broadcasts (round 1 loaded each light's record with scalar loads inside that loop: one exposed L2
        // latency per listed light and tile).
        PixelPair X[NPAIR]; f2 sax[NPAIR], say[NPAIR], saz[NPAIR];                // pixels (0, 1), (2, 3) .. of the lane
#pragma unroll
        for (int q = 0; q < NPAIR; q++) {
            const int k0 = 2 * q, k1 = 2 * q + 1;
            X[q].fx.x = frag[k0].x; X[q].fx.y = frag[k1].x; X[q].fy.x = frag[k0].y; X[q].fy.y = frag[k1].y; X[q].fz.x = frag[k0].z; X[q].fz.y = frag[k1].z;
            X[q].nx.x = nrm[k0].x; X[q].nx.y = nrm[k1].x; X[q].ny.x = nrm[k0].y; X[q].ny.y = nrm[k1].y; X[q].nz.x = nrm[k0].z; X[q].nz.y = nrm[k1].z;
            X[q].ox.x = od[k0].x; X[q].ox.y = od[k1].x; X[q].oy.x = od[k0].y; X[q].oy.y = od[k1].y; X[q].oz.x = od[k0].z; X[q].oz.y = od[k1].z;
            X[q].cx.x = camdir[k0].x; X[q].cx.y = camdir[k1].x; X[q].cy.x = camdir[k0].y; X[q].cy.y = camdir[k1].y; X[q].cz.x = camdir[k0].z; X[q].cz.y = camdir[k1].z;
            sax[q] = say[q] = saz[q] = (f2)(0.0f);
        }
        uint32_t n = 0;                                                        // listed lights (uniform)
        auto shade_list = [&]() {                                              // (the records are in s_rec: the lane that listed a light put its record there, behind the barrier of the round)
            for (uint32_t j = 0; j < n; j++) {
                const float4 A = s_rec[j * 4u], B = s_rec[j * 4u + 1u], C = s_rec[j * 4u + 2u], D = s_rec[j * 4u + 3u];
#pragma unroll
                for (int q = 0; q < NPAIR; q++) shade_pair(X[q], live[2 * q], live[2 * q + 1], A, B, C, D, sax[q], say[q], saz[q]);
            }
            __syncthreads();                                                   // (s_rec is refilled afterwards)
            n = 0;
        };
        // the records a light of this tile can be among: the slabs [lo - rmax, hi + rmax] along the sort axis (a light farther away along that axis alone misses the tile)
        const float reach = P.rmax * 1.00001f + 1e-3f;                         // (the margin of the test below)
        const float alo = P.axis == 0 ? lo[0] : (P.axis == 1 ? lo[1] : lo[2]), ahi = P.axis == 0 ? hi[0] : (P.axis == 1 ? hi[1] : hi[2]);
        uint32_t l_begin = 0, l_end = 0;
        if (P.n_spot && ahi >= alo) { l_begin = slab_start[light_bucket(alo - reach, P.kmin, P.inv_w)]; l_end = slab_start[light_bucket(ahi + reach, P.kmin, P.inv_w) + 1u]; }
        for (uint32_t c0 = l_begin; c0 < l_end; c0 += LT_THREADS * CULL_CHUNK) {
            // the WHOLE record of every light tested (64 B, L2-resident: 4,096 lights are 256 KB), not its position alone: a lane that lists its light stores the record into LDS at once,
            // and the tile does not wait for a second round trip (list -> records) before it can shade
            float4 Ac[CULL_CHUNK], Bc[CULL_CHUNK], Cc[CULL_CHUNK], Dc[CULL_CHUNK];
#pragma unroll
            for (uint32_t r = 0; r < CULL_CHUNK; r++) {
                const uint32_t li = c0 + r * LT_THREADS + tid;
                Ac[r] = make_float4(0.f, 0.f, 0.f, -1.f); Bc[r] = Cc[r] = Dc[r] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (li < l_end) { Ac[r] = spot[(size_t)li * 4]; Bc[r] = spot[(size_t)li * 4 + 1]; Cc[r] = spot[(size_t)li * 4 + 2]; Dc[r] = spot[(size_t)li * 4 + 3]; }
            }
#pragma unroll
            for (uint32_t r = 0; r < CULL_CHUNK; r++) {
                const uint32_t i0 = c0 + r * LT_THREADS;
                if (i0 >= l_end) break;                                         // uniform
                const uint32_t li = i0 + tid; bool hit = false;
                if (li < l_end) {
                    const float4 A = Ac[r];
                    float dx = fmaxf(fmaxf(lo[0] - A.x, A.x - hi[0]), 0.0f), dy = fmaxf(fmaxf(lo[1] - A.y, A.y - hi[1]), 0.0f), dz = fmaxf(fmaxf(lo[2] - A.z, A.z - hi[2]), 0.0f);
                    float rr = A.w * 1.00001f + 1e-3f;                              // conservative: the exact per-pixel radius test follows
                    hit = ((dx * dx + dy * dy) + dz * dz) <= rr * rr;
                }
                const uint64_t m = __ballot(hit);
                if (lane == 0) s_wcnt[wid] = (uint32_t)__popcll(m);
                __syncthreads();
                uint32_t base = n, tot = 0;
                for (uint32_t w = 0; w < 4; w++) { if (w < wid) base += s_wcnt[w]; tot += s_wcnt[w]; }
                if (hit) {
                    const uint32_t at = (base + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u))) * 4u;
                    s_rec[at] = Ac[r]; s_rec[at + 1u] = Bc[r]; s_rec[at + 2u] = Cc[r]; s_rec[at + 3u] = Dc[r];
                }
                __syncthreads();
                n += tot;
                if (n + LT_THREADS > LIST_CAP) shade_list();                    // uniform decision: the next round could overflow the list
            }
        }
        if (n) shade_list();
        // ---- cone lights (calculatePointLights :72-91): no radius, every pixel evaluates every light ----
        float3 point_acc[NPX]; for (int k = 0; k < NPX; k++) point_acc[k] = f3(0.f, 0.f, 0.f);
        for (uint32_t l = 0; l < P.n_point; l++) {
            float4 A = point[(size_t)l * 6], B = point[(size_t)l * 6 + 1], C = point[(size_t)l * 6 + 2], D = point[(size_t)l * 6 + 3], E = point[(size_t)l * 6 + 4], F = point[(size_t)l * 6 + 5];
#pragma unroll
            for (int k = 0; k < NPX; k++) if (live[k]) {
                float3 fn = norm3v(frag[k]);
                float angle = dot3(sub3(fn, f3(A.x, A.y, A.z)), f3(E.x, E.y, E.z));
                float intensity = fminf(fmaxf((angle - F.y) / (F.x - F.y), 0.0f), 1.0f);
                shade(frag[k], nrm[k], od[k], camdir[k], A, B, C, D, false, intensity, point_acc[k]);
            }
        }
        float3 spot_acc[NPX];
#pragma unroll
        for (int q = 0; q < NPAIR; q++) { spot_acc[2 * q] = f3(sax[q].x, say[q].x, saz[q].x); spot_acc[2 * q + 1] = f3(sax[q].y, say[q].y, saz[q].y); }
#pragma unroll
        for (int k = 0; k < NPX; k++) {
            // main() :42-44: lightColour = spot; lightColour += point; lightColour += spot
            acc[k].x = (spot_acc[k].x + point_acc[k].x) + spot_acc[k].x; acc[k].y = (spot_acc[k].y + point_acc[k].y) + spot_acc[k].y; acc[k].z = (spot_acc[k].z + point_acc[k].z) + spot_acc[k].z;
            acc[k].x += (acc[k].x < P.cutoff ? 1.0f : 0.0f) * od[k].x * P.default_diffuse;
            acc[k].y += (acc[k].y < P.cutoff ? 1.0f : 0.0f) * od[k].y * P.default_diffuse;
            acc[k].z += (acc[k].z < P.cutoff ? 1.0f : 0.0f) * od[k].z * P.default_diffuse;
            acc[k].x = fminf(fmaxf(acc[k].x, 0.0f), 1.0f); acc[k].y = fminf(fmaxf(acc[k].y, 0.0f), 1.0f); acc[k].z = fminf(fmaxf(acc[k].z, 0.0f), 1.0f);
        }
    } else {
#pragma unroll
        for (int k = 0; k < NPX; k++) acc[k] = f3(od[k].x * 1.0f * P.default_diffuse, od[k].y * 1.0f * P.default_diffuse, od[k].z * 1.0f * P.default_diffuse);   // :30-34
    }
#pragma unroll
    for (int k = 0; k < NPX; k++) if (live[k]) {
        typedef float v4f __attribute__((ext_vector_type(4)));
        v4f o; o.x = acc[k].x; o.y = acc[k].y; o.z = acc[k].z; o.w = 1.0f;
        __builtin_nontemporal_store(o, reinterpret_cast<v4f *>(out) + ((size_t)(ty0 + 4u * k) * P.width + tx));
    }
}

__global__ void k_gather_pixels(const float4 *img, const uint32_t *idx, uint32_t n, float4 *out) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = img[idx[i]];
}

}  // namespace

struct re_lighting {
    re_lighting_config cfg{};
    hipStream_t stream = nullptr;
    std::string err;
    float4 *d_pos = nullptr, *d_nrm = nullptr, *d_out = nullptr, *d_spot = nullptr, *d_point = nullptr; uchar4 *d_alb = nullptr; uint32_t *d_slab = nullptr;
    LightParams P{};
    int fail(int code, const char *fmt, ...) { char buf[512]; va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap); err = buf; return code; }
};
static thread_local std::string g_lt_error;
#define LCHK(ctx, call) do { hipError_t e_ = (call); if (e_ != hipSuccess) return (ctx)->fail(RE_E_HIP, "%s failed: %s", #call, hipGetErrorString(e_)); } while (0)

extern "C" const char *re_lighting_last_error(const re_lighting *l) { return l ? l->err.c_str() : g_lt_error.c_str(); }

extern "C" int re_lighting_create(const re_lighting_config *cfg, re_lighting **out) try {
    if (!cfg || !out || !cfg->width || !cfg->height) { g_lt_error = "re_lighting_create: bad argument"; return RE_E_ARG; }
    int ndev = 0; hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev == 0) { g_lt_error = std::string("re_lighting_create: no HIP device (") + hipGetErrorString(e) + "); no CPU path"; return RE_E_HIP; }
    if (cfg->device < 0 || cfg->device >= ndev) { g_lt_error = "re_lighting_create: device ordinal out of range"; return RE_E_ARG; }
    re_lighting *l = new re_lighting(); l->cfg = *cfg;
    size_t np = (size_t)cfg->width * cfg->height;
    if (hipSetDevice(cfg->device) != hipSuccess || hipStreamCreateWithFlags(&l->stream, hipStreamNonBlocking) != hipSuccess ||
        hipMalloc(&l->d_pos, np * 16) != hipSuccess || hipMalloc(&l->d_nrm, np * 16) != hipSuccess || hipMalloc(&l->d_out, np * 16) != hipSuccess || hipMalloc(&l->d_alb, np * 4) != hipSuccess ||
        hipMalloc(&l->d_spot, (size_t)(cfg->max_spot_lights + 1) * 64) != hipSuccess || hipMalloc(&l->d_slab, (size_t)(LIGHT_BUCKETS + 1u) * 4) != hipSuccess || hipMalloc(&l->d_point, (size_t)(cfg->max_point_lights + 1) * 96) != hipSuccess) {
        g_lt_error = "re_lighting_create: device allocation failed"; delete l; return RE_E_HIP;
    }
    l->P.width = cfg->width; l->P.height = cfg->height;
    *out = l; return RE_OK;
} RE_ABI_GUARD_NOCTX(g_lt_error, "re_lighting_create")
extern "C" void re_lighting_destroy(re_lighting *l) try {
    if (!l) return;
    (void)hipSetDevice(l->cfg.device);
    if (l->stream) (void)hipStreamSynchronize(l->stream);
    (void)hipFree(l->d_pos); (void)hipFree(l->d_nrm); (void)hipFree(l->d_out); (void)hipFree(l->d_alb); (void)hipFree(l->d_spot); (void)hipFree(l->d_point); (void)hipFree(l->d_slab);
    if (l->stream) (void)hipStreamDestroy(l->stream);
    delete l;
} catch (...) {}
extern "C" int re_lighting_upload_gbuffer(re_lighting *l, const float *g_position, const float *g_normal, const uint8_t *g_albedo_spec) try {
    if (!l || !g_position || !g_normal || !g_albedo_spec) return RE_E_ARG;
    LCHK(l, hipSetDevice(l->cfg.device));
    size_t np = (size_t)l->cfg.width * l->cfg.height;
    LCHK(l, hipMemcpyAsync(l->d_pos, g_position, np * 16, hipMemcpyHostToDevice, l->stream));
    LCHK(l, hipMemcpyAsync(l->d_nrm, g_normal, np * 16, hipMemcpyHostToDevice, l->stream));
    LCHK(l, hipMemcpyAsync(l->d_alb, g_albedo_spec, np * 4, hipMemcpyHostToDevice, l->stream));
    LCHK(l, hipStreamSynchronize(l->stream));
    return RE_OK;
} RE_ABI_GUARD(l, "re_lighting_upload_gbuffer")
extern "C" int re_lighting_set_lights(re_lighting *l, const re_lights *L) try {
    if (!l || !L) return RE_E_ARG;
    if (L->n_spot > l->cfg.max_spot_lights || L->n_point > l->cfg.max_point_lights) return l->fail(RE_E_CAPACITY, "more lights than configured");
    LCHK(l, hipSetDevice(l->cfg.device));
    std::vector<float> s((size_t)L->n_spot * 16), p((size_t)L->n_point * 24);
    // the radius lights go to the device in slab order along the axis of their largest extent (see k_deferred_lighting): counting sort by light_bucket,
    // ascending light index inside a slab
    uint32_t axis = 0; float kmin = 0.0f, inv_w = 0.0f, rmax = 0.0f;
    std::vector<uint32_t> slab_start(LIGHT_BUCKETS + 1u, 0u), place(L->n_spot);
    {
        float mn[3] = { 3.4e38f, 3.4e38f, 3.4e38f }, mx[3] = { -3.4e38f, -3.4e38f, -3.4e38f };
        for (uint32_t i = 0; i < L->n_spot; i++) {
            for (int k = 0; k < 3; k++) { const float v = L->spot_pos[3 * i + k]; if (v < mn[k]) mn[k] = v; if (v > mx[k]) mx[k] = v; }   // (comparisons with NaN are false: it takes no part)
            if (L->spot_radius[i] > rmax) rmax = L->spot_radius[i];
        }
        float best = -1.0f;
        for (int k = 0; k < 3; k++) { const float e = mx[k] - mn[k]; if (e > best && e < 3.0e38f) { best = e; axis = (uint32_t)k; } }
        if (best > 0.0f) { kmin = mn[axis]; inv_w = (float)LIGHT_BUCKETS / best; if (!(inv_w < 3.0e38f)) inv_w = 0.0f; }                // (all lights at one coordinate, or an extent too small to divide by: one slab)
        for (uint32_t i = 0; i < L->n_spot; i++) slab_start[light_bucket(L->spot_pos[3 * i + axis], kmin, inv_w) + 1u]++;
        for (uint32_t b = 0; b < LIGHT_BUCKETS; b++) slab_start[b + 1u] += slab_start[b];
        std::vector<uint32_t> fill(slab_start.begin(), slab_start.end() - 1);
        for (uint32_t i = 0; i < L->n_spot; i++) place[i] = fill[light_bucket(L->spot_pos[3 * i + axis], kmin, inv_w)]++;
    }
    for (uint32_t i = 0; i < L->n_spot; i++) {
        float *o = &s[(size_t)place[i] * 16];
        o[0] = L->spot_pos[3 * i]; o[1] = L->spot_pos[3 * i + 1]; o[2] = L->spot_pos[3 * i + 2]; o[3] = L->spot_radius[i];
        o[4] = L->spot_diffuse[3 * i]; o[5] = L->spot_diffuse[3 * i + 1]; o[6] = L->spot_diffuse[3 * i + 2]; o[7] = L->spot_linear[i];
        o[8] = L->spot_specular[3 * i]; o[9] = L->spot_specular[3 * i + 1]; o[10] = L->spot_specular[3 * i + 2]; o[11] = L->spot_quadratic[i];
        o[12] = L->spot_ambient[4 * i]; o[13] = L->spot_ambient[4 * i + 1]; o[14] = L->spot_ambient[4 * i + 2]; o[15] = L->spot_ambient[4 * i + 3];
    }
    for (uint32_t i = 0; i < L->n_point; i++) {
        float *o = &p[(size_t)i * 24];
        o[0] = L->point_pos[3 * i]; o[1] = L->point_pos[3 * i + 1]; o[2] = L->point_pos[3 * i + 2]; o[3] = 0.f;
        o[4] = L->point_diffuse[3 * i]; o[5] = L->point_diffuse[3 * i + 1]; o[6] = L->point_diffuse[3 * i + 2]; o[7] = L->point_linear[i];
        o[8] = L->point_specular[3 * i]; o[9] = L->point_specular[3 * i + 1]; o[10] = L->point_specular[3 * i + 2]; o[11] = L->point_quadratic[i];
        o[12] = L->point_ambient[4 * i]; o[13] = L->point_ambient[4 * i + 1]; o[14] = L->point_ambient[4 * i + 2]; o[15] = L->point_ambient[4 * i + 3];
        const float *d = L->point_dir + 3 * i; float n = sqrtf((d[0] * d[0] + d[1] * d[1]) + d[2] * d[2]);     // normalize(pointLightDirection[i])
        o[16] = d[0] / n; o[17] = d[1] / n; o[18] = d[2] / n; o[19] = 0.f;
        o[20] = L->point_cutoff[i]; o[21] = L->point_outer_cutoff[i]; o[22] = o[23] = 0.f;
    }
    if (L->n_spot) LCHK(l, hipMemcpyAsync(l->d_spot, s.data(), s.size() * 4, hipMemcpyHostToDevice, l->stream));
    if (L->n_point) LCHK(l, hipMemcpyAsync(l->d_point, p.data(), p.size() * 4, hipMemcpyHostToDevice, l->stream));
    LCHK(l, hipMemcpyAsync(l->d_slab, slab_start.data(), slab_start.size() * 4, hipMemcpyHostToDevice, l->stream));
    LCHK(l, hipStreamSynchronize(l->stream));
    l->P.n_spot = L->n_spot; l->P.n_point = L->n_point; l->P.axis = axis; l->P.kmin = kmin; l->P.inv_w = inv_w; l->P.rmax = rmax;
    for (int k = 0; k < 3; k++) l->P.cam[k] = L->camera_pos[k];
    l->P.cutoff = L->no_light_source_cutoff; l->P.default_diffuse = L->default_diffuse_factor; l->P.any_visible = L->any_light_source_visible;
    return RE_OK;
} RE_ABI_GUARD(l, "re_lighting_set_lights")
extern "C" int re_lighting_run(re_lighting *l, float *kernel_us) try {
    if (!l) return RE_E_ARG;
    LCHK(l, hipSetDevice(l->cfg.device));
    hipEvent_t a = nullptr, b = nullptr;
    if (kernel_us) { LCHK(l, hipEventCreate(&a)); LCHK(l, hipEventCreate(&b)); }
    dim3 grid((l->cfg.width + TILE - 1) / TILE, (l->cfg.height + TILE_H - 1) / TILE_H);
    hipExtLaunchKernelGGL(k_deferred_lighting, grid, dim3(LT_THREADS), 0, l->stream, a, b, 0, l->P, l->d_pos, l->d_nrm, l->d_alb, l->d_spot, l->d_slab, l->d_point, l->d_out);
    LCHK(l, hipGetLastError());
    LCHK(l, hipStreamSynchronize(l->stream));
    if (kernel_us) { float ms = 0; LCHK(l, hipEventElapsedTime(&ms, a, b)); *kernel_us = ms * 1000.f; (void)hipEventDestroy(a); (void)hipEventDestroy(b); }
    return RE_OK;
} RE_ABI_GUARD(l, "re_lighting_run")
extern "C" int re_lighting_read(re_lighting *l, float *out_rgba) try {
    if (!l || !out_rgba) return RE_E_ARG;
    LCHK(l, hipSetDevice(l->cfg.device));
    LCHK(l, hipMemcpy(out_rgba, l->d_out, (size_t)l->cfg.width * l->cfg.height * 16, hipMemcpyDeviceToHost));
    return RE_OK;
} RE_ABI_GUARD(l, "re_lighting_read")
extern "C" int re_lighting_read_pixels(re_lighting *l, const uint32_t *idx, uint32_t n, float *out_rgba) try {
    if (!l || !idx || !out_rgba) return RE_E_ARG;
    LCHK(l, hipSetDevice(l->cfg.device));
    uint32_t *d_idx = nullptr; float4 *d_o = nullptr;
    LCHK(l, hipMalloc(&d_idx, (size_t)n * 4 + 4)); LCHK(l, hipMalloc(&d_o, (size_t)n * 16 + 16));
    LCHK(l, hipMemcpy(d_idx, idx, (size_t)n * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_gather_pixels, dim3((n + 255) / 256), dim3(256), 0, l->stream, l->d_out, d_idx, n, d_o);
    LCHK(l, hipStreamSynchronize(l->stream));
    LCHK(l, hipMemcpy(out_rgba, d_o, (size_t)n * 16, hipMemcpyDeviceToHost));
    (void)hipFree(d_idx); (void)hipFree(d_o);
    return RE_OK;
} RE_ABI_GUARD(l, "re_lighting_read_pixels")
