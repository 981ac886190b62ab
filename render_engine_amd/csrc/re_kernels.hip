// re_kernels.hip -- gfx950 (MI355X / CDNA4) kernels of the visible-set pipeline.
//
// Written for 64-wide wavefronts: ballots are 64-bit, in-wave prefix sums use mbcnt, one atomic
// per wave reserves output space.  No MFMA: the work is branchy per-section / per-entity integer
// and f32 math, bound by HBM (the section-key scan) -- see DESIGN.md for the roofline of each kernel.
// Built with -ffp-contract=off: every f32 result that feeds a visibility decision must round
// exactly like the reference's Rust code (see re_math.h).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "re_kernels.h"
#include "re_math.h"

namespace re {

__device__ __forceinline__ uint32_t lane_id() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }
__device__ __forceinline__ uint32_t mbcnt(uint64_t mask) {
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}
// inclusive scan of v over the 64 lanes of the wave
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v) {
    uint32_t l = lane_id();
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { uint32_t o = __shfl_up(v, d, 64); if (l >= (uint32_t)d) v += o; }
    return v;
}

// ---------------------------------------------------------------------------------------------
// K_transform: per entity row, TRS -> TransformationMatrix + StaticAABB, then the spatial-hash
// section decision of BoundingBoxTree::add_entity (world/bounding_box_tree_v2.rs:563-579).
// mode 0 = registration (EntityTransformationBuilder::write_components: only supplied factors)
// ---------------------------------------------------------------------------------------------
// Rows [row0, row0 + n): the whole world at upload (row0 = 0), or the rows an add-entity batch appended (re_add_entities / RE_CHANGE_ADD_ENTITY);
// row_key / row_nk are indexed by the row's position in the range.
__global__ __launch_bounds__(256) void k_transform_assign(RowArrays R, uint32_t row0, uint32_t n, uint32_t outline, uint32_t atomic,
                                                           uint64_t *row_key, uint8_t *row_nk, SharedRec *shrec, uint32_t *shrec_count, uint32_t shrec_cap) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t r = row0 + i;
    uint32_t fl = R.flags[r];
    float pos[3] = { R.pos[r * 3 + 0], R.pos[r * 3 + 1], R.pos[r * 3 + 2] };
    float axis[3] = { R.rot[r * 4 + 0], R.rot[r * 4 + 1], R.rot[r * 4 + 2] }; float angle = R.rot[r * 4 + 3];
    float scl[3] = { R.scale[r * 3 + 0], R.scale[r * 3 + 1], R.scale[r * 3 + 2] };
    float m[16];
    Aabb orig = R.orig[r], a;
    if (fl & F_USER) {
        // the user entity (Pipeline::register_user_entity, flows/pipeline.rs:125-144): identity matrix, OriginalAABB.translate(position)
        for (int k = 0; k < 16; k++) m[k] = (k % 5 == 0) ? 1.0f : 0.0f;
        a = orig; a.xmin += pos[0]; a.xmax += pos[0]; a.ymin += pos[1]; a.ymax += pos[1]; a.zmin += pos[2]; a.zmax += pos[2];
    } else {
        trs_matrix(pos, (fl & F_HAS_ROT) != 0, axis, angle, (fl & F_HAS_SCALE) != 0, scl, m);
        a = apply_transformation(orig, m);
    }
    float4 *mo = reinterpret_cast<float4 *>(R.mat + (size_t)r * 16);
    mo[0] = make_float4(m[0], m[1], m[2], m[3]); mo[1] = make_float4(m[4], m[5], m[6], m[7]);
    mo[2] = make_float4(m[8], m[9], m[10], m[11]); mo[3] = make_float4(m[12], m[13], m[14], m[15]);
    R.aabb[r] = a;
    Aabb bv = a;
    bool oob = normalize_aabb(&bv, (float)outline);
    uint64_t keys[8];
    int nk = oob ? 0 : assign_sections(bv, atomic, keys);      // apply_choices adds with add_if_out_bounds = false
    if (nk < 0) nk = 0;
    row_nk[i] = (uint8_t)nk;
    row_key[i] = nk >= 1 ? keys[0] : 0ull;
    if (nk > 1) {
        uint32_t slot = atomicAdd(shrec_count, 1u);
        if (slot < shrec_cap) {
            SharedRec rec; rec.row = r; rec.nk = (uint32_t)nk;
            for (int k = 0; k < 8; k++) rec.keys[k] = k < nk ? keys[k] : 0ull;
            shrec[slot] = rec;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// K_fold: BoundingBoxTree::end_of_changes (:1055-1130).  One lane per world section folds the
// epsilon-biased combine over local_entities.chain(static_entities) in CSR order (ascending id),
// or takes back_up_aabb when too many sections changed and the section is crowded.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_fold_tight(uint32_t ncells, const uint64_t *cell_key, const uint32_t *cell_begin, const uint32_t *cell_nlocal,
                                                    const uint32_t *cell_nstatic, const uint32_t *rows, const Aabb *ent_aabb, Aabb *cell_tight,
                                                    uint32_t atomic, int too_many) {
    uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= ncells) return;
    uint64_t key = cell_key[c];
    uint32_t n = cell_nlocal[c] + cell_nstatic[c];
    uint32_t adj = 20u + key_level(key) * 5u; if (adj > 50u) adj = 50u;
    Aabb u = { 0.f, 0.f, 0.f, 0.f, 0.f, 0.f };                    // StaticAABB::point_aabb()
    if (too_many && n > adj) u = key_to_aabb(key, atomic);        // back_up_aabb
    else {
        uint32_t b = cell_begin[c];
        for (uint32_t i = 0; i < n; i++) {
            Aabb e = ent_aabb[rows[b + i]];
            u = (i == 0) ? e : combine_aabb(u, e);
        }
    }
    cell_tight[c] = u;
}
// shared-section branch of end_of_changes (:1104-1125): first_entity is never cleared, so the AABB
// is that of the last entity iterated (entities, then static_entities)
__global__ __launch_bounds__(256) void k_fold_shared(uint32_t nsh, const uint32_t *sh_begin, const uint32_t *sh_nact, const uint32_t *sh_nstat,
                                                     const uint32_t *rows, const Aabb *ent_aabb, Aabb *sh_aabb) {
    uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= nsh) return;
    uint32_t n = sh_nact[s] + sh_nstat[s];
    Aabb u = { 0.f, 0.f, 0.f, 0.f, 0.f, 0.f };
    if (n) u = ent_aabb[rows[sh_begin[s] + n - 1]];
    sh_aabb[s] = u;
}

// ---------------------------------------------------------------------------------------------
// K0: static render cache (flows/render_flow.rs:549-594).  A unique section whose static set
// changed is re-cached with the camera of this frame: its static entities are cached only if
// distance_to_aabb(section.aabb) < far at that moment (sort_unique_world_sections :749-754).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_static_cache_cells(uint32_t ncells, const Aabb *cell_tight, uint8_t *cell_flags, FrameParams P) {
    uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= ncells) return;
    uint8_t f = cell_flags[c];
    if (!(f & CF_STATIC_DIRTY)) return;
    float d = distance_to_aabb(cell_tight[c], P.cam[0], P.cam[1], P.cam[2]);
    f &= ~CF_STATIC_CACHED;                                   // the dirty bit stays until the frame ends (pipeline.rs:271)
    if (d < P.far_draw) f |= CF_STATIC_CACHED;
    cell_flags[c] = f;
}
// Pipeline::execute -> clear_changed_static_unique (flows/pipeline.rs:271)
__global__ __launch_bounds__(256) void k_clear_static_dirty(uint32_t ncells, uint8_t *cell_flags, uint32_t nsh, uint8_t *sh_dirty) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < ncells) { uint8_t f = cell_flags[i]; if (f & CF_STATIC_DIRTY) cell_flags[i] = f & ~CF_STATIC_DIRTY; }
    if (i < nsh) sh_dirty[i] = 0;
}
// shared sections reached from a re-cached unique section (sort_shared_world_sections(is_static) :808-866):
// the first unique section (ascending key) that reaches it caches its static entities
__global__ __launch_bounds__(256) void k_static_cache_shared(uint32_t nsh, const int32_t *sh_cells, const uint64_t *cell_key, const Aabb *sh_aabb, uint8_t *sh_dirty,
                                                             int32_t *sh_owner, uint8_t *sh_cached, FrameParams P) {
    uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= nsh) return;
    if (!sh_dirty[s]) return;
    int32_t owner = -1; uint64_t okey = 0;                        // the linking section with the smallest KEY (sections patched in since the last full build sit in spare slots: slot order is not key order)
    for (int k = 0; k < 8; k++) { int32_t c = sh_cells[s * 8 + k]; if (c >= 0) { const uint64_t kk = cell_key[c]; if (owner < 0 || kk < okey) { owner = c; okey = kk; } } }
    float d2 = distance_to_aabb(sh_aabb[s], P.cam[0], P.cam[1], P.cam[2]);
    sh_owner[s] = owner; sh_cached[s] = (uint8_t)(d2 < P.far_draw);
}

// Appends the instances of up to 64 visible sections (one per lane; cnt == 0: nothing) to the frame's
// instance list: one 64-bit atomic per wave reserves {emitting sections, instances}, an in-wave prefix
// sum gives every lane its offset, then every lane expands its own rows (RenderFlow::add_entities,
// render_flow.rs:872-933: ModelId + LOD -> group slot).  Sections with many rows are expanded by the
// whole wave so one crowded section does not serialise behind a single lane.
// segments of the instance list (see FrameHeader::cursors)
struct ShardMap { uint32_t n[CURSOR_SHARDS]; uint32_t total; };
__device__ __forceinline__ ShardMap load_shard_map(const FrameHeader *hdr, uint32_t nshards, uint32_t seg_cap) {
    ShardMap m; m.total = 0;
#pragma unroll
    for (uint32_t k = 0; k < CURSOR_SHARDS; k++) {
        uint32_t v = k < nshards ? (uint32_t)(hdr->cursors[k * CURSOR_STRIDE] >> 32) : 0u;
        m.n[k] = v < seg_cap ? v : seg_cap; m.total += m.n[k];
    }
    return m;
}
__device__ __forceinline__ uint32_t shard_item_index(uint32_t t, const ShardMap &m, uint32_t seg_cap) {
    uint32_t idx = t;
#pragma unroll
    for (uint32_t k = 0; k < CURSOR_SHARDS; k++) { if (t < m.n[k]) { idx = k * seg_cap + t; break; } t -= m.n[k]; }
    return idx;
}

constexpr uint32_t EMIT_MAX = 4u;                              // sections one lane can hold per reservation (rounds of 64 visible sections per slice)

// row0 / gc0: entry rb of the pool, fetched by the caller ahead of the cursor atomic (k == 0 when first == 0)
// hist: the wave's LDS histogram of group slots (in-scan counting, see ItemSink), or nullptr: then a counting frame adds straight to
// K.group_count[shard][slot] (the few instances of shared sections)
// dist: the section's distance (LOD input); only read for group classes with custom level-of-view bands (K.gc_lodtab)
__device__ __forceinline__ void expand_rows(uint32_t rb, uint32_t cnt, uint32_t first, uint32_t stride, uint32_t n, uint32_t off, uint32_t lod, uint32_t seg_base, const ItemSink &K,
                                            uint32_t *hist, uint32_t shard, float dist, bool have0 = false, uint32_t row0 = 0, uint32_t gc0 = 0) {
    for (uint32_t k = first; k < n; k += stride) {
        uint32_t t = off + k;
        if (t < K.seg_cap) {
            const uint32_t e = rb + (k % cnt);
            const bool pre = have0 && e == rb;
            const uint32_t row = pre ? row0 : K.rows[e], gc = pre ? gc0 : K.rows_gc[e];
            uint32_t lod_e = lod;
            if (K.gc_lodtab && gc != 0xFFFFFFFFu) { const uint32_t tb = K.gc_lodtab[gc]; if (tb) lod_e = lod_index(dist, K.lod_n[tb], K.lod_min + tb * 8u, K.lod_max + tb * 8u); }   // level_views.custom
            const uint32_t slot = gc == 0xFFFFFFFFu ? 0xFFFFFFFFu : gc * 8u + lod_e;
            K.item_row[seg_base + t] = row;
            if (K.slot_write_through) __hip_atomic_store(&K.item_slot[seg_base + t], slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // (wave-uniform: a kernel argument)
            else K.item_slot[seg_base + t] = slot;
            if (K.group_count && slot < K.count_nslots) { if (hist) atomicAdd(&hist[slot], 1u); else atomicAdd(&K.group_count[shard * K.count_nslots + slot], 1u); }
        }
    }
}

// Every lane brings up to EMIT_MAX visible sections {row range, count, lod | multiplicity << 8}.  ONE 64-bit
// atomic per call reserves their instances.
// dist_lds: with custom level-of-view bands, the distance of the section in slot j of lane l at dist_lds[j * 64 + l] (wave-private LDS, float bits), so that no
// register stays live for it across the reservation; dist0: the same for callers that hold one section per lane in registers (shared sections)
__device__ __forceinline__ void emit_sections_multi(const uint32_t (&rb)[EMIT_MAX], const uint32_t (&cnt)[EMIT_MAX], const uint32_t (&lodm)[EMIT_MAX],
                                                    FrameHeader *hdr, const ItemSink &K, uint32_t shard_hint, uint32_t *hist = nullptr,
                                                    const uint32_t *dist_lds = nullptr, float dist0 = 0.0f) {
    uint32_t mine = 0, nsec = 0;
#pragma unroll
    for (uint32_t j = 0; j < EMIT_MAX; j++) { uint32_t n = cnt[j] * ((lodm[j] >> 8) & 3u); mine += n; nsec += n ? 1u : 0u; }
    uint64_t mask = __ballot(mine > 0);
    if (!mask) return;
    // the first pool entry of every section is requested before the reservation: its round trip overlaps the atomic's
    uint32_t row0[EMIT_MAX], gc0[EMIT_MAX];
#pragma unroll
    for (uint32_t j = 0; j < EMIT_MAX; j++) { const bool on = cnt[j] != 0; row0[j] = on ? K.rows[rb[j]] : 0u; gc0[j] = on ? K.rows_gc[rb[j]] : 0u; }
    uint32_t incl = wave_incl_scan(mine), incs = wave_incl_scan(nsec);
    uint32_t tot = __shfl(incl, 63, 64), tots = __shfl(incs, 63, 64);
    const uint32_t shard = K.nshards > 1u ? (shard_hint & (CURSOR_SHARDS - 1u)) : 0u;      // wave-uniform (list / section-block index)
    unsigned long long base = 0;
    if (lane_id() == 0) base = atomicAdd(&hdr->cursors[shard * CURSOR_STRIDE], (unsigned long long)tots | ((unsigned long long)tot << 32));
    base = __shfl(base, 0, 64);
    uint32_t off = (uint32_t)(base >> 32) + (incl - mine);
    const uint32_t seg_base = shard * K.seg_cap;
    const uint32_t WIDE = 32u;
#pragma unroll
    for (uint32_t j = 0; j < EMIT_MAX; j++) {
        const uint32_t m = (lodm[j] >> 8) & 3u, lod = lodm[j] & 7u, n = cnt[j] * m;
        const float dj = (!K.gc_lodtab || !n) ? 0.0f : (dist_lds ? __uint_as_float(dist_lds[j * 64u + lane_id()]) : dist0);
        if (n && n <= WIDE) expand_rows(rb[j], cnt[j], 0u, 1u, n, off, lod, seg_base, K, hist, shard, dj, true, row0[j], gc0[j]);
        uint64_t wide = __ballot(n > WIDE);
        while (wide) {                                          // wave-cooperative expansion of crowded sections
            int src = __ffsll((long long)wide) - 1; wide &= wide - 1;
            expand_rows(__shfl(rb[j], src, 64), __shfl(cnt[j], src, 64), lane_id(), 64u, __shfl(n, src, 64), __shfl(off, src, 64), __shfl(lod, src, 64), seg_base, K, hist, shard, __shfl(dj, src, 64));
        }
        off += n;
    }
}
__device__ __forceinline__ void emit_sections(uint32_t rb0, uint32_t cnt0, uint32_t lodm0, FrameHeader *hdr, const ItemSink &K, uint32_t shard_hint, float dist) {
    uint32_t rb[EMIT_MAX] = {}, cnt[EMIT_MAX] = {}, lodm[EMIT_MAX] = {};
    rb[0] = rb0; cnt[0] = cnt0; lodm[0] = lodm0;
    emit_sections_multi(rb, cnt, lodm, hdr, K, shard_hint, nullptr, nullptr, dist);
}

// ---------------------------------------------------------------------------------------------
// K1: visibility query over the spatial hash == VisibleWorldFlow::find_visible_world_ids for both
// cullers (flows/visible_world_flow.rs:40-146, flows/pipeline.rs:216-229) fused with the
// per-section part of RenderFlow (distance test, LOD; render_flow.rs:475-492, 749-754).
//
// The reference enumerates every candidate id of a camera-centred box per level and probes a hash
// map.  Here the occupied sections are one key-sorted array: a section is a candidate iff its own
// index lies inside the box of its level, so the query is ONE streaming pass over the 8-byte keys
// (the only per-section bytes read for non-candidates).
//
// ONE launch (k_scan_cull).  Every wave streams 512 keys and runs the packed 16-bit box tests; the few
// waves that hold a candidate (about 1 in 20 at the reference's draw distance) continue in place with
// the exact predicates and the instance expansion, while the rest of the grid keeps streaming around
// them.  Nothing is handed over through HBM between the two phases and no launch boundary, completion
// ticket or fence sits between them; the dependent-load chain of the candidate waves (about four memory
// round trips) hides under the stream.
// ---------------------------------------------------------------------------------------------
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pk_sub_u16(uint32_t a, uint32_t b) { u16x2 r = __builtin_bit_cast(u16x2, a) - __builtin_bit_cast(u16x2, b); return __builtin_bit_cast(uint32_t, r); }
__device__ __forceinline__ uint32_t pk_min_u16(uint32_t a, uint32_t b) { u16x2 r = __builtin_elementwise_min(__builtin_bit_cast(u16x2, a), __builtin_bit_cast(u16x2, b)); return __builtin_bit_cast(uint32_t, r); }
__device__ __forceinline__ bool pk_in_box(uint32_t hi, uint32_t lo, const PBox &b) {
    uint32_t dh = pk_sub_u16(hi, b.sub_hi), dl = pk_sub_u16(lo, b.sub_lo);
    return (pk_min_u16(dh, b.min_hi) == dh) && (pk_min_u16(dl, b.min_lo) == dl);
}
__device__ __forceinline__ bool in_box(uint32_t x, uint32_t y, uint32_t z, const LevelBox &b) {
    return ((x - b.bx) & 0xFFFFu) < b.nx && ((y - b.by) & 0xFFFFu) < b.ny && ((z - b.bz) & 0xFFFFu) < b.nz;
}

// Visibility of one world section of level-box pair (a = logic, b = render): 0 = not visible, 1 = in one of the two
// query results, 2 = in both (the section then appears twice in visible_sections_vec).  *candidate: inside a candidate box.
__device__ __forceinline__ uint32_t section_multiplicity_boxes(uint64_t key, const LevelBox &a, const LevelBox &b, const FrameParams &P, bool *candidate) {
    uint32_t x = key_x(key), y = key_y(key), z = key_z(key);
    bool inl = in_box(x, y, z, a), inr = in_box(x, y, z, b);
    if (candidate) *candidate = inl | inr;
    // candidate AABB as visible_world_flow.rs:73-82: base = (base_unique + i) as f32 * level_length
    float ll = a.level_length;
    bool visl = false, visr = false;
    if (inl) {
        float fx = (float)(a.bx + ((x - a.bx) & 0xFFFFu)) * ll, fy = (float)(a.by + ((y - a.by) & 0xFFFFu)) * ll, fz = (float)(a.bz + ((z - a.bz) & 0xFFFFu)) * ll;
        Aabb g = { fx, fx + ll, fy, fy + ll, fz, fz + ll };
        visl = logic_aabb_in_view(P.lookahead, P.cam[0], P.cam[1], P.cam[2], g);
    }
    if (inr) {
        float fx = (float)(b.bx + ((x - b.bx) & 0xFFFFu)) * ll, fy = (float)(b.by + ((y - b.by) & 0xFFFFu)) * ll, fz = (float)(b.bz + ((z - b.bz) & 0xFFFFu)) * ll;
        Aabb g = { fx, fx + ll, fy, fy + ll, fz, fz + ll };
        visr = frustum_aabb_visible(P.planes, g);
    }
    return (visl && visr) ? 2u : ((visl || visr) ? 1u : 0u);
}
__device__ __forceinline__ uint32_t section_multiplicity(uint64_t key, const FrameParams &P, bool *candidate) {
    uint32_t lv = key_level(key);
    if (lv >= P.max_level) { if (candidate) *candidate = false; return 0u; }
    LevelBox a = P.box[0][lv], b = P.box[1][lv];
    return section_multiplicity_boxes(key, a, b, P, candidate);
}

__device__ __forceinline__ void cull_shared_section(uint32_t s, const SharedArrays &S, const uint64_t *__restrict__ cell_key, const uint8_t *__restrict__ cell_flags,
                                                    const Aabb *__restrict__ cell_tight, const ItemSink &K, FrameHeader *hdr, const FrameParams &P);

// The streaming part is lean on purpose: every wave issues its CULL_ITERS x 16-byte key loads immediately and runs the
// packed tests (~12 VALU per key) against the two boxes of its level held in SGPRs.  Level runs are padded to whole
// wave chunks on the host, so the level of the first key is the level of every real key of the wave; keys of any other
// level fail the packed test by themselves.  Frame parameters live in the kernel-argument segment and are read with
// scalar loads by candidate waves only.
// The body of the scan + cull kernel for workgroup `bid` of `nblk` (the kernel's own grid, or the scan part of the fused launch below).
// Force-inlined into its kernels: `A` is the kernel's by-value argument, and the kernel-argument offsets used below are those of the
// common leading signature.
template <bool K32, bool EAGER = false>
__device__ __forceinline__ void scan_cull_body(const uint32_t bid, const uint32_t nblk, const void *__restrict__ keys, uint32_t ncells, uint32_t nsp, uint32_t s0, uint32_t c0, uint32_t s1, uint32_t c1,
                                               uint32_t s2, uint32_t c2, uint32_t s3, uint32_t c3, const uint32_t *__restrict__ chunk_level, const ScanCullArgs &A) {
    constexpr uint32_t WK = WAVE_KEYS, NB = WK / 64u, NLD = K32 ? 2u : CULL_ITERS;   // keys per wave; ballots per wave (one per key a lane holds); 16-byte loads per lane
    // per-wave candidate list (no barrier: wave-private): section index + its key (one word compact, two words full); the visible
    // sections are compacted in place over its front
    __shared__ uint32_t s_idx[CULL_THREADS / 64][WK], s_key[CULL_THREADS / 64][K32 ? WK : 2u * WK];
#ifdef RE_EXP_STAMPS
    const unsigned long long tl_start = wall_clock64(); unsigned long long tl_keys = tl_start, tl_pred = 0, tl_emit = 0; uint32_t tl_cand = 0;
#endif
    // workgroup -> key chunk: the candidate spans first (see ScanSpans); all scalar
    uint32_t chunk = bid;
    {
        const uint32_t st[4] = { s0, s1, s2, s3 }, ct[4] = { c0, c1, c2, c3 };
        uint32_t acc = 0; bool in_span = false;
#pragma unroll
        for (uint32_t i = 0; i < 4; i++)
            if (i < nsp && !in_span) { if (chunk < acc + ct[i]) { chunk = st[i] + (chunk - acc); in_span = true; } else acc += ct[i]; }
        if (!in_span) {
            chunk -= acc;
#pragma unroll
            for (uint32_t i = 0; i < 4; i++) if (i < nsp && chunk >= st[i]) chunk += ct[i];
        }
    }
    const uint32_t lane = lane_id(), wid = threadIdx.x >> 6, wave = chunk * (CULL_THREADS / 64) + wid;
    // the stream: 512 keys per wave, as 4 x 16 B (full 64-bit keys) or 2 x 16 B (compact 32-bit keys) per lane; one ballot per key a lane holds
    const uint32_t wave_key0 = wave * WK;
    if (wave_key0 < ncells) {                                               // wave-uniform
        uint64_t m[NB]; uint64_t any = 0; uint32_t lv0, qn = 0;
        uint32_t *q_idx = s_idx[wid], *q_key = s_key[wid];
        if constexpr (K32) {
            const uint32_t nquads = (ncells + 3u) >> 2;                      // the compact array is padded to whole quads with padding keys
            const uint4 *kp = reinterpret_cast<const uint4 *>(keys);
            uint4 kk[NLD];
#pragma unroll
            for (uint32_t it = 0; it < NLD; it++) { uint32_t q = (wave_key0 >> 2) + it * 64u + lane; kk[it] = kp[q < nquads ? q : nquads - 1u]; }
            const uint32_t lv_word = chunk_level[__builtin_amdgcn_readfirstlane(wave)];      // scalar load, in flight together with the keys
            // EAGER (k_scan_cull_wide, frames with a large visible set): nothing that consumes a key is scheduled between the three requests.  Left to itself the compiler puts the
            // first use of load 0's keys in front of load 1 (one key load in flight per wave, the level word behind the first) -- the faster stream when 1 % of the waves hold
            // candidates (far = 1000: scan 12.2 vs 12.8 us), the slower one when half of them do (far = 8192: 29.3 vs 28.4 us).  profiles/r03_k1_code_layout.log
            if constexpr (EAGER) __builtin_amdgcn_sched_barrier(0);
            lv0 = lv_word & (MAX_LEVELS - 1);
            const PBox32 a0 = A.B32.box[lv0];
            const uint32_t hig = a0.hi | KEY32_GUARDS;
#pragma unroll
            for (uint32_t it = 0; it < NLD; it++) {
                const bool valid = ((wave_key0 >> 2) + it * 64u + lane) < nquads;
                const uint32_t v4[4] = { kk[it].x, kk[it].y, kk[it].z, kk[it].w };
#pragma unroll
                for (int h = 0; h < 4; h++) {
                    const uint32_t v = v4[h];
                    const uint32_t in = ((v | KEY32_GUARDS) - a0.lo) & (hig - v) & KEY32_GUARDS;      // 5 VALU per key; padding keys carry bit 31
                    const bool c = valid && in == KEY32_GUARDS && (int32_t)v >= 0;
                    m[it * 4 + h] = __ballot(c); any |= m[it * 4 + h];
                }
            }
            if (any) {                                                      // wave-uniform: stash the candidates (index, key) while the keys are in registers
#pragma unroll
                for (uint32_t it = 0; it < NLD; it++) {
                    const uint32_t v4[4] = { kk[it].x, kk[it].y, kk[it].z, kk[it].w };
#pragma unroll
                    for (uint32_t h = 0; h < 4; h++) {
                        const uint64_t mk = m[it * 4 + h];
                        if ((mk >> lane) & 1ull) { const uint32_t pos = qn + mbcnt(mk); q_idx[pos] = wave_key0 + (it * 64u + lane) * 4u + h; q_key[pos] = v4[h]; }
                        qn += (uint32_t)__popcll(mk);
                    }
                }
            }
        } else {
            const uint32_t npairs = (ncells + 1u) >> 1;                      // key array is padded to an even count with never-candidate keys
            const uint32_t wave_pair0 = wave * (64u * CULL_ITERS);
            const ulonglong2 *kp = reinterpret_cast<const ulonglong2 *>(keys);
            ulonglong2 kk[CULL_ITERS];
#pragma unroll
            for (uint32_t it = 0; it < CULL_ITERS; it++) {
                uint32_t pair = wave_pair0 + it * 64u + lane;
                kk[it] = kp[pair < npairs ? pair : npairs - 1u];
            }
            lv0 = __builtin_amdgcn_readfirstlane(key_level(kk[0].x)) & (MAX_LEVELS - 1);
            const PBox a0 = A.B.box[0][lv0], b0 = A.B.box[1][lv0];           // uniform index: scalar loads into SGPRs
#pragma unroll
            for (uint32_t it = 0; it < CULL_ITERS; it++) {
                const bool valid = (wave_pair0 + it * 64u + lane) < npairs;
#pragma unroll
                for (int h = 0; h < 2; h++) {
                    uint64_t key = h ? kk[it].y : kk[it].x;
                    uint32_t hi = (uint32_t)(key >> 32), lo = (uint32_t)key;
                    bool c = valid && (pk_in_box(hi, lo, a0) || pk_in_box(hi, lo, b0));
                    m[it * 2 + h] = __ballot(c); any |= m[it * 2 + h];
                }
            }
            if (any) {
#pragma unroll
                for (uint32_t it = 0; it < CULL_ITERS; it++)
#pragma unroll
                    for (uint32_t h = 0; h < 2; h++) {
                        const uint64_t mk = m[it * 2 + h], key = h ? kk[it].y : kk[it].x;
                        if ((mk >> lane) & 1ull) { const uint32_t pos = qn + mbcnt(mk); q_idx[pos] = wave_key0 + (it * 64u + lane) * 2u + h; q_key[2u * pos] = (uint32_t)key; q_key[2u * pos + 1u] = (uint32_t)(key >> 32); }
                        qn += (uint32_t)__popcll(mk);
                    }
            }
        }
#ifdef RE_EXP_STAMPS
        tl_keys = wall_clock64(); tl_cand = any ? 1u : 0u;
#endif
#ifdef RE_EXP_STAGES
        if (A.P.pad & 8u) any = 0;                                          // (tools/stage_stop.py: the stream and its tests alone -- costs every wave a scalar load, development builds only)
#endif
        if (__builtin_expect(any != 0ull, 0)) {                             // wave-uniform (scalar) branch: ~1% of the waves (laid out as the cold path: the 99% run straight through a few cache lines of code)
            // Candidate waves read the rest of the kernel-argument segment through a pointer the compiler cannot see through, so that
            // none of those scalar loads is hoisted in front of the key loads of the other 99%.
            typedef __attribute__((address_space(4))) const char *kernarg_ptr;
            kernarg_ptr ka = (kernarg_ptr)__builtin_amdgcn_kernarg_segment_ptr();
            asm volatile("" : "+s"(ka));
            const ScanCullArgs &R = *(const ScanCullArgs *)(ka + SCAN_CULL_ARGS_OFFSET);
            const FrameParams &P = R.P;
            const ItemSink K = R.K; FrameHeader *hdr = R.hdr;
            const Aabb *__restrict__ cell_tight = R.cell_tight; const uint32_t *__restrict__ cell_begin = R.cell_begin, *__restrict__ cell_nlocal = R.cell_nlocal, *__restrict__ cell_nstatic = R.cell_nstatic, *__restrict__ cell_nghost = R.cell_nghost;
            const uint8_t *__restrict__ cell_flags = R.cell_flags; uint32_t *__restrict__ cell_stamp = R.cell_stamp;
            if (lv0 < P.max_level && !R.spec->stale) {                       // (a stale tree cancels the frame: see SpecState)
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                const LevelBox la = P.box[0][lv0], lb = P.box[1][lv0];      // scalar loads from the kernel-argument segment
                uint32_t vis_map_acc = 0, vis_vec_acc = 0, cand_acc = 0, nv = 0;
                // in-scan counting of a large visible set (ItemSink::group_count): this wave's histogram of group slots in dynamic LDS
                extern __shared__ uint32_t s_dyn[];
                uint32_t *hist = K.group_count ? s_dyn + wid * K.count_nslots : nullptr;
                if (hist) for (uint32_t i = lane; i < K.count_nslots; i += 64u) hist[i] = 0u;
                // stage A -- pure arithmetic on the keys (no memory round trip): the exact box tests and the two cullers on the section's grid
                // box; the visible sections (index | multiplicity << 30) are compacted in place over the front of the list
#pragma unroll 1
                for (uint32_t base = 0; base < qn; base += 64u) {
                    const uint32_t i = base + lane; const bool on = i < qn;
                    const uint32_t idx = q_idx[on ? i : 0u];
                    uint64_t key;
                    if constexpr (K32) { const uint32_t v = q_key[on ? i : 0u]; key = pack_key(lv0, (v >> 20) & 0x1FFu, (v >> 10) & 0x1FFu, v & 0x1FFu); }
                    else { const uint32_t j2 = 2u * (on ? i : 0u); key = (uint64_t)q_key[j2] | ((uint64_t)q_key[j2 + 1u] << 32); }
                    const bool pad = !K32 && (key & 0xFFFFFFFFFFFFull) == 0xFFFFFFFFFFFFull;     // compact padding keys never get here (sign bit)
                    bool is_cand = false;
                    const uint32_t mult = (!on || pad) ? 0u : section_multiplicity_boxes(key, la, lb, P, &is_cand);
                    cand_acc += (on && !pad && is_cand) ? 1u : 0u;
                    const uint64_t vb = __ballot(mult != 0u);
                    if (mult) q_idx[nv + mbcnt(vb)] = idx | (mult << 30);   // positions < base + 64: only entries every lane has already read
                    nv += (uint32_t)__popcll(vb);
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#ifdef RE_EXP_STAMPS
                tl_pred = wall_clock64();
#endif
#ifdef RE_EXP_STAGES
                if (P.pad & 2u) nv = 0;                                     // (tools/stage_stop.py: the launch without stage B)
#endif
                // stage B -- the visible sections only (usually one round of 64): everything indexed by the section in one memory round trip,
                // then distance, LOD, active / cached-static row ranges, and the instance expansion
#pragma unroll 1
                for (uint32_t vbase = 0; vbase < nv; vbase += EMIT_MAX * 64u) {
                    uint32_t rbv[EMIT_MAX], cntv[EMIT_MAX], lodv[EMIT_MAX];
#pragma unroll
                    for (uint32_t j = 0; j < EMIT_MAX; j++) {
                        const uint32_t i = vbase + j * 64u + lane;
                        rbv[j] = 0; cntv[j] = 0; lodv[j] = 0;
                        if (vbase + j * 64u < nv) {                         // wave-uniform
                            const bool on = i < nv;
                            const uint32_t e = q_idx[on ? i : 0u], c = e & 0x3FFFFFFFu, mult = e >> 30;
                            const uint8_t f = cell_flags[c];
                            const Aabb t = cell_tight[c];
                            const uint32_t nl = cell_nlocal[c], ns = cell_nstatic[c] + cell_nghost[c], cb = cell_begin[c];   // ghosts: snapshot copies the cached data still holds, stored behind the static rows
                            if (on && !(f & CF_PAD)) {
                                cell_stamp[c] = (P.frame << 2) | mult;
                                vis_map_acc += 1; vis_vec_acc += mult;
                                float d = distance_to_aabb(t, P.cam[0], P.cam[1], P.cam[2]);
                                bool act = !(f & CF_STATIC_SECTION) && (d < P.far_draw);     // is_section_active && render_flow.rs:754
                                bool sta = (f & CF_STATIC_CACHED) && !(d > P.far_draw);      // cached && render_flow.rs:489
                                rbv[j] = cb + (act ? 0u : nl);
                                cntv[j] = (act ? nl : 0u) + (sta ? ns : 0u);
                                uint32_t mm = P.emit_duplicates ? mult : 1u;
                                lodv[j] = lod_index(d, P.n_lod, P.lod_min, P.lod_max) | (mm << 8);
                                if (K.gc_lodtab) q_key[j * 64u + lane] = __float_as_uint(d);       // (the keys are no longer needed: stage A is done)
                            }
                        }
                    }
#ifdef RE_EXP_STAGES
                    if (P.pad & 4u) { for (uint32_t j = 0; j < EMIT_MAX; j++) cntv[j] = 0; }      // (tools/stage_stop.py: stage B without the instance expansion)
#endif
                    emit_sections_multi(rbv, cntv, lodv, hdr, K, wave, hist, K.gc_lodtab ? q_key : nullptr);     // one reservation per slice of 256 visible sections
                }
                if (hist && nv) {                                           // flush: one atomic per non-empty group slot and wave, into the counts of this wave's cursor shard
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                    uint32_t *gc = K.group_count + (K.nshards > 1u ? (wave & (CURSOR_SHARDS - 1u)) : 0u) * K.count_nslots;
                    for (uint32_t i = lane; i < K.count_nslots; i += 64u) { const uint32_t v = hist[i]; if (v) atomicAdd(&gc[i], v); }
                }
#ifdef RE_EXP_STAMPS
                tl_emit = wall_clock64();
#endif
                for (int d = 32; d >= 1; d >>= 1) { vis_map_acc += __shfl_down(vis_map_acc, d, 64); vis_vec_acc += __shfl_down(vis_vec_acc, d, 64); cand_acc += __shfl_down(cand_acc, d, 64); }
                if (lane == 0) {                                            // sharded frame counters: fire-and-forget atomics, <= 3 per candidate wave
                    uint32_t *cnt = hdr->counters + (wave & (COUNTER_SHARDS - 1u)) * 16u;
                    if (cand_acc) atomicAdd(cnt + 0, cand_acc);
                    if (vis_map_acc) { atomicAdd(cnt + 1, vis_map_acc); atomicAdd(cnt + 2, vis_vec_acc); }
                }
            }
        }
    }
    // shared world sections: their visibility follows from the keys of the sections linking them, so the first
    // workgroups process them next to their share of the stream.  The last workgroup (it owns the short tail of the
    // key array) stages the frame parameters into device memory for the tick.
    {
        typedef __attribute__((address_space(4))) const char *kernarg_ptr;
        kernarg_ptr ka = (kernarg_ptr)__builtin_amdgcn_kernarg_segment_ptr();
        asm volatile("" : "+s"(ka));
        const ScanCullArgs &R = *(const ScanCullArgs *)(ka + SCAN_CULL_ARGS_OFFSET);
        const uint32_t nsh = R.S.n;
        if (__builtin_expect(bid * CULL_THREADS < nsh, 0) && !R.spec->stale) {
            const SharedArrays S = R.S; const ItemSink K = R.K;
            for (uint32_t s0 = bid * CULL_THREADS; s0 < nsh; s0 += nblk * CULL_THREADS)
                cull_shared_section(s0 + threadIdx.x, S, R.cell_key64, R.cell_flags, R.cell_tight, K, R.hdr, R.P);
        }
        if (__builtin_expect(bid == nblk - 1u, 0) && !R.spec->stale) {
            const uint32_t *src = reinterpret_cast<const uint32_t *>(&R.P); uint32_t *dst = reinterpret_cast<uint32_t *>(R.P_dev);
            for (uint32_t i = threadIdx.x; i < sizeof(FrameParams) / 4u; i += CULL_THREADS) dst[i] = src[i];
        }
#ifdef RE_EXP_STAMPS
        if (R.timeline && lane == 0) { unsigned long long *t = R.timeline + (size_t)wave * 8u; t[0] = tl_start; t[1] = tl_keys; t[2] = wall_clock64(); t[3] = tl_cand; t[4] = tl_pred; t[5] = tl_emit; }
#endif
    }
}

template <bool K32>
__global__ __launch_bounds__(CULL_THREADS) void k_scan_cull(const void *__restrict__ keys, uint32_t ncells, uint32_t nsp, uint32_t s0, uint32_t c0, uint32_t s1, uint32_t c1,
                                                            uint32_t s2, uint32_t c2, uint32_t s3, uint32_t c3, const uint32_t *__restrict__ chunk_level, ScanCullArgs A) {
    scan_cull_body<K32>(blockIdx.x, gridDim.x, keys, ncells, nsp, s0, c0, s1, c1, s2, c2, s3, c3, chunk_level, A);
}
template __global__ void k_scan_cull<false>(const void *, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t, const uint32_t *, ScanCullArgs);
template __global__ void k_scan_cull<true>(const void *, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t, const uint32_t *, ScanCullArgs);
// the same kernel for frames with a large visible set (compact keys): both key loads of a wave and its level word requested together (scan_cull_body: EAGER)
__global__ __launch_bounds__(CULL_THREADS) void k_scan_cull_wide(const void *__restrict__ keys, uint32_t ncells, uint32_t nsp, uint32_t s0, uint32_t c0, uint32_t s1, uint32_t c1,
                                                                 uint32_t s2, uint32_t c2, uint32_t s3, uint32_t c3, const uint32_t *__restrict__ chunk_level, ScanCullArgs A) {
    scan_cull_body<true, true>(blockIdx.x, gridDim.x, keys, ncells, nsp, s0, c0, s1, c1, s2, c2, s3, c3, chunk_level, A);
}

// ---------------------------------------------------------------------------------------------
// Probe path (RE_CFG_PROBE): the same stage A / stage B as k_scan_cull, fed by hash probes of the candidate cells
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t key_hash(unsigned long long x, uint32_t mask) { x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 29; return (uint32_t)x & mask; }
__global__ __launch_bounds__(256) void k_hash_build(uint32_t ncells, const uint64_t *__restrict__ cell_key, HashEntry *tab, uint32_t mask) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= ncells) return;
    const unsigned long long k = cell_key[i];
    if ((k & 0xFFFFFFFFFFFFull) == 0xFFFFFFFFFFFFull) return;                // padding / spare slot
    for (uint32_t h = key_hash(k, mask), probe = 0; probe <= mask; probe++, h = (h + 1u) & mask) {
        const unsigned long long prev = atomicCAS(&tab[h].key, ~0ull, k);
        if (prev == ~0ull || prev == k) { tab[h].slot = i; return; }
    }
}
// in-place table patches: pass 0 retires the old key of every slot whose key changes, pass 1 enters the new keys (two launches, so a
// section that leaves one slot and appears in another within one batch ends up present)
__global__ __launch_bounds__(256) void k_hash_patch(uint32_t m, const Pair64 *__restrict__ slot_newkey, const uint64_t *__restrict__ cell_key_old, HashEntry *tab, uint32_t mask, uint32_t insert_pass) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    const uint32_t slot = slot_newkey[i].idx;
    const unsigned long long k = insert_pass ? slot_newkey[i].val : cell_key_old[slot];
    if ((k & 0xFFFFFFFFFFFFull) == 0xFFFFFFFFFFFFull) return;
    for (uint32_t h = key_hash(k, mask), probe = 0; probe <= mask; probe++, h = (h + 1u) & mask) {
        if (insert_pass) {
            const unsigned long long prev = atomicCAS(&tab[h].key, ~0ull, k);
            if (prev == ~0ull || prev == k) { tab[h].slot = slot; return; }
        } else {
            const unsigned long long cur = tab[h].key;
            if (cur == k) { if (tab[h].slot == slot) tab[h].slot = 0xFFFFFFFFu; return; }
            if (cur == ~0ull) return;
        }
    }
}

__global__ __launch_bounds__(CULL_THREADS) void k_probe_cull(ProbeArgs Q, ScanCullArgs R) {
    constexpr uint32_t WK = PROBE_KEYS, NIT = WK / 64u;
    __shared__ uint32_t s_idx[CULL_THREADS / 64][WK], s_key[CULL_THREADS / 64][2u * WK];
    const uint32_t lane = lane_id(), wid = threadIdx.x >> 6, wave = blockIdx.x * (CULL_THREADS / 64) + wid;
    const FrameParams &P = R.P;
    if (wave < Q.nwaves && !R.spec->stale) {                                 // wave-uniform
        uint32_t lv0 = 0;
        for (uint32_t l = 1; l < (uint32_t)MAX_LEVELS; l++) if (wave >= Q.wave0[l]) lv0 = l;     // wave0[] is non-decreasing; levels without cells own no wave
        const LevelBox ub = Q.ubox[lv0];
        const uint32_t cells = ub.nx * ub.ny * ub.nz, t0 = (wave - Q.wave0[lv0]) * WK;
        uint32_t *q_idx = s_idx[wid], *q_key = s_key[wid];
        unsigned long long key[NIT]; HashEntry e[NIT]; uint32_t h[NIT]; bool valid[NIT];
#pragma unroll
        for (uint32_t it = 0; it < NIT; it++) {                             // all first probes in flight together
            const uint32_t t = t0 + it * 64u + lane;
            valid[it] = t < cells;
            const uint32_t tt = valid[it] ? t : 0u, iy = tt % ub.ny, r = tt / ub.ny, iz = r % ub.nz, ix = r / ub.nz;
            key[it] = pack_key(lv0, (ub.bx + ix) & 0xFFFFu, (ub.bz + iz) & 0xFFFFu, (ub.by + iy) & 0xFFFFu);
            h[it] = key_hash(key[it], Q.mask);
            e[it] = Q.tab[h[it]];
        }
        uint32_t qn = 0;
#pragma unroll
        for (uint32_t it = 0; it < NIT; it++) {
            while (valid[it] && e[it].key != key[it] && e[it].key != ~0ull) { h[it] = (h[it] + 1u) & Q.mask; e[it] = Q.tab[h[it]]; }   // collisions: rare at load 0.5
            const bool found = valid[it] && e[it].key == key[it] && e[it].slot != 0xFFFFFFFFu;
            const uint64_t mk = __ballot(found);
            if (found) { const uint32_t pos = qn + mbcnt(mk); q_idx[pos] = e[it].slot; q_key[2u * pos] = (uint32_t)key[it]; q_key[2u * pos + 1u] = (uint32_t)(key[it] >> 32); }
            qn += (uint32_t)__popcll(mk);
        }
        if (qn && lv0 < P.max_level) {
            const ItemSink K = R.K; FrameHeader *hdr = R.hdr;
            const Aabb *__restrict__ cell_tight = R.cell_tight; const uint32_t *__restrict__ cell_begin = R.cell_begin, *__restrict__ cell_nlocal = R.cell_nlocal, *__restrict__ cell_nstatic = R.cell_nstatic, *__restrict__ cell_nghost = R.cell_nghost;
            const uint8_t *__restrict__ cell_flags = R.cell_flags; uint32_t *__restrict__ cell_stamp = R.cell_stamp;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            const LevelBox la = P.box[0][lv0], lb = P.box[1][lv0];
            uint32_t vis_map_acc = 0, vis_vec_acc = 0, cand_acc = 0, nv = 0;
            // stage A (as k_scan_cull): exact box tests and the two cullers on the section's grid box; visible sections compacted in place
#pragma unroll 1
            for (uint32_t base = 0; base < qn; base += 64u) {
                const uint32_t i = base + lane; const bool on = i < qn;
                const uint32_t idx = q_idx[on ? i : 0u];
                const uint32_t j2 = 2u * (on ? i : 0u);
                const uint64_t k = (uint64_t)q_key[j2] | ((uint64_t)q_key[j2 + 1u] << 32);
                bool is_cand = false;
                const uint32_t mult = !on ? 0u : section_multiplicity_boxes(k, la, lb, P, &is_cand);
                cand_acc += (on && is_cand) ? 1u : 0u;
                const uint64_t vb = __ballot(mult != 0u);
                if (mult) q_idx[nv + mbcnt(vb)] = idx | (mult << 30);
                nv += (uint32_t)__popcll(vb);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            // stage B (as k_scan_cull): the visible sections
#pragma unroll 1
            for (uint32_t vbase = 0; vbase < nv; vbase += EMIT_MAX * 64u) {
                uint32_t rbv[EMIT_MAX], cntv[EMIT_MAX], lodv[EMIT_MAX];
#pragma unroll
                for (uint32_t j = 0; j < EMIT_MAX; j++) {
                    const uint32_t i = vbase + j * 64u + lane;
                    rbv[j] = 0; cntv[j] = 0; lodv[j] = 0;
                    if (vbase + j * 64u < nv) {
                        const bool on = i < nv;
                        const uint32_t en = q_idx[on ? i : 0u], c = en & 0x3FFFFFFFu, mult = en >> 30;
                        const uint8_t f = cell_flags[c];
                        const Aabb t = cell_tight[c];
                        const uint32_t nl = cell_nlocal[c], ns = cell_nstatic[c] + cell_nghost[c], cb = cell_begin[c];
                        if (on && !(f & CF_PAD)) {
                            cell_stamp[c] = (P.frame << 2) | mult;
                            vis_map_acc += 1; vis_vec_acc += mult;
                            float d = distance_to_aabb(t, P.cam[0], P.cam[1], P.cam[2]);
                            bool act = !(f & CF_STATIC_SECTION) && (d < P.far_draw);
                            bool sta = (f & CF_STATIC_CACHED) && !(d > P.far_draw);
                            rbv[j] = cb + (act ? 0u : nl);
                            cntv[j] = (act ? nl : 0u) + (sta ? ns : 0u);
                            uint32_t mm = P.emit_duplicates ? mult : 1u;
                            lodv[j] = lod_index(d, P.n_lod, P.lod_min, P.lod_max) | (mm << 8);
                            if (K.gc_lodtab) q_key[j * 64u + lane] = __float_as_uint(d);
                        }
                    }
                }
                emit_sections_multi(rbv, cntv, lodv, hdr, K, wave, nullptr, K.gc_lodtab ? q_key : nullptr);
            }
            for (int d = 32; d >= 1; d >>= 1) { vis_map_acc += __shfl_down(vis_map_acc, d, 64); vis_vec_acc += __shfl_down(vis_vec_acc, d, 64); cand_acc += __shfl_down(cand_acc, d, 64); }
            if (lane == 0) {
                uint32_t *cnt = hdr->counters + (wave & (COUNTER_SHARDS - 1u)) * 16u;
                if (cand_acc) atomicAdd(cnt + 0, cand_acc);
                if (vis_map_acc) { atomicAdd(cnt + 1, vis_map_acc); atomicAdd(cnt + 2, vis_vec_acc); }
            }
        }
    }
    // shared world sections and the frame parameters for the tick, as in k_scan_cull
    const uint32_t nsh = R.S.n;
    if (blockIdx.x * CULL_THREADS < nsh && !R.spec->stale) {
        const SharedArrays S = R.S; const ItemSink K = R.K;
        for (uint32_t s0 = blockIdx.x * CULL_THREADS; s0 < nsh; s0 += gridDim.x * CULL_THREADS)
            cull_shared_section(s0 + threadIdx.x, S, R.cell_key64, R.cell_flags, R.cell_tight, K, R.hdr, R.P);
    }
    if (blockIdx.x == gridDim.x - 1u && !R.spec->stale) {
        const uint32_t *src = reinterpret_cast<const uint32_t *>(&R.P); uint32_t *dst = reinterpret_cast<uint32_t *>(R.P_dev);
        for (uint32_t i = threadIdx.x; i < sizeof(FrameParams) / 4u; i += CULL_THREADS) dst[i] = src[i];
    }
}

// Shared world sections (render_flow.rs:808-866): emitted once per frame when some linking unique
// section is visible and active; static members through the unique section that cached them.
// Called by whole waves (lanes with s >= S.n contribute nothing).
__device__ __forceinline__ void cull_shared_section(uint32_t s, const SharedArrays &S, const uint64_t *__restrict__ cell_key, const uint8_t *__restrict__ cell_flags,
                                                    const Aabb *__restrict__ cell_tight, const ItemSink &K, FrameHeader *hdr, const FrameParams &P) {
    uint32_t rbA = 0, cntA = 0, lodA = 0, rbS = 0, cntS = 0, lodS = 0; float distA = 0.0f, distS = 0.0f;
    if (s < S.n) {
        bool act = false;
        uint32_t na = S.nact[s], ns = S.nstat[s], b = S.begin[s];
        if (na)
            for (int k = 0; k < 8; k++) {
                int32_t c = S.cells[s * 8 + k];
                if (c >= 0 && !act && !(cell_flags[c] & CF_STATIC_SECTION) && section_multiplicity(cell_key[c], P, nullptr)) act = true;
            }
        if (act) {
            float d2 = distance_to_aabb(S.aabb[s], P.cam[0], P.cam[1], P.cam[2]);
            if (d2 < P.far_draw) { rbA = b; cntA = na; lodA = lod_index(d2, P.n_lod, P.lod_min, P.lod_max) | (1u << 8); distA = d2; }
        }
        int32_t ow = S.owner[s];
        if (ns && ow >= 0 && S.cached[s]) {
            uint32_t mult = section_multiplicity(cell_key[ow], P, nullptr);
            if (mult) {
                float d = distance_to_aabb(cell_tight[ow], P.cam[0], P.cam[1], P.cam[2]);      // extract_static_data uses the unique section's distance
                if (!(d > P.far_draw)) { rbS = b + na; cntS = ns; uint32_t m = P.emit_duplicates ? mult : 1u; lodS = lod_index(d, P.n_lod, P.lod_min, P.lod_max) | (m << 8); distS = d; }
            }
        }
    }
    const uint32_t hint = __builtin_amdgcn_readfirstlane(s) >> 6;
    emit_sections(rbA, cntA, lodA, hdr, K, hint, distA);          // active members
    emit_sections(rbS, cntS, lodS, hdr, K, hint + 1u, distS);     // static members
}



// The optional header of a caller-provided output slab (re_set_output_count; the all-gather send slab of the multi-GPU exchange): 4 words
// {instances written, instances of the frame (before truncation to the slab), frame number, 0}.  A frame cancelled by cross-frame
// speculation (SpecState) marks the header instead of leaving the previous frame's numbers there for a collective to ship.
constexpr uint32_t SLAB_CANCELLED = 0xFFFFFFFFu;
__device__ __forceinline__ void write_slab_header(uint32_t *out_count, uint32_t total, uint32_t cap, uint32_t frame) {
    if (!out_count) return;
    out_count[0] = total < cap ? total : cap; out_count[1] = total; out_count[2] = frame; out_count[3] = 0u;
}
__device__ __forceinline__ void cancel_slab_header(uint32_t *out_count, uint32_t frame) {
    if (!out_count) return;
    out_count[0] = SLAB_CANCELLED; out_count[1] = 0u; out_count[2] = frame; out_count[3] = 0u;
}

// ---------------------------------------------------------------------------------------------
// K2a (large visible sets): per-group instance counts from the expanded item list.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_emit_count(const FrameHeader *hdr, const uint32_t *__restrict__ item_slot, uint32_t nshards, uint32_t seg_cap,
                                                    uint32_t *__restrict__ group_count, uint32_t nslots, const SpecState *spec) {
    extern __shared__ uint32_t s_hist[];
    if (spec->stale) return;
    const bool use_lds = nslots <= LDS_HIST_SLOTS;
    if (use_lds) { for (uint32_t i = threadIdx.x; i < nslots; i += blockDim.x) s_hist[i] = 0; __syncthreads(); }
    const ShardMap sm = load_shard_map(hdr, nshards, seg_cap);
    const uint32_t T = sm.total;
    for (uint32_t t = blockIdx.x * blockDim.x + threadIdx.x; t < T; t += gridDim.x * blockDim.x) {
        uint32_t slot = item_slot[shard_item_index(t, sm, seg_cap)];
        if (slot != 0xFFFFFFFFu) { if (use_lds) atomicAdd(&s_hist[slot], 1u); else atomicAdd(&group_count[slot], 1u); }
    }
    if (use_lds) {
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < nslots; i += blockDim.x) { uint32_t v = s_hist[i]; if (v) atomicAdd(&group_count[i], v); }
    }
}

__device__ __forceinline__ FrameCounts load_frame_counts(const FrameHeader *hdr);
// K2b: exclusive scan of the group counts -> InstanceRange table (upload_instance_data_to_render_system,
// render_flow.rs:964-983).  One workgroup; also resets the per-frame counters.
__global__ __launch_bounds__(1024) void k_group_scan(uint32_t *__restrict__ group_count, uint32_t *__restrict__ group_begin, uint32_t *__restrict__ group_fill,
                                                     uint32_t nslots, const uint32_t *__restrict__ gc_model, const uint32_t *__restrict__ gc_rs, const uint32_t *__restrict__ gc_sort,
                                                     InstanceRange *__restrict__ ranges, uint32_t range_cap, FrameHeader *hdr, FrameHeader *hdr_next, TickHeader *th, HostResult *hres, const SpecState *spec,
                                                     uint32_t *out_count, uint32_t out_cap, uint32_t frame, uint32_t seg_cap) {
    __shared__ uint32_t s_wsum[16], s_wcnt[16], s_whash[16];
    if (spec->stale) { if (threadIdx.x == 0) { HostResult r = {}; r.overflow = 2u; *hres = r; cancel_slab_header(out_count, frame); publish_to_host(&hres->done_frame, frame); } return; }
    __shared__ uint32_t s_carry, s_gcarry;
    if (threadIdx.x == 0) { s_carry = 0; s_gcarry = 0; }
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63u, wid = threadIdx.x >> 6;
    uint32_t my_hash = 0;                                       // hash of the table words this thread stores (see HostResult::table_hash)
    for (uint32_t base = 0; base < nslots; base += 1024) {
        uint32_t i = base + threadIdx.x;
        uint32_t v = i < nslots ? group_count[i] : 0u;
        uint32_t nz = v ? 1u : 0u;
        uint32_t incl = wave_incl_scan(v), incn = wave_incl_scan(nz);
        if (lane == 63) { s_wsum[wid] = incl; s_wcnt[wid] = incn; }
        __syncthreads();
        uint32_t woff = 0, wcn = 0;
        for (uint32_t w = 0; w < wid; w++) { woff += s_wsum[w]; wcn += s_wcnt[w]; }
        uint32_t begin = s_carry + woff + incl - v;
        uint32_t gidx = s_gcarry + wcn + incn - nz;
        if (i < nslots) {
            group_begin[i] = begin; group_fill[i] = 0; group_count[i] = 0;
            if (v && gidx < range_cap) {
                uint32_t gc = i >> 3, lod = i & 7u;
                InstanceRange r; r.model_index = gc_model[gc] | (lod << 25); r.render_system = gc_rs[gc]; r.sortable = gc_sort[gc]; r.begin = begin; r.count = v;
                ranges[gidx] = r;
                const uint32_t w0 = gidx * (uint32_t)(sizeof(InstanceRange) / 4u);
                my_hash ^= table_word_hash(r.model_index, w0) ^ table_word_hash(r.render_system, w0 + 1u) ^ table_word_hash(r.sortable, w0 + 2u) ^ table_word_hash(r.begin, w0 + 3u) ^ table_word_hash(r.count, w0 + 4u);
            }
        }
        __syncthreads();
        if (threadIdx.x == 1023) { s_carry = begin + v; s_gcarry = gidx + nz; }
        __syncthreads();
    }
    // every wave's table stores have left the wave before the barrier in front of the publication (publish_to_host)
    for (int d = 32; d >= 1; d >>= 1) my_hash ^= __shfl_xor(my_hash, d, 64);
    if (lane == 0) s_whash[wid] = my_hash;
    wait_own_stores();
    __syncthreads();
    FrameCounts fc = {};
    if (wid == 0) fc = load_frame_counts(hdr);
    if (threadIdx.x == 0) {
        uint32_t nsec = 0, nitems = 0, table_hash = 0; bool seg_over = false;
        for (uint32_t k = 0; k < CURSOR_SHARDS; k++) { unsigned long long cur = hdr->cursors[k * CURSOR_STRIDE]; nsec += (uint32_t)cur; nitems += (uint32_t)(cur >> 32); seg_over |= (uint32_t)(cur >> 32) > seg_cap; }
        for (uint32_t w = 0; w < 16u; w++) table_hash ^= s_whash[w];
        HostResult r = {}; r.n_vis_map = fc.n_vis_map; r.n_vis_vec = fc.n_vis_vec; r.n_groups = s_gcarry < range_cap ? s_gcarry : range_cap; r.total = s_carry; r.n_candidates = fc.n_candidates;
        r.overflow = seg_over ? RESULT_SEGMENT_OVERFLOW : 0u; r.n_entries = nsec; r.n_items = nitems;
        r.table_hash = result_seal(table_hash | 1u, frame, r.n_groups, r.total, r.n_vis_map, r.n_vis_vec, r.n_items);
        *hres = r;                                              // mapped pinned host memory
        if (seg_over) cancel_slab_header(out_count, frame); else write_slab_header(out_count, s_carry, out_cap, frame);
        publish_to_host(&hres->done_frame, frame);
    }
    for (uint32_t i = threadIdx.x; i < sizeof(FrameHeader) / 4u; i += 1024u) reinterpret_cast<uint32_t *>(hdr_next)[i] = 0u;   // next frame's cursor/counters
    for (uint32_t i = threadIdx.x; i < sizeof(TickHeader) / 4u; i += 1024u) reinterpret_cast<uint32_t *>(th)[i] = 0u;
}

// K2c: scatter -- the instance pack (specify_type_ids! callback + MappedBuffer::write_data_serialized,
// prelude/layout_update_macros.rs:15-21, render_components/mapped_buffer.rs:166-189).
// Per workgroup iteration 256 instances: (A) one lane per instance ranks it inside its group with an
// LDS histogram, one global atomic per (workgroup, group) reserves the slots; (B) 4 lanes per
// instance each move one float4 of the 64-byte column-major matrix, so a wave instruction
// reads/writes 16 whole 64-byte rows.
__global__ __launch_bounds__(256) void k_emit_scatter(const FrameHeader *hdr, const uint32_t *__restrict__ item_row, const uint32_t *__restrict__ item_slot, uint32_t nshards, uint32_t seg_cap,
                                                      const uint32_t *__restrict__ group_begin, uint32_t *__restrict__ group_fill, uint32_t nslots,
                                                      const uint32_t *__restrict__ row_id, const float *__restrict__ row_mat,
                                                      uint32_t *__restrict__ out_ids, float *__restrict__ out_mats, uint32_t out_cap, const SpecState *spec) {
    extern __shared__ uint32_t s_hist[];                       // [nslots] local counts, then the reserved bases
    if (spec->stale) return;
    __shared__ uint32_t s_pos[256], s_row[256];
    const bool use_lds = nslots <= LDS_HIST_SLOTS;
    const ShardMap sm = load_shard_map(hdr, nshards, seg_cap);
    const uint32_t T = sm.total;
    const uint32_t tid = threadIdx.x;
    for (uint32_t t0 = blockIdx.x * 256u; t0 < T; t0 += gridDim.x * 256u) {   // uniform trip count per workgroup
        uint32_t t = t0 + tid;
        uint32_t slot = 0xFFFFFFFFu, row = 0, rank = 0, pos = 0xFFFFFFFFu;
        if (t < T) { uint32_t ii = shard_item_index(t, sm, seg_cap); slot = item_slot[ii]; row = item_row[ii]; }
        if (use_lds) {
            for (uint32_t i = tid; i < nslots; i += 256u) s_hist[i] = 0;
            __syncthreads();
            if (slot != 0xFFFFFFFFu) rank = atomicAdd(&s_hist[slot], 1u);
            __syncthreads();
            for (uint32_t i = tid; i < nslots; i += 256u) { uint32_t v = s_hist[i]; if (v) s_hist[i] = atomicAdd(&group_fill[i], v); }
            __syncthreads();
            if (slot != 0xFFFFFFFFu) pos = group_begin[slot] + s_hist[slot] + rank;
        } else if (slot != 0xFFFFFFFFu) pos = group_begin[slot] + atomicAdd(&group_fill[slot], 1u);
        s_pos[tid] = pos; s_row[tid] = row;
        if (pos < out_cap) out_ids[pos] = row_id[row];
        __syncthreads();
        const uint32_t part = tid & 3u;
#pragma unroll
        for (uint32_t q = 0; q < 4; q++) {
            uint32_t li = q * 64u + (tid >> 2);
            uint32_t p = s_pos[li];
            if (p < out_cap) {
                const float4 *src = reinterpret_cast<const float4 *>(row_mat + (size_t)s_row[li] * 16);
                float4 *dst = reinterpret_cast<float4 *>(out_mats + (size_t)p * 16);
                dst[part] = src[part];
            }
        }
        __syncthreads();
    }
}

// Fallback counting pass for k_pack_large when the scan did not count (the host predicted a small visible set, or the probe path ran):
// per (cursor shard, group slot), LDS histogram per workgroup over one shard's segment.
__global__ __launch_bounds__(256) void k_emit_count_sharded(const FrameHeader *hdr, const uint32_t *__restrict__ item_slot, uint32_t nshards, uint32_t seg_cap,
                                                            uint32_t *__restrict__ group_count, uint32_t nslots, const SpecState *spec) {
    extern __shared__ uint32_t s_hist[];
    if (spec->stale) return;
    const uint32_t shard = blockIdx.x % nshards, part = blockIdx.x / nshards, parts = gridDim.x / nshards;
    uint32_t n = (uint32_t)(hdr->cursors[shard * CURSOR_STRIDE] >> 32); if (n > seg_cap) n = seg_cap;
    for (uint32_t i = threadIdx.x; i < nslots; i += blockDim.x) s_hist[i] = 0;
    __syncthreads();
    for (uint32_t t = part * blockDim.x + threadIdx.x; t < n; t += parts * blockDim.x) {
        const uint32_t slot = item_slot[shard * seg_cap + t];
        if (slot < nslots) atomicAdd(&s_hist[slot], 1u);
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < nslots; i += blockDim.x) { const uint32_t v = s_hist[i]; if (v) atomicAdd(&group_count[shard * nslots + i], v); }
}

// The pack of a large visible set (specify_type_ids! callback + MappedBuffer::write_data_serialized + the InstanceRange table:
// prelude/layout_update_macros.rs:15-21, render_components/mapped_buffer.rs:166-189, flows/render_flow.rs:939-992) in ONE launch.
// The instance counts per (cursor shard, group slot) are there already (counted by the scan, or by k_emit_count_sharded), so
//   * every workgroup scans them itself (<= 8 x 512 words from L2): group begins, and where its own shard starts inside every group;
//   * workgroup 0 writes the InstanceRange table + the frame result to the host (publish_to_host) and clears the next frame's headers;
//   * every workgroup moves tiles of PACK_LARGE_TILE instances of ONE shard: rank inside the tile with LDS atomics, ONE global atomic per
//     (tile, non-empty group) on the shard's own fill counter (contention: tiles of one shard only), then id + 64-byte matrix with 4 lanes per
//     instance; the matrix loads are issued before the atomics return, so the tile costs two dependent round trips (list entry -> matrix).
__global__ __launch_bounds__(PACK_LARGE_THREADS) void k_pack_large(PackLargeArgs A) {
    constexpr uint32_t NT = PACK_LARGE_THREADS, NW = NT / 64u, TILE = PACK_LARGE_TILE, PER = TILE / NT, IPP = NT / 4u, PASSES = TILE / IPP, CHUNK = PACK_LARGE_CHUNK;   // IPP: instances per matrix pass (4 lanes each)
    static_assert(PASSES % CHUNK == 0, "matrix passes go in chunks");
    __shared__ uint32_t s_gbase[COUNT_SLOTS_MAX];             // first instance of each group (absolute)
    __shared__ uint32_t s_hist[COUNT_SLOTS_MAX], s_tbase[COUNT_SLOTS_MAX];
    __shared__ uint32_t s_row[TILE], s_pos[TILE], s_id[TILE], s_tpre[COUNT_SLOTS_MAX];
    __shared__ uint32_t s_wsum[NW], s_wcnt[NW], s_whash[NW], s_carry, s_gcarry;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wid = tid >> 6, bid = blockIdx.x, nslots = A.nslots, nsh = A.nshards;
    if (A.spec->stale) {                                    // cancelled frame (SpecState)
        if (bid == 0 && tid == 0) { HostResult r = {}; r.overflow = 2u; *A.hres = r; cancel_slab_header(A.out_count, A.frame); publish_to_host(&A.hres->done_frame, A.frame); }
        return;
    }
    // ---- workgroup b works on cursor shard b & 7, tiles (b >> 3), (b >> 3) + gridDim / 8, ... of that shard's segment: the mapping does not depend on
    // the counts, so the first tile's list entries are requested together with the cursors and the group counts (entries beyond the shard's
    // count are stale but harmless: masked below)
    const uint32_t shard = bid & (CURSOR_SHARDS - 1u), tstride = (gridDim.x >> 3) ? (gridDim.x >> 3) : 1u;   // (never 0: the tile loop below must advance whatever grid it was launched with)
    uint32_t tile = bid >> 3, slot[PER], row[PER];
#pragma unroll
    for (uint32_t q = 0; q < PER; q++) {
        const uint32_t j = tile * TILE + q * NT + tid;
        slot[q] = 0xFFFFFFFFu; row[q] = 0;
        if (shard < nsh && j < A.seg_cap) { const uint32_t ii = shard * A.seg_cap + j; slot[q] = A.item_slot[ii]; row[q] = A.item_row[ii]; }
    }
    uint32_t n_sh = 0, raw_items = 0, raw_sec = 0; bool seg_over = false;
#pragma unroll
    for (uint32_t k = 0; k < CURSOR_SHARDS; k++) {
        const unsigned long long cur = k < nsh ? A.hdr->cursors[k * CURSOR_STRIDE] : 0ull;
        const uint32_t v = (uint32_t)(cur >> 32);
        raw_sec += (uint32_t)cur; raw_items += v; seg_over |= v > A.seg_cap;
        if (k == shard) n_sh = v < A.seg_cap ? v : A.seg_cap;
    }
    if (seg_over) {                                         // a cursor segment overflowed (clustered sections: the shard is the wave index mod 8): nothing is packed, the host redoes the frame with one whole-list segment
        if (bid == 0) {
            if (tid == 0) { HostResult r = {}; r.overflow = RESULT_SEGMENT_OVERFLOW; r.n_entries = raw_sec; r.n_items = raw_items; *A.hres = r; cancel_slab_header(A.out_count, A.frame); publish_to_host(&A.hres->done_frame, A.frame); }
            for (uint32_t i = tid; i < sizeof(FrameHeader) / 4u; i += NT) reinterpret_cast<uint32_t *>(A.hdr_next)[i] = 0u;
            for (uint32_t i = tid; i < sizeof(TickHeader) / 4u; i += NT) reinterpret_cast<uint32_t *>(A.th)[i] = 0u;
            for (uint32_t i = tid; i < A.zero_words; i += NT) { if (A.zero_a) A.zero_a[i] = 0u; if (A.zero_b) A.zero_b[i] = 0u; }
        }
        return;
    }
    if (bid != 0 && tile * TILE >= n_sh) return;
    if (tid == 0) { s_carry = 0; s_gcarry = 0; }
    __syncthreads();
    // ---- scan of the group counts (every workgroup): group begins; workgroup 0 also emits the InstanceRange table
    uint32_t my_hash = 0;
    for (uint32_t base = 0; base < nslots; base += NT) {
        const uint32_t i = base + tid;
        uint32_t v = 0, before = 0;                         // before: instances of this group in the shards in front of this workgroup's
        if (i < nslots) for (uint32_t k = 0; k < nsh; k++) { const uint32_t cnt = A.gcount[k * nslots + i]; v += cnt; if (k < shard) before += cnt; }
        const uint32_t nz = v ? 1u : 0u;
        const uint32_t incl = wave_incl_scan(v), incn = wave_incl_scan(nz);
        if (lane == 63) { s_wsum[wid] = incl; s_wcnt[wid] = incn; }
        __syncthreads();
        uint32_t woff = 0, wcn = 0;
        for (uint32_t w = 0; w < wid; w++) { woff += s_wsum[w]; wcn += s_wcnt[w]; }
        const uint32_t begin = s_carry + woff + incl - v, gidx = s_gcarry + wcn + incn - nz;
        if (i < nslots) {
            s_gbase[i] = begin + before;                    // where this workgroup's shard starts inside the group
            if (v && bid == 0 && gidx < A.range_cap) {
                const uint32_t gc = i >> 3, lod = i & 7u;
                InstanceRange r; r.model_index = A.gc_model[gc] | (lod << 25); r.render_system = A.gc_rs[gc]; r.sortable = A.gc_sort[gc]; r.begin = begin; r.count = v;
                A.ranges[gidx] = r;
                const uint32_t w0 = gidx * (uint32_t)(sizeof(InstanceRange) / 4u);
                my_hash ^= table_word_hash(r.model_index, w0) ^ table_word_hash(r.render_system, w0 + 1u) ^ table_word_hash(r.sortable, w0 + 2u) ^ table_word_hash(r.begin, w0 + 3u) ^ table_word_hash(r.count, w0 + 4u);
            }
        }
        __syncthreads();
        if (tid == NT - 1) { s_carry = begin + v; s_gcarry = gidx + nz; }
        __syncthreads();
    }
    if (bid == 0) {
        // every wave's table stores have left the wave before the barrier in front of the publication (publish_to_host)
        for (int d = 32; d >= 1; d >>= 1) my_hash ^= __shfl_xor(my_hash, d, 64);
        if (lane == 0) s_whash[wid] = my_hash;
        wait_own_stores();
        __syncthreads();
        if (wid == 0) {
            const FrameCounts fc = load_frame_counts(A.hdr);
            if (lane == 0) {
                uint32_t table_hash = 0; for (uint32_t w = 0; w < NW; w++) table_hash ^= s_whash[w];
                HostResult r = {}; r.n_vis_map = fc.n_vis_map; r.n_vis_vec = fc.n_vis_vec; r.n_candidates = fc.n_candidates;
                r.n_groups = s_gcarry < A.range_cap ? s_gcarry : A.range_cap; r.total = s_carry; r.overflow = 0; r.n_entries = raw_sec; r.n_items = raw_items;
                r.table_hash = result_seal(table_hash | 1u, A.frame, r.n_groups, r.total, r.n_vis_map, r.n_vis_vec, r.n_items);
                *A.hres = r;
                write_slab_header(A.out_count, s_carry, A.out_cap, A.frame);
                publish_to_host(&A.hres->done_frame, A.frame);
            }
        }
        for (uint32_t i = tid; i < sizeof(FrameHeader) / 4u; i += NT) reinterpret_cast<uint32_t *>(A.hdr_next)[i] = 0u;      // next frame's cursors / counters
        for (uint32_t i = tid; i < sizeof(TickHeader) / 4u; i += NT) reinterpret_cast<uint32_t *>(A.th)[i] = 0u;
        for (uint32_t i = tid; i < A.zero_words; i += NT) { if (A.zero_a) A.zero_a[i] = 0u; if (A.zero_b) A.zero_b[i] = 0u; }   // the other parity's counts / fills: nobody reads or adds to them during this launch
    }
    // ---- tiles of PER entries per thread: few, fat tiles -- one reservation per (tile, non-empty group) on the shard's own fill counters.  The tile is
    // counting-sorted by group in LDS first, so that neighbouring lanes move neighbouring output rows: the 4-byte ids leave as full lines instead of
    // one 64-byte write per id (the PMC pass of round 2 showed 1.9x the written bytes for the list-order move), the matrix rows as 1 KB runs.
    // Every thread keeps CHUNK 16-byte matrix loads in flight while it moves the tile (4 lanes per instance).
    const uint32_t part = tid & 3u, li = tid >> 2;
    for (bool first = true; tile * TILE < n_sh; tile += tstride, first = false) {   // workgroup-uniform
#pragma unroll
        for (uint32_t q = 0; q < PER; q++) {
            const uint32_t j = tile * TILE + q * NT + tid;
            if (!first) { slot[q] = 0xFFFFFFFFu; row[q] = 0; if (j < n_sh) { const uint32_t ii = shard * A.seg_cap + j; slot[q] = A.item_slot[ii]; row[q] = A.item_row[ii]; } }
            else if (j >= n_sh) { slot[q] = 0xFFFFFFFFu; row[q] = 0; }            // a stale entry beyond the shard's count
            if (slot[q] >= nslots) slot[q] = 0xFFFFFFFFu;
        }
        __syncthreads();                                                        // (the previous tile is done with the LDS arrays)
        for (uint32_t i = tid; i < nslots; i += NT) s_hist[i] = 0;
        for (uint32_t i = tid; i < TILE; i += NT) { s_pos[i] = 0xFFFFFFFFu; s_row[i] = 0u; }   // tile-sorted entries past the tile's live count load row 0 and store nothing
        __syncthreads();
        uint32_t rank[PER], ids[PER];
#pragma unroll
        for (uint32_t q = 0; q < PER; q++) {
            rank[q] = slot[q] != 0xFFFFFFFFu ? atomicAdd(&s_hist[slot[q]], 1u) : 0u;
            ids[q] = slot[q] != 0xFFFFFFFFu ? A.row_id[row[q]] : 0u;              // in flight with everything up to the id stores below
        }
        __syncthreads();
        // exclusive scan of the tile's group histogram (<= 512 slots: two per thread) -> first tile-sorted position of each group
        {
            const uint32_t i0 = tid * 2u, a = i0 < nslots ? s_hist[i0] : 0u, b = i0 + 1u < nslots ? s_hist[i0 + 1u] : 0u;
            const uint32_t incl = wave_incl_scan(a + b);
            if (lane == 63) s_wsum[wid] = incl;
            __syncthreads();
            uint32_t woff = 0; for (uint32_t w = 0; w < wid; w++) woff += s_wsum[w];
            const uint32_t ex = woff + incl - (a + b);
            if (i0 < nslots) s_tpre[i0] = ex;
            if (i0 + 1u < nslots) s_tpre[i0 + 1u] = ex + a;
        }
        __syncthreads();
        uint32_t tp[PER];
#pragma unroll
        for (uint32_t q = 0; q < PER; q++) { tp[q] = slot[q] != 0xFFFFFFFFu ? s_tpre[slot[q]] + rank[q] : 0xFFFFFFFFu; if (tp[q] != 0xFFFFFFFFu) s_row[tp[q]] = row[q]; }
        __syncthreads();
        // requests in flight together from here: the ids (above), the first CHUNK of matrix loads (tile-sorted order), the reservations on the shard's fill counters
        typedef float f32x4 __attribute__((ext_vector_type(4)));                // (a native vector: an array of HIP's float4 class stays in scratch memory here -- 64 B written and read back per instance)
        f32x4 mat[CHUNK];
#pragma unroll
        for (uint32_t ps = 0; ps < CHUNK; ps++)
            mat[ps] = reinterpret_cast<const f32x4 *>(A.row_mat + (size_t)s_row[ps * IPP + li] * 16)[part];
        for (uint32_t i = tid; i < nslots; i += NT) {
            const uint32_t cnt = s_hist[i];
            s_tbase[i] = s_gbase[i] + (cnt ? atomicAdd(&A.gfill[shard * nslots + i], cnt) : 0u);
        }
        __syncthreads();
#pragma unroll
        for (uint32_t q = 0; q < PER; q++) if (tp[q] != 0xFFFFFFFFu) { s_pos[tp[q]] = s_tbase[slot[q]] + rank[q]; s_id[tp[q]] = ids[q]; }
        __syncthreads();
#pragma unroll
        for (uint32_t q = 0; q < PER; q++) { const uint32_t e = q * NT + tid, pos = s_pos[e]; if (pos < A.out_cap) A.out_ids[pos] = s_id[e]; }   // neighbouring lanes, neighbouring words
#pragma unroll                                                                  // (both chunks unrolled: with a rolled loop the compiler keeps mat[] in scratch memory -- 64 B written and read back per instance)
        for (uint32_t c0 = 0; c0 < PASSES; c0 += CHUNK) {
            if (c0) {
#pragma unroll
                for (uint32_t ps = 0; ps < CHUNK; ps++)
                    mat[ps] = reinterpret_cast<const f32x4 *>(A.row_mat + (size_t)s_row[(c0 + ps) * IPP + li] * 16)[part];
            }
#pragma unroll
            for (uint32_t ps = 0; ps < CHUNK; ps++) {
                const uint32_t pp = s_pos[(c0 + ps) * IPP + li];
                if (pp < A.out_cap) reinterpret_cast<f32x4 *>(A.out_mats + (size_t)pp * 16)[part] = mat[ps];
            }
        }
    }
}

// K2 (small visible sets, the common case at the reference's draw distance: ~10^3 instances): the whole pack in ONE
// launch of a few workgroups with no communication between them.  == specify_type_ids! callback +
// MappedBuffer::write_data_serialized + the InstanceRange table (prelude/layout_update_macros.rs:15-21,
// render_components/mapped_buffer.rs:166-189, flows/render_flow.rs:964-983).
// Every workgroup reads the slot of every instance (a few KB, L2 hits) into two LDS histograms -- all instances
// (-> group begins, by a scan) and the instances before its own chunk (-> where its chunk starts inside each group)
// -- then ranks its own chunk with LDS atomics and moves the matrices, 4 lanes per instance, one float4 each.
// The redundant counting costs less than any cross-workgroup hand-over would (a ticket, a fence, or a launch).
// Workgroup 0 also writes the group table and the frame counters to mapped host memory and clears the next frame's header.
__device__ __forceinline__ FrameCounts load_frame_counts(const FrameHeader *hdr) {          // whole wave; result valid in every lane
    const uint32_t lane = lane_id();
    const uint32_t *cnt = hdr->counters + (lane & (COUNTER_SHARDS - 1u)) * 16u;
    uint32_t a = lane < COUNTER_SHARDS ? cnt[0] : 0u, b = lane < COUNTER_SHARDS ? cnt[1] : 0u, c = lane < COUNTER_SHARDS ? cnt[2] : 0u;
    for (int d = 32; d >= 1; d >>= 1) { a += __shfl_xor(a, d, 64); b += __shfl_xor(b, d, 64); c += __shfl_xor(c, d, 64); }
    FrameCounts r; r.n_candidates = a; r.n_vis_map = b; r.n_vis_vec = c; return r;
}
// Workgroup b owns the 64 instances [ (b >> 3) * 64, +64 ) of cursor shard b & 7.  Order of the instances inside a group: by shard,
// then by position in the shard (any order is as good as the reference's hash order).  Nothing a workgroup loads first depends on
// another load: the cursors, the first 256 slots of every shard (the whole list in the common case) and the workgroup's own 64
// (row, slot) pairs are requested together -- entries beyond a shard's count are stale but harmless, they are masked once the
// cursors are known -- and the 64-byte matrices follow one round trip later, in flight while the histograms are built:
// two dependent memory round trips from launch to store.
__device__ __forceinline__ void pack_small_body(const uint32_t bid, const uint32_t nblk, FrameHeader *hdr, FrameHeader *hdr_next, TickHeader *th, const PackArgs &A, const ItemSink &K, uint32_t nrows) {
    extern __shared__ uint32_t s_dyn[];                       // [nslots] all instances -> group begins, [nslots] instances before this chunk -> running fill
    __shared__ uint32_t s_wsum[4], s_wcnt[4], s_whash[4], s_carry, s_gcarry;
    __shared__ uint32_t s_pos[64], s_row[64];
    uint32_t direct_hash = 0;                                 // hash of the table words this thread stores straight to the host (a table too large to stage)
    const uint32_t NT = 256, tid = threadIdx.x, lane = tid & 63u, wid = tid >> 6;
    const bool reporter = bid == 0 && !(A.flags & PACK_NO_PUBLISH);       // (k_scan_cull_sync: the scan's last workgroup has reported the frame already)
    if (A.spec->stale) {                                    // cancelled frame (SpecState): report it, touch nothing
        if (reporter && tid == 0) { HostResult r = {}; r.overflow = 2u; *A.hres = r; cancel_slab_header(A.out_count, A.frame); publish_to_host(&A.hres->done_frame, A.frame); }
        return;
    }
    const uint32_t nslots = A.nslots;
    uint32_t *s_tot = s_dyn, *s_fill = s_dyn + nslots;
    const uint32_t my_shard = bid & (CURSOR_SHARDS - 1u), my_j0 = (bid >> 3) * 64u;
    const uint32_t part = tid & 3u, li = tid >> 2;
    // ---- round trip 1: everything that needs no other load
    unsigned long long cur[CURSOR_SHARDS];
#pragma unroll
    for (uint32_t k = 0; k < CURSOR_SHARDS; k++) cur[k] = hdr->cursors[k * CURSOR_STRIDE];
    uint32_t sl[CURSOR_SHARDS];
#pragma unroll
    for (uint32_t k = 0; k < CURSOR_SHARDS; k++) sl[k] = (k < K.nshards && tid < K.seg_cap) ? K.item_slot[k * K.seg_cap + tid] : 0xFFFFFFFFu;   // (a list of ONE segment -- the redo of a frame whose clustered sections overflowed a segment -- has no entries behind it)
    uint32_t my_slot = 0xFFFFFFFFu, my_row = 0;
    if (tid < 64u && my_shard < K.nshards && my_j0 + tid < K.seg_cap) { const uint32_t ii = my_shard * K.seg_cap + my_j0 + tid; my_slot = K.item_slot[ii]; my_row = K.item_row[ii]; }
    if (my_row >= nrows) my_row = 0;                          // a stale entry of an earlier world
    // workgroup 0 reports the frame: its counter shards travel with this first round trip, and its InstanceRange table is staged in
    // LDS so that one wave writes everything the host reads (table, counts, "frame done") behind a single system-scope fence
    __shared__ InstanceRange s_rng[PACK_STAGED_GROUPS];
    const bool staged = nslots <= PACK_STAGED_GROUPS;         // at most nslots groups
    uint32_t fc_a = 0, fc_b = 0, fc_c = 0;
    if (reporter && wid == 0 && lane < COUNTER_SHARDS) { const uint32_t *cnt = hdr->counters + lane * 16u; fc_a = cnt[0]; fc_b = cnt[1]; fc_c = cnt[2]; }
    for (uint32_t i = tid; i < 2u * nslots && nslots <= LDS_HIST_SLOTS; i += NT) s_dyn[i] = 0;
    if (tid == 0) { s_carry = 0; s_gcarry = 0; }
    if (tid < 64u) s_row[tid] = my_row;
    __syncthreads();
    // ---- round trip 2 (requested now, consumed after the histograms): id + matrix of the own instances
    const uint32_t my_id = tid < 64u ? A.row_id[my_row] : 0u;
    const float4 pre = reinterpret_cast<const float4 *>(A.row_mat + (size_t)s_row[li] * 16)[part];
    // the cursors are here: counts per shard, capacity checks
    uint32_t n[CURSOR_SHARDS], raw_items = 0, raw_sec = 0, T = 0; bool seg_over = false;
#pragma unroll
    for (uint32_t k = 0; k < CURSOR_SHARDS; k++) {
        const uint32_t v = (uint32_t)(cur[k] >> 32);
        raw_sec += (uint32_t)cur[k]; raw_items += v; seg_over |= v > K.seg_cap;
        n[k] = v < K.seg_cap ? v : K.seg_cap; T += n[k];
    }
    const bool overflow = seg_over || T > PACK_SMALL_ITEMS || nslots > LDS_HIST_SLOTS;
    uint32_t my_n = 0;
#pragma unroll
    for (uint32_t k = 0; k < CURSOR_SHARDS; k++) if (k == my_shard) my_n = n[k];
    const bool have = my_j0 < my_n;                           // this workgroup owns instances
    if (bid != 0 && (overflow || !have)) return;
    if (!overflow) {
        // histograms: all instances -> s_tot; the instances ordered before this workgroup's chunk -> s_fill
#pragma unroll
        for (uint32_t k = 0; k < CURSOR_SHARDS; k++) {
            if (tid < n[k] && sl[k] != 0xFFFFFFFFu) { atomicAdd(&s_tot[sl[k]], 1u); if (k < my_shard || (k == my_shard && tid < my_j0)) atomicAdd(&s_fill[sl[k]], 1u); }
            for (uint32_t j = NT + tid; j < n[k]; j += NT) {  // shards longer than the speculative batch
                const uint32_t s2 = K.item_slot[k * K.seg_cap + j];
                if (s2 != 0xFFFFFFFFu) { atomicAdd(&s_tot[s2], 1u); if (k < my_shard || (k == my_shard && j < my_j0)) atomicAdd(&s_fill[s2], 1u); }
            }
        }
        __syncthreads();
        // exclusive scan of the group counts -> group begins (+ the InstanceRange table, workgroup 0)
        for (uint32_t base = 0; base < nslots; base += NT) {
            uint32_t i = base + tid;
            uint32_t v = i < nslots ? s_tot[i] : 0u;
            uint32_t nz = v ? 1u : 0u;
            uint32_t incl = wave_incl_scan(v), incn = wave_incl_scan(nz);
            if (lane == 63) { s_wsum[wid] = incl; s_wcnt[wid] = incn; }
            __syncthreads();
            uint32_t woff = 0, wcn = 0;
            for (uint32_t w = 0; w < wid; w++) { woff += s_wsum[w]; wcn += s_wcnt[w]; }
            uint32_t begin = s_carry + woff + incl - v;
            uint32_t gidx = s_gcarry + wcn + incn - nz;
            if (i < nslots) {
                s_tot[i] = begin;
                if (v && reporter) {
                    uint32_t gc = i >> 3, lod = i & 7u; InstanceRange r; r.model_index = A.gc_model[gc] | (lod << 25); r.render_system = A.gc_rs[gc]; r.sortable = A.gc_sort[gc]; r.begin = begin; r.count = v;
                    if (staged) s_rng[gidx] = r;
                    else {
                        A.ranges[gidx] = r;
                        const uint32_t w0 = gidx * (uint32_t)(sizeof(InstanceRange) / 4u);
                        direct_hash ^= table_word_hash(r.model_index, w0) ^ table_word_hash(r.render_system, w0 + 1u) ^ table_word_hash(r.sortable, w0 + 2u) ^ table_word_hash(r.begin, w0 + 3u) ^ table_word_hash(r.count, w0 + 4u);
                    }
                }
            }
            __syncthreads();
            if (tid == NT - 1) { s_carry = begin + v; s_gcarry = gidx + nz; }
            __syncthreads();
        }
    }
    if (bid == 0) {
        // the host polls done_frame while this kernel runs: every wave's InstanceRange stores (the table that is not staged in LDS) must have
        // left the wave before the barrier in front of lane 0's publication (a workgroup barrier alone does not wait for stores in flight)
        if (!staged) {
            for (int d = 32; d >= 1; d >>= 1) direct_hash ^= __shfl_xor(direct_hash, d, 64);
            if (lane == 0) s_whash[wid] = direct_hash;
            wait_own_stores();
        }
        __syncthreads();
        if (wid == 0 && reporter) {
            for (int d = 32; d >= 1; d >>= 1) { fc_a += __shfl_xor(fc_a, d, 64); fc_b += __shfl_xor(fc_b, d, 64); fc_c += __shfl_xor(fc_c, d, 64); }
            FrameCounts fc; fc.n_candidates = fc_a; fc.n_vis_map = fc_b; fc.n_vis_vec = fc_c;
            uint32_t table_hash = 0;
            if (staged && !overflow) {                          // the table, by this wave alone
                const uint32_t nw = s_gcarry * (uint32_t)(sizeof(InstanceRange) / 4u);
                const uint32_t *src = reinterpret_cast<const uint32_t *>(s_rng); uint32_t *dst = reinterpret_cast<uint32_t *>(A.ranges);
                for (uint32_t w = lane; w < nw; w += 64u) { const uint32_t v = src[w]; dst[w] = v; table_hash ^= table_word_hash(v, w); }
                for (int d = 32; d >= 1; d >>= 1) table_hash ^= __shfl_xor(table_hash, d, 64);
                table_hash |= 1u;
            } else if (!overflow) table_hash = (s_whash[0] ^ s_whash[1] ^ s_whash[2] ^ s_whash[3]) | 1u;     // a table too large to stage: written by all waves above
            wait_own_stores();                                  // the whole wave: every lane's table words have left before lane 0 publishes
            if (lane == 0) {
                HostResult r = {}; r.n_vis_map = fc.n_vis_map; r.n_vis_vec = fc.n_vis_vec; r.n_candidates = fc.n_candidates;   // (never read the mapped host struct: a PCIe round trip)
                r.n_groups = overflow ? 0u : s_gcarry; r.total = overflow ? 0u : s_carry;
                r.overflow = seg_over ? RESULT_SEGMENT_OVERFLOW : (overflow ? 1u : 0u); r.n_entries = raw_sec; r.n_items = raw_items;
                r.table_hash = table_hash ? result_seal(table_hash, A.frame, r.n_groups, r.total, r.n_vis_map, r.n_vis_vec, r.n_items) : 0u;
                *A.hres = r;                                        // mapped pinned host memory
                if (!overflow) write_slab_header(A.out_count, s_carry, A.out_cap, A.frame);
                else if (seg_over) cancel_slab_header(A.out_count, A.frame);      // the host redoes the frame with one whole-list segment (finish_cull) into the same slab
                publish_to_host(&A.hres->done_frame, A.frame);      // the group table and the counters above are complete
            }
        }
        // next frame's cursors / counters (this frame's header stays readable), also when this pack declines (overflow): frames enqueued
        // behind this one must find clean headers; the redo of this frame through the large path reads this frame's own header
        for (uint32_t i = tid; i < sizeof(FrameHeader) / 4u; i += NT) reinterpret_cast<uint32_t *>(hdr_next)[i] = 0u;
        for (uint32_t i = tid; i < sizeof(TickHeader) / 4u; i += NT) reinterpret_cast<uint32_t *>(th)[i] = 0u;       // the tick of this frame starts from zero counters
    }
    if (overflow || !have) return;
    // ---- the own 64 instances: rank inside the group, then id + matrix (4 lanes per instance, one float4 each)
    if (tid < 64u) {
        uint32_t pos = 0xFFFFFFFFu;
        if (my_j0 + tid < my_n && my_slot != 0xFFFFFFFFu) pos = s_tot[my_slot] + atomicAdd(&s_fill[my_slot], 1u);
        s_pos[tid] = pos;
        if (pos < A.out_cap) A.out_ids[pos] = my_id;
    }
    __syncthreads();
    {
        const uint32_t pp = s_pos[li];
        if (pp < A.out_cap) reinterpret_cast<float4 *>(A.out_mats + (size_t)pp * 16)[part] = pre;
    }
    // ---- a shard longer than the launch anticipated (the grid is sized from the previous frame): further chunks of this workgroup,
    // without the speculative loads; the "before this chunk" histogram is rebuilt per chunk
    const uint32_t stride = ((nblk >> 3) ? (nblk >> 3) : 1u) * 64u;     // (never 0: the loop must advance whatever grid it was launched with)
    for (uint32_t j0 = my_j0 + stride; j0 < my_n; j0 += stride) {           // workgroup-uniform
        __syncthreads();
        for (uint32_t i = tid; i < nslots; i += NT) s_fill[i] = 0;
        __syncthreads();
#pragma unroll
        for (uint32_t k = 0; k < CURSOR_SHARDS; k++) {
            const uint32_t lim = k < my_shard ? n[k] : (k == my_shard ? j0 : 0u);
            for (uint32_t j = tid; j < lim; j += NT) { const uint32_t s2 = K.item_slot[k * K.seg_cap + j]; if (s2 != 0xFFFFFFFFu) atomicAdd(&s_fill[s2], 1u); }
        }
        __syncthreads();
        if (tid < 64u) {
            uint32_t pos = 0xFFFFFFFFu, row = 0;
            if (j0 + tid < my_n) {
                const uint32_t ii = my_shard * K.seg_cap + j0 + tid, slot = K.item_slot[ii]; row = K.item_row[ii];
                if (slot != 0xFFFFFFFFu) pos = s_tot[slot] + atomicAdd(&s_fill[slot], 1u);
            }
            s_pos[tid] = pos; s_row[tid] = row;
            if (pos < A.out_cap) A.out_ids[pos] = A.row_id[row];
        }
        __syncthreads();
        const uint32_t pp = s_pos[li];
        if (pp < A.out_cap) reinterpret_cast<float4 *>(A.out_mats + (size_t)pp * 16)[part] = reinterpret_cast<const float4 *>(A.row_mat + (size_t)s_row[li] * 16)[part];
    }
}

__global__ __launch_bounds__(256) void k_pack_small(FrameHeader *hdr, FrameHeader *hdr_next, TickHeader *th, PackArgs A, ItemSink K, uint32_t nrows) {
    pack_small_body(blockIdx.x, gridDim.x, hdr, hdr_next, th, A, K, nrows);
}

// Software-pipelined frame loop of a static world: the launch of frame f + 1 carries the pack of frame f in its first workgroups
// (`npack` of them, passed in the upper bits of the preloaded `nsp` word so that no wave waits for another scalar load), the rest is
// the scan of frame f + 1.  The pack's chain of dependent round trips overlaps the key stream instead of following it in a launch of
// its own; one launch per frame.  The host defers a pack only when nothing can touch the rows between the two culls (no dynamic
// entities), see issue_cull.
template <bool K32>
__global__ __launch_bounds__(CULL_THREADS) void k_scan_cull_fused(const void *__restrict__ keys, uint32_t ncells, uint32_t nsp_npack, uint32_t s0, uint32_t c0, uint32_t s1, uint32_t c1,
                                                                  uint32_t s2, uint32_t c2, uint32_t s3, uint32_t c3, const uint32_t *__restrict__ chunk_level, ScanCullArgs A, FusedPack F) {
    const uint32_t npack = nsp_npack >> 8;
    if (blockIdx.x < npack) { pack_small_body(blockIdx.x, npack, F.hdr, F.hdr_next, F.th, F.A, F.K, F.nrows); return; }
    scan_cull_body<K32>(blockIdx.x - npack, gridDim.x - npack, keys, ncells, nsp_npack & 0xFFu, s0, c0, s1, c1, s2, c2, s3, c3, chunk_level, A);
}
template __global__ void k_scan_cull_fused<false>(const void *, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t, const uint32_t *, ScanCullArgs, FusedPack);
template __global__ void k_scan_cull_fused<true>(const void *, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t, const uint32_t *, ScanCullArgs, FusedPack);

// ---------------------------------------------------------------------------------------------
// One launch between a SYNCHRONOUS call and its answer (small visible sets).  A synchronous frame used to be two dependent launches -- the scan, then
// k_pack_small, whose workgroup 0 publishes the result after its own chain of round trips -- and the host's wait spanned the gap between them.  Here every
// workgroup of the scan signs off when its waves are done (their item_slot stores go through to memory: ItemSink::slot_write_through, and every wave has
// waited for its own stores in front of the workgroup's barrier), on a ticket counter in the line of one of the 64 frame-counter shards; the workgroup that
// completes a shard signs the top-level counter, and the one that completes that -- the last of the launch, nobody waits for anybody -- publishes the frame
// with its first wave: cursors, frame counters and the first 256 group slots of every list segment in ONE round trip (agent-scope loads: everything it reads
// was written by atomics or write-through stores of this same launch), the histogram in LDS, the InstanceRange table and the result block straight into
// mapped host memory (publish_to_host).  k_pack_small follows on the stream with PACK_NO_PUBLISH and moves the instances while the host is already back.
// ---------------------------------------------------------------------------------------------
constexpr uint32_t SYNC_TAIL_ARGS_OFFSET = SCAN_CULL_ARGS_OFFSET + (uint32_t)((sizeof(ScanCullArgs) + 7u) & ~(size_t)7u);
__device__ __forceinline__ uint32_t load_agent(const uint32_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void sync_frame_publish(FrameHeader *hdr, const ItemSink &K, const PackArgs &A) {          // ONE wave
    extern __shared__ uint32_t s_dyn[];                       // [nslots]: instances per group slot
    const uint32_t lane = lane_id();
    if (A.spec->stale) {                                      // cancelled frame (SpecState)
        if (lane == 0) { HostResult r = {}; r.overflow = 2u; *A.hres = r; cancel_slab_header(A.out_count, A.frame); publish_to_host(&A.hres->done_frame, A.frame); }
        return;
    }
    const uint32_t nslots = A.nslots;
    // ---- the one round trip: cursors (lane k), this lane's counter shard, 4 group slots per lane and segment
    unsigned long long mycur = 0;
    if (lane < CURSOR_SHARDS) mycur = __hip_atomic_load(&hdr->cursors[lane * CURSOR_STRIDE], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const uint32_t *cnt = hdr->counters + (lane & (COUNTER_SHARDS - 1u)) * 16u;
    uint32_t fc_a = load_agent(cnt + 0), fc_b = load_agent(cnt + 1), fc_c = load_agent(cnt + 2);
    uint32_t sl[CURSOR_SHARDS][4];
#pragma unroll
    for (uint32_t k = 0; k < CURSOR_SHARDS; k++)
#pragma unroll
        for (uint32_t q = 0; q < 4; q++) { const uint32_t j = q * 64u + lane; sl[k][q] = (k < K.nshards && j < K.seg_cap) ? load_agent(&K.item_slot[k * K.seg_cap + j]) : 0xFFFFFFFFu; }
    for (uint32_t i = lane; i < nslots && nslots <= SYNC_TAIL_SLOTS; i += 64u) s_dyn[i] = 0u;
    uint32_t n[CURSOR_SHARDS], raw_items = 0, raw_sec = 0, T = 0; bool seg_over = false;
    const uint32_t cur_lo = (uint32_t)mycur, cur_hi = (uint32_t)(mycur >> 32);
#pragma unroll
    for (uint32_t k = 0; k < CURSOR_SHARDS; k++) {
        const uint32_t v = __shfl(cur_hi, (int)k, 64);
        raw_sec += __shfl(cur_lo, (int)k, 64); raw_items += v; seg_over |= v > K.seg_cap;
        n[k] = v < K.seg_cap ? v : K.seg_cap; T += n[k];
    }
    const bool overflow = seg_over || T > PACK_SMALL_ITEMS || nslots > SYNC_TAIL_SLOTS;
    uint32_t table_hash = 0, carry = 0, gcarry = 0;
    if (!overflow) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (uint32_t k = 0; k < CURSOR_SHARDS; k++) {
#pragma unroll
            for (uint32_t q = 0; q < 4; q++) if (q * 64u + lane < n[k] && sl[k][q] < nslots) atomicAdd(&s_dyn[sl[k][q]], 1u);
            for (uint32_t j = 256u + lane; j < n[k]; j += 64u) { const uint32_t s2 = load_agent(&K.item_slot[k * K.seg_cap + j]); if (s2 < nslots) atomicAdd(&s_dyn[s2], 1u); }   // segments longer than the speculative batch
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // exclusive scan of the group counts -> the InstanceRange table, every entry stored by the lane that holds it (this wave alone writes what the host reads)
        for (uint32_t base = 0; base < nslots; base += 64u) {
            const uint32_t i = base + lane, v = i < nslots ? s_dyn[i] : 0u, nz = v ? 1u : 0u;
            const uint32_t incl = wave_incl_scan(v), incn = wave_incl_scan(nz);
            const uint32_t begin = carry + incl - v, gidx = gcarry + incn - nz;
            if (v) {
                const uint32_t gc = i >> 3, lod = i & 7u; InstanceRange r; r.model_index = A.gc_model[gc] | (lod << 25); r.render_system = A.gc_rs[gc]; r.sortable = A.gc_sort[gc]; r.begin = begin; r.count = v;
                A.ranges[gidx] = r;
                const uint32_t w0 = gidx * (uint32_t)(sizeof(InstanceRange) / 4u);
                table_hash ^= table_word_hash(r.model_index, w0) ^ table_word_hash(r.render_system, w0 + 1u) ^ table_word_hash(r.sortable, w0 + 2u) ^ table_word_hash(r.begin, w0 + 3u) ^ table_word_hash(r.count, w0 + 4u);
            }
            carry += __shfl(incl, 63, 64); gcarry += __shfl(incn, 63, 64);
        }
        for (int d = 32; d >= 1; d >>= 1) table_hash ^= __shfl_xor(table_hash, d, 64);
        table_hash |= 1u;
    }
    for (int d = 32; d >= 1; d >>= 1) { fc_a += __shfl_xor(fc_a, d, 64); fc_b += __shfl_xor(fc_b, d, 64); fc_c += __shfl_xor(fc_c, d, 64); }
    wait_own_stores();                                        // the whole wave: every lane's table words have left before lane 0 publishes
    if (lane == 0) {
        HostResult r = {}; r.n_vis_map = fc_b; r.n_vis_vec = fc_c; r.n_candidates = fc_a;
        r.n_groups = overflow ? 0u : gcarry; r.total = overflow ? 0u : carry;
        r.overflow = seg_over ? RESULT_SEGMENT_OVERFLOW : (overflow ? 1u : 0u); r.n_entries = raw_sec; r.n_items = raw_items;
        r.table_hash = table_hash ? result_seal(table_hash, A.frame, r.n_groups, r.total, r.n_vis_map, r.n_vis_vec, r.n_items) : 0u;
        *A.hres = r;
        if (!overflow) write_slab_header(A.out_count, carry, A.out_cap, A.frame);
        else if (seg_over) cancel_slab_header(A.out_count, A.frame);
        publish_to_host(&A.hres->done_frame, A.frame);
    }
}
template <bool K32>
__global__ __launch_bounds__(CULL_THREADS) void k_scan_cull_sync(const void *__restrict__ keys, uint32_t ncells, uint32_t nsp, uint32_t s0, uint32_t c0, uint32_t s1, uint32_t c1,
                                                                 uint32_t s2, uint32_t c2, uint32_t s3, uint32_t c3, const uint32_t *__restrict__ chunk_level, ScanCullArgs A, PackArgs T) {
    scan_cull_body<K32>(blockIdx.x, gridDim.x, keys, ncells, nsp, s0, c0, s1, c1, s2, c2, s3, c3, chunk_level, A);
    wait_own_stores();                                        // this wave's list entries (write-through) and atomics have been performed ...
    __syncthreads();                                          // ... before the workgroup signs off
    if (threadIdx.x >= 64u) return;
    typedef __attribute__((address_space(4))) const char *kernarg_ptr;
    kernarg_ptr ka = (kernarg_ptr)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(ka));                              // (as in the body: none of these scalar loads in front of the key loads)
    const ScanCullArgs &R = *(const ScanCullArgs *)(ka + SCAN_CULL_ARGS_OFFSET);
    FrameHeader *hdr = R.hdr;
    uint32_t last = 0;
    if (lane_id() == 0) {
        const uint32_t nblk = gridDim.x, sh = blockIdx.x & (COUNTER_SHARDS - 1u), expect = (nblk - sh + COUNTER_SHARDS - 1u) / COUNTER_SHARDS;
        if (atomicAdd(&hdr->counters[sh * 16u + 3u], 1u) + 1u == expect) {
            const uint32_t ntop = nblk < COUNTER_SHARDS ? nblk : COUNTER_SHARDS;
            if (atomicAdd(&hdr->counters[4], 1u) + 1u == ntop) last = 1u;           // (issued after the atomic above has returned)
        }
    }
    if (!__shfl(last, 0, 64)) return;
    const PackArgs &P2 = *(const PackArgs *)(ka + SYNC_TAIL_ARGS_OFFSET);
    sync_frame_publish(hdr, R.K, P2);
}
template __global__ void k_scan_cull_sync<false>(const void *, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t, const uint32_t *, ScanCullArgs, PackArgs);
template __global__ void k_scan_cull_sync<true>(const void *, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t, const uint32_t *, ScanCullArgs, PackArgs);

// ---------------------------------------------------------------------------------------------
// K3: the ECS tick.  LogicFlow::apply_kinematics (flows/logic_flow.rs:366-448) + the component math of
// exports/movement_components.rs:210-299 + update_aabb_after_kinematic_change
// (helper_things/entity_change_helpers.rs:217-262), in place on the SoA columns.
// One lane per dynamic entity (entities carrying Velocity or VelocityRotation).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void normalize3(const float v[3], float o[3]) {
    float n = norm3(v[0], v[1], v[2]); o[0] = v[0] / n; o[1] = v[1] / n; o[2] = v[2] / n;
}

// update_aabb_after_kinematic_change (entity_change_helpers.rs:217-262) + the decision of update_entity_in_tree (:325-351) for one entity whose
// Position / Rotation / Scale changed: new TransformationMatrix and StaticAABB; returns 0 when the entity stays in its spatial-hash section
// (entity_exists_in_section, bounding_box_tree_v2.rs:765-782), 1 when it changes section (re-bucket list), 2 when it leaves the world without
// OutOfBoundsLogic (ecs.remove_entity, :347; the tree keeps the stale entry -- the caller marks the row dead).  No atomics here.
// what place_core reads from memory, so that a caller can request it early together with its own loads
struct PlaceInputs { Aabb orig; float scl[3]; float c3w; uint64_t key; };
// (nothing here depends on another load: the key column holds the key of the row's own section -- == cell_key[row_cell[r]] for a row in a unique
// section, meaningless otherwise: place_core looks at it only then)
__device__ __forceinline__ void place_prefetch(PlaceInputs &in, bool on, uint32_t r, const RowArrays &R) {
    in.orig = Aabb{ 0.f, 0.f, 0.f, 0.f, 0.f, 0.f }; in.scl[0] = in.scl[1] = in.scl[2] = 1.f; in.c3w = 1.f; in.key = 0;
    if (!on) return;
    in.orig = R.orig[r];
    in.scl[0] = R.scale[r * 3 + 0]; in.scl[1] = R.scale[r * 3 + 1]; in.scl[2] = R.scale[r * 3 + 2];
    in.c3w = R.mat[(size_t)r * 16 + 15];
    in.key = R.key[r];
}
__device__ __forceinline__ uint32_t place_core(uint32_t r, uint32_t fl, uint32_t rc, const float pos[3], const float rot[4], bool translation_only,
                                                RowArrays R, const uint64_t *__restrict__ cell_key, const int32_t *__restrict__ sh_cells, uint32_t outline, uint32_t atomic,
                                                const PlaceInputs *pre = nullptr) {
    Aabb orig = pre ? pre->orig : R.orig[r], a;
    float4 *mo = reinterpret_cast<float4 *>(R.mat + (size_t)r * 16);
    if (translation_only) {
        // translation-only fast path: column 3 xyz overwritten, OriginalAABB translated (rotation/scale ignored, :221-240)
        float4 c3; c3.w = pre ? pre->c3w : mo[3].w; c3.x = pos[0]; c3.y = pos[1]; c3.z = pos[2]; mo[3] = c3;
        a.xmin = orig.xmin + pos[0]; a.xmax = orig.xmax + pos[0]; a.ymin = orig.ymin + pos[1]; a.ymax = orig.ymax + pos[1]; a.zmin = orig.zmin + pos[2]; a.zmax = orig.zmax + pos[2];
    } else {
        float scl[3] = { pre ? pre->scl[0] : R.scale[r * 3 + 0], pre ? pre->scl[1] : R.scale[r * 3 + 1], pre ? pre->scl[2] : R.scale[r * 3 + 2] };   // Scale::default() when absent (set at upload)
        float m[16];
        trs_matrix(pos, true, rot, rot[3], true, scl, m);                               // all three factors, defaults when absent (:245-250)
        mo[0] = make_float4(m[0], m[1], m[2], m[3]); mo[1] = make_float4(m[4], m[5], m[6], m[7]);
        mo[2] = make_float4(m[8], m[9], m[10], m[11]); mo[3] = make_float4(m[12], m[13], m[14], m[15]);
        a = apply_transformation(orig, m);
    }
    R.aabb[r] = a;
    // update_entity_in_tree -> add_entity (:325-351): same section => nothing else happens
    Aabb bv = a;
    bool oob = normalize_aabb(&bv, (float)outline);
    if (oob && !(fl & F_OOB_LOGIC)) { R.gclass[r] = 0xFFFFFFFFu; return 2u; }
    // Short cut for the common case, an entity of a level-0 section that stays inside it (power-of-two atomic length): when both corners of
    // the clamped box truncate to the section's own indices, add_entity's decision is that section again -- the extents are below one section
    // length, so find_aabb_level_from_length_and_origin stops at level 0 with one section per axis, and find_unique_world_section_id
    // divides the same minimum corner (bounding_box_tree_v2.rs:451-551).  Everything else takes the full decision below.
    if (pre && rc != ROW_CELL_NONE && !(rc & ROW_CELL_SHARED) && (atomic & (atomic - 1u)) == 0u && key_level(pre->key) == 0u) {
        const float inv = 1.0f / (float)atomic;
        const float cx = (float)key_x(pre->key), cy = (float)key_y(pre->key), cz = (float)key_z(pre->key);
        if (truncf(bv.xmin * inv) == cx && truncf(bv.xmax * inv) == cx && truncf(bv.ymin * inv) == cy && truncf(bv.ymax * inv) == cy &&
            truncf(bv.zmin * inv) == cz && truncf(bv.zmax * inv) == cz && bv.xmin <= bv.xmax && bv.ymin <= bv.ymax && bv.zmin <= bv.zmax) return 0u;
    }
    uint64_t keys[8];
    int nk = assign_sections(bv, atomic, keys);
    bool same;
    if (rc == ROW_CELL_NONE) same = false;
    else if (!(rc & ROW_CELL_SHARED)) same = (nk == 1) && keys[0] == (pre ? pre->key : cell_key[rc]);
    else {
        uint32_t s = rc & ~ROW_CELL_SHARED;
        same = nk > 1;
        for (int k = 0; k < 8 && same; k++) {
            int32_t c = sh_cells[s * 8 + k];
            if (k < nk) same = c >= 0 && cell_key[c] == keys[k]; else same = c < 0;
        }
    }
    return same ? 0u : 1u;
}
// the same for one entity of a change batch (k_apply_rows): per-lane list reservations (a batch is small)
__device__ __forceinline__ void place_changed_entity(uint32_t r, uint32_t fl, uint32_t nfl, uint32_t rc, const float pos[3], const float rot[4], bool translation_only,
                                                     RowArrays R, const uint64_t *__restrict__ cell_key, const int32_t *__restrict__ sh_cells,
                                                     uint32_t outline, uint32_t atomic, TickHeader *th, uint32_t *__restrict__ mover_rows, uint32_t *__restrict__ oob_rows, uint32_t list_cap) {
    const uint32_t status = place_core(r, fl, rc, pos, rot, translation_only, R, cell_key, sh_cells, outline, atomic);
    if (status == 2u) {
        uint32_t slot = atomicAdd(&th->n_oob, 1u);
        if (slot < list_cap) oob_rows[slot] = r;
        R.flags[r] = nfl | F_DEAD;
    } else if (status == 1u) {
        uint32_t slot = atomicAdd(&th->n_rebucket, 1u);
        if (slot < list_cap) mover_rows[slot] = r | (translation_only ? 0x80000000u : 0u);   // bit 31: translation-only mover
    }
}

// K3.  The rows of the dynamic entities (Velocity or VelocityRotation) are the FIRST ndyn rows of every per-entity column (the upload
// sorts them to the front), so lane j works on row j: flags, section slot, Position, Rotation, Scale, OriginalAABB and the velocities are
// read -- and TransformationMatrix, StaticAABB, Rotation written -- as contiguous streams (80 B read + 104 B written per
// ticking entity); the only gathers are the section stamp (the visibility gate) and, for ticking entities, the key of their section.
// Counters: one atomic per wave (ballot + mbcnt), as in the cull kernel.
__device__ __forceinline__ void tick_body(uint32_t ndyn, float *__restrict__ dyn_vel, const float *__restrict__ dyn_acc,
                                              float *__restrict__ dyn_rotvel, const float *__restrict__ dyn_rotacc,
                                              RowArrays R, const uint32_t *__restrict__ row_cell,
                                              const uint64_t *__restrict__ cell_key, const uint32_t *__restrict__ cell_stamp, const uint8_t *__restrict__ cell_flags,
                                              const int32_t *__restrict__ sh_cells, const Aabb *__restrict__ sh_aabb,
                                              const FrameParams *__restrict__ Pp, float dt, uint32_t tick_all, uint32_t outline, uint32_t atomic,
                                              TickHeader *th, uint32_t *__restrict__ mover_rows, uint32_t *__restrict__ oob_rows, uint32_t list_cap,
                                              SpecState *spec, SpecState *h_spec, uint32_t tick_frame, uint32_t ndyn0, const uint32_t *__restrict__ dyn_row) {
    // An EARLIER tick left the tree stale: this frame is replayed by the host.  The flag a workgroup of THIS tick raises when it finds a
    // mover must not stop the workgroups of the same tick that start later (they would skip their entities for good): the frame travels
    // with the flag in one 64-bit word.
    // Dynamic entity j < ndyn0 lives in row j (the upload's leading block); entities that became dynamic later -- added with a velocity, or given one
    // by a change request -- sit in any row, listed in dyn_row (one more coalesced load for the waves of the tail only).
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    const bool in = j < ndyn;
    const uint32_t r = (j < ndyn0 || !in) ? j : dyn_row[j];
    const FrameParams &P = *Pp;
    // ---- round trip 1: flag word and section slot of the row (coalesced), the frame number, and -- in the same round trip, not in front of it -- the speculation word
    const uint32_t fl = in ? R.flags[r] : F_DEAD, rc = in ? row_cell[r] : ROW_CELL_NONE;
    const uint32_t cull_frame = P.frame;
    // RE_TICK_ALL_DYNAMIC: every live entity ticks, so every component is requested NOW, together with the flag word (one round trip from launch to
    // arithmetic instead of two); a lane that turns out not to use a component -- or not to tick -- ignores what arrives.
    float pos[3] = { 0.f, 0.f, 0.f }, rot[4] = { 1.f, 0.f, 0.f, 0.f }, v[3] = { 0.f, 0.f, 0.f }, a[3] = { 0.f, 0.f, 0.f };
    float4 wq = make_float4(1.f, 0.f, 0.f, 0.f), aq = make_float4(1.f, 0.f, 0.f, 0.f);
    PlaceInputs pin;                                                              // what place_core needs from memory, requested in the same round trip
    auto load_components = [&](bool on, bool by_flags) {
        if (on) {
            pos[0] = R.pos[r * 3 + 0]; pos[1] = R.pos[r * 3 + 1]; pos[2] = R.pos[r * 3 + 2];
            const float4 q = reinterpret_cast<const float4 *>(R.rot)[r]; rot[0] = q.x; rot[1] = q.y; rot[2] = q.z; rot[3] = q.w;
            if (!by_flags || (fl & F_HAS_VEL)) { v[0] = dyn_vel[j * 3 + 0]; v[1] = dyn_vel[j * 3 + 1]; v[2] = dyn_vel[j * 3 + 2]; }
            if (!by_flags || (fl & F_HAS_ACC)) { a[0] = dyn_acc[j * 3 + 0]; a[1] = dyn_acc[j * 3 + 1]; a[2] = dyn_acc[j * 3 + 2]; }
            if (!by_flags || (fl & F_HAS_ROTVEL)) wq = reinterpret_cast<const float4 *>(dyn_rotvel)[j];
            if (!by_flags || (fl & F_HAS_ROTACC)) aq = reinterpret_cast<const float4 *>(dyn_rotacc)[j];
        }
        place_prefetch(pin, on, r, R);
    };
    if (tick_all) load_components(in, false);
    {
        const unsigned long long w = *reinterpret_cast<const volatile unsigned long long *>(spec);
        if ((uint32_t)w != 0u && (uint32_t)(w >> 32) != tick_frame) return;     // tick_frame: the frame this tick was issued for (a cancelled frame never wrote its parameters, so Pp->frame would be the stale one's)
    }
    // ---- the visibility gate (logic_flow.rs:216-223, 308-358): one gathered word per entity, none with RE_TICK_ALL_DYNAMIC
    const bool unique_cell = rc != ROW_CELL_NONE && !(rc & ROW_CELL_SHARED);
    const uint32_t stamp = (unique_cell && !tick_all) ? cell_stamp[rc] : 0u;
    // reset_has_changed_component (logic_flow.rs:776-801)
    uint32_t nfl = fl & ~(F_HAS_MOVED | F_HAS_ROTATED);
    bool run = false;
    if (fl & F_DEAD) run = false;
    else if (tick_all) run = rc != ROW_CELL_NONE;
    else if (rc == ROW_CELL_NONE) run = false;
    else if (!(rc & ROW_CELL_SHARED)) {
        const bool vis = (stamp >> 2) == cull_frame;
        // visible loop: local (non-static) entities of active visible sections; always-execute entities
        // only when their section is NOT in visible_sections_map (find_always_execute_entities :803-836)
        run = (!(fl & F_STATIC) && vis) || ((fl & F_ALWAYS_EXEC) && !vis);
    } else {
        const uint32_t s = rc & ~ROW_CELL_SHARED;
        bool anyvis = false, act = false;
        for (int k = 0; k < 8; k++) {
            const int32_t c = sh_cells[s * 8 + k];
            if (c >= 0 && (cell_stamp[c] >> 2) == cull_frame) { anyvis = true; if (!(cell_flags[c] & CF_STATIC_SECTION)) act = true; }
        }
        bool inview = false;
        if (act && !(fl & F_STATIC)) {
            const Aabb sa = sh_aabb[s];
            inview = logic_aabb_in_view(P.lookahead, P.cam[0], P.cam[1], P.cam[2], sa) || frustum_aabb_visible(P.planes, sa);   // logic_flow.rs:338-339
        }
        run = (!(fl & F_STATIC) && act && inview) || ((fl & F_ALWAYS_EXEC) && !anyvis);
    }
    if (!__ballot(run)) {                                                         // nobody in this wave ticks (the common wave of a visibility-gated tick): only the markers
        if (in && nfl != fl) R.flags[r] = nfl;
        return;
    }
    // ---- round trip 2 (the visibility-gated tick; RE_TICK_ALL_DYNAMIC asked for all of it in round trip 1): every component a ticking lane may need,
    // requested together (all contiguous by row) -- no load below depends on another
    if (!tick_all) load_components(run, true);
    // ---- apply_kinematics (logic_flow.rs:366-448)
    bool pos_set = false, rot_set = false;
    if (run) {
        if (fl & F_HAS_VEL) {
            if ((fl & F_HAS_ACC) && norm3(a[0], a[1], a[2]) != 0.0f) {            // :384  velocity += acceleration * dt
                dyn_vel[j * 3 + 0] = v[0] + a[0] * dt; dyn_vel[j * 3 + 1] = v[1] + a[1] * dt; dyn_vel[j * 3 + 2] = v[2] + a[2] * dt;
            }
            if (norm3(v[0], v[1], v[2]) != 0.0f) {                                // :394  position += OLD velocity * dt (change requests are deferred)
                pos[0] = pos[0] + v[0] * dt; pos[1] = pos[1] + v[1] * dt; pos[2] = pos[2] + v[2] * dt;
                pos_set = true;
            }
        }
        if (fl & F_HAS_ROTVEL) {
            const float w[4] = { wq.x, wq.y, wq.z, wq.w };
            if ((fl & F_HAS_ROTACC) && aq.w != 0.0f) {                            // :418
                float sc[3] = { aq.x * dt, aq.y * dt, aq.z * dt }, nrm[3], sum[3], out[3];
                normalize3(sc, nrm);
                sum[0] = w[0] + nrm[0]; sum[1] = w[1] + nrm[1]; sum[2] = w[2] + nrm[2];
                normalize3(sum, out);
                reinterpret_cast<float4 *>(dyn_rotvel)[j] = make_float4(out[0], out[1], out[2], w[3] + aq.w * dt);
            }
            if (w[3] != 0.0f) {                                                   // :429  rotation += OLD rotation velocity * dt
                float sc[3] = { w[0] * dt, w[1] * dt, w[2] * dt }, nrm[3], sum[3], out[3];
                normalize3(sc, nrm);
                sum[0] = rot[0] + nrm[0]; sum[1] = rot[1] + nrm[1]; sum[2] = rot[2] + nrm[2];
                normalize3(sum, out);
                rot[0] = out[0]; rot[1] = out[1]; rot[2] = out[2]; rot[3] = rot[3] + w[3] * dt;
                rot_set = true;
            }
        }
    }
    if (pos_set) nfl |= F_HAS_MOVED;
    if (rot_set) nfl |= F_HAS_ROTATED;
    const bool changed = pos_set || rot_set;
    uint32_t status = 0;                                                          // 1: the entity changes section (re-bucket list), 2: it leaves the world
    if (changed) {
        if (pos_set) { R.pos[r * 3 + 0] = pos[0]; R.pos[r * 3 + 1] = pos[1]; R.pos[r * 3 + 2] = pos[2]; }
        if (rot_set) reinterpret_cast<float4 *>(R.rot)[r] = make_float4(rot[0], rot[1], rot[2], rot[3]);
        status = place_core(r, fl, rc, pos, rot, pos_set && !rot_set, R, cell_key, sh_cells, outline, atomic, &pin);
        if (status == 2u) nfl |= F_DEAD;
    }
    if (in && nfl != fl) R.flags[r] = nfl;
    // ---- counters and lists: one atomic per wave and counter
    const uint64_t mc = __ballot(changed), mr = __ballot(status == 1u), mo = __ballot(status == 2u);
    if (mc) {
        const uint32_t lane = lane_id();
        uint32_t base_r = 0, base_o = 0;
        if (lane == 0) {
            atomicAdd(&th->shard[((blockIdx.x * 4u + (threadIdx.x >> 6)) & (TICK_TICKET_SHARDS - 1u)) * TICK_SHARD_STRIDE], (uint32_t)__popcll(mc));   // n_changed, sharded: ONE address takes ~88 atomics per us (measured: 1,575 waves = 18 us of a 25 us launch)
            if (mr) base_r = atomicAdd(&th->n_rebucket, (uint32_t)__popcll(mr));
            if (mo) base_o = atomicAdd(&th->n_oob, (uint32_t)__popcll(mo));
            if (mr | mo) {                                                        // {stale = 1, stale_frame = frame} in one store: the host must patch the tree / retire the rows before any later frame runs
                const unsigned long long w = 1ull | ((unsigned long long)tick_frame << 32);
                *reinterpret_cast<volatile unsigned long long *>(spec) = w; post_to_host64(reinterpret_cast<unsigned long long *>(h_spec), w);
            }
        }
        if (mr) { base_r = __shfl(base_r, 0, 64); if (status == 1u) { const uint32_t slot = base_r + mbcnt(mr); if (slot < list_cap) mover_rows[slot] = r | ((pos_set && !rot_set) ? 0x80000000u : 0u); } }   // bit 31: translation-only mover
        if (mo) { base_o = __shfl(base_o, 0, 64); if (status == 2u) { const uint32_t slot = base_o + mbcnt(mo); if (slot < list_cap) oob_rows[slot] = r; } }
    }
}

// The tick's counters for the host, published by the LAST wave of the tick itself (synchronous ticks: seq != 0), so that no second launch sits between
// the tick and the host that polls for it.  Every wave signs off when it is done -- after a fence that puts its counter atomics in front of the
// signature -- on one of 32 counters (each in the 128-byte line of a changed-count shard: atomics serialise per line), the wave that completes a
// counter signs the top-level one, and the wave that completes that one reads the counters (agent-scope loads: the atomics live in L2), copies them
// into mapped host memory and publishes the tick's sequence number there (publish_to_host).  Nobody waits for anybody: no wave spins.
__device__ __forceinline__ void tick_sign_off(TickHeader *th, TickHeader *h_th, uint32_t seq) {
    if (lane_id() != 0) return;
    // Only ATOMICS are handed from wave to wave here (agent scope: performed at the device's coherence point, not in the XCD's L2), so the order
    // that matters is "this wave's counter atomics have been performed before its signature is": s_waitcnt vmcnt(0).  A release FENCE in this
    // place writes the XCD's L2 back once per wave -- measured: 57 us instead of 11 for 1,575 waves, 674 instead of 53 for 15,747.  (The mover /
    // out-of-bounds lists are plain stores; their reader is the host behind the end of the kernel.)
    wait_own_stores();
    const uint32_t wpb = blockDim.x >> 6, W = gridDim.x * wpb, wv = blockIdx.x * wpb + (threadIdx.x >> 6), sh = wv & (TICK_TICKET_SHARDS - 1u);
    const uint32_t expect = (W - sh + TICK_TICKET_SHARDS - 1u) / TICK_TICKET_SHARDS;          // waves whose index is sh modulo 32
    if (atomicAdd(&th->shard[sh * TICK_SHARD_STRIDE + 1u], 1u) + 1u != expect) return;
    const uint32_t ntop = W < TICK_TICKET_SHARDS ? W : TICK_TICKET_SHARDS;
    if (atomicAdd(&th->pad[1], 1u) + 1u != ntop) return;                    // (issued after the atomic above has returned)
    uint32_t a = __hip_atomic_load(&th->n_changed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);       // (n_changed itself: the per-lane adds of a change batch, k_apply_rows)
    for (uint32_t k2 = 0; k2 < TICK_TICKET_SHARDS; k2++) a += __hip_atomic_load(&th->shard[k2 * TICK_SHARD_STRIDE], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const uint32_t b = __hip_atomic_load(&th->n_rebucket, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), c2 = __hip_atomic_load(&th->n_oob, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    h_th->n_changed = a; h_th->n_rebucket = b; h_th->n_oob = c2; h_th->pad[0] = table_word_hash(a, 1u) ^ table_word_hash(b, 2u) ^ table_word_hash(c2, 3u) ^ table_word_hash(seq, 4u);   // seal: the reader checks it
    publish_to_host(&h_th->ticket, seq);
}
__global__ __launch_bounds__(256) void k_tick(uint32_t ndyn, float *__restrict__ dyn_vel, const float *__restrict__ dyn_acc,
                                              float *__restrict__ dyn_rotvel, const float *__restrict__ dyn_rotacc,
                                              RowArrays R, const uint32_t *__restrict__ row_cell,
                                              const uint64_t *__restrict__ cell_key, const uint32_t *__restrict__ cell_stamp, const uint8_t *__restrict__ cell_flags,
                                              const int32_t *__restrict__ sh_cells, const Aabb *__restrict__ sh_aabb,
                                              const FrameParams *__restrict__ Pp, float dt, uint32_t tick_all, uint32_t outline, uint32_t atomic,
                                              TickHeader *th, uint32_t *__restrict__ mover_rows, uint32_t *__restrict__ oob_rows, uint32_t list_cap,
                                              SpecState *spec, SpecState *h_spec, uint32_t tick_frame, uint32_t ndyn0, const uint32_t *__restrict__ dyn_row,
                                              TickHeader *h_th, uint32_t publish_seq) {
    tick_body(ndyn, dyn_vel, dyn_acc, dyn_rotvel, dyn_rotacc, R, row_cell, cell_key, cell_stamp, cell_flags, sh_cells, sh_aabb, Pp, dt, tick_all, outline, atomic, th, mover_rows, oob_rows,
              list_cap, spec, h_spec, tick_frame, ndyn0, dyn_row);
    if (publish_seq) tick_sign_off(th, h_th, publish_seq);                  // every wave, whichever way it left the body
}

// section decision (add_entity with add_if_out_bounds = true: the box is clamped) for a list of rows, from their current StaticAABB;
// also clears RE_F_STATIC: update_entity_in_tree re-adds movers with is_static = false (entity_change_helpers.rs:330)
__global__ __launch_bounds__(256) void k_assign_rows(uint32_t m, const uint32_t *__restrict__ rows, RowArrays R, uint32_t outline, uint32_t atomic,
                                                     uint8_t *__restrict__ out_nk, uint64_t *__restrict__ out_keys) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    uint32_t r = rows[i] & 0x7FFFFFFFu;
    Aabb bv = R.aabb[r];
    normalize_aabb(&bv, (float)outline);
    uint64_t keys[8];
    int nk = assign_sections(bv, atomic, keys);
    if (nk < 0) nk = 0;
    out_nk[i] = (uint8_t)nk;
    for (int k = 0; k < 8; k++) out_keys[(size_t)i * 8 + k] = k < nk ? keys[k] : 0ull;
    R.flags[r] &= ~F_STATIC;
}
// end_of_changes restricted to the changed sections of an incremental update: unchanged sections keep their (possibly stale) AABB
// Component writes of user change requests (EntityChangeRequest::apply_changes -> ECS::write_component, objects/ecs.rs:457-474),
// resolved to "last write wins" per (entity, component) on the host.  comp: RE_C_* (0..6) or WRITE_FLAGS (and-mask, or-mask[, kill]).
__device__ __forceinline__ void write_component(const WriteOp &w, const RowArrays &R, float *__restrict__ dyn_vel, float *__restrict__ dyn_acc, float *__restrict__ dyn_rotvel, float *__restrict__ dyn_rotacc) {
    const float *v = reinterpret_cast<const float *>(w.v);
    switch (w.comp) {
        case 0: for (int k = 0; k < 3; k++) R.pos[(size_t)w.index * 3 + k] = v[k]; break;
        case 1: for (int k = 0; k < 4; k++) R.rot[(size_t)w.index * 4 + k] = v[k]; break;
        case 2: for (int k = 0; k < 3; k++) R.scale[(size_t)w.index * 3 + k] = v[k]; break;
        case 3: for (int k = 0; k < 3; k++) dyn_vel[(size_t)w.index * 3 + k] = v[k]; break;
        case 4: for (int k = 0; k < 3; k++) dyn_acc[(size_t)w.index * 3 + k] = v[k]; break;
        case 5: for (int k = 0; k < 4; k++) dyn_rotvel[(size_t)w.index * 4 + k] = v[k]; break;
        case 6: for (int k = 0; k < 4; k++) dyn_rotacc[(size_t)w.index * 4 + k] = v[k]; break;
        case WRITE_GCLASS: R.gclass[w.index] = w.v[0]; break;
        case WRITE_FLAGS: R.flags[w.index] = (R.flags[w.index] & w.v[0]) | w.v[1]; if (w.v[2]) R.gclass[w.index] = 0xFFFFFFFFu; break;
        default: break;
    }
}
__global__ __launch_bounds__(256) void k_write_components(uint32_t m, const WriteOp *__restrict__ ops, RowArrays R, float *__restrict__ dyn_vel, float *__restrict__ dyn_acc,
                                                          float *__restrict__ dyn_rotvel, float *__restrict__ dyn_rotacc) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    const WriteOp w = ops[i];
    write_component(w, R, dyn_vel, dyn_acc, dyn_rotvel, dyn_rotacc);
}

// update_aabb_after_kinematic_change for the entities of one apply_change batch: rows[i] bit 31 = translation-only set
__global__ __launch_bounds__(256) void k_apply_rows(uint32_t m, const uint32_t *__restrict__ rows, RowArrays R, const uint32_t *__restrict__ row_cell,
                                                    const uint64_t *__restrict__ cell_key, const int32_t *__restrict__ sh_cells, uint32_t outline, uint32_t atomic,
                                                    TickHeader *th, uint32_t *__restrict__ mover_rows, uint32_t *__restrict__ oob_rows, uint32_t list_cap) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    const uint32_t r = rows[i] & 0x7FFFFFFFu; const bool translation_only = (rows[i] >> 31) != 0;
    const uint32_t fl = R.flags[r];
    if (fl & F_DEAD) return;
    const float pos[3] = { R.pos[r * 3 + 0], R.pos[r * 3 + 1], R.pos[r * 3 + 2] };
    const float rot[4] = { R.rot[r * 4 + 0], R.rot[r * 4 + 1], R.rot[r * 4 + 2], R.rot[r * 4 + 3] };
    atomicAdd(&th->n_changed, 1u);
    place_changed_entity(r, fl, fl, row_cell[r], pos, rot, translation_only, R, cell_key, sh_cells, outline, atomic, th, mover_rows, oob_rows, list_cap);
}

// A small change batch in ONE launch of one workgroup (the common case: the user entity re-inserted at the camera position every frame, logic_flow.rs:246-251, a handful of
// requests from the collision callbacks): the tick counters zeroed, the component writes, a workgroup barrier (the writes are this workgroup's own: workgroup scope is enough),
// update_aabb_after_kinematic_change for the moved entities, and the counters published to the host's mapped block -- the caller polls that word instead of a memset, two
// copies, two launches, a copy back and a stream synchronise.  ops / rows are read straight from mapped host memory (a few hundred bytes).
__global__ __launch_bounds__(256) void k_apply_small(uint32_t n_ops, const WriteOp *__restrict__ ops, uint32_t n_rows, const uint32_t *__restrict__ rows, RowArrays R,
                                                     float *__restrict__ dyn_vel, float *__restrict__ dyn_acc, float *__restrict__ dyn_rotvel, float *__restrict__ dyn_rotacc,
                                                     const uint32_t *__restrict__ row_cell, const uint64_t *__restrict__ cell_key, const int32_t *__restrict__ sh_cells, uint32_t outline, uint32_t atomic,
                                                     TickHeader *th, uint32_t *__restrict__ mover_rows, uint32_t *__restrict__ oob_rows, uint32_t list_cap, TickHeader *h_th, uint32_t seq) {
    const uint32_t tid = threadIdx.x;
    for (uint32_t i = tid; i < sizeof(TickHeader) / 4u; i += 256u) reinterpret_cast<uint32_t *>(th)[i] = 0u;
    if (tid < n_ops) { const WriteOp w = ops[tid]; write_component(w, R, dyn_vel, dyn_acc, dyn_rotvel, dyn_rotacc); }
    __threadfence_block();
    __syncthreads();
    if (tid < n_rows) {
        const uint32_t r = rows[tid] & 0x7FFFFFFFu; const bool translation_only = (rows[tid] >> 31) != 0;
        const uint32_t fl = R.flags[r];
        if (!(fl & F_DEAD)) {
            const float pos[3] = { R.pos[r * 3 + 0], R.pos[r * 3 + 1], R.pos[r * 3 + 2] };
            const float rot[4] = { R.rot[r * 4 + 0], R.rot[r * 4 + 1], R.rot[r * 4 + 2], R.rot[r * 4 + 3] };
            atomicAdd(&th->n_changed, 1u);
            place_changed_entity(r, fl, fl, row_cell[r], pos, rot, translation_only, R, cell_key, sh_cells, outline, atomic, th, mover_rows, oob_rows, list_cap);
        }
    }
    wait_own_stores();                                        // every wave: its stores and counter atomics have been performed
    __syncthreads();
    if (tid == 0) {
        const uint32_t a = __hip_atomic_load(&th->n_changed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), b = __hip_atomic_load(&th->n_rebucket, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT),
                       c2 = __hip_atomic_load(&th->n_oob, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        h_th->n_changed = a; h_th->n_rebucket = b; h_th->n_oob = c2; h_th->pad[0] = table_word_hash(a, 1u) ^ table_word_hash(b, 2u) ^ table_word_hash(c2, 3u) ^ table_word_hash(seq, 4u);
        publish_to_host(&h_th->ticket, seq);
    }
}

__global__ __launch_bounds__(256) void k_fold_tight_masked(uint32_t ncells, const uint64_t *cell_key, const uint32_t *cell_begin, const uint32_t *cell_nlocal,
                                                           const uint32_t *cell_nstatic, const uint32_t *rows, const Aabb *ent_aabb, Aabb *cell_tight,
                                                           uint32_t atomic, int too_many, const uint8_t *refold, const Aabb *carried) {
    uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= ncells) return;
    if (!refold[c]) { cell_tight[c] = carried[c]; return; }
    uint64_t key = cell_key[c];
    uint32_t n = cell_nlocal[c] + cell_nstatic[c];
    uint32_t adj = 20u + key_level(key) * 5u; if (adj > 50u) adj = 50u;
    Aabb u = { 0.f, 0.f, 0.f, 0.f, 0.f, 0.f };
    if (too_many && n > adj) u = key_to_aabb(key, atomic);
    else {
        uint32_t b = cell_begin[c];
        for (uint32_t i = 0; i < n; i++) { Aabb e = ent_aabb[rows[b + i]]; u = (i == 0) ? e : combine_aabb(u, e); }
    }
    cell_tight[c] = u;
}

// ---- incremental patches of the resident section table (re-bucket of movers, host-assisted bookkeeping) ----
// A ghost instance: the id and the 64 matrix bytes the reference's static render cache still holds for an entity that has since been
// woken, deleted or rewritten (render_flow.rs:549-594 is a snapshot; pipeline.rs:271 keeps the logic phase from refreshing it).
__global__ __launch_bounds__(256) void k_clone_rows(uint32_t m, const Pair32 *__restrict__ src_dst, uint32_t *__restrict__ row_id, float *__restrict__ row_mat) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m * 4u) return;
    const Pair32 p = src_dst[i >> 2];
    reinterpret_cast<float4 *>(row_mat + (size_t)p.val * 16)[i & 3u] = reinterpret_cast<const float4 *>(row_mat + (size_t)p.idx * 16)[i & 3u];
    if ((i & 3u) == 0) row_id[p.val] = row_id[p.idx];
}
__global__ __launch_bounds__(256) void k_shift_rows(uint32_t m, uint32_t *__restrict__ rows, uint32_t from, uint32_t delta) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < m) { const uint32_t r = rows[i]; if (r >= from && r != 0xFFFFFFFFu) rows[i] = r + delta; }
}
__global__ __launch_bounds__(256) void k_scatter32(uint32_t m, const Pair32 *__restrict__ pairs, uint32_t *__restrict__ dst) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < m) dst[pairs[i].idx] = pairs[i].val;
}
__global__ __launch_bounds__(256) void k_scatter64(uint32_t m, const Pair64 *__restrict__ pairs, uint64_t *__restrict__ dst) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < m) dst[pairs[i].idx] = pairs[i].val;
}
__global__ __launch_bounds__(256) void k_flag_ops(uint32_t m, const FlagOp *__restrict__ ops, uint8_t *__restrict__ flags) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < m) flags[ops[i].idx] = (uint8_t)((flags[ops[i].idx] & ops[i].and_mask) | ops[i].or_mask);      // one op per slot (the host merges)
}
// end_of_changes for the changed sections only (bounding_box_tree_v2.rs:1055-1130): same fold as k_fold_tight
__global__ __launch_bounds__(256) void k_fold_tight_list(uint32_t m, const uint32_t *__restrict__ slots, const uint64_t *cell_key, const uint32_t *cell_begin, const uint32_t *cell_nlocal,
                                                         const uint32_t *cell_nstatic, const uint32_t *rows, const Aabb *ent_aabb, Aabb *cell_tight, uint32_t atomic, int too_many) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    const uint32_t c = slots[i];
    if (c == 0xFFFFFFFFu) return;                                             // (an entry of the device-side re-bucket whose section was emptied)
    uint64_t key = cell_key[c];
    uint32_t n = cell_nlocal[c] + cell_nstatic[c];
    uint32_t adj = 20u + key_level(key) * 5u; if (adj > 50u) adj = 50u;
    Aabb u = { 0.f, 0.f, 0.f, 0.f, 0.f, 0.f };
    if (too_many && n > adj) u = key_to_aabb(key, atomic);
    else {
        uint32_t b = cell_begin[c];
        for (uint32_t k = 0; k < n; k++) { Aabb e = ent_aabb[rows[b + k]]; u = (k == 0) ? e : combine_aabb(u, e); }
    }
    cell_tight[c] = u;
}

// ECS::get_indexes_for_components (objects/ecs.rs:238-285) on the presence column: wave-ballot compaction of the matching live rows' ids
__global__ __launch_bounds__(256) void k_query_flags(uint32_t n, const uint32_t *__restrict__ flags, const uint32_t *__restrict__ row_id, uint32_t need_mask,
                                                     uint32_t *__restrict__ out_ids, uint32_t cap, uint32_t *count) {
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t fl = r < n ? flags[r] : F_DEAD;
    const bool hit = !(fl & (F_DEAD | F_PHANTOM)) && (fl & need_mask) == need_mask;      // (halo replicas belong to another shard's ECS)
    const uint64_t m = __ballot(hit);
    if (!m) return;
    uint32_t base = 0;
    if (lane_id() == 0) base = atomicAdd(count, (uint32_t)__popcll(m));
    base = __shfl(base, 0, 64);
    if (hit) { const uint32_t slot = base + mbcnt(m); if (slot < cap) out_ids[slot] = row_id[r]; }
}

// ---------------------------------------------------------------------------------------------
// Helpers of the device-side re-bucket (re_rebucket.hip): the overlay of sections created since the last full build, the gather of keys in sorted order,
// and the kernels that fetch the host mirrors of the sections the device patched (re_api.hip: sync_mirrors).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_rb_ovl_insert(uint32_t n, const Pair64 *__restrict__ pairs, RbTables T) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) rb_ovl_put(T, pairs[i].val, pairs[i].idx);
}
__global__ __launch_bounds__(256) void k_rb_gather_keys(uint32_t n, const uint32_t *__restrict__ perm, const uint64_t *__restrict__ op_key, uint64_t *__restrict__ key_sorted) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) key_sorted[i] = op_key[perm[i]];
}
// host mirrors on demand: the state of the sections the device patched (re_api.hip: sync_mirrors)
__global__ __launch_bounds__(256) void k_rb_gather_cells(uint32_t n, const uint32_t *__restrict__ slots, RbCells C, uint64_t *__restrict__ out_key, uint32_t *__restrict__ out_hdr) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t sl = slots[i];
    out_key[i] = C.cell_key[sl]; out_hdr[i * 4 + 0] = C.cell_begin[sl]; out_hdr[i * 4 + 1] = C.cell_cap[sl]; out_hdr[i * 4 + 2] = C.cell_nl[sl]; out_hdr[i * 4 + 3] = C.cell_ns[sl];
}
__global__ __launch_bounds__(256) void k_rb_gather_rows(uint32_t n, const uint32_t *__restrict__ slots, const uint32_t *__restrict__ offs, RbCells C, uint32_t *__restrict__ out_rows) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t sl = slots[i], b = C.cell_begin[sl], cnt = offs[i + 1] - offs[i];
    for (uint32_t k = 0; k < cnt; k++) out_rows[offs[i] + k] = C.rows[b + k];
}

// The lights RenderFlow::render finds near the camera (flows/render_flow.rs:249-254 -> shadow_flow.rs:494-513 find_nearby_world_sections_maps: the
// whole-world visibility query with an AABB culler of radius far_draw; then find_nearby_lights :455-487: the light sets of those unique sections and
// of the shared sections linked to them, world/bounding_box_tree_v2.rs:157-228).  One thread per light entity: is the section that holds it (or, for
// an entity of a shared section, any of the linked sections) a candidate of its level's box whose grid AABB intersects the culler?
__device__ __forceinline__ bool light_section_visible(uint64_t key, const LightQuery &Q) {
    const uint32_t lv = key_level(key);
    if (lv >= Q.max_level) return false;
    const LevelBox b = Q.box[lv];
    const uint32_t x = key_x(key), y = key_y(key), z = key_z(key);
    if (!in_box(x, y, z, b)) return false;
    const float ll = b.level_length;
    const float fx = (float)(b.bx + ((x - b.bx) & 0xFFFFu)) * ll, fy = (float)(b.by + ((y - b.by) & 0xFFFFu)) * ll, fz = (float)(b.bz + ((z - b.bz) & 0xFFFFu)) * ll;   // visible_world_flow.rs:73-82
    const Aabb &c = Q.culler;
    return c.xmin <= fx + ll && c.xmax >= fx && c.ymin <= fy + ll && c.ymax >= fy && c.zmin <= fz + ll && c.zmax >= fz;      // StaticAABB::intersect (aabb.rs:68-73)
}
__global__ __launch_bounds__(256) void k_visible_lights(uint32_t n, const uint32_t *__restrict__ light_rows, const uint32_t *__restrict__ flags, const uint32_t *__restrict__ row_id,
                                                        const uint32_t *__restrict__ row_cell, const uint64_t *__restrict__ cell_key, const uint8_t *__restrict__ cell_flags,
                                                        const int32_t *__restrict__ sh_cells, LightQuery Q, uint32_t *__restrict__ out_ids, uint32_t cap, uint32_t *count) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t r = light_rows[i], fl = flags[r], rc = row_cell[r];
    if ((fl & F_DEAD) || !(fl & Q.type_flag) || rc == ROW_CELL_NONE) return;
    bool vis = false;
    if (!(rc & ROW_CELL_SHARED)) vis = !(cell_flags[rc] & CF_PAD) && light_section_visible(cell_key[rc], Q);
    else {
        const uint32_t s = rc & ~ROW_CELL_SHARED;
        for (int k = 0; k < 8 && !vis; k++) { const int32_t c = sh_cells[s * 8 + k]; if (c >= 0 && !(cell_flags[c] & CF_PAD)) vis = light_section_visible(cell_key[c], Q); }
    }
    if (vis) { const uint32_t o = atomicAdd(count, 1u); if (o < cap) out_ids[o] = row_id[r]; }
}

// re_export_entities: the complete state of a few entities (a migrant's record for another GPU), gathered row by row
__global__ __launch_bounds__(256) void k_export_rows(uint32_t m, const uint32_t *__restrict__ rows, const uint32_t *__restrict__ dyn_slot, RowArrays R, const float *__restrict__ dyn_vel,
                                                     const float *__restrict__ dyn_acc, const float *__restrict__ dyn_rotvel, const float *__restrict__ dyn_rotacc, ExportRec *__restrict__ out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    const uint32_t r = rows[i], j = dyn_slot[i];
    ExportRec e{};
    e.id = R.id[r]; e.flags = R.flags[r];
    const Aabb o = R.orig[r]; e.orig[0] = o.xmin; e.orig[1] = o.xmax; e.orig[2] = o.ymin; e.orig[3] = o.ymax; e.orig[4] = o.zmin; e.orig[5] = o.zmax;
    for (int q = 0; q < 3; q++) { e.pos[q] = R.pos[(size_t)r * 3 + q]; e.scale[q] = R.scale[(size_t)r * 3 + q]; }
    for (int q = 0; q < 4; q++) e.rot[q] = R.rot[(size_t)r * 4 + q];
    e.rotvel[0] = e.rotacc[0] = 1.0f;
    if (j != 0xFFFFFFFFu) {
        for (int q = 0; q < 3; q++) { e.vel[q] = dyn_vel[(size_t)j * 3 + q]; e.acc[q] = dyn_acc[(size_t)j * 3 + q]; }
        for (int q = 0; q < 4; q++) { e.rotvel[q] = dyn_rotvel[(size_t)j * 4 + q]; e.rotacc[q] = dyn_rotacc[(size_t)j * 4 + q]; }
    }
    out[i] = e;
}

// The slab headers of an all-gathered frame for the host (re_allgather_visible): one wave behind the collective copies the 4-word header of every rank's slab
// into mapped host memory and publishes the exchange's sequence number there -- the host polls that word (one PCIe write's latency) instead of
// synchronising the stream and issuing one blocking 16-byte copy per rank.
__global__ __launch_bounds__(64) void k_gather_headers(uint32_t n_ranks, const uint32_t *__restrict__ recv, uint32_t words_per_rank, uint32_t *h_hdr, uint32_t *h_seq, uint32_t seq) {
    for (uint32_t r = threadIdx.x; r < n_ranks; r += 64u) {
        const uint4 v = *reinterpret_cast<const uint4 *>(recv + (size_t)r * words_per_rank);
        *reinterpret_cast<uint4 *>(h_hdr + (size_t)r * 4u) = v;
    }
    wait_own_stores();                                       // (one wave: every lane's host stores have left before lane 0 publishes)
    if (threadIdx.x == 0) publish_to_host(h_seq, seq);
}

// gathers the visible sections of the last cull for re_debug_get_visible_sections
__global__ __launch_bounds__(256) void k_collect_visible(uint32_t ncells, const uint32_t *cell_stamp, uint32_t frame, uint32_t *out_idx, uint8_t *out_mult, uint32_t cap, uint32_t *count) {
    uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= ncells) return;
    uint32_t st = cell_stamp[c];
    if ((st >> 2) == frame) { uint32_t s = atomicAdd(count, 1u); if (s < cap) { out_idx[s] = c; out_mult[s] = (uint8_t)(st & 3u); } }
}

}  // namespace re
