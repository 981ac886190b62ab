// Collision broad phase on the resident world: LogicFlow::handle_collisions (flows/logic_flow.rs:452-651) over
// BoundingBoxTree::find_related_entities (world/bounding_box_tree_v2.rs:950-1048), for gfx950.
//
// The reference walks, per world section that holds a moved entity, the related_world_sections graph (every existing ancestor
// and descendant section, transitively), keeps the sections whose AABB lies within 200 units of the camera, and tests the moved
// entities of that section against the local entities of those sections and the entities of their shared sections.
// The transitive closure of "ancestor or descendant" from a section R is the subtree of R's topmost existing ancestor T(R), so
//     X is related to R   <=>   T(X) is R or an ancestor of R,
// and because only sections near the camera survive the 200-unit test, T() is needed only for the few sections around the
// camera.  No hash probes: ancestry is key arithmetic (parent = level + 1, index / 2), existence comes from one pass over the
// section table that collects the sections around the camera (all levels).  Everything after that pass works on short lists.
#include "re_kernels.h"

namespace re {

__device__ __forceinline__ bool key_pad(uint64_t k) { return (k & 0xFFFFFFFFFFFFull) == 0xFFFFFFFFFFFFull; }
// a is d or an ancestor of d
__device__ __forceinline__ bool key_covers(uint64_t a, uint64_t d) {
    const uint32_t la = (uint32_t)(a >> 48), ld = (uint32_t)(d >> 48);
    if (la < ld || la - ld > 15u) return false;
    const uint32_t s = la - ld;
    return (((uint32_t)(d >> 32) & 0xFFFFu) >> s) == ((uint32_t)(a >> 32) & 0xFFFFu) && (((uint32_t)(d >> 16) & 0xFFFFu) >> s) == ((uint32_t)(a >> 16) & 0xFFFFu) &&
           (((uint32_t)d & 0xFFFFu) >> s) == ((uint32_t)a & 0xFFFFu);
}
__device__ __forceinline__ bool aabb_intersect(const Aabb &a, const Aabb &b) {   // StaticAABB::intersect (aabb.rs:68-73): closed intervals (range.rs:71)
    return a.xmin <= b.xmax && a.xmax >= b.xmin && a.ymin <= b.ymax && a.ymax >= b.ymin && a.zmin <= b.zmax && a.zmax >= b.zmin;
}

// Pass 1: the existing sections whose grid cell comes within 200 + 4 sides of the camera (per level; every section the 200-unit
// test can keep, every linking section of a shared section it can keep, and all their ancestors lie inside), with the result of
// the reference's test on the stored section AABB (:553-558).  K32: the compact stream keys of k_scan_cull (x:9|z:9|y:9 with guard
// bits, the level of a 512-key chunk in chunk_level), four per lane.
__device__ __forceinline__ void region_consider(uint64_t k, uint32_t slot, const Aabb *__restrict__ cell_tight, const FrameParams *__restrict__ Pp, uint32_t atomic,
                                                ColHeader *hdr, ColRegion *region, uint32_t region_cap, uint32_t *high, uint32_t high_cap) {
    const uint32_t level = (uint32_t)(k >> 48);
    if (level >= 24u) return;
    const float side = (float)atomic * (float)(1u << level), reach = COLLISION_DISTANCE + 4.0f * side;
    const float x0 = side * (float)((uint32_t)(k >> 32) & 0xFFFFu), z0 = side * (float)((uint32_t)(k >> 16) & 0xFFFFu), y0 = side * (float)((uint32_t)k & 0xFFFFu);
    const float cx = Pp->cam[0], cy = Pp->cam[1], cz = Pp->cam[2];
    if (x0 > cx + reach || x0 + side < cx - reach || y0 > cy + reach || y0 + side < cy - reach || z0 > cz + reach || z0 + side < cz - reach) return;
    const Aabb t = cell_tight[slot];
    ColRegion e; e.key = k; e.slot = slot; e.near = !(distance_to_aabb(t, cx, cy, cz) > COLLISION_DISTANCE) ? 1u : 0u; e.top = k;
    const uint32_t at = atomicAdd(&hdr->n_region, 1u);
    if (at < region_cap) region[at] = e;
    if (level && at < region_cap) { const uint32_t h = atomicAdd(&hdr->n_high, 1u); if (h < high_cap) high[h] = at; }
}
template <bool K32>
__global__ __launch_bounds__(256) void k_col_region(uint32_t ncells, const void *__restrict__ keys, const uint32_t *__restrict__ chunk_level, const Aabb *__restrict__ cell_tight,
                                                    const FrameParams *__restrict__ Pp, uint32_t atomic, ColHeader *hdr, ColRegion *region, uint32_t region_cap, uint32_t *high, uint32_t high_cap) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if constexpr (K32) {
        const uint32_t base = i * 4u;
        if (base >= ncells) return;
        const uint4 q = reinterpret_cast<const uint4 *>(keys)[i];            // the compact array is padded to whole quads with padding keys
        const uint32_t level = chunk_level[base / WAVE_KEYS32] & (MAX_LEVELS - 1);
        const uint32_t v4[4] = { q.x, q.y, q.z, q.w };
#pragma unroll
        for (uint32_t h = 0; h < 4; h++) {
            const uint32_t v = v4[h];
            if ((int32_t)v < 0 || base + h >= ncells) continue;             // padding / spare slot
            region_consider(pack_key(level, (v >> 20) & 0x1FFu, (v >> 10) & 0x1FFu, v & 0x1FFu), base + h, cell_tight, Pp, atomic, hdr, region, region_cap, high, high_cap);
        }
    } else {
        if (i >= ncells) return;
        const uint64_t k = reinterpret_cast<const uint64_t *>(keys)[i];
        if (key_pad(k)) return;
        region_consider(k, i, cell_tight, Pp, atomic, hdr, region, region_cap, high, high_cap);
    }
}
template __global__ void k_col_region<false>(uint32_t, const void *, const uint32_t *, const Aabb *, const FrameParams *, uint32_t, ColHeader *, ColRegion *, uint32_t, uint32_t *, uint32_t);
template __global__ void k_col_region<true>(uint32_t, const void *, const uint32_t *, const Aabb *, const FrameParams *, uint32_t, ColHeader *, ColRegion *, uint32_t, uint32_t *, uint32_t);

// the shared sections the 200-unit test keeps (:561-566), with their linking sections
__global__ __launch_bounds__(256) void k_col_shared(uint32_t nsh, const Aabb *__restrict__ sh_aabb, const int32_t *__restrict__ sh_cells, const uint32_t *__restrict__ sh_nact,
                                                    const uint32_t *__restrict__ sh_nstat, const uint64_t *__restrict__ cell_key, const FrameParams *__restrict__ Pp,
                                                    ColHeader *hdr, ColShared *out, uint32_t cap) {
    const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= nsh || sh_nact[s] == 0u) return;                                // only the non-static members are searched
    if (distance_to_aabb(sh_aabb[s], Pp->cam[0], Pp->cam[1], Pp->cam[2]) > COLLISION_DISTANCE) return;
    ColShared e; e.s = s; e.nk = 0;
    for (int k = 0; k < 8; k++) { const int32_t c = sh_cells[s * 8 + k]; if (c >= 0) e.top[e.nk++] = cell_key[c]; }
    const uint32_t at = atomicAdd(&hdr->n_shared, 1u);
    if (at < cap) out[at] = e;
}

// topmost existing ancestor of every listed section (itself when it has none)
__device__ __forceinline__ uint64_t top_of(uint64_t k, const ColRegion *region, const uint32_t *high, uint32_t nhigh) {
    uint64_t top = k;
    for (uint32_t h = 0; h < nhigh; h++) { const uint64_t a = region[high[h]].key; if ((a >> 48) > (top >> 48) && key_covers(a, k)) top = a; }
    return top;
}
// ... and the short list the pair kernel walks: the near sections that hold non-static entities
__global__ __launch_bounds__(256) void k_col_tops(ColHeader *hdr, ColRegion *region, uint32_t region_cap, const uint32_t *high, uint32_t high_cap, ColShared *shared, uint32_t shared_cap,
                                                  const uint32_t *__restrict__ cell_begin, const uint32_t *__restrict__ cell_nlocal, ColNear *near, uint32_t near_cap) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t nr = min(hdr->n_region, region_cap), nh = min(hdr->n_high, high_cap), ns = min(hdr->n_shared, shared_cap);
    if (i < nr) {
        const ColRegion x = region[i];
        const uint32_t nl = x.near ? cell_nlocal[x.slot] : 0u;
        if (nl) {
            ColNear e; e.top = top_of(x.key, region, high, nh); e.begin = cell_begin[x.slot]; e.n = nl;
            const uint32_t at = atomicAdd(&hdr->n_near, 1u);
            if (at < near_cap) near[at] = e;
        }
    }
    if (i < ns) for (uint32_t k = 0; k < shared[i].nk; k++) shared[i].top[k] = top_of(shared[i].top[k], region, high, nh);
}

// 64-bit open-addressing table: section key -> smallest order word of the moved entities that touch it
__device__ __forceinline__ uint32_t col_hash(uint64_t x, uint32_t mask) { x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; return (uint32_t)x & mask; }
__device__ __forceinline__ uint32_t first_touch_insert(unsigned long long *tab_key, unsigned long long *tab_min, uint32_t mask, uint64_t key, unsigned long long order) {
    uint32_t h = col_hash(key, mask);
    for (uint32_t probe = 0; probe <= mask; probe++, h = (h + 1u) & mask) {
        const unsigned long long prev = atomicCAS(&tab_key[h], ~0ull, (unsigned long long)key);
        if (prev == ~0ull || prev == key) { atomicMin(&tab_min[h], order); return h; }
    }
    return 0xFFFFFFFFu;
}
__device__ __forceinline__ unsigned long long first_touch_lookup(const unsigned long long *tab_key, const unsigned long long *tab_min, uint32_t mask, uint64_t key) {
    uint32_t h = col_hash(key, mask);
    for (uint32_t probe = 0; probe <= mask; probe++, h = (h + 1u) & mask) { if (tab_key[h] == key) return tab_min[h]; if (tab_key[h] == ~0ull) break; }
    return ~0ull;
}

// Pass 2: moved_entities (update_positions -> apply_kinematics, logic_flow.rs:308-358, 443-446): every entity the tick of this
// frame processes that carries Velocity or VelocityRotation and CanCauseCollisions -- the predicate of k_tick -- once per listing
// of its section in visible_sections_vec; plus the user entity (UserAlwaysCausesCollisions, :236-240).  One entry per section the
// entity is stored under (one, or the linking sections of its shared section: relevant_world_sections, :479-514).
__global__ __launch_bounds__(256) void k_col_moved(uint32_t ndyn, const uint32_t *__restrict__ dyn_row, const uint32_t *__restrict__ dyn_cell, uint32_t user_row, uint32_t user_cell,
                                                   RowArrays R, const uint64_t *__restrict__ cell_key, const uint32_t *__restrict__ cell_stamp, const uint8_t *__restrict__ cell_flags,
                                                   const int32_t *__restrict__ sh_cells, const Aabb *__restrict__ sh_aabb, const FrameParams *__restrict__ Pp,
                                                   ColHeader *hdr, ColMoved *moved, uint32_t moved_cap, uint8_t *row_moved, unsigned long long *tab_key, unsigned long long *tab_min, uint32_t tab_mask) {
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j > ndyn || (j == ndyn && user_row == ROW_CELL_NONE)) return;
    const bool is_user = j == ndyn;
    const uint32_t r = is_user ? user_row : dyn_row[j], rc = is_user ? user_cell : dyn_cell[r];      // (dyn_cell: the row -> section column; dynamic entities added later sit in any row)
    const uint32_t fl = R.flags[r];
    if ((fl & F_DEAD) || rc == ROW_CELL_NONE) return;
    const FrameParams &P = *Pp;
    uint32_t mult = 0;
    if (is_user) mult = 1;
    else {
        if (!(fl & F_CAN_COLLIDE) || !(fl & (F_HAS_VEL | F_HAS_ROTVEL))) return;
        if (!(rc & ROW_CELL_SHARED)) {
            const uint32_t stamp = cell_stamp[rc];
            const bool vis = (stamp >> 2) == P.frame;
            if (!(fl & F_STATIC) && vis) mult = stamp & 3u;                  // the section is listed once or twice (logic box and frustum)
            else if ((fl & F_ALWAYS_EXEC) && !vis) mult = 1;
        } else {
            const uint32_t s = rc & ~ROW_CELL_SHARED;
            bool anyvis = false, act = false;
            for (int k = 0; k < 8; k++) {
                const int32_t c = sh_cells[s * 8 + k];
                if (c >= 0 && (cell_stamp[c] >> 2) == P.frame) { anyvis = true; if (!(cell_flags[c] & CF_STATIC_SECTION)) act = true; }
            }
            bool inview = false;
            if (act && !(fl & F_STATIC)) { const Aabb sa = sh_aabb[s]; inview = logic_aabb_in_view(P.lookahead, P.cam[0], P.cam[1], P.cam[2], sa) || frustum_aabb_visible(P.planes, sa); }
            if ((!(fl & F_STATIC) && act && inview) || ((fl & F_ALWAYS_EXEC) && !anyvis)) mult = 1;
        }
    }
    if (!mult) return;
    row_moved[r] = 1;
    const unsigned long long order = is_user ? (1ull << 32) : (unsigned long long)R.id[r];   // ascending EntityId, the user entity last
    if (!(rc & ROW_CELL_SHARED)) {
        const uint32_t ts = first_touch_insert(tab_key, tab_min, tab_mask, cell_key[rc], order);
        const uint32_t at = atomicAdd(&hdr->n_moved, 1u);
        if (at < moved_cap) { ColMoved m; m.key = cell_key[rc]; m.row = r; m.info = mult << 1; m.order = order; m.tslot = ts; moved[at] = m; }
    } else {
        const uint32_t s = rc & ~ROW_CELL_SHARED;
        for (int k = 0; k < 8; k++) {
            const int32_t c = sh_cells[s * 8 + k];
            if (c < 0) continue;
            const uint32_t ts = first_touch_insert(tab_key, tab_min, tab_mask, cell_key[c], order);
            const uint32_t at = atomicAdd(&hdr->n_moved, 1u);
            if (at < moved_cap) { ColMoved m; m.key = cell_key[c]; m.row = r; m.info = (mult << 1) | 1u; m.order = order; m.tslot = ts; moved[at] = m; }
        }
    }
}

// Pass 3: one wave per (section, moved entity) entry: the collision tests of collision_fn (:619-650)
__device__ __forceinline__ void col_emit(ColHeader *hdr, uint2 *pairs, uint32_t cap, uint32_t a, uint32_t b) {
    const uint32_t at = atomicAdd(&hdr->n_pairs, 1u);
    if (at < cap) pairs[at] = make_uint2(a, b);
}
__device__ __forceinline__ void col_test(uint32_t me_row, uint32_t me_id, const Aabb &a, uint32_t o, uint32_t mult, RowArrays R, const uint8_t *row_moved, ColHeader *hdr, uint2 *pairs, uint32_t cap) {
    if (R.flags[o] & F_DEAD) return;
    const bool self = row_moved[o] != 0;
    if (self && o == me_row) return;                                         // :627-630
    if (!aabb_intersect(a, R.aabb[o])) return;
    const uint32_t oid = R.id[o];
    for (uint32_t m = 0; m < mult; m++) {
        col_emit(hdr, pairs, cap, me_id, oid);                               // apply_collision_only_to_self(moved, other)
        if (!self) col_emit(hdr, pairs, cap, oid, me_id);                    // both directions for an entity that did not move (:640-647)
    }
}
__global__ __launch_bounds__(256) void k_col_pairs(ColHeader *hdr, const ColMoved *moved, uint32_t moved_cap, const ColNear *near, uint32_t near_cap, const ColShared *shared,
                                                   uint32_t shared_cap, RowArrays R, const uint32_t *__restrict__ sh_begin, const uint32_t *__restrict__ sh_nact,
                                                   const uint32_t *__restrict__ rows, const uint8_t *__restrict__ row_moved, const unsigned long long *tab_min,
                                                   uint2 *pairs, uint32_t pair_cap) {
    const uint32_t lane = threadIdx.x & 63u, nm = min(hdr->n_moved, moved_cap);
    const uint32_t nn = min(hdr->n_near, near_cap), ns = min(hdr->n_shared, shared_cap);
    for (uint32_t e = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); e < nm; e += gridDim.x * (blockDim.x >> 6)) {   // wave-uniform loop
        const ColMoved m = moved[e];
        // the entity that creates a section's entry through a Shared lookup is not pushed into it (:488-498)
        if ((m.info & 1u) && m.tslot != 0xFFFFFFFFu && tab_min[m.tslot] == m.order) continue;
        const uint32_t mult = (m.info >> 1) & 3u, me_id = R.id[m.row];
        const Aabb a = R.aabb[m.row];
        for (uint32_t i = lane; i < nn; i += 64u) {
            const ColNear x = near[i];
            if (!key_covers(x.top, m.key)) continue;
            for (uint32_t k = 0; k < x.n; k++) col_test(m.row, me_id, a, rows[x.begin + k], mult, R, row_moved, hdr, pairs, pair_cap);
        }
        for (uint32_t i = lane; i < ns; i += 64u) {
            const ColShared s = shared[i];
            bool rel = false;
            for (uint32_t k = 0; k < s.nk; k++) rel |= key_covers(s.top[k], m.key);
            if (!rel) continue;
            const uint32_t b = sh_begin[s.s], n = sh_nact[s.s];              // SharedWorldSectionEntities.entities: the non-static members (bounding_box_tree_v2.rs:292-301)
            for (uint32_t k = 0; k < n; k++) col_test(m.row, me_id, a, rows[b + k], mult, R, row_moved, hdr, pairs, pair_cap);
        }
    }
}

// leaves the "moved" row bytes and the first-touch table clean for the next call (instead of clearing 10 MB + the table every time)
__global__ __launch_bounds__(256) void k_col_clear(const ColHeader *hdr, const ColMoved *moved, uint32_t moved_cap, uint8_t *row_moved, unsigned long long *tab_key, unsigned long long *tab_min,
                                                   ColHeader *h_hdr, uint32_t call) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) {                                                            // the counts, into mapped host memory: the host polls pad[1] instead of synchronising the stream
        ColHeader h = *hdr; h.pad[1] = 0;
        h.pad[0] = table_word_hash(h.n_region, 1u) ^ table_word_hash(h.n_high, 2u) ^ table_word_hash(h.n_shared, 3u) ^ table_word_hash(h.n_moved, 4u) ^ table_word_hash(h.n_pairs, 5u) ^ table_word_hash(h.n_near, 6u) ^ table_word_hash(call, 7u);   // seal
        *h_hdr = h; publish_to_host(&h_hdr->pad[1], call);
    }
    if (i >= min(hdr->n_moved, moved_cap)) return;
    const ColMoved m = moved[i];
    row_moved[m.row] = 0;
    if (m.tslot != 0xFFFFFFFFu) { tab_key[m.tslot] = ~0ull; tab_min[m.tslot] = ~0ull; }
}

}  // namespace re
