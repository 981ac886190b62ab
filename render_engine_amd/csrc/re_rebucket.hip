// re_rebucket.hip -- the re-bucket bookkeeping of a tick's movers on the device, shared world sections included.
//
// update_entity_in_tree -> BoundingBoxTree::add_entity (which first removes the entity from its previous section) for every mover whose section
// changed, then end_of_changes and update_static_world_sections (helper_things/entity_change_helpers.rs:217-262, 325-351;
// world/bounding_box_tree_v2.rs:563-942, 1055-1213).  Round 2 (re_kernels.hip: k_rb_*) took the movers between UNIQUE sections that no shared section
// links and left the rest -- about half of the placement changes of a scene whose boxes straddle section borders for a tick -- to the host.  Here a
// placement is either kind:
//   * two ops per mover, remove from the old placement and add to the new one, keyed by the placement -- a unique section's key, or the hash of a
//     shared section's id (its 2..8 linked keys) with the top bit set -- and by the reference's order (translation-only movers first, then ascending
//     EntityId, remove before add); sorted, one thread replays a placement's ops in order;
//   * the SHARED placements first: what the section ends as, and every time it is created or emptied on the way it emits a link op (+1 / -1) for each
//     unique section it links, stamped with the order of the op that caused it;
//   * then the unique placements with those link ops merged in by order, so that a section's existence (members or links) and its share of
//     total_world_aabb_combining evolve exactly as in the reference's one sequential pass;
//   * the host reads one status block (is the batch eligible, is there slack: free slots per level, free shared entries, row pool), hands the free
//     slots over, and the apply kernels rewrite the segments of the row pool, create / retire sections and shared entries (stable indices, holes
//     allowed), recompute the links' slots, the shared AABBs (last member), the static-section flags (both loops of update_static_world_sections) and
//     the tight AABBs of the changed sections.
// The host only notes what changed; its mirrors follow on demand (re_api.hip: sync_mirrors).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "re_kernels.h"
#include "re_math.h"

namespace re {

__device__ __forceinline__ uint64_t sh_id_pkey(const uint64_t *keys, uint32_t nk) {      // placement key of a shared section: top bit set, never ~0
    uint64_t h = 0x9E3779B97F4A7C15ull ^ nk;
    for (uint32_t k = 0; k < nk; k++) { h ^= keys[k]; h *= 0xFF51AFD7ED558CCDull; h ^= h >> 32; }
    uint64_t p = RB2_SHARED_BIT | (h >> 1);
    if (p == ~0ull) p ^= 1ull;
    return p;
}
__device__ __forceinline__ bool sh_same_id(const uint64_t *a, uint32_t na, const uint64_t *b, uint32_t nb) {
    if (na != nb) return false;
    for (uint32_t k = 0; k < na; k++) if (a[k] != b[k]) return false;
    return true;
}
// SharedIdPub::operator< (re_api.hip): keys lexicographic, then the count -- the order update_static_world_sections visits changed shared sections in
__device__ __forceinline__ bool sh_id_less(const uint64_t *a, uint32_t na, const uint64_t *b, uint32_t nb) {
    const uint32_t m = na < nb ? na : nb;
    for (uint32_t i = 0; i < m; i++) if (a[i] != b[i]) return a[i] < b[i];
    return na < nb;
}
// index of a shared section by id: -1 absent, -2 another id with the same placement key (the batch goes to the host path)
__device__ __forceinline__ int32_t sh_lookup(const ShTable &S, uint64_t pkey, const uint64_t *keys, uint32_t nk) {
    for (uint32_t h = rb_hash(pkey) & S.hmask;; h = (h + 1u) & S.hmask) {
        const unsigned long long k = S.hkeys[h];
        if (k == ~0ull) return -1;
        if (k == pkey) {
            const uint32_t idx = S.hidx[h];
            if (idx == 0xFFFFFFFFu) return -1;                              // retired
            return sh_same_id(S.keys + (size_t)idx * 8, S.nk[idx], keys, nk) ? (int32_t)idx : -2;
        }
    }
}
__device__ __forceinline__ void sh_hash_put(const ShTable &S, uint64_t pkey, uint32_t idx) {
    for (uint32_t h = rb_hash(pkey) & S.hmask;; h = (h + 1u) & S.hmask) {
        const unsigned long long prev = atomicCAS(&S.hkeys[h], ~0ull, (unsigned long long)pkey);
        if (prev == ~0ull || prev == pkey) { S.hidx[h] = idx; return; }
    }
}
__global__ __launch_bounds__(256) void k_rb2_hash_insert(uint32_t n, ShTable S) {
    const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n || !S.nk[s]) return;
    sh_hash_put(S, sh_id_pkey(S.keys + (size_t)s * 8, S.nk[s]), s);
}

// ---- small batches: (placement key, reference order) sorted by ONE workgroup (a bitonic network in LDS) instead of two radix sorts and a gather -- a batch of a
// handful of movers (a change request that carries the user entity across a section border) is bound by the number of launches, not by the sorting
__device__ __forceinline__ void rb2_sort_block(uint32_t n, const uint64_t *key, const uint64_t *ord, uint64_t *key_sorted, uint32_t *perm, uint64_t *s_key, uint64_t *s_ord, uint32_t *s_idx) {      // a workgroup of 1,024
    const uint32_t tid = threadIdx.x;
    uint32_t m = 1; while (m < n) m <<= 1;                                  // (n <= RB2_SORT_SMALL: the host chose this kernel)
    for (uint32_t i = tid; i < m; i += 1024u) { const bool on = i < n; s_key[i] = on ? key[i] : ~0ull; s_ord[i] = on ? (ord ? ord[i] : 0ull) : ~0ull; s_idx[i] = on ? i : 0xFFFFFFFFu; }
    __syncthreads();
    for (uint32_t k = 2; k <= m; k <<= 1)
        for (uint32_t j = k >> 1; j > 0; j >>= 1) {
            for (uint32_t i = tid; i < m; i += 1024u) {
                const uint32_t l = i ^ j;
                if (l > i) {
                    const bool up = (i & k) == 0;
                    const uint64_t ka = s_key[i], kb = s_key[l], oa = s_ord[i], ob = s_ord[l]; const uint32_t ia = s_idx[i], ib = s_idx[l];
                    const bool a_gt_b = ka != kb ? ka > kb : (oa != ob ? oa > ob : ia > ib);      // (the op index last: a total order, so the network's result is the stable order)
                    if (a_gt_b == up) { s_key[i] = kb; s_key[l] = ka; s_ord[i] = ob; s_ord[l] = oa; s_idx[i] = ib; s_idx[l] = ia; }
                }
            }
            __syncthreads();
        }
    for (uint32_t i = tid; i < n; i += 1024u) { key_sorted[i] = s_key[i]; perm[i] = s_idx[i]; }
}
__global__ __launch_bounds__(1024) void k_rb2_sort_small(uint32_t n, const uint64_t *__restrict__ key, const uint64_t *__restrict__ ord, uint64_t *__restrict__ key_sorted, uint32_t *__restrict__ perm, const uint32_t *n_extra) {
    __shared__ uint64_t s_key[RB2_SORT_SMALL], s_ord[RB2_SORT_SMALL]; __shared__ uint32_t s_idx[RB2_SORT_SMALL];
    if (n_extra) n = min(n + *n_extra, RB2_SORT_SMALL);                    // (a count an earlier kernel of the batch left on the device: the host has not read it)
    rb2_sort_block(n, key, ord, key_sorted, perm, s_key, s_ord, s_idx);
}

// ---- phase 1: two ops per mover --------------------------------------------------------------------------------------------------------------
// The list may end with the rows a change batch DELETES (RB2_MOVER_DELETED; DeleteRequest -> remove_entity, entity_change_helpers.rs:109-136): they only have a remove op, ordered in
// front of every mover in the order of the batch (apply_change removes them inline, before the kinematic re-adds).
// (the bodies of phases 1-3 are functions of one op / one segment head: a kernel of their own for batches of any size, steps of k_rb2_plan_small for small ones.  No __restrict__
// here: in the fused kernel one step reads what the step before wrote)
__device__ __forceinline__ void rb2_ops_one(uint32_t i, uint32_t m, const uint32_t *movers, const RowArrays &R, const RbCells &C, const ShTable &S, uint32_t outline, uint32_t atomic,
                                            uint64_t *op_key, uint64_t *op_key2, uint64_t *op_ord, uint32_t *op_row, uint32_t *op_idx, uint64_t *mk, uint8_t *mnk, uint32_t *host_list, Rb2Status *st) {
    if (i >= m) return;
    if (movers[i] & RB2_MOVER_DELETED) {
        const uint32_t r = movers[i] & 0x1FFFFFFFu, rc = C.row_cell[r];
        uint64_t pold = ~0ull;                                              // (an entity that is in no section -- it left the world earlier -- has nothing to remove)
        if (rc != ROW_CELL_NONE) {
            if (rc & ROW_CELL_SHARED) { const uint32_t s = rc & ~ROW_CELL_SHARED; if (s < S.cap && S.nk[s]) pold = sh_id_pkey(S.keys + (size_t)s * 8, S.nk[s]); else st->fallback = 1u; }
            else pold = C.cell_key[rc];
        }
        for (int k = 0; k < 8; k++) mk[(size_t)i * 8 + k] = 0ull;
        mnk[i] = 0;
        op_key[2 * i] = pold; op_ord[2 * i] = (uint64_t)i << 1; op_row[2 * i] = r | RB_REMOVE; op_idx[2 * i] = 2 * i;
        op_key[2 * i + 1] = ~0ull; op_ord[2 * i + 1] = ((uint64_t)i << 1) | 1ull; op_row[2 * i + 1] = r; op_idx[2 * i + 1] = 2 * i + 1;
        op_key2[2 * i] = (pold & RB2_SHARED_BIT) ? ~0ull : pold; op_key2[2 * i + 1] = ~0ull;
        return;
    }
    if (movers[i] & RB2_MOVER_ADDED) {                                       // AddEntity -> add_entity(id, aabb, add_if_out_bounds = false, ..): out of bounds -> not inserted
        const uint32_t r = movers[i] & 0x1FFFFFFFu;
        Aabb bv = R.aabb[r];
        const bool oob = normalize_aabb(&bv, (float)outline);
        uint64_t keys[8];
        int nk = oob ? 0 : assign_sections(bv, atomic, keys);
        if (nk < 0) nk = 0;
        if (C.row_cell[r] != ROW_CELL_NONE || (R.flags[r] & F_STATIC)) st->fallback = 1u;      // (not a row the host may have listed as added here)
        const uint64_t pnew = nk == 0 ? ~0ull : (nk == 1 ? keys[0] : sh_id_pkey(keys, (uint32_t)nk));
        for (int k = 0; k < 8; k++) mk[(size_t)i * 8 + k] = k < nk ? keys[k] : 0ull;
        mnk[i] = (uint8_t)nk;
        op_key[2 * i] = ~0ull; op_ord[2 * i] = (uint64_t)i << 1; op_row[2 * i] = r | RB_REMOVE; op_idx[2 * i] = 2 * i;
        op_key[2 * i + 1] = pnew; op_ord[2 * i + 1] = ((uint64_t)i << 1) | 1ull; op_row[2 * i + 1] = r; op_idx[2 * i + 1] = 2 * i + 1;
        op_key2[2 * i] = ~0ull; op_key2[2 * i + 1] = (pnew & RB2_SHARED_BIT) ? ~0ull : pnew;
        return;
    }
    const uint32_t w = movers[i], r = w & 0x1FFFFFFFu, fl = R.flags[r], rc = C.row_cell[r];
    Aabb bv = R.aabb[r];
    normalize_aabb(&bv, (float)outline);
    uint64_t keys[8];
    const int nk = assign_sections(bv, atomic, keys);                      // add_entity with add_if_out_bounds = true: the box is clamped
    if (rc == ROW_CELL_NONE || (fl & F_DEAD) || nk < 1) st->fallback = 1u;  // (not a mover the tick can have listed)
    const bool host = (fl & F_STATIC) || rc == ROW_CELL_NONE || nk < 1;     // static rows (an always-execute entity of a static set) keep the host path: their removal touches the changed-static set
    if (host) host_list[atomicAdd(&st->n_host, 1u)] = w;
    uint64_t pold = ~0ull, pnew = ~0ull;
    if (!host) {
        if (rc & ROW_CELL_SHARED) { const uint32_t s = rc & ~ROW_CELL_SHARED; pold = s < S.cap ? sh_id_pkey(S.keys + (size_t)s * 8, S.nk[s]) : ~0ull; if (s >= S.cap || !S.nk[s]) st->fallback = 1u; }
        else pold = C.cell_key[rc];
        pnew = nk == 1 ? keys[0] : sh_id_pkey(keys, (uint32_t)nk);
    }
    for (int k = 0; k < 8; k++) mk[(size_t)i * 8 + k] = k < nk ? keys[k] : 0ull;
    mnk[i] = (uint8_t)(nk < 0 ? 0 : nk);
    const uint64_t ord = (1ull << 34) | ((uint64_t)((w >> 31) ? 0u : 1u) << 33) | ((uint64_t)R.id[r] << 1);   // (behind the batch's deletions) translation-only movers first, then ascending EntityId; remove before add
    op_key[2 * i] = pold; op_ord[2 * i] = ord; op_row[2 * i] = r | RB_REMOVE; op_idx[2 * i] = 2 * i;
    op_key[2 * i + 1] = pnew; op_ord[2 * i + 1] = ord | 1ull; op_row[2 * i + 1] = r; op_idx[2 * i + 1] = 2 * i + 1;
    // the second sort (unique placements + the link ops of the shared ones) leaves the shared placements' member ops behind every section
    op_key2[2 * i] = (pold & RB2_SHARED_BIT) ? ~0ull : pold;
    op_key2[2 * i + 1] = (pnew & RB2_SHARED_BIT) ? ~0ull : pnew;
}
__global__ __launch_bounds__(256) void k_rb2_ops(uint32_t m, const uint32_t *__restrict__ movers, RowArrays R, RbCells C, ShTable S, uint32_t outline, uint32_t atomic,
                                                 uint64_t *__restrict__ op_key, uint64_t *__restrict__ op_key2, uint64_t *__restrict__ op_ord, uint32_t *__restrict__ op_row,
                                                 uint32_t *__restrict__ op_idx, uint64_t *__restrict__ mk, uint8_t *__restrict__ mnk, uint32_t *__restrict__ host_list, Rb2Status *st) {
    rb2_ops_one(blockIdx.x * blockDim.x + threadIdx.x, m, movers, R, C, S, outline, atomic, op_key, op_key2, op_ord, op_row, op_idx, mk, mnk, host_list, st);
}

// ---- phase 2: the shared placements.  One thread per placement (segment head of the first sort) replays remove_entity / add_entity on the shared section's
// counts (re_api.hip: rebucket, `replay`); every creation / emptying on the way becomes link ops for the sections it links ------------------------------------
__device__ __forceinline__ void rb2_shared_segment_one(uint32_t t, uint32_t n, uint32_t m, const uint32_t *perm, const uint64_t *key_sorted, const uint32_t *op_row, uint64_t *op_ord, const uint64_t *mk,
                                                       const uint8_t *mnk, const ShTable &S, const RbCells &C, uint64_t *op_key2, uint32_t *op_row_w, uint32_t *op_idx, uint32_t link_cap, Rb2ShSeg *segs, Rb2Status *st) {
    if (t >= n) return;
    const uint64_t pkey = key_sorted[t];
    if (!(pkey & RB2_SHARED_BIT) || pkey == ~0ull) return;                 // unique placements: phase 3; ~0: the movers left to the host path
    if (t > 0 && key_sorted[t - 1] == pkey) return;                         // segment heads only
    uint32_t e = t + 1u; while (e < n && key_sorted[e] == pkey) e++;
    // the id of the placement, from its first op: the section a leaving row is still registered in, or the new keys of an arriving one
    Rb2ShSeg G{}; G.pkey = pkey; G.op_begin = t; G.op_count = e - t;
    {
        const uint32_t o = perm[t], w = op_row[o];
        if (w & RB_REMOVE) { const uint32_t s = C.row_cell[w & 0x7FFFFFFFu] & ~ROW_CELL_SHARED; G.nk = S.nk[s]; for (uint32_t k = 0; k < 8; k++) G.keys[k] = k < G.nk ? S.keys[(size_t)s * 8 + k] : 0ull; }
        else { const uint32_t i = o >> 1; G.nk = mnk[i]; for (uint32_t k = 0; k < 8; k++) G.keys[k] = mk[(size_t)i * 8 + k]; }
    }
    const int32_t idx0 = sh_lookup(S, pkey, G.keys, G.nk);
    if (idx0 == -2) { st->fallback = 1u; return; }
    G.idx = idx0; G.exists0 = idx0 >= 0;
    uint32_t na = G.exists0 ? S.nact[idx0] : 0u, nst = G.exists0 ? S.nstat[idx0] : 0u;
    bool exists = G.exists0, relink = false;
    for (uint32_t q = t; q < e; q++) {
        const uint32_t o = perm[q], w = op_row[o];
        const uint64_t ord = op_ord[o];
        int dir = 0;
        if (w & RB_REMOVE) {
            const uint32_t s = C.row_cell[w & 0x7FFFFFFFu] & ~ROW_CELL_SHARED;
            if (!sh_same_id(S.keys + (size_t)s * 8, S.nk[s], G.keys, G.nk)) { st->fallback = 1u; return; }      // two ids, one placement key
            if (na) na--;
            if (exists && na == 0 && nst == 0) { exists = false; dir = -1; }
        } else {
            const uint32_t i = o >> 1;
            if (!sh_same_id(mk + (size_t)i * 8, mnk[i], G.keys, G.nk)) { st->fallback = 1u; return; }
            if (!exists) { exists = true; na = 0; nst = 0; dir = 1; relink = true; }
            na++;
        }
        if (dir) {                                                          // the shared section appears / disappears: its linked sections gain / lose a link, at this point of the sequence
            const uint32_t at = atomicAdd(&st->n_link, G.nk);
            if (at + G.nk > link_cap) { st->fallback = 1u; return; }
            for (uint32_t k = 0; k < G.nk; k++) {
                const uint32_t j = 2u * m + at + k;
                op_key2[j] = G.keys[k]; op_ord[j] = ord; op_row_w[j] = dir > 0 ? RB2_LINK_INC : RB2_LINK_DEC; op_idx[j] = j;
            }
        }
    }
    G.na1 = na; G.nst = nst; G.exists1 = exists; G.relink = relink;
    segs[atomicAdd(&st->nseg_s, 1u)] = G;
    if (exists) {
        const uint32_t size = na + nst;
        if (idx0 < 0) atomicAdd(&st->need_sh, 1u);
        if (idx0 < 0 || size > S.rowcap[idx0]) { const uint32_t cap = size * 2u > 4u ? size * 2u : 4u; atomicAdd(&st->need_pool, cap); }
    }
}
__global__ __launch_bounds__(256) void k_rb2_shared_segments(uint32_t n, uint32_t m, const uint32_t *__restrict__ perm, const uint64_t *__restrict__ key_sorted, const uint32_t *__restrict__ op_row,
                                                             uint64_t *__restrict__ op_ord, const uint64_t *__restrict__ mk, const uint8_t *__restrict__ mnk, ShTable S, RbCells C,
                                                             uint64_t *__restrict__ op_key2, uint32_t *__restrict__ op_row_w, uint32_t *__restrict__ op_idx, uint32_t link_cap,
                                                             Rb2ShSeg *__restrict__ segs, Rb2Status *st) {
    rb2_shared_segment_one(blockIdx.x * blockDim.x + threadIdx.x, n, m, perm, key_sorted, op_row, op_ord, mk, mnk, S, C, op_key2, op_row_w, op_idx, link_cap, segs, st);
}

// ---- phase 3: the unique placements, member ops and link ops merged by the reference's order -----------------------------------------------------------
__device__ __forceinline__ void rb2_unique_segment_one(uint32_t t, uint32_t n, const uint32_t *perm, const uint64_t *key_sorted, const uint32_t *op_row, const RbTables &T, const RbCells &C,
                                                       const uint8_t *cell_links, Rb2Seg *segs, Rb2Status *st) {
    if (t >= n) return;
    const uint64_t key = key_sorted[t];
    if (key == ~0ull) return;
    if (t > 0 && key_sorted[t - 1] == key) return;                           // segment heads only
    uint32_t e = t + 1u; while (e < n && key_sorted[e] == key) e++;
    const int32_t slot = rb_find(T, C.cell_key, key);
    const bool exists0 = slot >= 0;
    uint32_t nl = exists0 ? C.cell_nl[slot] : 0u, ns = exists0 ? C.cell_ns[slot] : 0u, links = exists0 ? cell_links[slot] : 0u, total = 0;
    if (links >= 255u) st->fallback = 1u;                                    // (a saturated link count: the host path counts exactly)
    bool exists = exists0, changed = false;
    for (uint32_t q = t; q < e; q++) {
        const uint32_t w = op_row[perm[q]];
        if (w == RB2_LINK_INC) { if (!exists) { nl = 0; ns = 0; links = 0; exists = true; } links++; }
        else if (w == RB2_LINK_DEC) { if (links) links--; if (nl == 0 && ns == 0 && links == 0) exists = false; }
        else if (w & RB_REMOVE) {
            if (nl) nl--;
            if (nl == 0 && ns == 0 && links == 0) exists = false; else total += changed ? 1u : nl + ns;
            changed = true;
        } else {
            if (exists) { nl++; total += changed ? 1u : nl + ns; }
            else { nl = 1u; ns = 0u; links = 0u; exists = true; total += 1u; }
            changed = true;
        }
    }
    Rb2Seg S{}; S.key = key; S.slot = slot; S.op_begin = t; S.op_count = e - t; S.nl1 = nl; S.ns = ns; S.links1 = links; S.exists0 = exists0; S.exists1 = exists; S.changed = changed;
    segs[atomicAdd(&st->nseg_u, 1u)] = S;
    if (total) atomicAdd(&st->total, total);
    if (links > 254u) st->fallback = 1u;
    if (slot >= 0 && (changed || !exists) && C.cell_ng[slot] != 0u) st->fallback = 1u;     // ghost instances of the frozen render cache are parked in this section: the host path keeps their books
    if (exists) {
        const uint32_t size = nl + ns;
        if (slot < 0) atomicAdd(&st->need_slots[key_level(key) & (MAX_LEVELS - 1)], 1u);
        if (changed && (slot < 0 || size > C.cell_cap[slot])) { const uint32_t cap = size * 2u > 4u ? size * 2u : 4u; atomicAdd(&st->need_pool, cap); }
    }
}
__global__ __launch_bounds__(256) void k_rb2_unique_segments(uint32_t n, const uint32_t *__restrict__ perm, const uint64_t *__restrict__ key_sorted, const uint32_t *__restrict__ op_row,
                                                             RbTables T, RbCells C, uint8_t *__restrict__ cell_links, Rb2Seg *__restrict__ segs, Rb2Status *st) {
    rb2_unique_segment_one(blockIdx.x * blockDim.x + threadIdx.x, n, perm, key_sorted, op_row, T, C, cell_links, segs, st);
}

// members of one placement rewritten in place: the active rows that stay (compacted), the arrivals merged in ascending EntityId from the back, the static
// rows behind them; relocated to the end of the pool when the segment outgrew its capacity (the host checked the room).  Returns the new active count.
__device__ __forceinline__ uint32_t rb2_rewrite_members(const uint32_t *__restrict__ perm, const uint32_t *__restrict__ op_row, uint32_t op_begin, uint32_t op_count, const RbCells &C, const RowArrays &R,
                                                        uint32_t *begin_io, uint32_t *cap_io, uint32_t na0, uint32_t nst, uint32_t *__restrict__ tmp_row, Rb2Status *st, uint32_t place_word) {
    uint32_t begin = *begin_io, cap = *cap_io;
    uint32_t a = 0;                                                           // the rows that arrive, in ascending EntityId (insertion sort into this placement's share of the scratch array)
    for (uint32_t q = op_begin; q < op_begin + op_count; q++) {
        const uint32_t w = op_row[perm[q]];
        if ((w & RB_REMOVE) || w == RB2_LINK_INC || w == RB2_LINK_DEC) continue;
        const uint32_t id = R.id[w];
        uint32_t k = a;
        while (k > 0 && R.id[tmp_row[op_begin + k - 1u]] > id) { tmp_row[op_begin + k] = tmp_row[op_begin + k - 1u]; k--; }
        tmp_row[op_begin + k] = w; a++;
    }
    auto leaves = [&](uint32_t r) { for (uint32_t q = op_begin; q < op_begin + op_count; q++) if (op_row[perm[q]] == (r | RB_REMOVE)) return true; return false; };
    uint32_t kept = 0; for (uint32_t i = 0; i < na0; i++) if (!leaves(C.rows[begin + i])) kept++;
    const uint32_t na1 = kept + a, size = na1 + nst;
    if (size > cap) {
        const uint32_t ncap = size * 2u > 4u ? size * 2u : 4u, nb = atomicAdd(&st->pool_used, ncap);
        if ((uint64_t)nb + ncap > C.pool_cap) { st->err = 1u; return na0; }      // (cannot happen: the host checked need_pool)
        for (uint32_t i = 0; i < na0 + nst; i++) { C.rows[nb + i] = C.rows[begin + i]; C.rows_gc[nb + i] = C.rows_gc[begin + i]; }
        begin = nb; cap = ncap;
    }
    uint32_t mm = 0;
    for (uint32_t i = 0; i < na0; i++) { const uint32_t r = C.rows[begin + i]; if (!leaves(r)) { C.rows[begin + mm] = r; C.rows_gc[begin + mm] = C.rows_gc[begin + i]; mm++; } }
    if (na1 > na0) for (uint32_t i = nst; i-- > 0;) { C.rows[begin + na1 + i] = C.rows[begin + na0 + i]; C.rows_gc[begin + na1 + i] = C.rows_gc[begin + na0 + i]; }
    else if (na1 < na0) for (uint32_t i = 0; i < nst; i++) { C.rows[begin + na1 + i] = C.rows[begin + na0 + i]; C.rows_gc[begin + na1 + i] = C.rows_gc[begin + na0 + i]; }
    for (int32_t i = (int32_t)mm - 1, j = (int32_t)a - 1, k = (int32_t)na1 - 1; j >= 0; k--) {
        if (i >= 0 && R.id[C.rows[begin + i]] > R.id[tmp_row[op_begin + j]]) { C.rows[begin + k] = C.rows[begin + i]; C.rows_gc[begin + k] = C.rows_gc[begin + i]; i--; }
        else { const uint32_t r = tmp_row[op_begin + j]; C.rows[begin + k] = r; C.rows_gc[begin + k] = R.gclass[r]; C.row_cell[r] = place_word; j--; }
    }
    *begin_io = begin; *cap_io = cap;
    return na1;
}

// The same rewrite by a whole WAVE (round 3, second step): the serial version above walks the ops and the members with dependent loads -- two per op, one per member, the ops again for
// every member ("does it leave?"), the ids again for every comparison of the merge -- about a hundred round trips for a section with five members and two ops, 145 us for the apply
// phase of a batch however small.  Here lane l holds op l AND member l (both at most 64; larger placements fall back to the serial version in lane 0): every load is one round
// trip of the wave, "leaves", the ranks among the arrivals and the merge positions are loops over lanes (shuffles, no memory), and every lane stores its element where it ends up:
//   position of a kept active row  = its index among the kept rows + the arrivals with a smaller EntityId
//   position of an arriving row    = its rank among the arrivals + the kept rows with a smaller EntityId          (ids are unique; the kept rows are in ascending id)
//   position of a static row       = new active count + its index among the static rows
struct Rb2Rewrite { uint32_t begin, cap, na1, last_row; };      // last_row: the row that ends last in the segment (active rows, then static rows), ~0 when it is empty
__device__ __forceinline__ Rb2Rewrite rb2_rewrite_members_wave(const uint32_t *__restrict__ perm, const uint32_t *__restrict__ op_row, uint32_t op_begin, uint32_t op_count, const RbCells &C, const RowArrays &R,
                                                               uint32_t begin, uint32_t cap, uint32_t na0, uint32_t nst, uint32_t *__restrict__ tmp_row, Rb2Status *st, uint32_t place_word,
                                                               bool set_key, uint64_t key) {
    const uint32_t lane = threadIdx.x & 63u;
    Rb2Rewrite out; out.begin = begin; out.cap = cap; out.na1 = na0; out.last_row = 0xFFFFFFFFu;
    if (op_count > 64u || na0 + nst > 64u) {                                 // wave-uniform: a crowded placement, the serial version in lane 0
        uint32_t b2 = begin, c2 = cap, n1 = 0, last = 0xFFFFFFFFu;
        if (lane == 0) {
            n1 = rb2_rewrite_members(perm, op_row, op_begin, op_count, C, R, &b2, &c2, na0, nst, tmp_row, st, place_word);
            if (set_key) for (uint32_t q = op_begin; q < op_begin + op_count; q++) { const uint32_t w = op_row[perm[q]]; if (!(w & RB_REMOVE) && w != RB2_LINK_INC && w != RB2_LINK_DEC) C.row_key[w] = key; }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (n1 + nst) last = __hip_atomic_load(&C.rows[b2 + n1 + nst - 1u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // (this lane's own store a moment ago: past the L1)
        }
        out.begin = __shfl(b2, 0, 64); out.cap = __shfl(c2, 0, 64); out.na1 = __shfl(n1, 0, 64); out.last_row = __shfl(last, 0, 64);
        return out;
    }
    // ---- one round trip each: the ops, the members, then the ids
    const bool has_op = lane < op_count, has_mem = lane < na0 + nst;
    const uint32_t w = has_op ? op_row[perm[op_begin + lane]] : 0xFFFFFFFFu;
    const bool arrives = has_op && !(w & RB_REMOVE) && w != RB2_LINK_INC && w != RB2_LINK_DEC;
    const uint32_t mrow = has_mem ? C.rows[begin + lane] : 0xFFFFFFFFu, mgc = has_mem ? C.rows_gc[begin + lane] : 0u;
    const bool active = lane < na0, is_static = has_mem && !active;
    const uint32_t aid = arrives ? R.id[w] : 0u, agc = arrives ? R.gclass[w] : 0u;
    const uint32_t mid = active ? R.id[mrow] : 0u;
    // ---- which active rows leave (an op "remove this row"); loops over lanes
    bool leaves = false;
    for (uint32_t q = 0; q < op_count; q++) { const uint32_t wq = __shfl(w, (int)q, 64); if (active && wq == (mrow | RB_REMOVE)) leaves = true; }
    const bool kept = active && !leaves;
    const uint64_t kmask = __ballot(kept), amask = __ballot(arrives);
    const uint32_t nkept = (uint32_t)__popcll(kmask), narr = (uint32_t)__popcll(amask), na1 = nkept + narr, size = na1 + nst;
    uint32_t pos_m = 0xFFFFFFFFu, pos_a = 0xFFFFFFFFu;
    if (kept) pos_m = (uint32_t)__popcll(kmask & ((1ull << lane) - 1ull));
    if (is_static) pos_m = na1 + (lane - na0);
    uint32_t a_rank = 0, a_before_m = 0;
    for (uint32_t q = 0; q < op_count; q++) {
        const uint32_t idq = __shfl(aid, (int)q, 64); const bool aq = (amask >> q) & 1ull;
        if (aq && arrives && idq < aid) a_rank++;
        if (aq && kept && idq < mid) a_before_m++;
    }
    uint32_t k_before_a = 0;
    for (uint32_t m2 = 0; m2 < na0; m2++) { const uint32_t idm = __shfl(mid, (int)m2, 64); if (((kmask >> m2) & 1ull) && arrives && idm < aid) k_before_a++; }
    if (kept) pos_m += a_before_m;
    if (arrives) pos_a = a_rank + k_before_a;
    // ---- room: relocated to the end of the pool when the segment outgrew its capacity (the host checked the total)
    uint32_t nb = begin, ncap = cap;
    if (size > cap) {
        ncap = size * 2u > 4u ? size * 2u : 4u;
        if (lane == 0) nb = atomicAdd(&st->pool_used, ncap);
        nb = __shfl(nb, 0, 64);
        if ((uint64_t)nb + ncap > C.pool_cap) { if (lane == 0) st->err = 1u; return out; }      // (cannot happen)
    }
    // ---- every element to its place (all of them were read above: the old and the new segment may overlap)
    if (kept || is_static) { C.rows[nb + pos_m] = mrow; C.rows_gc[nb + pos_m] = mgc; }
    if (arrives) { C.rows[nb + pos_a] = w; C.rows_gc[nb + pos_a] = agc; C.row_cell[w] = place_word; if (set_key) C.row_key[w] = key; }
    // the row that ends last
    uint32_t last = 0xFFFFFFFFu;
    if (size) {
        const uint64_t lm = __ballot((kept || is_static) && pos_m == size - 1u), la = __ballot(arrives && pos_a == size - 1u);
        if (lm) last = __shfl(mrow, __ffsll((long long)lm) - 1, 64); else if (la) last = __shfl(w, __ffsll((long long)la) - 1, 64);
    }
    out.begin = nb; out.cap = ncap; out.na1 = na1; out.last_row = last;
    return out;
}

// ---- phase 4a: the unique sections (one WAVE per section) -------------------------------------------------------------------------------------------------
// (as with phases 1-3: the body is a function of one segment and one wave -- a kernel with a workgroup per segment for batches of any size, a step of k_rb2_apply_small for small ones)
__device__ __forceinline__ void rb2_apply_unique_one(uint32_t s, uint32_t lane, const uint32_t *perm, const uint32_t *op_row, const RbTables &T, const RbCells &C, const RowArrays &R, uint8_t *cell_links,
                                                     Rb2Seg *segs, Rb2Status *st, const uint32_t *free_slots, const uint32_t *free_off, uint32_t *tmp_row, uint32_t *refold) {
    Rb2Seg S = segs[s];
    const uint32_t lv = key_level(S.key) & (MAX_LEVELS - 1);
    if (!S.exists1) {
        if (lane == 0) {
            refold[s] = 0xFFFFFFFFu;
            if (S.slot >= 0) {                                                // no member and no link left: a padding slot from now on
                const uint32_t sl = (uint32_t)S.slot;
                C.cell_key[sl] = pack_key(lv, 0xFFFFu, 0xFFFFu, 0xFFFFu); C.cell_key32[sl] = KEY32_PAD | 0x1FF7FDFFu;
                C.cell_nl[sl] = 0; C.cell_ns[sl] = 0; C.cell_ng[sl] = 0; C.cell_flags[sl] = (uint8_t)(CF_PAD | CF_STATIC_SECTION); cell_links[sl] = 0;
                S.freed = 1; atomicAdd(&st->n_freed, 1u); segs[s] = S;
            }
        }
        return;
    }
    uint32_t sl = 0; const bool create = S.slot < 0;
    if (create) {
        if (lane == 0) {
            sl = free_slots[free_off[lv] + atomicAdd(&st->popped[lv], 1u)];
            atomicAdd(&st->n_created, 1u);
            C.cell_key[sl] = S.key; C.cell_key32[sl] = (key_x(S.key) << 20) | (key_z(S.key) << 10) | key_y(S.key);
            C.cell_stamp[sl] = 0; C.cell_cap[sl] = 0; C.cell_begin[sl] = 0; C.cell_nl[sl] = 0; C.cell_ns[sl] = 0; C.cell_ng[sl] = 0; C.cell_flags[sl] = 0;
            rb_ovl_put(T, S.key, sl);
        }
        sl = __shfl(sl, 0, 64); S.slot = (int32_t)sl; S.created = 1;
    } else sl = (uint32_t)S.slot;
    if (lane == 0) cell_links[sl] = (uint8_t)(S.links1 > 255u ? 255u : S.links1);
    if (S.changed) {
        const uint32_t begin = create ? 0u : C.cell_begin[sl], cap = create ? 0u : C.cell_cap[sl], nl0 = create ? 0u : C.cell_nl[sl], ns0 = create ? 0u : C.cell_ns[sl];
        const Rb2Rewrite r = rb2_rewrite_members_wave(perm, op_row, S.op_begin, S.op_count, C, R, begin, cap, nl0, ns0, tmp_row, st, sl, true, S.key);
        if (lane == 0) { C.cell_begin[sl] = r.begin; C.cell_cap[sl] = r.cap; C.cell_nl[sl] = r.na1; }
        S.nl1 = r.na1;
    }
    if (lane == 0) { segs[s] = S; refold[s] = (S.changed || S.created) ? sl : 0xFFFFFFFFu; }      // end_of_changes / update_static_world_sections touch the changed (and the new) sections
}
__global__ __launch_bounds__(64) void k_rb2_apply_unique(const uint32_t *__restrict__ perm, const uint32_t *__restrict__ op_row, RbTables T, RbCells C, RowArrays R, uint8_t *__restrict__ cell_links,
                                                         Rb2Seg *__restrict__ segs, Rb2Status *st, const uint32_t *__restrict__ free_slots, const uint32_t *__restrict__ free_off,
                                                         uint32_t *__restrict__ tmp_row, uint32_t *__restrict__ refold) {
    if (blockIdx.x >= st->nseg_u) return;
    rb2_apply_unique_one(blockIdx.x, threadIdx.x, perm, op_row, T, C, R, cell_links, segs, st, free_slots, free_off, tmp_row, refold);
}

// ---- phase 4b: the shared sections (stable indices: a retired entry is a hole, a new one takes a free index the host handed over); one WAVE per section ---------
__device__ __forceinline__ void rb2_apply_shared_one(uint32_t s, uint32_t lane, const uint32_t *perm, const uint32_t *op_row, const RbTables &T, const RbCells &C, const RowArrays &R, const ShTable &S,
                                                     Rb2ShSeg *segs, Rb2Status *st, const uint32_t *free_sh, uint32_t *tmp_row) {
    Rb2ShSeg G = segs[s];
    if (!G.exists1) {
        if (G.idx >= 0 && lane == 0) {
            const uint32_t idx = (uint32_t)G.idx;
            for (uint32_t h = rb_hash(G.pkey) & S.hmask;; h = (h + 1u) & S.hmask) { const unsigned long long k = S.hkeys[h]; if (k == ~0ull) break; if (k == G.pkey) { S.hidx[h] = 0xFFFFFFFFu; break; } }
            S.nact[idx] = 0; S.nstat[idx] = 0; S.nk[idx] = 0; S.owner[idx] = -1; S.cached[idx] = 0; S.dirty[idx] = 0;
            for (uint32_t k = 0; k < 8; k++) { S.cells[(size_t)idx * 8 + k] = -1; S.keys[(size_t)idx * 8 + k] = 0ull; }
            G.freed = 1; atomicAdd(&st->n_sh_freed, 1u); segs[s] = G;
        }
        return;
    }
    uint32_t idx = 0; const bool create = G.idx < 0;
    if (create) {
        if (lane == 0) {
            idx = free_sh[atomicAdd(&st->popped_sh, 1u)];
            atomicAdd(&st->n_sh_created, 1u);
            S.nk[idx] = (uint8_t)G.nk; for (uint32_t k = 0; k < 8; k++) S.keys[(size_t)idx * 8 + k] = G.keys[k];
            S.begin[idx] = 0; S.rowcap[idx] = 0; S.nact[idx] = 0; S.nstat[idx] = 0; S.owner[idx] = -1; S.cached[idx] = 0; S.dirty[idx] = 0;
            sh_hash_put(S, G.pkey, idx);
        }
        idx = __shfl(idx, 0, 64); G.idx = (int32_t)idx; G.created = 1;
    } else idx = (uint32_t)G.idx;
    if (G.relink || G.created) {                                             // (a section emptied and refilled within the batch may find a linked section in another slot); lane k looks key k up
        if (lane < 8u) { int32_t c = -1; if (lane < G.nk) { c = rb_find(T, C.cell_key, G.keys[lane]); if (c < 0) st->err = 2u; } S.cells[(size_t)idx * 8 + lane] = c; }
    }
    const uint32_t begin = create ? 0u : S.begin[idx], cap = create ? 0u : S.rowcap[idx], na0 = create ? 0u : S.nact[idx], nst = create ? 0u : S.nstat[idx];
    const Rb2Rewrite r = rb2_rewrite_members_wave(perm, op_row, G.op_begin, G.op_count, C, R, begin, cap, na0, nst, tmp_row, st, ROW_CELL_SHARED | idx, false, 0ull);
    if (lane == 0) {
        S.begin[idx] = r.begin; S.rowcap[idx] = r.cap; S.nact[idx] = r.na1;
        Aabb u = { 0.f, 0.f, 0.f, 0.f, 0.f, 0.f };                               // end_of_changes, shared branch (:1104-1125): the AABB of the last entity iterated (entities, then static_entities)
        if (r.last_row != 0xFFFFFFFFu) u = R.aabb[r.last_row];
        S.aabb[idx] = u;
        G.na1 = r.na1; segs[s] = G;
    }
}
__global__ __launch_bounds__(64) void k_rb2_apply_shared(const uint32_t *__restrict__ perm, const uint32_t *__restrict__ op_row, RbTables T, RbCells C, RowArrays R, ShTable S,
                                                         Rb2ShSeg *__restrict__ segs, Rb2Status *st, const uint32_t *__restrict__ free_sh, uint32_t *__restrict__ tmp_row) {
    if (blockIdx.x >= st->nseg_s) return;
    rb2_apply_shared_one(blockIdx.x, threadIdx.x, perm, op_row, T, C, R, S, segs, st, free_sh, tmp_row);
}

// ---- phase 4c: update_static_world_sections (bounding_box_tree_v2.rs:1133-1213) --------------------------------------------------------------------------
// which sections a shared section WITHOUT active entities links (first loop: an empty section is a static section when nothing links it, or when one of
// the shared sections linking it has no active entity)
// (value 1 in front of the first loop, value 0 behind it: the array is all zero between batches -- clearing its ncells bytes per batch was a 10 MB memset in a 10 M-section world)
__global__ __launch_bounds__(256) void k_rb2_mark_inactive(uint32_t nsh, ShTable S, uint8_t *__restrict__ cell_inact, uint8_t value) {
    const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= nsh || !S.nk[s] || S.nact[s] != 0u) return;
    for (uint32_t k = 0; k < 8; k++) { const int32_t c = S.cells[(size_t)s * 8 + k]; if (c >= 0) cell_inact[c] = value; }
}
__global__ __launch_bounds__(256) void k_rb2_static_first(RbCells C, const uint8_t *__restrict__ cell_links, const uint8_t *__restrict__ cell_inact, const Rb2Seg *__restrict__ segs, const Rb2Status *st) {
    const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= st->nseg_u) return;
    const Rb2Seg S = segs[s];
    if (!S.exists1 || S.slot < 0 || !(S.changed || S.created)) return;
    const uint32_t sl = (uint32_t)S.slot;
    const bool stat = C.cell_nl[sl] == 0u && (cell_links[sl] == 0u || cell_inact[sl] != 0u);
    C.cell_flags[sl] = (uint8_t)((C.cell_flags[sl] & ~CF_STATIC_SECTION) | (stat ? CF_STATIC_SECTION : 0));
}
// second loop: the changed shared sections in canonical id order; for each linked section: no active entity -> a static section if it holds no active
// entity itself, else not a static section.  (slot, shared segment) pairs sorted by slot; one thread folds a slot's rules in that order: with active
// entities of its own only "not static" can apply (any rule with active entities), without them the LAST rule decides.
__global__ __launch_bounds__(256) void k_rb2_static_pairs(const Rb2ShSeg *__restrict__ segs, const Rb2Status *st, ShTable S, uint64_t *__restrict__ pair_key, uint32_t *__restrict__ pair_seg, Rb2Status *stw) {
    const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= st->nseg_s) return;
    const Rb2ShSeg G = segs[s];
    if (!G.exists1 || G.idx < 0) return;
    const uint32_t at = atomicAdd(&stw->n_pairs, G.nk);
    for (uint32_t k = 0; k < G.nk; k++) { pair_key[at + k] = (uint64_t)(uint32_t)S.cells[(size_t)G.idx * 8 + k]; pair_seg[at + k] = s; }
}
__global__ __launch_bounds__(256) void k_rb2_static_second(uint32_t n, const uint64_t *__restrict__ slot_sorted, const uint32_t *__restrict__ seg_sorted, const Rb2ShSeg *__restrict__ segs, RbCells C) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const uint64_t sl64 = slot_sorted[t];
    if (t > 0 && slot_sorted[t - 1] == sl64) return;
    if (sl64 >= 0x80000000ull) return;                                       // (a link that could not be resolved: reported by the apply kernel)
    const uint32_t sl = (uint32_t)sl64;
    bool any_active = false; uint32_t last = seg_sorted[t];
    for (uint32_t q = t; q < n && slot_sorted[q] == sl64; q++) {
        const Rb2ShSeg &G = segs[seg_sorted[q]];
        if (G.na1 != 0u) any_active = true;
        const Rb2ShSeg &L = segs[last];
        if (sh_id_less(L.keys, L.nk, G.keys, G.nk)) last = seg_sorted[q];
    }
    uint8_t f = C.cell_flags[sl];
    if (C.cell_nl[sl] != 0u) { if (any_active) f &= (uint8_t)~CF_STATIC_SECTION; }
    else f = (uint8_t)((f & ~CF_STATIC_SECTION) | (segs[last].na1 == 0u ? CF_STATIC_SECTION : 0));
    C.cell_flags[sl] = f;
}
// small batches: both loops in ONE workgroup (mark, first loop, unmark, the pairs, their sort in LDS, second loop) -- eight launches and a read-back of the pair count less.
// The arrays are read and written through plain pointers (no __restrict__): later steps read what earlier steps of this kernel wrote, ordered by the workgroup barriers.
__device__ __forceinline__ void rb2_static_block(uint32_t nsh, const ShTable &S, const RbCells &C, const uint8_t *cell_links, uint8_t *cell_inact, const Rb2Seg *segs_u, const Rb2ShSeg *segs_s, uint32_t nu, uint32_t ns,
                                                 uint64_t *s_key, uint32_t &s_np) {      // a workgroup of 1,024; s_key: RB2_STATIC_SMALL_PAIRS words of LDS
    const uint32_t tid = threadIdx.x;
    if (tid == 0) s_np = 0u;
    if (nu) {
        for (uint32_t v = 0; v < 2u; v++) {                                  // v = 0: mark, then the first loop; v = 1: unmark
            for (uint32_t s = tid; s < nsh; s += 1024u) {
                if (!S.nk[s] || S.nact[s] != 0u) continue;
                for (uint32_t k = 0; k < 8; k++) { const int32_t c = S.cells[(size_t)s * 8 + k]; if (c >= 0) cell_inact[c] = (uint8_t)(v ^ 1u); }
            }
            __syncthreads();
            if (v == 0u) {
                for (uint32_t s = tid; s < nu; s += 1024u) {
                    const Rb2Seg G = segs_u[s];
                    if (!G.exists1 || G.slot < 0 || !(G.changed || G.created)) continue;
                    const uint32_t sl = (uint32_t)G.slot;
                    const bool stat = C.cell_nl[sl] == 0u && (cell_links[sl] == 0u || cell_inact[sl] != 0u);
                    C.cell_flags[sl] = (uint8_t)((C.cell_flags[sl] & ~CF_STATIC_SECTION) | (stat ? CF_STATIC_SECTION : 0));
                }
                __syncthreads();
            }
        }
    }
    __syncthreads();
    if (!ns) return;
    // (slot, shared segment) pairs: slot in the high word, so equal slots are neighbours after the sort (a segment links a slot at most once; which of a slot's pairs comes first does not matter to the fold below)
    for (uint32_t s = tid; s < ns; s += 1024u) {
        const Rb2ShSeg &G = segs_s[s];
        if (!G.exists1 || G.idx < 0) continue;
        const uint32_t at = atomicAdd(&s_np, (uint32_t)G.nk);
        for (uint32_t k = 0; k < G.nk && at + k < RB2_STATIC_SMALL_PAIRS; k++) s_key[at + k] = ((uint64_t)(uint32_t)S.cells[(size_t)G.idx * 8 + k] << 32) | s;
    }
    __syncthreads();
    const uint32_t np = min(s_np, RB2_STATIC_SMALL_PAIRS);                   // (the host chose this kernel for 8 * nseg_s <= RB2_STATIC_SMALL_PAIRS)
    uint32_t m = 1; while (m < np) m <<= 1;
    for (uint32_t i = np + tid; i < m; i += 1024u) s_key[i] = ~0ull;
    __syncthreads();
    for (uint32_t k = 2; k <= m; k <<= 1)
        for (uint32_t j = k >> 1; j > 0; j >>= 1) {
            for (uint32_t i = tid; i < m; i += 1024u) {
                const uint32_t l = i ^ j;
                if (l > i) { const uint64_t a = s_key[i], b = s_key[l]; if ((a > b) == ((i & k) == 0)) { s_key[i] = b; s_key[l] = a; } }
            }
            __syncthreads();
        }
    for (uint32_t t = tid; t < np; t += 1024u) {
        const uint32_t sl = (uint32_t)(s_key[t] >> 32);
        if (t > 0 && (uint32_t)(s_key[t - 1] >> 32) == sl) continue;
        if (sl >= 0x80000000u) continue;                                     // (a link that could not be resolved: reported by the apply kernel)
        bool any_active = false; uint32_t last = (uint32_t)s_key[t];
        for (uint32_t q = t; q < np && (uint32_t)(s_key[q] >> 32) == sl; q++) {
            const uint32_t g = (uint32_t)s_key[q];
            const Rb2ShSeg &G = segs_s[g];
            if (G.na1 != 0u) any_active = true;
            const Rb2ShSeg &L = segs_s[last];
            if (sh_id_less(L.keys, L.nk, G.keys, G.nk)) last = g;
        }
        uint8_t f = C.cell_flags[sl];
        if (C.cell_nl[sl] != 0u) { if (any_active) f &= (uint8_t)~CF_STATIC_SECTION; }
        else f = (uint8_t)((f & ~CF_STATIC_SECTION) | (segs_s[last].na1 == 0u ? CF_STATIC_SECTION : 0));
        C.cell_flags[sl] = f;
    }
}
__global__ __launch_bounds__(1024) void k_rb2_static_small(uint32_t nsh, ShTable S, RbCells C, const uint8_t *cell_links, uint8_t *cell_inact, const Rb2Seg *segs_u, const Rb2ShSeg *segs_s, const Rb2Status *st) {
    __shared__ uint64_t s_key[RB2_STATIC_SMALL_PAIRS]; __shared__ uint32_t s_np;
    rb2_static_block(nsh, S, C, cell_links, cell_inact, segs_u, segs_s, st->nseg_u, st->nseg_s, s_key, s_np);
}

__global__ __launch_bounds__(256) void k_rb2_gather_u32(uint32_t n, const uint32_t *__restrict__ perm, const uint32_t *__restrict__ src, uint32_t *__restrict__ dst) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[perm[i]];
}
// the status block for the host: copied into mapped pinned memory and announced by a sequence word the host polls (publish_to_host) -- a stream-ordered copy plus a stream
// synchronise cost ~15 us per read-back, three times a batch
// ... and, for the last read-back of a small batch, the segment lists with it (words_a / words_b 32-bit words) and the block reset for the next batch (everything but pool_used, which the
// device keeps current)
__device__ __forceinline__ void rb2_publish_block(Rb2Status *st, Rb2Status *h_st, uint32_t *h_seq, uint32_t seq, const uint32_t *src_a, uint32_t *dst_a, uint32_t words_a,
                                                  const uint32_t *src_b, uint32_t *dst_b, uint32_t words_b, uint32_t reset) {      // the whole workgroup
    static_assert(sizeof(Rb2Status) % 4u == 0 && sizeof(Rb2Status) / 4u <= 256u, "one word per thread");
    uint32_t *src = reinterpret_cast<uint32_t *>(st); uint32_t *dst = reinterpret_cast<uint32_t *>(h_st);
    if (threadIdx.x < sizeof(Rb2Status) / 4u) {
        dst[threadIdx.x] = __hip_atomic_load(src + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (reset && threadIdx.x != offsetof(Rb2Status, pool_used) / 4u) src[threadIdx.x] = 0u;
    }
    for (uint32_t i = threadIdx.x; i < words_a; i += blockDim.x) dst_a[i] = __hip_atomic_load(src_a + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    for (uint32_t i = threadIdx.x; i < words_b; i += blockDim.x) dst_b[i] = __hip_atomic_load(src_b + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    wait_own_stores();
    __syncthreads();                                                          // every wave's host stores have left before thread 0 publishes
    if (threadIdx.x == 0) publish_to_host(h_seq, seq);
}
__global__ __launch_bounds__(256) void k_rb2_publish_status(Rb2Status *st, Rb2Status *h_st, uint32_t *h_seq, uint32_t seq, const uint32_t *src_a, uint32_t *dst_a, uint32_t words_a,
                                                            const uint32_t *src_b, uint32_t *dst_b, uint32_t words_b, uint32_t reset) {
    rb2_publish_block(st, h_st, h_seq, seq, src_a, dst_a, words_a, src_b, dst_b, words_b, reset);
}

// ---- small batches (<= RB2_PLAN_SMALL movers: their ops fit the one-workgroup sort, and so do, nearly always, the link ops they emit): phases 1-3 and the status read-back as ONE launch of
// ONE workgroup.  A batch that carries the user entity across a section border took five launches and a read-back to get here; each dependent launch costs 5-7 us, whatever it does.
// Between the steps: a workgroup barrier with its workgroup-scope fence -- the waves of a workgroup share one L1, which is write-through, and atomics are performed in the L2 behind it,
// so nothing more is needed for one step to see what the step before wrote.  (An AGENT-scope fence here writes back and invalidates the XCD's L2: measured, it made this kernel slower
// than the five launches it replaces.)
__device__ __forceinline__ void rb2_phase_barrier() { __syncthreads(); }
__global__ __launch_bounds__(1024) void k_rb2_plan_small(uint32_t m, const uint32_t *movers, RowArrays R, RbCells C, ShTable S, RbTables T, uint32_t outline, uint32_t atomic,
                                                         uint64_t *op_key, uint64_t *op_key2, uint64_t *op_ord, uint32_t *op_row, uint32_t *op_idx, uint64_t *mk, uint8_t *mnk, uint32_t *host_list,
                                                         uint64_t *ksorted1, uint32_t *perm1, uint64_t *ksorted2, uint32_t *perm2, uint32_t link_cap, const uint8_t *cell_links,
                                                         Rb2ShSeg *segs_s, Rb2Seg *segs_u, Rb2Status *st, Rb2Status *h_st, uint32_t *h_seq, uint32_t seq, uint32_t *h_segs_u) {
    __shared__ uint64_t s_key[RB2_SORT_SMALL], s_ord[RB2_SORT_SMALL]; __shared__ uint32_t s_idx[RB2_SORT_SMALL];
    const uint32_t tid = threadIdx.x, n1 = 2u * m;
    for (uint32_t i = tid; i < m; i += 1024u) rb2_ops_one(i, m, movers, R, C, S, outline, atomic, op_key, op_key2, op_ord, op_row, op_idx, mk, mnk, host_list, st);
    rb2_phase_barrier();
    rb2_sort_block(n1, op_key, op_ord, ksorted1, perm1, s_key, s_ord, s_idx);
    rb2_phase_barrier();
    for (uint32_t t = tid; t < n1; t += 1024u) rb2_shared_segment_one(t, n1, m, perm1, ksorted1, op_row, op_ord, mk, mnk, S, C, op_key2, op_row, op_idx, link_cap, segs_s, st);
    rb2_phase_barrier();
    const uint32_t n2 = min(n1 + __hip_atomic_load(&st->n_link, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), RB2_SORT_SMALL);      // (more link ops than fit: phase 3 below plans on a truncated list, nothing is touched, and the host -- n1 + n_link > RB2_SORT_SMALL in the status block -- plans again)
    rb2_sort_block(n2, op_key2, op_ord, ksorted2, perm2, s_key, s_ord, s_idx);
    rb2_phase_barrier();
    for (uint32_t t = tid; t < n2; t += 1024u) rb2_unique_segment_one(t, n2, perm2, ksorted2, op_row, T, C, cell_links, segs_u, st);
    rb2_phase_barrier();
    const uint32_t nu = __hip_atomic_load(&st->nseg_u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // (the planned unique segments with the status: the host checks the sections they create or retire against its ghost books)
    rb2_publish_block(st, h_st, h_seq, seq, reinterpret_cast<const uint32_t *>(segs_u), h_segs_u, nu * (uint32_t)(sizeof(Rb2Seg) / 4u), nullptr, nullptr, 0u, 0u);
}
// ... and phase 4 of such a batch: the unique sections (a wave each, 16 at a time), the shared sections, update_static_world_sections, the tight AABBs of end_of_changes, the rows the
// batch deleted, then the status block and both segment lists for the host's bookkeeping, and the status block reset for the next batch -- six to eleven launches as one
__global__ __launch_bounds__(1024) void k_rb2_apply_small(uint32_t m, const uint32_t *movers, uint32_t has_deleted, const uint32_t *perm1, const uint32_t *perm2, const uint32_t *op_row, RbTables T, RbCells C, RowArrays R,
                                                          ShTable S, uint8_t *cell_links, uint8_t *cell_inact, Aabb *cell_tight, Rb2Seg *segs_u, Rb2ShSeg *segs_s, Rb2Status *st, const uint32_t *free_u,
                                                          const uint32_t *free_off, const uint32_t *free_s, uint32_t *tmp_u, uint32_t *tmp_s, uint32_t *refold, uint32_t nsh, uint32_t atomic, uint32_t too_many, uint32_t segments_done,
                                                          Rb2Status *h_st, uint32_t *h_seq, uint32_t seq, uint32_t *h_segs_u, uint32_t *h_segs_s) {
    __shared__ uint64_t s_key[RB2_STATIC_SMALL_PAIRS]; __shared__ uint32_t s_np;
    const uint32_t tid = threadIdx.x, wave = tid >> 6, lane = tid & 63u;
    const uint32_t nu = st->nseg_u, ns = st->nseg_s;                          // (the plan of k_rb2_plan_small: not written here)
    if (!segments_done) {                                                     // (more than a few segments: k_rb2_apply_unique / k_rb2_apply_shared ran in front of this kernel, a workgroup per segment -- 16 waves at a time are 9 rounds of dependent loads for 143 segments)
        for (uint32_t s = wave; s < nu; s += 16u) rb2_apply_unique_one(s, lane, perm2, op_row, T, C, R, cell_links, segs_u, st, free_u, free_off, tmp_u, refold);
        rb2_phase_barrier();
        for (uint32_t s = wave; s < ns; s += 16u) rb2_apply_shared_one(s, lane, perm1, op_row, T, C, R, S, segs_s, st, free_s, tmp_s);
        rb2_phase_barrier();
    }
    rb2_static_block(nsh, S, C, cell_links, cell_inact, segs_u, segs_s, nu, ns, s_key, s_np);
    rb2_phase_barrier();
    for (uint32_t i = tid; i < nu; i += 1024u) {                              // end_of_changes (k_fold_tight_list)
        const uint32_t c = refold[i];
        if (c == 0xFFFFFFFFu) continue;
        const uint64_t key = C.cell_key[c];
        const uint32_t n = C.cell_nl[c] + C.cell_ns[c];
        uint32_t adj = 20u + key_level(key) * 5u; if (adj > 50u) adj = 50u;
        Aabb u = { 0.f, 0.f, 0.f, 0.f, 0.f, 0.f };
        if (too_many && n > adj) u = key_to_aabb(key, atomic);
        else { const uint32_t b = C.cell_begin[c]; for (uint32_t k = 0; k < n; k++) { const Aabb e = R.aabb[C.rows[b + k]]; u = (k == 0) ? e : combine_aabb(u, e); } }
        cell_tight[c] = u;
    }
    if (has_deleted) for (uint32_t i = tid; i < m; i += 1024u) if (movers[i] & RB2_MOVER_DELETED) C.row_cell[movers[i] & 0x1FFFFFFFu] = ROW_CELL_NONE;
    rb2_phase_barrier();
    rb2_publish_block(st, h_st, h_seq, seq, reinterpret_cast<const uint32_t *>(segs_u), h_segs_u, nu * (uint32_t)(sizeof(Rb2Seg) / 4u),
                      reinterpret_cast<const uint32_t *>(segs_s), h_segs_s, ns * (uint32_t)(sizeof(Rb2ShSeg) / 4u), 1u);
}
__global__ __launch_bounds__(256) void k_rb2_clear_deleted(uint32_t n, const uint32_t *__restrict__ movers, uint32_t *__restrict__ row_cell) {      // the deleted rows are in no section any more
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && (movers[i] & RB2_MOVER_DELETED)) row_cell[movers[i] & 0x1FFFFFFFu] = ROW_CELL_NONE;
}
// ---- host mirrors on demand: the state of the shared entries the device changed (re_api.hip: sync_mirrors) ----------------------------------------------------
__global__ __launch_bounds__(256) void k_rb2_gather_shared(uint32_t n, const uint32_t *__restrict__ idxs, ShTable S, uint64_t *__restrict__ out_keys, uint32_t *__restrict__ out_hdr, int32_t *__restrict__ out_cells) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t s = idxs[i];
    for (uint32_t k = 0; k < 8; k++) { out_keys[(size_t)i * 8 + k] = S.keys[(size_t)s * 8 + k]; out_cells[(size_t)i * 8 + k] = S.cells[(size_t)s * 8 + k]; }
    out_hdr[i * 5 + 0] = S.begin[s]; out_hdr[i * 5 + 1] = S.rowcap[s]; out_hdr[i * 5 + 2] = S.nact[s]; out_hdr[i * 5 + 3] = S.nstat[s]; out_hdr[i * 5 + 4] = S.nk[s];
}
__global__ __launch_bounds__(256) void k_rb2_gather_shared_rows(uint32_t n, const uint32_t *__restrict__ idxs, const uint32_t *__restrict__ offs, ShTable S, const uint32_t *__restrict__ rows, uint32_t *__restrict__ out_rows) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t s = idxs[i], b = S.begin[s], cnt = offs[i + 1] - offs[i];
    for (uint32_t k = 0; k < cnt; k++) out_rows[offs[i] + k] = rows[b + k];
}
__global__ __launch_bounds__(256) void k_rb2_gather_links(uint32_t n, const uint32_t *__restrict__ slots, const uint8_t *__restrict__ cell_links, uint8_t *__restrict__ out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = cell_links[slots[i]];
}

}  // namespace re
