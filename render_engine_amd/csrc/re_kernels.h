// re_kernels.h -- device-side data layout shared by the kernels (re_kernels.hip) and the C-ABI
// host code (re_api.hip).  All arrays are struct-of-arrays in HBM; see DESIGN.md "HBM layout".
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "re_math.h"

namespace re {

// entity flag bits (== RE_F_* of include/re_hip.h) + internal
constexpr uint32_t F_PHANTOM = 0x10000;
constexpr uint32_t F_LIGHT_DIRECTIONAL = 0x2000, F_LIGHT_POINT = 0x4000, F_LIGHT_SPOT = 0x8000, F_LIGHT_ANY = 0xE000;
constexpr uint32_t F_STATIC = 0x001, F_HAS_VEL = 0x002, F_HAS_ACC = 0x004, F_HAS_ROT = 0x008, F_HAS_ROTVEL = 0x010,
                   F_HAS_ROTACC = 0x020, F_HAS_SCALE = 0x040, F_ALWAYS_EXEC = 0x080, F_OOB_LOGIC = 0x100,
                   F_HAS_MOVED = 0x200, F_HAS_ROTATED = 0x400, F_USER = 0x800, F_CAN_COLLIDE = 0x1000, F_DEAD = 0x80000000u;
// world-section flag bits
constexpr uint8_t CF_STATIC_SECTION = 1;   // member of static_world_sections (bounding_box_tree_v2.rs:1133-1213)
constexpr uint8_t CF_STATIC_CACHED = 2;    // its static entities are in the render cache (render_flow.rs:549-594)
constexpr uint8_t CF_STATIC_DIRTY = 4;     // member of changed_static_unique_sections
constexpr uint8_t CF_PAD = 8;              // padding slot that aligns a level run to a wave chunk (not a world section)

constexpr uint32_t ROW_CELL_NONE = 0xFFFFFFFFu, ROW_CELL_SHARED = 0x80000000u;
constexpr int MAX_LEVELS = 16;
constexpr int CULL_THREADS = 256;
constexpr int CULL_ITERS = 4;               // 16-byte key loads in flight per lane
constexpr uint32_t LDS_HIST_SLOTS = 4096;   // group slots (gclass*8+lod) whose two histograms fit the LDS of k_pack_small

struct RowArrays {                          // one row per entity, in upload order
    uint32_t *id, *gclass, *flags;
    float *mat;                             // 16 floats, column-major, 64 B per row
    Aabb *aabb, *orig;                      // StaticAABB, OriginalAABB
    float *pos, *rot, *scale;               // 3, 4 (axis+angle), 3 floats
    uint64_t *key;                          // the key of the row's own (unique) world section, kept in step with row_cell: the tick streams it instead of gathering cell_key[row_cell]
};
struct SharedRec { uint32_t row, nk; uint64_t keys[8]; };

struct LevelBox { uint32_t bx, by, bz, nx, ny, nz; float level_length; uint32_t pad; };
// the same box packed for 16-bit SIMD-within-a-register tests on the key halves (hi = level:16|x:16, lo = z:16|y:16):
// inside  <=>  pk_min_u16(pk_sub_u16(word, sub), min) == pk_sub_u16(word, sub) for both words
struct PBox { uint32_t sub_hi, sub_lo, min_hi, min_lo; };
struct PBoxTable { PBox box[2][16]; };
// Compact keys (worlds of at most 512 sections per axis): 32-bit  x:9 | guard | z:9 | guard | y:9 | guard  (x in bits 20-28), bit 31 = padding slot;
// the level is not in the key (level runs are chunk-aligned: one level word per 512-key chunk).  A box as per-field lower / upper bounds in
// the same layout: inside <=> the guard bits survive both (v | G) - lo and (hi | G) - v.
constexpr uint32_t KEY32_GUARDS = 0x20080200u, KEY32_PAD = 0x80000000u;
struct PBox32 { uint32_t lo, hi; };
struct PBox32Table { PBox32 box[16]; };  // one box per level: the bounding box of the logic and render candidate boxes (the exact tests follow on the full key)
constexpr uint32_t WAVE_KEYS32 = 64u * 4u * 2u;  // compact 32-bit keys: the same 512 per wave (2 x 16 B per lane)
constexpr uint32_t WAVE_KEYS = 64u * 4u * 2u;   // keys one wave of k_scan_cull owns (64 lanes x CULL_ITERS x 2); level runs are padded to it

struct FrameParams {
    float planes[24];
    float cam[3]; float far_draw;
    float lookahead;
    uint32_t n_lod; float lod_min[8], lod_max[8];
    uint32_t max_level, frame, emit_duplicates, pad;
    LevelBox box[2][MAX_LEVELS];            // [0] logic box, [1] render box, per level
};
constexpr uint32_t COUNTER_SHARDS = 64;
constexpr uint32_t PACK_STAGED_GROUPS = 256;   // k_pack_small stages up to this many InstanceRanges in LDS before one wave writes them to the host
constexpr uint32_t CURSOR_SHARDS = 8, CURSOR_STRIDE = 16;
struct FrameHeader {
    unsigned long long cursors[CURSOR_SHARDS * CURSOR_STRIDE];   // one per 128-byte line (global atomics serialise per LINE, ~87 per us: tools/cpp/atomic_line_probe.hip); low 32: emitting sections, high 32: instances.  Sharded by
                                            // wave index mod 8, because a single word saturates near 88 atomics/us
    uint32_t counters[COUNTER_SHARDS * 16]; // one shard per 64-byte line: [0] sections inside a candidate box (== hash probes of the
                                            // reference), [1] visible sections (map), [2] visible sections (vec, with duplicates)
};
struct FrameCounts { uint32_t n_candidates, n_vis_map, n_vis_vec; };
constexpr uint32_t TICK_TICKET_SHARDS = 32, TICK_SHARD_STRIDE = 32;   // counters 128 bytes apart: atomics serialise per 128-byte line
struct TickHeader { uint32_t n_changed, n_rebucket, n_oob, ticket; uint32_t pad[12]; uint32_t shard[TICK_TICKET_SHARDS * TICK_SHARD_STRIDE]; };   // shard[k * STRIDE]: k_tick's share of n_changed, one counter per 128-byte line (a single line serialises the waves' atomics); readers add them up.  shard[k * STRIDE + 1] and pad[1] (device copy): the waves' sign-off counters (tick_sign_off); pad[0] (host copy): the seal
// Speculation across frames of a world with dynamic entities: frames are enqueued without waiting for the previous tick; a tick that
// finds entities that change section (or leave the world) raises `stale`, and every kernel enqueued after it cancels itself until the
// host has patched the tree and replayed those frames.
struct alignas(8) SpecState { uint32_t stale, stale_frame; };   // written and read as one 64-bit word by the tick (k_tick)
// per-frame results the kernels write straight into mapped pinned host memory (no copy kernels)
// position-dependent hash of one table word (device and host)
RE_HD uint32_t table_word_hash(uint32_t v, uint32_t w) { uint32_t x = (v ^ (w * 0x9E3779B1u)) * 0x85EBCA6Bu; return x ^ (x >> 15); }
// Publication of a result block in mapped pinned host memory to the host thread that polls one of its words.
// The block's words are plain stores; the polled word must NOT be: a plain store behind __threadfence_system() carries no scope, and on
// MI355X it was observed to reach host memory up to 6.5 us BEFORE the fenced stores in front of it (tools/cpp/publish_order_test.hip:
// 216 torn blocks in 533,000 polled frames with the plain word, none with the system-scope atomic; it was the cause of round 1's
// `group table inconsistent`).  Form (MI355X_MICROARCH.md, inter-workgroup visibility, "Valid forms", at system scope): every storing
// wave has waited for its own stores (s_waitcnt vmcnt(0)) and joined a workgroup barrier; then ONE lane runs a system-scope release
// fence, waits for it behind an s_waitcnt the compiler cannot drop, and stores the word with a system-scope atomic release store.
#if defined(__HIPCC__)
__device__ __forceinline__ void wait_own_stores() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void publish_to_host(uint32_t *word, uint32_t value) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");                  // "" = system scope: buffer_wbl2 sc0 sc1 + s_waitcnt
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __hip_atomic_store(word, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);   // global_store_dword ... sc0 sc1 (the release is the fence above)
}
__device__ __forceinline__ void post_to_host64(unsigned long long *word, unsigned long long value) {   // a single self-contained word (no block in front of it): scope only, no release
    __hip_atomic_store(word, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
#endif
struct HostResult;
RE_HD uint32_t result_seal(uint32_t table_hash, uint32_t frame, uint32_t n_groups, uint32_t total, uint32_t n_vis_map, uint32_t n_vis_vec, uint32_t n_items) {   // ties the table hash to the frame and the counts
    return (table_hash ^ table_word_hash(frame, 0x101u) ^ table_word_hash(n_groups, 0x102u) ^ table_word_hash(total, 0x103u) ^ table_word_hash(n_vis_map, 0x104u) ^ table_word_hash(n_vis_vec, 0x105u) ^ table_word_hash(n_items, 0x106u)) | 1u;
}
struct HostResult {
    uint32_t n_vis_map, n_vis_vec, n_groups, total, n_candidates, overflow, n_entries, n_items;
    uint32_t done_frame, table_hash;        // table_hash: hash of the InstanceRange table words k_pack_small wrote to host memory (0: none), so the reader can tell a table whose
                                            // posted writes have not all landed yet.  done_frame: written last (after a system-scope fence): the frame whose results above are complete -- a synchronous
                                            // call polls this word instead of paying the driver's stream-synchronise latency
};
constexpr uint32_t RESULT_SEGMENT_OVERFLOW = 3u;   // HostResult::overflow: a cursor segment of the instance list overflowed (0 = packed, 1 = k_pack_small declined: too many instances for it, 2 = frame cancelled by cross-frame speculation)
constexpr uint32_t PACK_SMALL_ITEMS = 16383;   // instances k_pack_small takes (every workgroup counts all of them); more go through the count/scan/scatter path
struct SharedArrays {                       // shared world sections (bounding_box_tree_v2.rs:113-155, 253-316)
    uint32_t n;
    const int32_t *cells; const Aabb *aabb; const uint32_t *begin, *nact, *nstat; const int32_t *owner; const uint8_t *cached;
};
struct InstanceRange { uint32_t model_index, render_system, sortable, begin, count; };

__global__ void k_transform_assign(RowArrays R, uint32_t row0, uint32_t n, uint32_t outline, uint32_t atomic, uint64_t *row_key, uint8_t *row_nk,
                                   SharedRec *shrec, uint32_t *shrec_count, uint32_t shrec_cap);
__global__ void k_fold_tight(uint32_t ncells, const uint64_t *cell_key, const uint32_t *cell_begin, const uint32_t *cell_nlocal, const uint32_t *cell_nstatic,
                             const uint32_t *rows, const Aabb *ent_aabb, Aabb *cell_tight, uint32_t atomic, int too_many);
__global__ void k_fold_shared(uint32_t nsh, const uint32_t *sh_begin, const uint32_t *sh_nact, const uint32_t *sh_nstat, const uint32_t *rows, const Aabb *ent_aabb, Aabb *sh_aabb);
__global__ void k_static_cache_cells(uint32_t ncells, const Aabb *cell_tight, uint8_t *cell_flags, FrameParams P);
__global__ void k_clear_static_dirty(uint32_t ncells, uint8_t *cell_flags, uint32_t nsh, uint8_t *sh_dirty);
__global__ void k_static_cache_shared(uint32_t nsh, const int32_t *sh_cells, const uint64_t *cell_key, const Aabb *sh_aabb, uint8_t *sh_dirty, int32_t *sh_owner, uint8_t *sh_cached, FrameParams P);
struct ItemSink {
    uint32_t *item_row, *item_slot; uint32_t item_cap;
    uint32_t nshards, seg_cap;              // instance list = nshards segments of seg_cap slots, one cursor each
    const uint32_t *rows, *rows_gc;         // the row pool and, entry for entry, the row's group class (0xFFFFFFFF: not drawn -- removed or hidden)
    const uint32_t *gc_lodtab; const uint32_t *lod_n; const float *lod_min, *lod_max;   // per-model level-of-view bands (level_views.custom, render_flow.rs:889-893): band table of each
                                            // group class (0: the camera's default bands), tables of 8 bands each; nullptr while no model has custom bands
    uint32_t slot_write_through;            // the one-launch synchronous frame (k_scan_cull_sync): the last workgroup of the SAME launch reads item_slot, so its stores go through to memory (sc1)
    uint32_t *group_count; uint32_t count_nslots;   // large visible sets: the expansion also counts the instances per (cursor shard, group slot) -- [nshards][count_nslots],
                                            // through a per-wave LDS histogram flushed once per wave -- so that the pack needs no counting pass (nullptr / 0: off)
};
constexpr uint32_t COUNT_SLOTS_MAX = 512;   // group slots the in-scan counting (and k_pack_large) handle; larger tables take the count / scan / scatter kernels
#ifndef RE_PACK_THREADS
#define RE_PACK_THREADS 512
#endif
#ifndef RE_PACK_CHUNK
#define RE_PACK_CHUNK 8
#endif
// k_pack_large: threads per workgroup, 16-byte matrix loads each thread keeps in flight, instances per tile.  512 / 8 / 1024: every thread has its whole
// share of the tile's matrices in flight at once (256 threads needed two dependent rounds: 21.9 -> 19.6 us at 501 K instances), and the 624 workgroups of that
// frame are all resident together (1024 threads or 512-instance tiles need a second round of workgroups: 28.6 / 26.0 us).  tools/pack_variants.py, round 2.
constexpr uint32_t PACK_LARGE_THREADS = RE_PACK_THREADS, PACK_LARGE_CHUNK = RE_PACK_CHUNK;
#ifndef RE_PACK_TILE
#define RE_PACK_TILE 1024
#endif
constexpr uint32_t PACK_LARGE_TILE = RE_PACK_TILE;  // instances per workgroup iteration of k_pack_large
struct PackArgs {                           // what k_pack_small needs besides the item list
    uint32_t nslots, out_cap;
    const uint32_t *row_id; const float *row_mat; uint32_t *out_ids; float *out_mats;
    const uint32_t *gc_model, *gc_rs, *gc_sort; InstanceRange *ranges; HostResult *hres; const SpecState *spec;
    uint32_t frame;                         // this frame's number (HostResult::done_frame)
    uint32_t *out_count;                    // optional device word: instances written to the output buffers (the all-gather slab header)
    uint32_t flags;                         // PACK_NO_PUBLISH: the frame's result block has been published by the scan's own last workgroup (k_scan_cull_sync); this launch only moves the instances
};
constexpr uint32_t PACK_NO_PUBLISH = 1u;
constexpr uint32_t SYNC_TAIL_SLOTS = 256;      // group slots the tail of k_scan_cull_sync takes (its histogram is dynamic LDS of every workgroup of the scan: 1 KB)
struct ScanCullArgs {                       // the kernel-argument segment of k_scan_cull after its two leading scalars (the kernel addresses it explicitly)
    PBoxTable B; PBox32Table B32;           // read by every wave (one of the two); everything below by candidate waves only
    const uint64_t *cell_key64;             // the full keys (candidate waves; the stream may run over the compact 32-bit keys)
    FrameParams P; FrameParams *P_dev;
    const Aabb *cell_tight; const uint32_t *cell_begin, *cell_nlocal, *cell_nstatic, *cell_nghost; const uint8_t *cell_flags; uint32_t *cell_stamp;
    ItemSink K; FrameHeader *hdr; SharedArrays S; const SpecState *spec;
#ifdef RE_EXP_STAMPS
    unsigned long long *timeline;           // development builds: [wave] = {start, keys arrived, end} 100 MHz stamps
#endif
};
// Workgroup order of k_scan_cull: up to 4 disjoint, ascending spans of key chunks (one chunk = the 2048 keys of a workgroup) run first,
// the other chunks follow in key order.  The host puts the x-slabs of the candidate boxes there, so the few long-running
// candidate waves start at once and finish under the stream instead of after it.
struct ScanSpans { uint32_t n, start[4], count[4]; };
constexpr uint32_t SCAN_CULL_ARGS_OFFSET = 56;   // keys (8 bytes) + ncells (4) + the 9 span scalars (36) + chunk_level (8), already a multiple of the 8-byte alignment of ScanCullArgs
template <bool K32> __global__ void k_scan_cull(const void *keys, uint32_t ncells, uint32_t nsp, uint32_t s0, uint32_t c0, uint32_t s1, uint32_t c1, uint32_t s2, uint32_t c2,
                                                uint32_t s3, uint32_t c3, const uint32_t *chunk_level, ScanCullArgs A);   // the leading scalars arrive preloaded in SGPRs
extern template __global__ void k_scan_cull<false>(const void *, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t, const uint32_t *, ScanCullArgs);
extern template __global__ void k_scan_cull<true>(const void *, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t, const uint32_t *, ScanCullArgs);
__global__ void k_scan_cull_wide(const void *keys, uint32_t ncells, uint32_t nsp, uint32_t s0, uint32_t c0, uint32_t s1, uint32_t c1, uint32_t s2, uint32_t c2,
                                 uint32_t s3, uint32_t c3, const uint32_t *chunk_level, ScanCullArgs A);   // k_scan_cull<true> for frames with a large visible set
__global__ void k_pack_small(FrameHeader *hdr, FrameHeader *hdr_next, TickHeader *th, PackArgs A, ItemSink K, uint32_t nrows);
// One launch between the call and the host's answer for a SYNCHRONOUS frame with a small visible set: the scan, whose last workgroup to finish publishes the
// InstanceRange table and the counts (k_pack_small then only moves the instances, stream-ordered behind it, while the host is already back).
template <bool K32> __global__ void k_scan_cull_sync(const void *keys, uint32_t ncells, uint32_t nsp, uint32_t s0, uint32_t c0, uint32_t s1, uint32_t c1, uint32_t s2, uint32_t c2,
                                                     uint32_t s3, uint32_t c3, const uint32_t *chunk_level, ScanCullArgs A, PackArgs T);
extern template __global__ void k_scan_cull_sync<false>(const void *, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t, const uint32_t *, ScanCullArgs, PackArgs);
extern template __global__ void k_scan_cull_sync<true>(const void *, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t, const uint32_t *, ScanCullArgs, PackArgs);
struct FusedPack { FrameHeader *hdr, *hdr_next; TickHeader *th; PackArgs A; ItemSink K; uint32_t nrows, pad; };   // the previous frame's pack, carried by the next frame's launch
template <bool K32> __global__ void k_scan_cull_fused(const void *keys, uint32_t ncells, uint32_t nsp_npack, uint32_t s0, uint32_t c0, uint32_t s1, uint32_t c1, uint32_t s2, uint32_t c2,
                                                      uint32_t s3, uint32_t c3, const uint32_t *chunk_level, ScanCullArgs A, FusedPack F);
extern template __global__ void k_scan_cull_fused<false>(const void *, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t, const uint32_t *, ScanCullArgs, FusedPack);
extern template __global__ void k_scan_cull_fused<true>(const void *, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t, const uint32_t *, ScanCullArgs, FusedPack);
__global__ void k_emit_count(const FrameHeader *hdr, const uint32_t *item_slot, uint32_t nshards, uint32_t seg_cap, uint32_t *group_count, uint32_t nslots, const SpecState *spec);
__global__ void k_emit_count_sharded(const FrameHeader *hdr, const uint32_t *item_slot, uint32_t nshards, uint32_t seg_cap, uint32_t *group_count, uint32_t nslots, const SpecState *spec);
// The pack of a large visible set in ONE launch behind the scan (which counted the instances per (shard, group), see ItemSink): every workgroup
// scans the counts itself, workgroup 0 publishes the InstanceRange table and the frame result, and each workgroup moves tiles of one cursor shard.
struct PackLargeArgs {
    const FrameHeader *hdr; FrameHeader *hdr_next; TickHeader *th;
    const uint32_t *gcount; uint32_t *gfill;            // [nshards][nslots]: this frame's counts (read only), running fill per (shard, group)
    uint32_t *zero_a, *zero_b; uint32_t zero_words;     // the other frame parity's count / fill arrays: cleared here for the next frame (may be null)
    uint32_t nslots, out_cap, range_cap, frame;
    const uint32_t *item_row, *item_slot; uint32_t nshards, seg_cap;
    const uint32_t *row_id; const float *row_mat; uint32_t *out_ids; float *out_mats;
    const uint32_t *gc_model, *gc_rs, *gc_sort; InstanceRange *ranges; HostResult *hres; const SpecState *spec; uint32_t *out_count;
};
__global__ void k_pack_large(PackLargeArgs A);
__global__ void k_group_scan(uint32_t *group_count, uint32_t *group_begin, uint32_t *group_fill, uint32_t nslots, const uint32_t *gc_model, const uint32_t *gc_rs,
                             const uint32_t *gc_sort, InstanceRange *ranges, uint32_t range_cap, FrameHeader *hdr, FrameHeader *hdr_next, TickHeader *th, HostResult *hres, const SpecState *spec,
                             uint32_t *out_count, uint32_t out_cap, uint32_t frame, uint32_t seg_cap);
__global__ void k_emit_scatter(const FrameHeader *hdr, const uint32_t *item_row, const uint32_t *item_slot, uint32_t nshards, uint32_t seg_cap, const uint32_t *group_begin,
                               uint32_t *group_fill, uint32_t nslots, const uint32_t *row_id, const float *row_mat, uint32_t *out_ids, float *out_mats, uint32_t out_cap, const SpecState *spec);
__global__ void k_tick(uint32_t ndyn, float *dyn_vel, const float *dyn_acc, float *dyn_rotvel, const float *dyn_rotacc, RowArrays R,
                       const uint32_t *row_cell, const uint64_t *cell_key, const uint32_t *cell_stamp, const uint8_t *cell_flags, const int32_t *sh_cells,
                       const Aabb *sh_aabb, const FrameParams *P, float dt, uint32_t tick_all, uint32_t outline, uint32_t atomic, TickHeader *th,
                       uint32_t *mover_rows, uint32_t *oob_rows, uint32_t list_cap, SpecState *spec, SpecState *h_spec, uint32_t tick_frame,
                       uint32_t ndyn0, const uint32_t *dyn_row, TickHeader *h_th, uint32_t publish_seq);
__global__ void k_shift_rows(uint32_t m, uint32_t *rows, uint32_t from, uint32_t delta);   // pool entries >= from (ghost instances) move up by delta when the row columns grow
// Probe path of the visibility query (opt-in, RE_CFG_PROBE): instead of streaming every section key, enumerate the cells of the two
// candidate boxes (their bounding box per level) and look each one up in a device hash table key -> slot, like the reference's
// own contains_key probes (visible_world_flow.rs:96-104).  Work is O(candidates), not O(sections).
constexpr uint32_t PROBE_KEYS = 64;        // candidate cells per wave of k_probe_cull: few, so that the dependent memory round trips of the hits spread over many waves
struct HashEntry { unsigned long long key; uint32_t slot, pad; };          // empty: key == ~0; removed section: slot == ~0 (the key stays as a tombstone)
struct ProbeArgs { const HashEntry *tab; uint32_t mask, nwaves; uint32_t wave0[MAX_LEVELS + 1]; LevelBox ubox[MAX_LEVELS]; };
__global__ void k_hash_build(uint32_t ncells, const uint64_t *cell_key, HashEntry *tab, uint32_t mask);
struct Pair64;
__global__ void k_hash_patch(uint32_t m, const Pair64 *slot_newkey, const uint64_t *cell_key_old, HashEntry *tab, uint32_t mask, uint32_t insert_pass);
__global__ void k_probe_cull(ProbeArgs Q, ScanCullArgs A);
// collision broad phase (re_collide.hip)
constexpr float COLLISION_DISTANCE = 200.0f;           // handle_collisions keeps sections within this distance of the camera (logic_flow.rs:553-566)
struct ColHeader { uint32_t n_region, n_high, n_shared, n_moved, n_pairs, n_near, pad[2]; };
struct ColRegion { uint64_t key, top; uint32_t slot, near; };            // an existing section around the camera; top: its topmost existing ancestor
struct ColNear { uint64_t top; uint32_t begin, n; };                      // a section within the distance that holds non-static entities: its rows in the pool
struct ColShared { uint64_t top[8]; uint32_t s, nk; };                    // a shared section within the distance; top[k]: topmost ancestor of its k-th linking section
struct ColMoved { uint64_t key; unsigned long long order; uint32_t row, info, tslot, pad; };   // (section, moved entity); info bit 0: Shared lookup, bits 1-2: listings; tslot: its first-touch table entry
template <bool K32> __global__ void k_col_region(uint32_t ncells, const void *keys, const uint32_t *chunk_level, const Aabb *cell_tight, const FrameParams *P, uint32_t atomic,
                                                 ColHeader *hdr, ColRegion *region, uint32_t region_cap, uint32_t *high, uint32_t high_cap);
extern template __global__ void k_col_region<false>(uint32_t, const void *, const uint32_t *, const Aabb *, const FrameParams *, uint32_t, ColHeader *, ColRegion *, uint32_t, uint32_t *, uint32_t);
extern template __global__ void k_col_region<true>(uint32_t, const void *, const uint32_t *, const Aabb *, const FrameParams *, uint32_t, ColHeader *, ColRegion *, uint32_t, uint32_t *, uint32_t);
__global__ void k_col_shared(uint32_t nsh, const Aabb *sh_aabb, const int32_t *sh_cells, const uint32_t *sh_nact, const uint32_t *sh_nstat, const uint64_t *cell_key,
                             const FrameParams *P, ColHeader *hdr, ColShared *out, uint32_t cap);
__global__ void k_col_tops(ColHeader *hdr, ColRegion *region, uint32_t region_cap, const uint32_t *high, uint32_t high_cap, ColShared *shared, uint32_t shared_cap,
                           const uint32_t *cell_begin, const uint32_t *cell_nlocal, ColNear *near, uint32_t near_cap);
__global__ void k_col_moved(uint32_t ndyn, const uint32_t *dyn_row, const uint32_t *dyn_cell, uint32_t user_row, uint32_t user_cell, RowArrays R, const uint64_t *cell_key,
                            const uint32_t *cell_stamp, const uint8_t *cell_flags, const int32_t *sh_cells, const Aabb *sh_aabb, const FrameParams *P, ColHeader *hdr,
                            ColMoved *moved, uint32_t moved_cap, uint8_t *row_moved, unsigned long long *tab_key, unsigned long long *tab_min, uint32_t tab_mask);
__global__ void k_col_pairs(ColHeader *hdr, const ColMoved *moved, uint32_t moved_cap, const ColNear *near, uint32_t near_cap, const ColShared *shared, uint32_t shared_cap,
                            RowArrays R, const uint32_t *sh_begin, const uint32_t *sh_nact, const uint32_t *rows, const uint8_t *row_moved, const unsigned long long *tab_min,
                            uint2 *pairs, uint32_t pair_cap);
__global__ void k_col_clear(const ColHeader *hdr, const ColMoved *moved, uint32_t moved_cap, uint8_t *row_moved, unsigned long long *tab_key, unsigned long long *tab_min,
                            ColHeader *h_hdr, uint32_t call);
struct WriteOp { uint32_t comp, index; uint32_t v[4]; };
constexpr uint32_t WRITE_GCLASS = 101;    // v[0] = group class of the row (0xFFFFFFFF hides it from the pack)
constexpr uint32_t WRITE_FLAGS = 100;     // v[0] = and-mask, v[1] = or-mask, v[2] != 0: also retire the row's group class (entity removed)
__global__ void k_write_components(uint32_t m, const WriteOp *ops, RowArrays R, float *dyn_vel, float *dyn_acc, float *dyn_rotvel, float *dyn_rotacc);
constexpr uint32_t APPLY_SMALL_MAX = 256;   // component writes / moved entities of a change batch that k_apply_small takes in ONE launch of one workgroup
__global__ void k_apply_small(uint32_t n_ops, const WriteOp *ops, uint32_t n_rows, const uint32_t *rows, RowArrays R, float *dyn_vel, float *dyn_acc, float *dyn_rotvel, float *dyn_rotacc,
                              const uint32_t *row_cell, const uint64_t *cell_key, const int32_t *sh_cells, uint32_t outline, uint32_t atomic,
                              TickHeader *th, uint32_t *mover_rows, uint32_t *oob_rows, uint32_t list_cap, TickHeader *h_th, uint32_t seq);
__global__ void k_apply_rows(uint32_t m, const uint32_t *rows, RowArrays R, const uint32_t *row_cell, const uint64_t *cell_key, const int32_t *sh_cells, uint32_t outline,
                             uint32_t atomic, TickHeader *th, uint32_t *mover_rows, uint32_t *oob_rows, uint32_t list_cap);
__global__ void k_assign_rows(uint32_t m, const uint32_t *rows, RowArrays R, uint32_t outline, uint32_t atomic, uint8_t *out_nk, uint64_t *out_keys);
__global__ void k_fold_tight_masked(uint32_t ncells, const uint64_t *cell_key, const uint32_t *cell_begin, const uint32_t *cell_nlocal, const uint32_t *cell_nstatic,
                                    const uint32_t *rows, const Aabb *ent_aabb, Aabb *cell_tight, uint32_t atomic, int too_many, const uint8_t *refold, const Aabb *carried);
// incremental section-table patches (host-assisted re-bucket): staged (index, value) pairs scattered into the resident arrays
struct Pair32 { uint32_t idx, val; };
struct Pair64 { uint32_t idx, pad; uint64_t val; };
struct FlagOp { uint32_t idx; uint8_t and_mask, or_mask, pad[2]; };
// ---- device-side re-bucket bookkeeping (re_api.hip: rebucket_on_device) ----
constexpr uint32_t RB_REMOVE = 0x80000000u;                  // op row word: bit 31 = remove_entity (else add_entity)
struct RbTables {                                            // key -> slot of the resident table: the immutable sorted keys of the last full build + an overlay of sections created since
    const uint64_t *base_keys; uint32_t nbase;
    unsigned long long *ovl_keys; uint32_t *ovl_slots; uint32_t ovl_mask;
};
struct RbCells {
    uint64_t *cell_key; uint32_t *cell_key32; uint32_t *cell_begin, *cell_cap, *cell_nl, *cell_ns, *cell_ng, *cell_stamp; uint8_t *cell_flags;
    uint32_t *rows, *rows_gc, *row_cell; uint64_t *row_key; uint32_t pool_cap;
    const uint8_t *cell_links;                               // shared sections linking each unique section (maintained by the host paths)
};
// ---- the device-side re-bucket with shared sections (re_rebucket.hip; re_api.hip: rebucket_on_device) ----
constexpr uint64_t RB2_SHARED_BIT = 0x8000000000000000ull;   // placement key of a shared section: this bit + 63 bits of the hash of its id (a unique section's key has a level < 16 up there)
constexpr uint32_t RB2_LINK_INC = 0x40000000u, RB2_LINK_DEC = 0x20000000u;   // op words of the link ops a shared section's creation / emptying sends to the sections it links
struct ShTable {                                             // shared world sections with STABLE indices: cap entries, a retired one is a hole (nk == 0, no members, no links)
    int32_t *cells; Aabb *aabb; uint32_t *begin, *nact, *nstat, *rowcap; int32_t *owner; uint8_t *cached, *dirty;
    uint64_t *keys; uint8_t *nk;                             // the id (SharedWorldSectionId): nk linked section keys
    unsigned long long *hkeys; uint32_t *hidx; uint32_t hmask;   // placement key -> index (open addressing; a retired entry keeps its key with index ~0)
    uint32_t cap;
};
struct Rb2Status {                                           // one block the host reads between the phases
    uint32_t fallback, err;                                  // fallback: a mover / a collision of placement keys the device path does not handle (nothing has been touched); err: accounting broke in the apply phase
    uint32_t total;                                          // total_world_aabb_combining of the batch
    uint32_t nseg_u, nseg_s, n_link, n_pairs, n_host;
    uint32_t need_pool, need_sh, need_slots[MAX_LEVELS];
    uint32_t popped[MAX_LEVELS], popped_sh, pool_used;
    uint32_t n_created, n_freed, n_sh_created, n_sh_freed;
};
struct Rb2Seg { uint64_t key; int32_t slot; uint32_t op_begin, op_count, nl1, ns, links1; uint8_t exists0, exists1, created, freed, changed, pad[3]; };
struct Rb2ShSeg { uint64_t pkey; uint64_t keys[8]; int32_t idx; uint32_t op_begin, op_count, na1, nst, nk; uint8_t exists0, exists1, created, freed, relink, pad[3]; };
constexpr uint32_t RB2_MOVER_DELETED = 0x40000000u;   // word of the mover list: a row the batch deletes (remove op only); bit 31 = translation-only mover as before
constexpr uint32_t RB2_MOVER_ADDED = 0x20000000u;     // ... a row the batch adds (AddEntity: add op only, where its StaticAABB puts it)
__global__ void k_rb2_clear_deleted(uint32_t n, const uint32_t *movers, uint32_t *row_cell);
__global__ void k_rb2_publish_status(Rb2Status *st, Rb2Status *h_st, uint32_t *h_seq, uint32_t seq, const uint32_t *src_a, uint32_t *dst_a, uint32_t words_a,
                                     const uint32_t *src_b, uint32_t *dst_b, uint32_t words_b, uint32_t reset);
constexpr uint32_t RB2_STATIC_SMALL_PAIRS = 2048, RB2_STATIC_SMALL_SHARED = 32768;   // k_rb2_static_small: update_static_world_sections of a batch with up to 256 changed shared sections in a world of up to 32,768, by one workgroup
__global__ void k_rb2_static_small(uint32_t nsh, ShTable S, RbCells C, const uint8_t *cell_links, uint8_t *cell_inact, const Rb2Seg *segs_u, const Rb2ShSeg *segs_s, const Rb2Status *st);
constexpr uint32_t RB2_INLINE_WORDS = 2048;   // segment lists up to this many words travel with the status block (k_rb2_publish_status) instead of as stream copies; free lists up to this many entries are read by the apply kernels from the mapped block
constexpr uint32_t RB2_SORT_SMALL = 2048;   // ops one workgroup sorts in LDS (k_rb2_sort_small: 40 KB)
__global__ void k_rb2_sort_small(uint32_t n, const uint64_t *key, const uint64_t *ord, uint64_t *key_sorted, uint32_t *perm, const uint32_t *n_extra);
__global__ void k_rb2_gather_u32(uint32_t n, const uint32_t *perm, const uint32_t *src, uint32_t *dst);
__global__ void k_rb2_hash_insert(uint32_t n, ShTable S);
__global__ void k_rb2_ops(uint32_t m, const uint32_t *movers, RowArrays R, RbCells C, ShTable S, uint32_t outline, uint32_t atomic, uint64_t *op_key, uint64_t *op_key2, uint64_t *op_ord,
                          uint32_t *op_row, uint32_t *op_idx, uint64_t *mk, uint8_t *mnk, uint32_t *host_list, Rb2Status *st);
__global__ void k_rb2_shared_segments(uint32_t n, uint32_t m, const uint32_t *perm, const uint64_t *key_sorted, const uint32_t *op_row, uint64_t *op_ord, const uint64_t *mk, const uint8_t *mnk,
                                      ShTable S, RbCells C, uint64_t *op_key2, uint32_t *op_row_w, uint32_t *op_idx, uint32_t link_cap, Rb2ShSeg *segs, Rb2Status *st);
__global__ void k_rb2_unique_segments(uint32_t n, const uint32_t *perm, const uint64_t *key_sorted, const uint32_t *op_row, RbTables T, RbCells C, uint8_t *cell_links, Rb2Seg *segs, Rb2Status *st);
__global__ void k_rb2_apply_small(uint32_t m, const uint32_t *movers, uint32_t has_deleted, const uint32_t *perm1, const uint32_t *perm2, const uint32_t *op_row, RbTables T, RbCells C, RowArrays R, ShTable S,
                                  uint8_t *cell_links, uint8_t *cell_inact, Aabb *cell_tight, Rb2Seg *segs_u, Rb2ShSeg *segs_s, Rb2Status *st, const uint32_t *free_u, const uint32_t *free_off, const uint32_t *free_s,
                                  uint32_t *tmp_u, uint32_t *tmp_s, uint32_t *refold, uint32_t nsh, uint32_t atomic, uint32_t too_many, uint32_t segments_done, Rb2Status *h_st, uint32_t *h_seq, uint32_t seq, uint32_t *h_segs_u, uint32_t *h_segs_s);
constexpr uint32_t RB2_PLAN_SMALL = RB2_SORT_SMALL / 2u;   // movers of a batch whose ops (2 each) fit the one-workgroup sort: phases 1-3 in one launch (k_rb2_plan_small) -- if the link ops they emit (up to 8 per op) fit too; the host sees from the status block when they did not and plans the batch again with the kernels of large batches
__global__ void k_rb2_plan_small(uint32_t m, const uint32_t *movers, RowArrays R, RbCells C, ShTable S, RbTables T, uint32_t outline, uint32_t atomic, uint64_t *op_key, uint64_t *op_key2, uint64_t *op_ord,
                                 uint32_t *op_row, uint32_t *op_idx, uint64_t *mk, uint8_t *mnk, uint32_t *host_list, uint64_t *ksorted1, uint32_t *perm1, uint64_t *ksorted2, uint32_t *perm2, uint32_t link_cap,
                                 const uint8_t *cell_links, Rb2ShSeg *segs_s, Rb2Seg *segs_u, Rb2Status *st, Rb2Status *h_st, uint32_t *h_seq, uint32_t seq, uint32_t *h_segs_u);
__global__ void k_rb2_apply_unique(const uint32_t *perm, const uint32_t *op_row, RbTables T, RbCells C, RowArrays R, uint8_t *cell_links, Rb2Seg *segs, Rb2Status *st,
                                   const uint32_t *free_slots, const uint32_t *free_off, uint32_t *tmp_row, uint32_t *refold);
__global__ void k_rb2_apply_shared(const uint32_t *perm, const uint32_t *op_row, RbTables T, RbCells C, RowArrays R, ShTable S, Rb2ShSeg *segs, Rb2Status *st, const uint32_t *free_sh, uint32_t *tmp_row);
__global__ void k_rb2_mark_inactive(uint32_t nsh, ShTable S, uint8_t *cell_inact, uint8_t value);
__global__ void k_rb2_static_first(RbCells C, const uint8_t *cell_links, const uint8_t *cell_inact, const Rb2Seg *segs, const Rb2Status *st);
__global__ void k_rb2_static_pairs(const Rb2ShSeg *segs, const Rb2Status *st, ShTable S, uint64_t *pair_key, uint32_t *pair_seg, Rb2Status *stw);
__global__ void k_rb2_static_second(uint32_t n, const uint64_t *slot_sorted, const uint32_t *seg_sorted, const Rb2ShSeg *segs, RbCells C);
__global__ void k_rb2_gather_shared(uint32_t n, const uint32_t *idxs, ShTable S, uint64_t *out_keys, uint32_t *out_hdr, int32_t *out_cells);
__global__ void k_rb2_gather_shared_rows(uint32_t n, const uint32_t *idxs, const uint32_t *offs, ShTable S, const uint32_t *rows, uint32_t *out_rows);
__global__ void k_rb2_gather_links(uint32_t n, const uint32_t *slots, const uint8_t *cell_links, uint8_t *out);
#if defined(__HIPCC__)
// key -> slot of the resident section table (shared by the re-bucket kernels of re_kernels.hip and re_rebucket.hip)
__device__ __forceinline__ uint32_t rb_hash(uint64_t k) { k ^= k >> 33; k *= 0xFF51AFD7ED558CCDull; k ^= k >> 29; return (uint32_t)k; }
__device__ __forceinline__ int32_t rb_find(const RbTables &T, const uint64_t *cell_key, uint64_t key) {
    for (uint32_t h = rb_hash(key) & T.ovl_mask;; h = (h + 1u) & T.ovl_mask) {          // sections created since the last full build (authoritative for the keys they hold)
        const unsigned long long k = T.ovl_keys[h];
        if (k == ~0ull) break;
        if (k == key) { const uint32_t sl = T.ovl_slots[h]; return cell_key[sl] == key ? (int32_t)sl : -1; }
    }
    uint32_t lo = 0, hi = T.nbase;                                                     // lower_bound in the immutable sorted keys of the last full build
    while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (T.base_keys[mid] < key) lo = mid + 1u; else hi = mid; }
    return (lo < T.nbase && T.base_keys[lo] == key && cell_key[lo] == key) ? (int32_t)lo : -1;
}
__device__ __forceinline__ void rb_ovl_put(const RbTables &T, uint64_t key, uint32_t slot) {
    for (uint32_t h = rb_hash(key) & T.ovl_mask;; h = (h + 1u) & T.ovl_mask) {
        const unsigned long long prev = atomicCAS(&T.ovl_keys[h], ~0ull, (unsigned long long)key);
        if (prev == ~0ull || prev == key) { T.ovl_slots[h] = slot; return; }
    }
}
#endif
hipError_t sort_pairs_u64_u32(void *tmp, size_t *tmp_bytes, const uint64_t *keys_in, uint64_t *keys_out, const uint32_t *vals_in, uint32_t *vals_out,
                              uint32_t n, unsigned begin_bit, unsigned end_bit, hipStream_t stream);      // re_sort.hip (rocPRIM radix sort)
__global__ void k_rb_ovl_insert(uint32_t n, const Pair64 *pairs, RbTables T);
__global__ void k_rb_gather_keys(uint32_t n, const uint32_t *perm, const uint64_t *key, uint64_t *out);
__global__ void k_rb_gather_cells(uint32_t n, const uint32_t *slots, RbCells C, uint64_t *out_key, uint32_t *out_hdr);
__global__ void k_rb_gather_rows(uint32_t n, const uint32_t *slots, const uint32_t *offs, RbCells C, uint32_t *out_rows);
struct LightQuery { LevelBox box[MAX_LEVELS]; Aabb culler; uint32_t max_level, type_flag; };
__global__ void k_visible_lights(uint32_t n, const uint32_t *light_rows, const uint32_t *flags, const uint32_t *row_id, const uint32_t *row_cell, const uint64_t *cell_key,
                                 const uint8_t *cell_flags, const int32_t *sh_cells, LightQuery Q, uint32_t *out_ids, uint32_t cap, uint32_t *count);
__global__ void k_scatter32(uint32_t m, const Pair32 *pairs, uint32_t *dst);
__global__ void k_clone_rows(uint32_t m, const Pair32 *src_dst, uint32_t *row_id, float *row_mat);   // ghost instances: (source row, ghost row)
__global__ void k_scatter64(uint32_t m, const Pair64 *pairs, uint64_t *dst);
__global__ void k_flag_ops(uint32_t m, const FlagOp *ops, uint8_t *flags);
__global__ void k_fold_tight_list(uint32_t m, const uint32_t *slots, const uint64_t *cell_key, const uint32_t *cell_begin, const uint32_t *cell_nlocal, const uint32_t *cell_nstatic,
                                  const uint32_t *rows, const Aabb *ent_aabb, Aabb *cell_tight, uint32_t atomic, int too_many);
// ECS::get_indexes_for_components over the presence column: ids of the live rows whose flag word has every bit of need_mask
__global__ void k_query_flags(uint32_t n, const uint32_t *flags, const uint32_t *row_id, uint32_t need_mask, uint32_t *out_ids, uint32_t cap, uint32_t *count);
struct ExportRec { uint32_t id, model_index, render_system, sortable, flags; float orig[6], pos[3], rot[4], scale[3], vel[3], acc[3], rotvel[4], rotacc[4]; };   // == re_entity_state (include/re_hip.h)
__global__ void k_export_rows(uint32_t m, const uint32_t *rows, const uint32_t *dyn_slot, RowArrays R, const float *dyn_vel, const float *dyn_acc, const float *dyn_rotvel,
                              const float *dyn_rotacc, ExportRec *out);
__global__ void k_gather_headers(uint32_t n_ranks, const uint32_t *recv, uint32_t words_per_rank, uint32_t *h_hdr, uint32_t *h_seq, uint32_t seq);
__global__ void k_collect_visible(uint32_t ncells, const uint32_t *cell_stamp, uint32_t frame, uint32_t *out_idx, uint8_t *out_mult, uint32_t cap, uint32_t *count);

}  // namespace re
