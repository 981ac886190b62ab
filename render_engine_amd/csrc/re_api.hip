// re_api.hip -- host side of librender_engine_hip.so: the extern "C" entry points declared in
// include/re_hip.h, device-memory ownership, world build and per-frame kernel orchestration on one
// HIP stream per context.  No CPU fallback exists: every entry point needs a HIP device.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <array>
#include <set>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <thread>
#include <unordered_map>
#include <unordered_set>
#include <vector>

#include "re_hip.h"
#include "re_guard.h"
#include "re_kernels.h"
#include "re_math.h"

using namespace re;

static thread_local std::string g_create_error;

namespace {

template <typename T> struct DevBuf {
    T *p = nullptr; size_t n = 0;
    hipError_t alloc(size_t count, uint64_t *acct) {
        release(acct);
        if (count == 0) count = 1;
        hipError_t e = hipMalloc(reinterpret_cast<void **>(&p), count * sizeof(T));
        if (e == hipSuccess) { n = count; if (acct) *acct += count * sizeof(T); } else p = nullptr;
        return e;
    }
    void release(uint64_t *acct) { if (p) { (void)hipFree(p); if (acct) *acct -= n * sizeof(T); } p = nullptr; n = 0; }
};

struct GroupKey { uint32_t model, rs, sort; bool operator==(const GroupKey &o) const { return model == o.model && rs == o.rs && sort == o.sort; } };
struct GroupKeyHash { size_t operator()(const GroupKey &k) const { return (size_t)k.model * 0x9E3779B97F4A7C15ull ^ ((size_t)k.rs << 32) ^ ((size_t)k.sort << 20); } };

}  // namespace

struct SharedIdPub { uint32_t nk; uint64_t keys[8];
                     // canonical order: keys lexicographic (numeric), then count -- the order the oracle visits changed shared sections in
                     bool operator<(const SharedIdPub &o) const { uint32_t m = nk < o.nk ? nk : o.nk; for (uint32_t i = 0; i < m; i++) if (keys[i] != o.keys[i]) return keys[i] < o.keys[i]; return nk < o.nk; }
                     bool operator==(const SharedIdPub &o) const { return nk == o.nk && memcmp(keys, o.keys, sizeof keys) == 0; } };

// State of the structure before an incremental re-bucket (apply_change semantics): unchanged world sections keep their tight AABB
// (which may be stale: add_entity returns early when an entity stays in its section), static-section flag and render-cache bits.
struct Carry {
    std::vector<uint64_t> keys; std::vector<Aabb> tight; std::vector<uint8_t> flags;
    std::vector<SharedIdPub> shids; std::vector<Aabb> sh_aabb; std::vector<int32_t> sh_owner_key_idx; std::vector<uint64_t> sh_owner_key; std::vector<uint8_t> sh_cached;
    std::set<uint64_t> changed_cells, changed_static; std::vector<SharedIdPub> changed_shared; std::set<SharedIdPub> changed_shared_set;
    bool too_many = false;
    std::set<uint64_t> ghost_touched;                   // sections that received a ghost instance: their row segment is rewritten, nothing else changes
};

constexpr uint32_t NUM_FRAME_HEADERS = 3;   // see frame_header()
struct re_ctx;
static int flush_deferred_pack(re_ctx *c);
static int drain_other_lane(re_ctx *c);
static void free_second_lane(re_ctx *c);
static void comm_release(re_ctx *c);
struct re_ctx {
    re_config cfg{};
    int device = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev[6] = {};
    std::string err;
    uint64_t dev_bytes = 0;
    uint32_t maxlevel = 0;

    // rows.  n = rows in use (every entity ever registered: removed ones keep their row, marked F_DEAD); row_cap = rows the per-entity columns hold
    // (grown by ensure_row_capacity when entities are added after the upload); the ghost instances of the frozen static cache live in rows
    // [ghost_base, ghost_base + ghost_cap) of the id / matrix columns, ghost_base == row_cap.  n_base / ndyn0: rows / dynamic entities of the upload
    // (dynamic entity j < ndyn0 lives in row j; later ones sit in any row, listed in d_dyn_row), dyn_cap: slots of the dynamic table.
    uint32_t n = 0, ndyn = 0, row_cap = 0, ghost_base = 0, n_base = 0, ndyn0 = 0, dyn_cap = 0;
    std::unordered_map<uint32_t, uint32_t> id_extra;     // entities added after the upload (and ids reused after a removal): id -> row
    std::unordered_map<uint32_t, uint32_t> dyn_extra;    // rows that got a slot of the dynamic table after the upload: row -> slot
    std::unordered_map<GroupKey, uint32_t, GroupKeyHash> gmap;   // (ModelId, sortable) -> group class
    uint32_t nslots_cap = 0;                             // InstanceRange capacity of the host result block(s)
    struct AddKeys { uint8_t nk; bool is_static; uint64_t keys[8]; };
    std::unordered_map<uint32_t, AddKeys> add_keys;      // rows an add batch created, with their section decision: consumed by rebucket() (TreeOp kind 4)
    bool dyn_index(uint32_t r, uint32_t &j) const {      // slot of row r in the dynamic table
        if (r < ndyn0) { j = r; return true; }
        auto e = dyn_extra.find(r); if (e == dyn_extra.end()) return false;
        j = e->second; return true;
    }
    DevBuf<uint32_t> d_id, d_gclass, d_flags, d_row_cell;
    DevBuf<float> d_mat, d_pos, d_rot, d_scale;
    DevBuf<Aabb> d_aabb, d_orig;
    DevBuf<uint32_t> d_dyn_row, d_dyn_cell; DevBuf<float> d_dyn_vel, d_dyn_acc, d_dyn_rotvel, d_dyn_rotacc;
    DevBuf<uint64_t> d_row_key; DevBuf<uint8_t> d_row_nk; DevBuf<SharedRec> d_shrec; DevBuf<uint32_t> d_counter;
    std::vector<uint32_t> h_id, h_flags, h_dyn_row;      // host mirrors of the immutable id column / upload flags / dynamic-row list
    uint32_t n_rebuilds = 0;
    bool has_rotvel = false; uint32_t n_dead = 0;         // n_dead: rows removed by out-of-bounds ticks (upper bound on uncounted reservations)
    bool ids_identity = false;                           // entity id == row index (dense ids): no lookup table needed
    std::vector<uint32_t> id_to_row;                     // otherwise, ids below 4n: direct table (rows are not in id order once the dynamic entities lead)
    std::vector<std::pair<uint32_t, uint32_t>> id_rows;  // otherwise (id, row) sorted by id
    bool row_of(uint32_t id, uint32_t *row) const {
        if (!id_extra.empty()) { auto e = id_extra.find(id); if (e != id_extra.end()) { *row = e->second; return true; } }
        if (ids_identity) { if (id >= n_base) return false; *row = id; return true; }
        if (!id_to_row.empty()) { if (id >= id_to_row.size() || id_to_row[id] == 0xFFFFFFFFu) return false; *row = id_to_row[id]; return true; }
        auto p = std::lower_bound(id_rows.begin(), id_rows.end(), std::make_pair(id, 0u));
        if (p == id_rows.end() || p->first != id) return false;
        *row = p->second; return true;
    }
    // sections
    uint32_t ncells = 0, nsh = 0, nrows_csr = 0, n_real_sections = 0;   // ncells counts padding slots too
    DevBuf<uint64_t> d_cell_key; DevBuf<Aabb> d_cell_tight; DevBuf<uint32_t> d_cell_begin, d_cell_nlocal, d_cell_nstatic, d_cell_stamp, d_rows;
    DevBuf<uint8_t> d_cell_flags;
    DevBuf<int32_t> d_sh_cells, d_sh_owner; DevBuf<Aabb> d_sh_aabb; DevBuf<uint32_t> d_sh_begin, d_sh_nact, d_sh_nstat; DevBuf<uint8_t> d_sh_cached, d_sh_dirty;
    std::vector<uint64_t> h_cell_key;
    // host copies the incremental re-bucket needs: per-row section decision, and the shared-section ids in device order
    std::vector<uint64_t> h_row_key; std::vector<uint8_t> h_row_nk; std::vector<uint32_t> h_row_cell, h_gclass;
    std::unordered_map<uint32_t, std::array<uint64_t, 8>> h_row_shared_keys;
    std::vector<SharedIdPub> h_shids; std::vector<uint32_t> h_sh_nact, h_sh_nstat; std::vector<int32_t> h_sh_cells;   // (h_sh_cells: the slots each shared section links, [nsh][8])
    bool dirty_pending = false, timings_pending = false;
    // host mirrors of the section table for incremental patches (see patch_sections)
    std::vector<uint32_t> h_cell_nl, h_cell_ns, h_cell_begin, h_cell_cap, h_rows;   // per slot; h_rows mirrors the row pool
    std::vector<uint64_t> base_keys;                    // slot keys of the last full build (sorted: the lookup base and the span hint)
    std::vector<uint64_t> base_index;                   // every 1024th base key: the cache-resident first level of the lookup
    DevBuf<uint8_t> d_stage;                            // staging area of patch uploads
    std::unordered_map<uint64_t, uint32_t> extra_slots;  // sections created since, key -> slot
    std::vector<std::vector<uint32_t>> free_slots;       // per level: padding / emptied slots a new section of that level may take
    uint32_t pool_used = 0, pool_cap = 0, n_patches = 0;
    // ghost instances (snapshot copies of the frozen static render cache): rows n .. n + n_ghost of the id / matrix columns
    uint32_t ghost_cap = 0, n_ghost = 0; std::vector<uint32_t> h_ghost_gc; std::map<uint64_t, std::vector<uint32_t>> ghost_map;   // per section key: its ghost rows
    DevBuf<uint32_t> d_cell_nghost; std::vector<uint32_t> h_cell_ng;
    // collision broad phase (re_collide): lists allocated at the first call
    uint32_t user_row = ROW_CELL_NONE; DevBuf<ColHeader> d_col_hdr; DevBuf<ColRegion> d_col_region; DevBuf<uint32_t> d_col_high; DevBuf<ColShared> d_col_shared; DevBuf<ColMoved> d_col_moved; DevBuf<ColNear> d_col_near; ColHeader *h_col = nullptr, *d_hcol = nullptr; uint32_t col_calls = 0;
    DevBuf<uint8_t> d_row_moved; DevBuf<unsigned long long> d_col_tab; DevBuf<uint2> d_col_pairs; uint32_t col_moved_cap = 0, col_tab_size = 0, col_pair_cap = 0;
    DevBuf<HashEntry> d_htab; uint32_t htab_mask = 0, htab_keys = 0; uint32_t probe_frames = 0;   // RE_CFG_PROBE: key -> slot table of the probe path (k_probe_cull)
    std::set<uint64_t> dormant_cached;                  // sections with ghosts that were cached when they were emptied: the reference's cache entry outlives the section and shows again when the section is re-created
    std::set<uint32_t> h_uncached;                       // rows made static after the static render cache froze: in the tree's static sets, not drawn
    // groups
    uint32_t ngclass = 0, nslots = 0;
    DevBuf<uint32_t> d_gc_model, d_gc_rs, d_gc_sort, d_group_count, d_group_begin, d_group_fill;
    // large visible sets, group tables of <= COUNT_SLOTS_MAX slots: instance counts / running fills per (cursor shard, group slot), two parities alternating by
    // large-pack frame.  k_pack_large of one frame clears the other parity's arrays for the next; `dirty` tracks arrays that hold something nobody will clear.
    DevBuf<uint32_t> d_gcount, d_gfill; bool gc_dirty[2] = { false, false }; uint32_t large_seq = 0;
    // per-model level-of-view bands (level_views.custom): kept across uploads, like a model registration; device tables rebuilt when they or the group classes change
    struct CustomLod { uint32_t model, rs, n; float lmin[8], lmax[8]; };
    std::vector<CustomLod> custom_lod; std::vector<GroupKey> h_gkeys;
    DevBuf<uint32_t> d_gc_lodtab, d_lod_n; DevBuf<float> d_lod_min, d_lod_max; bool lod_tables_on = false;
    // frame
    uint32_t frame = 0; bool have_cull = false;
    DevBuf<uint32_t> d_rows_gc; std::vector<uint32_t> h_sh_begin;   // group class per row-pool entry; pool offsets of the shared sections' members
    DevBuf<uint32_t> d_cell_key32, d_chunk_level; bool key32 = false; PBox32Table PB32{};   // compact keys for the stream (worlds of <= 512 sections per axis)
    FrameParams P{}; PBoxTable PB{}; DevBuf<FrameParams> d_params;
#ifdef RE_EXP_STAMPS
    DevBuf<unsigned long long> d_timeline;
#endif
    uint32_t nlists = 0, pred_candidates = 0;             // nlists: 512-key chunks == waves of k_scan_cull
    uint32_t item_cap = 0, out_cap = 0, list_cap = 0;
    DevBuf<uint32_t> d_item_row, d_item_slot, d_out_ids; DevBuf<float> d_out_mats;
    uint32_t *ext_out_ids = nullptr; float *ext_out_mats = nullptr; uint32_t ext_out_cap = 0; uint32_t *ext_out_count = nullptr;
    DevBuf<FrameHeader> d_hdr; DevBuf<TickHeader> d_th; DevBuf<uint32_t> d_movers, d_oob;
    // results land in mapped pinned host memory, written directly by the kernels (d_* = device view)
    SpecState *h_spec = nullptr, *d_hspec = nullptr; DevBuf<SpecState> d_spec;     // cross-frame speculation (see SpecState)
    struct PendingCall { uint8_t kind; uint32_t frame; re_camera cam; uint32_t flags; float dt; uint32_t *out_ids; float *out_mats; uint32_t out_cap; uint32_t *out_count; };   // kind 0 = cull_pack, 1 = tick
    std::vector<uint32_t> h_oob_ids;                    // entities removed because they left the world, since the last re_get_out_of_bounds
    std::vector<PendingCall> pending;                   // calls enqueued since the last resolved synchronisation, in order
    HostResult *h_res = nullptr, *d_hres = nullptr; InstanceRange *h_ranges = nullptr, *d_hranges = nullptr; TickHeader *h_th = nullptr, *d_hth = nullptr;
    void *h_block = nullptr;                           // the one mapped, coherent host block those four live in (alloc_host_block)
    uint32_t n_seal_waits = 0, n_sync_fallbacks = 0;  // a polled block whose seal did not agree at first sight / that needed a stream synchronise: both stay 0 when the publication protocol holds
    bool th_clean = true; uint32_t pred_total = 0;
    std::vector<re_instance_range> groups_out;
    bool cull_inflight = false, tick_inflight = false;
    bool tick_published = false;                      // the tick in flight publishes its counters itself (synchronous ticks; asynchronous ones are settled by resolve())
    uint32_t tick_frame = 0xFFFFFFFFu;                // frame of the last tick issued
    uint32_t tick_seq = 0;                            // synchronous ticks issued with a kernel: the last wave of k_tick writes the number into h_th->ticket (tick_sign_off)
    bool timings_on = false;                          // re_get_timings was asked for: synchronous frames record their kernel events (5 event records cost ~12 us per frame)
    bool deferred_pack = false; FusedPack deferred{}; uint32_t deferred_grid = 0, n_fused_frames = 0;
    // RE_CULL_TWO_LANES: a second set of per-frame resources ("lane": stream, frame headers, instance lists, section stamps, frame
    // parameters, packed output, result block, deferred pack).  The members above/below always are those of the CURRENT lane; the other
    // lane's are parked here and swapped in by switch_lane().  Frames of a static fly-through alternate lanes, so the launches of
    // consecutive frames sit on different streams and overlap; everything else first drains the other lane (drain_other_lane).
    struct LanePark {
        hipStream_t stream = nullptr; DevBuf<FrameHeader> d_hdr; DevBuf<uint32_t> d_item_row, d_item_slot, d_out_ids, d_cell_stamp; DevBuf<float> d_out_mats; DevBuf<FrameParams> d_params;
        HostResult *h_res = nullptr, *d_hres = nullptr; InstanceRange *h_ranges = nullptr, *d_hranges = nullptr; void *h_block = nullptr;
        bool deferred_pack = false; FusedPack deferred{}; uint32_t deferred_grid = 0, lane_seq = 0; bool busy = false;
    } park;
    bool park_ready = false, lane_busy = false; uint32_t lane_seq = 0, lane_id = 0, n_lane_switches = 0;   // RE_CULL_DEFER_PACK: the pack of the last frame, waiting for the next launch
    re_tick_result last_tick{};
    // The instance list is CURSOR_SHARDS segments with one cursor each, and a wave reserves in segment (wave index mod 8): sections that sit in few
    // waves (one crowded section, a cluster) can fill one segment while the list as a whole has room.  The pack kernels report that
    // (RESULT_SEGMENT_OVERFLOW); finish_cull then redoes the frame with the list as ONE segment -- which holds every instance of the world twice
    // (duplicates mode) by construction -- and the context stays in that mode until the next upload.
    uint64_t shard_lo = 0, shard_hi = 0; std::vector<uint32_t> moved_rows;   // re_set_shard_range: the keys this context owns; rows re-bucketed since the last re_list_migrants
    bool single_shard = false; uint32_t n_segment_redos = 0; re_camera last_cam{}; uint32_t last_cull_flags = 0;
    // what the last frame's pack was launched with, so that it can be run again into other output buffers (the second, variable-length round of the
    // multi-GPU exchange when a rank's visible set outgrew its slab)
    struct LastPack { int kind = 0; FrameHeader *hdr = nullptr, *hdr_next = nullptr; PackArgs A{}; ItemSink K{}; uint32_t grid = 0, nrows = 0, par = 0; PackLargeArgs L{}; } last_pack;   // kind: 1 k_pack_small, 2 k_pack_large, 3 count / scan / scatter
    // multi-GPU exchange (re_comm_*, re_allgather_visible): an RCCL communicator, two send slabs [4-word header | pad to 16 words | ids[cap] | matrices[cap * 16]]
    // alternating by frame, their receive buffers, and the buffers of the variable-length second round
    struct Comm {
        void *comm = nullptr; bool owned = false; int rank = 0, n = 1; uint32_t cap = 0, words = 0, seq = 0; int last = -1, pending = -1;
        DevBuf<uint32_t> slab[2], recv[2], big_ids; DevBuf<float> big_mats; uint32_t big_cap = 0;
        std::vector<uint32_t> counts; uint32_t n_second_rounds = 0, n_regathers = 0;
        uint32_t *h_hdr = nullptr, *d_hhdr = nullptr, hdr_seq = 0;          // the gathered slab headers in mapped host memory ([n ranks][4 words], then the sequence word k_gather_headers publishes)
    } comm;
    float t_cull = 0, t_pack = 0, t_tick = 0; bool timed_frame = false, timed_tick = false;
    // device-side re-bucket bookkeeping (rebucket_on_device): lookup tables, scratch, and the sections whose host mirrors are behind the device
    DevBuf<uint32_t> d_cell_cap; DevBuf<uint8_t> d_cell_links; std::vector<uint32_t> h_linked_slots;   // (links: shared sections linking each unique section; non-zero keeps a batch on the host)
    DevBuf<uint64_t> d_base_keys; DevBuf<unsigned long long> d_ovl_keys; DevBuf<uint32_t> d_ovl_slots; uint32_t ovl_cap = 0, ovl_count = 0;
    bool rb_base_dirty = true, rb_ovl_dirty = true;
    uint8_t *h_chg = nullptr, *d_chg = nullptr; DevBuf<WriteOp> d_chg_ops; DevBuf<uint32_t> d_chg_list;   // change batches: mapped staging block of k_apply_small; device scratch of larger batches
    DevBuf<uint32_t> d_hrb_list; DevBuf<uint8_t> d_hrb_nk; DevBuf<uint64_t> d_hrb_keys;   // scratch of the host-path re-bucket (the movers' new section decisions)
    // shared world sections on the device path: the table has sh_cap entries with STABLE indices (a retired entry is a hole until a host path rebuilds the
    // table compactly); ids, row capacities and the id -> index hash are device-only, rebuilt from the host mirrors whenever a host path has touched the table
    uint32_t sh_cap = 0, sh_hmask = 0, sh_hash_used = 0; bool rb_sh_dirty = true;   // (sh_hash_used: entries of the id -> index hash, retired ones included)
    DevBuf<uint64_t> d_sh_keys; DevBuf<uint8_t> d_sh_nk, d_cell_inact; DevBuf<uint32_t> d_sh_rowcap, d_sh_hidx; DevBuf<unsigned long long> d_sh_hkeys;
    std::vector<uint32_t> sh_free, stale_shared;        // holes of the table (indices a new shared section may take); entries the device changed since the host mirrors were brought up to date
    struct Rb2Scratch {
        uint32_t cap = 0;                                 // movers the buffers are sized for
        DevBuf<uint64_t> key, key2, ord, ord_s, kgath, ksorted1, ksorted2, mk, pair_key, pair_key_s; DevBuf<uint32_t> row, idx, perm_a, perm1, perm2, host_list, refold, tmp_u, tmp_s, free_u, free_off, free_s, pair_seg, pair_seg_s;
        DevBuf<uint8_t> mnk, tmp; DevBuf<Rb2Seg> segs_u; DevBuf<Rb2ShSeg> segs_s; DevBuf<Rb2Status> status;
        // pinned host staging of everything that travels between the host and the device inside one batch (status blocks, free slots, the segments' outcome): copies from / to
        // pageable memory cost 15-20 us each, and a batch made eight of them -- most of a small batch's time
        uint8_t *pin = nullptr, *d_pin = nullptr; size_t pin_bytes = 0;     // (d_pin: the device's address of the mapped block)
        bool status_clean = false; uint32_t status_pool = 0;                  // the device status block was reset by the last batch's final read-back and holds this pool fill
        Rb2Status *h_status = nullptr, *d_h_status = nullptr; uint32_t *h_seq = nullptr, *d_h_seq = nullptr, seq = 0;      // (d_h_*: the device's view of the mapped block)
        uint32_t *h_free_off = nullptr, *h_free_u = nullptr, *h_free_s = nullptr, *h_keep = nullptr; Rb2Seg *h_segs_u = nullptr; Rb2ShSeg *h_segs_s = nullptr;
    } rb2;
    std::vector<uint32_t> stale_slots;                   // sections patched on the device since the host mirrors (h_cell_*, h_rows, h_row_*, extra_slots) were last brought up to date
    uint32_t n_device_rebuckets = 0, n_host_rebuckets = 0, n_phantom = 0, last_added_rejected = 0, slack_boost = 1;
    std::vector<uint32_t> h_light_rows; DevBuf<uint32_t> d_light_rows, d_light_out; bool light_rows_dirty = true;   // rows that carry a FindLightType (members of their section's light set)
    std::vector<hipEvent_t> k1_events; uint32_t k1_used = 0, k1_every = 1, k1_seen = 0, k1_kind = 0; bool k1_timing = false;   // per-launch timing of one kernel (re_timing_begin): k_scan_cull, k_tick or k_pack_large

    int fail(int code, const char *fmt, ...) {
        char buf[512]; va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
        err = buf; return code;
    }
};

#define HIPCHK(ctx, call) do { hipError_t e_ = (call); if (e_ != hipSuccess) return (ctx)->fail(RE_E_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); } while (0)

// Every host wait on a stream of the context has a deadline: a kernel that never finishes (a wave spinning on a condition no other wave will
// satisfy, a loop whose stride came out as zero) must surface as RE_E_HIP from the entry point -- "no aborts cross the ABI" also means the caller
// is not left blocked inside the library until something outside kills the process (the likeliest reading of round 2's unexplained
// `Fatal Python error: Aborted` inside re_cull_pack: a work-in-progress kernel that hung and an external `timeout` that ended the run with
// SIGABRT).  RE_SYNC_TIMEOUT_MS overrides the 60 s default; 0 = wait without a deadline.
static hipError_t sync_stream(hipStream_t st) {
    static const long limit_ms = [] { const char *e = getenv("RE_SYNC_TIMEOUT_MS"); return e ? atol(e) : 60000L; }();
    if (limit_ms <= 0) return hipStreamSynchronize(st);
    hipError_t e = hipStreamQuery(st);
    if (e != hipErrorNotReady) return e;
    const auto t0 = std::chrono::steady_clock::now();
    for (uint32_t spins = 0;; spins++) {
        e = hipStreamQuery(st);
        if (e != hipErrorNotReady) return e;
        if ((spins & 63u) == 63u) {
            const auto dt = std::chrono::steady_clock::now() - t0;
            if (dt > std::chrono::milliseconds(limit_ms)) return hipErrorLaunchTimeOut;
            if (dt > std::chrono::milliseconds(2)) std::this_thread::sleep_for(std::chrono::microseconds(50));   // a long wait (an upload, a rebuild): stop burning the core
        }
    }
}

extern "C" uint32_t re_abi_version(void) { return 3u; }

// RE_EXP_TIME_ISSUE=1: host time of a frame's phases (printed by re_destroy): what the calling thread spends per frame outside the kernels
namespace {
struct IssueClock {
    bool on = getenv("RE_EXP_TIME_ISSUE") != nullptr; uint64_t n = 0; double params = 0, spans = 0, scan = 0, pack = 0, wait = 0, finish = 0;
    static double now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
    void report() const { if (on && n) fprintf(stderr, "host us per frame over %llu frames: parameters %.2f  spans %.2f  scan launch %.2f  pack launch %.2f  wait for the result %.2f  finish %.2f\n",
                                                 (unsigned long long)n, params / n, spans / n, scan / n, pack / n, wait / n, finish / n); }
};
IssueClock g_issue_clock;
}
static void report_issue_clock() { g_issue_clock.report(); g_issue_clock.n = 0; g_issue_clock.params = g_issue_clock.spans = g_issue_clock.scan = g_issue_clock.pack = g_issue_clock.wait = g_issue_clock.finish = 0; }

// Everything the kernels publish to the polling host thread lives in ONE block of mapped, coherent pinned host memory per frame lane
// (hipHostMallocMapped | hipHostMallocCoherent): the frame result at 0, the speculation word at 128, the tick counters at 256, the
// InstanceRange table from 8192 on.  (Round 1 used four separate hipHostMalloc(Mapped) blocks; what made its group table arrive after
// the "frame done" word was not the number of blocks but the plain store of that word -- see publish_to_host in re_kernels.h.)
constexpr size_t HB_RES = 0, HB_SPEC = 128, HB_TICK = 256, HB_RANGES = 8192;
static_assert(sizeof(HostResult) <= HB_SPEC && sizeof(SpecState) <= HB_TICK - HB_SPEC && HB_TICK + sizeof(TickHeader) <= HB_RANGES, "host block layout");
static hipError_t alloc_host_block(uint32_t nslots, void **host, void **dev) {
    const size_t bytes = HB_RANGES + sizeof(InstanceRange) * (size_t)std::max(nslots, 1u);
    hipError_t e = hipHostMalloc(host, bytes, hipHostMallocMapped | hipHostMallocCoherent);
    if (e != hipSuccess) return e;
    memset(*host, 0, bytes);
    return hipHostGetDevicePointer(dev, *host, 0);
}
template <typename T> static inline T *hb_at(void *base, size_t off) { return reinterpret_cast<T *>(static_cast<char *>(base) + off); }

extern "C" const char *re_last_error(const re_ctx *ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

extern "C" int re_create(const re_config *cfg, re_ctx **out) try {
    if (!cfg || !out) { g_create_error = "re_create: null argument"; return RE_E_ARG; }
    if (cfg->atomic_length == 0 || cfg->outline_length < cfg->atomic_length) { g_create_error = "re_create: outline_length must be >= atomic_length > 0"; return RE_E_ARG; }
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev == 0) { g_create_error = std::string("re_create: no HIP device available (") + hipGetErrorString(e) + "); this library has no CPU path"; return RE_E_HIP; }
    if (cfg->device < 0 || cfg->device >= ndev) { g_create_error = "re_create: device ordinal out of range"; return RE_E_ARG; }
    re_ctx *c = new re_ctx();
    c->cfg = *cfg; c->device = cfg->device;
    c->maxlevel = max_level(cfg->outline_length, cfg->atomic_length);
    if (c->maxlevel > (uint32_t)MAX_LEVELS || cfg->outline_length / cfg->atomic_length > 32768u) { g_create_error = "re_create: outline/atomic must give at most 16 levels and 32768 sections per axis"; delete c; return RE_E_ARG; }
    if ((e = hipSetDevice(c->device)) != hipSuccess || (e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking)) != hipSuccess) {
        g_create_error = std::string("re_create: ") + hipGetErrorString(e); delete c; return RE_E_HIP;
    }
    for (auto &ev : c->ev) (void)hipEventCreate(&ev);
    *out = c;
    return RE_OK;
} RE_ABI_GUARD_NOCTX(g_create_error, "re_create")

static void release_rb2(re_ctx *c) {
    re_ctx::Rb2Scratch &B = c->rb2;
    for (DevBuf<uint64_t> *b : { &B.key, &B.key2, &B.ord, &B.ord_s, &B.kgath, &B.ksorted1, &B.ksorted2, &B.mk, &B.pair_key, &B.pair_key_s }) b->release(nullptr);
    for (DevBuf<uint32_t> *b : { &B.row, &B.idx, &B.perm_a, &B.perm1, &B.perm2, &B.host_list, &B.refold, &B.tmp_u, &B.tmp_s, &B.free_u, &B.free_off, &B.free_s, &B.pair_seg, &B.pair_seg_s }) b->release(nullptr);
    B.mnk.release(nullptr); B.tmp.release(nullptr); B.segs_u.release(nullptr); B.segs_s.release(nullptr); B.status.release(nullptr); B.cap = 0;
    if (B.pin) { (void)hipHostFree(B.pin); B.pin = nullptr; B.pin_bytes = 0; }
}
static void free_world(re_ctx *c) {
    uint64_t *a = &c->dev_bytes;
    c->d_light_rows.release(nullptr); c->d_light_out.release(nullptr); c->light_rows_dirty = true;
    c->d_cell_cap.release(a); c->d_cell_links.release(a); c->h_linked_slots.clear(); c->d_base_keys.release(a); c->d_ovl_keys.release(nullptr); c->d_ovl_slots.release(nullptr); c->rb_base_dirty = c->rb_ovl_dirty = true; c->stale_slots.clear();
    c->d_hrb_list.release(nullptr); c->d_hrb_nk.release(nullptr); c->d_hrb_keys.release(nullptr); c->d_chg_ops.release(nullptr); c->d_chg_list.release(nullptr);
    if (c->h_chg) { (void)hipHostFree(c->h_chg); c->h_chg = nullptr; c->d_chg = nullptr; }
    c->d_sh_keys.release(nullptr); c->d_sh_nk.release(nullptr); c->d_cell_inact.release(nullptr); c->d_sh_rowcap.release(nullptr); c->d_sh_hidx.release(nullptr); c->d_sh_hkeys.release(nullptr);
    c->sh_cap = 0; c->rb_sh_dirty = true; c->sh_free.clear(); c->stale_shared.clear(); release_rb2(c);
    c->d_id.release(a); c->d_gclass.release(a); c->d_flags.release(a); c->d_row_cell.release(a); c->d_mat.release(a); c->d_pos.release(a); c->d_rot.release(a);
    c->d_scale.release(a); c->d_aabb.release(a); c->d_orig.release(a); c->d_dyn_row.release(a); c->d_dyn_cell.release(a); c->d_dyn_vel.release(a); c->d_dyn_acc.release(a);
    c->d_dyn_rotvel.release(a); c->d_dyn_rotacc.release(a); c->d_row_key.release(a); c->d_row_nk.release(a); c->d_shrec.release(a); c->d_counter.release(a);
    c->d_cell_key.release(a); c->d_cell_tight.release(a); c->d_cell_begin.release(a); c->d_cell_nlocal.release(a); c->d_cell_nstatic.release(a);
    c->d_cell_stamp.release(a); c->d_rows.release(a); c->d_cell_flags.release(a); c->d_sh_cells.release(a); c->d_sh_owner.release(a); c->d_sh_aabb.release(a);
    c->d_sh_begin.release(a); c->d_sh_nact.release(a); c->d_sh_nstat.release(a); c->d_sh_cached.release(a); c->d_sh_dirty.release(a);
    c->d_gc_model.release(a); c->d_gc_rs.release(a); c->d_gc_sort.release(a); c->d_group_count.release(a); c->d_group_begin.release(a); c->d_group_fill.release(a);
    c->d_gcount.release(a); c->d_gfill.release(a); c->gc_dirty[0] = c->gc_dirty[1] = false;
    c->d_item_row.release(a); c->d_item_slot.release(a); c->d_out_ids.release(a); c->d_out_mats.release(a);
    c->d_hdr.release(a); c->d_th.release(a); c->d_params.release(a); c->d_movers.release(a); c->d_oob.release(a);
    if (c->h_block) { (void)hipHostFree(c->h_block); c->h_block = nullptr; }      // one block: frame result, speculation word, tick counters, group table
    c->h_res = nullptr; c->h_ranges = nullptr; c->h_th = nullptr; c->h_spec = nullptr;
    c->d_spec.release(nullptr); c->pending.clear();
    c->single_shard = false; c->id_extra.clear(); c->dyn_extra.clear(); c->gmap.clear(); c->add_keys.clear();
    c->row_cap = c->ghost_base = c->n_base = c->ndyn0 = c->dyn_cap = 0;
    c->n = c->ndyn = c->ncells = c->nsh = 0; c->have_cull = false; c->cull_inflight = c->tick_inflight = false; c->deferred_pack = false;
}

extern "C" void re_destroy(re_ctx *c) try {
    if (!c) return;
    report_issue_clock();
    (void)hipSetDevice(c->device);
    if (c->park_ready) { (void)drain_other_lane(c); free_second_lane(c); }
    if (c->stream) (void)sync_stream(c->stream);
    if (c->comm.comm) comm_release(c);
    free_world(c);
    if (c->h_col) { (void)hipHostFree(c->h_col); c->h_col = nullptr; }        // lives with the collision scratch lists (kept across uploads)
    for (auto &ev : c->ev) if (ev) (void)hipEventDestroy(ev);
    for (auto &ev : c->k1_events) (void)hipEventDestroy(ev);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
} catch (...) {}

static RowArrays row_arrays(re_ctx *c) {
    RowArrays R; R.id = c->d_id.p; R.gclass = c->d_gclass.p; R.flags = c->d_flags.p; R.mat = c->d_mat.p; R.aabb = c->d_aabb.p; R.orig = c->d_orig.p;
    R.pos = c->d_pos.p; R.rot = c->d_rot.p; R.scale = c->d_scale.p; R.key = c->d_row_key.p; return R;
}

// group class a row-pool entry carries for row r: hidden while the row is not to be drawn (removed, or made static after the cache froze)
static inline uint32_t effective_gclass(const re_ctx *c, uint32_t r) {
    if (r >= c->ghost_base) return c->h_ghost_gc[r - c->ghost_base];          // a ghost instance keeps the group class it was cloned with
    return ((c->h_flags[r] & (F_DEAD | F_PHANTOM)) || c->h_uncached.count(r)) ? 0xFFFFFFFFu : c->h_gclass[r];      // (a halo replica is never drawn here)
}
// pool positions of row r (one: its section's segment or its shared section's) get the row's current effective group class
static void collect_row_gc(const re_ctx *c, uint32_t r, std::vector<Pair32> &out) {
    const uint32_t rc = c->h_row_cell[r];
    if (rc == ROW_CELL_NONE) return;
    uint32_t b, n;
    if (rc & ROW_CELL_SHARED) { const uint32_t s2 = rc & ~ROW_CELL_SHARED; if (s2 >= c->h_sh_begin.size()) return; b = c->h_sh_begin[s2]; n = c->h_sh_nact[s2] + c->h_sh_nstat[s2]; }
    else { b = c->h_cell_begin[rc]; n = c->h_cell_nl[rc] + c->h_cell_ns[rc]; }
    for (uint32_t i = 0; i < n; i++) if (c->h_rows[b + i] == r) out.push_back(Pair32{ b + i, effective_gclass(c, r) });
}
static int upload_row_gc(re_ctx *c, const std::vector<Pair32> &pairs) {
    if (pairs.empty()) return RE_OK;
    Pair32 *d = nullptr; HIPCHK(c, hipMalloc(reinterpret_cast<void **>(&d), pairs.size() * sizeof(Pair32)));
    HIPCHK(c, hipMemcpyAsync(d, pairs.data(), pairs.size() * sizeof(Pair32), hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(k_scatter32, dim3(((uint32_t)pairs.size() + 255) / 256), dim3(256), 0, c->stream, (uint32_t)pairs.size(), d, c->d_rows_gc.p);
    HIPCHK(c, sync_stream(c->stream)); (void)hipFree(d);
    return RE_OK;
}

// (the dynamic entities are the first ndyn rows: the tick reads row_cell[j] itself, there is no separate copy to keep in step)
static int upload_dyn_cells(re_ctx *) { return RE_OK; }

// ------------------------------------------------------------------------------------------------
// World-section structure from the per-row section keys (the spatial hash as key-sorted arrays).
// Bulk semantics of one registration batch followed by BoundingBoxTree::end_of_changes.
// ------------------------------------------------------------------------------------------------
static inline uint32_t to_key32(uint64_t k) {
    if ((k & 0xFFFFFFFFFFFFull) == 0xFFFFFFFFFFFFull) return KEY32_PAD | 0x1FF7FDFFu;      // padding slot
    return ((uint32_t)key_x(k) << 20) | ((uint32_t)key_z(k) << 10) | (uint32_t)key_y(k);
}

namespace {
struct SortRec { uint64_t key; uint64_t sub; uint32_t row; };   // sub = static << 32 | entity id
using SharedId = SharedIdPub;
}

// The shared-section arrays hold sh_cap entries (the table itself [0, nsh), holes included); grown, never shrunk.  Callers re-upload the table afterwards.
static int ensure_shared_capacity(re_ctx *c, uint32_t need) {
    if (need <= c->sh_cap && c->d_sh_cells.p) return RE_OK;
    uint64_t *acct = &c->dev_bytes;
    const uint32_t cap = std::max(std::max(need * 2u + 1024u, c->sh_cap), std::min(c->ndyn, 1u << 20));      // (every dynamic entity may come to straddle a section border: room for the device path to create its shared section, ~200 bytes an entry)
    HIPCHK(c, sync_stream(c->stream));
    HIPCHK(c, c->d_sh_cells.alloc((size_t)cap * 8, acct)); HIPCHK(c, c->d_sh_owner.alloc(cap, acct)); HIPCHK(c, c->d_sh_aabb.alloc(cap, acct));
    HIPCHK(c, c->d_sh_begin.alloc(cap, acct)); HIPCHK(c, c->d_sh_nact.alloc(cap, acct)); HIPCHK(c, c->d_sh_nstat.alloc(cap, acct));
    HIPCHK(c, c->d_sh_cached.alloc(cap, acct)); HIPCHK(c, c->d_sh_dirty.alloc(cap, acct));
    HIPCHK(c, c->d_sh_keys.alloc((size_t)cap * 8, nullptr)); HIPCHK(c, c->d_sh_nk.alloc(cap, nullptr)); HIPCHK(c, c->d_sh_rowcap.alloc(cap, nullptr));
    uint32_t hs = 1024; while (hs < 4u * cap) hs <<= 1;
    HIPCHK(c, c->d_sh_hkeys.alloc(hs, nullptr)); HIPCHK(c, c->d_sh_hidx.alloc(hs, nullptr)); c->sh_hmask = hs - 1u;
    c->sh_cap = cap; c->rb_sh_dirty = true;
    return RE_OK;
}
static int build_sections(re_ctx *c, const std::vector<uint64_t> &row_key, const std::vector<uint8_t> &row_nk, std::vector<SharedRec> &shrec,
                          const std::vector<uint32_t> &flags, const Carry *carry = nullptr) {
    const uint32_t n = c->n;
    uint64_t *acct = &c->dev_bytes;
    // --- unique rows sorted by (section key, static, entity id): CSR order == the oracle's iteration order
    std::vector<SortRec> recs; recs.reserve(n);
    for (uint32_t r = 0; r < n; r++) if (row_nk[r] == 1) recs.push_back({ row_key[r], ((uint64_t)((flags[r] & F_STATIC) ? 1 : 0) << 32) | c->h_id[r], r });
    auto cmp = [](const SortRec &a, const SortRec &b) { return a.key != b.key ? a.key < b.key : a.sub < b.sub; };
    if (!std::is_sorted(recs.begin(), recs.end(), cmp)) {
        // rows are the dynamic entities followed by the others, each part in upload order: two sorted runs in a key-ordered world
        auto mid = std::partition_point(recs.begin(), recs.end(), [&](const SortRec &a) { return a.row < c->ndyn0; });
        if (std::is_sorted(recs.begin(), mid, cmp) && std::is_sorted(mid, recs.end(), cmp)) std::inplace_merge(recs.begin(), mid, recs.end(), cmp);
        else std::sort(recs.begin(), recs.end(), cmp);
    }
    // --- shared sections, indexed by first appearance in row order (== creation order of a fresh tree)
    std::sort(shrec.begin(), shrec.end(), [](const SharedRec &a, const SharedRec &b) { return a.row < b.row; });
    std::map<SharedId, uint32_t> shmap; std::vector<SharedId> shids; std::vector<std::vector<uint32_t>> sh_act, sh_sta;
    if (carry) {                                                        // surviving shared sections keep their creation order
        std::set<SharedId> alive;
        for (const SharedRec &sr : shrec) { SharedId id; id.nk = sr.nk; memcpy(id.keys, sr.keys, sizeof id.keys); alive.insert(id); }
        for (const SharedId &id : carry->shids) if (alive.count(id)) { shmap.emplace(id, (uint32_t)shids.size()); shids.push_back(id); sh_act.emplace_back(); sh_sta.emplace_back(); }
    }
    for (const SharedRec &sr : shrec) {
        SharedId id; id.nk = sr.nk; memcpy(id.keys, sr.keys, sizeof id.keys);
        auto it = shmap.find(id); uint32_t s;
        if (it == shmap.end()) { s = (uint32_t)shids.size(); shmap.emplace(id, s); shids.push_back(id); sh_act.emplace_back(); sh_sta.emplace_back(); } else s = it->second;
        ((flags[sr.row] & F_STATIC) ? sh_sta[s] : sh_act[s]).push_back(sr.row);
    }
    const uint32_t nsh = (uint32_t)shids.size();
    auto by_id = [&](uint32_t a, uint32_t b) { return c->h_id[a] < c->h_id[b]; };
    for (uint32_t s = 0; s < nsh; s++) { std::sort(sh_act[s].begin(), sh_act[s].end(), by_id); std::sort(sh_sta[s].begin(), sh_sta[s].end(), by_id); }
    // --- section keys = keys of unique rows U keys linked by shared sections
    std::vector<uint64_t> keys; keys.reserve(recs.size());
    for (size_t i = 0; i < recs.size(); i++) if (i == 0 || recs[i].key != recs[i - 1].key) keys.push_back(recs[i].key);
    if (nsh) {
        std::vector<uint64_t> lk;
        for (const SharedId &id : shids) for (uint32_t k = 0; k < id.nk; k++) lk.push_back(id.keys[k]);
        std::sort(lk.begin(), lk.end()); lk.erase(std::unique(lk.begin(), lk.end()), lk.end());
        std::vector<uint64_t> merged; merged.reserve(keys.size() + lk.size());
        std::set_union(keys.begin(), keys.end(), lk.begin(), lk.end(), std::back_inserter(merged));
        keys.swap(merged);
    }
    // pad every level run to whole wave chunks of k_scan_cull (so the level is uniform inside a wave); a pad slot
    // carries the largest key of its level (x = z = y = 0xFFFF: never a world section, outline/atomic <= 32768)
    c->n_real_sections = (uint32_t)keys.size();
    bool has_movers = false; for (uint32_t r = 0; r < n && !has_movers; r++) has_movers = (flags[r] & F_HAS_VEL) != 0;
    // compact 32-bit stream keys when every section index fits 9 bits; a wave then owns 1024 keys, so level runs are padded to that
    c->key32 = (c->cfg.outline_length + c->cfg.atomic_length - 1) / c->cfg.atomic_length <= 512u && getenv("RE_EXP_KEY64") == nullptr;
    const size_t wave_keys = c->key32 ? WAVE_KEYS32 : WAVE_KEYS;
    {
        std::vector<uint64_t> padded; padded.reserve(keys.size() + MAX_LEVELS * WAVE_KEYS32);
        size_t i = 0;
        while (i < keys.size()) {
            uint32_t lv = key_level(keys[i]);
            const size_t run0 = padded.size();
            while (i < keys.size() && key_level(keys[i]) == lv) padded.push_back(keys[i++]);
            // spare slots per level run (sections created by re-bucket patches, change requests and added entities live there): at least one chunk for every world -- a table
            // without any made every section a change request created a full rebuild (1.4 s at 10 M sections: round 3, bench.py change_request) --, ~0.8 % of the run for worlds
            // with movers; doubled (slack_boost) every time a rebuild was forced by exhausted slack
            size_t spare = (c->cfg.flags & RE_CFG_TIGHT_SLACK) ? 0 : std::max<size_t>(wave_keys, has_movers ? (padded.size() - run0) / 128 : 0) * c->slack_boost;
            for (size_t k = 0; k < spare; k++) padded.push_back(pack_key(lv, 0xFFFFu, 0xFFFFu, 0xFFFFu));
            while (padded.size() % wave_keys) padded.push_back(pack_key(lv, 0xFFFFu, 0xFFFFu, 0xFFFFu));
        }
        keys.swap(padded);
    }
    auto is_pad = [](uint64_t k) { return (k & 0xFFFFFFFFFFFFull) == 0xFFFFFFFFFFFFull; };
    const uint32_t ncells = (uint32_t)keys.size();
    std::vector<uint32_t> begin(ncells + 1, 0), nlocal(ncells, 0), nstatic(ncells, 0), nghost(ncells, 0), rows; rows.reserve(n);
    std::vector<uint32_t> row_cell(n, ROW_CELL_NONE);
    {
        size_t i = 0;
        for (uint32_t ci = 0; ci < ncells; ci++) {
            begin[ci] = (uint32_t)rows.size();
            while (i < recs.size() && recs[i].key == keys[ci]) {
                if (recs[i].sub >> 32) nstatic[ci]++; else nlocal[ci]++;
                rows.push_back(recs[i].row); row_cell[recs[i].row] = ci; i++;
            }
            if (!c->ghost_map.empty()) { auto g = c->ghost_map.find(keys[ci]); if (g != c->ghost_map.end()) { nghost[ci] = (uint32_t)g->second.size(); for (uint32_t gr : g->second) rows.push_back(gr); } }
        }
        begin[ncells] = (uint32_t)rows.size();
    }
    std::vector<int32_t> sh_cells((size_t)nsh * 8 + 8, -1); std::vector<uint32_t> sh_begin(nsh + 1, 0), sh_nact(nsh + 1, 0), sh_nstat(nsh + 1, 0);
    std::vector<std::vector<uint32_t>> cell_links;                         // only for sections that link shared sections
    std::unordered_map<uint32_t, uint32_t> cell_link_idx;
    for (uint32_t s = 0; s < nsh; s++) {
        for (uint32_t k = 0; k < shids[s].nk; k++) {
            uint32_t ci = (uint32_t)(std::lower_bound(keys.begin(), keys.end(), shids[s].keys[k]) - keys.begin());
            sh_cells[(size_t)s * 8 + k] = (int32_t)ci;
            auto it = cell_link_idx.find(ci);
            if (it == cell_link_idx.end()) { cell_link_idx.emplace(ci, (uint32_t)cell_links.size()); cell_links.emplace_back(); it = cell_link_idx.find(ci); }
            cell_links[it->second].push_back(s);
        }
        sh_begin[s] = (uint32_t)rows.size(); sh_nact[s] = (uint32_t)sh_act[s].size(); sh_nstat[s] = (uint32_t)sh_sta[s].size();
        for (uint32_t r : sh_act[s]) { rows.push_back(r); row_cell[r] = ROW_CELL_SHARED | s; }
        for (uint32_t r : sh_sta[s]) { rows.push_back(r); row_cell[r] = ROW_CELL_SHARED | s; }
    }
    // --- update_static_world_sections (bounding_box_tree_v2.rs:1133-1213).  Fresh build: every section and shared section changed.
    // Incremental (carry): only the changed sections are recomputed, everything else keeps its previous flag.
    std::vector<uint8_t> cflags(ncells + 1, 0), refold(ncells + 1, 1); std::vector<Aabb> carried_tight(ncells + 1);
    auto loop1 = [&](uint32_t ci) {
        bool st = false;
        if (nlocal[ci] == 0) {
            auto it = cell_link_idx.find(ci);
            if (it == cell_link_idx.end()) st = true;
            else for (uint32_t s : cell_links[it->second]) if (sh_nact[s] == 0) st = true;
        }
        return st;
    };
    for (uint32_t ci = 0; ci < ncells; ci++) {
        if (is_pad(keys[ci])) { cflags[ci] = (uint8_t)(CF_PAD | CF_STATIC_SECTION); continue; }
        if (!carry) { cflags[ci] = (uint8_t)((loop1(ci) ? CF_STATIC_SECTION : 0) | CF_STATIC_DIRTY); continue; }
        auto old = std::lower_bound(carry->keys.begin(), carry->keys.end(), keys[ci]);
        bool existed = old != carry->keys.end() && *old == keys[ci];
        bool changed = carry->changed_cells.count(keys[ci]) != 0;
        uint8_t f = existed ? (uint8_t)(carry->flags[old - carry->keys.begin()] & (CF_STATIC_SECTION | CF_STATIC_CACHED | CF_STATIC_DIRTY)) : (uint8_t)0;
        if (!existed && c->dormant_cached.erase(keys[ci])) f |= CF_STATIC_CACHED;   // a re-created section whose cache entry (ghosts) survived it
        if (changed || !existed) f = (uint8_t)((f & ~CF_STATIC_SECTION) | (loop1(ci) ? CF_STATIC_SECTION : 0));
        if (carry->changed_static.count(keys[ci])) f |= CF_STATIC_DIRTY;
        cflags[ci] = f;
        if (existed && !changed) { refold[ci] = 0; carried_tight[ci] = carry->tight[old - carry->keys.begin()]; }
    }
    if (carry && !c->ghost_map.empty())                                      // sections that disappear with this rebuild while their cache entry holds ghosts
        for (size_t i = 0; i < carry->keys.size(); i++)
            if ((carry->flags[i] & CF_STATIC_CACHED) && c->ghost_map.count(carry->keys[i]) && !std::binary_search(keys.begin(), keys.end(), carry->keys[i])) c->dormant_cached.insert(carry->keys[i]);
    std::vector<uint32_t> sh_order(nsh); for (uint32_t s = 0; s < nsh; s++) sh_order[s] = s;
    std::sort(sh_order.begin(), sh_order.end(), [&](uint32_t a, uint32_t b) { return shids[a] < shids[b]; });   // canonical id order
    for (uint32_t s : sh_order) {
        if (carry && !carry->changed_shared_set.count(shids[s])) continue;      // second loop: changed shared sections only
        for (uint32_t k = 0; k < shids[s].nk; k++) {
            uint32_t ci = (uint32_t)sh_cells[(size_t)s * 8 + k];
            if (sh_nact[s] == 0) { if (nlocal[ci] == 0) cflags[ci] |= CF_STATIC_SECTION; }
            else cflags[ci] &= ~CF_STATIC_SECTION;
        }
    }
    // --- upload
    c->ncells = ncells; c->nsh = nsh; c->nrows_csr = (uint32_t)rows.size();
    c->h_cell_key = keys; c->h_row_cell = row_cell; c->h_shids = shids; c->h_sh_nact = sh_nact; c->h_sh_nstat = sh_nstat;
    c->base_keys = keys; c->extra_slots.clear(); c->base_index.clear(); for (size_t i = 0; i < keys.size(); i += 1024) c->base_index.push_back(keys[i]); c->h_cell_nl = nlocal; c->h_cell_ns = nstatic; c->h_cell_begin.assign(begin.begin(), begin.begin() + ncells);
    c->h_cell_cap.resize(ncells); for (uint32_t ci = 0; ci < ncells; ci++) c->h_cell_cap[ci] = nlocal[ci] + nstatic[ci] + nghost[ci];
    c->h_cell_ng = nghost;
    c->rb_base_dirty = true; c->rb_ovl_dirty = true; c->stale_slots.clear();      // (a full build is made from the host mirrors, which its callers bring up to date first)
    c->h_rows = rows; c->free_slots.assign(MAX_LEVELS, {}); c->h_sh_begin.assign(sh_begin.begin(), sh_begin.begin() + nsh); c->h_sh_cells.assign(sh_cells.begin(), sh_cells.begin() + (size_t)nsh * 8);
    for (uint32_t ci = ncells; ci-- > 0;) if (is_pad(keys[ci])) c->free_slots[key_level(keys[ci]) & (MAX_LEVELS - 1)].push_back(ci);   // popped from the back: lowest slot first
    std::vector<uint64_t> keys_padded(keys); keys_padded.resize((size_t)((ncells + 1) & ~1u) + 2, 0xFFFFFFFFFFFFFFFFull);
    HIPCHK(c, c->d_cell_key.alloc(keys_padded.size(), acct));
    HIPCHK(c, c->d_cell_tight.alloc(ncells, acct)); HIPCHK(c, c->d_cell_begin.alloc(ncells + 1, acct)); HIPCHK(c, c->d_cell_nlocal.alloc(ncells, acct));
    HIPCHK(c, c->d_cell_nstatic.alloc(ncells, acct)); HIPCHK(c, c->d_cell_stamp.alloc(ncells, acct)); HIPCHK(c, c->d_cell_flags.alloc(ncells, acct));
    if (c->park_ready) { HIPCHK(c, c->park.d_cell_stamp.alloc(ncells, acct)); HIPCHK(c, hipMemset(c->park.d_cell_stamp.p, 0, (size_t)std::max(ncells, 1u) * 4)); }   // the parked lane's stamps follow a rebuilt table (it is idle: rebuilds happen behind drain_other_lane)
    c->pool_used = (uint32_t)rows.size(); c->pool_cap = c->pool_used + ((c->cfg.flags & RE_CFG_TIGHT_SLACK) ? 96u : c->pool_used / 4u + 65536u);      // slack: re-bucket patches append relocated segments
    HIPCHK(c, c->d_rows.alloc(c->pool_cap, acct)); HIPCHK(c, c->d_row_cell.alloc(std::max(n, c->row_cap), acct));
    { int rcs = ensure_shared_capacity(c, nsh); if (rcs != RE_OK) return rcs; }
    c->rb_sh_dirty = true; c->sh_free.clear(); c->stale_shared.clear();
    hipStream_t st = c->stream;
    HIPCHK(c, hipMemcpyAsync(c->d_cell_key.p, keys_padded.data(), keys_padded.size() * 8, hipMemcpyHostToDevice, st));
    {   // compact stream keys + the level of every 512-key chunk, when every section index fits 9 bits
        const size_t nchunks = (keys.size() + wave_keys - 1) / wave_keys;
        std::vector<uint32_t> k32(((keys.size() + 3) & ~(size_t)3) + 8, KEY32_PAD | 0x1FF7FDFFu), lvl(nchunks + 1, 0);
        for (size_t i = 0; i < keys.size(); i++) k32[i] = to_key32(keys[i]);
        for (size_t ch = 0; ch < nchunks; ch++) lvl[ch] = key_level(keys[ch * wave_keys]) & (MAX_LEVELS - 1);
        HIPCHK(c, c->d_cell_key32.alloc(k32.size(), acct)); HIPCHK(c, c->d_chunk_level.alloc(lvl.size(), acct));
        HIPCHK(c, hipMemcpyAsync(c->d_cell_key32.p, k32.data(), k32.size() * 4, hipMemcpyHostToDevice, st));
        HIPCHK(c, hipMemcpyAsync(c->d_chunk_level.p, lvl.data(), lvl.size() * 4, hipMemcpyHostToDevice, st));
        HIPCHK(c, sync_stream(st));
    }
    HIPCHK(c, hipMemcpyAsync(c->d_cell_begin.p, begin.data(), (size_t)(ncells + 1) * 4, hipMemcpyHostToDevice, st));
    if (ncells) {
        HIPCHK(c, hipMemcpyAsync(c->d_cell_nlocal.p, nlocal.data(), (size_t)ncells * 4, hipMemcpyHostToDevice, st));
        HIPCHK(c, hipMemcpyAsync(c->d_cell_nstatic.p, nstatic.data(), (size_t)ncells * 4, hipMemcpyHostToDevice, st));
        HIPCHK(c, c->d_cell_nghost.alloc(ncells, acct)); HIPCHK(c, hipMemcpyAsync(c->d_cell_nghost.p, nghost.data(), (size_t)ncells * 4, hipMemcpyHostToDevice, st));
        HIPCHK(c, c->d_cell_cap.alloc(ncells, acct)); HIPCHK(c, hipMemcpyAsync(c->d_cell_cap.p, c->h_cell_cap.data(), (size_t)ncells * 4, hipMemcpyHostToDevice, st));
        {   // shared sections linking each unique section (the device-side re-bucket leaves linked sections to the host path)
            std::vector<uint8_t> links(std::max(ncells, 1u), 0); c->h_linked_slots.clear();
            for (uint32_t s2 = 0; s2 < nsh; s2++) for (uint32_t k = 0; k < 8; k++) { const int32_t ci = sh_cells[(size_t)s2 * 8 + k]; if (ci >= 0) { if (links[ci] < 255) links[ci]++; c->h_linked_slots.push_back((uint32_t)ci); } }
            HIPCHK(c, c->d_cell_links.alloc(ncells, acct)); HIPCHK(c, hipMemcpy(c->d_cell_links.p, links.data(), std::max(ncells, 1u), hipMemcpyHostToDevice));
        }
        HIPCHK(c, hipMemcpyAsync(c->d_cell_flags.p, cflags.data(), (size_t)ncells, hipMemcpyHostToDevice, st));
        HIPCHK(c, hipMemsetAsync(c->d_cell_stamp.p, 0, (size_t)ncells * 4, st));
    }
    if (!rows.empty()) HIPCHK(c, hipMemcpyAsync(c->d_rows.p, rows.data(), rows.size() * 4, hipMemcpyHostToDevice, st));
    {
        std::vector<uint32_t> gc(rows.size());
        for (size_t i = 0; i < rows.size(); i++) gc[i] = effective_gclass(c, rows[i]);
        HIPCHK(c, c->d_rows_gc.alloc(c->pool_cap, acct));
        if (!gc.empty()) HIPCHK(c, hipMemcpy(c->d_rows_gc.p, gc.data(), gc.size() * 4, hipMemcpyHostToDevice));
    }
    if (n) HIPCHK(c, hipMemcpyAsync(c->d_row_cell.p, row_cell.data(), (size_t)n * 4, hipMemcpyHostToDevice, st));
    if (n) {                                                                 // the key of every row's own world section, row by row: the tick reads it as a stream instead of gathering cell_key[row_cell]
        if (c->d_row_key.n < std::max(n, c->row_cap)) HIPCHK(c, c->d_row_key.alloc(std::max(n, c->row_cap), acct));
        HIPCHK(c, hipMemcpyAsync(c->d_row_key.p, row_key.data(), (size_t)n * 8, hipMemcpyHostToDevice, st));
    }
    if (nsh) {
        HIPCHK(c, hipMemcpyAsync(c->d_sh_cells.p, sh_cells.data(), (size_t)nsh * 8 * 4, hipMemcpyHostToDevice, st));
        HIPCHK(c, hipMemcpyAsync(c->d_sh_begin.p, sh_begin.data(), (size_t)nsh * 4, hipMemcpyHostToDevice, st));
        HIPCHK(c, hipMemcpyAsync(c->d_sh_nact.p, sh_nact.data(), (size_t)nsh * 4, hipMemcpyHostToDevice, st));
        HIPCHK(c, hipMemcpyAsync(c->d_sh_nstat.p, sh_nstat.data(), (size_t)nsh * 4, hipMemcpyHostToDevice, st));
        std::vector<int32_t> owner(nsh, -1); std::vector<uint8_t> cached(nsh, 0), dirty(nsh, carry ? 0 : 1);
        if (carry)
            for (uint32_t s = 0; s < nsh; s++) {
                auto it = std::find(carry->shids.begin(), carry->shids.end(), shids[s]);
                if (it == carry->shids.end()) continue;
                size_t o = it - carry->shids.begin();
                cached[s] = carry->sh_cached[o];
                if (carry->sh_owner_key_idx[o] >= 0) { auto kk = std::lower_bound(keys.begin(), keys.end(), carry->sh_owner_key[o]); if (kk != keys.end() && *kk == carry->sh_owner_key[o]) owner[s] = (int32_t)(kk - keys.begin()); }
            }
        HIPCHK(c, hipMemcpyAsync(c->d_sh_owner.p, owner.data(), (size_t)nsh * 4, hipMemcpyHostToDevice, st));
        HIPCHK(c, hipMemcpyAsync(c->d_sh_cached.p, cached.data(), nsh, hipMemcpyHostToDevice, st));
        HIPCHK(c, hipMemcpyAsync(c->d_sh_dirty.p, dirty.data(), nsh, hipMemcpyHostToDevice, st));
        HIPCHK(c, sync_stream(st));
    }
    // --- end_of_changes: tight AABBs.  total_world_aabb_combining of a fresh batch == number of unique adds (:710-744)
    int too_many = carry ? (int)carry->too_many : (int)(recs.size() > 500);
    DevBuf<uint8_t> d_refold; DevBuf<Aabb> d_carried;
    if (ncells && !carry) hipLaunchKernelGGL(k_fold_tight, dim3((ncells + 255) / 256), dim3(256), 0, st, ncells, c->d_cell_key.p, c->d_cell_begin.p, c->d_cell_nlocal.p,
                                             c->d_cell_nstatic.p, c->d_rows.p, c->d_aabb.p, c->d_cell_tight.p, c->cfg.atomic_length, too_many);
    if (ncells && carry) {
        HIPCHK(c, d_refold.alloc(ncells, nullptr)); HIPCHK(c, d_carried.alloc(ncells, nullptr));
        HIPCHK(c, hipMemcpyAsync(d_refold.p, refold.data(), ncells, hipMemcpyHostToDevice, st));
        HIPCHK(c, hipMemcpyAsync(d_carried.p, carried_tight.data(), (size_t)ncells * sizeof(Aabb), hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(k_fold_tight_masked, dim3((ncells + 255) / 256), dim3(256), 0, st, ncells, c->d_cell_key.p, c->d_cell_begin.p, c->d_cell_nlocal.p, c->d_cell_nstatic.p,
                           c->d_rows.p, c->d_aabb.p, c->d_cell_tight.p, c->cfg.atomic_length, too_many, d_refold.p, d_carried.p);
    }
    if (nsh) hipLaunchKernelGGL(k_fold_shared, dim3((nsh + 255) / 256), dim3(256), 0, st, nsh, c->d_sh_begin.p, c->d_sh_nact.p, c->d_sh_nstat.p, c->d_rows.p, c->d_aabb.p, c->d_sh_aabb.p);
    if (nsh && carry) {                                                     // unchanged shared sections keep their AABB
        std::vector<Aabb> sa(nsh);
        HIPCHK(c, sync_stream(st));
        HIPCHK(c, hipMemcpy(sa.data(), c->d_sh_aabb.p, (size_t)nsh * sizeof(Aabb), hipMemcpyDeviceToHost));
        for (uint32_t s2 = 0; s2 < nsh; s2++) {
            if (carry->changed_shared_set.count(shids[s2])) continue;
            auto it = std::find(carry->shids.begin(), carry->shids.end(), shids[s2]);
            if (it != carry->shids.end()) sa[s2] = carry->sh_aabb[it - carry->shids.begin()];
        }
        HIPCHK(c, hipMemcpy(c->d_sh_aabb.p, sa.data(), (size_t)nsh * sizeof(Aabb), hipMemcpyHostToDevice));
    }
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, sync_stream(st));
    c->nlists = std::max(1u, (uint32_t)((ncells + wave_keys - 1) / wave_keys));       // waves of k_scan_cull
    HIPCHK(c, sync_stream(st));
    d_refold.release(nullptr); d_carried.release(nullptr);
    if (c->cfg.flags & RE_CFG_PROBE) {                                      // key -> slot table at load <= 0.5 (tombstones of later patches included until the next full build)
        uint32_t sz = 1024; while (sz < 2u * (uint32_t)std::min<size_t>(ncells + 1u, 1u << 30)) sz <<= 1;
        if (c->htab_mask + 1u != sz) { c->d_htab.release(acct); HIPCHK(c, c->d_htab.alloc(sz, acct)); c->htab_mask = sz - 1u; }
        HIPCHK(c, hipMemsetAsync(c->d_htab.p, 0xFF, (size_t)sz * sizeof(HashEntry), st));
        if (ncells) hipLaunchKernelGGL(k_hash_build, dim3((ncells + 255) / 256), dim3(256), 0, st, ncells, c->d_cell_key.p, c->d_htab.p, c->htab_mask);
        HIPCHK(c, hipGetLastError()); HIPCHK(c, sync_stream(st));
        c->htab_keys = (uint32_t)keys.size();
    }
    if (!carry) { c->dirty_pending = true; c->have_cull = false; }
    else if (!carry->changed_static.empty()) c->dirty_pending = true;
    return RE_OK;
}

static int resolve(re_ctx *c);
static int size_frame_buffers(re_ctx *c);
// device tables of the per-model level-of-view bands: table 0 = "default bands", table t >= 1 = custom_lod[t - 1]; one table index per group class
static int upload_lod_tables(re_ctx *c) {
    c->lod_tables_on = false;
    if (c->custom_lod.empty() || !c->ngclass) return RE_OK;
    std::vector<uint32_t> tab(c->ngclass, 0u), ln(c->custom_lod.size() + 1, 0u); std::vector<float> lmin((c->custom_lod.size() + 1) * 8, 0.f), lmax((c->custom_lod.size() + 1) * 8, 0.f);
    bool any = false;
    for (size_t t = 0; t < c->custom_lod.size(); t++) {
        const auto &x = c->custom_lod[t];
        ln[t + 1] = x.n; for (int k = 0; k < 8; k++) { lmin[(t + 1) * 8 + k] = x.lmin[k]; lmax[(t + 1) * 8 + k] = x.lmax[k]; }
        for (uint32_t g = 0; g < c->ngclass; g++) if (c->h_gkeys[g].model == x.model && c->h_gkeys[g].rs == x.rs) { tab[g] = (uint32_t)t + 1u; any = true; }
    }
    if (!any) return RE_OK;
    HIPCHK(c, c->d_gc_lodtab.alloc(tab.size(), nullptr)); HIPCHK(c, c->d_lod_n.alloc(ln.size(), nullptr)); HIPCHK(c, c->d_lod_min.alloc(lmin.size(), nullptr)); HIPCHK(c, c->d_lod_max.alloc(lmax.size(), nullptr));
    HIPCHK(c, hipMemcpy(c->d_gc_lodtab.p, tab.data(), tab.size() * 4, hipMemcpyHostToDevice)); HIPCHK(c, hipMemcpy(c->d_lod_n.p, ln.data(), ln.size() * 4, hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(c->d_lod_min.p, lmin.data(), lmin.size() * 4, hipMemcpyHostToDevice)); HIPCHK(c, hipMemcpy(c->d_lod_max.p, lmax.data(), lmax.size() * 4, hipMemcpyHostToDevice));
    c->lod_tables_on = true;
    return RE_OK;
}

extern "C" int re_upload_entities(re_ctx *c, const re_entities *E, uint32_t *n_rejected) try {
    if (!c) return RE_E_ARG;
    if (!E || (E->n && (!E->entity_id || !E->model_index || !E->flags || !E->original_aabb || !E->position))) return c->fail(RE_E_ARG, "re_upload_entities: missing required array");
    HIPCHK(c, hipSetDevice(c->device));
    if (c->park_ready) { (void)drain_other_lane(c); free_second_lane(c); }
    HIPCHK(c, sync_stream(c->stream));
    free_world(c);
    const uint32_t n = E->n;
    c->n = n;
    uint64_t *acct = &c->dev_bytes;
    // host staging: defaults for absent components, normalised axes (Rotation::new etc., exports/movement_components.rs:108-164)
    std::vector<float> rot((size_t)n * 4), scl((size_t)n * 3);
    std::vector<uint32_t> flags(n), gclass(n);
    std::vector<uint32_t> dyn_row; std::vector<float> dvel, dacc, drv, dra;
    std::unordered_map<GroupKey, uint32_t, GroupKeyHash> &gmap = c->gmap; std::vector<GroupKey> gkeys;
    // Row order: the dynamic entities (Velocity or VelocityRotation) first, otherwise upload order.  Row j of every per-entity column then IS
    // dynamic entity j, and the tick reads and writes contiguous streams (k_tick) instead of gathering 12..64-byte pieces per entity.
    std::vector<uint32_t> perm; bool permuted = false;
    {
        auto dynamic = [&](uint32_t i) { return (E->flags[i] & (F_HAS_VEL | F_HAS_ROTVEL)) && !(E->flags[i] & F_PHANTOM); };   // (a halo replica never ticks here)
        uint32_t nd = 0; bool prefix = true;
        for (uint32_t i = 0; i < n; i++) if (dynamic(i)) { if (i != nd) prefix = false; nd++; }
        if (nd && !prefix) {
            perm.resize(n); uint32_t a = 0, b = nd;
            for (uint32_t i = 0; i < n; i++) { if (dynamic(i)) perm[a++] = i; else perm[b++] = i; }
            permuted = true;
        }
    }
    auto src = [&](uint32_t r) -> size_t { return permuted ? perm[r] : r; };
    c->h_id.resize(n); for (uint32_t r = 0; r < n; r++) c->h_id[r] = E->entity_id[src(r)];
    c->ids_identity = true; c->id_rows.clear(); c->id_to_row.clear();
    for (uint32_t r = 0; r < n && c->ids_identity; r++) if (c->h_id[r] != r) c->ids_identity = false;
    for (uint32_t r = 0; r < n; r++) {
        const size_t i = src(r);
        uint32_t fl = E->flags[i] & ~(F_HAS_MOVED | F_HAS_ROTATED | F_DEAD);
        if (fl & F_PHANTOM) fl &= ~(F_HAS_VEL | F_HAS_ACC | F_HAS_ROTVEL | F_HAS_ROTACC | F_ALWAYS_EXEC | F_USER | F_LIGHT_ANY);   // RE_F_PHANTOM: in the tree, otherwise inert (its owner shard ticks, draws and lists it)
        flags[r] = fl;
        if ((fl & F_HAS_ROT) && !E->rotation) return c->fail(RE_E_ARG, "entity %u has RE_F_HAS_ROT but rotation == NULL", r);
        if ((fl & F_HAS_SCALE) && !E->scale) return c->fail(RE_E_ARG, "entity %u has RE_F_HAS_SCALE but scale == NULL", r);
        if ((fl & F_HAS_VEL) && !E->velocity) return c->fail(RE_E_ARG, "entity %u has RE_F_HAS_VEL but velocity == NULL", r);
        if ((fl & F_HAS_ACC) && !E->acceleration) return c->fail(RE_E_ARG, "entity %u has RE_F_HAS_ACC but acceleration == NULL", r);
        if ((fl & F_HAS_ROTVEL) && !E->rotation_velocity) return c->fail(RE_E_ARG, "entity %u has RE_F_HAS_ROTVEL but rotation_velocity == NULL", r);
        if ((fl & F_HAS_ROTACC) && !E->rotation_acceleration) return c->fail(RE_E_ARG, "entity %u has RE_F_HAS_ROTACC but rotation_acceleration == NULL", r);
        if (fl & F_HAS_ROT) {
            const float *a = E->rotation + i * 4; float nn = norm3(a[0], a[1], a[2]);
            rot[r * 4 + 0] = a[0] / nn; rot[r * 4 + 1] = a[1] / nn; rot[r * 4 + 2] = a[2] / nn; rot[r * 4 + 3] = a[3];
        } else { rot[r * 4 + 0] = 1.f; rot[r * 4 + 1] = 0.f; rot[r * 4 + 2] = 0.f; rot[r * 4 + 3] = 0.f; }           // Rotation::default
        if (fl & F_HAS_SCALE) { scl[r * 3 + 0] = E->scale[i * 3 + 0]; scl[r * 3 + 1] = E->scale[i * 3 + 1]; scl[r * 3 + 2] = E->scale[i * 3 + 2]; }
        else { scl[r * 3 + 0] = scl[r * 3 + 1] = scl[r * 3 + 2] = 1.f; }                                          // Scale::default
        GroupKey gk{ E->model_index[i], E->render_system ? E->render_system[i] : 0u, E->sortable ? E->sortable[i] : 0u };
        auto it = gmap.find(gk);
        if (it == gmap.end()) { it = gmap.emplace(gk, (uint32_t)gkeys.size()).first; gkeys.push_back(gk); }
        gclass[r] = it->second;
        if (fl & (F_HAS_VEL | F_HAS_ROTVEL)) {
            dyn_row.push_back(r);
            for (int k = 0; k < 3; k++) { dvel.push_back((fl & F_HAS_VEL) ? E->velocity[i * 3 + k] : 0.f); dacc.push_back((fl & F_HAS_ACC) ? E->acceleration[i * 3 + k] : 0.f); }
            float rv[4] = { 1.f, 0.f, 0.f, 0.f }, ra[4] = { 1.f, 0.f, 0.f, 0.f };
            if (fl & F_HAS_ROTVEL) { const float *a = E->rotation_velocity + i * 4; float nn = norm3(a[0], a[1], a[2]); rv[0] = a[0] / nn; rv[1] = a[1] / nn; rv[2] = a[2] / nn; rv[3] = a[3]; }
            if (fl & F_HAS_ROTACC) { const float *a = E->rotation_acceleration + i * 4; float nn = norm3(a[0], a[1], a[2]); ra[0] = a[0] / nn; ra[1] = a[1] / nn; ra[2] = a[2] / nn; ra[3] = a[3]; }
            for (int k = 0; k < 4; k++) { drv.push_back(rv[k]); dra.push_back(ra[k]); }
        }
    }
    if (!c->ids_identity) {
        uint32_t max_id = 0; for (uint32_t r = 0; r < n; r++) max_id = std::max(max_id, c->h_id[r]);
        if ((uint64_t)max_id < 4ull * n + 1024ull) {                           // dense enough: a direct id -> row table
            c->id_to_row.assign((size_t)max_id + 1, 0xFFFFFFFFu);
            for (uint32_t r = 0; r < n; r++) { if (c->id_to_row[c->h_id[r]] != 0xFFFFFFFFu) return c->fail(RE_E_ARG, "re_upload_entities: duplicate entity id %u", c->h_id[r]); c->id_to_row[c->h_id[r]] = r; }
        } else {
            c->id_rows.resize(n);
            for (uint32_t r = 0; r < n; r++) c->id_rows[r] = { c->h_id[r], r };
            std::sort(c->id_rows.begin(), c->id_rows.end());
            for (uint32_t r = 1; r < n; r++) if (c->id_rows[r].first == c->id_rows[r - 1].first) return c->fail(RE_E_ARG, "re_upload_entities: duplicate entity id %u", c->id_rows[r].first);
        }
    }
    c->h_flags = flags; c->h_dyn_row = dyn_row; c->has_rotvel = false;
    c->h_light_rows.clear(); for (uint32_t r = 0; r < n; r++) if (flags[r] & F_LIGHT_ANY) c->h_light_rows.push_back(r);
    c->light_rows_dirty = true;
    c->n_phantom = 0; for (uint32_t r = 0; r < n; r++) if (flags[r] & F_PHANTOM) c->n_phantom++;      // halo replicas: listed by the scan, hidden from the pack (twice each in duplicates mode)
    c->user_row = ROW_CELL_NONE; for (uint32_t r = 0; r < n; r++) if (flags[r] & F_USER) { c->user_row = r; break; }
    c->d_row_moved.release(nullptr); c->d_col_moved.release(nullptr); c->d_col_tab.release(nullptr); c->col_moved_cap = 0;
    for (uint32_t f : flags) if (f & F_HAS_ROTVEL) { c->has_rotvel = true; break; }
    c->ndyn = c->ndyn0 = c->dyn_cap = (uint32_t)dyn_row.size();
    c->ngclass = (uint32_t)gkeys.size(); c->nslots = c->ngclass * 8u; c->h_gkeys = gkeys;
    hipStream_t st = c->stream;
    c->ghost_cap = std::max(2048u, n / 8u);   // ghost instances of the frozen static cache (68 bytes each)
    c->n_ghost = 0; c->h_ghost_gc.clear(); c->ghost_map.clear(); c->dormant_cached.clear();      // (round 2 had these four resets inside the comment above: a second upload on a context that held ghosts kept them)
    c->row_cap = c->ghost_base = c->n_base = n;
    HIPCHK(c, c->d_id.alloc((size_t)n + c->ghost_cap, acct)); HIPCHK(c, c->d_gclass.alloc(n, acct)); HIPCHK(c, c->d_flags.alloc(n, acct)); HIPCHK(c, c->d_mat.alloc(((size_t)n + c->ghost_cap) * 16, acct));
    HIPCHK(c, c->d_pos.alloc((size_t)n * 3, acct)); HIPCHK(c, c->d_rot.alloc((size_t)n * 4, acct)); HIPCHK(c, c->d_scale.alloc((size_t)n * 3, acct));
    HIPCHK(c, c->d_aabb.alloc(n, acct)); HIPCHK(c, c->d_orig.alloc(n, acct));
    HIPCHK(c, c->d_row_key.alloc(n, acct)); HIPCHK(c, c->d_row_nk.alloc(n, acct)); HIPCHK(c, c->d_shrec.alloc(n, acct)); HIPCHK(c, c->d_counter.alloc(4, acct));
    HIPCHK(c, c->d_dyn_row.alloc(c->ndyn, acct)); HIPCHK(c, c->d_dyn_vel.alloc((size_t)c->ndyn * 3, acct)); HIPCHK(c, c->d_dyn_acc.alloc((size_t)c->ndyn * 3, acct));
    HIPCHK(c, c->d_dyn_rotvel.alloc((size_t)c->ndyn * 4, acct)); HIPCHK(c, c->d_dyn_rotacc.alloc((size_t)c->ndyn * 4, acct));
    std::vector<float> pos_p, orig_p;                                          // the caller's arrays in row order (only when the rows were permuted)
    if (permuted) {
        pos_p.resize((size_t)n * 3); orig_p.resize((size_t)n * 6);
        for (uint32_t r = 0; r < n; r++) { const size_t i = perm[r]; memcpy(&pos_p[(size_t)r * 3], E->position + i * 3, 12); memcpy(&orig_p[(size_t)r * 6], E->original_aabb + i * 6, 24); }
    }
    if (n) {
        HIPCHK(c, hipMemcpyAsync(c->d_id.p, c->h_id.data(), (size_t)n * 4, hipMemcpyHostToDevice, st));
        HIPCHK(c, hipMemcpyAsync(c->d_gclass.p, gclass.data(), (size_t)n * 4, hipMemcpyHostToDevice, st));
        HIPCHK(c, hipMemcpyAsync(c->d_flags.p, flags.data(), (size_t)n * 4, hipMemcpyHostToDevice, st));
        HIPCHK(c, hipMemcpyAsync(c->d_pos.p, permuted ? pos_p.data() : E->position, (size_t)n * 12, hipMemcpyHostToDevice, st));
        HIPCHK(c, hipMemcpyAsync(c->d_rot.p, rot.data(), (size_t)n * 16, hipMemcpyHostToDevice, st));
        HIPCHK(c, hipMemcpyAsync(c->d_scale.p, scl.data(), (size_t)n * 12, hipMemcpyHostToDevice, st));
        HIPCHK(c, hipMemcpyAsync(c->d_orig.p, permuted ? orig_p.data() : E->original_aabb, (size_t)n * 24, hipMemcpyHostToDevice, st));
    }
    if (c->ndyn) {
        HIPCHK(c, hipMemcpyAsync(c->d_dyn_row.p, dyn_row.data(), (size_t)c->ndyn * 4, hipMemcpyHostToDevice, st));
        HIPCHK(c, hipMemcpyAsync(c->d_dyn_vel.p, dvel.data(), (size_t)c->ndyn * 12, hipMemcpyHostToDevice, st));
        HIPCHK(c, hipMemcpyAsync(c->d_dyn_acc.p, dacc.data(), (size_t)c->ndyn * 12, hipMemcpyHostToDevice, st));
        HIPCHK(c, hipMemcpyAsync(c->d_dyn_rotvel.p, drv.data(), (size_t)c->ndyn * 16, hipMemcpyHostToDevice, st));
        HIPCHK(c, hipMemcpyAsync(c->d_dyn_rotacc.p, dra.data(), (size_t)c->ndyn * 16, hipMemcpyHostToDevice, st));
    }
    // groups
    {
        std::vector<uint32_t> gm(c->ngclass + 1), gr(c->ngclass + 1), gs(c->ngclass + 1);
        for (uint32_t g = 0; g < c->ngclass; g++) { gm[g] = gkeys[g].model; gr[g] = gkeys[g].rs; gs[g] = gkeys[g].sort; }
        HIPCHK(c, c->d_gc_model.alloc(c->ngclass, acct)); HIPCHK(c, c->d_gc_rs.alloc(c->ngclass, acct)); HIPCHK(c, c->d_gc_sort.alloc(c->ngclass, acct));
        HIPCHK(c, c->d_group_count.alloc(c->nslots, acct)); HIPCHK(c, c->d_group_begin.alloc(c->nslots, acct)); HIPCHK(c, c->d_group_fill.alloc(c->nslots, acct));
        if (c->ngclass) {
            HIPCHK(c, hipMemcpyAsync(c->d_gc_model.p, gm.data(), (size_t)c->ngclass * 4, hipMemcpyHostToDevice, st));
            HIPCHK(c, hipMemcpyAsync(c->d_gc_rs.p, gr.data(), (size_t)c->ngclass * 4, hipMemcpyHostToDevice, st));
            HIPCHK(c, hipMemcpyAsync(c->d_gc_sort.p, gs.data(), (size_t)c->ngclass * 4, hipMemcpyHostToDevice, st));
        }
        HIPCHK(c, hipMemsetAsync(c->d_group_count.p, 0, (size_t)std::max(c->nslots, 1u) * 4, st));
        HIPCHK(c, hipMemsetAsync(c->d_group_fill.p, 0, (size_t)std::max(c->nslots, 1u) * 4, st));
        if (c->nslots <= COUNT_SLOTS_MAX) {
            const size_t words = 2u * CURSOR_SHARDS * (size_t)std::max(c->nslots, 1u);
            HIPCHK(c, c->d_gcount.alloc(words, acct)); HIPCHK(c, c->d_gfill.alloc(words, acct));
            HIPCHK(c, hipMemsetAsync(c->d_gcount.p, 0, words * 4, st)); HIPCHK(c, hipMemsetAsync(c->d_gfill.p, 0, words * 4, st));
        }
    }
    // TRS -> matrix, AABB, section keys on the GPU
    HIPCHK(c, hipMemsetAsync(c->d_counter.p, 0, 16, st));
    if (n) hipLaunchKernelGGL(k_transform_assign, dim3((n + 255) / 256), dim3(256), 0, st, row_arrays(c), 0u, n, c->cfg.outline_length, c->cfg.atomic_length,
                              c->d_row_key.p, c->d_row_nk.p, c->d_shrec.p, c->d_counter.p, n);
    HIPCHK(c, hipGetLastError());
    std::vector<uint64_t> row_key(n); std::vector<uint8_t> row_nk(n); uint32_t nshrec = 0;
    if (n) {
        HIPCHK(c, hipMemcpyAsync(row_key.data(), c->d_row_key.p, (size_t)n * 8, hipMemcpyDeviceToHost, st));
        HIPCHK(c, hipMemcpyAsync(row_nk.data(), c->d_row_nk.p, (size_t)n, hipMemcpyDeviceToHost, st));
    }
    HIPCHK(c, hipMemcpyAsync(&nshrec, c->d_counter.p, 4, hipMemcpyDeviceToHost, st));
    HIPCHK(c, sync_stream(st));
    std::vector<SharedRec> shrec(nshrec);
    if (nshrec) HIPCHK(c, hipMemcpy(shrec.data(), c->d_shrec.p, (size_t)nshrec * sizeof(SharedRec), hipMemcpyDeviceToHost));
    uint32_t rejected = 0;
    for (uint32_t r = 0; r < n; r++) if (row_nk[r] == 0) rejected++;
    if (n_rejected) *n_rejected = rejected;
    c->d_shrec.release(acct);
    c->h_row_key = row_key; c->h_row_nk = row_nk; c->h_gclass = gclass; c->h_row_shared_keys.clear(); c->h_uncached.clear(); c->h_oob_ids.clear(); c->pending.clear();
    for (const SharedRec &sr : shrec) { std::array<uint64_t, 8> a; memcpy(a.data(), sr.keys, sizeof sr.keys); c->h_row_shared_keys[sr.row] = a; }
    int rc = build_sections(c, row_key, row_nk, shrec, flags);
    if (rc != RE_OK) return rc;
    rc = upload_dyn_cells(c);
    if (rc != RE_OK) return rc;
    // frame buffers
    c->item_cap = c->out_cap = c->list_cap = 0;
    { int rc2 = size_frame_buffers(c); if (rc2 != RE_OK) return rc2; }
    HIPCHK(c, c->d_hdr.alloc(NUM_FRAME_HEADERS, acct)); HIPCHK(c, c->d_th.alloc(1, acct)); HIPCHK(c, c->d_params.alloc(1, acct));
    {
        void *hb = nullptr, *db = nullptr;
        c->nslots_cap = std::max(2u * c->nslots, 1024u);                           // head-room: entities added later may bring new (ModelId, sortable) groups
        HIPCHK(c, alloc_host_block(c->nslots_cap, &hb, &db));
        c->h_block = hb;
        c->h_res = hb_at<HostResult>(hb, HB_RES); c->d_hres = hb_at<HostResult>(db, HB_RES);
        c->h_spec = hb_at<SpecState>(hb, HB_SPEC); c->d_hspec = hb_at<SpecState>(db, HB_SPEC);
        c->h_th = hb_at<TickHeader>(hb, HB_TICK); c->d_hth = hb_at<TickHeader>(db, HB_TICK);
        c->h_ranges = hb_at<InstanceRange>(hb, HB_RANGES); c->d_hranges = hb_at<InstanceRange>(db, HB_RANGES);
    }
    HIPCHK(c, c->d_spec.alloc(1, nullptr)); HIPCHK(c, hipMemset(c->d_spec.p, 0, sizeof(SpecState)));
    HIPCHK(c, hipMemsetAsync(c->d_hdr.p, 0, NUM_FRAME_HEADERS * sizeof(FrameHeader), c->stream));
    HIPCHK(c, hipMemsetAsync(c->d_th.p, 0, sizeof(TickHeader), c->stream));
    HIPCHK(c, sync_stream(c->stream));
    c->frame = 0; c->lane_seq = 0; c->th_clean = true; c->pred_total = 0;
    return upload_lod_tables(c);
} RE_ABI_GUARD(c, "re_upload_entities")

// level_views.custom (flows/render_flow.rs:495-499, 889-893; registered by register_model_with_render_system :1069-1076): a model with custom_level_of_view
// uses its own bands instead of the render system's default ones.  n == 0 removes a model's bands.
extern "C" int re_set_model_lod(re_ctx *c, uint32_t model_index, uint32_t render_system, uint32_t n_lod, const float *lod_min, const float *lod_max) try {
    if (!c || (n_lod && (!lod_min || !lod_max))) return RE_E_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    if (c->h_res) { int rc_ = resolve(c); if (rc_ != RE_OK) return rc_; }
    auto it = std::find_if(c->custom_lod.begin(), c->custom_lod.end(), [&](const re_ctx::CustomLod &x) { return x.model == model_index && x.rs == render_system; });
    if (!n_lod) { if (it != c->custom_lod.end()) c->custom_lod.erase(it); }
    else {
        if (it == c->custom_lod.end()) { c->custom_lod.push_back(re_ctx::CustomLod{}); it = c->custom_lod.end() - 1; }
        it->model = model_index; it->rs = render_system; it->n = std::min(n_lod, 8u);
        for (uint32_t k = 0; k < 8; k++) { it->lmin[k] = k < it->n ? lod_min[k] : 0.f; it->lmax[k] = k < it->n ? lod_max[k] : 0.f; }
    }
    return c->h_res ? upload_lod_tables(c) : RE_OK;
} RE_ABI_GUARD(c, "re_set_model_lod")

// ------------------------------------------------------------------------------------------------
// frame parameters: RenderFrustumCuller::new(P*V), LogicFrustumCuller::new(wsl, pos) and the two
// candidate boxes of flows/pipeline.rs:219-226 / flows/visible_world_flow.rs:47-57,117-145
// ------------------------------------------------------------------------------------------------
static void fill_level_boxes(LevelBox *out, uint32_t maxlevel, float wsl, float xmin, float xmax, float ymin, float ymax, float zmin, float zmax) {
    for (uint32_t level = 0; level < (uint32_t)MAX_LEVELS; level++) {
        LevelBox b{};
        if (level < maxlevel) {
            float ll = wsl * ldexpf(1.0f, (int)level);                              // world_section_length * 2.0_f32.powf(level)
            b.nx = f2u32(ceilf((xmax - xmin) / ll)); b.ny = f2u32(ceilf((ymax - ymin) / ll)); b.nz = f2u32(ceilf((zmax - zmin) / ll));
            b.bx = f2u32(xmin / ll); b.by = f2u32(ymin / ll); b.bz = f2u32(zmin / ll);
            b.level_length = ll;
        }
        out[level] = b;
    }
}

static void fill_packed_boxes(PBox *out, const LevelBox *in, uint32_t maxlevel) {
    for (uint32_t l = 0; l < (uint32_t)MAX_LEVELS; l++) {
        const LevelBox &b = in[l]; PBox p;
        bool ok = l < maxlevel && b.nx && b.ny && b.nz;
        auto m1 = [](uint32_t n) { return (n > 65536u ? 65536u : n) - 1u; };
        p.sub_hi = ((ok ? l : (l ^ 0x8000u)) << 16) | (b.bx & 0xFFFFu);          // an empty box gets an unmatchable level field
        p.sub_lo = ((b.bz & 0xFFFFu) << 16) | (b.by & 0xFFFFu);
        p.min_hi = ok ? m1(b.nx) : 0u;                                           // level difference must be 0
        p.min_lo = ok ? ((m1(b.nz) << 16) | m1(b.ny)) : 0u;
        out[l] = p;
    }
}

// Key chunks (2048 keys = one workgroup of k_scan_cull) that can hold candidates: per level, the x-slab [bx, bx+nx) of the union of
// the two candidate boxes is one contiguous run of the key-sorted section array.  At most 4 disjoint ascending spans (see ScanSpans).
static ScanSpans candidate_spans(re_ctx *c, uint32_t nchunks) {
    ScanSpans SP{}; const FrameParams &P = c->P;
    const uint32_t per = (uint32_t)(CULL_THREADS / 64) * (c->key32 ? WAVE_KEYS32 : WAVE_KEYS);
    std::vector<std::pair<uint32_t, uint32_t>> sp;                            // [first chunk, end chunk)
    const std::vector<uint64_t> &K = c->base_keys;                            // slot order of the last full build (sorted); sections patched in since sit in padding slots and are only a hint short
    for (uint32_t l = 0; l < P.max_level && l < (uint32_t)MAX_LEVELS; l++) {
        uint32_t x0 = 0xFFFFFFFFu, x1 = 0;
        for (int w = 0; w < 2; w++) {
            const LevelBox &b = P.box[w][l];
            if (!b.nx || !b.ny || !b.nz) continue;
            if (b.bx + b.nx > 0xFFFFu) { x0 = 0; x1 = 0xFFFFu; continue; }      // wrapped box: the whole level run
            x0 = std::min(x0, b.bx); x1 = std::max(x1, b.bx + b.nx);
        }
        if (x0 >= x1) continue;
        uint64_t k0 = ((uint64_t)l << 48) | ((uint64_t)x0 << 32), k1 = ((uint64_t)l << 48) | ((uint64_t)x1 << 32);
        // two-level search (the block index of every 1024th key stays in cache; a plain binary search over the 80 MB key array costs ~0.5 us of cache misses per bound,
        // in front of every frame's launch)
        auto bound = [&](uint64_t k) -> size_t {
            if (c->base_index.empty()) return (size_t)(std::lower_bound(K.begin(), K.end(), k) - K.begin());
            const size_t blk = std::lower_bound(c->base_index.begin(), c->base_index.end(), k) - c->base_index.begin();     // first block whose first key is >= k
            const size_t lo = blk ? (blk - 1) * 1024 : 0, hi = std::min(blk * 1024 + 1, K.size());
            return (size_t)(std::lower_bound(K.begin() + lo, K.begin() + hi, k) - K.begin());
        };
        size_t i0 = bound(k0), i1 = bound(k1);
        if (i0 >= i1) continue;
        sp.push_back({ (uint32_t)(i0 / per), std::min(nchunks, (uint32_t)((i1 + per - 1) / per)) });
    }
    std::sort(sp.begin(), sp.end());
    std::vector<std::pair<uint32_t, uint32_t>> m;
    for (auto &s : sp) { if (!m.empty() && s.first <= m.back().second) m.back().second = std::max(m.back().second, s.second); else m.push_back(s); }
    while (m.size() > 4) {                                                    // merge the two closest spans
        size_t best = 0; for (size_t i = 1; i + 1 < m.size(); i++) if (m[i + 1].first - m[i].second < m[best + 1].first - m[best].second) best = i;
        m[best].second = m[best + 1].second; m.erase(m.begin() + best + 1);
    }
    SP.n = (uint32_t)m.size();
    for (size_t i = 0; i < m.size(); i++) { SP.start[i] = m[i].first; SP.count[i] = m[i].second - m[i].first; }
    return SP;
}

// The stream filter for the compact 32-bit keys: ONE box per level, the bounding box of the logic and the render candidate box
// (per-field lower / upper bounds; indices above 511 cannot exist in such a world).  It only has to be conservative: the waves
// that find a candidate run the exact per-box tests on the full key (section_multiplicity_boxes), so a section of the union that
// lies in neither box is dropped there and is not counted as a candidate.  Halves the VALU work of the stream.
static void fill_union_boxes32(PBox32 *out, const LevelBox *a, const LevelBox *b, uint32_t maxlevel) {
    for (uint32_t l = 0; l < (uint32_t)MAX_LEVELS; l++) {
        PBox32 p{ 0x1FF7FDFFu, 0u };                                              // empty: lower bound above upper bound in every field
        uint32_t lo[3] = { 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu }, hi[3] = { 0, 0, 0 }; bool any = false;
        for (const LevelBox *bx : { &a[l], &b[l] }) {
            if (l >= maxlevel || !bx->nx || !bx->ny || !bx->nz) continue;
            const uint32_t base[3] = { bx->bx, bx->bz, bx->by }, n[3] = { bx->nx, bx->nz, bx->ny };     // field order x | z | y
            if (base[0] > 511u || base[1] > 511u || base[2] > 511u) continue;      // the box starts outside the world
            for (int k = 0; k < 3; k++) { lo[k] = std::min(lo[k], base[k]); hi[k] = std::max(hi[k], std::min(base[k] + n[k] - 1u, 511u)); }
            any = true;
        }
        if (any) { p.lo = (lo[0] << 20) | (lo[1] << 10) | lo[2]; p.hi = (hi[0] << 20) | (hi[1] << 10) | hi[2]; }
        out[l] = p;
    }
}

static void make_frame_params(re_ctx *c, const re_camera *cam, uint32_t flags) {
    FrameParams &P = c->P;
    make_planes(cam->projection_view, P.planes);
    for (int k = 0; k < 3; k++) P.cam[k] = cam->position[k];
    P.far_draw = cam->far_draw;
    float wsl = (float)c->cfg.atomic_length;
    P.lookahead = wsl;
    P.n_lod = cam->n_lod > 8 ? 8 : cam->n_lod;
    for (int k = 0; k < 8; k++) { P.lod_min[k] = cam->lod_min[k]; P.lod_max[k] = cam->lod_max[k]; }
    P.max_level = c->maxlevel; P.frame = c->frame; P.emit_duplicates = (flags & RE_CULL_EMIT_DUPLICATES) ? 1u : 0u;
#ifdef RE_EXP_STAGES
    { static const uint32_t stop = getenv("RE_EXP_STAGE_STOP") ? (uint32_t)atoi(getenv("RE_EXP_STAGE_STOP")) : 0u; P.pad = stop; }      // tools/stage_stop.py (development builds only)
#endif
    float draw = wsl * 2.0f;                                                    // find_visible_world_ids_entire_world(.., wsl * 2.0, ..)
    fill_level_boxes(P.box[0], c->maxlevel, wsl, rmax(cam->position[0] - draw, 0.0f), cam->position[0] + draw, rmax(cam->position[1] - draw, 0.0f), cam->position[1] + draw,
                     rmax(cam->position[2] - draw, 0.0f), cam->position[2] + draw);
    float half = cam->far_draw / 2.0f;                                          // find_visible_world_ids_frustum_aabb
    float cx = cam->direction[0] * half + cam->position[0], cy = cam->direction[1] * half + cam->position[1], cz = cam->direction[2] * half + cam->position[2];
    fill_level_boxes(P.box[1], c->maxlevel, wsl, rmax(cx - half, 0.0f), cx + half, rmax(cy - half, 0.0f), cy + half, rmax(cz - half, 0.0f), cz + half);
    fill_packed_boxes(c->PB.box[0], P.box[0], c->maxlevel); fill_packed_boxes(c->PB.box[1], P.box[1], c->maxlevel);
    fill_union_boxes32(c->PB32.box, P.box[0], P.box[1], c->maxlevel);
}

static int finish_tick(re_ctx *c, re_tick_result *out);
static int resolve(re_ctx *c);

static void fill_visible(re_ctx *c, re_visible *out) {
    const HostResult &h = *c->h_res;
    uint32_t cap = c->ext_out_ids ? c->ext_out_cap : c->out_cap;
    c->groups_out.resize(std::min(h.n_groups, c->nslots));
    for (uint32_t g = 0; g < h.n_groups && g < c->nslots; g++) {
        const InstanceRange &r = c->h_ranges[g];
        c->groups_out[g] = re_instance_range{ r.model_index, r.render_system, r.sortable, r.begin, r.count };
    }
    if (!out) return;
    out->n_visible_sections = h.n_vis_map; out->n_visible_vec = h.n_vis_vec;
    out->n_instances = h.total; out->n_written = std::min(h.total, cap);
    out->n_groups = (uint32_t)c->groups_out.size(); out->groups = c->groups_out.data();
    out->d_entity_ids = c->ext_out_ids ? c->ext_out_ids : c->d_out_ids.p;
    out->d_matrices = c->ext_out_mats ? c->ext_out_mats : c->d_out_mats.p;
}

// Frame headers rotate through three buffers: frame f accumulates into header f % 3, its pack reads it and clears header (f + 2) % 3 for
// the frame after next -- so the pack of frame f may run inside the launch of frame f + 1, which is filling header (f + 1) % 3.
static FrameHeader *frame_header(re_ctx *c, uint32_t frame) { return c->d_hdr.p + frame % NUM_FRAME_HEADERS; }
static ItemSink item_sink(re_ctx *c, uint32_t frame) {
    const size_t half = (size_t)(frame & 1u) * c->item_cap;
    ItemSink K; K.item_row = c->d_item_row.p + half; K.item_slot = c->d_item_slot.p + half; K.item_cap = c->item_cap; K.rows = c->d_rows.p; K.rows_gc = c->d_rows_gc.p;
    K.nshards = c->single_shard ? 1u : CURSOR_SHARDS; K.seg_cap = c->item_cap / K.nshards; K.group_count = nullptr; K.count_nslots = 0; K.slot_write_through = 0;
    K.gc_lodtab = c->lod_tables_on ? c->d_gc_lodtab.p : nullptr; K.lod_n = c->d_lod_n.p; K.lod_min = c->d_lod_min.p; K.lod_max = c->d_lod_max.p;
    return K;
}
static SharedArrays shared_arrays(re_ctx *c) {
    SharedArrays S; S.n = c->nsh; S.cells = c->d_sh_cells.p; S.aabb = c->d_sh_aabb.p; S.begin = c->d_sh_begin.p; S.nact = c->d_sh_nact.p; S.nstat = c->d_sh_nstat.p;
    S.owner = c->d_sh_owner.p; S.cached = c->d_sh_cached.p; return S;
}

// multi-kernel pack for large visible sets: count -> scan -> scatter
// The counts of the large-pack frame about to be issued live in parity `par` of d_gcount / d_gfill: both must be clear before anything adds to them.
static int prepare_group_counts(re_ctx *c, uint32_t par) {
    if (!c->gc_dirty[par]) return RE_OK;
    const size_t words = (size_t)CURSOR_SHARDS * std::max(c->nslots, 1u);
    HIPCHK(c, hipMemsetAsync(c->d_gcount.p + par * words, 0, words * 4, c->stream)); HIPCHK(c, hipMemsetAsync(c->d_gfill.p + par * words, 0, words * 4, c->stream));
    c->gc_dirty[par] = false;
    return RE_OK;
}
// re_timing_begin: the begin / end events of every `every`-th launch of the selected kernel (hipExtLaunchKernelGGL ties them to the dispatch itself)
static void take_timing_events(re_ctx *c, hipEvent_t *a, hipEvent_t *b) {
    if ((c->k1_seen++ % c->k1_every) == 0 && c->k1_used + 2 <= c->k1_events.size()) { *a = c->k1_events[c->k1_used]; *b = c->k1_events[c->k1_used + 1]; c->k1_used += 2; }
}

static int launch_pack_large(re_ctx *c, FrameHeader *hdr, FrameHeader *hdr_next, bool counted_by_scan = false) {
    const ItemSink KS = item_sink(c, c->lane_seq);
    const uint32_t nshards = KS.nshards, seg_cap = KS.seg_cap;
    hipStream_t st = c->stream;
    uint32_t *out_ids = c->ext_out_ids ? c->ext_out_ids : c->d_out_ids.p; float *out_mats = c->ext_out_mats ? c->ext_out_mats : c->d_out_mats.p;
    uint32_t out_cap = c->ext_out_ids ? c->ext_out_cap : c->out_cap;
    if (c->nslots <= COUNT_SLOTS_MAX && c->d_gcount.p) {
        // one launch: the counts per (cursor shard, group slot) come from the scan itself (counted_by_scan) or from a counting pass over the instance list
        const uint32_t par = c->large_seq & 1u; const size_t words = (size_t)CURSOR_SHARDS * std::max(c->nslots, 1u);
        if (!counted_by_scan) {
            int rc = prepare_group_counts(c, par); if (rc != RE_OK) return rc;
            const uint32_t parts = std::max(1u, std::min(64u, (c->pred_total / nshards + 2047u) / 2048u));
            hipLaunchKernelGGL(k_emit_count_sharded, dim3(nshards * parts), dim3(256), (size_t)std::max(c->nslots, 1u) * 4, st, hdr, KS.item_slot, nshards, seg_cap, c->d_gcount.p + par * words, c->nslots, c->d_spec.p);
        }
        PackLargeArgs A{}; A.hdr = hdr; A.hdr_next = hdr_next; A.th = c->d_th.p; A.gcount = c->d_gcount.p + par * words; A.gfill = c->d_gfill.p + par * words;
        A.zero_a = c->d_gcount.p + (par ^ 1u) * words; A.zero_b = c->d_gfill.p + (par ^ 1u) * words; A.zero_words = (uint32_t)words;
        A.nslots = c->nslots; A.out_cap = out_cap; A.range_cap = c->nslots; A.frame = c->frame; A.item_row = KS.item_row; A.item_slot = KS.item_slot; A.nshards = nshards; A.seg_cap = seg_cap;
        A.row_id = c->d_id.p; A.row_mat = c->d_mat.p; A.out_ids = out_ids; A.out_mats = out_mats; A.gc_model = c->d_gc_model.p; A.gc_rs = c->d_gc_rs.p; A.gc_sort = c->d_gc_sort.p;
        A.ranges = c->d_hranges; A.hres = c->d_hres; A.spec = c->d_spec.p; A.out_count = c->ext_out_count;
        // workgroup b takes tiles b >> 3, (b >> 3) + grid / 8, ... of cursor shard b & 7: enough rounds of 8 workgroups for the predicted shard length
        // (+25 %; a longer shard makes its workgroups loop, any grid that is a multiple of 8 is correct)
        const uint32_t per_shard = (c->pred_total + c->pred_total / 4u) / nshards + PACK_LARGE_TILE;
        const uint32_t grid = CURSOR_SHARDS * std::min(2048u, std::max(4u, (per_shard + PACK_LARGE_TILE - 1u) / PACK_LARGE_TILE));   // (always a multiple of 8: workgroup b serves segment b & 7, also when only segment 0 exists)
        hipEvent_t ta = nullptr, tb = nullptr;
        if (c->k1_timing && c->k1_kind == RE_TIME_PACK_LARGE) take_timing_events(c, &ta, &tb);
        hipExtLaunchKernelGGL(k_pack_large, dim3(grid), dim3(PACK_LARGE_THREADS), 0, st, ta, tb, 0, A);
        HIPCHK(c, hipGetLastError());
        c->last_pack.kind = 2; c->last_pack.L = A; c->last_pack.grid = grid; c->last_pack.par = par; c->last_pack.hdr = hdr; c->last_pack.hdr_next = hdr_next;
        c->gc_dirty[par] = true; c->gc_dirty[par ^ 1u] = false;                // this frame's arrays stay as they are; the other parity's were cleared by the launch
        c->large_seq++;
        return RE_OK;
    }
    uint32_t grid = std::min(2048u, (c->item_cap + 255u) / 256u);
    size_t lds = c->nslots <= LDS_HIST_SLOTS ? (size_t)std::max(c->nslots, 1u) * 4 : 4;
    // few workgroups for the count: every workgroup flushes its LDS histogram with one global atomic per non-empty group,
    // and a handful of hot (model, LOD) groups saturate near 88 atomics/us per address
    hipLaunchKernelGGL(k_emit_count, dim3(std::min(grid, 256u)), dim3(256), lds, st, hdr, KS.item_slot, nshards, seg_cap, c->d_group_count.p, c->nslots, c->d_spec.p);
    hipLaunchKernelGGL(k_group_scan, dim3(1), dim3(1024), 0, st, c->d_group_count.p, c->d_group_begin.p, c->d_group_fill.p, c->nslots, c->d_gc_model.p, c->d_gc_rs.p, c->d_gc_sort.p,
                       c->d_hranges, c->nslots, hdr, hdr_next, c->d_th.p, c->d_hres, c->d_spec.p, c->ext_out_count, c->ext_out_ids ? c->ext_out_cap : c->out_cap, c->frame, seg_cap);
    hipLaunchKernelGGL(k_emit_scatter, dim3(grid), dim3(256), lds, st, hdr, KS.item_row, KS.item_slot, nshards, seg_cap, c->d_group_begin.p, c->d_group_fill.p, c->nslots,
                       c->d_id.p, c->d_mat.p, out_ids, out_mats, out_cap, c->d_spec.p);
    HIPCHK(c, hipGetLastError());
    c->last_pack.kind = 3; c->last_pack.hdr = hdr; c->last_pack.hdr_next = hdr_next;
    return RE_OK;
}

static int issue_cull(re_ctx *c, const re_camera *cam, uint32_t flags);
static int finish_cull(re_ctx *c, re_visible *out) {
    { int rc = drain_other_lane(c); if (rc != RE_OK) return rc; }
    { int rc = flush_deferred_pack(c); if (rc != RE_OK) return rc; }
    // Fast completion: the pack publishes "frame done" into mapped host memory after the group table and the counters; polling that
    // word costs a PCIe write's latency instead of the driver's stream-synchronise wake-up.  The packed instances are device
    // resident, and whatever reads them next is ordered behind the pack on the stream.  Not when a tick that may leave the tree
    // stale is in flight: that needs resolve().
    bool done = false;
    const double t_wait0 = g_issue_clock.on ? IssueClock::now() : 0.0;
    if (!(c->tick_inflight && c->ndyn)) {
        const volatile uint32_t *flag = &c->h_res->done_frame;
        const auto t0 = std::chrono::steady_clock::now();
        for (uint32_t spins = 0; !(done = (*flag == c->frame)); spins++)
            if ((spins & 1023u) == 1023u && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(2)) break;   // long frame: let the driver wait
        if (done) { std::atomic_thread_fence(std::memory_order_acquire); if (c->h_res->overflow == 2u || (c->h_spec && c->h_spec->stale)) done = false; else c->pending.clear(); }
    }
    if (!done) { int rc = resolve(c); if (rc != RE_OK) return rc; }
    const double t_wait1 = g_issue_clock.on ? IssueClock::now() : 0.0; g_issue_clock.wait += t_wait1 - t_wait0;
    c->cull_inflight = false; c->lane_busy = false;
    if (c->h_res->overflow == 1) {
        // k_pack_small declined (visible set larger than predicted): run the multi-kernel pack on this frame's entries
        int rc = launch_pack_large(c, frame_header(c, c->lane_seq), frame_header(c, c->lane_seq + 2u));
        if (rc != RE_OK) return rc;
        HIPCHK(c, sync_stream(c->stream));
    }
    if (c->h_res->overflow == RESULT_SEGMENT_OVERFLOW) {
        // A cursor segment of the instance list overflowed: nothing was packed.  Redo the frame with the list as one segment (see re_ctx::single_shard).
        if (c->single_shard) return c->fail(RE_E_CAPACITY, "instance list overflow (%u instances reserved, list capacity %u)", c->h_res->n_items, c->item_cap);
        c->single_shard = true; c->n_segment_redos++;
        HIPCHK(c, hipMemsetAsync(c->d_hdr.p, 0, NUM_FRAME_HEADERS * sizeof(FrameHeader), c->stream)); c->th_clean = false;
        if (c->d_gcount.p) {
            HIPCHK(c, hipMemsetAsync(c->d_gcount.p, 0, c->d_gcount.n * 4, c->stream)); HIPCHK(c, hipMemsetAsync(c->d_gfill.p, 0, c->d_gfill.n * 4, c->stream));
            c->gc_dirty[0] = c->gc_dirty[1] = false;
        }
        c->pending.clear();
        const re_camera cam = c->last_cam;
        int rc = issue_cull(c, &cam, c->last_cull_flags & ~(RE_CULL_ASYNC | RE_CULL_DEFER_PACK | RE_CULL_TWO_LANES));
        if (rc != RE_OK) return rc;
        return finish_cull(c, out);
    }
    c->timings_pending = c->timed_frame;                                      // the events are read in re_get_timings (they may still be in flight here)
    c->pred_total = c->h_res->total; c->pred_candidates = c->h_res->n_candidates;
    {   // The group table carries a seal: a hash over every table word, tied to the frame number and the counts (result_seal).  With the
        // publication protocol of publish_to_host the block is complete when "frame done" is visible, so the check below passes at first
        // sight; n_seal_waits / n_sync_fallbacks (re_stats) count the times it did not, and the GPU tests assert that both stay 0.
        // Every instance the cull reserved must also have been counted into a group (dead rows excepted): otherwise a cursor segment overflowed.
        auto table_ok = [&]() -> bool {
            uint32_t n = 0, hsh = 0; const volatile HostResult *hr = c->h_res; const volatile uint32_t *wds = reinterpret_cast<const volatile uint32_t *>(c->h_ranges);
            const uint32_t ng = std::min((uint32_t)hr->n_groups, c->nslots), nw = (uint32_t)(sizeof(InstanceRange) / 4u);
            for (uint32_t g = 0; g < ng; g++) {
                for (uint32_t k = 0; k < nw; k++) hsh ^= table_word_hash(wds[g * nw + k], g * nw + k);
                n += wds[g * nw + 4u];                                         // InstanceRange::count
            }
            if (hr->table_hash && result_seal(hsh | 1u, c->frame, hr->n_groups, hr->total, hr->n_vis_map, hr->n_vis_vec, hr->n_items) != hr->table_hash) return false;
            return n == hr->total;
        };
        if (!table_ok()) {
            c->n_seal_waits++;
            const auto t0 = std::chrono::steady_clock::now();
            bool ok = false;
            while (!(ok = table_ok()) && std::chrono::steady_clock::now() - t0 < std::chrono::microseconds(200)) {}
            if (!ok) { c->n_sync_fallbacks++; HIPCHK(c, sync_stream(c->stream)); std::atomic_thread_fence(std::memory_order_acquire); ok = table_ok(); }
            if (!ok) return c->fail(RE_E_STATE, "group table inconsistent with its seal (frame %u, %u groups, %u instances)", c->frame, c->h_res->n_groups, c->h_res->total);
        }
        if (c->h_res->n_items > c->h_res->total + c->n_dead + (uint32_t)c->h_uncached.size() + 2u * c->n_phantom) return c->fail(RE_E_CAPACITY, "instance-list segment overflow (%u reserved, %u packed)", c->h_res->n_items, c->h_res->total);
    }
    if (c->h_res->n_items > c->item_cap) return c->fail(RE_E_CAPACITY, "instance expansion capacity exceeded (%u > %u)", c->h_res->n_items, c->item_cap);
    fill_visible(c, out);
    if (g_issue_clock.on) g_issue_clock.finish += IssueClock::now() - t_wait1;
    return RE_OK;
}

// enqueue one frame's cull + pack (no synchronisation)
// ---- frame lanes (RE_CULL_TWO_LANES) ----
static void switch_lane(re_ctx *c) {
    re_ctx::LanePark &k = c->park;
    std::swap(c->stream, k.stream); std::swap(c->d_hdr, k.d_hdr); std::swap(c->d_item_row, k.d_item_row); std::swap(c->d_item_slot, k.d_item_slot);
    std::swap(c->d_out_ids, k.d_out_ids); std::swap(c->d_out_mats, k.d_out_mats); std::swap(c->d_cell_stamp, k.d_cell_stamp); std::swap(c->d_params, k.d_params);
    std::swap(c->h_res, k.h_res); std::swap(c->d_hres, k.d_hres); std::swap(c->h_ranges, k.h_ranges); std::swap(c->d_hranges, k.d_hranges);   // (h_block stays: lane 0's block also holds the tick counters and the speculation word)
    std::swap(c->deferred_pack, k.deferred_pack); std::swap(c->deferred, k.deferred); std::swap(c->deferred_grid, k.deferred_grid);
    std::swap(c->lane_seq, k.lane_seq); std::swap(c->lane_busy, k.busy);
    c->lane_id ^= 1u; c->n_lane_switches++;
}
// second lane: same sizes as the first one's per-frame buffers
static int ensure_second_lane(re_ctx *c) {
    if (c->park_ready) return RE_OK;
    re_ctx::LanePark &k = c->park; uint64_t *acct = &c->dev_bytes;
    HIPCHK(c, hipStreamCreateWithFlags(&k.stream, hipStreamNonBlocking));
    HIPCHK(c, k.d_hdr.alloc(NUM_FRAME_HEADERS, acct)); HIPCHK(c, k.d_item_row.alloc((size_t)c->item_cap * 2, acct)); HIPCHK(c, k.d_item_slot.alloc((size_t)c->item_cap * 2, acct));
    HIPCHK(c, k.d_out_ids.alloc(c->out_cap, acct)); HIPCHK(c, k.d_out_mats.alloc((size_t)c->out_cap * 16, acct));
    HIPCHK(c, k.d_cell_stamp.alloc(std::max<size_t>(c->d_cell_stamp.n, 1), acct)); HIPCHK(c, k.d_params.alloc(1, acct));
    HIPCHK(c, hipMemset(k.d_hdr.p, 0, NUM_FRAME_HEADERS * sizeof(FrameHeader))); HIPCHK(c, hipMemset(k.d_item_row.p, 0, (size_t)c->item_cap * 8)); HIPCHK(c, hipMemset(k.d_item_slot.p, 0xFF, (size_t)c->item_cap * 8));
    HIPCHK(c, hipMemset(k.d_cell_stamp.p, 0, k.d_cell_stamp.n * 4));
    {
        void *hb = nullptr, *db = nullptr;
        HIPCHK(c, alloc_host_block(c->nslots_cap, &hb, &db));
        k.h_block = hb;
        k.h_res = hb_at<HostResult>(hb, HB_RES); k.d_hres = hb_at<HostResult>(db, HB_RES);
        k.h_ranges = hb_at<InstanceRange>(hb, HB_RANGES); k.d_hranges = hb_at<InstanceRange>(db, HB_RANGES);
    }
    k.deferred_pack = false; k.lane_seq = 0; k.busy = false;
    c->park_ready = true;
    return RE_OK;
}
static void free_second_lane(re_ctx *c) {
    if (c->lane_id) switch_lane(c);                                          // lane 0 is the one the rest of the context owns
    re_ctx::LanePark &k = c->park; uint64_t *acct = &c->dev_bytes;
    if (k.stream) { (void)sync_stream(k.stream); (void)hipStreamDestroy(k.stream); k.stream = nullptr; }
    k.d_hdr.release(acct); k.d_item_row.release(acct); k.d_item_slot.release(acct); k.d_out_ids.release(acct); k.d_out_mats.release(acct); k.d_cell_stamp.release(acct); k.d_params.release(acct);
    if (k.h_block) { (void)hipHostFree(k.h_block); k.h_block = nullptr; }
    k.h_res = nullptr; k.d_hres = nullptr; k.h_ranges = nullptr; k.d_hranges = nullptr;
    k.deferred_pack = false; k.busy = false; c->park_ready = false;
}
static int flush_deferred_pack(re_ctx *c);
// Everything except the alternating frames themselves works on one lane: send the other lane's pending pack off and wait for its stream.
// The current lane holds the newest frame (lanes are switched before a frame is issued), so results, stamps and frame parameters
// of "the last frame" are those of the current lane afterwards.
static int drain_other_lane(re_ctx *c) {
    if (!c->park_ready || !(c->park.busy || c->park.deferred_pack)) return RE_OK;
    switch_lane(c);
    int rc = flush_deferred_pack(c);
    hipError_t e = sync_stream(c->stream);
    c->lane_busy = false;
    switch_lane(c);
    if (rc != RE_OK) return rc;
    HIPCHK(c, e);
    return RE_OK;
}

// A pack deferred by RE_CULL_DEFER_PACK that no later frame picked up: launch it on its own (anything that needs the frame's result, or
// is about to change what the pack reads, calls this first).
static int flush_deferred_pack(re_ctx *c) {
    if (!c->deferred_pack) return RE_OK;
    c->deferred_pack = false;
    const FusedPack &F = c->deferred;
    hipLaunchKernelGGL(k_pack_small, dim3(c->deferred_grid), dim3(256), (size_t)std::max(c->nslots, 1u) * 8, c->stream, F.hdr, F.hdr_next, F.th, F.A, F.K, F.nrows);
    HIPCHK(c, hipGetLastError());
    return RE_OK;
}

static int issue_cull(re_ctx *c, const re_camera *cam, uint32_t flags) {
    hipStream_t st = c->stream;
    c->last_cam = *cam; c->last_cull_flags = flags;
    IssueClock &IC = g_issue_clock; double t_ic = IC.on ? IssueClock::now() : 0.0;
    auto lap = [&](double &acc) { if (IC.on) { const double t = IssueClock::now(); acc += t - t_ic; t_ic = t; } };
    if (c->cull_inflight && c->h_res->overflow == 0) { c->pred_total = std::max(c->pred_total, c->h_res->total); c->pred_candidates = std::max(c->pred_candidates, c->h_res->n_candidates); }   // hint from an earlier async frame, if it has landed
    c->frame += 1;
    make_frame_params(c, cam, flags);
    const FrameParams &P = c->P;
    c->lane_seq += 1;                                                           // frames issued on this lane: rotates its headers and instance lists
    FrameHeader *hdr = frame_header(c, c->lane_seq), *hdr_next = frame_header(c, c->lane_seq + 2u);     // the pack clears the header of the frame after next
    c->timed_frame = c->timings_on && !(flags & RE_CULL_ASYNC);
    if (c->timed_frame) HIPCHK(c, hipEventRecord(c->ev[0], st));
    if (c->dirty_pending) {
        if (c->ncells) hipLaunchKernelGGL(k_static_cache_cells, dim3((c->ncells + 255) / 256), dim3(256), 0, st, c->ncells, c->d_cell_tight.p, c->d_cell_flags.p, P);
        if (c->nsh) hipLaunchKernelGGL(k_static_cache_shared, dim3((c->nsh + 255) / 256), dim3(256), 0, st, c->nsh, c->d_sh_cells.p, c->d_cell_key.p, c->d_sh_aabb.p, c->d_sh_dirty.p, c->d_sh_owner.p, c->d_sh_cached.p, P);
        // the changed-static set is consumed by every render until the frame ends (re_tick clears it)
    }
    hipEvent_t k1a = nullptr, k1b = nullptr;
    if (c->k1_timing && c->k1_kind == RE_TIME_SCAN) take_timing_events(c, &k1a, &k1b);
    uint32_t *out_ids = c->ext_out_ids ? c->ext_out_ids : c->d_out_ids.p; float *out_mats = c->ext_out_mats ? c->ext_out_mats : c->d_out_mats.p;
    uint32_t out_cap = c->ext_out_ids ? c->ext_out_cap : c->out_cap;
    bool small = c->nslots <= LDS_HIST_SLOTS && c->nsh <= 65536u && (uint64_t)c->pred_total * 2u <= PACK_SMALL_ITEMS && !(flags & RE_CULL_FORCE_LARGE_PACK);
    PackArgs A{}; A.nslots = c->nslots; A.out_cap = out_cap; A.row_id = c->d_id.p; A.row_mat = c->d_mat.p; A.out_ids = out_ids; A.out_mats = out_mats;
    A.gc_model = c->d_gc_model.p; A.gc_rs = c->d_gc_rs.p; A.gc_sort = c->d_gc_sort.p; A.ranges = c->d_hranges; A.hres = c->d_hres; A.spec = c->d_spec.p; A.out_count = c->ext_out_count; A.frame = c->frame;
    // K1: key scan + candidate cull + instance expansion in one launch (the dominant kernel).  hipExtLaunchKernelGGL ties the two
    // timing events to this dispatch's own begin/end timestamps.
    uint32_t scan_grid = std::max(1u, (c->nlists + (CULL_THREADS / 64) - 1) / (CULL_THREADS / 64));
    ScanCullArgs SA; SA.B = c->PB; SA.B32 = c->PB32; SA.cell_key64 = c->d_cell_key.p; SA.P = P; SA.P_dev = c->d_params.p;
    SA.cell_tight = c->d_cell_tight.p; SA.cell_begin = c->d_cell_begin.p; SA.cell_nlocal = c->d_cell_nlocal.p; SA.cell_nstatic = c->d_cell_nstatic.p; SA.cell_nghost = c->d_cell_nghost.p;
    SA.cell_flags = c->d_cell_flags.p; SA.cell_stamp = c->d_cell_stamp.p; SA.K = item_sink(c, c->lane_seq); SA.hdr = hdr; SA.S = shared_arrays(c); SA.spec = c->d_spec.p;
    // a large visible set is expected: the scan counts the instances per (cursor shard, group slot) while it expands them, so the pack is one launch
    const bool count_in_scan = !small && c->nslots <= COUNT_SLOTS_MAX && c->d_gcount.p != nullptr;
    size_t scan_lds = 0;
    if (count_in_scan) {
        const uint32_t par = c->large_seq & 1u;
        int rc = prepare_group_counts(c, par); if (rc != RE_OK) return rc;
        SA.K.group_count = c->d_gcount.p + (size_t)par * CURSOR_SHARDS * std::max(c->nslots, 1u); SA.K.count_nslots = c->nslots;
        scan_lds = (size_t)(CULL_THREADS / 64) * c->nslots * 4;
    }
    static_assert(alignof(ScanCullArgs) == 8, "SCAN_CULL_ARGS_OFFSET assumes 8-byte alignment");
#ifdef RE_EXP_STAMPS
    if (!c->d_timeline.p) HIPCHK(c, c->d_timeline.alloc((size_t)scan_grid * 4 * 8, nullptr));
    SA.timeline = c->d_timeline.p;
#endif
    // Probe path (RE_CFG_PROBE): when the candidate boxes hold far fewer cells than the table has sections, look the cells up
    // instead of streaming the keys
    bool probed = false;
    if ((c->cfg.flags & RE_CFG_PROBE) && c->d_htab.p && !(flags & RE_CULL_FORCE_STREAM)) {
        ProbeArgs Q{}; Q.tab = c->d_htab.p; Q.mask = c->htab_mask;
        uint64_t waves = 0; bool ok = true;
        for (uint32_t l = 0; l < (uint32_t)MAX_LEVELS; l++) {
            Q.wave0[l] = (uint32_t)waves;
            LevelBox u{}; u.level_length = P.box[0][l].level_length;
            if (l < P.max_level) {
                const LevelBox &a = P.box[0][l], &b = P.box[1][l];
                const bool ea = !a.nx || !a.ny || !a.nz, eb = !b.nx || !b.ny || !b.nz;
                auto lo = [&](uint32_t x, uint32_t y) { return ea ? y : eb ? x : std::min(x, y); };
                auto hi = [&](uint64_t x, uint64_t y) { return ea ? y : eb ? x : std::max(x, y); };
                if (!(ea && eb)) {
                    u.bx = lo(a.bx, b.bx); u.by = lo(a.by, b.by); u.bz = lo(a.bz, b.bz);
                    const uint64_t ex = hi((uint64_t)a.bx + a.nx, (uint64_t)b.bx + b.nx) - u.bx, ey = hi((uint64_t)a.by + a.ny, (uint64_t)b.by + b.ny) - u.by, ez = hi((uint64_t)a.bz + a.nz, (uint64_t)b.bz + b.nz) - u.bz;
                    if (ex > 65536u || ey > 65536u || ez > 65536u || ex * ey * ez > (1ull << 31)) ok = false;     // (u16 wrap-around of the ids: leave it to the stream)
                    else { u.nx = (uint32_t)ex; u.ny = (uint32_t)ey; u.nz = (uint32_t)ez; waves += (ex * ey * ez + PROBE_KEYS - 1) / PROBE_KEYS; }
                }
            }
            Q.ubox[l] = u;
        }
        Q.wave0[MAX_LEVELS] = (uint32_t)waves; Q.nwaves = (uint32_t)waves;
        if (ok && waves && (waves * PROBE_KEYS * 4u <= (uint64_t)c->ncells + 4096u || ((c->cfg.flags & RE_CFG_PROBE_ALWAYS) && waves <= (1u << 20)))) {
            const uint32_t wgs = (uint32_t)((waves + (CULL_THREADS / 64) - 1) / (CULL_THREADS / 64));
            const uint32_t grid = std::max(std::max(wgs, 1u), std::min((c->nsh + CULL_THREADS - 1) / CULL_THREADS, 2048u));
            hipExtLaunchKernelGGL(k_probe_cull, dim3(grid), dim3(CULL_THREADS), 0, st, k1a, k1b, 0, Q, SA);
            probed = true; c->probe_frames++;
        }
    }
    lap(IC.params);
    const ScanSpans SP = probed ? ScanSpans{} : candidate_spans(c, scan_grid);
    lap(IC.spans);
    // a pack deferred by the previous frame rides in the first workgroups of this frame's scan (one launch per frame); any other
    // kind of launch here sends it off on its own first
    const bool fuse = c->deferred_pack && !probed && c->deferred_grid < (1u << 20);
    if (c->deferred_pack && !fuse) { int rc = flush_deferred_pack(c); if (rc != RE_OK) return rc; }
    const size_t fused_lds = (size_t)std::max(c->nslots, 1u) * 8;
    // a synchronous frame with a small visible set: the scan's last workgroup publishes the result itself (k_scan_cull_sync), the pack launch behind it only moves the instances
    static const bool always_one = getenv("RE_EXP_ONE_LAUNCH_SYNC") != nullptr;       // (testing: every eligible synchronous frame of the process)
    const bool one_launch = ((flags & RE_CULL_ONE_LAUNCH) || always_one) && small && !(flags & RE_CULL_ASYNC) && !probed && !fuse && !count_in_scan && c->nslots <= SYNC_TAIL_SLOTS;
    if (one_launch) SA.K.slot_write_through = 1u;
    if (probed) {}
    else if (one_launch && c->key32)
        hipExtLaunchKernelGGL(k_scan_cull_sync<true>, dim3(scan_grid), dim3(CULL_THREADS), (size_t)std::max(c->nslots, 1u) * 4, st, k1a, k1b, 0, (const void *)c->d_cell_key32.p, c->ncells, SP.n, SP.start[0], SP.count[0],
                              SP.start[1], SP.count[1], SP.start[2], SP.count[2], SP.start[3], SP.count[3], (const uint32_t *)c->d_chunk_level.p, SA, A);
    else if (one_launch)
        hipExtLaunchKernelGGL(k_scan_cull_sync<false>, dim3(scan_grid), dim3(CULL_THREADS), (size_t)std::max(c->nslots, 1u) * 4, st, k1a, k1b, 0, (const void *)c->d_cell_key.p, c->ncells, SP.n, SP.start[0], SP.count[0],
                              SP.start[1], SP.count[1], SP.start[2], SP.count[2], SP.start[3], SP.count[3], (const uint32_t *)c->d_chunk_level.p, SA, A);
    else if (fuse && c->key32)
        hipExtLaunchKernelGGL(k_scan_cull_fused<true>, dim3(scan_grid + c->deferred_grid), dim3(CULL_THREADS), fused_lds, st, k1a, k1b, 0, (const void *)c->d_cell_key32.p, c->ncells, SP.n | (c->deferred_grid << 8),
                              SP.start[0], SP.count[0], SP.start[1], SP.count[1], SP.start[2], SP.count[2], SP.start[3], SP.count[3], (const uint32_t *)c->d_chunk_level.p, SA, c->deferred);
    else if (fuse)
        hipExtLaunchKernelGGL(k_scan_cull_fused<false>, dim3(scan_grid + c->deferred_grid), dim3(CULL_THREADS), fused_lds, st, k1a, k1b, 0, (const void *)c->d_cell_key.p, c->ncells, SP.n | (c->deferred_grid << 8),
                              SP.start[0], SP.count[0], SP.start[1], SP.count[1], SP.start[2], SP.count[2], SP.start[3], SP.count[3], (const uint32_t *)c->d_chunk_level.p, SA, c->deferred);
    else if (c->key32 && !small)      // a large visible set is expected: the schedule that has a wave's requests in flight together (k_scan_cull_wide)
        hipExtLaunchKernelGGL(k_scan_cull_wide, dim3(scan_grid), dim3(CULL_THREADS), scan_lds, st, k1a, k1b, 0, (const void *)c->d_cell_key32.p, c->ncells, SP.n, SP.start[0], SP.count[0],
                              SP.start[1], SP.count[1], SP.start[2], SP.count[2], SP.start[3], SP.count[3], (const uint32_t *)c->d_chunk_level.p, SA);
    else if (c->key32)
        hipExtLaunchKernelGGL(k_scan_cull<true>, dim3(scan_grid), dim3(CULL_THREADS), scan_lds, st, k1a, k1b, 0, (const void *)c->d_cell_key32.p, c->ncells, SP.n, SP.start[0], SP.count[0],
                              SP.start[1], SP.count[1], SP.start[2], SP.count[2], SP.start[3], SP.count[3], (const uint32_t *)c->d_chunk_level.p, SA);
    else
        hipExtLaunchKernelGGL(k_scan_cull<false>, dim3(scan_grid), dim3(CULL_THREADS), scan_lds, st, k1a, k1b, 0, (const void *)c->d_cell_key.p, c->ncells, SP.n, SP.start[0], SP.count[0],
                              SP.start[1], SP.count[1], SP.start[2], SP.count[2], SP.start[3], SP.count[3], (const uint32_t *)c->d_chunk_level.p, SA);
    if (fuse) { c->deferred_pack = false; c->n_fused_frames++; }
    HIPCHK(c, hipGetLastError());
    lap(IC.scan);
    if (c->timed_frame) HIPCHK(c, hipEventRecord(c->ev[1], st));
    if (small) {
        // workgroup b packs 64 instances of cursor shard b & 7: enough rounds of 8 workgroups for the predicted shard length (+50 %),
        // and never fewer than cover 16 K instances spread evenly would need is not required -- a shard longer than the grid covers
        // makes the pack decline (overflow) and the frame is redone through the large path
        uint32_t per_shard = (c->pred_total + c->pred_total / 2u) / (c->single_shard ? 1u : CURSOR_SHARDS) + 64u;
        uint32_t pgrid = CURSOR_SHARDS * std::min(32u, (per_shard + 63u) / 64u);
        // RE_CULL_DEFER_PACK: in a world without dynamic entities nothing changes what the pack reads before the next visibility query,
        // so an asynchronous frame may leave its pack to the launch of the next one (k_scan_cull_fused)
        if ((flags & RE_CULL_DEFER_PACK) && (flags & RE_CULL_ASYNC) && c->ndyn == 0 && !c->dirty_pending && !c->comm.comm) {
            FusedPack F{}; F.hdr = hdr; F.hdr_next = hdr_next; F.th = c->d_th.p; F.A = A; F.K = item_sink(c, c->lane_seq); F.nrows = c->ghost_base + c->ghost_cap;
            c->deferred = F; c->deferred_grid = pgrid; c->deferred_pack = true;
        } else {
            PackArgs Am = A; if (one_launch) Am.flags |= PACK_NO_PUBLISH;
            hipLaunchKernelGGL(k_pack_small, dim3(pgrid), dim3(256), (size_t)std::max(c->nslots, 1u) * 8, st, hdr, hdr_next, c->d_th.p, Am, item_sink(c, c->lane_seq), c->ghost_base + c->ghost_cap);
            c->last_pack.kind = 1; c->last_pack.hdr = hdr; c->last_pack.hdr_next = hdr_next; c->last_pack.A = A; c->last_pack.K = item_sink(c, c->lane_seq); c->last_pack.grid = pgrid; c->last_pack.nrows = c->ghost_base + c->ghost_cap;
        }
    } else {
        int rc = launch_pack_large(c, hdr, hdr_next, count_in_scan);
        if (rc != RE_OK) return rc;
    }
    HIPCHK(c, hipGetLastError());
    lap(IC.pack); if (IC.on) IC.n++;
    if (c->timed_frame) HIPCHK(c, hipEventRecord(c->ev[2], st));
    c->have_cull = true; c->cull_inflight = true; c->th_clean = true;
    c->pending.push_back(re_ctx::PendingCall{ 0, c->frame, *cam, flags, 0.f, c->ext_out_ids, c->ext_out_mats, c->ext_out_cap, c->ext_out_count });
    return RE_OK;
}

static int resolve(re_ctx *c);
extern "C" int re_cull_pack(re_ctx *c, const re_camera *cam, uint32_t flags, re_visible *out) try {
    if (!c) return RE_E_ARG;
    if (!cam) return c->fail(RE_E_ARG, "re_cull_pack: camera is NULL");
    if (!c->h_res) return c->fail(RE_E_STATE, "re_cull_pack: no world uploaded");
    HIPCHK(c, hipSetDevice(c->device));
    // Movers of the previous tick may change the section table.  A synchronous call waits for that tick; an asynchronous one is
    // enqueued speculatively: if the tick does find movers, this frame's kernels cancel themselves and resolve() replays it.
    if (!(flags & RE_CULL_ASYNC) && c->tick_inflight && c->ndyn) { int rc = finish_tick(c, nullptr); if (rc != RE_OK) return rc; }
    {   // frame lanes: a deferred asynchronous frame of a static world may go to the other lane (its launch then overlaps the previous
        // frame's on the GPU); anything else runs on one lane only and waits for the other one first
        const uint32_t need = RE_CULL_ASYNC | RE_CULL_DEFER_PACK | RE_CULL_TWO_LANES;
        const bool small = c->nslots <= LDS_HIST_SLOTS && c->nsh <= 65536u && (uint64_t)c->pred_total * 2u <= PACK_SMALL_ITEMS && !(flags & RE_CULL_FORCE_LARGE_PACK);
        if ((flags & need) == need && c->ndyn == 0 && !c->dirty_pending && small && c->have_cull && !c->comm.comm) {
            int rc = ensure_second_lane(c); if (rc != RE_OK) return rc;
            switch_lane(c);
            c->lane_busy = true;
        } else { int rc = drain_other_lane(c); if (rc != RE_OK) return rc; }
    }
    if (c->comm.comm) {                                                       // multi-GPU exchange: this frame packs straight into its send slab
        const int b = (int)(c->comm.seq & 1u); uint32_t *sl = c->comm.slab[b].p;
        c->ext_out_count = sl; c->ext_out_ids = sl + 16; c->ext_out_mats = reinterpret_cast<float *>(sl + 16 + c->comm.cap); c->ext_out_cap = c->comm.cap;
        c->comm.last = b; c->comm.seq++;
    }
    int rc = issue_cull(c, cam, flags);
    if (rc != RE_OK || (flags & RE_CULL_ASYNC)) return rc;
    return finish_cull(c, out);
} RE_ABI_GUARD(c, "re_cull_pack")

// slot of a world section in the resident table, -1 when it does not exist (base = last full build, overlay = created since)
static int32_t find_slot(const re_ctx *c, uint64_t key) {
    auto e = c->extra_slots.find(key);
    if (e != c->extra_slots.end()) return c->h_cell_key[e->second] == key ? (int32_t)e->second : -1;
    if (c->base_index.empty()) return -1;
    // two-level binary search: the block index stays in cache, the second level touches a few lines of the 8 B/section key array
    size_t blk = std::upper_bound(c->base_index.begin(), c->base_index.end(), key) - c->base_index.begin();
    if (blk == 0) return -1;
    const size_t lo = (blk - 1) * 1024, hi = std::min(lo + 1024, c->base_keys.size());
    auto o = std::lower_bound(c->base_keys.begin() + lo, c->base_keys.begin() + hi, key);
    if (o != c->base_keys.begin() + hi && *o == key) { size_t sl = o - c->base_keys.begin(); if (c->h_cell_key[sl] == key) return (int32_t)sl; }
    return -1;
}

// ------------------------------------------------------------------------------------------------
// patch_sections: the result of a re-bucket written INTO the resident section table instead of rebuilding it -- cost
// proportional to the sections the movers touched, not to the world.  Sections keep their slot; a section that gains
// entities beyond its segment's capacity is relocated to the end of the row pool; a new section takes a free (padding or
// emptied) slot of its level run, so every 512-key chunk of the scan stays level-uniform; an emptied, unlinked section
// becomes a padding slot.  The (small) shared-section table is rebuilt whole.  Only changed sections are re-folded and
// re-flagged, exactly what end_of_changes / update_static_world_sections touch (bounding_box_tree_v2.rs:1055-1213).
// Returns 1 when the slack is exhausted (no free slot of a level, row pool full): the caller rebuilds from scratch.
// ------------------------------------------------------------------------------------------------
static int patch_sections(re_ctx *c, const Carry &carry, const std::map<uint64_t, std::vector<uint32_t>> &arrive, const std::vector<uint32_t> &removed_rows) {
    hipStream_t st = c->stream;
    static const bool timing = getenv("RE_EXP_TIME_REBUCKET") != nullptr;
    auto t_begin = std::chrono::steady_clock::now(); auto lap = [&](const char *what) { if (timing) { auto t = std::chrono::steady_clock::now(); fprintf(stderr, "    patch %-10s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(t - t_begin).count()); t_begin = t; } };
    auto is_pad = [](uint64_t k) { return (k & 0xFFFFFFFFFFFFull) == 0xFFFFFFFFFFFFull; };
    auto by_id = [&](uint32_t a, uint32_t b) { return c->h_id[a] < c->h_id[b]; };
    // find_slot costs ~0.25 us a call (a hash probe + a two-level search that misses the cache in the 80 MB key array) and a patch asks for the slot of
    // every affected section several times: remembered per key for the duration of the patch (the two places that change a section's slot update the memo)
    std::unordered_map<uint64_t, int32_t> slot_memo; slot_memo.reserve(carry.changed_cells.size() * 2 + 64);
    auto slot_of = [&](uint64_t K) -> int32_t { auto it = slot_memo.find(K); if (it != slot_memo.end()) return it->second; const int32_t sl = find_slot(c, K); slot_memo.emplace(K, sl); return sl; };
    // ---- A. shared-section table from the per-row decisions (surviving sections keep their creation order)
    std::vector<SharedRec> shrec; shrec.reserve(c->h_row_shared_keys.size());
    for (auto &kv : c->h_row_shared_keys) { SharedRec sr; sr.row = kv.first; sr.nk = c->h_row_nk[kv.first]; memcpy(sr.keys, kv.second.data(), sizeof sr.keys); shrec.push_back(sr); }
    std::sort(shrec.begin(), shrec.end(), [](const SharedRec &a, const SharedRec &b) { return a.row < b.row; });
    std::map<SharedIdPub, uint32_t> shmap; std::vector<SharedIdPub> shids; std::vector<std::vector<uint32_t>> sh_act, sh_sta;
    {
        std::set<SharedIdPub> alive;
        for (const SharedRec &sr : shrec) { SharedIdPub id; id.nk = sr.nk; memcpy(id.keys, sr.keys, sizeof id.keys); alive.insert(id); }
        for (const SharedIdPub &id : c->h_shids) if (alive.count(id)) { shmap.emplace(id, (uint32_t)shids.size()); shids.push_back(id); sh_act.emplace_back(); sh_sta.emplace_back(); }
    }
    for (const SharedRec &sr : shrec) {
        SharedIdPub id; id.nk = sr.nk; memcpy(id.keys, sr.keys, sizeof id.keys);
        auto it = shmap.find(id); uint32_t s2;
        if (it == shmap.end()) { s2 = (uint32_t)shids.size(); shmap.emplace(id, s2); shids.push_back(id); sh_act.emplace_back(); sh_sta.emplace_back(); } else s2 = it->second;
        ((c->h_flags[sr.row] & F_STATIC) ? sh_sta[s2] : sh_act[s2]).push_back(sr.row);
    }
    const uint32_t nsh = (uint32_t)shids.size(), old_nsh = c->nsh;
    for (uint32_t s2 = 0; s2 < nsh; s2++) { std::sort(sh_act[s2].begin(), sh_act[s2].end(), by_id); std::sort(sh_sta[s2].begin(), sh_sta[s2].end(), by_id); }
    // (everything below that looks at the whole shared table is per patch O(shared sections); what can be is restricted to the shared sections that
    // appeared or vanished with this batch: the survivors' linked sections exist and keep their slots)
    std::unordered_set<uint64_t> linked; linked.reserve((size_t)nsh * 4 + 16);
    for (const SharedIdPub &id : shids) for (uint32_t k = 0; k < id.nk; k++) linked.insert(id.keys[k]);
    std::vector<int32_t> old_of(nsh, -1);                                     // index of a surviving shared section in the previous table
    {
        std::map<SharedIdPub, uint32_t> prev; for (uint32_t o = 0; o < old_nsh; o++) prev.emplace(c->h_shids[o], o);
        for (uint32_t s2 = 0; s2 < nsh; s2++) { auto it = prev.find(shids[s2]); if (it != prev.end()) old_of[s2] = (int32_t)it->second; }
    }
    lap("A shared");
    // ---- B. unique sections that may change: changed ones, newly linked ones, formerly linked ones
    std::set<uint64_t> affected(carry.changed_cells.begin(), carry.changed_cells.end());
    affected.insert(carry.ghost_touched.begin(), carry.ghost_touched.end());
    auto ghosts_of = [&](uint64_t K) -> const std::vector<uint32_t> * { auto g = c->ghost_map.find(K); return g == c->ghost_map.end() ? nullptr : &g->second; };
    for (uint32_t s2 = 0; s2 < nsh; s2++) if (old_of[s2] < 0) for (uint32_t k = 0; k < shids[s2].nk; k++) if (slot_of(shids[s2].keys[k]) < 0) affected.insert(shids[s2].keys[k]);   // sections a NEW shared section links
    {
        std::vector<uint8_t> survives(old_nsh, 0); for (uint32_t s2 = 0; s2 < nsh; s2++) if (old_of[s2] >= 0) survives[old_of[s2]] = 1;
        for (uint32_t o = 0; o < old_nsh; o++) if (!survives[o]) for (uint32_t k = 0; k < c->h_shids[o].nk; k++) if (!linked.count(c->h_shids[o].keys[k])) affected.insert(c->h_shids[o].keys[k]);   // sections only a VANISHED one linked
    }
    std::vector<Pair64> p_key, p_rowkey; std::vector<Pair32> p_begin, p_nl, p_ns, p_ng, p_rows, p_rowcell, p_stamp, p_cap;
    std::map<uint32_t, FlagOp> fops;                                          // one merged op per slot
    auto fop = [&](uint32_t slot) -> FlagOp & { auto it = fops.find(slot); if (it == fops.end()) { FlagOp f{}; f.idx = slot; f.and_mask = 0xFF; f.or_mask = 0; it = fops.emplace(slot, f).first; } return it->second; };
    std::vector<uint32_t> refold; std::set<uint32_t> created; std::vector<std::pair<uint32_t, uint32_t>> freed;   // (level, slot): reusable from the next patch on
    int32_t n_real_delta = 0;
    // ---- capacity check BEFORE anything is touched (a patch that gave up half-way would leave the mirrors ahead of the device):
    // free slots per level for the sections to create, row-pool room for every segment that may be relocated + the shared region
    {
        uint32_t need_slots[MAX_LEVELS] = {}; uint64_t need_pool = 0;
        for (uint64_t K : affected) {
            const int32_t slot = slot_of(K);
            uint32_t size = 0;
            if (slot >= 0) for (uint32_t i = 0, b = c->h_cell_begin[slot], e = c->h_cell_nl[slot] + c->h_cell_ns[slot]; i < e; i++) { uint32_t r = c->h_rows[b + i]; if (c->h_row_nk[r] == 1 && c->h_row_key[r] == K) size++; }
            auto ar = arrive.find(K);
            if (ar != arrive.end()) size += (uint32_t)ar->second.size();     // upper bound (an arriving row may already be counted)
            if (size == 0 && !linked.count(K)) continue;
            if (const auto *g = ghosts_of(K)) size += (uint32_t)g->size();
            if (slot < 0) need_slots[key_level(K) & (MAX_LEVELS - 1)]++;
            if (slot < 0 || size > c->h_cell_cap[slot]) need_pool += std::max(4u, size * 2u);
        }
        for (const SharedRec &sr : shrec) { (void)sr; need_pool += 1; }
        for (uint32_t l = 0; l < (uint32_t)MAX_LEVELS; l++) if (need_slots[l] > c->free_slots[l].size()) return 1;
        if ((uint64_t)c->pool_used + need_pool > c->pool_cap) return 1;
        if ((c->cfg.flags & RE_CFG_PROBE) && (uint64_t)c->htab_keys + affected.size() > (uint64_t)(c->htab_mask + 1u) * 7u / 10u) return 1;   // key -> slot table too full of tombstones: rebuild
    }
    for (uint32_t r : removed_rows) if (c->h_row_cell[r] != ROW_CELL_NONE) { c->h_row_cell[r] = ROW_CELL_NONE; p_rowcell.push_back(Pair32{ r, ROW_CELL_NONE }); }
    for (uint64_t K : affected) {
        int32_t slot = slot_of(K);
        std::vector<uint32_t> mem;
        if (slot >= 0) for (uint32_t i = 0, b = c->h_cell_begin[slot], e = c->h_cell_nl[slot] + c->h_cell_ns[slot]; i < e; i++) { uint32_t r = c->h_rows[b + i]; if (c->h_row_nk[r] == 1 && c->h_row_key[r] == K) mem.push_back(r); }
        auto ar = arrive.find(K);
        if (ar != arrive.end()) for (uint32_t r : ar->second) if (c->h_row_nk[r] == 1 && c->h_row_key[r] == K) mem.push_back(r);
        std::sort(mem.begin(), mem.end()); mem.erase(std::unique(mem.begin(), mem.end()), mem.end());
        std::sort(mem.begin(), mem.end(), [&](uint32_t a, uint32_t b) { bool sa = (c->h_flags[a] & F_STATIC) != 0, sb = (c->h_flags[b] & F_STATIC) != 0; return sa != sb ? sb : c->h_id[a] < c->h_id[b]; });
        const bool exists = !mem.empty() || linked.count(K);
        if (!exists) {
            if (slot < 0) continue;
            const uint32_t lv = key_level(K) & (MAX_LEVELS - 1);
            const uint64_t padk = pack_key(lv, 0xFFFFu, 0xFFFFu, 0xFFFFu);
            if (ghosts_of(K)) { uint8_t f0 = 0; HIPCHK(c, hipMemcpy(&f0, c->d_cell_flags.p + slot, 1, hipMemcpyDeviceToHost)); if (f0 & CF_STATIC_CACHED) c->dormant_cached.insert(K); }
            slot_memo[K] = -1;
            c->h_cell_key[slot] = padk; c->h_cell_nl[slot] = 0; c->h_cell_ns[slot] = 0; c->h_cell_ng[slot] = 0; p_ng.push_back(Pair32{ (uint32_t)slot, 0 }); freed.push_back({ lv, (uint32_t)slot }); c->extra_slots.erase(K);
            p_key.push_back(Pair64{ (uint32_t)slot, 0, padk }); p_nl.push_back(Pair32{ (uint32_t)slot, 0 }); p_ns.push_back(Pair32{ (uint32_t)slot, 0 });
            FlagOp &f = fop((uint32_t)slot); f.and_mask = 0; f.or_mask = (uint8_t)(CF_PAD | CF_STATIC_SECTION);
            n_real_delta--;
            continue;
        }
        if (slot < 0) {
            const uint32_t lv = key_level(K) & (MAX_LEVELS - 1);
            if (c->free_slots[lv].empty()) return c->fail(RE_E_STATE, "patch_sections: free-slot accounting");
            slot = (int32_t)c->free_slots[lv].back(); c->free_slots[lv].pop_back(); slot_memo[K] = slot;
            c->h_cell_key[slot] = K; c->extra_slots[K] = (uint32_t)slot; c->h_cell_cap[slot] = 0; c->h_cell_begin[slot] = 0; p_cap.push_back(Pair32{ (uint32_t)slot, 0u });
            p_key.push_back(Pair64{ (uint32_t)slot, 0, K }); p_stamp.push_back(Pair32{ (uint32_t)slot, 0 });
            FlagOp &f = fop((uint32_t)slot); f.and_mask = 0; f.or_mask = 0;
            if (c->dormant_cached.erase(K)) f.or_mask |= CF_STATIC_CACHED;          // its cache entry (now ghosts only) is reachable again
            created.insert((uint32_t)slot); n_real_delta++;
        }
        const std::vector<uint32_t> *gh = ghosts_of(K);                        // snapshot copies parked in this section, behind its static rows
        const uint32_t nmem = (uint32_t)mem.size(), ng = gh ? (uint32_t)gh->size() : 0u;
        if (gh) mem.insert(mem.end(), gh->begin(), gh->end());
        const uint32_t size = (uint32_t)mem.size();
        bool relocated = false;
        if (size > c->h_cell_cap[slot]) {
            relocated = true;
            const uint32_t cap = std::max(4u, size * 2u);
            if ((uint64_t)c->pool_used + cap > c->pool_cap) return c->fail(RE_E_STATE, "patch_sections: row-pool accounting");
            c->h_cell_begin[slot] = c->pool_used; c->h_cell_cap[slot] = cap; c->pool_used += cap;
            if (c->h_rows.size() < c->pool_used) c->h_rows.resize(c->pool_used, 0);
            p_begin.push_back(Pair32{ (uint32_t)slot, c->h_cell_begin[slot] }); p_cap.push_back(Pair32{ (uint32_t)slot, cap });
        }
        uint32_t nl = 0; for (uint32_t i = 0; i < nmem; i++) if (!(c->h_flags[mem[i]] & F_STATIC)) nl++;
        c->h_cell_nl[slot] = nl; c->h_cell_ns[slot] = nmem - nl;
        p_nl.push_back(Pair32{ (uint32_t)slot, nl }); p_ns.push_back(Pair32{ (uint32_t)slot, nmem - nl });
        if (c->h_cell_ng[slot] != ng || created.count((uint32_t)slot)) { c->h_cell_ng[slot] = ng; p_ng.push_back(Pair32{ (uint32_t)slot, ng }); }
        for (uint32_t i = 0; i < size; i++) {
            const uint32_t pos = c->h_cell_begin[slot] + i, r = mem[i];
            if (c->h_rows[pos] != r || relocated) { c->h_rows[pos] = r; p_rows.push_back(Pair32{ pos, r }); }
            if (i < nmem && c->h_row_cell[r] != (uint32_t)slot) { c->h_row_cell[r] = (uint32_t)slot; p_rowcell.push_back(Pair32{ r, (uint32_t)slot }); p_rowkey.push_back(Pair64{ r, 0, K }); }
        }
        if (carry.changed_cells.count(K) || created.count((uint32_t)slot)) refold.push_back((uint32_t)slot);
    }
    lap("B cells");
    // ---- C. shared sections: members in one fresh region at the end of the pool, table arrays re-uploaded whole
    uint32_t sh_total = 0; for (uint32_t s2 = 0; s2 < nsh; s2++) sh_total += (uint32_t)(sh_act[s2].size() + sh_sta[s2].size());
    if ((uint64_t)c->pool_used + sh_total > c->pool_cap) return c->fail(RE_E_STATE, "patch_sections: row-pool accounting (shared region)");
    std::vector<int32_t> sh_cells((size_t)nsh * 8 + 8, -1); std::vector<uint32_t> sh_begin(nsh + 1, 0), sh_nact(nsh + 1, 0), sh_nstat(nsh + 1, 0);
    std::unordered_map<uint32_t, std::vector<uint32_t>> cell_links;
    const uint32_t sh_region = c->pool_used;
    if (c->h_rows.size() < (size_t)c->pool_used + sh_total) c->h_rows.resize((size_t)c->pool_used + sh_total, 0);
    for (uint32_t s2 = 0; s2 < nsh; s2++) {
        for (uint32_t k = 0; k < shids[s2].nk; k++) {
            int32_t ci = -1;
            if (old_of[s2] >= 0 && (size_t)old_of[s2] * 8 + k < c->h_sh_cells.size()) { ci = c->h_sh_cells[(size_t)old_of[s2] * 8 + k]; if (ci < 0 || (uint32_t)ci >= c->ncells || c->h_cell_key[ci] != shids[s2].keys[k]) ci = -1; }   // a survivor's linked section keeps its slot (checked: it may have been emptied and re-created)
            if (ci < 0) ci = slot_of(shids[s2].keys[k]);
            if (ci < 0) return c->fail(RE_E_STATE, "patch_sections: linked section missing");
            sh_cells[(size_t)s2 * 8 + k] = ci; cell_links[(uint32_t)ci].push_back(s2);
        }
        sh_begin[s2] = c->pool_used; sh_nact[s2] = (uint32_t)sh_act[s2].size(); sh_nstat[s2] = (uint32_t)sh_sta[s2].size();
        for (int pass = 0; pass < 2; pass++) for (uint32_t r : (pass ? sh_sta[s2] : sh_act[s2])) {
            c->h_rows[c->pool_used++] = r;
            if (c->h_row_cell[r] != (ROW_CELL_SHARED | s2)) { c->h_row_cell[r] = ROW_CELL_SHARED | s2; p_rowcell.push_back(Pair32{ r, ROW_CELL_SHARED | s2 }); }
        }
    }
    std::vector<FlagOp> link_ops;                                             // the link counts the device-side re-bucket reads: formerly linked slots to 0, then the new counts
    {
        std::map<uint32_t, uint32_t> cnt;
        for (uint32_t sl : c->h_linked_slots) cnt[sl] = 0;
        c->h_linked_slots.clear();
        for (auto &kv : cell_links) { cnt[kv.first] = (uint32_t)std::min<size_t>(kv.second.size(), 255); for (size_t q = 0; q < kv.second.size(); q++) c->h_linked_slots.push_back(kv.first); }
        for (auto &kv : cnt) { FlagOp f{}; f.idx = kv.first; f.and_mask = 0; f.or_mask = (uint8_t)kv.second; link_ops.push_back(f); }
    }
    // rows that left the tree altogether (DeleteRequest)
    // (their row_cell was set by the caller through `arrive` being empty: handled below by the caller-provided list)
    // ---- D. update_static_world_sections for the changed sections (first loop), then for the changed shared sections (second loop)
    auto loop1 = [&](uint32_t slot) {
        if (c->h_cell_nl[slot] != 0) return false;
        auto it = cell_links.find(slot);
        if (it == cell_links.end()) return true;
        for (uint32_t s2 : it->second) if (sh_nact[s2] == 0) return true;
        return false;
    };
    auto set_static = [&](uint32_t slot, bool v) { FlagOp &f = fop(slot); f.and_mask &= (uint8_t)~CF_STATIC_SECTION; f.or_mask = (uint8_t)((f.or_mask & ~CF_STATIC_SECTION) | (v ? CF_STATIC_SECTION : 0)); };
    for (uint64_t K : affected) {
        int32_t slot = slot_of(K);
        if (slot < 0) continue;
        if (carry.changed_cells.count(K) || created.count((uint32_t)slot)) set_static((uint32_t)slot, loop1((uint32_t)slot));
    }
    for (uint64_t K : carry.changed_static) { int32_t slot = slot_of(K); if (slot >= 0) fop((uint32_t)slot).or_mask |= CF_STATIC_DIRTY; }
    {
        std::vector<uint32_t> order(nsh); for (uint32_t s2 = 0; s2 < nsh; s2++) order[s2] = s2;
        std::sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return shids[a] < shids[b]; });   // canonical id order
        // the running value of the flag: what the first loop decided, else unknown -> the ops below are pure set / clear, no read needed
        for (uint32_t s2 : order) {
            if (!carry.changed_shared_set.count(shids[s2])) continue;
            for (uint32_t k = 0; k < shids[s2].nk; k++) {
                const uint32_t ci = (uint32_t)sh_cells[(size_t)s2 * 8 + k];
                if (sh_nact[s2] == 0) { if (c->h_cell_nl[ci] == 0) set_static(ci, true); }
                else set_static(ci, false);
            }
        }
    }
    // ---- E/F. previous shared-section state (AABB of unchanged sections, cache owner) keyed by id; tiny arrays
    std::vector<Aabb> old_aabb(old_nsh); std::vector<int32_t> old_owner(old_nsh); std::vector<uint8_t> old_cached(old_nsh);
    if (old_nsh) {
        HIPCHK(c, hipMemcpyAsync(old_aabb.data(), c->d_sh_aabb.p, (size_t)old_nsh * sizeof(Aabb), hipMemcpyDeviceToHost, st));
        HIPCHK(c, hipMemcpyAsync(old_owner.data(), c->d_sh_owner.p, (size_t)old_nsh * 4, hipMemcpyDeviceToHost, st));
        HIPCHK(c, hipMemcpyAsync(old_cached.data(), c->d_sh_cached.p, old_nsh, hipMemcpyDeviceToHost, st));
        HIPCHK(c, sync_stream(st));
    }
    std::map<SharedIdPub, uint32_t> old_index; for (uint32_t s2 = 0; s2 < old_nsh; s2++) old_index.emplace(c->h_shids[s2], s2);
    std::vector<int32_t> owner(nsh, -1); std::vector<uint8_t> cached(nsh, 0), dirty(nsh, 0);
    for (uint32_t s2 = 0; s2 < nsh; s2++) {
        auto it = old_index.find(shids[s2]);
        if (it == old_index.end()) continue;
        cached[s2] = old_cached[it->second];
        const int32_t oo = old_owner[it->second];                               // slots are stable; the owner may have been emptied
        if (oo >= 0 && !is_pad(c->h_cell_key[oo])) owner[s2] = oo;
    }
    lap("C-F");
    // ---- G. upload + kernels
    {
        std::vector<FlagOp> vf; vf.reserve(fops.size()); for (auto &kv : fops) vf.push_back(kv.second);
        std::vector<Pair32> p_rowsgc; p_rowsgc.reserve(p_rows.size());         // the group class travels with every pool entry written
        for (const Pair32 &pr : p_rows) p_rowsgc.push_back(Pair32{ pr.idx, effective_gclass(c, pr.val) });
        std::vector<Pair32> *v32[9] = { &p_begin, &p_nl, &p_ns, &p_stamp, &p_rows, &p_rowcell, &p_cap, &p_rowsgc, &p_ng };      // (p_cap: the capacities the device-side re-bucket reads)
        uint32_t *dst32[9] = { c->d_cell_begin.p, c->d_cell_nlocal.p, c->d_cell_nstatic.p, c->d_cell_stamp.p, c->d_rows.p, c->d_row_cell.p, c->d_cell_cap.p, c->d_rows_gc.p, c->d_cell_nghost.p };
        std::vector<Pair32> p_key32; p_key32.reserve(p_key.size()); for (const Pair64 &pk : p_key) p_key32.push_back(Pair32{ pk.idx, to_key32(pk.val) });   // the compact stream keys follow
        size_t bytes = p_rowkey.size() * sizeof(Pair64) + 32 + p_key.size() * (sizeof(Pair64) + sizeof(Pair32)) + (vf.size() + link_ops.size()) * sizeof(FlagOp) + refold.size() * 4 + p_rows.size() * sizeof(Pair32) + 256;
        for (auto *v : v32) bytes += v->size() * sizeof(Pair32) + 16;
        if (c->d_stage.n < bytes) HIPCHK(c, c->d_stage.alloc(bytes * 2, nullptr));
        std::vector<uint8_t> host(bytes); size_t off = 0;
        auto put = [&](const void *src, size_t nb) { size_t o = off; if (nb) memcpy(host.data() + off, src, nb); off = (off + nb + 15) & ~(size_t)15; return o; };
        const size_t o_key = put(p_key.data(), p_key.size() * sizeof(Pair64)), o_fl = put(vf.data(), vf.size() * sizeof(FlagOp)), o_rf = put(refold.data(), refold.size() * 4);
        size_t o32[9]; for (int k = 0; k < 9; k++) o32[k] = put(v32[k]->data(), v32[k]->size() * sizeof(Pair32));
        const size_t o_k32 = put(p_key32.data(), p_key32.size() * sizeof(Pair32));
        const size_t o_lk = put(link_ops.data(), link_ops.size() * sizeof(FlagOp));
        const size_t o_rk = put(p_rowkey.data(), p_rowkey.size() * sizeof(Pair64));
        HIPCHK(c, hipMemcpyAsync(c->d_stage.p, host.data(), off, hipMemcpyHostToDevice, st));
        if ((c->cfg.flags & RE_CFG_PROBE) && !p_key.empty()) {                // the key -> slot table follows: retire the old keys of those slots, then enter the new ones
            for (uint32_t pass = 0; pass < 2; pass++)
                hipLaunchKernelGGL(k_hash_patch, dim3(((uint32_t)p_key.size() + 255) / 256), dim3(256), 0, st, (uint32_t)p_key.size(), reinterpret_cast<const Pair64 *>(c->d_stage.p + o_key), c->d_cell_key.p, c->d_htab.p, c->htab_mask, pass);
            c->htab_keys += (uint32_t)p_key.size();                         // upper bound on the keys (live + tombstones) the table holds
        }
        if (!p_key32.empty()) hipLaunchKernelGGL(k_scatter32, dim3(((uint32_t)p_key32.size() + 255) / 256), dim3(256), 0, st, (uint32_t)p_key32.size(), reinterpret_cast<const Pair32 *>(c->d_stage.p + o_k32), c->d_cell_key32.p);
        if (!p_key.empty()) hipLaunchKernelGGL(k_scatter64, dim3(((uint32_t)p_key.size() + 255) / 256), dim3(256), 0, st, (uint32_t)p_key.size(), reinterpret_cast<const Pair64 *>(c->d_stage.p + o_key), c->d_cell_key.p);
        if (!p_rowkey.empty()) hipLaunchKernelGGL(k_scatter64, dim3(((uint32_t)p_rowkey.size() + 255) / 256), dim3(256), 0, st, (uint32_t)p_rowkey.size(), reinterpret_cast<const Pair64 *>(c->d_stage.p + o_rk), c->d_row_key.p);
        for (int k = 0; k < 9; k++) if (!v32[k]->empty()) hipLaunchKernelGGL(k_scatter32, dim3(((uint32_t)v32[k]->size() + 255) / 256), dim3(256), 0, st, (uint32_t)v32[k]->size(), reinterpret_cast<const Pair32 *>(c->d_stage.p + o32[k]), dst32[k]);
        std::vector<uint32_t> sh_gc(sh_total);
        for (uint32_t i = 0; i < sh_total; i++) sh_gc[i] = effective_gclass(c, c->h_rows[sh_region + i]);
        if (sh_total) { HIPCHK(c, hipMemcpyAsync(c->d_rows.p + sh_region, c->h_rows.data() + sh_region, (size_t)sh_total * 4, hipMemcpyHostToDevice, st));
                        HIPCHK(c, hipMemcpyAsync(c->d_rows_gc.p + sh_region, sh_gc.data(), (size_t)sh_total * 4, hipMemcpyHostToDevice, st)); }
        if (!vf.empty()) hipLaunchKernelGGL(k_flag_ops, dim3(((uint32_t)vf.size() + 255) / 256), dim3(256), 0, st, (uint32_t)vf.size(), reinterpret_cast<const FlagOp *>(c->d_stage.p + o_fl), c->d_cell_flags.p);
        if (!link_ops.empty()) hipLaunchKernelGGL(k_flag_ops, dim3(((uint32_t)link_ops.size() + 255) / 256), dim3(256), 0, st, (uint32_t)link_ops.size(), reinterpret_cast<const FlagOp *>(c->d_stage.p + o_lk), c->d_cell_links.p);
        // end_of_changes: tight AABBs of the changed sections (stream order: after the table patches above)
        if (!refold.empty()) hipLaunchKernelGGL(k_fold_tight_list, dim3(((uint32_t)refold.size() + 255) / 256), dim3(256), 0, st, (uint32_t)refold.size(), reinterpret_cast<const uint32_t *>(c->d_stage.p + o_rf), c->d_cell_key.p,
                                                c->d_cell_begin.p, c->d_cell_nlocal.p, c->d_cell_nstatic.p, c->d_rows.p, c->d_aabb.p, c->d_cell_tight.p, c->cfg.atomic_length, carry.too_many ? 1 : 0);
        HIPCHK(c, hipGetLastError());
        HIPCHK(c, sync_stream(st));                                 // `host` goes out of scope
    }
    lap("G upload");
    { int rcs = ensure_shared_capacity(c, nsh); if (rcs != RE_OK) return rcs; }      // (the table is rebuilt compactly below: no holes, the device-only parts follow at their next use)
    c->rb_sh_dirty = true; c->sh_free.clear(); c->stale_shared.clear();
    if (nsh) {
        HIPCHK(c, hipMemcpyAsync(c->d_sh_cells.p, sh_cells.data(), (size_t)nsh * 8 * 4, hipMemcpyHostToDevice, st));
        HIPCHK(c, hipMemcpyAsync(c->d_sh_begin.p, sh_begin.data(), (size_t)nsh * 4, hipMemcpyHostToDevice, st));
        HIPCHK(c, hipMemcpyAsync(c->d_sh_nact.p, sh_nact.data(), (size_t)nsh * 4, hipMemcpyHostToDevice, st));
        HIPCHK(c, hipMemcpyAsync(c->d_sh_nstat.p, sh_nstat.data(), (size_t)nsh * 4, hipMemcpyHostToDevice, st));
        HIPCHK(c, hipMemcpyAsync(c->d_sh_owner.p, owner.data(), (size_t)nsh * 4, hipMemcpyHostToDevice, st));
        HIPCHK(c, hipMemcpyAsync(c->d_sh_cached.p, cached.data(), nsh, hipMemcpyHostToDevice, st));
        HIPCHK(c, hipMemcpyAsync(c->d_sh_dirty.p, dirty.data(), nsh, hipMemcpyHostToDevice, st));
    }
    // AABBs of the changed shared sections
    if (nsh) {
        hipLaunchKernelGGL(k_fold_shared, dim3((nsh + 255) / 256), dim3(256), 0, st, nsh, c->d_sh_begin.p, c->d_sh_nact.p, c->d_sh_nstat.p, c->d_rows.p, c->d_aabb.p, c->d_sh_aabb.p);
        std::vector<Aabb> sa(nsh);
        HIPCHK(c, hipMemcpyAsync(sa.data(), c->d_sh_aabb.p, (size_t)nsh * sizeof(Aabb), hipMemcpyDeviceToHost, st));
        HIPCHK(c, sync_stream(st));
        for (uint32_t s2 = 0; s2 < nsh; s2++) {
            if (carry.changed_shared_set.count(shids[s2])) continue;
            auto it = old_index.find(shids[s2]);
            if (it != old_index.end()) sa[s2] = old_aabb[it->second];
        }
        HIPCHK(c, hipMemcpy(c->d_sh_aabb.p, sa.data(), (size_t)nsh * sizeof(Aabb), hipMemcpyHostToDevice));
    }
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, sync_stream(st));
    for (auto &f : freed) c->free_slots[f.first].push_back(f.second);
    c->nsh = nsh; c->h_shids = shids; sh_nact.resize(nsh); sh_nstat.resize(nsh); c->h_sh_nact = sh_nact; c->h_sh_nstat = sh_nstat; sh_begin.resize(nsh); c->h_sh_begin = sh_begin;
    c->h_sh_cells.assign(sh_cells.begin(), sh_cells.begin() + (size_t)nsh * 8);
    c->n_real_sections = (uint32_t)((int32_t)c->n_real_sections + n_real_delta);
    c->nrows_csr = c->pool_used;
    if (!carry.changed_static.empty()) c->dirty_pending = true;
    c->n_patches++; c->rb_ovl_dirty = true;                                   // (extra_slots changed: the device overlay follows before its next use)
    return 0;
}

// ------------------------------------------------------------------------------------------------
// The same bookkeeping on the device for the common batch -- movers between unique world sections of a world without shared
// sections, ghosts or hidden rows: two ops per mover sorted by (section key, reference order), one thread per affected section (re_kernels.hip:
// k_rb_*).  The host reads one status block between the two phases (is the batch eligible, is there slack for the new sections and the
// relocated segments), hands over the free slots of the levels that need one, and afterwards only notes WHICH sections changed: its mirrors
// of the section table (h_cell_*, h_rows, h_row_*, extra_slots) are brought up to date from the device when a host path next needs them
// (sync_mirrors).  Returns RE_OK, 1 when the batch is left to the host path (nothing has been touched then), or an error.
// ------------------------------------------------------------------------------------------------
static RbCells rb_cells(re_ctx *c) {
    RbCells C; C.cell_key = c->d_cell_key.p; C.cell_key32 = c->d_cell_key32.p; C.cell_begin = c->d_cell_begin.p; C.cell_cap = c->d_cell_cap.p; C.cell_nl = c->d_cell_nlocal.p;
    C.cell_ns = c->d_cell_nstatic.p; C.cell_ng = c->d_cell_nghost.p; C.cell_stamp = c->d_cell_stamp.p; C.cell_flags = c->d_cell_flags.p;
    C.rows = c->d_rows.p; C.rows_gc = c->d_rows_gc.p; C.row_cell = c->d_row_cell.p; C.row_key = c->d_row_key.p; C.pool_cap = c->pool_cap; C.cell_links = c->d_cell_links.p;
    return C;
}
static RbTables rb_tables(re_ctx *c) { RbTables T; T.base_keys = c->d_base_keys.p; T.nbase = (uint32_t)c->base_keys.size(); T.ovl_keys = c->d_ovl_keys.p; T.ovl_slots = c->d_ovl_slots.p; T.ovl_mask = c->ovl_cap - 1u; return T; }

// host mirrors of the sections the device patched, fetched when a host path needs them
static ShTable sh_table(re_ctx *c);
static int sync_shared_mirrors(re_ctx *c);
static int sync_mirrors(re_ctx *c) {
    { int rc = sync_shared_mirrors(c); if (rc != RE_OK) return rc; }          // (first: a row that moved from a shared to a unique section ends as the unique part below leaves it)
    if (c->stale_slots.empty()) return RE_OK;
    hipStream_t st = c->stream;
    std::sort(c->stale_slots.begin(), c->stale_slots.end()); c->stale_slots.erase(std::unique(c->stale_slots.begin(), c->stale_slots.end()), c->stale_slots.end());
    const uint32_t n = (uint32_t)c->stale_slots.size();
    DevBuf<uint32_t> d_slots, d_hdr, d_offs, d_rows; DevBuf<uint64_t> d_keys;
    auto done = [&](int rc) { d_slots.release(nullptr); d_hdr.release(nullptr); d_offs.release(nullptr); d_rows.release(nullptr); d_keys.release(nullptr); return rc; };
    if (d_slots.alloc(n, nullptr) != hipSuccess || d_hdr.alloc((size_t)n * 4, nullptr) != hipSuccess || d_offs.alloc((size_t)n + 1, nullptr) != hipSuccess || d_keys.alloc(n, nullptr) != hipSuccess) return done(c->fail(RE_E_HIP, "sync_mirrors: out of device memory"));
    std::vector<uint64_t> keys(n); std::vector<uint32_t> hdr((size_t)n * 4), offs((size_t)n + 1, 0);
    if (hipMemcpyAsync(d_slots.p, c->stale_slots.data(), (size_t)n * 4, hipMemcpyHostToDevice, st) != hipSuccess) return done(c->fail(RE_E_HIP, "sync_mirrors: copy"));
    hipLaunchKernelGGL(k_rb_gather_cells, dim3((n + 255) / 256), dim3(256), 0, st, n, d_slots.p, rb_cells(c), d_keys.p, d_hdr.p);
    (void)hipMemcpyAsync(keys.data(), d_keys.p, (size_t)n * 8, hipMemcpyDeviceToHost, st); (void)hipMemcpyAsync(hdr.data(), d_hdr.p, (size_t)n * 16, hipMemcpyDeviceToHost, st);
    if (sync_stream(st) != hipSuccess) return done(c->fail(RE_E_HIP, "sync_mirrors: section headers"));
    for (uint32_t i = 0; i < n; i++) offs[i + 1] = offs[i] + hdr[(size_t)i * 4 + 2] + hdr[(size_t)i * 4 + 3];
    std::vector<uint32_t> rows(std::max<uint32_t>(offs[n], 1u));
    if (offs[n]) {
        if (d_rows.alloc(offs[n], nullptr) != hipSuccess) return done(c->fail(RE_E_HIP, "sync_mirrors: out of device memory"));
        (void)hipMemcpyAsync(d_offs.p, offs.data(), ((size_t)n + 1) * 4, hipMemcpyHostToDevice, st);
        hipLaunchKernelGGL(k_rb_gather_rows, dim3((n + 255) / 256), dim3(256), 0, st, n, d_slots.p, d_offs.p, rb_cells(c), d_rows.p);
        (void)hipMemcpyAsync(rows.data(), d_rows.p, (size_t)offs[n] * 4, hipMemcpyDeviceToHost, st);
        if (sync_stream(st) != hipSuccess) return done(c->fail(RE_E_HIP, "sync_mirrors: section rows"));
    }
    auto is_pad = [](uint64_t k) { return (k & 0xFFFFFFFFFFFFull) == 0xFFFFFFFFFFFFull; };
    for (uint32_t i = 0; i < n; i++) {
        const uint32_t slot = c->stale_slots[i], begin = hdr[(size_t)i * 4], cap = hdr[(size_t)i * 4 + 1], nl = hdr[(size_t)i * 4 + 2], ns = hdr[(size_t)i * 4 + 3];
        const uint64_t ko = c->h_cell_key[slot], kn = keys[i];
        if (ko != kn) {                                                        // find_slot: base = last full build, overlay = created since
            auto e = c->extra_slots.find(ko); if (e != c->extra_slots.end() && e->second == slot) c->extra_slots.erase(e);
            if (!is_pad(kn) && !(slot < c->base_keys.size() && c->base_keys[slot] == kn)) c->extra_slots[kn] = slot;
        }
        c->h_cell_key[slot] = kn; c->h_cell_begin[slot] = begin; c->h_cell_cap[slot] = cap; c->h_cell_nl[slot] = nl; c->h_cell_ns[slot] = ns; c->h_cell_ng[slot] = 0;
        if (c->h_rows.size() < (size_t)begin + std::max(cap, nl + ns)) c->h_rows.resize((size_t)begin + std::max(cap, nl + ns), 0);
        // the rest of the segment holds whatever the device's compaction left there: never "already in place" for the host path, which uploads only the
        // pool entries that differ from this mirror (patch_sections)
        for (uint32_t k = nl + ns; k < cap; k++) c->h_rows[begin + k] = 0xFFFFFFFFu;
        for (uint32_t k = 0; k < nl + ns; k++) {
            const uint32_t r = rows[offs[i] + k];
            c->h_rows[begin + k] = r;
            if (r < c->ghost_base) { c->h_row_cell[r] = slot; c->h_row_key[r] = kn; if (c->h_row_nk[r] > 1) c->h_row_shared_keys.erase(r); c->h_row_nk[r] = 1; }
        }
    }
    c->stale_slots.clear();
    return done(RE_OK);
}
// the shared world sections the device created, changed or retired (rebucket_on_device2): ids, counts, linked slots, members
static int sync_shared_mirrors(re_ctx *c) {
    const uint32_t nsh = c->nsh;
    if (c->h_shids.size() < nsh) { c->h_shids.resize(nsh, SharedIdPub{}); c->h_sh_nact.resize(nsh, 0); c->h_sh_nstat.resize(nsh, 0); c->h_sh_begin.resize(nsh, 0); c->h_sh_cells.resize((size_t)nsh * 8, -1); }
    if (c->stale_shared.empty()) return RE_OK;
    hipStream_t st = c->stream;
    std::sort(c->stale_shared.begin(), c->stale_shared.end()); c->stale_shared.erase(std::unique(c->stale_shared.begin(), c->stale_shared.end()), c->stale_shared.end());
    const uint32_t n = (uint32_t)c->stale_shared.size();
    DevBuf<uint32_t> d_idx, d_hdr, d_offs, d_rows; DevBuf<uint64_t> d_keys; DevBuf<int32_t> d_cells;
    auto done = [&](int rc) { d_idx.release(nullptr); d_hdr.release(nullptr); d_offs.release(nullptr); d_rows.release(nullptr); d_keys.release(nullptr); d_cells.release(nullptr); return rc; };
    if (d_idx.alloc(n, nullptr) != hipSuccess || d_hdr.alloc((size_t)n * 5, nullptr) != hipSuccess || d_offs.alloc((size_t)n + 1, nullptr) != hipSuccess || d_keys.alloc((size_t)n * 8, nullptr) != hipSuccess
        || d_cells.alloc((size_t)n * 8, nullptr) != hipSuccess) return done(c->fail(RE_E_HIP, "sync_mirrors: out of device memory"));
    std::vector<uint64_t> keys((size_t)n * 8); std::vector<uint32_t> hdr((size_t)n * 5), offs((size_t)n + 1, 0); std::vector<int32_t> cells((size_t)n * 8);
    if (hipMemcpyAsync(d_idx.p, c->stale_shared.data(), (size_t)n * 4, hipMemcpyHostToDevice, st) != hipSuccess) return done(c->fail(RE_E_HIP, "sync_mirrors: copy"));
    hipLaunchKernelGGL(k_rb2_gather_shared, dim3((n + 255) / 256), dim3(256), 0, st, n, (const uint32_t *)d_idx.p, sh_table(c), d_keys.p, d_hdr.p, d_cells.p);
    (void)hipMemcpyAsync(keys.data(), d_keys.p, (size_t)n * 64, hipMemcpyDeviceToHost, st); (void)hipMemcpyAsync(hdr.data(), d_hdr.p, (size_t)n * 20, hipMemcpyDeviceToHost, st);
    (void)hipMemcpyAsync(cells.data(), d_cells.p, (size_t)n * 32, hipMemcpyDeviceToHost, st);
    if (sync_stream(st) != hipSuccess) return done(c->fail(RE_E_HIP, "sync_mirrors: shared section headers"));
    for (uint32_t i = 0; i < n; i++) offs[i + 1] = offs[i] + hdr[(size_t)i * 5 + 2] + hdr[(size_t)i * 5 + 3];
    std::vector<uint32_t> rows(std::max<uint32_t>(offs[n], 1u));
    if (offs[n]) {
        if (d_rows.alloc(offs[n], nullptr) != hipSuccess) return done(c->fail(RE_E_HIP, "sync_mirrors: out of device memory"));
        (void)hipMemcpyAsync(d_offs.p, offs.data(), ((size_t)n + 1) * 4, hipMemcpyHostToDevice, st);
        hipLaunchKernelGGL(k_rb2_gather_shared_rows, dim3((n + 255) / 256), dim3(256), 0, st, n, (const uint32_t *)d_idx.p, (const uint32_t *)d_offs.p, sh_table(c), (const uint32_t *)c->d_rows.p, d_rows.p);
        (void)hipMemcpyAsync(rows.data(), d_rows.p, (size_t)offs[n] * 4, hipMemcpyDeviceToHost, st);
        if (sync_stream(st) != hipSuccess) return done(c->fail(RE_E_HIP, "sync_mirrors: shared section rows"));
    }
    for (uint32_t i = 0; i < n; i++) {
        const uint32_t s2 = c->stale_shared[i];
        if (s2 >= nsh) continue;
        const uint32_t begin = hdr[(size_t)i * 5], na = hdr[(size_t)i * 5 + 2], nst = hdr[(size_t)i * 5 + 3], nk = hdr[(size_t)i * 5 + 4];
        SharedIdPub id{}; id.nk = nk; for (uint32_t k = 0; k < nk && k < 8; k++) id.keys[k] = keys[(size_t)i * 8 + k];
        c->h_shids[s2] = id; c->h_sh_nact[s2] = na; c->h_sh_nstat[s2] = nst; c->h_sh_begin[s2] = begin;
        for (uint32_t k = 0; k < 8; k++) {                                   // (the link counts a host patch resets: every slot linked before or now)
            const int32_t was = c->h_sh_cells[(size_t)s2 * 8 + k], now = cells[(size_t)i * 8 + k];
            if (was >= 0) c->h_linked_slots.push_back((uint32_t)was);
            if (now >= 0) c->h_linked_slots.push_back((uint32_t)now);
            c->h_sh_cells[(size_t)s2 * 8 + k] = now;
        }
        if (c->h_rows.size() < (size_t)begin + na + nst) c->h_rows.resize((size_t)begin + na + nst, 0);
        std::array<uint64_t, 8> a; memcpy(a.data(), id.keys, sizeof id.keys);
        for (uint32_t k = 0; k < na + nst; k++) {
            const uint32_t r = rows[offs[i] + k];
            c->h_rows[begin + k] = r;
            if (r < c->ghost_base) { c->h_row_cell[r] = ROW_CELL_SHARED | s2; c->h_row_key[r] = id.keys[0]; c->h_row_nk[r] = (uint8_t)nk; c->h_row_shared_keys[r] = a; }
        }
    }
    c->stale_shared.clear();
    return done(RE_OK);
}

static bool device_rebucket_applicable(const re_ctx *c) {
    static const bool off = getenv("RE_EXP_HOST_REBUCKET") != nullptr;         // A/B switch of tools/rebucket_cost.py
    // (worlds whose frozen render cache holds ghost instances or hidden rows are eligible since round 3: a batch falls back only when it changes a section that parks
    // ghosts -- k_rb2_unique_segments -- or creates / retires a section whose key the host's ghost books know -- rebucket_on_device2; hidden rows are static and never move here)
    return !off && c->ncells && !(c->cfg.flags & (RE_CFG_PROBE | RE_CFG_FULL_REBUILD));
}

// ------------------------------------------------------------------------------------------------
// rebucket_on_device, round 3: the whole batch of a tick's movers on the device, shared world sections included (re_rebucket.hip).  Returns RE_OK (the
// movers listed in host_list -- static rows -- are left for the host path as a second batch), 1 when the batch is left to the host path whole (nothing
// has been touched then), or an error.
// ------------------------------------------------------------------------------------------------
static ShTable sh_table(re_ctx *c) {
    ShTable S; S.cells = c->d_sh_cells.p; S.aabb = c->d_sh_aabb.p; S.begin = c->d_sh_begin.p; S.nact = c->d_sh_nact.p; S.nstat = c->d_sh_nstat.p; S.rowcap = c->d_sh_rowcap.p;
    S.owner = c->d_sh_owner.p; S.cached = c->d_sh_cached.p; S.dirty = c->d_sh_dirty.p; S.keys = c->d_sh_keys.p; S.nk = c->d_sh_nk.p;
    S.hkeys = c->d_sh_hkeys.p; S.hidx = c->d_sh_hidx.p; S.hmask = c->sh_hmask; S.cap = c->sh_cap;
    return S;
}
// the device-only parts of the shared table (ids, row capacities, id -> index hash, free indices) from the host mirrors, after a host path rebuilt the table
static int ensure_device_shared(re_ctx *c) {
    if (!c->rb_sh_dirty) return RE_OK;
    { int rc = ensure_shared_capacity(c, c->nsh); if (rc != RE_OK) return rc; }
    hipStream_t st = c->stream;
    const uint32_t n = c->nsh;
    std::vector<uint64_t> keys((size_t)std::max(n, 1u) * 8, 0ull); std::vector<uint8_t> nk(std::max(n, 1u), 0); std::vector<uint32_t> rowcap(std::max(n, 1u), 0);
    c->sh_free.clear();
    for (uint32_t s2 = 0; s2 < n; s2++) {
        const SharedIdPub &id = c->h_shids[s2];
        nk[s2] = (uint8_t)id.nk; for (uint32_t k = 0; k < id.nk && k < 8; k++) keys[(size_t)s2 * 8 + k] = id.keys[k];
        rowcap[s2] = c->h_sh_nact[s2] + c->h_sh_nstat[s2];                   // (a host path packs the members without slack)
        if (!id.nk) c->sh_free.push_back(s2);
    }
    if (n) {
        HIPCHK(c, hipMemcpyAsync(c->d_sh_keys.p, keys.data(), (size_t)n * 64, hipMemcpyHostToDevice, st)); HIPCHK(c, hipMemcpyAsync(c->d_sh_nk.p, nk.data(), n, hipMemcpyHostToDevice, st));
        HIPCHK(c, hipMemcpyAsync(c->d_sh_rowcap.p, rowcap.data(), (size_t)n * 4, hipMemcpyHostToDevice, st));
    }
    HIPCHK(c, hipMemsetAsync(c->d_sh_hkeys.p, 0xFF, ((size_t)c->sh_hmask + 1u) * 8, st));
    if (n) hipLaunchKernelGGL(k_rb2_hash_insert, dim3((n + 255) / 256), dim3(256), 0, st, n, sh_table(c));
    HIPCHK(c, hipGetLastError()); HIPCHK(c, sync_stream(st));                // (the staging vectors go out of scope)
    c->rb_sh_dirty = false; c->sh_hash_used = n - (uint32_t)c->sh_free.size();
    return RE_OK;
}
static int rebucket_on_device2(re_ctx *c, uint32_t M, std::vector<uint32_t> *host_list, uint32_t n_deleted = 0) {      // n_deleted: rows of a DeleteRequest behind the M movers of the list (RB2_MOVER_DELETED)
    host_list->clear();
    if (!M || !device_rebucket_applicable(c)) return 1;
    hipStream_t st = c->stream;
    static const bool timing = getenv("RE_EXP_TIME_REBUCKET") != nullptr;
    auto t_begin = std::chrono::steady_clock::now(); auto lap = [&](const char *what) { if (timing) { auto t = std::chrono::steady_clock::now(); fprintf(stderr, "  device rebucket %-10s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(t - t_begin).count()); t_begin = t; } };
    // ---- lookup tables: sorted keys of the last full build + overlay of the sections created since; the shared table's device-only parts
    if (c->rb_base_dirty) {
        HIPCHK(c, c->d_base_keys.alloc(c->base_keys.size(), &c->dev_bytes));
        HIPCHK(c, hipMemcpyAsync(c->d_base_keys.p, c->base_keys.data(), c->base_keys.size() * 8, hipMemcpyHostToDevice, st));
        c->rb_base_dirty = false;
    }
    if (!c->ovl_cap) { c->ovl_cap = 1u << 17; HIPCHK(c, c->d_ovl_keys.alloc(c->ovl_cap, nullptr)); HIPCHK(c, c->d_ovl_slots.alloc(c->ovl_cap, nullptr)); c->rb_ovl_dirty = true; }
    if (((uint64_t)c->sh_hash_used + std::min(2u * M, c->sh_cap)) * 2u > (uint64_t)c->sh_hmask + 1u) c->rb_sh_dirty = true;      // retired ids keep their hash entry: rebuild before the probes get long
    if (c->rb_ovl_dirty || c->rb_sh_dirty) { int rc = sync_mirrors(c); if (rc != RE_OK) return rc; }
    if (c->rb_ovl_dirty) {
        if (c->extra_slots.size() * 4u > c->ovl_cap) return 1;
        HIPCHK(c, hipMemsetAsync(c->d_ovl_keys.p, 0xFF, (size_t)c->ovl_cap * 8, st));
        std::vector<Pair64> pr; pr.reserve(c->extra_slots.size());
        for (auto &kv : c->extra_slots) pr.push_back(Pair64{ kv.second, 0u, kv.first });
        if (!pr.empty()) {
            if (c->d_stage.n < pr.size() * sizeof(Pair64)) HIPCHK(c, c->d_stage.alloc(pr.size() * sizeof(Pair64) * 2, nullptr));
            HIPCHK(c, hipMemcpyAsync(c->d_stage.p, pr.data(), pr.size() * sizeof(Pair64), hipMemcpyHostToDevice, st));
            hipLaunchKernelGGL(k_rb_ovl_insert, dim3(((uint32_t)pr.size() + 255) / 256), dim3(256), 0, st, (uint32_t)pr.size(), reinterpret_cast<const Pair64 *>(c->d_stage.p), rb_tables(c));
            HIPCHK(c, sync_stream(st));
        }
        c->ovl_count = (uint32_t)c->extra_slots.size(); c->rb_ovl_dirty = false;
    }
    { int rc = ensure_device_shared(c); if (rc != RE_OK) return rc; }
    if (((uint64_t)c->sh_hash_used + std::min(2u * M, c->sh_cap)) * 2u > (uint64_t)c->sh_hmask + 1u) return 1;      // (cannot happen: the hash has 4 entries per table entry and a batch creates at most as many sections as the table has free entries)
    // ---- scratch: 2 member ops per mover + up to 8 link ops per op of a shared placement
    re_ctx::Rb2Scratch &B = c->rb2;
    const uint32_t n1 = 2u * M, link_cap = 16u * M;
    if (B.cap < M) {
        const uint32_t mc = std::max(2u * M, 2048u), oc = 18u * mc;
        for (DevBuf<uint64_t> *b : { &B.key, &B.key2, &B.ord, &B.ord_s, &B.kgath, &B.ksorted1, &B.ksorted2 }) HIPCHK(c, b->alloc(oc, nullptr));
        for (DevBuf<uint32_t> *b : { &B.row, &B.idx, &B.perm_a, &B.perm1, &B.perm2, &B.refold, &B.tmp_u, &B.tmp_s }) HIPCHK(c, b->alloc(oc, nullptr));
        HIPCHK(c, B.mk.alloc((size_t)mc * 8, nullptr)); HIPCHK(c, B.mnk.alloc(mc, nullptr)); HIPCHK(c, B.host_list.alloc(mc, nullptr));
        HIPCHK(c, B.segs_u.alloc(oc, nullptr)); HIPCHK(c, B.segs_s.alloc(2u * mc, nullptr));
        HIPCHK(c, B.free_u.alloc(oc, nullptr)); HIPCHK(c, B.free_off.alloc(MAX_LEVELS, nullptr)); HIPCHK(c, B.free_s.alloc(2u * mc, nullptr));
        HIPCHK(c, B.pair_key.alloc(16u * mc, nullptr)); HIPCHK(c, B.pair_key_s.alloc(16u * mc, nullptr)); HIPCHK(c, B.pair_seg.alloc(16u * mc, nullptr)); HIPCHK(c, B.pair_seg_s.alloc(16u * mc, nullptr));
        if (!B.status.p) { HIPCHK(c, B.status.alloc(1, nullptr)); B.status_clean = false; }
        size_t t1 = 0, t2 = 0;
        HIPCHK(c, re::sort_pairs_u64_u32(nullptr, &t1, B.ord.p, B.ord_s.p, B.idx.p, B.perm_a.p, oc, 0, 35, st));
        HIPCHK(c, re::sort_pairs_u64_u32(nullptr, &t2, B.kgath.p, B.ksorted1.p, B.perm_a.p, B.perm1.p, oc, 0, 64, st));
        HIPCHK(c, B.tmp.alloc(std::max(t1, t2) + 256, nullptr));
        {   // the pinned staging block, laid out for mc movers
            auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
            const size_t o_seq = up(sizeof(Rb2Status)), o_off = o_seq + 256, o_fu = o_off + up(MAX_LEVELS * 4), o_fs = o_fu + up((size_t)oc * 4), o_keep = o_fs + up((size_t)2 * mc * 4),
                         o_su = o_keep + up((size_t)mc * 4), o_ss = o_su + up((size_t)oc * sizeof(Rb2Seg)), total = o_ss + up((size_t)2 * mc * sizeof(Rb2ShSeg));
            if (B.pin) { (void)hipHostFree(B.pin); B.pin = nullptr; }
            void *hp = nullptr, *dp = nullptr; HIPCHK(c, hipHostMalloc(&hp, total, hipHostMallocMapped)); HIPCHK(c, hipHostGetDevicePointer(&dp, hp, 0));
            B.pin = static_cast<uint8_t *>(hp); B.d_pin = static_cast<uint8_t *>(dp); B.pin_bytes = total;
            B.h_status = reinterpret_cast<Rb2Status *>(B.pin); B.d_h_status = reinterpret_cast<Rb2Status *>(dp);
            B.h_seq = reinterpret_cast<uint32_t *>(B.pin + o_seq); B.d_h_seq = reinterpret_cast<uint32_t *>(static_cast<uint8_t *>(dp) + o_seq); *B.h_seq = 0; B.seq = 0; B.h_free_off = reinterpret_cast<uint32_t *>(B.pin + o_off); B.h_free_u = reinterpret_cast<uint32_t *>(B.pin + o_fu);
            B.h_free_s = reinterpret_cast<uint32_t *>(B.pin + o_fs); B.h_keep = reinterpret_cast<uint32_t *>(B.pin + o_keep);
            B.h_segs_u = reinterpret_cast<Rb2Seg *>(B.pin + o_su); B.h_segs_s = reinterpret_cast<Rb2ShSeg *>(B.pin + o_ss);
        }
        B.cap = mc;
    }
    if (!c->d_cell_inact.p || c->d_cell_inact.n < c->ncells) { HIPCHK(c, c->d_cell_inact.alloc(std::max(c->ncells, 1u), nullptr)); HIPCHK(c, hipMemsetAsync(c->d_cell_inact.p, 0, std::max(c->ncells, 1u), st)); }      // (all zero between batches)
    Rb2Status &hs = *B.h_status; hs = Rb2Status{}; hs.pool_used = c->pool_used;
    if (!(B.status_clean && B.status_pool == c->pool_used)) HIPCHK(c, hipMemcpyAsync(B.status.p, &hs, sizeof hs, hipMemcpyHostToDevice, st));      // (else: the last batch's final read-back left the block ready)
    B.status_clean = false;
    auto mapped = [&](const void *h) { return reinterpret_cast<uint32_t *>(B.d_pin + (reinterpret_cast<const uint8_t *>(h) - B.pin)); };
    const RbTables T = rb_tables(c); const RbCells C = rb_cells(c); const ShTable S = sh_table(c);
    // the status block as the kernels so far left it: published into the mapped block by one small kernel and polled (a copy plus a stream synchronise costs ~15 us more, three times a batch)
    auto wait_status = [&](uint32_t seq) -> int {
        HIPCHK(c, hipGetLastError());
        const volatile uint32_t *flag = B.h_seq; bool done = false;
        const auto t0 = std::chrono::steady_clock::now();
        for (uint32_t spins = 0; !(done = (*flag == seq)); spins++)
            if ((spins & 1023u) == 1023u && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(5)) break;   // a long batch: let the driver wait
        if (!done) { HIPCHK(c, sync_stream(st)); if (*flag != seq) return c->fail(RE_E_STATE, "device re-bucket: the status block was not published"); }
        std::atomic_thread_fence(std::memory_order_acquire);
        return RE_OK;
    };
    auto read_status = [&](const void *src_a = nullptr, void *h_a = nullptr, uint32_t words_a = 0, const void *src_b = nullptr, void *h_b = nullptr, uint32_t words_b = 0, bool reset = false) -> int {
        const uint32_t seq = ++B.seq;
        hipLaunchKernelGGL(k_rb2_publish_status, dim3(1), dim3(256), 0, st, B.status.p, B.d_h_status, B.d_h_seq, seq, static_cast<const uint32_t *>(src_a), words_a ? mapped(h_a) : nullptr, words_a,
                           static_cast<const uint32_t *>(src_b), words_b ? mapped(h_b) : nullptr, words_b, reset ? 1u : 0u);
        return wait_status(seq);
    };
    auto sort_ops = [&](uint32_t n, const uint64_t *key_src, uint64_t *ksorted, uint32_t *perm, const uint32_t *n_extra = nullptr) -> int {      // by (placement key, reference order): two stable radix sorts
        if (n <= RB2_SORT_SMALL) { hipLaunchKernelGGL(k_rb2_sort_small, dim3(1), dim3(1024), 0, st, n, key_src, (const uint64_t *)B.ord.p, ksorted, perm, n_extra); return RE_OK; }      // (one launch of one workgroup)
        size_t tb = B.tmp.n;
        HIPCHK(c, re::sort_pairs_u64_u32(B.tmp.p, &tb, B.ord.p, B.ord_s.p, B.idx.p, B.perm_a.p, n, 0, 35, st));
        hipLaunchKernelGGL(k_rb_gather_keys, dim3((n + 255) / 256), dim3(256), 0, st, n, (const uint32_t *)B.perm_a.p, key_src, B.kgath.p);
        tb = B.tmp.n;
        HIPCHK(c, re::sort_pairs_u64_u32(B.tmp.p, &tb, B.kgath.p, ksorted, B.perm_a.p, perm, n, 0, 64, st));
        return RE_OK;
    };
    // ---- phases 1-3: the ops, the shared placements (link ops for the sections they link), the unique placements with the link ops merged in
    const bool general = getenv("RE_EXP_RB2_GENERAL") != nullptr;            // (tests: small batches through the kernels of large ones; read per batch)
    bool plan_small = M <= RB2_PLAN_SMALL && !general;
    static_assert(RB2_PLAN_SMALL * 2u <= RB2_SORT_SMALL, "2 member ops per mover");
    if (plan_small) {                                                         // one launch of one workgroup, the status block published behind it
        const uint32_t seq = ++B.seq;
        hipLaunchKernelGGL(k_rb2_plan_small, dim3(1), dim3(1024), 0, st, M, (const uint32_t *)c->d_movers.p, row_arrays(c), C, S, T, c->cfg.outline_length, c->cfg.atomic_length,
                           B.key.p, B.key2.p, B.ord.p, B.row.p, B.idx.p, B.mk.p, B.mnk.p, B.host_list.p, B.ksorted1.p, B.perm1.p, B.ksorted2.p, B.perm2.p, link_cap, (const uint8_t *)c->d_cell_links.p,
                           B.segs_s.p, B.segs_u.p, B.status.p, B.d_h_status, B.d_h_seq, seq, mapped(B.h_segs_u));
        { int rc = wait_status(seq); if (rc != RE_OK) return rc; }
        if (n1 + hs.n_link > RB2_SORT_SMALL) {                                // the link ops did not fit the one-workgroup sort (many shared sections created / emptied): the plan is void, again below
            plan_small = false;
            hs = Rb2Status{}; hs.pool_used = c->pool_used;
            HIPCHK(c, hipMemcpyAsync(B.status.p, &hs, sizeof hs, hipMemcpyHostToDevice, st));
        }
    }
    if (!plan_small) {
        hipLaunchKernelGGL(k_rb2_ops, dim3((M + 255) / 256), dim3(256), 0, st, M, c->d_movers.p, row_arrays(c), C, S, c->cfg.outline_length, c->cfg.atomic_length,
                           B.key.p, B.key2.p, B.ord.p, B.row.p, B.idx.p, B.mk.p, B.mnk.p, B.host_list.p, B.status.p);
        { int rc = sort_ops(n1, B.key.p, B.ksorted1.p, B.perm1.p); if (rc != RE_OK) return rc; }
        hipLaunchKernelGGL(k_rb2_shared_segments, dim3((n1 + 255) / 256), dim3(256), 0, st, n1, M, (const uint32_t *)B.perm1.p, (const uint64_t *)B.ksorted1.p, (const uint32_t *)B.row.p, B.ord.p,
                           (const uint64_t *)B.mk.p, (const uint8_t *)B.mnk.p, S, C, B.key2.p, B.row.p, B.idx.p, link_cap, B.segs_s.p, B.status.p);
        { int rc = read_status(); if (rc != RE_OK) return rc; }
        lap("shared");
        if (hs.fallback || hs.n_link > link_cap) return 1;
        const uint32_t n2 = n1 + hs.n_link;
        { int rc = sort_ops(n2, B.key2.p, B.ksorted2.p, B.perm2.p); if (rc != RE_OK) return rc; }
        hipLaunchKernelGGL(k_rb2_unique_segments, dim3((n2 + 255) / 256), dim3(256), 0, st, n2, (const uint32_t *)B.perm2.p, (const uint64_t *)B.ksorted2.p, (const uint32_t *)B.row.p, T, C, c->d_cell_links.p, B.segs_u.p, B.status.p);
        { int rc = read_status(); if (rc != RE_OK) return rc; }
    }
    lap("unique");
    if (hs.fallback || hs.n_link > link_cap) return 1;
    if (!c->ghost_map.empty() || !c->dormant_cached.empty()) {               // sections this batch would create or retire: none may be one the ghost books of the frozen cache know
        const uint32_t nq = hs.nseg_u;
        if (nq && !plan_small) { HIPCHK(c, hipMemcpyAsync(B.h_segs_u, B.segs_u.p, (size_t)nq * sizeof(Rb2Seg), hipMemcpyDeviceToHost, st)); HIPCHK(c, sync_stream(st)); }      // (k_rb2_plan_small publishes the list with the status)
        for (uint32_t i2 = 0; i2 < nq; i2++) {
            const Rb2Seg &G = B.h_segs_u[i2];
            if (G.exists0 != G.exists1 && (c->ghost_map.count(G.key) || c->dormant_cached.count(G.key))) return 1;
        }
    }
    // movers the host path keeps (static rows): as a second batch behind this one, only where the threshold of total_world_aabb_combining cannot depend on the split
    if (hs.n_host >= M || (hs.n_host && hs.total <= 500u) || (hs.n_host && n_deleted)) return 1;      // (a batch with deletions is not split: the host's second batch would count them again)
    std::vector<uint32_t> keep(hs.n_host);
    if (hs.n_host) { HIPCHK(c, hipMemcpyAsync(B.h_keep, B.host_list.p, (size_t)hs.n_host * 4, hipMemcpyDeviceToHost, st)); HIPCHK(c, sync_stream(st)); memcpy(keep.data(), B.h_keep, (size_t)hs.n_host * 4); }
    uint32_t need_total = 0;
    for (uint32_t l = 0; l < (uint32_t)MAX_LEVELS; l++) { if (hs.need_slots[l] > c->free_slots[l].size()) return 1; need_total += hs.need_slots[l]; }
    if ((uint64_t)c->pool_used + hs.need_pool > c->pool_cap) return 1;
    if (((uint64_t)c->ovl_count + need_total) * 2u > c->ovl_cap) return 1;
    if (hs.need_sh > c->sh_free.size() + (c->sh_cap - c->nsh)) return 1;      // (the host path rebuilds the table compactly and with more room)
    // ---- phase 4: free slots / free shared indices for what is created, then the patch itself
    std::vector<uint32_t> fs; uint32_t nfl = 0;
    for (uint32_t l = 0; l < (uint32_t)MAX_LEVELS; l++) { B.h_free_off[l] = nfl; for (uint32_t j = 0; j < hs.need_slots[l]; j++) B.h_free_u[nfl++] = c->free_slots[l][c->free_slots[l].size() - 1u - j]; }
    { uint32_t bump = c->nsh; for (uint32_t j = 0; j < hs.need_sh; j++) { fs.push_back(j < c->sh_free.size() ? c->sh_free[c->sh_free.size() - 1u - j] : bump++); B.h_free_s[j] = fs.back(); } }
    const bool inline_free = nfl + fs.size() <= RB2_INLINE_WORDS;            // a small batch: the apply kernels read the lists from the mapped block (three stream copies less)
    if (!inline_free) {
        if (nfl) HIPCHK(c, hipMemcpyAsync(B.free_u.p, B.h_free_u, (size_t)nfl * 4, hipMemcpyHostToDevice, st));
        if (!fs.empty()) HIPCHK(c, hipMemcpyAsync(B.free_s.p, B.h_free_s, fs.size() * 4, hipMemcpyHostToDevice, st));
        HIPCHK(c, hipMemcpyAsync(B.free_off.p, B.h_free_off, MAX_LEVELS * 4, hipMemcpyHostToDevice, st));
    }
    const uint32_t *a_free_u = inline_free ? mapped(B.h_free_u) : B.free_u.p, *a_free_s = inline_free ? mapped(B.h_free_s) : B.free_s.p, *a_free_off = inline_free ? mapped(B.h_free_off) : B.free_off.p;
    const uint32_t nu = hs.nseg_u, ns = hs.nseg_s;
    uint32_t nsh_after = c->nsh; for (uint32_t x : fs) nsh_after = std::max(nsh_after, x + 1u);
    const Rb2Status planned = hs;                                            // (the plan of the phases above; the block is read back once more below)
    Rb2Status &h2 = *B.h_status;
    const bool apply_small = !general && inline_free && 8u * ns <= RB2_STATIC_SMALL_PAIRS && nsh_after <= RB2_STATIC_SMALL_SHARED && nu + ns <= 2048u;      // (k_rb2_static_small's limits; segment lists the workgroup can publish)
    if (apply_small) {                                                        // the whole of phase 4 and its read-back: one launch of one workgroup (behind the per-segment kernels when there are more than a few)
        const bool wide = nu + ns > 32u;                                       // more segments than two rounds of the workgroup's 16 waves: a workgroup per segment first
        if (wide && nu) hipLaunchKernelGGL(k_rb2_apply_unique, dim3(nu), dim3(64), 0, st, (const uint32_t *)B.perm2.p, (const uint32_t *)B.row.p, T, C, row_arrays(c), c->d_cell_links.p, B.segs_u.p, B.status.p,
                                           a_free_u, a_free_off, B.tmp_u.p, B.refold.p);
        if (wide && ns) hipLaunchKernelGGL(k_rb2_apply_shared, dim3(ns), dim3(64), 0, st, (const uint32_t *)B.perm1.p, (const uint32_t *)B.row.p, T, C, row_arrays(c), S, B.segs_s.p, B.status.p, a_free_s, B.tmp_s.p);
        const uint32_t seq = ++B.seq;
        hipLaunchKernelGGL(k_rb2_apply_small, dim3(1), dim3(1024), 0, st, M, (const uint32_t *)c->d_movers.p, n_deleted ? 1u : 0u, (const uint32_t *)B.perm1.p, (const uint32_t *)B.perm2.p, (const uint32_t *)B.row.p,
                           T, C, row_arrays(c), S, c->d_cell_links.p, c->d_cell_inact.p, c->d_cell_tight.p, B.segs_u.p, B.segs_s.p, B.status.p, a_free_u, a_free_off, a_free_s, B.tmp_u.p, B.tmp_s.p, B.refold.p,
                           nsh_after, c->cfg.atomic_length, hs.total > 500u ? 1u : 0u, wide ? 1u : 0u, B.d_h_status, B.d_h_seq, seq, mapped(B.h_segs_u), mapped(B.h_segs_s));
        { int rc = wait_status(seq); if (rc != RE_OK) return rc; }
    } else {
    if (nu) hipLaunchKernelGGL(k_rb2_apply_unique, dim3(nu), dim3(64), 0, st, (const uint32_t *)B.perm2.p, (const uint32_t *)B.row.p, T, C, row_arrays(c), c->d_cell_links.p, B.segs_u.p, B.status.p,
                               a_free_u, a_free_off, B.tmp_u.p, B.refold.p);
    if (ns) hipLaunchKernelGGL(k_rb2_apply_shared, dim3(ns), dim3(64), 0, st, (const uint32_t *)B.perm1.p, (const uint32_t *)B.row.p, T, C, row_arrays(c), S, B.segs_s.p, B.status.p,
                               a_free_s, B.tmp_s.p);
    // update_static_world_sections: first loop (changed / new unique sections), second loop (changed shared sections in canonical order)
    const bool static_small = 8u * ns <= RB2_STATIC_SMALL_PAIRS && nsh_after <= RB2_STATIC_SMALL_SHARED && nu <= 16384u && !general;
    if (static_small) {
        if (nu || ns) hipLaunchKernelGGL(k_rb2_static_small, dim3(1), dim3(1024), 0, st, nsh_after, S, C, (const uint8_t *)c->d_cell_links.p, c->d_cell_inact.p, (const Rb2Seg *)B.segs_u.p, (const Rb2ShSeg *)B.segs_s.p, (const Rb2Status *)B.status.p);
    } else {
    if (nsh_after && nu) hipLaunchKernelGGL(k_rb2_mark_inactive, dim3((nsh_after + 255) / 256), dim3(256), 0, st, nsh_after, S, c->d_cell_inact.p, (uint8_t)1);
    if (nu) hipLaunchKernelGGL(k_rb2_static_first, dim3((nu + 255) / 256), dim3(256), 0, st, C, (const uint8_t *)c->d_cell_links.p, (const uint8_t *)c->d_cell_inact.p, (const Rb2Seg *)B.segs_u.p, (const Rb2Status *)B.status.p);
    if (nsh_after && nu) hipLaunchKernelGGL(k_rb2_mark_inactive, dim3((nsh_after + 255) / 256), dim3(256), 0, st, nsh_after, S, c->d_cell_inact.p, (uint8_t)0);
    if (ns) {
        hipLaunchKernelGGL(k_rb2_static_pairs, dim3((ns + 255) / 256), dim3(256), 0, st, (const Rb2ShSeg *)B.segs_s.p, (const Rb2Status *)B.status.p, S, B.pair_key.p, B.pair_seg.p, B.status.p);
        const uint32_t np_max = 8u * ns;                                      // (pairs beyond n_pairs carry stale keys: sort only what was written -- the count comes back with the status below, so sort the bound and let the kernel stop at n_pairs)
        { int rc = read_status(); if (rc != RE_OK) return rc; }
        const uint32_t np = std::min(hs.n_pairs, np_max);
        if (np && np <= RB2_SORT_SMALL) {      // (a stable sort by slot: the pair index is the tie-break, and pair i belongs to segment pair_seg[i])
            hipLaunchKernelGGL(k_rb2_sort_small, dim3(1), dim3(1024), 0, st, np, (const uint64_t *)B.pair_key.p, (const uint64_t *)nullptr, B.pair_key_s.p, B.perm_a.p, (const uint32_t *)nullptr);
            hipLaunchKernelGGL(k_rb2_gather_u32, dim3((np + 255) / 256), dim3(256), 0, st, np, (const uint32_t *)B.perm_a.p, (const uint32_t *)B.pair_seg.p, B.pair_seg_s.p);
            hipLaunchKernelGGL(k_rb2_static_second, dim3((np + 255) / 256), dim3(256), 0, st, np, (const uint64_t *)B.pair_key_s.p, (const uint32_t *)B.pair_seg_s.p, (const Rb2ShSeg *)B.segs_s.p, C);
        } else if (np) {
            size_t tb = B.tmp.n;
            HIPCHK(c, re::sort_pairs_u64_u32(B.tmp.p, &tb, B.pair_key.p, B.pair_key_s.p, B.pair_seg.p, B.pair_seg_s.p, np, 0, 32, st));
            hipLaunchKernelGGL(k_rb2_static_second, dim3((np + 255) / 256), dim3(256), 0, st, np, (const uint64_t *)B.pair_key_s.p, (const uint32_t *)B.pair_seg_s.p, (const Rb2ShSeg *)B.segs_s.p, C);
        }
    }
    }
    // end_of_changes: tight AABBs of the changed sections (bounding_box_tree_v2.rs:1055-1130)
    if (nu) hipLaunchKernelGGL(k_fold_tight_list, dim3((nu + 255) / 256), dim3(256), 0, st, nu, (const uint32_t *)B.refold.p, c->d_cell_key.p, c->d_cell_begin.p, c->d_cell_nlocal.p, c->d_cell_nstatic.p, c->d_rows.p,
                               c->d_aabb.p, c->d_cell_tight.p, c->cfg.atomic_length, hs.total > 500u ? 1 : 0);
    if (n_deleted) hipLaunchKernelGGL(k_rb2_clear_deleted, dim3((M + 255) / 256), dim3(256), 0, st, M, (const uint32_t *)c->d_movers.p, c->d_row_cell.p);
    HIPCHK(c, hipGetLastError());
    static_assert(sizeof(Rb2Seg) % 4u == 0 && sizeof(Rb2ShSeg) % 4u == 0, "segment lists are copied by words");
    const uint32_t wu = nu * (uint32_t)(sizeof(Rb2Seg) / 4u), ws = ns * (uint32_t)(sizeof(Rb2ShSeg) / 4u);
    if ((uint64_t)wu + ws <= RB2_INLINE_WORDS) {
        int rc = read_status(B.segs_u.p, B.h_segs_u, wu, B.segs_s.p, B.h_segs_s, ws, true); if (rc != RE_OK) return rc;
    } else {
        if (nu) HIPCHK(c, hipMemcpyAsync(B.h_segs_u, B.segs_u.p, (size_t)nu * sizeof(Rb2Seg), hipMemcpyDeviceToHost, st));
        if (ns) HIPCHK(c, hipMemcpyAsync(B.h_segs_s, B.segs_s.p, (size_t)ns * sizeof(Rb2ShSeg), hipMemcpyDeviceToHost, st));
        int rc = read_status(nullptr, nullptr, 0, nullptr, nullptr, 0, true); if (rc != RE_OK) return rc;      // (stream order: the segment copies above have landed when the sequence word arrives)
    }
    }
    B.status_clean = h2.err == 0; B.status_pool = h2.pool_used;
    const Rb2Seg *su = B.h_segs_u; const Rb2ShSeg *ss = B.h_segs_s;
    lap("apply");
    // ---- what the host keeps in step at once: free slots / indices, pool fill, counts; everything else waits for sync_mirrors
    if (h2.err) return c->fail(RE_E_STATE, "device re-bucket: accounting error %u in the apply phase", h2.err);
    for (uint32_t l = 0; l < (uint32_t)MAX_LEVELS; l++) {
        if (h2.popped[l] != planned.need_slots[l]) return c->fail(RE_E_STATE, "device re-bucket: free-slot accounting (level %u: %u taken, %u planned)", l, h2.popped[l], planned.need_slots[l]);
        c->free_slots[l].resize(c->free_slots[l].size() - planned.need_slots[l]);
    }
    if (h2.popped_sh != planned.need_sh) return c->fail(RE_E_STATE, "device re-bucket: shared-index accounting (%u taken, %u planned)", h2.popped_sh, planned.need_sh);
    { const uint32_t from_holes = std::min<uint32_t>(planned.need_sh, (uint32_t)c->sh_free.size()); c->sh_free.resize(c->sh_free.size() - from_holes); }
    int32_t delta = 0;
    for (uint32_t i2 = 0; i2 < nu; i2++) { const Rb2Seg &G = su[i2];
        if (G.slot < 0) continue;
        c->stale_slots.push_back((uint32_t)G.slot);
        if (G.freed) { c->free_slots[key_level(G.key) & (MAX_LEVELS - 1)].push_back((uint32_t)G.slot); delta--; }
        if (G.created) delta++;
    }
    for (uint32_t i2 = 0; i2 < ns; i2++) { const Rb2ShSeg &G = ss[i2];
        if (G.idx < 0) continue;
        c->stale_shared.push_back((uint32_t)G.idx);
        if (G.freed) c->sh_free.push_back((uint32_t)G.idx);
    }
    if (h2.pool_used > c->pool_cap) return c->fail(RE_E_STATE, "device re-bucket: row-pool accounting");
    c->nsh = nsh_after; c->sh_hash_used += h2.n_sh_created;
    c->pool_used = h2.pool_used; c->nrows_csr = c->pool_used; c->ovl_count += h2.n_created;
    c->n_real_sections = (uint32_t)((int32_t)c->n_real_sections + delta);
    c->n_patches++; c->n_device_rebuckets++;
    host_list->swap(keep);
    lap("bookkeeping");
    return RE_OK;
}

// ------------------------------------------------------------------------------------------------
// Incremental re-bucket after a tick: update_entity_in_tree -> BoundingBoxTree::add_entity (which removes the entity from
// its previous section) for every mover whose section changed, then end_of_changes (helper_things/entity_change_helpers.rs:
// 217-262, 325-351; world/bounding_box_tree_v2.rs:563-942, 1055-1213).  Order of the reference: translation-only movers, then
// kinematic movers, each set in ascending EntityId (stand-in for hash order).  The sequential bookkeeping that decides
// total_world_aabb_combining (> 500 => crowded changed sections fall back to their grid AABB) is replayed exactly on the
// few affected sections; the result is then patched into the resident table (patch_sections), or -- when its slack is used
// up -- the key-sorted arrays are rebuilt, carrying over everything the reference leaves untouched.
// ------------------------------------------------------------------------------------------------
// pre: tree operations an apply_change batch performs inline, before the kinematic re-adds (MakeObjectStatic / WakeUpRequest:
// remove + add with the other static flag into the same section; DeleteRequest: remove only), in list order.
struct TreeOp { uint32_t row; uint8_t kind; };                          // kind: 1 = make static, 2 = wake up, 3 = remove, 4 = add (an entity of this batch: re_ctx::add_keys holds its section decision)
static int rebucket(re_ctx *c, uint32_t n_movers, const std::vector<TreeOp> *pre = nullptr, const std::set<uint64_t> *ghost_touched = nullptr) {
    hipStream_t st = c->stream;
    static const bool timing = getenv("RE_EXP_TIME_REBUCKET") != nullptr;
    auto t_begin = std::chrono::steady_clock::now(); auto lap = [&](const char *what) { if (timing) { auto t = std::chrono::steady_clock::now(); fprintf(stderr, "  rebucket %-10s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(t - t_begin).count()); t_begin = t; } };
    if (n_movers > c->list_cap) return c->fail(RE_E_CAPACITY, "mover list overflow");
    std::vector<uint32_t> movers; bool second_batch = false;              // second_batch: the device took the movers between unique sections, these are the rest
    if (c->shard_hi && n_movers) {                                          // a sharded world: remember who moved (re_list_migrants looks at where they went)
        const uint32_t m0 = std::min(n_movers, c->list_cap); const size_t at = c->moved_rows.size();
        c->moved_rows.resize(at + m0);
        HIPCHK(c, hipMemcpy(c->moved_rows.data() + at, c->d_movers.p, (size_t)m0 * 4, hipMemcpyDeviceToHost));
    }
    // The device takes a tick's movers, and a change batch that moved and / or DELETED non-static entities (no make-static / wake-up / add, no ghost of the frozen cache touched):
    // the deleted rows are appended to the mover list (RB2_MOVER_DELETED) in the order of the batch.
    bool only_deletes = true; uint32_t n_del = 0;                             // (deletions and additions of non-static entities)
    if (pre) for (const TreeOp &op : *pre) {
        bool ok = (op.kind == 3 && !(c->h_flags[op.row] & F_STATIC));
        if (op.kind == 4) { auto ak = c->add_keys.find(op.row); ok = ak != c->add_keys.end() && !ak->second.is_static && !(c->h_flags[op.row] & F_STATIC); }
        if (!ok) { only_deletes = false; break; }
        n_del++;
    }
    if (only_deletes && (!ghost_touched || ghost_touched->empty()) && n_movers + n_del <= c->list_cap && (n_movers + n_del) != 0) {
        if (n_del) {
            std::vector<uint32_t> del; del.reserve(n_del);
            for (const TreeOp &op : *pre) del.push_back(op.row | (op.kind == 4 ? RB2_MOVER_ADDED : RB2_MOVER_DELETED));
            HIPCHK(c, hipMemcpyAsync(c->d_movers.p + n_movers, del.data(), (size_t)n_del * 4, hipMemcpyHostToDevice, st));
            HIPCHK(c, sync_stream(st));                                       // (`del` goes out of scope)
        }
        int drc = rebucket_on_device2(c, n_movers + n_del, &movers, n_del);
        if (drc < 0) return drc;
        if (drc == 0) {
            for (uint32_t i2 = 0; i2 < n_del; i2++) {                         // (a deleted row is in no section: no later sync_mirrors will look at it; an added one is a member of a noted section)
                if ((*pre)[i2].kind != 3) continue;
                const uint32_t r = (*pre)[i2].row; c->h_row_shared_keys.erase(r); c->h_row_nk[r] = 0; c->h_row_key[r] = 0; c->h_row_cell[r] = ROW_CELL_NONE;
            }
            if (movers.empty()) return RE_OK;
            second_batch = true; pre = nullptr;                               // (the deletions are done; cannot happen with deletions: such a batch is not split)
        }
    }
    c->n_host_rebuckets++;
    { int src = sync_mirrors(c); if (src != RE_OK) return src; }
    const uint32_t M = second_batch ? (uint32_t)movers.size() : std::min(n_movers, c->list_cap);
    if (!second_batch) { movers.resize(M); HIPCHK(c, hipMemcpy(movers.data(), c->d_movers.p, (size_t)M * 4, hipMemcpyDeviceToHost)); }
    std::sort(movers.begin(), movers.end(), [&](uint32_t a, uint32_t b) {
        bool ta = (a >> 31) != 0, tb = (b >> 31) != 0;                 // translation-only first
        if (ta != tb) return ta;
        return c->h_id[a & 0x7FFFFFFFu] < c->h_id[b & 0x7FFFFFFFu];
    });
    DevBuf<uint32_t> &d_list = c->d_hrb_list; DevBuf<uint8_t> &d_nk = c->d_hrb_nk; DevBuf<uint64_t> &d_keys = c->d_hrb_keys;      // kept across calls (three hipMalloc / hipFree pairs per batch cost ~0.4 ms)
    if (d_list.n < std::max(M, 1u)) { const uint32_t cap = std::max(2u * M, 4096u); HIPCHK(c, d_list.alloc(cap, nullptr)); HIPCHK(c, d_nk.alloc(cap, nullptr)); HIPCHK(c, d_keys.alloc((size_t)cap * 8, nullptr)); }
    std::vector<uint8_t> nk(M); std::vector<uint64_t> nkeys((size_t)M * 8);
    if (M) {
        HIPCHK(c, hipMemcpyAsync(d_list.p, movers.data(), (size_t)M * 4, hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(k_assign_rows, dim3((M + 255) / 256), dim3(256), 0, st, M, d_list.p, row_arrays(c), c->cfg.outline_length, c->cfg.atomic_length, d_nk.p, d_keys.p);
        HIPCHK(c, hipMemcpyAsync(nk.data(), d_nk.p, M, hipMemcpyDeviceToHost, st));
        HIPCHK(c, hipMemcpyAsync(nkeys.data(), d_keys.p, (size_t)M * 64, hipMemcpyDeviceToHost, st));
    }
    Carry carry;
    if (ghost_touched) carry.ghost_touched = *ghost_touched;
    if (M) HIPCHK(c, sync_stream(st));
    lap("assign");
    const uint32_t os = c->nsh;
    std::map<uint64_t, std::vector<uint32_t>> arrive; std::vector<uint32_t> removed_rows;
    // ---- replay of remove_entity / add_entity on the affected sections (counts only)
    struct CS { uint32_t nl, ns, links; bool exists; };
    struct SS { uint32_t na, nst; bool exists; };
    std::unordered_map<uint64_t, CS> cs; cs.reserve((size_t)M * 4u + 64u); std::map<SharedIdPub, SS> ss;
    std::unordered_map<uint64_t, uint32_t> link_count; link_count.reserve((size_t)os * 4u + 16u);   // shared sections linking each section, from the previous structure
    std::map<SharedIdPub, uint32_t> shid_index;                           // shared section id -> its index in the previous table (a linear search per lookup cost 6 ms per tick with ~1,000 shared sections)
    for (uint32_t s = 0; s < os; s++) shid_index.emplace(c->h_shids[s], s);
    for (uint32_t s = 0; s < os; s++) for (uint32_t k = 0; k < c->h_shids[s].nk; k++) link_count[c->h_shids[s].keys[k]]++;
    auto cell = [&](uint64_t key) -> CS & {
        auto it = cs.find(key);
        if (it != cs.end()) return it->second;
        CS v{ 0, 0, 0, false };
        const int32_t sl = find_slot(c, key);
        if (sl >= 0) { v.nl = c->h_cell_nl[sl]; v.ns = c->h_cell_ns[sl]; v.exists = true; auto lc = link_count.find(key); v.links = lc == link_count.end() ? 0u : lc->second; }
        return cs.emplace(key, v).first->second;
    };
    auto shared = [&](const SharedIdPub &id) -> SS & {
        auto it = ss.find(id);
        if (it != ss.end()) return it->second;
        SS v{ 0, 0, false };
        auto o = shid_index.find(id);
        if (o != shid_index.end()) { const size_t i = o->second; v.na = c->h_sh_nact[i]; v.nst = c->h_sh_nstat[i]; v.exists = true; }
        return ss.emplace(id, v).first->second;
    };
    auto mark_shared = [&](const SharedIdPub &id) { if (carry.changed_shared_set.insert(id).second) carry.changed_shared.push_back(id); };
    uint32_t total = 0;
    // one tree operation: remove_entity (:787-942) from the row's current section, then (unless remove_only) add_entity (:563-762)
    // into the section given by (new_nk, new_keys) with the given static flag
    auto replay = [&](uint32_t r, bool remove_only, uint32_t new_nk, const uint64_t *new_keys, bool new_static) {
        const bool was_static = (c->h_flags[r] & F_STATIC) != 0;
        if (c->h_row_nk[r] > 1) {
            SharedIdPub id; id.nk = c->h_row_nk[r]; memcpy(id.keys, c->h_row_shared_keys[r].data(), sizeof id.keys);
            SS &sh = shared(id);
            if (was_static) { for (uint32_t k = 0; k < id.nk; k++) carry.changed_static.insert(id.keys[k]); if (sh.nst) sh.nst--; } else if (sh.na) sh.na--;
            if (sh.na == 0 && sh.nst == 0) {
                for (uint32_t k = 0; k < id.nk; k++) { CS &cl = cell(id.keys[k]); if (cl.links) cl.links--; if (cl.nl == 0 && cl.ns == 0 && cl.links == 0) cl.exists = false; }
                sh.exists = false;
            }
            mark_shared(id);
        } else if (c->h_row_nk[r] == 1) {
            const uint64_t key = c->h_row_key[r];
            CS &cl = cell(key);
            if (was_static) { if (cl.ns) cl.ns--; carry.changed_static.insert(key); } else if (cl.nl) cl.nl--;
            if (cl.nl == 0 && cl.ns == 0 && cl.links == 0) cl.exists = false;
            else total += carry.changed_cells.count(key) ? 1u : cl.nl + cl.ns;
            carry.changed_cells.insert(key);
        }
        if (remove_only) { c->h_row_shared_keys.erase(r); c->h_row_nk[r] = 0; c->h_row_key[r] = 0; removed_rows.push_back(r); return; }
        if (new_nk > 1) {
            SharedIdPub id; id.nk = new_nk; memcpy(id.keys, new_keys, sizeof id.keys);
            SS &sh = shared(id);
            if (new_static) for (uint32_t k = 0; k < id.nk; k++) carry.changed_static.insert(id.keys[k]);            // :590-596
            if (!sh.exists) { sh = SS{ 0, 0, true }; for (uint32_t k = 0; k < id.nk; k++) { CS &cl = cell(id.keys[k]); if (!cl.exists) cl = CS{ 0, 0, 0, true }; cl.links++; } }
            if (new_static) sh.nst++; else sh.na++;
            mark_shared(id);
            std::array<uint64_t, 8> a; memcpy(a.data(), id.keys, sizeof id.keys); c->h_row_shared_keys[r] = a;
            c->h_row_key[r] = id.keys[0];
        } else if (new_nk == 1) {
            const uint64_t key = new_keys[0];
            CS &cl = cell(key);
            if (cl.exists) { if (new_static) cl.ns++; else cl.nl++; total += carry.changed_cells.count(key) ? 1u : cl.nl + cl.ns; }
            else { cl = CS{ new_static ? 0u : 1u, new_static ? 1u : 0u, 0, true }; total += 1; }
            if (new_static) carry.changed_static.insert(key);
            carry.changed_cells.insert(key);
            c->h_row_shared_keys.erase(r);
            c->h_row_key[r] = key;
            arrive[key].push_back(r);
        }
        c->h_row_nk[r] = (uint8_t)new_nk;
        if (new_static) c->h_flags[r] |= F_STATIC; else c->h_flags[r] &= ~F_STATIC;
    };
    if (pre) for (const TreeOp &op : *pre) {
        const uint32_t r = op.row;
        if (op.kind == 4) {                                                  // apply_choices -> add_entity(id, aabb, false, is_static, ..): nothing to remove first; out of bounds -> not inserted
            auto ak = c->add_keys.find(r);
            if (ak != c->add_keys.end() && ak->second.nk) replay(r, false, ak->second.nk, ak->second.keys, ak->second.is_static);
            continue;
        }
        uint64_t cur[8] = {}; const uint32_t cnk = c->h_row_nk[r];
        if (cnk > 1) memcpy(cur, c->h_row_shared_keys[r].data(), sizeof cur); else cur[0] = c->h_row_key[r];
        if (op.kind == 3) replay(r, true, 0, nullptr, false);
        else if (cnk) replay(r, false, cnk, cur, op.kind == 1);            // same StaticAABB => same section(s)
    }
    for (uint32_t i = 0; i < M; i++) {
        const uint32_t r = movers[i] & 0x7FFFFFFFu;
        // add_entity returns early for an entity that stays where it is (entity_exists_in_section, bounding_box_tree_v2.rs:765-782).  The device lists
        // only entities that change section -- except an entity ADDED by this very batch and changed again in it: the device did not know its section yet.
        if (nk[i] && c->h_row_nk[r] == nk[i] && (nk[i] == 1 ? c->h_row_key[r] == nkeys[(size_t)i * 8] : memcmp(c->h_row_shared_keys[r].data(), &nkeys[(size_t)i * 8], (size_t)nk[i] * 8) == 0) && c->add_keys.count(r)) continue;
        replay(r, false, nk[i], &nkeys[(size_t)i * 8], false);               // update_entity_in_tree: is_static = false (:330)
    }
    // a hidden (made-static-after-the-cache-froze) row that a re-add turned non-static again is drawn again
    std::vector<uint32_t> reveal;
    for (auto it = c->h_uncached.begin(); it != c->h_uncached.end();) { if (!(c->h_flags[*it] & F_STATIC)) { reveal.push_back(*it); it = c->h_uncached.erase(it); } else ++it; }
    for (uint32_t r : reveal) if (!(c->h_flags[r] & F_DEAD)) HIPCHK(c, hipMemcpyAsync(c->d_gclass.p + r, &c->h_gclass[r], 4, hipMemcpyHostToDevice, st));
    // (their pool entries are rewritten by the patch / rebuild below with the row's now visible group class: they moved)
    carry.too_many = total > 500 || second_batch;                             // (second batch: the device part alone was above the threshold)
    // ---- write the result into the resident table; rebuild everything only when its slack is exhausted
    lap("replay");
    {
        int prc = (c->cfg.flags & RE_CFG_FULL_REBUILD) ? 1 : patch_sections(c, carry, arrive, removed_rows);
        lap("patch");
        if (prc == 0) {
            std::vector<Pair32> gc; for (uint32_t r : reveal) collect_row_gc(c, r, gc);      // revealed rows that kept their pool position
            return upload_row_gc(c, gc);
        }
        if (prc < 0) return prc;
    }
    // full rebuild: key-sorted arrays from the patched per-row decisions, carrying over what the reference leaves untouched.
    // The previous state is fetched by key (slots are no longer key-ordered after patches).
    if (!(c->cfg.flags & RE_CFG_FULL_REBUILD) && c->slack_boost < 64u) c->slack_boost *= 2u;      // (the slack was used up: the rebuilt table gets twice the spare slots)
    {
        const uint32_t oc = c->ncells;
        std::vector<Aabb> tight(oc); std::vector<uint8_t> fl(oc);
        if (oc) { HIPCHK(c, hipMemcpy(tight.data(), c->d_cell_tight.p, (size_t)oc * sizeof(Aabb), hipMemcpyDeviceToHost)); HIPCHK(c, hipMemcpy(fl.data(), c->d_cell_flags.p, oc, hipMemcpyDeviceToHost)); }
        std::vector<uint32_t> order; order.reserve(oc);
        for (uint32_t i = 0; i < oc; i++) if ((c->h_cell_key[i] & 0xFFFFFFFFFFFFull) != 0xFFFFFFFFFFFFull) order.push_back(i);
        std::sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b2) { return c->h_cell_key[a] < c->h_cell_key[b2]; });
        carry.keys.resize(order.size()); carry.tight.resize(order.size()); carry.flags.resize(order.size());
        for (size_t i = 0; i < order.size(); i++) { carry.keys[i] = c->h_cell_key[order[i]]; carry.tight[i] = tight[order[i]]; carry.flags[i] = fl[order[i]]; }
        carry.shids = c->h_shids; carry.sh_aabb.resize(os); carry.sh_cached.resize(os);
        std::vector<int32_t> sh_owner(os);
        if (os) {
            HIPCHK(c, hipMemcpy(carry.sh_aabb.data(), c->d_sh_aabb.p, (size_t)os * sizeof(Aabb), hipMemcpyDeviceToHost));
            HIPCHK(c, hipMemcpy(carry.sh_cached.data(), c->d_sh_cached.p, os, hipMemcpyDeviceToHost));
            HIPCHK(c, hipMemcpy(sh_owner.data(), c->d_sh_owner.p, (size_t)os * 4, hipMemcpyDeviceToHost));
        }
        carry.sh_owner_key_idx = sh_owner; carry.sh_owner_key.resize(os, 0);
        for (uint32_t s2 = 0; s2 < os; s2++) if (sh_owner[s2] >= 0) carry.sh_owner_key[s2] = c->h_cell_key[sh_owner[s2]];
    }
    std::vector<SharedRec> shrec; shrec.reserve(c->h_row_shared_keys.size());
    for (auto &kv : c->h_row_shared_keys) { SharedRec sr; sr.row = kv.first; sr.nk = c->h_row_nk[kv.first]; memcpy(sr.keys, kv.second.data(), sizeof sr.keys); shrec.push_back(sr); }
    std::vector<uint32_t> flags(c->h_flags);
    int rc = build_sections(c, c->h_row_key, c->h_row_nk, shrec, flags, &carry);
    if (rc != RE_OK) return rc;
    rc = upload_dyn_cells(c);
    if (rc != RE_OK) return rc;
    c->n_rebuilds++;
    return RE_OK;
}

// rows the last tick / change batch removed because they left the world: mirror RE_F_DEAD on the host
static int absorb_out_of_bounds(re_ctx *c, uint32_t n_oob) {
    uint32_t cnt = std::min(n_oob, c->list_cap);
    if (!cnt) return RE_OK;
    { int src = sync_mirrors(c); if (src != RE_OK) return src; }
    std::vector<uint32_t> rows(cnt);
    HIPCHK(c, hipMemcpy(rows.data(), c->d_oob.p, (size_t)cnt * 4, hipMemcpyDeviceToHost));
    std::vector<Pair32> gc;
    for (uint32_t r : rows) if (r < c->n) { c->h_flags[r] |= F_DEAD; c->h_oob_ids.push_back(c->h_id[r]); collect_row_gc(c, r, gc); }   // the tree keeps the stale entry: stop drawing it
    return upload_row_gc(c, gc);
}

static int finish_tick(re_ctx *c, re_tick_result *out) {
    if (!c->ndyn) {                                                           // nothing ticks: whatever re_tick enqueued (clearing the changed-static set) is ordered on the stream
        c->tick_inflight = false; c->last_tick = re_tick_result{ 0, 0, 0 };
        if (out) *out = c->last_tick;
        return RE_OK;
    }
    // fast completion: the counters arrive in mapped host memory with the last wave of the tick (tick_sign_off).  A tick that found movers or entities
    // leaving the world (the stale word is raised before the counters are published) still needs resolve(): patch the tree, replay.
    bool done = false;
    if (c->tick_published && (!c->park_ready || !(c->park.busy || c->park.deferred_pack))) {
        const volatile uint32_t *flag = &c->h_th->ticket;
        const auto t0 = std::chrono::steady_clock::now();
        for (uint32_t spins = 0; !(done = (*flag == c->tick_seq)); spins++)
            if ((spins & 1023u) == 1023u && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(2)) break;
        if (done) {
            std::atomic_thread_fence(std::memory_order_acquire);
            auto sealed = [&]() { const volatile TickHeader *t = c->h_th; return t->pad[0] == (table_word_hash(t->n_changed, 1u) ^ table_word_hash(t->n_rebucket, 2u) ^ table_word_hash(t->n_oob, 3u) ^ table_word_hash(c->tick_seq, 4u)); };
            if (!sealed()) {                                                  // (does not happen with publish_to_host; counted, and asserted to be 0 by the tests)
                c->n_seal_waits++;
                const auto t1 = std::chrono::steady_clock::now();
                while (!sealed() && std::chrono::steady_clock::now() - t1 < std::chrono::microseconds(200)) {}
                if (!sealed()) c->n_sync_fallbacks++;
            }
            if (!sealed() || (c->h_spec && c->h_spec->stale)) done = false;
        }
    }
    if (!done) {
        int rc = resolve(c);
        if (rc != RE_OK) return rc;
        // (after resolve the stream has drained: the published counters are those of the last tick that ran, replays included)
        const volatile uint32_t *flag = &c->h_th->ticket;
        const auto t0 = std::chrono::steady_clock::now();
        while (c->tick_published && *flag != c->tick_seq && std::chrono::steady_clock::now() - t0 < std::chrono::milliseconds(200)) {}
    } else c->pending.clear();
    c->tick_inflight = false;
    if (c->timed_tick) { (void)hipEventSynchronize(c->ev[4]); (void)hipEventElapsedTime(&c->t_tick, c->ev[3], c->ev[4]); c->t_tick *= 1000.f; }
    if (c->ndyn) { c->last_tick.n_changed = c->h_th->n_changed; c->last_tick.n_rebucket = c->h_th->n_rebucket; c->last_tick.n_out_of_bounds = c->h_th->n_oob; }
    else c->last_tick = re_tick_result{ 0, 0, 0 };
    if (out) *out = c->last_tick;
    return RE_OK;
}

static int issue_tick(re_ctx *c, float dt, uint32_t flags) {
    hipStream_t st = c->stream;
    c->timed_tick = c->timings_on && !(flags & RE_TICK_ASYNC);
    if (c->timed_tick) HIPCHK(c, hipEventRecord(c->ev[3], st));
    if (c->ndyn) {
        if (!c->th_clean) HIPCHK(c, hipMemsetAsync(c->d_th.p, 0, sizeof(TickHeader), st));     // normally zeroed by the pack kernel of the frame
        hipEvent_t ta = nullptr, tb = nullptr;
        if (c->k1_timing && c->k1_kind == RE_TIME_TICK) take_timing_events(c, &ta, &tb);
        hipExtLaunchKernelGGL(k_tick, dim3((c->ndyn + 255) / 256), dim3(256), 0, st, ta, tb, 0, c->ndyn, c->d_dyn_vel.p, c->d_dyn_acc.p, c->d_dyn_rotvel.p, c->d_dyn_rotacc.p,
                           row_arrays(c), c->d_row_cell.p, c->d_cell_key.p, c->d_cell_stamp.p, c->d_cell_flags.p, c->d_sh_cells.p, c->d_sh_aabb.p, c->d_params.p, dt,
                           (flags & RE_TICK_ALL_DYNAMIC) ? 1u : 0u, c->cfg.outline_length, c->cfg.atomic_length, c->d_th.p, c->d_movers.p, c->d_oob.p, c->list_cap, c->d_spec.p, c->d_hspec, c->frame,
                           c->ndyn0, c->d_dyn_row.p, c->d_hth, (flags & RE_TICK_ASYNC) ? 0u : c->tick_seq + 1u);      // a synchronous tick publishes its counters itself (tick_sign_off): no second launch
        c->th_clean = false;
        c->tick_published = !(flags & RE_TICK_ASYNC);
        if (c->tick_published) ++c->tick_seq;
    }
    if (c->dirty_pending) {                                                     // Pipeline::execute: clear_changed_static_unique (pipeline.rs:271)
        uint32_t m = std::max(c->ncells, c->nsh);
        if (m) hipLaunchKernelGGL(k_clear_static_dirty, dim3((m + 255) / 256), dim3(256), 0, st, c->ncells, c->d_cell_flags.p, c->nsh, c->d_sh_dirty.p);
        c->dirty_pending = false;
    }
    HIPCHK(c, hipGetLastError());
    if (c->timed_tick) HIPCHK(c, hipEventRecord(c->ev[4], st));
    c->tick_inflight = true;
    c->pending.push_back(re_ctx::PendingCall{ 1, c->frame, re_camera{}, flags, dt, nullptr, nullptr, 0, nullptr });
    return RE_OK;
}

extern "C" int re_tick(re_ctx *c, float dt, uint32_t flags, re_tick_result *out) try {
    if (!c) return RE_E_ARG;
    if (!c->h_res) return c->fail(RE_E_STATE, "re_tick: no world uploaded");
    if (!(flags & RE_TICK_ALL_DYNAMIC) && !c->have_cull) return c->fail(RE_E_STATE, "re_tick: reference semantics tick entities of the last visibility query; call re_cull_pack first or pass RE_TICK_ALL_DYNAMIC");
    if (dt == 0.0f && c->has_rotvel) return c->fail(RE_E_ARG, "re_tick: delta_time == 0 with rotating entities (the reference asserts, exports/movement_components.rs:287)");
    HIPCHK(c, hipSetDevice(c->device));
    // (a second tick of the same frame while the first is still in flight would carry the same frame number in the stale word: settle the first)
    if (c->ndyn && c->tick_inflight && c->tick_frame == c->frame) { int rc0 = finish_tick(c, nullptr); if (rc0 != RE_OK) return rc0; }
    c->tick_frame = c->frame;
    int rc = issue_tick(c, dt, flags);
    if (rc != RE_OK || (flags & RE_TICK_ASYNC)) return rc;
    return finish_tick(c, out);
} RE_ABI_GUARD(c, "re_tick")

// Synchronise and settle speculation: when a tick raised `stale` (entities changed section or left the world), everything enqueued
// after it has cancelled itself; patch the tree from that tick's lists, then replay the cancelled calls (which may go stale again).
// the counters of the last tick that ran, copied from the device (stream-ordered); k_tick counts n_changed in shards
static int fetch_tick_counters(re_ctx *c) {
    TickHeader t;
    HIPCHK(c, hipMemcpyAsync(&t, c->d_th.p, sizeof t, hipMemcpyDeviceToHost, c->stream)); HIPCHK(c, sync_stream(c->stream));
    for (uint32_t k = 0; k < TICK_TICKET_SHARDS; k++) t.n_changed += t.shard[k * TICK_SHARD_STRIDE];
    c->h_th->n_changed = t.n_changed; c->h_th->n_rebucket = t.n_rebucket; c->h_th->n_oob = t.n_oob;
    return RE_OK;
}
static int resolve(re_ctx *c) {
    { int rc = drain_other_lane(c); if (rc != RE_OK) return rc; }
    { int rc = flush_deferred_pack(c); if (rc != RE_OK) return rc; }
    HIPCHK(c, sync_stream(c->stream));
    // n_changed, n_rebucket, n_oob of the last tick that ran: a synchronous tick has published them into the mapped block itself (tick_sign_off); otherwise a stream-ordered copy
    auto published = [&]() { return c->tick_published && c->h_th && *reinterpret_cast<const volatile uint32_t *>(&c->h_th->ticket) == c->tick_seq; };
    if (c->tick_inflight && c->ndyn && c->h_th && !published()) { int rc_ = fetch_tick_counters(c); if (rc_ != RE_OK) return rc_; }
    while (c->h_spec && c->h_spec->stale) {
        const uint32_t sf = c->h_spec->stale_frame;
        c->h_spec->stale = 0; HIPCHK(c, hipMemsetAsync(c->d_spec.p, 0, sizeof(SpecState), c->stream));      // (stream-ordered: in front of whatever is launched next)
        const TickHeader th = *c->h_th;
        if (th.n_oob) { int rc = absorb_out_of_bounds(c, th.n_oob); if (rc != RE_OK) return rc; c->n_dead += th.n_oob; }
        c->last_tick = re_tick_result{ th.n_changed, th.n_rebucket, th.n_oob };
        if (th.n_rebucket) { int rc = rebucket(c, th.n_rebucket); if (rc != RE_OK) return rc; }
        // calls enqueued after that tick did nothing: run them again on the patched tree
        std::vector<re_ctx::PendingCall> replay; bool after = false;
        for (const auto &pc : c->pending) { if (after) replay.push_back(pc); else if (pc.kind == 1 && pc.frame == sf) after = true; }
        c->pending.clear();
        if (!replay.empty()) {
            HIPCHK(c, hipMemsetAsync(c->d_hdr.p, 0, NUM_FRAME_HEADERS * sizeof(FrameHeader), c->stream)); c->th_clean = false;
            if (c->d_gcount.p) {                                              // a cancelled pack cleared nothing: start the replay from clean count / fill arrays
                HIPCHK(c, hipMemsetAsync(c->d_gcount.p, 0, c->d_gcount.n * 4, c->stream)); HIPCHK(c, hipMemsetAsync(c->d_gfill.p, 0, c->d_gfill.n * 4, c->stream));
                c->gc_dirty[0] = c->gc_dirty[1] = false;
            }
            c->cull_inflight = false;
            uint32_t *keep_ids = c->ext_out_ids; float *keep_mats = c->ext_out_mats; uint32_t keep_cap = c->ext_out_cap, *keep_cnt = c->ext_out_count;
            for (const auto &pc : replay) {
                if (pc.kind == 0) { c->ext_out_ids = pc.out_ids; c->ext_out_mats = pc.out_mats; c->ext_out_cap = pc.out_cap; c->ext_out_count = pc.out_count; }   // the buffers that frame was issued with
                int rc = pc.kind == 0 ? issue_cull(c, &pc.cam, pc.flags | RE_CULL_ASYNC) : issue_tick(c, pc.dt, pc.flags | RE_TICK_ASYNC);
                if (rc != RE_OK) return rc;
            }
            c->ext_out_ids = keep_ids; c->ext_out_mats = keep_mats; c->ext_out_cap = keep_cap; c->ext_out_count = keep_cnt;
        }
        if (!replay.empty()) {                                              // (without a replay nothing has run since the counters above were taken)
            HIPCHK(c, sync_stream(c->stream));
            if (c->ndyn) { int rc_ = fetch_tick_counters(c); if (rc_ != RE_OK) return rc_; }
        }
    }
    c->pending.clear();
    return RE_OK;
}

// ------------------------------------------------------------------------------------------------
// Entities added after the upload: Pipeline::register_model_instances at any time (flows/pipeline.rs:186-208) and the AddEntity arm of
// apply_change (helper_things/entity_change_helpers.rs:48-107).  A new entity takes the next row of the per-entity columns (grown in
// place, amortised: ensure_row_capacity), a slot of the dynamic table when it carries Velocity / VelocityRotation (k_tick reaches rows
// outside the leading block through d_dyn_row), possibly a new (ModelId, sortable) group class (regrow_groups), and enters the tree through
// the same re-bucket as a mover -- an add_entity without the remove_entity in front of it (rebucket(): TreeOp kind 4).
// ------------------------------------------------------------------------------------------------
template <typename T> static hipError_t grow_buf(DevBuf<T> &b, size_t keep, size_t count, uint64_t *acct, hipStream_t st) {
    DevBuf<T> nb; hipError_t e = nb.alloc(count, acct);
    if (e != hipSuccess) return e;
    if (keep && b.p) { e = hipMemcpyAsync(nb.p, b.p, keep * sizeof(T), hipMemcpyDeviceToDevice, st); if (e == hipSuccess) e = sync_stream(st); }
    if (e != hipSuccess) { nb.release(acct); return e; }
    b.release(acct); b = nb;
    return hipSuccess;
}
// per-frame buffers whose size follows the number of rows: instance lists, packed output (when the caller gave no capacity), mover / out-of-bounds lists
static int size_frame_buffers(re_ctx *c) {
    uint64_t *acct = &c->dev_bytes;
    const uint32_t n = c->row_cap;
    const uint32_t out_cap = c->cfg.max_instances ? c->cfg.max_instances : std::max(n, 1u);
    uint32_t item_cap = (std::max(4u * n, 64u) + 64u * CURSOR_SHARDS) / CURSOR_SHARDS * CURSOR_SHARDS;   // 2n instances (duplicates mode) with 2x head-room per cursor segment
    if (n <= 65536u) item_cap = CURSOR_SHARDS * (2u * (n + std::max(2048u, n / 8u)) + 64u);   // a small world's sections sit in a few waves, i.e. in a few cursor shards: every segment holds the whole world incl. its ghost instances (twice: duplicates mode)
    const uint32_t list_cap = std::max(std::max(c->dyn_cap, std::min(n, 65536u)), 1u);   // movers of one tick (dynamic rows) or of one change batch (any row)
    if (item_cap > c->item_cap || !c->d_item_row.p) {
        // two instance lists, alternating by frame: a deferred pack (RE_CULL_DEFER_PACK) reads the list of frame f while the scan of frame f + 1 fills the other
        c->item_cap = item_cap;
        HIPCHK(c, c->d_item_row.alloc((size_t)c->item_cap * 2, acct)); HIPCHK(c, c->d_item_slot.alloc((size_t)c->item_cap * 2, acct));
        HIPCHK(c, hipMemset(c->d_item_row.p, 0, (size_t)c->item_cap * 8)); HIPCHK(c, hipMemset(c->d_item_slot.p, 0xFF, (size_t)c->item_cap * 8));   // the pack reads speculatively past the cursors
    }
    if (out_cap > c->out_cap || !c->d_out_ids.p) { c->out_cap = out_cap; HIPCHK(c, c->d_out_ids.alloc(c->out_cap, acct)); HIPCHK(c, c->d_out_mats.alloc((size_t)c->out_cap * 16, acct)); }
    if (list_cap > c->list_cap || !c->d_movers.p) { c->list_cap = list_cap; HIPCHK(c, c->d_movers.alloc(c->list_cap, acct)); HIPCHK(c, c->d_oob.alloc(c->list_cap, acct)); }
    return RE_OK;
}
static int ensure_row_capacity(re_ctx *c, uint32_t need) {
    if (need <= c->row_cap) return RE_OK;
    if (c->park_ready) { (void)drain_other_lane(c); free_second_lane(c); }
    hipStream_t st = c->stream; uint64_t *a = &c->dev_bytes;
    HIPCHK(c, sync_stream(st));
    const uint32_t old_cap = c->row_cap, new_cap = std::max(need, old_cap + old_cap / 4u + 1024u), n = c->n;
    const uint32_t new_ghost_cap = std::max(c->ghost_cap, std::max(2048u, new_cap / 8u));
    HIPCHK(c, grow_buf(c->d_gclass, n, new_cap, a, st)); HIPCHK(c, grow_buf(c->d_flags, n, new_cap, a, st));
    HIPCHK(c, grow_buf(c->d_pos, (size_t)n * 3, (size_t)new_cap * 3, a, st)); HIPCHK(c, grow_buf(c->d_rot, (size_t)n * 4, (size_t)new_cap * 4, a, st)); HIPCHK(c, grow_buf(c->d_scale, (size_t)n * 3, (size_t)new_cap * 3, a, st));
    HIPCHK(c, grow_buf(c->d_aabb, n, new_cap, a, st)); HIPCHK(c, grow_buf(c->d_orig, n, new_cap, a, st));
    HIPCHK(c, grow_buf(c->d_row_cell, n, new_cap, a, st)); HIPCHK(c, grow_buf(c->d_row_key, n, new_cap, a, st));
    HIPCHK(c, hipMemsetAsync(c->d_row_cell.p + n, 0xFF, (size_t)(new_cap - n) * 4, st));                  // ROW_CELL_NONE: not in the tree
    {   // id / matrix columns: rows, then the ghost instances behind the row capacity
        DevBuf<uint32_t> nid; DevBuf<float> nmat;
        HIPCHK(c, nid.alloc((size_t)new_cap + new_ghost_cap, a)); HIPCHK(c, nmat.alloc(((size_t)new_cap + new_ghost_cap) * 16, a));
        if (n) { HIPCHK(c, hipMemcpyAsync(nid.p, c->d_id.p, (size_t)n * 4, hipMemcpyDeviceToDevice, st)); HIPCHK(c, hipMemcpyAsync(nmat.p, c->d_mat.p, (size_t)n * 64, hipMemcpyDeviceToDevice, st)); }
        if (c->n_ghost) {
            HIPCHK(c, hipMemcpyAsync(nid.p + new_cap, c->d_id.p + old_cap, (size_t)c->n_ghost * 4, hipMemcpyDeviceToDevice, st));
            HIPCHK(c, hipMemcpyAsync(nmat.p + (size_t)new_cap * 16, c->d_mat.p + (size_t)old_cap * 16, (size_t)c->n_ghost * 64, hipMemcpyDeviceToDevice, st));
        }
        HIPCHK(c, sync_stream(st));
        c->d_id.release(a); c->d_mat.release(a); c->d_id = nid; c->d_mat = nmat;
    }
    if (c->n_ghost) {                                                        // the row pool names ghost instances by row: they moved up with the capacity
        const uint32_t delta = new_cap - old_cap;
        if (c->pool_used) hipLaunchKernelGGL(k_shift_rows, dim3((c->pool_used + 255) / 256), dim3(256), 0, st, c->pool_used, c->d_rows.p, old_cap, delta);
        for (uint32_t i = 0; i < c->pool_used && i < c->h_rows.size(); i++) if (c->h_rows[i] >= old_cap && c->h_rows[i] != 0xFFFFFFFFu) c->h_rows[i] += delta;
        for (auto &kv : c->ghost_map) for (uint32_t &g : kv.second) g += delta;
        HIPCHK(c, hipGetLastError()); HIPCHK(c, sync_stream(st));
    }
    c->row_cap = c->ghost_base = new_cap; c->ghost_cap = new_ghost_cap;
    c->last_pack.kind = 0;                                                   // (what the last pack was launched with names the old columns)
    c->d_row_moved.release(nullptr); c->d_col_moved.release(nullptr); c->d_col_tab.release(nullptr); c->col_moved_cap = 0;   // collision scratch: sized at its next use
    return size_frame_buffers(c);
}
static int ensure_dyn_capacity(re_ctx *c, uint32_t need) {
    if (need <= c->dyn_cap) return RE_OK;
    hipStream_t st = c->stream; uint64_t *a = &c->dev_bytes;
    HIPCHK(c, sync_stream(st));
    const uint32_t new_cap = std::max(need, c->dyn_cap + c->dyn_cap / 4u + 256u), k = c->ndyn;
    HIPCHK(c, grow_buf(c->d_dyn_row, k, new_cap, a, st)); HIPCHK(c, grow_buf(c->d_dyn_vel, (size_t)k * 3, (size_t)new_cap * 3, a, st)); HIPCHK(c, grow_buf(c->d_dyn_acc, (size_t)k * 3, (size_t)new_cap * 3, a, st));
    HIPCHK(c, grow_buf(c->d_dyn_rotvel, (size_t)k * 4, (size_t)new_cap * 4, a, st)); HIPCHK(c, grow_buf(c->d_dyn_rotacc, (size_t)k * 4, (size_t)new_cap * 4, a, st));
    c->dyn_cap = new_cap;
    c->col_moved_cap = 0; c->d_col_moved.release(nullptr); c->d_col_tab.release(nullptr); c->d_row_moved.release(nullptr);
    return size_frame_buffers(c);
}
// a slot of the dynamic table for a row that had none (an added entity with a velocity, or Velocity written to an entity registered without one:
// the reference registers the component on write, objects/entity_change_request.rs:29-30); the velocities start as the defaults of the upload
static int alloc_dyn_slot(re_ctx *c, uint32_t row, uint32_t *slot) {
    uint32_t j = 0;
    if (c->dyn_index(row, j)) { *slot = j; return RE_OK; }
    { int rc = ensure_dyn_capacity(c, c->ndyn + 1u); if (rc != RE_OK) return rc; }
    j = c->ndyn++;
    const float z3[3] = { 0.f, 0.f, 0.f }, ax[4] = { 1.f, 0.f, 0.f, 0.f };
    HIPCHK(c, hipMemcpy(c->d_dyn_row.p + j, &row, 4, hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(c->d_dyn_vel.p + (size_t)j * 3, z3, 12, hipMemcpyHostToDevice)); HIPCHK(c, hipMemcpy(c->d_dyn_acc.p + (size_t)j * 3, z3, 12, hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(c->d_dyn_rotvel.p + (size_t)j * 4, ax, 16, hipMemcpyHostToDevice)); HIPCHK(c, hipMemcpy(c->d_dyn_rotacc.p + (size_t)j * 4, ax, 16, hipMemcpyHostToDevice));
    c->h_dyn_row.push_back(row); c->dyn_extra[row] = j;
    *slot = j;
    return RE_OK;
}
// group class of a (ModelId, sortable) pair; a new pair is appended to the host table (*grew) and reaches the device with regrow_groups
static uint32_t group_class_of(re_ctx *c, const GroupKey &gk, bool *grew) {
    auto it = c->gmap.find(gk);
    if (it != c->gmap.end()) return it->second;
    const uint32_t g = (uint32_t)c->h_gkeys.size();
    c->gmap.emplace(gk, g); c->h_gkeys.push_back(gk); *grew = true;
    return g;
}
static int upload_lod_tables(re_ctx *c);
static int regrow_groups(re_ctx *c) {
    uint64_t *acct = &c->dev_bytes; hipStream_t st = c->stream;
    if (c->park_ready) { (void)drain_other_lane(c); free_second_lane(c); }
    HIPCHK(c, sync_stream(st));
    c->ngclass = (uint32_t)c->h_gkeys.size(); c->nslots = c->ngclass * 8u;
    std::vector<uint32_t> gm(c->ngclass + 1), gr(c->ngclass + 1), gs(c->ngclass + 1);
    for (uint32_t g = 0; g < c->ngclass; g++) { gm[g] = c->h_gkeys[g].model; gr[g] = c->h_gkeys[g].rs; gs[g] = c->h_gkeys[g].sort; }
    HIPCHK(c, c->d_gc_model.alloc(c->ngclass, acct)); HIPCHK(c, c->d_gc_rs.alloc(c->ngclass, acct)); HIPCHK(c, c->d_gc_sort.alloc(c->ngclass, acct));
    HIPCHK(c, c->d_group_count.alloc(c->nslots, acct)); HIPCHK(c, c->d_group_begin.alloc(c->nslots, acct)); HIPCHK(c, c->d_group_fill.alloc(c->nslots, acct));
    HIPCHK(c, hipMemcpy(c->d_gc_model.p, gm.data(), (size_t)c->ngclass * 4, hipMemcpyHostToDevice)); HIPCHK(c, hipMemcpy(c->d_gc_rs.p, gr.data(), (size_t)c->ngclass * 4, hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(c->d_gc_sort.p, gs.data(), (size_t)c->ngclass * 4, hipMemcpyHostToDevice));
    HIPCHK(c, hipMemset(c->d_group_count.p, 0, (size_t)std::max(c->nslots, 1u) * 4)); HIPCHK(c, hipMemset(c->d_group_fill.p, 0, (size_t)std::max(c->nslots, 1u) * 4));
    c->d_gcount.release(acct); c->d_gfill.release(acct); c->gc_dirty[0] = c->gc_dirty[1] = false;
    if (c->nslots <= COUNT_SLOTS_MAX) {
        const size_t words = 2u * CURSOR_SHARDS * (size_t)std::max(c->nslots, 1u);
        HIPCHK(c, c->d_gcount.alloc(words, acct)); HIPCHK(c, c->d_gfill.alloc(words, acct));
        HIPCHK(c, hipMemset(c->d_gcount.p, 0, words * 4)); HIPCHK(c, hipMemset(c->d_gfill.p, 0, words * 4));
    }
    if (c->nslots > c->nslots_cap) {                                        // the InstanceRange table of the result block
        void *hb = nullptr, *db = nullptr;
        const uint32_t cap = std::max(2u * c->nslots, 1024u);
        HIPCHK(c, alloc_host_block(cap, &hb, &db));
        memcpy(hb, c->h_block, HB_RANGES);                                  // frame result, speculation word, tick counters as they are
        (void)hipHostFree(c->h_block);
        c->h_block = hb; c->nslots_cap = cap;
        c->h_res = hb_at<HostResult>(hb, HB_RES); c->d_hres = hb_at<HostResult>(db, HB_RES);
        c->h_spec = hb_at<SpecState>(hb, HB_SPEC); c->d_hspec = hb_at<SpecState>(db, HB_SPEC);
        c->h_th = hb_at<TickHeader>(hb, HB_TICK); c->d_hth = hb_at<TickHeader>(db, HB_TICK);
        c->h_ranges = hb_at<InstanceRange>(hb, HB_RANGES); c->d_hranges = hb_at<InstanceRange>(db, HB_RANGES);
    }
    c->last_pack.kind = 0;
    return upload_lod_tables(c);
}

// The rows of an add batch: entity E[src] becomes row `row` (consecutive from c->n).  Columns staged and copied, TransformationMatrix / StaticAABB /
// section decision on the device (the upload's kernel over the new row range), host mirrors extended; the tree does not know them yet (add_keys).
struct NewRow { uint32_t src, row; };
static int create_rows(re_ctx *c, const re_entities *E, const std::vector<NewRow> &rows, uint32_t *n_rejected) {
    const uint32_t m = (uint32_t)rows.size(), row0 = c->n;
    if (n_rejected) *n_rejected = 0;
    if (!m) return RE_OK;
    hipStream_t st = c->stream;
    { int rc = ensure_row_capacity(c, row0 + m); if (rc != RE_OK) return rc; }
    std::vector<uint32_t> ids(m), gcl(m), fls(m); std::vector<float> pos((size_t)m * 3), rot((size_t)m * 4), scl((size_t)m * 3), orig((size_t)m * 6);
    struct Dyn { uint32_t k; float v[3], a[3], rv[4], ra[4]; }; std::vector<Dyn> dyn;
    bool grew = false;
    for (uint32_t k = 0; k < m; k++) {
        const size_t i = rows[k].src;
        uint32_t fl = E->flags[i] & ~F_DEAD;                                 // (HasMoved / HasRotated as given: an entity that arrives from another GPU keeps the markers of its last tick)
        if (fl & F_PHANTOM) fl &= ~(F_HAS_VEL | F_HAS_ACC | F_HAS_ROTVEL | F_HAS_ROTACC | F_ALWAYS_EXEC | F_USER | F_LIGHT_ANY);
        if ((fl & F_HAS_ROT) && !E->rotation) return c->fail(RE_E_ARG, "added entity %u has RE_F_HAS_ROT but rotation == NULL", (uint32_t)i);
        if ((fl & F_HAS_SCALE) && !E->scale) return c->fail(RE_E_ARG, "added entity %u has RE_F_HAS_SCALE but scale == NULL", (uint32_t)i);
        if ((fl & F_HAS_VEL) && !E->velocity) return c->fail(RE_E_ARG, "added entity %u has RE_F_HAS_VEL but velocity == NULL", (uint32_t)i);
        if ((fl & F_HAS_ACC) && !E->acceleration) return c->fail(RE_E_ARG, "added entity %u has RE_F_HAS_ACC but acceleration == NULL", (uint32_t)i);
        if ((fl & F_HAS_ROTVEL) && !E->rotation_velocity) return c->fail(RE_E_ARG, "added entity %u has RE_F_HAS_ROTVEL but rotation_velocity == NULL", (uint32_t)i);
        if ((fl & F_HAS_ROTACC) && !E->rotation_acceleration) return c->fail(RE_E_ARG, "added entity %u has RE_F_HAS_ROTACC but rotation_acceleration == NULL", (uint32_t)i);
        ids[k] = E->entity_id[i]; fls[k] = fl;
        memcpy(&pos[(size_t)k * 3], E->position + i * 3, 12); memcpy(&orig[(size_t)k * 6], E->original_aabb + i * 6, 24);
        if (fl & F_HAS_ROT) { const float *a = E->rotation + i * 4; const float nn = norm3(a[0], a[1], a[2]); rot[k * 4 + 0] = a[0] / nn; rot[k * 4 + 1] = a[1] / nn; rot[k * 4 + 2] = a[2] / nn; rot[k * 4 + 3] = a[3]; }
        else { rot[k * 4 + 0] = 1.f; rot[k * 4 + 1] = rot[k * 4 + 2] = rot[k * 4 + 3] = 0.f; }
        if (fl & F_HAS_SCALE) memcpy(&scl[(size_t)k * 3], E->scale + i * 3, 12); else scl[k * 3 + 0] = scl[k * 3 + 1] = scl[k * 3 + 2] = 1.f;
        gcl[k] = group_class_of(c, GroupKey{ E->model_index[i], E->render_system ? E->render_system[i] : 0u, E->sortable ? E->sortable[i] : 0u }, &grew);
        if (fl & (F_HAS_VEL | F_HAS_ROTVEL)) {
            Dyn d{}; d.k = k; d.rv[0] = d.ra[0] = 1.f;
            for (int q = 0; q < 3; q++) { d.v[q] = (fl & F_HAS_VEL) ? E->velocity[i * 3 + q] : 0.f; d.a[q] = (fl & F_HAS_ACC) ? E->acceleration[i * 3 + q] : 0.f; }
            if (fl & F_HAS_ROTVEL) { const float *a = E->rotation_velocity + i * 4; const float nn = norm3(a[0], a[1], a[2]); d.rv[0] = a[0] / nn; d.rv[1] = a[1] / nn; d.rv[2] = a[2] / nn; d.rv[3] = a[3]; }
            if (fl & F_HAS_ROTACC) { const float *a = E->rotation_acceleration + i * 4; const float nn = norm3(a[0], a[1], a[2]); d.ra[0] = a[0] / nn; d.ra[1] = a[1] / nn; d.ra[2] = a[2] / nn; d.ra[3] = a[3]; }
            dyn.push_back(d);
        }
    }
    if (grew) { int rc = regrow_groups(c); if (rc != RE_OK) return rc; }
    HIPCHK(c, hipMemcpyAsync(c->d_id.p + row0, ids.data(), (size_t)m * 4, hipMemcpyHostToDevice, st)); HIPCHK(c, hipMemcpyAsync(c->d_gclass.p + row0, gcl.data(), (size_t)m * 4, hipMemcpyHostToDevice, st));
    HIPCHK(c, hipMemcpyAsync(c->d_flags.p + row0, fls.data(), (size_t)m * 4, hipMemcpyHostToDevice, st)); HIPCHK(c, hipMemcpyAsync(c->d_pos.p + (size_t)row0 * 3, pos.data(), (size_t)m * 12, hipMemcpyHostToDevice, st));
    HIPCHK(c, hipMemcpyAsync(c->d_rot.p + (size_t)row0 * 4, rot.data(), (size_t)m * 16, hipMemcpyHostToDevice, st)); HIPCHK(c, hipMemcpyAsync(c->d_scale.p + (size_t)row0 * 3, scl.data(), (size_t)m * 12, hipMemcpyHostToDevice, st));
    HIPCHK(c, hipMemcpyAsync(c->d_orig.p + row0, orig.data(), (size_t)m * 24, hipMemcpyHostToDevice, st));
    // TRS -> matrix, AABB, section decision: the upload's kernel over the new rows
    DevBuf<uint64_t> d_key; DevBuf<uint8_t> d_nk; DevBuf<SharedRec> d_sr; DevBuf<uint32_t> d_cnt;
    auto done = [&](int rc) { d_key.release(nullptr); d_nk.release(nullptr); d_sr.release(nullptr); d_cnt.release(nullptr); return rc; };
    if (d_key.alloc(m, nullptr) != hipSuccess || d_nk.alloc(m, nullptr) != hipSuccess || d_sr.alloc(m, nullptr) != hipSuccess || d_cnt.alloc(4, nullptr) != hipSuccess) return done(c->fail(RE_E_HIP, "create_rows: out of device memory"));
    (void)hipMemsetAsync(d_cnt.p, 0, 16, st);
    hipLaunchKernelGGL(k_transform_assign, dim3((m + 255) / 256), dim3(256), 0, st, row_arrays(c), row0, m, c->cfg.outline_length, c->cfg.atomic_length, d_key.p, d_nk.p, d_sr.p, d_cnt.p, m);
    (void)hipMemsetAsync(c->d_row_cell.p + row0, 0xFF, (size_t)m * 4, st);      // ROW_CELL_NONE: in no section yet (the device-side re-bucket reads it; rows beyond the upload were never written)
    std::vector<uint64_t> key(m); std::vector<uint8_t> nk(m); uint32_t nsr = 0;
    (void)hipMemcpyAsync(key.data(), d_key.p, (size_t)m * 8, hipMemcpyDeviceToHost, st); (void)hipMemcpyAsync(nk.data(), d_nk.p, m, hipMemcpyDeviceToHost, st); (void)hipMemcpyAsync(&nsr, d_cnt.p, 4, hipMemcpyDeviceToHost, st);
    if (hipGetLastError() != hipSuccess || sync_stream(st) != hipSuccess) return done(c->fail(RE_E_HIP, "create_rows: kernel / copy failed"));
    std::vector<SharedRec> sr(std::min(nsr, m));
    if (!sr.empty() && hipMemcpy(sr.data(), d_sr.p, sr.size() * sizeof(SharedRec), hipMemcpyDeviceToHost) != hipSuccess) return done(c->fail(RE_E_HIP, "create_rows: copy failed"));
    // the dynamic table
    for (const Dyn &d : dyn) {
        uint32_t j = 0; int rc = alloc_dyn_slot(c, row0 + d.k, &j); if (rc != RE_OK) return done(rc);
        if (hipMemcpy(c->d_dyn_vel.p + (size_t)j * 3, d.v, 12, hipMemcpyHostToDevice) != hipSuccess || hipMemcpy(c->d_dyn_acc.p + (size_t)j * 3, d.a, 12, hipMemcpyHostToDevice) != hipSuccess ||
            hipMemcpy(c->d_dyn_rotvel.p + (size_t)j * 4, d.rv, 16, hipMemcpyHostToDevice) != hipSuccess || hipMemcpy(c->d_dyn_rotacc.p + (size_t)j * 4, d.ra, 16, hipMemcpyHostToDevice) != hipSuccess) return done(c->fail(RE_E_HIP, "create_rows: copy failed"));
    }
    // host mirrors
    uint32_t rejected = 0;
    for (uint32_t k = 0; k < m; k++) {
        const uint32_t r = row0 + k;
        c->h_id.push_back(ids[k]); c->h_flags.push_back(fls[k] & ~F_STATIC); c->h_gclass.push_back(gcl[k]);      // (the static bit follows the tree replay)
        c->h_row_key.push_back(0); c->h_row_nk.push_back(0); c->h_row_cell.push_back(ROW_CELL_NONE);
        c->id_extra[ids[k]] = r;
        if (fls[k] & F_LIGHT_ANY) { c->h_light_rows.push_back(r); c->light_rows_dirty = true; }
        if (fls[k] & F_PHANTOM) c->n_phantom++;
        if ((fls[k] & F_USER) && c->user_row == ROW_CELL_NONE) c->user_row = r;
        if (fls[k] & F_HAS_ROTVEL) c->has_rotvel = true;
        re_ctx::AddKeys ak{}; ak.nk = nk[k]; ak.keys[0] = key[k]; ak.is_static = (fls[k] & F_STATIC) != 0;
        if (!nk[k]) rejected++;
        c->add_keys[r] = ak;
    }
    for (const SharedRec &q : sr) { auto it = c->add_keys.find(q.row); if (it != c->add_keys.end()) memcpy(it->second.keys, q.keys, sizeof q.keys); }
    c->n += m;
    if (n_rejected) *n_rejected = rejected;
    return done(RE_OK);
}

// ------------------------------------------------------------------------------------------------
// re_apply_changes == apply_change (helper_things/entity_change_helpers.rs:32-189) for the change requests user logic returns
// (LogicFunction / CollisionFunction -> Vec<EntityChangeInformation>): ModifyRequest of the kinematic components, DeleteRequest,
// MakeObjectStatic, WakeUpRequest.  The list is replayed on the host exactly as the reference does it (the three HashSets of
// :34-36, last write per component wins, deleted entities ignore later requests); the component values, the new matrices /
// AABBs / section decisions run on the GPU (k_write_components, k_apply_rows = the tail of the tick kernel), and the tree is
// patched by the same re-bucket as after a tick.
// ------------------------------------------------------------------------------------------------
// in_frame: the batch belongs to a frame's logic phase (apply_change); false: Pipeline::register_model_instances between frames (re_add_entities) -- the
// changed-static set then survives until the next render, which re-caches those sections.
static int apply_changes_impl(re_ctx *c, const re_change *changes, uint32_t n, const re_entities *added, re_tick_result *out, bool in_frame) {
    if (!c->h_res) return c->fail(RE_E_STATE, "re_apply_changes: no world uploaded");
    if (n && !changes) return c->fail(RE_E_ARG, "re_apply_changes: changes is NULL");
    if (in_frame && !c->have_cull) return c->fail(RE_E_STATE, "re_apply_changes: apply_change runs inside a frame, after its render (flows/pipeline.rs:212-271); call re_cull_pack first");
    if (added && added->n && (!added->entity_id || !added->model_index || !added->flags || !added->original_aabb || !added->position)) return c->fail(RE_E_ARG, "re_apply_changes: added entities miss a required array");
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t st = c->stream;
    if (c->tick_inflight) { int rc = finish_tick(c, nullptr); if (rc != RE_OK) return rc; }
    { int rc = resolve(c); if (rc != RE_OK) return rc; }
    // The host mirrors of the section table (behind the device after a device-side re-bucket) are fetched only by the parts of a batch that read them: ghosts of the frozen
    // cache, added entities, hidden / shown rows.  A batch that only writes components of non-static entities -- the per-frame re-insertion of the user entity -- does not
    // (h_flags, which every batch reads, is never behind: the device path moves non-static rows and leaves their flags alone).
    bool mirrors_fresh = false;
    auto need_mirrors = [&]() -> int { if (mirrors_fresh) return RE_OK; mirrors_fresh = true; return sync_mirrors(c); };
    std::set<uint32_t> kin, trans, deleted;                                   // rows; the mover list is put in the reference's order (ascending EntityId) by rebucket()
    std::map<std::pair<uint32_t, uint32_t>, std::array<float, 4>> writes;     // (row, component) -> last value
    std::map<uint32_t, std::pair<uint32_t, uint32_t>> flag_ops;               // row -> (and-mask, or-mask)
    std::vector<TreeOp> pre;
    auto flag_op = [&](uint32_t r, uint32_t clear, uint32_t set) {       // recorded only: nothing of the context changes before the whole list is validated
        auto &f = flag_ops.emplace(r, std::make_pair(0xFFFFFFFFu, 0u)).first->second;
        f.first &= ~clear; f.second = (f.second & ~clear) | set;
    };
    auto dyn_index = [&](uint32_t r, uint32_t &j) -> bool { return c->dyn_index(r, j); };
    auto normalized = [](const float *v) { std::array<float, 4> o; float nn = norm3(v[0], v[1], v[2]); o[0] = v[0] / nn; o[1] = v[1] / nn; o[2] = v[2] / nn; o[3] = v[3]; return o; };
    // The static render cache of the reference is a snapshot that the logic phase can never refresh (render_flow.rs:549-594 reads
    // changed_static_unique, which pipeline.rs:271 clears before the next render): a cached static entity that is woken, deleted,
    // moved or rewritten stays in the picture with its old bytes, and an entity made static later is never drawn.  Both are
    // modelled: the first as a ghost instance parked in the caching section, the second by hiding the row's group class.
    // entities this batch adds (RE_CHANGE_ADD_ENTITY): they take rows c->n, c->n + 1, ... when the list has been validated; until then the rows are
    // only planned, and everything below that looks at a row's flags goes through flags_of
    std::vector<NewRow> new_rows; std::vector<uint32_t> new_flags; std::unordered_map<uint32_t, uint32_t> new_ids;
    auto flags_of = [&](uint32_t r) -> uint32_t { return r < c->n ? c->h_flags[r] : new_flags[r - c->n]; };
    std::set<uint32_t> dyn_alloc;                                             // rows that need a slot of the dynamic table (a velocity-type component written to an entity registered without one)
    std::map<uint32_t, uint32_t> sortable_ops;                                // row -> sortable index (Add / RemoveSortableComponent, last one wins)
    std::map<uint32_t, bool> is_static;                                       // static bit as the batch evolves (the host mirror changes during the replay)
    auto stat = [&](uint32_t r) -> bool & { auto it = is_static.find(r); if (it == is_static.end()) it = is_static.emplace(r, (flags_of(r) & F_STATIC) != 0).first; return it->second; };
    std::set<uint32_t> hide, unhide; bool new_rotvel = false;
    auto uncached = [&](uint32_t r) { return (c->h_uncached.count(r) && !unhide.count(r)) || hide.count(r); };
    // First touch of a static entity that the frozen cache holds: plan a ghost instance (a copy of id + matrix as they are now,
    // parked behind the static rows of the section that cached it) and from then on treat the entity as one made static after the
    // freeze: hidden while it stays static.  The cache keeps drawing the copy whatever happens to the entity.
    std::vector<std::pair<uint32_t, uint64_t>> ghosts;                        // (row, key of the caching section)
    std::set<uint32_t> ghosted; std::set<uint64_t> shared_ghost_owners;      // sections that park a ghost of a shared section's static member
    std::vector<int32_t> sh_owner_h; std::vector<uint8_t> sh_cached_h, cell_flags_h;
    auto plan_ghost = [&](uint32_t r) -> int {
        if (r >= c->n) return RE_OK;                                            // an entity this batch added: no cache entry can hold it
        if (ghosted.count(r)) return RE_OK;
        { int rcm = need_mirrors(); if (rcm != RE_OK) return rcm; }
        ghosted.insert(r); hide.insert(r);
        if (c->h_row_nk[r] == 1) {                                              // cached by its own section -- unless that lay beyond the draw distance when the cache froze (an empty entry)
            if (cell_flags_h.empty() && c->ncells) { cell_flags_h.resize(c->ncells); HIPCHK(c, hipMemcpy(cell_flags_h.data(), c->d_cell_flags.p, c->ncells, hipMemcpyDeviceToHost)); }
            const uint32_t slot = c->h_row_cell[r];
            if (slot < cell_flags_h.size() && (cell_flags_h[slot] & CF_STATIC_CACHED)) ghosts.push_back({ r, c->h_row_key[r] });
            return RE_OK;
        }
        if (c->h_row_nk[r] > 1) {                                              // a shared section's static members are cached by one of its linking sections
            if (sh_owner_h.empty() && c->nsh) {
                sh_owner_h.resize(c->nsh); sh_cached_h.resize(c->nsh);
                HIPCHK(c, hipMemcpy(sh_owner_h.data(), c->d_sh_owner.p, (size_t)c->nsh * 4, hipMemcpyDeviceToHost)); HIPCHK(c, hipMemcpy(sh_cached_h.data(), c->d_sh_cached.p, c->nsh, hipMemcpyDeviceToHost));
            }
            const uint32_t s2 = c->h_row_cell[r] & ~ROW_CELL_SHARED;
            if (s2 < sh_owner_h.size() && sh_cached_h[s2] && sh_owner_h[s2] >= 0) { ghosts.push_back({ r, c->h_cell_key[sh_owner_h[s2]] }); shared_ghost_owners.insert(c->h_cell_key[sh_owner_h[s2]]); }
        }
        return RE_OK;
    };
    for (uint32_t i = 0; i < n; i++) {
        const re_change &ch = changes[i];
        uint32_t r = 0;
        if (ch.kind == RE_CHANGE_ADD_ENTITY) {
            // AddEntity (entity_change_helpers.rs:48-107): create_entity + apply_choices -- components, matrix, StaticAABB, add_entity(.., false, is_static) -- inline, in list order
            if (!added || ch.reserved >= added->n) return c->fail(RE_E_ARG, "re_apply_changes: change %u adds entity %u of %u", i, ch.reserved, added ? added->n : 0u);
            const uint32_t id = added->entity_id[ch.reserved];
            if (ch.entity_id != id) return c->fail(RE_E_ARG, "re_apply_changes: change %u names entity %u, the entity it adds has id %u", i, ch.entity_id, id);
            uint32_t r0 = 0;
            { auto ni = new_ids.find(id); if (ni != new_ids.end() && !deleted.count(ni->second)) return c->fail(RE_E_ARG, "re_apply_changes: entity id %u is in use (change %u)", id, i); }
            if (new_ids.find(id) == new_ids.end() && c->row_of(id, &r0) && !(c->h_flags[r0] & F_DEAD) && !deleted.count(r0)) return c->fail(RE_E_ARG, "re_apply_changes: entity id %u is in use (change %u)", id, i);
            r = c->n + (uint32_t)new_rows.size();
            uint32_t fl = added->flags[ch.reserved] & ~(F_HAS_MOVED | F_HAS_ROTATED | F_DEAD);
            if (fl & F_PHANTOM) fl &= ~(F_HAS_VEL | F_HAS_ACC | F_HAS_ROTVEL | F_HAS_ROTACC | F_ALWAYS_EXEC | F_USER | F_LIGHT_ANY);
            new_rows.push_back(NewRow{ ch.reserved, r }); new_flags.push_back(fl); new_ids[id] = r;
            pre.push_back({ r, 4 });
            is_static[r] = (fl & F_STATIC) != 0;
            if ((fl & F_STATIC) && in_frame) hide.insert(r);                   // static after the cache froze: in the tree's static set, drawn only once its section is re-cached
            if (fl & F_HAS_ROTVEL) new_rotvel = true;
            continue;
        }
        { auto nr = new_ids.find(ch.entity_id); if (nr != new_ids.end()) r = nr->second; else if (!c->row_of(ch.entity_id, &r)) return c->fail(RE_E_ARG, "re_apply_changes: unknown entity %u (change %u)", ch.entity_id, i); }
        if ((flags_of(r) & F_DEAD) || (deleted.count(r) && ch.kind != RE_CHANGE_MODIFY)) continue;   // removed earlier (out of bounds or deleted)
        switch (ch.kind) {
            case RE_CHANGE_MODIFY: {
                if (deleted.count(r)) break;                                   // :281-284
                bool pos = false, rot = false, scl = false; uint32_t j = 0;
                std::array<float, 4> v = { ch.value[0], ch.value[1], ch.value[2], ch.value[3] };
                switch (ch.component) {
                    case RE_C_POSITION: pos = true; break;
                    case RE_C_ROTATION: rot = true; v = normalized(ch.value); flag_op(r, 0, F_HAS_ROT); break;       // Rotation::new normalises the axis (movement_components.rs:108-118)
                    case RE_C_SCALE: scl = true; flag_op(r, 0, F_HAS_SCALE); break;
                    case RE_C_VELOCITY: case RE_C_ACCELERATION: case RE_C_ROTATION_VEL: case RE_C_ROTATION_ACC:
                        if (!dyn_index(r, j) && !(r >= c->n && (flags_of(r) & (F_HAS_VEL | F_HAS_ROTVEL)))) dyn_alloc.insert(r);   // registered on write (objects/entity_change_request.rs:29-30): a slot of the dynamic table
                        if (ch.component == RE_C_ROTATION_VEL || ch.component == RE_C_ROTATION_ACC) v = normalized(ch.value);
                        flag_op(r, 0, ch.component == RE_C_VELOCITY ? F_HAS_VEL : ch.component == RE_C_ACCELERATION ? F_HAS_ACC : ch.component == RE_C_ROTATION_VEL ? F_HAS_ROTVEL : F_HAS_ROTACC);
                        if (ch.component == RE_C_ROTATION_VEL) new_rotvel = true;
                        break;
                    default: return c->fail(RE_E_ARG, "re_apply_changes: component %u cannot be modified (change %u)", ch.component, i);
                }
                if ((pos || rot || scl) && stat(r) && !uncached(r)) { int rc = plan_ghost(r); if (rc != RE_OK) return rc; }
                writes[{ r, ch.component }] = v;
                if (pos && !rot && !scl) { if (!kin.count(r)) trans.insert(r); }
                else if (pos || rot || scl) { kin.insert(r); trans.erase(r); }
                break;
            }
            case RE_CHANGE_REMOVE_COMPONENT: {                                 // ecs.remove_component_type_id_internal: presence bit off, nothing recomputed
                if (deleted.count(r)) break;
                uint32_t j = 0;
                switch (ch.component) {
                    case RE_C_ROTATION: writes[{ r, (uint32_t)RE_C_ROTATION }] = { 1.f, 0.f, 0.f, 0.f }; flag_op(r, F_HAS_ROT, 0); break;       // later reads see Rotation::default
                    case RE_C_SCALE: writes[{ r, (uint32_t)RE_C_SCALE }] = { 1.f, 1.f, 1.f, 0.f }; flag_op(r, F_HAS_SCALE, 0); break;         // Scale::default
                    case RE_C_VELOCITY: case RE_C_ACCELERATION: case RE_C_ROTATION_VEL: case RE_C_ROTATION_ACC:
                        writes.erase({ r, ch.component });
                        if (dyn_index(r, j) || dyn_alloc.count(r) || r >= c->n) flag_op(r, ch.component == RE_C_VELOCITY ? F_HAS_VEL : ch.component == RE_C_ACCELERATION ? F_HAS_ACC : ch.component == RE_C_ROTATION_VEL ? F_HAS_ROTVEL : F_HAS_ROTACC, 0);
                        break;                                                  // (an entity without a slot in the dynamic table never carried it: no effect, like the reference)
                    default: return c->fail(RE_E_ARG, "re_apply_changes: component %u cannot be removed (change %u)", ch.component, i);
                }
                break;
            }
            case RE_CHANGE_DELETE:
                if (stat(r) && !uncached(r)) { int rc = plan_ghost(r); if (rc != RE_OK) return rc; }
                pre.push_back({ r, 3 }); kin.erase(r); trans.erase(r); deleted.insert(r);
                flag_op(r, 0, F_DEAD);
                break;
            case RE_CHANGE_MAKE_STATIC:
                pre.push_back({ r, 1 }); flag_op(r, 0, F_STATIC);
                if (!stat(r)) { stat(r) = true; if (unhide.count(r)) unhide.erase(r); else hide.insert(r); }
                break;
            case RE_CHANGE_WAKE_UP:
                if (stat(r) && !uncached(r)) { int rc = plan_ghost(r); if (rc != RE_OK) return rc; }
                pre.push_back({ r, 2 }); flag_op(r, F_STATIC, 0);
                if (stat(r)) { stat(r) = false; if (hide.count(r)) hide.erase(r); else unhide.insert(r); }
                break;
            case RE_CHANGE_ADD_SORTABLE: case RE_CHANGE_REMOVE_SORTABLE: {
                // AddSortableComponent / RemoveSortableComponent (entity_change_helpers.rs:138-146 -> ECS::write_sortable_component / remove_sortable_component,
                // objects/ecs.rs:202-213): the entity moves to another sortable bucket, i.e. to another (ModelId, sortable) group; a cached static entity stays in the
                // snapshot under its old bucket (a ghost), the live one is hidden like any entity that became static after the freeze
                if (deleted.count(r)) break;
                if (ch.kind == RE_CHANGE_ADD_SORTABLE && ch.component > 0xFFFFu) return c->fail(RE_E_ARG, "re_apply_changes: sortable index %u (change %u)", ch.component, i);
                if (stat(r) && !uncached(r)) { int rc = plan_ghost(r); if (rc != RE_OK) return rc; }
                sortable_ops[r] = ch.kind == RE_CHANGE_ADD_SORTABLE ? ch.component : 0u;
                break;
            }
            default: return c->fail(RE_E_ARG, "re_apply_changes: unknown change kind %u (change %u)", ch.kind, i);
        }
    }
    if (trans.size() + kin.size() > c->list_cap) return c->fail(RE_E_CAPACITY, "re_apply_changes: %zu moved entities exceed the mover list (%u); split the batch", trans.size() + kin.size(), c->list_cap);
    if (c->n_ghost + ghosts.size() > c->ghost_cap) return c->fail(RE_E_CAPACITY, "re_apply_changes: more than %u ghost instances of the frozen static cache", c->ghost_cap);
    // ---- the list is valid: from here on the context changes
    if (!new_rows.empty() || !pre.empty() || !hide.empty() || !unhide.empty() || !sortable_ops.empty() || !dyn_alloc.empty()) { int rcm = need_mirrors(); if (rcm != RE_OK) return rcm; }
    std::set<uint64_t> ghost_touched;
    if (!ghosts.empty()) {
        std::vector<Pair32> cl; cl.reserve(ghosts.size());
        for (auto &g : ghosts) { const uint32_t gr = c->ghost_base + c->n_ghost++; cl.push_back(Pair32{ g.first, gr }); c->h_ghost_gc.push_back(c->h_gclass[g.first]); c->ghost_map[g.second].push_back(gr); ghost_touched.insert(g.second); }
        Pair32 *d = nullptr; HIPCHK(c, hipMalloc(reinterpret_cast<void **>(&d), cl.size() * sizeof(Pair32)));
        HIPCHK(c, hipMemcpyAsync(d, cl.data(), cl.size() * sizeof(Pair32), hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(k_clone_rows, dim3(((uint32_t)cl.size() * 4u + 255) / 256), dim3(256), 0, st, (uint32_t)cl.size(), d, c->d_id.p, c->d_mat.p);   // before any component write of this batch
        HIPCHK(c, sync_stream(st)); (void)hipFree(d);
    }
    // ---- the entities this batch adds: rows, columns, matrices, section decisions (the tree sees them in rebucket(): TreeOp kind 4, in list order)
    if (!new_rows.empty()) { int rc = create_rows(c, added, new_rows, nullptr); if (rc != RE_OK) return rc; }
    for (uint32_t r : dyn_alloc) { uint32_t j = 0; int rc = alloc_dyn_slot(c, r, &j); if (rc != RE_OK) return rc; }
    std::vector<uint32_t> regrouped;                                          // rows whose group class changed: device column + their entry of the row pool
    if (!sortable_ops.empty()) {
        bool grew = false;
        for (auto &kv : sortable_ops) {
            const GroupKey old = c->h_gkeys[c->h_gclass[kv.first]];
            const uint32_t g = group_class_of(c, GroupKey{ old.model, old.rs, kv.second }, &grew);
            if (g != c->h_gclass[kv.first]) { c->h_gclass[kv.first] = g; regrouped.push_back(kv.first); }
        }
        if (grew) { int rc = regrow_groups(c); if (rc != RE_OK) return rc; }
    }
    if (!in_frame) {
        // Between frames the changed-static set is still there at the next render (nothing clears it before: pipeline.rs:271 runs inside execute), so the
        // sections that received a static entity are RE-CACHED then, from their live static sets: ghost instances parked there go, rows hidden there show again.
        for (const NewRow &nr : new_rows) {
            auto ak = c->add_keys.find(nr.row);
            if (ak == c->add_keys.end() || !ak->second.is_static) continue;
            for (uint32_t k = 0; k < ak->second.nk; k++) {
                const uint64_t K = ak->second.keys[k];
                if (c->ghost_map.erase(K)) ghost_touched.insert(K);
                c->dormant_cached.erase(K);
                const int32_t sl = find_slot(c, K);
                if (sl < 0) continue;
                for (uint32_t q = c->h_cell_nl[sl], e = c->h_cell_nl[sl] + c->h_cell_ns[sl], b = c->h_cell_begin[sl]; q < e; q++) { const uint32_t r2 = c->h_rows[b + q]; if (r2 < c->n && c->h_uncached.count(r2)) unhide.insert(r2); }
            }
        }
    }
    if (new_rotvel) c->has_rotvel = true;
    for (auto &kv : flag_ops) c->h_flags[kv.first] = (c->h_flags[kv.first] & (kv.second.first | F_STATIC)) | (kv.second.second & ~F_STATIC);   // the static bit follows the tree replay
    for (uint32_t r : hide) c->h_uncached.insert(r);
    for (uint32_t r : unhide) c->h_uncached.erase(r);
    for (uint32_t r : deleted) c->h_uncached.erase(r);
    std::vector<WriteOp> ops; ops.reserve(writes.size() + flag_ops.size() + hide.size() + unhide.size());
    for (auto &kv : writes) {
        WriteOp w{}; w.comp = kv.first.second; w.index = kv.first.first;
        if (w.comp >= RE_C_VELOCITY && w.comp <= RE_C_ROTATION_ACC) { uint32_t j = 0; dyn_index(kv.first.first, j); w.index = j; }
        memcpy(w.v, kv.second.data(), 16); ops.push_back(w);
    }
    for (uint32_t r : hide) { WriteOp w{}; w.comp = WRITE_GCLASS; w.index = r; w.v[0] = 0xFFFFFFFFu; ops.push_back(w); }
    for (uint32_t r : unhide) if (!deleted.count(r)) { WriteOp w{}; w.comp = WRITE_GCLASS; w.index = r; w.v[0] = c->h_gclass[r]; ops.push_back(w); }
    for (uint32_t r : regrouped) if (!deleted.count(r) && !c->h_uncached.count(r)) { WriteOp w{}; w.comp = WRITE_GCLASS; w.index = r; w.v[0] = c->h_gclass[r]; ops.push_back(w); }
    for (auto &kv : flag_ops) { WriteOp w{}; w.comp = WRITE_FLAGS; w.index = kv.first; w.v[0] = kv.second.first; w.v[1] = kv.second.second; w.v[2] = (kv.second.second & F_DEAD) ? 1u : 0u; ops.push_back(w); }
    std::vector<uint32_t> list; list.reserve(trans.size() + kin.size());
    for (uint32_t r : trans) list.push_back(r | 0x80000000u);
    for (uint32_t r : kin) list.push_back(r);
    TickHeader th{};
    static const bool no_small = getenv("RE_EXP_NO_APPLY_SMALL") != nullptr;      // A/B switch of tools/change_cost.py
    if ((!ops.empty() || !list.empty()) && ops.size() <= APPLY_SMALL_MAX && list.size() <= APPLY_SMALL_MAX && c->h_th && !no_small) {
        // the common small batch: ONE launch that reads its input from a mapped staging block and publishes the counters (k_apply_small)
        if (!c->h_chg) {
            void *hp = nullptr, *dp = nullptr;
            HIPCHK(c, hipHostMalloc(&hp, APPLY_SMALL_MAX * (sizeof(WriteOp) + 4u), hipHostMallocMapped));
            HIPCHK(c, hipHostGetDevicePointer(&dp, hp, 0));
            c->h_chg = static_cast<uint8_t *>(hp); c->d_chg = static_cast<uint8_t *>(dp);
        }
        if (!ops.empty()) memcpy(c->h_chg, ops.data(), ops.size() * sizeof(WriteOp));
        if (!list.empty()) memcpy(c->h_chg + APPLY_SMALL_MAX * sizeof(WriteOp), list.data(), list.size() * 4);
        std::atomic_thread_fence(std::memory_order_release);
        const uint32_t seq = ++c->tick_seq;
        hipLaunchKernelGGL(k_apply_small, dim3(1), dim3(256), 0, st, (uint32_t)ops.size(), reinterpret_cast<const WriteOp *>(c->d_chg), (uint32_t)list.size(),
                           reinterpret_cast<const uint32_t *>(c->d_chg + APPLY_SMALL_MAX * sizeof(WriteOp)), row_arrays(c), c->d_dyn_vel.p, c->d_dyn_acc.p, c->d_dyn_rotvel.p, c->d_dyn_rotacc.p,
                           c->d_row_cell.p, c->d_cell_key.p, c->d_sh_cells.p, c->cfg.outline_length, c->cfg.atomic_length, c->d_th.p, c->d_movers.p, c->d_oob.p, c->list_cap, c->d_hth, seq);
        HIPCHK(c, hipGetLastError());
        const volatile TickHeader *t = c->h_th; bool ok = false;
        const auto t0 = std::chrono::steady_clock::now();
        for (uint32_t spins = 0; !(ok = (t->ticket == seq)); spins++)
            if ((spins & 1023u) == 1023u && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(5)) break;
        if (ok) {
            std::atomic_thread_fence(std::memory_order_acquire);
            th.n_changed = t->n_changed; th.n_rebucket = t->n_rebucket; th.n_oob = t->n_oob;
            ok = t->pad[0] == (table_word_hash(th.n_changed, 1u) ^ table_word_hash(th.n_rebucket, 2u) ^ table_word_hash(th.n_oob, 3u) ^ table_word_hash(seq, 4u));
        }
        if (!ok) {                                                              // (a slow box, or a block that did not agree with its seal at first sight: the stream's end is authoritative)
            HIPCHK(c, sync_stream(st));
            HIPCHK(c, hipMemcpy(&th, c->d_th.p, sizeof th, hipMemcpyDeviceToHost));
        }
        c->th_clean = false;
    } else if (!ops.empty() || !list.empty()) {
        DevBuf<WriteOp> &d_ops = c->d_chg_ops; DevBuf<uint32_t> &d_list = c->d_chg_list;        // kept across calls
        if (d_ops.n < ops.size()) HIPCHK(c, d_ops.alloc(ops.size() * 2, nullptr));
        if (d_list.n < list.size()) HIPCHK(c, d_list.alloc(list.size() * 2, nullptr));
        HIPCHK(c, hipMemsetAsync(c->d_th.p, 0, sizeof(TickHeader), st));
        if (!ops.empty()) {
            HIPCHK(c, hipMemcpyAsync(d_ops.p, ops.data(), ops.size() * sizeof(WriteOp), hipMemcpyHostToDevice, st));
            hipLaunchKernelGGL(k_write_components, dim3(((uint32_t)ops.size() + 255) / 256), dim3(256), 0, st, (uint32_t)ops.size(), d_ops.p, row_arrays(c),
                               c->d_dyn_vel.p, c->d_dyn_acc.p, c->d_dyn_rotvel.p, c->d_dyn_rotacc.p);
        }
        if (!list.empty()) {
            HIPCHK(c, hipMemcpyAsync(d_list.p, list.data(), list.size() * 4, hipMemcpyHostToDevice, st));
            hipLaunchKernelGGL(k_apply_rows, dim3(((uint32_t)list.size() + 255) / 256), dim3(256), 0, st, (uint32_t)list.size(), d_list.p, row_arrays(c), c->d_row_cell.p,
                               c->d_cell_key.p, c->d_sh_cells.p, c->cfg.outline_length, c->cfg.atomic_length, c->d_th.p, c->d_movers.p, c->d_oob.p, c->list_cap);
        }
        HIPCHK(c, hipGetLastError());
        HIPCHK(c, hipMemcpyAsync(&th, c->d_th.p, sizeof th, hipMemcpyDeviceToHost, st));
        HIPCHK(c, sync_stream(st));
        c->th_clean = false;
    }
    if (th.n_oob) { int rc = absorb_out_of_bounds(c, th.n_oob); if (rc != RE_OK) return rc; }
    c->n_dead += th.n_oob + (uint32_t)deleted.size();
    c->last_tick = re_tick_result{ th.n_changed, th.n_rebucket, th.n_oob };
    // (Before the tree replay: a section emptied by this very batch must carry the flag into its dormant entry.)
    // A ghost of a shared section's static member is drawn whenever the section that cached it is -- also when that section's own
    // static entities were NOT cached (it lay beyond the draw distance when the cache froze: an empty entry plus the shared members).
    // The ghost sits behind the section's static rows and is emitted with them, so such a section is flagged as cached and its own
    // static rows are hidden instead: the same picture, and a later wake-up of one of them shows it again like any hidden row.
    std::vector<uint32_t> newly_hidden;
    for (uint64_t K : shared_ghost_owners) {
        const int32_t os = find_slot(c, K);
        if (os < 0) continue;                                                  // (cannot happen: the owner links the shared section)
        uint8_t f = 0; HIPCHK(c, hipMemcpy(&f, c->d_cell_flags.p + os, 1, hipMemcpyDeviceToHost));
        if (f & CF_STATIC_CACHED) continue;
        f |= CF_STATIC_CACHED; HIPCHK(c, hipMemcpy(c->d_cell_flags.p + os, &f, 1, hipMemcpyHostToDevice));
        for (uint32_t i = c->h_cell_nl[os], e = c->h_cell_nl[os] + c->h_cell_ns[os], b = c->h_cell_begin[os]; i < e; i++) {
            const uint32_t r = c->h_rows[b + i];
            if (c->h_uncached.insert(r).second) { newly_hidden.push_back(r); const uint32_t none = 0xFFFFFFFFu; HIPCHK(c, hipMemcpy(c->d_gclass.p + r, &none, 4, hipMemcpyHostToDevice)); }
        }
    }
    if (th.n_rebucket || !pre.empty() || !ghost_touched.empty()) {
        int rc = rebucket(c, th.n_rebucket, &pre, &ghost_touched);
        if (rc != RE_OK) return rc;
    }
    {   // rows hidden / shown by this batch that kept their place in the row pool
        std::vector<Pair32> gc;
        for (uint32_t r : newly_hidden) collect_row_gc(c, r, gc);
        for (uint32_t r : hide) collect_row_gc(c, r, gc);
        for (uint32_t r : unhide) if (!deleted.count(r)) collect_row_gc(c, r, gc);
        for (uint32_t r : regrouped) if (!deleted.count(r)) collect_row_gc(c, r, gc);
        int rc = upload_row_gc(c, gc);
        if (rc != RE_OK) return rc;
    }
    c->last_added_rejected = 0; for (auto &kv : c->add_keys) if (!kv.second.nk) c->last_added_rejected++;      // (out of bounds: created, not inserted -- re_add_entities reports the count)
    c->add_keys.clear();
    if (in_frame && c->dirty_pending) {                                         // Pipeline::execute: clear_changed_static_unique after the logic flow (pipeline.rs:271)
        uint32_t m = std::max(c->ncells, c->nsh);
        if (m) hipLaunchKernelGGL(k_clear_static_dirty, dim3((m + 255) / 256), dim3(256), 0, st, c->ncells, c->d_cell_flags.p, c->nsh, c->d_sh_dirty.p);
        c->dirty_pending = false;
    }
    if (out) *out = c->last_tick;
    return RE_OK;
}
extern "C" int re_apply_changes(re_ctx *c, const re_change *changes, uint32_t n, uint32_t flags, re_tick_result *out) try {
    (void)flags;
    if (!c) return RE_E_ARG;
    return apply_changes_impl(c, changes, n, nullptr, out, true);
} RE_ABI_GUARD(c, "re_apply_changes")
extern "C" int re_apply_changes_ex(re_ctx *c, const re_change *changes, uint32_t n, const re_entities *added, uint32_t flags, re_tick_result *out) try {
    (void)flags;
    if (!c) return RE_E_ARG;
    return apply_changes_impl(c, changes, n, added, out, true);
} RE_ABI_GUARD(c, "re_apply_changes_ex")
// Pipeline::register_model_instances at any time (flows/pipeline.rs:186-208): create_entity + apply_choices per instance, then end_of_changes.
extern "C" int re_add_entities(re_ctx *c, const re_entities *E, uint32_t *n_rejected) try {
    if (!c) return RE_E_ARG;
    if (!E) return c->fail(RE_E_ARG, "re_add_entities: entities is NULL");
    if (!c->h_res) return re_upload_entities(c, E, n_rejected);               // the first registration of a context
    if (n_rejected) *n_rejected = 0;
    if (!E->n) return RE_OK;
    if (!E->entity_id) return c->fail(RE_E_ARG, "re_add_entities: missing required array");
    std::vector<re_change> ch(E->n);
    for (uint32_t i = 0; i < E->n; i++) { ch[i] = re_change{}; ch[i].kind = RE_CHANGE_ADD_ENTITY; ch[i].entity_id = E->entity_id[i]; ch[i].reserved = i; }
    const uint32_t row0 = c->n;
    int rc = apply_changes_impl(c, ch.data(), E->n, E, nullptr, false);
    if (rc != RE_OK) return rc;
    (void)row0;
    if (n_rejected) *n_rejected = c->last_added_rejected;                      // (not from the host mirrors: the section inserts may have run on the device, whose mirrors follow on demand)
    return RE_OK;
} RE_ABI_GUARD(c, "re_add_entities")

// ------------------------------------------------------------------------------------------------
// re_collide == LogicFlow::handle_collisions (flows/logic_flow.rs:452-651): the broad phase and the AABB tests; the collision
// logic of the entity types (CollisionFunction callbacks) stays with the caller, which gets the argument pairs.
// ------------------------------------------------------------------------------------------------
constexpr uint32_t COL_REGION_CAP = 1u << 16, COL_SHARED_CAP = 1u << 14;
extern "C" int re_collide(re_ctx *c, uint32_t flags, re_collision *pairs, uint32_t capacity, uint32_t *n_total) try {
    (void)flags;
    if (!c) return RE_E_ARG;
    if (!c->h_res) return c->fail(RE_E_STATE, "re_collide: no world uploaded");
    if (!c->have_cull) return c->fail(RE_E_STATE, "re_collide: the collision pass works on the visibility query of the frame; call re_cull_pack first");
    if (capacity && !pairs) return c->fail(RE_E_ARG, "re_collide: capacity without a buffer");
    HIPCHK(c, hipSetDevice(c->device));
    { int rc = c->cull_inflight ? finish_cull(c, nullptr) : resolve(c); if (rc != RE_OK) return rc; }
    if (c->tick_inflight) { int rc = finish_tick(c, nullptr); if (rc != RE_OK) return rc; }
    { int rc = sync_mirrors(c); if (rc != RE_OK) return rc; }
    hipStream_t st = c->stream;
    if (!c->d_col_hdr.p) {
        HIPCHK(c, c->d_col_hdr.alloc(1, nullptr)); HIPCHK(c, c->d_col_region.alloc(COL_REGION_CAP, nullptr)); HIPCHK(c, c->d_col_high.alloc(COL_REGION_CAP, nullptr));
        HIPCHK(c, c->d_col_shared.alloc(COL_SHARED_CAP, nullptr)); HIPCHK(c, c->d_col_near.alloc(COL_REGION_CAP, nullptr));
        HIPCHK(c, hipHostMalloc(reinterpret_cast<void **>(&c->h_col), sizeof(ColHeader), hipHostMallocMapped | hipHostMallocCoherent)); memset(c->h_col, 0, sizeof(ColHeader));
        HIPCHK(c, hipHostGetDevicePointer(reinterpret_cast<void **>(&c->d_hcol), c->h_col, 0));
    }
    if (!c->col_moved_cap) {
        c->col_moved_cap = (uint32_t)std::min<uint64_t>(((uint64_t)c->ndyn + 1u) * 8u, 1u << 20);
        c->col_tab_size = 64; while (c->col_tab_size < 2u * c->col_moved_cap) c->col_tab_size <<= 1;
        HIPCHK(c, c->d_col_moved.alloc(c->col_moved_cap, nullptr)); HIPCHK(c, c->d_col_tab.alloc((size_t)c->col_tab_size * 2, nullptr)); HIPCHK(c, c->d_row_moved.alloc(std::max(c->n, 1u), nullptr));
        // cleared once; every call leaves them clean again (k_col_clear)
        HIPCHK(c, hipMemsetAsync(c->d_row_moved.p, 0, std::max(c->n, 1u), c->stream)); HIPCHK(c, hipMemsetAsync(c->d_col_tab.p, 0xFF, (size_t)c->col_tab_size * 16, c->stream));
    }
    const uint32_t want = std::max(capacity, 1u << 12);
    if (want > c->col_pair_cap) { c->d_col_pairs.release(nullptr); HIPCHK(c, c->d_col_pairs.alloc(want, nullptr)); c->col_pair_cap = want; }
    HIPCHK(c, hipMemsetAsync(c->d_col_hdr.p, 0, sizeof(ColHeader), st));
    unsigned long long *tab_key = c->d_col_tab.p, *tab_min = c->d_col_tab.p + c->col_tab_size;
    const uint32_t user_cell = c->user_row != ROW_CELL_NONE ? c->h_row_cell[c->user_row] : ROW_CELL_NONE;
    if (c->ncells && c->key32) hipLaunchKernelGGL(k_col_region<true>, dim3(((c->ncells + 3) / 4 + 255) / 256), dim3(256), 0, st, c->ncells, (const void *)c->d_cell_key32.p, (const uint32_t *)c->d_chunk_level.p,
                                                  c->d_cell_tight.p, c->d_params.p, c->cfg.atomic_length, c->d_col_hdr.p, c->d_col_region.p, COL_REGION_CAP, c->d_col_high.p, COL_REGION_CAP);
    else if (c->ncells) hipLaunchKernelGGL(k_col_region<false>, dim3((c->ncells + 255) / 256), dim3(256), 0, st, c->ncells, (const void *)c->d_cell_key.p, (const uint32_t *)nullptr,
                                           c->d_cell_tight.p, c->d_params.p, c->cfg.atomic_length, c->d_col_hdr.p, c->d_col_region.p, COL_REGION_CAP, c->d_col_high.p, COL_REGION_CAP);
    if (c->nsh) hipLaunchKernelGGL(k_col_shared, dim3((c->nsh + 255) / 256), dim3(256), 0, st, c->nsh, c->d_sh_aabb.p, c->d_sh_cells.p, c->d_sh_nact.p, c->d_sh_nstat.p, c->d_cell_key.p,
                                   c->d_params.p, c->d_col_hdr.p, c->d_col_shared.p, COL_SHARED_CAP);
    hipLaunchKernelGGL(k_col_tops, dim3(COL_REGION_CAP / 256), dim3(256), 0, st, c->d_col_hdr.p, c->d_col_region.p, COL_REGION_CAP, c->d_col_high.p, COL_REGION_CAP, c->d_col_shared.p, COL_SHARED_CAP,
                       c->d_cell_begin.p, c->d_cell_nlocal.p, c->d_col_near.p, COL_REGION_CAP);
    hipLaunchKernelGGL(k_col_moved, dim3((c->ndyn + 1 + 255) / 256), dim3(256), 0, st, c->ndyn, c->d_dyn_row.p, c->d_row_cell.p /* dynamic rows are the first ndyn rows */, c->user_row, user_cell, row_arrays(c), c->d_cell_key.p,
                       c->d_cell_stamp.p, c->d_cell_flags.p, c->d_sh_cells.p, c->d_sh_aabb.p, c->d_params.p, c->d_col_hdr.p, c->d_col_moved.p, c->col_moved_cap, c->d_row_moved.p,
                       tab_key, tab_min, c->col_tab_size - 1u);
    hipLaunchKernelGGL(k_col_pairs, dim3(256), dim3(256), 0, st, c->d_col_hdr.p, c->d_col_moved.p, c->col_moved_cap, c->d_col_near.p, COL_REGION_CAP, c->d_col_shared.p, COL_SHARED_CAP,
                       row_arrays(c), c->d_sh_begin.p, c->d_sh_nact.p, c->d_rows.p, c->d_row_moved.p, tab_min, c->d_col_pairs.p, c->col_pair_cap);
    hipLaunchKernelGGL(k_col_clear, dim3((c->col_moved_cap + 255) / 256), dim3(256), 0, st, c->d_col_hdr.p, c->d_col_moved.p, c->col_moved_cap, c->d_row_moved.p, tab_key, tab_min,
                       c->d_hcol, ++c->col_calls);
    HIPCHK(c, hipGetLastError());
    ColHeader h{};
    {   // completion: poll the counts the last kernel publishes in mapped host memory (a stream synchronise costs tens of microseconds)
        const volatile uint32_t *flag = &c->h_col->pad[1];
        const auto t0 = std::chrono::steady_clock::now(); bool done = false;
        for (uint32_t spins = 0; !(done = (*flag == c->col_calls)); spins++)
            if ((spins & 1023u) == 1023u && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(5)) break;
        if (!done) HIPCHK(c, sync_stream(st));
        std::atomic_thread_fence(std::memory_order_acquire);
        auto sealed = [&]() { const volatile ColHeader *q = c->h_col; return q->pad[0] == (table_word_hash(q->n_region, 1u) ^ table_word_hash(q->n_high, 2u) ^ table_word_hash(q->n_shared, 3u) ^ table_word_hash(q->n_moved, 4u) ^ table_word_hash(q->n_pairs, 5u) ^ table_word_hash(q->n_near, 6u) ^ table_word_hash(c->col_calls, 7u)); };
        if (!sealed()) {                                                      // (as in finish_cull: counted, asserted to be 0 by the tests)
            c->n_seal_waits++;
            const auto t1 = std::chrono::steady_clock::now();
            while (!sealed() && std::chrono::steady_clock::now() - t1 < std::chrono::microseconds(500)) {}
            if (!sealed()) { c->n_sync_fallbacks++; HIPCHK(c, sync_stream(st)); std::atomic_thread_fence(std::memory_order_acquire); }
        }
        h = *c->h_col;
    }
    if (h.n_region > COL_REGION_CAP || h.n_high > COL_REGION_CAP) return c->fail(RE_E_CAPACITY, "re_collide: %u world sections around the camera exceed the region list (%u)", h.n_region, COL_REGION_CAP);
    if (h.n_shared > COL_SHARED_CAP) return c->fail(RE_E_CAPACITY, "re_collide: %u shared sections within the collision distance exceed the list (%u)", h.n_shared, COL_SHARED_CAP);
    if (h.n_moved > c->col_moved_cap) {                                     // entries beyond the list were not cleaned up by k_col_clear
        HIPCHK(c, hipMemset(c->d_row_moved.p, 0, std::max(c->n, 1u))); HIPCHK(c, hipMemset(c->d_col_tab.p, 0xFF, (size_t)c->col_tab_size * 16));
    }
    if (h.n_moved > c->col_moved_cap) return c->fail(RE_E_CAPACITY, "re_collide: %u (section, moved entity) entries exceed the list (%u)", h.n_moved, c->col_moved_cap);
    const uint32_t nw = std::min(std::min(h.n_pairs, capacity), c->col_pair_cap);
    if (nw) HIPCHK(c, hipMemcpy(pairs, c->d_col_pairs.p, (size_t)nw * sizeof(re_collision), hipMemcpyDeviceToHost));
    if (n_total) *n_total = h.n_pairs;
    return RE_OK;
} RE_ABI_GUARD(c, "re_collide")

// ------------------------------------------------------------------------------------------------
// Multi-GPU exchange behind the C ABI (BASELINE configs[3]): one process per GPU, sections sharded by contiguous key range, and
// ONE exchange step per frame -- the all-gather of every GPU's packed visible-instance slab over RCCL / xGMI.  RCCL is loaded at run time
// (a single-GPU host does not need it); the communicator is either created here from a unique id the host distributes over its own
// channel (re_comm_unique_id / re_comm_init) or adopted from the host (re_comm_adopt).
// ------------------------------------------------------------------------------------------------
#include <dlfcn.h>
namespace rccl {
typedef struct ncclComm *ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef int ncclResult_t;                                    // ncclSuccess == 0
enum { Int32 = 2, Uint32 = 3, Float32 = 7 };                 // ncclDataType_t (rccl.h)
static ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
static ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
static ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
static ncclResult_t (*AllGather)(const void *, void *, size_t, int, ncclComm_t, hipStream_t) = nullptr;
static const char *(*GetErrorString)(ncclResult_t) = nullptr;
static std::string load_error;
static bool load() {
    if (AllGather) return true;
    void *h = nullptr;
    const char *env = getenv("RE_RCCL_LIBRARY");
    if (env) h = dlopen(env, RTLD_NOW | RTLD_LOCAL);
    // The RCCL that belongs to the HIP runtime THIS library is bound to: the one installed next to it.  A process may hold a second ROCm stack -- a
    // PyTorch wheel brings its own libamdhip64.so / libhsa-runtime64.so / librccl.so under other sonames --, and a communicator created by that stack's
    // RCCL cannot take this library's streams and buffers (its HSA runtime may not even be initialised: ncclCommInitRank then fails with "no
    // ROCm-capable device is detected").  Round 2 looked for an already loaded librccl first and found exactly that copy.
    if (!h) {
        Dl_info di{};
        if (dladdr(reinterpret_cast<const void *>(&hipGetDeviceCount), &di) && di.dli_fname) {
            std::string dir(di.dli_fname); const size_t sl = dir.rfind('/');
            if (sl != std::string::npos) {
                dir.resize(sl + 1);
                for (const char *name : { "librccl.so.1", "librccl.so" }) if (!h) h = dlopen((dir + name).c_str(), RTLD_NOW | RTLD_LOCAL);
            }
        }
    }
    if (!h) h = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_LOCAL);
    if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
    if (!h) { load_error = std::string("RCCL could not be loaded: ") + (dlerror() ? dlerror() : "?"); return false; }
    GetUniqueId = reinterpret_cast<decltype(GetUniqueId)>(dlsym(h, "ncclGetUniqueId"));
    CommInitRank = reinterpret_cast<decltype(CommInitRank)>(dlsym(h, "ncclCommInitRank"));
    CommDestroy = reinterpret_cast<decltype(CommDestroy)>(dlsym(h, "ncclCommDestroy"));
    GetErrorString = reinterpret_cast<decltype(GetErrorString)>(dlsym(h, "ncclGetErrorString"));
    auto ag = reinterpret_cast<decltype(AllGather)>(dlsym(h, "ncclAllGather"));
    if (!GetUniqueId || !CommInitRank || !CommDestroy || !GetErrorString || !ag) { load_error = "RCCL: missing symbols"; return false; }
    AllGather = ag;
    return true;
}
}  // namespace rccl
#define NCCLCHK(ctx, call) do { rccl::ncclResult_t r_ = (call); if (r_ != 0) return (ctx)->fail(RE_E_HIP, "%s failed: %s", #call, rccl::GetErrorString ? rccl::GetErrorString(r_) : "?"); } while (0)

static void comm_release(re_ctx *c) {
    re_ctx::Comm &m = c->comm;
    if (m.comm && m.owned && rccl::CommDestroy) (void)rccl::CommDestroy(reinterpret_cast<rccl::ncclComm_t>(m.comm));
    for (int b = 0; b < 2; b++) { m.slab[b].release(&c->dev_bytes); m.recv[b].release(&c->dev_bytes); }
    m.big_ids.release(&c->dev_bytes); m.big_mats.release(&c->dev_bytes);
    if (m.h_hdr) (void)hipHostFree(m.h_hdr);
    m = re_ctx::Comm{};
    c->ext_out_ids = nullptr; c->ext_out_mats = nullptr; c->ext_out_cap = 0; c->ext_out_count = nullptr;
}
static int comm_setup(re_ctx *c, void *comm, bool owned, int rank, int n_ranks, uint32_t slab_instances) {
    re_ctx::Comm &m = c->comm;
    m.comm = comm; m.owned = owned; m.rank = rank; m.n = n_ranks; m.cap = std::max(slab_instances, 1u);
    m.words = 16u + m.cap * 17u;
    for (int b = 0; b < 2; b++) {
        HIPCHK(c, m.slab[b].alloc(m.words, &c->dev_bytes)); HIPCHK(c, m.recv[b].alloc((size_t)m.words * n_ranks, &c->dev_bytes));
        HIPCHK(c, hipMemset(m.slab[b].p, 0, (size_t)m.words * 4)); HIPCHK(c, hipMemset(m.recv[b].p, 0, (size_t)m.words * n_ranks * 4));
    }
    HIPCHK(c, hipHostMalloc(reinterpret_cast<void **>(&m.h_hdr), ((size_t)n_ranks * 4 + 32) * 4, hipHostMallocMapped | hipHostMallocCoherent));
    memset(m.h_hdr, 0, ((size_t)n_ranks * 4 + 32) * 4);
    HIPCHK(c, hipHostGetDevicePointer(reinterpret_cast<void **>(&m.d_hhdr), m.h_hdr, 0));
    m.counts.assign(n_ranks, 0u); m.seq = 0; m.last = -1; m.pending = -1; m.hdr_seq = 0;
    return RE_OK;
}
extern "C" int re_comm_unique_id(uint8_t *id) try {
    if (!id) return RE_E_ARG;
    if (!rccl::load()) { g_create_error = rccl::load_error; return RE_E_UNSUPPORTED; }
    rccl::ncclUniqueId u; memset(&u, 0, sizeof u);
    rccl::ncclResult_t r = rccl::GetUniqueId(&u);
    if (r != 0) { g_create_error = std::string("ncclGetUniqueId: ") + rccl::GetErrorString(r); return RE_E_HIP; }
    memcpy(id, u.internal, RE_COMM_ID_BYTES);
    return RE_OK;
} RE_ABI_GUARD_NOCTX(g_create_error, "re_comm_unique_id")
extern "C" int re_comm_init(re_ctx *c, const uint8_t *id, int rank, int n_ranks, uint32_t slab_instances) try {
    if (!c || !id || n_ranks < 1 || rank < 0 || rank >= n_ranks) return c ? c->fail(RE_E_ARG, "re_comm_init: bad arguments") : RE_E_ARG;
    if (!rccl::load()) return c->fail(RE_E_UNSUPPORTED, "%s", rccl::load_error.c_str());
    HIPCHK(c, hipSetDevice(c->device));
    if (c->comm.comm) comm_release(c);
    rccl::ncclUniqueId u; memcpy(u.internal, id, RE_COMM_ID_BYTES);
    rccl::ncclComm_t comm = nullptr;
    NCCLCHK(c, rccl::CommInitRank(&comm, n_ranks, u, rank));
    return comm_setup(c, comm, true, rank, n_ranks, slab_instances);
} RE_ABI_GUARD(c, "re_comm_init")
extern "C" int re_comm_adopt(re_ctx *c, void *nccl_comm, int rank, int n_ranks, uint32_t slab_instances) try {
    if (!c || !nccl_comm || n_ranks < 1 || rank < 0 || rank >= n_ranks) return c ? c->fail(RE_E_ARG, "re_comm_adopt: bad arguments") : RE_E_ARG;
    if (!rccl::load()) return c->fail(RE_E_UNSUPPORTED, "%s", rccl::load_error.c_str());
    HIPCHK(c, hipSetDevice(c->device));
    if (c->comm.comm) comm_release(c);
    return comm_setup(c, nccl_comm, false, rank, n_ranks, slab_instances);
} RE_ABI_GUARD(c, "re_comm_adopt")
extern "C" int re_comm_destroy(re_ctx *c) try {
    if (!c) return RE_E_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    if (c->stream) HIPCHK(c, sync_stream(c->stream));
    comm_release(c);
    return RE_OK;
} RE_ABI_GUARD(c, "re_comm_destroy")

// The last frame's pack once more, into other output buffers (everything it reads -- instance list, cursors, group counts -- is still there until the
// next visibility query): the second round of the exchange needs a rank's FULL packed set, not the slab's truncated prefix.
static int repack_last_frame(re_ctx *c, uint32_t *ids, float *mats, uint32_t cap) {
    hipStream_t st = c->stream;
    re_ctx::LastPack &P = c->last_pack;
    if (P.kind == 1) {
        PackArgs A = P.A; A.out_ids = ids; A.out_mats = mats; A.out_cap = cap; A.out_count = nullptr;
        hipLaunchKernelGGL(k_pack_small, dim3(P.grid), dim3(256), (size_t)std::max(c->nslots, 1u) * 8, st, P.hdr, P.hdr_next, c->d_th.p, A, P.K, P.nrows);
    } else if (P.kind == 2) {
        PackLargeArgs A = P.L; A.out_ids = ids; A.out_mats = mats; A.out_cap = cap; A.out_count = nullptr; A.zero_a = nullptr; A.zero_b = nullptr; A.zero_words = 0;
        HIPCHK(c, hipMemsetAsync(A.gfill, 0, (size_t)CURSOR_SHARDS * std::max(c->nslots, 1u) * 4, st));      // the fills of the first run
        hipLaunchKernelGGL(k_pack_large, dim3(P.grid), dim3(PACK_LARGE_THREADS), 0, st, A);
    } else if (P.kind == 3) {
        uint32_t *ki = c->ext_out_ids; float *km = c->ext_out_mats; uint32_t kc = c->ext_out_cap, *kn = c->ext_out_count;
        c->ext_out_ids = ids; c->ext_out_mats = mats; c->ext_out_cap = cap; c->ext_out_count = nullptr;
        int rc = launch_pack_large(c, P.hdr, P.hdr_next);                       // count / scan / scatter leave their scratch clean: they can simply run again
        c->ext_out_ids = ki; c->ext_out_mats = km; c->ext_out_cap = kc; c->ext_out_count = kn;
        if (rc != RE_OK) return rc;
    } else return c->fail(RE_E_STATE, "repack: no frame has been packed yet");
    HIPCHK(c, hipGetLastError());
    return RE_OK;
}

static int gather_enqueue(re_ctx *c, int b) {
    re_ctx::Comm &m = c->comm;
    NCCLCHK(c, rccl::AllGather(m.slab[b].p, m.recv[b].p, m.words, rccl::Int32, reinterpret_cast<rccl::ncclComm_t>(m.comm), c->stream));
    // the gathered headers follow the collective into mapped host memory (no stream synchronise, no per-rank copy)
    hipLaunchKernelGGL(k_gather_headers, dim3(1), dim3(64), 0, c->stream, (uint32_t)m.n, (const uint32_t *)m.recv[b].p, m.words, m.d_hhdr, m.d_hhdr + (size_t)m.n * 4, ++m.hdr_seq);
    HIPCHK(c, hipGetLastError());
    return RE_OK;
}
static int gather_finish(re_ctx *c, re_gathered *out) {
    re_ctx::Comm &m = c->comm;
    const int b = m.pending >= 0 ? m.pending : m.last;
    if (b < 0) return c->fail(RE_E_STATE, "re_allgather_visible: no frame has been packed since re_comm_init");
    m.pending = -1;
    auto read_headers = [&]() -> int {                                       // poll the sequence word k_gather_headers publishes behind the collective
        const volatile uint32_t *flag = m.h_hdr + (size_t)m.n * 4;
        const auto t0 = std::chrono::steady_clock::now();
        bool done = false;
        for (uint32_t spins = 0; !(done = (*flag == m.hdr_seq)); spins++)
            if ((spins & 1023u) == 1023u && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(5)) break;   // a long collective: let the driver wait
        if (!done) HIPCHK(c, sync_stream(c->stream));
        std::atomic_thread_fence(std::memory_order_acquire);
        return RE_OK;
    };
    { int rc = read_headers(); if (rc != RE_OK) return rc; }
    // A frame cancelled by cross-frame speculation left a marker in its slab header instead of a count (every rank reads the same headers, so every
    // rank takes this branch together): settle the speculation here -- the replay packs into the same slab -- and gather again.
    for (int round = 0; round < 4; round++) {
        bool cancelled = false;
        for (int r = 0; r < m.n; r++) cancelled |= m.h_hdr[(size_t)r * 4] == 0xFFFFFFFFu;
        if (!cancelled) break;
        { int rc = c->cull_inflight ? finish_cull(c, nullptr) : resolve(c); if (rc != RE_OK) return rc; }
        m.n_regathers++;
        { int rc = gather_enqueue(c, b); if (rc != RE_OK) return rc; }
        { int rc = read_headers(); if (rc != RE_OK) return rc; }
        if (round == 3) return c->fail(RE_E_STATE, "re_allgather_visible: a rank's frame stayed cancelled");
    }
    uint32_t max_total = 0; bool overflow = false;
    for (int r = 0; r < m.n; r++) { const uint32_t total = m.h_hdr[(size_t)r * 4 + 1]; m.counts[r] = total; max_total = std::max(max_total, total); overflow |= total > m.cap; }
    if (out) { out->n_ranks = (uint32_t)m.n; out->counts = m.counts.data(); out->overflowed = overflow ? 1u : 0u; }
    if (!overflow) {
        if (out) {
            out->d_entity_ids = m.recv[b].p + 16; out->ids_rank_stride = m.words;
            out->d_matrices = reinterpret_cast<const float *>(m.recv[b].p + 16 + m.cap); out->matrices_rank_stride = m.words;
        }
        return RE_OK;
    }
    // Second, variable-length round (rare: a slab is sized for the expected visible set of a rank): every rank packs its frame again, untruncated,
    // into its full-size output buffer, and the buffers are gathered padded to the largest count.
    if (max_total > c->out_cap) return c->fail(RE_E_CAPACITY, "re_allgather_visible: a rank packed %u instances, more than this rank's output buffer holds (%u)", max_total, c->out_cap);
    { int rc = repack_last_frame(c, c->d_out_ids.p, c->d_out_mats.p, c->out_cap); if (rc != RE_OK) return rc; }
    if (m.big_cap < max_total) {
        m.big_ids.release(&c->dev_bytes); m.big_mats.release(&c->dev_bytes);
        m.big_cap = max_total + max_total / 4u;
        HIPCHK(c, m.big_ids.alloc((size_t)m.big_cap * m.n, &c->dev_bytes)); HIPCHK(c, m.big_mats.alloc((size_t)m.big_cap * m.n * 16, &c->dev_bytes));
    }
    NCCLCHK(c, rccl::AllGather(c->d_out_ids.p, m.big_ids.p, max_total, rccl::Uint32, reinterpret_cast<rccl::ncclComm_t>(m.comm), c->stream));
    NCCLCHK(c, rccl::AllGather(c->d_out_mats.p, m.big_mats.p, (size_t)max_total * 16, rccl::Float32, reinterpret_cast<rccl::ncclComm_t>(m.comm), c->stream));
    HIPCHK(c, sync_stream(c->stream));
    m.n_second_rounds++;
    if (out) { out->d_entity_ids = m.big_ids.p; out->ids_rank_stride = max_total; out->d_matrices = m.big_mats.p; out->matrices_rank_stride = max_total * 16u; }
    return RE_OK;
}
extern "C" int re_allgather_visible(re_ctx *c, uint32_t flags, re_gathered *out) try {
    if (!c) return RE_E_ARG;
    if (!c->comm.comm) return c->fail(RE_E_STATE, "re_allgather_visible: no communicator (re_comm_init / re_comm_adopt)");
    if (c->comm.last < 0) return c->fail(RE_E_STATE, "re_allgather_visible: call re_cull_pack first");
    HIPCHK(c, hipSetDevice(c->device));
    if (c->comm.pending >= 0) { int rc = gather_finish(c, nullptr); if (rc != RE_OK) return rc; }
    // synchronous: the frame is settled first (a cancelled frame is replayed into the same slab), so the slab that goes out is the final one
    if (!(flags & RE_GATHER_ASYNC) && c->cull_inflight) { int rc = finish_cull(c, nullptr); if (rc != RE_OK) return rc; }
    { int rc = gather_enqueue(c, c->comm.last); if (rc != RE_OK) return rc; }       // on the context's stream: ordered behind the pack that fills the slab
    c->comm.pending = c->comm.last;
    if (flags & RE_GATHER_ASYNC) return RE_OK;
    return gather_finish(c, out);
} RE_ABI_GUARD(c, "re_allgather_visible")
extern "C" int re_gather_wait(re_ctx *c, re_gathered *out) try {
    if (!c) return RE_E_ARG;
    if (!c->comm.comm) return c->fail(RE_E_STATE, "re_gather_wait: no communicator");
    HIPCHK(c, hipSetDevice(c->device));
    return gather_finish(c, out);
} RE_ABI_GUARD(c, "re_gather_wait")

extern "C" int re_set_shard_range(re_ctx *c, uint64_t key_lo, uint64_t key_hi) try {
    if (!c) return RE_E_ARG;
    if (key_lo > key_hi) return c->fail(RE_E_ARG, "re_set_shard_range: key_lo > key_hi");
    c->shard_lo = key_lo; c->shard_hi = key_hi; c->moved_rows.clear();
    return RE_OK;
} RE_ABI_GUARD(c, "re_set_shard_range")
extern "C" int re_list_migrants(re_ctx *c, uint32_t *ids, uint32_t capacity, uint32_t *n) try {
    if (!c || !n || (capacity && !ids)) return RE_E_ARG;
    *n = 0;
    if (!c->h_res) return c->fail(RE_E_STATE, "re_list_migrants: no world uploaded");
    HIPCHK(c, hipSetDevice(c->device));
    if (c->tick_inflight) { int rc = finish_tick(c, nullptr); if (rc != RE_OK) return rc; }
    { int rc = resolve(c); if (rc != RE_OK) return rc; }
    { int rc = sync_mirrors(c); if (rc != RE_OK) return rc; }
    if (!c->shard_hi) { c->moved_rows.clear(); return RE_OK; }
    std::sort(c->moved_rows.begin(), c->moved_rows.end(), [](uint32_t a, uint32_t b) { return (a & 0x7FFFFFFFu) < (b & 0x7FFFFFFFu); });
    uint32_t cnt = 0, prev = 0xFFFFFFFFu;
    for (uint32_t w : c->moved_rows) {
        const uint32_t r = w & 0x7FFFFFFFu;                                  // (bit 31 of a mover word: translation-only)
        if (r == prev || r >= c->n) continue;
        prev = r;
        if ((c->h_flags[r] & (F_DEAD | F_PHANTOM)) || !c->h_row_nk[r]) continue;
        uint64_t first = c->h_row_key[r];
        if (c->h_row_nk[r] > 1) { const auto &ks = c->h_row_shared_keys[r]; first = ks[0]; for (uint32_t k = 1; k < c->h_row_nk[r]; k++) first = std::min(first, ks[k]); }
        if (first >= c->shard_lo && first < c->shard_hi) continue;
        if (cnt < capacity) ids[cnt] = c->h_id[r];
        cnt++;
    }
    *n = cnt;
    if (cnt <= capacity) c->moved_rows.clear();                              // (a list that did not fit is kept for a second call with room)
    return RE_OK;
} RE_ABI_GUARD(c, "re_list_migrants")
extern "C" int re_export_entities(re_ctx *c, const uint32_t *ids, uint32_t n, re_entity_state *out) try {
    if (!c || (n && (!ids || !out))) return RE_E_ARG;
    if (!c->h_res) return c->fail(RE_E_STATE, "re_export_entities: no world uploaded");
    if (!n) return RE_OK;
    static_assert(sizeof(re_entity_state) == sizeof(ExportRec) && sizeof(ExportRec) == 140, "re_entity_state layout");
    HIPCHK(c, hipSetDevice(c->device));
    if (c->tick_inflight) { int rc = finish_tick(c, nullptr); if (rc != RE_OK) return rc; }
    { int rc = resolve(c); if (rc != RE_OK) return rc; }
    std::vector<uint32_t> rows(n), slots(n);
    for (uint32_t i = 0; i < n; i++) {
        if (!c->row_of(ids[i], &rows[i]) || (c->h_flags[rows[i]] & F_DEAD)) return c->fail(RE_E_ARG, "re_export_entities: unknown entity %u", ids[i]);
        uint32_t j = 0; slots[i] = c->dyn_index(rows[i], j) ? j : 0xFFFFFFFFu;
    }
    DevBuf<uint32_t> d_rows, d_slots; DevBuf<ExportRec> d_out;
    auto done = [&](int rc) { d_rows.release(nullptr); d_slots.release(nullptr); d_out.release(nullptr); return rc; };
    if (d_rows.alloc(n, nullptr) != hipSuccess || d_slots.alloc(n, nullptr) != hipSuccess || d_out.alloc(n, nullptr) != hipSuccess) return done(c->fail(RE_E_HIP, "re_export_entities: out of device memory"));
    hipStream_t st = c->stream;
    (void)hipMemcpyAsync(d_rows.p, rows.data(), (size_t)n * 4, hipMemcpyHostToDevice, st); (void)hipMemcpyAsync(d_slots.p, slots.data(), (size_t)n * 4, hipMemcpyHostToDevice, st);
    hipLaunchKernelGGL(k_export_rows, dim3((n + 255) / 256), dim3(256), 0, st, n, d_rows.p, d_slots.p, row_arrays(c), c->d_dyn_vel.p, c->d_dyn_acc.p, c->d_dyn_rotvel.p, c->d_dyn_rotacc.p, d_out.p);
    (void)hipMemcpyAsync(out, d_out.p, (size_t)n * sizeof(ExportRec), hipMemcpyDeviceToHost, st);
    if (hipGetLastError() != hipSuccess || sync_stream(st) != hipSuccess) return done(c->fail(RE_E_HIP, "re_export_entities: kernel / copy failed"));
    for (uint32_t i = 0; i < n; i++) {                                       // what the device rows do not hold: the group's ModelId / sortable index, and the tree's static bit
        const GroupKey &g = c->h_gkeys[c->h_gclass[rows[i]]];
        out[i].model_index = g.model; out[i].render_system = g.rs; out[i].sortable = g.sort;
        out[i].flags = (out[i].flags & ~(F_STATIC | F_DEAD)) | (c->h_flags[rows[i]] & F_STATIC);
    }
    return done(RE_OK);
} RE_ABI_GUARD(c, "re_export_entities")

extern "C" int re_wait(re_ctx *c, re_visible *out_visible, re_tick_result *out_tick) try {
    if (!c) return RE_E_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    int rc = RE_OK;
    if (c->cull_inflight) rc = finish_cull(c, out_visible); else { rc = resolve(c); if (rc == RE_OK && out_visible && c->h_res) fill_visible(c, out_visible); }
    if (rc != RE_OK) return rc;
    if (c->tick_inflight) rc = finish_tick(c, out_tick); else if (out_tick) *out_tick = c->last_tick;
    return rc;
} RE_ABI_GUARD(c, "re_wait")

// The frame loop of the reference (threads/render_thread.rs:217-250 -> Pipeline::execute, flows/pipeline.rs:212-276) driven from native code,
// for measurement without an interpreter between the calls: exactly the two public entry points, n times.
extern "C" int re_run_frames(re_ctx *c, const re_camera *cam, float dt, uint32_t cull_flags, uint32_t tick_flags, uint32_t n, float *wall_us, re_visible *last_visible, re_tick_result *last_tick) try {
    if (!c) return RE_E_ARG;
    if (!cam) return c->fail(RE_E_ARG, "re_run_frames: camera is NULL");
    re_visible vis{}; re_tick_result tr{};
    for (uint32_t f = 0; f < n; f++) {
        const auto t0 = std::chrono::steady_clock::now();
        int rc = re_cull_pack(c, cam, cull_flags, &vis); if (rc != RE_OK) return rc;
        if (c->comm.comm) { rc = re_allgather_visible(c, (cull_flags & RE_CULL_ASYNC) ? RE_GATHER_ASYNC : 0u, nullptr); if (rc != RE_OK) return rc; }   // the frame's one exchange step
        rc = re_tick(c, dt, tick_flags, &tr); if (rc != RE_OK) return rc;
        if (wall_us) wall_us[f] = std::chrono::duration<float, std::micro>(std::chrono::steady_clock::now() - t0).count();
    }
    if (last_visible && !(cull_flags & RE_CULL_ASYNC)) *last_visible = vis;
    if (last_tick && !(tick_flags & RE_TICK_ASYNC)) *last_tick = tr;
    return RE_OK;
} RE_ABI_GUARD(c, "re_run_frames")

extern "C" int re_copy_visible(re_ctx *c, uint32_t *ids_host, float *mats_host, uint32_t capacity, uint32_t *n_written) try {
    if (!c || !c->h_res) return RE_E_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    if (c->cull_inflight) { int rc = finish_cull(c, nullptr); if (rc) return rc; }
    uint32_t cap = c->ext_out_ids ? c->ext_out_cap : c->out_cap;
    uint32_t nw = std::min(std::min(c->h_res->total, cap), capacity);      // truncate and report (mapped_buffer.rs:171-186)
    const uint32_t *src_ids = c->ext_out_ids ? c->ext_out_ids : c->d_out_ids.p; const float *src_m = c->ext_out_mats ? c->ext_out_mats : c->d_out_mats.p;
    if (nw && ids_host) HIPCHK(c, hipMemcpyAsync(ids_host, src_ids, (size_t)nw * 4, hipMemcpyDeviceToHost, c->stream));
    if (nw && mats_host) HIPCHK(c, hipMemcpyAsync(mats_host, src_m, (size_t)nw * 64, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, sync_stream(c->stream));
    if (n_written) *n_written = nw;
    return RE_OK;
} RE_ABI_GUARD(c, "re_copy_visible")

extern "C" int re_set_output_buffers(re_ctx *c, uint32_t *d_ids, float *d_mats, uint32_t capacity) try {
    if (!c) return RE_E_ARG;
    if ((d_ids == nullptr) != (d_mats == nullptr)) return c->fail(RE_E_ARG, "re_set_output_buffers: both pointers or neither");
    c->ext_out_ids = d_ids; c->ext_out_mats = d_mats; c->ext_out_cap = d_ids ? capacity : 0;
    return RE_OK;
} RE_ABI_GUARD(c, "re_set_output_buffers")

extern "C" int re_set_output_count(re_ctx *c, uint32_t *d_count) try {
    if (!c) return RE_E_ARG;
    c->ext_out_count = d_count;
    return RE_OK;
} RE_ABI_GUARD(c, "re_set_output_count")

extern "C" int re_read_component(re_ctx *c, uint32_t entity_id, int component, void *dst) try {
    if (!c || !dst) return RE_E_ARG;
    uint32_t r = 0;
    if (!c->row_of(entity_id, &r)) return c->fail(RE_E_ARG, "re_read_component: unknown entity %u", entity_id);
    HIPCHK(c, hipSetDevice(c->device));
    { int rc_ = resolve(c); if (rc_ != RE_OK) return rc_; }
    auto dynidx = [&](uint32_t &j) -> bool { return c->dyn_index(r, j); };
    uint32_t j = 0;
    {   // ECS::get_copy -> None for a component the entity does not carry (objects/ecs.rs:653-664, check_component_written :348-367)
        static const uint32_t need[] = { 0, F_HAS_ROT, F_HAS_SCALE, F_HAS_VEL, F_HAS_ACC, F_HAS_ROTVEL, F_HAS_ROTACC };
        if (component >= RE_C_ROTATION && component <= RE_C_ROTATION_ACC && !(c->h_flags[r] & need[component]))
            return c->fail(RE_E_ARG, "re_read_component: entity %u does not carry component %d", entity_id, component);
        if ((c->h_flags[r] & F_DEAD) && component != RE_C_FLAGS) return c->fail(RE_E_ARG, "re_read_component: entity %u was removed", entity_id);
    }
    switch (component) {
        case RE_C_POSITION: HIPCHK(c, hipMemcpy(dst, c->d_pos.p + (size_t)r * 3, 12, hipMemcpyDeviceToHost)); break;
        case RE_C_ROTATION: HIPCHK(c, hipMemcpy(dst, c->d_rot.p + (size_t)r * 4, 16, hipMemcpyDeviceToHost)); break;
        case RE_C_SCALE: HIPCHK(c, hipMemcpy(dst, c->d_scale.p + (size_t)r * 3, 12, hipMemcpyDeviceToHost)); break;
        case RE_C_TRANSFORMATION: HIPCHK(c, hipMemcpy(dst, c->d_mat.p + (size_t)r * 16, 64, hipMemcpyDeviceToHost)); break;
        case RE_C_STATIC_AABB: HIPCHK(c, hipMemcpy(dst, c->d_aabb.p + r, 24, hipMemcpyDeviceToHost)); break;
        case RE_C_ORIGINAL_AABB: HIPCHK(c, hipMemcpy(dst, c->d_orig.p + r, 24, hipMemcpyDeviceToHost)); break;
        case RE_C_FLAGS: HIPCHK(c, hipMemcpy(dst, c->d_flags.p + r, 4, hipMemcpyDeviceToHost)); break;
        case RE_C_VELOCITY: if (!dynidx(j)) return c->fail(RE_E_ARG, "entity %u has no Velocity", entity_id); HIPCHK(c, hipMemcpy(dst, c->d_dyn_vel.p + (size_t)j * 3, 12, hipMemcpyDeviceToHost)); break;
        case RE_C_ACCELERATION: if (!dynidx(j)) return c->fail(RE_E_ARG, "entity %u has no Acceleration", entity_id); HIPCHK(c, hipMemcpy(dst, c->d_dyn_acc.p + (size_t)j * 3, 12, hipMemcpyDeviceToHost)); break;
        case RE_C_ROTATION_VEL: if (!dynidx(j)) return c->fail(RE_E_ARG, "entity %u has no VelocityRotation", entity_id); HIPCHK(c, hipMemcpy(dst, c->d_dyn_rotvel.p + (size_t)j * 4, 16, hipMemcpyDeviceToHost)); break;
        case RE_C_ROTATION_ACC: if (!dynidx(j)) return c->fail(RE_E_ARG, "entity %u has no AccelerationRotation", entity_id); HIPCHK(c, hipMemcpy(dst, c->d_dyn_rotacc.p + (size_t)j * 4, 16, hipMemcpyDeviceToHost)); break;
        default: return c->fail(RE_E_ARG, "re_read_component: unknown component %d", component);
    }
    return RE_OK;
} RE_ABI_GUARD(c, "re_read_component")

// ---- ECS presence semantics (objects/ecs.rs): see include/re_hip.h
static uint32_t ecs_bits_of_flags(uint32_t fl) {
    if (fl & F_DEAD) return 0u;                                               // remove_entity clears every component (ecs.rs:557-600)
    uint32_t b = (1u << RE_ECS_BIT_POSITION) | (1u << RE_ECS_BIT_TRANSFORMATION) | (1u << RE_ECS_BIT_MODEL_ID) | (1u << RE_ECS_BIT_STATIC_AABB) | (1u << RE_ECS_BIT_ORIGINAL_AABB);
    if (fl & F_CAN_COLLIDE) b |= 1u << RE_ECS_BIT_CAN_CAUSE_COLLISIONS;
    if (fl & F_HAS_MOVED) b |= 1u << RE_ECS_BIT_HAS_MOVED;
    if (fl & F_HAS_VEL) b |= 1u << RE_ECS_BIT_VELOCITY;
    if (fl & F_HAS_ACC) b |= 1u << RE_ECS_BIT_ACCELERATION;
    if (fl & F_HAS_ROTATED) b |= 1u << RE_ECS_BIT_HAS_ROTATED;
    if (fl & F_HAS_ROT) b |= 1u << RE_ECS_BIT_ROTATION;
    if (fl & F_HAS_ROTVEL) b |= 1u << RE_ECS_BIT_VELOCITY_ROTATION;
    if (fl & F_HAS_ROTACC) b |= 1u << RE_ECS_BIT_ACCELERATION_ROTATION;
    if (fl & F_HAS_SCALE) b |= 1u << RE_ECS_BIT_SCALE;
    if (fl & F_ALWAYS_EXEC) b |= 1u << RE_ECS_BIT_ALWAYS_EXECUTE_LOGIC;
    return b;
}
extern "C" int re_ecs_bitset(re_ctx *c, uint32_t entity_id, uint32_t *bits) try {
    if (!c || !bits) return RE_E_ARG;
    uint32_t r = 0;
    if (!c->row_of(entity_id, &r)) return c->fail(RE_E_ARG, "re_ecs_bitset: unknown entity %u", entity_id);
    HIPCHK(c, hipSetDevice(c->device));
    { int rc_ = resolve(c); if (rc_ != RE_OK) return rc_; }
    uint32_t fl = 0;
    HIPCHK(c, hipMemcpy(&fl, c->d_flags.p + r, 4, hipMemcpyDeviceToHost));     // the device column is the truth (HasMoved / HasRotated are maintained by the tick)
    *bits = ecs_bits_of_flags(fl);
    return RE_OK;
} RE_ABI_GUARD(c, "re_ecs_bitset")
// The world sections an entity is registered in -- one key (its unique section) or the 2..8 keys its shared section links --, for a loader that spreads
// a world over several GPUs.  The owner of an entity is the shard whose key range holds the SMALLEST of its keys: all
// entities of a unique section, and a shared section together with the unique section that caches its static entities, then live on one shard (a
// section's tight AABB -- distance test, LOD -- folds all of its entities).  The other keys tell which entities a shard needs as halo replicas
// (RE_F_PHANTOM).  Host arithmetic with the functions the upload kernel runs (re_math.h); no device needed.
extern "C" int re_section_keys(const re_config *cfg, const re_entities *E, uint64_t *keys, uint8_t *n_keys) try {
    if (!cfg || !E || (E->n && (!keys || !n_keys || !E->flags || !E->original_aabb || !E->position))) return RE_E_ARG;
    if (!cfg->outline_length || !cfg->atomic_length) return RE_E_ARG;
    for (uint32_t i = 0; i < E->n; i++) {
        const uint32_t fl = E->flags[i];
        const float *p = E->position + (size_t)i * 3;
        float axis[3] = { 1.f, 0.f, 0.f }, angle = 0.f, scl[3] = { 1.f, 1.f, 1.f }, m[16];
        if ((fl & F_HAS_ROT) && E->rotation) { const float *a = E->rotation + (size_t)i * 4; const float nn = norm3(a[0], a[1], a[2]); axis[0] = a[0] / nn; axis[1] = a[1] / nn; axis[2] = a[2] / nn; angle = a[3]; }   // Rotation::new normalises the axis (as re_upload_entities does)
        if ((fl & F_HAS_SCALE) && E->scale) { const float *q = E->scale + (size_t)i * 3; scl[0] = q[0]; scl[1] = q[1]; scl[2] = q[2]; }
        Aabb orig; memcpy(&orig, E->original_aabb + (size_t)i * 6, sizeof orig);
        Aabb a;
        if (fl & F_USER) { a = orig; a.xmin += p[0]; a.xmax += p[0]; a.ymin += p[1]; a.ymax += p[1]; a.zmin += p[2]; a.zmax += p[2]; }
        else { trs_matrix(p, (fl & F_HAS_ROT) != 0, axis, angle, (fl & F_HAS_SCALE) != 0, scl, m); a = apply_transformation(orig, m); }
        Aabb bv = a;
        const bool oob = normalize_aabb(&bv, (float)cfg->outline_length);
        uint64_t k8[8];
        int nk = oob ? 0 : assign_sections(bv, cfg->atomic_length, k8);
        if (nk < 0) nk = 0;
        n_keys[i] = (uint8_t)nk;
        for (int k = 0; k < 8; k++) keys[(size_t)i * 8 + k] = k < nk ? k8[k] : 0ull;
    }
    return RE_OK;
} RE_ABI_GUARD_NOCTX(g_create_error, "re_section_keys")

// The lights of one type RenderFlow::render finds near the camera (flows/render_flow.rs:249-254 -> flows/shadow_flow.rs:455-513): k_visible_lights over the
// entities uploaded with RE_F_LIGHT_*.  Stands alone (own visibility test with the AABB culler of radius far_draw); ids in ascending order.
extern "C" int re_visible_lights(re_ctx *c, const re_camera *cam, uint32_t light_type, uint32_t *ids, uint32_t capacity, uint32_t *n_out) try {
    if (!c) return RE_E_ARG;
    if (!cam || !n_out || (capacity && !ids)) return c->fail(RE_E_ARG, "re_visible_lights: NULL argument");
    if (light_type != RE_F_LIGHT_DIRECTIONAL && light_type != RE_F_LIGHT_POINT && light_type != RE_F_LIGHT_SPOT) return c->fail(RE_E_ARG, "re_visible_lights: light_type must be one RE_F_LIGHT_* bit");
    if (!c->h_res) return c->fail(RE_E_STATE, "re_visible_lights: no world uploaded");
    HIPCHK(c, hipSetDevice(c->device));
    { int rc_ = resolve(c); if (rc_ != RE_OK) return rc_; }                     // the section table is settled (movers of the last tick are in)
    *n_out = 0;
    const uint32_t nl = (uint32_t)c->h_light_rows.size();
    if (!nl) return RE_OK;
    if (c->light_rows_dirty) { HIPCHK(c, c->d_light_rows.alloc(nl, nullptr)); HIPCHK(c, hipMemcpy(c->d_light_rows.p, c->h_light_rows.data(), (size_t)nl * 4, hipMemcpyHostToDevice)); c->light_rows_dirty = false; }
    if (c->d_light_out.n < (size_t)nl + 1) HIPCHK(c, c->d_light_out.alloc((size_t)nl + 1, nullptr));
    LightQuery Q{}; const float r = cam->far_draw, wsl = (float)c->cfg.atomic_length;
    Q.culler = Aabb{ cam->position[0] - r, cam->position[0] + r, cam->position[1] - r, cam->position[1] + r, cam->position[2] - r, cam->position[2] + r };
    fill_level_boxes(Q.box, c->maxlevel, wsl, rmax(cam->position[0] - r, 0.0f), cam->position[0] + r, rmax(cam->position[1] - r, 0.0f), cam->position[1] + r,
                     rmax(cam->position[2] - r, 0.0f), cam->position[2] + r);   // generate_original_culling_aabb (visible_world_flow.rs:131-145)
    Q.max_level = c->maxlevel; Q.type_flag = light_type;
    HIPCHK(c, hipMemsetAsync(c->d_light_out.p + nl, 0, 4, c->stream));
    hipLaunchKernelGGL(k_visible_lights, dim3((nl + 255) / 256), dim3(256), 0, c->stream, nl, c->d_light_rows.p, c->d_flags.p, c->d_id.p, c->d_row_cell.p, c->d_cell_key.p, c->d_cell_flags.p,
                       c->d_sh_cells.p, Q, c->d_light_out.p, nl, c->d_light_out.p + nl);
    HIPCHK(c, hipGetLastError());
    std::vector<uint32_t> out((size_t)nl + 1);
    HIPCHK(c, hipMemcpyAsync(out.data(), c->d_light_out.p, ((size_t)nl + 1) * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, sync_stream(c->stream));
    const uint32_t cnt = std::min(out[nl], nl);
    std::sort(out.begin(), out.begin() + cnt);
    *n_out = cnt;
    for (uint32_t i = 0; i < cnt && i < capacity; i++) ids[i] = out[i];
    return RE_OK;
} RE_ABI_GUARD(c, "re_visible_lights")

extern "C" int re_ecs_query(re_ctx *c, const int *components, uint32_t n_components, uint32_t *ids, uint32_t capacity, uint32_t *n_out) try {
    if (!c || (n_components && !components) || (capacity && !ids)) return RE_E_ARG;
    if (!c->h_res) return c->fail(RE_E_STATE, "re_ecs_query: no world uploaded");
    uint32_t need = 0;
    for (uint32_t k = 0; k < n_components; k++) {
        switch (components[k]) {
            case RE_C_POSITION: case RE_C_TRANSFORMATION: case RE_C_STATIC_AABB: case RE_C_ORIGINAL_AABB: break;     // written for every registered entity
            case RE_C_ROTATION: need |= F_HAS_ROT; break; case RE_C_SCALE: need |= F_HAS_SCALE; break;
            case RE_C_VELOCITY: need |= F_HAS_VEL; break; case RE_C_ACCELERATION: need |= F_HAS_ACC; break;
            case RE_C_ROTATION_VEL: need |= F_HAS_ROTVEL; break; case RE_C_ROTATION_ACC: need |= F_HAS_ROTACC; break;
            default: return c->fail(RE_E_ARG, "re_ecs_query: component %d was never registered", components[k]);   // the reference panics (get_indexes_for_components unwraps)
        }
    }
    HIPCHK(c, hipSetDevice(c->device));
    { int rc_ = resolve(c); if (rc_ != RE_OK) return rc_; }
    const uint32_t cap = std::max(c->n, 1u);
    DevBuf<uint32_t> d_out, d_cnt; HIPCHK(c, d_out.alloc(cap, nullptr)); HIPCHK(c, d_cnt.alloc(1, nullptr));
    HIPCHK(c, hipMemsetAsync(d_cnt.p, 0, 4, c->stream));
    if (c->n) hipLaunchKernelGGL(k_query_flags, dim3((c->n + 255) / 256), dim3(256), 0, c->stream, c->n, c->d_flags.p, c->d_id.p, need, d_out.p, cap, d_cnt.p);
    HIPCHK(c, hipGetLastError());
    uint32_t cnt = 0;
    HIPCHK(c, hipMemcpyAsync(&cnt, d_cnt.p, 4, hipMemcpyDeviceToHost, c->stream)); HIPCHK(c, sync_stream(c->stream));
    std::vector<uint32_t> found(cnt);
    if (cnt) HIPCHK(c, hipMemcpy(found.data(), d_out.p, (size_t)cnt * 4, hipMemcpyDeviceToHost));
    d_out.release(nullptr); d_cnt.release(nullptr);
    std::sort(found.begin(), found.end());                                     // BTreeSet<EntityId>: ascending
    for (uint32_t i = 0; i < cnt && i < capacity; i++) ids[i] = found[i];
    if (n_out) *n_out = cnt;
    return RE_OK;
} RE_ABI_GUARD(c, "re_ecs_query")

extern "C" int re_get_out_of_bounds(re_ctx *c, uint32_t *ids, uint32_t capacity, uint32_t *n) try {
    if (!c || !c->h_th) return RE_E_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    if (c->tick_inflight) { int rc = finish_tick(c, nullptr); if (rc) return rc; } else { int rc = resolve(c); if (rc) return rc; }
    const uint32_t cnt = (uint32_t)c->h_oob_ids.size();
    for (uint32_t i = 0; i < cnt && i < capacity; i++) if (ids) ids[i] = c->h_oob_ids[i];
    if (n) *n = cnt;
    c->h_oob_ids.clear();
    return RE_OK;
} RE_ABI_GUARD(c, "re_get_out_of_bounds")

extern "C" int re_get_stats(re_ctx *c, re_stats *out) try {
    if (!c || !out) return RE_E_ARG;
    out->n_entities = c->n; out->n_dynamic = c->ndyn; out->n_sections = c->n_real_sections; out->n_shared_sections = c->nsh - (uint32_t)c->sh_free.size();   /* (retired entries of the table are holes until a host path rebuilds it) */ out->max_level = c->maxlevel; out->device_bytes = c->dev_bytes; out->n_probe_frames = c->probe_frames; out->n_table_rebuilds = c->n_rebuilds; out->n_fused_frames = c->n_fused_frames; out->reserved = c->n_lane_switches; out->n_seal_waits = c->n_seal_waits; out->n_sync_fallbacks = c->n_sync_fallbacks; out->n_section_slots = c->ncells; out->n_device_rebuckets = c->n_device_rebuckets; out->n_segment_redos = c->n_segment_redos; out->n_host_rebuckets = c->n_host_rebuckets;
    return RE_OK;
} RE_ABI_GUARD(c, "re_get_stats")

extern "C" int re_debug_get_sections(re_ctx *c, uint32_t capacity, uint64_t *keys, float *tight, uint32_t *n_local, uint32_t *n_static, uint8_t *is_static_section, uint32_t *n) try {
    if (!c) return RE_E_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    { int rc_ = resolve(c); if (rc_ != RE_OK) return rc_; }
    { int rc_ = sync_mirrors(c); if (rc_ != RE_OK) return rc_; }
    if (n) *n = c->n_real_sections;
    const uint32_t m = c->ncells;
    if (!m || !capacity) return RE_OK;
    std::vector<Aabb> t(m); std::vector<uint32_t> nl(m), ns(m); std::vector<uint8_t> f(m);
    HIPCHK(c, hipMemcpy(t.data(), c->d_cell_tight.p, (size_t)m * 24, hipMemcpyDeviceToHost));
    HIPCHK(c, hipMemcpy(nl.data(), c->d_cell_nlocal.p, (size_t)m * 4, hipMemcpyDeviceToHost));
    HIPCHK(c, hipMemcpy(ns.data(), c->d_cell_nstatic.p, (size_t)m * 4, hipMemcpyDeviceToHost));
    HIPCHK(c, hipMemcpy(f.data(), c->d_cell_flags.p, m, hipMemcpyDeviceToHost));
    std::vector<uint32_t> order; order.reserve(m);
    for (uint32_t i = 0; i < m; i++) if (!(f[i] & CF_PAD)) order.push_back(i);            // padding slots are not world sections
    std::sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return c->h_cell_key[a] < c->h_cell_key[b]; });   // ascending key (slots are not, after patches)
    for (uint32_t o = 0; o < order.size() && o < capacity; o++) {
        const uint32_t i = order[o];
        if (keys) keys[o] = c->h_cell_key[i];
        if (tight) memcpy(tight + (size_t)o * 6, &t[i], 24);
        if (n_local) n_local[o] = nl[i];
        if (n_static) n_static[o] = ns[i];
        if (is_static_section) is_static_section[o] = f[i] & CF_STATIC_SECTION;
    }
    return RE_OK;
} RE_ABI_GUARD(c, "re_debug_get_sections")

// the shared world sections in canonical id order (keys lexicographic, then count): ids, AABB, members (active, then static; each in ascending EntityId).
// Read from the DEVICE table; the host mirrors are checked against it on the way (RE_E_STATE when they are out of step).
extern "C" int re_debug_get_shared_sections(re_ctx *c, uint32_t capacity, uint64_t *keys, uint8_t *n_keys, float *aabb, uint32_t *n_active, uint32_t *n_static,
                                            uint32_t member_capacity, uint32_t *member_ids, uint32_t *member_offsets, uint32_t *n) try {
    if (!c) return RE_E_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    { int rc_ = resolve(c); if (rc_ != RE_OK) return rc_; }
    { int rc_ = sync_mirrors(c); if (rc_ != RE_OK) return rc_; }
    const uint32_t m = c->nsh;
    std::vector<uint32_t> order; order.reserve(m);
    for (uint32_t i = 0; i < m; i++) if (c->h_shids[i].nk) order.push_back(i);
    std::sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return c->h_shids[a] < c->h_shids[b]; });
    if (n) *n = (uint32_t)order.size();
    if (!m || !capacity) return RE_OK;
    std::vector<Aabb> box(m); std::vector<uint32_t> na(m), ns(m), bg(m), rows(std::max(c->pool_used, 1u)); std::vector<int32_t> cells((size_t)m * 8);
    HIPCHK(c, hipMemcpy(box.data(), c->d_sh_aabb.p, (size_t)m * 24, hipMemcpyDeviceToHost));
    HIPCHK(c, hipMemcpy(na.data(), c->d_sh_nact.p, (size_t)m * 4, hipMemcpyDeviceToHost)); HIPCHK(c, hipMemcpy(ns.data(), c->d_sh_nstat.p, (size_t)m * 4, hipMemcpyDeviceToHost));
    HIPCHK(c, hipMemcpy(bg.data(), c->d_sh_begin.p, (size_t)m * 4, hipMemcpyDeviceToHost)); HIPCHK(c, hipMemcpy(cells.data(), c->d_sh_cells.p, (size_t)m * 32, hipMemcpyDeviceToHost));
    if (c->pool_used) HIPCHK(c, hipMemcpy(rows.data(), c->d_rows.p, (size_t)c->pool_used * 4, hipMemcpyDeviceToHost));
    uint32_t off = 0;
    for (uint32_t o = 0; o < order.size(); o++) {
        const uint32_t i = order[o];
        if (na[i] != c->h_sh_nact[i] || ns[i] != c->h_sh_nstat[i] || bg[i] != c->h_sh_begin[i]) return c->fail(RE_E_STATE, "shared section %u: host mirror out of step (%u+%u members at %u on the device, %u+%u at %u on the host)", i, na[i], ns[i], bg[i], c->h_sh_nact[i], c->h_sh_nstat[i], c->h_sh_begin[i]);
        if ((uint64_t)bg[i] + na[i] + ns[i] > c->pool_used) return c->fail(RE_E_STATE, "shared section %u: members outside the row pool", i);
        for (uint32_t k = 0; k < 8; k++) {
            const int32_t ci = cells[(size_t)i * 8 + k];
            if (k < c->h_shids[i].nk ? (ci < 0 || (uint32_t)ci >= c->ncells || c->h_cell_key[ci] != c->h_shids[i].keys[k]) : ci >= 0) return c->fail(RE_E_STATE, "shared section %u: link %u does not lead to its section (slot %d)", i, k, ci);
            if (ci != c->h_sh_cells[(size_t)i * 8 + k]) return c->fail(RE_E_STATE, "shared section %u: host mirror of link %u out of step", i, k);
        }
        for (uint32_t k = 0; k < na[i] + ns[i]; k++) {
            const uint32_t r = rows[bg[i] + k];
            if (r >= c->ghost_base) continue;
            if (c->h_row_cell[r] != (ROW_CELL_SHARED | i) || c->h_rows[bg[i] + k] != r) return c->fail(RE_E_STATE, "shared section %u: host mirror of member %u out of step", i, k);
        }
        if (o >= capacity) continue;
        if (keys) memcpy(keys + (size_t)o * 8, c->h_shids[i].keys, 64);
        if (n_keys) n_keys[o] = (uint8_t)c->h_shids[i].nk;
        if (aabb) memcpy(aabb + (size_t)o * 6, &box[i], 24);
        if (n_active) n_active[o] = na[i];
        if (n_static) n_static[o] = ns[i];
        if (member_offsets) member_offsets[o] = off;
        for (uint32_t k = 0; k < na[i] + ns[i]; k++, off++) if (member_ids && off < member_capacity) { const uint32_t r = rows[bg[i] + k]; member_ids[off] = r < c->n ? c->h_id[r] : 0xFFFFFFFFu; }
        if (member_offsets) member_offsets[o + 1] = off;
    }
    return RE_OK;
} RE_ABI_GUARD(c, "re_debug_get_shared_sections")

extern "C" int re_debug_get_visible_sections(re_ctx *c, uint32_t capacity, uint64_t *keys, uint8_t *multiplicity, uint32_t *n) try {
    if (!c || !c->have_cull) return RE_E_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    if (c->cull_inflight) { int rc = finish_cull(c, nullptr); if (rc) return rc; }
    uint32_t cap = std::max(c->ncells, 1u);
    uint32_t *d_idx = nullptr, *d_cnt = nullptr; uint8_t *d_mult = nullptr;
    HIPCHK(c, hipMalloc(reinterpret_cast<void **>(&d_idx), (size_t)cap * 4)); HIPCHK(c, hipMalloc(reinterpret_cast<void **>(&d_mult), cap)); HIPCHK(c, hipMalloc(reinterpret_cast<void **>(&d_cnt), 4));
    HIPCHK(c, hipMemsetAsync(d_cnt, 0, 4, c->stream));
    if (c->ncells) hipLaunchKernelGGL(k_collect_visible, dim3((c->ncells + 255) / 256), dim3(256), 0, c->stream, c->ncells, c->d_cell_stamp.p, c->frame, d_idx, d_mult, cap, d_cnt);
    uint32_t cnt = 0;
    HIPCHK(c, hipMemcpyAsync(&cnt, d_cnt, 4, hipMemcpyDeviceToHost, c->stream));
    { int rc_ = resolve(c); if (rc_ != RE_OK) return rc_; }
    { int rc_ = sync_mirrors(c); if (rc_ != RE_OK) return rc_; }
    std::vector<uint32_t> idx(cnt); std::vector<uint8_t> mult(cnt);
    if (cnt) { HIPCHK(c, hipMemcpy(idx.data(), d_idx, (size_t)cnt * 4, hipMemcpyDeviceToHost)); HIPCHK(c, hipMemcpy(mult.data(), d_mult, cnt, hipMemcpyDeviceToHost)); }
    (void)hipFree(d_idx); (void)hipFree(d_mult); (void)hipFree(d_cnt);
    std::vector<uint32_t> order(cnt); for (uint32_t i = 0; i < cnt; i++) order[i] = i;
    std::sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return c->h_cell_key[idx[a]] < c->h_cell_key[idx[b]]; });
    for (uint32_t i = 0; i < cnt && i < capacity; i++) { if (keys) keys[i] = c->h_cell_key[idx[order[i]]]; if (multiplicity) multiplicity[i] = mult[order[i]]; }
    if (n) *n = cnt;
    return RE_OK;
} RE_ABI_GUARD(c, "re_debug_get_visible_sections")

extern "C" int re_get_timings(re_ctx *c, float *cull_us, float *pack_us, float *tick_us) try {
    if (!c) return RE_E_ARG;
    if (!cull_us && !pack_us && !tick_us) { c->timings_on = false; c->timings_pending = false; return RE_OK; }   // all NULL: switch the recording off again
    c->timings_on = true;                                                     // from now on synchronous frames are timed
    if (c->timings_pending) {
        HIPCHK(c, hipSetDevice(c->device)); HIPCHK(c, sync_stream(c->stream));
        (void)hipEventElapsedTime(&c->t_cull, c->ev[0], c->ev[1]); (void)hipEventElapsedTime(&c->t_pack, c->ev[1], c->ev[2]);
        c->t_cull *= 1000.f; c->t_pack *= 1000.f; c->timings_pending = false;
    }
    if (cull_us) *cull_us = c->t_cull; if (pack_us) *pack_us = c->t_pack; if (tick_us) *tick_us = c->t_tick;
    return RE_OK;
} RE_ABI_GUARD(c, "re_get_timings")

extern "C" int re_debug_copy_to_host(re_ctx *c, const void *d_src, void *dst, uint64_t bytes) try {
    if (!c || (bytes && (!d_src || !dst))) return RE_E_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    if (bytes) HIPCHK(c, hipMemcpyAsync(dst, d_src, bytes, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, sync_stream(c->stream));
    return RE_OK;
} RE_ABI_GUARD(c, "re_debug_copy_to_host")
extern "C" void *re_get_stream(re_ctx *c) { return c ? (void *)c->stream : nullptr; }

// per-launch HIP-event timing of the dominant kernel (k_scan_cull) over a timed region, every `every`-th launch (timed dispatches
// carry completion signals that cost a few microseconds of queue time each, so a throughput run samples):
// re_timing_begin(ctx, max_launches, every) ... frames ... re_timing_collect(ctx, us[], cap, &n)
extern "C" int re_timing_begin(re_ctx *c, uint32_t max_launches, uint32_t every) try {
    if (!c) return RE_E_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    while (c->k1_events.size() < (size_t)max_launches * 2) { hipEvent_t e; HIPCHK(c, hipEventCreate(&e)); c->k1_events.push_back(e); }
    c->k1_used = 0; c->k1_timing = max_launches > 0; c->k1_every = std::max(every & 0xFFFFu, 1u); c->k1_kind = every >> 16; c->k1_seen = 0;
    if (c->k1_kind > RE_TIME_PACK_LARGE) return c->fail(RE_E_ARG, "re_timing_begin: unknown kernel selector");
    return RE_OK;
} RE_ABI_GUARD(c, "re_timing_begin")
extern "C" int re_timing_collect(re_ctx *c, float *us, uint32_t capacity, uint32_t *n) try {
    if (!c) return RE_E_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    { int rc_ = resolve(c); if (rc_ != RE_OK) return rc_; }
    uint32_t launches = c->k1_used / 2;
    for (uint32_t i = 0; i < launches && i < capacity; i++) { float ms = 0; HIPCHK(c, hipEventElapsedTime(&ms, c->k1_events[2 * i], c->k1_events[2 * i + 1])); if (us) us[i] = ms * 1000.f; }
    if (n) *n = launches;
    c->k1_timing = false; c->k1_used = 0;
    return RE_OK;
} RE_ABI_GUARD(c, "re_timing_collect")
#ifdef RE_EXP_STAMPS
extern "C" int re_debug_get_timeline(re_ctx *c, unsigned long long *out, uint32_t nwaves) try {
    if (!c || !out || !c->d_timeline.p) return RE_E_ARG;
    HIPCHK(c, sync_stream(c->stream));
    HIPCHK(c, hipMemcpy(out, c->d_timeline.p, (size_t)std::min(nwaves, c->nlists) * 64, hipMemcpyDeviceToHost));
    return RE_OK;
} RE_ABI_GUARD(c, "re_debug_get_timeline")
#endif
extern "C" int re_get_last_candidates(re_ctx *c, uint32_t *n_candidates) try {
    if (!c || !c->h_res || !n_candidates) return RE_E_ARG;
    *n_candidates = c->h_res->n_candidates; return RE_OK;
} RE_ABI_GUARD(c, "re_get_last_candidates")
