// k1_tune.hip -- standalone micro-benchmark of key-scan variants (development tool, not part of the library).
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/k1_tune tools/k1_tune.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>
#include <string>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

struct Box { uint32_t bx, by, bz, nx, ny, nz; };
struct Params { Box a, b; uint32_t maxlevel; };

__device__ __forceinline__ bool cand(uint64_t key, const Params &P) {
    uint32_t lv = (uint32_t)(key >> 48);
    if (lv >= P.maxlevel) return false;
    uint32_t x = (uint32_t)(key >> 32) & 0xFFFF, z = (uint32_t)(key >> 16) & 0xFFFF, y = (uint32_t)key & 0xFFFF;
    // level-0 boxes shifted by level (this tool only has level-0 keys; same instruction mix as a table lookup from SGPRs)
    bool inl = ((x - P.a.bx) & 0xFFFF) < P.a.nx && ((y - P.a.by) & 0xFFFF) < P.a.ny && ((z - P.a.bz) & 0xFFFF) < P.a.nz;
    bool inr = ((x - P.b.bx) & 0xFFFF) < P.b.nx && ((y - P.b.by) & 0xFFFF) < P.b.ny && ((z - P.b.bz) & 0xFFFF) < P.b.nz;
    return inl | inr;
}

// V0: pure streaming read floor (xor-reduce so the loads are not dead)
template <int ITERS> __global__ __launch_bounds__(256) void v_read(const ulonglong2 *__restrict__ kp, uint32_t npairs, unsigned long long *out) {
    unsigned long long acc = 0;
    uint32_t base = blockIdx.x * (256 * ITERS) + threadIdx.x;
    ulonglong2 kk[ITERS];
#pragma unroll
    for (int it = 0; it < ITERS; it++) { uint32_t p = base + it * 256; kk[it] = kp[p < npairs ? p : npairs - 1]; }
#pragma unroll
    for (int it = 0; it < ITERS; it++) acc ^= kk[it].x ^ kk[it].y;
    if (acc == 0x1234567ull) out[0] = acc;
}
// V0g: grid-stride read floor
__global__ __launch_bounds__(256) void v_read_gs(const ulonglong2 *__restrict__ kp, uint32_t npairs, unsigned long long *out) {
    unsigned long long acc = 0;
    for (uint32_t p = blockIdx.x * 256 + threadIdx.x; p < npairs; p += gridDim.x * 256) { ulonglong2 k = kp[p]; acc ^= k.x ^ k.y; }
    if (acc == 0x1234567ull) out[0] = acc;
}
// V1: candidate test, batched loads, count only via ballot (no queue)
template <int ITERS> __global__ __launch_bounds__(256) void v_test(const ulonglong2 *__restrict__ kp, uint32_t npairs, Params P, uint32_t *count) {
    uint32_t base = blockIdx.x * (256 * ITERS) + threadIdx.x;
    ulonglong2 kk[ITERS];
#pragma unroll
    for (int it = 0; it < ITERS; it++) { uint32_t p = base + it * 256; kk[it] = kp[p < npairs ? p : npairs - 1]; }
    uint32_t n = 0;
#pragma unroll
    for (int it = 0; it < ITERS; it++) { uint32_t p = base + it * 256; if (p < npairs) { n += cand(kk[it].x, P); n += cand(kk[it].y, P); } }
    unsigned long long m = __ballot(n != 0);
    if (m) { for (int d = 32; d >= 1; d >>= 1) n += __shfl_down(n, d, 64); if ((threadIdx.x & 63) == 0) atomicAdd(count, n); }
}
// V1w: as V1 but every wave owns a contiguous run of 64*ITERS pairs (instead of the block-strided layout)
template <int ITERS> __global__ __launch_bounds__(256) void v_test_w(const ulonglong2 *__restrict__ kp, uint32_t npairs, Params P, uint32_t *count) {
    uint32_t lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    uint32_t base = (blockIdx.x * 4 + wid) * (64 * ITERS) + lane;
    ulonglong2 kk[ITERS];
#pragma unroll
    for (int it = 0; it < ITERS; it++) { uint32_t p = base + it * 64; kk[it] = kp[p < npairs ? p : npairs - 1]; }
    uint32_t n = 0;
#pragma unroll
    for (int it = 0; it < ITERS; it++) { uint32_t p = base + it * 64; if (p < npairs) { n += cand(kk[it].x, P); n += cand(kk[it].y, P); } }
    unsigned long long m = __ballot(n != 0);
    if (m) { for (int d = 32; d >= 1; d >>= 1) n += __shfl_down(n, d, 64); if (lane == 0) atomicAdd(count, n); }
}
// V1b: as V1 with the big by-value parameter block (same 1.2 KB kernarg as the real kernel)
struct LevelBoxT2 { uint32_t bx, by, bz, nx, ny, nz; float ll; uint32_t pad; };
struct BigParams2 { float planes[24]; float cam[4]; float misc[20]; uint32_t maxlevel, frame, e, pad; LevelBoxT2 box[2][16]; };
template <int ITERS> __global__ __launch_bounds__(256) void v_test_big(const ulonglong2 *__restrict__ kp, uint32_t npairs, BigParams2 BP, uint32_t *count) {
    Params P; P.maxlevel = BP.maxlevel;
    P.a = { BP.box[0][0].bx, BP.box[0][0].by, BP.box[0][0].bz, BP.box[0][0].nx, BP.box[0][0].ny, BP.box[0][0].nz };
    P.b = { BP.box[1][0].bx, BP.box[1][0].by, BP.box[1][0].bz, BP.box[1][0].nx, BP.box[1][0].ny, BP.box[1][0].nz };
    uint32_t base = blockIdx.x * (256 * ITERS) + threadIdx.x;
    ulonglong2 kk[ITERS];
#pragma unroll
    for (int it = 0; it < ITERS; it++) { uint32_t p = base + it * 256; kk[it] = kp[p < npairs ? p : npairs - 1]; }
    uint32_t n = 0;
#pragma unroll
    for (int it = 0; it < ITERS; it++) { uint32_t p = base + it * 256; if (p < npairs) { n += cand(kk[it].x, P); n += cand(kk[it].y, P); } }
    unsigned long long m = __ballot(n != 0);
    if (m) { for (int d = 32; d >= 1; d >>= 1) n += __shfl_down(n, d, 64); if ((threadIdx.x & 63) == 0) atomicAdd(count, n); }
}
// V2: grid-stride with 2-deep unroll
template <int UNR> __global__ __launch_bounds__(256) void v_test_gs(const ulonglong2 *__restrict__ kp, uint32_t npairs, Params P, uint32_t *count) {
    uint32_t n = 0;
    uint32_t stride = gridDim.x * 256;
    for (uint32_t p0 = blockIdx.x * 256 + threadIdx.x; p0 < npairs; p0 += stride * UNR) {
        ulonglong2 kk[UNR];
#pragma unroll
        for (int u = 0; u < UNR; u++) { uint32_t p = p0 + u * stride; kk[u] = kp[p < npairs ? p : npairs - 1]; }
#pragma unroll
        for (int u = 0; u < UNR; u++) { uint32_t p = p0 + u * stride; if (p < npairs) { n += cand(kk[u].x, P); n += cand(kk[u].y, P); } }
    }
    unsigned long long m = __ballot(n != 0);
    if (m) { for (int d = 32; d >= 1; d >>= 1) n += __shfl_down(n, d, 64); if ((threadIdx.x & 63) == 0) atomicAdd(count, n); }
}

// V3..V5: the real kernel's phase-1 structure, pieces switched on one at a time
struct LevelBoxT { uint32_t bx, by, bz, nx, ny, nz; float ll; uint32_t pad; };
struct BigParams { float planes[24]; float cam[4]; float misc[20]; uint32_t maxlevel, frame, e, pad; LevelBoxT box[2][16]; };
__device__ __forceinline__ bool inb(uint32_t x, uint32_t y, uint32_t z, const LevelBoxT &b) { return ((x - b.bx) & 0xFFFF) < b.nx && ((y - b.by) & 0xFFFF) < b.ny && ((z - b.bz) & 0xFFFF) < b.nz; }
template <int ITERS, int MODE> __global__ __launch_bounds__(256) void v_real(const ulonglong2 *__restrict__ kp, uint32_t npairs, const uint64_t *__restrict__ keys, BigParams P, uint32_t *count) {
    __shared__ uint32_t s_queue[4][64 * ITERS * 2];
    const uint32_t tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const uint32_t wave_pair0 = (blockIdx.x * 4 + wid) * (64u * ITERS);
    if (wave_pair0 >= npairs) return;
    ulonglong2 kk[ITERS];
#pragma unroll
    for (int it = 0; it < ITERS; it++) { uint32_t p = wave_pair0 + it * 64 + lane; kk[it] = kp[p < npairs ? p : npairs - 1]; }
    uint32_t lv0 = MODE >= 1 ? __builtin_amdgcn_readfirstlane((uint32_t)(kk[0].x >> 48)) : 0u;
    bool ok = lv0 < P.maxlevel; uint32_t lvi = ok ? lv0 : 0;
    const LevelBoxT a0 = P.box[0][lvi], b0 = P.box[1][lvi];
    uint32_t cmask = 0;
#pragma unroll
    for (int it = 0; it < ITERS; it++) {
        uint32_t p = wave_pair0 + it * 64 + lane;
        if (p < npairs) {
#pragma unroll
            for (int h = 0; h < 2; h++) {
                uint64_t key = h ? kk[it].y : kk[it].x;
                uint32_t lv = (uint32_t)(key >> 48); bool c;
                if (lv == lv0) { uint32_t x = (uint32_t)(key >> 32) & 0xFFFF, z = (uint32_t)(key >> 16) & 0xFFFF, y = (uint32_t)key & 0xFFFF; c = ok && (inb(x, y, z, a0) || inb(x, y, z, b0)); }
                else c = lv < P.maxlevel;
                if (c) cmask |= 1u << (it * 2 + h);
            }
        }
    }
    uint32_t cnt = __popc(cmask), qn = 0;
    if (MODE >= 2) {
        if (__ballot(cnt != 0)) {
            uint32_t v = cnt; for (int d = 1; d < 64; d <<= 1) { uint32_t o = __shfl_up(v, d, 64); if (lane >= (uint32_t)d) v += o; }
            qn = __shfl(v, 63, 64); uint32_t base = v - cnt;
            while (cmask) { uint32_t bit = __ffs(cmask) - 1; cmask &= cmask - 1; s_queue[wid][base++] = (wave_pair0 + (bit >> 1) * 64 + lane) * 2 + (bit & 1); }
        }
        uint32_t acc = 0;
        if (MODE >= 3) {
            for (uint32_t i0 = 0; i0 < qn; i0 += 64) {
                uint32_t i = i0 + lane;
                if (i < qn) { uint32_t c = s_queue[wid][i]; uint64_t key = keys[c]; uint32_t lv = (uint32_t)(key >> 48); LevelBoxT a = P.box[0][lv]; acc += inb((uint32_t)(key >> 32) & 0xFFFF, (uint32_t)key & 0xFFFF, (uint32_t)(key >> 16) & 0xFFFF, a) ? 1 : 2; }
            }
        } else acc = qn ? 1 : 0;
        if (__ballot(acc != 0)) { for (int d = 32; d >= 1; d >>= 1) acc += __shfl_down(acc, d, 64); if (lane == 0 && acc) atomicAdd(count, acc); }
    } else {
        unsigned long long m = __ballot(cnt != 0);
        if (m) { for (int d = 32; d >= 1; d >>= 1) cnt += __shfl_down(cnt, d, 64); if (lane == 0) atomicAdd(count, cnt); }
    }
}

// V6: packed 16-bit box test.  key = lv:16|x:16 (hi word), z:16|y:16 (lo word).
// in box  <=>  pk_min_u16(pk_sub_u16(word, base), n-1) == pk_sub_u16(word, base) for both words (level diff must be 0).
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pk_sub(uint32_t a, uint32_t b) { u16x2 r = __builtin_bit_cast(u16x2, a) - __builtin_bit_cast(u16x2, b); return __builtin_bit_cast(uint32_t, r); }
__device__ __forceinline__ uint32_t pk_min(uint32_t a, uint32_t b) { u16x2 r = __builtin_elementwise_min(__builtin_bit_cast(u16x2, a), __builtin_bit_cast(u16x2, b)); return __builtin_bit_cast(uint32_t, r); }
struct PBox { uint32_t sub_hi, sub_lo, min_hi, min_lo; };
struct PParams { PBox box[2][16]; uint32_t maxlevel; };
__device__ __forceinline__ bool pk_in(uint32_t hi, uint32_t lo, const PBox &b) {
    uint32_t dh = pk_sub(hi, b.sub_hi), dl = pk_sub(lo, b.sub_lo);
    return (pk_min(dh, b.min_hi) == dh) & (pk_min(dl, b.min_lo) == dl);
}
template <int ITERS, int MODE> __global__ __launch_bounds__(256) void v_packed(const ulonglong2 *__restrict__ kp, uint32_t npairs, const uint64_t *__restrict__ keys, PParams P, uint32_t *count) {
    __shared__ uint32_t s_queue[4][64 * ITERS * 2];
    const uint32_t tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const uint32_t wave_pair0 = (blockIdx.x * 4 + wid) * (64u * ITERS);
    if (wave_pair0 >= npairs) return;
    ulonglong2 kk[ITERS];
#pragma unroll
    for (int it = 0; it < ITERS; it++) { uint32_t p = wave_pair0 + it * 64 + lane; kk[it] = kp[p < npairs ? p : npairs - 1]; }
    uint32_t lv0 = __builtin_amdgcn_readfirstlane((uint32_t)(kk[0].x >> 48)) & 15u;
    const PBox a0 = P.box[0][lv0], b0 = P.box[1][lv0];
    unsigned long long m[ITERS * 2]; unsigned long long any = 0;
#pragma unroll
    for (int it = 0; it < ITERS; it++) {
        bool valid = (wave_pair0 + it * 64 + lane) < npairs;
#pragma unroll
        for (int h = 0; h < 2; h++) {
            uint64_t key = h ? kk[it].y : kk[it].x;
            uint32_t hi = (uint32_t)(key >> 32), lo = (uint32_t)key;
            bool c = valid & (pk_in(hi, lo, a0) | pk_in(hi, lo, b0));
            m[it * 2 + h] = __ballot(c); any |= m[it * 2 + h];
        }
    }
    uint32_t acc = 0;
    if (any) {
        uint32_t base = 0;
#pragma unroll
        for (int k = 0; k < ITERS * 2; k++) {
            if ((m[k] >> lane) & 1ull) s_queue[wid][base + __builtin_amdgcn_mbcnt_hi((uint32_t)(m[k] >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m[k], 0u))] = (wave_pair0 + (k >> 1) * 64 + lane) * 2 + (k & 1);
            base += __popcll(m[k]);
        }
        uint32_t qn = base;
        if (MODE >= 1) {
            for (uint32_t i0 = 0; i0 < qn; i0 += 64) { uint32_t i = i0 + lane; if (i < qn) { uint32_t c = s_queue[wid][i]; uint64_t key = keys[c]; acc += (key & 1) ? 1 : 2; } }
            for (int d = 32; d >= 1; d >>= 1) acc += __shfl_down(acc, d, 64);
        } else acc = qn;
        if (lane == 0) atomicAdd(count, acc);
    }
}

template <typename F> static void run(const char *name, F launch, size_t bytes, int reps = 60) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    std::vector<float> t;
    for (int i = 0; i < reps; i++) { launch(a, b); CK(hipDeviceSynchronize()); float ms; CK(hipEventElapsedTime(&ms, a, b)); if (i >= 10) t.push_back(ms * 1000.f); }
    std::sort(t.begin(), t.end());
    float med = t[t.size() / 2], mn = t[0];
    printf("%-28s median %7.2f us  min %7.2f us   %6.2f TB/s (median)\n", name, med, mn, bytes / (med * 1e-6) / 1e12);
}

int main(int argc, char **argv) {
    int axis = argc > 1 ? atoi(argv[1]) : 216;
    size_t n = (size_t)axis * axis * axis;
    std::vector<uint64_t> keys(n + 2, ~0ull);
    size_t i = 0;
    for (int x = 0; x < axis; x++) for (int z = 0; z < axis; z++) for (int y = 0; y < axis; y++) keys[i++] = ((uint64_t)(x + 20) << 32) | ((uint64_t)(z + 20) << 16) | (uint64_t)(y + 20);
    uint32_t npairs = (uint32_t)((n + 1) / 2);
    ulonglong2 *d; CK(hipMalloc(&d, (size_t)npairs * 16 + 64)); CK(hipMemcpy(d, keys.data(), (size_t)npairs * 16, hipMemcpyHostToDevice));
    unsigned long long *dout; CK(hipMalloc(&dout, 64)); uint32_t *dcnt; CK(hipMalloc(&dcnt, 64)); CK(hipMemset(dcnt, 0, 64));
    Params P; P.a = { 126, 126, 126, 4, 4, 4 }; P.b = { 120, 120, 112, 16, 16, 16 }; P.maxlevel = 8;
    size_t bytes = (size_t)npairs * 16;
    printf("keys %zu (%.1f MB)\n", n, bytes / 1e6);
    hipStream_t st; CK(hipStreamCreate(&st));
#define EXT(k, grid, ...) [&](hipEvent_t a, hipEvent_t b) { hipExtLaunchKernelGGL(k, dim3(grid), dim3(256), 0, st, a, b, 0, __VA_ARGS__); }
    run("read  batched x8", EXT(v_read<8>, (npairs + 2047) / 2048, d, npairs, dout), bytes);
    run("read  batched x4", EXT(v_read<4>, (npairs + 1023) / 1024, d, npairs, dout), bytes);
    run("read  batched x2", EXT(v_read<2>, (npairs + 511) / 512, d, npairs, dout), bytes);
    run("read  batched x1", EXT(v_read<1>, (npairs + 255) / 256, d, npairs, dout), bytes);
    run("read  batched x16", EXT(v_read<16>, (npairs + 4095) / 4096, d, npairs, dout), bytes);
    for (int g : { 1024, 2048, 4096, 8192 }) { std::string nm = "read  grid-stride g=" + std::to_string(g); run(nm.c_str(), EXT(v_read_gs, g, d, npairs, dout), bytes); }
    run("test  batched x8", EXT(v_test<8>, (npairs + 2047) / 2048, d, npairs, P, dcnt), bytes);
    run("test  batched x4", EXT(v_test<4>, (npairs + 1023) / 1024, d, npairs, P, dcnt), bytes);
    run("test  batched x2", EXT(v_test<2>, (npairs + 511) / 512, d, npairs, P, dcnt), bytes);
    for (int g : { 2048, 4096 }) {
        std::string nm = "test  grid-stride u2 g=" + std::to_string(g); run(nm.c_str(), EXT(v_test_gs<2>, g, d, npairs, P, dcnt), bytes);
        nm = "test  grid-stride u4 g=" + std::to_string(g); run(nm.c_str(), EXT(v_test_gs<4>, g, d, npairs, P, dcnt), bytes);
    }
    run("test  wave-contig x4", EXT(v_test_w<4>, (npairs + 1023) / 1024, d, npairs, P, dcnt), bytes);
    run("test  wave-contig x2", EXT(v_test_w<2>, (npairs + 511) / 512, d, npairs, P, dcnt), bytes);
    { BigParams2 B2{}; B2.maxlevel = 8; B2.box[0][0] = { 126, 126, 126, 4, 4, 4, 64.f, 0 }; B2.box[1][0] = { 120, 120, 112, 16, 16, 16, 64.f, 0 };
      run("test  x4 big kernarg", EXT(v_test_big<4>, (npairs + 1023) / 1024, d, npairs, B2, dcnt), bytes); }
    BigParams BP{}; BP.maxlevel = 8;
    for (int l = 0; l < 16; l++) { BP.box[0][l] = { 126u >> l, 126u >> l, 126u >> l, 4, 4, 4, 64.f, 0 }; BP.box[1][l] = { 120u >> l, 120u >> l, 112u >> l, 16, 16, 16, 64.f, 0 }; }
    const uint64_t *dk = reinterpret_cast<const uint64_t *>(d);
    run("real x4 mode0 (kernarg box)", EXT((v_real<4, 0>), (npairs + 1023) / 1024, d, npairs, dk, BP, dcnt), bytes);
    run("real x4 mode1 (+lv0 from key)", EXT((v_real<4, 1>), (npairs + 1023) / 1024, d, npairs, dk, BP, dcnt), bytes);
    run("real x4 mode2 (+LDS queue)", EXT((v_real<4, 2>), (npairs + 1023) / 1024, d, npairs, dk, BP, dcnt), bytes);
    run("real x4 mode3 (+phase2 lite)", EXT((v_real<4, 3>), (npairs + 1023) / 1024, d, npairs, dk, BP, dcnt), bytes);
    run("real x2 mode3", EXT((v_real<2, 3>), (npairs + 511) / 512, d, npairs, dk, BP, dcnt), bytes);
    run("real x8 mode3", EXT((v_real<8, 3>), (npairs + 2047) / 2048, d, npairs, dk, BP, dcnt), bytes);
    PParams PP{}; PP.maxlevel = 8;
    for (int l = 0; l < 16; l++) {
        auto mk = [&](uint32_t bx, uint32_t by, uint32_t bz, uint32_t nx, uint32_t ny, uint32_t nz) { PBox b; b.sub_hi = ((uint32_t)l << 16) | bx; b.sub_lo = (bz << 16) | by; b.min_hi = nx - 1; b.min_lo = ((nz - 1) << 16) | (ny - 1); return b; };
        PP.box[0][l] = mk(126 >> l, 126 >> l, 126 >> l, 4, 4, 4); PP.box[1][l] = mk(120 >> l, 120 >> l, 112 >> l, 16, 16, 16);
    }
    CK(hipMemset(dcnt, 0, 64));
    run("packed x4 mode0", EXT((v_packed<4, 0>), (npairs + 1023) / 1024, d, npairs, dk, PP, dcnt), bytes, 11);
    { uint32_t c1; CK(hipMemcpy(&c1, dcnt, 4, hipMemcpyDeviceToHost)); printf("   packed candidates per launch: %u (expected %u)\n", c1 / 11, 16 * 16 * 16 + 0); }
    run("packed x4 mode0", EXT((v_packed<4, 0>), (npairs + 1023) / 1024, d, npairs, dk, PP, dcnt), bytes);
    run("packed x4 mode1", EXT((v_packed<4, 1>), (npairs + 1023) / 1024, d, npairs, dk, PP, dcnt), bytes);
    for (int ldsk : {0, 12, 18, 24, 30, 38}) {   // extra dynamic LDS (KiB) throttles resident workgroups per CU: 160 KiB / (8+ldsk)
        std::string nm = "packed x4 mode1 +" + std::to_string(ldsk) + "KiB LDS";
        run(nm.c_str(), [&](hipEvent_t a, hipEvent_t b) { hipExtLaunchKernelGGL((v_packed<4, 1>), dim3((npairs + 1023) / 1024), dim3(256), ldsk * 1024, st, a, b, 0, d, npairs, dk, PP, dcnt); }, bytes);
    }
    run("packed x2 mode1", EXT((v_packed<2, 1>), (npairs + 511) / 512, d, npairs, dk, PP, dcnt), bytes);
    run("packed x8 mode1", EXT((v_packed<8, 1>), (npairs + 2047) / 2048, d, npairs, dk, PP, dcnt), bytes);
    uint32_t cnt; CK(hipMemcpy(&cnt, dcnt, 4, hipMemcpyDeviceToHost)); printf("(candidate count accumulator %u)\n", cnt);
    return 0;
}
