// stream_floor.hip -- how fast can ANY kernel read the 40 MB key stream of configs[1] once?  (development tool)
// Variants: bytes per wave (16-byte loads per lane, all in flight), one chunk per wave vs a resident grid striding over the array.
// Build: hipcc --offload-arch=gfx950 -O3 -o stream_floor tools/cpp/stream_floor.hip
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int LOADS> __global__ __launch_bounds__(256) void k_chunk(const uint4 *__restrict__ p, uint32_t n16, uint32_t *out) {
    const uint32_t lane = threadIdx.x & 63u, wave = blockIdx.x * 4u + (threadIdx.x >> 6);
    const uint32_t base = wave * (64u * LOADS) + lane;
    uint4 v[LOADS];
#pragma unroll
    for (int i = 0; i < LOADS; i++) { const uint32_t q = base + i * 64u; v[i] = q < n16 ? p[q] : make_uint4(0, 0, 0, 0); }
    uint32_t acc = 0;
#pragma unroll
    for (int i = 0; i < LOADS; i++) acc |= (v[i].x ^ 0x9E3779B9u) & (v[i].y ^ 0x85EBCA6Bu) & v[i].z & v[i].w;
    if (acc == 0xFFFFFFFFu) out[0] = acc;
}
template <int LOADS> __global__ __launch_bounds__(256) void k_stride(const uint4 *__restrict__ p, uint32_t n16, uint32_t *out) {
    const uint32_t lane = threadIdx.x & 63u, wave = blockIdx.x * 4u + (threadIdx.x >> 6), nwaves = gridDim.x * 4u;
    uint32_t acc = 0;
    for (uint32_t base = wave * (64u * LOADS) + lane; base < n16 + 64u * LOADS; base += nwaves * (64u * LOADS)) {
        uint4 v[LOADS];
#pragma unroll
        for (int i = 0; i < LOADS; i++) { const uint32_t q = base + i * 64u; v[i] = q < n16 ? p[q] : make_uint4(0, 0, 0, 0); }
#pragma unroll
        for (int i = 0; i < LOADS; i++) acc |= (v[i].x ^ 0x9E3779B9u) & (v[i].y ^ 0x85EBCA6Bu) & v[i].z & v[i].w;
    }
    if (acc == 0xFFFFFFFFu) out[0] = acc;
}
// the 2 KB-per-wave chunk kernel with the scan kernel's resource signature: LDS bytes per workgroup, registers per lane (a clobbered high register)
#define RES_KERNEL(NAME, LDS_BYTES, TOPREG) \
__global__ __launch_bounds__(256) void NAME(const uint4 *__restrict__ p, uint32_t n16, uint32_t *out) { \
    __shared__ uint32_t pad[LDS_BYTES / 4 + 1]; \
    const uint32_t lane = threadIdx.x & 63u, wave = blockIdx.x * 4u + (threadIdx.x >> 6); \
    const uint32_t base = wave * 128u + lane; \
    uint4 v[2]; \
    for (int i = 0; i < 2; i++) { const uint32_t q = base + i * 64u; v[i] = q < n16 ? p[q] : make_uint4(0, 0, 0, 0); } \
    asm volatile("v_mov_b32 " TOPREG ", 0" ::: TOPREG); \
    uint32_t acc = 0; \
    for (int i = 0; i < 2; i++) acc |= (v[i].x ^ 0x9E3779B9u) & (v[i].y ^ 0x85EBCA6Bu) & v[i].z & v[i].w; \
    if (acc == 0xFFFFFFFFu) { pad[threadIdx.x] = acc; __syncthreads(); out[0] = pad[(threadIdx.x + 1) & 255]; } \
}
RES_KERNEL(k_res_0_32, 0, "v31") RES_KERNEL(k_res_0_80, 0, "v79") RES_KERNEL(k_res_0_128, 0, "v127")
RES_KERNEL(k_res_16k_32, 16384, "v31") RES_KERNEL(k_res_16k_80, 16384, "v79") RES_KERNEL(k_res_32k_80, 32768, "v79")
__global__ void k_empty(uint32_t *out) { if (threadIdx.x == 9999) out[0] = 1; }
// round 3: the 2 KB-per-wave chunk kernel with, step by step, what a wave of k_scan_cull that holds no candidate does besides reading its keys
//   MODE bit 0: the level word of the wave's chunk (a scalar global load) and, behind it, the box of that level from the kernel-argument segment
//   MODE bit 1: the guard-bit test of the 8 keys of a lane + one ballot per key
//   MODE bit 2: the trailing block (two more words of the ~1.4 KB kernel-argument segment, the last workgroup copies 600 bytes of it)
//   MODE bit 3: the workgroup -> chunk mapping through four spans (scalar arithmetic in front of everything)
struct LikeBox { uint32_t lo, hi; };
struct LikeArgs { LikeBox box[16]; uint32_t nsh, stale, words[320]; };
template <int MODE> __global__ __launch_bounds__(256) void k_like(const uint4 *__restrict__ p, uint32_t n16, uint32_t nsp, uint32_t s0, uint32_t c0, uint32_t s1, uint32_t c1, uint32_t s2, uint32_t c2,
                                                                  uint32_t s3, uint32_t c3, const uint32_t *__restrict__ chunk_level, uint32_t *out, LikeArgs A) {
    __shared__ uint32_t lds[4096];
    uint32_t chunk = blockIdx.x;
    if (MODE & 8) {
        const uint32_t st[4] = { s0, s1, s2, s3 }, ct[4] = { c0, c1, c2, c3 };
        uint32_t acc = 0; bool in_span = false;
#pragma unroll
        for (uint32_t i = 0; i < 4; i++) if (i < nsp && !in_span) { if (chunk < acc + ct[i]) { chunk = st[i] + (chunk - acc); in_span = true; } else acc += ct[i]; }
        if (!in_span) { chunk -= acc;
#pragma unroll
            for (uint32_t i = 0; i < 4; i++) if (i < nsp && chunk >= st[i]) chunk += ct[i]; }
    }
    const uint32_t lane = threadIdx.x & 63u, wave = chunk * 4u + (threadIdx.x >> 6);
    const uint32_t base = wave * 128u + lane;
    uint4 v[2];
#pragma unroll
    for (int i = 0; i < 2; i++) { const uint32_t q = base + i * 64u; v[i] = p[q < n16 ? q : n16 - 1u]; }
    uint32_t lv = 0;
    if (MODE & 1) lv = chunk_level[__builtin_amdgcn_readfirstlane(wave)] & 15u;
    const LikeBox b = A.box[lv];
    uint64_t any = 0;
    if (MODE & 2) {
        const uint32_t G = 0x20080200u, hig = b.hi | G;
#pragma unroll
        for (int i = 0; i < 2; i++) { const uint32_t w4[4] = { v[i].x, v[i].y, v[i].z, v[i].w };
#pragma unroll
            for (int h = 0; h < 4; h++) { const uint32_t x = w4[h]; const uint32_t in = ((x | G) - b.lo) & (hig - x) & G; any |= __ballot(in == G && (int32_t)x >= 0); } }
    } else { uint32_t acc = 0; for (int i = 0; i < 2; i++) acc |= (v[i].x ^ 0x9E3779B9u) & (v[i].y ^ 0x85EBCA6Bu) & v[i].z & v[i].w; any = acc == 0xFFFFFFFFu ? 1 : 0; }
    if (any) { lds[threadIdx.x] = (uint32_t)any; __syncthreads(); out[0] = lds[(threadIdx.x + 1) & 255]; }
    if (MODE & 4) {
        if (blockIdx.x * 256u < A.nsh && !A.stale) out[1] = A.words[0];
        if (blockIdx.x == gridDim.x - 1u && !A.stale) for (uint32_t i = threadIdx.x; i < 150u; i += 256u) out[2 + i] = A.words[i];
    }
}

template <typename F> static float median_us(F launch, hipStream_t st, int reps) {
    std::vector<float> us;
    for (int r = 0; r < reps + 10; r++) {
        hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
        launch(a, b);
        (void)hipStreamSynchronize(st);
        float ms = 0; (void)hipEventElapsedTime(&ms, a, b);
        if (r >= 10) us.push_back(ms * 1000.f);
        (void)hipEventDestroy(a); (void)hipEventDestroy(b);
    }
    std::sort(us.begin(), us.end());
    return us[us.size() / 2];
}

int main() {
    const uint32_t nkeys = 216u * 216u * 216u + 1300u;             // configs[1]: 10,077,696 section keys (+ padding slots), 4 bytes each
    const uint32_t n16 = (nkeys + 3u) / 4u;
    uint4 *d = nullptr; uint32_t *out = nullptr;
    CK(hipMalloc(&d, (size_t)n16 * 16 + 65536)); CK(hipMalloc(&out, 64));
    CK(hipMemset(d, 0x5A, (size_t)n16 * 16 + 65536));
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    const double mb = n16 * 16.0 / 1e6;
    auto report = [&](const char *name, float us) { printf("%-44s %7.2f us  %6.2f TB/s\n", name, us, mb / us / 1e6 * 1e6 / 1e6); };
    printf("%.2f MB\n", mb);
    report("empty kernel, 1 workgroup", median_us([&](hipEvent_t a, hipEvent_t b) { hipExtLaunchKernelGGL(k_empty, dim3(1), dim3(256), 0, st, a, b, 0, out); }, st, 50));
    report("empty kernel, 3,400 workgroups", median_us([&](hipEvent_t a, hipEvent_t b) { hipExtLaunchKernelGGL(k_empty, dim3(3400), dim3(256), 0, st, a, b, 0, out); }, st, 50));
    report("empty kernel, 13,600 workgroups", median_us([&](hipEvent_t a, hipEvent_t b) { hipExtLaunchKernelGGL(k_empty, dim3(13600), dim3(256), 0, st, a, b, 0, out); }, st, 50));
#define CHUNK(L) { const uint32_t waves = (n16 + 64u * L - 1u) / (64u * L), wgs = (waves + 3u) / 4u; char nm[96]; snprintf(nm, sizeof nm, "one chunk per wave, %d B/wave (%u waves)", L * 1024, waves); \
        report(nm, median_us([&](hipEvent_t a, hipEvent_t b) { hipExtLaunchKernelGGL(k_chunk<L>, dim3(wgs), dim3(256), 0, st, a, b, 0, d, n16, out); }, st, 50)); }
    CHUNK(1) CHUNK(2) CHUNK(4) CHUNK(8)
#define STRIDE(L, G) { char nm[96]; snprintf(nm, sizeof nm, "resident grid %d workgroups, %d B per step", G, L * 1024); \
        report(nm, median_us([&](hipEvent_t a, hipEvent_t b) { hipExtLaunchKernelGGL(k_stride<L>, dim3(G), dim3(256), 0, st, a, b, 0, d, n16, out); }, st, 50)); }
#define RES(K, LABEL) { const uint32_t waves = (n16 + 127u) / 128u, wgs = (waves + 3u) / 4u; \
        report("2048 B/wave, " LABEL, median_us([&](hipEvent_t a, hipEvent_t b) { hipExtLaunchKernelGGL(K, dim3(wgs), dim3(256), 0, st, a, b, 0, d, n16, out); }, st, 50)); }
    RES(k_res_0_32, "no LDS, 32 VGPRs") RES(k_res_0_80, "no LDS, 80 VGPRs") RES(k_res_0_128, "no LDS, 128 VGPRs")
    RES(k_res_16k_32, "16 KB LDS, 32 VGPRs") RES(k_res_16k_80, "16 KB LDS, 80 VGPRs") RES(k_res_32k_80, "32 KB LDS, 80 VGPRs")
    {   // round 3: what a candidate-free wave of k_scan_cull adds to the read
        uint32_t *lvl = nullptr; const uint32_t waves = (n16 + 127u) / 128u, wgs = (waves + 3u) / 4u;
        CK(hipMalloc(&lvl, (size_t)(waves + 8) * 4)); CK(hipMemset(lvl, 0, (size_t)(waves + 8) * 4));
        uint32_t *out2 = nullptr; CK(hipMalloc(&out2, 4096));
        LikeArgs A{}; for (int i = 0; i < 16; i++) { A.box[i].lo = 0x1F07C1F0u; A.box[i].hi = 0x00000001u; }      // (no key of the 0x5A fill passes)
        A.nsh = 0; A.stale = 0;
        const uint32_t s0 = wgs / 3, c0 = wgs / 7;
#define LIKE(M, LABEL) report("k_scan_cull-like: " LABEL, median_us([&](hipEvent_t a, hipEvent_t b) { hipExtLaunchKernelGGL(k_like<M>, dim3(wgs), dim3(256), 0, st, a, b, 0, d, n16, 1u, s0, c0, 0u, 0u, 0u, 0u, 0u, 0u, (const uint32_t *)lvl, out2, A); }, st, 50));
        LIKE(0, "read only (16 KB LDS, 1.4 KB kernarg)") LIKE(1, "+ level word -> box") LIKE(2, "+ tests and ballots") LIKE(3, "+ level word, tests") LIKE(7, "+ trailing block") LIKE(15, "+ span mapping (all)")
    }
    STRIDE(2, 1024) STRIDE(2, 2048) STRIDE(4, 1024) STRIDE(4, 2048) STRIDE(8, 512) STRIDE(8, 1024)
    return 0;
}
