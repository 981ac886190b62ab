// Does a result block written by a kernel into mapped pinned host memory ever become visible to the polling host AFTER the "done" word
// that the kernel stores behind a system-scope fence?  (The question behind `group table inconsistent (645 vs 718)`)
//
// The kernel mimics the publication of k_pack_small / k_group_scan: a workgroup of 256 or 1024 threads stores an InstanceRange-like table
// (20-byte records, one per thread) into host memory, fences, a barrier, then thread 0 stores a 40-byte result struct, fences and stores
// the sequence number.  The host polls the sequence number and then verifies EVERY word.  Variants:
//   alloc 0: table and result in two hipHostMalloc(Mapped) blocks      alloc 1: one hipHostMalloc(Mapped | Coherent) block
//   fence 0: every thread __threadfence_system() before the barrier    fence 1: only thread 0 fences (the k_group_scan bug pattern)
//   flag  0: __threadfence_system() + PLAIN store of the done word     flag  1: release store at system scope (__hip_atomic_store)
//   flag  2: __threadfence_system() + RELAXED system-scope atomic store (the scope bits alone)
//   flag  3: NO fence anywhere: every thread waits for its own stores (s_waitcnt vmcnt(0)) in front of the barrier, thread 0 stores the result, waits again and
//            stores the done word as a relaxed system-scope atomic -- the block lives in host memory, which the device does not cache, so the L2 write-back a
//            system-scope release performs (every dirty line of the XCD: megabytes behind a tick) has nothing of the block to write (round 3: tick_sign_off)
//            RESULT (round 3, MI355X): the TABLE is torn in ~99 % of the polled frames -- plain stores to mapped host memory do sit in the L2 until a system-scope
//            write-back --, the result struct (same 64-byte line as the done word, stored by the same thread) never
//   flag  4: as 3, but the result struct is stored word by word with relaxed SYSTEM-scope atomic stores (write-through), then the wait, then the done word:
//            the form tick_sign_off uses for its four counters (look at "torn result structs"; the table is stored as in 3 and tears)
//   load  0: idle device                                               load  1: a streaming kernel on a second stream keeps HBM busy
// build: hipcc --offload-arch=gfx950 -O2 tools/cpp/publish_order_test.hip -o tools/cpp/publish_order_test
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>

struct Rec { uint32_t w[5]; };
struct Res { uint32_t v[8]; uint32_t done, seal; };

__host__ __device__ inline uint32_t mix(uint32_t a, uint32_t b, uint32_t c) { uint32_t x = (a * 0x9E3779B1u) ^ (b * 0x85EBCA6Bu) ^ (c * 0xC2B2AE35u); x ^= x >> 15; x *= 0x2C1B3C6Du; return x ^ (x >> 12); }

__global__ void k_publish(Rec *table, uint32_t nrec, Res *res, uint32_t seq, int fence_all, int release_flag) {
    const uint32_t t = threadIdx.x;
    for (uint32_t i = t; i < nrec; i += blockDim.x) {
        Rec r; for (int k = 0; k < 5; k++) r.w[k] = mix(seq, i, (uint32_t)k);
        table[i] = r;
    }
    if (release_flag == 3 || release_flag == 4) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if (fence_all) __threadfence_system();
    __syncthreads();
    if (t == 0) {
        Res r = {}; for (int k = 0; k < 8; k++) r.v[k] = mix(seq, 0xFFFFu, (uint32_t)k);
        if (release_flag == 4) { for (int k = 0; k < 8; k++) __hip_atomic_store(&res->v[k], r.v[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
        else *res = r;
        if (release_flag == 3 || release_flag == 4) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __hip_atomic_store(&res->done, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
        else if (release_flag == 1) { __hip_atomic_store(&res->done, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); }
        else if (release_flag == 2) { __threadfence_system(); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __hip_atomic_store(&res->done, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
        else { __threadfence_system(); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); res->done = seq; }
    }
}
__global__ void k_stream(float4 *buf, size_t n, int rounds) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
    for (int r = 0; r < rounds; r++) for (size_t j = i; j < n; j += stride) { float4 v = buf[j]; v.x += 1.0f; buf[j] = v; }
}

int main(int argc, char **argv) {
    const uint32_t iters = argc > 1 ? (uint32_t)atoi(argv[1]) : 100000u;
    const uint32_t nrec = argc > 2 ? (uint32_t)atoi(argv[2]) : 320u;
    const int flag_lo = argc > 3 ? atoi(argv[3]) : 0, flag_hi = argc > 4 ? atoi(argv[4]) : 2;
    hipStream_t st, st2; hipStreamCreateWithFlags(&st, hipStreamNonBlocking); hipStreamCreateWithFlags(&st2, hipStreamNonBlocking);
    float4 *big = nullptr; const size_t nbig = (size_t)64 << 20; hipMalloc((void **)&big, nbig * sizeof(float4)); hipMemset(big, 0, nbig * sizeof(float4));
    int total_bad = 0;
    for (int load = 0; load < 2; load++)
    for (int alloc = 0; alloc < 2; alloc++)
    for (int fence_all = 1; fence_all >= 0; fence_all--)
    for (int threads = 256; threads <= 1024; threads *= 4)
    for (int rel = flag_lo; rel <= flag_hi; rel++) {
        Rec *h_tab = nullptr; Res *h_res = nullptr; void *block = nullptr;
        if (alloc == 0) { hipHostMalloc((void **)&h_tab, sizeof(Rec) * nrec, hipHostMallocMapped); hipHostMalloc((void **)&h_res, sizeof(Res), hipHostMallocMapped); }
        else { hipHostMalloc(&block, 4096 + sizeof(Rec) * nrec, hipHostMallocMapped | hipHostMallocCoherent); h_res = (Res *)block; h_tab = (Rec *)((char *)block + 4096); }
        memset(h_tab, 0, sizeof(Rec) * nrec); memset(h_res, 0, sizeof(Res));
        Rec *d_tab; Res *d_res; hipHostGetDevicePointer((void **)&d_tab, h_tab, 0); hipHostGetDevicePointer((void **)&d_res, h_res, 0);
        uint32_t bad_iters = 0, bad_words = 0, bad_res = 0; double worst_us = 0;
        if (load) hipLaunchKernelGGL(k_stream, dim3(2048), dim3(256), 0, st2, big, nbig, 1 + (int)(iters / 30u));
        const auto t_begin = std::chrono::steady_clock::now();
        for (uint32_t seq = 1; seq <= iters; seq++) {
            // an "asynchronous frame" nobody waits for, then the frame the host polls: two tables land back to back in the same memory
            if (seq % 3u) { hipLaunchKernelGGL(k_publish, dim3(1), dim3(threads), 0, st, d_tab, nrec, d_res, seq, fence_all, rel); continue; }
            hipLaunchKernelGGL(k_publish, dim3(1), dim3(threads), 0, st, d_tab, nrec, d_res, seq, fence_all, rel);
            const volatile uint32_t *flag = &h_res->done;
            while (*flag != seq) {}
            __atomic_thread_fence(__ATOMIC_ACQUIRE);
            auto check = [&](uint32_t *nw) { uint32_t b = 0; const volatile uint32_t *w = (const volatile uint32_t *)h_tab; for (uint32_t i = 0; i < nrec; i++) for (uint32_t k = 0; k < 5; k++) if (w[i * 5 + k] != mix(seq, i, k)) b++; *nw = b; return b == 0; };
            uint32_t nw = 0, rb = 0;
            for (int k = 0; k < 8; k++) if (((const volatile uint32_t *)h_res->v)[k] != mix(seq, 0xFFFFu, (uint32_t)k)) rb++;
            if (rb) bad_res++;
            if (!check(&nw)) {
                bad_iters++; bad_words += nw;
                const auto t0 = std::chrono::steady_clock::now(); uint32_t dummy;
                while (!check(&dummy) && std::chrono::steady_clock::now() - t0 < std::chrono::milliseconds(50)) {}
                const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
                if (us > worst_us) worst_us = us;
                if (bad_iters <= 3) printf("    seq %u: %u stale table words right after the done word; consistent %.1f us later\n", seq, nw, us);
            }
        }
        hipStreamSynchronize(st);
        const double el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count();
        printf("load %d alloc %s fence %s threads %4d flag %s: %u polled frames, torn tables %u (%u words, worst settle %.1f us), torn result structs %u  [%.2f s]\n",
               load, alloc ? "one-coherent-block" : "two-blocks", fence_all ? "every-thread" : "thread0-only", threads, rel == 1 ? "release-atomic" : rel == 2 ? "fence+relaxed-system-atomic" : rel == 3 ? "NO-fence: waits + relaxed-system-atomic" : rel == 4 ? "NO-fence: result words system-atomic + wait + relaxed-system-atomic" : "fence+plain", iters / 3u, bad_iters, bad_words, worst_us, bad_res, el);
        fflush(stdout);
        total_bad += (int)bad_iters + (int)bad_res;
        if (load) hipStreamSynchronize(st2);
        if (alloc == 0) { hipHostFree(h_tab); hipHostFree(h_res); } else hipHostFree(block);
    }
    printf("TOTAL torn observations: %d\n", total_bad);
    return 0;
}
