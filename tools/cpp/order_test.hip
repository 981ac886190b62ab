// Does a kernel launched behind another one on the same HIP stream ever see the first one's fire-and-forget atomics incomplete?
// build: hipcc --offload-arch=gfx950 -O2 tools/cpp/order_test.hip -o tools/cpp/order_test
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <chrono>
__global__ void k_count(uint32_t *ctr, uint32_t n, uint32_t spin) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (spin) { unsigned long long t0 = wall_clock64(); while (wall_clock64() - t0 < spin * (i % 7u)) {} }
    atomicAdd(ctr, 1u);                       // result unused: no wait for the atomic's return
}
__global__ void k_publish(const uint32_t *ctr, uint32_t *h_out, uint32_t *h_seq, uint32_t seq) {
    if (threadIdx.x || blockIdx.x) return;
    *h_out = *ctr; __threadfence_system(); *h_seq = seq;
}
__global__ void k_zero(uint32_t *ctr) { if (!threadIdx.x && !blockIdx.x) *ctr = 0; }
int main(int argc, char **argv) {
    const uint32_t n = argc > 1 ? atoi(argv[1]) : 100000, iters = argc > 2 ? atoi(argv[2]) : 20000, spin = argc > 3 ? atoi(argv[3]) : 0;
    hipStream_t st; hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    uint32_t *d_ctr; hipMalloc((void **)&d_ctr, 4);
    uint32_t *h_out, *h_seq, *d_out, *d_seq;
    hipHostMalloc((void **)&h_out, 4, hipHostMallocMapped); hipHostMalloc((void **)&h_seq, 4, hipHostMallocMapped); *h_out = 0; *h_seq = 0;
    hipHostGetDevicePointer((void **)&d_out, h_out, 0); hipHostGetDevicePointer((void **)&d_seq, h_seq, 0);
    uint32_t bad_poll = 0, bad_sync = 0;
    for (uint32_t it = 1; it <= iters; it++) {
        hipLaunchKernelGGL(k_zero, dim3(1), dim3(64), 0, st, d_ctr);
        hipLaunchKernelGGL(k_count, dim3((n + 255) / 256), dim3(256), 0, st, d_ctr, n, spin);
        if (it & 1) {   // variant 1: publish kernel + poll
            hipLaunchKernelGGL(k_publish, dim3(1), dim3(64), 0, st, (const uint32_t *)d_ctr, d_out, d_seq, it);
            while (*(volatile uint32_t *)h_seq != it) {}
            if (*(volatile uint32_t *)h_out != n) { if (bad_poll < 5) printf("poll: iteration %u saw %u of %u\n", it, *h_out, n); bad_poll++; }
        } else {        // variant 2: stream synchronise + copy on the null stream
            hipStreamSynchronize(st);
            uint32_t v = 0; hipMemcpy(&v, d_ctr, 4, hipMemcpyDeviceToHost);
            if (v != n) { if (bad_sync < 5) printf("sync: iteration %u saw %u of %u\n", it, v, n); bad_sync++; }
        }
    }
    hipStreamSynchronize(st);
    printf("n %u iterations %u spin %u: short counts with publish+poll %u, with synchronise+copy %u\n", n, iters, spin, bad_poll, bad_sync);
    return 0;
}
