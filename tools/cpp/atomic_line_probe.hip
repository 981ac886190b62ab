// How do global atomics serialise: per address or per cache line?  W waves, lane 0 of each does one atomicAdd on word ((wave % A) * S) of a buffer.
// (A, S) = (1, 1): one address;  (64, 1): 64 adjacent words = two 128-byte lines;  (64, 32): 64 words in 64 different lines;  (512, 1) / (512, 32) likewise.
// build: hipcc --offload-arch=gfx950 -O2 tools/cpp/atomic_line_probe.hip -o tools/cpp/atomic_line_probe
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
__global__ void k_atomics(uint32_t *buf, uint32_t A, uint32_t S, uint32_t per_wave) {
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63u;
    if (lane < per_wave) atomicAdd(&buf[(size_t)((wave * per_wave + lane) % A) * S], 1u);
}
int main() {
    uint32_t *buf; hipMalloc((void **)&buf, 1u << 22); hipMemset(buf, 0, 1u << 22);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const uint32_t W = 16384;
    const uint32_t cfg[][3] = { {1, 1, 1}, {8, 1, 1}, {8, 32, 1}, {64, 1, 1}, {64, 32, 1}, {512, 1, 1}, {512, 32, 1}, {512, 1, 64}, {512, 32, 64}, {4096, 1, 64}, {4096, 32, 64} };
    for (auto &c : cfg) {
        float best = 1e9f;
        for (int rep = 0; rep < 5; rep++) {
            hipEventRecord(e0, 0);
            hipLaunchKernelGGL(k_atomics, dim3(W / 4), dim3(256), 0, 0, buf, c[0], c[1], c[2]);
            hipEventRecord(e1, 0); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
        }
        printf("A = %4u addresses, stride %2u words, %2u atomics per wave: %8.2f us for %u atomics  (%.1f per us)\n", c[0], c[1], c[2], best * 1e3f, W * c[2], W * c[2] / (best * 1e3f));
    }
    return 0;
}
