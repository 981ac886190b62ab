// What does the WRITE_SIZE counter report for the store patterns of the instance pack?  (k_pack_large shows 65 MB for 34 MB of output.)
// Each kernel writes exactly 64 MiB.  Run under:  rocprofv3 --pmc WRITE_SIZE --output-format csv -d <dir> -- tools/cpp/write_size_probe
//   k_w_dword_seq     coalesced 4-byte stores                      k_w_x4_seq      coalesced 16-byte stores
//   k_w_row64_perm    64-byte rows (4 lanes x 16 B) to a random permutation of the row slots (each 128-byte line gets its halves at different times)
//   k_w_row64_pairs   the same, but rows 2i and 2i+1 (one 128-byte line) written by neighbouring lane quads
//   k_w_run1k_perm    runs of 16 consecutive rows (1 KiB) to random places, run start aligned to 64 B only
// build: hipcc --offload-arch=gfx950 -O2 tools/cpp/write_size_probe.hip -o tools/cpp/write_size_probe
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>
#include <numeric>
#include <algorithm>
#include <random>
constexpr size_t BYTES = 64ull << 20, ROWS = BYTES / 64;
__global__ void k_w_dword_seq(uint32_t *p) { size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; p[i] = (uint32_t)i; }
__global__ void k_w_x4_seq(float4 *p) { size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; p[i] = make_float4((float)i, 1.f, 2.f, 3.f); }
__global__ void k_w_row64(float4 *p, const uint32_t *perm) { size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; size_t row = perm[t >> 2]; p[row * 4 + (t & 3)] = make_float4((float)t, 1.f, 2.f, 3.f); }
int main() {
    float4 *d; hipMalloc((void **)&d, BYTES + 4096); uint32_t *perm; hipMalloc((void **)&perm, ROWS * 4);
    std::vector<uint32_t> h(ROWS); std::mt19937 rng(7);
    auto upload = [&]() { hipMemcpy(perm, h.data(), ROWS * 4, hipMemcpyHostToDevice); };
    for (int rep = 0; rep < 3; rep++) {
        hipLaunchKernelGGL(k_w_dword_seq, dim3((unsigned)(BYTES / 4 / 256)), dim3(256), 0, 0, (uint32_t *)d);
        hipLaunchKernelGGL(k_w_x4_seq, dim3((unsigned)(BYTES / 16 / 256)), dim3(256), 0, 0, d);
        hipDeviceSynchronize();
    }
    // 1: random permutation of single rows
    std::iota(h.begin(), h.end(), 0u); std::shuffle(h.begin(), h.end(), rng); upload();
    for (int rep = 0; rep < 3; rep++) hipLaunchKernelGGL(k_w_row64, dim3((unsigned)(ROWS * 4 / 256)), dim3(256), 0, 0, d, perm);
    hipDeviceSynchronize();
    // 2: pairs of rows (whole 128-byte lines) permuted
    { std::vector<uint32_t> pr(ROWS / 2); std::iota(pr.begin(), pr.end(), 0u); std::shuffle(pr.begin(), pr.end(), rng); for (size_t i = 0; i < ROWS / 2; i++) { h[2 * i] = pr[i] * 2; h[2 * i + 1] = pr[i] * 2 + 1; } upload(); }
    for (int rep = 0; rep < 3; rep++) hipLaunchKernelGGL(k_w_row64, dim3((unsigned)(ROWS * 4 / 256)), dim3(256), 0, 0, d, perm);
    hipDeviceSynchronize();
    // 3: runs of 16 rows, run starts at odd row offsets (64-byte aligned, not 128)
    { std::vector<uint32_t> rn(ROWS / 16 - 1); std::iota(rn.begin(), rn.end(), 0u); std::shuffle(rn.begin(), rn.end(), rng);
      for (size_t i = 0; i < ROWS / 16 - 1; i++) for (uint32_t k = 0; k < 16; k++) h[i * 16 + k] = rn[i] * 16 + 1 + k;
      for (uint32_t k = 0; k < 16; k++) h[(ROWS / 16 - 1) * 16 + k] = (uint32_t)(ROWS - 16 + k); upload(); }
    for (int rep = 0; rep < 3; rep++) hipLaunchKernelGGL(k_w_row64, dim3((unsigned)(ROWS * 4 / 256)), dim3(256), 0, 0, d, perm);
    hipDeviceSynchronize();
    printf("done: launches in order: 3 x (dword_seq, x4_seq), 3 x row64 single-row permutation, 3 x row64 line pairs, 3 x row64 1 KiB runs at odd rows; each writes %zu bytes\n", BYTES);
    return 0;
}
