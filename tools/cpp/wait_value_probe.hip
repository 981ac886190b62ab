// wait_value_probe.hip -- can a second stream pick up work as soon as a kernel on the first stream has WRITTEN a word (hipStreamWaitValue32), i.e. before that kernel has
// ended and without the end-of-kernel barrier?  Measures, in GPU wall-clock ticks (100 MHz), the time from the producer's flag store to the start of the consumer kernel
//   (a) consumer behind hipStreamWaitValue32 on a second stream     (b) consumer on the same stream (the ordinary dependent launch)
// build: hipcc --offload-arch=gfx950 -O2 tools/cpp/wait_value_probe.hip -o /tmp/wvp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
__global__ void k_producer(uint32_t *flag, uint32_t seq, unsigned long long *t_store, uint32_t spin_after) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        t_store[0] = wall_clock64();
        __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    // the rest of the kernel keeps running for a while (the stream of a scan kernel behind its last candidate wave)
    unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < spin_after) {}
}
__global__ void k_consumer(unsigned long long *t_start) { if (threadIdx.x == 0) t_start[0] = wall_clock64(); }
int main() {
    hipStream_t a, b; CK(hipStreamCreateWithFlags(&a, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&b, hipStreamNonBlocking));
    uint32_t *flag = nullptr; unsigned long long *ts = nullptr, *h = nullptr;
    CK(hipMalloc(&flag, 64)); CK(hipMemset(flag, 0, 64)); CK(hipMalloc(&ts, 64)); CK(hipHostMalloc((void **)&h, 64, 0));
    for (int mode = 0; mode < 2; mode++) for (uint32_t spin : { 0u, 500u }) {
        std::vector<double> us;
        for (uint32_t seq = 1; seq <= 300; seq++) {
            if (mode == 0) {
                hipError_t e = hipStreamWaitValue32(b, flag, seq, hipStreamWaitValueEq, 0xFFFFFFFFu);
                if (e != hipSuccess) { printf("hipStreamWaitValue32: %s\n", hipGetErrorString(e)); return 0; }
                hipLaunchKernelGGL(k_consumer, dim3(1), dim3(64), 0, b, ts + 1);
                hipLaunchKernelGGL(k_producer, dim3(256), dim3(64), 0, a, flag, seq, ts, spin);
            } else {
                hipLaunchKernelGGL(k_producer, dim3(256), dim3(64), 0, a, flag, seq, ts, spin);
                hipLaunchKernelGGL(k_consumer, dim3(1), dim3(64), 0, a, ts + 1);
            }
            CK(hipStreamSynchronize(a)); CK(hipStreamSynchronize(b));
            CK(hipMemcpy(h, ts, 16, hipMemcpyDeviceToHost));
            if (seq > 20) us.push_back((double)(long long)(h[1] - h[0]) / 100.0);
        }
        std::sort(us.begin(), us.end());
        printf("%s, producer keeps running %u ticks after its store: flag store -> consumer start  median %.2f us  min %.2f  p90 %.2f\n",
               mode == 0 ? "second stream behind hipStreamWaitValue32" : "same stream (dependent launch)        ", spin, us[us.size() / 2], us[0], us[us.size() * 9 / 10]);
    }
    return 0;
}
