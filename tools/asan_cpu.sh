#!/bin/bash
# Host-side AddressSanitizer + UndefinedBehaviorSanitizer build of the library (device code untouched: GPU sanitizers are not available on this pool)
# and the CPU test suite under it.  The suite reaches the host-only entry points (re_section_keys, re_history_*, the loader); the host bookkeeping
# behind the frame calls needs a device and is exercised by the GPU suite with the normal build.
#   tools/asan_cpu.sh            -> build into /tmp/re_asan, run `pytest -m "not gpu"`, print the number of sanitizer reports
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=${RE_ASAN_DIR:-/tmp/re_asan}; mkdir -p "$OUT"
ASAN_LIB=$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so | head -1)
C=$ROOT/render_engine_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O1 -g -std=c++17 -fPIC -shared -ffp-contract=off -fno-fast-math -fgpu-rdc -Wno-unused-result -mllvm -amdgpu-kernarg-preload-count=14 \
    -Xarch_host -fsanitize=address,undefined -Xarch_host -fno-omit-frame-pointer -shared-libsan -I "$ROOT/include" -I "$C" \
    "$C/re_kernels.hip" "$C/re_rebucket.hip" "$C/re_api.hip" "$C/re_lighting.hip" "$C/re_collide.hip" "$C/re_sort.hip" "$C/re_history.cpp" -o "$OUT/librender_engine_hip_asan.so"
cd "$ROOT"
# (test_no_exception_crosses_the_abi asks for a 2^62-byte vector to see the guard turn std::bad_alloc into a status code: ASan's allocator ends the process for
# that request instead of returning, so the witness runs with the normal build only)
RE_HIP_LIBRARY="$OUT/librender_engine_hip_asan.so" LD_PRELOAD="$ASAN_LIB" ASAN_OPTIONS=detect_leaks=0 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=0 \
    python -m pytest tests -q -m "not gpu" -p no:cacheprovider --deselect tests/test_c_abi.py::test_no_exception_crosses_the_abi 2>&1 | tee "$OUT/run.log" | tail -4
echo "sanitizer reports: $(grep -c 'runtime error\|AddressSanitizer' "$OUT/run.log" || true)"
