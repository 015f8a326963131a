#!/bin/bash
# CPU-only sanitizer run of libslamhip's host side (reference CI: .github/workflows/tests.yml).
# Builds the library with AddressSanitizer + UBSan on the HOST code only (-fno-gpu-sanitize: device ASan needs xnack+
# code objects, which this pool does not run) and drives every C-ABI entry point through its no-GPU paths.
# Needs hipcc (ROCm); exits 0 with a notice where it is absent (hosted CI runners).
set -e
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
if ! [ -x "$HIPCC" ]; then echo "asan_host.sh: no hipcc here, skipped"; exit 0; fi
cd "$(dirname "$0")/.."
mkdir -p build/asan
SAN="-fsanitize=address,undefined -fno-gpu-sanitize -fno-omit-frame-pointer -g -O1"
$HIPCC --offload-arch=gfx950 -std=c++17 -fPIC -shared $SAN -shared-libsan -Wno-unused-function \
    -o build/asan/libslamhip_asan.so slam_decomposition_amd/csrc/slam_hip.hip slam_decomposition_amd/csrc/slam_comm.hip -ldl
$HIPCC -std=c++17 $SAN -shared-libsan -x c++ tools/asan_host_driver.cpp -o build/asan/driver -Lbuild/asan -lslamhip_asan -Wl,-rpath,'$ORIGIN' -lpthread
RT=$(dirname "$($HIPCC -print-file-name=libclang_rt.asan-x86_64.so 2>/dev/null || true)")
[ -d "$RT" ] || RT=$(dirname "$(find "$(dirname "$HIPCC")/../lib/llvm" -name 'libclang_rt.asan-x86_64.so' | head -1)")
LD_LIBRARY_PATH="$RT:$LD_LIBRARY_PATH" ASAN_OPTIONS=detect_leaks=0:abort_on_error=1 UBSAN_OPTIONS=halt_on_error=1 HIP_VISIBLE_DEVICES=-1 build/asan/driver
