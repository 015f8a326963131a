#!/bin/bash
# Dev tool (CPU): compile ONE instantiation of the optimizer kernel and print its resource usage and instruction mix.
# usage: tools/kcompile.sh <K> <GC> [extra hipcc flags]     (GC: 0 dense, 1 xgen, 2 xri, 3 cx, 4 xri1 = RiSwap family)
K=${1:-3}; GC=${2:-2}; shift 2
mkdir -p build/kc
cat > build/kc/one.hip <<SRC
#include "../../slam_decomposition_amd/csrc/slam_kernels.hpp"
template __global__ void slamdev::minimize_kernel<$K, $GC, ${MQ:-false}>(slamdev::MinimizeArgs<$K>, const slamdev::MinimizeArgs<$K>*, int);
SRC
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -c --cuda-device-only -save-temps=obj -Rpass-analysis=kernel-resource-usage "$@" -o build/kc/one.o build/kc/one.hip 2> build/kc/usage.txt || { cat build/kc/usage.txt | head -30; exit 1; }
grep -E "VGPRs:|ScratchSize|VGPRs Spill|SGPRs Spill|Occupancy" build/kc/usage.txt | sed 's/.*:     //; s/ \[-Rpass.*//' | tr '\n' ';'; echo
S=build/kc/one-hip-amdgcn-amd-amdhsa-gfx950.s
for pat in v_fma_f64 v_mul_f64 v_add_f64 v_pk_fma_f32 v_pk_mul_f32 v_cndmask _dpp "v_mov_b32 " scratch_load scratch_store ds_read ds_write s_load v_readlane v_writelane "v_cvt" "v_lshl\|v_and_b32\|v_add_u32\|v_or_b32\|v_mul_lo\|v_mad_u\|v_add3\|v_lshl_add\|v_mul_u32\|v_mul_hi"; do
  printf "%-14s %s;  " "${pat:0:14}" "$(grep -c -- "$pat" $S)"
done; echo; echo "total instr lines: $(grep -cE '^\s+(v_|s_|ds_|global_|scratch_|buffer_)' $S)  valu: $(grep -cE '^\s+v_' $S)"
