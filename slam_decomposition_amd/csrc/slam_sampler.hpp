// slam_sampler.hpp -- Haar-random 4x4 unitaries generated on the device.
//
// Reference: HaarSample._get_unitary (src/slam/sampler.py:62-71) = qiskit random_unitary = SciPy's
// unitary_group: Ginibre matrix Z (i.i.d. complex normals) -> QR -> Q diag(R_ii / |R_ii|), i.e. the QR
// factorisation with a positive real diagonal of R (Mezzadri 2007), which is exactly what Gram-Schmidt
// produces.  SciPy draws Z from NumPy's PCG64 + ziggurat, one target at a time on the host (~20-50 us
// each, i.e. 100x slower than the GPU decomposes them); here Z comes from Philox4x32-10 + Box-Muller,
// keyed on (seed, target index), so the SAMPLE differs from SciPy's but the DISTRIBUTION is the same
// (tests/test_gpu_sampler.py: moments, Weyl-coordinate statistics, KAT-4 span fraction).
// oracle/slam_oracle.py:haar_philox_port restates this generator in NumPy.
#pragma once
#include "slam_device.hpp"

namespace slamdev {

// normals 2p, 2p + 1 of target `idx`
__device__ __forceinline__ void philox_normal_pair(uint64_t seed, uint64_t idx, uint32_t p, double& n0, double& n1) {
    uint32_t w[4];
    philox4x32_10(p, 0x48414152u /* "HAAR" */, (uint32_t)idx, (uint32_t)(idx >> 32), (uint32_t)seed, (uint32_t)(seed >> 32), w);
    const uint64_t m0 = ((uint64_t)(w[0] >> 5) << 26) + (uint64_t)(w[1] >> 6);
    const uint64_t m1 = ((uint64_t)(w[2] >> 5) << 26) + (uint64_t)(w[3] >> 6);
    const double u0 = ((double)m0 + 0.5) * (1.0 / 9007199254740992.0);  // (0, 1)
    const double u1 = ((double)m1 + 0.5) * (1.0 / 9007199254740992.0);
    const double r = sqrt(-2.0 * log(u0));
    double s, c;
    sincos(6.283185307179586476925286766559 * u1, &s, &c);
    n0 = r * c;
    n1 = r * s;
}

__global__ void __launch_bounds__(128) haar_targets_kernel(double* targets, int64_t first_index, int64_t n, uint64_t seed) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const uint64_t idx = (uint64_t)(first_index + t);
    // Ginibre matrix, entry (r, c) = normals (2e, 2e + 1), e = 4r + c (scale is irrelevant for Q)
    double zr[4][4], zi[4][4];
#pragma unroll
    for (int e = 0; e < 16; ++e) philox_normal_pair(seed, idx, (uint32_t)e, zr[e >> 2][e & 3], zi[e >> 2][e & 3]);
    // Gram-Schmidt on the columns, two passes per column (CGS2: orthogonal to rounding even for ill-conditioned Z)
#pragma unroll
    for (int c = 0; c < 4; ++c) {
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
#pragma unroll
            for (int p = 0; p < c; ++p) {
                double pr = 0.0, pi = 0.0;  // <q_p, v> = sum conj(q_p[r]) v[r]
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    pr += zr[r][p] * zr[r][c] + zi[r][p] * zi[r][c];
                    pi += zr[r][p] * zi[r][c] - zi[r][p] * zr[r][c];
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    zr[r][c] -= pr * zr[r][p] - pi * zi[r][p];
                    zi[r][c] -= pr * zi[r][p] + pi * zr[r][p];
                }
            }
        }
        double nn = 0.0;
#pragma unroll
        for (int r = 0; r < 4; ++r) nn += zr[r][c] * zr[r][c] + zi[r][c] * zi[r][c];
        const double inv = 1.0 / sqrt(nn);
#pragma unroll
        for (int r = 0; r < 4; ++r) { zr[r][c] *= inv; zi[r][c] *= inv; }
    }
    double2* out = reinterpret_cast<double2*>(targets + t * 32);
#pragma unroll
    for (int e = 0; e < 16; ++e) out[e] = make_double2(zr[e >> 2][e & 3], zi[e >> 2][e & 3]);
}

}  // namespace slamdev
