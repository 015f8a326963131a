// Dev tool (GPU): issue cost of the VALU / LDS instructions the optimizer kernel is made of, on gfx950.
// One 64-lane workgroup per wave; `waves_per_simd` waves share every SIMD.  Each test runs a loop of 64
// independent chains x 8 instructions and reports cycles per wave-instruction as seen by one SIMD
// (s_memtime ticks / instructions issued on that SIMD).
//   hipcc --offload-arch=gfx950 -O3 -o build/ubench_valu tools/ubench_valu.hip && build/ubench_valu
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

typedef float f32x2 __attribute__((ext_vector_type(2)));

enum { T_FMA64, T_MUL64, T_ADD64, T_FMA32, T_PKFMA32, T_MOVDPP, T_CNDMASK, T_MIX64_32, T_LDSR128, T_FMA64_SGPR, T_N };
static const char* kNames[T_N] = {"v_fma_f64", "v_mul_f64", "v_add_f64", "v_fma_f32", "v_pk_fma_f32", "v_mov_b32 dpp", "v_cndmask_b32",
                                  "fma_f64 + fma_f32 alternating", "ds_read_b128", "v_fma_f64 (sgpr operand)"};

template <int T>
__global__ void __launch_bounds__(64) bench(double* out, long long* ticks, int iters, double seed, const double* sg) {
    __shared__ __attribute__((aligned(16))) double lds[64 * 4];
    const int lane = threadIdx.x;
    double a[8], b = seed + lane * 1e-9, c = 1.0 - 1e-9;
    float fa[8], fb = (float)b, fc = 0.999f;
    f32x2 pa[8], pb = {fb, fb}, pc = {fc, fc};
    int ia[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[i] = seed * i; fa[i] = (float)i; pa[i] = f32x2{(float)i, 1.0f}; ia[i] = lane + i; }
    lds[lane] = b; lds[64 + lane] = c; lds[128 + lane] = b; lds[192 + lane] = c;
    __syncthreads();
    const double s0 = sg[0];
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (T == T_FMA64) a[i] = __builtin_fma(a[i], c, b);
                if (T == T_FMA64_SGPR) a[i] = __builtin_fma(a[i], s0, b);
                if (T == T_MUL64) a[i] = a[i] * c;
                if (T == T_ADD64) a[i] = a[i] + b;
                if (T == T_FMA32) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(fa[i]) : "v"(fc), "v"(fb));
                if (T == T_PKFMA32) pa[i] = __builtin_elementwise_fma(pa[i], pc, pb);
                if (T == T_MOVDPP) ia[i] = __builtin_amdgcn_update_dpp(0, ia[i], 0xB1, 0xF, 0xF, true);
                if (T == T_CNDMASK) ia[i] = (ia[(i + 1) & 7] & 1) ? ia[i] : lane;
                if (T == T_MIX64_32) { if (i & 1) a[i] = __builtin_fma(a[i], c, b); else asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(fa[i]) : "v"(fc), "v"(fb)); }
                if (T == T_LDSR128) {
                    double2 v = *reinterpret_cast<const double2*>(&lds[((lane + i + r + (int)a[i]) & 63) * 2]);
                    asm volatile("" : "+v"(v.x), "+v"(v.y));
                    a[i] = v.x * 0.0;
                }
            }
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    double acc = 0.0;
#pragma unroll
    for (int i = 0; i < 8; ++i) acc += a[i] + fa[i] + pa[i].x + pa[i].y + ia[i];
    out[blockIdx.x * 64 + lane] = acc;
    if (lane == 0) ticks[blockIdx.x] = t1 - t0;
}

template <int T>
int run(int waves_per_simd, double* d_out, long long* d_ticks, const double* d_sg) {
    const int blocks = 256 * 4 * waves_per_simd;
    const int iters = 2000;
    hipLaunchKernelGGL(bench<T>, dim3(blocks), dim3(64), 0, 0, d_out, d_ticks, 10, 1.0, d_sg);
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    CHECK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(bench<T>, dim3(blocks), dim3(64), 0, 0, d_out, d_ticks, iters, 1.0, d_sg);
    CHECK(hipEventRecord(e1, 0));
    CHECK(hipDeviceSynchronize());
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<long long> t(blocks);
    CHECK(hipMemcpy(t.data(), d_ticks, blocks * sizeof(long long), hipMemcpyDeviceToHost));
    double mean = 0;
    for (long long v : t) mean += (double)v;
    mean /= blocks;
    const double inst = (double)iters * 64;  // per wave (the LDS test adds one v_add_f64 per read)
    // s_memtime counts at 100 MHz on gfx9 (constant clock): convert with the wall time of the launch
    printf("%-32s %d wave(s)/SIMD: %8.3f ms, %7.2f ns per wave-instruction per SIMD  (s_memtime ticks/inst/wave %.3f)\n", kNames[T], waves_per_simd,
           ms, ms * 1e6 / (inst * waves_per_simd), mean / inst);
    return 0;
}

int main() {
    double* d_out; long long* d_ticks; double* d_sg;
    CHECK(hipMalloc(&d_out, 256 * 4 * 4 * 64 * sizeof(double)));
    CHECK(hipMalloc(&d_ticks, 256 * 4 * 4 * sizeof(long long)));
    CHECK(hipMalloc(&d_sg, 64));
    double one = 1.0 - 1e-9;
    CHECK(hipMemcpy(d_sg, &one, 8, hipMemcpyHostToDevice));
    for (int w = 1; w <= 2; ++w) {
        if (run<T_FMA64>(w, d_out, d_ticks, d_sg)) return 1;
        if (run<T_FMA64_SGPR>(w, d_out, d_ticks, d_sg)) return 1;
        if (run<T_MUL64>(w, d_out, d_ticks, d_sg)) return 1;
        if (run<T_ADD64>(w, d_out, d_ticks, d_sg)) return 1;
        if (run<T_FMA32>(w, d_out, d_ticks, d_sg)) return 1;
        if (run<T_PKFMA32>(w, d_out, d_ticks, d_sg)) return 1;
        if (run<T_MOVDPP>(w, d_out, d_ticks, d_sg)) return 1;
        if (run<T_CNDMASK>(w, d_out, d_ticks, d_sg)) return 1;
        if (run<T_MIX64_32>(w, d_out, d_ticks, d_sg)) return 1;
        if (run<T_LDSR128>(w, d_out, d_ticks, d_sg)) return 1;
    }
    printf("(at 2.4 GHz: 1 cycle = 0.417 ns; v_fma_f64 at the 78.6 TF/s peak = 4 cycles = 1.67 ns per wave-instruction per SIMD)\n");
    return 0;
}
