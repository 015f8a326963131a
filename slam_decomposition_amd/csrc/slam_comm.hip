// slam_comm.hip -- the multi-GPU side of libslamhip.so: one process per GPU, RCCL over xGMI, reached through the
// C ABI (include/slam_hip.h) -- no torch, no MPI.  The path shards by target; the only exchange is
// the final min-all-reduce of the best-loss vector (the running minimum of TemplateOptimizer._run,
// src/slam/optimizer.py:281-284, taken over all ranks) plus a few scalars (barrier, slowest rank's time, counts).
//
// librccl.so is loaded with dlopen the first time a communicator is asked for: a single-GPU process never maps
// its 0.5 GB.  gfx950 only.
#include "../../include/slam_hip.h"

#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <new>

extern "C" __attribute__((visibility("hidden"))) void slam_set_last_error(const char* msg);  // slam_hip.hip: thread-local message of slam_last_error()

namespace {

int cfail(int code, const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    slam_set_last_error(buf);
    return code;
}

#define CHIP_TRY(expr)                                                                                          \
    do {                                                                                                        \
        hipError_t _e = (expr);                                                                                 \
        if (_e != hipSuccess)                                                                                   \
            return cfail(SLAM_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
    } while (0)

struct Rccl {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
    ncclResult_t (*CommUserRank)(const ncclComm_t, int*) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    char why[512] = {0};
};

Rccl g_rccl;
std::once_flag g_rccl_once;

void load_rccl() {
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* n : names) {
        g_rccl.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL);
        if (g_rccl.handle) break;
    }
    if (!g_rccl.handle) {
        snprintf(g_rccl.why, sizeof(g_rccl.why), "dlopen(librccl.so.1): %s", dlerror());
        return;
    }
#define SLAM_SYM(field, name)                                                              \
    g_rccl.field = reinterpret_cast<decltype(g_rccl.field)>(dlsym(g_rccl.handle, name));   \
    if (!g_rccl.field) { snprintf(g_rccl.why, sizeof(g_rccl.why), "librccl has no %s", name); g_rccl.handle = nullptr; return; }
    SLAM_SYM(GetUniqueId, "ncclGetUniqueId")
    SLAM_SYM(CommInitRank, "ncclCommInitRank")
    SLAM_SYM(CommDestroy, "ncclCommDestroy")
    SLAM_SYM(AllReduce, "ncclAllReduce")
    SLAM_SYM(CommCount, "ncclCommCount")
    SLAM_SYM(CommUserRank, "ncclCommUserRank")
    SLAM_SYM(GetErrorString, "ncclGetErrorString")
#undef SLAM_SYM
}

int need_rccl() {
    std::call_once(g_rccl_once, load_rccl);
    if (!g_rccl.handle) return cfail(SLAM_ERR_UNSUPPORTED, "RCCL is not available: %s", g_rccl.why);
    return SLAM_OK;
}

#define RCCL_TRY(expr)                                                                                            \
    do {                                                                                                          \
        ncclResult_t _r = (expr);                                                                                 \
        if (_r != ncclSuccess)                                                                                    \
            return cfail(SLAM_ERR_HIP, "%s failed: %s (%s:%d)", #expr, g_rccl.GetErrorString(_r), __FILE__, __LINE__); \
    } while (0)

__global__ void fill_inf_kernel(double* v, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) v[i] = INFINITY;
}

// dst[i] = min(dst[i], src[i]); NaN in src never wins (a target without a result keeps +inf)
__global__ void min_into_kernel(double* dst, const double* src, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        const double s = src[i];
        if (s < dst[i]) dst[i] = s;
    }
}

__global__ void count_below_kernel(const double* v, int64_t n, double threshold, unsigned long long* out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long c = (i < n && v[i] < threshold) ? 1ull : 0ull;
    for (int off = 32; off > 0; off >>= 1) c += __shfl_down(c, off);
    if ((threadIdx.x & 63) == 0 && c) atomicAdd(out, c);
}

}  // namespace

struct slam_comm {
    int device = 0, rank = 0, world = 1;
    ncclComm_t comm = nullptr;
    hipStream_t stream = nullptr;
    double* d_merge = nullptr;   // [merge_cap] the job-wide best-loss vector of this rank
    int64_t merge_cap = 0, merge_n = 0;
    double* d_small = nullptr;   // staging of slam_comm_allreduce_f64
    int64_t small_cap = 0;
    double* d_stage = nullptr;   // staging of slam_comm_merge_add_host
    int64_t stage_cap = 0;
    unsigned long long* d_count = nullptr;
};

namespace {

int grow(double** p, int64_t* cap, int64_t n) {
    if (n <= *cap) return SLAM_OK;
    if (*p) CHIP_TRY(hipFree(*p));
    *p = nullptr;
    *cap = 0;
    CHIP_TRY(hipMalloc(reinterpret_cast<void**>(p), (size_t)n * sizeof(double)));
    *cap = n;
    return SLAM_OK;
}

}  // namespace

extern "C" {

int slam_comm_get_unique_id(void* id) {
    if (!id) return cfail(SLAM_ERR_INVALID, "id is NULL");
    int rc = need_rccl();
    if (rc) return rc;
    static_assert(sizeof(ncclUniqueId) == SLAM_COMM_ID_BYTES, "ncclUniqueId size");
    ncclUniqueId u;
    RCCL_TRY(g_rccl.GetUniqueId(&u));
    std::memcpy(id, &u, sizeof(u));
    return SLAM_OK;
}

int slam_comm_init(int device, int rank, int world, const void* id, slam_comm** out) {
    if (!out) return cfail(SLAM_ERR_INVALID, "out is NULL");
    *out = nullptr;
    if (!id) return cfail(SLAM_ERR_INVALID, "id is NULL");
    if (world < 1 || rank < 0 || rank >= world) return cfail(SLAM_ERR_INVALID, "bad rank %d / world %d", rank, world);
    int rc = need_rccl();
    if (rc) return rc;
    int n = 0;
    CHIP_TRY(hipGetDeviceCount(&n));
    if (device < 0 || device >= n) return cfail(SLAM_ERR_INVALID, "device %d out of range (%d visible)", device, n);
    CHIP_TRY(hipSetDevice(device));
    slam_comm* c = new (std::nothrow) slam_comm();
    if (!c) return cfail(SLAM_ERR_NOMEM, "out of host memory");
    c->device = device;
    c->rank = rank;
    c->world = world;
    ncclUniqueId u;
    std::memcpy(&u, id, sizeof(u));
    hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&c->d_count), sizeof(unsigned long long));
    if (e != hipSuccess) {
        slam_comm_destroy(c);
        return cfail(SLAM_ERR_HIP, "communicator setup failed: %s", hipGetErrorString(e));
    }
    ncclResult_t r = g_rccl.CommInitRank(&c->comm, world, u, rank);
    if (r != ncclSuccess) {
        c->comm = nullptr;
        slam_comm_destroy(c);
        return cfail(SLAM_ERR_HIP, "ncclCommInitRank(rank %d of %d, device %d) failed: %s", rank, world, device, g_rccl.GetErrorString(r));
    }
    *out = c;
    return SLAM_OK;
}

int slam_comm_destroy(slam_comm* c) {
    if (!c) return SLAM_OK;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->comm && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(c->comm);
    if (c->d_merge) (void)hipFree(c->d_merge);
    if (c->d_small) (void)hipFree(c->d_small);
    if (c->d_stage) (void)hipFree(c->d_stage);
    if (c->d_count) (void)hipFree(c->d_count);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
    return SLAM_OK;
}

int slam_comm_rank(slam_comm* c, int* rank, int* world) {
    if (!c) return cfail(SLAM_ERR_INVALID, "comm is NULL");
    // what RCCL itself reports for this communicator (ncclCommUserRank / ncclCommCount), not what the caller passed in:
    // a job whose ranks did not all join the same communicator shows up here
    int r = -1, w = -1;
    RCCL_TRY(g_rccl.CommUserRank(c->comm, &r));
    RCCL_TRY(g_rccl.CommCount(c->comm, &w));
    if (r != c->rank || w != c->world)
        return cfail(SLAM_ERR_STATE, "RCCL reports rank %d of %d, the communicator was created as rank %d of %d", r, w, c->rank, c->world);
    if (rank) *rank = r;
    if (world) *world = w;
    return SLAM_OK;
}

int slam_comm_allreduce_f64(slam_comm* c, double* inout, int64_t n, int op) {
    if (!c) return cfail(SLAM_ERR_INVALID, "comm is NULL");
    if (n < 0 || (n > 0 && !inout)) return cfail(SLAM_ERR_INVALID, "bad buffer");
    if (op != SLAM_OP_SUM && op != SLAM_OP_MAX && op != SLAM_OP_MIN) return cfail(SLAM_ERR_INVALID, "op must be SLAM_OP_SUM / MAX / MIN");
    if (n == 0) return SLAM_OK;
    CHIP_TRY(hipSetDevice(c->device));
    int rc = grow(&c->d_small, &c->small_cap, n);
    if (rc) return rc;
    CHIP_TRY(hipMemcpyAsync(c->d_small, inout, (size_t)n * sizeof(double), hipMemcpyHostToDevice, c->stream));
    RCCL_TRY(g_rccl.AllReduce(c->d_small, c->d_small, (size_t)n, ncclFloat64, (ncclRedOp_t)op, c->comm, c->stream));
    CHIP_TRY(hipMemcpyAsync(inout, c->d_small, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    CHIP_TRY(hipStreamSynchronize(c->stream));
    return SLAM_OK;
}

int slam_comm_barrier(slam_comm* c) {
    double one = 1.0;
    return slam_comm_allreduce_f64(c, &one, 1, SLAM_OP_SUM);
}

int slam_comm_merge_begin(slam_comm* c, int64_t n_global) {
    if (!c) return cfail(SLAM_ERR_INVALID, "comm is NULL");
    if (n_global <= 0) return cfail(SLAM_ERR_INVALID, "n_global must be > 0");
    CHIP_TRY(hipSetDevice(c->device));
    int rc = grow(&c->d_merge, &c->merge_cap, n_global);
    if (rc) return rc;
    c->merge_n = n_global;
    hipLaunchKernelGGL(fill_inf_kernel, dim3((unsigned)((n_global + 255) / 256)), dim3(256), 0, c->stream, c->d_merge, n_global);
    CHIP_TRY(hipGetLastError());
    return SLAM_OK;
}

static int merge_window(slam_comm* c, const double* d_src, int64_t count, int64_t first_global) {
    if (first_global < 0 || count < 0 || first_global + count > c->merge_n)
        return cfail(SLAM_ERR_INVALID, "window [%lld, %lld) outside the merge vector [0, %lld)", (long long)first_global,
                     (long long)(first_global + count), (long long)c->merge_n);
    if (count == 0) return SLAM_OK;
    hipLaunchKernelGGL(min_into_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, c->stream, c->d_merge + first_global, d_src, count);
    CHIP_TRY(hipGetLastError());
    return SLAM_OK;
}

int slam_comm_merge_add(slam_comm* c, slam_ctx* ctx, int64_t first_local, int64_t count, int64_t first_global) {
    if (!c || !ctx) return cfail(SLAM_ERR_INVALID, "NULL argument");
    if (c->merge_n <= 0) return cfail(SLAM_ERR_STATE, "call slam_comm_merge_begin first");
    void* p = nullptr;
    int64_t n = 0;
    int rc = slam_best_loss_device_ptr(ctx, &p, &n);  // resident best_loss of the context
    if (rc) return rc;
    int ctx_device = -1;
    rc = slam_ctx_device(ctx, &ctx_device);
    if (rc) return rc;
    if (ctx_device != c->device)
        return cfail(SLAM_ERR_INVALID, "context lives on device %d, the communicator on device %d: a rank merges its own GPU's results", ctx_device, c->device);
    if (first_local < 0 || count < 0 || first_local + count > n) return cfail(SLAM_ERR_INVALID, "window outside the context's resident results");
    CHIP_TRY(hipSetDevice(c->device));
    // every API call on the context ends with its stream drained, so its results are complete here ...
    rc = merge_window(c, static_cast<const double*>(p) + first_local, count, first_global);
    if (rc) return rc;
    // ... and the read of the context's buffer is complete when this call returns: the context's next call (which resets
    // the window to +inf on ITS stream) cannot race with it
    CHIP_TRY(hipStreamSynchronize(c->stream));
    return SLAM_OK;
}

int slam_comm_merge_add_host(slam_comm* c, const double* loss, int64_t count, int64_t first_global) {
    if (!c || (!loss && count > 0)) return cfail(SLAM_ERR_INVALID, "NULL argument");
    if (c->merge_n <= 0) return cfail(SLAM_ERR_STATE, "call slam_comm_merge_begin first");
    if (count <= 0) return count == 0 ? SLAM_OK : cfail(SLAM_ERR_INVALID, "count < 0");
    CHIP_TRY(hipSetDevice(c->device));
    CHIP_TRY(hipStreamSynchronize(c->stream));  // the staging buffer may still feed an earlier window's kernel
    int rc = grow(&c->d_stage, &c->stage_cap, count);
    if (rc) return rc;
    CHIP_TRY(hipMemcpyAsync(c->d_stage, loss, (size_t)count * sizeof(double), hipMemcpyHostToDevice, c->stream));
    rc = merge_window(c, c->d_stage, count, first_global);
    if (rc) return rc;
    CHIP_TRY(hipStreamSynchronize(c->stream));  // the caller's array is free again
    return SLAM_OK;
}

int slam_allreduce_min(slam_comm* c, double threshold, int64_t* n_below, double* merged, int64_t merged_capacity) {
    if (!c) return cfail(SLAM_ERR_INVALID, "comm is NULL");
    if (c->merge_n <= 0) return cfail(SLAM_ERR_STATE, "call slam_comm_merge_begin first");
    if (merged && merged_capacity < c->merge_n)
        return cfail(SLAM_ERR_INVALID, "merged holds %lld doubles, the job vector has %lld (slam_comm_merge_begin)", (long long)merged_capacity,
                     (long long)c->merge_n);
    CHIP_TRY(hipSetDevice(c->device));
    RCCL_TRY(g_rccl.AllReduce(c->d_merge, c->d_merge, (size_t)c->merge_n, ncclFloat64, ncclMin, c->comm, c->stream));
    unsigned long long cnt = 0;
    if (n_below) {
        CHIP_TRY(hipMemsetAsync(c->d_count, 0, sizeof(unsigned long long), c->stream));
        hipLaunchKernelGGL(count_below_kernel, dim3((unsigned)((c->merge_n + 255) / 256)), dim3(256), 0, c->stream, c->d_merge, c->merge_n,
                           threshold, c->d_count);
        CHIP_TRY(hipGetLastError());
        CHIP_TRY(hipMemcpyAsync(&cnt, c->d_count, sizeof(cnt), hipMemcpyDeviceToHost, c->stream));
    }
    if (merged) CHIP_TRY(hipMemcpyAsync(merged, c->d_merge, (size_t)c->merge_n * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    CHIP_TRY(hipStreamSynchronize(c->stream));
    if (n_below) *n_below = (int64_t)cnt;
    return SLAM_OK;
}

}  // extern "C"
