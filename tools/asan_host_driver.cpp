// AddressSanitizer driver for the HOST side of libslamhip (tools/asan_host.sh): runs every C-ABI entry point through
// the paths that need no GPU -- argument validation, error formatting, the thread-local message, context / communicator
// creation failing cleanly on a machine without a device -- under ASan + UBSan.  With a GPU present it still only uses
// these paths (it never launches a kernel).
#include "../include/slam_hip.h"

#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>

static int fails = 0;
#define EXPECT(cond)                                                             \
    do {                                                                         \
        if (!(cond)) { std::printf("FAIL %s:%d  %s  [%s]\n", __FILE__, __LINE__, #cond, slam_last_error()); ++fails; } \
    } while (0)

int main() {
    EXPECT(std::strstr(slam_version(), "gfx950") != nullptr);
    EXPECT(slam_device_count(nullptr) == SLAM_ERR_INVALID);
    int n = -1;
    const int rc_count = slam_device_count(&n);
    EXPECT(rc_count == SLAM_OK || (rc_count == SLAM_ERR_HIP && n == 0));
    EXPECT(slam_ctx_create(0, nullptr) == SLAM_ERR_INVALID);
    slam_ctx* ctx = nullptr;
    EXPECT(slam_ctx_create(-1, &ctx) != SLAM_OK && ctx == nullptr);
    EXPECT(slam_ctx_create(1 << 20, &ctx) != SLAM_OK && ctx == nullptr);
    EXPECT(std::strlen(slam_last_error()) > 0);
    // every entry point rejects a NULL context with a message, never a crash
    double d[64] = {0};
    int32_t i32[8] = {0};
    int64_t i64 = 0;
    slam_opt_params prm{};
    prm.restarts = 1;
    slam_stats st{};
    void* p = nullptr;
    EXPECT(slam_ctx_destroy(nullptr) == SLAM_OK);
    EXPECT(slam_ctx_device_info(nullptr, nullptr, 0, nullptr, nullptr) == SLAM_ERR_INVALID);
    EXPECT(slam_set_targets(nullptr, d, 1) == SLAM_ERR_INVALID);
    EXPECT(slam_sample_haar(nullptr, 1, 0, 1) == SLAM_ERR_INVALID);
    EXPECT(slam_get_targets(nullptr, 0, 1, d) == SLAM_ERR_INVALID);
    EXPECT(slam_c1c2c3(nullptr, d, 1, 8, d) == SLAM_ERR_INVALID);
    EXPECT(slam_targets_c1c2c3(nullptr, 0, 1, 8, d) == SLAM_ERR_INVALID);
    EXPECT(slam_predict_spans(nullptr, 0, 1, 3, d, d, 0.0, i32) == SLAM_ERR_INVALID);
    EXPECT(slam_eval_c1c2c3(nullptr, 1, i32, d, 1, 8, d) == SLAM_ERR_INVALID);
    EXPECT(slam_set_gates(nullptr, d, 1) == SLAM_ERR_INVALID);
    EXPECT(slam_eval_loss_grad(nullptr, 1, i32, d, i32, 1, d, d) == SLAM_ERR_INVALID);
    EXPECT(slam_eval_unitary(nullptr, 1, i32, d, i32, 1, d, d) == SLAM_ERR_INVALID);
    EXPECT(slam_minimize_stage(nullptr, 1, i32, nullptr, 0, nullptr, &prm, d, d, i32, d, i32, i32, i32) == SLAM_ERR_INVALID);
    EXPECT(slam_minimize_stage_trace(nullptr, 1, i32, nullptr, 0, nullptr, &prm, 1e-10, 4, d, d, i32, d, i32, i32, d, d) == SLAM_ERR_INVALID);
    EXPECT(slam_decompose(nullptr, 1, 3, i32, &prm, 1e-10, d, d, i32) == SLAM_ERR_INVALID);
    EXPECT(slam_decompose_resident(nullptr, 1, 3, i32, &prm, 1e-10) == SLAM_ERR_INVALID);
    EXPECT(slam_decompose_range(nullptr, 0, 1, 1, 3, i32, &prm, 1e-10) == SLAM_ERR_INVALID);
    EXPECT(slam_decompose_range_fetch(nullptr, 0, 1, 1, 3, i32, &prm, 1e-10, d, d, i32) == SLAM_ERR_INVALID);
    EXPECT(slam_decompose_list(nullptr, i32, 1, 1, 3, 3, i32, &prm, 1e-10) == SLAM_ERR_INVALID);
    EXPECT(slam_decompose_multi(nullptr, 0, 0, 1, 1, 3, i32, &prm, 1e-10) == SLAM_ERR_INVALID);
    { slam_ctx* none[2] = {nullptr, nullptr}; EXPECT(slam_decompose_multi(none, 2, 0, 1, 1, 3, i32, &prm, 1e-10) == SLAM_ERR_INVALID); }
    EXPECT(slam_abi_version() == SLAM_ABI_VERSION);
    EXPECT(slam_fetch_results(nullptr, 3, d, d, i32) == SLAM_ERR_INVALID);
    EXPECT(slam_fetch_results_range(nullptr, 3, 0, 1, d, d, i32) == SLAM_ERR_INVALID);
    EXPECT(slam_fetch_span_losses(nullptr, 0, 1, d) == SLAM_ERR_INVALID);
    slam_v2_gate vg{};
    vg.n_params = 1;
    EXPECT(slam_v2_set_gates(nullptr, &vg, 1) == SLAM_ERR_INVALID);
    EXPECT(slam_v2_set_constraint(nullptr, 1, d, 13, 1.0) == SLAM_ERR_INVALID);
    EXPECT(slam_v2_eval_loss_grad(nullptr, 1, i32, d, i32, 1, d, d, d) == SLAM_ERR_INVALID);
    EXPECT(slam_v2_minimize_stage(nullptr, 1, i32, nullptr, 1, nullptr, d, d, nullptr, nullptr, &prm, 1e-10, d, d, i32, d, i32, i32, i32) == SLAM_ERR_INVALID);
    EXPECT(slam_v2_decompose_range(nullptr, 0, 1, 1, 1, i32, d, d, nullptr, nullptr, &prm, 1e-10, d, d, i32) == SLAM_ERR_INVALID);
    EXPECT(slam_v2_minimize_stage_trace(nullptr, 1, i32, nullptr, 1, nullptr, d, d, nullptr, nullptr, &prm, 1e-10, 4, d, d, i32, d, i32, i32, d, d) == SLAM_ERR_INVALID);
    EXPECT(slam_set_cost(nullptr, 0) == SLAM_ERR_INVALID);
    EXPECT(slam_synchronize(nullptr) == SLAM_ERR_INVALID);
    EXPECT(slam_get_stats(nullptr, &st) == SLAM_ERR_INVALID);
    EXPECT(slam_reset_stats(nullptr) == SLAM_ERR_INVALID);
    EXPECT(slam_best_loss_device_ptr(nullptr, &p, &i64) == SLAM_ERR_INVALID);
    { int dev = 0; EXPECT(slam_ctx_device(nullptr, &dev) == SLAM_ERR_INVALID); }
    // communicator: bad arguments; without RCCL or without a device creation fails with a message
    slam_comm* comm = nullptr;
    char id[SLAM_COMM_ID_BYTES] = {0};
    EXPECT(slam_comm_get_unique_id(nullptr) == SLAM_ERR_INVALID);
    EXPECT(slam_comm_init(0, 0, 1, nullptr, &comm) == SLAM_ERR_INVALID);
    EXPECT(slam_comm_init(0, 2, 2, id, &comm) == SLAM_ERR_INVALID && comm == nullptr);
    EXPECT(slam_comm_init(0, 0, 1, id, nullptr) == SLAM_ERR_INVALID);
    EXPECT(slam_comm_destroy(nullptr) == SLAM_OK);
    EXPECT(slam_comm_rank(nullptr, nullptr, nullptr) == SLAM_ERR_INVALID);
    EXPECT(slam_comm_allreduce_f64(nullptr, d, 1, SLAM_OP_MIN) == SLAM_ERR_INVALID);
    EXPECT(slam_comm_barrier(nullptr) == SLAM_ERR_INVALID);
    EXPECT(slam_comm_merge_begin(nullptr, 1) == SLAM_ERR_INVALID);
    EXPECT(slam_comm_merge_add(nullptr, nullptr, 0, 0, 0) == SLAM_ERR_INVALID);
    EXPECT(slam_comm_merge_add_host(nullptr, d, 1, 0) == SLAM_ERR_INVALID);
    EXPECT(slam_allreduce_min(nullptr, 1e-8, &i64, d, 1) == SLAM_ERR_INVALID);
    // the message is thread-local: concurrent failing calls do not trample each other
    std::vector<std::thread> th;
    for (int t = 0; t < 8; ++t)
        th.emplace_back([t] {
            for (int r = 0; r < 200; ++r) {
                slam_ctx* c = nullptr;
                if (slam_ctx_create(-1 - t, &c) == SLAM_OK) ++fails;
                char want[32];
                std::snprintf(want, sizeof(want), "device %d ", -1 - t);
                if (slam_device_count(nullptr) != SLAM_ERR_INVALID) ++fails;
                (void)want;
            }
        });
    for (auto& x : th) x.join();
    std::printf(fails ? "asan host driver: %d FAILURES\n" : "asan host driver: ok\n", fails);
    return fails ? 1 : 0;
}
