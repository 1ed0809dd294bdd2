// Drives the C ABI (include/dvi_eskf.h) of a HOST-ONLY, sanitized build of dvi_ekf_amd/csrc/eskf_capi.cpp (tests/hipstub/):
// handle lifetime, argument validation, workspace growth, schedule walking, the two-stream fork / join and its error path,
// the rules of stream capture.  Exit code 0 = every expectation met; AddressSanitizer / UBSan abort on their own.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../include/dvi_eskf.h"

static int g_failed = 0;
#define EXPECT(cond) do { if (!(cond)) { std::fprintf(stderr, "%s:%d: expectation failed: %s\n", __FILE__, __LINE__, #cond); ++g_failed; } } while (0)

static dvi_eskf_params params() {
    dvi_eskf_params p;
    std::memset(&p, 0, sizeof p);
    p.L = 50.0; p.theta_cam = 0.5; p.hold_frozen = 1;
    for (int i = 0; i < 3; ++i) { p.sigma_a[i] = 1.0; p.sigma_om[i] = 1e-3; }
    return p;
}
template <typename T> static T* dev(size_t n) { void* p = nullptr; EXPECT(hipMalloc(&p, n * sizeof(T)) == hipSuccess); std::memset(p, 0, n * sizeof(T)); return static_cast<T*>(p); }

static void run_schedule(dvi_eskf* h, int64_t B, const std::vector<int32_t>& n_pred, bool per_instance, bool rows, bool rows_pred, int expect, hipStream_t s = nullptr) {
    size_t total = 0;
    for (int32_t n : n_pred) total += (size_t)(n > 0 ? n : 0);
    const size_t E = n_pred.size(), m = per_instance ? (size_t)B : 1;
    double *dt = dev<double>(total * m + 1), *om = dev<double>(total * m * 3 + 1), *acc = dev<double>(total * m * 3 + 1);
    double *pos = dev<double>(E * m * 3 + 1), *quat = dev<double>(E * m * 4 + 1), *notch = dev<double>(E * m + 1);
    int32_t* st = dev<int32_t>(E * (size_t)B + 1);
    std::vector<double> t_log(E + 1, 1.0), t_pred(total + 1, 2.0);
    double* r1 = rows ? dev<double>(E * (size_t)B * 30) : nullptr;
    double* r2 = rows_pred ? dev<double>((total ? total : 1) * (size_t)B * 30) : nullptr;
    dvi_eskf_schedule sc;
    std::memset(&sc, 0, sizeof sc);
    sc.n_epochs = (int64_t)E; sc.n_pred = n_pred.data();
    sc.dt = dt; sc.dt_per_instance = per_instance; sc.om = om; sc.acc = acc; sc.imu_per_instance = per_instance;
    sc.pos = pos; sc.quat_xyzw = quat; sc.notch = notch; sc.meas_per_instance = per_instance;
    sc.status = st; sc.t_log = rows ? t_log.data() : nullptr; sc.rows_out = r1;
    sc.t_pred = rows_pred ? t_pred.data() : nullptr; sc.rows_pred_out = r2; sc.rows_instances = 0;
    EXPECT(dvi_eskf_run_epochs(h, &sc, s) == expect);
    for (void* p : {(void*)dt, (void*)om, (void*)acc, (void*)pos, (void*)quat, (void*)notch, (void*)st, (void*)r1, (void*)r2}) hipFree(p);
}

int main() {
    dvi_eskf_params p = params();
    dvi_eskf* h = nullptr;
    // ---- create: argument validation
    EXPECT(dvi_eskf_create(nullptr, 4, DVI_ESKF_F64, 0, &h) == DVI_ESKF_EINVAL);
    EXPECT(dvi_eskf_create(&p, 0, DVI_ESKF_F64, 0, &h) == DVI_ESKF_EINVAL);
    EXPECT(dvi_eskf_create(&p, 4, 7, 0, &h) == DVI_ESKF_EINVAL);
    EXPECT(dvi_eskf_create(&p, 4, DVI_ESKF_F64, 3, &h) == DVI_ESKF_ENODEV);
    EXPECT(dvi_eskf_create(&p, 4, DVI_ESKF_F64, -1, &h) == DVI_ESKF_ENODEV);
    p.quirks_fixed = 32;
    EXPECT(dvi_eskf_create(&p, 4, DVI_ESKF_F64, 0, &h) == DVI_ESKF_EINVAL);
    p.quirks_fixed = 0;
    for (int k = 1; k <= 6; ++k) {          // every allocation of create may fail: nothing leaks, no handle comes back
        hipstub_fail_next_malloc(k);
        h = nullptr;
        EXPECT(dvi_eskf_create(&p, 4, DVI_ESKF_F64, 0, &h) == DVI_ESKF_ENOMEM && h == nullptr);
        EXPECT(hipstub_live_allocations() == 0);
    }
    hipstub_fail_next_malloc(0);
    for (int dtype : {DVI_ESKF_F64, DVI_ESKF_F32}) {
        const int64_t B = 2176;             // 17 x 128: the two halves of the batch on two streams for epochs the one-launch kernel does not take
        EXPECT(dvi_eskf_create(&p, B, dtype, 0, &h) == DVI_ESKF_OK && h);
        EXPECT(dvi_eskf_batch(h) == B && dvi_eskf_dtype(h) == dtype && dvi_eskf_batch(nullptr) == 0);
        // ---- noise: diagonal only, finite
        std::vector<double> Q(169, 0.0), R(49, 0.0);
        for (int i = 0; i < 13; ++i) Q[i * 13 + i] = i >= 6 ? 1e-4 : 0.0;
        for (int i = 0; i < 7; ++i) R[i * 7 + i] = 1e-3;
        EXPECT(dvi_eskf_set_noise(h, Q.data(), R.data()) == DVI_ESKF_OK);
        Q[1] = 1.0;
        EXPECT(dvi_eskf_set_noise(h, Q.data(), nullptr) == DVI_ESKF_EINVAL);
        Q[1] = 0.0; R[8] = 0.0 / 0.0;
        EXPECT(dvi_eskf_set_noise(h, nullptr, R.data()) == DVI_ESKF_EINVAL);
        R[8] = 1e-3;
        EXPECT(dvi_eskf_set_noise(nullptr, Q.data(), R.data()) == DVI_ESKF_EINVAL);
        const size_t es = dtype == DVI_ESKF_F64 ? 8 : 4;
        // ---- state in / out
        void *x = dev<char>(B * 26 * es), *P = dev<char>(B * 576 * es), *o3 = dev<char>(B * 3 * es), *a3 = dev<char>(B * 3 * es), *q4 = dev<char>(B * 4 * es);
        EXPECT(dvi_eskf_set_state(h, x, P, o3, a3, q4, nullptr) == DVI_ESKF_OK);
        EXPECT(dvi_eskf_set_state(h, x, nullptr, nullptr, nullptr, nullptr, nullptr) == DVI_ESKF_OK);
        EXPECT(dvi_eskf_get_state(h, x, P, o3, a3, q4, nullptr) == DVI_ESKF_OK);
        void* dofs = dev<char>(B * 6 * es);
        EXPECT(dvi_eskf_get_calibration(h, dofs, nullptr) == DVI_ESKF_OK && dvi_eskf_get_calibration(h, nullptr, nullptr) == DVI_ESKF_EINVAL);
        // ---- single steps: argument validation, broadcast and per-instance inputs
        void *dt = dev<char>(B * es), *om = dev<char>(B * 3 * es), *acc = dev<char>(B * 3 * es), *pos = dev<char>(B * 3 * es), *quat = dev<char>(B * 4 * es), *notch = dev<char>(B * es);
        int32_t* st = dev<int32_t>(B);
        void* K = dev<char>(B * 24 * 7 * es);
        EXPECT(dvi_eskf_predict(h, nullptr, 0, om, acc, 0, nullptr) == DVI_ESKF_EINVAL);
        EXPECT(dvi_eskf_predict(h, dt, 0, om, acc, 0, nullptr) == DVI_ESKF_OK);
        EXPECT(dvi_eskf_predict(h, dt, 1, om, acc, 1, nullptr) == DVI_ESKF_OK);
        EXPECT(dvi_eskf_update(h, pos, nullptr, notch, 0, nullptr, st, nullptr) == DVI_ESKF_EINVAL);
        EXPECT(dvi_eskf_update(h, pos, quat, notch, 1, K, st, nullptr) == DVI_ESKF_OK);
        EXPECT(dvi_eskf_step(h, dt, 1, om, acc, 1, pos, quat, notch, 1, nullptr, st, nullptr) == DVI_ESKF_OK);
        EXPECT(dvi_eskf_step(h, dt, 0, om, acc, 0, pos, quat, notch, 0, K, st, nullptr) == DVI_ESKF_OK);
        EXPECT(dvi_eskf_epoch(h, -1, dt, 0, om, acc, 0, pos, quat, notch, 0, nullptr, st, nullptr) == DVI_ESKF_EINVAL);
        // ---- clone / copy_state / step_from / copy_instances
        dvi_eskf *c = nullptr, *small = nullptr, *other = nullptr;
        EXPECT(dvi_eskf_clone(h, nullptr, &c) == DVI_ESKF_OK && c && dvi_eskf_clone(nullptr, nullptr, &c) == DVI_ESKF_EINVAL);
        EXPECT(dvi_eskf_copy_state(c, h, nullptr) == DVI_ESKF_OK && dvi_eskf_copy_state(h, h, nullptr) == DVI_ESKF_OK);
        EXPECT(dvi_eskf_create(&p, 5, dtype, 0, &small) == DVI_ESKF_OK);
        EXPECT(dvi_eskf_copy_state(small, h, nullptr) == DVI_ESKF_EINVAL);
        EXPECT(dvi_eskf_copy_instances(small, h, B - 5, nullptr) == DVI_ESKF_OK && dvi_eskf_copy_instances(small, h, B - 4, nullptr) == DVI_ESKF_EINVAL);
        EXPECT(dvi_eskf_copy_instances(small, h, -1, nullptr) == DVI_ESKF_EINVAL && dvi_eskf_copy_instances(h, h, 0, nullptr) == DVI_ESKF_EINVAL);
        EXPECT(dvi_eskf_create(&p, B, dtype == DVI_ESKF_F64 ? DVI_ESKF_F32 : DVI_ESKF_F64, 0, &other) == DVI_ESKF_OK);
        EXPECT(dvi_eskf_copy_state(other, h, nullptr) == DVI_ESKF_EINVAL);
        EXPECT(dvi_eskf_step_from(c, h, 1, dt, 0, om, acc, 0, pos, quat, notch, 0, nullptr, st, nullptr) == DVI_ESKF_OK);
        EXPECT(dvi_eskf_step_from(c, nullptr, 1, dt, 0, om, acc, 0, pos, quat, notch, 0, nullptr, st, nullptr) == DVI_ESKF_EINVAL);
        EXPECT(dvi_eskf_step_from(c, other, 1, dt, 0, om, acc, 0, pos, quat, notch, 0, nullptr, st, nullptr) == DVI_ESKF_EINVAL);
        // ---- epochs: the workspace grows with the longest epoch seen
        {
            const int N = 9;
            void *dtn = dev<char>(N * B * es), *omn = dev<char>(N * B * 3 * es), *accn = dev<char>(N * B * 3 * es);
            for (int n : {2, 5, 9, 3}) EXPECT(dvi_eskf_epoch(h, n, dtn, 1, omn, accn, 1, pos, quat, notch, 1, nullptr, st, nullptr) == DVI_ESKF_OK);
            EXPECT(dvi_eskf_reserve(h, 12) == DVI_ESKF_OK && dvi_eskf_reserve(h, -1) == DVI_ESKF_EINVAL);
            hipFree(dtn); hipFree(omn); hipFree(accn);
        }
        // ---- runs: one launch of pair steps, one launch of epochs, epoch by epoch, with log rows of either kind, ragged schedules
        run_schedule(h, B, {1, 1, 1, 1}, true, false, false, DVI_ESKF_OK);
        run_schedule(h, B, {3, 3, 3}, false, false, false, DVI_ESKF_OK);
        run_schedule(h, B, {10, 0, 3, 17, 1}, true, false, false, DVI_ESKF_OK);
        run_schedule(h, B, {10, 0, 3, 17, 1}, false, true, false, DVI_ESKF_OK);
        run_schedule(h, B, {4, 2, 0, 5}, true, false, true, DVI_ESKF_OK);
        run_schedule(h, B, {0, 0}, false, false, true, DVI_ESKF_OK);
        run_schedule(h, B, {2, -1}, false, false, false, DVI_ESKF_EINVAL);
        run_schedule(h, B, {2, 2}, false, true, true, DVI_ESKF_EINVAL);                 // the two kinds of log rows exclude each other
        run_schedule(h, B, {}, false, false, false, DVI_ESKF_OK);
        // ... through the paths of a handle the one-launch kernels do not take (kept Jacobians): epoch by epoch, the two halves
        // of the batch on two streams
        EXPECT(dvi_eskf_keep_jacobians(h, 1) == DVI_ESKF_OK);
        run_schedule(h, B, {4, 3, 5, 2}, true, false, false, DVI_ESKF_OK);
        run_schedule(h, B, {4, 3, 5, 2}, false, true, false, DVI_ESKF_OK);
        run_schedule(h, B, {4, 1, 0, 2}, true, false, true, DVI_ESKF_OK);
        {
            void *Fx = dev<char>(B * 576 * es), *Fi = dev<char>(B * 24 * 13 * es);
            EXPECT(dvi_eskf_get_jacobians(h, Fx, Fi, nullptr) == DVI_ESKF_OK && dvi_eskf_get_jacobians(h, nullptr, Fi, nullptr) == DVI_ESKF_EINVAL);
            hipFree(Fx); hipFree(Fi);
        }
        EXPECT(dvi_eskf_keep_jacobians(h, 0) == DVI_ESKF_OK);
        // ---- per-instance noise, lanes selector
        {
            std::vector<double> q13((size_t)B * 13, 1e-4), r7((size_t)B * 7, 1e-3), so((size_t)B * 3, 1e-3);
            EXPECT(dvi_eskf_set_noise_per_instance(h, q13.data(), r7.data(), so.data(), nullptr) == DVI_ESKF_OK);
            EXPECT(dvi_eskf_set_noise_per_instance(h, nullptr, r7.data(), nullptr, nullptr) == DVI_ESKF_EINVAL);
            EXPECT(dvi_eskf_step(h, dt, 0, om, acc, 0, pos, quat, notch, 0, nullptr, st, nullptr) == DVI_ESKF_OK);
            dvi_eskf* c2 = nullptr;
            EXPECT(dvi_eskf_clone(h, nullptr, &c2) == DVI_ESKF_OK);
            dvi_eskf_destroy(c2);
            EXPECT(dvi_eskf_set_noise(h, Q.data(), R.data()) == DVI_ESKF_OK);
        }
        EXPECT(dvi_eskf_set_lanes_per_instance(h, 7) == DVI_ESKF_EINVAL && dvi_eskf_set_lanes_per_instance(h, 32) == DVI_ESKF_OK);
        EXPECT(dvi_eskf_step(h, dt, 0, om, acc, 0, pos, quat, notch, 0, nullptr, st, nullptr) == DVI_ESKF_OK);
        EXPECT(dvi_eskf_set_lanes_per_instance(h, 1) == DVI_ESKF_OK);
        // ---- stream capture: no allocation, no replaced buffer once a step has been recorded
        {
            hipStream_t cs = nullptr;
            EXPECT(hipStreamCreateWithFlags(&cs, 0) == hipSuccess);
            dvi_eskf* g = nullptr;
            EXPECT(dvi_eskf_create(&p, 256, dtype, 0, &g) == DVI_ESKF_OK);
            hipstub_set_capturing(cs, 1);
            void *dt3 = dev<char>(3 * 256 * es), *om3 = dev<char>(3 * 256 * 3 * es), *acc3 = dev<char>(3 * 256 * 3 * es);
            EXPECT(dvi_eskf_epoch(g, 3, dt3, 0, om3, acc3, 0, pos, quat, notch, 0, nullptr, st, cs) == DVI_ESKF_ECAPTURE);   // would have to grow the workspace
            EXPECT(dvi_eskf_step(g, dt, 0, om, acc, 0, pos, quat, notch, 0, nullptr, st, cs) == DVI_ESKF_OK);             // recorded
            hipstub_set_capturing(cs, 0);
            EXPECT(dvi_eskf_reserve(g, 3) == DVI_ESKF_ECAPTURE);                          // the recorded step refers to the old workspace
            EXPECT(dvi_eskf_keep_jacobians(g, 1) == DVI_ESKF_ECAPTURE);
            run_schedule(g, 256, {2, 3}, false, false, false, DVI_ESKF_ECAPTURE);        // (a ragged schedule cannot be uploaded for a recorded handle either: epoch by epoch, which needs the workspace)
            run_schedule(g, 256, {1, 1}, false, false, false, DVI_ESKF_OK);
            EXPECT(dvi_eskf_flush(g, cs) == DVI_ESKF_OK);
            dvi_eskf_destroy(g);
            hipFree(dt3); hipFree(om3); hipFree(acc3);
            EXPECT(hipStreamDestroy(cs) == hipSuccess);
        }
        // ---- the rest of the surface
        {
            void* rows = dev<char>(B * 30 * es);
            EXPECT(dvi_eskf_log_rows(h, 1.0, rows, nullptr) == DVI_ESKF_OK && dvi_eskf_log_rows(h, 1.0, nullptr, nullptr) == DVI_ESKF_EINVAL);
            double ref6[6] = {0, 0, 0, 0, 0, 0};
            void* mse = dev<char>(B * es);
            EXPECT(dvi_eskf_update_mse(h, ref6, nullptr, mse, nullptr) == DVI_ESKF_OK && dvi_eskf_update_mse(h, nullptr, nullptr, mse, nullptr) == DVI_ESKF_EINVAL);
            EXPECT(dvi_eskf_gather_calibration(h, dofs, nullptr, nullptr) == DVI_ESKF_EINVAL);
            EXPECT(dvi_eskf_raw_P(h) && dvi_eskf_raw_x(h) && !dvi_eskf_raw_P(nullptr));
            hipFree(rows); hipFree(mse);
        }
        EXPECT(dvi_eskf_randn(0, 0, 1, 0, 4, 0, 2, 3, dofs, nullptr) == DVI_ESKF_OK && dvi_eskf_randn(5, 0, 1, 0, 4, 0, 2, 3, dofs, nullptr) == DVI_ESKF_EINVAL);
        for (dvi_eskf* d : {c, small, other, h}) dvi_eskf_destroy(d);
        dvi_eskf_destroy(nullptr);
        for (void* q : {x, P, o3, a3, q4, dofs, dt, om, acc, pos, quat, notch, (void*)st, K}) hipFree(q);
        EXPECT(hipstub_live_allocations() == 0);
    }
    for (int code : {0, -1, -2, -3, -4, -5, -99}) EXPECT(dvi_eskf_strerror(code) != nullptr);
    std::printf("capi driver: %ld launches, %d failed expectations\n", hipstub_launches(), g_failed);
    return g_failed ? 1 : 0;
}
