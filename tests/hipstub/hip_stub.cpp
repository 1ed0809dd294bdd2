// Host-only stand-ins behind tests/hipstub/hip/hip_runtime.h and for the kernel launchers of eskf_kernels.hip /
// eskf_epochs.hip / eskf_noise.hip (declared in dvi_ekf_amd/csrc/eskf_device.h).  "Device" memory is heap memory; a launcher
// TOUCHES every buffer the real kernels would touch, over the extent the real kernels would touch it -- so that an
// undersized workspace, a stale pointer after a re-allocation or a wrong stride is an AddressSanitizer report -- and
// computes nothing.  TEST INFRASTRUCTURE (tests/test_capi_sanitized.py), never linked into the product.
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <cstring>
#include <set>

#include "../../dvi_ekf_amd/csrc/eskf_device.h"

struct hipstub_stream { int capturing; };
struct hipstub_event { int dummy; };
static long g_live = 0, g_launches = 0;
static int g_fail_malloc = 0;
static std::set<hipstub_stream*> g_capturing;

extern "C" {
hipError_t hipGetDeviceCount(int* n) { *n = 1; return hipSuccess; }
hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
hipError_t hipSetDevice(int d) { return d == 0 ? hipSuccess : hipErrorInvalidValue; }
hipError_t hipMalloc(void** p, size_t bytes) {
    if (g_fail_malloc > 0 && --g_fail_malloc == 0) { *p = nullptr; return hipErrorOutOfMemory; }
    *p = std::malloc(bytes ? bytes : 1);
    if (!*p) return hipErrorOutOfMemory;
    ++g_live;
    return hipSuccess;
}
hipError_t hipFree(void* p) { if (p) { std::free(p); --g_live; } return hipSuccess; }
hipError_t hipMallocAsync(void** p, size_t bytes, hipStream_t) { return hipMalloc(p, bytes); }
hipError_t hipFreeAsync(void* p, hipStream_t) { return hipFree(p); }
hipError_t hipMemsetAsync(void* p, int value, size_t bytes, hipStream_t) { std::memset(p, value, bytes); return hipSuccess; }
hipError_t hipMemcpyAsync(void* dst, const void* src, size_t bytes, hipMemcpyKind, hipStream_t) { std::memmove(dst, src, bytes); return hipSuccess; }
hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
hipError_t hipDeviceSynchronize(void) { return hipSuccess; }
hipError_t hipStreamIsCapturing(hipStream_t s, hipStreamCaptureStatus* st) {
    *st = (s && g_capturing.count(s)) ? hipStreamCaptureStatusActive : hipStreamCaptureStatusNone;
    return hipSuccess;
}
hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = new hipstub_stream{0}; ++g_live; return hipSuccess; }
hipError_t hipStreamDestroy(hipStream_t s) { g_capturing.erase(s); delete s; --g_live; return hipSuccess; }
hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { *e = new hipstub_event{0}; ++g_live; return hipSuccess; }
hipError_t hipEventDestroy(hipEvent_t e) { delete e; --g_live; return hipSuccess; }
hipError_t hipEventRecord(hipEvent_t e, hipStream_t) { return e ? hipSuccess : hipErrorInvalidValue; }
hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t e, unsigned) { return e ? hipSuccess : hipErrorInvalidValue; }
void hipstub_set_capturing(hipStream_t s, int on) { if (on) g_capturing.insert(s); else g_capturing.erase(s); }
void hipstub_fail_next_malloc(int n) { g_fail_malloc = n; }
long hipstub_live_allocations(void) { return g_live; }
long hipstub_launches(void) { return g_launches; }
}

namespace eskf {
namespace {
size_t es_of(int dtype) { return dtype == 0 ? 8 : 4; }
// read / write every byte of [p, p + n): what a kernel over that extent does as far as the sanitizer is concerned
void rd(const void* p, size_t n) {
    if (!p || !n) return;
    volatile unsigned char acc = 0;
    const unsigned char* c = static_cast<const unsigned char*>(p);
    for (size_t i = 0; i < n; i += 64) acc ^= c[i];
    acc ^= c[n - 1];
    (void)acc;
}
void wr(void* p, size_t n) { if (p && n) std::memset(p, 0, n); }
void touch_step(int dtype, const StepArgs& a, bool with_J) {
    const size_t es = es_of(dtype), B = (size_t)a.B;
    const size_t first = (size_t)a.sub_first, cnt = a.sub_count ? (size_t)a.sub_count : B;
    rd(static_cast<const char*>(a.P_src) + first * 300 * es, cnt * 300 * es);
    rd(a.xs_src, 36 * B * es);
    wr(static_cast<char*>(a.P) + first * 300 * es, cnt * 300 * es);
    wr(a.xs, 36 * B * es);
    const size_t n = (size_t)(a.n_pred > 0 ? a.n_pred : 0);
    if (n) {
        rd(a.dt, (a.dt_stride ? B : 1) * n * es);
        rd(a.om, (a.imu_stride ? B : 1) * n * 3 * es);
        rd(a.acc, (a.imu_stride ? B : 1) * n * 3 * es);
    }
    if (a.do_update) {
        const size_t m = a.meas_stride ? B : 1;
        rd(a.pos, m * 3 * es); rd(a.quat, m * 4 * es); rd(a.notch, m * es);
        if (a.status) wr(a.status, B * sizeof(int32_t));
        if (a.K_out) wr(a.K_out, B * 24 * 7 * es);
    }
    if (with_J) {
        wr(a.J, (n ? n : 1) * B * JAC_WORDS * es);
        if (n >= 2 && a.aux) wr(a.aux, (AUX_WORDS0 + AUX_WORDSK * n) * B * es);
    }
    wr(a.e, B * 24 * es);
    wr(a.status_ws, B * sizeof(int32_t));
    if (a.noise) rd(a.noise, B * NOISE_WORDS * es);
    ++g_launches;
}
}  // namespace

int run_kernel_block(int, int lpi, const StepArgs& a, const HostParams& hp) {
    return (lpi == 1 && a.n_pred == 1 && !a.noise && !a.K_out && !hp.keep_jac && (hp.quirks & QK_Q1)) ? 64 : 0;
}
int step_kernel_block(int dtype, int lpi, const StepArgs& a, const HostParams& hp) { return run_kernel_block(dtype, lpi, a, hp); }
hipError_t launch_step_fused(int dtype, const StepArgs& a, const HostParams&, hipStream_t) { touch_step(dtype, a, false); return hipSuccess; }
hipError_t launch_run(int dtype, int, const StepArgs& a0, const RunArgs& ra, const HostParams&, hipStream_t) {
    for (int k = 0; k < ra.n_steps; ++k) {
        StepArgs a = a0;
        a.dt = static_cast<const char*>(a0.dt) + k * ra.dt_step;
        a.om = static_cast<const char*>(a0.om) + k * ra.imu_step;
        a.acc = static_cast<const char*>(a0.acc) + k * ra.imu_step;
        a.pos = static_cast<const char*>(a0.pos) + k * ra.pos_step;
        a.quat = static_cast<const char*>(a0.quat) + k * ra.quat_step;
        a.notch = static_cast<const char*>(a0.notch) + k * ra.notch_step;
        if (a.status) a.status = a0.status + k * ra.status_step;
        if (k > 0) { a.P_src = a.P; a.xs_src = a.xs; }
        touch_step(dtype, a, false);
    }
    return hipSuccess;
}
int epochs_kernel_block(int, int lpi, const StepArgs& a, const HostParams& hp) {
    return (lpi == 1 && !a.noise && !a.K_out && !hp.keep_jac && (hp.quirks & QK_Q1) && a.B <= 6144) ? 4 : 0;
}
hipError_t launch_epochs(int dtype, int, const EpochsArgs& ea, const HostParams&, hipStream_t) {
    const size_t es = es_of(dtype), B = (size_t)ea.a.B;
    if (ea.n_pred) rd(ea.n_pred, (size_t)ea.n_epochs * sizeof(int32_t));
    size_t g = 0;
    for (int e = 0; e < ea.n_epochs; ++e) {
        const int n = ea.n_pred ? ea.n_pred[e] : ea.n_uniform;
        StepArgs a = ea.a;
        a.n_pred = n;
        a.dt = static_cast<const char*>(ea.a.dt) + g * (a.dt_stride ? B : 1) * es;
        a.om = static_cast<const char*>(ea.a.om) + g * (a.imu_stride ? B : 1) * 3 * es;
        a.acc = static_cast<const char*>(ea.a.acc) + g * (a.imu_stride ? B : 1) * 3 * es;
        const size_t m = a.meas_stride ? B : 1;
        a.pos = static_cast<const char*>(ea.a.pos) + (size_t)e * m * 3 * es;
        a.quat = static_cast<const char*>(ea.a.quat) + (size_t)e * m * 4 * es;
        a.notch = static_cast<const char*>(ea.a.notch) + (size_t)e * m * es;
        if (a.status) a.status = ea.a.status + (size_t)e * B;
        if (e > 0) { a.P_src = a.P; a.xs_src = a.xs; }
        touch_step(dtype, a, false);
        g += (size_t)n;
    }
    if (g != (size_t)ea.n_total) std::abort();
    if (ea.xlog_mode) wr(ea.xlog, (size_t)(ea.xlog_mode == 1 ? ea.n_total : ea.n_epochs) * 26 * (size_t)ea.n_log * es);
    return hipSuccess;
}
hipError_t launch_log_rows_run(int dtype, int64_t n_rows, int64_t n_log, const void* xlog, const void* xs, int64_t B, const double* t_rows, void* rows, hipStream_t) {
    const size_t es = es_of(dtype);
    rd(xlog, (size_t)n_rows * 26 * (size_t)n_log * es); rd(xs, 36 * (size_t)B * es); rd(t_rows, (size_t)n_rows * sizeof(double));
    wr(rows, (size_t)n_rows * (size_t)n_log * 30 * es);
    ++g_launches;
    return hipSuccess;
}
hipError_t launch_step(int dtype, int lpi, const StepArgs& a, const HostParams& hp, hipStream_t s) {
    if (a.sub_count && !epoch_sub_batches_ok(lpi, a.sub_first, a.sub_count, a.n_pred, a.noise != nullptr, a.K_out != nullptr, a.aux != nullptr))
        return hipErrorInvalidValue;
    if (step_kernel_block(dtype, lpi, a, hp)) return launch_step_fused(dtype, a, hp, s);
    touch_step(dtype, a, true);
    return hipSuccess;
}
hipError_t launch_flush(int dtype, const StepArgs& a, const HostParams&, hipStream_t) {
    const size_t es = es_of(dtype), B = (size_t)a.B;
    rd(a.e, B * 24 * es); wr(a.status_ws, B * sizeof(int32_t)); wr(a.xs, 36 * B * es);
    ++g_launches;
    return hipSuccess;
}
bool epoch_sub_batches_ok(int lpi, int64_t sub_first, int64_t sub_count, int n_pred, bool pin, bool gain_out, bool have_aux) {
    return have_aux && lpi == 1 && !pin && !gain_out && n_pred >= 2 && sub_count > 0 && (sub_count & 63) == 0 && (sub_first & 63) == 0;
}
hipError_t launch_copy_state(const void* srcP, void* dstP, size_t bytesP, const void* srcx, void* dstx, size_t bytesx, hipStream_t) {
    std::memmove(dstP, srcP, bytesP); std::memmove(dstx, srcx, bytesx); ++g_launches; return hipSuccess;
}
hipError_t launch_copy_instances(int dtype, const void* srcP, const void* srcx, int64_t srcB, int64_t first, void* dstP, void* dstx, int64_t dstB, hipStream_t) {
    const size_t es = es_of(dtype);
    rd(static_cast<const char*>(srcP) + (size_t)first * 300 * es, (size_t)dstB * 300 * es); rd(srcx, 36 * (size_t)srcB * es);
    wr(dstP, (size_t)dstB * 300 * es); wr(dstx, 36 * (size_t)dstB * es);
    ++g_launches;
    return hipSuccess;
}
hipError_t launch_pack(int dtype, int64_t B, const void* x26, const void* P576, const void* om3, const void* acc3, const void* qo4, void* xs, void* Ppk, hipStream_t) {
    const size_t es = es_of(dtype), b = (size_t)B;
    rd(x26, b * 26 * es); rd(P576, b * 576 * es); rd(om3, b * 3 * es); rd(acc3, b * 3 * es); rd(qo4, b * 4 * es);
    wr(xs, 36 * b * es); if (P576) wr(Ppk, b * 300 * es);
    ++g_launches;
    return hipSuccess;
}
hipError_t launch_unpack(int dtype, int64_t B, const void* xs, const void* Ppk, void* x26, void* P576, void* om3, void* acc3, void* qo4, void* dofs6, hipStream_t) {
    const size_t es = es_of(dtype), b = (size_t)B;
    rd(xs, 36 * b * es); rd(Ppk, b * 300 * es);
    wr(x26, b * 26 * es); wr(P576, b * 576 * es); wr(om3, b * 3 * es); wr(acc3, b * 3 * es); wr(qo4, b * 4 * es); wr(dofs6, b * 6 * es);
    ++g_launches;
    return hipSuccess;
}
hipError_t launch_imu(int, int64_t, const void*, const void*, const void*, const void*, const void*, int, const void*, const void*, void*, void*, void*, void*, void*,
                      const double*, const HostParams&, hipStream_t) { ++g_launches; return hipSuccess; }
hipError_t launch_log_rows(int dtype, int64_t B, int64_t n_log, const void* xs, double, void* rows, hipStream_t) {
    rd(xs, 36 * (size_t)B * es_of(dtype)); wr(rows, (size_t)n_log * 30 * es_of(dtype)); ++g_launches; return hipSuccess;
}
hipError_t launch_update_mse(int dtype, int64_t B, const void* xs, const double* cam6, const double* imu6, void* mse, hipStream_t) {
    rd(xs, 36 * (size_t)B * es_of(dtype)); rd(cam6, 48); if (imu6) rd(imu6, 48); wr(mse, (size_t)B * es_of(dtype)); ++g_launches; return hipSuccess;
}
hipError_t launch_jacobians(int dtype, int64_t B, const void* J, int ip, void* Fx, void* Fi, int, hipStream_t) {
    const size_t es = es_of(dtype), b = (size_t)B;
    rd(static_cast<const char*>(J) + (size_t)ip * b * JAC_WORDS * es, b * JAC_WORDS * es); wr(Fx, b * 576 * es); wr(Fi, b * 24 * 13 * es);
    ++g_launches;
    return hipSuccess;
}
hipError_t launch_randn(int kind, uint64_t, int64_t, int64_t n_instances, int64_t, int64_t n_steps, int lanes, void* out, hipStream_t) {
    wr(out, (size_t)n_instances * (size_t)n_steps * (size_t)lanes * (kind == 0 ? 8 : kind == 1 ? 4 : 16)); ++g_launches; return hipSuccess;
}
}  // namespace eskf
