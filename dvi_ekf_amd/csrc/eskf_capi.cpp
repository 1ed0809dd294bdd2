// eskf_capi.cpp -- C ABI (include/dvi_eskf.h) over the HIP kernels.  Owns the device
// records of a batch of filter instances; never throws, never frees caller memory.
#include <hip/hip_runtime.h>

#include <dlfcn.h>

#include <cmath>
#include <cstdlib>
#include <vector>
#include <cstdio>
#include <cstring>
#include <new>

#include "../../include/dvi_eskf.h"
#include "eskf_device.h"

#ifdef ESKF_X_STAMPS
namespace eskf { hipError_t x_read_stamps(unsigned long long* out); hipError_t x_read_cstamps(unsigned long long* out); }
#endif
#ifdef ESKF_X_RUNSTAMPS
namespace eskf { hipError_t x_read_run_stamps(unsigned long long* out); hipError_t x_read_pre_stamps(unsigned long long* out); }
#endif

#if defined(ESKF_X_EPDUMP) || defined(ESKF_X_EPSTAMPS)
static unsigned long long* g_x_ep_buffer = nullptr;     // diagnostic builds: where the long-horizon kernel leaves its dump / stamps
#endif

struct dvi_eskf {
    int64_t B;
    int dtype;
    int device;
    int lpi;
    eskf::HostParams hp;
    void* d_P;   // [B][300] packed covariance
    void* d_x;   // [36][B]  state record, structure-of-arrays
    // workspace of the step pipeline (pre -> cov -> post kernels)
    void* d_J;          // [jcap][B][JAC_WORDS]
    void* d_aux;        // [AUX_WORDS0 + AUX_WORDSK jcap][B] hand-over of the epoch's recurrences to its Jacobian kernel
    int jcap;           // predicts per launch the Jacobian workspace can hold
    void* d_e;          // [B][24]
    int32_t* d_status;  // [B]
    // Deferred injection: WHETHER an instance's error state still awaits injection is device state (bit
    // DVI_STATUS_PENDING of d_status, set by the covariance kernel, cleared by the kernel that injects).
    // The host only keeps a hint so that it can skip work it knows to be pointless:
    bool maybe_pending; // an update has been enqueued since the last flush
    bool captured;      // a step of this handle was recorded into a HIP graph: replays change the device
                        // state behind the host's back, so from then on the hint is always "maybe"
    void* d_noise;      // nullable [B][NOISE_WORDS]: per-instance diag(Q) | diag(R) | sigma_om
    int jac_ip;         // record of the last predict in d_J when it was written in full (keep_jac), else -1
    // dvi_eskf_run_epochs on an under-filled batch: parts of the batch (two to MAX_SIDE) run their epochs on as many streams
    // of the handle's own, forked from and joined to the caller's stream by events (created at first use)
    static constexpr int MAX_SIDE = 8;
    hipStream_t side[MAX_SIDE];
    hipEvent_t ev_fork, ev_join[MAX_SIDE];
    bool side_ready;
    // dvi_eskf_run_epochs through the long-horizon kernel: the schedule's predict counts on the device
    int32_t* d_npred;
    int64_t npred_cap;
    std::vector<int32_t> npred_host;    // what d_npred holds (a run with the schedule of the run before uploads nothing)
    // ... its state snapshots for the trajectory log ([rows][26][n_log]) and the rows' time stamps (device doubles)
    void* d_xlog;
    size_t xlog_bytes;
    double* d_trows;
    int64_t trows_cap;
};

namespace {

size_t elem_size(int dtype) { return dtype == DVI_ESKF_F64 ? sizeof(double) : sizeof(float); }

void free_side_streams(dvi_eskf* h) {
    if (!h->side_ready) return;
    for (int i = 0; i < dvi_eskf::MAX_SIDE; ++i) { (void)hipStreamDestroy(h->side[i]); (void)hipEventDestroy(h->ev_join[i]); }
    (void)hipEventDestroy(h->ev_fork);
    h->side_ready = false;
}
bool ensure_side_streams(dvi_eskf* h) {
    if (h->side_ready) return true;
    bool ok = hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming) == hipSuccess;
    for (int i = 0; i < dvi_eskf::MAX_SIDE && ok; ++i)
        ok = hipStreamCreateWithFlags(&h->side[i], hipStreamNonBlocking) == hipSuccess &&
             hipEventCreateWithFlags(&h->ev_join[i], hipEventDisableTiming) == hipSuccess;
    h->side_ready = ok;     // (a partial failure leaks a handle or two of a process that is out of resources anyway)
    return ok;
}

void free_buffers(dvi_eskf* h) {
    if (h->d_P) (void)hipFree(h->d_P);
    if (h->d_x) (void)hipFree(h->d_x);
    if (h->d_J) (void)hipFree(h->d_J);
    if (h->d_aux) (void)hipFree(h->d_aux);
    h->d_aux = nullptr;
    if (h->d_e) (void)hipFree(h->d_e);
    if (h->d_status) (void)hipFree(h->d_status);
    if (h->d_noise) (void)hipFree(h->d_noise);
    h->d_noise = nullptr;
    if (h->d_npred) (void)hipFree(h->d_npred);
    h->d_npred = nullptr;
    h->npred_cap = 0;
    h->npred_host.clear();
    if (h->d_xlog) (void)hipFree(h->d_xlog);
    if (h->d_trows) (void)hipFree(h->d_trows);
    h->d_xlog = nullptr; h->xlog_bytes = 0;
    h->d_trows = nullptr; h->trows_cap = 0;
    h->d_P = h->d_x = h->d_J = h->d_e = nullptr;
    h->d_status = nullptr;
    h->jcap = 0;
}

// state records + single-predict workspace; false on allocation failure (buffers freed)
bool alloc_buffers(dvi_eskf* h) {
    const size_t es = elem_size(h->dtype);
    const size_t B = (size_t)h->B;
    h->d_P = h->d_x = h->d_J = h->d_e = nullptr;
    h->d_status = nullptr;
    h->d_aux = nullptr;
    h->jcap = 0;
    h->d_npred = nullptr;
    h->npred_cap = 0;
    h->npred_host.clear();
    h->d_xlog = nullptr; h->xlog_bytes = 0;
    h->d_trows = nullptr; h->trows_cap = 0;
    if (hipMalloc(&h->d_aux, B * (eskf::AUX_WORDS0 + eskf::AUX_WORDSK) * es) != hipSuccess ||
        hipMalloc(&h->d_P, B * 300 * es) != hipSuccess || hipMalloc(&h->d_x, B * 36 * es) != hipSuccess ||
        hipMalloc(&h->d_J, B * eskf::JAC_WORDS * es) != hipSuccess ||
        hipMalloc(&h->d_e, B * 24 * es) != hipSuccess ||
        hipMalloc((void**)&h->d_status, B * sizeof(int32_t)) != hipSuccess) {
        free_buffers(h);
        return false;
    }
    h->jcap = 1;
    return true;
}

bool stream_is_capturing(hipStream_t s) {
    // (the legacy NULL stream can never be capturing, and asking about it while ANOTHER stream is in a global-mode
    //  capture returns hipErrorStreamCaptureImplicit and invalidates that capture)
    if (s == nullptr) return false;
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &st) != hipSuccess) return false;
    return st != hipStreamCaptureStatusNone;
}

// Grow the Jacobian workspace for an n_pred-step epoch.  An allocation and a device-wide
// synchronisation: neither is allowed while a stream is being captured, so a capturing caller must have
// sized the workspace beforehand (dvi_eskf_reserve, or one eager epoch of that length) -- DVI_ESKF_ECAPTURE
// otherwise.  The same answer once ANY step of the handle has been captured: the address of the old workspace
// is baked into the recorded kernel arguments, and a replay after the old buffer has been freed would write
// Jacobian records into freed memory.  Happens at most once per handle and epoch length.
int ensure_jcap(dvi_eskf* h, int n_pred, hipStream_t stream) {
    if (n_pred <= h->jcap) return DVI_ESKF_OK;
    if (h->captured || stream_is_capturing(stream)) return DVI_ESKF_ECAPTURE;
    void* nj = nullptr;
    void* na = nullptr;
    if (hipMalloc(&nj, (size_t)n_pred * (size_t)h->B * eskf::JAC_WORDS * elem_size(h->dtype)) != hipSuccess) return DVI_ESKF_ENOMEM;
    if (hipMalloc(&na, (size_t)(eskf::AUX_WORDS0 + eskf::AUX_WORDSK * n_pred) * (size_t)h->B * elem_size(h->dtype)) != hipSuccess) {
        (void)hipFree(nj);
        return DVI_ESKF_ENOMEM;
    }
    if (hipDeviceSynchronize() != hipSuccess) {   // earlier launches may still read the old workspace
        (void)hipFree(nj);
        (void)hipFree(na);
        return DVI_ESKF_EHIP;
    }
    (void)hipFree(h->d_J);
    (void)hipFree(h->d_aux);
    h->d_J = nj;
    h->d_aux = na;
    h->jcap = n_pred;
    h->jac_ip = -1;
    return DVI_ESKF_OK;
}

struct DeviceGuard {
    int prev = -1;
    bool ok = true;
    explicit DeviceGuard(int dev) {
        if (hipGetDevice(&prev) != hipSuccess) { ok = false; return; }
        if (prev != dev && hipSetDevice(dev) != hipSuccess) ok = false;
    }
    ~DeviceGuard() {
        int cur = -1;
        if (prev >= 0 && hipGetDevice(&cur) == hipSuccess && cur != prev) (void)hipSetDevice(prev);
    }
};

eskf::StepArgs base_args(const dvi_eskf* h) {
    eskf::StepArgs a;
    std::memset(&a, 0, sizeof a);
    a.P = h->d_P; a.xs = h->d_x;
    a.P_src = h->d_P; a.xs_src = h->d_x;
    a.J = h->d_J; a.aux = h->d_aux; a.e = h->d_e; a.status_ws = h->d_status; a.noise = h->d_noise;
    a.B = h->B;
    a.rounds = 1;
    return a;
}

// apply a deferred injection so that d_x is the state the caller expects to see: the flush kernel
// injects where the device says an injection is due and does nothing elsewhere
int flush_pending(dvi_eskf* h, hipStream_t stream) {
    if (stream_is_capturing(stream)) h->captured = true;
    if (!h->maybe_pending && !h->captured) return DVI_ESKF_OK;
    eskf::StepArgs a = base_args(h);
    if (eskf::launch_flush(h->dtype, a, h->hp, stream) != hipSuccess) return DVI_ESKF_EHIP;
    h->maybe_pending = false;
    return DVI_ESKF_OK;
}

int step_common(dvi_eskf* h, int n_pred, int do_update, const void* dt, int dt_pi, const void* om, const void* acc,
                int imu_pi, const void* pos, const void* quat, const void* notch, int meas_pi, void* K_out,
                int32_t* status, void* stream, dvi_eskf* src = nullptr, int64_t sub_first = 0, int64_t sub_count = 0) {
    if (!h) return DVI_ESKF_EINVAL;
    if (src == h) src = nullptr;
    if (src && (src->B != h->B || src->dtype != h->dtype || src->device != h->device)) return DVI_ESKF_EINVAL;
    if (n_pred < 0) return DVI_ESKF_EINVAL;
    if (n_pred > 0 && (!dt || !om || !acc)) return DVI_ESKF_EINVAL;
    if (do_update && (!pos || !quat || !notch)) return DVI_ESKF_EINVAL;
    if (n_pred == 0 && !do_update) return DVI_ESKF_OK;
    // quirk Q1 switched off (dvi_eskf_params.quirks_fixed): predicts only through the covariance kernel that knows the other
    // column layout of the camera rows -- row-resident, 32 lanes per instance, shared noise model, no gyro block in Q, no gain output
    if (!(h->hp.quirks & eskf::QK_Q1) && n_pred > 0) {
        const bool q_gyro = h->hp.qd[3] != 0.0 || h->hp.qd[4] != 0.0 || h->hp.qd[5] != 0.0 || h->hp.q_gyro_any;
        if (h->lpi != 1 || h->d_noise || q_gyro || (do_update && K_out)) return DVI_ESKF_EINVAL;
    }
    DeviceGuard g(h->device);
    if (!g.ok) return DVI_ESKF_EHIP;
    if (int rc = ensure_jcap(h, n_pred, (hipStream_t)stream)) return rc;
    if (stream_is_capturing((hipStream_t)stream)) h->captured = true;
    eskf::StepArgs a = base_args(h);
    a.dt = dt; a.om = om; a.acc = acc;
    a.pos = pos; a.quat = quat; a.notch = notch;
    a.K_out = do_update ? K_out : nullptr;
    a.status = do_update ? status : nullptr;
    a.n_pred = n_pred;
    a.do_update = do_update ? 1 : 0;
    a.dt_stride = dt_pi ? 1 : 0;
    a.imu_stride = imu_pi ? 1 : 0;
    a.meas_stride = meas_pi ? 1 : 0;
    a.pending_hint = (h->maybe_pending || h->captured) ? 1 : 0;
    a.sub_first = sub_first;
    a.sub_count = sub_count;
    if (src) {      // read the other handle's (flushed) state; whatever this handle held is replaced
        if (flush_pending(src, (hipStream_t)stream) != DVI_ESKF_OK) return DVI_ESKF_EHIP;
        a.P_src = src->d_P;
        a.xs_src = src->d_x;
        a.drop_pending = 1;
    }
    hipError_t e = eskf::launch_step(h->dtype, h->lpi, a, h->hp, (hipStream_t)stream);
    if (e != hipSuccess) return DVI_ESKF_EHIP;
    // the pre kernel has consumed what was due; this step's update (if any) leaves its own error state behind
    h->maybe_pending = do_update != 0;
    if (n_pred > 0) h->jac_ip = h->hp.keep_jac ? n_pred - 1 : -1;
    return DVI_ESKF_OK;
}

}  // namespace

extern "C" {

int dvi_eskf_create(const dvi_eskf_params* p, int64_t batch, int dtype, int device, dvi_eskf** out) {
    if (!p || !out || batch <= 0) return DVI_ESKF_EINVAL;
    if (dtype != DVI_ESKF_F64 && dtype != DVI_ESKF_F32) return DVI_ESKF_EINVAL;
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev) return DVI_ESKF_ENODEV;
    DeviceGuard g(device);
    if (!g.ok) return DVI_ESKF_ENODEV;
    dvi_eskf* h = new (std::nothrow) dvi_eskf();
    if (!h) return DVI_ESKF_ENOMEM;
    h->B = batch; h->dtype = dtype; h->device = device; h->lpi = 1; h->d_noise = nullptr;
    h->maybe_pending = false; h->captured = false; h->jac_ip = -1;
    h->side_ready = false;
    h->hp.q_gyro_any = 0; h->hp.keep_jac = 0;
    h->hp.L = p->L; h->hp.theta_cam = p->theta_cam;
    h->hp.frozen_mask = 0;
    for (int i = 0; i < 6; ++i) if (p->frozen[i]) h->hp.frozen_mask |= (1 << i);
    h->hp.hold_frozen = p->hold_frozen ? 1 : 0;
    if (p->quirks_fixed & ~(DVI_ESKF_FIX_Q1 | DVI_ESKF_FIX_Q3 | DVI_ESKF_FIX_Q6 | DVI_ESKF_FIX_Q9)) { delete h; return DVI_ESKF_EINVAL; }
    h->hp.quirks = eskf::QK_V1 & ~((p->quirks_fixed & DVI_ESKF_FIX_Q1 ? eskf::QK_Q1 : 0) | (p->quirks_fixed & DVI_ESKF_FIX_Q3 ? eskf::QK_Q3 : 0) |
                                   (p->quirks_fixed & DVI_ESKF_FIX_Q6 ? eskf::QK_Q6 : 0) | (p->quirks_fixed & DVI_ESKF_FIX_Q9 ? eskf::QK_Q9 : 0));
    for (int i = 0; i < 3; ++i) { h->hp.sig_a[i] = p->sigma_a[i]; h->hp.sig_om[i] = p->sigma_om[i]; }
    for (int i = 0; i < 13; ++i) h->hp.qd[i] = 1.0;   // Q = eye before set_noise (Filter.py:68)
    for (int i = 0; i < 7; ++i) h->hp.r[i] = 1.0;
    const size_t es = elem_size(dtype);
    if (!alloc_buffers(h)) {
        delete h;
        return DVI_ESKF_ENOMEM;
    }
    // zeroed records, complete before this call returns: the caller's first call may run on any stream
    // (torch side streams do not synchronise with the null stream these memsets run on)
    if (hipMemsetAsync(h->d_P, 0, (size_t)batch * 300 * es, nullptr) != hipSuccess ||
        hipMemsetAsync(h->d_x, 0, (size_t)batch * 36 * es, nullptr) != hipSuccess ||
        hipMemsetAsync(h->d_status, 0, (size_t)batch * sizeof(int32_t), nullptr) != hipSuccess ||
        hipStreamSynchronize(nullptr) != hipSuccess) {
        free_buffers(h);
        delete h;
        return DVI_ESKF_EHIP;
    }
    *out = h;
    return DVI_ESKF_OK;
}

int dvi_eskf_clone(const dvi_eskf* src, void* stream, dvi_eskf** out) {
    if (!src || !out) return DVI_ESKF_EINVAL;
    *out = nullptr;
    DeviceGuard g(src->device);
    if (!g.ok) return DVI_ESKF_EHIP;
    if (flush_pending(const_cast<dvi_eskf*>(src), (hipStream_t)stream) != DVI_ESKF_OK) return DVI_ESKF_EHIP;
    dvi_eskf* h = new (std::nothrow) dvi_eskf(*src);
    if (!h) return DVI_ESKF_ENOMEM;
    const size_t es = elem_size(src->dtype);
    h->d_noise = nullptr;             // (the struct copy aliased the source's buffers)
    h->d_npred = nullptr;
    h->npred_cap = 0;
    h->d_xlog = nullptr; h->xlog_bytes = 0;
    h->d_trows = nullptr; h->trows_cap = 0;
    h->side_ready = false;            // (... and its streams: the copy makes its own when it needs them)
    if (!alloc_buffers(h)) {
        delete h;
        return DVI_ESKF_ENOMEM;
    }
    hipStream_t s = (hipStream_t)stream;
    h->maybe_pending = false;         // the copy starts from the flushed state: nothing is due on it
    h->captured = stream_is_capturing(s);
    h->jac_ip = -1;
    const size_t noise_bytes = (size_t)src->B * eskf::NOISE_WORDS * es;
    if (src->d_noise && hipMalloc(&h->d_noise, noise_bytes) != hipSuccess) {
        h->d_noise = nullptr;
        free_buffers(h);
        delete h;
        return DVI_ESKF_ENOMEM;
    }
    if (hipMemsetAsync(h->d_status, 0, (size_t)src->B * sizeof(int32_t), s) != hipSuccess ||
        hipMemcpyAsync(h->d_P, src->d_P, (size_t)src->B * 300 * es, hipMemcpyDeviceToDevice, s) != hipSuccess ||
        hipMemcpyAsync(h->d_x, src->d_x, (size_t)src->B * 36 * es, hipMemcpyDeviceToDevice, s) != hipSuccess ||
        (src->d_noise && hipMemcpyAsync(h->d_noise, src->d_noise, noise_bytes, hipMemcpyDeviceToDevice, s) != hipSuccess)) {
        free_buffers(h);
        delete h;
        return DVI_ESKF_EHIP;
    }
    *out = h;
    return DVI_ESKF_OK;
}

int dvi_eskf_copy_state(dvi_eskf* dst, const dvi_eskf* src, void* stream) {
    if (!dst || !src || dst->B != src->B || dst->dtype != src->dtype || dst->device != src->device) return DVI_ESKF_EINVAL;
    DeviceGuard g(src->device);
    if (!g.ok) return DVI_ESKF_EHIP;
    if (flush_pending(const_cast<dvi_eskf*>(src), (hipStream_t)stream) != DVI_ESKF_OK) return DVI_ESKF_EHIP;
    const size_t es = elem_size(src->dtype);
    hipStream_t s = (hipStream_t)stream;
    if (stream_is_capturing(s)) dst->captured = true;
    // whatever was due on dst is overwritten together with the state it belonged to
    if ((dst->maybe_pending || dst->captured) &&
        hipMemsetAsync(dst->d_status, 0, (size_t)dst->B * sizeof(int32_t), s) != hipSuccess) return DVI_ESKF_EHIP;
    dst->maybe_pending = false;
    if (dst == src) return DVI_ESKF_OK;
    if (eskf::launch_copy_state(src->d_P, dst->d_P, (size_t)src->B * 300 * es, src->d_x, dst->d_x, (size_t)src->B * 36 * es, s) != hipSuccess)
        return DVI_ESKF_EHIP;
    return DVI_ESKF_OK;
}

int dvi_eskf_copy_instances(dvi_eskf* dst, const dvi_eskf* src, int64_t first, void* stream) {
    if (!dst || !src || dst == src || dst->dtype != src->dtype || dst->device != src->device) return DVI_ESKF_EINVAL;
    if (first < 0 || first + dst->B > src->B) return DVI_ESKF_EINVAL;
    DeviceGuard g(src->device);
    if (!g.ok) return DVI_ESKF_EHIP;
    hipStream_t s = (hipStream_t)stream;
    if (flush_pending(const_cast<dvi_eskf*>(src), s) != DVI_ESKF_OK) return DVI_ESKF_EHIP;
    if (stream_is_capturing(s)) dst->captured = true;
    // whatever was due on dst is overwritten together with the state it belonged to
    if ((dst->maybe_pending || dst->captured) && hipMemsetAsync(dst->d_status, 0, (size_t)dst->B * sizeof(int32_t), s) != hipSuccess) return DVI_ESKF_EHIP;
    dst->maybe_pending = false;
    if (eskf::launch_copy_instances(src->dtype, src->d_P, src->d_x, src->B, first, dst->d_P, dst->d_x, dst->B, s) != hipSuccess) return DVI_ESKF_EHIP;
    return DVI_ESKF_OK;
}

void dvi_eskf_destroy(dvi_eskf* h) {
    if (!h) return;
    DeviceGuard g(h->device);
    free_side_streams(h);
    free_buffers(h);
    delete h;
}

int64_t dvi_eskf_batch(const dvi_eskf* h) { return h ? h->B : 0; }
int dvi_eskf_dtype(const dvi_eskf* h) { return h ? h->dtype : DVI_ESKF_EINVAL; }

int dvi_eskf_set_noise(dvi_eskf* h, const double* Q, const double* R) {
    if (!h) return DVI_ESKF_EINVAL;
    if (Q) {
        for (int i = 0; i < 13; ++i)
            for (int j = 0; j < 13; ++j)
                if (i != j && Q[i * 13 + j] != 0.0) return DVI_ESKF_EINVAL;
        for (int i = 0; i < 13; ++i) if (!std::isfinite(Q[i * 13 + i])) return DVI_ESKF_EINVAL;
    }
    if (R) {
        for (int i = 0; i < 7; ++i)
            for (int j = 0; j < 7; ++j)
                if (i != j && R[i * 7 + j] != 0.0) return DVI_ESKF_EINVAL;
        for (int i = 0; i < 7; ++i) if (!std::isfinite(R[i * 7 + i])) return DVI_ESKF_EINVAL;
    }
    // (a captured step has the per-instance noise buffer among its recorded arguments: it cannot be freed any more)
    if (h->d_noise && h->captured) return DVI_ESKF_ECAPTURE;
    if (Q) for (int i = 0; i < 13; ++i) h->hp.qd[i] = Q[i * 13 + i];
    if (R) for (int i = 0; i < 7; ++i) h->hp.r[i] = R[i * 7 + i];
    if (h->d_noise) {                 // back to one noise model for the whole batch
        DeviceGuard g(h->device);
        if (!g.ok || hipDeviceSynchronize() != hipSuccess) return DVI_ESKF_EHIP;
        (void)hipFree(h->d_noise);
        h->d_noise = nullptr;
        h->hp.q_gyro_any = 0;
    }
    return DVI_ESKF_OK;
}

int dvi_eskf_set_noise_per_instance(dvi_eskf* h, const double* q13, const double* r7, const double* sigma_om3,
                                    void* stream) {
    if (!h || !q13 || !r7) return DVI_ESKF_EINVAL;
    const size_t B = (size_t)h->B;
    for (size_t i = 0; i < B * 13; ++i) if (!std::isfinite(q13[i])) return DVI_ESKF_EINVAL;
    for (size_t i = 0; i < B * 7; ++i) if (!std::isfinite(r7[i])) return DVI_ESKF_EINVAL;
    DeviceGuard g(h->device);
    if (!g.ok) return DVI_ESKF_EHIP;
    const size_t es = elem_size(h->dtype);
    const int W = eskf::NOISE_WORDS;
    if (!h->d_noise && hipMalloc(&h->d_noise, B * W * es) != hipSuccess) { h->d_noise = nullptr; return DVI_ESKF_ENOMEM; }
    std::vector<unsigned char> host(B * W * es);
    int any = 0;
    for (size_t b = 0; b < B; ++b) {
        double rec[eskf::NOISE_WORDS] = {0};
        for (int k = 0; k < 13; ++k) rec[k] = q13[b * 13 + k];
        for (int k = 0; k < 7; ++k) rec[13 + k] = r7[b * 7 + k];
        for (int k = 0; k < 3; ++k) rec[20 + k] = sigma_om3 ? sigma_om3[b * 3 + k] : h->hp.sig_om[k];
        if (rec[3] != 0.0 || rec[4] != 0.0 || rec[5] != 0.0) any = 1;
        for (int k = 0; k < W; ++k) {
            if (h->dtype == DVI_ESKF_F64) reinterpret_cast<double*>(host.data())[b * W + k] = rec[k];
            else reinterpret_cast<float*>(host.data())[b * W + k] = (float)rec[k];
        }
    }
    hipStream_t s = (hipStream_t)stream;
    if (hipMemcpyAsync(h->d_noise, host.data(), host.size(), hipMemcpyHostToDevice, s) != hipSuccess ||
        hipStreamSynchronize(s) != hipSuccess)
        return DVI_ESKF_EHIP;
    h->hp.q_gyro_any = any;
    for (int k = 0; k < 13; ++k) h->hp.qd[k] = q13[k];     // instance 0, for dvi_eskf_* calls that report the model
    for (int k = 0; k < 7; ++k) h->hp.r[k] = r7[k];
    return DVI_ESKF_OK;
}

int dvi_eskf_set_state(dvi_eskf* h, const void* x26, const void* P576, const void* om3, const void* acc3,
                       const void* qo4, void* stream) {
    if (!h) return DVI_ESKF_EINVAL;
    DeviceGuard g(h->device);
    if (!g.ok) return DVI_ESKF_EHIP;
    if (flush_pending(h, (hipStream_t)stream) != DVI_ESKF_OK) return DVI_ESKF_EHIP;
    hipError_t e = eskf::launch_pack(h->dtype, h->B, x26, P576, om3, acc3, qo4, h->d_x, h->d_P, (hipStream_t)stream);
    return e == hipSuccess ? DVI_ESKF_OK : DVI_ESKF_EHIP;
}

int dvi_eskf_get_state(const dvi_eskf* h, void* x26, void* P576, void* om3, void* acc3, void* qo4, void* stream) {
    if (!h) return DVI_ESKF_EINVAL;
    DeviceGuard g(h->device);
    if (!g.ok) return DVI_ESKF_EHIP;
    if (flush_pending(const_cast<dvi_eskf*>(h), (hipStream_t)stream) != DVI_ESKF_OK) return DVI_ESKF_EHIP;
    hipError_t e = eskf::launch_unpack(h->dtype, h->B, h->d_x, h->d_P, x26, P576, om3, acc3, qo4, nullptr,
                                       (hipStream_t)stream);
    return e == hipSuccess ? DVI_ESKF_OK : DVI_ESKF_EHIP;
}

int dvi_eskf_get_calibration(const dvi_eskf* h, void* dofs_out, void* stream) {
    if (!h || !dofs_out) return DVI_ESKF_EINVAL;
    DeviceGuard g(h->device);
    if (!g.ok) return DVI_ESKF_EHIP;
    if (flush_pending(const_cast<dvi_eskf*>(h), (hipStream_t)stream) != DVI_ESKF_OK) return DVI_ESKF_EHIP;
    hipError_t e = eskf::launch_unpack(h->dtype, h->B, h->d_x, h->d_P, nullptr, nullptr, nullptr, nullptr, nullptr,
                                       dofs_out, (hipStream_t)stream);
    return e == hipSuccess ? DVI_ESKF_OK : DVI_ESKF_EHIP;
}

int dvi_eskf_predict(dvi_eskf* h, const void* dt, int dt_pi, const void* om, const void* acc, int imu_pi,
                     void* stream) {
    return step_common(h, 1, 0, dt, dt_pi, om, acc, imu_pi, nullptr, nullptr, nullptr, 0, nullptr, nullptr, stream);
}

int dvi_eskf_update(dvi_eskf* h, const void* pos, const void* quat, const void* notch, int meas_pi, void* K_out,
                    int32_t* status, void* stream) {
    return step_common(h, 0, 1, nullptr, 0, nullptr, nullptr, 0, pos, quat, notch, meas_pi, K_out, status, stream);
}

int dvi_eskf_step(dvi_eskf* h, const void* dt, int dt_pi, const void* om, const void* acc, int imu_pi,
                  const void* pos, const void* quat, const void* notch, int meas_pi, void* K_out, int32_t* status,
                  void* stream) {
    return step_common(h, 1, 1, dt, dt_pi, om, acc, imu_pi, pos, quat, notch, meas_pi, K_out, status, stream);
}

int dvi_eskf_epoch(dvi_eskf* h, int n_pred, const void* dt, int dt_pi, const void* om, const void* acc, int imu_pi,
                   const void* pos, const void* quat, const void* notch, int meas_pi, void* K_out, int32_t* status,
                   void* stream) {
    return step_common(h, n_pred, 1, dt, dt_pi, om, acc, imu_pi, pos, quat, notch, meas_pi, K_out, status, stream);
}

int dvi_eskf_step_from(dvi_eskf* h, dvi_eskf* src, int n_pred, const void* dt, int dt_pi, const void* om,
                       const void* acc, int imu_pi, const void* pos, const void* quat, const void* notch, int meas_pi,
                       void* K_out, int32_t* status, void* stream) {
    if (!src) return DVI_ESKF_EINVAL;
    return step_common(h, n_pred, 1, dt, dt_pi, om, acc, imu_pi, pos, quat, notch, meas_pi, K_out, status, stream, src);
}

}  // extern "C"

namespace {
// A run whose every epoch is ONE predict + the update (the 1:1 schedule of the saved trajectory and of the Monte-Carlo
// batches) as ONE launch of eskf_run_kernel: the time loop of Filter.run (Filter.py:144-168) inside the kernel, every
// block stepping its own instances.  Returns 1 if the run was enqueued that way, 0 if it has to go epoch by epoch
// (log rows, other schedules, configurations the kernel does not cover, DVI_ESKF_RUN=0), < 0 on error.
int run_pairs_in_one_launch(dvi_eskf* h, dvi_eskf* src, const dvi_eskf_schedule* sc, void* stream) {
    if (sc->rows_out || sc->rows_pred_out) return 0;
    if (sc->n_epochs > 0x7fffffffLL) return 0;
    for (int64_t e = 0; e < sc->n_epochs; ++e)
        if (sc->n_pred[e] != 1) return 0;
    if (src == h) src = nullptr;
    if (src && (src->B != h->B || src->dtype != h->dtype || src->device != h->device)) return DVI_ESKF_EINVAL;
    DeviceGuard g(h->device);
    if (!g.ok) return DVI_ESKF_EHIP;
    const size_t es = elem_size(h->dtype);
    const size_t B = (size_t)h->B;
    eskf::StepArgs a = base_args(h);
    a.dt = sc->dt; a.om = sc->om; a.acc = sc->acc;
    a.pos = sc->pos; a.quat = sc->quat_xyzw; a.notch = sc->notch;
    a.status = sc->status;
    a.n_pred = 1;
    a.do_update = 1;
    a.dt_stride = sc->dt_per_instance ? 1 : 0;
    a.imu_stride = sc->imu_per_instance ? 1 : 0;
    a.meas_stride = sc->meas_per_instance ? 1 : 0;
    if (!eskf::run_kernel_block(h->dtype, h->lpi, a, h->hp)) return 0;
    if (stream_is_capturing((hipStream_t)stream)) h->captured = true;
    a.pending_hint = (h->maybe_pending || h->captured) ? 1 : 0;
    if (src) {
        if (flush_pending(src, (hipStream_t)stream) != DVI_ESKF_OK) return DVI_ESKF_EHIP;
        a.P_src = src->d_P;
        a.xs_src = src->d_x;
        a.drop_pending = 1;
    }
    eskf::RunArgs ra;
    std::memset(&ra, 0, sizeof ra);
    ra.n_steps = (int)sc->n_epochs;
    ra.dt_step = (int64_t)((sc->dt_per_instance ? B : 1) * es);
    ra.imu_step = (int64_t)((sc->imu_per_instance ? B : 1) * 3 * es);
    const size_t mrow = sc->meas_per_instance ? B : 1;
    ra.pos_step = (int64_t)(mrow * 3 * es);
    ra.quat_step = (int64_t)(mrow * 4 * es);
    ra.notch_step = (int64_t)(mrow * es);
    ra.status_step = (int64_t)B;
    if (eskf::launch_run(h->dtype, h->lpi, a, ra, h->hp, (hipStream_t)stream) != hipSuccess) return DVI_ESKF_EHIP;
    h->maybe_pending = true;        // the last step's update leaves its error state behind
    h->jac_ip = -1;
    return 1;
}

// A run with epochs of any length as ONE launch of eskf_epochs_kernel (eskf_epochs.hip): the time loop of Filter.run and
// the predicts of propagate_imu (Filter.py:144-217) inside the kernel, the covariances in registers from the first epoch
// to the last.  Returns 1 if the run was enqueued that way, 0 if it has to go epoch by epoch (log rows, configurations
// the kernel does not cover, a schedule that would have to be uploaded while the stream is being captured,
// DVI_ESKF_EPOCHS=0), < 0 on error.
int run_epochs_in_one_launch(dvi_eskf* h, dvi_eskf* src, const dvi_eskf_schedule* sc, void* stream) {
    if (sc->n_epochs > 0x7fffffffLL) return 0;
    int64_t total = 0;
    bool uniform = true;
    for (int64_t e = 0; e < sc->n_epochs; ++e) {
        total += sc->n_pred[e];
        uniform = uniform && sc->n_pred[e] == sc->n_pred[0];
    }
    if (total > 0x7fffffffLL) return 0;
    if (src == h) src = nullptr;
    if (src && (src->B != h->B || src->dtype != h->dtype || src->device != h->device)) return DVI_ESKF_EINVAL;
    DeviceGuard g(h->device);
    if (!g.ok) return DVI_ESKF_EHIP;
    hipStream_t s = (hipStream_t)stream;
    eskf::EpochsArgs ea;
    std::memset(&ea, 0, sizeof ea);
    eskf::StepArgs& a = ea.a;
    a = base_args(h);
    a.dt = sc->dt; a.om = sc->om; a.acc = sc->acc;
    a.pos = sc->pos; a.quat = sc->quat_xyzw; a.notch = sc->notch;
    a.status = sc->status;
    a.n_pred = 0;
    a.do_update = 1;
    a.dt_stride = sc->dt_per_instance ? 1 : 0;
    a.imu_stride = sc->imu_per_instance ? 1 : 0;
    a.meas_stride = sc->meas_per_instance ? 1 : 0;
    if (!eskf::epochs_kernel_block(h->dtype, h->lpi, a, h->hp)) return 0;
    const bool capturing = stream_is_capturing(s);
    if (!uniform) {
        // the predict counts go to the device: an allocation (first time, or a longer schedule than any before) and a copy
        // from pageable host memory -- neither can be recorded into a graph, nor may the buffer of a recorded launch move
        const bool same = (int64_t)h->npred_host.size() == sc->n_epochs && h->d_npred &&
                          std::memcmp(h->npred_host.data(), sc->n_pred, (size_t)sc->n_epochs * sizeof(int32_t)) == 0;
        // (a recorded launch reads the buffer at every replay: once a run of the handle has been captured it is never written again)
        if (!same && (capturing || h->captured)) return 0;
        if (!same && sc->n_epochs > h->npred_cap) {
            int32_t* nb = nullptr;
            if (hipMalloc((void**)&nb, (size_t)sc->n_epochs * sizeof(int32_t)) != hipSuccess) return DVI_ESKF_ENOMEM;
            if (h->d_npred) {
                if (hipDeviceSynchronize() != hipSuccess) { (void)hipFree(nb); return DVI_ESKF_EHIP; }   // (an earlier run may still read it)
                (void)hipFree(h->d_npred);
            }
            h->d_npred = nb;
            h->npred_cap = sc->n_epochs;
        }
        // (stream-ordered behind the runs that read the buffer before; the source is the caller's host array, consumed
        //  before this call returns: pageable memory is staged synchronously)
        if (!same) {
            h->npred_host.clear();
            if (hipMemcpyAsync(h->d_npred, sc->n_pred, (size_t)sc->n_epochs * sizeof(int32_t), hipMemcpyHostToDevice, s) != hipSuccess) return DVI_ESKF_EHIP;
            if (hipStreamSynchronize(s) != hipSuccess) return DVI_ESKF_EHIP;
            h->npred_host.assign(sc->n_pred, sc->n_pred + sc->n_epochs);
        }
        ea.n_pred = h->d_npred;
    }
    // the trajectory log: state snapshots written by the kernel, turned into rows afterwards (allocations and a copy from
    // host memory: not while the stream is being captured)
    const bool log_pred = sc->rows_pred_out != nullptr, log_epoch = sc->rows_out != nullptr;
    const int64_t n_rows = log_pred ? total : log_epoch ? sc->n_epochs : 0;
    const int64_t n_log = log_pred ? (sc->rows_instances > 0 && sc->rows_instances < h->B ? sc->rows_instances : h->B) : h->B;
    if (n_rows > 0) {
        if (capturing || h->captured) return 0;
        const size_t need = (size_t)n_rows * 26 * (size_t)n_log * elem_size(h->dtype);
        if (need > h->xlog_bytes || n_rows > h->trows_cap) {
            if (hipDeviceSynchronize() != hipSuccess) return DVI_ESKF_EHIP;
            if (need > h->xlog_bytes) {
                if (h->d_xlog) (void)hipFree(h->d_xlog);
                h->d_xlog = nullptr; h->xlog_bytes = 0;
                if (hipMalloc(&h->d_xlog, need) != hipSuccess) { h->d_xlog = nullptr; return DVI_ESKF_ENOMEM; }
                h->xlog_bytes = need;
            }
            if (n_rows > h->trows_cap) {
                if (h->d_trows) (void)hipFree(h->d_trows);
                h->d_trows = nullptr; h->trows_cap = 0;
                if (hipMalloc((void**)&h->d_trows, (size_t)n_rows * sizeof(double)) != hipSuccess) { h->d_trows = nullptr; return DVI_ESKF_ENOMEM; }
                h->trows_cap = n_rows;
            }
        }
        if (hipMemcpyAsync(h->d_trows, log_pred ? sc->t_pred : sc->t_log, (size_t)n_rows * sizeof(double), hipMemcpyHostToDevice, s) != hipSuccess ||
            hipStreamSynchronize(s) != hipSuccess)
            return DVI_ESKF_EHIP;
        ea.xlog = h->d_xlog;
        ea.xlog_mode = log_pred ? 1 : 2;
        ea.n_log = n_log;
    }
    if (capturing) h->captured = true;
    a.pending_hint = (h->maybe_pending || h->captured) ? 1 : 0;
    if (src) {
        if (flush_pending(src, s) != DVI_ESKF_OK) return DVI_ESKF_EHIP;
        a.P_src = src->d_P;
        a.xs_src = src->d_x;
        a.drop_pending = 1;
    }
#if defined(ESKF_X_EPDUMP) || defined(ESKF_X_EPSTAMPS)
    ea.stamps = g_x_ep_buffer;
#endif
    ea.n_epochs = (int32_t)sc->n_epochs;
    ea.n_total = (int32_t)total;
    ea.n_uniform = uniform ? sc->n_pred[0] : -1;
    if (eskf::launch_epochs(h->dtype, h->lpi, ea, h->hp, s) != hipSuccess) return DVI_ESKF_EHIP;
    h->maybe_pending = true;        // the last epoch's update leaves its error state behind
    h->jac_ip = -1;
    if (n_rows > 0) {               // the last row is the state after the run's last update: injected now, read from the state record
        if (flush_pending(h, s) != DVI_ESKF_OK) return DVI_ESKF_EHIP;
        if (eskf::launch_log_rows_run(h->dtype, n_rows, n_log, h->d_xlog, h->d_x, h->B, h->d_trows, log_pred ? sc->rows_pred_out : sc->rows_out, s) != hipSuccess)
            return DVI_ESKF_EHIP;
    }
    return 1;
}

int run_epochs_impl(dvi_eskf* h, dvi_eskf* src, const dvi_eskf_schedule* sc, void* stream);
}  // namespace

extern "C" {

int dvi_eskf_run_epochs(dvi_eskf* h, const dvi_eskf_schedule* sc, void* stream) { return run_epochs_impl(h, nullptr, sc, stream); }

int dvi_eskf_run_epochs_from(dvi_eskf* h, dvi_eskf* src, const dvi_eskf_schedule* sc, void* stream) {
    if (!src) return DVI_ESKF_EINVAL;
    return run_epochs_impl(h, src, sc, stream);
}

}  // extern "C"

namespace {
int run_epochs_impl(dvi_eskf* h, dvi_eskf* src, const dvi_eskf_schedule* sc, void* stream) {
    if (!h || !sc || sc->n_epochs < 0) return DVI_ESKF_EINVAL;
    if (sc->n_epochs == 0) return DVI_ESKF_OK;
    if (!sc->n_pred || !sc->pos || !sc->quat_xyzw || !sc->notch) return DVI_ESKF_EINVAL;
    if (sc->rows_out && !sc->t_log) return DVI_ESKF_EINVAL;
    if (sc->rows_pred_out && (!sc->t_pred || sc->rows_out || sc->rows_instances < 0 || sc->rows_instances > h->B)) return DVI_ESKF_EINVAL;
    int64_t total = 0;
    for (int64_t e = 0; e < sc->n_epochs; ++e) {
        if (sc->n_pred[e] < 0) return DVI_ESKF_EINVAL;
        total += sc->n_pred[e];
    }
    if (total > 0 && (!sc->dt || !sc->om || !sc->acc)) return DVI_ESKF_EINVAL;
    if (src == h) src = nullptr;
    if (src && (src->B != h->B || src->dtype != h->dtype || src->device != h->device)) return DVI_ESKF_EINVAL;
    if (int one = run_pairs_in_one_launch(h, src, sc, stream)) return one < 0 ? one : DVI_ESKF_OK;
    if (int one = run_epochs_in_one_launch(h, src, sc, stream)) return one < 0 ? one : DVI_ESKF_OK;
    if (sc->rows_pred_out) {
        // a row per predict where the long-horizon kernel does not take the run: the reference's own call sequence, predict by
        // predict (Filter.propagate + log row), then the update (which replaces the row of the epoch's last predict)
        const size_t es_ = elem_size(h->dtype);
        const size_t B_ = (size_t)h->B;
        const size_t nl = (size_t)((sc->rows_instances > 0 && sc->rows_instances < h->B) ? sc->rows_instances : h->B);
        auto log_row = [&](size_t gg) -> int {        // row gg of the run: the flushed state of the first nl instances
            DeviceGuard dg(h->device);
            if (!dg.ok || flush_pending(h, (hipStream_t)stream) != DVI_ESKF_OK) return DVI_ESKF_EHIP;
            return eskf::launch_log_rows(h->dtype, h->B, (int64_t)nl, h->d_x, sc->t_pred[gg], static_cast<char*>(sc->rows_pred_out) + gg * nl * 30 * es_,
                                         (hipStream_t)stream) == hipSuccess ? DVI_ESKF_OK : DVI_ESKF_EHIP;
        };
        const size_t dt_row_ = (sc->dt_per_instance ? B_ : 1) * es_, imu_row_ = (sc->imu_per_instance ? B_ : 1) * 3 * es_;
        const size_t mrow_ = sc->meas_per_instance ? B_ : 1;
        size_t g = 0;
        for (int64_t e = 0; e < sc->n_epochs; ++e) {
            for (int k = 0; k < sc->n_pred[e]; ++k, ++g) {
                int rc = step_common(h, 1, 0, static_cast<const char*>(sc->dt) + g * dt_row_, sc->dt_per_instance, static_cast<const char*>(sc->om) + g * imu_row_,
                                     static_cast<const char*>(sc->acc) + g * imu_row_, sc->imu_per_instance, nullptr, nullptr, nullptr, 0, nullptr, nullptr,
                                     stream, (e == 0 && k == 0) ? src : nullptr);
                if (rc != DVI_ESKF_OK) return rc;
                rc = log_row(g);
                if (rc != DVI_ESKF_OK) return rc;
            }
            int rc = step_common(h, 0, 1, nullptr, 0, nullptr, nullptr, 0, static_cast<const char*>(sc->pos) + (size_t)e * mrow_ * 3 * es_,
                                 static_cast<const char*>(sc->quat_xyzw) + (size_t)e * mrow_ * 4 * es_, static_cast<const char*>(sc->notch) + (size_t)e * mrow_ * es_,
                                 sc->meas_per_instance, nullptr, sc->status ? sc->status + (size_t)e * B_ : nullptr, stream, (e == 0 && sc->n_pred[0] == 0) ? src : nullptr);
            if (rc != DVI_ESKF_OK) return rc;
            if (g > 0) {
                rc = log_row(g - 1);
                if (rc != DVI_ESKF_OK) return rc;
            }
        }
        return DVI_ESKF_OK;
    }
    {   // size the Jacobian workspace for the longest epoch once, not epoch by epoch
        int longest = 0;
        for (int64_t e = 0; e < sc->n_epochs; ++e) longest = sc->n_pred[e] > longest ? sc->n_pred[e] : longest;
        DeviceGuard g(h->device);
        if (!g.ok) return DVI_ESKF_EHIP;
        if (int rc = ensure_jcap(h, longest, (hipStream_t)stream)) return rc;
    }
    const size_t es = elem_size(h->dtype);
    const size_t B = (size_t)h->B;
    const size_t dt_row = (sc->dt_per_instance ? B : 1) * es;
    const size_t imu_row = (sc->imu_per_instance ? B : 1) * 3 * es;
    const size_t mrow = sc->meas_per_instance ? B : 1;
    const char* dt = static_cast<const char*>(sc->dt);
    const char* om = static_cast<const char*>(sc->om);
    const char* acc = static_cast<const char*>(sc->acc);
    const char* pos = static_cast<const char*>(sc->pos);
    const char* quat = static_cast<const char*>(sc->quat_xyzw);
    const char* notch = static_cast<const char*>(sc->notch);
    // An under-filled batch (the serial part of an epoch -- its recurrences -- leaves most of the chip idle): parts of the
    // batch run their epochs on streams of their own, so that one part's recurrences and Jacobian kernel run next to the
    // other parts' covariance kernels.  Instances do not interact, the kernels of a part touch that part's records only:
    // the results are the same bits.  DVI_ESKF_EPOCH_STREAMS=0 switches it off, =k (2..8) asks for k parts; two by default.
    // (Measured, config #5's schedule, us per camera frame fp32 / fp64 at B = 8 192 and 16 384: two parts 106 / 147 and 167 / 244,
    //  three 106 / 145 and 167 / 237, four 162 / 216 and 223 / 310, eight 241 / 284 and 270 / 370 -- from four parts on the
    //  enqueue, three launches per part and frame from one host thread, is what a frame waits for.)
    static const int streams_env = []() { const char* e = getenv("DVI_ESKF_EPOCH_STREAMS"); return e ? atoi(e) : -1; }();
    static const int64_t two_streams_max = []() { const char* e = getenv("DVI_ESKF_EPOCH_STREAMS_MAX"); return e ? (int64_t)atoll(e) : (int64_t)16384; }();
    // (measured, config #5's schedule, us per camera frame, one stream -> two: B = 1 024 72 -> 75 fp32 / 80 -> 82 fp64; 2 048
    //  78 -> 79 / 98 -> 86; 4 096 100 -> 83 / 115 -> 106; 8 192 129 -> 114 / 148 -> 146; 16 384 184 -> 167 / 256 -> 246)
    static const int64_t two_streams_min = []() { const char* e = getenv("DVI_ESKF_EPOCH_STREAMS_MIN"); return e ? (int64_t)atoll(e) : (int64_t)2048; }();
    int shortest = 0x7fffffff;
    for (int64_t e = 0; e < sc->n_epochs; ++e) shortest = sc->n_pred[e] < shortest ? sc->n_pred[e] : shortest;
    int n_parts = streams_env >= 0 ? streams_env : 2;
    if (n_parts > dvi_eskf::MAX_SIDE) n_parts = dvi_eskf::MAX_SIDE;
    while (n_parts > 1 && (h->B % (64 * n_parts)) != 0) --n_parts;       // (parts are multiples of a wavefront of instances)
    const int64_t part = n_parts > 1 ? h->B / n_parts : 0;
    // (whether the kernels can run a part of the batch: asked where the launcher decides it, eskf_kernels.hip)
    const bool split = !src && n_parts > 1 && !sc->rows_out && h->B >= two_streams_min && h->B <= two_streams_max &&
                       sc->n_epochs >= 2 &&
                       eskf::epoch_sub_batches_ok(h->lpi, part, part, shortest, h->d_noise != nullptr, false, h->d_aux != nullptr);
    if (split) {
        static const bool verbose = []() { const char* e = getenv("DVI_ESKF_VERBOSE"); return e && e[0] == '1'; }();
        static bool said = false;
        if (verbose && !said) {
            said = true;
            fprintf(stderr, "dvi_eskf: runs of epochs go epoch by epoch, %d parts of the batch on %d streams\n", n_parts, n_parts);
        }
        DeviceGuard g(h->device);
        hipStream_t cs = (hipStream_t)stream;
        if (!g.ok || !ensure_side_streams(h)) return DVI_ESKF_EHIP;
        if (hipEventRecord(h->ev_fork, cs) != hipSuccess) return DVI_ESKF_EHIP;     // (nothing forked yet: nothing to join)
        // From here on EVERY exit goes through the join below: a side stream that has been made to wait for the fork
        // event holds work that is unordered with respect to `stream` (and, under capture, belongs to the capture: an
        // unjoined stream makes hipStreamEndCapture fail) until `stream` waits for its join event.  The first error is
        // what the call returns.
        int rc_first = DVI_ESKF_OK;
        bool forked[dvi_eskf::MAX_SIDE] = {};
        for (int s2 = 0; s2 < n_parts && rc_first == DVI_ESKF_OK; ++s2) {
            if (hipStreamWaitEvent(h->side[s2], h->ev_fork, 0) != hipSuccess) rc_first = DVI_ESKF_EHIP;
            else forked[s2] = true;
        }
#ifdef ESKF_TEST_HOOKS      // (test build only, tests/test_gpu_capi.py: DVI_ESKF_TEST_FAIL_SPLIT_EPOCH=<e> makes the enqueue of epoch e fail)
        const char* fail_env = getenv("DVI_ESKF_TEST_FAIL_SPLIT_EPOCH");
        const int64_t fail_epoch = fail_env ? (int64_t)atoll(fail_env) : -1;
#else
        const int64_t fail_epoch = -1;
#endif
        size_t off2 = 0;
        for (int64_t e = 0; e < sc->n_epochs && rc_first == DVI_ESKF_OK; ++e) {     // (enqueued epoch by epoch, in turn: every queue always has work)
            const int n = sc->n_pred[e];
            for (int s2 = 0; s2 < n_parts && rc_first == DVI_ESKF_OK; ++s2) {
                if (e == fail_epoch && s2 == 1) { rc_first = DVI_ESKF_EHIP; break; }
                rc_first = step_common(h, n, 1, dt + off2 * dt_row, sc->dt_per_instance, om + off2 * imu_row, acc + off2 * imu_row,
                                       sc->imu_per_instance, pos + (size_t)e * mrow * 3 * es, quat + (size_t)e * mrow * 4 * es,
                                       notch + (size_t)e * mrow * es, sc->meas_per_instance, nullptr,
                                       sc->status ? sc->status + (size_t)e * B : nullptr, h->side[s2], nullptr, s2 * part, part);
            }
            off2 += (size_t)n;
        }
        for (int s2 = 0; s2 < n_parts; ++s2) {
            if (!forked[s2]) continue;
            if (hipEventRecord(h->ev_join[s2], h->side[s2]) != hipSuccess || hipStreamWaitEvent(cs, h->ev_join[s2], 0) != hipSuccess) {
                if (rc_first == DVI_ESKF_OK) rc_first = DVI_ESKF_EHIP;
            }
        }
        if (stream_is_capturing(cs)) h->captured = true;
        return rc_first;
    }
    size_t off = 0;
    for (int64_t e = 0; e < sc->n_epochs; ++e) {
        const int n = sc->n_pred[e];
        int rc = step_common(h, n, 1, n ? dt + off * dt_row : nullptr, sc->dt_per_instance, n ? om + off * imu_row : nullptr,
                             n ? acc + off * imu_row : nullptr, sc->imu_per_instance, pos + (size_t)e * mrow * 3 * es,
                             quat + (size_t)e * mrow * 4 * es, notch + (size_t)e * mrow * es, sc->meas_per_instance,
                             nullptr, sc->status ? sc->status + (size_t)e * B : nullptr, stream, e == 0 ? src : nullptr);
        if (rc != DVI_ESKF_OK) return rc;
        if (sc->rows_out) {
            rc = dvi_eskf_log_rows(h, sc->t_log[e], static_cast<char*>(sc->rows_out) + (size_t)e * B * 30 * es, stream);
            if (rc != DVI_ESKF_OK) return rc;
        }
        off += (size_t)n;
    }
    return DVI_ESKF_OK;
}
}  // namespace

extern "C" {

namespace {
// RCCL is bound at first use: the library must call into the SAME RCCL image that made the
// caller's communicator (a process that imported torch already holds one), and a filter
// process that never gathers should not have to load RCCL at all.
typedef int (*nccl_all_gather_fn)(const void*, void*, size_t, int, void*, hipStream_t);
nccl_all_gather_fn bind_all_gather() {
    static nccl_all_gather_fn fn = []() -> nccl_all_gather_fn {
        void* sym = dlsym(RTLD_DEFAULT, "ncclAllGather");
        if (!sym) {
            void* lib = dlopen("librccl.so.1", RTLD_LAZY | RTLD_NOLOAD);     // already in the process?
            if (!lib) lib = dlopen("librccl.so", RTLD_LAZY | RTLD_NOLOAD);
            if (!lib) lib = dlopen("librccl.so.1", RTLD_LAZY | RTLD_LOCAL);
            if (!lib) lib = dlopen("librccl.so", RTLD_LAZY | RTLD_LOCAL);
            if (lib) sym = dlsym(lib, "ncclAllGather");
        }
        return reinterpret_cast<nccl_all_gather_fn>(sym);
    }();
    return fn;
}
}  // namespace

int dvi_eskf_gather_calibration(dvi_eskf* h, void* dofs_all, void* rccl_comm, void* stream) {
    if (!h || !dofs_all || !rccl_comm) return DVI_ESKF_EINVAL;
    DeviceGuard g(h->device);
    if (!g.ok) return DVI_ESKF_EHIP;
    nccl_all_gather_fn all_gather = bind_all_gather();
    if (!all_gather) return DVI_ESKF_EHIP;
    const size_t es = elem_size(h->dtype);
    const size_t count = (size_t)h->B * 6;
    void* local = nullptr;                                  // the send block must not alias the receive buffer
    if (hipMallocAsync(&local, count * es, (hipStream_t)stream) != hipSuccess) return DVI_ESKF_ENOMEM;
    int rc = dvi_eskf_get_calibration(h, local, stream);
    if (rc == DVI_ESKF_OK) {
        const int nccl_dtype = h->dtype == DVI_ESKF_F64 ? 8 /* ncclFloat64 */ : 7 /* ncclFloat32 */;
        if (all_gather(local, dofs_all, count, nccl_dtype, rccl_comm, (hipStream_t)stream) != 0) rc = DVI_ESKF_EHIP;
    }
    (void)hipFreeAsync(local, (hipStream_t)stream);
    return rc;
}

int dvi_eskf_reserve(dvi_eskf* h, int n_pred) {
    if (!h || n_pred < 0) return DVI_ESKF_EINVAL;
    DeviceGuard g(h->device);
    if (!g.ok) return DVI_ESKF_EHIP;
    return ensure_jcap(h, n_pred, nullptr);
}

int dvi_eskf_keep_jacobians(dvi_eskf* h, int on) {
    if (!h) return DVI_ESKF_EINVAL;
    // the record format (with / without the N6 block) is a recorded kernel argument of every captured step: a
    // replay after a change would leave a record that is half one format, half the other
    if (h->captured && (on ? 1 : 0) != h->hp.keep_jac) return DVI_ESKF_ECAPTURE;
    h->hp.keep_jac = on ? 1 : 0;
    h->jac_ip = -1;
    return DVI_ESKF_OK;
}

int dvi_eskf_get_jacobians(const dvi_eskf* h, void* Fx_out, void* Fi_out, void* stream) {
    if (!h || !Fx_out || !Fi_out) return DVI_ESKF_EINVAL;
    if (h->jac_ip < 0) return DVI_ESKF_EINVAL;     // no predict since dvi_eskf_keep_jacobians(h, 1)
    DeviceGuard g(h->device);
    if (!g.ok) return DVI_ESKF_EHIP;
    hipError_t e = eskf::launch_jacobians(h->dtype, h->B, h->d_J, h->jac_ip, Fx_out, Fi_out, h->hp.quirks, (hipStream_t)stream);
    return e == hipSuccess ? DVI_ESKF_OK : DVI_ESKF_EHIP;
}

int dvi_eskf_flush(dvi_eskf* h, void* stream) {
    if (!h) return DVI_ESKF_EINVAL;
    DeviceGuard g(h->device);
    if (!g.ok) return DVI_ESKF_EHIP;
    return flush_pending(h, (hipStream_t)stream);
}

void* dvi_eskf_raw_P(dvi_eskf* h) { return h ? h->d_P : nullptr; }
void* dvi_eskf_raw_x(dvi_eskf* h) { return h ? h->d_x : nullptr; }

int dvi_eskf_set_lanes_per_instance(dvi_eskf* h, int lanes) {
    if (!h) return DVI_ESKF_EINVAL;
    if (lanes != 1 && lanes != 2 && lanes != 32) return DVI_ESKF_EINVAL;
    h->lpi = lanes;
    return DVI_ESKF_OK;
}

int dvi_eskf_log_rows(dvi_eskf* h, double t, void* rows_out, void* stream) {
    if (!h || !rows_out) return DVI_ESKF_EINVAL;
    DeviceGuard g(h->device);
    if (!g.ok) return DVI_ESKF_EHIP;
    if (flush_pending(h, (hipStream_t)stream) != DVI_ESKF_OK) return DVI_ESKF_EHIP;
    hipError_t e = eskf::launch_log_rows(h->dtype, h->B, h->B, h->d_x, t, rows_out, (hipStream_t)stream);
    return e == hipSuccess ? DVI_ESKF_OK : DVI_ESKF_EHIP;
}

int dvi_eskf_update_mse(dvi_eskf* h, const double* cam_ref6, const double* imu_ref6, void* mse_out, void* stream) {
    if (!h || !cam_ref6 || !mse_out) return DVI_ESKF_EINVAL;
    DeviceGuard g(h->device);
    if (!g.ok) return DVI_ESKF_EHIP;
    if (flush_pending(h, (hipStream_t)stream) != DVI_ESKF_OK) return DVI_ESKF_EHIP;
    hipError_t e = eskf::launch_update_mse(h->dtype, h->B, h->d_x, cam_ref6, imu_ref6, mse_out, (hipStream_t)stream);
    return e == hipSuccess ? DVI_ESKF_OK : DVI_ESKF_EHIP;
}

int dvi_eskf_imu_synthesize(const dvi_eskf_params* p, int dtype, int device, int64_t n, const void* R_WC,
                            const void* om_CW, const void* acc_CW, const void* alp_CW, const void* notch,
                            int notch_per_sample, const double* dofs6, const void* p_CW, const void* v_CW, void* om_B,
                            void* acc_B, void* p_BW, void* R_WB, void* v_BW, void* stream) {
    if (!p || n < 0 || !dofs6) return DVI_ESKF_EINVAL;
    if (dtype != DVI_ESKF_F64 && dtype != DVI_ESKF_F32) return DVI_ESKF_EINVAL;
    if (n == 0) return DVI_ESKF_OK;
    if (!R_WC || !om_CW || !acc_CW || !alp_CW || !notch || !om_B || !acc_B) return DVI_ESKF_EINVAL;
    if ((p_BW && !p_CW) || (v_BW && !v_CW)) return DVI_ESKF_EINVAL;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev) return DVI_ESKF_ENODEV;
    DeviceGuard g(device);
    if (!g.ok) return DVI_ESKF_EHIP;
    eskf::HostParams hp{};
    hp.L = p->L; hp.theta_cam = p->theta_cam;
    hp.quirks = eskf::QK_V1;
    hipError_t e = eskf::launch_imu(dtype, n, R_WC, om_CW, acc_CW, alp_CW, notch, notch_per_sample ? 1 : 0, p_CW, v_CW,
                                    om_B, acc_B, p_BW, R_WB, v_BW, dofs6, hp, (hipStream_t)stream);
    return e == hipSuccess ? DVI_ESKF_OK : DVI_ESKF_EHIP;
}

int dvi_eskf_randn(int kind, int device, uint64_t seed, int64_t first_instance, int64_t n_instances, int64_t first_step,
                   int64_t n_steps, int lanes, void* out, void* stream) {
    if (kind < 0 || kind > 2 || first_instance < 0 || n_instances < 0 || first_step < 0 || n_steps < 0 || lanes < 0)
        return DVI_ESKF_EINVAL;
    if (first_step + n_steps > (int64_t)1 << 32) return DVI_ESKF_EINVAL;       // the step is one counter word
    if (n_instances == 0 || n_steps == 0 || lanes == 0) return DVI_ESKF_OK;
    if (!out) return DVI_ESKF_EINVAL;
    if ((double)n_instances * (double)n_steps * (double)lanes >= 9.0e18) return DVI_ESKF_EINVAL;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev) return DVI_ESKF_ENODEV;
    DeviceGuard g(device);
    if (!g.ok) return DVI_ESKF_EHIP;
    hipError_t e = eskf::launch_randn(kind, seed, first_instance, n_instances, first_step, n_steps, lanes, out,
                                      (hipStream_t)stream);
    return e == hipSuccess ? DVI_ESKF_OK : DVI_ESKF_EHIP;
}

#if defined(ESKF_X_EPDUMP) || defined(ESKF_X_EPSTAMPS)
int dvi_eskf_x_ep_buffer(void* device_buffer) { g_x_ep_buffer = static_cast<unsigned long long*>(device_buffer); return 0; }
void* dvi_eskf_x_raw_J(dvi_eskf* h) { return h ? h->d_J : nullptr; }
#endif
#ifdef ESKF_X_RUNSTAMPS
int dvi_eskf_x_run_stamps(unsigned long long* out) { return eskf::x_read_run_stamps(out) == hipSuccess ? 0 : -4; }
int dvi_eskf_x_pre_stamps(unsigned long long* out) { return eskf::x_read_pre_stamps(out) == hipSuccess ? 0 : -4; }
#endif
#ifdef ESKF_X_STAMPS
int dvi_eskf_x_stamps(unsigned long long* out) { return eskf::x_read_stamps(out) == hipSuccess ? 0 : -4; }
int dvi_eskf_x_cstamps(unsigned long long* out) { return eskf::x_read_cstamps(out) == hipSuccess ? 0 : -4; }
#endif

const char* dvi_eskf_strerror(int code) {
    switch (code) {
        case DVI_ESKF_OK: return "ok";
        case DVI_ESKF_EINVAL: return "invalid argument";
        case DVI_ESKF_ENODEV: return "no usable HIP device";
        case DVI_ESKF_ENOMEM: return "device memory allocation failed";
        case DVI_ESKF_EHIP: return "HIP runtime call or kernel launch failed";
        case DVI_ESKF_ECAPTURE: return "a step of this handle is being or has been captured into a HIP graph and the call would have to "
                                       "replace a buffer the graph refers to: reserve / set_noise / keep_jacobians before the first capture";
        default: return "unknown error";
    }
}

}  // extern "C"
