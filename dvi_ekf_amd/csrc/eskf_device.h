// eskf_device.h -- types shared by the HIP kernels (eskf_kernels.hip) and the C-ABI
// host layer (eskf_capi.cpp).  Internal; the public interface is include/dvi_eskf.h.
#pragma once

#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

namespace eskf {

constexpr int DVI_STATUS_SINGULAR = 1;
constexpr int DVI_STATUS_NONFINITE = 2;
constexpr int DVI_STATUS_PENDING = 1 << 30;   // workspace word only: the update's error state is not injected yet
constexpr int JAC_WORDS = 96;  // words per Jacobian record (eskf_kernels.hip: JW)
constexpr int AUX_WORDS0 = 12, AUX_WORDSK = 8;   // eskf_kernels.hip: AUX0, AUXK
// The quirks of the v1 source that can be switched off on the device (include/dvi_eskf.h): bit set = the v1
// behaviour is in force.  Q3 / Q6 / Q9 are wave-uniform branches of the lane-per-instance chains; Q1 changes where the camera
// rows of Fx take their operands and is a template parameter of the covariance kernels that serve it.
constexpr int QK_Q1 = 1, QK_Q3 = 2, QK_Q6 = 4, QK_Q9 = 8, QK_V1 = 15;
constexpr int NOISE_WORDS = 24; // per-instance noise record: diag(Q) 13 | diag(R) 7 | sigma_om 3 | pad   // words per Jacobian record (eskf_kernels.hip: JW)

// constants of one filter configuration, in the kernel's working precision; passed by value
// as a kernel argument (lives in SGPRs / the kernarg segment)
template <typename T>
struct DevParams {
    T L, st, ct;        // probe length, sin / cos of the camera angle
    T sig_om[3];        // gyro noise std (Jacobian evaluation point, Filter.py:330-333)
    T qd[13];           // diag(Q)  (Filter.py:68-72)
    T r[7];             // diag(R)  (Filter.py:75)
    int frozen_mask;    // bit i: DOF i frozen
    int hold_frozen;    // 0: zero frozen DOFs after predict (Filter.py:243-245); 1: hold
    int q_gyro;         // diag(Q)[3:6] has a non-zero entry
    int jfull;          // the pre kernels write all 85 block words of a record (with the N6 block): q_gyro, or the
                        // handle keeps the Jacobians for dvi_eskf_get_jacobians
    int quirks;         // QK_* bits: which of the switchable v1 quirks are in force (QK_V1 = all: the reference as it is)
};

struct HostParams {
    double L, theta_cam;
    double sig_a[3], sig_om[3];
    double qd[13], r[7];
    int frozen_mask, hold_frozen;
    int q_gyro_any;     // per-instance noise: some instance has a non-zero gyro block in Q
    int keep_jac;       // dvi_eskf_keep_jacobians: full Jacobian records although Q has no gyro block
    int quirks;         // QK_* bits in force (QK_V1 unless dvi_eskf_params.quirks_fixed switches some off)
};

template <typename T>
inline DevParams<T> make_dev_params(const HostParams& h) {
    DevParams<T> d;
    d.L = (T)h.L;
    d.st = (T)sin(h.theta_cam);
    d.ct = (T)cos(h.theta_cam);
    for (int i = 0; i < 3; ++i) d.sig_om[i] = (T)h.sig_om[i];
    for (int i = 0; i < 13; ++i) d.qd[i] = (T)h.qd[i];
    for (int i = 0; i < 7; ++i) d.r[i] = (T)h.r[i];
    d.frozen_mask = h.frozen_mask;
    d.hold_frozen = h.hold_frozen;
    d.q_gyro = (h.qd[3] != 0.0 || h.qd[4] != 0.0 || h.qd[5] != 0.0 || h.q_gyro_any) ? 1 : 0;
    d.jfull = (d.q_gyro || h.keep_jac) ? 1 : 0;
    d.quirks = h.quirks;
    return d;
}

// arguments of one launch of the step kernel: n_pred predicts followed by an optional update
struct StepArgs {
    void* P;            // [B][300] packed covariance
    void* xs;           // [B][36]  state record
    const void* dt;     // [n_pred][B] (dt_stride = 1) or [n_pred]
    const void* om;     // [n_pred][B][3] (imu_stride = 1) or [n_pred][3]
    const void* acc;
    const void* pos;    // [B][3] (meas_stride = 1) or [3]
    const void* quat;   // [B][4] or [4]
    const void* notch;  // [B] or [1]
    void* K_out;        // nullable [B][24][7]
    int32_t* status;    // nullable [B] (caller's)
    // workspace owned by the handle
    void* J;            // [n_pred][B][96] records written by the pre kernel: residual of the update (7 + non-finite flag;
                        // first predict's record) | Jacobian blocks
    void* aux;          // [12 + 8 n_pred][B] what eskf_nominal_kernel hands to eskf_jac_kernel (epochs)
    void* e;            // [B][24] error state of the update
    int32_t* status_ws; // [B]
    int64_t B;
    int n_pred;
    int do_update;
    int dt_stride, imu_stride, meas_stride;
    int rounds;         // instances per wave slot in the covariance kernel (set by the launcher)
    int pending_hint;   // 0: the host knows that no instance has DVI_STATUS_PENDING set (nothing to fetch);
                        // 1: look at the status words
    int drop_pending;   // step_from: a pending injection of this handle is dropped with the state it belonged to
    const void* noise;  // nullable [B][NOISE_WORDS]: per-instance diag(Q) 13 | diag(R) 7 | sigma_om 3 | pad
    // where the step READS the covariance and the state record (another handle's buffers for
    // dvi_eskf_step_from; otherwise the handle's own P / xs -- the step is then in place)
    const void* P_src;
    const void* xs_src;
    // a launch over a PART of the batch: instances [sub_first, sub_first + sub_count), both multiples of 64 (sub_count = 0:
    // the whole batch).  B stays the stride of every array.  Honoured by eskf_nominal_kernel, eskf_jac_kernel and the
    // row-resident covariance kernel with 32 lanes per instance (dvi_eskf_run_epochs pipelines two halves of an
    // under-filled batch on two streams with it).
    int64_t sub_first, sub_count;
};

// A run of n_steps consecutive predict(+update) steps in ONE launch (eskf_run_kernel): the StepArgs describe step 0, the
// inputs of step k lie k strides further on (bytes; 0 = the same sample for every step), the caller's status words of
// step k at status + k * status_step (elements).  From step 1 on the step reads the handle's own P / xs.
struct RunArgs {
    int n_steps, pad;
    int64_t dt_step, imu_step, pos_step, quat_step, notch_step;
    int64_t status_step;
};
// whether launch_run would take this run (same coverage as the one-launch step: one predict per step, shared noise
// model, default covariance kernel, no gain output, no kept Jacobians); 0 = no, else instances per block
int run_kernel_block(int dtype, int lpi, const StepArgs& a, const HostParams& hp);
// the one-launch step (eskf_step_kernel): instances per block if it takes this launch, else 0; and its launcher
int step_kernel_block(int dtype, int lpi, const StepArgs& a, const HostParams& hp);
hipError_t launch_step_fused(int dtype, const StepArgs& a, const HostParams& hp, hipStream_t stream);
hipError_t launch_run(int dtype, int lpi, const StepArgs& a, const RunArgs& ra, const HostParams& hp, hipStream_t stream);

// A whole run of epochs -- n_pred[e] >= 0 predicts, then the update with measurement e -- in ONE launch of
// eskf_epochs_kernel (eskf_epochs.hip): a block keeps its instances for the run, producer wavefronts run the nominal
// recurrences and the Jacobian blocks ahead of the consumer wavefronts that hold the covariances in registers.  The
// StepArgs describe epoch 0 / predict 0 (inputs as [total] / [n_epochs] arrays, strides as in StepArgs: predict g of the
// run at g rows, epoch e's measurement and status words at e rows).
struct EpochsArgs {
    StepArgs a;
    const int32_t* n_pred;   // DEVICE [n_epochs], or null when every epoch has n_uniform predicts
    int32_t n_epochs;
    int32_t n_total;         // sum of n_pred
    int32_t n_uniform;
    int32_t xlog_mode;       // 0: no log | 1: a state snapshot after every predict (row g of the run) | 2: after every epoch (row e)
    void* xlog;              // [rows][26][n_log]: x26 of instances [0, n_log) -- after predict g the propagated state (Filter.py:230);
                             // the snapshot an update belongs to (the epoch's last predict / the epoch) is overwritten with the
                             // corrected state when the next epoch has injected it (Filter.py:393); the run's last one is left to the
                             // caller (dvi_eskf_flush + the state record)
    int64_t n_log;
    unsigned long long* stamps;   // diagnostic builds (ESKF_X_EPSTAMPS): [blocks][16 waves][8] cycle counters
};
// instances per block if the long-horizon kernel takes this run (shared noise model, default covariance kernel, no gain
// output, no kept Jacobians, no gyro block in Q), else 0
int epochs_kernel_block(int dtype, int lpi, const StepArgs& a, const HostParams& hp);
hipError_t launch_epochs(int dtype, int lpi, const EpochsArgs& ea, const HostParams& hp, hipStream_t stream);

hipError_t launch_step(int dtype, int lpi, const StepArgs& a, const HostParams& hp, hipStream_t stream);
hipError_t launch_flush(int dtype, const StepArgs& a, const HostParams& hp, hipStream_t stream);
bool epoch_sub_batches_ok(int lpi, int64_t sub_first, int64_t sub_count, int n_pred, bool per_instance_noise, bool gain_out, bool have_aux);
hipError_t launch_copy_instances(int dtype, const void* srcP, const void* srcx, int64_t srcB, int64_t first, void* dstP, void* dstx, int64_t dstB,
                                 hipStream_t stream);
hipError_t launch_copy_state(const void* srcP, void* dstP, size_t bytesP, const void* srcx, void* dstx, size_t bytesx, hipStream_t stream);
hipError_t launch_pack(int dtype, int64_t B, const void* x26, const void* P576, const void* om3, const void* acc3,
                       const void* qo4, void* xs, void* Ppk, hipStream_t stream);
hipError_t launch_unpack(int dtype, int64_t B, const void* xs, const void* Ppk, void* x26, void* P576, void* om3,
                         void* acc3, void* qo4, void* dofs6, hipStream_t stream);

hipError_t launch_imu(int dtype, int64_t n, const void* R_WC, const void* om, const void* acc, const void* alp,
                      const void* notch, int notch_stride, const void* p_CW, const void* v_CW, void* om_B, void* acc_B,
                      void* p_BW, void* R_WB, void* v_BW, const double* dofs6, const HostParams& hp, hipStream_t stream);

// rows [n_log][30] of the batch's first n_log instances (n_log <= B) from the state record xs [36][B]
hipError_t launch_log_rows(int dtype, int64_t B, int64_t n_log, const void* xs, double t, void* rows, hipStream_t stream);
// rows [n_rows][n_log][30] from the snapshots xlog [n_rows][26][n_log] of a run (EpochsArgs::xlog) -- the last row from the state
// record xs [36][B] itself (flushed) -- with the time stamps t_rows (device, double [n_rows])
hipError_t launch_log_rows_run(int dtype, int64_t n_rows, int64_t n_log, const void* xlog, const void* xs, int64_t B, const double* t_rows,
                               void* rows, hipStream_t stream);

// Filter.calculate_update_mse per instance: cam6 / imu6 host references (imu6 nullable), mse [B] device
hipError_t launch_update_mse(int dtype, int64_t B, const void* xs, const double* cam6, const double* imu6, void* mse,
                             hipStream_t stream);

// Fx [B,24,24] and Fi [B,24,13] of the predict whose full record (85 block words) is J[ip]
hipError_t launch_jacobians(int dtype, int64_t B, const void* J, int ip, void* Fx, void* Fi, int quirks, hipStream_t stream);

// eskf_noise.hip -- kind: 0 = normals f64, 1 = normals f32, 2 = raw Philox words (uint32 x 4 per triple)
hipError_t launch_randn(int kind, uint64_t seed, int64_t first_instance, int64_t n_instances, int64_t first_step,
                        int64_t n_steps, int lanes, void* out, hipStream_t stream);

}  // namespace eskf
