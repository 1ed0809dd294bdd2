/* dvi_eskf.h -- C ABI of the MI355X batched error-state Kalman filter.
 *
 * The reference (salehahr/dvi-ekf @ v1) is a pure-Python package with no FFI; the
 * boundary this library sits behind is the Python class `Filter`
 * (dvi_ekf/filter/Filter.py:28-476).  Each entry point below names the reference
 * method(s) it replaces.  Python binding: dvi_ekf_amd/_capi.py (ctypes); see
 * INTEGRATION.md for the stub a maintainer of the reference would add.
 *
 * Conventions
 *  - every function returns 0 on success or a negative DVI_ESKF_E* code; nothing
 *    throws across the ABI; dvi_eskf_strerror() explains a code.
 *  - all array arguments are HIP *device* pointers (e.g. torch tensor.data_ptr())
 *    of the handle's dtype, C-contiguous, batch-major; `stream` is a hipStream_t
 *    (NULL = default stream).  Work is enqueued asynchronously on that stream.
 *    Small constants (params, Q, R) are host pointers.
 *  - the library never frees caller buffers; one handle is not thread-safe,
 *    distinct handles are independent.
 *  - state vector x26 = p(3) v(3) q(xyzw 4) dofs(6) notch(3) p_cam(3) q_cam(xyzw 4)
 *    (dvi_ekf/filter/state.py:11-86); error state = 24 (state.py:89-129).
 *  - behaviour: the arithmetic is that of the v1 source including its quirks Q1-Q10 of SURVEY.md
 *    App. A.6 (column cursor of the camera rows, v_tr := p_tr, Jacobians at n = sigma, half-angle
 *    residual, stale R_WB_old, ...): the reference's saved trajectory pins every one of them, and that
 *    is the default.  The four a maintainer of the reference would fix first can be switched off per
 *    handle, dvi_eskf_params.quirks_fixed (DVI_ESKF_FIX_Q1 / Q3 / Q6 / Q9 below) -- there is no reference
 *    artefact for a "fixed" filter; those variants are checked against the restatement with the same quirk
 *    switched off (oracle/eskf_ref.py, tests/test_gpu_quirks.py).  The plumbing switch is hold_frozen (G1).
 *
 * HIP graphs.  predict / update / step / epoch / step_from / run_epochs / run_epochs_from / flush / get_* / copy_state only
 * enqueue kernels and async copies on `stream` and may be recorded with hipStreamBeginCapture and
 * replayed any number of times: whether an update's error injection is still due is kept on the DEVICE
 * (a bit of the per-instance status word), not in the handle, so a replayed sequence continues from
 * whatever the previous replay left behind exactly as the same calls made eagerly would.  Four rules:
 *  (1) an epoch longer than any the handle has run needs a larger workspace -- an allocation, which a
 *      capturing stream cannot do: call dvi_eskf_reserve(h, n_pred) (or run one such epoch) before the
 *      capture, else the call returns DVI_ESKF_ECAPTURE and nothing is recorded;
 *  (2) create, clone, destroy, set_noise*, reserve, imu_synthesize and randn allocate or synchronise: not
 *      capturable;
 *  (3) pointers recorded in a graph (inputs, K_out, status, handles) must stay valid for its replays;
 *  (4) so must the handle's OWN buffers: once a step of a handle has been captured, every call that would
 *      replace a buffer or a record format the graph refers to -- a longer epoch or dvi_eskf_reserve beyond
 *      the workspace size (the Jacobian workspace would be re-allocated), dvi_eskf_set_noise on a handle
 *      that carries per-instance noise (that buffer would be freed), a change of dvi_eskf_keep_jacobians
 *      (the record format is a recorded argument) -- returns DVI_ESKF_ECAPTURE and changes nothing, for the
 *      rest of the handle's life: reserve, choose the noise model and keep_jacobians BEFORE the first capture.
 *      dvi_eskf_get_jacobians refers to the last predict enqueued through the API, not to what a replay ran last.
 */
#ifndef DVI_ESKF_H
#define DVI_ESKF_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DVI_ESKF_NX 26
#define DVI_ESKF_NE 24
#define DVI_ESKF_NN 13
#define DVI_ESKF_NM 7

#define DVI_ESKF_F64 0
#define DVI_ESKF_F32 1

#define DVI_ESKF_OK 0
#define DVI_ESKF_EINVAL (-1)    /* bad argument (null handle, bad dtype, non-diagonal Q/R, ...) */
#define DVI_ESKF_ENODEV (-2)    /* no usable HIP device                                         */
#define DVI_ESKF_ENOMEM (-3)    /* device allocation failed                                     */
#define DVI_ESKF_EHIP (-4)      /* a HIP runtime call / kernel launch failed                    */
#define DVI_ESKF_ECAPTURE (-5)  /* a step of the handle is being / has been captured into a HIP graph and
                                   the call would have to allocate, synchronise or replace a buffer
                                   the graph refers to (see "HIP graphs" above, rules 1 and 4)   */

/* per-instance status word written by update / step */
#define DVI_ESKF_STATUS_SINGULAR 1  /* innovation covariance not factorisable: state and P left
                                       untouched -- the `return None` path of Filter.py:356-361 */
#define DVI_ESKF_STATUS_NONFINITE 2 /* asin argument clamped, or a NaN / inf residual (measurement
                                       or state): it propagates into THIS instance as NumPy lets it
                                       in the reference; other instances are unaffected          */

typedef struct dvi_eskf dvi_eskf;

/* Constants that Filter.__init__ takes from the config / Simulator (Filter.py:34-93). */
typedef struct {
    double L;             /* probe length, config.yaml:33 (model.length)                   */
    double theta_cam;     /* camera angle [rad], config.yaml:34 / config.py:131-133        */
    int32_t frozen[6];    /* frozen-DOF mask, config.yaml:8; Filter.py:56,243-245,377-379  */
    int32_t hold_frozen;  /* 0: v1 source -- frozen DOFs are zeroed after every predict
                             (Filter.py:243-245); 1: frozen DOFs keep their value (what the
                             reference's saved kf_best_mandala0_mono.txt needs)            */
    double sigma_a[3];    /* accelerometer noise std (config.py:116-120)                   */
    double sigma_om[3];   /* gyro noise std (config.py:106-113); the Jacobians are evaluated
                             at n_om = sigma_om (Filter.py:330-333)                        */
    int32_t quirks_fixed; /* 0: the v1 source as it is.  DVI_ESKF_FIX_* bits switch a quirk off: */
} dvi_eskf_params;
/* Q1, kinematics/symbols.py:103,107 + Filter.py:279-285: the columns of the camera rows' Jacobian are placed one per
 *     VARIABLE (err_q counts as one) instead of one per error-state column: delta-p_C's identity lands in columns 16:19, the
 *     delta-theta_C block in 19:22.  Fixed: identity at 18:21, the block at 21:24.  Predicts of such a handle run through the
 *     row-resident covariance kernel only (default lanes, shared noise model, no gyro block in Q, no gain output together
 *     with a predict: DVI_ESKF_EINVAL otherwise) and one launch sequence per step / epoch.
 * Q3, filter/state.py:124: the camera error quaternion is built about the IMU's delta-theta axis.  Fixed: about delta-theta_C.
 * Q6, tools/Quaternion.py:138-147: the rotation residual is asin|v| x axis, HALF the rotation angle.  Fixed: the full angle.
 * Q9, kinematics/equations.py:50,86: the camera attitude is propagated with om_old.  Fixed: with the averaged rate, as the
 *     IMU attitude is. */
#define DVI_ESKF_FIX_Q1 1
#define DVI_ESKF_FIX_Q3 2
#define DVI_ESKF_FIX_Q6 4
#define DVI_ESKF_FIX_Q9 8

/* Filter.__init__ (Filter.py:34-93): allocates device state for `batch` independent
 * filter instances on HIP device `device`.  dtype: DVI_ESKF_F64 | DVI_ESKF_F32. */
int dvi_eskf_create(const dvi_eskf_params* params, int64_t batch, int dtype, int device, dvi_eskf** out);

/* copy.deepcopy(kf) (Simulator.py:150): independent copy incl. device buffers. */
int dvi_eskf_clone(const dvi_eskf* src, void* stream, dvi_eskf** out);

/* dst <- src (device-to-device copy of the state records; same batch and dtype).  The
 * checkpoint/restore primitive behind Filter.reset (Filter.py:95-108) and
 * Simulator.reset_kf (Simulator.py:160-161) when the initial state is kept on the device. */
int dvi_eskf_copy_state(dvi_eskf* dst, const dvi_eskf* src, void* stream);

/* dst <- the instances [first, first + dvi_eskf_batch(dst)) of src (device-to-device; same dtype and device; a pending
 * injection of src is applied first, one of dst is dropped).  How a driver that logs or plots ONE trajectory of a batch
 * (the reference-shaped Filter with config.batch > 1) starts a one-instance filter from instance 0's state without
 * unpacking the batch. */
int dvi_eskf_copy_instances(dvi_eskf* dst, const dvi_eskf* src, int64_t first, void* stream);

void dvi_eskf_destroy(dvi_eskf* h);

int64_t dvi_eskf_batch(const dvi_eskf* h);
int dvi_eskf_dtype(const dvi_eskf* h);

/* Filter.update_noise_matrices / the Q, R attributes (Filter.py:67-75,110-117).  Host
 * pointers, row-major, double.  Both must be diagonal (the reference only ever builds
 * diagonal Q and R); otherwise DVI_ESKF_EINVAL. */
int dvi_eskf_set_noise(dvi_eskf* h, const double* Q13x13, const double* R7x7);

/* A population of differently tuned filters in one batch (Simulator.optimise,
 * Simulator.py:163-245, evaluates one candidate of the differential evolution per run -- here one
 * candidate per instance).  HOST arrays: q13 [B,13] = diag(Q) and r7 [B,7] = diag(R) per instance
 * (Filter.py:67-75,110-117), sigma_om3 [B,3] = the gyro noise std-dev at which the Jacobians are
 * evaluated (Filter.py:330-333) or NULL to keep the one of dvi_eskf_create.  Steps then run through
 * the per-instance-noise build of the selected covariance kernel (row-resident by default: it reads the
 * instance's 24-word noise record next to the residual); dvi_eskf_set_noise returns to one model for
 * the whole batch.  Copies and synchronises: not capturable. */
int dvi_eskf_set_noise_per_instance(dvi_eskf* h, const double* q13, const double* r7, const double* sigma_om3,
                                    void* stream);

/* Filter.reset / the _states, _P, om_old, acc_old, R_WB_old attributes (Filter.py:44-45,
 * 78-80,95-108).  x26 [B,26], P576 [B,24,24] (symmetric; the lower triangle is stored),
 * om_old3 / acc_old3 [B,3]; q_old4 [B,4] is the attitude whose rotation matrix is
 * R_WB_old -- NULL means "take it from x" (Filter.py:80).  Any other NULL leaves that
 * part unchanged. */
int dvi_eskf_set_state(dvi_eskf* h, const void* x26, const void* P576, const void* om_old3,
                       const void* acc_old3, const void* q_old4, void* stream);
int dvi_eskf_get_state(const dvi_eskf* h, void* x26, void* P576, void* om_old3, void* acc_old3,
                       void* q_old4, void* stream);

/* Filter.propagate (Filter.py:219-230): _predict_nominal + _predict_error +
 * _predict_error_covariance + buffer swap, for every instance.
 * dt [B] or [1]; om, acc [B,3] or [1,3] (per_instance = 0 broadcasts one sample). */
int dvi_eskf_predict(dvi_eskf* h, const void* dt, int dt_per_instance, const void* om, const void* acc,
                     int imu_per_instance, void* stream);

/* Filter.update (Filter.py:351-395).  pos [B,3] | [1,3]; quat_xyzw [B,4] | [1,4] raw
 * (un-normalised) measurement quaternion; notch [B] | [1].  K_out: NULL or [B,24,7]
 * (the return value of Filter.update); status: NULL or int32 [B]. */
int dvi_eskf_update(dvi_eskf* h, const void* pos, const void* quat_xyzw, const void* notch,
                    int meas_per_instance, void* K_out, int32_t* status, void* stream);

/* One predict followed by one update in a single launch (P makes one HBM round trip):
 * the 1:1 epoch of Filter.run_one_epoch (Filter.py:170-185). */
int dvi_eskf_step(dvi_eskf* h, const void* dt, int dt_per_instance, const void* om, const void* acc,
                  int imu_per_instance, const void* pos, const void* quat_xyzw, const void* notch,
                  int meas_per_instance, void* K_out, int32_t* status, void* stream);

/* Filter.run_one_epoch with N IMU samples per camera frame (Filter.py:170-217): n_pred
 * predicts then one update in a single launch, P resident on-chip in between.
 * dt [n_pred,B] | [n_pred]; om/acc [n_pred,B,3] | [n_pred,3]. */
int dvi_eskf_epoch(dvi_eskf* h, int n_pred, const void* dt, int dt_per_instance, const void* om,
                   const void* acc, int imu_per_instance, const void* pos, const void* quat_xyzw,
                   const void* notch, int meas_per_instance, void* K_out, int32_t* status, void* stream);

/* dvi_eskf_epoch whose predicts start from ANOTHER handle's state: h <- epoch(src), src untouched.  What
 * `copy.deepcopy(kf)` / Simulator.reset_kf (Simulator.py:150,160-161) followed by a run does, without
 * the copy: a Monte-Carlo restart or a tuning candidate reads the common snapshot directly (P and the
 * state record are read from src's buffers and written to h's).  Same batch size, dtype and device;
 * h keeps its own noise model and parameters; a pending injection of src is applied to src first, one
 * of h is dropped with the state it belonged to.  n_pred >= 0; the update always runs. */
int dvi_eskf_step_from(dvi_eskf* h, dvi_eskf* src, int n_pred, const void* dt, int dt_per_instance, const void* om,
                       const void* acc, int imu_per_instance, const void* pos, const void* quat_xyzw,
                       const void* notch, int meas_per_instance, void* K_out, int32_t* status, void* stream);

/* Filter.run (Filter.py:144-168): a whole run -- every camera frame's epoch -- enqueued by one
 * call, so that a long run costs one trip through the host binding instead of one per frame.
 * Epoch e consumes n_pred[e] consecutive IMU samples and measurement e; same kernels, same
 * results, bit for bit, as n_epochs calls of dvi_eskf_epoch. */
typedef struct dvi_eskf_schedule {
    int64_t n_epochs;
    const int32_t* n_pred;  /* HOST  [n_epochs]  IMU samples in front of each camera frame (>= 0) */
    const void* dt;         /* device [total] | [total,B],     total = sum(n_pred)                 */
    int dt_per_instance;
    const void* om;         /* device [total,3] | [total,B,3]                                      */
    const void* acc;
    int imu_per_instance;
    const void* pos;        /* device [n_epochs,3] | [n_epochs,B,3]                                */
    const void* quat_xyzw;  /* device [n_epochs,4] | [n_epochs,B,4]  raw measurement quaternions   */
    const void* notch;      /* device [n_epochs]   | [n_epochs,B]                                  */
    int meas_per_instance;
    int32_t* status;        /* NULL or device int32 [n_epochs,B]: the status word of every update  */
    const double* t_log;    /* HOST  [n_epochs] time stamps of the log rows; needed with rows_out  */
    void* rows_out;         /* NULL or device [n_epochs,B,30]: dvi_eskf_log_rows after every update
                               (Filter.update appends to the trajectory log, Filter.py:394)       */
    /* The reference's trajectory log has one row per IMU sample (Filter.propagate appends the propagated state,
     * Filter.py:230) and Filter.update overwrites the last one with the corrected state (Filter.py:393, FilterTraj.py:66-69):
     * rows_pred_out [total, rows_instances, 30] is exactly that log for the instances [0, rows_instances) of the batch
     * (0 = all B; a driver that plots one trajectory asks for one), with the time stamps t_pred (HOST, [total]).  An epoch
     * without a predict overwrites the row of the predict before it.  Exclusive with rows_out. */
    const double* t_pred;
    void* rows_pred_out;
    int64_t rows_instances;
} dvi_eskf_schedule;
/* For an under-filled batch (2 048 <= B <= 16 384, B a multiple of 128, every epoch with at least two predicts, no
 * log rows, shared noise model, default covariance kernel) the two halves of the batch are enqueued on two streams
 * of the handle's own, forked from and joined to `stream` by events: one half's recurrences and Jacobian kernel
 * then run next to the other half's covariance kernel.  Instances do not interact: the results are bit for bit those
 * of the one-stream order, and the call stays capturable. */
int dvi_eskf_run_epochs(dvi_eskf* h, const dvi_eskf_schedule* sched, void* stream);
/* A run whose every epoch is one predict + the update (n_pred[e] == 1 throughout: the 1:1 schedule of the reference's
 * saved trajectory, kf_best_mandala0_mono.txt, and of the Monte-Carlo batches), no log rows, shared noise model, default
 * covariance kernel, is ONE launch: the time loop of Filter.run (Filter.py:144-168) runs inside the kernel
 * (eskf_run_kernel), every thread block stepping its own instances from frame to frame; results are bit for bit those of
 * the epoch-by-epoch calls.  Capturable like any other call.  Within such a run the covariances of a quarter of the
 * instances (half / all of them for batches up to 28 672 / 8 192) stay in on-chip memory from one frame to the next; what
 * the handle holds in device memory is complete again when the launch has finished -- as after any other call. */

/* dvi_eskf_run_epochs whose first epoch starts from the (flushed) state of `src`, as dvi_eskf_step_from does for one
 * step: h == copy_state(src) followed by run_epochs, bit for bit, without the copy.  src != NULL; same batch, dtype
 * and device as h. */
int dvi_eskf_run_epochs_from(dvi_eskf* h, dvi_eskf* src, const dvi_eskf_schedule* sched, void* stream);

/* The estimated calibration parameters, i.e. state.dofs (Filter.calculate_dof_metric,
 * Filter.py:452-455): dofs_out [B,6]. */
int dvi_eskf_get_calibration(const dvi_eskf* h, void* dofs_out, void* stream);

/* The one collective of a multi-GPU run (instances are independent): all-gather of
 * the calibration estimates of every rank.  dofs_all [world*B,6] in the handle's dtype, rank r's
 * block at rows r*B..; every rank must hold the same batch size.  rccl_comm is the caller's
 * ncclComm_t; the call is ncclAllGather on `stream`, RCCL being bound at first use from the
 * library image already in the process (or librccl.so.1).  DVI_ESKF_EHIP if RCCL cannot be
 * bound or reports an error.  Python callers with torch.distributed use dvi_ekf_amd/dist.py
 * (same collective through the process group). */
int dvi_eskf_gather_calibration(dvi_eskf* h, void* dofs_all, void* rccl_comm, void* stream);

/* The error injection of an update (state.py:46-60) is deferred: it is applied by the next
 * predict/update/step/epoch call, or by any call that reads or replaces the state (get_state,
 * set_state, get_calibration, clone, copy_state).  dvi_eskf_flush applies it explicitly -- needed
 * only before looking at dvi_eskf_raw_x().  Which instances still await their injection is device
 * state, so all of this also holds across HIP-graph replays (see "HIP graphs" above). */
int dvi_eskf_flush(dvi_eskf* h, void* stream);

/* Sizes the per-handle workspace for epochs of up to n_pred predicts (one Jacobian record per predict
 * and instance).  Optional -- epoch / run_epochs grow it on demand -- except before capturing such an
 * epoch into a HIP graph.  Allocates and synchronises the device. */
int dvi_eskf_reserve(dvi_eskf* h, int n_pred);

/* Filter.Fx / Filter.Fi (Filter.py:249-268; filled by _cam_error_jacobian, Filter.py:279-285,334-342): the
 * dense 24x24 error-state transition and 24x13 noise Jacobian of the LAST predict, per instance.  The
 * kernels never form these matrices; with dvi_eskf_keep_jacobians(h, 1) every predict leaves its blocks
 * (incl. the noise rows, which are otherwise only computed in tune mode) in the workspace and
 * dvi_eskf_get_jacobians expands them: Fx_out [B,24,24], Fi_out [B,24,13], row-major, handle dtype.
 * DVI_ESKF_EINVAL when no predict has run since keeping was switched on.  Keeping costs 18 more words
 * of workspace traffic per predict and instance; off by default. */
int dvi_eskf_keep_jacobians(dvi_eskf* h, int on);
int dvi_eskf_get_jacobians(const dvi_eskf* h, void* Fx_out, void* Fi_out, void* stream);

/* Raw device buffers (zero-copy views for profiling / checkpointing): packed lower
 * triangle of P [B,300] and the persistent state block, structure-of-arrays [36,B] with
 * words x26 | om_old3 | acc_old3 | q_old4. */
void* dvi_eskf_raw_P(dvi_eskf* h);
void* dvi_eskf_raw_x(dvi_eskf* h);

/* Trajectory log, the step behind the path: one 30-column row per instance in the
 * layout of trajectory/FilterTraj.py:12-41 (what Filter.save()/plot() and the saved golden
 * trajectory use): t | x y z vx vy vz rx ry rz qw qx qy qz | dof1..3 [deg] dof4..6 |
 * xc yc zc rx_c ry_c rz_c qwc qxc qyc qzc, Euler angles extrinsic xyz in degrees.
 * rows_out [B,30] in the handle's dtype; applies a pending injection first. */
int dvi_eskf_log_rows(dvi_eskf* h, double t, void* rows_out, void* stream);

/* Filter.calculate_update_mse (Filter.py:397-418), per instance, as a reduction on the device: the mean of
 * the 12 squared differences between the filter's camera pose (position, extrinsic-xyz Euler angles in
 * degrees) and the camera frame's, and between the filter's IMU velocity / Euler angles and the IMU
 * reference sample's.  HOST arrays: cam_ref6 = (x, y, z, rx_deg, ry_deg, rz_deg) of the camera frame
 * (VisualTraj columns, Filter.py:402-406), imu_ref6 = (vx, vy, vz, rx, ry, rz) of the last IMU reference
 * sample (ImuRefTraj columns, Filter.py:409-413) or NULL when no reference trajectory is kept (the IMU half
 * then contributes 0).  mse_out: device [B] in the handle's dtype.  Applies a pending injection first. */
int dvi_eskf_update_mse(dvi_eskf* h, const double* cam_ref6, const double* imu_ref6, void* mse_out, void* stream);

/* Input synthesis, the step in front of the path: noise-free IMU samples from the
 * camera trajectory through the probe kinematics -- f_imu_meas / f_imu of
 * kinematics/equations.py:8-41,54-69 as driven by Imu.eval_expr_single (models/Imu.py:141-196)
 * and Imu.ref_vals (Imu.py:305-315).  Stateless; one sample per lane.  Device pointers:
 *   R_WC [n,9] row-major camera attitude, om_CW / acc_CW / alp_CW [n,3] camera angular rate,
 *   acceleration and angular acceleration in W, notch [n,3] = (q7, q7', q7'') per sample or [3];
 *   dofs6: HOST array, the ground-truth joint values q1..q6 (Probe.py:382-410).
 * Outputs: om_B, acc_B [n,3] (body-frame rate and acceleration; no gravity, bias or noise), and --
 * when not NULL -- the reference pose p_BW [n,3] (needs p_CW), R_WB [n,9], v_BW [n,3] (needs v_CW).
 * Only L and theta_cam of the parameter block are used. */
int dvi_eskf_imu_synthesize(const dvi_eskf_params* p, int dtype, int device, int64_t n, const void* R_WC,
                            const void* om_CW, const void* acc_CW, const void* alp_CW, const void* notch,
                            int notch_per_sample, const double* dofs6, const void* p_CW, const void* v_CW,
                            void* om_B, void* acc_B, void* p_BW, void* R_WB, void* v_BW, void* stream);

/* Noise for the synthetic batch workloads: standard normals that are a pure function
 * of (seed, instance, step, lane) -- Philox4x32-10 with key = seed and counter = (instance lo,
 * instance hi, step, lane), Box-Muller (cosine branch) on two 53-bit uniforms -- so that instance i
 * of a sharded Monte-Carlo run (Simulator / montecarlo, configs #2 and #4) draws the numbers it
 * draws in the unsharded run.  Stateless.  out: device [n_steps, n_instances, lanes];
 * kind DVI_ESKF_RANDN_F64 / _F32: normals; DVI_ESKF_RANDN_BITS: the four raw 32-bit Philox words of
 * every triple, uint32 [n_steps, n_instances, lanes, 4] (for known-answer tests). */
#define DVI_ESKF_RANDN_F64 0
#define DVI_ESKF_RANDN_F32 1
#define DVI_ESKF_RANDN_BITS 2
int dvi_eskf_randn(int kind, int device, uint64_t seed, int64_t first_instance, int64_t n_instances,
                   int64_t first_step, int64_t n_steps, int lanes, void* out, void* stream);

/* Covariance kernel selector (for experiments and cross-checks).
 * DVI_ESKF_LANES_ROWS (default): row-resident kernel -- lane i of an instance keeps row i of P in
 * registers, 32 lanes per instance, two instances per wavefront.
 * 32: LDS-image kernel, 32 lanes per instance (the kernel that also forms the gain matrix; requests with a gain
 * output run through it whatever is selected).  All variants compute the same step; they differ in rounding only. */
#define DVI_ESKF_LANES_ROWS 1
#define DVI_ESKF_LANES_ROWS24 2   /* row-resident kernel, 24 lanes per instance: eight instances per block of three wavefronts */
int dvi_eskf_set_lanes_per_instance(dvi_eskf* h, int lanes);

const char* dvi_eskf_strerror(int code);

#ifdef __cplusplus
}
#endif
#endif /* DVI_ESKF_H */
