// eskf_kernels.hip -- CDNA4 (gfx950) kernels of the batched error-state Kalman filter.
//
// Replaces, per filter instance, the per-step arithmetic of the reference's
// Filter.propagate / Filter.update (dvi_ekf/filter/Filter.py:219-395) including the
// CasADi-evaluated Jacobians (Filter.py:270-342, kinematics/symbols.py:134-200) and the
// probe forward kinematics (models/Probe.py:147-167,431-480), in closed form
// (SURVEY.md App. A/B).
//
// One predict(+update) step = a pre kernel (lane per instance: nominal state, Jacobian blocks, residual --
// eskf_pre4_kernel for launches with one predict, eskf_nominal_kernel + eskf_jac_kernel for epochs of an under-filled
// batch, eskf_pre_kernel otherwise) -> eskf_cov_rows_kernel (32 lanes per instance, rows of P in registers: covariance
// propagation, the measurements one after the other, reset; eskf_cov_kernel<LPI> is the same step
// on an LDS image of P and also forms the gain matrix) ; the error injection of an update is applied
// by the next step's pre kernel or, when the state is read in between, by eskf_post_kernel.  See
// the "Work split" comment below and DESIGN.md section 4.
//
// One launch instead of two or three (round 4): eskf_step_kernel runs the same pre chains and the same covariance body
// in ONE launch -- a block's pre phase hands the records of its 16 / 32 / 64 instances to its own covariance rounds
// through LDS staging rows and registers, nothing of them crosses memory --, and eskf_run_kernel puts the time loop of
// Filter.run (Filter.py:144-168) around that for runs with one IMU sample per camera frame: a whole run is one launch,
// the error states and status words of an update reach the next step's pre phase through registers and LDS as well.
// Both compute the bits of the kernels above; the launcher picks by batch size (launch_step / launch_run).
//
// HBM layout:
//   P  [B][300]  symmetric-packed, one contiguous 2.4 KB record per instance: 12 "double rows" r
//                of 25 words, word c of double row r is P[r][c] for c <= r, else P[23-r][c-r-1]
//   xs [36][B]   structure-of-arrays state: words x26 | om_old3 | acc_old3 | q_old4
//                (R_WB_old = Rm(q_old), Filter.py:227)
//   J  [n][B][96]  records between the kernels: residual of the launch's update (8 words, first predict's record) |
//                  Jacobian blocks (67 words; 85 with N6); e [B][24], status [B]; aux [12 + 8 n][B] (epochs)
//                (status bit 30 = "this instance's error state still awaits its injection": device state,
//                 so that captured step sequences replay correctly -- see eskf_pre_kernel)
#include <hip/hip_runtime.h>

#include <type_traits>
#include <stdint.h>
#include <stdlib.h>
#include <stddef.h>
#include <stdio.h>

#include "eskf_device.h"

// Where a*b+c becomes one fused operation is decided by the SOURCE -- within an expression, by the front end -- and not by
// the back end's view of the code around it (hipcc's default, -ffp-contract=fast, lets the instruction selector fuse
// across statements, depending on how often a product is used in whatever kernel the statement was inlined into).  The
// same device function then yields the same bits in every kernel that uses it: the paths of DESIGN.md section 4 are
// compared bit for bit with each other (tests/test_gpu_step_kernel.py, test_gpu_epochs_kernel.py), and a multi-GPU shard
// must compute what the single GPU computes whichever kernel its batch size selects.  Functions that spell their
// operations out (`#pragma clang fp contract(off)` + fma) are not affected.
#pragma clang fp contract(on)

namespace eskf {

// ----------------------------------------------------------------------------------
// small math on register arrays
// ----------------------------------------------------------------------------------
__device__ __forceinline__ void t_sincos(double a, double& s, double& c) { sincos(a, &s, &c); }
__device__ __forceinline__ void t_sincos(float a, float& s, float& c) { sincosf(a, &s, &c); }
__device__ __forceinline__ double t_sqrt(double a) { return sqrt(a); }
__device__ __forceinline__ float t_sqrt(float a) { return sqrtf(a); }
__device__ __forceinline__ double t_asin(double a) { return asin(a); }
__device__ __forceinline__ float t_asin(float a) { return asinf(a); }
__device__ __forceinline__ double t_fma(double a, double b, double c) { return fma(a, b, c); }
__device__ __forceinline__ float t_fma(float a, float b, float c) { return fmaf(a, b, c); }
__device__ __forceinline__ bool t_finite(double a) { return isfinite(a); }
__device__ __forceinline__ bool t_finite(float a) { return isfinite(a); }

// (cross3, quat_nrm, quat_mul, quat_about_axis: no contraction of a*b+c into an fma -- the error injection is
//  evaluated by three different kernels (pre kernel on the fly, eskf_pre4_kernel's chains, eskf_post_kernel) and has
//  to give the same bits in all of them, whatever the surrounding code lets the compiler fuse)
template <typename T>
__device__ __forceinline__ void cross3(const T* a, const T* b, T* c) {
#pragma clang fp contract(off)
    c[0] = a[1] * b[2] - a[2] * b[1];
    c[1] = a[2] * b[0] - a[0] * b[2];
    c[2] = a[0] * b[1] - a[1] * b[0];
}
template <typename T>
__device__ __forceinline__ void mat3_vec(const T* A, const T* x, T* y) {
#pragma unroll
    for (int i = 0; i < 3; ++i) y[i] = A[3 * i] * x[0] + A[3 * i + 1] * x[1] + A[3 * i + 2] * x[2];
}
template <typename T>
__device__ __forceinline__ void mat3T_vec(const T* A, const T* x, T* y) {
#pragma unroll
    for (int i = 0; i < 3; ++i) y[i] = A[i] * x[0] + A[3 + i] * x[1] + A[6 + i] * x[2];
}
// C = A * [w]x   (columns of the result are A (e_k x ... ) -- written out)
template <typename T>
__device__ __forceinline__ void mat3_skew(const T* A, const T* w, T* C) {
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const T a0 = A[3 * i], a1 = A[3 * i + 1], a2 = A[3 * i + 2];
        C[3 * i + 0] = a1 * w[2] - a2 * w[1];
        C[3 * i + 1] = a2 * w[0] - a0 * w[2];
        C[3 * i + 2] = a0 * w[1] - a1 * w[0];
    }
}
template <typename T>
__device__ __forceinline__ T norm3(const T* v) {
#pragma clang fp contract(off)
    return t_sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
}
// Quaternion.rot (tools/Quaternion.py:101-107): rotation matrix of q/|q|, xyzw
template <typename T>
__device__ __forceinline__ void quat_to_mat(const T* q, T* M) {
    const T n = T(1) / t_sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    const T x = q[0] * n, y = q[1] * n, z = q[2] * n, w = q[3] * n;
    const T x2 = x * x, y2 = y * y, z2 = z * z, w2 = w * w;
    const T xy = x * y, zw = z * w, xz = x * z, yw = y * w, yz = y * z, xw = x * w;
    M[0] = x2 - y2 - z2 + w2; M[1] = T(2) * (xy - zw);   M[2] = T(2) * (xz + yw);
    M[3] = T(2) * (xy + zw);  M[4] = -x2 + y2 - z2 + w2; M[5] = T(2) * (yz - xw);
    M[6] = T(2) * (xz - yw);  M[7] = T(2) * (yz + xw);   M[8] = -x2 - y2 + z2 + w2;
}
// Quaternion.normalise (Quaternion.py:195-206): unit length, scalar part >= 0
template <typename T>
__device__ __forceinline__ void quat_nrm(T* q) {
#pragma clang fp contract(off)
    T n = T(1) / t_sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    if (q[3] < T(0)) n = -n;
#pragma unroll
    for (int i = 0; i < 4; ++i) q[i] *= n;
}
// Hamilton product xyzw (Quaternion.py:170-173)
template <typename T>
__device__ __forceinline__ void quat_mul(const T* a, const T* b, T* o) {
#pragma clang fp contract(off)
    T c[3];
    cross3(a, b, c);
    o[3] = a[3] * b[3] - (a[0] * b[0] + a[1] * b[1] + a[2] * b[2]);
#pragma unroll
    for (int i = 0; i < 3; ++i) o[i] = a[3] * b[i] + b[3] * a[i] + c[i];
}
// Quaternion(val=3x3, do_normalise=True) (Quaternion.py:45-46,69): Markley's method on M as
// given -- NO orthogonalisation (scipy 1.10.1, the reference's pin) -- then unit, w >= 0.
template <typename T>
__device__ __forceinline__ void mat_to_quat(const T* M, T* q) {
    const T tr = M[0] + M[4] + M[8];
    // first maximum of (M00, M11, M22, tr), like numpy.argmax
    int c = 0;
    T best = M[0];
    if (M[4] > best) { best = M[4]; c = 1; }
    if (M[8] > best) { best = M[8]; c = 2; }
    if (tr > best) { c = 3; }
    if (c == 3) {
        q[0] = M[7] - M[5]; q[1] = M[2] - M[6]; q[2] = M[3] - M[1]; q[3] = T(1) + tr;
    } else if (c == 0) {
        q[0] = T(1) - tr + T(2) * M[0]; q[1] = M[3] + M[1]; q[2] = M[6] + M[2]; q[3] = M[7] - M[5];
    } else if (c == 1) {
        q[1] = T(1) - tr + T(2) * M[4]; q[2] = M[7] + M[5]; q[0] = M[1] + M[3]; q[3] = M[2] - M[6];
    } else {
        q[2] = T(1) - tr + T(2) * M[8]; q[0] = M[2] + M[6]; q[1] = M[5] + M[7]; q[3] = M[3] - M[1];
    }
    quat_nrm(q);
}
// Quaternion.about_axis (Quaternion.py:209-224)
template <typename T>
__device__ __forceinline__ void quat_about_axis(T angle, const T* axis, T* q) {
#pragma clang fp contract(off)
    const T len = t_sqrt(axis[0] * axis[0] + axis[1] * axis[1] + axis[2] * axis[2]);
    T s, c;
    t_sincos(T(0.5) * angle, s, c);
    // guard constant is 4*eps of float64 in the reference, whatever the working precision
    const T f = (len > T(8.881784197001252e-16)) ? s / len : T(1);
    q[0] = axis[0] * f; q[1] = axis[1] * f; q[2] = axis[2] * f; q[3] = c;
    quat_nrm(q);
}

// Quaternion.angle (Quaternion.py:138-147) of the unit error quaternion with vector part of length nv: asin(nv), HALF the
// rotation angle (quirk Q6) -- the full angle when the quirk is switched off.  (v1: times one, the same bits.)
template <typename T>
__device__ __forceinline__ T residual_angle(T nv, int quirks) {
    const T k6 = (quirks & QK_Q6) ? T(1) : T(2);
    return k6 * t_asin(nv);
}
// ----------------------------------------------------------------------------------
// probe kinematics in closed form (models/Probe.py:147-167)
// ----------------------------------------------------------------------------------
template <typename T>
struct ProbeTrig {
    T s1, c1, s2, c2, s3, c3, sp, cp;   // sin/cos of q1, q2, q3, phi = q3 - q7
};
template <typename T>
__device__ __forceinline__ ProbeTrig<T> probe_trig(const T* dofs, T q7) {
    ProbeTrig<T> t;
    t_sincos(dofs[0], t.s1, t.c1);
    t_sincos(dofs[1], t.s2, t.c2);
    t_sincos(dofs[2], t.s3, t.c3);
    t_sincos(dofs[2] - q7, t.sp, t.cp);
    return t;
}
// The sines / cosines of q1, q2, q3 only change when the extrinsics do -- at an injection, or when frozen DOFs
// are zeroed -- not from one predict of an epoch to the next; phi = q3 - q7 moves with the notch angle.  A lane
// keeps the values together with the arguments they belong to and re-evaluates what has changed (same
// arguments, same library calls: bitwise the values of probe_trig).
template <typename T>
struct TrigCache {
    T a1, a2, a3, ph;   // the arguments the cached values belong to (NaN = nothing cached)
    ProbeTrig<T> t;
    __device__ __forceinline__ TrigCache() {
        a1 = a2 = a3 = ph = T(NAN);
        t.s1 = t.c1 = t.s2 = t.c2 = t.s3 = t.c3 = t.sp = t.cp = T(0);
    }
};
template <typename T>
__device__ __forceinline__ ProbeTrig<T> probe_trig(const T* dofs, T q7, TrigCache<T>& tc) {
    if (!(tc.a1 == dofs[0])) { tc.a1 = dofs[0]; t_sincos(dofs[0], tc.t.s1, tc.t.c1); }
    if (!(tc.a2 == dofs[1])) { tc.a2 = dofs[1]; t_sincos(dofs[1], tc.t.s2, tc.t.c2); }
    if (!(tc.a3 == dofs[2])) { tc.a3 = dofs[2]; t_sincos(dofs[2], tc.t.s3, tc.t.c3); }
    const T ph = dofs[2] - q7;
    if (!(tc.ph == ph)) { tc.ph = ph; t_sincos(ph, tc.t.sp, tc.t.cp); }
    return tc.t;
}
#define E5(s, c, o)   { o[0] = t.s1 * t.s2 * (s) - t.c1 * (c); o[1] = t.s1 * (c) + t.c1 * t.s2 * (s); o[2] = t.c2 * (s); }
#define E6(s, c, o)   { o[0] = t.s1 * t.s2 * (c) + t.c1 * (s); o[1] = t.c1 * t.s2 * (c) - t.s1 * (s); o[2] = t.c2 * (c); }
#define E5_1(s, c, o) { o[0] = t.c1 * t.s2 * (s) + t.s1 * (c); o[1] = t.c1 * (c) - t.s1 * t.s2 * (s); o[2] = T(0); }
#define E6_1(s, c, o) { o[0] = t.c1 * t.s2 * (c) - t.s1 * (s); o[1] = -t.s1 * t.s2 * (c) - t.c1 * (s); o[2] = T(0); }
#define E5_2(s, c, o) { o[0] = t.s1 * t.c2 * (s); o[1] = t.c1 * t.c2 * (s); o[2] = -t.s2 * (s); }
#define E6_2(s, c, o) { o[0] = t.s1 * t.c2 * (c); o[1] = t.c1 * t.c2 * (c); o[2] = -t.s2 * (c); }

// p_CB, R_BC, z6 -- the outputs of SymProbe.get_est_fwkin that are not identically zero
template <typename T>
__device__ __forceinline__ void probe_pose(const ProbeTrig<T>& t, const T* dofs, T L, T st, T ct, T* p, T* R, T* z6) {
    T e5[3], e6[3], a[3], g[3];
    z6[0] = -t.s1 * t.c2; z6[1] = -t.c1 * t.c2; z6[2] = t.s2;
    E5(t.s3, t.c3, e5); E6(t.s3, t.c3, e6);
    E5(t.sp, t.cp, a);  E6(t.sp, t.cp, g);
    const T lq = L - dofs[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        p[i] = lq * z6[i] + dofs[4] * e5[i] + dofs[5] * e6[i];
        R[3 * i + 0] = -a[i];
        R[3 * i + 1] = ct * g[i] + st * z6[i];
        R[3 * i + 2] = -st * g[i] + ct * z6[i];
    }
}

// ----------------------------------------------------------------------------------
// per-instance register state
// ----------------------------------------------------------------------------------
template <typename T>
struct State {
    T p[3], v[3], q[4], dofs[6], notch[3], pc[3], qc[4];   // x26 (state.py:11-29)
    T om_o[3], acc_o[3], qo[4];                            // Filter.py:78-80 buffers
};

// Blocks of the error-state transition Fx that are not structural constants
// (SURVEY.md App. A.3).  Row 18+r: dt at col 3+r | Bm[r][:] at 6:9 | Cm[r][:] at 9:15 | 1 at col
// 16+r.  Row 21+r: Dm[r][0:3] at 9:12 | Dm[r][3] at col 15 | Em[r][:] at 19:22 | 1 at (22,22),(23,23).
template <typename T>
struct Jac {
    T A[9];     // Fx[3:6,6:9]  = -dt R_o [a_o]x            (Filter.py:254)
    T Th[9];    // Fx[6:9,6:9]  = Rm(Om_old)^T              (Filter.py:255,132-134)
    T Bm[9], Cm[18], Dm[12], Em[9];
    T dt;
    T N6[18];   // Fi[18:24,3:6] (6x3): dt R_o [p]x ; -dt R_BC^T
};              // member order == word order of the Jacobian record in HBM (J_A .. J_N6 below)

// Filter._predict_nominal (Filter.py:232-247) + eqns.f_predict (equations.py:44-50,72-100)
// + State.from_array (state.py:62-74), then the Jacobian blocks of Filter._predict_error /
// _cam_error_jacobian (Filter.py:249-342) evaluated -- as the reference does -- at the
// post-step dofs/notch with the pre-step buffers om_old, acc_old, R_WB_old.
// NOMINAL = false: only what the Jacobians of the following steps depend on is advanced
// (attitude q, dofs, notch); JAC = false: the blocks are not formed.  The two halves are
// independent, so the pre kernel gives them to different waves.
// ATT = false (Jacobian role of a single-predict launch): the attitude is not advanced either -- the blocks of
// THIS predict only need the buffers (om_old, acc_old, q_old), dofs and notch; q would only matter as the
// next predict's q_old.
// CACHE: reuse the trigonometry of the previous predict of the launch where its arguments have not moved
template <typename T, bool NOMINAL = true, bool JAC = true, bool ATT = true, bool CACHE = true>
__device__ __forceinline__ void predict_scalar(State<T>& s, const DevParams<T>& prm, const T* sig_om, T dt,
                                               const T* om_n, const T* acc_n, Jac<T>& J, TrigCache<T>& tc) {
    static_assert(ATT || (!NOMINAL && JAC), "ATT = false is the Jacobian role only");
    T R[9], Ro[9];
    if (ATT) quat_to_mat(s.q, R);
    if (JAC) quat_to_mat(s.qo, Ro);

    // --- nominal state -------------------------------------------------------------
    if (ATT) {
        T om_avg[3], w[3], RS[9], Rn[9];
#pragma unroll
        for (int i = 0; i < 3; ++i) { om_avg[i] = (s.om_o[i] + om_n[i]) * T(0.5); w[i] = dt * om_avg[i]; }
        mat3_skew(R, w, RS);
#pragma unroll
        for (int i = 0; i < 9; ++i) Rn[i] = R[i] + RS[i];                           // equations.py:46
        if (NOMINAL) {
            T Rc[9];
            quat_to_mat(s.qc, Rc);
            const ProbeTrig<T> t = CACHE ? probe_trig(s.dofs, s.notch[0], tc) : probe_trig(s.dofs, s.notch[0]);
            T pB[3], RBC[9], z6[3];
            probe_pose(t, s.dofs, prm.L, prm.st, prm.ct, pB, RBC, z6);
            T a1[3], a2[3], acc_avg[3];
            mat3_vec(R, s.acc_o, a1);
            mat3_vec(Rn, acc_n, a2);
#pragma unroll
            for (int i = 0; i < 3; ++i) acc_avg[i] = (a1[i] + a2[i]) * T(0.5);      // equations.py:47
            T u[3], C_om[3], cr[3], Rcr[3], wc[3], RcS[9], Rcn[9];
#pragma unroll
            for (int i = 0; i < 3; ++i) u[i] = ((prm.quirks & QK_Q9) ? s.om_o[i] : om_avg[i]) + z6[i] * s.notch[1];   // (Q9: om_old, not the average)
            mat3T_vec(RBC, u, C_om);                                                // equations.py:50
            cross3(om_avg, pB, cr);                                                 // BB_v_CB == 0
            mat3_vec(R, cr, Rcr);
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const T vi = s.v[i];
                s.p[i] = s.p[i] + dt * vi + (dt * dt * T(0.5)) * acc_avg[i];        // equations.py:76
                s.v[i] = vi + dt * acc_avg[i];                                      // equations.py:77
                s.pc[i] = s.pc[i] + dt * vi + dt * Rcr[i];                          // equations.py:83-85
                wc[i] = dt * C_om[i];
            }
            mat3_skew(Rc, wc, RcS);
#pragma unroll
            for (int i = 0; i < 9; ++i) Rcn[i] = Rc[i] + RcS[i];                    // equations.py:86
            mat_to_quat(Rcn, s.qc);                                                 // state.py:72
        }
        s.notch[0] += dt * s.notch[1];                                              // equations.py:80-82
        s.notch[1] += dt * s.notch[2];
        if (!prm.hold_frozen) {                                                     // Filter.py:243-245
#pragma unroll
            for (int i = 0; i < 6; ++i)
                if ((prm.frozen_mask >> i) & 1) s.dofs[i] = T(0);
        }
        mat_to_quat(Rn, s.q);                                                       // state.py:67
    } else {
        s.notch[0] += dt * s.notch[1];                                              // equations.py:80-82
        s.notch[1] += dt * s.notch[2];
        if (!prm.hold_frozen) {                                                     // Filter.py:243-245
#pragma unroll
            for (int i = 0; i < 6; ++i)
                if ((prm.frozen_mask >> i) & 1) s.dofs[i] = T(0);
        }
    }
    if (!JAC) {
#pragma unroll
        for (int i = 0; i < 3; ++i) { s.om_o[i] = om_n[i]; s.acc_o[i] = acc_n[i]; }
#pragma unroll
        for (int i = 0; i < 4; ++i) s.qo[i] = s.q[i];
        return;
    }

    // --- Jacobian blocks -----------------------------------------------------------
    J.dt = dt;
    {
        T w[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) w[i] = -dt * s.acc_o[i];
        mat3_skew(Ro, w, J.A);                                                      // -dt R_o [a_o]x
        T Om[4] = {T(0.5) * dt * s.om_o[0], T(0.5) * dt * s.om_o[1], T(0.5) * dt * s.om_o[2], T(1)};
        T OmR[9];
        quat_to_mat(Om, OmR);                                                       // Filter.py:132-134
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) J.Th[3 * i + j] = OmR[3 * j + i];
    }
    {
        const ProbeTrig<T> t = CACHE ? probe_trig(s.dofs, s.notch[0], tc) : probe_trig(s.dofs, s.notch[0]);
        const T st = prm.st, ct = prm.ct;
        T z6[3], z6_1[3], z6_2[3], e5[3], e6[3], e5_1[3], e5_2[3], e6_1[3], e6_2[3];
        T a[3], g[3], a_1[3], a_2[3], g_1[3], g_2[3];
        z6[0] = -t.s1 * t.c2;   z6[1] = -t.c1 * t.c2;  z6[2] = t.s2;
        z6_1[0] = -t.c1 * t.c2; z6_1[1] = t.s1 * t.c2; z6_1[2] = T(0);
        z6_2[0] = t.s1 * t.s2;  z6_2[1] = t.c1 * t.s2; z6_2[2] = t.c2;
        E5(t.s3, t.c3, e5);     E6(t.s3, t.c3, e6);
        E5_1(t.s3, t.c3, e5_1); E6_1(t.s3, t.c3, e6_1);
        E5_2(t.s3, t.c3, e5_2); E6_2(t.s3, t.c3, e6_2);
        E5(t.sp, t.cp, a);      E6(t.sp, t.cp, g);
        E5_1(t.sp, t.cp, a_1);  E6_1(t.sp, t.cp, g_1);
        E5_2(t.sp, t.cp, a_2);  E6_2(t.sp, t.cp, g_2);

        const T lq = prm.L - s.dofs[3], q5 = s.dofs[4], q6 = s.dofs[5], thd = s.notch[1];
        T p[3], Jp[18];   // Jp[3][6]: d p_CB / d q1..q6 (the q7 column is identically zero)
        T RBC[9];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            p[i] = lq * z6[i] + q5 * e5[i] + q6 * e6[i];
            Jp[6 * i + 0] = lq * z6_1[i] + q5 * e5_1[i] + q6 * e6_1[i];
            Jp[6 * i + 1] = lq * z6_2[i] + q5 * e5_2[i] + q6 * e6_2[i];
            Jp[6 * i + 2] = q5 * e6[i] - q6 * e5[i];
            Jp[6 * i + 3] = -z6[i];
            Jp[6 * i + 4] = e5[i];
            Jp[6 * i + 5] = e6[i];
            RBC[3 * i + 0] = -a[i];
            RBC[3 * i + 1] = ct * g[i] + st * z6[i];
            RBC[3 * i + 2] = -st * g[i] + ct * z6[i];
        }
        // noise symbols are evaluated at their std-devs (Filter.py:330-333); "true" v_CB := p_CB
        // (Probe.py:458)
        T om_t[3], wv[3], cr[3], u_t[3], u_o[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) om_t[i] = s.om_o[i] - sig_om[i];
        cross3(om_t, p, cr);
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            wv[i] = -dt * (p[i] + cr[i]);
            u_t[i] = om_t[i] + z6[i] * thd;
            u_o[i] = s.om_o[i] + z6[i] * thd;
        }
        mat3_skew(Ro, wv, J.Bm);                                                    // -dt R_o [p + om_t x p]x
        // M = dt R_o (I + [om_t]x);  Cm = M Jp
        T RoS[9], M[9];
        mat3_skew(Ro, om_t, RoS);
#pragma unroll
        for (int i = 0; i < 9; ++i) M[i] = dt * (Ro[i] + RoS[i]);
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int c = 0; c < 6; ++c)
                J.Cm[6 * r + c] = M[3 * r] * Jp[c] + M[3 * r + 1] * Jp[6 + c] + M[3 * r + 2] * Jp[12 + c];
        // delta-theta_C rows: d/dq_i [ R_BC^T (om_t + z6 thd) ] with only R_BC and z6 depending on q
        // columns of dR_BC/dq_i are (-da, ct dg + st dz, -st dg + ct dz)
        auto dcol = [&](const T* da, const T* dg, const T* dz, const T* dzr, int col) {
            T d0 = T(0), d1 = T(0), d2 = T(0);
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                d0 += -da[k] * u_t[k];
                d1 += (ct * dg[k] + st * dz[k]) * u_t[k];
                d2 += (-st * dg[k] + ct * dz[k]) * u_t[k];
            }
            T rz[3];
            mat3T_vec(RBC, dzr, rz);
            J.Dm[0 * 4 + col] = dt * (d0 + rz[0] * thd);
            J.Dm[1 * 4 + col] = dt * (d1 + rz[1] * thd);
            J.Dm[2 * 4 + col] = dt * (d2 + rz[2] * thd);
        };
        const T zero3[3] = {T(0), T(0), T(0)};
        T ma[3] = {-a[0], -a[1], -a[2]}, mg[3] = {-g[0], -g[1], -g[2]};
        dcol(a_1, g_1, z6_1, z6_1, 0);      // q1
        dcol(a_2, g_2, z6_2, z6_2, 1);      // q2
        dcol(g, ma, zero3, zero3, 2);       // q3:  da/dphi = g, dg/dphi = -a
        dcol(mg, a, zero3, zero3, 3);       // q7:  phi = q3 - q7
        T ac[3], bc[3], hs[3];
        mat3T_vec(RBC, u_t, ac);
        mat3T_vec(RBC, u_o, bc);
#pragma unroll
        for (int i = 0; i < 3; ++i) hs[i] = T(0.5) * dt * (ac[i] + bc[i]);
        // Em = I - [hs]x
        J.Em[0] = T(1);   J.Em[1] = hs[2];  J.Em[2] = -hs[1];
        J.Em[3] = -hs[2]; J.Em[4] = T(1);   J.Em[5] = hs[0];
        J.Em[6] = hs[1];  J.Em[7] = -hs[0]; J.Em[8] = T(1);
        // noise Jacobian camera rows (only needed when the gyro block of Q is non-zero)
        if (prm.jfull) {    // (without it the record ends after dt: the caller's staging rows are 67 words)
            T dp[3] = {dt * p[0], dt * p[1], dt * p[2]};
            mat3_skew(Ro, dp, J.N6);
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int j = 0; j < 3; ++j) J.N6[9 + 3 * i + j] = -dt * RBC[3 * j + i];
        }
    }
    if (!ATT) return;
    // buffer swap (Filter.py:225-227)
#pragma unroll
    for (int i = 0; i < 3; ++i) { s.om_o[i] = om_n[i]; s.acc_o[i] = acc_n[i]; }
#pragma unroll
    for (int i = 0; i < 4; ++i) s.qo[i] = s.q[i];
}

// ----------------------------------------------------------------------------------
// Work split.  The per-instance "scalar" arithmetic -- probe trigonometry,
// nominal propagation, Jacobian blocks, residual, injection -- does not depend on the
// covariance, and CDNA4 has no scalar fp64 ALU: evaluated inside a wave that is shared by a
// few instances it costs a full wave issue per instruction.  It therefore runs in two small
// lane-per-instance kernels (64 instances per wave, every lane useful):
//
//   eskf_pre_kernel   x, IMU samples, measurement -> predicted x, Jacobian blocks J[n], residual
//   eskf_cov_*_kernel P, J[n], residual           -> P, error state e, status      (32 lanes / instance)
//   eskf_post_kernel  predicted x, e, status      -> x                              (injection)
//
// The nominal state never reads P, so one pre-launch covers all n_pred predicts of an epoch
// and the covariance kernel runs them back to back with P resident in LDS.
// ----------------------------------------------------------------------------------
template <typename T> struct Vec2;
template <> struct Vec2<double> { using type = double2; };
template <> struct Vec2<float> { using type = float2; };
template <typename T> struct Vec16;     // a 16-byte piece of a record
template <> struct Vec16<double> { using type = double2; };
template <> struct Vec16<float> { using type = float4; };

// One record per instance and predict: the residual of the launch's update (7 words + the non-finite flag; record
// of the FIRST predict only) followed by the blocks in the member order of Jac<T>:
//   res8 | A9 Th9 Bm9 Cm18 Dm12 Em9 dt1 | N6 18 | pad      (96 words = 768 B: six 128-byte lines, a group of 64
//   instances starts on a line of its own)
constexpr int JW = 96;
constexpr int J_RES = 0, J_B0 = 8;      // J_B0: first word of the blocks
constexpr int J_A = J_B0 + 0, J_TH = J_B0 + 9, J_BM = J_B0 + 18, J_CM = J_B0 + 27, J_DM = J_B0 + 45, J_EM = J_B0 + 57,
              J_DT = J_B0 + 66, J_N6 = J_B0 + 67;
constexpr int J_CORE = J_B0 + 67, J_FULL = J_B0 + 85;   // record words in use without / with the N6 block
constexpr int JB_CORE = 67, JB_FULL = 85;               // ... counted from J_B0 (the staging rows of eskf_pre_kernel)
static_assert(JW == JAC_WORDS, "eskf_device.h sizes the workspace");

// state record, structure-of-arrays: word w of instance b at xs[w * B + b]
// words: x26 | om_old3 | acc_old3 | q_old4
template <typename T>
__device__ __forceinline__ void load_state(const T* xs, int64_t B, int64_t b, State<T>& s) {
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        s.p[i] = xs[(0 + i) * B + b]; s.v[i] = xs[(3 + i) * B + b]; s.notch[i] = xs[(16 + i) * B + b];
        s.pc[i] = xs[(19 + i) * B + b]; s.om_o[i] = xs[(26 + i) * B + b]; s.acc_o[i] = xs[(29 + i) * B + b];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) { s.q[i] = xs[(6 + i) * B + b]; s.qc[i] = xs[(22 + i) * B + b]; s.qo[i] = xs[(32 + i) * B + b]; }
#pragma unroll
    for (int i = 0; i < 6; ++i) s.dofs[i] = xs[(10 + i) * B + b];
}
// what the Jacobian blocks of ONE predict depend on: dofs, notch, om_old, acc_old, q_old (19 words); the
// rest of the register state is left unset and must not be used
template <typename T>
__device__ __forceinline__ void load_state_jac(const T* xs, int64_t B, int64_t b, State<T>& s) {
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        s.notch[i] = xs[(16 + i) * B + b]; s.om_o[i] = xs[(26 + i) * B + b]; s.acc_o[i] = xs[(29 + i) * B + b];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) s.qo[i] = xs[(32 + i) * B + b];
#pragma unroll
    for (int i = 0; i < 6; ++i) s.dofs[i] = xs[(10 + i) * B + b];
}
template <typename T>
__device__ __forceinline__ void store_state(T* xs, int64_t B, int64_t b, const State<T>& s) {
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        xs[(0 + i) * B + b] = s.p[i]; xs[(3 + i) * B + b] = s.v[i]; xs[(16 + i) * B + b] = s.notch[i];
        xs[(19 + i) * B + b] = s.pc[i]; xs[(26 + i) * B + b] = s.om_o[i]; xs[(29 + i) * B + b] = s.acc_o[i];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) { xs[(6 + i) * B + b] = s.q[i]; xs[(22 + i) * B + b] = s.qc[i]; xs[(32 + i) * B + b] = s.qo[i]; }
#pragma unroll
    for (int i = 0; i < 6; ++i) xs[(10 + i) * B + b] = s.dofs[i];
}

// IMU sample `ip` of instance b; zeros beyond the last one (assigned on every path, so that the
// registers of the sample before are free as soon as it has been used)
template <typename T, bool SHARED = false>
__device__ __forceinline__ void load_imu_sample(const StepArgs& a, int ip, int64_t b, T& dt, T (&om)[3], T (&acc)[3]) {
    if (ip < a.n_pred) {
        if constexpr (SHARED) {
            // one IMU stream for the whole batch: the sample is the same for every lane -- fetched with SCALAR
            // loads (constant address space).  Besides saving vector registers this keeps it off the vector-memory
            // counter: a vector load issued behind an epoch's record stores is only known to have arrived when
            // all of those stores have been acknowledged (loads and stores retire in order on one counter), which
            // exposed the full store latency once per predict.
            typedef const T __attribute__((address_space(4))) * cptr;
            cptr cdt = (cptr)(uintptr_t)a.dt;
            cptr com = (cptr)(uintptr_t)a.om;
            cptr cacc = (cptr)(uintptr_t)a.acc;
            dt = cdt[ip];
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                om[i] = com[ip * 3 + i];
                acc[i] = cacc[ip * 3 + i];
            }
            return;
        }
        const int64_t so = a.imu_stride ? ((int64_t)ip * a.B + b) * 3 : (int64_t)ip * 3;
        dt = reinterpret_cast<const T*>(a.dt)[a.dt_stride ? (int64_t)ip * a.B + b : ip];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            om[i] = reinterpret_cast<const T*>(a.om)[so + i];
            acc[i] = reinterpret_cast<const T*>(a.acc)[so + i];
        }
    } else {
        dt = T(0);
#pragma unroll
        for (int i = 0; i < 3; ++i) { om[i] = T(0); acc[i] = T(0); }
    }
}

// The 64 Jacobian records a wavefront has staged in LDS (row stride JS words, one row per instance) -> HBM, in
// 16-byte pieces, as many whole records per store instruction as fit the wavefront (fp64: one record = 34 pieces;
// fp32: three records of 17 pieces).  Eight instructions at a time: all their LDS reads are issued, THEN the stores
// -- the compiler fence in between keeps it from pairing every read with its store and a wait (one LDS round trip
// per record: 64 x ~120 cycles).  Measured per 33-predict epoch of B = 8192, fp32 (pre kernel): one record per
// instruction, read-wait-store 144 us; this version 118 us; no stores at all 72 us; the same bytes as 67
// one-word-per-lane instructions (conflict-free LDS reads) 213 us; as 17 instructions with all 64 lanes busy
// (pieces of up to four records per instruction) 183 us.
template <typename P>
__device__ __forceinline__ void st_piece(P* dst, const P& v) {
    *dst = v;
}
template <typename T, int JS>
__device__ __forceinline__ void store_records(const T* sJ, T* gJ, int nrec, int lane, int first = 0) {
    constexpr int PW = 16 / (int)sizeof(T);             // words per piece
    struct alignas(16) Piece { T w[PW]; };
    constexpr int NP = (JS + PW - 1) / PW;              // pieces per record
    constexpr int RPI = 64 / NP;                        // records per store instruction
    const int rs = lane / NP, pc = lane - rs * NP;
    if (rs >= RPI) return;
    const T* src = sJ + pc * PW;
    const int nw = (JS - pc * PW) > PW ? PW : (JS - pc * PW);   // valid words of this lane's piece
    auto fetch = [&](int r, Piece& v) {
#pragma unroll
        for (int k = 0; k < PW; ++k) v.w[k] = (k < nw) ? src[r * JS + k] : T(0);
    };
    T* dst = gJ + pc * PW;
    int r0 = first;     // (records [first, nrec): several wavefronts may share the rows of a block)
    for (; r0 + 8 * RPI <= nrec; r0 += 8 * RPI) {
        Piece v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) fetch(r0 + u * RPI + rs, v[u]);
        asm volatile("" ::: "memory");
#pragma unroll
        for (int u = 0; u < 8; ++u) st_piece(reinterpret_cast<Piece*>(dst + (int64_t)(r0 + u * RPI + rs) * JW), v[u]);
    }
    for (; r0 < nrec; r0 += RPI) {                      // the rest (and the last block of a ragged batch)
        const int r = r0 + rs;
        if (r < nrec) {
            Piece v;
            fetch(r, v);
            st_piece(reinterpret_cast<Piece*>(dst + (int64_t)r * JW), v);
        }
    }
}

// ---- lane-per-instance: nominal propagation + residual (blockIdx.y = 0) and Jacobian blocks
// (blockIdx.y = 1).  The two halves are independent chains of about equal length; giving them to
// different waves halves the latency of this kernel (it runs at one or two waves per SIMD).
// ONE: the launch has exactly one predict (the pair step, a single predict) -- its own build, so that the slim
// Jacobian role below does not share registers and code with the epoch loop
// SIMU (epoch builds): dt, om, acc are one stream shared by the batch (dt_stride = imu_stride = 0): the samples after
// the first are fetched with scalar loads, see load_imu_sample
template <typename T, bool ONE, bool SIMU = false>
__global__ __launch_bounds__(128, 2) void eskf_pre_kernel(StepArgs a, DevParams<T> prm) {
    // Two wavefronts per block work on the same 64 instances: wave 0 propagates the nominal state
    // (and forms the residual), wave 1 forms the Jacobian blocks.  Both read the state before wave 0
    // overwrites it (block barrier).  Wave 1 stages the 67 block words of its records in LDS (odd row stride:
    // conflict-free) and writes them out as whole records with 16-byte stores -- a lane-per-instance
    // store of an array-of-structures would touch 64 different cache lines per instruction.
    static_assert(sizeof(Jac<T>) == JB_FULL * sizeof(T), "Jac<T> must be the 85 block words of the record");
    // staging rows of 85 words, or 67 when the gyro block of Q is zero and the N6 block is not formed
    // (both odd -> conflict-free); sized by the launcher: 34 KB lets four blocks share a CU
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    T* sJ = reinterpret_cast<T*>(smem_raw);
    const int JS = prm.jfull ? JB_FULL : JB_CORE;
    const int lane = threadIdx.x & 63;
    const bool jac_role = threadIdx.x >= 64;           // wave-uniform
    const int64_t b0 = (int64_t)blockIdx.x * 64;
    const int64_t b_raw = b0 + lane;
    const bool live = b_raw < a.B;
    const int64_t b = live ? b_raw : a.B - 1;      // tail lanes recompute the last instance, write nothing
    T* xs = reinterpret_cast<T*>(a.xs);
    // The Jacobian role of a single-predict launch (the pair step) needs 19 of the 36 state words --
    // dofs, notch, om_old, acc_old, q_old -- and 9 of the 24 words of a pending error state (ddofs,
    // dnotch): its blocks are evaluated with the stale buffers (Filter.py:254-255,327: R_WB_old is not
    // refreshed by an update) and the attitude q only matters as the NEXT predict's q_old.
    const bool jac_slim = ONE && jac_role;                 // wave-uniform
    State<T> s;
    if (jac_slim) load_state_jac(reinterpret_cast<const T*>(a.xs_src), a.B, b, s);
    else load_state(reinterpret_cast<const T*>(a.xs_src), a.B, b, s);
    // Deferred injection.  Whether the workspace holds an update whose error state is still to be
    // injected is DEVICE state -- bit DVI_STATUS_PENDING of the instance's status word, set by the
    // covariance kernel, cleared here (or by the flush kernel) -- so that a captured sequence of steps
    // replays correctly whatever the host knew when it was captured.  The launcher only passes a hint:
    // a.pending_hint == 0 promises that no instance has the bit set (predict-only sequences then do not
    // fetch 24 words per instance for nothing).  The error state is fetched together with the status
    // word and the state -- one round trip to HBM, not two -- whether or not the status lets it be applied.
    T epend[24];
    int pend_status = 0;
    if (a.pending_hint) {
        pend_status = a.status_ws[b];
        if (jac_slim) {
#pragma unroll
            for (int i = 9; i < 18; ++i) epend[i] = reinterpret_cast<const T*>(a.e)[b * 24 + i];
        } else {
#pragma unroll
            for (int i = 0; i < 24; ++i) epend[i] = reinterpret_cast<const T*>(a.e)[b * 24 + i];
        }
    }
    // (dvi_eskf_step_from reads another handle's flushed state: whatever was due on this one is dropped)
    const bool was_pending = (pend_status & DVI_STATUS_PENDING) != 0;
    const bool inject_now = was_pending && !a.drop_pending && !(pend_status & DVI_STATUS_SINGULAR);
    // ... and so is the first IMU sample (the following ones are fetched at the end of the predict before)
    T dt0, om0[3], acc0[3];
    load_imu_sample(a, 0, b, dt0, om0, acc0);
    // wave 0 also fetches the measurement now and parks it in LDS until the residual is formed
    // (same round trip; kept in registers it would cost eight values through the whole propagation)
    T* sM = sJ + 64 * JS + lane * 9;
    if (!jac_role && a.do_update) {
        const int64_t mo = a.meas_stride ? b : 0;
        const T* gpos = reinterpret_cast<const T*>(a.pos) + mo * 3;
        const T* gq = reinterpret_cast<const T*>(a.quat) + mo * 4;
        T m[8] = {gpos[0], gpos[1], gpos[2], gq[0], gq[1], gq[2], gq[3], reinterpret_cast<const T*>(a.notch)[mo]};
#pragma unroll
        for (int i = 0; i < 8; ++i) sM[i] = m[i];
    }
    __syncthreads();                               // every read of the old state (and of the pending
                                                   // error state) precedes this step's writes
    if (inject_now) {                              // the previous update's injection (both roles)
        if (jac_slim) {
#pragma unroll
            for (int i = 0; i < 6; ++i) s.dofs[i] += epend[9 + i];
#pragma unroll
            for (int i = 0; i < 3; ++i) s.notch[i] += epend[15 + i];
        } else {
            inject_error(s, epend, prm.quirks);
        }
    }
    if (jac_role) {
        // ---- wave 1: Jacobian blocks of every predict of this launch ---------------------------
        // Jacobian evaluation point n = sigma (Filter.py:330-333): per instance when the handle carries
        // per-instance noise
        T sig_om[3] = {prm.sig_om[0], prm.sig_om[1], prm.sig_om[2]};
        if (a.noise) {
#pragma unroll
            for (int i = 0; i < 3; ++i) sig_om[i] = reinterpret_cast<const T*>(a.noise)[b * NOISE_WORDS + 20 + i];
        }
        T dt = dt0, om_n[3] = {om0[0], om0[1], om0[2]}, acc_n[3] = {acc0[0], acc0[1], acc0[2]};
        TrigCache<T> tc;
        for (int ip = 0; ip < a.n_pred; ++ip) {
            // the blocks are written straight into this lane's staging row (the struct IS the record),
            // so they do not occupy registers; only this wavefront touches sJ: wave-level ordering
            Jac<T>& J = *reinterpret_cast<Jac<T>*>(sJ + lane * JS);
            if constexpr (ONE) predict_scalar<T, false, true, false, false>(s, prm, sig_om, dt, om_n, acc_n, J, tc);
            else predict_scalar<T, false, true>(s, prm, sig_om, dt, om_n, acc_n, J, tc);
            // (only this wavefront touches the staging rows, and the LDS executes one wavefront's operations in
            //  order: what is needed is that the COMPILER keeps the staging writes, the reads of the store loop
            //  and the next predict's writes in program order -- not a wait for outstanding vector-memory
            //  operations, which a release fence would also imply)
            asm volatile("" ::: "memory");
            __builtin_amdgcn_wave_barrier();
            asm volatile("" ::: "memory");
            T* gJ = reinterpret_cast<T*>(a.J) + ((int64_t)ip * a.B + b0) * JW + J_B0;     // (the blocks follow the residual words)
            const int nrec = (a.B - b0 < 64) ? (int)(a.B - b0) : 64;
            if (prm.jfull) store_records<T, JB_FULL>(sJ, gJ, nrec, lane);
            else store_records<T, JB_CORE>(sJ, gJ, nrec, lane);
            // (only this wavefront touches the staging rows, and the LDS executes one wavefront's operations in
            //  order: what is needed is that the COMPILER keeps the staging writes, the reads of the store loop
            //  and the next predict's writes in program order -- not a wait for outstanding vector-memory
            //  operations, which a release fence would also imply)
            asm volatile("" ::: "memory");
            __builtin_amdgcn_wave_barrier();
            asm volatile("" ::: "memory");
            load_imu_sample<T, SIMU>(a, ip + 1, b, dt, om_n, acc_n);
        }
        return;
    }
    // ---- wave 0: nominal state through every predict, then the residual -------------------------
    T dt = dt0, om_n[3] = {om0[0], om0[1], om0[2]}, acc_n[3] = {acc0[0], acc0[1], acc0[2]};
    TrigCache<T> tc;
    for (int ip = 0; ip < a.n_pred; ++ip) {
        Jac<T> J;
        predict_scalar<T, true, false, true, !ONE>(s, prm, prm.sig_om, dt, om_n, acc_n, J, tc);
        load_imu_sample<T, SIMU>(a, ip + 1, b, dt, om_n, acc_n);
    }
    if (!live) return;
    // (output addresses are formed here, from an opaque copy of the instance index: formed in the
    // prologue they would sit in registers -- in scratch, in fact -- through the whole propagation)
    int64_t bo = b;
    asm volatile("" : "+v"(bo));
    if (a.do_update) {
        // measurement correction and residual (Filter.py:364-375)
        const T gpos[3] = {sM[0], sM[1], sM[2]};
        const T ang_notch = sM[7];
        T qraw[4] = {sM[3], sM[4], sM[5], sM[6]};
        T nq[4] = {T(0), T(0), T(0), T(1)};
        t_sincos(T(0.5) * ang_notch, nq[2], nq[3]);          // Quaternion(val=[0,0,ang], euler="xyz")
        T qm[4], qe[4];
        quat_mul(nq, qraw, qm);
        quat_nrm(qm);
        qm[0] = -qm[0]; qm[1] = -qm[1]; qm[2] = -qm[2];      // conjugate
        quat_mul(qm, s.qc, qe);
        quat_nrm(qe);
        T nv = t_sqrt(qe[0] * qe[0] + qe[1] * qe[1] + qe[2] * qe[2]);
        T flag = T(0);
        if (nv > T(1)) { nv = T(1); flag = T(1); }
        const T ang = residual_angle(nv, prm.quirks);             // half-angle residual (Quaternion.py:138-147) unless Q6 is off
        const T f = (nv == T(0)) ? T(0) : ang / nv;
        T* gr = reinterpret_cast<T*>(a.J) + bo * JW + J_RES;      // (record of the launch's first predict)
        T r[7];
#pragma unroll
        for (int i = 0; i < 3; ++i) { r[i] = gpos[i] - s.pc[i]; r[3 + i] = f * qe[i]; }
        r[6] = ang_notch - s.notch[0];
        // a NaN / inf measurement or state propagates exactly as it does in the reference (NumPy
        // arithmetic, no check); the instance is flagged, its neighbours are not affected
        if (!t_finite(r[0] + r[1] + r[2] + r[3] + r[4] + r[5] + r[6])) flag = T(1);
#pragma unroll
        for (int i = 0; i < 7; ++i) gr[i] = r[i];
        gr[7] = flag;                                         // asin argument was clamped, or non-finite residual
    }
    if (a.n_pred > 0 || a.pending_hint || a.xs_src != a.xs) store_state(xs, a.B, bo, s);   // (unchanged in place otherwise)
    if (was_pending) a.status_ws[bo] = pend_status & ~DVI_STATUS_PENDING;                  // the injection is done (or dropped)
}

// ---- the Jacobian blocks of one predict (Filter.py:249-342), in three groups that only share
// their inputs -- evaluated, as the reference does, at the post-predict dofs / notch with the pre-predict buffers
// om_old, acc_old, R_WB_old (Filter.py:220-221,325-327).  eskf_pre4_kernel gives the groups to different wavefronts,
// eskf_jac_kernel to the two halves of a block.
// (a) what only needs the buffers: A = -dt R_o [a_o]x, Theta = Rm(Om_old)^T, dt; returns R_o
template <typename T>
__device__ __forceinline__ void jac_blocks_old_buffers(const T* qo, const T* acc_o, const T* om_o, T dt, Jac<T>& J, T* Ro) {
    quat_to_mat(qo, Ro);
    J.dt = dt;
    T w[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) w[i] = -dt * acc_o[i];
    mat3_skew(Ro, w, J.A);                                                  // -dt R_o [a_o]x
    T Om[4] = {T(0.5) * dt * om_o[0], T(0.5) * dt * om_o[1], T(0.5) * dt * om_o[2], T(1)};
    T OmR[9];
    quat_to_mat(Om, OmR);                                                   // Filter.py:132-134
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) J.Th[3 * i + j] = OmR[3 * j + i];
}
// (b) the delta-p_C rows: Bm, Cm (and the first half of N6) from p_CB and its derivatives
template <typename T>
__device__ __forceinline__ void jac_blocks_position(const DevParams<T>& prm, const ProbeTrig<T>& t, const T* dofs, const T* om_t,
                                                    const T* Ro, T dt, Jac<T>& J) {
    T z6[3], z6_1[3], z6_2[3], e5[3], e6[3], e5_1[3], e5_2[3], e6_1[3], e6_2[3];
    z6[0] = -t.s1 * t.c2;   z6[1] = -t.c1 * t.c2;  z6[2] = t.s2;
    z6_1[0] = -t.c1 * t.c2; z6_1[1] = t.s1 * t.c2; z6_1[2] = T(0);
    z6_2[0] = t.s1 * t.s2;  z6_2[1] = t.c1 * t.s2; z6_2[2] = t.c2;
    E5(t.s3, t.c3, e5);     E6(t.s3, t.c3, e6);
    E5_1(t.s3, t.c3, e5_1); E6_1(t.s3, t.c3, e6_1);
    E5_2(t.s3, t.c3, e5_2); E6_2(t.s3, t.c3, e6_2);
    const T lq = prm.L - dofs[3], q5 = dofs[4], q6 = dofs[5];
    T p[3], Jp[18];   // Jp[3][6]: d p_CB / d q1..q6 (the q7 column is identically zero)
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        p[i] = lq * z6[i] + q5 * e5[i] + q6 * e6[i];
        Jp[6 * i + 0] = lq * z6_1[i] + q5 * e5_1[i] + q6 * e6_1[i];
        Jp[6 * i + 1] = lq * z6_2[i] + q5 * e5_2[i] + q6 * e6_2[i];
        Jp[6 * i + 2] = q5 * e6[i] - q6 * e5[i];
        Jp[6 * i + 3] = -z6[i];
        Jp[6 * i + 4] = e5[i];
        Jp[6 * i + 5] = e6[i];
    }
    // noise symbols are evaluated at their std-devs (Filter.py:330-333); "true" v_CB := p_CB (Probe.py:458)
    T wv[3], cr[3];
    cross3(om_t, p, cr);
#pragma unroll
    for (int i = 0; i < 3; ++i) wv[i] = -dt * (p[i] + cr[i]);
    mat3_skew(Ro, wv, J.Bm);                                                // -dt R_o [p + om_t x p]x
    T RoS[9], M[9];                                                         // M = dt R_o (I + [om_t]x);  Cm = M Jp
    mat3_skew(Ro, om_t, RoS);
#pragma unroll
    for (int i = 0; i < 9; ++i) M[i] = dt * (Ro[i] + RoS[i]);
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 6; ++c)
            J.Cm[6 * r + c] = M[3 * r] * Jp[c] + M[3 * r + 1] * Jp[6 + c] + M[3 * r + 2] * Jp[12 + c];
    if (prm.jfull) {    // noise Jacobian, camera position rows: dt R_o [p]x
        T dp[3] = {dt * p[0], dt * p[1], dt * p[2]};
        mat3_skew(Ro, dp, J.N6);
    }
}
// (c) the delta-theta_C rows: Dm, Em (and the second half of N6) from R_BC and its derivatives
template <typename T>
__device__ __forceinline__ void jac_blocks_attitude(const DevParams<T>& prm, const ProbeTrig<T>& t, T thd, const T* om_t, const T* om_o,
                                                    T dt, Jac<T>& J) {
    const T st = prm.st, ct = prm.ct;
    T z6[3], z6_1[3], z6_2[3], aa[3], g[3], a_1[3], a_2[3], g_1[3], g_2[3];
    z6[0] = -t.s1 * t.c2;   z6[1] = -t.c1 * t.c2;  z6[2] = t.s2;
    z6_1[0] = -t.c1 * t.c2; z6_1[1] = t.s1 * t.c2; z6_1[2] = T(0);
    z6_2[0] = t.s1 * t.s2;  z6_2[1] = t.c1 * t.s2; z6_2[2] = t.c2;
    E5(t.sp, t.cp, aa);     E6(t.sp, t.cp, g);
    E5_1(t.sp, t.cp, a_1);  E6_1(t.sp, t.cp, g_1);
    E5_2(t.sp, t.cp, a_2);  E6_2(t.sp, t.cp, g_2);
    T RBC[9], u_t[3], u_o[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        RBC[3 * i + 0] = -aa[i];
        RBC[3 * i + 1] = ct * g[i] + st * z6[i];
        RBC[3 * i + 2] = -st * g[i] + ct * z6[i];
        u_t[i] = om_t[i] + z6[i] * thd;
        u_o[i] = om_o[i] + z6[i] * thd;
    }
    // delta-theta_C rows: d/dq_i [ R_BC^T (om_t + z6 thd) ] with only R_BC and z6 depending on q
    // columns of dR_BC/dq_i are (-da, ct dg + st dz, -st dg + ct dz)
    auto dcol = [&](const T* da, const T* dg, const T* dz, const T* dzr, int col) {
        T d0 = T(0), d1 = T(0), d2 = T(0);
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            d0 += -da[k] * u_t[k];
            d1 += (ct * dg[k] + st * dz[k]) * u_t[k];
            d2 += (-st * dg[k] + ct * dz[k]) * u_t[k];
        }
        T rz[3];
        mat3T_vec(RBC, dzr, rz);
        J.Dm[0 * 4 + col] = dt * (d0 + rz[0] * thd);
        J.Dm[1 * 4 + col] = dt * (d1 + rz[1] * thd);
        J.Dm[2 * 4 + col] = dt * (d2 + rz[2] * thd);
    };
    const T zero3[3] = {T(0), T(0), T(0)};
    T ma[3] = {-aa[0], -aa[1], -aa[2]}, mg[3] = {-g[0], -g[1], -g[2]};
    dcol(a_1, g_1, z6_1, z6_1, 0);      // q1
    dcol(a_2, g_2, z6_2, z6_2, 1);      // q2
    dcol(g, ma, zero3, zero3, 2);       // q3:  da/dphi = g, dg/dphi = -a
    dcol(mg, aa, zero3, zero3, 3);      // q7:  phi = q3 - q7
    T ac[3], bc[3], hs[3];
    mat3T_vec(RBC, u_t, ac);
    mat3T_vec(RBC, u_o, bc);
#pragma unroll
    for (int i = 0; i < 3; ++i) hs[i] = T(0.5) * dt * (ac[i] + bc[i]);
    // Em = I - [hs]x
    J.Em[0] = T(1);   J.Em[1] = hs[2];  J.Em[2] = -hs[1];
    J.Em[3] = -hs[2]; J.Em[4] = T(1);   J.Em[5] = hs[0];
    J.Em[6] = hs[1];  J.Em[7] = -hs[0]; J.Em[8] = T(1);
    if (prm.jfull) {    // noise Jacobian, camera attitude rows: -dt R_BC^T
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) J.N6[9 + 3 * i + j] = -dt * RBC[3 * j + i];
    }
}

// ----------------------------------------------------------------------------------
// eskf_pre4_kernel: the pre kernel of a launch with exactly ONE predict (the pair step, a single predict),
// re-cut into four short chains.  Four wavefronts share 64 instances (lane = instance); what one chain needs from
// another crosses LDS at two block barriers:
//
//   phase A (before barrier 1)                                        phase B (after it)
//   W0  p, v, q: injection, R, R+, acc_avg, p+, v+, q+ (Markley)      p_CB from the trigonometry -> pc+, r_p
//   W1  qc: injection, Rm(qc); measurement quaternion turned by the   R_BC, z6 -> C_om -> qc+ (Markley) -> r_theta,
//       notch angle; notch states                                     r_notch
//   W2  dofs: injection; sincos(q1), sincos(q2); R_o, blocks A,       p_CB, J_p at dofs+ -> blocks Bm, Cm
//       Theta, dt
//   W3  sincos(q3), sincos(phi), sincos(phi+)                         R_BC+ and its derivatives -> blocks Dm, Em
//
// (barrier 1b, right behind 1: every wavefront has copied the trigonometry out of the staging rows, whose block words
// carry it between the two) then barrier 2 and all four wavefronts move the 64 staged records -- the residual of the
// update in front, then the blocks -- to HBM.  The eight sincos evaluations of a
// pair step (two injections, four joint angles, phi after the predict, the notch measurement) -- a third of the
// old kernel's dependent chain -- are spread two per wavefront, and every chain fits 128 VGPRs (four blocks per CU,
// sixteen wavefronts; the two-role kernel above needs 217 and runs at two per SIMD).  Same helper functions, same
// operation order per output as predict_scalar: the results are the same bits.
// The trigonometry of the Jacobian blocks is evaluated at the post-predict dofs / notch (Filter.py:220-221,325-326):
// q1..q3 only differ from the nominal ones where a frozen DOF is zeroed (then sin = 0, cos = 1 exactly),
// phi+ = q3+ - (theta + dt theta') gets its own evaluation unless it equals phi.
// ----------------------------------------------------------------------------------
#if defined(ESKF_X_STAMPS) && !defined(ESKF_TU_FUSED)   // diagnostic build only: wall-clock stamps (100 MHz) of every block of the last pre4 launch
__device__ unsigned long long g_x_stamps[8 * 4096];
__device__ unsigned long long g_c_stamps[8 * 4096];      // covariance kernel: every eighth block
hipError_t x_read_cstamps(unsigned long long* out) { return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_c_stamps), sizeof(g_c_stamps)); }
#define ESKF_STAMP(K) { if (lane == 0 && vblock < 4096) g_x_stamps[(K) * 4096 + vblock] = __builtin_amdgcn_s_memrealtime(); }
#define ESKF_CSTAMP(K) { if (lane == 0 && (vblock & 7) == 0 && (vblock >> 3) < 4096) g_c_stamps[(K) * 4096 + (vblock >> 3)] = __builtin_amdgcn_s_memrealtime(); }
hipError_t x_read_stamps(unsigned long long* out) { return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_x_stamps), sizeof(g_x_stamps)); }
#elif defined(ESKF_X_RUNSTAMPS) && defined(ESKF_TU_FUSED)   // run-kernel stamps: the pre phase of a block's LAST step, wavefront 0
__device__ unsigned long long g_pre_stamps[1024 * 8];
hipError_t x_read_pre_stamps(unsigned long long* out) { return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_pre_stamps), sizeof(g_pre_stamps)); }
#define ESKF_STAMP(K) { if (lane == 0 && vblock < 1024) g_pre_stamps[vblock * 8 + (K)] = __builtin_amdgcn_s_memrealtime(); }
#define ESKF_CSTAMP(K) {}
#else
#define ESKF_STAMP(K) {}
#define ESKF_CSTAMP(K) {}
#endif
// The kernel's own arguments as they lie in the kernarg segment (constant address space).  A wavefront of
// eskf_step_kernel that works through several pairs re-reads what it needs from there at three points of every round
// (cov_rows_body<WAVE>::refresh_args) through a pointer the compiler cannot see through: 106 dwords of arguments are more
// than a wavefront has scalar registers, and read once in front of the round loop they are all kept live across it --
// what does not fit is parked in lanes of vector registers, 260 v_readlane / v_writelane per round (a third on top of
// the round's own vector instructions; measured on the first build of that kernel).
template <typename T>
struct StepKernArgs { StepArgs a; DevParams<T> prm; };
template <typename T>
struct RunKernArgs { StepArgs a; DevParams<T> prm; RunArgs ra; };     // eskf_run_kernel: the same two, then the run
// Pointers that come out of the kernarg segment through load_kernarg are generic to the compiler (it cannot see that
// they were kernel arguments): every access through them would be a FLAT instruction -- which counts on the LDS counter
// as well as on the vector-memory one, so that a wait for an LDS exchange also waits for the covariance stores in
// flight.  globalise() tells it what they are: device memory.
template <typename P>
__device__ __forceinline__ P* globalise(P* p) {
    // the pointer's bits through an opaque statement, back as a GLOBAL pointer, cast to generic: the address-space
    // inference then turns every access derived from it into a global one (a global address is the same number in both
    // address spaces; neither a generic -> global -> generic cast pair nor llvm.assume(!is_shared / !is_private) survived
    // the optimiser here)
    typedef P __attribute__((address_space(1))) * gp_t;
    unsigned long long bits = (unsigned long long)p;
    asm("" : "+s"(bits));
    return (P*)(gp_t)bits;
}
__device__ __forceinline__ void globalise(StepArgs& a) {
    a.P = globalise(a.P); a.xs = globalise(a.xs); a.dt = globalise(a.dt); a.om = globalise(a.om); a.acc = globalise(a.acc);
    a.pos = globalise(a.pos); a.quat = globalise(a.quat); a.notch = globalise(a.notch); a.K_out = globalise(a.K_out);
    a.status = globalise(a.status); a.J = globalise(a.J); a.aux = globalise(a.aux); a.e = globalise(a.e);
    a.status_ws = globalise(a.status_ws); a.noise = globalise(a.noise); a.P_src = globalise(a.P_src); a.xs_src = globalise(a.xs_src);
}
template <typename S>
__device__ __forceinline__ S load_kernarg(const char __attribute__((address_space(4))) * p) {
    static_assert(sizeof(S) % 4 == 0, "whole dwords");
    S r;
    const uint32_t __attribute__((address_space(4))) * w = (const uint32_t __attribute__((address_space(4))) *)p;
#pragma unroll
    for (unsigned i = 0; i < sizeof(S) / 4; ++i) reinterpret_cast<uint32_t*>(&r)[i] = w[i];
    return r;
}
// eskf_run_kernel: the arguments of step k of the run -- inputs moved on by k strides; from the second step on the step
// is in place and injects what the step before left behind
template <typename T>
__device__ __forceinline__ void load_run_step_args(int k, StepArgs& a, DevParams<T>& prm) {
    const char __attribute__((address_space(4))) * ka = (const char __attribute__((address_space(4))) *)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(ka));
    a = load_kernarg<StepArgs>(ka + offsetof(RunKernArgs<T>, a));
    globalise(a);
    prm = load_kernarg<DevParams<T>>(ka + offsetof(RunKernArgs<T>, prm));
    const RunArgs ra = load_kernarg<RunArgs>(ka + offsetof(RunKernArgs<T>, ra));
    a.dt = reinterpret_cast<const char*>(a.dt) + (int64_t)k * ra.dt_step;
    a.om = reinterpret_cast<const char*>(a.om) + (int64_t)k * ra.imu_step;
    a.acc = reinterpret_cast<const char*>(a.acc) + (int64_t)k * ra.imu_step;
    if (a.do_update) {
        a.pos = reinterpret_cast<const char*>(a.pos) + (int64_t)k * ra.pos_step;
        a.quat = reinterpret_cast<const char*>(a.quat) + (int64_t)k * ra.quat_step;
        a.notch = reinterpret_cast<const char*>(a.notch) + (int64_t)k * ra.notch_step;
        if (a.status) a.status += (int64_t)k * ra.status_step;
    }
    if (k > 0) {
        a.P_src = a.P;
        a.xs_src = a.xs;
        a.drop_pending = 0;
        a.pending_hint = a.do_update ? 1 : a.pending_hint;
    }
}
// exchange words of a lane sit in its own staging row, on block words that are only written after barrier 1b
constexpr int PRE4_XOFF = J_BM;
// The one-launch kernels (IMG != 0): the staging rows do not lie one behind the other but in the UPPER halves of the
// block's exchange images (IMG words per instance, sixteen images per block; four rows of 75 words = words 300..599 of an
// image), so that the lower halves -- the packed covariance of the sixteen instances a block's last round has just
// worked on -- survive the next step's pre phase (eskf_run_kernel: those covariances never leave the chip between two
// steps, see step_block).  Row of the block's instance s:
template <int IMG>
__device__ __forceinline__ int staging_row_word(int s) {
    if constexpr (IMG == 0) return s * J_CORE;
    else return (s >> 2) * IMG + 300 + (s & 3) * J_CORE;
}
// eskf_run_kernel, between two steps of a run: the error state of a step's update and its status word wait for the next
// step's pre phase in LDS (ES words per instance: e 24 | status word), not in memory -- see step_block.  They lie IN the
// instance's staging row, on record words that the pre phase only writes after its barrier 1b (Cm[1..17], Dm[0..7]:
// formed by W2 / W3 from the published trigonometry), while every chain reads its part of the error state before barrier 1.
constexpr int ES = 25;
constexpr int E_ROW = J_CM + 1;             // first word of the ES words within the staging row
static_assert(E_ROW > J_BM + 9 && E_ROW + ES <= J_EM, "error state words: behind the exchange words, on Cm / Dm only");
__device__ __forceinline__ double status_as_word(int st, double) { return __longlong_as_double((long long)(unsigned)st); }
__device__ __forceinline__ float status_as_word(int st, float) { return __int_as_float(st); }
__device__ __forceinline__ int word_as_status(double w) { return (int)(unsigned)__double_as_longlong(w); }
__device__ __forceinline__ int word_as_status(float w) { return __float_as_int(w); }
template <typename T>
__device__ __forceinline__ void load_epend(const StepArgs& a, int64_t b, int i0, int n, T* dst, const bool e_lds = false, const T* sE_row = nullptr) {
    if (e_lds) {
#pragma unroll
        for (int i = 0; i < n; ++i) dst[i] = sE_row[i0 + i];
        return;
    }
#pragma unroll
    for (int i = 0; i < n; ++i) dst[i] = reinterpret_cast<const T*>(a.e)[b * 24 + i0 + i];
}
// KEEP (eskf_step_kernel): the block's first NI lanes carry its NI instances and the records STAY in the staging rows --
// the covariance rounds of the same block take them from there; nothing is written to a.J.
template <typename T, bool KEEP = false, int NI = 64, int IMG = 0>
__device__ __forceinline__ void pre4_body(const StepArgs& a, const DevParams<T>& prm, unsigned char* smem_raw, const int vblock,
                                          const bool e_lds = false, const int lane_in = -1, const int role_in = -1) {
    static_assert(IMG == 0 || KEEP, "rows inside the exchange images: the one-launch kernels");
    // e_lds (eskf_run_kernel, from the second step of a run on): the error states and status words the block's own rounds
    // left in LDS for this pre phase (words E_ROW.. of the instance's staging row); else they are in memory (a.e, a.status_ws)
    // (No re-reading of the arguments from the kernarg segment in here, as the covariance body does: tried for the run
    //  kernel, it changed where the compiler contracts a*b+c in the chains -- one ulp in x after one step -- and with
    //  MachineLICM off for these kernels it is not needed.)
    T* sJ = reinterpret_cast<T*>(smem_raw);
    const int JS = (!KEEP && prm.jfull) ? J_FULL : J_CORE;   // staging rows = whole records: residual words, then the blocks (odd: conflict-free)
    // (the one-launch kernels hand their own lane / wavefront numbers in: inside a time loop the thread index of the
    //  kernel's entry would have to be kept -- spilled -- across the whole step)
    const int lane = KEEP && lane_in >= 0 ? lane_in : (int)(threadIdx.x & 63);
    const int role = KEEP && role_in >= 0 ? role_in : __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);     // wave-uniform, and known to be
    const int64_t b0 = (int64_t)vblock * NI;
    const int64_t b_raw = b0 + lane;
    const bool live = (NI == 64 || lane < NI) && b_raw < a.B;
    const int64_t b_last = (b0 + NI - 1 < a.B) ? b0 + NI - 1 : a.B - 1;
    const int64_t b = live ? b_raw : b_last;           // tail lanes recompute the block's last instance, write nothing
    const T* xi = reinterpret_cast<const T*>(a.xs_src);
    const int64_t B = a.B;
    auto X = [&](int w) { return xi[(int64_t)w * B + b]; };
    // deferred injection: device state, see eskf_pre_kernel
    const T* sE_row = sJ + staging_row_word<IMG>(NI == 64 || lane < NI ? lane : NI - 1) + E_ROW;
    const bool e_wanted = e_lds || a.pending_hint;
    int pend_status = 0;
    if (e_lds) pend_status = word_as_status(sE_row[24]);
    else if (a.pending_hint) pend_status = a.status_ws[b];
    const bool was_pending = (pend_status & DVI_STATUS_PENDING) != 0;
    const bool inject_now = was_pending && !a.drop_pending && !(pend_status & DVI_STATUS_SINGULAR);
    const T dt = reinterpret_cast<const T*>(a.dt)[a.dt_stride ? b : 0];
    T* row = sJ + (IMG ? staging_row_word<IMG>(lane) : lane * JS);     // this lane's staging row
    Jac<T>& J = *reinterpret_cast<Jac<T>*>(row + J_B0);        // its blocks (W2, W3)
    T* xw = row + PRE4_XOFF;                                   // exchange words (trigonometry): on Bm / Cm, which W2 only forms after barrier 1b
    // the non-finite flag of the residual is OR-ed together by W0 (r_p) and W1 (the rest); W2 clears it before barrier 1
    auto raise_flag = [&]() {
        if constexpr (sizeof(T) == 8) __hip_atomic_fetch_or(reinterpret_cast<unsigned long long*>(row + J_RES + 7), 0x3ff0000000000000ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        else __hip_atomic_fetch_or(reinterpret_cast<unsigned*>(row + J_RES + 7), 0x3f800000u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    };
    if (role == 0) ESKF_STAMP(0)

    if (role == 0) {
        // ---- W0: position, velocity, attitude --------------------------------------------------------
        T p[3], v[3], q[4], pc[3], om_o[3], acc_o[3], q456[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            p[i] = X(0 + i); v[i] = X(3 + i); pc[i] = X(19 + i); om_o[i] = X(26 + i); acc_o[i] = X(29 + i);
            q456[i] = X(13 + i);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) q[i] = X(6 + i);
        T e9[9], e1215[3], e1821[3];
        if (e_wanted) { load_epend(a, b, 0, 9, e9, e_lds, sE_row); load_epend(a, b, 12, 3, e1215, e_lds, sE_row); load_epend(a, b, 18, 3, e1821, e_lds, sE_row); }
        const int64_t so = a.imu_stride ? b * 3 : 0;
        T om_n[3], acc_n[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) { om_n[i] = reinterpret_cast<const T*>(a.om)[so + i]; acc_n[i] = reinterpret_cast<const T*>(a.acc)[so + i]; }
        T gpos[3] = {T(0), T(0), T(0)};
        if (a.do_update) {
            const T* gp = reinterpret_cast<const T*>(a.pos) + (a.meas_stride ? b : 0) * 3;
#pragma unroll
            for (int i = 0; i < 3; ++i) gpos[i] = gp[i];
        }
#if defined(ESKF_X_STAMPS) || defined(ESKF_X_RUNSTAMPS)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
        ESKF_STAMP(1)
        if (inject_now) {                               // state.py:46-60,105-129
            T th[3] = {e9[6], e9[7], e9[8]}, dq[4], qn[4];
            quat_about_axis(norm3(th), th, dq);
            quat_mul(q, dq, qn);
            quat_nrm(qn);
#pragma unroll
            for (int i = 0; i < 3; ++i) { p[i] += e9[i]; v[i] += e9[3 + i]; pc[i] += e1821[i]; q456[i] += e1215[i]; }
#pragma unroll
            for (int i = 0; i < 4; ++i) q[i] = qn[i];
        }
        T R[9], om_avg[3], w[3], RS[9], Rn[9], a1[3], a2[3], acc_avg[3], qn[4];
        quat_to_mat(q, R);
#pragma unroll
        for (int i = 0; i < 3; ++i) { om_avg[i] = (om_o[i] + om_n[i]) * T(0.5); w[i] = dt * om_avg[i]; }
        mat3_skew(R, w, RS);
#pragma unroll
        for (int i = 0; i < 9; ++i) Rn[i] = R[i] + RS[i];                           // equations.py:46
        mat3_vec(R, acc_o, a1);
        mat3_vec(Rn, acc_n, a2);
#pragma unroll
        for (int i = 0; i < 3; ++i) acc_avg[i] = (a1[i] + a2[i]) * T(0.5);          // equations.py:47
        mat_to_quat(Rn, qn);                                                        // state.py:67
        ESKF_STAMP(2)
        __syncthreads();                                // barrier 1: old state read by everyone, trigonometry published
        ESKF_STAMP(3)
        ProbeTrig<T> t;
        t.s1 = xw[0]; t.c1 = xw[1]; t.s2 = xw[2]; t.c2 = xw[3];
        t.s3 = xw[4]; t.c3 = xw[5];
        __syncthreads();                                // barrier 1b: everyone holds the trigonometry, the rows may be written
        T pB[3];
        {
            T z6[3], e5[3], e6[3];
            z6[0] = -t.s1 * t.c2; z6[1] = -t.c1 * t.c2; z6[2] = t.s2;
            E5(t.s3, t.c3, e5); E6(t.s3, t.c3, e6);
            const T lq = prm.L - q456[0];
#pragma unroll
            for (int i = 0; i < 3; ++i) pB[i] = lq * z6[i] + q456[1] * e5[i] + q456[2] * e6[i];
        }
        T cr[3], Rcr[3];
        cross3(om_avg, pB, cr);                                                     // BB_v_CB == 0
        mat3_vec(R, cr, Rcr);
        T rp[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const T vi = v[i];
            p[i] = p[i] + dt * vi + (dt * dt * T(0.5)) * acc_avg[i];                // equations.py:76
            v[i] = vi + dt * acc_avg[i];                                            // equations.py:77
            pc[i] = pc[i] + dt * vi + dt * Rcr[i];                                  // equations.py:83-85
            rp[i] = gpos[i] - pc[i];                                                // Filter.py:369
        }
        T* xo = reinterpret_cast<T*>(a.xs);
        const int64_t Bo = a.B;
        if (a.do_update) {
#pragma unroll
            for (int i = 0; i < 3; ++i) row[J_RES + i] = rp[i];
            // a NaN / inf measurement or state propagates exactly as it does in the reference (NumPy arithmetic, no
            // check); the instance is flagged, its neighbours are not affected
            if (!t_finite(rp[0] + rp[1] + rp[2])) raise_flag();
        }
        if (live) {
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                xo[(0 + i) * Bo + b] = p[i]; xo[(3 + i) * Bo + b] = v[i]; xo[(19 + i) * Bo + b] = pc[i];
                xo[(26 + i) * Bo + b] = om_n[i]; xo[(29 + i) * Bo + b] = acc_n[i];    // Filter.py:225-227
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) { xo[(6 + i) * Bo + b] = qn[i]; xo[(32 + i) * Bo + b] = qn[i]; }
        }
    } else if (role == 1) {
        // ---- W1: camera attitude, notch states, rotation and notch residuals -------------------------------
        T qc[4], notch[3], om_o[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) { notch[i] = X(16 + i); om_o[i] = X(26 + i); }
#pragma unroll
        for (int i = 0; i < 4; ++i) qc[i] = X(22 + i);
        T e69[3], e1518[3], e2124[3];
        if (e_wanted) { load_epend(a, b, 6, 3, e69, e_lds, sE_row); load_epend(a, b, 15, 3, e1518, e_lds, sE_row); load_epend(a, b, 21, 3, e2124, e_lds, sE_row); }
        T om_w[3] = {om_o[0], om_o[1], om_o[2]};        // what the camera attitude is propagated with: om_old (quirk Q9) ...
        if (!(prm.quirks & QK_Q9)) {                    // ... or the average of the old and the new sample
            const int64_t so = a.imu_stride ? b * 3 : 0;
#pragma unroll
            for (int i = 0; i < 3; ++i) om_w[i] = (om_o[i] + reinterpret_cast<const T*>(a.om)[so + i]) * T(0.5);
        }
        T qraw[4] = {T(0), T(0), T(0), T(1)}, ang_notch = T(0);
        if (a.do_update) {
            const int64_t mo = a.meas_stride ? b : 0;
            const T* gq = reinterpret_cast<const T*>(a.quat) + mo * 4;
#pragma unroll
            for (int i = 0; i < 4; ++i) qraw[i] = gq[i];
            ang_notch = reinterpret_cast<const T*>(a.notch)[mo];
        }
        if (inject_now) {
            T dqc[4], qcn[4];
            quat_about_axis(norm3(e2124), (prm.quirks & QK_Q3) ? e69 : e2124, dqc);   // sic (state.py:124) unless Q3 is off
            quat_mul(qc, dqc, qcn);
            quat_nrm(qcn);
#pragma unroll
            for (int i = 0; i < 4; ++i) qc[i] = qcn[i];
#pragma unroll
            for (int i = 0; i < 3; ++i) notch[i] += e1518[i];
        }
        T Rc[9], qm[4];
        quat_to_mat(qc, Rc);
        if (a.do_update) {                                                          // Filter.py:364-365
            T nq[4] = {T(0), T(0), T(0), T(1)};
            t_sincos(T(0.5) * ang_notch, nq[2], nq[3]);          // Quaternion(val=[0,0,ang], euler="xyz")
            quat_mul(nq, qraw, qm);
            quat_nrm(qm);
            qm[0] = -qm[0]; qm[1] = -qm[1]; qm[2] = -qm[2];      // conjugate
        }
        __syncthreads();                                // barrier 1
        ProbeTrig<T> t;
        t.s1 = xw[0]; t.c1 = xw[1]; t.s2 = xw[2]; t.c2 = xw[3];
        t.sp = xw[6]; t.cp = xw[7];
        __syncthreads();                                // barrier 1b
        T RBC[9], z6[3];
        {
            T aa[3], g[3];
            z6[0] = -t.s1 * t.c2; z6[1] = -t.c1 * t.c2; z6[2] = t.s2;
            E5(t.sp, t.cp, aa); E6(t.sp, t.cp, g);
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                RBC[3 * i + 0] = -aa[i];
                RBC[3 * i + 1] = prm.ct * g[i] + prm.st * z6[i];
                RBC[3 * i + 2] = -prm.st * g[i] + prm.ct * z6[i];
            }
        }
        T u[3], C_om[3], wc[3], RcS[9], Rcn[9], qcn[4];
#pragma unroll
        for (int i = 0; i < 3; ++i) u[i] = om_w[i] + z6[i] * notch[1];
        mat3T_vec(RBC, u, C_om);                                                    // equations.py:50
#pragma unroll
        for (int i = 0; i < 3; ++i) wc[i] = dt * C_om[i];
        mat3_skew(Rc, wc, RcS);
#pragma unroll
        for (int i = 0; i < 9; ++i) Rcn[i] = Rc[i] + RcS[i];                        // equations.py:86
        mat_to_quat(Rcn, qcn);                                                      // state.py:72
        notch[0] += dt * notch[1];                                                  // equations.py:80-82
        notch[1] += dt * notch[2];
        T r[4] = {T(0), T(0), T(0), T(0)}, flag = T(0);
        if (a.do_update) {                                                          // Filter.py:370-372
            T qe[4];
            quat_mul(qm, qcn, qe);
            quat_nrm(qe);
            T nv = t_sqrt(qe[0] * qe[0] + qe[1] * qe[1] + qe[2] * qe[2]);
            if (nv > T(1)) { nv = T(1); flag = T(1); }
            const T ang = residual_angle(nv, prm.quirks);             // half-angle residual (Quaternion.py:138-147) unless Q6 is off
            const T f = (nv == T(0)) ? T(0) : ang / nv;
#pragma unroll
            for (int i = 0; i < 3; ++i) r[i] = f * qe[i];
            r[3] = ang_notch - notch[0];
        }
        T* xo = reinterpret_cast<T*>(a.xs);
        const int64_t Bo = a.B;
        if (live) {
#pragma unroll
            for (int i = 0; i < 3; ++i) xo[(16 + i) * Bo + b] = notch[i];
#pragma unroll
            for (int i = 0; i < 4; ++i) xo[(22 + i) * Bo + b] = qcn[i];
        }
        if (a.do_update) {
#pragma unroll
            for (int i = 0; i < 4; ++i) row[J_RES + 3 + i] = r[i];
            if (!t_finite(r[0] + r[1] + r[2] + r[3])) flag = T(1);
            if (flag != T(0)) raise_flag();                       // asin argument was clamped, or non-finite residual
        }
        // (with the status word in LDS the one in memory is not touched: the rounds of the run's last step write it anew)
        if (live && was_pending && !e_lds) a.status_ws[b] = pend_status & ~DVI_STATUS_PENDING;  // the injection is done (or dropped)
        __syncthreads();                                // barrier 2: the 64 staged records are complete
    } else {
        // ---- W2 / W3: Jacobian blocks (Filter.py:249-342), evaluated at the post-predict dofs / notch with the
        // pre-predict buffers om_old, acc_old, R_WB_old
        T dofs[6], notch[3], om_o[3], e918[9];
#pragma unroll
        for (int i = 0; i < 6; ++i) dofs[i] = X(10 + i);
#pragma unroll
        for (int i = 0; i < 3; ++i) { notch[i] = X(16 + i); om_o[i] = X(26 + i); }
        if (e_wanted) load_epend(a, b, 9, 9, e918, e_lds, sE_row);
        T sig_om[3] = {prm.sig_om[0], prm.sig_om[1], prm.sig_om[2]};
        if (a.noise) {
#pragma unroll
            for (int i = 0; i < 3; ++i) sig_om[i] = reinterpret_cast<const T*>(a.noise)[b * NOISE_WORDS + 20 + i];
        }
        T acc_o[3], qo[4];
        if (role == 2) {
#pragma unroll
            for (int i = 0; i < 3; ++i) acc_o[i] = X(29 + i);
#pragma unroll
            for (int i = 0; i < 4; ++i) qo[i] = X(32 + i);
        }
        if (inject_now) {
#pragma unroll
            for (int i = 0; i < 6; ++i) dofs[i] += e918[i];
#pragma unroll
            for (int i = 0; i < 3; ++i) notch[i] += e918[6 + i];
        }
        // what the nominal propagation evaluates the probe at (pre-predict) ...
        const T q1n = dofs[0], q2n = dofs[1], q3n = dofs[2];
        const T phi = q3n - notch[0];
        // ... and what the Jacobians are evaluated at (post-predict): frozen DOFs zeroed, notch advanced
        bool zeroed[3] = {false, false, false};
        if (!prm.hold_frozen) {                                                     // Filter.py:243-245
#pragma unroll
            for (int i = 0; i < 6; ++i)
                if ((prm.frozen_mask >> i) & 1) { dofs[i] = T(0); if (i < 3) zeroed[i] = true; }
        }
        notch[0] += dt * notch[1];                                                  // equations.py:80-82
        notch[1] += dt * notch[2];
        const T phi_n = dofs[2] - notch[0];
        T Ro[9];
        if (role == 2) {
            T s1, c1, s2, c2;
            t_sincos(q1n, s1, c1);
            t_sincos(q2n, s2, c2);
            xw[0] = s1; xw[1] = c1; xw[2] = s2; xw[3] = c2;
            row[J_RES + 7] = T(0);                      // (the residual's flag word: raised by W0 / W1 after barrier 1b)
            jac_blocks_old_buffers(qo, acc_o, om_o, dt, J, Ro);
        } else {
            T s3, c3, sp, cp, spn, cpn;
            t_sincos(q3n, s3, c3);
            t_sincos(phi, sp, cp);
            if (phi_n == phi) { spn = sp; cpn = cp; }
            else t_sincos(phi_n, spn, cpn);
            xw[4] = s3; xw[5] = c3; xw[6] = sp; xw[7] = cp;
            xw[8] = spn; xw[9] = cpn;
        }
        __syncthreads();                                // barrier 1
        ProbeTrig<T> t;                                 // at the post-predict dofs / notch
        {
            const T s1 = xw[0], c1 = xw[1], s2 = xw[2], c2 = xw[3], s3 = xw[4], c3 = xw[5];
            t.s1 = zeroed[0] ? T(0) : s1; t.c1 = zeroed[0] ? T(1) : c1;
            t.s2 = zeroed[1] ? T(0) : s2; t.c2 = zeroed[1] ? T(1) : c2;
            t.s3 = zeroed[2] ? T(0) : s3; t.c3 = zeroed[2] ? T(1) : c3;
            t.sp = xw[8]; t.cp = xw[9];
        }
        __syncthreads();                                // barrier 1b
        const T thd = notch[1];
        T om_t[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) om_t[i] = om_o[i] - sig_om[i];
        if (role == 2) {
            T* xo = reinterpret_cast<T*>(a.xs);
            const int64_t Bo = a.B;
            if (live) {
#pragma unroll
                for (int i = 0; i < 6; ++i) xo[(10 + i) * Bo + b] = dofs[i];
            }
            jac_blocks_position(prm, t, dofs, om_t, Ro, dt, J);
        } else {
            jac_blocks_attitude(prm, t, thd, om_t, om_o, dt, J);
        }
    }
    if (role == 0) ESKF_STAMP(4)
    if (role != 1) __syncthreads();                     // barrier 2: the 64 staged records are complete
    if (role == 0) ESKF_STAMP(5)
    if constexpr (!KEEP) {
        T* gJ = reinterpret_cast<T*>(a.J) + b0 * JW;
        const int nrec = (a.B - b0 < 64) ? (int)(a.B - b0) : 64;
        const int first = role * 16, last = (first + 16 < nrec) ? first + 16 : nrec;
        if (prm.jfull) store_records<T, J_FULL>(sJ, gJ, last, lane, first);
        else store_records<T, J_CORE>(sJ, gJ, last, lane, first);
    }
    if (role == 0) ESKF_STAMP(6)
#ifdef ESKF_X_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (role == 0) ESKF_STAMP(7)
#endif
}
template <typename T>
__global__ __launch_bounds__(256, 4) void eskf_pre4_kernel(StepArgs a, DevParams<T> prm) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    pre4_body<T>(a, prm, smem_raw, (int)blockIdx.x);
}

// ----------------------------------------------------------------------------------
// Epochs (launches with n >= 2 predicts: Filter.run_one_epoch / propagate_imu, Filter.py:170-217): the pre work of a
// launch as two kernels.
//
//   eskf_nominal_kernel   lane per instance, serial over the predicts -- but only what IS a recurrence: wavefront 0
//                         carries attitude, position and velocity (q -> R -> R+ -> Markley, ~150 operations per
//                         predict), wavefront 1 the camera attitude and the notch states (and forms the residual at
//                         the end).  p_CB does not depend on the notch angle, so it is a constant of the epoch; the
//                         sines / cosines of q1..q3 are evaluated once, phi = q3 - q7 only when the notch moves.  Per
//                         predict the kernel leaves behind what that predict's Jacobian blocks need and only a
//                         recurrence can know: q_old(k), the post-predict notch angle and rate, sin / cos of the
//                         post-predict phi (8 words in `aux`, lane-contiguous).
//   eskf_jac_kernel       lane per (predict, instance): the Jacobian blocks of all n predicts of all instances at
//                         once -- they depend on q_old(k), the notch state after predict k, om(k-1), acc(k-1), dt(k) and
//                         the extrinsics of the epoch, not on each other -- with no transcendental left in it.
//
// The two-role eskf_pre_kernel ran the blocks of the n predicts one after the other in ONE wavefront per 64 instances
// next to a 900-operation nominal chain (config #5, 33 predicts, B = 8 192: 118 us for 79 us of covariance kernel; 128
// wavefronts on a chip that holds 4 096).
// aux layout, word w of instance b at aux[w * B + b]:   0..2 om_old(0) | 3..5 acc_old(0) | 6..11 s1 c1 s2 c2 s3 c3 at
// the post-predict dofs | then per predict k, 8 words from 12 + 8 k: q_old(k) 4 | notch+ 2 (angle, rate) | sin, cos phi+
// ----------------------------------------------------------------------------------
constexpr int AUX0 = 12, AUXK = 8;
// position, velocity and camera position of one predict (equations.py:76-77,83-85), shared by eskf_nominal_kernel and the
// long-horizon kernel (eskf_epochs.hip) with every fusion spelled out: where a*b+c is contracted must not depend on the
// kernel the statement is inlined into -- the two paths are compared bit for bit
template <typename T>
__device__ __forceinline__ void advance_p_v_pc(T dt, const T (&acc_avg)[3], const T (&Rcr)[3], T (&p)[3], T (&v)[3], T (&pc)[3]) {
#pragma clang fp contract(off)
    // (the fusions of `p + dt * v + (dt * dt / 2) * a` etc. as the front end contracts these expressions where they are written
    //  out -- pre4_body, predict_scalar --: a single predict gives the same bits through any of the pre kernels)
    const T h = dt * dt * T(0.5);
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const T vi = v[i];
        p[i] = t_fma(h, acc_avg[i], t_fma(dt, vi, p[i]));                       // equations.py:76
        v[i] = t_fma(dt, acc_avg[i], vi);                                       // equations.py:77
        pc[i] = t_fma(dt, Rcr[i], t_fma(dt, vi, pc[i]));                        // equations.py:83-85
    }
}
template <typename T, bool SIMU>
__global__ __launch_bounds__(128, 2) void eskf_nominal_kernel(StepArgs a, DevParams<T> prm) {
    __shared__ T sX[9 * 64];                            // s1 c1 s2 c2 s3 c3 (pre-predict dofs) | p_cam after the last predict
    const int lane = threadIdx.x & 63;
    const int role = threadIdx.x >> 6;                 // wave-uniform
    const int64_t hi = a.sub_count ? a.sub_first + a.sub_count : a.B;       // (a launch may cover a part of the batch)
    const int64_t b_raw = a.sub_first + (int64_t)blockIdx.x * 64 + lane;
    const bool live = b_raw < hi;
    const int64_t b = live ? b_raw : hi - 1;           // tail lanes recompute the last instance, write nothing
    const T* xi = reinterpret_cast<const T*>(a.xs_src);
    T* xo = reinterpret_cast<T*>(a.xs);
    T* aux = reinterpret_cast<T*>(a.aux);
    const int64_t B = a.B;
    const int n = a.n_pred;
    auto X = [&](int w) { return xi[(int64_t)w * B + b]; };
    int pend_status = 0;
    if (a.pending_hint) pend_status = a.status_ws[b];
    const bool was_pending = (pend_status & DVI_STATUS_PENDING) != 0;
    const bool inject_now = was_pending && !a.drop_pending && !(pend_status & DVI_STATUS_SINGULAR);
    bool zeroed[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) zeroed[i] = !prm.hold_frozen && ((prm.frozen_mask >> i) & 1);   // Filter.py:243-245

    if (role == 0) {
        // ---- position, velocity, attitude ---------------------------------------------------------------
        T p[3], v[3], q[4], pc[3], om_o[3], acc_o[3], qo[4], dofs[6];
#pragma unroll
        for (int i = 0; i < 3; ++i) { p[i] = X(i); v[i] = X(3 + i); pc[i] = X(19 + i); om_o[i] = X(26 + i); acc_o[i] = X(29 + i); }
#pragma unroll
        for (int i = 0; i < 4; ++i) { q[i] = X(6 + i); qo[i] = X(32 + i); }
#pragma unroll
        for (int i = 0; i < 6; ++i) dofs[i] = X(10 + i);
        T e15[15], e1821[3];
        if (a.pending_hint) { load_epend(a, b, 0, 15, e15); load_epend(a, b, 18, 3, e1821); }
        T dt, om_n[3], acc_n[3];
        load_imu_sample(a, 0, b, dt, om_n, acc_n);
        if (inject_now) {                               // state.py:46-60,105-129
            T th[3] = {e15[6], e15[7], e15[8]}, dq[4], qn[4];
            quat_about_axis(norm3(th), th, dq);
            quat_mul(q, dq, qn);
            quat_nrm(qn);
#pragma unroll
            for (int i = 0; i < 3; ++i) { p[i] += e15[i]; v[i] += e15[3 + i]; pc[i] += e1821[i]; }
#pragma unroll
            for (int i = 0; i < 6; ++i) dofs[i] += e15[9 + i];
#pragma unroll
            for (int i = 0; i < 4; ++i) q[i] = qn[i];
        }
        // probe: p_CB at the dofs the first predict sees and -- frozen DOFs are zeroed by it -- at the ones every later
        // predict (and every Jacobian block) sees; the trigonometry of the latter goes to the Jacobian kernel
        ProbeTrig<T> t0, t1;
        t_sincos(dofs[0], t0.s1, t0.c1);
        t_sincos(dofs[1], t0.s2, t0.c2);
        t_sincos(dofs[2], t0.s3, t0.c3);
        t0.sp = t0.cp = T(0);
        t1 = t0;
        if (zeroed[0]) { t1.s1 = T(0); t1.c1 = T(1); }
        if (zeroed[1]) { t1.s2 = T(0); t1.c2 = T(1); }
        if (zeroed[2]) { t1.s3 = T(0); t1.c3 = T(1); }
        sX[0 * 64 + lane] = t0.s1; sX[1 * 64 + lane] = t0.c1; sX[2 * 64 + lane] = t0.s2; sX[3 * 64 + lane] = t0.c2;
        sX[4 * 64 + lane] = t0.s3; sX[5 * 64 + lane] = t0.c3;
        auto probe_p = [&](const ProbeTrig<T>& t, const T* d, T* pB) {
            T z6[3], e5[3], e6[3];
            z6[0] = -t.s1 * t.c2; z6[1] = -t.c1 * t.c2; z6[2] = t.s2;
            E5(t.s3, t.c3, e5); E6(t.s3, t.c3, e6);
            const T lq = prm.L - d[3];
#pragma unroll
            for (int i = 0; i < 3; ++i) pB[i] = lq * z6[i] + d[4] * e5[i] + d[5] * e6[i];
        };
        T pB0[3], pB1[3];
        probe_p(t0, dofs, pB0);
#pragma unroll
        for (int i = 0; i < 6; ++i) if (zeroed[i]) dofs[i] = T(0);
        probe_p(t1, dofs, pB1);
        if (live) {
#pragma unroll
            for (int i = 0; i < 3; ++i) { aux[(0 + i) * B + b] = om_o[i]; aux[(3 + i) * B + b] = acc_o[i]; }
            aux[6 * B + b] = t1.s1; aux[7 * B + b] = t1.c1; aux[8 * B + b] = t1.s2; aux[9 * B + b] = t1.c2;
            aux[10 * B + b] = t1.s3; aux[11 * B + b] = t1.c3;
#pragma unroll
            for (int i = 0; i < 4; ++i) aux[(AUX0 + i) * B + b] = qo[i];       // q_old of the first predict: the stale buffer (Filter.py:227)
        }
        __syncthreads();                                // A: trigonometry published, old state read by both wavefronts
        if (live) {
#pragma unroll
            for (int i = 0; i < 6; ++i) xo[(10 + i) * B + b] = dofs[i];
        }
        for (int k = 0; k < n; ++k) {
            T dt2, om2[3], acc2[3];                     // the next sample is on its way while this predict is computed
            load_imu_sample<T, SIMU>(a, k + 1, b, dt2, om2, acc2);
            T R[9], om_avg[3], w[3], RS[9], Rn[9], a1[3], a2[3], acc_avg[3], cr[3], Rcr[3];
            quat_to_mat(q, R);
#pragma unroll
            for (int i = 0; i < 3; ++i) { om_avg[i] = (om_o[i] + om_n[i]) * T(0.5); w[i] = dt * om_avg[i]; }
            mat3_skew(R, w, RS);
#pragma unroll
            for (int i = 0; i < 9; ++i) Rn[i] = R[i] + RS[i];                       // equations.py:46
            mat3_vec(R, acc_o, a1);
            mat3_vec(Rn, acc_n, a2);
#pragma unroll
            for (int i = 0; i < 3; ++i) acc_avg[i] = (a1[i] + a2[i]) * T(0.5);      // equations.py:47
            cross3(om_avg, k == 0 ? pB0 : pB1, cr);                                 // BB_v_CB == 0
            mat3_vec(R, cr, Rcr);
            advance_p_v_pc(dt, acc_avg, Rcr, p, v, pc);
#pragma unroll
            for (int i = 0; i < 3; ++i) { om_o[i] = om_n[i]; acc_o[i] = acc_n[i]; }  // Filter.py:225-226
            mat_to_quat(Rn, q);                                                     // state.py:67
            if (k + 1 < n && live) {
#pragma unroll
                for (int i = 0; i < 4; ++i) aux[(AUX0 + AUXK * (k + 1) + i) * B + b] = q[i];   // = R_WB_old of predict k + 1 (Filter.py:227)
            }
            dt = dt2;
#pragma unroll
            for (int i = 0; i < 3; ++i) { om_n[i] = om2[i]; acc_n[i] = acc2[i]; }
        }
#pragma unroll
        for (int i = 0; i < 3; ++i) sX[(6 + i) * 64 + lane] = pc[i];
        __syncthreads();                                // B: p_cam for the residual
        if (live) {
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                xo[(0 + i) * B + b] = p[i]; xo[(3 + i) * B + b] = v[i]; xo[(19 + i) * B + b] = pc[i];
                xo[(26 + i) * B + b] = om_o[i]; xo[(29 + i) * B + b] = acc_o[i];
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) { xo[(6 + i) * B + b] = q[i]; xo[(32 + i) * B + b] = q[i]; }
        }
        return;
    }
    // ---- camera attitude, notch states, residual ---------------------------------------------------------
    T qc[4], notch[3], om_o[3], q3;
#pragma unroll
    for (int i = 0; i < 3; ++i) { notch[i] = X(16 + i); om_o[i] = X(26 + i); }
#pragma unroll
    for (int i = 0; i < 4; ++i) qc[i] = X(22 + i);
    q3 = X(12);
    T e69[3], e11 = T(0), e1518[3], e2124[3];
    if (a.pending_hint) {
        load_epend(a, b, 6, 3, e69); load_epend(a, b, 11, 1, &e11); load_epend(a, b, 15, 3, e1518); load_epend(a, b, 21, 3, e2124);
    }
    T gpos[3] = {T(0), T(0), T(0)}, qraw[4] = {T(0), T(0), T(0), T(1)}, ang_notch = T(0);
    if (a.do_update) {
        const int64_t mo = a.meas_stride ? b : 0;
        const T* gp = reinterpret_cast<const T*>(a.pos) + mo * 3;
        const T* gq = reinterpret_cast<const T*>(a.quat) + mo * 4;
#pragma unroll
        for (int i = 0; i < 3; ++i) gpos[i] = gp[i];
#pragma unroll
        for (int i = 0; i < 4; ++i) qraw[i] = gq[i];
        ang_notch = reinterpret_cast<const T*>(a.notch)[mo];
    }
    T dt, om_n[3], acc_n[3];
    load_imu_sample(a, 0, b, dt, om_n, acc_n);
    if (inject_now) {
        T dqc[4], qcn[4];
        quat_about_axis(norm3(e2124), (prm.quirks & QK_Q3) ? e69 : e2124, dqc);   // sic (state.py:124) unless Q3 is off
        quat_mul(qc, dqc, qcn);
        quat_nrm(qcn);
#pragma unroll
        for (int i = 0; i < 4; ++i) qc[i] = qcn[i];
#pragma unroll
        for (int i = 0; i < 3; ++i) notch[i] += e1518[i];
        q3 += e11;
    }
    T qm[4] = {T(0), T(0), T(0), T(1)};
    if (a.do_update) {                                                              // Filter.py:364-365
        T nq[4] = {T(0), T(0), T(0), T(1)};
        t_sincos(T(0.5) * ang_notch, nq[2], nq[3]);          // Quaternion(val=[0,0,ang], euler="xyz")
        quat_mul(nq, qraw, qm);
        quat_nrm(qm);
        qm[0] = -qm[0]; qm[1] = -qm[1]; qm[2] = -qm[2];      // conjugate
    }
    __syncthreads();                                    // A
    T s1 = sX[0 * 64 + lane], c1 = sX[1 * 64 + lane], s2 = sX[2 * 64 + lane], c2 = sX[3 * 64 + lane];
    T ph_arg = T(NAN), sp = T(0), cp = T(1);            // sin / cos of the phi they were last evaluated at
    for (int k = 0; k < n; ++k) {
        T dt2, om2[3], acc2[3];
        load_imu_sample<T, SIMU>(a, k + 1, b, dt2, om2, acc2);
        if (k == 1) {                                   // (from the second predict on the frozen DOFs are zero)
            if (zeroed[0]) { s1 = T(0); c1 = T(1); }
            if (zeroed[1]) { s2 = T(0); c2 = T(1); }
            if (zeroed[2]) q3 = T(0);
        }
        const T phi = q3 - notch[0];
        if (!(phi == ph_arg)) { ph_arg = phi; t_sincos(phi, sp, cp); }
        ProbeTrig<T> t;
        t.s1 = s1; t.c1 = c1; t.s2 = s2; t.c2 = c2; t.s3 = t.c3 = T(0); t.sp = sp; t.cp = cp;
        T RBC[9], z6[3];
        {
            T aa[3], g[3];
            z6[0] = -t.s1 * t.c2; z6[1] = -t.c1 * t.c2; z6[2] = t.s2;
            E5(t.sp, t.cp, aa); E6(t.sp, t.cp, g);
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                RBC[3 * i + 0] = -aa[i];
                RBC[3 * i + 1] = prm.ct * g[i] + prm.st * z6[i];
                RBC[3 * i + 2] = -prm.st * g[i] + prm.ct * z6[i];
            }
        }
        T u[3], C_om[3], wc[3], Rc[9], RcS[9], Rcn[9];
#pragma unroll
        for (int i = 0; i < 3; ++i) u[i] = ((prm.quirks & QK_Q9) ? om_o[i] : (om_o[i] + om_n[i]) * T(0.5)) + z6[i] * notch[1];   // (Q9)
        mat3T_vec(RBC, u, C_om);                                                    // equations.py:50
#pragma unroll
        for (int i = 0; i < 3; ++i) wc[i] = dt * C_om[i];
        quat_to_mat(qc, Rc);
        mat3_skew(Rc, wc, RcS);
#pragma unroll
        for (int i = 0; i < 9; ++i) Rcn[i] = Rc[i] + RcS[i];                        // equations.py:86
        mat_to_quat(Rcn, qc);                                                       // state.py:72
        notch[0] += dt * notch[1];                                                  // equations.py:80-82
        notch[1] += dt * notch[2];
        // what the Jacobian blocks of this predict are evaluated at: the dofs after it (frozen ones zeroed), the notch after it
        const T phi_n = (zeroed[2] ? T(0) : q3) - notch[0];
        if (!(phi_n == ph_arg)) { ph_arg = phi_n; t_sincos(phi_n, sp, cp); }
        if (live) {
            T* ak = aux + (int64_t)(AUX0 + AUXK * k) * B + b;
            ak[4 * B] = notch[0]; ak[5 * B] = notch[1]; ak[6 * B] = sp; ak[7 * B] = cp;
        }
#pragma unroll
        for (int i = 0; i < 3; ++i) { om_o[i] = om_n[i]; om_n[i] = om2[i]; }
        dt = dt2;
    }
    __syncthreads();                                    // B
    if (!live) return;
    if (a.do_update) {                                                              // Filter.py:369-375
        T qe[4];
        quat_mul(qm, qc, qe);
        quat_nrm(qe);
        T nv = t_sqrt(qe[0] * qe[0] + qe[1] * qe[1] + qe[2] * qe[2]);
        T flag = T(0);
        if (nv > T(1)) { nv = T(1); flag = T(1); }
        const T ang = residual_angle(nv, prm.quirks);             // half-angle residual (Quaternion.py:138-147) unless Q6 is off
        const T f = (nv == T(0)) ? T(0) : ang / nv;
        T r[7];
#pragma unroll
        for (int i = 0; i < 3; ++i) { r[i] = gpos[i] - sX[(6 + i) * 64 + lane]; r[3 + i] = f * qe[i]; }
        r[6] = ang_notch - notch[0];
        // a NaN / inf measurement or state propagates exactly as it does in the reference (NumPy arithmetic, no
        // check); the instance is flagged, its neighbours are not affected
        if (!t_finite(r[0] + r[1] + r[2] + r[3] + r[4] + r[5] + r[6])) flag = T(1);
        T* gr = reinterpret_cast<T*>(a.J) + b * JW + J_RES;      // (record of the launch's first predict)
#pragma unroll
        for (int i = 0; i < 7; ++i) gr[i] = r[i];
        gr[7] = flag;                                         // asin argument was clamped, or non-finite residual
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) xo[(16 + i) * B + b] = notch[i];
#pragma unroll
    for (int i = 0; i < 4; ++i) xo[(22 + i) * B + b] = qc[i];
    if (was_pending) a.status_ws[b] = pend_status & ~DVI_STATUS_PENDING;              // the injection is done (or dropped)
}

template <typename T>
__global__ __launch_bounds__(128, 4) void eskf_jac_kernel(StepArgs a, DevParams<T> prm) {
    // 64 (predict, instance) jobs per block, predict-major like the records J[n][B][JW]: wavefront 0 forms A, Theta,
    // Bm, Cm of its job, wavefront 1 Dm and Em; the block words of the 64 records are staged in LDS (odd row stride)
    // and written out whole, behind the residual words (which eskf_nominal_kernel writes for the first predict)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    T* sJ = reinterpret_cast<T*>(smem_raw);
    const int JS = prm.jfull ? JB_FULL : JB_CORE;
    const int lane = threadIdx.x & 63;
    const int role = threadIdx.x >> 6;
    const int64_t B = a.B;
    const int64_t cnt = a.sub_count ? a.sub_count : B;              // instances of this launch: [sub_first, sub_first + cnt)
    const int64_t njobs = (int64_t)a.n_pred * cnt;
    const int64_t job0 = (int64_t)blockIdx.x * 64;
    const int64_t job_raw = job0 + lane;
    const int64_t job = job_raw < njobs ? job_raw : njobs - 1;      // tail lanes recompute the last job, write nothing
    const int k = (int)(job / cnt);
    const int64_t b = a.sub_first + (job - (int64_t)k * cnt);
    const T* aux = reinterpret_cast<const T*>(a.aux);
    const T* ak = aux + (int64_t)(AUX0 + AUXK * k) * B + b;
    const T dt = reinterpret_cast<const T*>(a.dt)[a.dt_stride ? (int64_t)k * B + b : k];
    T sig_om[3] = {prm.sig_om[0], prm.sig_om[1], prm.sig_om[2]};
    if (a.noise) {
#pragma unroll
        for (int i = 0; i < 3; ++i) sig_om[i] = reinterpret_cast<const T*>(a.noise)[b * NOISE_WORDS + 20 + i];
    }
    // the buffers predict k is linearised with: the sample before it (the state's buffers for the first predict)
    T om_o[3], om_t[3];
    const int64_t so = a.imu_stride ? ((int64_t)(k - 1) * B + b) * 3 : (int64_t)(k - 1) * 3;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        om_o[i] = k == 0 ? aux[(0 + i) * B + b] : reinterpret_cast<const T*>(a.om)[so + i];
        om_t[i] = om_o[i] - sig_om[i];
    }
    ProbeTrig<T> t;
    t.s1 = aux[6 * B + b]; t.c1 = aux[7 * B + b]; t.s2 = aux[8 * B + b]; t.c2 = aux[9 * B + b];
    t.s3 = aux[10 * B + b]; t.c3 = aux[11 * B + b];
    t.sp = ak[6 * B]; t.cp = ak[7 * B];
    Jac<T>& J = *reinterpret_cast<Jac<T>*>(sJ + lane * JS);
    if (role == 0) {
        T acc_o[3], qo[4], dofs[6], Ro[9];
#pragma unroll
        for (int i = 0; i < 3; ++i) acc_o[i] = k == 0 ? aux[(3 + i) * B + b] : reinterpret_cast<const T*>(a.acc)[so + i];
#pragma unroll
        for (int i = 0; i < 4; ++i) qo[i] = ak[i * B];
#pragma unroll
        for (int i = 0; i < 6; ++i) dofs[i] = reinterpret_cast<const T*>(a.xs)[(10 + i) * B + b];   // (post-predict: stored by eskf_nominal_kernel)
        jac_blocks_old_buffers(qo, acc_o, om_o, dt, J, Ro);
        jac_blocks_position(prm, t, dofs, om_t, Ro, dt, J);
    } else {
        jac_blocks_attitude(prm, t, ak[5 * B], om_t, om_o, dt, J);
    }
    __syncthreads();
    // (the block's records are consecutive: predict-major jobs over the whole batch, or a part whose size is a multiple of 64)
    const int64_t k0 = job0 / cnt;
    T* gJ = reinterpret_cast<T*>(a.J) + (k0 * B + a.sub_first + (job0 - k0 * cnt)) * JW + J_B0;
    const int nrec = (njobs - job0 < 64) ? (int)(njobs - job0) : 64;
    const int first = role * 32, last = (first + 32 < nrec) ? first + 32 : nrec;
    if (prm.jfull) store_records<T, JB_FULL>(sJ, gJ, last, lane, first);
    else store_records<T, JB_CORE>(sJ, gJ, last, lane, first);
}

// ---- error injection: ErrorState.from_array (state.py:105-129) + State.__add__ (state.py:46-60)
template <typename T>
__device__ __forceinline__ void inject_error(State<T>& s, const T* e, const int quirks) {
    T th[3] = {e[6], e[7], e[8]}, thc[3] = {e[21], e[22], e[23]};
    T dq[4], dqc[4], qn[4], qcn[4];
    quat_about_axis(norm3(th), th, dq);
    quat_about_axis(norm3(thc), (quirks & QK_Q3) ? th : thc, dqc);   // sic (state.py:124: the axis is the IMU's delta-theta) unless Q3 is off
    quat_mul(s.q, dq, qn);
    quat_nrm(qn);
    quat_mul(s.qc, dqc, qcn);
    quat_nrm(qcn);
#pragma unroll
    for (int i = 0; i < 3; ++i) { s.p[i] += e[i]; s.v[i] += e[3 + i]; s.notch[i] += e[15 + i]; s.pc[i] += e[18 + i]; }
#pragma unroll
    for (int i = 0; i < 6; ++i) s.dofs[i] += e[9 + i];
#pragma unroll
    for (int i = 0; i < 4; ++i) { s.q[i] = qn[i]; s.qc[i] = qcn[i]; }
}

// ---- lane-per-instance: apply a pending injection to the stored state.  A step leaves the error
// state of its update in the workspace; the next step's pre kernel injects it on the fly, and this
// kernel is only launched when the state is read or replaced in between (dvi_eskf_flush).
template <typename T>
__global__ __launch_bounds__(256) void eskf_post_kernel(StepArgs a, DevParams<T> prm) {
    const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= a.B) return;
    const int st = a.status_ws[b];
    if (!(st & DVI_STATUS_PENDING)) return;                    // nothing due for this instance
    a.status_ws[b] = st & ~DVI_STATUS_PENDING;
    if (st & DVI_STATUS_SINGULAR) return;                      // Filter.py:356-361: nothing changes
    T* xs = reinterpret_cast<T*>(a.xs);
    State<T> s;
    load_state(xs, a.B, b, s);
    inject_error(s, reinterpret_cast<const T*>(a.e) + b * 24, prm.quirks);
    store_state(xs, a.B, b, s);
}

// ----------------------------------------------------------------------------------
// covariance kernel: LDS region of one instance (units of T)
// ----------------------------------------------------------------------------------
constexpr int PS = 25;             // row stride of P
constexpr int OFF_P = 0;           // 24 x 25
constexpr int ZS = 8;              // row stride of the Z panel (16-byte aligned rows)
constexpr int OFF_S = 600;         // scratch: U 6x24 [600,744) in predict | Z 24x8 [600,792) in update
constexpr int OFF_J = 744;         // Jacobian record of the current predict (85 words, predict only:
                                   //   the update's Z panel grows over it once it is dead)
constexpr int OFF_E = 792;         // 24 error-state words (update)
constexpr int REGION_MIN = 832;
// regions of the instances sharing a wave are staggered so that their windows fall on
// different banks: REGION = LPI (mod 32) (conflict-free for the 8-byte accesses below)
__host__ __device__ constexpr int region_words(int lpi) {
    int r = REGION_MIN;
    while ((r % 32) != (lpi % 32)) ++r;
    return r;
}

__device__ __forceinline__ void group_sync() { __syncthreads(); }   // block == one wavefront


// packed index -> (i, j), i >= j
__device__ __forceinline__ void unpack_ij(int e, int& i, int& j) {
    const int r = e / 25, c = e - 25 * r;
    if (c <= r) { i = r; j = c; } else { i = 23 - r; j = c - r - 1; }
}

// PIN: per-instance noise -- diag(Q), diag(R) are read from the instance's noise record instead of the
// parameter block (a population of differently tuned filters in one batch)
template <typename T, int LPI, bool PIN = false>
__global__ __launch_bounds__(64) void eskf_cov_kernel(StepArgs a, DevParams<T> prm) {
    constexpr int G = 64 / LPI;
    constexpr int REGION = region_words(LPI);
    __shared__ T smem[G * REGION];

    const int lane = threadIdx.x;
    const int grp = lane / LPI;
    const int lig = lane - grp * LPI;
    if (grp >= G) return;                   // lanes beyond the last whole instance slot idle

    T* sP = smem + grp * REGION + OFF_P;
    T* sS = smem + grp * REGION + OFF_S;
    T* sJ = smem + grp * REGION + OFF_J;
    T* sE = smem + grp * REGION + OFF_E;

    const int jwords = prm.q_gyro ? JB_FULL : JB_CORE;     // (block words: the LDS copy starts at J_B0)
    constexpr int NJ = (JB_FULL + LPI - 1) / LPI;     // Jacobian words per lane (upper bound)
    constexpr int NV = (150 + LPI - 1) / LPI;        // 16-byte (2-word) pieces of packed P per lane
    using T2 = typename Vec2<T>::type;

    // A wave works through a.rounds instances per slot, one after the other.  The packed
    // covariance and the first Jacobian record of round r+1 are fetched into registers while
    // round r computes, so only the very first HBM round trip of a wave is exposed.
    const int64_t round_stride = (int64_t)gridDim.x * G;
    const int64_t slot = (int64_t)blockIdx.x * G + grp;
    T2 pv[NV];
    T jv[NJ];
#define ESKF_ISSUE_LOADS(INST)                                                                     \
    {                                                                                              \
        const T2* gP2_ = reinterpret_cast<const T2*>(reinterpret_cast<const T*>(a.P_src) + (INST) * 300); \
        _Pragma("unroll") for (int k = 0; k < NV; ++k) {                                           \
            const int e2 = lig + k * LPI;                                                          \
            if (e2 < 150) pv[k] = gP2_[e2];                                                        \
        }                                                                                          \
        if (a.n_pred > 0) {                                                                        \
            const T* gJ_ = reinterpret_cast<const T*>(a.J) + (INST) * JW + J_B0;                          \
            _Pragma("unroll") for (int k = 0; k < NJ; ++k) {                                       \
                const int e = lig + k * LPI;                                                       \
                if (e < jwords) jv[k] = gJ_[e];                                                    \
            }                                                                                      \
        }                                                                                          \
    }
    {
        const int64_t i0 = slot < a.B ? slot : a.B - 1;
        ESKF_ISSUE_LOADS(i0)
    }

  for (int rnd = 0; rnd < a.rounds; ++rnd) {
    const int64_t inst_raw = (int64_t)rnd * round_stride + slot;
    const bool valid = inst_raw < a.B;      // tail: recompute the last instance, write nothing
    const int64_t inst = valid ? inst_raw : a.B - 1;
    T* gP = reinterpret_cast<T*>(a.P) + inst * 300;

    // ---- packed P (registers) -> full symmetric LDS image; Jacobian record -> LDS ------------
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const int e2 = lig + k * LPI;
        if (e2 < 150) {
            int i, j;
            unpack_ij(2 * e2, i, j);
            sP[i * PS + j] = pv[k].x;
            sP[j * PS + i] = pv[k].x;
            unpack_ij(2 * e2 + 1, i, j);
            sP[i * PS + j] = pv[k].y;
            sP[j * PS + i] = pv[k].y;
        }
    }
    if (a.n_pred > 0) {
#pragma unroll
        for (int k = 0; k < NJ; ++k) {
            const int e = lig + k * LPI;
            if (e < jwords) sJ[e] = jv[k];
        }
    }
    if (rnd + 1 < a.rounds) {               // prefetch the next round's instance
        const int64_t nraw = (int64_t)(rnd + 1) * round_stride + slot;
        const int64_t ni = nraw < a.B ? nraw : a.B - 1;
        ESKF_ISSUE_LOADS(ni)
    }

    // ================================ predict(s) =======================================
    for (int ip = 0; ip < a.n_pred; ++ip) {
        if (ip > 0) {   // (the record of the first predict was fetched together with P)
            const T* gJ = reinterpret_cast<const T*>(a.J) + ((int64_t)ip * a.B + inst) * JW + J_B0;
            T jv[NJ];
#pragma unroll
            for (int k = 0; k < NJ; ++k) {
                const int e = lig + k * LPI;
                if (e < jwords) jv[k] = gJ[e];
            }
#pragma unroll
            for (int k = 0; k < NJ; ++k) {
                const int e = lig + k * LPI;
                if (e < jwords) sJ[e] = jv[k];
            }
        }
        group_sync();
        const T dt = sJ[J_DT - J_B0];
        const T* A = sJ + J_A - J_B0;
        const T* Th = sJ + J_TH - J_B0;
        const T* Bm = sJ + J_BM - J_B0;
        const T* Cm = sJ + J_CM - J_B0;
        const T* Dm = sJ + J_DM - J_B0;
        const T* Em = sJ + J_EM - J_B0;

        // ---- phase 1: lane owns column j.  U[:,j] = Fx[18:24,:] P[:,j]; then the row
        // operations of Faa = Fx[0:18,0:18] on column j (rows 0:9, 15, 16), all from the
        // old column held in registers -> no intra-phase hazard.
        for (int j = lig; j < 24; j += LPI) {
            T c[24];
#pragma unroll
            for (int k = 0; k < 24; ++k) c[k] = sP[k * PS + j];
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                T u = dt * c[3 + r] + c[16 + r];
#pragma unroll
                for (int k = 0; k < 3; ++k) u += Bm[3 * r + k] * c[6 + k];
#pragma unroll
                for (int k = 0; k < 6; ++k) u += Cm[6 * r + k] * c[9 + k];
                sS[r * 24 + j] = u;
                T w = Dm[4 * r + 3] * c[15];
#pragma unroll
                for (int k = 0; k < 3; ++k) w += Dm[4 * r + k] * c[9 + k];
#pragma unroll
                for (int k = 0; k < 3; ++k) w += Em[3 * r + k] * c[19 + k];
                if (r == 1) w += c[22];
                if (r == 2) w += c[23];
                sS[(3 + r) * 24 + j] = w;
            }
            if (j < 18) {
#pragma unroll
                for (int r = 0; r < 3; ++r) {
                    sP[r * PS + j] = c[r] + dt * c[3 + r];
                    sP[(3 + r) * PS + j] = c[3 + r] + A[3 * r] * c[6] + A[3 * r + 1] * c[7] + A[3 * r + 2] * c[8];
                    sP[(6 + r) * PS + j] = Th[3 * r] * c[6] + Th[3 * r + 1] * c[7] + Th[3 * r + 2] * c[8];
                }
                sP[15 * PS + j] = c[15] + dt * c[16];
                sP[16 * PS + j] = c[16] + dt * c[17];
            }
        }
        group_sync();

        // ---- phase 2: lane owns a row.  Items 0..17: column operations of Faa^T on row i plus
        // the diagonal of Fi Q Fi^T, and two entries of P'cc = U Fx[18:24,:]^T (the 36 entries
        // are spread over these 18 lanes).  Items 18..23 (U row a): P'ca = U Faa^T, written to
        // row 18+a and mirrored into column 18+a.
        for (int it = lig; it < 24; it += LPI) {
            if (it < 18) {
                const int i = it;
                T c[9], c15, c16, c17;
#pragma unroll
                for (int k = 0; k < 9; ++k) c[k] = sP[i * PS + k];
                c15 = sP[i * PS + 15]; c16 = sP[i * PS + 16]; c17 = sP[i * PS + 17];
                T n[9];
#pragma unroll
                for (int r = 0; r < 3; ++r) {
                    n[r] = c[r] + dt * c[3 + r];
                    n[3 + r] = c[3 + r] + A[3 * r] * c[6] + A[3 * r + 1] * c[7] + A[3 * r + 2] * c[8];
                    n[6 + r] = Th[3 * r] * c[6] + Th[3 * r + 1] * c[7] + Th[3 * r + 2] * c[8];
                }
                const T n15 = c15 + dt * c16, n16 = c16 + dt * c17;
#pragma unroll
                for (int k = 0; k < 9; ++k) sP[i * PS + k] = n[k];
                sP[i * PS + 15] = n15; sP[i * PS + 16] = n16;
                // Fi Q Fi^T diagonal: rows 3..14 <- Q[i-3], row 17 <- Q[12]  (Filter.py:263-264,349)
                // (select chain: a runtime index into the by-value parameter block would spill it)
                {
                    T qi = T(0);
                    if (PIN) {
                        const T* gn = reinterpret_cast<const T*>(a.noise) + inst * NOISE_WORDS;
                        if (i >= 3 && i < 15) qi = gn[i - 3];
                        if (i == 17) qi = gn[12];
                    } else {
#pragma unroll
                        for (int k = 0; k < 12; ++k) qi = (i == 3 + k) ? prm.qd[k] : qi;
                        qi = (i == 17) ? prm.qd[12] : qi;
                    }
                    if (i >= 3) sP[i * PS + i] += qi;
                }
                // P'cc[a2][b2] and P'cc[a2][3 + b2]
                const int a2 = it / 3, b2 = it - 3 * a2;
                const T* u = sS + a2 * 24;
                T x = dt * u[3 + b2] + u[16 + b2];
#pragma unroll
                for (int k = 0; k < 3; ++k) x += Bm[3 * b2 + k] * u[6 + k];
#pragma unroll
                for (int k = 0; k < 6; ++k) x += Cm[6 * b2 + k] * u[9 + k];
                T y = Dm[4 * b2 + 3] * u[15];
#pragma unroll
                for (int k = 0; k < 3; ++k) y += Dm[4 * b2 + k] * u[9 + k];
#pragma unroll
                for (int k = 0; k < 3; ++k) y += Em[3 * b2 + k] * u[19 + k];
                if (b2 >= 1) y += u[21 + b2];   // the (22,22) and (23,23) ones of the identity survive
                if (prm.q_gyro) {   // camera block of Fi Q Fi^T (gyro noise enters rows 18:24 through N6)
                    const T* N6 = sJ + J_N6 - J_B0;
#pragma unroll
                    for (int k = 0; k < 3; ++k) {
                        const T nk = N6[3 * a2 + k] * (PIN ? reinterpret_cast<const T*>(a.noise)[inst * NOISE_WORDS + 3 + k] : prm.qd[3 + k]);
                        x += nk * N6[3 * b2 + k];
                        y += nk * N6[3 * (3 + b2) + k];
                    }
                }
                sP[(18 + a2) * PS + 18 + b2] = x;
                sP[(18 + a2) * PS + 21 + b2] = y;
            } else {
                const int ar = it - 18;
                T u[18];
#pragma unroll
                for (int k = 0; k < 18; ++k) u[k] = sS[ar * 24 + k];
                T row[18];
                // P'ca[ar][0:18] = U[ar][0:18] Faa^T
#pragma unroll
                for (int r = 0; r < 3; ++r) {
                    row[r] = u[r] + dt * u[3 + r];
                    row[3 + r] = u[3 + r] + A[3 * r] * u[6] + A[3 * r + 1] * u[7] + A[3 * r + 2] * u[8];
                    row[6 + r] = Th[3 * r] * u[6] + Th[3 * r + 1] * u[7] + Th[3 * r + 2] * u[8];
                }
#pragma unroll
                for (int k = 9; k < 15; ++k) row[k] = u[k];
                row[15] = u[15] + dt * u[16];
                row[16] = u[16] + dt * u[17];
                row[17] = u[17];
                if (prm.q_gyro) {   // Fi[6:9,3:6] = I couples the gyro noise into the camera rows
                    const T* N6 = sJ + J_N6 - J_B0;
#pragma unroll
                    for (int k = 0; k < 3; ++k) row[6 + k] += N6[3 * ar + k] * (PIN ? reinterpret_cast<const T*>(a.noise)[inst * NOISE_WORDS + 3 + k] : prm.qd[3 + k]);
                }
#pragma unroll
                for (int k = 0; k < 18; ++k) {
                    sP[(18 + ar) * PS + k] = row[k];
                    sP[k * PS + 18 + ar] = row[k];
                }
            }
        }
        group_sync();
    }

    // ================================ update ===========================================
    if (a.do_update) {
        // ---- S = H P H^T + R (H selects rows 18..23,15: Filter.py:83-85), LDL^T.  Every lane
        // of the instance factorises redundantly (7 pivots, ~60 fma) ---------------------
        constexpr int sel[7] = {18, 19, 20, 21, 22, 23, 15};
        T Lm[21], d[7], rd[7];   // unit-lower L packed by rows: L[i][j] at i(i-1)/2 + j
        bool ok = true;
        {
            T S[28];
#pragma unroll
            for (int i = 0; i < 7; ++i)
#pragma unroll
                for (int j = 0; j <= i; ++j) S[i * (i + 1) / 2 + j] = sP[sel[i] * PS + sel[j]] + (i == j ? (PIN ? reinterpret_cast<const T*>(a.noise)[inst * NOISE_WORDS + 13 + i] : prm.r[i]) : T(0));
#pragma unroll
            for (int j = 0; j < 7; ++j) {
                T dj = S[j * (j + 1) / 2 + j];
#pragma unroll
                for (int k = 0; k < j; ++k) dj -= Lm[j * (j - 1) / 2 + k] * Lm[j * (j - 1) / 2 + k] * d[k];
                d[j] = dj;
                ok = ok && (dj != T(0)) && t_finite(dj);
                rd[j] = T(1) / dj;
#pragma unroll
                for (int i = j + 1; i < 7; ++i) {
                    T v = S[i * (i + 1) / 2 + j];
#pragma unroll
                    for (int k = 0; k < j; ++k) v -= Lm[i * (i - 1) / 2 + k] * Lm[j * (j - 1) / 2 + k] * d[k];
                    Lm[i * (i - 1) / 2 + j] = v * rd[j];
                }
            }
        }
        const T* gres = reinterpret_cast<const T*>(a.J) + inst * JW + J_RES;
        T res[7];
#pragma unroll
        for (int m = 0; m < 7; ++m) res[m] = gres[m];
        int status = ok ? 0 : DVI_STATUS_SINGULAR;
        if (gres[7] != T(0)) status |= DVI_STATUS_NONFINITE;

        // ---- phase 3: lane owns row i: Z[i,:] L^T = P[i,sel]; K[i,:] = (Z D^-1) L^-1; e_i = K[i,:] res
        if (ok) {
            for (int i = lig; i < 24; i += LPI) {
                T z[7], k[7];
#pragma unroll
                for (int m = 0; m < 7; ++m) {
                    T v = sP[i * PS + sel[m]];
#pragma unroll
                    for (int n = 0; n < m; ++n) v -= z[n] * Lm[m * (m - 1) / 2 + n];
                    z[m] = v;
                }
#pragma unroll
                for (int m = 6; m >= 0; --m) {
                    T v = z[m] * rd[m];
#pragma unroll
                    for (int n = m + 1; n < 7; ++n) v -= k[n] * Lm[n * (n - 1) / 2 + m];
                    k[m] = v;
                }
                T e = T(0);
#pragma unroll
                for (int m = 0; m < 7; ++m) { e += k[m] * res[m]; sS[i * ZS + m] = z[m]; }
                if (i >= 9 && i < 15 && ((prm.frozen_mask >> (i - 9)) & 1)) e = T(0);   // Filter.py:377-379
                sE[i] = e;
                if (valid) reinterpret_cast<T*>(a.e)[inst * 24 + i] = e;
                if (a.K_out && valid) {
                    T* gK = reinterpret_cast<T*>(a.K_out) + (inst * 24 + i) * 7;
#pragma unroll
                    for (int m = 0; m < 7; ++m) gK[m] = k[m];
                }
            }
        }
        group_sync();
        if (ok) {
            // ---- phase 4: P -= Z D^-1 Z^T (Joseph form with the optimal gain, Filter.py:383-384),
            // lower triangle only: lane j updates the 13 (12 for j >= 12) elements (j+d mod 24, j),
            // d = 0..12 -- every unordered pair exactly once -- and writes both mirror positions.
            for (int j = lig; j < 24; j += LPI) {
                T zd[7];
#pragma unroll
                for (int m = 0; m < 7; ++m) zd[m] = sS[j * ZS + m] * rd[m];
                const int nd = (j < 12) ? 13 : 12;
#pragma unroll
                for (int dd = 0; dd < 13; ++dd) {
                    if (dd < nd) {
                        int i = j + dd;
                        if (i >= 24) i -= 24;
                        const T* zi = sS + i * ZS;
                        T v = sP[i * PS + j];
#pragma unroll
                        for (int m = 0; m < 7; ++m) v -= zi[m] * zd[m];
                        sP[i * PS + j] = v;
                        sP[j * PS + i] = v;
                    }
                }
            }
        }
        group_sync();
        // half error angles of the reset Jacobian G = I - [theta/2]x blocks (Filter.py:387-390)
        const T Gt[3] = {T(0.5) * sE[6], T(0.5) * sE[7], T(0.5) * sE[8]};
        const T Gc[3] = {T(0.5) * sE[21], T(0.5) * sE[22], T(0.5) * sE[23]};
        if (ok) {
            // ---- phase 5a: lane owns column j: rows 6:9 and 21:24 of G P
            for (int j = lig; j < 24; j += LPI) {
                const T a0 = sP[6 * PS + j], a1 = sP[7 * PS + j], a2 = sP[8 * PS + j];
                sP[6 * PS + j] = a0 + Gt[2] * a1 - Gt[1] * a2;
                sP[7 * PS + j] = a1 - Gt[2] * a0 + Gt[0] * a2;
                sP[8 * PS + j] = a2 + Gt[1] * a0 - Gt[0] * a1;
                const T b0 = sP[21 * PS + j], b1 = sP[22 * PS + j], b2 = sP[23 * PS + j];
                sP[21 * PS + j] = b0 + Gc[2] * b1 - Gc[1] * b2;
                sP[22 * PS + j] = b1 - Gc[2] * b0 + Gc[0] * b2;
                sP[23 * PS + j] = b2 + Gc[1] * b0 - Gc[0] * b1;
            }
        }
        group_sync();
        if (ok) {
            // ---- phase 5b: lane owns row i: columns 6:9 and 21:24 of (G P) G^T
            for (int i = lig; i < 24; i += LPI) {
                const T a0 = sP[i * PS + 6], a1 = sP[i * PS + 7], a2 = sP[i * PS + 8];
                sP[i * PS + 6] = a0 + Gt[2] * a1 - Gt[1] * a2;
                sP[i * PS + 7] = a1 - Gt[2] * a0 + Gt[0] * a2;
                sP[i * PS + 8] = a2 + Gt[1] * a0 - Gt[0] * a1;
                const T b0 = sP[i * PS + 21], b1 = sP[i * PS + 22], b2 = sP[i * PS + 23];
                sP[i * PS + 21] = b0 + Gc[2] * b1 - Gc[1] * b2;
                sP[i * PS + 22] = b1 - Gc[2] * b0 + Gc[0] * b2;
                sP[i * PS + 23] = b2 + Gc[1] * b0 - Gc[0] * b1;
            }
        }
        group_sync();
        if (lig == 0 && valid) {
            a.status_ws[inst] = status | DVI_STATUS_PENDING;   // the error state of this update awaits injection
            if (a.status) a.status[inst] = status;
        }
    }

    // ---- store: lower triangle -> packed record (16-byte stores) --------------------------
    if (valid) {
        T2* gP2 = reinterpret_cast<T2*>(gP);
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            const int e2 = lig + k * LPI;
            if (e2 < 150) {
                int i, j;
                T2 v;
                unpack_ij(2 * e2, i, j);
                v.x = sP[i * PS + j];
                unpack_ij(2 * e2 + 1, i, j);
                v.y = sP[i * PS + j];
                gP2[e2] = v;
            }
        }
    }
    group_sync();   // the next round overwrites the LDS image
  }
#undef ESKF_ISSUE_LOADS
}

// ----------------------------------------------------------------------------------
// row-resident covariance kernel (selector DVI_ESKF_LANES_ROWS): lane i of an instance keeps row i
// of P (= column i, P is symmetric) in registers for the whole step; 32 lanes per instance (24
// active), two instances per wave.  LDS is used for exchanges only:
//   load     packed record -> lower-triangle image, each lane gathers its row
//   predict  P' = Fx P Fx^T as two applications of v -> Fx v: lane i computes (Fx P)[:,i] and
//            writes it as row i of W = P Fx^T; lane j reads column j of W and applies Fx again,
//            which leaves row j of P' in its registers.  One 24x24 transposition per predict.
//   update   the 7 measurements are processed one after the other (R is diagonal, so this is the
//            joint update of Filter.py:351-385 with S = L D L^T factorised on the fly: the scalar
//            innovation variances are the pivots D): every lane publishes P[i][sel_m], reads the
//            column back and applies the rank-1 downdate to its row
//   reset    G P G^T: column part in registers; the six rows it touches are gathered from the
//            other lanes' column results by symmetry
// The Jacobian record is read from LDS with group-uniform (broadcast) addresses.
// ----------------------------------------------------------------------------------
// 1/x by v_rcp + two Newton steps (5 dependent operations instead of the 11 of the IEEE division
// sequence; the result is within an ulp or two, and only ever multiplies: pivots of S = L D L^T)
__device__ __forceinline__ double fast_rcp(double x) {
    double r = __builtin_amdgcn_rcp(x);
    r = fma(fma(-x, r, 1.0), r, r);
    r = fma(fma(-x, r, 1.0), r, r);
    return r;
}
__device__ __forceinline__ float fast_rcp(float x) {
    float r = __builtin_amdgcn_rcpf(x);
    r = fmaf(fmaf(-x, r, 1.0f), r, r);
    return r;
}

namespace rowk {
// exchange image: row stride PS = 25 (conflict-free row writes and column reads)
constexpr int XS = 9;              // row stride of the [24][6] panel of the reset (odd: conflict-free 8-byte accesses)
constexpr int R_BUF = 0;           // 24 x 25 exchange image | [24][XS] panel of the reset
constexpr int R_C = 600;           // two 24-word column buffers (alternating)
constexpr int R_J = 648;           // Jacobian record, blocks re-laid on even offsets (16-byte reads)
constexpr int L_A = 0, L_TH = 10, L_BM = 20, L_CM = 30, L_DM = 48, L_EM = 60, L_DT = 70, L_N6 = 72;
constexpr int R_RES = 744;         // residual of the update (7) + non-finite flag [+ the instance's noise record, 24 words]
// words per instance: even, and = 24 (mod 32) -- with 24 lanes per instance consecutive instances of a block
// then continue each other's bank pattern (row stride 25: lane t of the block hits bank 25 t mod 32)
constexpr int R_REGION = 760;
constexpr int R_REGION_PIN = 792;  // with per-instance noise
constexpr int L_DUMP = 71;         // record words that have no place in the LDS copy (the residual) land on this pad word
__device__ __forceinline__ int lds_word(int e) {   // HBM record word -> LDS word
    return e < J_B0 ? L_DUMP : (e - J_B0) + (e >= J_TH) + (e >= J_BM) + (e >= J_CM) + (e >= J_DT) + (e >= J_N6);
}

// o = Fx v  (Fx of Filter.py:270-342 with the structure of SURVEY.md App. A; same blocks as
// phases 1/2 of eskf_cov_kernel)
template <typename T>
__device__ __forceinline__ void fx_apply(const T* sJ, const T (&v)[24], T (&o)[24]) {
    const T dt = sJ[L_DT];
    {
        T A[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) A[k] = sJ[L_A + k];
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            o[r] = v[r] + dt * v[3 + r];
            o[3 + r] = v[3 + r] + A[3 * r] * v[6] + A[3 * r + 1] * v[7] + A[3 * r + 2] * v[8];
        }
    }
    {
        T Th[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) Th[k] = sJ[L_TH + k];
#pragma unroll
        for (int r = 0; r < 3; ++r) o[6 + r] = t_fma(Th[3 * r + 2], v[8], t_fma(Th[3 * r + 1], v[7], Th[3 * r] * v[6]));   // (explicit order: same bits as the DPP path)
    }
#pragma unroll
    for (int k = 9; k < 15; ++k) o[k] = v[k];
    o[15] = v[15] + dt * v[16];
    o[16] = v[16] + dt * v[17];
    o[17] = v[17];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        T Bm[3], Cm[6];
#pragma unroll
        for (int k = 0; k < 3; ++k) Bm[k] = sJ[L_BM + 3 * r + k];
#pragma unroll
        for (int k = 0; k < 6; ++k) Cm[k] = sJ[L_CM + 6 * r + k];
        T u = dt * v[3 + r] + v[16 + r];
#pragma unroll
        for (int k = 0; k < 3; ++k) u += Bm[k] * v[6 + k];
#pragma unroll
        for (int k = 0; k < 6; ++k) u += Cm[k] * v[9 + k];
        o[18 + r] = u;
    }
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        T Dm[4], Em[3];
#pragma unroll
        for (int k = 0; k < 4; ++k) Dm[k] = sJ[L_DM + 4 * r + k];
#pragma unroll
        for (int k = 0; k < 3; ++k) Em[k] = sJ[L_EM + 3 * r + k];
        T w = Dm[3] * v[15];
#pragma unroll
        for (int k = 0; k < 3; ++k) w += Dm[k] * v[9 + k];
#pragma unroll
        for (int k = 0; k < 3; ++k) w += Em[k] * v[19 + k];
        if (r == 1) w += v[22];
        if (r == 2) w += v[23];
        o[21 + r] = w;
    }
}
// one 8-byte LDS store that the compiler will not pair with its neighbour: ds_write2_b64 costs 13 LDS
// cycles, two ds_write_b64 cost 12 (MI355X_MICROARCH.md, LDS table)
template <typename T>
__device__ __forceinline__ void lds_store1(T* p, T v) {
    if (sizeof(T) == 8) {
        typedef volatile __attribute__((address_space(3))) T lds_t;
        *(lds_t*)p = v;
    } else {
        *p = v;     // (four-byte accesses: the paired forms are the cheaper ones)
    }
}

// ... and one 8-byte LDS load that will not be paired: ds_read2_b64 costs 8 LDS cycles, two ds_read_b64 cost 4
template <typename T>
__device__ __forceinline__ T lds_load1(const T* p) {
    if (sizeof(T) == 8) {
        typedef const volatile __attribute__((address_space(3))) T lds_t;
        return *(lds_t*)p;
    }
    return *p;
}

// dst[0:24] = Fx v (v in registers), written out piece by piece so that neither the outputs nor the
// Jacobian blocks of the later pieces pile up in registers
template <typename T>
__device__ __forceinline__ void fx_apply_store(const T* sJ, const T (&v)[24], T* dst, bool active) {
    const T dt = sJ[L_DT];
    {
        T A[9], Th[9], o[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) { A[k] = sJ[L_A + k]; Th[k] = sJ[L_TH + k]; }
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            // (fusions and order spelled out where the source form leaves the compiler a choice: the DPP
            //  path performs exactly these operations, and the two lane layouts must give the same bits)
            o[r] = t_fma(dt, v[3 + r], v[r]);
            o[3 + r] = v[3 + r] + A[3 * r] * v[6] + A[3 * r + 1] * v[7] + A[3 * r + 2] * v[8];
            o[6 + r] = t_fma(Th[3 * r + 2], v[8], t_fma(Th[3 * r + 1], v[7], Th[3 * r] * v[6]));
        }
        if (active) {
#pragma unroll
            for (int k = 0; k < 9; ++k) lds_store1(dst + k, o[k]);
#pragma unroll
            for (int k = 9; k < 15; ++k) lds_store1(dst + k, v[k]);
            lds_store1(dst + 15, v[15] + dt * v[16]);
            lds_store1(dst + 16, v[16] + dt * v[17]);
            lds_store1(dst + 17, v[17]);
        }
    }
    {
        T o[3];
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            T Bm[3], Cm[6];
#pragma unroll
            for (int k = 0; k < 3; ++k) Bm[k] = sJ[L_BM + 3 * r + k];
#pragma unroll
            for (int k = 0; k < 6; ++k) Cm[k] = sJ[L_CM + 6 * r + k];
            T u = t_fma(dt, v[3 + r], v[16 + r]);
#pragma unroll
            for (int k = 0; k < 3; ++k) u += Bm[k] * v[6 + k];
#pragma unroll
            for (int k = 0; k < 6; ++k) u += Cm[k] * v[9 + k];
            o[r] = u;
        }
        if (active) {
#pragma unroll
            for (int r = 0; r < 3; ++r) lds_store1(dst + 18 + r, o[r]);
        }
    }
    {
        T o[3];
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            T Dm[4], Em[3];
#pragma unroll
            for (int k = 0; k < 4; ++k) Dm[k] = sJ[L_DM + 4 * r + k];
#pragma unroll
            for (int k = 0; k < 3; ++k) Em[k] = sJ[L_EM + 3 * r + k];
            T w = Dm[3] * v[15];
#pragma unroll
            for (int k = 0; k < 3; ++k) w += Dm[k] * v[9 + k];
#pragma unroll
            for (int k = 0; k < 3; ++k) w += Em[k] * v[19 + k];
            if (r == 1) w += v[22];
            if (r == 2) w += v[23];
            o[r] = w;
        }
        if (active) {
#pragma unroll
            for (int r = 0; r < 3; ++r) lds_store1(dst + 21 + r, o[r]);
        }
    }
}

// o = Fx c for the column c[k] = col[k * PS] of the exchange image, read in three pieces so that
// at most nine inputs are live next to the outputs (the camera rows 18:24 are accumulated)
template <typename T>
__device__ __forceinline__ void fx_apply_col(const T* sJ, const T* col, T (&o)[24]) {
    const T dt = sJ[L_DT];
    T u[3], w[3];
    {   // inputs 0..8
        T c[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) c[k] = lds_load1(col + k * PS);
        T A[9], Th[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) { A[k] = sJ[L_A + k]; Th[k] = sJ[L_TH + k]; }
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            T x;
            {
#pragma clang fp contract(off)
                x = dt * c[3 + r];          // rounded once, used twice (not fused into o[r])
                o[r] = c[r] + x;
            }
            o[3 + r] = c[3 + r] + A[3 * r] * c[6] + A[3 * r + 1] * c[7] + A[3 * r + 2] * c[8];
            o[6 + r] = t_fma(Th[3 * r + 2], c[8], t_fma(Th[3 * r + 1], c[7], Th[3 * r] * c[6]));
            T Bm[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) Bm[k] = sJ[L_BM + 3 * r + k];
#pragma unroll
            for (int k = 0; k < 3; ++k) x += Bm[k] * c[6 + k];
            u[r] = x;
        }
    }
    {   // inputs 9..17
        T c[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) c[k] = lds_load1(col + (9 + k) * PS);
#pragma unroll
        for (int k = 0; k < 6; ++k) o[9 + k] = c[k];
        o[15] = c[6] + dt * c[7];
        o[16] = c[7] + dt * c[8];
        o[17] = c[8];
        u[0] += c[7];                       // c[16 + r], r = 0, 1 (r = 2 is input 18)
        u[1] += c[8];
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            T Cm[6], Dm[4];
#pragma unroll
            for (int k = 0; k < 6; ++k) Cm[k] = sJ[L_CM + 6 * r + k];
#pragma unroll
            for (int k = 0; k < 4; ++k) Dm[k] = sJ[L_DM + 4 * r + k];
#pragma unroll
            for (int k = 0; k < 6; ++k) u[r] += Cm[k] * c[k];
            T y = Dm[3] * c[6];
#pragma unroll
            for (int k = 0; k < 3; ++k) y += Dm[k] * c[k];
            w[r] = y;
        }
    }
    {   // inputs 18..23
        T c[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) c[k] = lds_load1(col + (18 + k) * PS);
        u[2] += c[0];
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            T Em[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) Em[k] = sJ[L_EM + 3 * r + k];
#pragma unroll
            for (int k = 0; k < 3; ++k) w[r] += Em[k] * c[1 + k];
        }
        w[1] += c[4];
        w[2] += c[5];
    }
#pragma unroll
    for (int r = 0; r < 3; ++r) { o[18 + r] = u[r]; o[21 + r] = w[r]; }
}

// P' += Fi Q Fi^T on the row-resident P: diagonal entries 3..14 <- Q[i-3], 17 <- Q[12]
// (Filter.py:263-264,349); with gyro noise (tune mode) also the camera rows through N6 and theta
// through Fi[6:9,3:6] = I
template <int J0, int J1, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (J0 < J1) {
        f(std::integral_constant<int, J0>{});
        static_for<J0 + 1, J1>(f);
    }
}
// p += q in the lanes that own row ROW of an instance in the 32-lanes-per-instance layout (lanes ROW and 32 + ROW): a
// constant of the layout, so the add runs under a constant EXEC mask -- one vector instruction per diagonal element
// instead of a compare, two selects and an add (the select chain below spends ~78 instructions per predict on
// adding one number to the lane's diagonal element)
template <int ROW>
__device__ __forceinline__ void add_in_row_lanes(double& p, double q) {
    constexpr unsigned long long mask = (1ull << ROW) | (1ull << (32 + ROW));
    unsigned long long saved;
    // (s_and_saveexec writes SCC: without the clobber a compare of the surrounding loop was scheduled in front of this
    //  statement and its branch behind it -- the long-horizon kernel left its predict loop after the first predict)
    asm("s_and_saveexec_b64 %1, %3\n\tv_add_f64 %0, %0, %2\n\ts_mov_b64 exec, %1" : "+v"(p), "=&s"(saved) : "s"(q), "s"(mask) : "scc");
}
template <int ROW>
__device__ __forceinline__ void add_in_row_lanes(float& p, float q) {
    constexpr unsigned long long mask = (1ull << ROW) | (1ull << (32 + ROW));
    unsigned long long saved;
    asm("s_and_saveexec_b64 %1, %3\n\tv_add_f32 %0, %0, %2\n\ts_mov_b64 exec, %1" : "+v"(p), "=&s"(saved) : "s"(q), "s"(mask) : "scc");
}
// P' += Fi Q Fi^T on the row-resident P (Filter.py:263-264,349): diagonal entries 3..14 <- Q[i - 3], 17 <- Q[12], each
// added in the two lanes that own the row -- the adds of rowk::noise_rows under the same constant lane sets, but with
// EXEC written directly (two scalar moves of an inline constant per entry and one restore at the end instead of a mask
// pair, a save and a restore per entry: 40 instructions instead of 65 -- at one covariance wavefront per SIMD every
// instruction of whatever kind is a slot of the wavefront's own time).  All lanes are active on entry.
__device__ __forceinline__ void noise_diag_exec(const DevParams<double>& prm, double (&p)[24]) {
#define EP_NQ(K, ROW) "s_mov_b32 exec_lo, %[m" #ROW "]\n s_mov_b32 exec_hi, %[m" #ROW "]\n v_add_f64 %[p" #ROW "], %[p" #ROW "], %[q" #K "]\n"
    asm volatile(EP_NQ(0, 3) EP_NQ(1, 4) EP_NQ(2, 5) EP_NQ(3, 6) EP_NQ(4, 7) EP_NQ(5, 8) EP_NQ(6, 9) EP_NQ(7, 10) EP_NQ(8, 11) EP_NQ(9, 12)
                 EP_NQ(10, 13) EP_NQ(11, 14) EP_NQ(12, 17) "s_mov_b64 exec, -1\n"
                 : [p3] "+v"(p[3]), [p4] "+v"(p[4]), [p5] "+v"(p[5]), [p6] "+v"(p[6]), [p7] "+v"(p[7]), [p8] "+v"(p[8]), [p9] "+v"(p[9]),
                   [p10] "+v"(p[10]), [p11] "+v"(p[11]), [p12] "+v"(p[12]), [p13] "+v"(p[13]), [p14] "+v"(p[14]), [p17] "+v"(p[17])
                 : [q0] "s"(prm.qd[0]), [q1] "s"(prm.qd[1]), [q2] "s"(prm.qd[2]), [q3] "s"(prm.qd[3]), [q4] "s"(prm.qd[4]), [q5] "s"(prm.qd[5]),
                   [q6] "s"(prm.qd[6]), [q7] "s"(prm.qd[7]), [q8] "s"(prm.qd[8]), [q9] "s"(prm.qd[9]), [q10] "s"(prm.qd[10]), [q11] "s"(prm.qd[11]),
                   [q12] "s"(prm.qd[12]),
                   [m3] "n"(1 << 3), [m4] "n"(1 << 4), [m5] "n"(1 << 5), [m6] "n"(1 << 6), [m7] "n"(1 << 7), [m8] "n"(1 << 8), [m9] "n"(1 << 9),
                   [m10] "n"(1 << 10), [m11] "n"(1 << 11), [m12] "n"(1 << 12), [m13] "n"(1 << 13), [m14] "n"(1 << 14), [m17] "n"(1 << 17));
#undef EP_NQ
}
__device__ __forceinline__ void noise_diag_exec(const DevParams<float>& prm, float (&p)[24]) {
#define EP_NQ(K, ROW) "s_mov_b32 exec_lo, %[m" #ROW "]\n s_mov_b32 exec_hi, %[m" #ROW "]\n v_add_f32 %[p" #ROW "], %[p" #ROW "], %[q" #K "]\n"
    asm volatile(EP_NQ(0, 3) EP_NQ(1, 4) EP_NQ(2, 5) EP_NQ(3, 6) EP_NQ(4, 7) EP_NQ(5, 8) EP_NQ(6, 9) EP_NQ(7, 10) EP_NQ(8, 11) EP_NQ(9, 12)
                 EP_NQ(10, 13) EP_NQ(11, 14) EP_NQ(12, 17) "s_mov_b64 exec, -1\n"
                 : [p3] "+v"(p[3]), [p4] "+v"(p[4]), [p5] "+v"(p[5]), [p6] "+v"(p[6]), [p7] "+v"(p[7]), [p8] "+v"(p[8]), [p9] "+v"(p[9]),
                   [p10] "+v"(p[10]), [p11] "+v"(p[11]), [p12] "+v"(p[12]), [p13] "+v"(p[13]), [p14] "+v"(p[14]), [p17] "+v"(p[17])
                 : [q0] "s"(prm.qd[0]), [q1] "s"(prm.qd[1]), [q2] "s"(prm.qd[2]), [q3] "s"(prm.qd[3]), [q4] "s"(prm.qd[4]), [q5] "s"(prm.qd[5]),
                   [q6] "s"(prm.qd[6]), [q7] "s"(prm.qd[7]), [q8] "s"(prm.qd[8]), [q9] "s"(prm.qd[9]), [q10] "s"(prm.qd[10]), [q11] "s"(prm.qd[11]),
                   [q12] "s"(prm.qd[12]),
                   [m3] "n"(1 << 3), [m4] "n"(1 << 4), [m5] "n"(1 << 5), [m6] "n"(1 << 6), [m7] "n"(1 << 7), [m8] "n"(1 << 8), [m9] "n"(1 << 9),
                   [m10] "n"(1 << 10), [m11] "n"(1 << 11), [m12] "n"(1 << 12), [m13] "n"(1 << 13), [m14] "n"(1 << 14), [m17] "n"(1 << 17));
#undef EP_NQ
}

template <typename T, bool PIN, bool GYRO = true, bool LANES32 = false>
__device__ __forceinline__ void noise_rows(const DevParams<T>& prm, const T* sJ, const T* sN, int row, T (&p)[24]) {
    if constexpr (!PIN && LANES32) {
        noise_diag_exec(prm, p);            // (every lane of the wavefront is active where the predict runs)
    } else if (PIN) {      // diag(Q) of this instance sits in LDS (sN[0:13]); each lane needs the entry of its own row
        T qrow = T(0);
        if (row >= 3 && row < 15) qrow = sN[row - 3];
        if (row == 17) qrow = sN[12];
#pragma unroll
        for (int k = 0; k < 12; ++k) p[3 + k] += (row == 3 + k) ? qrow : T(0);
        p[17] += (row == 17) ? qrow : T(0);
    } else {        // (one selected value per lane: thirteen hoisted selects would sit in registers through the predict)
        T qrow = T(0);
#pragma unroll
        for (int k = 0; k < 12; ++k) qrow = (row == 3 + k) ? prm.qd[k] : qrow;
        qrow = (row == 17) ? prm.qd[12] : qrow;
#pragma unroll
        for (int k = 0; k < 12; ++k) p[3 + k] += (row == 3 + k) ? qrow : T(0);
        p[17] += (row == 17) ? qrow : T(0);
    }
    if (GYRO && prm.q_gyro) {
        const T* N6 = sJ + L_N6;
        if (row >= 18) {
            const int ar = row - 18;
            T na[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                na[k] = N6[3 * ar + k] * (PIN ? sN[3 + k] : prm.qd[3 + k]);
                p[6 + k] += na[k];
            }
#pragma unroll
            for (int b = 0; b < 6; ++b)
#pragma unroll
                for (int k = 0; k < 3; ++k) p[18 + b] += na[k] * N6[3 * b + k];
        } else if (row >= 6 && row < 9) {
            const int k = row - 6;
            const T qk = PIN ? sN[3 + k] : ((k == 0) ? prm.qd[3] : (k == 1) ? prm.qd[4] : prm.qd[5]);
#pragma unroll
            for (int b = 0; b < 6; ++b) p[18 + b] += N6[3 * b + k] * qk;
        }
    }
}

// scheduling fence: every element of the row is final before anything after it is issued
template <typename T>
__device__ __forceinline__ void pin_row(T (&p)[24]) {
    asm volatile("" : "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3]), "+v"(p[4]), "+v"(p[5]), "+v"(p[6]), "+v"(p[7]),
                      "+v"(p[8]), "+v"(p[9]), "+v"(p[10]), "+v"(p[11]), "+v"(p[12]), "+v"(p[13]), "+v"(p[14]), "+v"(p[15]),
                      "+v"(p[16]), "+v"(p[17]), "+v"(p[18]), "+v"(p[19]), "+v"(p[20]), "+v"(p[21]), "+v"(p[22]), "+v"(p[23])
                 : : "memory");
}

// lanes whose value is zero, an infinity or a NaN -- v_cmp_class with the result in scalar registers
__device__ __forceinline__ unsigned long long unusable_pivot_mask(double v) {
    unsigned long long m;
    asm("v_cmp_class_f64_e64 %0, %1, %2" : "=s"(m) : "v"(v), "v"(0x267));     // sNaN qNaN -inf -0 +0 +inf
    return m;
}
__device__ __forceinline__ unsigned long long unusable_pivot_mask(float v) {
    unsigned long long m;
    asm("v_cmp_class_f32_e64 %0, %1, %2" : "=s"(m) : "v"(v), "v"(0x267));
    return m;
}

// The gather of a row from the packed-order LDS image (32 lanes per instance) without address arithmetic: for element j
// the lanes whose own row holds it (row >= j, bits j..31 of each half: rows, and the idle lanes that shadow row 23) read
// at row start + j, the others at start of row j + row -- two reads under constant execution masks, the element's
// constants in the offset fields, instead of one read behind an add and a select (-46 vector instructions per
// wavefront; the kernel pays for vector instructions, not for LDS reads or scalar moves: +1.4 %, six alternations).
// va = LDS byte address of the lane's packed row, vb = LDS byte address of word `row` of the image.  The reads are
// invisible to the compiler's counter logic: the block waits for them itself.  (Text generated: row h of the packed
// record starts at word 25 h for h < 12, else 26 (23 - h) + 1.)
__device__ __forceinline__ void gather_row_masked(unsigned va, unsigned vb, double (&p)[24]) {
    asm volatile(
        "ds_read_b64 %0, %24\n"
        "s_mov_b32 exec_lo, 0xfffffffe\n s_mov_b32 exec_hi, 0xfffffffe\n ds_read_b64 %1, %24 offset:8\n"
        "s_mov_b32 exec_lo, 0x00000001\n s_mov_b32 exec_hi, 0x00000001\n ds_read_b64 %1, %25 offset:200\n"
        "s_mov_b32 exec_lo, 0xfffffffc\n s_mov_b32 exec_hi, 0xfffffffc\n ds_read_b64 %2, %24 offset:16\n"
        "s_mov_b32 exec_lo, 0x00000003\n s_mov_b32 exec_hi, 0x00000003\n ds_read_b64 %2, %25 offset:400\n"
        "s_mov_b32 exec_lo, 0xfffffff8\n s_mov_b32 exec_hi, 0xfffffff8\n ds_read_b64 %3, %24 offset:24\n"
        "s_mov_b32 exec_lo, 0x00000007\n s_mov_b32 exec_hi, 0x00000007\n ds_read_b64 %3, %25 offset:600\n"
        "s_mov_b32 exec_lo, 0xfffffff0\n s_mov_b32 exec_hi, 0xfffffff0\n ds_read_b64 %4, %24 offset:32\n"
        "s_mov_b32 exec_lo, 0x0000000f\n s_mov_b32 exec_hi, 0x0000000f\n ds_read_b64 %4, %25 offset:800\n"
        "s_mov_b32 exec_lo, 0xffffffe0\n s_mov_b32 exec_hi, 0xffffffe0\n ds_read_b64 %5, %24 offset:40\n"
        "s_mov_b32 exec_lo, 0x0000001f\n s_mov_b32 exec_hi, 0x0000001f\n ds_read_b64 %5, %25 offset:1000\n"
        "s_mov_b32 exec_lo, 0xffffffc0\n s_mov_b32 exec_hi, 0xffffffc0\n ds_read_b64 %6, %24 offset:48\n"
        "s_mov_b32 exec_lo, 0x0000003f\n s_mov_b32 exec_hi, 0x0000003f\n ds_read_b64 %6, %25 offset:1200\n"
        "s_mov_b32 exec_lo, 0xffffff80\n s_mov_b32 exec_hi, 0xffffff80\n ds_read_b64 %7, %24 offset:56\n"
        "s_mov_b32 exec_lo, 0x0000007f\n s_mov_b32 exec_hi, 0x0000007f\n ds_read_b64 %7, %25 offset:1400\n"
        "s_mov_b32 exec_lo, 0xffffff00\n s_mov_b32 exec_hi, 0xffffff00\n ds_read_b64 %8, %24 offset:64\n"
        "s_mov_b32 exec_lo, 0x000000ff\n s_mov_b32 exec_hi, 0x000000ff\n ds_read_b64 %8, %25 offset:1600\n"
        "s_mov_b32 exec_lo, 0xfffffe00\n s_mov_b32 exec_hi, 0xfffffe00\n ds_read_b64 %9, %24 offset:72\n"
        "s_mov_b32 exec_lo, 0x000001ff\n s_mov_b32 exec_hi, 0x000001ff\n ds_read_b64 %9, %25 offset:1800\n"
        "s_mov_b32 exec_lo, 0xfffffc00\n s_mov_b32 exec_hi, 0xfffffc00\n ds_read_b64 %10, %24 offset:80\n"
        "s_mov_b32 exec_lo, 0x000003ff\n s_mov_b32 exec_hi, 0x000003ff\n ds_read_b64 %10, %25 offset:2000\n"
        "s_mov_b32 exec_lo, 0xfffff800\n s_mov_b32 exec_hi, 0xfffff800\n ds_read_b64 %11, %24 offset:88\n"
        "s_mov_b32 exec_lo, 0x000007ff\n s_mov_b32 exec_hi, 0x000007ff\n ds_read_b64 %11, %25 offset:2200\n"
        "s_mov_b32 exec_lo, 0xfffff000\n s_mov_b32 exec_hi, 0xfffff000\n ds_read_b64 %12, %24 offset:96\n"
        "s_mov_b32 exec_lo, 0x00000fff\n s_mov_b32 exec_hi, 0x00000fff\n ds_read_b64 %12, %25 offset:2296\n"
        "s_mov_b32 exec_lo, 0xffffe000\n s_mov_b32 exec_hi, 0xffffe000\n ds_read_b64 %13, %24 offset:104\n"
        "s_mov_b32 exec_lo, 0x00001fff\n s_mov_b32 exec_hi, 0x00001fff\n ds_read_b64 %13, %25 offset:2088\n"
        "s_mov_b32 exec_lo, 0xffffc000\n s_mov_b32 exec_hi, 0xffffc000\n ds_read_b64 %14, %24 offset:112\n"
        "s_mov_b32 exec_lo, 0x00003fff\n s_mov_b32 exec_hi, 0x00003fff\n ds_read_b64 %14, %25 offset:1880\n"
        "s_mov_b32 exec_lo, 0xffff8000\n s_mov_b32 exec_hi, 0xffff8000\n ds_read_b64 %15, %24 offset:120\n"
        "s_mov_b32 exec_lo, 0x00007fff\n s_mov_b32 exec_hi, 0x00007fff\n ds_read_b64 %15, %25 offset:1672\n"
        "s_mov_b32 exec_lo, 0xffff0000\n s_mov_b32 exec_hi, 0xffff0000\n ds_read_b64 %16, %24 offset:128\n"
        "s_mov_b32 exec_lo, 0x0000ffff\n s_mov_b32 exec_hi, 0x0000ffff\n ds_read_b64 %16, %25 offset:1464\n"
        "s_mov_b32 exec_lo, 0xfffe0000\n s_mov_b32 exec_hi, 0xfffe0000\n ds_read_b64 %17, %24 offset:136\n"
        "s_mov_b32 exec_lo, 0x0001ffff\n s_mov_b32 exec_hi, 0x0001ffff\n ds_read_b64 %17, %25 offset:1256\n"
        "s_mov_b32 exec_lo, 0xfffc0000\n s_mov_b32 exec_hi, 0xfffc0000\n ds_read_b64 %18, %24 offset:144\n"
        "s_mov_b32 exec_lo, 0x0003ffff\n s_mov_b32 exec_hi, 0x0003ffff\n ds_read_b64 %18, %25 offset:1048\n"
        "s_mov_b32 exec_lo, 0xfff80000\n s_mov_b32 exec_hi, 0xfff80000\n ds_read_b64 %19, %24 offset:152\n"
        "s_mov_b32 exec_lo, 0x0007ffff\n s_mov_b32 exec_hi, 0x0007ffff\n ds_read_b64 %19, %25 offset:840\n"
        "s_mov_b32 exec_lo, 0xfff00000\n s_mov_b32 exec_hi, 0xfff00000\n ds_read_b64 %20, %24 offset:160\n"
        "s_mov_b32 exec_lo, 0x000fffff\n s_mov_b32 exec_hi, 0x000fffff\n ds_read_b64 %20, %25 offset:632\n"
        "s_mov_b32 exec_lo, 0xffe00000\n s_mov_b32 exec_hi, 0xffe00000\n ds_read_b64 %21, %24 offset:168\n"
        "s_mov_b32 exec_lo, 0x001fffff\n s_mov_b32 exec_hi, 0x001fffff\n ds_read_b64 %21, %25 offset:424\n"
        "s_mov_b32 exec_lo, 0xffc00000\n s_mov_b32 exec_hi, 0xffc00000\n ds_read_b64 %22, %24 offset:176\n"
        "s_mov_b32 exec_lo, 0x003fffff\n s_mov_b32 exec_hi, 0x003fffff\n ds_read_b64 %22, %25 offset:216\n"
        "s_mov_b32 exec_lo, 0xff800000\n s_mov_b32 exec_hi, 0xff800000\n ds_read_b64 %23, %24 offset:184\n"
        "s_mov_b32 exec_lo, 0x007fffff\n s_mov_b32 exec_hi, 0x007fffff\n ds_read_b64 %23, %25 offset:8\n"
        "s_mov_b64 exec, -1\n s_waitcnt lgkmcnt(0)\n"
        : "=&v"(p[0]), "=&v"(p[1]), "=&v"(p[2]), "=&v"(p[3]), "=&v"(p[4]), "=&v"(p[5]), "=&v"(p[6]), "=&v"(p[7]),
          "=&v"(p[8]), "=&v"(p[9]), "=&v"(p[10]), "=&v"(p[11]), "=&v"(p[12]), "=&v"(p[13]), "=&v"(p[14]), "=&v"(p[15]),
          "=&v"(p[16]), "=&v"(p[17]), "=&v"(p[18]), "=&v"(p[19]), "=&v"(p[20]), "=&v"(p[21]), "=&v"(p[22]), "=&v"(p[23])
        : "v"(va), "v"(vb)
        : "memory");
}
__device__ __forceinline__ void gather_row_masked(unsigned va, unsigned vb, float (&p)[24]) {
    asm volatile(
        "ds_read_b32 %0, %24\n"
        "s_mov_b32 exec_lo, 0xfffffffe\n s_mov_b32 exec_hi, 0xfffffffe\n ds_read_b32 %1, %24 offset:4\n"
        "s_mov_b32 exec_lo, 0x00000001\n s_mov_b32 exec_hi, 0x00000001\n ds_read_b32 %1, %25 offset:100\n"
        "s_mov_b32 exec_lo, 0xfffffffc\n s_mov_b32 exec_hi, 0xfffffffc\n ds_read_b32 %2, %24 offset:8\n"
        "s_mov_b32 exec_lo, 0x00000003\n s_mov_b32 exec_hi, 0x00000003\n ds_read_b32 %2, %25 offset:200\n"
        "s_mov_b32 exec_lo, 0xfffffff8\n s_mov_b32 exec_hi, 0xfffffff8\n ds_read_b32 %3, %24 offset:12\n"
        "s_mov_b32 exec_lo, 0x00000007\n s_mov_b32 exec_hi, 0x00000007\n ds_read_b32 %3, %25 offset:300\n"
        "s_mov_b32 exec_lo, 0xfffffff0\n s_mov_b32 exec_hi, 0xfffffff0\n ds_read_b32 %4, %24 offset:16\n"
        "s_mov_b32 exec_lo, 0x0000000f\n s_mov_b32 exec_hi, 0x0000000f\n ds_read_b32 %4, %25 offset:400\n"
        "s_mov_b32 exec_lo, 0xffffffe0\n s_mov_b32 exec_hi, 0xffffffe0\n ds_read_b32 %5, %24 offset:20\n"
        "s_mov_b32 exec_lo, 0x0000001f\n s_mov_b32 exec_hi, 0x0000001f\n ds_read_b32 %5, %25 offset:500\n"
        "s_mov_b32 exec_lo, 0xffffffc0\n s_mov_b32 exec_hi, 0xffffffc0\n ds_read_b32 %6, %24 offset:24\n"
        "s_mov_b32 exec_lo, 0x0000003f\n s_mov_b32 exec_hi, 0x0000003f\n ds_read_b32 %6, %25 offset:600\n"
        "s_mov_b32 exec_lo, 0xffffff80\n s_mov_b32 exec_hi, 0xffffff80\n ds_read_b32 %7, %24 offset:28\n"
        "s_mov_b32 exec_lo, 0x0000007f\n s_mov_b32 exec_hi, 0x0000007f\n ds_read_b32 %7, %25 offset:700\n"
        "s_mov_b32 exec_lo, 0xffffff00\n s_mov_b32 exec_hi, 0xffffff00\n ds_read_b32 %8, %24 offset:32\n"
        "s_mov_b32 exec_lo, 0x000000ff\n s_mov_b32 exec_hi, 0x000000ff\n ds_read_b32 %8, %25 offset:800\n"
        "s_mov_b32 exec_lo, 0xfffffe00\n s_mov_b32 exec_hi, 0xfffffe00\n ds_read_b32 %9, %24 offset:36\n"
        "s_mov_b32 exec_lo, 0x000001ff\n s_mov_b32 exec_hi, 0x000001ff\n ds_read_b32 %9, %25 offset:900\n"
        "s_mov_b32 exec_lo, 0xfffffc00\n s_mov_b32 exec_hi, 0xfffffc00\n ds_read_b32 %10, %24 offset:40\n"
        "s_mov_b32 exec_lo, 0x000003ff\n s_mov_b32 exec_hi, 0x000003ff\n ds_read_b32 %10, %25 offset:1000\n"
        "s_mov_b32 exec_lo, 0xfffff800\n s_mov_b32 exec_hi, 0xfffff800\n ds_read_b32 %11, %24 offset:44\n"
        "s_mov_b32 exec_lo, 0x000007ff\n s_mov_b32 exec_hi, 0x000007ff\n ds_read_b32 %11, %25 offset:1100\n"
        "s_mov_b32 exec_lo, 0xfffff000\n s_mov_b32 exec_hi, 0xfffff000\n ds_read_b32 %12, %24 offset:48\n"
        "s_mov_b32 exec_lo, 0x00000fff\n s_mov_b32 exec_hi, 0x00000fff\n ds_read_b32 %12, %25 offset:1148\n"
        "s_mov_b32 exec_lo, 0xffffe000\n s_mov_b32 exec_hi, 0xffffe000\n ds_read_b32 %13, %24 offset:52\n"
        "s_mov_b32 exec_lo, 0x00001fff\n s_mov_b32 exec_hi, 0x00001fff\n ds_read_b32 %13, %25 offset:1044\n"
        "s_mov_b32 exec_lo, 0xffffc000\n s_mov_b32 exec_hi, 0xffffc000\n ds_read_b32 %14, %24 offset:56\n"
        "s_mov_b32 exec_lo, 0x00003fff\n s_mov_b32 exec_hi, 0x00003fff\n ds_read_b32 %14, %25 offset:940\n"
        "s_mov_b32 exec_lo, 0xffff8000\n s_mov_b32 exec_hi, 0xffff8000\n ds_read_b32 %15, %24 offset:60\n"
        "s_mov_b32 exec_lo, 0x00007fff\n s_mov_b32 exec_hi, 0x00007fff\n ds_read_b32 %15, %25 offset:836\n"
        "s_mov_b32 exec_lo, 0xffff0000\n s_mov_b32 exec_hi, 0xffff0000\n ds_read_b32 %16, %24 offset:64\n"
        "s_mov_b32 exec_lo, 0x0000ffff\n s_mov_b32 exec_hi, 0x0000ffff\n ds_read_b32 %16, %25 offset:732\n"
        "s_mov_b32 exec_lo, 0xfffe0000\n s_mov_b32 exec_hi, 0xfffe0000\n ds_read_b32 %17, %24 offset:68\n"
        "s_mov_b32 exec_lo, 0x0001ffff\n s_mov_b32 exec_hi, 0x0001ffff\n ds_read_b32 %17, %25 offset:628\n"
        "s_mov_b32 exec_lo, 0xfffc0000\n s_mov_b32 exec_hi, 0xfffc0000\n ds_read_b32 %18, %24 offset:72\n"
        "s_mov_b32 exec_lo, 0x0003ffff\n s_mov_b32 exec_hi, 0x0003ffff\n ds_read_b32 %18, %25 offset:524\n"
        "s_mov_b32 exec_lo, 0xfff80000\n s_mov_b32 exec_hi, 0xfff80000\n ds_read_b32 %19, %24 offset:76\n"
        "s_mov_b32 exec_lo, 0x0007ffff\n s_mov_b32 exec_hi, 0x0007ffff\n ds_read_b32 %19, %25 offset:420\n"
        "s_mov_b32 exec_lo, 0xfff00000\n s_mov_b32 exec_hi, 0xfff00000\n ds_read_b32 %20, %24 offset:80\n"
        "s_mov_b32 exec_lo, 0x000fffff\n s_mov_b32 exec_hi, 0x000fffff\n ds_read_b32 %20, %25 offset:316\n"
        "s_mov_b32 exec_lo, 0xffe00000\n s_mov_b32 exec_hi, 0xffe00000\n ds_read_b32 %21, %24 offset:84\n"
        "s_mov_b32 exec_lo, 0x001fffff\n s_mov_b32 exec_hi, 0x001fffff\n ds_read_b32 %21, %25 offset:212\n"
        "s_mov_b32 exec_lo, 0xffc00000\n s_mov_b32 exec_hi, 0xffc00000\n ds_read_b32 %22, %24 offset:88\n"
        "s_mov_b32 exec_lo, 0x003fffff\n s_mov_b32 exec_hi, 0x003fffff\n ds_read_b32 %22, %25 offset:108\n"
        "s_mov_b32 exec_lo, 0xff800000\n s_mov_b32 exec_hi, 0xff800000\n ds_read_b32 %23, %24 offset:92\n"
        "s_mov_b32 exec_lo, 0x007fffff\n s_mov_b32 exec_hi, 0x007fffff\n ds_read_b32 %23, %25 offset:4\n"
        "s_mov_b64 exec, -1\n s_waitcnt lgkmcnt(0)\n"
        : "=&v"(p[0]), "=&v"(p[1]), "=&v"(p[2]), "=&v"(p[3]), "=&v"(p[4]), "=&v"(p[5]), "=&v"(p[6]), "=&v"(p[7]),
          "=&v"(p[8]), "=&v"(p[9]), "=&v"(p[10]), "=&v"(p[11]), "=&v"(p[12]), "=&v"(p[13]), "=&v"(p[14]), "=&v"(p[15]),
          "=&v"(p[16]), "=&v"(p[17]), "=&v"(p[18]), "=&v"(p[19]), "=&v"(p[20]), "=&v"(p[21]), "=&v"(p[22]), "=&v"(p[23])
        : "v"(va), "v"(vb)
        : "memory");
}

// ---- cross-lane broadcast without LDS (32 lanes per instance; fp64 shown, fp32 below) --------------
// An instance occupies two 16-lane DPP rows: rows 0..15 of P in the first, rows 16..23 (+ 8 shadow
// lanes) in the second.  `row_newbcast:n` feeds lane n of the reader's own 16-lane row to every lane
// of that row as a source operand of a DP-rate instruction at no extra cost, so a value that every lane
// of the instance holds for its own row -- a column of P -- reaches all rows as `v_fmac_f64_dpp`
// operands once both 16-lane rows carry it: spread_rows() makes X = (first row's values in both rows)
// and Y = (second row's values in both rows) with one v_permlane16_swap per dword.
// (Checked on gfx950 with tools/dpp_micro.hip and tools/dpp_semantics.hip: row_newbcast semantics, the
//  swap's row pairing -- and that it needs its wait states --, v_fmac_f64_dpp issuing at the rate of
//  v_fmac_f64.)
__device__ __forceinline__ void spread_rows(double v, double& X, double& Y) {
    // (two 64-bit copies of v made first and consumed in place by the swaps: v_mov_b64 x 2 instead of the four
    //  v_mov_b32 the compiler emits when it has to duplicate the halves of a value that stays live)
    double c0 = v, c1 = v;
    asm("" : "+v"(c0), "+v"(c1));
    const auto a = __builtin_amdgcn_permlane16_swap(__double2loint(c0), __double2loint(c1), false, false);
    const auto b = __builtin_amdgcn_permlane16_swap(__double2hiint(c0), __double2hiint(c1), false, false);
    X = __hiloint2double(b[0], a[0]);
    Y = __hiloint2double(b[1], a[1]);
    // A DPP read needs 2 wait states after a VALU write of its source, and inline assembly is opaque to
    // the compiler's hazard recognizer: every DPP user depends on the outputs of this statement, so the
    // wait states sit between the swap and all of them wherever the scheduler puts those
    // (tools/check_dpp_hazards.py verifies the generated code).
    asm("s_nop 1" : "+v"(X), "+v"(Y));
}
// acc += (lane LANE of the reader's 16-lane row of src) * mul.  Not volatile: a pure function of its
// operands, free to be interleaved with independent chains.
template <int LANE, bool FIRST = false>
__device__ __forceinline__ void dpp_fmac(double& acc, double src, double mul) {
    if constexpr (FIRST) acc = 0.0;     // (the accumulator's first term: whatever it held is not read)
    asm("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(src), "v"(mul), "n"(LANE));
}
template <int LANE>
__device__ __forceinline__ double dpp_bcast(double src) {
    double r;
    asm("v_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(src), "n"(LANE));
    return r;
}
// fp32: the same three primitives on one dword
__device__ __forceinline__ void spread_rows(float v, float& X, float& Y) {
    const int w = __float_as_int(v);
    const auto a = __builtin_amdgcn_permlane16_swap(w, w, false, false);
    X = __int_as_float(a[0]);
    Y = __int_as_float(a[1]);
    asm("s_nop 1" : "+v"(X), "+v"(Y));
}
template <int LANE, bool FIRST = false>
__device__ __forceinline__ void dpp_fmac(float& acc, float src, float mul) {
    // (the accumulator's first term is the product itself: one instruction instead of a zero and a multiply-add.  fp32 only --
    //  v_mul_f64 has no DPP form; differs from fma(a, b, +0) in the sign of a zero product, nowhere else)
    if constexpr (FIRST) asm("v_mul_f32_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "=v"(acc) : "v"(src), "v"(mul), "n"(LANE));
    else asm("v_fmac_f32_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(src), "v"(mul), "n"(LANE));
}
template <int LANE>
__device__ __forceinline__ float dpp_bcast(float src) {
    float r;
    asm("v_mov_b32_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(src), "n"(LANE));
    return r;
}
// float overloads: never instantiated on a taken path (the fp32 kernels keep the LDS exchange)
// The Jacobian record of an instance spread over its 32 lanes (lane l holds words l, 32+l, 64+l -- the
// registers the record is fetched into) and made reachable as DPP operands: X[k] / Y[k] carry the k-th
// word of the lanes of the first / second 16-lane row in both rows.
template <typename T>
struct JRegs {
    T X[3], Y[3];
    // acc += J[W] * v
    template <int W, bool FIRST = false>
    __device__ __forceinline__ void mac(T& acc, T v) const {
        constexpr int k = W >> 5, l = W & 31;
        dpp_fmac<(l & 15), FIRST>(acc, l < 16 ? X[k] : Y[k], v);
    }
};
// dst[0:24] = Fx v and o = Fx c: the operations of fx_apply_store / fx_apply_col in the same order (the two
// lane layouts give the same bits), with the Jacobian entries as DPP operands instead of LDS reads
// Q1 (symbols.py:103,107 / Filter.py:279-285): the v1 source places the columns of the camera rows' Jacobian
// after err_q one position per VARIABLE instead of one per error-state column -- delta-p_C's identity lands in columns 16:19
// (notch rate, notch acceleration, p_C x), Em in 19:22, and the ones of I_24 at (22,22), (23,23) survive.  Q1 = false: the
// columns where those variables live -- identity at 18:21, Em at 21:24, nothing from columns 16:18.
template <typename T, bool Q1 = true>
__device__ __forceinline__ void fx_apply_store_dpp(const JRegs<T>& J, const T (&v)[24], T* dst, bool active) {
    {
        T o[9];
        static_for<0, 3>([&](auto R) {
            constexpr int r = decltype(R)::value;
            T a = v[r];
            J.template mac<J_DT>(a, v[3 + r]);
            o[r] = a;
            T b = v[3 + r];
            J.template mac<J_A + 3 * r>(b, v[6]); J.template mac<J_A + 3 * r + 1>(b, v[7]); J.template mac<J_A + 3 * r + 2>(b, v[8]);
            o[3 + r] = b;
            T c;
            J.template mac<J_TH + 3 * r, true>(c, v[6]); J.template mac<J_TH + 3 * r + 1>(c, v[7]); J.template mac<J_TH + 3 * r + 2>(c, v[8]);
            o[6 + r] = c;
        });
        T o15 = v[15], o16 = v[16];
        J.template mac<J_DT>(o15, v[16]);
        J.template mac<J_DT>(o16, v[17]);
        {   // (every lane stores: the idle lanes shadow row 23 -- the same words, the same values)
#pragma unroll
            for (int k = 0; k < 9; ++k) lds_store1(dst + k, o[k]);
#pragma unroll
            for (int k = 9; k < 15; ++k) lds_store1(dst + k, v[k]);
            lds_store1(dst + 15, o15);
            lds_store1(dst + 16, o16);
            lds_store1(dst + 17, v[17]);
        }
    }
    {
        T o[3];
        static_for<0, 3>([&](auto R) {
            constexpr int r = decltype(R)::value;
            T u = v[(Q1 ? 16 : 18) + r];
            J.template mac<J_DT>(u, v[3 + r]);
            J.template mac<J_BM + 3 * r>(u, v[6]); J.template mac<J_BM + 3 * r + 1>(u, v[7]); J.template mac<J_BM + 3 * r + 2>(u, v[8]);
            static_for<0, 6>([&](auto K) { J.template mac<J_CM + 6 * r + decltype(K)::value>(u, v[9 + decltype(K)::value]); });
            o[r] = u;
        });
        {   // (every lane stores: the idle lanes shadow row 23 -- the same words, the same values)
#pragma unroll
            for (int r = 0; r < 3; ++r) lds_store1(dst + 18 + r, o[r]);
        }
    }
    {
        T o[3];
        static_for<0, 3>([&](auto R) {
            constexpr int r = decltype(R)::value;
            T w;
            J.template mac<J_DM + 4 * r + 3, true>(w, v[15]);
            J.template mac<J_DM + 4 * r>(w, v[9]); J.template mac<J_DM + 4 * r + 1>(w, v[10]); J.template mac<J_DM + 4 * r + 2>(w, v[11]);
            constexpr int e0 = Q1 ? 19 : 21;
            J.template mac<J_EM + 3 * r>(w, v[e0]); J.template mac<J_EM + 3 * r + 1>(w, v[e0 + 1]); J.template mac<J_EM + 3 * r + 2>(w, v[e0 + 2]);
            if (Q1 && r == 1) w += v[22];
            if (Q1 && r == 2) w += v[23];
            o[r] = w;
        });
        {   // (every lane stores: the idle lanes shadow row 23 -- the same words, the same values)
#pragma unroll
            for (int r = 0; r < 3; ++r) lds_store1(dst + 21 + r, o[r]);
        }
    }
}
template <typename T, bool Q1 = true>
__device__ __forceinline__ void fx_apply_col_dpp(const JRegs<T>& J, const T* col, T (&o)[24]) {
    T u[3], w[3];
    {   // inputs 0..8
        T c[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) c[k] = lds_load1(col + k * PS);
        static_for<0, 3>([&](auto R) {
            constexpr int r = decltype(R)::value;
            T x;                        // dt c[3+r], used twice (see fx_apply_store_dpp)
            J.template mac<J_DT, true>(x, c[3 + r]);
            o[r] = c[r] + x;
            T b = c[3 + r];
            J.template mac<J_A + 3 * r>(b, c[6]); J.template mac<J_A + 3 * r + 1>(b, c[7]); J.template mac<J_A + 3 * r + 2>(b, c[8]);
            o[3 + r] = b;
            T t;
            J.template mac<J_TH + 3 * r, true>(t, c[6]); J.template mac<J_TH + 3 * r + 1>(t, c[7]); J.template mac<J_TH + 3 * r + 2>(t, c[8]);
            o[6 + r] = t;
            J.template mac<J_BM + 3 * r>(x, c[6]); J.template mac<J_BM + 3 * r + 1>(x, c[7]); J.template mac<J_BM + 3 * r + 2>(x, c[8]);
            u[r] = x;
        });
    }
    {   // inputs 9..17
        T c[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) c[k] = lds_load1(col + (9 + k) * PS);
#pragma unroll
        for (int k = 0; k < 6; ++k) o[9 + k] = c[k];
        T o15 = c[6], o16 = c[7];
        J.template mac<J_DT>(o15, c[7]);
        J.template mac<J_DT>(o16, c[8]);
        o[15] = o15;
        o[16] = o16;
        o[17] = c[8];
        if constexpr (Q1) {
            u[0] += c[7];                   // c[16 + r], r = 0, 1 (r = 2 is input 18)
            u[1] += c[8];
        }
        static_for<0, 3>([&](auto R) {
            constexpr int r = decltype(R)::value;
            static_for<0, 6>([&](auto K) { J.template mac<J_CM + 6 * r + decltype(K)::value>(u[r], c[decltype(K)::value]); });
            T y;
            J.template mac<J_DM + 4 * r + 3, true>(y, c[6]);
            J.template mac<J_DM + 4 * r>(y, c[0]); J.template mac<J_DM + 4 * r + 1>(y, c[1]); J.template mac<J_DM + 4 * r + 2>(y, c[2]);
            w[r] = y;
        });
    }
    {   // inputs 18..23
        T c[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) c[k] = lds_load1(col + (18 + k) * PS);
        if constexpr (Q1) {
            u[2] += c[0];
            static_for<0, 3>([&](auto R) {
                constexpr int r = decltype(R)::value;
                J.template mac<J_EM + 3 * r>(w[r], c[1]); J.template mac<J_EM + 3 * r + 1>(w[r], c[2]); J.template mac<J_EM + 3 * r + 2>(w[r], c[3]);
            });
            w[1] += c[4];
            w[2] += c[5];
        } else {
            u[0] += c[0]; u[1] += c[1]; u[2] += c[2];
            static_for<0, 3>([&](auto R) {
                constexpr int r = decltype(R)::value;
                J.template mac<J_EM + 3 * r>(w[r], c[3]); J.template mac<J_EM + 3 * r + 1>(w[r], c[4]); J.template mac<J_EM + 3 * r + 2>(w[r], c[5]);
            });
        }
    }
#pragma unroll
    for (int r = 0; r < 3; ++r) { o[18 + r] = u[r]; o[21 + r] = w[r]; }
}
// The store of a row into the packed-order LDS image (32 lanes per instance): element j exists in rows >= j only -- lanes
// j..23 of each half, a constant per j -- so each element is one write under a constant execution mask written directly (two
// scalar moves), with the element's offset in the instruction: the twin of gather_row_masked.  (As a predicated C++ store
// each element cost a mask pair, a save and a restore of EXEC: 96 scalar instructions per row.)  va = LDS byte address of the
// lane's packed row.  All lanes are active on entry; the writes are invisible to the compiler's counter logic: the block
// waits for them itself.  (Text generated.)
__device__ __forceinline__ void store_row_masked(unsigned va, const double (&p)[24]) {
    asm volatile(
        "s_mov_b32 exec_lo, 0x00ffffff\n s_mov_b32 exec_hi, 0x00ffffff\n ds_write_b64 %24, %0 offset:0\n"
        "s_mov_b32 exec_lo, 0x00fffffe\n s_mov_b32 exec_hi, 0x00fffffe\n ds_write_b64 %24, %1 offset:8\n"
        "s_mov_b32 exec_lo, 0x00fffffc\n s_mov_b32 exec_hi, 0x00fffffc\n ds_write_b64 %24, %2 offset:16\n"
        "s_mov_b32 exec_lo, 0x00fffff8\n s_mov_b32 exec_hi, 0x00fffff8\n ds_write_b64 %24, %3 offset:24\n"
        "s_mov_b32 exec_lo, 0x00fffff0\n s_mov_b32 exec_hi, 0x00fffff0\n ds_write_b64 %24, %4 offset:32\n"
        "s_mov_b32 exec_lo, 0x00ffffe0\n s_mov_b32 exec_hi, 0x00ffffe0\n ds_write_b64 %24, %5 offset:40\n"
        "s_mov_b32 exec_lo, 0x00ffffc0\n s_mov_b32 exec_hi, 0x00ffffc0\n ds_write_b64 %24, %6 offset:48\n"
        "s_mov_b32 exec_lo, 0x00ffff80\n s_mov_b32 exec_hi, 0x00ffff80\n ds_write_b64 %24, %7 offset:56\n"
        "s_mov_b32 exec_lo, 0x00ffff00\n s_mov_b32 exec_hi, 0x00ffff00\n ds_write_b64 %24, %8 offset:64\n"
        "s_mov_b32 exec_lo, 0x00fffe00\n s_mov_b32 exec_hi, 0x00fffe00\n ds_write_b64 %24, %9 offset:72\n"
        "s_mov_b32 exec_lo, 0x00fffc00\n s_mov_b32 exec_hi, 0x00fffc00\n ds_write_b64 %24, %10 offset:80\n"
        "s_mov_b32 exec_lo, 0x00fff800\n s_mov_b32 exec_hi, 0x00fff800\n ds_write_b64 %24, %11 offset:88\n"
        "s_mov_b32 exec_lo, 0x00fff000\n s_mov_b32 exec_hi, 0x00fff000\n ds_write_b64 %24, %12 offset:96\n"
        "s_mov_b32 exec_lo, 0x00ffe000\n s_mov_b32 exec_hi, 0x00ffe000\n ds_write_b64 %24, %13 offset:104\n"
        "s_mov_b32 exec_lo, 0x00ffc000\n s_mov_b32 exec_hi, 0x00ffc000\n ds_write_b64 %24, %14 offset:112\n"
        "s_mov_b32 exec_lo, 0x00ff8000\n s_mov_b32 exec_hi, 0x00ff8000\n ds_write_b64 %24, %15 offset:120\n"
        "s_mov_b32 exec_lo, 0x00ff0000\n s_mov_b32 exec_hi, 0x00ff0000\n ds_write_b64 %24, %16 offset:128\n"
        "s_mov_b32 exec_lo, 0x00fe0000\n s_mov_b32 exec_hi, 0x00fe0000\n ds_write_b64 %24, %17 offset:136\n"
        "s_mov_b32 exec_lo, 0x00fc0000\n s_mov_b32 exec_hi, 0x00fc0000\n ds_write_b64 %24, %18 offset:144\n"
        "s_mov_b32 exec_lo, 0x00f80000\n s_mov_b32 exec_hi, 0x00f80000\n ds_write_b64 %24, %19 offset:152\n"
        "s_mov_b32 exec_lo, 0x00f00000\n s_mov_b32 exec_hi, 0x00f00000\n ds_write_b64 %24, %20 offset:160\n"
        "s_mov_b32 exec_lo, 0x00e00000\n s_mov_b32 exec_hi, 0x00e00000\n ds_write_b64 %24, %21 offset:168\n"
        "s_mov_b32 exec_lo, 0x00c00000\n s_mov_b32 exec_hi, 0x00c00000\n ds_write_b64 %24, %22 offset:176\n"
        "s_mov_b32 exec_lo, 0x00800000\n s_mov_b32 exec_hi, 0x00800000\n ds_write_b64 %24, %23 offset:184\n"
        "s_mov_b64 exec, -1\n s_waitcnt lgkmcnt(0)\n"
        :
        : "v"(p[0]), "v"(p[1]), "v"(p[2]), "v"(p[3]), "v"(p[4]), "v"(p[5]), "v"(p[6]), "v"(p[7]), "v"(p[8]), "v"(p[9]), "v"(p[10]), "v"(p[11]), "v"(p[12]), "v"(p[13]), "v"(p[14]), "v"(p[15]), "v"(p[16]), "v"(p[17]), "v"(p[18]), "v"(p[19]), "v"(p[20]), "v"(p[21]), "v"(p[22]), "v"(p[23]), "v"(va)
        : "memory");
}
__device__ __forceinline__ void store_row_masked(unsigned va, const float (&p)[24]) {
    asm volatile(
        "s_mov_b32 exec_lo, 0x00ffffff\n s_mov_b32 exec_hi, 0x00ffffff\n ds_write_b32 %24, %0 offset:0\n"
        "s_mov_b32 exec_lo, 0x00fffffe\n s_mov_b32 exec_hi, 0x00fffffe\n ds_write_b32 %24, %1 offset:4\n"
        "s_mov_b32 exec_lo, 0x00fffffc\n s_mov_b32 exec_hi, 0x00fffffc\n ds_write_b32 %24, %2 offset:8\n"
        "s_mov_b32 exec_lo, 0x00fffff8\n s_mov_b32 exec_hi, 0x00fffff8\n ds_write_b32 %24, %3 offset:12\n"
        "s_mov_b32 exec_lo, 0x00fffff0\n s_mov_b32 exec_hi, 0x00fffff0\n ds_write_b32 %24, %4 offset:16\n"
        "s_mov_b32 exec_lo, 0x00ffffe0\n s_mov_b32 exec_hi, 0x00ffffe0\n ds_write_b32 %24, %5 offset:20\n"
        "s_mov_b32 exec_lo, 0x00ffffc0\n s_mov_b32 exec_hi, 0x00ffffc0\n ds_write_b32 %24, %6 offset:24\n"
        "s_mov_b32 exec_lo, 0x00ffff80\n s_mov_b32 exec_hi, 0x00ffff80\n ds_write_b32 %24, %7 offset:28\n"
        "s_mov_b32 exec_lo, 0x00ffff00\n s_mov_b32 exec_hi, 0x00ffff00\n ds_write_b32 %24, %8 offset:32\n"
        "s_mov_b32 exec_lo, 0x00fffe00\n s_mov_b32 exec_hi, 0x00fffe00\n ds_write_b32 %24, %9 offset:36\n"
        "s_mov_b32 exec_lo, 0x00fffc00\n s_mov_b32 exec_hi, 0x00fffc00\n ds_write_b32 %24, %10 offset:40\n"
        "s_mov_b32 exec_lo, 0x00fff800\n s_mov_b32 exec_hi, 0x00fff800\n ds_write_b32 %24, %11 offset:44\n"
        "s_mov_b32 exec_lo, 0x00fff000\n s_mov_b32 exec_hi, 0x00fff000\n ds_write_b32 %24, %12 offset:48\n"
        "s_mov_b32 exec_lo, 0x00ffe000\n s_mov_b32 exec_hi, 0x00ffe000\n ds_write_b32 %24, %13 offset:52\n"
        "s_mov_b32 exec_lo, 0x00ffc000\n s_mov_b32 exec_hi, 0x00ffc000\n ds_write_b32 %24, %14 offset:56\n"
        "s_mov_b32 exec_lo, 0x00ff8000\n s_mov_b32 exec_hi, 0x00ff8000\n ds_write_b32 %24, %15 offset:60\n"
        "s_mov_b32 exec_lo, 0x00ff0000\n s_mov_b32 exec_hi, 0x00ff0000\n ds_write_b32 %24, %16 offset:64\n"
        "s_mov_b32 exec_lo, 0x00fe0000\n s_mov_b32 exec_hi, 0x00fe0000\n ds_write_b32 %24, %17 offset:68\n"
        "s_mov_b32 exec_lo, 0x00fc0000\n s_mov_b32 exec_hi, 0x00fc0000\n ds_write_b32 %24, %18 offset:72\n"
        "s_mov_b32 exec_lo, 0x00f80000\n s_mov_b32 exec_hi, 0x00f80000\n ds_write_b32 %24, %19 offset:76\n"
        "s_mov_b32 exec_lo, 0x00f00000\n s_mov_b32 exec_hi, 0x00f00000\n ds_write_b32 %24, %20 offset:80\n"
        "s_mov_b32 exec_lo, 0x00e00000\n s_mov_b32 exec_hi, 0x00e00000\n ds_write_b32 %24, %21 offset:84\n"
        "s_mov_b32 exec_lo, 0x00c00000\n s_mov_b32 exec_hi, 0x00c00000\n ds_write_b32 %24, %22 offset:88\n"
        "s_mov_b32 exec_lo, 0x00800000\n s_mov_b32 exec_hi, 0x00800000\n ds_write_b32 %24, %23 offset:92\n"
        "s_mov_b64 exec, -1\n s_waitcnt lgkmcnt(0)\n"
        :
        : "v"(p[0]), "v"(p[1]), "v"(p[2]), "v"(p[3]), "v"(p[4]), "v"(p[5]), "v"(p[6]), "v"(p[7]), "v"(p[8]), "v"(p[9]), "v"(p[10]), "v"(p[11]), "v"(p[12]), "v"(p[13]), "v"(p[14]), "v"(p[15]), "v"(p[16]), "v"(p[17]), "v"(p[18]), "v"(p[19]), "v"(p[20]), "v"(p[21]), "v"(p[22]), "v"(p[23]), "v"(va)
        : "memory");
}
// (row >= J) ? a : b for the 32-lanes-per-instance layout, where row = min(lane & 31, 23): the lanes with
// row >= J are a compile-time constant, so the select takes its condition from a scalar register pair instead of a
// vector compare (one VALU instruction per row element saved in front of the 24 row loads).  Any other layout
// compares with the lane's `row`.
template <bool LANES32>
__device__ __forceinline__ unsigned select_lanes_ge(int J, unsigned a, unsigned b, unsigned row) {
    if constexpr (!LANES32) return ((unsigned)J <= row) ? a : b;     // (24 lanes per instance: rows straddle wavefronts)
    if (J <= 0) return a;
    const unsigned m32 = J >= 24 ? 0u : ~((1u << J) - 1u);
    const unsigned long long mask = ((unsigned long long)m32 << 32) | m32;
    unsigned r;
    asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(b), "v"(a), "s"(mask));
    (void)row;
    return r;
}
}  // namespace rowk

// Index tables of the row-resident kernel, computed at compile time and read from device memory (a few
// cached 16-byte loads per lane instead of ~200 integer instructions per wave):
//   pos[lig][10]   image position (word offset i*25 + j) of the ten packed-record words lane lig moves
//                  (pieces lig, lig+32, ... of 16 bytes; the last piece is clamped to the record's end)
//   gat[row][24]   image position of P[row][j] in the lower-triangle image: max(row,j)*25 + min(row,j)
struct RowsTables {                  // [v]: lane layout v = 0: 32 lanes per instance, v = 1: 24 lanes per instance
    int pos[2][32][16];  // fp64: 16-byte pieces of two words
    int pos4[2][32][16]; // fp32: 16-byte pieces of four words (75 per record)
    int gat[24][24];
    int jsrc[2][32][4];  // Jacobian record: the record words lane lig moves (clamped to the record) ...
    int jdst[2][32][4];  // ... and where they go in the LDS copy (blocks re-laid on even offsets, lds_word)
};
constexpr int packed_image_pos(int e) {      // packed word e -> i*25 + j with i >= j (see unpack_ij)
    const int r = e / 25, c = e - 25 * r;
    return (c <= r) ? (r * 25 + c) : ((23 - r) * 25 + (c - r - 1));
}
constexpr RowsTables make_rows_tables() {
    RowsTables t{};
    for (int v = 0; v < 2; ++v) {
        const int lpi = v ? 24 : 32;
        for (int l = 0; l < lpi; ++l) {
            for (int k = 0; k < 8; ++k) {
                const int e2r = l + lpi * k, e2 = e2r < 150 ? e2r : 149;
                t.pos[v][l][2 * k] = packed_image_pos(2 * e2);
                t.pos[v][l][2 * k + 1] = packed_image_pos(2 * e2 + 1);
            }
            for (int k = 0; k < 4; ++k) {
                const int e4r = l + lpi * k, e4 = e4r < 75 ? e4r : 74;
                for (int c = 0; c < 4; ++c) t.pos4[v][l][4 * k + c] = packed_image_pos(4 * e4 + c);
            }
            for (int k = 0; k < 4; ++k) {
                const int wr = l + lpi * k, w = wr < JW ? wr : JW - 1, b = w - J_B0;
                t.jsrc[v][l][k] = w;
                t.jdst[v][l][k] = w < J_B0 ? 71 : b + (b >= 9) + (b >= 18) + (b >= 27) + (b >= 66) + (b >= 67);
            }
        }
    }
    for (int r = 0; r < 24; ++r)
        for (int j = 0; j < 24; ++j) t.gat[r][j] = (j <= r) ? (r * 25 + j) : (j * 25 + r);
    return t;
}
__device__ const RowsTables kRowsTables = make_rows_tables();

// waves per SIMD the general build of the row-resident kernel is made for: 3 (<= 168 VGPRs); SLIM: 4
#ifndef ROWS_WAVES
#define ROWS_WAVES 3
#endif
// PIN: per-instance noise (diag(Q), diag(R) from the instance's noise record, staged into LDS with the
// residual) instead of the parameter block
// LPI: lanes per instance.  32: one wavefront hosts two instances (8 of every 32 lanes idle, no barrier
// needed: a block is one wave).  24: a block of three wavefronts hosts eight instances, every lane owns a
// row; instances straddle wavefronts, so the exchanges are separated by real block barriers.
// SLIM (32 lanes per instance, no gyro block in Q -- the headline configuration and the tuning populations;
// sizes below are the fp64 ones):
// with the Jacobian record in registers (JRegs) and the update's columns handed around as DPP operands the
// kernel needs 640 LDS words per instance (exchange image 600, error state 24, residual 8) and <= 128
// VGPRs: four wavefronts per SIMD, sixteen per CU.
#ifndef ESKF_GATHER_MASKED
#define ESKF_GATHER_MASKED 1
#endif
#ifndef ESKF_ROWS_VIA_LDS
#define ESKF_ROWS_VIA_LDS 1
#endif
#ifndef ESKF_PACKED_IMAGE
#define ESKF_PACKED_IMAGE 1
#endif
#define ESKF_LDJ(p) (*(p))
// WAVE (eskf_step_kernel): the wavefront is one of several of its block, each working through pairs of instances on its
// own: the exchanges through LDS are wave-local (LDS instructions of one wavefront execute in order -- only the
// compiler has to be kept from moving them across), the region is the 600-word image alone, and the first predict's
// record comes from the registers `jpre` (words lig, 32 + lig, 64 + lig of the record the block's pre phase left in LDS)
// instead of from a.J.
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
template <typename T, bool PIN, int LPI, bool SLIM, bool WAVE = false, bool Q1 = true>
__device__ __forceinline__ void cov_rows_body(const StepArgs& a_in, const DevParams<T>& prm_in, T* smem, const int vblock, const int lane,
                                              const T* jpre = nullptr, const int kstep = -1, const bool e_keep = false, T* e_out = nullptr,
                                              const bool p_in_lds = false, const bool p_keep = false) {
    // p_in_lds / p_keep (eskf_run_kernel): the packed covariance of the wavefront's two instances is already in the lower
    // halves of their images (words 0..299, where the load phase puts the pieces anyway) -- left there by this wavefront's
    // last round of the previous step -- / stays there for the first round of the next step instead of being stored
    // e_keep / e_out (eskf_run_kernel, every step of a run but the last): the update's error state and status word are handed back
    // in a register -- lane `row` its component, lane 24 of the instance the status word -- instead of being stored to
    // a.e / a.status_ws: the block's next pre phase takes them from LDS (step_block)
    StepArgs a = a_in;
    DevParams<T> prm = prm_in;
    // WAVE: the arguments as of now, straight from the kernarg segment (see StepKernArgs); whatever was read before is dead.
    // kstep >= 0 (eskf_run_kernel): this is step kstep of a run -- from step 1 on the step is in place, and the caller's
    // status words of the step lie kstep strides further on.
    auto refresh_args = [&]() {
        if constexpr (WAVE) {
            const char __attribute__((address_space(4))) * ka = (const char __attribute__((address_space(4))) *)__builtin_amdgcn_kernarg_segment_ptr();
            asm volatile("" : "+s"(ka));
            a = load_kernarg<StepArgs>(ka + offsetof(StepKernArgs<T>, a));
            globalise(a);
            prm = load_kernarg<DevParams<T>>(ka + offsetof(StepKernArgs<T>, prm));
            if (kstep >= 0) {
                const RunArgs ra = load_kernarg<RunArgs>(ka + offsetof(RunKernArgs<T>, ra));
                if (kstep > 0) { a.P_src = a.P; a.xs_src = a.xs; }
                if (a.status) a.status += (int64_t)kstep * ra.status_step;
            }
        }
    };
    refresh_args();
    auto gsync = [&]() { if constexpr (WAVE) wave_sync(); else group_sync(); };
    using namespace rowk;
    static_assert(LPI == 32 || LPI == 24, "lane layouts: 32 or 24 lanes per instance");
    static_assert(!SLIM || LPI == 32, "SLIM: 32 lanes per instance");
    static_assert(!WAVE || SLIM, "WAVE: the slim build");
    static_assert(Q1 || (LPI == 32 && SLIM && !PIN && !WAVE), "Q1 switched off: the slim 32-lane kernel of the two-kernel step only");
    constexpr int G = LPI == 32 ? 2 : 8;    // instances per block
    constexpr int TV = LPI == 32 ? 0 : 1;   // table variant
    // (with per-instance noise the 24-word noise record follows the residual: 664 words, 15 wavefronts per CU)
    constexpr int REGION = WAVE ? (PIN ? 664 : 600) : SLIM ? (PIN ? 664 : 640) : PIN ? R_REGION_PIN : R_REGION;
    constexpr int R_C = SLIM ? 600 : rowk::R_C;
    constexpr int R_RES = SLIM ? 624 : rowk::R_RES;
    constexpr int R_J = SLIM ? 0 : rowk::R_J;      // (SLIM: never addressed)

    const int grp = LPI == 32 ? (lane >> 5) : (lane / 24);
    const int lig = LPI == 32 ? (lane & 31) : (lane - 24 * grp);
    const bool active = LPI == 24 || lig < 24;
    const int row0 = active ? lig : 23;     // (32-lane layout: idle lanes shadow row 23 and write nothing)

    T* sB = smem + grp * REGION + R_BUF;
    T* sC = smem + grp * REGION + R_C;
    T* sJ = smem + grp * REGION + R_J;
    T* sR = smem + grp * REGION + R_RES;
    // (WAVE without per-instance noise: REGION is the 600-word image alone, sC / sR then point PAST the instance's region --
    // into the neighbour's resident covariance, or past the block's allocation.  Every use of them sits in a branch that the
    // DPP paths replace; the static_asserts next to DPPJ / DPPU / DPPR keep it that way.)
    constexpr bool NO_C_RES = WAVE && !PIN;
    const T* sN = sR + 8;                   // PIN: diag(Q) 13 | diag(R) 7 | sigma_om 3 | pad
    // last word of the record in use (dt without the gyro block of Q, else the end of N6) and its LDS slot
    const int jlast = prm.q_gyro ? J_FULL - 1 : J_CORE - 1;
    const int jlast_lds = prm.q_gyro ? L_N6 + 17 : L_DUMP;   // (everything past dt goes to the pad word)
    constexpr int NJ = (JW + LPI - 1) / LPI;
    constexpr int PW = 16 / (int)sizeof(T);         // words per 16-byte piece of the packed record: 2 (fp64) or 4 (fp32)
    constexpr int NPC = 300 / PW;                   // pieces per record: 150 or 75
    constexpr int NV = (NPC + LPI - 1) / LPI;       // pieces per lane
    using T2 = typename Vec2<T>::type;
    using TP = typename Vec16<T>::type;
    auto piece_pos = [&](int k) { return sizeof(T) == 8 ? kRowsTables.pos[TV][lig][k] : kRowsTables.pos4[TV][lig][k]; };

    // one wavefront = two instances, start to finish (waves that loop over several instances with a
    // register prefetch of the next record were measured 2 % slower at three waves per SIMD: the other
    // waves cover the HBM latency, and 32 768 short blocks balance better than 8 192 long ones).
    // Instance indices are 32-bit (the launcher refuses B >= 2^31) and opaque: byte addresses are
    // formed where they are used instead of being kept in registers from the prologue on.
    const int nB = (int)(a.sub_count ? a.sub_first + a.sub_count : a.B);    // (one past the last instance of this launch)
    const int slot = vblock * G + grp;
    // Jacobian record IP of instance INST and the residual of the update -> registers -> LDS.  No
    // predicates either: every lane moves three record words (the 96-word record is padded; what
    // lies beyond the blocks in use lands in the idle N6 area) and one residual word.
#define ESKF_LOAD_J(DST, RES, INST, IP)                                                            \
    {                                                                                              \
        int lj_ = lig;                                                                             \
        asm volatile("" : "+v"(lj_));   /* (per-lane addresses formed here, not kept across rounds) */ \
        if constexpr (LPI == 32) {                                                                 \
            /* wave-uniform base + unsigned 32-bit lane offset, as for the rows; only the third word of a lane can lie \
               in the padding of the record (words >= 64 + lane), which is not fetched */                      \
            const int i0_ = vblock * G;                                                            \
            const char* gJw_ = reinterpret_cast<const char*>(reinterpret_cast<const T*>(a.J) + ((int64_t)(IP) * a.B + i0_) * JW); \
            const unsigned jo_ = (unsigned)((INST) - i0_) * (unsigned)(JW * sizeof(T));           \
            const unsigned lo_ = jo_ + (unsigned)lj_ * (unsigned)sizeof(T);                        \
            static_assert(NJ == 3 && J_CORE - 1 >= 64, "three words per lane, the first two never padding"); \
            DST[0] = ESKF_LDJ(reinterpret_cast<const T*>(gJw_ + (size_t)lo_));                      \
            DST[1] = ESKF_LDJ(reinterpret_cast<const T*>(gJw_ + (size_t)lo_ + (size_t)(32 * sizeof(T)))); \
            const int w2_ = lj_ + 64;                                                              \
            DST[2] = ESKF_LDJ(reinterpret_cast<const T*>(gJw_ + (size_t)(jo_ + (unsigned)(w2_ < jlast ? w2_ : jlast) * (unsigned)sizeof(T)))); \
        } else {                                                                                   \
        const T* gJ_ = reinterpret_cast<const T*>(a.J) + ((int64_t)(IP) * a.B + (INST)) * JW;      \
        _Pragma("unroll") for (int k = 0; k < NJ; ++k) {                                           \
            const int w_ = kRowsTables.jsrc[TV][lj_][k];                                           \
            DST[k] = ESKF_LDJ(gJ_ + (w_ < jlast ? w_ : jlast));   /* (the padding of the record is not fetched) */ \
        }                                                                                          \
        }                                                                                          \
        /* (32 lanes per instance: the residual is words 0..7 of the first record, i.e. already in DST[0]) */ \
        if constexpr (LPI != 32) RES.x = (reinterpret_cast<const T*>(a.J) + (int64_t)(INST) * JW + J_RES)[lj_ & 7]; \
        if (PIN) RES.y = reinterpret_cast<const T*>(a.noise)[(int64_t)(INST) * NOISE_WORDS + (lj_ < 24 ? lj_ : 23)]; \
    }
    // 32 lanes per instance: the record stays in the registers it was fetched into and is
    // handed to the Fx applications as DPP operands (JRegs); it is copied to LDS only for the gyro
    // block of Q (tune mode), whose N6 rows noise_rows() reads from there
    constexpr bool DPPJ = LPI == 32;
    static_assert(DPPJ || !NO_C_RES, "the residual words sR lie outside the LDS region of this build");
    JRegs<T> Jr;
#define ESKF_STAGE_J(SRC, RES)                                                                     \
    {                                                                                              \
        if constexpr (DPPJ) {                                                                      \
            _Pragma("unroll") for (int k = 0; k < 3; ++k) spread_rows(SRC[k], Jr.X[k], Jr.Y[k]);   \
        }                                                                                          \
        if (!SLIM && (!DPPJ || prm.q_gyro)) {                                                      \
            _Pragma("unroll") for (int k = 0; k < NJ; ++k) {                                       \
                const int w_ = kRowsTables.jdst[TV][lig][k];                                       \
                sJ[w_ < jlast_lds ? w_ : jlast_lds] = SRC[k];                                      \
            }                                                                                      \
        }                                                                                          \
        if constexpr (!DPPJ) sR[lig & 7] = RES.x;       /* (several lanes, same value) */          \
        if (PIN) sR[8 + (lig < 24 ? lig : 23)] = RES.y;                                            \
    }
    // row `row` of P from the packed record.  fp64: 24 eight-byte loads per lane with per-lane addresses (the
    // record's 38 cache lines are touched once and then hit in L1; no LDS involved).  fp32: the four-byte
    // version of that is texture-addresser bound (measured 18 % slower on the 1M-instance predict-only
    // configuration), so the record is fetched in 16-byte pieces, scattered into the lower-triangle LDS
    // image and gathered from there.
#define ESKF_ISSUE_ROW()                                                                           \
    {                                                                                              \
            /* wave-uniform base (scalar registers) + an unsigned 32-bit lane offset: one address  \
               instruction per load instead of a sign extension and a 64-bit add */                \
            const int i0_ = vblock * G;                                                   \
            const T* gPw_ = reinterpret_cast<const T*>(a.P_src) + (int64_t)i0_ * 300;              \
            const unsigned io_ = (unsigned)(inst - i0_) * (300u * 8u);                             \
            /* byte offset of P[row][j] = P[max][min]: packed row hi starts at word 25 hi (hi < 12) or    \
               26 (23 - hi) + 1, and holds its elements lo = 0..hi in order.  Formed with a compare and   \
               a select per element instead of being read from a table in memory: the loads of the rows   \
               then do not wait for a first round trip that only fetches their addresses */               \
            const unsigned rw_ = (unsigned)row;                                                    \
            const unsigned va_ = io_ + 8u * (rw_ < 12u ? 25u * rw_ : 26u * (23u - rw_) + 1u);      \
            const unsigned vb_ = io_ + 8u * rw_;                                                   \
            _Pragma("unroll") for (int j = 0; j < 24; ++j) {                                       \
                const unsigned cj_ = 8u * (unsigned)(j < 12 ? 25 * j : 26 * (23 - j) + 1);         \
                /* (j <= row) ? va + 8 j : vb + c_j  as  select(vb + (c_j - 8 j), va) + 8 j: which lanes hold a     \
                   row >= j is a CONSTANT of the lane layout -- handed to the select as a scalar mask, no       \
                   compare; the + 8 j rides in the load's offset field */                                      \
                /* The common part m_j = min(8 j, c_j) of the two alternatives is added AFTER the zero extension of the  \
                   lane offset, where the compiler can move it into the load's offset field (a 32-bit sum could wrap,    \
                   so it would not); what remains on either side is non-negative, so nothing wraps before it. */        \
                const unsigned e8_ = 8u * (unsigned)j;                                                               \
                const unsigned mj_ = cj_ < e8_ ? cj_ : e8_;                                                          \
                const unsigned o_ = rowk::select_lanes_ge<LPI == 32>(j, va_ + (e8_ - mj_), vb_ + (cj_ - mj_), rw_);   \
                p[j] = *reinterpret_cast<const T*>(reinterpret_cast<const char*>(gPw_) + (size_t)o_ + (size_t)mj_);   \
            }                                                                                      \
    }
#define ESKF_UNPACK_ROW()                                                                          \
    {                                                                                              \
        const T* gP_ = reinterpret_cast<const T*>(a.P_src) + (int64_t)inst * 300;                  \
        if (sizeof(T) == 8) {                                                                      \
            /* wave-uniform base (scalar registers) + an unsigned 32-bit lane offset: one address  \
               instruction per load instead of a sign extension and a 64-bit add */                \
            const int i0_ = vblock * G;                                                   \
            const T* gPw_ = reinterpret_cast<const T*>(a.P_src) + (int64_t)i0_ * 300;              \
            const unsigned io_ = (unsigned)(inst - i0_) * (300u * 8u);                             \
            /* byte offset of P[row][j] = P[max][min]: packed row hi starts at word 25 hi (hi < 12) or    \
               26 (23 - hi) + 1, and holds its elements lo = 0..hi in order.  Formed with a compare and   \
               a select per element instead of being read from a table in memory: the loads of the rows   \
               then do not wait for a first round trip that only fetches their addresses */               \
            const unsigned rw_ = (unsigned)row;                                                    \
            const unsigned va_ = io_ + 8u * (rw_ < 12u ? 25u * rw_ : 26u * (23u - rw_) + 1u);      \
            const unsigned vb_ = io_ + 8u * rw_;                                                   \
            _Pragma("unroll") for (int j = 0; j < 24; ++j) {                                       \
                const unsigned cj_ = 8u * (unsigned)(j < 12 ? 25 * j : 26 * (23 - j) + 1);         \
                /* (j <= row) ? va + 8 j : vb + c_j  as  select(vb + (c_j - 8 j), va) + 8 j: which lanes hold a     \
                   row >= j is a CONSTANT of the lane layout -- handed to the select as a scalar mask, no       \
                   compare; the + 8 j rides in the load's offset field */                                      \
                /* The common part m_j = min(8 j, c_j) of the two alternatives is added AFTER the zero extension of the  \
                   lane offset, where the compiler can move it into the load's offset field (a 32-bit sum could wrap,    \
                   so it would not); what remains on either side is non-negative, so nothing wraps before it. */        \
                const unsigned e8_ = 8u * (unsigned)j;                                                               \
                const unsigned mj_ = cj_ < e8_ ? cj_ : e8_;                                                          \
                const unsigned o_ = rowk::select_lanes_ge<LPI == 32>(j, va_ + (e8_ - mj_), vb_ + (cj_ - mj_), rw_);   \
                p[j] = *reinterpret_cast<const T*>(reinterpret_cast<const char*>(gPw_) + (size_t)o_ + (size_t)mj_);   \
            }                                                                                      \
            gsync();   /* (the staged Jacobian record and residual are visible to every lane) */ \
        } else {                                                                                   \
            const TP* gPp_ = reinterpret_cast<const TP*>(gP_);                                     \
            TP pv_[NV];                                                                            \
            int off_[PW * NV], gat_[24];                                                           \
            _Pragma("unroll") for (int k = 0; k < NV; ++k) {                                       \
                const int e_ = lig + k * LPI;                                                      \
                pv_[k] = gPp_[e_ < NPC ? e_ : NPC - 1];                                            \
            }                                                                                      \
            _Pragma("unroll") for (int k = 0; k < PW * NV; ++k) off_[k] = piece_pos(k);            \
            _Pragma("unroll") for (int j = 0; j < 24; ++j) gat_[j] = kRowsTables.gat[row][j];      \
            _Pragma("unroll") for (int k = 0; k < NV; ++k) {                                       \
                _Pragma("unroll") for (int c = 0; c < PW; ++c) sB[off_[PW * k + c]] = reinterpret_cast<const T*>(&pv_[k])[c]; \
            }                                                                                      \
            gsync();                                                                          \
            _Pragma("unroll") for (int j = 0; j < 24; ++j) p[j] = sB[gat_[j]];                     \
            gsync();                                                                          \
        }                                                                                          \
    }
    // one predict on the row-resident P with the Jacobian record that sits in LDS
#define ESKF_PREDICT_ROWS()                                                                        \
    {                                                                                              \
        if constexpr (DPPJ) fx_apply_store_dpp<T, Q1>(Jr, p, sB + row * PS, active);               \
        else fx_apply_store(sJ, p, sB + row * PS, active);   /* column `row` of Fx P = row `row` of W = P Fx^T */ \
        gsync();                                                                              \
        if constexpr (DPPJ) fx_apply_col_dpp<T, Q1>(Jr, sB + row, p);                              \
        else fx_apply_col(sJ, sB + row, p);  /* column `row` of Fx P Fx^T = row `row` of P' */     \
        refresh_args();                                                                            \
        noise_rows<T, PIN, !SLIM, LPI == 32>(prm, sJ, sN, row, p);                                 \
        gsync();                                                                              \
    }
    const bool valid = slot < nB;           // tail: recompute the last instance, write nothing
    int inst = valid ? slot : nB - 1;
    int row = row0;
    asm volatile("" : "+v"(row), "+v"(inst));
    constexpr int sel[7] = {18, 19, 20, 21, 22, 23, 15};
    T p[24];                                // row `row` of P
    ESKF_CSTAMP(0)
    if constexpr (DPPJ && (sizeof(T) == 8 || ESKF_ROWS_VIA_LDS)) {
        // everything this wavefront reads is requested in one go -- the packed covariance first, then the
        // Jacobian record and the residual -- and waited for once: one memory round trip, not two
        T j0[NJ];
        T2 r0;
        r0.x = r0.y = T(0);
#if ESKF_ROWS_VIA_LDS
        // The packed record arrives as coalesced 16-byte pieces (five per lane in fp64, three in fp32), goes into LDS in
        // its own order and is gathered from there: P[row][j] = packed[max(row, j)][min(row, j)], the lane set of either
        // case a constant of the layout (select_lanes_ge), the element's constant in the read's offset field.  Round 2
        // had replaced exactly this by 24 direct eight-byte loads per lane when the kernel was bound by its LDS traffic;
        // with the Jacobian entries and the update's columns on DPP operands it is the memory instructions that
        // count (24 + 3 loads -> 5 + 3: +4 % on the step, same box).
        typedef T row_piece_t __attribute__((ext_vector_type(PW)));
        row_piece_t pv_[NV];
        if (!(WAVE && p_in_lds)) {
            const int i0_ = vblock * G;
            const char* gPb_ = reinterpret_cast<const char*>(reinterpret_cast<const T*>(a.P_src) + (int64_t)i0_ * 300);
            const unsigned so_ = (unsigned)(inst - i0_) * (unsigned)(300 * sizeof(T));
#pragma unroll
            for (int k = 0; k < NV; ++k) {
                const int e_ = lig + k * LPI;
                pv_[k] = *reinterpret_cast<const row_piece_t*>(gPb_ + (size_t)(so_ + 16u * (unsigned)(e_ < NPC ? e_ : NPC - 1)));
            }
        }
        if constexpr (WAVE) {
#pragma unroll
            for (int k = 0; k < NJ; ++k) j0[k] = jpre[k];
            if (PIN) r0.y = reinterpret_cast<const T*>(a.noise)[(int64_t)inst * NOISE_WORDS + (lig < 24 ? lig : 23)];
        } else {
            ESKF_LOAD_J(j0, r0, inst, 0)
        }
        {
            if (!(WAVE && p_in_lds)) {
#pragma unroll
                for (int k = 0; k < NV; ++k) {
                    const int e_ = lig + k * LPI;
                    *reinterpret_cast<row_piece_t*>(sB + PW * (e_ < NPC ? e_ : NPC - 1)) = pv_[k];
                }
            }
            gsync();
            const unsigned rw_ = (unsigned)row;
            // (LDS byte addresses: the region's base is part of both alternatives, so nothing is added after the select)
            const unsigned lb_ = (unsigned)(size_t)(__attribute__((address_space(3))) T*)sB;
            constexpr unsigned ES = (unsigned)sizeof(T);
            const unsigned va_ = lb_ + ES * (rw_ < 12u ? 25u * rw_ : 26u * (23u - rw_) + 1u), vb_ = lb_ + ES * rw_;
            if constexpr (ESKF_GATHER_MASKED) rowk::gather_row_masked(va_, vb_, p);
            else
            static_for<0, 24>([&](auto JJ) {
                constexpr int j = decltype(JJ)::value;
                constexpr unsigned cj_ = ES * (unsigned)(j < 12 ? 25 * j : 26 * (23 - j) + 1), e8_ = ES * (unsigned)j;
                constexpr unsigned mj_ = cj_ < e8_ ? cj_ : e8_;
                const unsigned o_ = rowk::select_lanes_ge<true>(j, va_ + (e8_ - mj_), vb_ + (cj_ - mj_), rw_);
                typedef const volatile __attribute__((address_space(3))) T lds_elem_t;
                p[j] = *(lds_elem_t*)(size_t)(o_ + mj_);
            });
            gsync();
        }
#else
        ESKF_ISSUE_ROW()
        ESKF_LOAD_J(j0, r0, inst, 0)
#endif
        ESKF_STAGE_J(j0, r0)
        if (PIN || !SLIM) gsync();      // (what has been staged -- noise record, N6 rows -- is visible to every lane)
    } else {
        {   // first Jacobian record and residual into LDS, packed covariance into registers
            T j0[NJ];
            T2 r0;
            r0.x = r0.y = T(0);
            ESKF_LOAD_J(j0, r0, inst, 0)
            ESKF_STAGE_J(j0, r0)
        }
        ESKF_UNPACK_ROW()
    }
    // 32 lanes per instance: residual (7) and non-finite flag of the update are words 0..7 of the first predict's
    // record -- lanes 0..7 of the first 16-lane row of its first register, which spread_rows() has just made
    // available to both rows: the update takes them as DPP operands (no separate load, no LDS staging)
#ifdef ESKF_X_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    pin_row(p);
#endif
    ESKF_CSTAMP(1)
    T Xres = T(0);
    if constexpr (DPPJ) {
        Xres = Jr.X[0];
        asm("s_nop 1" : "+v"(Xres));    // (a copy, if the allocator makes one here, is two wait states old before any DPP read)
    }

    // ================================ predict(s) =======================================
    const int n_pred_run = a.n_pred;
    if constexpr (DPPJ) {
        // epochs: the record of the next predict is requested before this one is computed (six registers)
        // and put in place after it; the last (or only) predict of a launch has nothing to fetch
        for (int ip = 0; ip < n_pred_run; ++ip) {
            const bool more = ip + 1 < n_pred_run;    // uniform
            T jn[NJ];
            T2 rn;
            rn.x = rn.y = T(0);
#pragma unroll
            for (int k = 0; k < NJ; ++k) jn[k] = T(0);
            if (more) ESKF_LOAD_J(jn, rn, inst, ip + 1)
            ESKF_PREDICT_ROWS()
            if (more) {
                ESKF_STAGE_J(jn, rn)
                gsync();
            }
        }
    } else {
        for (int ip = 0; ip < n_pred_run; ++ip) {
            if (ip > 0) {                   // (the first record was staged before the loop)
                T jn[NJ];
                T2 rn;
                rn.x = rn.y = T(0);
                ESKF_LOAD_J(jn, rn, inst, ip)
                ESKF_STAGE_J(jn, rn)
                gsync();
            }
            ESKF_PREDICT_ROWS()
        }
    }

    // ================================ update ===========================================
    // The 7 measurements one after the other; sv is the pivot d_m of S = L D L^T.  A zero or
    // non-finite pivot is the reference's LinAlgError (Filter.py:356-361: nothing changes): the
    // remaining measurements are skipped and the ones already applied are taken back.
    // image positions of the packed-record words this lane will store, fetched now (table in memory): asked
    // for at the point of use they cost every wavefront an exposed round trip right before its stores
    int off[PW * NV];
    if constexpr (!(LPI == 32 && ESKF_PACKED_IMAGE)) {      // (32 lanes per instance: the store needs no positions, see there)
#pragma unroll
        for (int k = 0; k < PW * NV; ++k) off[k] = piece_pos(k);
#pragma unroll
        for (int k = 0; k < PW * NV; ++k) asm volatile("" : "+v"(off[k]));
    }

#ifdef ESKF_X_STAMPS
    pin_row(p);
#endif
    ESKF_CSTAMP(2)
    refresh_args();
    T e = T(0);                             // component `row` of the error state
    int status = 0;
    bool fail = false;
    const bool do_update_run = a.do_update != 0;
    if (do_update_run) {
        constexpr bool DPPU = LPI == 32;
        static_assert(DPPU || !NO_C_RES, "the residual words sR lie outside the LDS region of this build");
        if constexpr (DPPU) status = (dpp_bcast<7>(Xres) != T(0)) ? DVI_STATUS_NONFINITE : 0;
        else status = (sR[7] != T(0)) ? DVI_STATUS_NONFINITE : 0;
        T E[7];                             // error estimate at the measured components (group-uniform)
        if constexpr (!DPPU) {
#pragma unroll
            for (int m = 0; m < 7; ++m) E[m] = T(0);
        }
        int napplied = 0;
        unsigned long long okm[7] = {0, 0, 0, 0, 0, 0, 0};  // (DPPU) lanes whose instance has applied measurement m
        unsigned long long failm = 0;                       // (DPPU) lanes whose update has met an unusable pivot
        // 32 lanes per instance: the column every lane needs is handed around as DPP operands
        // (no LDS read-back, no barrier); the LDS copy is only written, for the take-back below
        constexpr int CS = DPPU ? 32 : 24;  // stride of the published columns (DPPU: every lane writes, no predicate)
        if constexpr (DPPU) {
            static_for<0, 7>([&](auto M) {
                constexpr int m = decltype(M)::value;
                constexpr int sm = sel[m];
                const T pm = p[sm];
                lds_store1(sB + m * CS + lig, pm);
                T X, Y;                     // column sel_m of the current P: X rows 0..15, Y rows 16..23
                spread_rows(pm, X, Y);
                const T csel = dpp_bcast<(sm & 15)>(sm < 16 ? X : Y);
                const T sv = csel + (PIN ? sN[13 + m] : prm.r[m]);
                // lanes whose pivot is zero or non-finite, as a scalar mask straight from the class compare and kept
                // as one: a `bool fail` carried through these branches is turned into a register and back by the
                // compiler at every measurement
                failm |= unusable_pivot_mask(sv);
                okm[m] = ~failm;                                // (the take-back reads it)
                if (!__builtin_amdgcn_inverse_ballot_w64(failm)) {
                    const T g = fast_rcp(sv);
                    const T rm = dpp_bcast<m>(Xres);
                    const T tau = (m == 0) ? rm * g : (rm - E[m]) * g;      // (E starts at zero: nothing to subtract yet)
                    const T nkk = pm * -g;  // minus the gain of this measurement, component `row`
                    if constexpr (m == 0) e = pm * tau;
                    else e += pm * tau;
                    static_for<m + 1, 7>([&](auto M2) {
                        constexpr int s2 = sel[decltype(M2)::value];
                        dpp_fmac<(s2 & 15), m == 0>(E[decltype(M2)::value], s2 < 16 ? X : Y, tau);
                    });
                    // Joseph form with the optimal gain, Filter.py:383-384: p[j] -= kk * P[j][sel_m]
                    static_for<12, 24>([&](auto J) {
                        constexpr int j = decltype(J)::value;
                        dpp_fmac<(j & 15)>(p[j], j < 16 ? X : Y, nkk);
                    });
                    static_for<0, 12>([&](auto J) {
                        constexpr int j = decltype(J)::value;
                        dpp_fmac<(j & 15)>(p[j], X, nkk);
                    });
                }
                pin_row(p);
                asm volatile("" : "+v"(e), "+v"(E[0]), "+v"(E[1]), "+v"(E[2]), "+v"(E[3]), "+v"(E[4]), "+v"(E[5]), "+v"(E[6]));
            });
            fail = __builtin_amdgcn_inverse_ballot_w64(failm);
            if (__builtin_expect(fail, 0)) gsync();   // (the published columns are read below)
        } else {
#pragma unroll
        for (int m = 0; m < 7; ++m) {
            T* cb = sB + m * 24;            // (the exchange image is idle during the update; the columns
                                            //  stay there for the take-back)
            const T pm = p[sel[m]];
            if (active) cb[row] = pm;       // column sel_m of the current P
            gsync();
            // upper half of the column first: it holds the measured components
            T c[12];
#pragma unroll
            for (int j = 0; j < 12; ++j) c[j] = cb[12 + j];
            const T sv = c[sel[m] - 12] + (PIN ? sN[13 + m] : prm.r[m]);
            fail = fail || !(sv != T(0) && t_finite(sv));
            if (!fail) {
                const T g = fast_rcp(sv);
                const T tau = (sR[m] - E[m]) * g;
                const T kk = pm * g;        // gain of this measurement, component `row`
                e += pm * tau;
#pragma unroll
                for (int m2 = m + 1; m2 < 7; ++m2) E[m2] += c[sel[m2] - 12] * tau;
#pragma unroll
                for (int j = 0; j < 12; ++j) p[12 + j] -= kk * c[j];   // Joseph form with the optimal gain, Filter.py:383-384
                asm volatile("" : : : "memory");
#pragma unroll
                for (int j = 0; j < 12; ++j) c[j] = cb[j];
#pragma unroll
                for (int j = 0; j < 12; ++j) p[j] -= kk * c[j];
                napplied = m + 1;
            }
            // one measurement at a time: without these fences the scheduler runs ahead along the
            // pivot chain and parks seven columns in scratch for the deferred row updates
            pin_row(p);
            asm volatile("" : "+v"(e), "+v"(E[0]), "+v"(E[1]), "+v"(E[2]), "+v"(E[3]), "+v"(E[4]), "+v"(E[5]), "+v"(E[6]));
        }
        }
        if (__builtin_expect(fail, 0)) {
            // take the applied measurements back, last first (the predicted covariance is restored
            // to rounding; bit-exact when the first pivot fails, e.g. P = 0 and R = 0)
            status |= DVI_STATUS_SINGULAR;
            if constexpr (DPPU) {           // (how many this lane's instance applied: from the scalar masks of the loop above)
#pragma unroll
                for (int m = 0; m < 7; ++m) napplied += (int)((okm[m] >> lane) & 1ull);
            }
#pragma unroll
            for (int m = 6; m >= 0; --m) {
                if (m < napplied) {
                    const T* cb = sB + m * CS;
                    const T g = fast_rcp(cb[sel[m]] + (PIN ? sN[13 + m] : prm.r[m]));
                    const T kk = cb[row] * g;
#pragma unroll
                    for (int j = 0; j < 24; ++j) p[j] += kk * cb[j];
                    pin_row(p);
                }
            }
        }
    }

#ifdef ESKF_X_STAMPS
    pin_row(p);
#endif
    ESKF_CSTAMP(3)
    if (do_update_run) {
        const bool upd = !fail;
        if (row >= 9 && row < 15 && ((prm.frozen_mask >> (row - 9)) & 1)) e = T(0);   // Filter.py:377-379
        constexpr bool DPPR = LPI == 32;    // the six attitude components of the error state reach every lane as DPP operands
        static_assert(DPPR || !NO_C_RES, "the error-state words sC lie outside the LDS region of this build");
        if (upd && active) {
            if constexpr (!DPPR) sC[row] = e;
            if (valid && !(WAVE && e_keep)) {
                if constexpr (LPI == 32) {      // (wave-uniform base + 32-bit lane offset)
                    const int i0_ = vblock * G;
                    char* gE_ = reinterpret_cast<char*>(reinterpret_cast<T*>(a.e) + (int64_t)i0_ * 24);
                    const unsigned eo_ = ((unsigned)(inst - i0_) * 24u + (unsigned)row) * (unsigned)sizeof(T);
                    *reinterpret_cast<T*>(gE_ + (size_t)eo_) = e;
                } else {
                    reinterpret_cast<T*>(a.e)[(int64_t)inst * 24 + row] = e;
                }
            }
        }
        T Gt[3] = {T(0), T(0), T(0)}, Gc[3] = {T(0), T(0), T(0)};
        if constexpr (DPPR) {
            // (one LDS round trip and one barrier less than publishing e: rows 6..8 sit in the first 16-lane row of
            //  the instance, rows 21..23 in lanes 5..7 of the second)
            T Xe, Ye;
            spread_rows(e, Xe, Ye);
            Gt[0] = T(0.5) * dpp_bcast<6>(Xe); Gt[1] = T(0.5) * dpp_bcast<7>(Xe); Gt[2] = T(0.5) * dpp_bcast<8>(Xe);
            Gc[0] = T(0.5) * dpp_bcast<5>(Ye); Gc[1] = T(0.5) * dpp_bcast<6>(Ye); Gc[2] = T(0.5) * dpp_bcast<7>(Ye);
        } else {
            gsync();
        }
        T* sY = sB;     // [24][XS]: columns 6:9 -> words 0:3, columns 21:24 -> words 4:7
        if (upd) {
            // ---- reset G P G^T (Filter.py:387-390), G = I - [theta/2]x on 6:9 and 21:24.
            // column part, in registers
            if constexpr (!DPPR) {
#pragma unroll
                for (int k = 0; k < 3; ++k) { Gt[k] = T(0.5) * sC[6 + k]; Gc[k] = T(0.5) * sC[21 + k]; }
            }
            const T a0 = p[6], a1 = p[7], a2 = p[8];
            p[6] = a0 + Gt[2] * a1 - Gt[1] * a2;
            p[7] = a1 - Gt[2] * a0 + Gt[0] * a2;
            p[8] = a2 + Gt[1] * a0 - Gt[0] * a1;
            const T b0 = p[21], b1 = p[22], b2 = p[23];
            p[21] = b0 + Gc[2] * b1 - Gc[1] * b2;
            p[22] = b1 - Gc[2] * b0 + Gc[0] * b2;
            p[23] = b2 + Gc[1] * b0 - Gc[0] * b1;
            if (active) {
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    sY[row * XS + k] = p[6 + k];
                    sY[row * XS + 4 + k] = p[21 + k];
                }
            }
        }
        gsync();
        // row part: only rows 6:9 and 21:24 change.  For a column j outside those blocks
        // P_new[i][j] = P_new[j][i], which lane j already holds (its column result); inside
        // them the 3x3 row combination of the column results of the lane's own triple.
        const bool lo = row >= 6 && row < 9, hi = row >= 21;
        if (upd && (lo || hi)) {
            const int tb = lo ? 6 : 21;
            const int t = row - tb;
            const int sw = lo ? t : 4 + t;
            T g0, g1, g2;
            if constexpr (DPPR) { g0 = lo ? Gt[0] : Gc[0]; g1 = lo ? Gt[1] : Gc[1]; g2 = lo ? Gt[2] : Gc[2]; }
            else { g0 = T(0.5) * sC[tb]; g1 = T(0.5) * sC[tb + 1]; g2 = T(0.5) * sC[tb + 2]; }   // (read again: not kept across)
            const T k0 = (t == 0) ? T(1) : (t == 1) ? -g2 : g1;
            const T k1 = (t == 0) ? g2 : (t == 1) ? T(1) : -g0;
            const T k2 = (t == 0) ? -g1 : (t == 1) ? g0 : T(1);
            const T* y0 = sY + tb * XS;
#pragma unroll
            for (int j = 0; j < 24; ++j) {
                const bool jl = j >= 6 && j < 9, jh = j >= 21;
                if (jl || jh) {
                    const int sj = jl ? j - 6 : 4 + j - 21;
                    p[j] = k0 * lds_load1(y0 + sj) + k1 * lds_load1(y0 + XS + sj) + k2 * lds_load1(y0 + 2 * XS + sj);
                } else {
                    p[j] = lds_load1(sY + j * XS + sw);
                }
            }
        }
        gsync();
        if (lig == 0 && valid) {
            if (!(WAVE && e_keep)) a.status_ws[inst] = status | DVI_STATUS_PENDING;   // the error state of this update awaits injection
            if (a.status) a.status[inst] = status;
        }
        if constexpr (WAVE) {
            *e_out = (lig == 24) ? status_as_word(valid ? (status | DVI_STATUS_PENDING) : 0, T(0)) : e;
        }
    }

#ifdef ESKF_X_STAMPS
    pin_row(p);
#endif
    ESKF_CSTAMP(4)
    // ---- store: rows -> lower-triangle image -> packed record (16-byte stores) ----------------
    if constexpr (LPI == 32 && ESKF_PACKED_IMAGE) {
        // 32 lanes per instance: the rows go into LDS in the ORDER OF THE PACKED RECORD (row h starts at word 25 h or
        // 26 (23 - h) + 1 and holds its elements 0..h), so the record leaves as a straight copy -- one 16-byte LDS read
        // and one 16-byte store per piece, no table of image positions (three loads), no per-word reads.  Element j
        // exists in rows >= j only: a constant set of lanes per j.
        const int st_ = row < 12 ? 25 * row : 26 * (23 - row) + 1;
        T* rp_ = sB + st_;
        store_row_masked((unsigned)(size_t)(__attribute__((address_space(3))) T*)rp_, p);
        gsync();
        if (valid && !(WAVE && p_keep)) {
            // (wave-uniform base + unsigned 32-bit lane offset, the piece's constant in the offset field -- as for the loads)
            const int i0_ = vblock * G;
            char* gPb_ = reinterpret_cast<char*>(reinterpret_cast<T*>(a.P) + (int64_t)i0_ * 300);
            const unsigned so_ = (unsigned)(inst - i0_) * (unsigned)(300 * sizeof(T)) + 16u * (unsigned)lig;
            TP v[NV];
#pragma unroll
            for (int k = 0; k < NV; ++k) {
                const int e_ = lig + k * LPI;
                // (volatile: all the reads stay in front of the stores, the clamped last one included -- read inside its
                //  store's predicate it is one more LDS round trip at the very end of the wavefront's life)
                typedef T piece_vec_t __attribute__((ext_vector_type(PW)));
                typedef const volatile __attribute__((address_space(3))) piece_vec_t lds_piece_t;
                const piece_vec_t t_ = *(lds_piece_t*)(sB + PW * (e_ < NPC ? e_ : NPC - 1));
                static_assert(sizeof(piece_vec_t) == sizeof(TP), "16-byte pieces");
                __builtin_memcpy(&v[k], &t_, sizeof(TP));
            }
#pragma unroll
            for (int k = 0; k < NV; ++k) {
                const int e_ = lig + k * LPI;
                if (e_ < NPC) *reinterpret_cast<TP*>(gPb_ + (size_t)so_ + (size_t)(16 * LPI * k)) = v[k];
            }
        }
    } else {
    if (active) {   // (the whole row: the words above the diagonal are never read)
#pragma unroll
        for (int j = 0; j < 24; ++j) lds_store1(sB + row * PS + j, p[j]);
    }
    gsync();
    if (valid) {
        // (all the reads of the image first, then the stores: read-wait-store per piece is five LDS round trips
        //  in a row at the very end of the wavefront's life; the last piece's index is clamped instead of
        //  predicated so that the reads are unconditional)
        TP* gPp = reinterpret_cast<TP*>(reinterpret_cast<T*>(a.P) + (int64_t)inst * 300);
        TP v[NV];
#pragma unroll
        for (int k = 0; k < NV; ++k) {
#pragma unroll
            for (int c = 0; c < PW; ++c) reinterpret_cast<T*>(&v[k])[c] = lds_load1(sB + off[PW * k + c]);
        }
        asm volatile("" ::: "memory");
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            const int e_ = lig + k * LPI;
            if (e_ < NPC) gPp[e_] = v[k];
        }
    }
    }
    ESKF_CSTAMP(5)
#ifdef ESKF_X_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    ESKF_CSTAMP(6)
#undef ESKF_LOAD_J
#undef ESKF_STAGE_J
#undef ESKF_UNPACK_ROW
#undef ESKF_ISSUE_ROW
#undef ESKF_PREDICT_ROWS
}

template <typename T, bool PIN = false, int LPI = 32, bool SLIM = false, bool Q1 = true>
__global__ __launch_bounds__(LPI == 32 ? 64 : 192, SLIM ? 4 : ROWS_WAVES) void eskf_cov_rows_kernel(StepArgs a, DevParams<T> prm) {
    constexpr int G = LPI == 32 ? 2 : 8;
    constexpr int REGION = SLIM ? (PIN ? 664 : 640) : PIN ? rowk::R_REGION_PIN : rowk::R_REGION;
    __shared__ __attribute__((aligned(16))) T smem[G * REGION];
    cov_rows_body<T, PIN, LPI, SLIM, false, Q1>(a, prm, smem, (int)blockIdx.x + (int)(a.sub_first / G), (int)threadIdx.x);
}

// ----------------------------------------------------------------------------------
// eskf_step_kernel: one predict(+update) step in ONE launch, the Jacobian records never leave the chip.
//
// A block of eight wavefronts owns NI = 16 ROUNDS instances.  Phase 1: wavefronts 0..3 are the four chains of
// eskf_pre4_kernel (pre4_body<KEEP>: same code, same bits) on the block's instances, one lane each -- nominal state to
// HBM, the records (residual + Jacobian blocks, 75 words) into LDS staging rows; wavefronts 4..7 only join the three
// barriers.  Then every wavefront takes the records of ITS 2 ROUNDS instances out of the staging rows into registers
// (three words per lane and round), a barrier hands the LDS over, and phase 2 runs ROUNDS rounds of the row-resident
// covariance step per wavefront (cov_rows_body<WAVE>: the body of eskf_cov_rows_kernel with wave-local exchanges and the
// first record in registers) on a 600-word image per instance that overlays the staging rows.
//
// What it saves over eskf_pre4_kernel + eskf_cov_rows_kernel: the record's round trip through memory (fp64, B = 65 536:
// 2 x 40 MB of the 477 MB a step moves; builds of the two kernels with the record stores / loads compiled out put the
// bound at 71.6 us against 87.0) and a kernel boundary.  What it costs: the block's phase 1 is serial with its rounds
// (two blocks per CU: while one is in phase 1 the other one's eight wavefronts have the CU).
// ROUNDS = 4: full batches (NI = 64: every lane of phase 1 busy); 3 / 2 / 1: batches whose 64-instance blocks would not come
// out as whole layers of one block per CU (phase 1 then runs with 48 / 32 / 16 of its 64 lanes) -- the launcher decides
// (pick_rounds).
// Not covered (the launcher falls back to the two kernels): per-instance noise at more than 12 288 instances, a gyro block in Q or kept Jacobians
// (85-word records), gain output, launches over a part of the batch, epochs.
// ----------------------------------------------------------------------------------
#if defined(ESKF_X_RUNSTAMPS) && defined(ESKF_TU_FUSED)     // diagnostic build: 100 MHz stamps of the run kernel's phases
__device__ unsigned long long g_run_stamps[1024 * 16 * 12];         // [block][step][slot]
hipError_t x_read_run_stamps(unsigned long long* out) { return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_run_stamps), sizeof(g_run_stamps)); }
#define ESKF_RSTAMP(SLOT) { if (kstep >= 0 && kstep < 16 && blockIdx.x < 1024 && lane == 0) g_run_stamps[((int)blockIdx.x * 16 + kstep) * 12 + (SLOT)] = __builtin_amdgcn_s_memrealtime(); }
#else
#define ESKF_RSTAMP(SLOT) {}
#endif
// One step of a block: phase 1, hand-over, rounds (see above).  kstep: -1 = the launch is this one step (eskf_step_kernel),
// else the step's number in the run (eskf_run_kernel).
// e_from_lds / e_to_lds (eskf_run_kernel): the error states and status words travel from this step's rounds to the
// next step's pre phase through LDS -- in registers until the block's rounds are over (in the record registers a round
// has used up), then as [instance][ES] behind the staging rows -- instead of through a.e / a.status_ws (25 MB per step at
// B = 65 536, read back with a 192-byte stride).  The first step of a run reads them from memory, the last one writes them
// there: what a run leaves behind is what its steps called one by one leave behind.
// PIN: per-instance noise (the tuning populations): the images are 664 words per instance (the noise record behind the
// exchange image), 85 KB of LDS per block in fp64 -- one block per CU, which is all a population needs.
template <typename T, int ROUNDS, bool PIN = false>
__device__ __forceinline__ void step_block(const StepArgs& a, const DevParams<T>& prm, unsigned char* smem_raw, const int wave,
                                           const int lane, const int kstep, const bool e_from_lds = false, const bool e_to_lds = false) {
    constexpr int NI = 16 * ROUNDS;
    constexpr int IMG = PIN ? 664 : 600;                 // words per exchange image (PIN: the noise record behind it)
    constexpr int WREGION = 2 * IMG;                     // words per wavefront: the images of its two instances
    static_assert(300 + 4 * J_CORE <= 600, "four staging rows in the upper half of an image, the packed covariance in the lower");
    // The order of the rounds alternates from step to step of a run (0 1 2 3, then 3 2 1 0, ...): the pair of instances a
    // wavefront works on LAST in one step is the pair it works on FIRST in the next, and the packed covariance of that
    // pair stays where the round's store phase has put it anyway -- words 0..299 of the two images -- instead of going to
    // memory and coming back (a quarter of the covariance traffic of a 64-instance block, all of it with 16 instances per
    // block).  The first step of a run loads every covariance, the last one stores every one.
    const bool rev = kstep > 0 && (kstep & 1);
    if (wave == 0) ESKF_RSTAMP(0)
    // ---- phase 1
    if (wave < 4) {
        // (the pre phase is what the block's eight wavefronts wait for: its four chains go first on their SIMDs, next to
        //  the other block's rounds -- same-box A/B of the run kernel: +1 % at B = 65 536, +5.5 % at 8 192)
        __builtin_amdgcn_s_setprio(2);
        pre4_body<T, true, NI, IMG>(a, prm, smem_raw, (int)blockIdx.x, e_from_lds, lane, wave);
        __builtin_amdgcn_s_setprio(0);
    } else {
        __syncthreads();                                // barrier 1
        __syncthreads();                                // barrier 1b
        __syncthreads();                                // barrier 2: the staged records are complete
    }
    if (wave == 0) ESKF_RSTAMP(1)
    // ---- hand-over: the records of this wavefront's instances, three words per lane and round
    T jr[ROUNDS][3];
    {
        const T* sJ = reinterpret_cast<const T*>(smem_raw);
        const int grp = lane >> 5, lig = lane & 31;
        const int w2 = lig + 64 < J_CORE - 1 ? lig + 64 : J_CORE - 1;     // (the record ends at word 74)
#pragma unroll
        for (int r = 0; r < ROUNDS; ++r) {      // (jr[r]: the record of the r-th round in THIS step's order)
            const int rr = rev ? ROUNDS - 1 - r : r;
            const T* row = sJ + staging_row_word<IMG>(wave * 2 * ROUNDS + 2 * rr + grp);
            jr[r][0] = row[lig];
            jr[r][1] = row[32 + lig];
            jr[r][2] = row[w2];
        }
    }
    __syncthreads();                                    // everyone has its records: the rows become exchange images
    if (wave == 0) ESKF_RSTAMP(2)
    // ---- phase 2
    T* smem = reinterpret_cast<T*>(smem_raw) + wave * WREGION;
    const int vb0 = (int)blockIdx.x * (NI / 2) + wave * ROUNDS;
    const int nB = (int)a.B;
#pragma unroll 1
    for (int r = 0; r < ROUNDS; ++r) {
        T e_r = T(0);       // (e_to_lds: what this round hands to the next pre phase)
        // a pair that lies entirely beyond the batch (ragged last block) has nothing to do -- and the body's
        // wave-uniform base + unsigned lane offset addressing needs the pair's first instance to exist
        const int rr = rev ? ROUNDS - 1 - r : r;
        if ((vb0 + rr) * 2 < nB) {
        // the round's record is always jr[0]: the later ones move up by one afterwards (a handful of register moves per
        // round; indexed by r -- or selected by compares on r, which the compiler turns back into an index -- the array
        // would live in scratch, one scratch round trip per round on the counter the round's loads and stores share)
        T j0[3] = {jr[0][0], jr[0][1], jr[0][2]};
        // (lane index opaque per round: what the body derives from it -- rows, LDS addresses -- is formed again every
        //  round instead of being hoisted out of the loop and spilled across it)
        int lane_r = lane;
        asm volatile("" : "+v"(lane_r));
        // The two wavefronts a block has on a SIMD (w and w + 4) take turns at going first.  Left alone the older one
        // always wins the issue arbitration: stamps of the run kernel showed wavefront 0 through its four rounds in 21 us,
        // wavefront 4 in 27, and wavefront 0 waiting 6 us at the end-of-step barrier (same-box A/B: +1.4 % at B = 65 536,
        // +2.5 % at 8 192).
        if ((r + (wave >> 2)) & 1) __builtin_amdgcn_s_setprio(1);
        cov_rows_body<T, PIN, 32, true, true>(a, prm, smem, vb0 + rr, lane_r, j0, kstep, e_to_lds, &e_r,
                                              /*p_in_lds=*/r == 0 && e_from_lds, /*p_keep=*/r == ROUNDS - 1 && e_to_lds);
        __builtin_amdgcn_s_setprio(0);
        }
#if defined(ESKF_X_RUNSTAMPS) && defined(ESKF_TU_FUSED)
        if (wave == 0) ESKF_RSTAMP(3 + r)
        if (wave == 4) ESKF_RSTAMP(7 + r)
#endif
#pragma unroll
        for (int q = 0; q + 1 < ROUNDS; ++q)
#pragma unroll
            for (int k = 0; k < 3; ++k) jr[q][k] = jr[q + 1][k];
        // (the slot the rotation has freed takes the round's error state: after the last round jr[q][0] is that of the
        //  q-th round in this step's order)
        jr[ROUNDS - 1][0] = e_r;
    }
    if (e_to_lds) {
        __syncthreads();                                // every wavefront is through its rounds: the images are dead -- but for
                                                        // the lower halves of the last round's, which nobody writes from here on
        T* sJ = reinterpret_cast<T*>(smem_raw);
        const int grp = lane >> 5, lig = lane & 31;
        if (lig <= 24) {
#pragma unroll
            for (int q = 0; q < ROUNDS; ++q) {
                const int qq = rev ? ROUNDS - 1 - q : q;
                sJ[staging_row_word<IMG>(wave * 2 * ROUNDS + 2 * qq + grp) + E_ROW + lig] = jr[q][0];
            }
        }
    }
}

// wavefronts per SIMD the one-launch step is made for: fp64 four (128 registers, two blocks per CU -- their LDS); fp32 six
// (80 registers, ten of them spilled: three blocks per CU -- config #3, predict-only launches at B = 1 048 576, 0.589 -> 0.608 of
// the roof, same box twice)
#ifndef ESKF_STEP_WAVES
#define ESKF_STEP_WAVES(T) (sizeof(T) == 4 ? 6 : 4)
#endif
template <typename T, int ROUNDS, bool PIN = false>
__global__ __launch_bounds__(512, ESKF_STEP_WAVES(T)) void eskf_step_kernel(StepArgs a, DevParams<T> prm) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    // (the wavefront's number as a SCALAR: everything derived from it -- the role, the image address, the pair index and
    //  with it the wave-uniform bases of the covariance body's addressing -- stays in scalar registers)
    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    const int lane = (int)threadIdx.x & 63;
    step_block<T, ROUNDS, PIN>(a, prm, smem_raw, wave, lane, -1);
}

// ----------------------------------------------------------------------------------
// eskf_run_kernel: the time loop in the kernel (Filter.run, Filter.py:144-168, for a schedule of one IMU sample per
// camera frame: the 1:1 schedule of the saved trajectory and of the Monte-Carlo batches).  A block keeps ITS instances
// for the whole run and steps them n_steps times: instances are independent, so no block ever waits for another -- the
// only synchronisation is the block's own barrier between a step's covariance rounds (error state, status words) and the
// next step's pre phase, which injects them.  Per step the same code as eskf_step_kernel (same bits as n_steps launches);
// what the run saves on top is the launch boundary per step, and blocks that drift apart run one block's pre phase next
// to its neighbour's rounds -- and the covariances of the sixteen instances a block works on last in a step, which it
// works on first in the next: they stay in the lower halves of the block's exchange images (step_block).  The other 48 of
// a 64-instance block still round-trip memory every step (64 x 2.4 KB per block do not fit a CU; with 16 instances per
// block -- batches up to 8 192 -- no covariance leaves the chip between the first and the last step of a run).
// ----------------------------------------------------------------------------------
// One step of the run.  (Inlined into the time loop -- which is only affordable because this translation unit is
// compiled without MachineLICM, see "host-side launchers": with the pass everything the step materialises is hoisted in
// front of the time loop and spilled across it.  As a real function call the step costs the callee-saved registers'
// round trip through scratch: 100 KB per block and step.)
typedef unsigned char __attribute__((address_space(3))) * lds_bytes_t;
template <typename T, int ROUNDS, bool PIN>
__device__ __forceinline__ void run_one_step(lds_bytes_t smem_lds, const int wave, const int lane, const int k) {
    unsigned char* smem_raw = (unsigned char*)smem_lds;
    StepArgs a;
    DevParams<T> prm;
    load_run_step_args<T>(k, a, prm);
    const RunArgs ra = load_kernarg<RunArgs>((const char __attribute__((address_space(4))) *)__builtin_amdgcn_kernarg_segment_ptr() + offsetof(RunKernArgs<T>, ra));
    step_block<T, ROUNDS, PIN>(a, prm, smem_raw, wave, lane, k, k > 0, k + 1 < ra.n_steps);
}
template <typename T, int ROUNDS, bool PIN = false>
__global__ __launch_bounds__(512, 4) void eskf_run_kernel(StepArgs a0, DevParams<T> prm0, RunArgs ra0) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    [[maybe_unused]] const int lane = (int)threadIdx.x & 63;      // (the stamps of the diagnostic build)
    const int n_steps = ra0.n_steps;
#pragma unroll 1
    for (int k = 0; k < n_steps; ++k) {
        // (the lane index opaque per step: the LDS addresses of the hand-over and of the error-state area are formed again
        //  every step -- hoisted in front of the time loop they are spilled and come back one scratch load after the other)
        // ... and the index itself comes from mbcnt (two instructions, no input register) rather than from the thread
        // index of the kernel's entry, which would be spilled across the step and reloaded -- a scratch round trip in
        // front of everything -- at the top of every one.
        int zero = 0;
        asm volatile("" : "+v"(zero));
        const int lane_k = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, (unsigned)zero));
        run_one_step<T, ROUNDS, PIN>((lds_bytes_t)smem_raw, wave, lane_k, k);
        // error states and status words of this step are visible to the pre phase of the next (and the images are free
        // to become staging rows again)
        __syncthreads();
#if defined(ESKF_X_RUNSTAMPS) && defined(ESKF_TU_FUSED)
        { const int kstep = k; if (wave == 0) ESKF_RSTAMP(11) }
#endif
    }
}

// state.dofs of every instance (Filter.calculate_dof_metric, Filter.py:452-455; the payload of the final gather)
template <typename T>
__global__ __launch_bounds__(256) void calibration_kernel(int64_t B, const T* xs, T* dofs6) {
    const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
#pragma unroll
    for (int i = 0; i < 6; ++i) dofs6[b * 6 + i] = xs[(10 + i) * B + b];
}

// ----------------------------------------------------------------------------------
// state import / export between the user-facing layout and the device records
// ----------------------------------------------------------------------------------
template <typename T>
__global__ void pack_state_kernel(int64_t B, const T* x26, const T* P576, const T* om3, const T* acc3, const T* qo4,
                                  T* xs, T* Ppk) {
    const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (P576) {
        for (int64_t t = tid; t < B * 300; t += (int64_t)gridDim.x * blockDim.x) {
            const int64_t b = t / 300;
            int i, j;
            unpack_ij((int)(t - b * 300), i, j);
            Ppk[t] = P576[b * 576 + i * 24 + j];
        }
    }
    for (int64_t t = tid; t < B * 36; t += (int64_t)gridDim.x * blockDim.x) {
        const int w = (int)(t / B);               // structure-of-arrays record: xs[w * B + b]
        const int64_t b = t - (int64_t)w * B;
        if (w < 26) { if (x26) xs[t] = x26[b * 26 + w]; }
        else if (w < 29) { if (om3) xs[t] = om3[b * 3 + (w - 26)]; }
        else if (w < 32) { if (acc3) xs[t] = acc3[b * 3 + (w - 29)]; }
        else {
            if (qo4) xs[t] = qo4[b * 4 + (w - 32)];
            else if (x26) xs[t] = x26[b * 26 + 6 + (w - 32)];     // R_WB_old = Rm(q) (Filter.py:80)
        }
    }
}

template <typename T>
__global__ void unpack_state_kernel(int64_t B, const T* xs, const T* Ppk, T* x26, T* P576, T* om3, T* acc3, T* qo4,
                                    T* dofs6) {
    const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (P576) {
        for (int64_t t = tid; t < B * 300; t += (int64_t)gridDim.x * blockDim.x) {
            const int64_t b = t / 300;
            int i, j;
            unpack_ij((int)(t - b * 300), i, j);
            const T v = Ppk[t];
            P576[b * 576 + i * 24 + j] = v;
            P576[b * 576 + j * 24 + i] = v;
        }
    }
    for (int64_t t = tid; t < B * 36; t += (int64_t)gridDim.x * blockDim.x) {
        const int w = (int)(t / B);
        const int64_t b = t - (int64_t)w * B;
        const T v = xs[t];
        if (w < 26) { if (x26) x26[b * 26 + w] = v; if (dofs6 && w >= 10 && w < 16) dofs6[b * 6 + (w - 10)] = v; }
        else if (w < 29) { if (om3) om3[b * 3 + (w - 26)] = v; }
        else if (w < 32) { if (acc3) acc3[b * 3 + (w - 29)] = v; }
        else { if (qo4) qo4[b * 4 + (w - 32)] = v; }
    }
}

// ----------------------------------------------------------------------------------
// input synthesis: noise-free IMU sample from the camera trajectory through the probe
// kinematics -- f_imu_meas / f_imu of kinematics/equations.py:8-41,54-69 as driven by
// Imu.eval_expr_single (models/Imu.py:141-196) and Imu.ref_vals (305-315).  Lane per sample.
// Only the notch joint moves (Probe.py:382-410), so relative velocity and acceleration of C in B
// vanish and om_CB = z6 q7', alp_CB = z6 q7''.
// ----------------------------------------------------------------------------------
struct ImuArgs {
    int64_t n;
    const void* R_WC;    // [n][9] row-major
    const void* om_CW;   // [n][3] camera angular rate in W
    const void* acc_CW;  // [n][3] camera acceleration in W
    const void* alp_CW;  // [n][3] camera angular acceleration in W
    const void* notch;   // [n][3] (q7, q7', q7'') or [3]
    int notch_stride;    // 1: per sample, 0: broadcast
    const void* p_CW;    // nullable [n][3]  (reference pose only)
    const void* v_CW;    // nullable [n][3]
    void* om_B;          // [n][3]
    void* acc_B;         // [n][3]
    void* p_BW;          // nullable [n][3]
    void* R_WB;          // nullable [n][9]
    void* v_BW;          // nullable [n][3]
    double dofs[6];      // ground-truth joint values q1..q6
};

template <typename T>
__global__ __launch_bounds__(256) void eskf_imu_kernel(ImuArgs a, DevParams<T> prm) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.n) return;
    const T* gn = reinterpret_cast<const T*>(a.notch) + (a.notch_stride ? i * 3 : 0);
    const T q7 = gn[0], q7d = gn[1], q7dd = gn[2];
    T dofs[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) dofs[k] = (T)a.dofs[k];
    const ProbeTrig<T> t = probe_trig(dofs, q7);
    T pB[3], RBC[9], z6[3];
    probe_pose(t, dofs, prm.L, prm.st, prm.ct, pB, RBC, z6);
    T Rc[9], om_c[3], acc_c[3], alp_c[3];
#pragma unroll
    for (int k = 0; k < 9; ++k) Rc[k] = reinterpret_cast<const T*>(a.R_WC)[i * 9 + k];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        om_c[k] = reinterpret_cast<const T*>(a.om_CW)[i * 3 + k];
        acc_c[k] = reinterpret_cast<const T*>(a.acc_CW)[i * 3 + k];
        alp_c[k] = reinterpret_cast<const T*>(a.alp_CW)[i * 3 + k];
    }
    // R_WB = R_WC R_BC^T   (equations.py:9)
    T R[9];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) R[3 * r + c] = Rc[3 * r] * RBC[3 * c] + Rc[3 * r + 1] * RBC[3 * c + 1] + Rc[3 * r + 2] * RBC[3 * c + 2];
    T omCB[3], alpCB[3], Rom[3], Ralp[3], lever[3], W_om[3], W_alp[3], W_acc[3], cr[3], cr2[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) { omCB[k] = z6[k] * q7d; alpCB[k] = z6[k] * q7dd; }
    mat3_vec(R, omCB, Rom);
    mat3_vec(R, alpCB, Ralp);
    mat3_vec(R, pB, lever);
#pragma unroll
    for (int k = 0; k < 3; ++k) W_om[k] = om_c[k] - Rom[k];                  // equations.py:10
    cross3(W_om, Rom, cr);
#pragma unroll
    for (int k = 0; k < 3; ++k) W_alp[k] = alp_c[k] - Ralp[k] - cr[k];       // equations.py:24-27
    T om_x_lever[3];
    cross3(W_om, lever, om_x_lever);
    cross3(W_alp, lever, cr);
    cross3(W_om, om_x_lever, cr2);
#pragma unroll
    for (int k = 0; k < 3; ++k) W_acc[k] = acc_c[k] - cr[k] - cr2[k];        // equations.py:29-37 (v_CB = acc_CB = 0)
    T oB[3], aB[3];
    mat3T_vec(R, W_om, oB);
    mat3T_vec(R, W_acc, aB);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        reinterpret_cast<T*>(a.om_B)[i * 3 + k] = oB[k];
        reinterpret_cast<T*>(a.acc_B)[i * 3 + k] = aB[k];
    }
    if (a.R_WB) {
#pragma unroll
        for (int k = 0; k < 9; ++k) reinterpret_cast<T*>(a.R_WB)[i * 9 + k] = R[k];
    }
    if (a.p_BW && a.p_CW) {                                                  // equations.py:11
#pragma unroll
        for (int k = 0; k < 3; ++k) reinterpret_cast<T*>(a.p_BW)[i * 3 + k] = reinterpret_cast<const T*>(a.p_CW)[i * 3 + k] - lever[k];
    }
    if (a.v_BW && a.v_CW) {                                                  // equations.py:12-13
#pragma unroll
        for (int k = 0; k < 3; ++k) reinterpret_cast<T*>(a.v_BW)[i * 3 + k] = reinterpret_cast<const T*>(a.v_CW)[i * 3 + k] - om_x_lever[k];
    }
}

// ----------------------------------------------------------------------------------
// trajectory log: the 30-column row of trajectory/FilterTraj.py:12-41 per instance,
// straight from the state record: t | p v euler_xyz_deg(q) q_wxyz | deg(dofs[0:3]) dofs[3:6] |
// p_cam euler_xyz_deg(q_cam) q_cam_wxyz.  Euler angles as scipy's as_euler('xyz') (extrinsic).
// ----------------------------------------------------------------------------------
__device__ __forceinline__ double t_atan2(double y, double x) { return atan2(y, x); }
__device__ __forceinline__ float t_atan2(float y, float x) { return atan2f(y, x); }

template <typename T>
__device__ __forceinline__ void euler_xyz_deg(const T* q, T* e) {
    T M[9];
    quat_to_mat(q, M);
    const T k = T(57.29577951308232087679815481410517);
    T sb = M[6];
    sb = sb > T(1) ? T(1) : (sb < T(-1) ? T(-1) : sb);
    e[0] = k * t_atan2(M[7], M[8]);
    e[1] = -k * t_asin(sb);
    e[2] = k * t_atan2(M[3], M[0]);
}

template <typename T>
__global__ __launch_bounds__(256) void eskf_log_rows_kernel(int64_t B, int64_t n_log, const T* xs, T tstamp, T* rows) {
    const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;      // (the rows of the batch's first n_log instances)
    if (b >= n_log) return;
    State<T> s;
    load_state(xs, B, b, s);
    T* r = rows + b * 30;
    const T k = T(57.29577951308232087679815481410517);
    T e[3];
    r[0] = tstamp;
#pragma unroll
    for (int i = 0; i < 3; ++i) { r[1 + i] = s.p[i]; r[4 + i] = s.v[i]; r[20 + i] = s.pc[i]; }
    euler_xyz_deg(s.q, e);
#pragma unroll
    for (int i = 0; i < 3; ++i) r[7 + i] = e[i];
    r[10] = s.q[3]; r[11] = s.q[0]; r[12] = s.q[1]; r[13] = s.q[2];
#pragma unroll
    for (int i = 0; i < 3; ++i) { r[14 + i] = k * s.dofs[i]; r[17 + i] = s.dofs[3 + i]; }
    euler_xyz_deg(s.qc, e);
#pragma unroll
    for (int i = 0; i < 3; ++i) r[23 + i] = e[i];
    r[26] = s.qc[3]; r[27] = s.qc[0]; r[28] = s.qc[1]; r[29] = s.qc[2];
}

// Filter.calculate_update_mse (Filter.py:397-418) as a reduction over the log row of every instance: the
// mean of the 12 squared differences between the camera columns xc yc zc rx_degc ry_degc rz_degc of the row
// and the camera frame's x y z rx_deg ry_deg rz_deg, and between the IMU columns vx vy vz rx ry rz of the row
// and the last IMU reference sample.  ref[0:6] = camera reference, ref[6:12] = IMU reference (shared by the
// batch; has_imu = 0: the IMU half is left out of the sum, as when no reference trajectory is kept).
struct MseRef { double v[12]; int has_imu; };
template <typename T>
__global__ __launch_bounds__(256) void eskf_update_mse_kernel(int64_t B, const T* xs, MseRef ref, T* mse) {
    const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    State<T> s;
    load_state(xs, B, b, s);
    T e[3], ec[3];
    euler_xyz_deg(s.q, e);
    euler_xyz_deg(s.qc, ec);
    T sum = T(0);
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const T d0 = T(ref.v[i]) - s.pc[i], d1 = T(ref.v[3 + i]) - ec[i];
        sum += d0 * d0 + d1 * d1;
    }
    if (ref.has_imu) {
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const T d0 = s.v[i] - T(ref.v[6 + i]), d1 = e[i] - T(ref.v[9 + i]);
            sum += d0 * d0 + d1 * d1;
        }
    }
    mse[b] = sum / T(12);
}

// ----------------------------------------------------------------------------------
// Filter.Fx / Filter.Fi (Filter.py:249-268,279-285): the dense matrices of one predict, materialised
// from its full record (85 block words) -- the kernels themselves never form them.  One thread per
// (instance, row).  Layout of the camera rows incl. the reference's column cursor (SURVEY.md App. A.3,
// Q1): row 18+r: dt at col 3+r | Bm[r] at 6:9 | Cm[r] at 9:15 | 1 at col 16+r; row 21+r: Dm[r][0:3] at
// 9:12 | Dm[r][3] at col 15 | Em[r] at 19:22 | the 1 of I_24 that survives at (22,22), (23,23).
// ----------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void eskf_jacobians_kernel(int64_t B, const T* __restrict__ J, T* __restrict__ Fx,
                                                             T* __restrict__ Fi, int quirks) {
    const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (tid >= B * 24) return;
    const int64_t b = tid / 24;
    const int r = (int)(tid - b * 24);
    const T* j = J + b * JW;
    const T dt = j[J_DT];
    T fx[24], fi[13];
#pragma unroll
    for (int c = 0; c < 24; ++c) fx[c] = T(0);
#pragma unroll
    for (int c = 0; c < 13; ++c) fi[c] = T(0);
    if (r < 18) {
        if (r < 6 || r > 8) fx[r] = T(1);
        if (r < 3) fx[3 + r] = dt;                                             // Filter.py:253
        else if (r < 6) { for (int k = 0; k < 3; ++k) fx[6 + k] = j[J_A + 3 * (r - 3) + k]; }    // Filter.py:254
        else if (r < 9) { for (int k = 0; k < 3; ++k) fx[6 + k] = j[J_TH + 3 * (r - 6) + k]; }   // Filter.py:255
        else if (r == 15 || r == 16) fx[r + 1] += dt;                          // Filter.py:256-257
        if (r >= 3 && r < 15) fi[r - 3] = T(1);                                // Filter.py:263
        if (r == 17) fi[12] = T(1);                                            // Filter.py:264
    } else if (r < 21) {
        const int q = r - 18;
        fx[3 + q] = dt;
        for (int k = 0; k < 3; ++k) fx[6 + k] = j[J_BM + 3 * q + k];
        for (int k = 0; k < 6; ++k) fx[9 + k] = j[J_CM + 6 * q + k];
        fx[((quirks & QK_Q1) ? 16 : 18) + q] = T(1);
        for (int k = 0; k < 3; ++k) fi[3 + k] = j[J_N6 + 3 * q + k];
    } else {
        const int q = r - 21;
        for (int k = 0; k < 3; ++k) fx[9 + k] = j[J_DM + 4 * q + k];
        fx[15] = j[J_DM + 4 * q + 3];
        if (quirks & QK_Q1) {
            for (int k = 0; k < 3; ++k) fx[19 + k] = j[J_EM + 3 * q + k];
            if (q > 0) fx[21 + q] = T(1);
        } else {
            for (int k = 0; k < 3; ++k) fx[21 + k] = j[J_EM + 3 * q + k];
        }
        for (int k = 0; k < 3; ++k) fi[3 + k] = j[J_N6 + 9 + 3 * q + k];
    }
    T* ox = Fx + (b * 24 + r) * 24;
    T* oi = Fi + (b * 24 + r) * 13;
#pragma unroll
    for (int c = 0; c < 24; ++c) ox[c] = fx[c];
#pragma unroll
    for (int c = 0; c < 13; ++c) oi[c] = fi[c];
}

// ----------------------------------------------------------------------------------
// host-side launchers
//
// This file is compiled TWICE (Makefile): as it is -- every kernel and launcher except the one-launch step and the run
// kernel -- and with -DESKF_TU_FUSED -mllvm -disable-machine-licm, which leaves only those two and their launchers.  In
// eskf_step_kernel / eskf_run_kernel the covariance body sits inside a loop over rounds (and that inside the time loop):
// MachineLICM hoists every constant the body materialises -- the polynomial coefficients of sincos / asin, the lane-mask
// constants, forty vector and sixty scalar registers of them -- in front of the loop, keeps them live across it and
// spills what does not fit (run kernel: 73 vector / 160 scalar registers spilled with it, 21 / 0 without; a step of
// the run was slower than a launch per step).  The other kernels keep the pass: their loops over the predicts of an
// epoch are what it is for.
// A third translation unit, eskf_epochs.hip (the long-horizon kernel), includes this file with ESKF_TU_EPOCHS defined
// for its device functions only: no launcher of this file is compiled there.
// ----------------------------------------------------------------------------------
#if !defined(ESKF_TU_FUSED) && !defined(ESKF_TU_EPOCHS)
template <typename T, int LPI, bool PIN = false>
static hipError_t launch_cov_t(const StepArgs& a, const DevParams<T>& prm, hipStream_t stream) {
    constexpr int G = 64 / LPI;
    if (a.sub_count) return hipErrorInvalidValue;          // (the LDS-image kernel covers the whole batch)
    if (!(prm.quirks & QK_Q1) && a.n_pred > 0) return hipErrorInvalidValue;     // (it only knows the v1 column layout of the camera rows)
    // rounds per wave: enough waves to keep every CU's LDS full several times over, few enough
    // that most HBM round trips hide behind the previous instance's arithmetic
    StepArgs b = a;
    int64_t blocks = (a.B + G - 1) / G;
    int rounds = 1;
    while (rounds < 4 && blocks / (rounds * 2) >= 256 * 16) rounds *= 2;
    b.rounds = rounds;
    blocks = (a.B + (int64_t)G * rounds - 1) / ((int64_t)G * rounds);
    if (blocks > 0x7fffffff) return hipErrorInvalidValue;
    hipLaunchKernelGGL((eskf_cov_kernel<T, LPI, PIN>), dim3((unsigned)blocks), dim3(64), 0, stream, b, prm);
    return hipGetLastError();
}

template <typename T, bool PIN = false, int LPI = 32>
static hipError_t launch_cov_rows(const StepArgs& a, const DevParams<T>& prm, hipStream_t stream) {
    constexpr int G = LPI == 32 ? 2 : 8;                    // instances per block
    if (a.B >= 0x7fffffffLL) return hipErrorInvalidValue;   // 32-bit instance indices in the kernel
    if (a.sub_count && LPI != 32) return hipErrorInvalidValue;
    const int64_t blocks = ((a.sub_count ? a.sub_count : a.B) + G - 1) / G;
    if constexpr (LPI == 32) {
        if (!prm.q_gyro) {      // no gyro block in Q (the normal case): the four-waves-per-SIMD build of the same kernel
            if constexpr (!PIN) {
                if (!(prm.quirks & QK_Q1) && a.n_pred > 0) {    // the camera rows' columns where their variables live (quirk Q1 off)
                    hipLaunchKernelGGL((eskf_cov_rows_kernel<T, false, 32, true, false>), dim3((unsigned)blocks), dim3(64), 0, stream, a, prm);
                    return hipGetLastError();
                }
            }
            if (!(prm.quirks & QK_Q1) && a.n_pred > 0) return hipErrorInvalidValue;
            hipLaunchKernelGGL((eskf_cov_rows_kernel<T, PIN, 32, true>), dim3((unsigned)blocks), dim3(64), 0, stream, a, prm);
            return hipGetLastError();
        }
    }
    if (!(prm.quirks & QK_Q1) && a.n_pred > 0) return hipErrorInvalidValue;     // (the other builds only know the v1 column layout)
    hipLaunchKernelGGL((eskf_cov_rows_kernel<T, PIN, LPI>), dim3((unsigned)blocks), dim3(LPI == 32 ? 64 : 192), 0, stream, a, prm);
    return hipGetLastError();
}

#endif  // !ESKF_TU_FUSED && !ESKF_TU_EPOCHS

#ifndef ESKF_TU_EPOCHS
// diagnostic switch (environment DVI_ESKF_PRE_LEGACY=1, read once): single-predict launches through the two-role
// pre kernel instead of eskf_pre4_kernel -- the A/B of the two and the bitwise test of one against the other
static const bool g_pre_legacy = []() { const char* e = getenv("DVI_ESKF_PRE_LEGACY"); return e && e[0] == '1'; }();
#endif

#if defined(ESKF_TU_EPOCHS)
// (device functions only)
#elif defined(ESKF_TU_FUSED)

// The one-launch step (eskf_step_kernel).  DVI_ESKF_FUSED=0 switches it off (A/B against the two kernels, and the bitwise
// test of one path against the other); DVI_ESKF_FUSED_ROUNDS=1..4 pins the block size (16 R instances).  Default: the
// block size whose LAYERS of blocks cost least (pick_rounds).  A CU holds two blocks; a launch of n x 256 blocks is n
// layers of one block per CU, which a CU works off in pairs.  T2[R] = what a pair of co-resident blocks takes for one
// step, T1[R] = what a block takes that has its CU to itself (us; tools/sweep_run_rounds.sh at the batches that are
// exactly one and two layers; only the ratios matter) -- with these eight numbers the model picks the measured best of
// R = 1..4 at 26 of the 28 points of that sweep (B = 4 096 ... 131 072, both kernels) and is within 1 % of it at the other
// two.  What it captures is the quantisation: 48-instance blocks at B = 24 576 are exactly two layers (8.2e8 pair-steps/s
// in a run against 7.7 / 7.1 for 32 / 64), at 32 768 they are 2.7 (7.5 against 8.2 for 64).
// (The first complete build of the kernel lost to the two kernels between 32 768 and 65 536 -- a launch there is two
// generations of blocks that are all in their pre phase at the same time -- and won below and above; what turned the middle
// around were the code-generation fixes of EXPERIMENTS.md, "Round 4": no FLAT accesses, no spilled scalar registers.)
static int pick_rounds(int64_t B, const float (&T1)[4], const float (&T2)[4]) {
    static const int n_cu = []() {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        return n;
    }();
    int best = 4;
    double best_t = 0;
    for (int R = 4; R >= 1; --R) {          // (ties go to the larger block)
        const int64_t blocks = (B + 16 * R - 1) / (16 * R);
        const int64_t full = blocks / n_cu, rem = blocks - full * n_cu;
        double t = (double)(full / 2) * T2[R - 1] + (double)(full % 2) * T1[R - 1];
        if (rem) t += (full % 2) ? T2[R - 1] - T1[R - 1] : T1[R - 1];      // (a partial layer pairs up with an unpaired full one)
        if (R == 4 || t < best_t * 0.99) { best = R; best_t = t; }
    }
    return best;
}
static const int g_fused = []() { const char* e = getenv("DVI_ESKF_FUSED"); return e ? atoi(e) : -1; }();     // -1: by batch
static const int g_fused_rounds = []() { const char* e = getenv("DVI_ESKF_FUSED_ROUNDS"); return e ? atoi(e) : 0; }();
static int step_kernel_rounds(int64_t B, bool pin) {          // 0: the two kernels
    if (g_fused == 0) return 0;
    if (g_fused_rounds >= 1 && g_fused_rounds <= 4) return g_fused_rounds;
    if (pin && B <= 12288) return 1;        // (per-instance noise records: the 16-instance blocks only)
    static const float T1[4] = {12.0f, 17.4f, 24.1f, 29.9f}, T2[4] = {17.5f, 26.9f, 36.5f, 46.6f};
    return pick_rounds(B, T1, T2);
}
// (the kernel asks for 75 KB of LDS per block in fp64: checked once per device against what a block may have)
static bool device_lds_fits(size_t bytes) {
    int dev = 0, lim = 0;
    if (hipGetDevice(&dev) != hipSuccess) return false;
    if (hipDeviceGetAttribute(&lim, hipDeviceAttributeMaxSharedMemoryPerBlock, dev) != hipSuccess) return false;
    return (size_t)lim >= bytes;
}
template <typename T>
static bool step_kernel_covers(int lpi, const StepArgs& a, const DevParams<T>& prm) {
    static const bool lds_ok = device_lds_fits((size_t)8 * 2 * 664 * sizeof(double));
    // (per-instance noise: the 16-instance blocks only -- a population is a small batch, and the 85 KB images are one block per CU)
    const int rounds = step_kernel_rounds(a.B, a.noise != nullptr);
    return lds_ok && rounds != 0 && !g_pre_legacy && a.n_pred == 1 && lpi == 1 && !a.K_out &&
           (!a.noise || rounds == 1) && !a.sub_count && !prm.jfull && !prm.q_gyro && (prm.quirks & QK_Q1) && a.B > 0 && a.B < 0x7fffffffLL;
}
template <typename T>
static hipError_t launch_step_kernel(const StepArgs& a, const DevParams<T>& prm, hipStream_t stream) {
    const int rounds = step_kernel_rounds(a.B, a.noise != nullptr);
    const size_t lds = (size_t)8 * 2 * (a.noise ? 664 : 600) * sizeof(T);     // 76 800 B (fp64): two blocks per CU
    const unsigned blocks = (unsigned)((a.B + 16 * rounds - 1) / (16 * rounds));
    // (DVI_ESKF_VERBOSE=1: one line on stderr the first time this path is taken -- the tests of the one-launch step
    //  make sure with it that they are not comparing the two-kernel path with itself)
    static const bool verbose = []() { const char* e = getenv("DVI_ESKF_VERBOSE"); return e && e[0] == '1'; }();
    static bool said = false;
    if (verbose && !said) {
        said = true;
        fprintf(stderr, "dvi_eskf: single-predict launches go through eskf_step_kernel (%d instances per block)\n", 16 * rounds);
    }
    if (a.noise) hipLaunchKernelGGL((eskf_step_kernel<T, 1, true>), dim3(blocks), dim3(512), lds, stream, a, prm);
    else if (rounds == 4) hipLaunchKernelGGL((eskf_step_kernel<T, 4>), dim3(blocks), dim3(512), lds, stream, a, prm);
    else if (rounds == 3) hipLaunchKernelGGL((eskf_step_kernel<T, 3>), dim3(blocks), dim3(512), lds, stream, a, prm);
    else if (rounds == 2) hipLaunchKernelGGL((eskf_step_kernel<T, 2>), dim3(blocks), dim3(512), lds, stream, a, prm);
    else hipLaunchKernelGGL((eskf_step_kernel<T, 1>), dim3(blocks), dim3(512), lds, stream, a, prm);
    return hipGetLastError();
}

// The run kernel takes what the one-launch step takes; its block size follows the same table unless DVI_ESKF_RUN_ROUNDS
// pins it (DVI_ESKF_RUN=0: runs are enqueued step by step)
static const int g_run = []() { const char* e = getenv("DVI_ESKF_RUN"); return e ? atoi(e) : -1; }();
static const int g_run_rounds = []() { const char* e = getenv("DVI_ESKF_RUN_ROUNDS"); return e ? atoi(e) : 0; }();
// (the layer model of pick_rounds with the run kernel's own figures: a quarter / a third / half / all of the covariances
//  resident between the steps with 64 / 48 / 32 / 16 instances per block -- sixteen-instance blocks are taken while they are one
//  generation, B <= 8 192: 6.9e8 pair-steps/s, and stop at 7.0e8 however large the batch, a pre phase per round)
static int run_kernel_rounds(int64_t B, bool pin) {
    if (g_run == 0) return 0;
    if (g_run_rounds >= 1 && g_run_rounds <= 4) return g_run_rounds;
    if (pin && B <= 8192) return 1;
    static const float T1[4] = {8.9f, 14.1f, 19.3f, 24.0f}, T2[4] = {11.8f, 20.5f, 30.1f, 40.1f};
    return pick_rounds(B, T1, T2);
}
template <typename T>
static int run_kernel_block_typed(int lpi, const StepArgs& a, const DevParams<T>& prm) {
    static const bool lds_ok = device_lds_fits((size_t)8 * 2 * 664 * sizeof(double));
    const int rounds = run_kernel_rounds(a.B, a.noise != nullptr);
    const bool ok = lds_ok && rounds != 0 && !g_pre_legacy && a.n_pred == 1 && lpi == 1 && (!a.noise || rounds == 1) && !a.K_out &&
                    !a.sub_count && !prm.jfull && !prm.q_gyro && (prm.quirks & QK_Q1) && a.B > 0 && a.B < 0x7fffffffLL;
    return ok ? 16 * rounds : 0;
}
int run_kernel_block(int dtype, int lpi, const StepArgs& a, const HostParams& hp) {
    if (dtype == 0) return run_kernel_block_typed<double>(lpi, a, make_dev_params<double>(hp));
    return run_kernel_block_typed<float>(lpi, a, make_dev_params<float>(hp));
}
template <typename T>
static hipError_t launch_run_typed(int lpi, const StepArgs& a, const RunArgs& ra, const DevParams<T>& prm, hipStream_t stream) {
    const int ni = run_kernel_block_typed<T>(lpi, a, prm);
    if (!ni || ra.n_steps < 1) return hipErrorInvalidValue;
    const RunArgs& ra_l = ra;
    const size_t lds = (size_t)8 * 2 * (a.noise ? 664 : 600) * sizeof(T);
    const unsigned blocks = (unsigned)((a.B + ni - 1) / ni);
    static const bool verbose = []() { const char* e = getenv("DVI_ESKF_VERBOSE"); return e && e[0] == '1'; }();
    static bool said = false;
    if (verbose && !said) {
        said = true;
        fprintf(stderr, "dvi_eskf: runs of pair steps go through eskf_run_kernel (%d instances per block)\n", ni);
    }
    if (a.noise) hipLaunchKernelGGL((eskf_run_kernel<T, 1, true>), dim3(blocks), dim3(512), lds, stream, a, prm, ra_l);
    else if (ni == 64) hipLaunchKernelGGL((eskf_run_kernel<T, 4>), dim3(blocks), dim3(512), lds, stream, a, prm, ra_l);
    else if (ni == 48) hipLaunchKernelGGL((eskf_run_kernel<T, 3>), dim3(blocks), dim3(512), lds, stream, a, prm, ra_l);
    else if (ni == 32) hipLaunchKernelGGL((eskf_run_kernel<T, 2>), dim3(blocks), dim3(512), lds, stream, a, prm, ra_l);
    else hipLaunchKernelGGL((eskf_run_kernel<T, 1>), dim3(blocks), dim3(512), lds, stream, a, prm, ra_l);
    return hipGetLastError();
}
hipError_t launch_run(int dtype, int lpi, const StepArgs& a, const RunArgs& ra, const HostParams& hp, hipStream_t stream) {
    if (dtype == 0) return launch_run_typed<double>(lpi, a, ra, make_dev_params<double>(hp), stream);
    return launch_run_typed<float>(lpi, a, ra, make_dev_params<float>(hp), stream);
}
int step_kernel_block(int dtype, int lpi, const StepArgs& a, const HostParams& hp) {
    const bool ok = dtype == 0 ? step_kernel_covers(lpi, a, make_dev_params<double>(hp)) : step_kernel_covers(lpi, a, make_dev_params<float>(hp));
    return ok ? 16 * step_kernel_rounds(a.B, a.noise != nullptr) : 0;
}
hipError_t launch_step_fused(int dtype, const StepArgs& a, const HostParams& hp, hipStream_t stream) {
    if (dtype == 0) return launch_step_kernel<double>(a, make_dev_params<double>(hp), stream);
    return launch_step_kernel<float>(a, make_dev_params<float>(hp), stream);
}

#else   // !ESKF_TU_FUSED

// largest batch whose epochs go through eskf_nominal_kernel + eskf_jac_kernel (DVI_ESKF_EPOCH_SPLIT_MAX; -1 = always)
static const int64_t g_epoch_split_max = []() { const char* e = getenv("DVI_ESKF_EPOCH_SPLIT_MAX"); return e ? (int64_t)atoll(e) : (int64_t)40960; }();

template <typename T>
static hipError_t launch_step_typed(int lpi, const StepArgs& a, const DevParams<T>& prm, hipStream_t stream) {
    if (a.B <= 0) return hipSuccess;
    // a part of the batch: ONE predicate says when that is possible (the host asks it before it splits a run), and every
    // branch below that launches a kernel which ignores sub_first / sub_count refuses such a launch on its own as well
    if (a.sub_count && !epoch_sub_batches_ok(lpi, a.sub_first, a.sub_count, a.n_pred, a.noise != nullptr, a.K_out != nullptr, a.aux != nullptr))
        return hipErrorInvalidValue;
    // 1. lane-per-instance: nominal propagation, Jacobian blocks, residual
    const size_t stage_bytes = (size_t)64 * ((prm.jfull ? JB_FULL : JB_CORE) + 9) * sizeof(T);    // + parked measurement
    if (a.n_pred == 1 && !g_pre_legacy) {
        if (a.sub_count) return hipErrorInvalidValue;      // (eskf_pre4_kernel covers the whole batch)
        const size_t stage4 = (size_t)64 * (prm.jfull ? J_FULL : J_CORE) * sizeof(T);
        hipLaunchKernelGGL((eskf_pre4_kernel<T>), dim3((unsigned)((a.B + 63) / 64)), dim3(256), stage4, stream, a, prm);
    } else if (a.n_pred == 1) {
        if (a.sub_count) return hipErrorInvalidValue;      // (the two-role kernel covers the whole batch)
        hipLaunchKernelGGL((eskf_pre_kernel<T, true>), dim3((unsigned)((a.B + 63) / 64)), dim3(128), stage_bytes, stream, a, prm);
    }
    else if (a.n_pred >= 2 && !g_pre_legacy && a.aux && (a.B <= g_epoch_split_max || g_epoch_split_max < 0 || a.sub_count)) {
        // epochs: the recurrences, then the Jacobian blocks of all predicts at once.  Pays while the batch under-fills
        // the chip for a lane-per-instance kernel (measured, epoch time against the two-role kernel below, fp64 / fp32,
        // 10 and 33 predicts: B = 1 024 x0.50-0.71, 8 192 x0.70-0.81, 16 384 x0.83-0.93, 32 768 x0.97-1.02, 65 536
        // x1.02-1.05: there 1 024 blocks fill every CU, the serial loop is no longer exposed, and the extra launch and
        // the hand-over words cost more than they save)
        const int64_t cnt = a.sub_count ? a.sub_count : a.B;
        const dim3 gb((unsigned)((cnt + 63) / 64));
        if (!a.imu_stride && !a.dt_stride) hipLaunchKernelGGL((eskf_nominal_kernel<T, true>), gb, dim3(128), 0, stream, a, prm);
        else hipLaunchKernelGGL((eskf_nominal_kernel<T, false>), gb, dim3(128), 0, stream, a, prm);
        hipError_t e0 = hipGetLastError();
        if (e0 != hipSuccess) return e0;
        const int64_t njobs = (int64_t)a.n_pred * cnt;
        if ((njobs + 63) / 64 > 0x7fffffffLL) return hipErrorInvalidValue;
        const size_t jl = (size_t)64 * (prm.jfull ? JB_FULL : JB_CORE) * sizeof(T);
        hipLaunchKernelGGL((eskf_jac_kernel<T>), dim3((unsigned)((njobs + 63) / 64)), dim3(128), jl, stream, a, prm);
    } else if (a.sub_count)
        return hipErrorInvalidValue;                       // (the two-role kernel covers the whole batch)
    else if (!a.imu_stride && !a.dt_stride)
        hipLaunchKernelGGL((eskf_pre_kernel<T, false, true>), dim3((unsigned)((a.B + 63) / 64)), dim3(128), stage_bytes, stream, a, prm);
    else
        hipLaunchKernelGGL((eskf_pre_kernel<T, false>), dim3((unsigned)((a.B + 63) / 64)), dim3(128), stage_bytes, stream, a, prm);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    // 2. covariance.  With per-instance noise: the PIN variants (row-resident, or the LDS-image kernel
    // with 32 lanes per instance when the gain matrix is asked for or that kernel is selected)
    if (a.noise) {
        if (a.K_out || (lpi != 1 && lpi != 2)) return launch_cov_t<T, 32, true>(a, prm, stream);
        return lpi == 2 ? launch_cov_rows<T, true, 24>(a, prm, stream) : launch_cov_rows<T, true, 32>(a, prm, stream);
    }
    switch (lpi) {
        case 2:  // DVI_ESKF_LANES_ROWS24
            e = a.K_out ? launch_cov_t<T, 32>(a, prm, stream) : launch_cov_rows<T, false, 24>(a, prm, stream);
            break;
        case 1:  // DVI_ESKF_LANES_ROWS; the gain matrix is only formed by the LDS-image kernel
            e = a.K_out ? launch_cov_t<T, 32>(a, prm, stream) : launch_cov_rows<T>(a, prm, stream);
            break;
        case 32: e = launch_cov_t<T, 32>(a, prm, stream); break;    // the LDS-image kernel (the other lane counts it was
                                                                     // written for -- 64, 24, 16, 12, 8 -- lost to this one and are gone)
        default: return hipErrorInvalidValue;
    }
    if (e != hipSuccess) return e;
    // 3. the error injection is deferred: the next pre kernel (or dvi_eskf_flush) applies it
    return e;
}

template <typename T>
static hipError_t launch_flush_typed(const StepArgs& a, const DevParams<T>& prm, hipStream_t stream) {
    if (a.B <= 0) return hipSuccess;
    hipLaunchKernelGGL(eskf_post_kernel<T>, dim3((unsigned)((a.B + 255) / 256)), dim3(256), 0, stream, a, prm);
    return hipGetLastError();
}

// Whether an epoch launch over the part [sub_first, sub_first + sub_count) of the batch is possible (see StepArgs::sub_first):
// the ONE statement of the conditions -- launch_step_typed refuses what it does not allow, dvi_eskf_run_epochs asks it
// before it puts the halves of a batch on two streams.  n_pred: the launch's predicts (the host passes the shortest
// epoch of the run).
bool epoch_sub_batches_ok(int lpi, int64_t sub_first, int64_t sub_count, int n_pred, bool per_instance_noise, bool gain_out, bool have_aux) {
    return !g_pre_legacy && have_aux && lpi == 1 && !per_instance_noise && !gain_out && n_pred >= 2 && sub_count > 0 &&
           (sub_count & 63) == 0 && (sub_first & 63) == 0;
}

hipError_t launch_flush(int dtype, const StepArgs& a, const HostParams& hp, hipStream_t stream) {
    if (dtype == 0) return launch_flush_typed<double>(a, make_dev_params<double>(hp), stream);
    return launch_flush_typed<float>(a, make_dev_params<float>(hp), stream);
}

hipError_t launch_step(int dtype, int lpi, const StepArgs& a, const HostParams& hp, hipStream_t stream) {
    if (a.B > 0 && step_kernel_block(dtype, lpi, a, hp)) return launch_step_fused(dtype, a, hp, stream);     // the one-launch step
    if (dtype == 0) return launch_step_typed<double>(lpi, a, make_dev_params<double>(hp), stream);
    return launch_step_typed<float>(lpi, a, make_dev_params<float>(hp), stream);
}

// state restore (dvi_eskf_copy_state): both buffers of a handle in one launch of 16-byte moves.  hipMemcpyAsync
// device-to-device moved the 176 MB of a 65 536-instance fp64 state in 0.26 ms; this takes what a copy takes on the
// part (tools/copy_ceiling.py).
__global__ __launch_bounds__(256) void copy_state_kernel(const uint4* __restrict__ sP, uint4* __restrict__ dP, size_t nP,
                                                         const uint4* __restrict__ sx, uint4* __restrict__ dx, size_t nx) {
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nP; i += stride) dP[i] = sP[i];
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nx; i += stride) dx[i] = sx[i];
}
hipError_t launch_copy_state(const void* srcP, void* dstP, size_t bytesP, const void* srcx, void* dstx, size_t bytesx, hipStream_t stream) {
    if ((bytesP | bytesx) & 15) return hipErrorInvalidValue;     // (300 and 36 words per instance: always multiples of 16 bytes)
    const size_t nP = bytesP / 16, nx = bytesx / 16;
    if (nP + nx == 0) return hipSuccess;
    size_t blocks = (nP + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    if (blocks == 0) blocks = 1;
    hipLaunchKernelGGL(copy_state_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, (const uint4*)srcP, (uint4*)dstP, nP,
                       (const uint4*)srcx, (uint4*)dstx, nx);
    return hipGetLastError();
}

// the records of the instances [first, first + dstB) of one handle into another (dvi_eskf_copy_instances)
template <typename T>
__global__ __launch_bounds__(256) void copy_instances_kernel(const T* __restrict__ sP, const T* __restrict__ sx, int64_t sB, int64_t first,
                                                             T* __restrict__ dP, T* __restrict__ dx, int64_t dB) {
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < dB * 300; i += stride) dP[i] = sP[first * 300 + i];
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < dB * 36; i += stride) {
        const int64_t w = i / dB, b = i - w * dB;
        dx[w * dB + b] = sx[w * sB + first + b];
    }
}
hipError_t launch_copy_instances(int dtype, const void* srcP, const void* srcx, int64_t srcB, int64_t first, void* dstP, void* dstx, int64_t dstB,
                                 hipStream_t stream) {
    if (dstB <= 0) return hipSuccess;
    int64_t blocks = (dstB * 300 + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    if (dtype == 0)
        hipLaunchKernelGGL(copy_instances_kernel<double>, dim3((unsigned)blocks), dim3(256), 0, stream, (const double*)srcP, (const double*)srcx, srcB, first,
                           (double*)dstP, (double*)dstx, dstB);
    else
        hipLaunchKernelGGL(copy_instances_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, stream, (const float*)srcP, (const float*)srcx, srcB, first,
                           (float*)dstP, (float*)dstx, dstB);
    return hipGetLastError();
}

hipError_t launch_pack(int dtype, int64_t B, const void* x26, const void* P576, const void* om3, const void* acc3,
                       const void* qo4, void* xs, void* Ppk, hipStream_t stream) {
    const int threads = 256;
    const int64_t work = B * 300;
    const unsigned blocks = (unsigned)((work + threads - 1) / threads > 65535 ? 65535 : (work + threads - 1) / threads);
    if (dtype == 0)
        hipLaunchKernelGGL(pack_state_kernel<double>, dim3(blocks), dim3(threads), 0, stream, B, (const double*)x26,
                           (const double*)P576, (const double*)om3, (const double*)acc3, (const double*)qo4, (double*)xs,
                           (double*)Ppk);
    else
        hipLaunchKernelGGL(pack_state_kernel<float>, dim3(blocks), dim3(threads), 0, stream, B, (const float*)x26,
                           (const float*)P576, (const float*)om3, (const float*)acc3, (const float*)qo4, (float*)xs,
                           (float*)Ppk);
    return hipGetLastError();
}

hipError_t launch_unpack(int dtype, int64_t B, const void* xs, const void* Ppk, void* x26, void* P576, void* om3,
                         void* acc3, void* qo4, void* dofs6, hipStream_t stream) {
    if (dofs6 && !x26 && !P576 && !om3 && !acc3 && !qo4) {      // the calibration estimates only
        const unsigned nb = (unsigned)((B + 255) / 256);
        if (dtype == 0) hipLaunchKernelGGL(calibration_kernel<double>, dim3(nb), dim3(256), 0, stream, B, (const double*)xs, (double*)dofs6);
        else hipLaunchKernelGGL(calibration_kernel<float>, dim3(nb), dim3(256), 0, stream, B, (const float*)xs, (float*)dofs6);
        return hipGetLastError();
    }
    const int threads = 256;
    const int64_t work = B * 300;
    const unsigned blocks = (unsigned)((work + threads - 1) / threads > 65535 ? 65535 : (work + threads - 1) / threads);
    if (dtype == 0)
        hipLaunchKernelGGL(unpack_state_kernel<double>, dim3(blocks), dim3(threads), 0, stream, B, (const double*)xs,
                           (const double*)Ppk, (double*)x26, (double*)P576, (double*)om3, (double*)acc3, (double*)qo4,
                           (double*)dofs6);
    else
        hipLaunchKernelGGL(unpack_state_kernel<float>, dim3(blocks), dim3(threads), 0, stream, B, (const float*)xs,
                           (const float*)Ppk, (float*)x26, (float*)P576, (float*)om3, (float*)acc3, (float*)qo4,
                           (float*)dofs6);
    return hipGetLastError();
}

hipError_t launch_imu(int dtype, int64_t n, const void* R_WC, const void* om, const void* acc, const void* alp,
                      const void* notch, int notch_stride, const void* p_CW, const void* v_CW, void* om_B, void* acc_B,
                      void* p_BW, void* R_WB, void* v_BW, const double* dofs6, const HostParams& hp, hipStream_t stream) {
    if (n <= 0) return hipSuccess;
    ImuArgs a;
    a.n = n; a.R_WC = R_WC; a.om_CW = om; a.acc_CW = acc; a.alp_CW = alp; a.notch = notch; a.notch_stride = notch_stride;
    a.p_CW = p_CW; a.v_CW = v_CW; a.om_B = om_B; a.acc_B = acc_B; a.p_BW = p_BW; a.R_WB = R_WB; a.v_BW = v_BW;
    for (int k = 0; k < 6; ++k) a.dofs[k] = dofs6[k];
    const unsigned blocks = (unsigned)((n + 255) / 256);
    if (dtype == 0)
        hipLaunchKernelGGL(eskf_imu_kernel<double>, dim3(blocks), dim3(256), 0, stream, a, make_dev_params<double>(hp));
    else
        hipLaunchKernelGGL(eskf_imu_kernel<float>, dim3(blocks), dim3(256), 0, stream, a, make_dev_params<float>(hp));
    return hipGetLastError();
}

hipError_t launch_update_mse(int dtype, int64_t B, const void* xs, const double* cam6, const double* imu6, void* mse,
                             hipStream_t stream) {
    if (B <= 0) return hipSuccess;
    MseRef r;
    for (int i = 0; i < 6; ++i) { r.v[i] = cam6[i]; r.v[6 + i] = imu6 ? imu6[i] : 0.0; }
    r.has_imu = imu6 ? 1 : 0;
    const unsigned blocks = (unsigned)((B + 255) / 256);
    if (dtype == 0)
        hipLaunchKernelGGL(eskf_update_mse_kernel<double>, dim3(blocks), dim3(256), 0, stream, B, (const double*)xs, r, (double*)mse);
    else
        hipLaunchKernelGGL(eskf_update_mse_kernel<float>, dim3(blocks), dim3(256), 0, stream, B, (const float*)xs, r, (float*)mse);
    return hipGetLastError();
}

hipError_t launch_jacobians(int dtype, int64_t B, const void* J, int ip, void* Fx, void* Fi, int quirks, hipStream_t stream) {
    if (B <= 0) return hipSuccess;
    const unsigned blocks = (unsigned)((B * 24 + 255) / 256);
    const size_t off = (size_t)ip * (size_t)B * JW;
    if (dtype == 0)
        hipLaunchKernelGGL(eskf_jacobians_kernel<double>, dim3(blocks), dim3(256), 0, stream, B, (const double*)J + off, (double*)Fx, (double*)Fi, quirks);
    else
        hipLaunchKernelGGL(eskf_jacobians_kernel<float>, dim3(blocks), dim3(256), 0, stream, B, (const float*)J + off, (float*)Fx, (float*)Fi, quirks);
    return hipGetLastError();
}

hipError_t launch_log_rows(int dtype, int64_t B, int64_t n_log, const void* xs, double t, void* rows, hipStream_t stream) {
    if (B <= 0 || n_log <= 0) return hipSuccess;
    if (n_log > B) return hipErrorInvalidValue;
    const unsigned blocks = (unsigned)((n_log + 255) / 256);
    if (dtype == 0)
        hipLaunchKernelGGL(eskf_log_rows_kernel<double>, dim3(blocks), dim3(256), 0, stream, B, n_log, (const double*)xs, t, (double*)rows);
    else
        hipLaunchKernelGGL(eskf_log_rows_kernel<float>, dim3(blocks), dim3(256), 0, stream, B, n_log, (const float*)xs, (float)t, (float*)rows);
    return hipGetLastError();
}

#endif  // !ESKF_TU_FUSED

}  // namespace eskf
