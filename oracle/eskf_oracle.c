/* CPU oracle of the ESKF predict/update hot path -- TEST INFRASTRUCTURE ONLY.
 * See eskf_oracle.h for scope and pinning.
 *
 * The camera rows of Fx / Fi are produced the way the reference produces them:
 * the reference differentiates its own expressions err_p_C_next /
 * err_theta_C_next with casadi.jacobian (Filter.py:270-342); here the same
 * expressions (symbols.py:129-200, Probe.py:147-167,456-468) are evaluated on
 * forward-mode dual numbers.  Dense matrix products are plain triple loops in
 * the reference's association order.
 */
#include "eskf_oracle.h"

#include <float.h>
#include <math.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* Scalar type of the restatement.  double: the oracle.  __float128 (-DESKF_ORACLE_QUAD, libquadmath):
 * the extended-precision ADJUDICATOR build of the very same code -- every intermediate (duals, matrix
 * products, the LU inverse of S, gain, Joseph form, reset) in quad precision, inputs and outputs
 * doubles -- which says, where the fp64 arithmetic of the reference is itself inaccurate (cond(S) up
 * to 1e10), how far each fp64 result is from the exact one.  Constants that the reference takes from
 * float64 (4 * eps in Quaternion.about_axis) keep their float64 value. */
#ifdef ESKF_ORACLE_QUAD
#include <quadmath.h>
typedef __float128 real;
#define r_sqrt sqrtq
#define r_sin sinq
#define r_cos cosq
#define r_asin asinq
#define r_fabs fabsq
#define r_isfinite(x) finiteq(x)
#define R_PI M_PIq
#else
typedef double real;
#define r_sqrt sqrt
#define r_sin sin
#define r_cos cos
#define r_asin asin
#define r_fabs fabs
#define r_isfinite(x) isfinite(x)
#define R_PI M_PI
#endif

/* ------------------------------------------------------------------------- */
/* small helpers (tools/math.py:29-39, tools/Quaternion.py)                   */
/* ------------------------------------------------------------------------- */
static void skew3(const real x[3], real S[9]) {
    S[0] = 0;     S[1] = -x[2]; S[2] = x[1];
    S[3] = x[2];  S[4] = 0;     S[5] = -x[0];
    S[6] = -x[1]; S[7] = x[0];  S[8] = 0;
}
static void mat3_mul(const real A[9], const real B[9], real C[9]) {
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            real s = 0;
            for (int k = 0; k < 3; ++k) s += A[3 * i + k] * B[3 * k + j];
            C[3 * i + j] = s;
        }
}
static void mat3_vec(const real A[9], const real x[3], real y[3]) {
    for (int i = 0; i < 3; ++i) y[i] = A[3 * i] * x[0] + A[3 * i + 1] * x[1] + A[3 * i + 2] * x[2];
}
static void mat3T_vec(const real A[9], const real x[3], real y[3]) {
    for (int i = 0; i < 3; ++i) y[i] = A[i] * x[0] + A[3 + i] * x[1] + A[6 + i] * x[2];
}
static void cross3(const real a[3], const real b[3], real c[3]) {
    c[0] = a[1] * b[2] - a[2] * b[1];
    c[1] = a[2] * b[0] - a[0] * b[2];
    c[2] = a[0] * b[1] - a[1] * b[0];
}

/* Hamilton product, xyzw (Quaternion.py:170-173) */
static void quat_mul(const real a[4], const real b[4], real o[4]) {
    real w = a[3] * b[3] - (a[0] * b[0] + a[1] * b[1] + a[2] * b[2]);
    real c[3];
    cross3(a, b, c);
    real v0 = a[3] * b[0] + b[3] * a[0] + c[0];
    real v1 = a[3] * b[1] + b[3] * a[1] + c[1];
    real v2 = a[3] * b[2] + b[3] * a[2] + c[2];
    o[0] = v0; o[1] = v1; o[2] = v2; o[3] = w;
}
/* Quaternion.normalise (Quaternion.py:195-206) */
static void quat_nrm(real q[4]) {
    real d = r_sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    for (int i = 0; i < 4; ++i) q[i] /= d;
    if (q[3] < 0)
        for (int i = 0; i < 4; ++i) q[i] = -q[i];
}
/* Quaternion.rot (Quaternion.py:101-107): scipy from_quat normalises, then as_matrix */
static void quat_to_mat(const real qq[4], real M[9]) {
    real n = r_sqrt(qq[0] * qq[0] + qq[1] * qq[1] + qq[2] * qq[2] + qq[3] * qq[3]);
    real x = qq[0] / n, y = qq[1] / n, z = qq[2] / n, w = qq[3] / n;
    real x2 = x * x, y2 = y * y, z2 = z * z, w2 = w * w;
    real xy = x * y, zw = z * w, xz = x * z, yw = y * w, yz = y * z, xw = x * w;
    M[0] = x2 - y2 - z2 + w2; M[1] = 2 * (xy - zw);      M[2] = 2 * (xz + yw);
    M[3] = 2 * (xy + zw);     M[4] = -x2 + y2 - z2 + w2; M[5] = 2 * (yz - xw);
    M[6] = 2 * (xz - yw);     M[7] = 2 * (yz + xw);      M[8] = -x2 - y2 + z2 + w2;
}
/* scipy 1.10.1 Rotation.from_matrix: Markley on M as given, then unit length;
 * followed by Quaternion.normalise (Quaternion.py:45-46,69) */
static void quat_from_mat(const real M[9], real q[4]) {
    real d[4] = {M[0], M[4], M[8], M[0] + M[4] + M[8]};
    int c = 0;
    for (int i = 1; i < 4; ++i)
        if (d[i] > d[c]) c = i;
    if (c != 3) {
        int i = c, j = (i + 1) % 3, k = (j + 1) % 3;
        q[i] = 1 - d[3] + 2 * M[3 * i + i];
        q[j] = M[3 * j + i] + M[3 * i + j];
        q[k] = M[3 * k + i] + M[3 * i + k];
        q[3] = M[3 * k + j] - M[3 * j + k];
    } else {
        q[0] = M[7] - M[5];
        q[1] = M[2] - M[6];
        q[2] = M[3] - M[1];
        q[3] = 1 + d[3];
    }
    real n = r_sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    for (int i = 0; i < 4; ++i) q[i] /= n;
    quat_nrm(q);
}
/* Quaternion.about_axis (Quaternion.py:209-224) */
static void quat_about_axis(real angle, const real axis[3], real q[4]) {
    real v[3] = {axis[0], axis[1], axis[2]};
    real qlen = r_sqrt(0.0 * 0.0 + v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
    if (qlen > DBL_EPSILON * 4.0) {
        real s = r_sin(angle / 2.0) / qlen;
        v[0] *= s; v[1] *= s; v[2] *= s;
    }
    q[0] = v[0]; q[1] = v[1]; q[2] = v[2]; q[3] = r_cos(angle / 2.0);
    quat_nrm(q);
}

/* ------------------------------------------------------------------------- */
/* forward-mode duals                                                         */
/* variables: err_p_B 0:3 | err_v_B 3:6 | err_theta 6:9 | err_q1..7 9:16 |     */
/*            err_p_C 16:19 | err_theta_C 19:22 | n_a 22:25 | n_om 25:28 |     */
/*            n_dofs 28:34 | n_notch_acc 34     (symbols.py:107,124)           */
/* ------------------------------------------------------------------------- */
#define ND 35
typedef struct { real v; real d[ND]; } dual;

static dual d_const(real c) { dual r; r.v = c; memset(r.d, 0, sizeof r.d); return r; }
static dual d_var(real v, int idx) { dual r = d_const(v); r.d[idx] = 1.0; return r; }
static dual d_add(dual a, dual b) { for (int i = 0; i < ND; ++i) a.d[i] += b.d[i]; a.v += b.v; return a; }
static dual d_sub(dual a, dual b) { for (int i = 0; i < ND; ++i) a.d[i] -= b.d[i]; a.v -= b.v; return a; }
static dual d_mul(dual a, dual b) {
    dual r;
    r.v = a.v * b.v;
    for (int i = 0; i < ND; ++i) r.d[i] = a.d[i] * b.v + a.v * b.d[i];
    return r;
}
static dual d_scale(dual a, real s) { for (int i = 0; i < ND; ++i) a.d[i] *= s; a.v *= s; return a; }
static dual d_sin(dual a) { real c = r_cos(a.v); dual r; r.v = r_sin(a.v); for (int i = 0; i < ND; ++i) r.d[i] = c * a.d[i]; return r; }
static dual d_cos(dual a) { real s = -r_sin(a.v); dual r; r.v = r_cos(a.v); for (int i = 0; i < ND; ++i) r.d[i] = s * a.d[i]; return r; }

static void dv_cross(const dual a[3], const dual b[3], dual c[3]) {
    c[0] = d_sub(d_mul(a[1], b[2]), d_mul(a[2], b[1]));
    c[1] = d_sub(d_mul(a[2], b[0]), d_mul(a[0], b[2]));
    c[2] = d_sub(d_mul(a[0], b[1]), d_mul(a[1], b[0]));
}

/* T <- T * DH(theta, d, a=0, alpha); T is 3x4 (last row implicit 0 0 0 1) */
static void chain_step(dual T[3][4], dual theta, dual dd, real alpha) {
    dual ct = d_cos(theta), st = d_sin(theta);
    real ca = r_cos(alpha), sa = r_sin(alpha);
    dual A[3][4];
    A[0][0] = ct; A[0][1] = d_scale(st, -ca); A[0][2] = d_scale(st, sa);  A[0][3] = d_const(0);
    A[1][0] = st; A[1][1] = d_scale(ct, ca);  A[1][2] = d_scale(ct, -sa); A[1][3] = d_const(0);
    A[2][0] = d_const(0); A[2][1] = d_const(sa); A[2][2] = d_const(ca);   A[2][3] = dd;
    dual N[3][4];
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 4; ++c) {
            dual s = d_mul(T[r][0], A[0][c]);
            s = d_add(s, d_mul(T[r][1], A[1][c]));
            s = d_add(s, d_mul(T[r][2], A[2][c]));
            if (c == 3) s = d_add(s, T[r][3]);
            N[r][c] = s;
        }
    memcpy(T, N, sizeof N);
}

/* Probe._gen_links (Probe.py:147-167), base Ry(-180 deg) (Probe.py:14,140), q8 = 0 (Probe.py:441) */
static void probe_chain(const dual q[7], real L, real theta_cam, dual p[3], dual R[3][3], dual z6[3]) {
    const real hp = R_PI / 2;
    dual T[3][4];
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 4; ++c) T[r][c] = d_const(0);
    T[0][0] = d_const(-1); T[1][1] = d_const(1); T[2][2] = d_const(-1);
    chain_step(T, d_add(q[0], d_const(hp)), d_const(0), hp);
    chain_step(T, d_add(q[1], d_const(-hp)), d_const(0), -hp);
    chain_step(T, q[2], d_const(0), 0.0);
    chain_step(T, d_const(0), q[3], hp);
    chain_step(T, d_const(hp), q[4], hp);
    chain_step(T, d_const(-hp), q[5], hp);
    for (int r = 0; r < 3; ++r) z6[r] = T[r][2];
    chain_step(T, q[6], d_const(L), theta_cam);
    chain_step(T, d_const(0), d_const(0), 0.0);
    for (int r = 0; r < 3; ++r) {
        p[r] = T[r][3];
        for (int c = 0; c < 3; ++c) R[r][c] = T[r][c];
    }
}

/* SymProbe.get_est_fwkin (Probe.py:470-480) in plain doubles: p, R, om = z6*qd7 (v = acc = 0) */
static void probe_fwkin(const real dofs[6], const real notch[3], real L, real theta_cam,
                        real p[3], real R[9], real om[3]) {
    dual q[7], dp[3], dR[3][3], dz[3];
    for (int i = 0; i < 6; ++i) q[i] = d_const(dofs[i]);
    q[6] = d_const(notch[0]);
    probe_chain(q, L, theta_cam, dp, dR, dz);
    for (int r = 0; r < 3; ++r) {
        p[r] = dp[r].v;
        om[r] = dz[r].v * notch[1];
        for (int c = 0; c < 3; ++c) R[3 * r + c] = dR[r][c].v;
    }
}

/* ------------------------------------------------------------------------- */
/* predict                                                                    */
/* ------------------------------------------------------------------------- */
void eskf_oracle_init_rwb(const double* x26, double* Rwb9) {
    real q[4], M[9];
    for (int i = 0; i < 4; ++i) q[i] = x26[6 + i];
    quat_to_mat(q, M);
    for (int i = 0; i < 9; ++i) Rwb9[i] = (double)M[i];
}

/* Filter._predict_nominal (232-247), eqns.f_predict (equations.py:44-50,72-100),
 * State.from_array (state.py:62-74) */
static void predict_nominal(const eskf_oracle_params* prm, real* x, const real* om_o,
                            const real* acc_o, real dt, const real* om_n, const real* acc_n) {
    real R[9], Rc[9], B_p[3], R_BC[9], B_om[3];
    quat_to_mat(x + 6, R);
    quat_to_mat(x + 22, Rc);
    probe_fwkin(x + 10, x + 16, prm->L, prm->theta_cam, B_p, R_BC, B_om);

    real om_avg[3], w[3], S[9], RS[9], Rn[9];
    for (int i = 0; i < 3; ++i) { om_avg[i] = (om_o[i] + om_n[i]) / 2; w[i] = dt * om_avg[i]; }
    skew3(w, S);
    mat3_mul(R, S, RS);
    for (int i = 0; i < 9; ++i) Rn[i] = R[i] + RS[i];                  /* equations.py:46 */
    real a1[3], a2[3], acc_avg[3];
    mat3_vec(R, acc_o, a1);
    mat3_vec(Rn, acc_n, a2);
    for (int i = 0; i < 3; ++i) acc_avg[i] = (a1[i] + a2[i]) / 2;      /* equations.py:47 */
    real u[3], C_om[3];
    for (int i = 0; i < 3; ++i) u[i] = om_o[i] + B_om[i];
    mat3T_vec(R_BC, u, C_om);                                          /* equations.py:50 */

    real p_n[3], v_n[3], pc_n[3], cr[3], Rcr[3];
    cross3(om_avg, B_p, cr);                                           /* BB_v_CB = 0 */
    mat3_vec(R, cr, Rcr);
    for (int i = 0; i < 3; ++i) {
        p_n[i] = x[i] + dt * x[3 + i] + (dt * dt / 2) * acc_avg[i];    /* equations.py:76 */
        v_n[i] = x[3 + i] + dt * acc_avg[i];                           /* equations.py:77 */
        pc_n[i] = x[19 + i] + dt * x[3 + i] + dt * Rcr[i];             /* equations.py:83-85 */
    }
    real wc[3], Sc[9], RcS[9], Rcn[9];
    for (int i = 0; i < 3; ++i) wc[i] = dt * C_om[i];
    skew3(wc, Sc);
    mat3_mul(Rc, Sc, RcS);
    for (int i = 0; i < 9; ++i) Rcn[i] = Rc[i] + RcS[i];               /* equations.py:86 */

    real n0 = x[16] + dt * x[17], n1 = x[17] + dt * x[18];           /* equations.py:80-82 */
    for (int i = 0; i < 3; ++i) { x[i] = p_n[i]; x[3 + i] = v_n[i]; x[19 + i] = pc_n[i]; }
    x[16] = n0; x[17] = n1;
    if (!prm->hold_frozen)                                             /* Filter.py:243-245 */
        for (int i = 0; i < 6; ++i)
            if (prm->frozen[i]) x[10 + i] = 0.0;
    quat_from_mat(Rn, x + 6);                                          /* state.py:67 */
    quat_from_mat(Rcn, x + 22);                                        /* state.py:72 */
}

/* err_p_C_next / err_theta_C_next (Filter.py:274-277) on duals, evaluated at
 * error = 0 and noise symbols = their std-devs (Filter.py:330-340).  out6[r].d
 * holds row r of the 6x35 Jacobian. */
static void err_cam_next(const eskf_oracle_params* prm, const real* x, const real* Rwb,
                         const real* om_b, real dt, dual out6[6]) {
    dual ev[ND];
    for (int i = 0; i < ND; ++i) ev[i] = d_var(0.0, i);
    for (int i = 0; i < 3; ++i) { ev[22 + i].v = prm->sigma_a[i]; ev[25 + i].v = prm->sigma_om[i]; }
    const dual* err_v_B = ev + 3;
    const dual* err_theta = ev + 6;
    const dual* err_q = ev + 9;
    const dual* err_p_C = ev + 16;
    const dual* err_theta_C = ev + 19;
    const dual* n_om = ev + 25;
    const real* dofs = x + 10;
    const real* notch = x + 16;

    /* estimated and "true" probe kinematics (Probe.py:447-461,465-468) */
    dual q_est[7], q_tr[7];
    for (int i = 0; i < 6; ++i) q_est[i] = d_const(dofs[i]);
    q_est[6] = d_const(notch[0]);
    for (int i = 0; i < 7; ++i) q_tr[i] = d_add(q_est[i], err_q[i]);
    dual p[3], R[3][3], z6[3], p_tr[3], R_tr[3][3], z6_tr[3];
    probe_chain(q_est, prm->L, prm->theta_cam, p, R, z6);
    probe_chain(q_tr, prm->L, prm->theta_cam, p_tr, R_tr, z6_tr);

    /* R_WB_tr = R_WB (I + [err_theta]x) (symbols.py:129); om_tr = om - n_om (131) */
    dual R_WB_tr[3][3];
    {
        dual Sk[3][3];
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c) Sk[r][c] = d_const(0);
        Sk[0][1] = d_scale(err_theta[2], -1); Sk[0][2] = err_theta[1];
        Sk[1][0] = err_theta[2];              Sk[1][2] = d_scale(err_theta[0], -1);
        Sk[2][0] = d_scale(err_theta[1], -1); Sk[2][1] = err_theta[0];
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c) {
                dual s = d_const(Rwb[3 * r + c]);
                for (int k = 0; k < 3; ++k) s = d_add(s, d_scale(Sk[k][c], Rwb[3 * r + k]));
                R_WB_tr[r][c] = s;
            }
    }
    dual om_tr[3], om_bd[3];
    for (int i = 0; i < 3; ++i) { om_bd[i] = d_const(om_b[i]); om_tr[i] = d_sub(om_bd[i], n_om[i]); }

    /* get_err_pc_dot (symbols.py:134-160); v = 0, v_tr := p_tr (Probe.py:458) */
    dual cr[3], cr_tr[3], in[3], in_tr[3], pcd[3], pcd_tr[3];
    dv_cross(om_bd, p, cr);
    dv_cross(om_tr, p_tr, cr_tr);
    for (int i = 0; i < 3; ++i) { in[i] = cr[i]; in_tr[i] = d_add(p_tr[i], cr_tr[i]); }
    for (int r = 0; r < 3; ++r) {
        dual s = d_const(0), t = d_const(0);
        for (int k = 0; k < 3; ++k) {
            s = d_add(s, d_scale(in[k], Rwb[3 * r + k]));
            t = d_add(t, d_mul(R_WB_tr[r][k], in_tr[k]));
        }
        pcd[r] = s; pcd_tr[r] = t;
    }
    for (int i = 0; i < 3; ++i) {
        dual e_dot = d_sub(d_add(err_v_B[i], pcd_tr[i]), pcd[i]);
        out6[i] = d_add(err_p_C[i], d_scale(e_dot, dt));
    }

    /* get_err_theta_c_dot (symbols.py:186-200) with quat_matrix (168-183), w = 0 */
    dual u[3], u_tr[3], om_c[3], om_c_tr[3];
    for (int i = 0; i < 3; ++i) {
        u[i] = d_add(om_bd[i], d_scale(z6[i], notch[1]));
        u_tr[i] = d_add(om_tr[i], d_scale(z6_tr[i], notch[1]));   /* Probe.py:459: only q substituted */
    }
    for (int c = 0; c < 3; ++c) {
        dual s = d_const(0), t = d_const(0);
        for (int k = 0; k < 3; ++k) {
            s = d_add(s, d_mul(R[k][c], u[k]));
            t = d_add(t, d_mul(R_tr[k][c], u_tr[k]));
        }
        om_c[c] = s; om_c_tr[c] = t;
    }
    dual half[3], x1[3], x2[3];
    for (int i = 0; i < 3; ++i) half[i] = d_scale(err_theta_C[i], 0.5);
    dv_cross(om_c_tr, half, x1);   /* skew(om_c_tr) @ (0.5 err_theta_C) */
    dv_cross(om_c, half, x2);
    for (int i = 0; i < 3; ++i) {
        dual e_dot = d_sub(d_sub(d_sub(om_c_tr[i], om_c[i]), x1[i]), x2[i]);
        out6[3 + i] = d_add(err_theta_C[i], d_scale(e_dot, dt));
    }
}

/* Filter._predict_error (249-268) + _cam_error_jacobian (270-342) */
static void predict_error(const eskf_oracle_params* prm, const real* x, const real* Rwb,
                          const real* om_o, const real* acc_o, real dt, real* Fx, real* Fi) {
    memset(Fx, 0, sizeof(real) * ESKF_NE * ESKF_NE);
    memset(Fi, 0, sizeof(real) * ESKF_NE * ESKF_NN);
    for (int i = 0; i < ESKF_NE; ++i) Fx[i * ESKF_NE + i] = 1.0;
    for (int i = 0; i < 3; ++i) Fx[i * ESKF_NE + 3 + i] = dt;
    real Sa[9], RSa[9];
    skew3(acc_o, Sa);
    mat3_mul(Rwb, Sa, RSa);
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) Fx[(3 + i) * ESKF_NE + 6 + j] = -RSa[3 * i + j] * dt;
    real Om[4] = {0.5 * dt * om_o[0], 0.5 * dt * om_o[1], 0.5 * dt * om_o[2], 1.0}, OmR[9];
    quat_nrm(Om);                                                      /* Filter.py:132-134 */
    quat_to_mat(Om, OmR);
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) Fx[(6 + i) * ESKF_NE + 6 + j] = OmR[3 * j + i];
    Fx[15 * ESKF_NE + 16] += dt;
    Fx[16 * ESKF_NE + 17] += dt;

    dual o[6];
    err_cam_next(prm, x, Rwb, om_o, dt, o);
    /* column cursor of Filter.py:279-285: widths 3,3,3,6,1,3,3 = 22 (symbols.py:107) */
    for (int r = 0; r < 6; ++r)
        for (int c = 0; c < 22; ++c) Fx[(18 + r) * ESKF_NE + c] = o[r].d[c];
    for (int i = 0; i < 12; ++i) Fi[(3 + i) * ESKF_NN + i] = 1.0;
    Fi[17 * ESKF_NN + 12] = 1.0;
    for (int r = 0; r < 6; ++r)
        for (int c = 0; c < ESKF_NN; ++c) Fi[(18 + r) * ESKF_NN + c] = o[r].d[22 + c];
}

static void matmul(const real* A, const real* B, real* C, int n, int k, int m) {
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < m; ++j) {
            real s = 0;
            for (int l = 0; l < k; ++l) s += A[i * k + l] * B[l * m + j];
            C[i * m + j] = s;
        }
}
static void matmul_bt(const real* A, const real* B, real* C, int n, int k, int m) { /* C = A B^T, B is m x k */
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < m; ++j) {
            real s = 0;
            for (int l = 0; l < k; ++l) s += A[i * k + l] * B[j * k + l];
            C[i * m + j] = s;
        }
}

static void predict_r(const eskf_oracle_params* prm, real* x, real* P, real* om_old,
                      real* acc_old, real* Rwb, real dt, const real* om, const real* acc,
                      real* Fx_out, real* Fi_out) {
    real Fx[ESKF_NE * ESKF_NE], Fi[ESKF_NE * ESKF_NN];
    predict_nominal(prm, x, om_old, acc_old, dt, om, acc);
    predict_error(prm, x, Rwb, om_old, acc_old, dt, Fx, Fi);
    /* P <- Fx P Fx^T + Fi Q Fi^T (Filter.py:349) */
    real T1[ESKF_NE * ESKF_NE], T2[ESKF_NE * ESKF_NE], T3[ESKF_NE * ESKF_NN], T4[ESKF_NE * ESKF_NE];
    matmul(Fx, P, T1, ESKF_NE, ESKF_NE, ESKF_NE);
    matmul_bt(T1, Fx, T2, ESKF_NE, ESKF_NE, ESKF_NE);
    real Qr[ESKF_NN * ESKF_NN];
    for (int i = 0; i < ESKF_NN * ESKF_NN; ++i) Qr[i] = prm->Q[i];
    matmul(Fi, Qr, T3, ESKF_NE, ESKF_NN, ESKF_NN);
    matmul_bt(T3, Fi, T4, ESKF_NE, ESKF_NN, ESKF_NE);
    for (int i = 0; i < ESKF_NE * ESKF_NE; ++i) P[i] = T2[i] + T4[i];
    for (int i = 0; i < 3; ++i) { om_old[i] = om[i]; acc_old[i] = acc[i]; }   /* Filter.py:225-226 */
    quat_to_mat(x + 6, Rwb);                                                   /* Filter.py:227 */
    if (Fx_out) memcpy(Fx_out, Fx, sizeof Fx);
    if (Fi_out) memcpy(Fi_out, Fi, sizeof Fi);
}

/* the API works on doubles; the arithmetic runs in `real` (a copy in, a rounding out: no-ops for the
 * double build) */
static void to_real(const double* a, real* r, int n) { for (int i = 0; i < n; ++i) r[i] = a[i]; }
static void to_double(const real* r, double* a, int n) { for (int i = 0; i < n; ++i) a[i] = (double)r[i]; }

void eskf_oracle_predict(const eskf_oracle_params* prm, double* x, double* P, double* om_old,
                         double* acc_old, double* Rwb, double dt, const double* om, const double* acc,
                         double* Fx_out, double* Fi_out) {
    real xr[ESKF_NX], Pr[ESKF_NE * ESKF_NE], oo[3], ao[3], Rr[9], on[3], an[3];
    real Fx[ESKF_NE * ESKF_NE], Fi[ESKF_NE * ESKF_NN];
    to_real(x, xr, ESKF_NX); to_real(P, Pr, ESKF_NE * ESKF_NE); to_real(om_old, oo, 3); to_real(acc_old, ao, 3);
    to_real(Rwb, Rr, 9); to_real(om, on, 3); to_real(acc, an, 3);
    predict_r(prm, xr, Pr, oo, ao, Rr, dt, on, an, Fx, Fi);
    to_double(xr, x, ESKF_NX); to_double(Pr, P, ESKF_NE * ESKF_NE); to_double(oo, om_old, 3); to_double(ao, acc_old, 3);
    to_double(Rr, Rwb, 9);
    if (Fx_out) to_double(Fx, Fx_out, ESKF_NE * ESKF_NE);
    if (Fi_out) to_double(Fi, Fi_out, ESKF_NE * ESKF_NN);
}

/* ------------------------------------------------------------------------- */
/* update                                                                     */
/* ------------------------------------------------------------------------- */
/* inverse by LU with partial pivoting (np.linalg.inv -> LAPACK getrf/getri);
 * returns 1 on an exactly zero pivot (LinAlgError "Singular matrix") */
static int inv7(const real* S, real* Sinv) {
    enum { N = ESKF_NM };
    real A[N * N];
    int piv[N];
    memcpy(A, S, sizeof A);
    for (int k = 0; k < N; ++k) {
        int p = k;
        real best = r_fabs(A[k * N + k]);
        for (int i = k + 1; i < N; ++i)
            if (r_fabs(A[i * N + k]) > best) { best = r_fabs(A[i * N + k]); p = i; }
        piv[k] = p;
        if (p != k)
            for (int j = 0; j < N; ++j) { real t = A[k * N + j]; A[k * N + j] = A[p * N + j]; A[p * N + j] = t; }
        if (A[k * N + k] == 0.0) return 1;
        for (int i = k + 1; i < N; ++i) {
            A[i * N + k] /= A[k * N + k];
            for (int j = k + 1; j < N; ++j) A[i * N + j] -= A[i * N + k] * A[k * N + j];
        }
    }
    for (int c = 0; c < N; ++c) {
        real b[N];
        for (int i = 0; i < N; ++i) b[i] = (i == c) ? 1.0 : 0.0;
        for (int k = 0; k < N; ++k)
            if (piv[k] != k) { real t = b[k]; b[k] = b[piv[k]]; b[piv[k]] = t; }
        for (int i = 1; i < N; ++i)
            for (int j = 0; j < i; ++j) b[i] -= A[i * N + j] * b[j];
        for (int i = N - 1; i >= 0; --i) {
            for (int j = i + 1; j < N; ++j) b[i] -= A[i * N + j] * b[j];
            b[i] /= A[i * N + i];
        }
        for (int i = 0; i < N; ++i) Sinv[i * N + c] = b[i];
    }
    return 0;
}

static int update_r(const eskf_oracle_params* prm, real* x, real* P, const real* pos,
                    const real* quat_raw, real ang_notch, real* K_out) {
    enum { NE = ESKF_NE, NM = ESKF_NM };
    int status = 0;
    real H[NM * NE];
    memset(H, 0, sizeof H);
    for (int i = 0; i < 6; ++i) H[i * NE + 18 + i] = 1.0;                /* Filter.py:83-85 */
    H[6 * NE + 15] = 1.0;

    real HP[NM * NE], S[NM * NM], Sinv[NM * NM], PHt[NE * NM], K[NE * NM];
    matmul(H, P, HP, NM, NE, NE);
    matmul_bt(HP, H, S, NM, NE, NM);
    real Rr[NM * NM];
    for (int i = 0; i < NM * NM; ++i) Rr[i] = prm->R[i];
    for (int i = 0; i < NM * NM; ++i) S[i] += Rr[i];                 /* Filter.py:355 */
    if (inv7(S, Sinv)) return ESKF_ORACLE_SINGULAR;                      /* Filter.py:356-361 */
    matmul_bt(P, H, PHt, NE, NE, NM);
    matmul(PHt, Sinv, K, NE, NM, NM);                                    /* Filter.py:357 */

    /* measurement correction and residual (Filter.py:364-375) */
    real nq[4] = {0, 0, r_sin(ang_notch / 2), r_cos(ang_notch / 2)};       /* Quaternion.py:73 */
    real qm[4], qe[4], qmc[4];
    quat_mul(nq, quat_raw, qm);
    quat_nrm(qm);
    qmc[0] = -qm[0]; qmc[1] = -qm[1]; qmc[2] = -qm[2]; qmc[3] = qm[3];
    quat_mul(qmc, x + 22, qe);
    quat_nrm(qe);
    real res[NM];
    for (int i = 0; i < 3; ++i) res[i] = pos[i] - x[19 + i];
    real nv = r_sqrt(qe[0] * qe[0] + qe[1] * qe[1] + qe[2] * qe[2]);
    if (nv > 1.0) { nv = 1.0; status |= ESKF_ORACLE_NONFINITE; }
    real ang = r_asin(nv);                                               /* Quaternion.py:138-147 */
    for (int i = 0; i < 3; ++i) res[3 + i] = (ang == 0.0) ? 0.0 : ang * (qe[i] / nv);
    res[6] = ang_notch - x[16];
    /* NumPy propagates a NaN / inf measurement or state silently (no check in Filter.update); the
     * status word records it */
    if (!r_isfinite(res[0] + res[1] + res[2] + res[3] + res[4] + res[5] + res[6])) status |= ESKF_ORACLE_NONFINITE;

    real e[NE];
    for (int i = 0; i < NE; ++i) {
        real s = 0;
        for (int j = 0; j < NM; ++j) s += K[i * NM + j] * res[j];
        e[i] = s;
    }
    for (int i = 0; i < 6; ++i)
        if (prm->frozen[i]) e[9 + i] = 0.0;                              /* Filter.py:377-379 */

    /* ErrorState.from_array (state.py:105-129), State.__add__ (state.py:46-60) */
    const real* th = e + 6;
    const real* thc = e + 21;
    real dq[4], dqc[4], qn[4], qcn[4];
    quat_about_axis(r_sqrt(th[0] * th[0] + th[1] * th[1] + th[2] * th[2]), th, dq);
    quat_about_axis(r_sqrt(thc[0] * thc[0] + thc[1] * thc[1] + thc[2] * thc[2]), th, dqc);   /* sic: axis = theta */
    quat_mul(x + 6, dq, qn);
    quat_nrm(qn); quat_nrm(qn);
    quat_mul(x + 22, dqc, qcn);
    quat_nrm(qcn); quat_nrm(qcn);
    for (int i = 0; i < 3; ++i) { x[i] += e[i]; x[3 + i] += e[3 + i]; x[16 + i] += e[15 + i]; x[19 + i] += e[18 + i]; }
    for (int i = 0; i < 6; ++i) x[10 + i] += e[9 + i];
    memcpy(x + 6, qn, sizeof qn);
    memcpy(x + 22, qcn, sizeof qcn);

    /* Joseph form (Filter.py:383-384) */
    real A[NE * NE], KH[NE * NE], T1[NE * NE], T2[NE * NE], KR[NE * NM], KRKt[NE * NE];
    matmul(K, H, KH, NE, NM, NE);
    for (int i = 0; i < NE; ++i)
        for (int j = 0; j < NE; ++j) A[i * NE + j] = (i == j ? 1.0 : 0.0) - KH[i * NE + j];
    matmul(A, P, T1, NE, NE, NE);
    matmul_bt(T1, A, T2, NE, NE, NE);
    matmul(K, Rr, KR, NE, NM, NM);
    matmul_bt(KR, K, KRKt, NE, NM, NE);
    for (int i = 0; i < NE * NE; ++i) T2[i] += KRKt[i];

    /* error reset (Filter.py:387-390) */
    real G[NE * NE];
    memset(G, 0, sizeof G);
    for (int i = 0; i < NE; ++i) G[i * NE + i] = 1.0;
    real h1[3] = {0.5 * th[0], 0.5 * th[1], 0.5 * th[2]}, h2[3] = {0.5 * thc[0], 0.5 * thc[1], 0.5 * thc[2]};
    real S1[9], S2[9];
    skew3(h1, S1);
    skew3(h2, S2);
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            G[(6 + i) * NE + 6 + j] = (i == j ? 1.0 : 0.0) - S1[3 * i + j];
            G[(21 + i) * NE + 21 + j] = (i == j ? 1.0 : 0.0) - S2[3 * i + j];
        }
    matmul(G, T2, T1, NE, NE, NE);
    matmul_bt(T1, G, P, NE, NE, NE);

    if (K_out) memcpy(K_out, K, sizeof K);
    return status;
}

int eskf_oracle_update(const eskf_oracle_params* prm, double* x, double* P, const double* pos,
                       const double* quat_raw, double ang_notch, double* K_out) {
    real xr[ESKF_NX], Pr[ESKF_NE * ESKF_NE], pr[3], qr[4], K[ESKF_NE * ESKF_NM];
    to_real(x, xr, ESKF_NX); to_real(P, Pr, ESKF_NE * ESKF_NE); to_real(pos, pr, 3); to_real(quat_raw, qr, 4);
    const int status = update_r(prm, xr, Pr, pr, qr, ang_notch, K);
    if (status & ESKF_ORACLE_SINGULAR) return status;                    /* nothing changes */
    to_double(xr, x, ESKF_NX); to_double(Pr, P, ESKF_NE * ESKF_NE);
    if (K_out) to_double(K, K_out, ESKF_NE * ESKF_NM);
    return status;
}

/* ------------------------------------------------------------------------- */
/* batched drivers                                                            */
/* ------------------------------------------------------------------------- */
void eskf_oracle_predict_batch(const eskf_oracle_params* prm, int64_t B, double* x, double* P,
                               double* om_old, double* acc_old, double* Rwb, const double* dt,
                               int dt_stride, const double* om, const double* acc) {
#pragma omp parallel for schedule(static)
    for (int64_t b = 0; b < B; ++b)
        eskf_oracle_predict(prm, x + b * ESKF_NX, P + b * ESKF_NE * ESKF_NE, om_old + 3 * b, acc_old + 3 * b,
                            Rwb + 9 * b, dt[b * dt_stride], om + 3 * b, acc + 3 * b, 0, 0);
}

void eskf_oracle_update_batch(const eskf_oracle_params* prm, int64_t B, double* x, double* P,
                              const double* pos, const double* quat, const double* notch, int32_t* status) {
#pragma omp parallel for schedule(static)
    for (int64_t b = 0; b < B; ++b) {
        int s = eskf_oracle_update(prm, x + b * ESKF_NX, P + b * ESKF_NE * ESKF_NE, pos + 3 * b, quat + 4 * b,
                                   notch[b], 0);
        if (status) status[b] = s;
    }
}

int eskf_oracle_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* threads the batched drivers use from now on (bench.py times the oracle on one thread as well as on all host cores);
 * returns the number in effect */
int eskf_oracle_set_num_threads(int n) {
#ifdef _OPENMP
    if (n >= 1) omp_set_num_threads(n);
    return omp_get_max_threads();
#else
    (void)n;
    return 1;
#endif
}
