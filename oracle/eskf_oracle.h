/* CPU oracle of the ESKF predict/update hot path -- TEST INFRASTRUCTURE ONLY.
 *
 * Plain-C restatement of salehahr/dvi-ekf @ v1, dvi_ekf/filter/Filter.py:219-395
 * and what it calls.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may link or load this; the product never does.
 *
 * Pinned by tests/test_oracle_golden.py against the reference's saved trajectory
 * tests/golden/data/trajs/kf_best_mandala0_mono.txt (300 values, 9 decimals) and
 * against the NumPy twin oracle/eskf_ref.py.
 */
#ifndef ESKF_ORACLE_H
#define ESKF_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ESKF_NX 26
#define ESKF_NE 24
#define ESKF_NN 13
#define ESKF_NM 7

typedef struct {
    double L;            /* scope length            config.yaml:33            */
    double theta_cam;    /* camera angle [rad]      config.yaml:34            */
    int32_t frozen[6];   /* frozen DOF mask         config.yaml:8             */
    int32_t hold_frozen; /* 0 = v1 literal (frozen dofs zeroed after predict,
                            Filter.py:243-245); 1 = golden-compat (held)      */
    double sigma_a[3];   /* accel noise std         config.py:116-120         */
    double sigma_om[3];  /* gyro noise std          config.py:106-113         */
    double Q[ESKF_NN * ESKF_NN]; /* Filter.py:68-72 */
    double R[ESKF_NM * ESKF_NM]; /* Filter.py:75    */
} eskf_oracle_params;

/* status bits returned per instance by update */
#define ESKF_ORACLE_SINGULAR 1  /* exact zero pivot: numpy LinAlgError path, Filter.py:356-361 */
#define ESKF_ORACLE_NONFINITE 2 /* asin argument clamped / non-finite state                    */

/* R_WB_old <- Rm(q) (Filter.py:80) */
void eskf_oracle_init_rwb(const double* x26, double* Rwb9);

/* Filter.propagate (Filter.py:219-230) for one instance.  Fx_out (576) / Fi_out
 * (312) may be NULL. */
void eskf_oracle_predict(const eskf_oracle_params* prm, double* x26, double* P576, double* om_old3,
                         double* acc_old3, double* Rwb9, double dt, const double* om3,
                         const double* acc3, double* Fx_out, double* Fi_out);

/* Filter.update (Filter.py:351-395) for one instance.  K_out (24x7) may be NULL.
 * Returns status bits; on ESKF_ORACLE_SINGULAR state and P are left untouched. */
int eskf_oracle_update(const eskf_oracle_params* prm, double* x26, double* P576, const double* pos3,
                       const double* quat4_raw, double ang_notch, double* K_out);

/* Batched drivers (OpenMP over instances).  dt: dt_stride = 0 -> scalar shared by
 * the batch, 1 -> one per instance. */
void eskf_oracle_predict_batch(const eskf_oracle_params* prm, int64_t B, double* x, double* P,
                               double* om_old, double* acc_old, double* Rwb, const double* dt,
                               int dt_stride, const double* om, const double* acc);
void eskf_oracle_update_batch(const eskf_oracle_params* prm, int64_t B, double* x, double* P,
                              const double* pos, const double* quat, const double* notch,
                              int32_t* status);
int eskf_oracle_num_threads(void);
int eskf_oracle_set_num_threads(int n);

#ifdef __cplusplus
}
#endif
#endif
