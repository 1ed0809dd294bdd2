"""The saved trajectory pins the oracle quirk by quirk.

``oracle/eskf_ref.py`` keeps every v1 behaviour that a "corrected" filter would not have behind a named flag; the
default set is the v1 source and reproduces ``kf_best_mandala0_mono.txt`` to its print rounding
(tests/test_oracle_golden.py).  Here each flag is switched off on its own: the trajectory then leaves the saved one by
orders of magnitude more than the 5e-10 the file resolves -- so an oracle (or a kernel, which is compared with this
oracle) that got any single one of them wrong could not have passed the golden test.  The survey measured the same
sensitivities with its own throw-away restatement (Q1 3.6, Q2 18, Q3 1.6, Q4 0.97, Q5 0.97, Q6 4.3,
Q7 4.8e-3, Q8 4.0e-6, Q9 0.13, Q10 2e-3); the numbers below are this oracle's."""
import numpy as np
import pytest

from oracle.eskf_ref import QUIRKS, V1, RefFilter, RefParams, ref_process_noise
from tests.helpers import default_run, load_golden, log_row

PRINT_ROUNDING = 5.0e-10 + 1e-12

# quirk -> smallest deviation from the saved file (max abs over its 300 values) this test accepts with the quirk off;
# a tenth of what the survey measured (the "corrected" variant of a quirk is not unique in every detail)
MIN_DEVIATION = {
    "Q1_jac_col_shift": 0.36, "Q2_vtr_is_ptr": 1.8, "Q3_dqc_axis_theta": 0.16, "Q4_q_imu_zero": 0.097,
    "Q5_jac_at_sigma": 0.097, "Q6_half_angle_residual": 0.43, "Q7_stale_Rwb_old": 4.8e-4, "Q8_jac_uses_new_dofs": 4.0e-7,
    "Q9_Rc_uses_om_old": 0.013, "Q10_markley_first_order": 2e-4,
}


def _rows(quirks):
    cfg, ri, Q, R, sa, so = default_run(golden_compat=True)
    Qn = ref_process_noise(ri.dt[0], sa, so, cfg.process_noise_rw_var, quirks)
    if quirks == V1:
        assert np.array_equal(Qn, Q)          # the helper is the product's run-mode Q (dvi_ekf_amd.schedule.noise_matrices)
    prm = RefParams(L=cfg.model.length, theta_cam=cfg.model.angle, frozen=tuple(cfg.frozen_dofs), hold_frozen=True,
                    sigma_a=sa, sigma_om=so, Q=Qn, R=R, quirks=quirks)
    f = RefFilter(prm, ri.x0, ri.P0, ri.om0, ri.acc0)
    rows = [log_row(ri.t0, f.x)]
    ip = 0
    for k in range(len(ri.t_meas)):
        for _ in range(ri.n_pred[k]):
            f.predict(ri.dt[ip], ri.om[ip], ri.acc[ip])
            ip += 1
        f.update(ri.pos[k], ri.quat[k], ri.notch[k])
        rows.append(log_row(ri.t_meas[k], f.x))
    return np.array(rows)


def test_flag_names_are_the_surveys():
    assert sorted(QUIRKS, key=lambda n: int(n[1:n.index("_")])) == list(MIN_DEVIATION)
    assert V1 == frozenset(MIN_DEVIATION)


def test_default_flags_are_the_saved_trajectory():
    assert np.abs(_rows(V1) - load_golden()).max() <= PRINT_ROUNDING


@pytest.mark.parametrize("quirk", list(MIN_DEVIATION))
def test_saved_trajectory_resolves_each_quirk(quirk):
    dev = _rows(V1 - {quirk}) - load_golden()
    # (angles are logged in degrees and may wrap: the state columns that cannot -- positions, velocities, quaternions --
    #  carry the bound on their own as well whenever they move at all)
    worst = float(np.nanmax(np.abs(dev)))
    print(f"{quirk}: off -> {worst:.3e} from the saved file (print rounding 5e-10)")
    assert np.isfinite(dev).all()
    assert worst >= MIN_DEVIATION[quirk], (quirk, worst)
    assert worst >= 500 * PRINT_ROUNDING
