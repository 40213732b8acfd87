"""Is the device step as accurate as the reference's fp64 arithmetic?  Asserted on a population.

For 32 seeded bundle-adjustment problems (16 per mode, lambda = 1e-3: cond(S) ~ 1e9) the device step and the fp64 oracle's
step are both measured against the EXACT step of the oracle's linearisation (oracle/ba_oracle.c, ora_solve_augmented_quad,
__float128; pinned by tests/test_oracle_referee.py).  Case by case the ratio e_gpu / e_64 is heavy-tailed (tests/referee.py
explains why), so the assertions are on the distribution:

    median(e_gpu / e_64) <= 1         the device is at least as accurate as the fp64 CPU path in the typical case
    geometric mean       <= 1         ... and on (log-)average
    >= 70 % of the cases within 2 x   ... and rarely much worse
    every e_gpu <= 1e-7, worst ratio <= 64

The 32 small problems make the device's Schur reduction and factorisation deterministic (no split blocks, conflict-free
update rounds): their numbers do not move from run to run.  Round 4 adds two members per mode with 320 and 420 cameras
(three tile rows and more: split blocks with the atomic flush, nested-dissection levels, the dataflow launch of the top
groups), which the small ones never reach; their last bits depend on the order of a few atomic adds.
"""
import numpy as np
import pytest

import apex_solver_amd as pkg
import referee
from apex_solver_amd.solver import GpuSchurComplementSolver, OptimizationType, Problem

pytestmark = pytest.mark.gpu

SHAPES = [(12, 500), (24, 1200), (40, 2000), (60, 3000)]
LARGE = [(320, 9000), (420, 12000)]


def test_device_step_is_as_accurate_as_fp64_on_a_population(oracle):
    lam = 1e-3
    rows = []
    for mode in ("selfcal", "ba"):
        ot = OptimizationType.SelfCalibration if mode == "selfcal" else OptimizationType.BundleAdjustment
        for k in range(16 + len(LARGE)):
            n_cam, n_pt = SHAPES[k % 4] if k < 16 else LARGE[k - 16]
            d = pkg.synthetic.make_problem(n_cam, n_pt, 3, 7, config_id=300 + k)
            prob = Problem.bundle_adjustment(d, ot, 1.0)
            s = GpuSchurComplementSolver(0).initialize_structure(prob)
            s.set_parameters(d.poses, d.intr, d.points)
            step = s.solve_augmented_equation(lam)
            s.close()
            o = oracle.from_data(d, prob.layout, mode=mode)
            o.linearize()
            ostep, _ = o.solve_augmented(lam, 0)
            exact, info = o.solve_augmented_quad(lam)
            assert info["residual"] < 1e-26
            rows.append((f"{mode} {n_cam}x{n_pt} #{k}", referee.rel(step, exact), referee.rel(ostep, exact)))
    e_gpu = np.array([r[1] for r in rows]); e_64 = np.array([r[2] for r in rows])
    ratio = e_gpu / e_64
    for (label, a, b), q in zip(rows, ratio):
        print(f"{label:26s} |gpu - exact| {a:.2e}  |fp64 oracle - exact| {b:.2e}  ratio {q:.2f}")
    med, gm, within2 = float(np.median(ratio)), float(np.exp(np.mean(np.log(ratio)))), float(np.mean(ratio <= 2.0))
    print(f"population of {len(rows)}: median ratio {med:.2f}  geometric mean {gm:.2f}  within 2x: {100 * within2:.0f} %  worst {ratio.max():.1f}"
          f"  | worst e_gpu {e_gpu.max():.1e}  worst e_64 {e_64.max():.1e}")
    if referee.RECORD:   # the other parity tests of this session, for the log (DESIGN.md section 2 quotes them)
        rr = np.array([a / b for _, a, b in referee.RECORD])
        print(f"parity cases refereed earlier in this session: {len(rr)}, median ratio {np.median(rr):.2f}, geometric mean "
              f"{np.exp(np.mean(np.log(rr))):.2f}, within 2x {100 * np.mean(rr <= 2):.0f} %, worst {rr.max():.1f}")
    assert med <= 1.0 and gm <= 1.0, (med, gm)
    assert within2 >= 0.70 and ratio.max() <= 64.0, (within2, ratio.max())
    assert e_gpu.max() <= referee.FP64_ENVELOPE
