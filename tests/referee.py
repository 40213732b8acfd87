"""Referee for the bundle-adjustment step (GPU tests).

The north star asks for the device step "within 1e-10 relative of CPU fp64".  On gauge-free bundle adjustment cond(S) is
1e9..1e10 and the CPU fp64 path itself is only 1e-10..1e-8 away from the exact solution of its own equations, so a fixed
1e-10 against the fp64 step cannot be a pass/fail line there.  What can: the EXACT step (oracle/ba_oracle.c,
ora_solve_augmented_quad: H, Hll^-1, S, g_red in __float128, refinement to 1e-28; pinned against a 50-digit mpmath
solve in tests/test_oracle_referee.py).  Every parity case prints the three distances

    e_gpu  = |step_gpu - exact| / |exact|      e_64 = |step_fp64oracle - exact| / |exact|
    e_own  = |step_gpu - exact(device's own exported r, J)|      (the equations the device solver was actually given)

and asserts e_gpu, e_own <= max(1e-7, 8 e_64) (1e-7: the measured envelope of fp64 on these systems, e_64 itself reaches
4.6e-8; the second term only matters on the deliberately pathological cheirality case, where e_64 = 1.8e-6).  The
claim "the device is at least as accurate as the fp64 CPU path" is asserted on a POPULATION
(tests/test_gpu_referee_population.py: 32 seeded problems; median and geometric mean of e_gpu / e_64 <= 1, >= 70 % of
the cases within 2 x), not case by case: both errors are dominated by the rounding's component along the one or two
weakest eigen-directions of S (the damped gauge), so e_gpu / e_64 is a ratio of two nearly one-dimensional random
variables -- heavy-tailed by construction.  Measured in round 3: the 32-problem population has median 0.38, geometric mean 0.43, 97 % within 2 x, worst 3.1 (the
device is typically 2-3 x CLOSER to the exact step than the fp64 oracle: FMA arithmetic, pairwise sums over lanes); the 22
ill-conditioned parity cases median 0.65, geometric mean 0.62, three beyond 2 x (2.2, 4.3, 9.9) and five below 0.13 x.
A per-case "<= 2 e_64" fails one small case in ten with nothing wrong.
Where cond(S) <= 1e5 (lambda = 1e4) the north star's 1e-10 against the fp64 oracle is asserted directly, every case.
"""
import numpy as np

NORTH_STAR = 1e-10
WELL_CONDITIONED_LAMBDA = 1e4
FP64_ENVELOPE = 1e-7          # forward error of ANY fp64 solve of these systems (oracle included) stays below this
RECORD = []                   # (label, e_gpu, e_64) of every refereed case of the session (printed by the population test)


def rel(a, b):
    a = np.ravel(a); b = np.ravel(b)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))


def device_blocks(s, dc):
    r = s.get_residual()
    jc, jl = s.get_jacobian_blocks()
    jp = np.ascontiguousarray(jc[:, :, :6])
    ji = np.ascontiguousarray(jc[:, :, 6:9]) if dc == 9 else np.zeros((jc.shape[0], 2, 3))
    return r, jp, jl, ji


def check_step(o, s, step, ostep, lam, dc, label="", own=True):
    """o: OracleProblem whose LAST linearize() was at the device's parameters (scaling set if the device's is);
    s: the device solver after solve_augmented_equation(lam) returned `step`; ostep: the fp64 oracle's step.
    Returns the measured errors (also printed: DESIGN.md section 2 quotes them)."""
    exact, info = o.solve_augmented_quad(lam)
    e = dict(gpu=rel(step, exact), fp64=rel(ostep, exact), lam=lam, sweeps=info["sweeps"])
    assert info["residual"] < 1e-26, info
    if own:
        r, jp, jl, ji = device_blocks(s, dc)
        o.set_linearization(r, jp, jl, ji)
        exact_own, info2 = o.solve_augmented_quad(lam)
        assert info2["residual"] < 1e-26, info2
        e["gpu_own"] = rel(step, exact_own)
        e["lin"] = rel(exact_own, exact)     # how far apart the two exact steps are: the device's J vs the oracle's J
        o.linearize()                         # put the oracle's own blocks back
    print(f"referee {label} lambda {lam:g}: |gpu - exact| {e['gpu']:.2e}  |fp64 oracle - exact| {e['fp64']:.2e}"
          + (f"  |gpu - exact(own J)| {e['gpu_own']:.2e}  |exact(own J) - exact| {e['lin']:.2e}" if own else ""))
    RECORD.append((label, e["gpu"], e["fp64"]))
    # (a pathological system -- a landmark almost on a camera centre, cond(S) ~ 1e12 -- takes the fp64 oracle itself past
    # the envelope: there the device is held to a small multiple of the oracle's own error)
    bound = max(FP64_ENVELOPE, 8.0 * e["fp64"])
    assert e["gpu"] <= bound, (label, e)
    if own:
        assert e["gpu_own"] <= bound, (label, e)
    return e


def check_well_conditioned(o, s, mode_dc, label="", lam=WELL_CONDITIONED_LAMBDA):
    """cond(S) <= 1e5: the north star's 1e-10 against the fp64 oracle, asserted directly (and against the exact step)."""
    for _ in range(3):   # (a landmark almost on a camera centre can leave cond(S) > 1e5 at 1e4: damp harder)
        ostep, ograd, oS, ogred = o.solve_augmented(lam, 0, want_schur=True)
        cond = float(np.linalg.cond(oS)) if oS.shape[0] <= 4000 else float("nan")
        if not (cond > 1e5):
            break
        lam *= 100.0
    step = s.solve_augmented_equation(lam)
    exact, info = o.solve_augmented_quad(lam)
    e = dict(vs_fp64=rel(step, ostep), vs_exact=rel(step, exact), fp64_vs_exact=rel(ostep, exact), cond=cond)
    print(f"referee {label} lambda {lam:g} (cond(S) {cond:.1e}): gpu vs fp64 oracle {e['vs_fp64']:.2e}  gpu vs exact {e['vs_exact']:.2e}"
          f"  fp64 oracle vs exact {e['fp64_vs_exact']:.2e}")
    assert not (cond > 1e5), cond
    assert e["vs_fp64"] < NORTH_STAR and e["vs_exact"] < NORTH_STAR, (label, e)
    return e



def sym_norm2(S, iters=60, seed=0):
    """|S|_2 of a symmetric matrix by power iteration on S: a LOWER estimate, within a fraction of a percent after 60 steps.  The
    backward-error checks divide by it, so an underestimate only makes them stricter; np.linalg.norm(S, 2) is a full SVD --
    a minute per call at 6,000 rows, half of the GPU suite's wall time on a slow box."""
    import numpy as np

    x = np.random.default_rng(seed).normal(size=S.shape[0])
    nrm = 0.0
    for _ in range(iters):
        x /= np.linalg.norm(x)
        x = S @ x
        nrm = float(np.linalg.norm(x))
    return nrm
