#!/usr/bin/env python3
"""bench.py -- ms per Levenberg-Marquardt iteration of the MI355X bundle-adjustment backend.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--workload NAME] [--scale S]
                  [--mode selfcal|ba] [--variant sparse|iterative] [--no-cpu-baseline]

A "step" is one LM iteration on a synthetic BA problem of the named BASELINE.json shape
(--workload sphere2500 runs the SE3 pose-graph path of BASELINE configs[1] instead):
linearise all factors + explicit Schur complement + damped Cholesky of S + back-substitution
(apexgpu_solve_augmented), step statistics, retraction to a trial point and its cost, and the
accept/reject bookkeeping -- exactly the body of optimize_with_mode's loop
(src/optimizer/levenberg_marquardt.rs:857-1029).  Inputs are resident in HBM before the timed
region.  Rank 0 prints ONE JSON line.

For N > 1 the driver launches this file under torch.distributed.run (one process per GPU, RANK / LOCAL_RANK /
WORLD_SIZE / MASTER_* in the environment).  Started plainly as `python bench.py --gpus N` (no WORLD_SIZE) it spawns that
launcher itself as a child process -- before anything in this process touches the GPU -- and exits with its code.
Landmarks and the Cholesky of S are distributed along the elimination tree (a rank's column tiles are complete locally; the top tiles, two
n-vectors and the reduced gradient are all-reduced over RCCL inside the library), so total work is fixed
("scaling": "strong").
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md): 8 TB/s


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default=os.environ.get("APEX_BENCH_WORKLOAD", "final-13682"))
    ap.add_argument("--scale", type=float, default=1.0)
    ap.add_argument("--mode", default="selfcal", choices=["selfcal", "ba"])
    ap.add_argument("--variant", default="sparse", choices=["sparse", "iterative", "implicit"])
    ap.add_argument("--cg", default="", help="max_iter,tol of the PCG variants (default: the reference's 200,1e-6 / implicit 500,1e-9)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-variants", action="store_true", help="skip the Iterative / matrix-free runs that follow the timed region of a Sparse run")
    ap.add_argument("--no-other-workloads", action="store_true", help="skip the short runs of the other BASELINE.json configurations that follow the headline's timed region")
    ap.add_argument("--comm", choices=("auto", "rccl", "shm"), default="auto", help="N > 1: the library's transport (auto: RCCL, host shared memory if RCCL cannot be initialised)")
    ap.add_argument("--opt", action="append", default=[], help="implementation switch name=value (apexgpu_set_option), repeatable")
    ap.add_argument("--cpu-sample-scale", type=float, default=0.0, help="0: full size when the reference's dense S fits (<= 2300 cameras), else a ~1000-camera sample")
    return ap.parse_args()


def lm_step(s, state):
    """One LM iteration through the C ABI (mirrors Solver::lm_optimize / the reference loop)."""
    s.solve_augmented_equation(state["lam"], want_step=False)
    state["pcg"].append(s.info()["pcg_iterations"])
    gn, sn, pred = s.step_stats()
    new_cost = s.eval_step()
    actual = state["cost"] - new_cost
    rho = (1.0 if actual > 0 else 0.0) if abs(pred) < 1e-15 else actual / pred
    if rho > 0.0:
        coff = 2.0 * rho - 1.0
        state["lam"] = max(state["lam"] * max(1.0 / 3.0, 1.0 - coff**3), 1e-12)
        state["nu"] = 2.0
        state["cost"] = new_cost
        s.commit_step()
        state["accepted"] += 1
    else:
        state["lam"] = min(state["lam"] * state["nu"], 1e12)
        state["nu"] *= 2.0
        s.discard_step()
    state["hist"].append((state["cost"], rho))


def usable_cores():
    """Hardware threads this process may actually run on: the affinity mask, cut by the cgroup CPU quota (a container on a
    256-thread host may be allowed far fewer: 256 OpenMP threads spinning on 32 CPUs' worth of quota is how a baseline
    gets SLOWER with "more cores")."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = min(n, max(1, int(float(q) / float(p) + 0.5)))
    except Exception:
        try:
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read()); p = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, int(q / p + 0.5)))
        except Exception:
            pass
    return n


def cpu_baseline_worker(workload, shape_scale, mode, dense_too=True, n_iter=2):
    """Runs in a child process (OMP_NUM_THREADS / OMP_PROC_BIND set by the parent): 2 LM iterations of the oracle with the
    SPARSE solve -- S sparsified at 1e-12 and factorised inside the envelope of a fill-reducing order, the reference's
    solve_with_cholesky contract (explicit_schur.rs:913-921, 544-550) -- and, beside it, one iteration with the dense LL^T
    that round 1-3 timed (it overstated the reference's cost on banded shapes)."""
    import apex_solver_amd as pkg
    from oracle import oracle as ora

    cores = int(os.environ.get("OMP_NUM_THREADS", os.cpu_count() or 1))
    d = pkg.datasets.load_named(workload, shape_scale)[0]
    lay = pkg.layout.reference_column_layout(d.n_cam, d.n_pt)
    o = ora.from_data(d, lay, mode=mode, native=True)

    def iterate(variant, n_iter):
        cost = o.residuals()[0]
        lam, t_solve = 1e-3, 0.0
        t0 = time.perf_counter()
        for _ in range(n_iter):
            o.linearize()
            t1 = time.perf_counter()
            step, grad = o.solve_augmented(lam, variant)
            t_solve += time.perf_counter() - t1
            o.apply_step(step, 1.0)
            new_cost = o.residuals()[0]
            if new_cost < cost:
                cost = new_cost
                lam = max(lam / 3.0, 1e-12)
            else:
                o.apply_step(step, -1.0)
                lam *= 2.0
        return (time.perf_counter() - t0) * 1e3 / n_iter, t_solve * 1e3 / n_iter

    p0 = o.get_params()
    ms, ms_solve = iterate(3, n_iter)
    stats = o.last_sparse_stats()
    out = {
        "value": ms, "unit": "ms per LM iter on the sample", "cores": cores, "kind": "port", "solve": "sparse",
        "sample": f"{d.name}: {d.n_cam} cameras / {d.n_pt} landmarks / {d.n_obs} observations; SINGLE-THREADED envelope Cholesky "
                  f"(the solve is serial: only the linearisation and the Schur formation use the cores); {n_iter} LM iteration(s) of oracle/ba_oracle.c "
                  f"(linearise + explicit Schur into a dense S + sparsify at 1e-12 + envelope Cholesky in reverse Cuthill-McKee order + trial cost)",
        "obs_per_s": d.n_obs / (ms * 1e-3),
        "solve_ms": ms_solve,   # H = J^T J, Schur complement, factorisation, back-substitution
        "sparse_s": dict(stats, n=o.cam_dof),
    }
    if dense_too:
        o.set_params(*p0)
        dms, dsolve = iterate(0, 1)
        out["dense_solve"] = {"value": dms, "solve_ms": dsolve, "note": "the same iteration with a DENSE LL^T of S (rounds 1-3's baseline)"}
    return out


def cpu_baseline(args, shape_scale, mode, full_size):
    """The oracle (kind "port": C restatement of the reference CPU path, OpenMP) timed on this box's host cores.

    full_size: the workload's dense S fits (ladybug-1723, venice-1778: ~2 GB) -- the oracle runs the SAME problem the GPU
    line is quoted on.  Otherwise (final-13682: the reference forms S as a dense n_c x n_c matrix, explicit_schur.rs:782,
    = 121 GB at 9 DOF per camera) a bounded sample of the same generator, and the line says so.
    Threads: `value` at min(usable, 32) (the restated path's Schur formation is serial like the reference's,
    explicit_schur.rs:801-898, so it stops scaling there), plus ONE thread and ALL usable cores (SURVEY §8d asks for both)
    on a third of the sample.  usable = affinity mask cut by the cgroup quota (usable_cores()); threads sleep at barriers
    (OMP_WAIT_POLICY=passive) instead of spinning.  Each run is its own child process so that the OpenMP runtime starts
    with the wanted thread count."""
    import subprocess

    cores = usable_cores()

    def run(threads, sc, dense_too=False, n_iter=2):
        env = dict(os.environ, OMP_NUM_THREADS=str(threads), OMP_PROC_BIND="false", OMP_WAIT_POLICY="passive", OMP_DYNAMIC="false")
        code = (f"import sys, json; sys.path.insert(0, {ROOT!r}); import bench; "
                f"print('CPUBASE ' + json.dumps(bench.cpu_baseline_worker({args.workload!r}, {sc!r}, {mode!r}, {dense_too!r}, {n_iter!r})))")
        p = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=900)
        line = [l for l in p.stdout.splitlines() if l.startswith("CPUBASE ")]
        if not line:
            raise RuntimeError(p.stderr[-500:])
        return json.loads(line[0][8:])

    out = run(min(cores, 32), shape_scale, True)
    out["usable_cores"] = cores
    out["hardware_threads"] = os.cpu_count() or 1
    if full_size:
        out["sample"] = "FULL SIZE, the GPU line's own problem -- " + out["sample"]
    else:
        out["sample"] += (f"; a SAMPLE of the same generator: {args.workload} itself is beyond the reference's CPU path, which forms S as a"
                          " dense n_c x n_c matrix (explicit_schur.rs:782) -- 121 GB at 13,682 cameras x 9 DOF")
    keep = ("value", "unit", "cores", "sample", "obs_per_s", "solve", "solve_ms")
    # ONE thread on the SAME sample as `value` where that sample is the bounded one (round 5: a third of it until then, which made
    # the two figures incomparable), one LM iteration; the full-size configurations keep a sixth (their one-thread run would
    # take minutes)
    third = shape_scale if not full_size else shape_scale / 6.0
    out["one_thread"] = {k: v for k, v in run(1, third, False, 1 if not full_size else 2).items() if k in keep}
    if cores > 32:
        out["all_cores"] = {k: v for k, v in run(cores, shape_scale / 3.0 if not full_size else shape_scale / 6.0).items() if k in keep}
    if not full_size and out["one_thread"].get("value"):   # (same sample: comparable)
        out["thread_scaling"] = out["one_thread"]["value"] / out["value"]   # x faster on `cores` threads than on one
        out["one_thread_ms"] = out["one_thread"]["value"]
    return out


def factor_roofline(info, f_ms, note=None):
    """The tile Cholesky against the fp64 MFMA peak (78.6 TF/s, MI355X_MICROARCH.md): flops of one factorisation from the
    plan's own operation counts (apexgpu_info[9..11]) -- a panel product or trailing update is 2 x 144^3, a diagonal tile's
    Cholesky + triangular inverse 2/3 x 144^3 -- `achieved` on ALL of them (comparable across rounds), `executed` without the
    36 of 81 block products per panel solve that multiply by the zero blocks of the triangular inverse."""
    t3 = 144.0 ** 3
    flops = 2.0 * t3 * (info["n_trsm"] + info["n_update"]) + info["n_potrf"] * (2.0 / 3.0) * t3
    executed = 2.0 * t3 * (info["n_trsm"] * 45.0 / 81.0 + info["n_update"]) + info["n_potrf"] * (2.0 / 3.0) * t3
    ach = flops / (f_ms * 1e-3) / 1e12 if f_ms > 0 else 0.0
    out = {"bound": "mfma", "kernel": "tile Cholesky of S (k_potrf_inv_mf + k_tile_gemm_nt + k_factor_flow)", "achieved": ach, "peak": 78.6,
           "unit": "TFLOP/s", "frac": ach / 78.6, "traffic": None, "flops_per_factorisation": flops,
           "flops_executed": executed, "executed_tflops": executed / (f_ms * 1e-3) / 1e12 if f_ms > 0 else 0.0,
           "tile_ops": {"potrf": info["n_potrf"], "panel_products": info["n_trsm"], "updates": info["n_update"]},
           "etree_levels": info["etree_levels"], "avg_factor_ms": f_ms}
    if note:
        out["note"] = note
    return out


def run_pose_graph(args, torch, dist, world, pg_dev, dev, steps, warmup, scale=1.0, workload="sphere2500", cpu_base=True, timing_barrier=True):
    """One pose-graph measurement (the body of bench_pose_graph; also a line of `other_workloads`)."""
    import apex_solver_amd as pkg
    from apex_solver_amd.pose_graph import GpuSparseCholeskySolver, PoseGraphProblem

    rank = int(os.environ.get("RANK", "0"))
    side = max(2, int(round(50 * scale ** 0.5)))
    if scale == 1.0:
        d, data_kind, data_src = pkg.datasets.load_pose_graph(workload, side, side)   # data/odometry/3d/sphere2500.g2o when present
    else:
        d, data_kind, data_src = pkg.synthetic.make_sphere(side, side), "synthetic", None
    prob = PoseGraphProblem.pose_graph(d)
    s = GpuSparseCholeskySolver(dev).initialize_structure(prob)
    s.set_parameters(d.poses)
    info = s.info()
    st8 = dict(lam=1e-3, nu=2.0, cost=s.compute_cost(), accepted=0)
    initial_cost = st8["cost"]

    def step():
        s.solve_augmented_equation(st8["lam"], want_step=False)
        gn, sn, pred = s.step_stats()
        nc = s.eval_step()
        actual = st8["cost"] - nc
        rho = (1.0 if actual > 0 else 0.0) if abs(pred) < 1e-15 else actual / pred
        if rho > 0.0:
            coff = 2.0 * rho - 1.0
            st8["lam"] = max(st8["lam"] * max(1.0 / 3.0, 1.0 - coff ** 3), 1e-12); st8["nu"] = 2.0
            st8["cost"] = nc; s.commit_step(); st8["accepted"] += 1
        else:
            st8["lam"] = min(st8["lam"] * st8["nu"], 1e12); st8["nu"] *= 2.0; s.discard_step()

    for _ in range(warmup):
        step()

    def barrier():
        torch.cuda.synchronize()
        if world > 1 and timing_barrier:
            dist.barrier()
        torch.cuda.synchronize()

    # (inside the timed region the factorisation alone is timed -- this workload's roofline; the other stages over a few
    # iterations behind it: see main())
    s.enable_stage_timing(2 << pkg.capi.PG_STAGE_NAMES.index("factor")); s.reset_stage_times()
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1 and timing_barrier:
        t = torch.tensor([elapsed], dtype=torch.float64, device=pg_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    stages_timed = s.stage_times()
    ms = elapsed * 1e3 / steps
    f_ms = stages_timed["factor"][0] / max(stages_timed["factor"][1], 1)
    cost_region, accepted_region = st8["cost"], st8["accepted"]
    stage_steps = max(2, min(5, steps))
    s.enable_stage_timing(True); s.reset_stage_times()
    for _ in range(stage_steps):
        step()
    barrier()
    stages = s.stage_times()
    per_step = {k: v[0] / stage_steps for k, v in stages.items()}
    per_step["factor"] = stages_timed["factor"][0] / steps
    out = {"metric": "ms per LM iter (Jacobian+JtJ+Cholesky)", "value": ms, "unit": "ms", "n_gpus": world, "steps": steps,
           "warmup": warmup, "ms_per_step": ms, "higher_is_better": False, "scaling": "weak", "vs_baseline": None,
           "dtype": "f64", "data": data_kind,
           "config": {"workload": f"{d.name} {data_kind} SE3 pose graph ({d.n_v} vertices / {d.n_e} edges)" + (f" from {data_src}" if data_src else ""), "tile_rows": info["tile_rows"],
                      "tiles": info["tiles"], "etree_levels": info["etree_levels"], "parallelism": f"replicas x{world}"},
           "roofline": factor_roofline(info, f_ms, "17-68 dependent levels: latency-bound at this size"),
           "stages_ms_per_step": per_step,
           "initial_cost": initial_cost, "final_cost": cost_region, "accepted_steps": accepted_region}
    if rank == 0 and world == 1 and cpu_base:
        try:
            from oracle import pg_oracle as po
            o = po.PgOracle.from_problem(prob)
            t0 = time.perf_counter()
            o.linearize(); rc, stp, _ = o.solve_augmented(1e-3); o.apply_step(stp, 1.0); o.residuals()
            out["cpu_baseline"] = {"value": (time.perf_counter() - t0) * 1e3, "unit": "ms per LM iter", "cores": 1,
                                   "kind": "port", "sample": f"the same {d.name} graph, 1 LM iteration of oracle/pg_oracle.c (envelope Cholesky, single-threaded solve)"}
        except Exception as e:
            out["cpu_baseline"] = {"error": repr(e)}
    s.close()
    return out


def bench_pose_graph(args):
    """BASELINE configs[1]: sphere2500-shaped SE3 pose graph, block-sparse J^T J + tile Cholesky (no Schur),
    replicas only (N > 1 runs N independent copies).  Same JSON contract; the dominant stage is the
    factorisation, so the roofline is priced against the dense fp64 MFMA peak."""
    import torch

    import apex_solver_amd as pkg
    from apex_solver_amd.pose_graph import GpuSparseCholeskySolver, PoseGraphProblem

    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback)")
    torch.cuda.set_device(local_rank % torch.cuda.device_count())
    dev, pg_dev = torch.cuda.current_device(), "cuda"
    if world > 1:
        import torch.distributed as dist
        dev, pg_dev = process_group(torch, local_rank)
    out = run_pose_graph(args, torch, dist if world > 1 else None, world, pg_dev, dev, args.steps, args.warmup, args.scale, args.workload,
                         cpu_base=not args.no_cpu_baseline)
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


def quick_ba(torch, dev, workload, variant, steps, warmup, mode="selfcal"):
    """A short run of another BASELINE.json configuration on this GPU, AFTER the timed region of the headline (a line of
    `other_workloads`): the same step as main() -- solve_augmented + step statistics + trial cost + accept / reject --
    `warmup` untimed and `steps` timed iterations, synchronised on both sides."""
    import apex_solver_amd as pkg
    from apex_solver_amd.solver import GpuSchurComplementSolver, OptimizationType, Problem, SchurVariant

    d, data_kind, data_src = pkg.datasets.load_named(workload, 1.0)
    ot = OptimizationType.SelfCalibration if mode == "selfcal" else OptimizationType.BundleAdjustment
    prob = Problem.bundle_adjustment(d, ot, 1.0)
    s = GpuSchurComplementSolver(dev)
    s.with_variant({"sparse": SchurVariant.Sparse, "iterative": SchurVariant.Iterative, "implicit": SchurVariant.Implicit}[variant])
    if variant == "implicit":
        s.with_cg_params(500, 1e-9)            # IterativeSchurSolver::new (implicit_schur.rs:94-95)
        s.with_option("matrix_free_only", 1)   # S is never formed
    t0 = time.perf_counter()
    s.initialize_structure(prob)
    s.set_parameters(d.poses, d.intr, d.points)
    setup_s = time.perf_counter() - t0
    info = s.info()
    state = dict(lam=1e-3, nu=2.0, cost=s.compute_cost(), accepted=0, hist=[], pcg=[])
    c0 = state["cost"]
    for _ in range(warmup):
        lm_step(s, state)
    # (as in main(): inside the timed region one stage is timed -- the factorisation resp. the PCG solve --, the rest behind it)
    s.enable_stage_timing(2 << pkg.capi.STAGE_NAMES.index("factor")); s.reset_stage_times()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        lm_step(s, state)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3 / steps
    stages_timed = s.stage_times()
    f_ms = stages_timed["factor"][0] / max(stages_timed["factor"][1], 1)
    cost_region, accepted_region, n_pcg_region = state["cost"], state["accepted"], len(state["pcg"])
    stage_steps = 2
    s.enable_stage_timing(True); s.reset_stage_times()
    for _ in range(stage_steps if variant == "sparse" else 0):   # (the matrix-free variant: 0.25 s per iteration -- its one stage is timed above)
        lm_step(s, state)
    torch.cuda.synchronize()
    stages = s.stage_times() if variant == "sparse" else stages_timed
    per_step = {k: round(v[0] / (stage_steps if variant == "sparse" else steps), 4) for k, v in stages.items()}
    per_step["factor"] = round(stages_timed["factor"][0] / steps, 4)
    out = {"ms_per_lm_iter": ms, "steps": steps, "warmup": warmup, "data": data_kind, "schur_variant": variant,
           "workload": f"{d.name} {data_kind} ({d.n_cam} cameras / {d.n_pt} landmarks / {d.n_obs} observations)" + (f" from {data_src}" if data_src else ""),
           "factor_ms": f_ms, "stages_ms_per_step": per_step, "setup_s": setup_s,
           "initial_cost": c0, "final_cost": cost_region, "accepted_steps": accepted_region, **s.variant_info()}
    if variant == "sparse":
        fr = factor_roofline(info, f_ms)
        out["factor_roofline"] = {k: fr[k] for k in ("achieved", "peak", "unit", "frac", "flops_per_factorisation", "etree_levels")}
    else:
        out["pcg_iterations_per_step"] = state["pcg"][warmup:n_pcg_region]
        out["factor_ms_is"] = "the PCG solve (no factorisation in this variant)"
    s.close()
    return out


def process_group(torch, local_rank):
    """torch.distributed for the barrier / max-over-ranks of the contract: RCCL, one rank per GPU.  Bring-up on a box with
    fewer GPUs than ranks (APEX_BENCH_PG=gloo): the ranks share the devices round-robin and the group runs over gloo with
    host tensors -- the library's own RCCL communicator then cannot be built and the shared-memory transport takes over,
    which is how the fall-back of main() is exercised on one GPU.  Returns (device index, device for group tensors)."""
    import torch.distributed as dist
    if os.environ.get("APEX_BENCH_PG", "nccl") == "gloo":
        dev = local_rank % torch.cuda.device_count()
        torch.cuda.set_device(dev)
        dist.init_process_group(backend="gloo")
        return dev, "cpu"
    torch.cuda.set_device(local_rank)
    dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    return local_rank, "cuda"


def spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher: run N ranks of this very command under torch.distributed.run as a
    CHILD process (never an exec: this process may not replace itself once a GPU runtime is loaded, and it has not
    touched the GPU yet -- torch.cuda.device_count() does not initialise it) and pass its exit code on."""
    import socket
    import subprocess

    import torch

    have = torch.cuda.device_count()
    if have < n and os.environ.get("APEX_BENCH_PG", "nccl") != "gloo":   # (gloo bring-up: the ranks share the devices)
        raise SystemExit(f"bench.py --gpus {n}: this node shows {have} GPU(s)")
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC only on this pool (RCCL needs it)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]]
    raise SystemExit(subprocess.run(cmd, env=env).returncode)


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        spawn_ranks(args.gpus)
    if args.workload.startswith("sphere"):
        return bench_pose_graph(args)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")

    import torch
    import torch.distributed as dist

    import apex_solver_amd as pkg
    from apex_solver_amd.solver import GpuSchurComplementSolver, OptimizationType, Problem, SchurVariant

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback)")
    torch.cuda.set_device(local_rank % torch.cuda.device_count())
    dev, pg_dev = torch.cuda.current_device(), "cuda"
    if world > 1:
        dev, pg_dev = process_group(torch, local_rank)

    d, data_kind, data_src = pkg.datasets.load_named(args.workload, args.scale)   # real BAL file when data/... holds it
    ot = OptimizationType.SelfCalibration if args.mode == "selfcal" else OptimizationType.BundleAdjustment
    prob = Problem.bundle_adjustment(d, ot, 1.0)
    s = GpuSchurComplementSolver(dev)
    s.with_variant({"sparse": SchurVariant.Sparse, "iterative": SchurVariant.Iterative, "implicit": SchurVariant.Implicit}[args.variant])
    if args.cg:
        s.with_cg_params(int(args.cg.split(",")[0]), float(args.cg.split(",")[1]))
    elif args.variant == "implicit":
        s.with_cg_params(500, 1e-9)  # IterativeSchurSolver::new (implicit_schur.rs:94-95)
    if args.variant == "implicit":
        s.with_option("matrix_free_only", 1)   # S is never formed: no tile structure beyond the diagonal, no pair list
    shared_gpu = world > 1 and os.environ.get("APEX_BENCH_PG", "nccl") == "gloo" and world > torch.cuda.device_count()
    if shared_gpu:
        # Bring-up mode only (several ranks on ONE GPU): two processes time-slice the device, and a launch whose workgroups wait
        # for each other's flags (the dataflow sweeps, the dataflow top of the factorisation) can then starve until its bounded
        # wait gives up -- handled (the solve is repeated by level launches, the handle stays on them), but each time-out costs
        # ~1 s and a second one fails the solve.  One rank per GPU, the production layout, keeps the dataflow launches.
        s.with_option("tri_dataflow", 0).with_option("factor_flow", 0)
    for o in args.opt:
        s.with_option(o.split("=")[0], int(o.split("=")[1]))
    comm_kind = "none"
    if world > 1:
        import ctypes as C

        def fresh_unique_id():
            uid = torch.zeros(128, dtype=torch.uint8, device=pg_dev)
            if rank == 0:
                buf = (C.c_char * 128)()
                rc = pkg.capi.load().apexgpu_get_unique_id(C.cast(buf, C.c_void_p))
                assert rc == 0
                uid = torch.frombuffer(bytearray(bytes(buf)), dtype=torch.uint8).to(pg_dev)
            dist.broadcast(uid, 0)
            return bytes(uid.cpu().numpy().tobytes())

        # The library's own RCCL communicator (csrc/comm.cpp) is probed on a throw-away handle first: if any rank cannot
        # join it, every rank takes the host shared-memory transport instead (one node: csrc/comm.h) and the line says so --
        # slower exchanges, same schedule, and a record instead of a dead run.  --comm rccl / shm force one or the other.
        # Two votes: the LOCAL preconditions first (a handle on this rank's device), so that a rank that fails before the
        # collective ncclCommInitRank cannot leave the others blocked inside it; then the collective initialisation itself.
        # (A rank that dies INSIDE ncclCommInitRank still hangs its peers until RCCL's own time-out: --comm shm is the way out.)
        ok, why, probe = 1, "", None
        if args.comm != "shm":
            try:
                probe = pkg.capi.Handle(1, 1, 1, 0, dev)
            except Exception as e:   # noqa: BLE001
                ok, why = 0, repr(e)[:200]
        else:
            ok = 0
        vote = torch.tensor([ok], dtype=torch.int32, device=pg_dev)
        dist.all_reduce(vote, op=dist.ReduceOp.MIN)
        if int(vote.item()) == 1:
            try:
                buf = (C.c_char * 128).from_buffer_copy(fresh_unique_id())
                probe.check(probe.L.apexgpu_comm_init(probe.h, world, rank, C.cast(buf, C.c_void_p)))
            except Exception as e:   # noqa: BLE001 -- whatever it is, the vote below decides
                ok, why = 0, repr(e)[:200]
            vote = torch.tensor([ok], dtype=torch.int32, device=pg_dev)
            dist.all_reduce(vote, op=dist.ReduceOp.MIN)
        if probe is not None:
            probe.close()
        if int(vote.item()) == 1:
            s.with_communicator(world, rank, fresh_unique_id()); comm_kind = "rccl"
        elif args.comm == "rccl":
            raise SystemExit(f"rank {rank}: the RCCL communicator could not be built ({why or 'another rank failed'})")
        else:
            tag = torch.tensor([os.getpid()], dtype=torch.int64, device=pg_dev)
            dist.broadcast(tag, 0)
            s.with_shm_communicator(world, rank, f"bench-{int(tag.item())}")
            comm_kind = "shm" if args.comm == "shm" else "shm (RCCL communicator failed: " + (why or "on another rank") + ")"
        if shared_gpu:
            comm_kind += "; ranks share a GPU: level launches instead of the dataflow launches"
    t_setup = time.perf_counter()
    s.initialize_structure(prob)
    s.set_parameters(d.poses, d.intr, d.points)
    setup_s = time.perf_counter() - t_setup
    info = s.info()

    state = dict(lam=1e-3, nu=2.0, cost=s.compute_cost(), accepted=0, hist=[], pcg=[])
    initial_cost = state["cost"]
    for _ in range(args.warmup):
        lm_step(s, state)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # Inside the timed region only the two stages with a roofline of their own are timed (HIP events on the solver's stream around
    # k_schur_pairs_r, the graded kernel, and around the factorisation: roofline.avg_launch_ms and factor.avg_factor_ms are measured
    # live, over these very steps); the eighteen events per iteration of the full stage table cost the stream ~0.1 ms per
    # iteration (round 5: 12.88 against 12.69 ms, three pairs of runs), so the other stages are timed over `stage_steps` further
    # iterations right behind the timed region (a later LM state: `stage_sum_ms` beside `value` says how far the table is off).
    sc_stage, fa_stage = pkg.capi.STAGE_NAMES.index("schur_scatter"), pkg.capi.STAGE_NAMES.index("factor")
    s.enable_stage_timing((2 << sc_stage) | ((2 << fa_stage) if args.variant == "sparse" else 0))
    s.reset_stage_times()
    accepted_before = state["accepted"]
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        lm_step(s, state)
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=pg_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    stages_timed = s.stage_times()
    ms_per_step = elapsed * 1e3 / args.steps
    accepted_in_region = state["accepted"] - accepted_before
    final_cost_region = state["cost"]
    stage_steps = max(2, min(5, args.steps))
    saved_params, saved_state = s.get_parameters(), {k: (list(v) if isinstance(v, list) else v) for k, v in state.items()}
    s.enable_stage_timing(True)
    s.reset_stage_times()
    for _ in range(stage_steps):
        lm_step(s, state)
    barrier()
    stages = s.stage_times()
    stages["schur_scatter"] = stages_timed["schur_scatter"]   # (the graded kernel: from the timed region itself)
    factor_in_region = args.variant == "sparse" and stages_timed["factor"][1] > 0
    if factor_in_region:
        stages["factor"] = stages_timed["factor"]
    # (what follows -- the other variants on this handle -- starts from the end of the timed region, as in rounds 1-4)
    s.set_parameters(*saved_params)
    state.clear(); state.update(saved_state)

    # ---- roofline of the graded Schur-reduction kernel, per launch -------------------------------------------------------
    dc = 9 if args.mode == "selfcal" else 6
    form = info.get("schur_form", 3)
    kernel = "k_schur_pairs_r"   # (form 4: the queued layout of the pair list, 3: one running block per wave)
    record_form = True
    n_obs_local = info["local_obs"]
    tile_bytes = 144 * 144 * 8
    # SURVEY §8(d), fused form (J never stored): each input read once, each output written once.  The record form (the
    # default pair kernel) reads a 32-byte projection record per observation instead of the 24-byte observation record;
    # the figure below stays the survey's fused-form one (the smaller of the two), so `frac` is comparable across rounds.
    alg_bytes = (24.0 * n_obs_local            # observation stream: cam idx, landmark idx, (u,v)
                 + 80.0 * d.n_cam + 24.0 * d.n_pt  # poses + intrinsics, points (each read once)
                 + 96.0 * d.n_pt                # Hll^-1 and g_l per landmark
                 + tile_bytes * info["touched_tiles"]  # S tiles that receive contributions
                 + 8.0 * dc * d.n_cam)          # g_red
    st = s.setup_times()
    pair_list_bytes = 16.0 * st["pair_slots"]   # the sorted pair list the default form streams (an index, like cam/pt idx)
    sc_ms, sc_n = stages["schur_scatter"]
    sc_avg = sc_ms / max(sc_n, 1)
    achieved = alg_bytes / (sc_avg * 1e-3) / 1e9 if sc_avg > 0 else 0.0
    # useful fp64 work of the reduction: every camera pair of a landmark once (rank-2 form: 12 + 36 + 162 FMA) and every
    # observation linearised once (~125 fp64 operations); the kernel re-linearises both observations of every pair
    off_pairs = info["pair_blocks"] - n_obs_local      # pair contributions without the self pairs
    useful_flop = 2.0 * (12 + 4 * dc + 2 * dc * dc) * off_pairs + 250.0 * n_obs_local
    executed_flop = ((2.0 * (12 + 4 * dc + 2 * dc * dc) + 2 * 120.0 + 36.0) if record_form else (2.0 * (12 + 4 * dc + 2 * dc * dc) + 2 * 250.0 + 36.0)) * off_pairs
    in_region = ("schur_scatter",) + (("factor",) if factor_in_region else ())
    per_step = {k: v[0] / (args.steps if k in in_region else stage_steps) for k, v in stages.items()}   # ms per LM iteration
    stage_ms = per_step["landmark_reduce"] + per_step["cam_reduce"] + sc_avg
    roofline = {"bound": "hbm", "kernel": kernel, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": None, "algorithmic_bytes": alg_bytes,
                "pair_list_bytes": pair_list_bytes if record_form else 0.0,
                "avg_launch_ms": sc_avg, "launches": sc_n,
                "actual_bound": ("the L1's window of outstanding misses: 2.75 64-byte L1->L2 requests per pair (landmark header + two projection "
                                 "records, compulsory in a pair-major kernel) = 16.9 GB per launch at 5.9 TB/s, ~62 requests in flight per CU at "
                                 "418 cycles each, the L1 stalled on pending data 63 % of its busy cycles: 0.5 ms of the launch; the other 2.3 ms are the "
                                 "vector and LDS phases of two waves per SIMD (the kernel without any gather: 2.33-2.37 ms) "
                                 "(profiles/r06_pairs_mem_counters.txt; DESIGN.md section 4)") if record_form else
                                "fp64 vector unit + LDS (the fused form moves 2 GB but executes ~60 GFLOP: DESIGN.md section 4)",
                "projection_record_bytes": 32.0 * n_obs_local if record_form else 0.0,
                "pair_contributions_per_launch": off_pairs,
                "fp64_useful_gflops": useful_flop / (sc_avg * 1e-3) / 1e9 if sc_avg > 0 else 0.0,
                "fp64_executed_gflops": executed_flop / (sc_avg * 1e-3) / 1e9 if sc_avg > 0 else 0.0,
                "fp64_vector_peak_gflops": 78600.0,
                "fp64_useful_frac": useful_flop / (sc_avg * 1e-3) / 78.6e12 if sc_avg > 0 else 0.0,
                # the roof that BINDS this kernel at d_c = 9: 40 % of HBM on the algorithmic bytes would need 75 of the chip's 78.6 TF/s
                # of fp64 (matrix and vector instructions share that pipe) -- DESIGN.md section 4
                "fp64_pipe_frac": useful_flop / (sc_avg * 1e-3) / 78.6e12 if sc_avg > 0 else 0.0,
                "schur_stage": {"kernels": "k_landmark_reduce + k_cam_reduce + " + kernel, "ms": stage_ms,
                                "achieved": alg_bytes / (stage_ms * 1e-3) / 1e9 if stage_ms > 0 else 0.0,
                                "frac": alg_bytes / (stage_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if stage_ms > 0 else 0.0}}

    # HBM traffic of the same kernel from the committed PMC pass (rocprofv3 cannot run inside this
    # process); only attached when the committed profile is of this very workload and kernel
    try:
        pm_path = next(p for p in (os.path.join(ROOT, "profiles", f"r0{r}_final13682_pmc_summary.json") for r in (6, 5, 4, 3, 2)) if os.path.exists(p))
        pm = json.load(open(pm_path))
        if world == 1 and args.workload == "final-13682" and args.scale == 1.0 and args.mode == "selfcal":
            k = [v for n, v in pm["kernels"].items() if kernel in n]
            if k:
                roofline["traffic"] = k[0]["hbm_bytes_per_launch_corrected"]
                roofline["traffic_raw"] = k[0].get("hbm_bytes_per_launch_raw")
                roofline["traffic_source"] = os.path.relpath(pm_path, ROOT)
    except Exception:
        pass

    fac = (factor_roofline(info, stages["factor"][0] / max(stages["factor"][1], 1),
                           "levels 0-8 at the tile GEMM's rate, the middle levels and the top latency-bound (profiles/r06_factor_timeline.txt)")
           if args.variant == "sparse" and not s.variant_info()["reason"] else None)
    if fac:   # (scalars the driver's record keeps: the factorisation against ITS roof, the fp64 MFMA peak)
        roofline["factor_ms"] = fac["avg_factor_ms"]
        roofline["factor_mfma_frac"] = fac["frac"]
    out = {
        "metric": "ms per LM iter (Jacobian+Schur+solve)", "value": ms_per_step, "unit": "ms", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": False,
        "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": data_kind,
        "config": {"workload": f"{d.name} {data_kind} ({d.n_cam} cameras / {d.n_pt} landmarks / {d.n_obs} observations)" + (f" from {data_src}" if data_src else ""),
                   "optimization_type": args.mode, "camera_dof": dc, "schur_variant": args.variant, "huber": 1.0,
                   "s_tile_rows": info["tile_rows"], "s_tiles": info["tiles"], "s_tiles_touched": info["touched_tiles"],
                   "etree_levels": info["etree_levels"], "border_cameras": st["hub_cameras"], "schur_form": form, "setup_s": setup_s,
                   "host_cache_bytes": int(pkg.capi.load().apexgpu_host_cache_bytes()), **s.variant_info(), **({"transport": comm_kind} if world > 1 else {}), "parallelism": f"landmark-shard x{world}" + (f", landmarks and Cholesky distributed by elimination subtree ({info['dist_top_columns']} shared top tile columns, tree_sharded={info.get('tree_sharded')})" if info.get("dist_top_columns") else "")},
        "roofline": roofline,
        # the stage that is half the step, against ITS roof (fp64 MFMA); only where a factorisation ran
        **({"factor": fac} if fac else {}),
        "stages_ms_per_step": per_step,
        "stage_launches": {k: int(v[1]) for k, v in stages.items()},
        "stage_sum_ms": sum(per_step.values()),
        "stages_measured": f"schur_scatter (the graded kernel) and factor: HIP events inside the timed region, {args.steps} iterations; the other stages: {stage_steps} "
                           "further iterations right behind it with every stage event on (the full table costs the stream ~0.1 ms per iteration)",
        "setup_s": setup_s, "setup_by_phase_s": {k: st[k] for k in ("order", "lists", "tile_plan", "schur_lists", "uploads", "total")},
        # the whole of setup_s by piece: handle creation (the process's first HIP call = runtime start-up), the host-side
        # argument arrays, apexgpu_set_structure (= setup_by_phase_s.total + its argument checks), the parameter upload
        "setup_wall_s": dict(getattr(s, "setup_wall", {})),
        "initial_cost": initial_cost, "final_cost": final_cost_region, "accepted_steps": accepted_in_region,
        "accepted_steps_incl_warmup": state["accepted"],
        "obs_per_s": d.n_obs / (ms_per_step * 1e-3),
    }
    if args.variant != "sparse":
        out["pcg_iterations_per_step"] = state["pcg"][args.warmup:args.warmup + args.steps]
    elif world == 1 and not args.no_other_variants:
        # OUTSIDE the timed region, same handle, same problem, the optimisation simply goes on: the reference's BA default
        # (SchurVariant::Iterative, 200 Jacobi-PCG iterations at 1e-6 on the explicit S: levenberg_marquardt.rs:519-530 -- what
        # every BASELINE.md row was published with) and the matrix-free IterativeSchurSolver at ITS defaults (500 / 1e-9,
        # implicit_schur.rs:94-95), whose cost does not depend on the fill of S: the structure-independent bound of an LM
        # iteration on this problem.
        other = {}
        for key, var, cg in (("iterative_ms", SchurVariant.Iterative, (200, 1e-6)), ("fallback_implicit", SchurVariant.Implicit, (500, 1e-9))):
            try:
                s.with_variant(var).with_cg_params(*cg)
                lm_step(s, state)
                torch.cuda.synchronize()
                n0 = len(state["pcg"])
                s.reset_stage_times()
                t1 = time.perf_counter()
                for _ in range(2):
                    lm_step(s, state)
                torch.cuda.synchronize()
                ms2 = (time.perf_counter() - t1) * 1e3 / 2
                its = state["pcg"][n0:]
                solve_ms = s.stage_times()["factor"][0] / 2      # the PCG loop of the two steps (booked under the factor stage)
                rec = {"ms_per_lm_iter": ms2, "cg_max_iter": cg[0], "cg_tol": cg[1], "pcg_iterations": its, "steps": 2, "cost_after": state["cost"]}
                if var == SchurVariant.Implicit:
                    # What the handle happened to need (`observed_*`: it depends on how close to convergence the steps above
                    # left the problem) and the figure that does not: one PCG iteration x the reference's iteration cap
                    # (implicit_schur.rs:94) + the rest of the LM iteration = the structure-independent BOUND.
                    per_it = solve_ms / max(sum(its) / 2.0, 1.0)
                    rest = ms2 - solve_ms
                    rec.update({"ms_per_pcg_iter": per_it, "cap": cg[0], "bound_ms": per_it * cg[0] + rest, "observed_iterations": its,
                                "observed_ms": ms2, "rest_of_lm_iter_ms": rest})
                    out["fallback_implicit"] = {k: rec[k] for k in ("ms_per_pcg_iter", "cap", "bound_ms", "observed_iterations", "observed_ms")}
                else:
                    out[key] = ms2
                other[key] = rec
            except Exception as e:
                other[key] = {"error": repr(e)}
        s.with_variant(SchurVariant.Sparse).with_cg_params(200, 1e-6)
        s.reset_stage_times()
        out["other_variants"] = other
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        sc = args.cpu_sample_scale
        # the reference's dense S (explicit_schur.rs:782) of ladybug-1723 / venice-1778 is ~2 GB: the oracle runs those at
        # FULL size; beyond ~2,300 cameras (8 (9 n_cam)^2 > ~3.5 GB, and a Cholesky of minutes) a ~1000-camera sample
        full = sc <= 0.0 and d.n_cam <= 2300
        if sc <= 0.0:
            sc = 1.0 if full else min(1.0, 1000.0 / max(d.n_cam, 1))
        try:
            out["cpu_baseline"] = cpu_baseline(args, sc * args.scale, args.mode, full)
        except Exception as e:  # the baseline is a reported number, never a reason to lose the GPU line
            out["cpu_baseline"] = {"error": repr(e)}
    s.close()
    if (rank == 0 and world == 1 and not args.no_other_workloads and args.workload == "final-13682" and args.scale == 1.0
            and args.variant == "sparse" and args.mode == "selfcal"):
        # The other four BASELINE.json configurations, OUTSIDE the timed region (the headline's handle is closed): short runs
        # of the same step, so that the driver's record carries every configuration and not only builder-run profiles.
        ow = {}
        t_ow = time.perf_counter()
        for key, fn in (("ladybug-1723", lambda: quick_ba(torch, dev, "ladybug-1723", "sparse", 5, 2)),
                        ("venice-1778", lambda: quick_ba(torch, dev, "venice-1778", "sparse", 5, 2)),
                        ("synthetic-10k", lambda: quick_ba(torch, dev, "synthetic-10k", "sparse", 5, 2)),
                        ("synthetic-10k implicit", lambda: quick_ba(torch, dev, "synthetic-10k", "implicit", 3, 1)),
                        ("sphere2500", lambda: {k: v for k, v in run_pose_graph(args, torch, None, 1, "cuda", dev, 5, 2, cpu_base=False).items()
                                                if k in ("value", "config", "roofline", "stages_ms_per_step", "data", "initial_cost", "final_cost", "accepted_steps")})):
            try:
                ow[key] = fn()
            except Exception as e:   # a reported extra, never a reason to lose the headline
                ow[key] = {"error": repr(e)[:300]}
        if "value" in ow.get("sphere2500", {}):
            r = ow["sphere2500"]
            r["ms_per_lm_iter"] = r.pop("value"); r["factor_ms"] = r["roofline"]["avg_factor_ms"]
        ow["wall_s"] = time.perf_counter() - t_ow
        out["other_workloads"] = ow
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
