"""apex-solver_amd: MI355X-native bundle-adjustment inner loop behind apex-solver's
Problem / LevenbergMarquardt / LinearSolverType surface.

Host-side modules:
  synthetic  deterministic BA problem generator
  layout     the reference's lexicographic global column layout
  capi       ctypes binding of the C-ABI library (include/apexgpu.h)
  solver     Python mirror of the reference's Problem / LevenbergMarquardt surface
  pose_graph SE3 pose-graph path: G2O reader, PoseGraphProblem, GpuSparseCholeskySolver
  datasets   real BAL / G2O files in the reference's data/ layout when present, the synthetic shapes otherwise
"""
from . import layout, synthetic  # noqa: F401

__all__ = ["layout", "synthetic"]
from . import bal, capi, datasets, pose_graph, solver  # noqa: F401,E402
from .pose_graph import G2oLoader, GpuSparseCholeskySolver, PoseGraphProblem  # noqa: F401,E402
from .solver import (GpuSchurComplementSolver, LevenbergMarquardt, LevenbergMarquardtConfig,  # noqa: F401,E402
                     LinearSolverType, OptimizationStatus, OptimizationType, Problem, SchurVariant, SolverResult)
