//! GPU linearizer backend: MI355X (gfx950) bundle adjustment through `libapexgpu.so`.
//!
//! Fills the slot this module reserved ("GPU-accelerated Jacobian assembly backends").  The device never materialises
//! a Jacobian: `GpuBaMode::assemble` uploads the current variable values and hands the solver a [`DeviceJacobian`]
//! *handle*; `GpuSchurComplementSolver` (src/linalg/gpu_schur.rs) then linearises every projection factor, forms the
//! Schur complement, factorises and back-substitutes on the device in one call (`apexgpu_solve_augmented`).
//!
//! C ABI: `include/apexgpu.h` of the backend repository; every entry below cites the reference interface it replaces.
#![cfg(feature = "gpu")]

use std::collections::HashMap;
use std::ffi::CStr;
use std::os::raw::{c_char, c_int};
use std::sync::{Arc, Mutex, OnceLock};

use faer::sparse::SparseColMat;
use faer::Mat;

use crate::core::problem::{Problem, SymbolicStructure, VariableEnum};
use crate::linalg::{LinAlgError, LinAlgResult};
use crate::linearizer::cpu::LinearizationMode;
use crate::linearizer::{AssemblyBackend, LinearizerError, LinearizerResult};

// ---------------------------------------------------------------------------------------------------------------------
// FFI (include/apexgpu.h)
// ---------------------------------------------------------------------------------------------------------------------
#[repr(C)]
pub struct ApexGpuSolver {
    _private: [u8; 0],
}

pub const APEXGPU_MODE_BUNDLE_ADJUSTMENT: c_int = 0;
pub const APEXGPU_MODE_SELF_CALIBRATION: c_int = 1;
pub const APEXGPU_MODE_ONLY_POSE: c_int = 2;
pub const APEXGPU_MODE_ONLY_LANDMARKS: c_int = 3;
pub const APEXGPU_MODE_ONLY_INTRINSICS: c_int = 4;
pub const APEXGPU_MODE_POSE_AND_INTRINSICS: c_int = 5;
pub const APEXGPU_MODE_LANDMARKS_AND_INTRINSICS: c_int = 6;

extern "C" {
    pub fn apexgpu_create(n_cam: i64, n_pt: i64, n_obs: i64, mode: c_int, device: c_int, out: *mut *mut ApexGpuSolver) -> c_int;
    pub fn apexgpu_destroy(h: *mut ApexGpuSolver);
    pub fn apexgpu_last_error(h: *const ApexGpuSolver) -> *const c_char;
    pub fn apexgpu_set_structure(
        h: *mut ApexGpuSolver, cam_idx: *const u32, pt_idx: *const u32, obs_uv: *const f64, intr_col: *const i64,
        pose_col: *const i64, pt_col: *const i64, fix_pose: *const u8, fix_intr: *const u8, fix_pt: *const u8, huber_delta: f64,
    ) -> c_int;
    pub fn apexgpu_set_cg_params(h: *mut ApexGpuSolver, max_iterations: c_int, tolerance: f64) -> c_int;
    /// which variant a solve asked with `asked_variant` runs on this handle, and why (round 5: automatic selection of the
    /// matrix-free PCG when the tile plan of S is refused -- the CPU path never fails on the fill of S)
    pub fn apexgpu_variant_info(h: *mut ApexGpuSolver, asked_variant: c_int, used_variant: *mut c_int, reason: *mut c_char, reason_len: c_int) -> c_int;
    /// what `apexgpu_set_structure` predicted for the two ways to the step and which one it built (round 6): out[0] ms per solve of the
    /// direct factorisation, out[1] of the matrix-free PCG at its cap, out[2] the choice (0 direct, 1 matrix-free by cost, 2 by refusal, 3 by option)
    pub fn apexgpu_variant_costs(h: *mut ApexGpuSolver, out4: *mut f64) -> c_int;
    /// hand the set-up's cached host blocks back to the system at once (the `apexgpu_destroy` of the last live handle does
    /// it by itself: the cache only lives while a solver does); `apexgpu_host_cache_bytes`: what is held right now
    pub fn apexgpu_trim_host_cache(released_bytes: *mut i64) -> c_int;
    pub fn apexgpu_host_cache_bytes() -> i64;
    pub fn apexgpu_set_params(h: *mut ApexGpuSolver, poses: *const f64, intr: *const f64, points: *const f64) -> c_int;
    pub fn apexgpu_solve_augmented(h: *mut ApexGpuSolver, lambda: f64, variant: c_int, step_out: *mut f64, grad_out: *mut f64) -> c_int;
    pub fn apexgpu_column_norms(h: *mut ApexGpuSolver, norms_out: *mut f64) -> c_int;
    pub fn apexgpu_set_column_scaling(h: *mut ApexGpuSolver, scaling: *const f64) -> c_int;
    pub fn apexgpu_get_hessian_csc(h: *mut ApexGpuSolver, nnz_out: *mut i64, colptr: *mut i64, rowidx: *mut i64, values: *mut f64) -> c_int;
    // level 2 only (patches/0001-additive-hooks.md): trial point and cost on the device
    pub fn apexgpu_cost(h: *mut ApexGpuSolver, cost: *mut f64) -> c_int;
    pub fn apexgpu_eval_step(h: *mut ApexGpuSolver, trial_cost: *mut f64) -> c_int;
    pub fn apexgpu_commit_step(h: *mut ApexGpuSolver) -> c_int;
    pub fn apexgpu_discard_step(h: *mut ApexGpuSolver) -> c_int;
    pub fn apexgpu_get_params(h: *mut ApexGpuSolver, poses: *mut f64, intr: *mut f64, points: *mut f64) -> c_int;
    // multi-rank: RCCL, or N processes on one device over host shared memory (bring-up)
    pub fn apexgpu_get_unique_id(out128: *mut core::ffi::c_void) -> c_int;
    pub fn apexgpu_comm_init(h: *mut ApexGpuSolver, world: c_int, rank: c_int, unique_id128: *const core::ffi::c_void) -> c_int;
    pub fn apexgpu_comm_init_shm(h: *mut ApexGpuSolver, world: c_int, rank: c_int, name: *const c_char) -> c_int;
    // [0] dataflow sweeps that timed out and were repeated level by level inside the same solve, [1] dataflow sweeps still on
    pub fn apexgpu_counters(h: *mut ApexGpuSolver, out4: *mut i64) -> c_int;
    pub fn apexgpu_set_option(h: *mut ApexGpuSolver, name: *const c_char, value: c_int) -> c_int;
}

/// Status code -> `LinAlgError` (the codes ARE the variants, src/linalg/mod.rs:76-101).
pub(crate) fn check(h: *mut ApexGpuSolver, rc: c_int) -> LinAlgResult<()> {
    if rc == 0 {
        return Ok(());
    }
    let msg = unsafe { CStr::from_ptr(apexgpu_last_error(h)) }.to_string_lossy().into_owned();
    Err(match rc {
        -1 => LinAlgError::FactorizationFailed(msg),
        -2 => LinAlgError::SingularMatrix(msg),
        -3 => LinAlgError::SparseMatrixCreation(msg),
        -4 => LinAlgError::MatrixConversion(msg),
        -5 => LinAlgError::InvalidInput(msg),
        _ => LinAlgError::InvalidState(msg), // -6 InvalidState, -10 device / RCCL failure
    })
}

// ---------------------------------------------------------------------------------------------------------------------
// Device descriptors: what a factor / loss tells a device backend about itself (patches/0001-additive-hooks.md)
// ---------------------------------------------------------------------------------------------------------------------
/// Returned by `Factor::device_descriptor()`; `ProjectionFactor<BALPinholeCameraStrict, _>` with a single observation
/// is the only factor that answers `Some` (`observations` and `camera` are `pub`, projection_factor.rs:67-80).
#[derive(Clone, Copy, Debug)]
pub struct DeviceFactorDesc {
    pub uv: [f64; 2],
    /// `[f, k1, k2]` of the factor's camera: the intrinsics the device uses when they are not variables (BundleAdjustment)
    pub intrinsics: [f64; 3],
    /// `OP::POSE`, `OP::LANDMARK`, `OP::INTRINSIC` of the factor's `OptimizeParams` (src/factors/mod.rs:66-101): the key
    /// list holds exactly the optimised ones, in this order
    pub optimizes: [bool; 3],
}
impl DeviceFactorDesc {
    /// The `mode` of `apexgpu_create` (include/apexgpu.h, `APEXGPU_MODE_*`) for this `OptimizeParams` configuration
    pub fn device_mode(&self) -> Option<i32> {
        match self.optimizes {
            [true, true, false] => Some(APEXGPU_MODE_BUNDLE_ADJUSTMENT),
            [true, true, true] => Some(APEXGPU_MODE_SELF_CALIBRATION),
            [true, false, false] => Some(APEXGPU_MODE_ONLY_POSE),
            [false, true, false] => Some(APEXGPU_MODE_ONLY_LANDMARKS),
            [false, false, true] => Some(APEXGPU_MODE_ONLY_INTRINSICS),
            [true, false, true] => Some(APEXGPU_MODE_POSE_AND_INTRINSICS),
            [false, true, true] => Some(APEXGPU_MODE_LANDMARKS_AND_INTRINSICS),
            [false, false, false] => None,
        }
    }
}
/// Returned by `LossFunction::device_descriptor()`; `HuberLoss` answers `Some(Huber { scale })`.
#[derive(Clone, Copy, Debug, PartialEq)]
pub enum DeviceLossDesc {
    Huber { scale: f64 },
}

// ---------------------------------------------------------------------------------------------------------------------
// GpuContext: one device handle per optimize(), shared by the assembly backend and the linear solver
// ---------------------------------------------------------------------------------------------------------------------
pub struct GpuContext {
    pub(crate) h: *mut ApexGpuSolver,
    pub(crate) total_dof: usize,
    /// camera-side and landmark variable names in device numbering (camera i <-> (pose name, intrinsics name))
    pub(crate) cams: Vec<(String, Option<String>)>,
    pub(crate) pts: Vec<String>,
    /// intrinsics of cameras whose intrinsics are not variables (BundleAdjustment mode), from the factors' camera models
    pub(crate) fixed_intr: Vec<[f64; 3]>,
    /// serialises the handle: the C ABI is not re-entrant per handle
    pub(crate) lock: Mutex<()>,
}
// The handle is only ever used under `lock`; the library keeps no thread-local state.
unsafe impl Send for GpuContext {}
unsafe impl Sync for GpuContext {}

impl Drop for GpuContext {
    fn drop(&mut self) {
        // (the last context to go also returns the cached set-up blocks of the library to the system: apexgpu.h)
        unsafe { apexgpu_destroy(self.h) }
    }
}

/// `AssemblyBackend::assemble` is a static function: it finds the context of the problem it is handed through this
/// registry (key = address of the `Problem`, registered by `GpuSchurComplementSolver::bind_problem`, removed when the
/// solver is dropped).
static CONTEXTS: OnceLock<Mutex<HashMap<usize, Arc<GpuContext>>>> = OnceLock::new();

pub(crate) fn registry() -> &'static Mutex<HashMap<usize, Arc<GpuContext>>> {
    CONTEXTS.get_or_init(|| Mutex::new(HashMap::new()))
}

impl GpuContext {
    /// Gathers the current values in device numbering and uploads them (`VariableEnum::to_vector`: SE3 =
    /// `[tx,ty,tz,qw,qx,qy,qz]`, src/core/problem.rs:161-173).  THIS is what keeps the device at the optimiser's point:
    /// the LM loop owns the variables and retracts them on the host (level 1).
    pub(crate) fn upload(&self, variables: &HashMap<String, VariableEnum>) -> LinAlgResult<()> {
        let missing = |n: &str| LinAlgError::InvalidInput(format!("variable {n} is not in the optimiser's state"));
        let mut poses = Vec::with_capacity(7 * self.cams.len());
        let mut intr = Vec::with_capacity(3 * self.cams.len());
        for (i, (pose, intr_name)) in self.cams.iter().enumerate() {
            poses.extend(variables.get(pose).ok_or_else(|| missing(pose))?.to_vector().iter().copied());
            match intr_name {
                Some(n) => intr.extend(variables.get(n).ok_or_else(|| missing(n))?.to_vector().iter().copied()),
                None => intr.extend_from_slice(&self.fixed_intr[i]),
            }
        }
        let mut points = Vec::with_capacity(3 * self.pts.len());
        for p in &self.pts {
            points.extend(variables.get(p).ok_or_else(|| missing(p))?.to_vector().iter().copied());
        }
        let _g = self.lock.lock().map_err(|_| LinAlgError::InvalidState("device handle poisoned".into()))?;
        check(self.h, unsafe { apexgpu_set_params(self.h, poses.as_ptr(), intr.as_ptr(), points.as_ptr()) })
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// GpuBaMode: LinearizationMode + AssemblyBackend
// ---------------------------------------------------------------------------------------------------------------------
/// The Jacobian of this mode is a handle: "the Jacobian of the parameters last uploaded to this context", optionally with a
/// column scaling in force.  `Send + Sync` as `LinearizationMode` requires (cpu/mod.rs:30-32).
pub struct DeviceJacobian {
    pub(crate) ctx: Arc<GpuContext>,
    pub(crate) scaling: Option<Arc<Vec<f64>>>,
}
/// H = J^T J exported on demand (observers, DogLeg): a host CSC copy.
pub struct DeviceHessian {
    pub csc: SparseColMat<usize, f64>,
}

pub struct GpuBaMode;

impl LinearizationMode for GpuBaMode {
    type Jacobian = DeviceJacobian;
    type Hessian = DeviceHessian;
}

impl AssemblyBackend for GpuBaMode {
    /// Replaces `assemble_sparse` (src/linearizer/cpu/sparse.rs:119-184).  Nothing is linearised here: the factors are
    /// re-linearised inside every device kernel from the uploaded parameters.  The residual vector the trait returns is
    /// only ever forwarded to `LinearSolver::solve_*` (levenberg_marquardt.rs:880-882), which ignores it on this
    /// backend, so an empty column is returned instead of copying 2 n_obs doubles per iteration.
    fn assemble(
        problem: &Problem,
        variables: &HashMap<String, VariableEnum>,
        _variable_index_map: &HashMap<String, usize>,
        _symbolic_structure: Option<&SymbolicStructure>,
        _total_dof: usize,
    ) -> LinearizerResult<(Mat<f64>, Self::Jacobian)> {
        let key = problem as *const Problem as usize;
        let ctx = registry()
            .lock()
            .map_err(|_| LinearizerError::ParallelComputation("GPU context registry poisoned".to_string()))?
            .get(&key)
            .cloned()
            .ok_or_else(|| LinearizerError::FactorLinearization("problem is not bound to a GPU context (bind_problem)".to_string()))?;
        ctx.upload(variables).map_err(|e| LinearizerError::FactorLinearization(e.to_string()))?;
        Ok((Mat::zeros(0, 1), DeviceJacobian { ctx, scaling: None }))
    }

    /// `compute_column_norms` (src/linearizer/mod.rs:229-239) at the parameters last uploaded.
    fn compute_column_norms(jacobian: &Self::Jacobian) -> Vec<f64> {
        let ctx = &jacobian.ctx;
        let mut norms = vec![0.0; ctx.total_dof];
        if let Ok(_g) = ctx.lock.lock() {
            let _ = unsafe { apexgpu_column_norms(ctx.h, norms.as_mut_ptr()) };
        }
        norms
    }

    /// "J * diag(scaling)" is a state of the device solver (apexgpu_set_column_scaling): the next solve works in the
    /// scaled variables exactly as the reference's does with a scaled Jacobian (src/optimizer/mod.rs:749-763).
    fn apply_column_scaling(jacobian: &Self::Jacobian, scaling: &[f64]) -> Self::Jacobian {
        let ctx = jacobian.ctx.clone();
        if let Ok(_g) = ctx.lock.lock() {
            let _ = unsafe { apexgpu_set_column_scaling(ctx.h, scaling.as_ptr()) };
        }
        DeviceJacobian { ctx: jacobian.ctx.clone(), scaling: Some(Arc::new(scaling.to_vec())) }
    }

    /// step_i *= scaling_i (src/linearizer/mod.rs:253-262): host arithmetic, as in SparseMode.
    fn apply_inverse_scaling(step: &Mat<f64>, scaling: &[f64]) -> Mat<f64> {
        Mat::from_fn(step.nrows(), 1, |i, _| step[(i, 0)] * scaling[i])
    }

    fn hessian_vec_product(hessian: &Self::Hessian, vec: &Mat<f64>) -> Mat<f64> {
        &hessian.csc * vec
    }
}
