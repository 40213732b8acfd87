//! `GpuSchurComplementSolver`: the reference's `SparseSchurComplementSolver` surface on the MI355X backend.
//!
//! `StructureAware::initialize_structure` + `bind_problem` replace `build_block_structure`
//! (src/linalg/sparse/explicit_schur.rs:244-323) and `build_symbolic_structure` (src/linearizer/cpu/sparse.rs:54-105);
//! `solve_augmented_equation` replaces explicit_schur.rs:1129-1234 (H = J^T J, block extraction, damping, 3x3 inversion with
//! the eigenvalue gate, Schur complement, Cholesky + ladder / PCG, back-substitution) with one C-ABI call.
#![cfg(feature = "gpu")]

use std::collections::HashMap;
use std::os::raw::{c_char, c_int};
use std::sync::Arc;

use apex_manifolds::ManifoldType;
use faer::sparse::{SparseColMat, SymbolicSparseColMat};
use faer::Mat;

use crate::core::problem::{Problem, VariableEnum};
use crate::linalg::sparse::explicit_schur::{SchurOrdering, SchurVariant};
use crate::linalg::{LinAlgError, LinAlgResult, LinearSolver, StructureAware};
use crate::linearizer::gpu::*;

pub struct GpuSchurComplementSolver {
    device: c_int,
    variant: SchurVariant,
    implicit: bool,                 // IterativeSchurSolver semantics (matrix-free PCG): APEXGPU_VARIANT_IMPLICIT
    cg: (c_int, f64),               // SparseSchurComplementSolver::new: (200, 1e-6), explicit_schur.rs:211-212
    ordering: SchurOrdering,
    // captured by initialize_structure, in the order of their global columns
    cam_vars: Vec<(String, usize, ManifoldType)>,   // (name, first column, manifold) of every camera-side variable
    pt_vars: Vec<(String, usize)>,
    fixed: HashMap<String, Vec<usize>>,
    total_dof: usize,
    ctx: Option<Arc<GpuContext>>,
    problem_key: Option<usize>,
    gradient: Option<Mat<f64>>,
    hessian: Option<DeviceHessian>,
}

impl GpuSchurComplementSolver {
    pub fn new(device: i32) -> Self {
        Self {
            device: device as c_int, variant: SchurVariant::Sparse, implicit: false, cg: (200, 1e-6),
            ordering: SchurOrdering::default(), cam_vars: Vec::new(), pt_vars: Vec::new(), fixed: HashMap::new(),
            total_dof: 0, ctx: None, problem_key: None, gradient: None, hessian: None,
        }
    }
    pub fn with_variant(mut self, v: SchurVariant) -> Self { self.variant = v; self }
    pub fn with_implicit_operator(mut self, on: bool) -> Self { self.implicit = on; if on { self.cg = (500, 1e-9); } self }
    pub fn with_cg_params(mut self, max_iterations: usize, tolerance: f64) -> Self { self.cg = (max_iterations as c_int, tolerance); self }

    fn variant_code(&self) -> c_int {
        if self.implicit { 2 } else { match self.variant { SchurVariant::Sparse => 0, SchurVariant::Iterative => 1 } }
    }

    /// Hands the factor list to the device: one BAL projection factor per residual block, in residual-block order
    /// (`residual_row_start_idx`, src/core/problem.rs:584-595), the observed pixel and the camera model from
    /// `Factor::device_descriptor()`, the Huber scale from `LossFunction::device_descriptor()`.  Any factor or loss that
    /// answers `None` makes the whole problem ineligible: the caller falls back to `SparseSchurComplement`.
    pub fn bind_problem(&mut self, problem: &Problem) -> LinAlgResult<()> {
        let bad = |m: String| LinAlgError::InvalidInput(m);
        if self.cam_vars.is_empty() || self.pt_vars.is_empty() {
            return Err(LinAlgError::InvalidState("Block structure not built. Call initialize_structure() first.".into()));
        }
        // camera-side variables: SE3 = pose, Rn(3) = intrinsics; paired into cameras by the factors' key lists
        let pose_index: HashMap<&str, usize> =
            self.cam_vars.iter().filter(|v| v.2 == ManifoldType::SE3).enumerate().map(|(i, v)| (v.0.as_str(), i)).collect();
        let n_cam = pose_index.len();
        let pt_index: HashMap<&str, usize> = self.pt_vars.iter().enumerate().map(|(i, v)| (v.0.as_str(), i)).collect();
        let col_of: HashMap<&str, usize> = self.cam_vars.iter().map(|v| (v.0.as_str(), v.1)).collect();
        let mut pose_name = vec![String::new(); n_cam];
        for (name, &i) in &pose_index { pose_name[i] = (*name).to_string(); }
        let mut intr_name: Vec<Option<String>> = vec![None; n_cam];
        let mut fixed_intr = vec![[0.0f64; 3]; n_cam];

        let mut blocks: Vec<_> = problem.residual_blocks().values().collect();
        blocks.sort_by_key(|b| b.residual_row_start_idx);
        let n_obs = blocks.len();
        let (mut cam_idx, mut pt_idx, mut uv) = (Vec::with_capacity(n_obs), Vec::with_capacity(n_obs), Vec::with_capacity(2 * n_obs));
        let mut mode: Option<i32> = None;
        let mut huber: Option<f64> = None; // Some(-1.0): no loss function on any block
        for b in &blocks {
            let d = b.factor.device_descriptor().ok_or_else(|| bad("a factor has no device descriptor".into()))?;
            let m = d.device_mode().ok_or_else(|| bad("a projection factor that optimises nothing".into()))?;
            if *mode.get_or_insert(m) != m { return Err(bad("mixed optimization types".into())); }
            // key lists as bin/bundle_adjustment.rs:391-441 builds them for EVERY OptimizeParams configuration:
            // [pose, pt] or [pose, pt, intr]; which of them the factor optimises is the device mode (Jacobian column masks)
            let keyed_intr = b.variable_key_list.len() == 3;
            if b.variable_key_list.len() != 2 && !keyed_intr { return Err(bad("unexpected key list of a projection factor".into())); }
            let ci = *pose_index.get(b.variable_key_list[0].as_str()).ok_or_else(|| bad(format!("{} is not a pose", b.variable_key_list[0])))?;
            let pi = *pt_index.get(b.variable_key_list[1].as_str()).ok_or_else(|| bad(format!("{} is not a landmark", b.variable_key_list[1])))?;
            if keyed_intr {
                let n = &b.variable_key_list[2];
                match &intr_name[ci] {
                    None => intr_name[ci] = Some(n.clone()),
                    Some(have) if have == n => {}
                    Some(_) => return Err(bad(format!("camera {} is keyed on two intrinsics variables", pose_name[ci]))),
                }
            } else {
                fixed_intr[ci] = d.intrinsics;
            }
            let scale = match &b.loss_func {
                None => -1.0,
                Some(l) => match l.device_descriptor() {
                    Some(DeviceLossDesc::Huber { scale }) => scale,
                    None => return Err(bad("a loss function has no device descriptor".into())),
                },
            };
            if *huber.get_or_insert(scale) != scale { return Err(bad("mixed loss functions".into())); }
            cam_idx.push(ci as u32); pt_idx.push(pi as u32); uv.extend_from_slice(&d.uv);
        }
        let mode = mode.unwrap_or(APEXGPU_MODE_SELF_CALIBRATION);
        // columns and fixed masks (Variable::fixed_indices: zeroed in the step at apply time, src/core/problem.rs:185-197)
        let mut pose_col = vec![0i64; n_cam]; let mut intr_col = vec![0i64; n_cam];
        let mut fix_pose = vec![0u8; 6 * n_cam]; let mut fix_intr = vec![0u8; 3 * n_cam]; let mut fix_pt = vec![0u8; 3 * self.pt_vars.len()];
        // intrinsics variables that no factor is keyed on (BundleAdjustment mode keeps them in the state): they still own
        // columns; the i-th Rn(3) camera-side variable belongs to the i-th pose in name order (intr_0007 <-> pose_0007)
        let spare: Vec<&(String, usize, ManifoldType)> = self.cam_vars.iter().filter(|v| v.2 == ManifoldType::RN).collect();
        for i in 0..n_cam {
            pose_col[i] = col_of[pose_name[i].as_str()] as i64;
            let iname = intr_name[i].clone().or_else(|| spare.get(i).map(|v| v.0.clone()));
            intr_col[i] = match &iname { Some(n) => col_of[n.as_str()] as i64, None => return Err(bad("camera without intrinsics columns".into())) };
            for &k in self.fixed.get(&pose_name[i]).map(|v| v.as_slice()).unwrap_or(&[]) { if k < 6 { fix_pose[6 * i + k] = 1; } }
            if let Some(n) = &iname { for &k in self.fixed.get(n).map(|v| v.as_slice()).unwrap_or(&[]) { if k < 3 { fix_intr[3 * i + k] = 1; } } }
        }
        let pt_col: Vec<i64> = self.pt_vars.iter().map(|v| v.1 as i64).collect();
        for (j, v) in self.pt_vars.iter().enumerate() {
            for &k in self.fixed.get(&v.0).map(|x| x.as_slice()).unwrap_or(&[]) { if k < 3 { fix_pt[3 * j + k] = 1; } }
        }
        let mut h: *mut ApexGpuSolver = std::ptr::null_mut();
        let rc = unsafe { apexgpu_create(n_cam as i64, self.pt_vars.len() as i64, n_obs as i64, mode, self.device, &mut h) };
        if rc != 0 { return Err(LinAlgError::InvalidState(format!("apexgpu_create failed ({rc}): no MI355X device {}?", self.device))); }
        let ctx = Arc::new(GpuContext {
            h, total_dof: self.total_dof,
            cams: (0..n_cam).map(|i| (pose_name[i].clone(), intr_name[i].clone())).collect(),
            pts: self.pt_vars.iter().map(|v| v.0.clone()).collect(),
            fixed_intr, lock: std::sync::Mutex::new(()),
        });
        if self.implicit {
            // IterativeSchurSolver never forms S (implicit_schur.rs:163-251); neither does a handle made for it: only the diagonal
            // tiles exist and no pair list is built, so set-up and iteration do not depend on the fill of S (round 4)
            check(h, unsafe { apexgpu_set_option(h, b"matrix_free_only\0".as_ptr() as *const c_char, 1) })?;
        }
        // This binding is level 1: step statistics, retraction and trial cost stay with the reference's
        // LM loop on the host, so the solve need not enqueue them (the backend does by default for its level-2 callers).
        check(h, unsafe { apexgpu_set_option(h, b"eager_step_eval\0".as_ptr() as *const c_char, 0) })?;
        check(h, unsafe { apexgpu_set_structure(h, cam_idx.as_ptr(), pt_idx.as_ptr(), uv.as_ptr(), intr_col.as_ptr(), pose_col.as_ptr(),
                                                 pt_col.as_ptr(), fix_pose.as_ptr(), fix_intr.as_ptr(), fix_pt.as_ptr(), huber.unwrap_or(-1.0)) })?;
        check(h, unsafe { apexgpu_set_cg_params(h, self.cg.0, self.cg.1) })?;
        {
            // A structure whose direct factorisation the backend refuses (S dense at tile granularity) does not fail: the handle
            // answers with the matrix-free PCG (IterativeSchurSolver semantics, implicit_schur.rs:835-946).  Say so once.
            let mut used: c_int = 0;
            let mut why = [0 as c_char; 512];
            let asked: c_int = if self.implicit { 2 } else if matches!(self.variant, SchurVariant::Iterative) { 1 } else { 0 };
            check(h, unsafe { apexgpu_variant_info(h, asked, &mut used, why.as_mut_ptr(), why.len() as c_int) })?;
            if used != asked {
                let msg = unsafe { std::ffi::CStr::from_ptr(why.as_ptr()) }.to_string_lossy().into_owned();
                // (round 6: also when the direct path was merely predicted to cost more than the matrix-free PCG at its cap)
                let mut c = [0f64; 4];
                check(h, unsafe { apexgpu_variant_costs(h, c.as_mut_ptr()) })?;
                tracing::warn!("GpuSchurComplementSolver: variant {asked} requested, variant {used} runs: {msg} \
                                (predicted ms per solve: direct {:.1}, matrix-free {:.1})", c[0], c[1]);
            }
        }
        let key = problem as *const Problem as usize;
        registry().lock().map_err(|_| LinAlgError::InvalidState("GPU context registry poisoned".into()))?.insert(key, ctx.clone());
        self.ctx = Some(ctx);
        self.problem_key = Some(key);
        Ok(())
    }

    fn context(&self) -> LinAlgResult<&Arc<GpuContext>> {
        self.ctx.as_ref().ok_or_else(|| LinAlgError::InvalidInput("Block structure not built. Call initialize_structure() first.".to_string()))
    }
}

impl StructureAware for GpuSchurComplementSolver {
    /// Same classification as `SchurOrdering::should_eliminate` (explicit_schur.rs:111-133): names starting with "pt_"
    /// (Rn, 3 DOF) are landmarks, everything else is camera-side; same errors for an empty side (:300-311).
    fn initialize_structure(&mut self, variables: &HashMap<String, VariableEnum>, variable_index_map: &HashMap<String, usize>) -> LinAlgResult<()> {
        let mut names: Vec<&String> = variables.keys().collect();
        names.sort(); // the global column order (src/optimizer/mod.rs:530-536)
        self.cam_vars.clear(); self.pt_vars.clear(); self.fixed.clear();
        self.total_dof = 0;
        for name in names {
            let var = &variables[name];
            let col = *variable_index_map.get(name).ok_or_else(|| LinAlgError::InvalidInput(format!("Variable {name} not found in index map")))?;
            let (mt, size) = (var.manifold_type(), var.get_size());
            self.total_dof = self.total_dof.max(col + size);
            let fixed: Vec<usize> = match var {
                VariableEnum::SE3(v) => v.fixed_indices.iter().copied().collect(),
                VariableEnum::Rn(v) => v.fixed_indices.iter().copied().collect(),
                _ => return Err(LinAlgError::InvalidInput(format!("variable {name}: the GPU backend handles SE3 poses and Rn blocks only"))),
            };
            if !fixed.is_empty() { self.fixed.insert(name.clone(), fixed); }
            if self.ordering.should_eliminate(name, &mt, size) { self.pt_vars.push((name.clone(), col)); } else { self.cam_vars.push((name.clone(), col, mt)); }
        }
        if self.cam_vars.is_empty() { return Err(LinAlgError::InvalidInput("No camera variables found".to_string())); }
        if self.pt_vars.is_empty() { return Err(LinAlgError::InvalidInput("No landmark variables found".to_string())); }
        Ok(())
    }
}

impl LinearSolver<GpuBaMode> for GpuSchurComplementSolver {
    fn solve_normal_equation(&mut self, residuals: &Mat<f64>, jacobian: &DeviceJacobian) -> LinAlgResult<Mat<f64>> {
        self.solve_augmented_equation(residuals, jacobian, 0.0)
    }

    /// (J^T J + lambda I) dx = -J^T r at the parameters `GpuBaMode::assemble` uploaded for `jacobian`; the returned step
    /// is the full step in global column order, length total_dof, as the trait demands (src/linalg/mod.rs:150-160).
    fn solve_augmented_equation(&mut self, _residuals: &Mat<f64>, jacobian: &DeviceJacobian, lambda: f64) -> LinAlgResult<Mat<f64>> {
        let ctx = self.context()?.clone();
        if !Arc::ptr_eq(&ctx, &jacobian.ctx) { return Err(LinAlgError::InvalidInput("Jacobian handle of another problem".to_string())); }
        let mut step = Mat::<f64>::zeros(self.total_dof, 1);
        let mut grad = Mat::<f64>::zeros(self.total_dof, 1);
        {
            let _g = ctx.lock.lock().map_err(|_| LinAlgError::InvalidState("device handle poisoned".into()))?;
            // an n x 1 faer::Mat is one contiguous column
            check(ctx.h, unsafe { apexgpu_solve_augmented(ctx.h, lambda, self.variant_code(), step.as_ptr_mut(), grad.as_ptr_mut()) })?;
        }
        self.gradient = Some(grad); // +J^T r: LM errors without it (levenberg_marquardt.rs:743-745)
        self.hessian = None;        // exported lazily, see get_hessian
        Ok(step)
    }

    /// `Some(H)` after a solve like SparseSchurComplementSolver (explicit_schur.rs:1236-1238).  The device never forms H;
    /// observers that ask for it pay for the export (apexgpu_get_hessian_csc) once per iteration.
    fn get_hessian(&self) -> Option<&DeviceHessian> {
        // `&self`: the cache is filled by `refresh_hessian`, which the observer hook of the LM arm calls when observers
        // are registered (patches/0001-additive-hooks.md) -- a `&self` getter cannot run the export itself.
        self.hessian.as_ref()
    }

    fn get_gradient(&self) -> Option<&Mat<f64>> { self.gradient.as_ref() }
}

impl GpuSchurComplementSolver {
    /// Fills the cache behind `get_hessian` (called by the LM arm when `!self.observers.is_empty()`).
    pub fn refresh_hessian(&mut self) -> LinAlgResult<()> {
        let ctx = self.context()?.clone();
        let _g = ctx.lock.lock().map_err(|_| LinAlgError::InvalidState("device handle poisoned".into()))?;
        let mut nnz: i64 = 0;
        check(ctx.h, unsafe { apexgpu_get_hessian_csc(ctx.h, &mut nnz, std::ptr::null_mut(), std::ptr::null_mut(), std::ptr::null_mut()) })?;
        let n = self.total_dof;
        let (mut colptr, mut rowidx, mut values) = (vec![0i64; n + 1], vec![0i64; nnz as usize], vec![0f64; nnz as usize]);
        check(ctx.h, unsafe { apexgpu_get_hessian_csc(ctx.h, &mut nnz, colptr.as_mut_ptr(), rowidx.as_mut_ptr(), values.as_mut_ptr()) })?;
        let sym = SymbolicSparseColMat::<usize>::new_checked(n, n, colptr.iter().map(|&x| x as usize).collect(), None,
                                                             rowidx.iter().map(|&x| x as usize).collect());
        self.hessian = Some(DeviceHessian { csc: SparseColMat::new(sym, values) });
        Ok(())
    }
}

impl Drop for GpuSchurComplementSolver {
    fn drop(&mut self) {
        if let Some(key) = self.problem_key.take() {
            if let Ok(mut r) = registry().lock() { r.remove(&key); }
        }
    }
}
