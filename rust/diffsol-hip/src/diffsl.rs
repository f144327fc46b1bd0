//! DiffSL text -> a device model (what `OdeBuilder::build_from_diffsl` does with the external `diffsl` JIT, crates/diffsol/src/ode_equations/diffsl.rs):
//! the front end of libdiffsol_hip_host.so (`dshs_diffsl_generate`) turns the text into HIP source, `dsh_model_compile` instantiates the library's own
//! kernel templates for it with hiprtc; the returned id plugs into `HipModelEquations::from_id`.
use crate::context::HipContext;
use crate::equations::HipModelEquations;
use crate::error::last_error;
use crate::ffi;
use diffsol_la::error::LaError;
use std::ffi::{CStr, CString};
use std::os::raw::{c_char, c_int};
use std::ptr;

// include/diffsol_hip_solver.h (host-side library): the DiffSL front end
const DSHS_DIFFSL_HIP_STATIC: c_int = 0;
const DSHS_DIFFSL_HIP_DYNAMIC: c_int = 1;
#[link(name = "diffsol_hip_host")]
extern "C" {
    fn dshs_last_error() -> *const c_char;
    fn dshs_diffsl_generate(code: *const c_char, target: c_int, source_out: *mut *mut c_char, dims: *mut i64, defaults_out: *mut f64, defaults_cap: i64) -> c_int;
    fn dshs_free_string(s: *mut c_char);
}

/// Dimensions and input defaults of a compiled model.
#[derive(Debug, Clone)]
pub struct DiffslInfo {
    pub model_id: i32,
    pub nstates: usize,
    pub nparams: usize,
    pub nroots: usize,
    pub nout: usize,
    pub has_mass: bool,
    pub defaults: Vec<f64>,
}

/// Compile DiffSL text for the device.  Models with n <= 8 states and <= 1 stop condition get the register-resident form (fused step kernels,
/// device-resident integrators), larger ones the run-time-sized form; the structural bandwidth of the Jacobian / mass matrix found by the front end's
/// dependency analysis is declared to the library, so banded models are assembled and factored on the band only.
pub fn compile(code: &str) -> Result<DiffslInfo, LaError> {
    let c = CString::new(code).map_err(|_| LaError::Other("DiffSL text contains a NUL byte".into()))?;
    let mut dims = [0i64; 10];
    let mut src: *mut c_char = ptr::null_mut();
    let host_err = || LaError::Other(unsafe { CStr::from_ptr(dshs_last_error()) }.to_string_lossy().into_owned());
    // dimensions first, then the form that fits them
    if unsafe { dshs_diffsl_generate(c.as_ptr(), DSHS_DIFFSL_HIP_DYNAMIC, &mut src, dims.as_mut_ptr(), ptr::null_mut(), 0) } != 0 {
        return Err(host_err());
    }
    unsafe { dshs_free_string(src) };
    let is_static = dims[0] <= 8 && dims[2] <= 1;
    let mut defaults = vec![0.0f64; (dims[1] as usize).max(1)];
    let target = if is_static { DSHS_DIFFSL_HIP_STATIC } else { DSHS_DIFFSL_HIP_DYNAMIC };
    if unsafe { dshs_diffsl_generate(c.as_ptr(), target, &mut src, dims.as_mut_ptr(), defaults.as_mut_ptr(), defaults.len() as i64) } != 0 {
        return Err(host_err());
    }
    let mut id: c_int = -1;
    let form = if is_static { ffi::DSH_JIT_FORM_STATIC } else { ffi::DSH_JIT_FORM_DYNAMIC };
    let rc = unsafe { ffi::dsh_model_compile(src, form, dims[0], dims[1], dims[2], dims[3], dims[4] as c_int, &mut id) };
    unsafe { dshs_free_string(src) };
    if rc != 0 {
        return Err(LaError::Other(last_error()));
    }
    unsafe { ffi::dsh_model_set_band(id, dims[6] as c_int, dims[7] as c_int, dims[8] as c_int, dims[9] as c_int) };
    if is_static && dims[0] >= 5 {
        // the register-resident integrators stop at n = 4: per-member device solves of this model run on its run-time-sized form, which the library compiles
        // at the first such request (dsh_model_member_twin)
        let mut dyn_src: *mut c_char = ptr::null_mut();
        let mut d2 = [0i64; 10];
        if unsafe { dshs_diffsl_generate(c.as_ptr(), DSHS_DIFFSL_HIP_DYNAMIC, &mut dyn_src, d2.as_mut_ptr(), ptr::null_mut(), 0) } == 0 {
            unsafe { ffi::dsh_model_set_member_twin_source(id, dyn_src, dims[0], dims[1], dims[2], dims[3]) };
            unsafe { dshs_free_string(dyn_src) };
        }
    }
    defaults.truncate(dims[1] as usize);
    Ok(DiffslInfo { model_id: id, nstates: dims[0] as usize, nparams: dims[1] as usize, nroots: dims[2] as usize, nout: dims[3] as usize, has_mass: dims[4] != 0, defaults })
}

/// `OdeBuilder::build_from_diffsl` for an ensemble: compile, then the equations with `params` (batch-major; empty = the model's defaults for every member).
pub fn equations_from_diffsl(code: &str, params: Vec<f64>, ctx: HipContext) -> Result<HipModelEquations, LaError> {
    use diffsol_la::Context;
    let info = compile(code)?;
    let p = if params.is_empty() { info.defaults.iter().cycle().take(info.nparams * ctx.nbatch()).cloned().collect() } else { params };
    Ok(HipModelEquations::from_id(info.model_id, 0, p, ctx, true))
}
