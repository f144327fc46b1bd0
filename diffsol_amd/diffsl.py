"""DiffSL models on the GPU: `OdeBuilder::build_from_diffsl` of the reference (crates/diffsol/src/ode_solver/builder.rs,
crates/diffsol/src/ode_equations/diffsl.rs) for the HIP backend.

    m = DiffslModel(code)                # DiffSL text -> model source (host front end) -> hiprtc -> model id usable wherever a registry id is
    s = Solver(m, p, nbatch=..., ...)    # p: [nbatch][ninputs] in the order of `in = [...]`

n <= 8 with at most one stop condition compiles to the register-resident form (fused Newton kernels, device-resident integrators); anything else to
the run-time-sized form (one thread per component and system)."""
import ctypes as C

import numpy as np

from . import _ffi
from ._ffi import check, vp

TARGET_HIP_STATIC, TARGET_HIP_DYNAMIC, TARGET_HOST_C = 0, 1, 2
FORM_STATIC, FORM_DYNAMIC, FORM_STATIC_BANDED = 0, 1, 2
FAMILY_OPERATORS, FAMILY_FUSED, FAMILY_RESIDENT_BDF, FAMILY_RESIDENT_SDIRK = 0, 1, 2, 3


def generate(code, target, model_index=0):
    """DiffSL text -> (source code, dims dict, input defaults): dshs_diffsl_generate_indexed.  model_index: the value of the scalar `N` in the text (the reference's
    DiffSlContext::model_index, 0 by default), a compile-time constant of the generated model."""
    L = _ffi.load_host_lib()
    k = int(model_index)
    out = vp()
    dims = (C.c_int64 * 10)()
    check(L.dshs_diffsl_generate_indexed(code.encode(), target, k, C.byref(out), dims, None, 0), host=True)  # dimensions first: one default per declared input
    L.dshs_free_string(out)
    ndef = max(int(dims[1]), 1)
    defaults = (C.c_double * ndef)()
    check(L.dshs_diffsl_generate_indexed(code.encode(), target, k, C.byref(out), dims, defaults, ndef), host=True)
    try:
        src = C.string_at(out).decode()
    finally:
        L.dshs_free_string(out)
    d = dict(n=int(dims[0]), nparams=int(dims[1]), nroots=int(dims[2]), nout=int(dims[3]), has_mass=bool(dims[4]), no_inputs=bool(dims[5]),
             band=(int(dims[6]), int(dims[7]), int(dims[8]), int(dims[9])))
    return src, d, np.array(defaults[: d["nparams"]])


class DiffslModel:
    def __init__(self, code, form=None, lane_resident=True, model_index=0):
        self.code = code
        self.model_index = int(model_index)
        _, d, self.defaults = generate(code, TARGET_HOST_C, model_index)
        if form is None:
            # static form (compile-time n: the fused host-driven kernels, and the register-resident integrators up to n = 4) for small models; a model with 5 <= n <= 8
            # that should run per member on the device is compiled with form=FORM_DYNAMIC (the wavefront-per-member kernels take run-time-sized models)
            form = FORM_STATIC if d["n"] <= 8 and d["nroots"] <= 1 else FORM_DYNAMIC
        self.form = form
        self.source, d, _ = generate(code, TARGET_HIP_STATIC if form == FORM_STATIC else TARGET_HIP_DYNAMIC, model_index)
        self.n, self.nparams, self.nroots, self.nout, self.has_mass, self.no_inputs = (d[k] for k in ("n", "nparams", "nroots", "nout", "has_mass", "no_inputs"))
        self._L = _ffi.load_device_lib()
        mid = C.c_int()
        check(self._L.dsh_model_compile(self.source.encode(), form, self.n, self.nparams, self.nroots, self.nout, 1 if self.has_mass else 0, C.byref(mid)))
        self.model_id = mid.value
        self.lane_model_id = None
        jkl, jku = d["band"][0], d["band"][1]
        mass_k = max(d["band"][2], d["band"][3]) if self.has_mass else 0  # the lane-per-member banded kernels take a banded mass matrix (round 4; diagonal: the fast path)
        if lane_resident and form == FORM_DYNAMIC and self.n <= 64 and mass_k <= 4 and max(jkl, jku) <= 4 and self.nroots <= 8:
            # banded Jacobian: the same model once more in the lane-per-member form; per-member device-resident BDF solves run on it (compiled on first use)
            lane_src = generate(code, TARGET_HIP_STATIC, model_index)[0]
            lid = C.c_int()
            check(self._L.dsh_model_compile(lane_src.encode(), FORM_STATIC_BANDED, self.n, self.nparams, self.nroots, self.nout, 1 if self.has_mass else 0, C.byref(lid)))
            self.lane_model_id = lid.value
            check(self._L.dsh_model_set_twin(self.model_id, self.lane_model_id))
        self.band = d["band"]  # (jac_kl, jac_ku, mass_kl, mass_ku): structural bandwidths, declared so that banded models are assembled / factored on the band
        check(self._L.dsh_model_set_band(self.model_id, *self.band))
        if form == FORM_STATIC and self.n >= 5:
            # the register-resident integrators stop at n = 4: per-member device solves of this model run on its run-time-sized form (the wavefront-per-member kernels),
            # compiled by the library at the first such request
            dyn_src = generate(code, TARGET_HIP_DYNAMIC, model_index)[0]
            check(self._L.dsh_model_set_member_twin_source(self.model_id, dyn_src.encode(), self.n, self.nparams, self.nroots, self.nout))

    def precompile(self, family):
        """Compile a kernel family now instead of at its first launch (needs no GPU)."""
        check(self._L.dsh_model_precompile(self.model_id, family))

    def release(self):
        if getattr(self, "model_id", None) is not None:
            self._L.dsh_model_release(self.model_id)
            self.model_id = None
        if getattr(self, "lane_model_id", None) is not None:
            self._L.dsh_model_release(self.lane_model_id)
            self.lane_model_id = None

    def host_source(self):
        """The same model as an `extern "C"` CPU library source (dsl_rhs, dsl_jac_mul, ...): what the parity tests hand to the CPU oracle."""
        return generate(self.code, TARGET_HOST_C)[0]
