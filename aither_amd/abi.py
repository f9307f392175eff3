"""ctypes mirror of include/aither_gfx950.h (plumbing only).

`bind(lib, prefix)` attaches argument/return types to every entry point the
header declares, for either the product library (prefix "agx_") or the CPU
oracle used by the tests (prefix "ora_", same signatures).
"""
import ctypes as C

RECON = {"constant": 0, "muscl": 1, "weno": 2, "wenoZ": 3}
LIMITER = {"none": 0, "vanAlbada": 1, "minmod": 2}
FLUX = {"roe": 0, "ausm": 1}
TIME = {"explicitEuler": 0, "rk4": 1, "implicitEuler": 2,
        "crankNicholson": 3, "bdf2": 4}
SOLVER = {"lusgs": 0, "dplur": 1, "blusgs": 2, "bdplur": 3}
EQN = {"euler": 0, "navierStokes": 1, "rans": 2}
JACOBIAN = {"rusanov": 0, "approximateRoe": 1}
VISC_RECON = {"central": 0, "centralFourth": 1}
TURB = {"none": 0, "sst2003": 1, "kOmegaWilcox2006": 2, "sstdes": 3, "wale": 4}
BC = {"slipWall": 0, "viscousWall": 1, "characteristic": 2, "inlet": 3,
      "supersonicInflow": 4, "supersonicOutflow": 5, "stagnationInlet": 6,
      "pressureOutlet": 7, "interblock": 8, "periodic": 9}
FIELD = {"state": 0, "residual": 1, "dt": 2, "spec_radius": 3, "cons_n": 4,
         "update": 5, "diagonal": 6, "temperature": 7, "viscosity": 8,
         "cons_nm1": 9, "vel_grad": 10, "temp_grad": 11, "dens_grad": 12,
         "press_grad": 13}
HALO_STATE, HALO_UPDATE, HALO_VELGRAD_A, HALO_VELGRAD_B, HALO_TURB = 0, 1, 2, 3, 4
# variables of a function file (AGX_OUT_*), under the reference's names (output.cpp:235-407)
OUT = {"density": 0, "vel_x": 1, "vel_y": 2, "vel_z": 3, "pressure": 4, "mach": 5, "sos": 6,
       "dt": 7, "temperature": 8, "energy": 9, "enthalpy": 10, "cp": 11, "cv": 12, "rank": 13,
       "globalPosition": 14, "viscosityRatio": 15, "turbulentViscosity": 16, "viscosity": 17,
       "tke": 18, "sdr": 19, "f1": 20, "f2": 21, "wallDistance": 22}
for _n, _base in (("velGrad", 23),):
    for _q, _c in enumerate(("ux", "vx", "wx", "uy", "vy", "wy", "uz", "vz", "wz")):
        OUT[f"{_n}_{_c}"] = _base + _q
for _n, _base in (("tempGrad", 32), ("densityGrad", 35), ("pressGrad", 38), ("tkeGrad", 41),
                  ("omegaGrad", 44)):
    for _q, _c in enumerate("xyz"):
        OUT[f"{_n}_{_c}"] = _base + _q
for _q, _c in enumerate(("mass", "mom_x", "mom_y", "mom_z", "energy", "tke", "sdr")):
    OUT[f"resid_{_c}"] = 47 + _q

c_dp = C.POINTER(C.c_double)


class Gas(C.Structure):
    _fields_ = [(n, C.c_double) for n in (
        "gas_constant", "n", "heat_of_formation", "visc_c1", "visc_s",
        "cond_c1", "cond_s", "t_ref", "rho_ref", "l_ref", "a_ref")]


class Config(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "n_eq", "n_ghost", "recon", "limiter", "inviscid_flux", "is_viscous",
        "time_integration", "matrix_solver", "matrix_sweeps",
        "nonlinear_iterations", "equation_set", "inv_flux_jacobian",
        "viscous_recon", "turbulence_model")] + [(n, C.c_double) for n in (
            "kappa", "theta", "zeta", "matrix_relaxation", "dual_time_cfl",
            "dt_nondim", "viscous_cfl_coeff")] + [("gas", Gas)]


class BlockGeom(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "ni", "nj", "nk", "ng", "parent_block", "global_pos")] + [
        (n, c_dp) for n in ("farea_i", "farea_j", "farea_k", "vol", "center",
                            "width_i", "width_j", "width_k", "wall_dist")]


class BcState(C.Structure):
    _fields_ = [("pressure", C.c_double), ("density", C.c_double),
                ("velocity", C.c_double * 3),
                ("stagnation_pressure", C.c_double),
                ("stagnation_temperature", C.c_double),
                ("direction", C.c_double * 3),
                ("wall_temperature", C.c_double),
                ("wall_heat_flux", C.c_double),
                ("length_scale", C.c_double),
                ("is_isothermal", C.c_int32), ("is_heat_flux", C.c_int32),
                ("is_nonreflecting", C.c_int32), ("pad_", C.c_int32),
                ("turb_intensity", C.c_double), ("eddy_visc_ratio", C.c_double),
                ("von_karman", C.c_double), ("wall_constant", C.c_double),
                ("is_wall_law", C.c_int32), ("pad2_", C.c_int32)]


class BcSurface(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "bc_type", "imin", "imax", "jmin", "jmax", "kmin", "kmax", "tag")] + [
        ("state", BcState)]


class Connection(C.Structure):
    _fields_ = [("rank", C.c_int32 * 2), ("block", C.c_int32 * 2),
                ("local_block", C.c_int32 * 2), ("boundary", C.c_int32 * 2),
                ("d1_start", C.c_int32 * 2), ("d1_end", C.c_int32 * 2),
                ("d2_start", C.c_int32 * 2), ("d2_end", C.c_int32 * 2),
                ("const_surf", C.c_int32 * 2), ("patch_border", C.c_int32 * 8),
                ("orientation", C.c_int32), ("is_interblock", C.c_int32)]


class Slab(C.Structure):
    _fields_ = [("peer", C.c_int32), ("tag", C.c_int32), ("count", C.c_int64),
                ("send", c_dp), ("recv", c_dp)]


SWAP_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.POINTER(Slab), C.c_void_p)
ALLGATHER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p)


class Exchange(C.Structure):
    _fields_ = [("user", C.c_void_p), ("swap", SWAP_FN), ("allgather", ALLGATHER_FN),
                ("nranks", C.c_int32), ("host_buffers", C.c_int32)]


class Linf(C.Structure):
    _fields_ = [("linf", C.c_double), ("block", C.c_int32), ("i", C.c_int32),
                ("j", C.c_int32), ("k", C.c_int32), ("eqn", C.c_int32),
                ("pad_", C.c_int32)]


# every symbol include/aither_gfx950.h declares: name -> (restype, argtypes)
_vp = C.c_void_p
_i = C.c_int
SYMBOLS = {
    "last_error": (C.c_char_p, []),
    "version": (C.c_char_p, []),
    "ctx_create": (_i, [_i, _i, C.POINTER(_vp)]),
    "ctx_destroy": (None, [_vp]),
    "ctx_set_stream": (_i, [_vp, _vp]),
    "config_set": (_i, [_vp, C.POINTER(Config)]),
    "block_create": (_i, [_vp, C.POINTER(BlockGeom), C.POINTER(_i)]),
    "block_set_bcs": (_i, [_vp, _i, _i, C.POINTER(BcSurface)]),
    "conn_create": (_i, [_vp, C.POINTER(Connection), C.POINTER(_i)]),
    "setup_finalize": (_i, [_vp]),
    "state_upload": (_i, [_vp, _i, c_dp]),
    "field_download": (_i, [_vp, _i, _i, c_dp]),
    "field_upload": (_i, [_vp, _i, _i, c_dp]),
    "output_pack": (_i, [_vp, _i, _i, C.POINTER(C.c_int32), c_dp]),
    "restart_pack": (_i, [_vp, _i, _i, c_dp]),
    "plot3d_metrics": (_i, [_vp, _i, _i, _i] + [c_dp] * 9),
    "nearest_wall_distance": (_i, [_vp, C.c_int64, c_dp, C.c_int64, c_dp, c_dp]),
    "mg_restrict": (_i, [_vp, _vp, _i, _i, C.POINTER(C.c_int32), c_dp]),
    "mg_matrix_residual": (_i, [_vp, c_dp]),
    "mg_invert_diagonal": (_i, [_vp]),
    "mg_save_update": (_i, [_vp]),
    "mg_reset_diagonal": (_i, [_vp]),
    "mg_prolong": (_i, [_vp, _vp, _i, C.POINTER(C.c_int32), c_dp]),
    "store_time_n": (_i, [_vp, _i]),
    "iterate": (_i, [_vp, _i, C.c_double, c_dp, C.POINTER(Linf), c_dp]),
    "phase_bc_faces": (_i, [_vp]),
    "phase_bc_edges": (_i, [_vp]),
    "phase_residual": (_i, [_vp, _i, C.c_double]),
    "phase_explicit_update": (_i, [_vp, _i, c_dp, C.POINTER(Linf)]),
    "phase_implicit_begin": (_i, [_vp]),
    "phase_relax_forward": (_i, [_vp, _i]),
    "phase_relax_backward": (_i, [_vp, _i]),
    "phase_matrix_residual": (_i, [_vp, c_dp]),
    "phase_implicit_update": (_i, [_vp, _i, c_dp, C.POINTER(Linf)]),
    "halo_swap_local": (_i, [_vp, _i]),
    "halo_count": (C.c_int64, [_vp, _i, _i]),
    "halo_pack": (_i, [_vp, _i, _i, _vp]),
    "halo_unpack": (_i, [_vp, _i, _i, _vp]),
    "set_exchange": (_i, [_vp, C.POINTER(Exchange)]),
    "rccl_unique_id": (_i, [_vp]),
    "rccl_exchange_create": (_i, [_vp, _vp, _i, _i]),
    "halo_exchange": (_i, [_vp, _i]),
    "timing_enable": (_i, [_vp, _i]),
    "timing_get": (_i, [_vp, _i, c_dp, C.POINTER(C.c_int64)]),
    "timing_reset": (_i, [_vp]),
    "sync": (_i, [_vp]),
}


class Api:
    """Namespace of bound functions without the prefix."""

    def __init__(self, lib, prefix):
        self.lib, self.prefix = lib, prefix
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(lib, prefix + name)   # AttributeError if missing
            fn.restype = res
            fn.argtypes = args
            setattr(self, name, fn)

    def check(self, rc, what=""):
        if rc != 0:
            msg = self.last_error()
            raise RuntimeError(f"{self.prefix}{what} failed ({rc}): "
                               f"{msg.decode() if msg else ''}")
