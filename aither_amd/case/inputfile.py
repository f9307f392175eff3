"""Parser for AITHER `.inp` files (setup-time, host side).

Mirrors the keyword set and defaults of the reference's input class
(src/input.cpp:49-157 defaults, :167-643 ReadInput) for the options on the hot
path.  Options that select out-of-scope physics (multi-species, chemistry,
wall laws, multigrid, RANS) are parsed and rejected loudly by `validate`.
"""
from dataclasses import dataclass, field
import re
from typing import Dict, List, Optional

MUSCL_KAPPA = {"upwind": -1.0, "fromm": 0.0, "quick": 0.5, "central": 1.0,
               "thirdOrder": 1.0 / 3.0}           # input.cpp:277-292


@dataclass
class Surface:
    """boundarySurface (include/boundaryConditions.hpp:55-150)."""
    bc_type: str
    imin: int
    imax: int
    jmin: int
    jmax: int
    kmin: int
    kmax: int
    tag: int

    def surface_type(self):
        # boundaryConditions.cpp:2424-2456
        if self.imin == self.imax:
            return 1 if self.imax == 0 else 2
        if self.jmin == self.jmax:
            return 3 if self.jmax == 0 else 4
        if self.kmin == self.kmax:
            return 5 if self.kmax == 0 else 6
        raise ValueError(f"surface {self} is not an i, j or k surface")

    def sort_key(self):
        # boundarySurface::operator< (boundaryConditions.cpp:92-105)
        return (self.surface_type(), self.imin, self.imax, self.jmin,
                self.jmax, self.kmin, self.kmax, self.tag)

    def is_connection(self):
        return self.bc_type in ("interblock", "periodic")

    def partner_surface(self):
        # boundaryConditions.cpp:2471-2498
        assert self.bc_type == "interblock"
        return self.tag // 1000

    def partner_block(self):
        return self.tag - 1000 * self.partner_surface()

    # direction helpers: dir3 normal, dir1/dir2 cyclic (cpp:2531-2577)
    def dir3(self):
        return "ijk"[(self.surface_type() - 1) // 2]

    def dir1(self):
        return "jki"[(self.surface_type() - 1) // 2]

    def dir2(self):
        return "kij"[(self.surface_type() - 1) // 2]

    def rng(self, d):
        lo, hi = {"i": (self.imin, self.imax), "j": (self.jmin, self.jmax),
                  "k": (self.kmin, self.kmax)}[d]
        return (lo, lo + 1) if lo == hi else (lo, hi)   # cpp:2657-2667


@dataclass
class State:
    """One entry of initialConditions / boundaryStates."""
    kind: str
    params: Dict[str, object] = field(default_factory=dict)

    def get(self, key, default=None):
        return self.params.get(key, default)


@dataclass
class InputDeck:
    grid_name: str = ""
    dt: float = -1.0
    iterations: int = 1
    rho_ref: float = -1.0
    t_ref: float = -1.0
    l_ref: float = 1.0
    fluids: List[str] = field(default_factory=lambda: ["air"])
    time_integration: str = "explicitEuler"
    face_reconstruction: str = "constant"
    viscous_face_reconstruction: str = "central"
    kappa: float = -2.0
    limiter: str = "none"
    output_frequency: int = 1
    equation_set: str = "euler"
    matrix_solver: str = "lusgs"
    matrix_sweeps: int = 1
    matrix_relaxation: float = 1.0
    theta: float = 1.0
    zeta: float = 0.0
    nonlinear_iterations: int = 1
    cfl_max: float = 1.0
    cfl_step: float = 0.0
    cfl_start: float = 1.0
    inv_flux_jac: str = "rusanov"
    dual_time_cfl: float = -1.0
    inviscid_flux: str = "roe"
    decomposition: str = "cubic"
    turbulence_model: str = "none"
    thermodynamic_model: str = "caloricallyPerfect"
    multigrid_levels: int = 1
    multigrid_cycle: str = "V"
    ics: List[State] = field(default_factory=list)
    bc_states: List[State] = field(default_factory=list)
    bcs: List[List[Surface]] = field(default_factory=list)
    num_surf: List[tuple] = field(default_factory=list)

    # ---- derived queries, same names/semantics as input.hpp -------------
    def is_implicit(self):
        return self.time_integration in ("implicitEuler", "crankNicholson",
                                         "bdf2")

    def is_viscous(self):
        return self.equation_set in ("navierStokes", "rans")

    def is_rans(self):
        return self.equation_set == "rans"

    def is_multilevel_in_time(self):
        return self.time_integration == "bdf2"

    def need_to_store_time_n(self):
        return self.is_implicit() or self.time_integration == "rk4"

    def using_muscl(self):
        return self.face_reconstruction in MUSCL_KAPPA

    def num_ghost_layers(self):
        # input.cpp:1127-1143
        if self.face_reconstruction == "constant":
            layers = 1
        elif self.using_muscl():
            layers = 2
        elif self.face_reconstruction in ("weno", "wenoZ"):
            layers = 3
        else:
            raise ValueError("unsupported faceReconstruction")
        visc = 2 if self.viscous_face_reconstruction == "centralFourth" else 1
        return max(layers, visc)

    def cfl(self, nn):
        # input::CalcCFL (input.cpp:637-639)
        return min(self.cfl_start + nn * self.cfl_step, self.cfl_max)

    def viscous_cfl_coefficient(self):
        # input.cpp:1110-1118
        if self.kappa == 1.0:
            return 4.0
        if self.kappa == -2.0:
            return 2.0
        return 1.0

    def matrix_requires_initialization(self):
        return self.matrix_solver in ("dplur", "bdplur") or \
            self.matrix_sweeps > 1

    def ic_for_block(self, block):
        # input::ICStateForBlock (input.cpp:1146-1172)
        default = None
        for ic in self.ics:
            tag = ic.get("tag")
            if tag == block:
                return ic
            if tag == -1 and default is None:
                default = ic
        if default is None:
            raise ValueError(f"no initial condition for block {block}")
        return default

    def bc_data(self, tag):
        # input::BCData (input.cpp:1175-1187)
        for st in self.bc_states:
            # periodic states carry startTag (= Tag()) and endTag
            if st.get("tag", st.get("startTag")) == tag or \
                    st.get("endTag") == tag:
                return st
        raise KeyError(f"no boundaryStates entry for tag {tag}")

    def validate(self):
        bad = []
        if len(self.fluids) != 1:
            bad.append("multi-species")
        if self.equation_set not in ("euler", "navierStokes", "rans"):
            bad.append(f"equationSet {self.equation_set}")
        if self.equation_set == "rans" and self.turbulence_model not in ("sst2003", "sstdes",
                                                                         "kOmegaWilcox2006"):
            bad.append(f"turbulenceModel {self.turbulence_model}")
        if self.thermodynamic_model != "caloricallyPerfect":
            bad.append("thermallyPerfect")
        if self.multigrid_levels != 1:
            bad.append("multigrid")
        if self.matrix_solver not in ("lusgs", "dplur", "blusgs", "bdplur"):
            bad.append(f"matrixSolver {self.matrix_solver}")
        if self.matrix_solver in ("blusgs", "bdplur") and self.inv_flux_jac != "rusanov":
            bad.append("block-matrix solver with inviscidFluxJacobian approximateRoe")
        if self.inv_flux_jac not in ("rusanov", "approximateRoe"):
            bad.append(f"inviscidFluxJacobian {self.inv_flux_jac}")
        if self.viscous_face_reconstruction not in ("central", "centralFourth"):
            bad.append(f"viscousFaceReconstruction {self.viscous_face_reconstruction}")
        if bad:
            raise NotImplementedError(
                "outside the accelerated hot path: " + ", ".join(bad))


def _strip(line):
    line = line.strip()
    pos = line.find("#")
    return line[:pos].strip() if pos >= 0 else line


def _parse_value(text):
    text = text.strip().rstrip(",").strip()
    if text.startswith("["):
        inner = text[1:text.index("]")]
        items = [t.strip() for t in inner.split(",") if t.strip()]
        if items and "=" in items[0]:
            return {k.strip(): float(v) for k, v in
                    (it.split("=") for it in items)}
        return [float(t) for t in items]
    if text in ("true", "false", "yes", "no"):
        return text in ("true", "yes")
    try:
        return int(text)
    except ValueError:
        pass
    try:
        return float(text)
    except ValueError:
        return text


def _parse_state_list(text):
    out = []
    for m in re.finditer(r"(\w+)\s*\(([^)]*)\)", text):
        kind, body = m.group(1), m.group(2)
        params = {}
        for tok in body.split(";"):
            tok = tok.strip()
            if not tok:
                continue
            key, val = tok.split("=", 1)
            params[key.strip()] = _parse_value(val)
        out.append(State(kind, params))
    return out


def parse_input(path):
    deck = InputDeck()
    with open(path) as fh:
        raw = [_strip(l) for l in fh]
    lines = [l for l in raw if l]
    it = iter(range(len(lines)))
    idx = 0

    def read_list(first):
        nonlocal idx
        text = first
        while ">" not in text:
            idx += 1
            text += " " + lines[idx]
        return text[text.index("<") + 1:text.index(">")]

    while idx < len(lines):
        line = lines[idx]
        if ":" not in line:
            idx += 1
            continue
        key, val = [t.strip() for t in line.split(":", 1)]
        if key == "gridName":
            deck.grid_name = val
        elif key == "timeStep":
            deck.dt = float(val)
        elif key == "iterations":
            deck.iterations = int(val)
        elif key == "referenceDensity":
            deck.rho_ref = float(val)
        elif key == "referenceTemperature":
            deck.t_ref = float(val)
        elif key == "referenceLength":
            deck.l_ref = float(val)
        elif key == "fluids":
            states = _parse_state_list(read_list(val))
            deck.fluids = [s.get("name") for s in states]
        elif key == "timeIntegration":
            deck.time_integration = val
            if val == "implicitEuler":
                deck.theta, deck.zeta = 1.0, 0.0
            elif val == "crankNicholson":
                deck.theta, deck.zeta = 0.5, 0.0
            elif val == "bdf2":
                deck.theta, deck.zeta = 1.0, 0.5
        elif key == "faceReconstruction":
            deck.face_reconstruction = val
            if val in MUSCL_KAPPA:
                deck.kappa = MUSCL_KAPPA[val]
            elif val not in ("constant", "weno", "wenoZ"):
                raise ValueError(f"faceReconstruction {val} not recognized")
        elif key == "viscousFaceReconstruction":
            deck.viscous_face_reconstruction = val
        elif key == "limiter":
            deck.limiter = val
        elif key == "outputFrequency":
            deck.output_frequency = int(val)
        elif key == "equationSet":
            deck.equation_set = val
        elif key == "matrixSolver":
            deck.matrix_solver = val
        elif key == "matrixSweeps":
            deck.matrix_sweeps = int(val)
        elif key == "matrixRelaxation":
            deck.matrix_relaxation = float(val)
        elif key == "nonlinearIterations":
            deck.nonlinear_iterations = int(val)
        elif key == "cflMax":
            deck.cfl_max = float(val)
        elif key == "cflStep":
            deck.cfl_step = float(val)
        elif key == "cflStart":
            deck.cfl_start = float(val)
        elif key == "inviscidFluxJacobian":
            deck.inv_flux_jac = val
        elif key == "dualTimeCFL":
            deck.dual_time_cfl = float(val)
        elif key == "inviscidFlux":
            deck.inviscid_flux = val
        elif key == "decompositionMethod":
            deck.decomposition = val
        elif key == "turbulenceModel":
            deck.turbulence_model = val
        elif key == "thermodynamicModel":
            deck.thermodynamic_model = val
        elif key == "multigridLevels":
            deck.multigrid_levels = int(val)
        elif key == "multigridCycle":
            deck.multigrid_cycle = val
        elif key in ("outputVariables", "wallOutputVariables"):
            read_list(val)
        elif key == "initialConditions":
            deck.ics = _parse_state_list(read_list(val))
        elif key == "boundaryStates":
            deck.bc_states = _parse_state_list(read_list(val))
        elif key == "boundaryConditions":
            nblk = int(val)
            for _ in range(nblk):
                idx += 1
                ns = tuple(int(t) for t in lines[idx].split())
                surfs = []
                for _ in range(sum(ns)):
                    idx += 1
                    tok = lines[idx].split()
                    surfs.append(Surface(tok[0], *[int(t) for t in tok[1:8]]))
                surfs.sort(key=Surface.sort_key)          # input.cpp:571-573
                deck.bcs.append(surfs)
                deck.num_surf.append(ns)
        idx += 1

    # input::CheckNonlinearIterations (input.cpp:872-888)
    if deck.time_integration == "rk4":
        deck.nonlinear_iterations = 4
    if deck.time_integration == "explicitEuler":
        deck.nonlinear_iterations = 1
    return deck
