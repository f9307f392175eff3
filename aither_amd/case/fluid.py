"""Fluid database and nondimensionalisation (setup-time, host side).

Restates, for the single-species calorically-perfect case, what the reference
does in fluid.cpp:83-97 (Nondimensionalize), input.cpp:593-613 (reference speed
of sound) and inputStates.cpp:464-473 (IC / BC state nondimensionalisation).
The arithmetic order of the reference is kept so that the nondimensional
constants agree with it to the last bit.
"""
from dataclasses import dataclass
import math

UNIVERSAL_GAS_CONST = 8.3144598  # J / mol-K, include/fluid.hpp:44

# Data of the reference's fluidDatabase/*.dat files (NIST values): n, molar
# mass [g/mol], Sutherland viscosity C1/S, Sutherland conductivity C1/S,
# heat of formation [J/mol].
FLUID_DATABASE = {
    "air": dict(n=2.5, molar_mass=28.97, visc_c1=1.458e-6, visc_s=110.4,
                cond_c1=2.495e-3, cond_s=194.0, heat_of_formation=0.0),
    "N2": dict(n=2.5, molar_mass=28.0134, visc_c1=1.4742e-06, visc_s=1.2846e+02,
               cond_c1=2.6834e-03, cond_s=2.5615e+02, heat_of_formation=0.0),
}


@dataclass
class Gas:
    """Nondimensional gas model handed to the solver (agx_gas)."""
    gas_constant: float
    n: float
    heat_of_formation: float
    visc_c1: float
    visc_s: float
    cond_c1: float
    cond_s: float
    t_ref: float
    rho_ref: float
    l_ref: float
    a_ref: float

    @property
    def gamma(self):
        r = self.gas_constant
        return (r * (self.n + 1.0)) / (r * self.n)


def make_gas(name, t_ref, rho_ref, l_ref=1.0):
    db = FLUID_DATABASE[name]
    n = db["n"]
    molar_mass = db["molar_mass"] / 1000.0          # fluid.cpp:133 (kg/mol)
    r_dim = UNIVERSAL_GAS_CONST / molar_mass         # fluid::GasConstant
    # input.cpp:608-613: aRef_ += mixRef * gamma * R * tRef; aRef_ = sqrt(aRef_)
    gamma = (n + 1) / n
    a_ref = 0.0
    a_ref += 1.0 * gamma * r_dim * t_ref
    a_ref = math.sqrt(a_ref)
    # fluid.cpp:83-97
    hf = db["heat_of_formation"]
    hf /= molar_mass * (a_ref * a_ref)
    molar_mass_nd = molar_mass / (rho_ref / math.pow(l_ref, 3.0))
    ugc_nd = UNIVERSAL_GAS_CONST / (a_ref * a_ref * rho_ref /
                                    (t_ref * math.pow(l_ref, 3.0)))
    r_nd = ugc_nd / molar_mass_nd
    return Gas(gas_constant=r_nd, n=n, heat_of_formation=hf,
               visc_c1=db["visc_c1"], visc_s=db["visc_s"],
               cond_c1=db["cond_c1"], cond_s=db["cond_s"],
               t_ref=t_ref, rho_ref=rho_ref, l_ref=l_ref, a_ref=a_ref)
