// agx_device.hpp -- device-side numerics primitives for gfx950 (fp64).
//
// Per-cell/per-face register work shared by all kernels: thermodynamics,
// face reconstruction, Riemann fluxes, spectral radii, ghost-state rules.
// Each function names the reference function it is the counterpart of
// (paths relative to the reference root); the arithmetic is free to differ in
// operand order / FMA contraction -- parity with the CPU oracle is asserted to
// 1e-10 relative by tests/test_parity_gpu.py, not bitwise.
#pragma once
#include <hip/hip_runtime.h>
#include "../../include/aither_gfx950.h"

#ifndef AGX_NEQ
#define AGX_NEQ 5
#endif
#define AGX_EPS 1.0e-30  // include/macros.hpp.in:20
#define AGX_TURB_MIN 1.0e-20

namespace agx {

// gas model constants resolved once on the host (agx_api.hip: derive_gas)
struct GasDev {
  double R;        // nondimensional gas constant
  double n;        // cv = n R
  double hf;       // heat of formation
  double gamma;    // cp / cv
  double cp, cv;
  double prandtl;  // 4 gamma / (9 gamma - 5)   thermodynamic.hpp:60-63
  double visc_c1, visc_s, cond_c1, cond_s, t_ref;
  double mu_ref;    // sutherland::muMixRef_     transport.cpp:64-66
  double k_nondim;  // sutherland::kNonDim_      transport.cpp:67
  double scaling;   // transport::NondimScaling  transport.hpp:43-46
  double inv_n;     // 1 / n
  double inv_prandtl;
  // rans: turbulence model (0: k-omega SST 2003, 1: k-omega Wilcox 2006) and its
  // turbulent Prandtl number (0.9 / 8/9; turbulence.hpp:500, :398)
  int wilcox;
  int sstdes;      // turbSstDes: SST 2003 whose k destruction is scaled by phi (turbulence.hpp:616-656)
  double turb_prandtl;
};

struct Prim {  // primitive: rho, u, v, w, p  (varArray.hpp:40-51)
  double v[AGX_NEQ];
};

__device__ __forceinline__ double dot3(const double* a, const double* b) {
  return a[0] * b[0] + a[1] * b[1] + a[2] * b[2];
}

// Reciprocal / reciprocal square root from the hardware seeds plus one Newton
// step.  Measured on gfx950 (tools/rcp_accuracy.hip, 4M samples over 2^+-60):
// v_rcp_f64 4.6e-8, v_rsq_f64 5.2e-8 raw; 2.2e-15 / 4.3e-15 after the step --
// five orders inside the 1e-10 parity budget, at 3-4 instructions instead of
// the ~14-instruction IEEE division / ~25-instruction square root sequences.
// Arguments are positive, normal, O(1) nondimensional quantities.
__device__ __forceinline__ double fast_rcp(double x) {
  const double r = __builtin_amdgcn_rcp(x);
  return fma(fma(-x, r, 1.0), r, r);
}
__device__ __forceinline__ double fast_div(double a, double b) { return a * fast_rcp(b); }
__device__ __forceinline__ double fast_rsqrt(double x) {
  const double y = __builtin_amdgcn_rsq(x);
  return fma(0.5 * y, fma(-x * y, y, 1.0), y);
}
__device__ __forceinline__ double fast_sqrt(double x) { return x * fast_rsqrt(x); }

// idealGas::Temperature eos.cpp:100-109
__device__ __forceinline__ double temperature(const GasDev& g, const double* s) {
  return s[4] * fast_rcp(s[0] * g.R);
}
// SpeedOfSound arrayView.hpp:383-391
__device__ __forceinline__ double sound_speed(const GasDev& g, const double* s) {
  return fast_sqrt(g.gamma * s[4] * fast_rcp(s[0]));
}
// rho * H = rho (hf + cp T + |v|^2/2) with rho cp T = (n+1) p: no division,
// no sqrt (EnthalpyFunc arrayView.hpp:401-409 takes |v| by sqrt and squares
// it again, eos.cpp:80-84 -- a last-ulp difference, inside the 1e-10 budget)
__device__ __forceinline__ double rho_enthalpy(const GasDev& g, const double* s) {
  return s[0] * (g.hf + 0.5 * dot3(s + 1, s + 1)) + (g.n + 1.0) * s[4];
}
// rho * E (InternalEnergy arrayView.hpp:434-443): rho cv T = n p
__device__ __forceinline__ double rho_energy(const GasDev& g, const double* s) {
  return s[0] * (g.hf + 0.5 * dot3(s + 1, s + 1)) + g.n * s[4];
}
// PrimToCons primitive.hpp:183-201
__device__ __forceinline__ void prim_to_cons(const GasDev& g, const double* s,
                                             double* u) {
  u[0] = s[0];
  u[1] = s[0] * s[1];
  u[2] = s[0] * s[2];
  u[3] = s[0] * s[3];
  u[4] = rho_energy(g, s);
#pragma unroll
  for (int e = 5; e < AGX_NEQ; ++e) u[e] = s[0] * s[e];    // rho k, rho omega
}
// primitive(cons, phys) primitive.hpp:152-178, idealGas::PressFromEnergy
// eos.cpp:40-52, TemperatureFromSpecEnergy thermodynamic.cpp:108-114
__device__ __forceinline__ void cons_to_prim(const GasDev& g, const double* u,
                                             double* s) {
  const double rho = u[0];
  const double ir = fast_rcp(rho);
  s[0] = rho;
  s[1] = u[1] * ir;
  s[2] = u[2] * ir;
  s[3] = u[3] * ir;
  // p = rho R T, T = (E/rho - |v|^2/2 - hf) / cv  =>  p = (rhoE - rho(...)) / n
  s[4] = (u[4] - rho * (g.hf + 0.5 * dot3(s + 1, s + 1))) * g.inv_n;
  // turbulence variables with primitive::LimitTurb (primitive.cpp:100-106;
  // turbModel::TkeMin / OmegaMin turbulence.hpp:72-73)
#pragma unroll
  for (int e = 5; e < AGX_NEQ; ++e) s[e] = fmax(u[e] / rho, AGX_TURB_MIN);
}
// UpdatePrimWithCons primitive.hpp:206-231
__device__ __forceinline__ void update_prim_with_cons(const GasDev& g,
                                                      const double* s,
                                                      const double* du,
                                                      double* out) {
  double u[AGX_NEQ];
  prim_to_cons(g, s, u);
#pragma unroll
  for (int e = 0; e < AGX_NEQ; ++e) u[e] += du[e];
  cons_to_prim(g, u, out);
}
// sutherland::SpeciesViscosity transport.cpp:114-122
__device__ __forceinline__ double viscosity(const GasDev& g, double t) {
  const double temp = t * g.t_ref;
  return (g.visc_c1 * temp * fast_sqrt(temp)) * fast_rcp((temp + g.visc_s) * g.mu_ref);
}
// sutherland::SpeciesConductivity transport.cpp:124-132
__device__ __forceinline__ double conductivity(const GasDev& g, double t) {
  const double temp = t * g.t_ref;
  return (g.cond_c1 * temp * fast_sqrt(temp)) * fast_rcp((temp + g.cond_s) * g.k_nondim);
}

// ---- limiters src/limiter.cpp:24-54 ---------------------------------------
template <int LIM>
__device__ __forceinline__ double limiter(double r) {
  if (LIM == AGX_LIMITER_VANALBADA) {
    const double r2 = r * r;
    return fmax(0.0, (r + r2) / (1.0 + r2));
  } else if (LIM == AGX_LIMITER_MINMOD) {
    return fmax(0.0, fmin(1.0, r));
  }
  return 1.0;
}

// FaceReconMUSCL reconstruction.hpp:110-154, one variable, ONE division:
// with num = EPS + (dw1-uw1) dPlus, den = EPS + (uw1-uw2) dMinus, r = num/den,
//   vanAlbada  L(r) = max(0, r(1+r)/(1+r^2)) = max(0, num s q)
//              L(1/r)                        = max(0, den s q)
//   with s = num + den, q = 1/(num^2 + den^2), so r L(1/r) = num s q when
//   den s > 0 and 0 otherwise;
//   minmod     r L(1/r) = L(r) = max(0, min(1, r)).
template <int LIM>
__device__ __forceinline__ double muscl(double uw2, double uw1, double dw1,
                                        double dPlus, double dMinus,
                                        double kappa) {
  const double dm = (uw1 - uw2) * dMinus;
  const double num = AGX_EPS + (dw1 - uw1) * dPlus;
  const double den = AGX_EPS + dm;
  double lim, rinv;
  if (LIM == AGX_LIMITER_VANALBADA) {
    const double sq = (num + den) * fast_rcp(num * num + den * den);
    const double nsq = num * sq;
    lim = fmax(0.0, nsq);
    rinv = den * sq > 0.0 ? nsq : 0.0;
  } else if (LIM == AGX_LIMITER_MINMOD) {
    lim = fmax(0.0, fmin(1.0, fast_div(num, den)));
    rinv = lim;
  } else {
    lim = 1.0;
    rinv = fast_div(num, den);
  }
  // 0.25 (1 -+ kappa) are wave-uniform scalars
  return uw1 + dm * ((0.25 * (1.0 - kappa)) * lim + (0.25 * (1.0 + kappa)) * rinv);
}

// ---- WENO5 on non-uniform widths, reconstruction.hpp:158-310 --------------
// LagrangeCoeff src/utility.cpp:449-483 (width-only; evaluated per face side)
__device__ __forceinline__ double stencil_width(const double* w, int s, int e) {
  double acc = 0.0;
  if (e > s) {
    for (int q = s; q < e; ++q) acc += w[q];
  } else if (s > e) {
    for (int q = e; q < s; ++q) acc += w[q];
    acc = -acc;
  }
  return acc;
}
template <int DEGREE>
__device__ __forceinline__ void lagrange_coeff(const double* w, int rr, int ii,
                                               double* coeffs) {
#pragma unroll
  for (int jj = 0; jj <= DEGREE; ++jj) {
    double c = 0.0;
#pragma unroll
    for (int mm = jj + 1; mm <= DEGREE + 1; ++mm) {
      double numer = 0.0, denom = 1.0;
#pragma unroll
      for (int ll = 0; ll <= DEGREE + 1; ++ll) {
        if (ll != mm) {
          double prod = 1.0;
#pragma unroll
          for (int qq = 0; qq <= DEGREE + 1; ++qq)
            if (qq != mm && qq != ll)
              prod *= stencil_width(w, ii - rr + qq, ii + 1);
          numer += prod;
          denom *= stencil_width(w, ii - rr + ll, ii - rr + mm);
        }
      }
      c += fast_div(numer, denom);
    }
    coeffs[jj] = c * w[ii - rr + jj];
  }
}
// FaceReconWENO reconstruction.hpp:244-310 in divided-difference (Newton) form.
//
// The reference evaluates Shu's formula 2.20 (LagrangeCoeff, utility.cpp:449-483: triple
// loops over products of stencil widths, nine coefficients for the three quadratic
// sub-stencils and five for the quartic one) per face side.  The same polynomials written
// on the divided differences of the cell averages need a handful of width sums and
// reciprocals instead: with the five widths w0..w4 (w2: the cell next to the face, w3, w4
// downwind), the face at the upper edge of cell 2, pair sums p_a = w_a + w_{a+1},
//     H_a = (u_{a+1} - u_a) * 2 / p_a               (twice the first divided difference)
//     E_m = H_{m+1} - H_m
//     s0 = u2 + w2/2 H_1 + [w2 p_1 / (2 (w0+w1+w2))] E_0
//     s1 = u2 + w2/2 H_2 - [w2 w3 / (2 (w1+w2+w3))] E_1
//     s2 = u2 + w2/2 H_2 - [w2 w3 / (2 (w2+w3+w4))] E_2
// (uniform widths: 1/3 u0 - 7/6 u1 + 11/6 u2 etc.), and the linear weights
//     fullCoeffs[0] / coeffs0[0] = w3 (w3+w4) / ((w0+..+w3)(w0+..+w4))
//     fullCoeffs[4] / coeffs2[2] = (w0+w1+w2)(w1+w2) / ((w0+..+w4)(w1+..+w4))
// (1/10 and 3/10; the reference gives the SECOND one to stencil 1 and 1 - both to stencil 2,
// reconstruction.hpp:281-283, kept).  H and E are also exactly what Derivative2nd
// (utility.hpp:114-120) and Beta0/1/2 (reconstruction.hpp:186-240) are made of:
//     deriv2nd_m = E_m * 4 / (p_m + p_{m+1}),   deriv1st = H_1 + w2/2 d2 | H_2 - w2/2 d2.
// 84 + 56 per variable fp64 instructions per face side instead of 258 + 71, and a
// width-only set of 16 doubles instead of 19 + the five widths (tools/weno_closed_form.py
// checks the algebra against formula 2.20 on random widths).
struct WenoCoeffs {  // everything that depends on cell widths only
  double ih[4];        // 2 / p_a
  double iq[3];        // 4 / (p_m + p_{m+1})
  double a2;           // w2 / 2
  double k0, k1, k2;   // the bracketed factors of E_0, E_1, E_2 above
  double lw0, lw1, lw2;
  double dx2, c13;     // w2^2, 13/12 w2^2
};
__device__ __forceinline__ void weno_coeffs(const double* cw, WenoCoeffs& k) {
  const double p0 = cw[0] + cw[1], p1 = cw[1] + cw[2], p2 = cw[2] + cw[3], p3 = cw[3] + cw[4];
  k.ih[0] = fast_rcp(0.5 * p0); k.ih[1] = fast_rcp(0.5 * p1);
  k.ih[2] = fast_rcp(0.5 * p2); k.ih[3] = fast_rcp(0.5 * p3);
  k.iq[0] = fast_rcp(0.25 * (p0 + p1));
  k.iq[1] = fast_rcp(0.25 * (p1 + p2));
  k.iq[2] = fast_rcp(0.25 * (p2 + p3));
  const double s012 = p0 + cw[2], s123 = p1 + cw[3], s234 = p2 + cw[4];
  k.a2 = 0.5 * cw[2];
  const double a23 = k.a2 * cw[3];
  k.k0 = k.a2 * p1 * fast_rcp(s012);
  k.k1 = a23 * fast_rcp(s123);
  k.k2 = a23 * fast_rcp(s234);
  const double w4 = p0 + p2, w5 = w4 + cw[4];
  k.lw0 = cw[3] * p3 * fast_rcp(w4 * w5);
  k.lw1 = s012 * p1 * fast_rcp(w5 * (p1 + p3));
  k.lw2 = 1.0 - k.lw0 - k.lw1;
  k.dx2 = cw[2] * cw[2];
  k.c13 = (13.0 / 12.0) * k.dx2;
}
// BetaIntegral reconstruction.hpp:158-183 between -dx/2 and +dx/2 (the only limits
// Beta0/1/2 use): the terms odd in x cancel exactly, leaving dx^2 (d1^2 + 13/12 d2^2 dx^2)
template <bool WENOZ>
__device__ __forceinline__ double weno(const WenoCoeffs& k, double u0, double u1, double u2,
                                       double u3, double u4) {
  const double h0 = (u1 - u0) * k.ih[0], h1 = (u2 - u1) * k.ih[1];
  const double h2 = (u3 - u2) * k.ih[2], h3 = (u4 - u3) * k.ih[3];
  const double e0 = h1 - h0, e1 = h2 - h1, e2 = h3 - h2;
  const double c2 = fma(k.a2, h2, u2);
  const double s0 = fma(k.k0, e0, fma(k.a2, h1, u2));
  const double s1 = fma(-k.k1, e1, c2);
  const double s2 = fma(-k.k2, e2, c2);
  const double dd0 = e0 * k.iq[0], dd1 = e1 * k.iq[1], dd2 = e2 * k.iq[2];
  const double df0 = fma(k.a2, dd0, h1), df1 = fma(-k.a2, dd1, h2), df2 = fma(-k.a2, dd2, h2);
  const double b0 = k.dx2 * fma(k.c13, dd0 * dd0, df0 * df0);
  const double b1 = k.dx2 * fma(k.c13, dd1 * dd1, df1 * df1);
  const double b2 = k.dx2 * fma(k.c13, dd2 * dd2, df2 * df2);
  double n0, n1, n2;
  if (WENOZ) {
    const double tau5 = fabs(b0 - b2);
    const double q0 = fast_div(tau5, 1.0e-40 + b0), q1 = fast_div(tau5, 1.0e-40 + b1),
                 q2 = fast_div(tau5, 1.0e-40 + b2);
    n0 = k.lw0 * (1.0 + q0 * q0);
    n1 = k.lw1 * (1.0 + q1 * q1);
    n2 = k.lw2 * (1.0 + q2 * q2);
  } else {
    // lw_m / (eps + b_m)^2, all three scaled by the product of the squares: one
    // reciprocal (of the sum) instead of four
    const double g0 = 1.0e-6 + b0, g1 = 1.0e-6 + b1, g2 = 1.0e-6 + b2;
    const double p12 = g1 * g2, p02 = g0 * g2, p01 = g0 * g1;
    n0 = k.lw0 * (p12 * p12);
    n1 = k.lw1 * (p02 * p02);
    n2 = k.lw2 * (p01 * p01);
  }
  const double inv = fast_rcp(n0 + n1 + n2);
  return fma(n0, s0, fma(n1, s1, n2 * s2)) * inv;
}

// ---- inviscid fluxes --------------------------------------------------------
// inviscidFlux::ConstructFromPrim inviscidFlux.hpp:129-160
__device__ __forceinline__ void phys_flux(const GasDev& g, const double* s,
                                          const double* n, double* f) {
  const double vn = dot3(s + 1, n);
  const double m = s[0] * vn;
  f[0] = m;
  f[1] = m * s[1] + s[4] * n[0];
  f[2] = m * s[2] + s[4] * n[1];
  f[3] = m * s[3] + s[4] * n[2];
  f[4] = vn * rho_enthalpy(g, s);
#pragma unroll
  for (int e = 5; e < AGX_NEQ; ++e) f[e] = m * s[e];
}

// RoeFlux inviscidFlux.hpp:260-382 with RoeAveragedState primitive.hpp:245-280
// (pressure is Roe-averaged, enthalpy derived from it) and Harten's fix 0.1.
// 2 reciprocals + 2 reciprocal square roots per face.
__device__ __forceinline__ void roe_flux(const GasDev& g, const double* l,
                                         const double* r, const double* n,
                                         double* flux) {
  double roe[AGX_NEQ];
  const double dr = r[0] * fast_rsqrt(r[0] * l[0]);      // sqrt(rhoR / rhoL)
  const double inv1 = fast_rcp(1.0 + dr);
  roe[0] = l[0] * dr;
#pragma unroll
  for (int e = 1; e < AGX_NEQ; ++e) roe[e] = (l[e] + dr * r[e]) * inv1;
  const double p_rho = roe[4] * fast_rcp(roe[0]);
  const double v2R = dot3(roe + 1, roe + 1);
  const double hR = g.hf + (g.n + 1.0) * p_rho + 0.5 * v2R;
  const double a2 = g.gamma * p_rho;
  const double ia = fast_rsqrt(a2);
  const double aR = a2 * ia;
  const double inv_a2 = ia * ia;
  const double rhoR = roe[0];
  const double vnR = dot3(roe + 1, n);
  double d[AGX_NEQ];
#pragma unroll
  for (int e = 0; e < AGX_NEQ; ++e) d[e] = r[e] - l[e];
  const double dvn = dot3(d + 1, n);
  const double fix = 0.1;
  double diss[AGX_NEQ];
  // left acoustic
  double ws = fabs(vnR - aR);
  if (ws < fix) ws = 0.5 * (ws * ws * (1.0 / fix) + fix);
  double wss = ws * (d[4] - rhoR * aR * dvn) * 0.5 * inv_a2;
  diss[0] = wss;
  diss[1] = wss * (roe[1] - aR * n[0]);
  diss[2] = wss * (roe[2] - aR * n[1]);
  diss[3] = wss * (roe[3] - aR * n[2]);
  diss[4] = wss * (hR - aR * vnR);
#pragma unroll
  for (int e = 5; e < AGX_NEQ; ++e) diss[e] = wss * roe[e];
  // entropy
  ws = fabs(vnR);
  wss = ws * (d[0] - d[4] * inv_a2);
  diss[0] += wss;
  diss[1] += wss * roe[1];
  diss[2] += wss * roe[2];
  diss[3] += wss * roe[3];
  diss[4] += wss * 0.5 * v2R;
  // shear
  wss = ws * rhoR;
  diss[1] += wss * (d[1] - dvn * n[0]);
  diss[2] += wss * (d[2] - dvn * n[1]);
  diss[3] += wss * (d[3] - dvn * n[2]);
  diss[4] += wss * (dot3(roe + 1, d + 1) - vnR * dvn);
  // right acoustic
  ws = fabs(vnR + aR);
  if (ws < fix) ws = 0.5 * (ws * ws * (1.0 / fix) + fix);
  wss = ws * (d[4] + rhoR * aR * dvn) * 0.5 * inv_a2;
  diss[0] += wss;
  diss[1] += wss * (roe[1] + aR * n[0]);
  diss[2] += wss * (roe[2] + aR * n[1]);
  diss[3] += wss * (roe[3] + aR * n[2]);
  diss[4] += wss * (hR + aR * vnR);
#pragma unroll
  for (int e = 5; e < AGX_NEQ; ++e) {
    diss[e] += wss * roe[e];
    // turbulence waves, inviscidFlux.hpp:363-372
    diss[e] += fabs(vnR) * (rhoR * d[e] + roe[e] * d[0] - d[4] * roe[e] * inv_a2);
  }
  double fl[AGX_NEQ], fr[AGX_NEQ];
  phys_flux(g, l, n, fl);
  phys_flux(g, r, n, fr);
#pragma unroll
  for (int e = 0; e < AGX_NEQ; ++e) flux[e] = 0.5 * (fl[e] + fr[e] - diss[e]);
}

// AUSMFlux (AUSMPW+) inviscidFlux.hpp:396-481 and member :162-209
__device__ __forceinline__ void ausm_flux(const GasDev& g, const double* l,
                                          const double* r, const double* n,
                                          double* f) {
  const double vnL = dot3(l + 1, n), vnR = dot3(r + 1, n);
  // sqrt(cL cR) = (gamma^2 pL pR / (rhoL rhoR))^(1/4)
  const double cS = fast_sqrt(fast_sqrt(g.gamma * g.gamma * (l[4] * r[4]) *
                                        fast_rcp(l[0] * r[0])));
  const double vel = 0.5 * (vnL + vnR);
  double c = cS;
  if (vel < 0.0) c = cS * cS * fast_rcp(fmax(vnR, cS));
  else if (vel > 0.0) c = cS * cS * fast_rcp(fmax(vnL, cS));
  const double ic = fast_rcp(c);
  const double ml = vnL * ic, mr = vnR * ic;
  const double sl = (ml > 0.0) - (ml < 0.0), sr = (mr > 0.0) - (mr < 0.0);
  const bool subl = fabs(ml) <= 1.0, subr = fabs(mr) <= 1.0;
  const double mPlusL = subl ? 0.25 * (ml + 1.0) * (ml + 1.0) : 0.5 * (ml + fabs(ml));
  const double mMinusR = subr ? -0.25 * (mr - 1.0) * (mr - 1.0) : 0.5 * (mr - fabs(mr));
  const double pPlus = subl ? 0.25 * (ml + 1.0) * (ml + 1.0) * (2.0 - ml) : 0.5 * (1.0 + sl);
  const double pMinus = subr ? 0.25 * (mr - 1.0) * (mr - 1.0) * (2.0 + mr) : 0.5 * (1.0 - sr);
  const double ps = pPlus * l[4] + pMinus * r[4];
  const double ips = fast_rcp(ps);
  const double pm = fmin(l[4], r[4]) * fast_rcp(fmax(l[4], r[4]));   // min(pL/pR, pR/pL)
  const double w = 1.0 - pm * pm * pm;
  const double fl = fabs(ml) < 1.0 ? l[4] * ips - 1.0 : 0.0;
  const double fr = fabs(mr) < 1.0 ? r[4] * ips - 1.0 : 0.0;
  const double mavg = mPlusL + mMinusR;
  const double mPlusLBar = mavg >= 0.0
      ? mPlusL + mMinusR * ((1.0 - w) * (1.0 + fr) - fl) : mPlusL * w * (1.0 + fl);
  const double mMinusRBar = mavg >= 0.0
      ? mMinusR * w * (1.0 + fr) : mMinusR + mPlusL * ((1.0 - w) * (1.0 + fl) - fr);
  const double vl = mPlusLBar * c, vr = mMinusRBar * c;
  f[0] = l[0] * vl + r[0] * vr;
  f[1] = l[0] * vl * l[1] + r[0] * vr * r[1] + ps * n[0];
  f[2] = l[0] * vl * l[2] + r[0] * vr * r[2] + ps * n[1];
  f[3] = l[0] * vl * l[3] + r[0] * vr * r[3] + ps * n[2];
  f[4] = vl * rho_enthalpy(g, l) + vr * rho_enthalpy(g, r);
#pragma unroll
  for (int e = 5; e < AGX_NEQ; ++e) f[e] = l[0] * vl * l[e] + r[0] * vr * r[e];
}

template <int FLUX>
__device__ __forceinline__ void inviscid_flux(const GasDev& g, const double* l,
                                              const double* r, const double* n,
                                              double* f) {
  if (FLUX == AGX_FLUX_ROE) roe_flux(g, l, r, n, f);
  else ausm_flux(g, l, r, n, f);
}

// InvCellSpectralRadius spectralRadius.hpp:44-64; al/au = {nx,ny,nz,|A|}
__device__ __forceinline__ double inv_cell_spec_rad(const GasDev& g,
                                                    const double* s,
                                                    const double* al,
                                                    const double* au) {
  double v[3] = {0.5 * (al[0] + au[0]), 0.5 * (al[1] + au[1]),
                 0.5 * (al[2] + au[2])};
  const double im = fast_rsqrt(dot3(v, v));
  const double fmag = 0.5 * (al[3] + au[3]);
  return (fabs(dot3(s + 1, v)) * im + sound_speed(g, s)) * fmag;
}
// viscous term of ViscCell/FaceSpectralRadius spectralRadius.hpp:94-160
__device__ __forceinline__ double visc_max_term(const GasDev& g, double rho) {
  const double ir = fast_rcp(rho);
  return fmax((4.0 / 3.0) * ir, g.gamma * ir);
}
__device__ __forceinline__ double visc_term(const GasDev& g, double mu) {
  return g.scaling * (mu * g.inv_prandtl);
}

// sigma_k / sigma_w of the k / omega diffusion (SST: blended; Wilcox: sigmaStar, sigma) and
// the eddy viscosity of that diffusion and of the turbulence spectral radii: the limited
// one (SST) or EddyViscosityNoLim = rho k / omega (Wilcox, UseUnlimitedEddyVisc)
__device__ __forceinline__ double turb_sigma_k(const GasDev& g, double f1) {
  return g.wilcox ? 0.6 : f1 * 0.85 + (1.0 - f1) * 1.0;
}
__device__ __forceinline__ double turb_sigma_w(const GasDev& g, double f1) {
  return g.wilcox ? 0.5 : f1 * 0.5 + (1.0 - f1) * 0.856;
}
__device__ __forceinline__ double turb_diff_visc(const GasDev& g, const double* s, double mut) {
  return (AGX_NEQ > 5 && g.wilcox) ? s[0] * s[AGX_NEQ - 2] / s[AGX_NEQ - 1] : mut;
}

// RusanovScalarOffDiagonal fluxJacobian.cpp:122-162 with FaceSpectralRadius
// spectralRadius.hpp:182-203 and ConvectiveFluxUpdate inviscidFlux.hpp:544-562
// diag != nullptr selects RoeOffDiagonal (fluxJacobian.cpp:240-291, inviscid): the
// change of the Roe flux between the neighbour and the cell `diag`
__device__ __forceinline__ void off_diagonal(const GasDev& g, bool viscous,
                                             const double* s, const double* du,
                                             const double* area, double mu,
                                             double dist, bool positive,
                                             double* out, const double* diag = nullptr,
                                             double mut = 0.0, double f1 = 0.0) {
  double su[AGX_NEQ], fo[AGX_NEQ], fn[AGX_NEQ];
  update_prim_with_cons(g, s, du, su);
  if (diag) {
    roe_flux(g, s, diag, area, fo);
    if (positive) roe_flux(g, su, diag, area, fn);
    else roe_flux(g, diag, su, area, fn);
#pragma unroll
    for (int e = 0; e < AGX_NEQ; ++e) out[e] = area[3] * (fn[e] - fo[e]);
    return;
  }
  phys_flux(g, s, area, fo);
  phys_flux(g, su, area, fn);
  double sr = 0.5 * area[3] * (fabs(dot3(s + 1, area)) + sound_speed(g, s));
  if (viscous)
    sr += area[3] * fast_rcp(dist) * visc_max_term(g, s[0]) *
          (AGX_NEQ > 5 ? g.scaling * (mu * g.inv_prandtl + mut / g.turb_prandtl) : visc_term(g, mu));
  const double sg = positive ? 1.0 : -1.0;
#pragma unroll
  for (int e = 0; e < 5; ++e)
    out[e] = 0.5 * area[3] * (fn[e] - fo[e]) + sg * du[e] * sr;
  if (AGX_NEQ > 5) {
    // turbulence entries: no flux change (fluxJacobian.cpp:145-148); spectral radius
    // turbModel::FaceSpectralRadius turbulence.hpp:309-330 = InviscidFaceSpectralRadius
    // (turbulence.cpp:174-186) + turbKWSst::ViscousFaceSpectralRadius (:817-831)
    const double vn = dot3(s + 1, area);
    double tsr = positive ? 0.5 * area[3] * fabs(vn + fabs(vn)) : 0.5 * area[3] * fabs(vn - fabs(vn));
    tsr += g.scaling * (area[3] / dist) / s[0] *
           (mu + turb_sigma_k(g, f1) * turb_diff_visc(g, s, mut));
#pragma unroll
    for (int e = 5; e < AGX_NEQ; ++e) out[e] = sg * du[e] * tsr;
  }
}

// ---- block-matrix solvers (blusgs / bdplur): 5 x 5 flow Jacobians, row major ----
constexpr int AGX_NF = 5;                 // flow equations (the block of the block solvers)
constexpr int AGX_NJ = AGX_NF * AGX_NF;
// fluxJacobian::InvFluxJacobian fluxJacobian.hpp:483-560 (one species, mf = 1)
__device__ inline void inv_flux_jacobian(const GasDev& g, const double* s, const double* area,
                                         double* J) {
  const double* n = area;
  const double vn = dot3(s + 1, n);
  const double gm1 = g.gamma - 1.0;
  const double phi = 0.5 * gm1 * dot3(s + 1, s + 1);
  double u[AGX_NF];
  prim_to_cons(g, s, u);
  const double a1 = g.gamma * (u[4] * fast_rcp(s[0])) - phi;    // primitive::Energy
  const double a3 = g.gamma - 2.0;
#pragma unroll
  for (int q = 0; q < AGX_NJ; ++q) J[q] = 0.0;
  J[0] = vn * (1.0 - 1.0);
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    J[1 + q] = 1.0 * n[q];
    J[AGX_NF * (1 + q)] = phi * n[q] - s[1 + q] * vn;
  }
  J[AGX_NF * 4] = vn * (phi - a1);
#pragma unroll
  for (int c = 0; c < 3; ++c) {
#pragma unroll
    for (int r = 0; r < 3; ++r)
      J[AGX_NF * (1 + r) + 1 + c] = r == c ? vn - a3 * n[c] * s[1 + c]
                                            : s[1 + r] * n[c] - gm1 * s[1 + c] * n[r];
    J[AGX_NF * 4 + 1 + c] = a1 * n[c] - gm1 * s[1 + c] * vn;
    J[AGX_NF * (1 + c) + 4] = gm1 * n[c];
  }
  J[AGX_NF * 4 + 4] = g.gamma * vn;
  const double h = 0.5 * area[3];
#pragma unroll
  for (int q = 0; q < AGX_NJ; ++q) J[q] *= h;
}
// fluxJacobian::RusanovFluxJacobian :446-479 with InvFaceSpectralRadius
// spectralRadius.hpp:67-80
__device__ inline void rusanov_flux_jacobian(const GasDev& g, const double* s, const double* area,
                                             bool positive, double* J) {
  const double sr = 0.5 * area[3] * (fabs(dot3(s + 1, area)) + sound_speed(g, s));
  inv_flux_jacobian(g, s, area, J);
#pragma unroll
  for (int e = 0; e < AGX_NF; ++e) J[AGX_NF * e + e] += positive ? sr : -sr;
}
// fluxJacobian::ApproxTSLJacobian :660-758 (laminar, one species) times
// DelprimitiveDelConservative :613-656; TauNormal utility.cpp:426-436.  vg[3 r + c]:
// velocity gradient (only its trace and symmetric part enter)
__device__ inline void tsl_jacobian(const GasDev& g, const double* s, double lam_visc,
                                    const double* area, double dist, bool left, const double* vg,
                                    double* J, double turb_visc = 0.0) {
  const double t = temperature(g, s);
  // (mu: laminar + eddy viscosity; the conductivity below gets its turbulent part)
  const double mut = g.scaling * turb_visc;
  const double mu = g.scaling * lam_visc + mut;
  const double* n = area;
  const double vn = dot3(s + 1, n);
  const double rho = s[0];
  const double k = conductivity(g, t) * g.scaling +
                   (AGX_NEQ > 5 ? mut * g.cp / g.turb_prandtl : 0.0);
  const double lambda = 0.0 - (2.0 / 3.0) * mu;
  const double trace = vg[0] + vg[4] + vg[8];
  double tau[3];
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    double mm = 0.0;
#pragma unroll
    for (int q = 0; q < 3; ++q) mm += (vg[3 * r + q] + vg[3 * q + r]) * n[q];
    tau[r] = lambda * trace * n[r] + mu * mm;
  }
  const double fac = left ? -1.0 : 1.0;
  const double third = 1.0 / 3.0;
  // T (the thin-shear-layer matrix in primitive variables) has entries in its velocity block
  // and its last row only, P = d(primitive)/d(conservative) in its first column, its diagonal
  // and its last row: the product MatrixMultiply (matrix.cpp:193-207) forms with 125
  // multiply-adds is written out on the 35 that are not products with a structural zero, in
  // the same order of summation (c ascending), so the result is the same number.
  // (three reciprocals -- 1 / mu, 1 / rho, 1 / dist -- for the function's eight quotients)
  const double imu = fast_rcp(mu), ir = fast_rcp(rho), imr = imu * ir;
  const double sc = area[3] * mu * fast_rcp(dist);
  double Tv[3][3], T4[AGX_NF];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
#pragma unroll
    for (int r = 0; r < 3; ++r) Tv[r][c] = (third * n[c] * n[r] + (r == c ? 1.0 : 0.0)) * sc;
    T4[1 + c] = (fac * 0.5 * dist * imu * tau[c] + third * n[c] * vn + s[1 + c]) * sc;
  }
  T4[0] = (-k * t * imr + 0.0) * sc;
  T4[4] = (k * imr) * sc;
  const double gm1 = g.gamma - 1.0;
  double Pq0[3], P4q[3];
#pragma unroll
  for (int q = 0; q < 3; ++q) { Pq0[q] = -ir * s[1 + q]; P4q[q] = -gm1 * s[1 + q]; }
  const double P40 = 0.5 * gm1 * dot3(s + 1, s + 1), P44 = gm1;
#pragma unroll
  for (int q = 0; q < AGX_NJ; ++q) J[q] = 0.0;
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    double a = 0.0;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      a = fma(Tv[r][c], Pq0[c], a);
      J[AGX_NF * (1 + r) + 1 + c] = Tv[r][c] * ir;
    }
    J[AGX_NF * (1 + r)] = a;
  }
  {
    double a = T4[0] * 1.0;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      a = fma(T4[1 + c], Pq0[c], a);
      J[AGX_NF * 4 + 1 + c] = fma(T4[4], P4q[c], T4[1 + c] * ir);
    }
    J[AGX_NF * 4] = fma(T4[4], P40, a);
    J[AGX_NF * 4 + 4] = T4[4] * P44;
  }
}
// MatrixInverse matrix.cpp:57-103 (Gauss-Jordan with row exchanges); m becomes its
// inverse; returns false for a singular matrix.  Fully unrolled, so the 50 doubles
// stay in registers (row exchanges are conditional swaps).
__device__ inline bool matrix_inverse5(double* m) {
  constexpr int N = AGX_NF;
  double I[AGX_NJ];
#pragma unroll
  for (int q = 0; q < AGX_NJ; ++q) I[q] = (q / N == q % N) ? 1.0 : 0.0;
  bool ok = true;
#pragma unroll
  for (int r = 0; r < N; ++r) {
    // FindMaxInColumn(mat, size, r, r, size - 1)
    double mx = 0.0;
    int rp = 0;
#pragma unroll
    for (int ii = r; ii < N; ++ii)
      if (fabs(m[ii * N + r]) > mx) { mx = fabs(m[ii * N + r]); rp = ii; }
#pragma unroll
    for (int ii = r + 1; ii < N; ++ii) {       // swap rows r and rp (rp == 0 only if column is 0)
      const bool sw = rp == ii;
#pragma unroll
      for (int q = 0; q < N; ++q) {
        const double a = m[r * N + q], bq = m[ii * N + q];
        m[r * N + q] = sw ? bq : a;
        m[ii * N + q] = sw ? a : bq;
        const double c = I[r * N + q], dq = I[ii * N + q];
        I[r * N + q] = sw ? dq : c;
        I[ii * N + q] = sw ? c : dq;
      }
    }
#pragma unroll
    for (int ii = 0; ii < r; ++ii) {
      const double factor = m[r * N + ii] / m[ii * N + ii];
#pragma unroll
      for (int q = 0; q < N; ++q) {
        m[r * N + q] = m[r * N + q] - factor * m[ii * N + q];
        I[r * N + q] = I[r * N + q] - factor * I[ii * N + q];
      }
    }
    if (m[r * N + r] == 0.0) ok = false;
    const double nf = 1.0 / m[r * N + r];
#pragma unroll
    for (int q = r; q < N; ++q) m[r * N + q] *= nf;
#pragma unroll
    for (int q = 0; q < N; ++q) I[r * N + q] *= nf;
  }
#pragma unroll
  for (int r = N - 2; r >= 0; --r)
#pragma unroll
    for (int ii = N - 1; ii > r; --ii) {
      const double factor = m[r * N + ii];
#pragma unroll
      for (int q = 0; q < N; ++q) {
        m[r * N + q] = m[r * N + q] - factor * m[ii * N + q];
        I[r * N + q] = I[r * N + q] - factor * I[ii * N + q];
      }
    }
#pragma unroll
  for (int q = 0; q < AGX_NJ; ++q) m[q] = I[q];
  return ok;
}
// ArrayMultiplication fluxJacobian.hpp:50-87 (block branch)
__device__ __forceinline__ void mat_vec5(const double* m, const double* v, double* out) {
#pragma unroll
  for (int r = 0; r < AGX_NF; ++r) {
    double a = 0.0;
#pragma unroll
    for (int c = 0; c < AGX_NF; ++c) a += m[AGX_NF * r + c] * v[c];
    out[r] = a;
  }
}
// RusanovBlockOffDiagonal fluxJacobian.cpp:164-194
// diagonal turbulence block of the Jacobians (rans + block-matrix solvers):
// turbModel::InvJac turbulence.cpp:117-160, turbKWSst::ViscJac :772-795
__device__ __forceinline__ double turb_inv_jac(const double* s, const double* area, bool positive) {
  const double vn = dot3(s + 1, area);
  return positive ? 0.5 * (vn * area[3] + fabs(vn) * area[3])
                  : 0.5 * (vn * area[3] - fabs(vn) * area[3]);
}
__device__ __forceinline__ void turb_visc_jac(const GasDev& g, const double* s, const double* area,
                                              double mu, double dist, double mut, double f1,
                                              double& jk, double& jw) {
  const double len = g.scaling * (area[3] * fast_rcp(dist)) * fast_rcp(s[0]);
  jk = len * (mu + turb_sigma_k(g, f1) * turb_diff_visc(g, s, mut));
  jw = len * (mu + turb_sigma_w(g, f1) * turb_diff_visc(g, s, mut));
}
__device__ inline void block_off_diagonal(const GasDev& g, bool viscous, const double* s,
                                          const double* du, const double* area, double mu,
                                          double dist, bool positive, const double* vg,
                                          double* out, double mut = 0.0, double f1 = 0.0) {
  double J[AGX_NJ];
  rusanov_flux_jacobian(g, s, area, positive, J);
  if (viscous) {
    double V[AGX_NJ];
    tsl_jacobian(g, s, mu, area, dist, positive, vg, V, mut);
#pragma unroll
    for (int q = 0; q < AGX_NJ; ++q) J[q] = positive ? J[q] - V[q] : J[q] + V[q];
  }
  mat_vec5(J, du, out);
  if (AGX_NEQ > 5) {
    // (both signs give InvJac + ViscJac: fac = -1 goes with the subtraction)
    double jk = 0.0, jw = 0.0;
    if (viscous) turb_visc_jac(g, s, area, mu, dist, mut, f1, jk, jw);
    const double tj = turb_inv_jac(s, area, positive);
    out[AGX_NEQ - 2] = (tj + jk) * du[AGX_NEQ - 2];
    out[AGX_NEQ - 1] = (tj + jw) * du[AGX_NEQ - 1];
  }
}

// ---- ghost states, ghostStates.cpp:62-708 ----------------------------------
__device__ __forceinline__ void extrap_hold(const double* bnd, double factor,
                                            const double* in, double* out) {
  // ExtrapolateHoldMixture ghostStates.cpp:691-708
  const double grho = factor * bnd[0] - in[0];
  if (grho <= 0.0) {
#pragma unroll
    for (int e = 0; e < AGX_NEQ; ++e) out[e] = bnd[e];
    return;
  }
  double t[AGX_NEQ];
#pragma unroll
  for (int e = 0; e < AGX_NEQ; ++e) t[e] = factor * bnd[e] - in[e];
  t[0] = fmax(grho, 0.0);
#pragma unroll
  for (int e = 0; e < AGX_NEQ; ++e) out[e] = t[e];
}

// what the nonreflecting (LODI) branches of GetGhostState read besides the interior
// state: dt and the state at time n of the adjacent cell, its pressure / velocity
// gradients of the last residual, Mach mean / maximum over the surface
// (procBlock.cpp:2497-2512, :6233-6262)
struct NrDev { double dt, sn[AGX_NEQ], pg[3], vg[9], avg_mach, max_mach; };

// GetGhostState; returns false for a BC variant this build does not cover
// primitive::ApplyFarfieldTurbBC primitive.cpp:83-98 (rans builds)
__device__ inline void apply_farfield_turb(const GasDev& g, double* s, const double* vel,
                                           double intensity, double ratio) {
  if (AGX_NEQ > 5) {
    const double q = intensity * sqrt(dot3(vel, vel));
    s[AGX_NEQ - 2] = fmax(1.5 * q * q, AGX_TURB_MIN);
    s[AGX_NEQ - 1] = fmax(s[0] * s[AGX_NEQ - 2] / (ratio * viscosity(g, temperature(g, s))),
                          AGX_TURB_MIN);
  }
}
// ---- wall functions (rans): wallLaw::AdiabaticBCs wallLaw.cpp:30-77 with its helpers
// (:182-289) and Ridder's FindRoot (utility.hpp:130-184).  The function whose root is
// sought has side effects: what is kept belongs to its LAST evaluation.
struct WallVars {     // wallVars wallData.hpp:33-62, what the solver reads (12 doubles)
  double yplus, heat_flux, density, temperature, viscosity, turb_eddy_visc, friction_velocity,
         shear[3], tke, sdr;
};
struct WallLawDev {
  double von_karman, wall_dist, yplus0, beta, gamma, q, phi, yplus_white, u_star, uplus, tw, rho_w,
         mu_w, k_w, recovery, vel_tan, heat_flux, yplus_last, cp;
  int mode;          // 0 adiabatic, 1 constant heat flux, 2 isothermal (wallLaw.cpp:31, :89, :147)
  double t_int, p_int;
  const GasDev* gas;
  // wallLaw::SetWallVars wallLaw.cpp:239-246
  __device__ void set_wall_vars(double t) {
    tw = t;
    rho_w = p_int / (gas->R * t);
    mu_w = viscosity(*gas, t) * gas->scaling;
    k_w = conductivity(*gas, t) * gas->scaling;
  }
  __device__ double func(double yplus) {
    uplus = (wall_dist * rho_w * vel_tan) / (mu_w * yplus);
    u_star = vel_tan / uplus;
    if (mode == 1) {
      // HeatFluxBCs :113-124: wall temperature from Crocco-Busemann with the wall properties
      // of the PREVIOUS evaluation (CalcWallTemperature :231-237), then SetWallVars
      set_wall_vars(t_int + recovery * u_star * u_star * uplus * uplus /
                                (2.0 * cp + heat_flux * mu_w / (rho_w * k_w * u_star)));
    }
    gamma = recovery * u_star * u_star / (2.0 * cp * tw);
    if (mode == 2) {     // IsothermalBCs :170-172, CalcHeatFlux :223-229
      const double tmp = (t_int / tw - 1.0 + gamma * uplus * uplus) / uplus;
      heat_flux = tmp * (rho_w * tw * k_w * u_star) / mu_w;
    }
    beta = heat_flux * mu_w / (rho_w * tw * k_w * u_star);
    q = sqrt(beta * beta + 4.0 * gamma);
    phi = asin(-beta / q);
    yplus_white = exp((von_karman / sqrt(gamma)) * (asin((2.0 * gamma * uplus - beta) / q) - phi)) *
                  yplus0;
    yplus_last = yplus;
    const double ku = von_karman * uplus;
    return yplus - (uplus + yplus_white -
                    yplus0 * (1.0 + ku + 0.5 * ku * ku + (1.0 / 6.0) * (ku * ku * ku)));
  }
};
__device__ __forceinline__ double sign_of(double v) { return (double)((0.0 < v) - (v < 0.0)); }
// mode 0: wallLaw::AdiabaticBCs :31-87; 1: HeatFluxBCs :89-145 (wall_value = q_w);
// 2: IsothermalBCs :147-200 (wall_value = T_w)
__device__ inline void wall_law_solve(const GasDev& g, const double* s, double wall_dist,
                                      const double* n, const double* vel_wall, bool is_lower,
                                      double von_karman, double wall_const, int mode,
                                      double wall_value, WallVars& wv) {
  WallLawDev w;
  w.mode = mode; w.gas = &g; w.p_int = s[4];
  w.von_karman = von_karman; w.wall_dist = wall_dist; w.yplus0 = exp(-von_karman * wall_const);
  w.heat_flux = 0.0; w.cp = g.cp; w.yplus_last = 0.0;
  w.beta = w.gamma = w.q = w.phi = w.yplus_white = w.u_star = w.uplus = 0.0;
  const double vel[3] = {s[1] - vel_wall[0], s[2] - vel_wall[1], s[3] - vel_wall[2]};
  const double vn = dot3(vel, n);
  const double vt[3] = {vel[0] - vn * n[0], vel[1] - vn * n[1], vel[2] - vn * n[2]};
  w.vel_tan = sqrt(dot3(vt, vt));
  const double t = s[4] / (s[0] * g.R);
  w.t_int = t;
  w.recovery = pow(g.prandtl, 1.0 / 3.0);
  if (mode == 0) {          // wall temperature from Crocco-Busemann, adiabatic
    w.set_wall_vars(t + 0.5 * w.recovery * w.vel_tan * w.vel_tan / g.cp);
  } else if (mode == 1) {   // guess: wall temperature equals interior temperature
    w.heat_flux = wall_value;
    w.set_wall_vars(t);
  } else {
    w.set_wall_vars(wall_value);
  }
  {   // FindRoot(func, 1.0e1, 1.0e4, 1.0e-8)
    double x1 = 1.0e1, x2 = 1.0e4;
    double f1 = w.func(x1), f2 = w.func(x2);
    if (!(sign_of(f1) == sign_of(f2) && sign_of(f1) != 0.0)) {
      for (int it = 0; it < 100; ++it) {
        const double x3 = 0.5 * (x1 + x2);
        const double f3 = w.func(x3);
        if (f3 == 0.0) break;
        const double denom = sqrt(fabs(f3 * f3 - f1 * f2));
        if (denom == 0.0) break;
        const double x4 = x3 + (x3 - x1) * (sign_of(f1 - f2) * f3) / denom;
        const double f4 = w.func(x4);
        if (f4 == 0.0) break;
        if (sign_of(f4) != sign_of(f3)) { x1 = x3; f1 = f3; x2 = x4; f2 = f4; }
        else if (sign_of(f4) != sign_of(f1)) { x2 = x4; f2 = f4; }
        else { x1 = x4; f1 = f4; }
        if (fabs(x2 - x1) <= 1.0e-8) break;
      }
    }
  }
  wv.yplus = w.yplus_last;
  wv.heat_flux = w.heat_flux;    // 0, q_w, or the last evaluation's Crocco-Busemann flux
  // CalcTurbVars with EddyVisc, wallLaw.cpp:243-279
  const double dyw = 2.0 * w.yplus_white * w.von_karman * sqrt(w.gamma) / w.q *
                     sqrt(fmax(1.0 - (2.0 * w.gamma * w.uplus - w.beta) *
                                         (2.0 * w.gamma * w.uplus - w.beta) / (w.q * w.q), 0.0));
  const double ku = w.von_karman * w.uplus;
  const double mut_w = fmax(w.mu_w * (1.0 + dyw - w.von_karman * w.yplus0 * (1.0 + ku + 0.5 * ku * ku)) -
                            viscosity(g, t) * g.scaling, 0.0);
  const double wi = 6.0 * w.mu_w / ((g.wilcox ? 0.0708 : 0.075) * w.rho_w * wall_dist * wall_dist) *
                    g.scaling;
  const double wo = w.u_star / (sqrt(0.09) * w.von_karman * wall_dist) * g.scaling;
  wv.sdr = sqrt(wi * wi + wo * wo);
  wv.tke = wv.sdr * mut_w / s[0] * (1.0 / g.scaling);
  wv.density = w.rho_w;
  wv.temperature = w.tw;
  wv.viscosity = w.mu_w;
  wv.turb_eddy_visc = mut_w;
  wv.friction_velocity = w.u_star;
  const double ssm = w.u_star * w.u_star * w.rho_w;
  for (int q = 0; q < 3; ++q) wv.shear[q] = (is_lower ? 1.0 : -1.0) * ssm * vt[q] / w.vel_tan;
}

// nu_w: kinematic viscosity of the wall-adjacent cell (rans viscous walls,
// procBlock.cpp:2814-2820)
__device__ inline bool ghost_state(const GasDev& g, const double* in, int bc,
                                   const double* area_unit, int surf,
                                   const agx_bc_state& d, int layer,
                                   double wall_dist, double* gh,
                                   const NrDev* nr = nullptr, double nu_w = 0.0,
                                   WallVars* wv = nullptr) {
#pragma unroll
  for (int e = 0; e < AGX_NEQ; ++e) gh[e] = in[e];
  const double sgn = (surf % 2 == 1) ? -1.0 : 1.0;
  const double n[3] = {sgn * area_unit[0], sgn * area_unit[1], sgn * area_unit[2]};
  switch (bc) {
    case AGX_BC_SLIPWALL: {
      const double vn = dot3(in + 1, n);
      for (int q = 0; q < 3; ++q) gh[1 + q] = in[1 + q] - 2.0 * n[q] * vn;
      return true;
    }
    case AGX_BC_VISCOUSWALL: {
      for (int q = 0; q < 3; ++q) gh[1 + q] = 2.0 * d.velocity[q] - in[1 + q];
      // ghostStates.cpp:144-260: per thermal wall type the low-Re ghost density, or -- with
      // wall functions -- the wall law first, then the ghost density it implies; k and omega
      // at the wall from the wall law unless y+ < 10 switches the face back to low-Re
      bool low_re = true;
      WallVars loc;
      WallVars& w = wv ? *wv : loc;
      const bool wl = AGX_NEQ > 5 && d.is_wall_law;
      if (d.is_isothermal) {
        double tg = 2.0 * d.wall_temperature - temperature(g, in);
        if (wl) {
          wall_law_solve(g, in, wall_dist, n, d.velocity, surf % 2 == 1, d.von_karman,
                         d.wall_constant, 2, d.wall_temperature, w);
          low_re = w.yplus < 10.0;
          if (!low_re) {
            // the wall law's heat flux with the turbulent conductivity (the eddy viscosity
            // is not zero at the wall), 2 x wall distance as gradient length :161-172
            const double kappa = conductivity(g, w.temperature) +
                                 w.turb_eddy_visc * g.cp / g.turb_prandtl;
            tg = d.wall_temperature - w.heat_flux / kappa * 2.0 * wall_dist;
          }
        }
        gh[0] = gh[4] / (g.R * tg);
      } else if (d.is_heat_flux) {
        // low-Re constant heat flux wall, ghostStates.cpp:228-242: the gradient
        // length is twice the wall distance of the wall-adjacent cell
        const double t = temperature(g, in);
        double tg = t - d.wall_heat_flux / conductivity(g, t) * 2.0 * wall_dist;
        if (wl) {
          wall_law_solve(g, in, wall_dist, n, d.velocity, surf % 2 == 1, d.von_karman,
                         d.wall_constant, 1, d.wall_heat_flux, w);
          low_re = w.yplus < 10.0;
          if (!low_re) tg = 2.0 * w.temperature - t;      // :213-219
        }
        gh[0] = gh[4] / (g.R * tg);
      } else if (wl) {
        wall_law_solve(g, in, wall_dist, n, d.velocity, surf % 2 == 1, d.von_karman,
                       d.wall_constant, 0, 0.0, w);
        low_re = w.yplus < 10.0;
      }
      if (wl && !low_re) {
        gh[AGX_NEQ - 2] = 2.0 * w.tke - in[AGX_NEQ - 2];
        gh[AGX_NEQ - 1] = 2.0 * w.sdr - in[AGX_NEQ - 1];
        if (layer > 1) {
          gh[AGX_NEQ - 2] = layer * gh[AGX_NEQ - 2] - w.tke;
          gh[AGX_NEQ - 1] = layer * gh[AGX_NEQ - 1] - w.sdr;
        }
      }
      if (AGX_NEQ > 5 && low_re) {
        // low-Re wall, ghostStates.cpp:261-279: k = 0 at the face, Menter's wall omega
        // (WallBeta = beta1 = 0.075)
        gh[AGX_NEQ - 2] = -1.0 * in[AGX_NEQ - 2];
        // (WallBeta: beta1 of SST, beta0 of Wilcox 2006)
        const double w_wall = g.scaling * g.scaling * 60.0 * nu_w /
                              (wall_dist * wall_dist * (g.wilcox ? 0.0708 : 0.075));
        gh[AGX_NEQ - 1] = 2.0 * w_wall - in[AGX_NEQ - 1];
        if (layer > 1) gh[AGX_NEQ - 1] = layer * gh[AGX_NEQ - 1] - w_wall;
      }
      return true;
    }
    case AGX_BC_CHARACTERISTIC:
    case AGX_BC_INLET: {
      if (bc == AGX_BC_INLET && d.is_nonreflecting && !nr) return false;
      const double fs[5] = {d.density, d.velocity[0], d.velocity[1], d.velocity[2],
                            d.pressure};
      const double vn = dot3(in + 1, n);
      const double c = sound_speed(g, in);
      const double mach = fabs(vn) / c;
      const bool inflow = vn < 0.0;
      bool extrap = true;
      if (mach >= 1.0 && (inflow || bc == AGX_BC_INLET)) {
        for (int e = 0; e < 5; ++e) gh[e] = fs[e];
        for (int e = 5; e < AGX_NEQ; ++e) gh[e] = 0.0;
        apply_farfield_turb(g, gh, fs + 1, d.turb_intensity, d.eddy_visc_ratio);
        if (bc == AGX_BC_INLET) extrap = false;   // ghostStates.cpp:412-424
      } else if (mach >= 1.0) {
        // supersonic outflow: interior
      } else if (inflow || bc == AGX_BC_INLET) {
        const double rc = in[0] * c;
        const double vd[3] = {fs[1] - in[1], fs[2] - in[2], fs[3] - in[3]};
        gh[4] = 0.5 * (fs[4] + in[4] - rc * dot3(n, vd));
        if (bc == AGX_BC_INLET && d.is_nonreflecting) {
          // LODI terms, ghostStates.cpp:435-462
          const double sigma = 0.25;
          const double rhoN = nr->sn[0], sosN = sound_speed(g, nr->sn), rcN = rhoN * sosN;
          const double dpn = gh[4] - nr->sn[4];
          const double alpha = sigma * sosN / d.length_scale;
          gh[0] = (rhoN + nr->dt * alpha * fs[0] + dpn / (sosN * sosN)) / (1.0 + nr->dt * alpha);
          const double kk = alpha * (1.0 - nr->max_mach * nr->max_mach);
          for (int q = 0; q < 3; ++q)
            gh[1 + q] = (nr->sn[1 + q] + nr->dt * kk * fs[1 + q] - n[q] * dpn / rcN) /
                        (1.0 + nr->dt * kk);
          // (reflecting and nonreflecting alike, ghostStates.cpp:475-479)
          apply_farfield_turb(g, gh, fs + 1, d.turb_intensity, d.eddy_visc_ratio);
        } else {
          const double dp = fs[4] - gh[4];
          gh[0] = fs[0] - dp / (c * c);
          for (int q = 0; q < 3; ++q) gh[1 + q] = fs[1 + q] - n[q] * dp / rc;
          apply_farfield_turb(g, gh, fs + 1, d.turb_intensity, d.eddy_visc_ratio);
        }
      } else {
        const double rc = in[0] * c;
        const double dp = in[4] - fs[4];
        gh[0] = in[0] - dp / (c * c);
        for (int q = 0; q < 3; ++q) gh[1 + q] = in[1 + q] + n[q] * dp / rc;
        gh[4] = fs[4];
      }
      if (extrap) {
        double t[AGX_NEQ];
        extrap_hold(gh, 2.0, in, t);
        for (int e = 0; e < AGX_NEQ; ++e) gh[e] = t[e];
        if (layer > 1) {
          extrap_hold(gh, (double)layer, in, t);
          for (int e = 0; e < AGX_NEQ; ++e) gh[e] = t[e];
          // (characteristic: whatever the flow direction, ghostStates.cpp:381-387; the
          // inlet's deeper layers are only extrapolated, :481-486)
          if (bc == AGX_BC_CHARACTERISTIC)
            apply_farfield_turb(g, gh, fs + 1, d.turb_intensity, d.eddy_visc_ratio);
        }
      }
      return true;
    }
    case AGX_BC_SUPERSONIC_INFLOW:
      gh[0] = d.density; gh[1] = d.velocity[0]; gh[2] = d.velocity[1];
      gh[3] = d.velocity[2]; gh[4] = d.pressure;
      apply_farfield_turb(g, gh, d.velocity, d.turb_intensity, d.eddy_visc_ratio);   // :513-517
      return true;
    case AGX_BC_SUPERSONIC_OUTFLOW:
      if (layer > 1)
        for (int e = 0; e < AGX_NEQ; ++e) gh[e] = layer * gh[e] - in[e];
      return true;
    case AGX_BC_STAGNATION_INLET: {
      const double gm1 = g.gamma - 1.0;
      const double c = sound_speed(g, in);
      const double vn = dot3(in + 1, n);
      const double v2 = dot3(in + 1, in + 1);
      const double rneg = vn - 2.0 * c / gm1;
      const double ct = -vn / sqrt(v2);
      const double c0sq = c * c + 0.5 * gm1 * v2;
      const double k = gm1 * ct * ct + 2.0;
      const double cb = -rneg * gm1 / k *
          (1.0 + ct * sqrt(k * c0sq / (gm1 * rneg * rneg) - 0.5 * gm1));
      const double ratio = cb * cb / c0sq;
      const double tb = d.stagnation_temperature * ratio;
      const double pb = d.stagnation_pressure * pow(ratio, g.gamma / gm1);
      const double vb = sqrt(2.0 / gm1 * (d.stagnation_temperature - tb));
      gh[0] = pb / (g.R * tb);
      gh[1] = vb * d.direction[0];
      gh[2] = vb * d.direction[1];
      gh[3] = vb * d.direction[2];
      gh[4] = pb;
      // farfield turbulence from the ghost velocity, ghostStates.cpp:581-585, :593-598
      apply_farfield_turb(g, gh, gh + 1, d.turb_intensity, d.eddy_visc_ratio);
      double t[AGX_NEQ];
      extrap_hold(gh, 2.0, in, t);
      for (int e = 0; e < AGX_NEQ; ++e) gh[e] = t[e];
      if (layer > 1) {
        extrap_hold(gh, (double)layer, in, t);
        for (int e = 0; e < AGX_NEQ; ++e) gh[e] = t[e];
        apply_farfield_turb(g, gh, gh + 1, d.turb_intensity, d.eddy_visc_ratio);
      }
      return true;
    }
    case AGX_BC_PRESSURE_OUTLET: {
      if (d.is_nonreflecting && !nr) return false;
      const double c = sound_speed(g, in);
      const double rc = in[0] * c;
      gh[4] = d.pressure;
      if (d.is_nonreflecting) {
        // LODI + transverse terms, ghostStates.cpp:614-643
        const double* sn = nr->sn;
        const double dv[3] = {in[1] - sn[1], in[2] - sn[2], in[3] - sn[3]};
        const double sigma = 0.25;
        const double rhoN = sn[0], sosN = sound_speed(g, sn), rcN = rhoN * sosN;
        const double kk = sigma * sosN * (1.0 - nr->max_mach * nr->max_mach) / d.length_scale;
        const double pgn = dot3(nr->pg, n), vnn = dot3(sn + 1, n);
        double tv[3], velT[3], vgt[9], dvn[3] = {0.0, 0.0, 0.0}, sum = 0.0;
        for (int r = 0; r < 3; ++r) {            // tensor::RemoveComponent (rows)
          const double rn = dot3(nr->vg + 3 * r, n);
          for (int q = 0; q < 3; ++q) { vgt[3 * r + q] = nr->vg[3 * r + q] - rn * n[q]; sum += vgt[3 * r + q]; }
        }
        for (int r = 0; r < 3; ++r)              // tensor::LinearCombination
          for (int q = 0; q < 3; ++q) dvn[q] += vgt[3 * r + q] * n[r];
        for (int q = 0; q < 3; ++q) {
          velT[q] = sn[1 + q] - vnn * n[q];
          tv[q] = (nr->pg[q] - pgn * n[q]) - rcN * dvn[q];
        }
        const double dvt = sum - (dvn[0] + dvn[1] + dvn[2]);
        const double trans = -0.5 * (dot3(velT, tv) + g.gamma * sn[4] * dvt);
        gh[4] = (sn[4] + rcN * dot3(dv, n) + nr->dt * kk * d.pressure -
                 nr->dt * nr->avg_mach * trans) / (1.0 + nr->dt * kk);
      }
      const double dp = in[4] - gh[4];
      gh[0] = in[0] - dp / (c * c);
      for (int q = 0; q < 3; ++q) gh[1 + q] = in[1 + q] + n[q] * dp / rc;
      if (dot3(gh + 1, n) / sound_speed(g, gh) >= 1.0)
        for (int e = 0; e < AGX_NEQ; ++e) gh[e] = in[e];
      for (int e = 0; e < AGX_NEQ; ++e) gh[e] = 2.0 * gh[e] - in[e];
      if (layer > 1)
        for (int e = 0; e < AGX_NEQ; ++e) gh[e] = layer * gh[e] - in[e];
      return true;
    }
    case AGX_BC_INTERBLOCK:
    case AGX_BC_PERIODIC:
      return true;
  }
  return false;
}

}  // namespace agx
