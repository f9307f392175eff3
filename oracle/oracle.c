/* oracle.c -- TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 * Plain-C scalar restatement of AITHER's hot path for the single-species,
 * calorically-perfect, laminar case (nEq = 5).  Data is kept in the
 * reference's own AoS layout and the arithmetic follows the reference
 * expression by expression (operand order included) so that differences with
 * the reference stay at round-off level.  Every function cites the reference
 * file:line it restates (paths relative to the reference root).
 *
 * Build: gcc -O2 -ffp-contract=off -fPIC -shared (oracle/Makefile).
 */
#include "oracle.h"
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdarg.h>
#include <float.h>

/* equations per cell: 5 (euler / navierStokes) or 7 (rans: + k, omega).  One value
 * for the whole library (test infrastructure, single-threaded): ora_config_set
 * refuses a second live context with another count.  NEQM sizes local arrays, NF is
 * the flow block of the block-matrix solvers. */
/* blocks smaller than this stay on one thread (the parity tests' blocks: a parallel region
 * per hyperplane costs more than it saves, and far more on an oversubscribed host) */
static long g_omp_min_cells = 40000L;   /* ORA_OMP_MIN_CELLS (read by ora_ctx_create) */
#define OMP_MIN_CELLS g_omp_min_cells
static int g_neq = 5, g_live_cfg = 0;
#define NEQ g_neq
#define NEQM 7
#define NF 5
#define TURB_MIN 1.0e-20
#define EPS 1.0e-30                /* include/macros.hpp.in:20 */
#define WALL_DIST_NEG_TOL -1.0e-10 /* include/macros.hpp.in:23 */
#define MAXBLK 64
#define MAXCONN 256

static char g_err[512] = "";
static int fail(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof g_err, fmt, ap);
  va_end(ap);
  return 1;
}

/* ------------------------------------------------------------------------ */
/* wallVars wallData.hpp:33-62 (what the solver reads of it) */
typedef struct wall_vars_s {
  double yplus, heat_flux, density, temperature, viscosity, turb_eddy_visc, friction_velocity,
         shear[3], tke, sdr;
} wall_vars;
typedef struct {
  int ni, nj, nk, ng, parent, gpos;
  int ci, cj, ck;        /* ghost-padded cell dims */
  double *state;         /* state_        [cells_g][NEQM]        */
  double *fa[3];         /* fAreaI/J/K_   [faces_g][4]          */
  double *vol, *center;  /* vol_ [cells_g], center_ [cells_g][3] */
  double *wid[3];        /* cellWidthI/J/K_ [cells_g]           */
  double *wdist;         /* wallDist_ [cells_g]                 */
  double *temp, *visc;   /* temperature_, viscosity_ [cells_g]  */
  double *velgrad;       /* velocityGrad_ [cells_g][9]          */
  double *grad18;        /* cell gradients of the last residual [cells][18], blocks with
                            nonreflecting surfaces only (pressureGrad_, velocityGrad_ fed
                            to the LODI terms, procBlock.cpp:2503-2505) */
  double *resid;         /* residual_ [cells][NEQM]              */
  double *specrad;       /* specRadius_ (flow part) [cells]     */
  double *dt;            /* dt_ [cells]                         */
  double *consn, *consnm1; /* consVarsN_, consVarsNm1_ [cells][NEQM] */
  double *x, *xold;      /* linearSolver x_ [cells_g][NEQM]      */
  /* multigrid (null on a level without): mgForcing_ [cells][NEQM], the matrix residual of
   * the last Relax [cells][NEQM], coarseDu [cells_g][NEQM] */
  double *forcing, *mres, *xsave;
  double *a, *ainv;      /* linearSolver a_, aInv_ (scalar) [cells] */
  double *am, *aminv;    /* block-matrix solvers: 5 x 5 per cell, row major (matMultiArray3d) */
  /* rans: turbulence part of specRadius_ / a_ / aInv_ (uncoupledScalar) [cells];
   * (eddyViscosity_, f1_, f2_) interleaved [cells_g][3]; tkeGrad_, omegaGrad_ [cells][3] */
  double *specrad_t, *a_t, *ainv_t, *turb3, *kgrad, *wgrad;
  double *am_t, *aminv_t; /* block-matrix solvers: diagonal of the turbulence block [cells][2] */
  /* wallData_ of the wall-law surfaces: per surface the offset of its faces in wallv
   * (-1: not a wall-law surface), wallVars of every face */
  long *wall_off;
  struct wall_vars_s *wallv;
  int nsurf;
  agx_bc_surface *surf;
  int nsurf_i, nsurf_j, nsurf_k;
  long ncell, ncell_g;
} blk_t;

typedef struct {
  agx_connection c;
  long n[2];            /* number of cells inserted into side s */
  long *dst[2], *src[2];/* dst[s]: cell index in block of side s (ghost);
                           src[s]: cell index in partner block           */
} conn_t;

struct ora_ctx {
  int rank;
  agx_config cfg;
  int have_cfg;
  int nblk;
  blk_t blk[MAXBLK];
  int nconn;
  conn_t conn[MAXCONN];
  /* derived gas constants */
  double gamma, cp, cv, mu_ref, k_nondim, scaling, prandtl;
  int have_time_n;   /* StoreOldSolution has run (consVarsN_ non-empty) */
  double mres_opsq;  /* sum of the squared OPERANDS of the last matrix residual (tests) */
  double mres_sumsq; /* ... and of the squared residual itself */
  agx_exchange ex;   /* multi-rank transport (host buffers) */
  int have_ex;
};

/* ------------------------------------------------------------------------ */
/* index helpers: signed reference-style indices (ghosts negative)           */
static inline long CI(const blk_t *b, int i, int j, int k) {
  return ((long)(k + b->ng) * b->cj + (j + b->ng)) * b->ci + (i + b->ng);
}
static inline long PI(const blk_t *b, int i, int j, int k) {
  return ((long)k * b->nj + j) * b->ni + i;
}
/* face arrays: dimension d has one more entry */
static inline long FI(const blk_t *b, int d, int i, int j, int k) {
  const int di = b->ci + (d == 0), dj = b->cj + (d == 1);
  return ((long)(k + b->ng) * dj + (j + b->ng)) * di + (i + b->ng);
}
static inline int is_physical(const blk_t *b, int i, int j, int k) {
  return i >= 0 && i < b->ni && j >= 0 && j < b->nj && k >= 0 && k < b->nk;
}
static inline int at_corner(const blk_t *b, int i, int j, int k) {
  return (i < 0 || i >= b->ni) && (j < 0 || j >= b->nj) &&
         (k < 0 || k >= b->nk);
}

/* ------------------------------------------------------------------------ */
/* thermodynamics: idealGas (src/eos.cpp), caloricallyPerfect
 * (include/thermodynamic.hpp:85-121, src/thermodynamic.cpp:66-114),
 * single species => mass fraction = rho/rho = 1.0 exactly                   */
static inline double dot3(const double *a, const double *b) {
  /* vector3d::DotProd = std::inner_product (vector3d.hpp:302-304) */
  return ((0.0 + a[0] * b[0]) + a[1] * b[1]) + a[2] * b[2];
}
static inline double mag3(const double *a) { return sqrt(dot3(a, a)); }

static inline double temperature(const ora_ctx *c, const double *s) {
  /* idealGas::Temperature eos.cpp:100-109 */
  const double rhoR = 0.0 + s[0] * c->cfg.gas.gas_constant;
  return s[4] / rhoR;
}
static inline double sos(const ora_ctx *c, const double *s) {
  /* SpeedOfSound arrayView.hpp:383-391 */
  return sqrt(c->gamma * s[4] / s[0]);
}
static inline double spec_energy(const ora_ctx *c, double t) {
  /* thermodynamic::SpecEnergy thermodynamic.cpp:83-92 */
  return 0.0 + 1.0 * (c->cfg.gas.heat_of_formation +
                      c->cfg.gas.gas_constant * c->cfg.gas.n * t);
}
static inline double energy(const ora_ctx *c, const double *s) {
  /* InternalEnergy arrayView.hpp:434-443; idealGas::Energy eos.cpp:70-72 */
  const double t = temperature(c, s);
  const double vel = mag3(s + 1);
  return spec_energy(c, t) + 0.5 * vel * vel;
}
static inline double enthalpy(const ora_ctx *c, const double *s) {
  /* EnthalpyFunc arrayView.hpp:401-409; idealGas::Enthalpy eos.cpp:80-84 */
  const double t = temperature(c, s);
  const double vel = mag3(s + 1);
  const double h = 0.0 + 1.0 * (c->cfg.gas.heat_of_formation +
                                c->cfg.gas.gas_constant *
                                    (c->cfg.gas.n + 1.0) * t);
  return h + 0.5 * vel * vel;
}
static void prim_to_cons(const ora_ctx *c, const double *s, double *u) {
  /* PrimToCons primitive.hpp:183-201 */
  const double rho = s[0];
  u[0] = s[0];
  u[1] = rho * s[1];
  u[2] = rho * s[2];
  u[3] = rho * s[3];
  u[4] = rho * energy(c, s);
  for (int e = NF; e < NEQ; ++e) u[e] = rho * s[e];     /* rho k, rho omega */
}
static void cons_to_prim(const ora_ctx *c, const double *u, double *s) {
  /* primitive::primitive(cons, phys) primitive.hpp:152-178;
   * idealGas::PressFromEnergy eos.cpp:40-52;
   * caloricallyPerfect::TemperatureFromSpecEnergy thermodynamic.cpp:108-114 */
  const double rho = u[0];
  s[0] = u[0];
  s[1] = u[1] / rho;
  s[2] = u[2] / rho;
  s[3] = u[3] / rho;
  const double en = u[4] / rho;
  const double vel = mag3(s + 1);
  const double spec = en - 0.5 * vel * vel;
  const double mf = s[0] / (0.0 + s[0]);
  const double hf = 0.0 + c->cfg.gas.heat_of_formation * mf;
  const double cv = 0.0 + mf * (c->cfg.gas.gas_constant * c->cfg.gas.n);
  const double t = (spec - hf) / cv;
  s[4] = 0.0 + s[0] * c->cfg.gas.gas_constant * t;
  /* turbulence variables, then primitive::LimitTurb primitive.cpp:100-106 with
   * turbModel::TkeMin / OmegaMin = 1e-20 (turbulence.hpp:72-73) */
  for (int e = NF; e < NEQ; ++e) {
    s[e] = u[e] / rho;
    if (!(s[e] > TURB_MIN)) s[e] = TURB_MIN;
  }
}
static void update_prim_with_cons(const ora_ctx *c, const double *s,
                                  const double *du, double *out) {
  /* UpdatePrimWithCons primitive.hpp:206-231 (single species: the mass
   * fraction clip/renormalise is the identity for positive density) */
  double u[NEQM];
  prim_to_cons(c, s, u);
  for (int e = 0; e < NEQ; ++e) u[e] = u[e] + du[e];
  const double rho = u[0];
  double mf = u[0] / rho;
  mf = mf > 0.0 ? mf : 0.0;
  const double total = 0.0 + mf;
  mf /= total;
  u[0] = rho * mf;
  cons_to_prim(c, u, out);
}
static inline double viscosity(const ora_ctx *c, double t) {
  /* sutherland::SpeciesViscosity transport.cpp:114-122 */
  const double temp = t * c->cfg.gas.t_ref;
  const double mu = (c->cfg.gas.visc_c1 * pow(temp, 1.5)) /
                    (temp + c->cfg.gas.visc_s);
  return mu / c->mu_ref;
}
static inline double conductivity(const ora_ctx *c, double t) {
  /* sutherland::SpeciesConductivity transport.cpp:124-132 */
  const double temp = t * c->cfg.gas.t_ref;
  const double k = (c->cfg.gas.cond_c1 * pow(temp, 1.5)) /
                   (temp + c->cfg.gas.cond_s);
  return k / c->k_nondim;
}

/* ------------------------------------------------------------------------ */
/* limiters src/limiter.cpp:24-54                                            */
static inline double lim_apply(int lim, double r) {
  if (lim == AGX_LIMITER_VANALBADA) {
    const double r2 = r * r;
    const double l = (r + r2) / (1.0 + r2);
    return l > 0.0 ? l : 0.0;          /* std::max(0.0, limiter) */
  } else if (lim == AGX_LIMITER_MINMOD) {
    const double m = 1.0 < r ? 1.0 : r; /* std::min(1.0, r) */
    return 0.0 < m ? m : 0.0;           /* std::max(0.0, .) */
  }
  return 1.0;
}

/* FaceReconMUSCL include/reconstruction.hpp:110-154 */
static void recon_muscl(const ora_ctx *c, const double *uw2, const double *uw1,
                        const double *dw1, double w2, double w, double wd,
                        double *out) {
  const double kappa = c->cfg.kappa;
  const int lim = c->cfg.limiter;
  const double dPlus = (w + w) / (w + wd);
  const double dMinus = (w + w) / (w + w2);
  for (int e = 0; e < NEQ; ++e) {
    const double r = (EPS + (dw1[e] - uw1[e]) * dPlus) /
                     (EPS + (uw1[e] - uw2[e]) * dMinus);
    double limiter, inv;
    if (lim == AGX_LIMITER_NONE) {
      limiter = 1.0;
      inv = 1.0;
    } else {
      limiter = lim_apply(lim, r);
      inv = lim_apply(lim, 1.0 / r);
    }
    out[e] = uw1[e] + 0.25 * ((uw1[e] - uw2[e]) * dMinus) *
                          ((1.0 - kappa) * limiter + (1.0 + kappa) * r * inv);
  }
}

/* StencilWidth include/utility.hpp:100-112 (std::accumulate from 0.0) */
static double stencil_width(const double *w, int start, int end) {
  double width = 0.0;
  if (end > start) {
    for (int q = start; q < end; ++q) width = width + w[q];
  } else if (start > end) {
    double acc = 0.0;
    for (int q = end; q < start; ++q) acc = acc + w[q];
    width = -1.0 * acc;
  }
  return width;
}
/* LagrangeCoeff src/utility.cpp:449-483 */
static void lagrange_coeff(const double *w, int degree, int rr, int ii,
                           double *coeffs) {
  for (int jj = 0; jj <= degree; ++jj) {
    coeffs[jj] = 0.0;
    for (int mm = jj + 1; mm <= degree + 1; ++mm) {
      double numer = 0.0, denom = 1.0;
      for (int ll = 0; ll <= degree + 1; ++ll) {
        if (ll != mm) {
          double numProd = 1.0;
          for (int qq = 0; qq <= degree + 1; ++qq) {
            if (qq != mm && qq != ll)
              numProd *= stencil_width(w, ii - rr + qq, ii + 1);
          }
          numer += numProd;
          denom *= stencil_width(w, ii - rr + ll, ii - rr + mm);
        }
      }
      coeffs[jj] += numer / denom;
    }
    coeffs[jj] *= w[ii - rr + jj];
  }
}
/* Derivative2nd include/utility.hpp:114-120 */
static inline double deriv2nd(double x0, double x1, double x2, double y0,
                              double y1, double y2) {
  const double fwd = (y2 - y1) / (0.5 * (x2 + x1));
  const double bck = (y1 - y0) / (0.5 * (x1 + x0));
  return (fwd - bck) / (0.25 * (x2 + x0) + 0.5 * x1);
}
/* BetaIntegral reconstruction.hpp:158-183 */
static inline double beta_int1(double d1, double d2, double dx, double x) {
  return (d1 * d1 * x + d1 * d2 * x * x + d2 * d2 * pow(x, 3.0) / 3.0) * dx +
         d2 * d2 * x * pow(dx, 3.0);
}
static inline double beta_int(double d1, double d2, double dx, double xl,
                              double xh) {
  return beta_int1(d1, d2, dx, xh) - beta_int1(d1, d2, dx, xl);
}
/* Beta0/1/2 reconstruction.hpp:186-240 */
static inline double beta0(double x0, double x1, double x2, double y0,
                           double y1, double y2) {
  const double d2 = deriv2nd(x0, x1, x2, y0, y1, y2);
  const double d1 = (y2 - y1) / (0.5 * (x2 + x1)) + 0.5 * x2 * d2;
  return beta_int(d1, d2, x2, -0.5 * x2, 0.5 * x2);
}
static inline double beta1(double x0, double x1, double x2, double y0,
                           double y1, double y2) {
  const double d2 = deriv2nd(x0, x1, x2, y0, y1, y2);
  const double d1 = (y2 - y1) / (0.5 * (x2 + x1)) - 0.5 * x1 * d2;
  return beta_int(d1, d2, x1, -0.5 * x1, 0.5 * x1);
}
static inline double beta2(double x0, double x1, double x2, double y0,
                           double y1, double y2) {
  const double d2 = deriv2nd(x0, x1, x2, y0, y1, y2);
  const double d1 = (y1 - y0) / (0.5 * (x1 + x0)) - 0.5 * x0 * d2;
  return beta_int(d1, d2, x0, -0.5 * x0, 0.5 * x0);
}
/* FaceReconWENO reconstruction.hpp:244-310 */
static void recon_weno(const double *u3, const double *u2, const double *u1,
                       const double *d1, const double *d2, double w3,
                       double w2, double w1, double wd1, double wd2,
                       int is_z, double *out) {
  const double cw[5] = {w3, w2, w1, wd1, wd2};
  double c0[3], c1[3], c2[3], cf[5];
  lagrange_coeff(cw, 2, 2, 2, c0);
  lagrange_coeff(cw, 2, 1, 2, c1);
  lagrange_coeff(cw, 2, 0, 2, c2);
  lagrange_coeff(cw, 4, 2, 2, cf);
  const double lw0 = cf[0] / c0[0];
  const double lw1 = cf[4] / c2[2];
  const double lw2 = 1.0 - lw0 - lw1;
  for (int e = 0; e < NEQ; ++e) {
    const double s0 = c0[0] * u3[e] + c0[1] * u2[e] + c0[2] * u1[e];
    const double s1 = c1[0] * u2[e] + c1[1] * u1[e] + c1[2] * d1[e];
    const double s2 = c2[0] * u1[e] + c2[1] * d1[e] + c2[2] * d2[e];
    const double b0 = beta0(w3, w2, w1, u3[e], u2[e], u1[e]);
    const double b1 = beta1(w2, w1, wd1, u2[e], u1[e], d1[e]);
    const double b2 = beta2(w1, wd1, wd2, u1[e], d1[e], d2[e]);
    double n0, n1, n2;
    if (is_z) {
      const double tau5 = fabs(b0 - b2);
      const double eps = 1.0e-40;
      const double q0 = tau5 / (eps + b0), q1 = tau5 / (eps + b1),
                   q2 = tau5 / (eps + b2);
      n0 = lw0 * (1.0 + q0 * q0);
      n1 = lw1 * (1.0 + q1 * q1);
      n2 = lw2 * (1.0 + q2 * q2);
    } else {
      const double eps = 1.0e-6;
      const double e0 = eps + b0, e1 = eps + b1, e2 = eps + b2;
      n0 = lw0 / (e0 * e0);
      n1 = lw1 / (e1 * e1);
      n2 = lw2 / (e2 * e2);
    }
    const double sum = n0 + n1 + n2;
    n0 /= sum;
    n1 /= sum;
    n2 /= sum;
    out[e] = n0 * s0 + n1 * s1 + n2 * s2;
  }
}

/* ------------------------------------------------------------------------ */
/* inviscidFlux::ConstructFromPrim inviscidFlux.hpp:129-160 */
static void phys_flux(const ora_ctx *c, const double *s, const double *n,
                      double *f) {
  const double velNorm = dot3(s + 1, n);
  const double rho = s[0];
  f[0] = s[0] * velNorm;
  f[1] = rho * velNorm * s[1] + s[4] * n[0];
  f[2] = rho * velNorm * s[2] + s[4] * n[1];
  f[3] = rho * velNorm * s[3] + s[4] * n[2];
  f[4] = rho * velNorm * enthalpy(c, s);
  for (int e = NF; e < NEQ; ++e) f[e] = rho * velNorm * s[e];
}

/* RoeFlux inviscidFlux.hpp:260-382, RoeAveragedState primitive.hpp:245-280 */
static void roe_flux(const ora_ctx *c, const double *l, const double *r,
                     const double *n, double *flux) {
  double roe[NEQM];
  const double denRatio = sqrt(r[0] / l[0]);
  roe[0] = l[0] * denRatio;
  roe[1] = (l[1] + denRatio * r[1]) / (1.0 + denRatio);
  roe[2] = (l[2] + denRatio * r[2]) / (1.0 + denRatio);
  roe[3] = (l[3] + denRatio * r[3]) / (1.0 + denRatio);
  roe[4] = (l[4] + denRatio * r[4]) / (1.0 + denRatio);
  for (int e = NF; e < NEQ; ++e) roe[e] = (l[e] + denRatio * r[e]) / (1.0 + denRatio);
  const double hR = enthalpy(c, roe);
  const double aR = sos(c, roe);
  const double rhoR = roe[0];
  const double velNormR = dot3(roe + 1, n);
  const double mfR = roe[0] / roe[0];
  double delta[NEQM];
  for (int e = 0; e < NEQ; ++e) delta[e] = r[e] - l[e];
  const double normVelDiff = dot3(delta + 1, n);
  double diss[NEQM] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
  const double entropyFix = 0.1;
  /* left moving acoustic wave */
  double waveSpeed = fabs(velNormR - aR);
  if (waveSpeed < entropyFix)
    waveSpeed = 0.5 * (waveSpeed * waveSpeed / entropyFix + entropyFix);
  double waveStrength = (delta[4] - rhoR * aR * normVelDiff) / (2.0 * aR * aR);
  double wss = waveSpeed * waveStrength;
  diss[0] += wss * mfR;
  diss[1] += wss * (roe[1] - aR * n[0]);
  diss[2] += wss * (roe[2] - aR * n[1]);
  diss[3] += wss * (roe[3] - aR * n[2]);
  diss[4] += wss * (hR - aR * velNormR);
  for (int e = NF; e < NEQ; ++e) diss[e] += wss * roe[e];
  /* entropy wave */
  waveSpeed = fabs(velNormR);
  waveStrength = -delta[4] / (aR * aR);
  wss = waveSpeed * waveStrength;
  diss[0] += wss * mfR + waveSpeed * delta[0];
  waveStrength = delta[0] - delta[4] / (aR * aR);
  wss = waveSpeed * waveStrength;
  diss[1] += wss * roe[1];
  diss[2] += wss * roe[2];
  diss[3] += wss * roe[3];
  diss[4] += wss * 0.5 * dot3(roe + 1, roe + 1);
  /* shear wave */
  waveStrength = rhoR;
  wss = waveSpeed * waveStrength;
  diss[1] += wss * (delta[1] - normVelDiff * n[0]);
  diss[2] += wss * (delta[2] - normVelDiff * n[1]);
  diss[3] += wss * (delta[3] - normVelDiff * n[2]);
  diss[4] += wss * (dot3(roe + 1, delta + 1) - velNormR * normVelDiff);
  /* right moving acoustic wave */
  waveSpeed = fabs(velNormR + aR);
  if (waveSpeed < entropyFix)
    waveSpeed = 0.5 * (waveSpeed * waveSpeed / entropyFix + entropyFix);
  waveStrength = (delta[4] + rhoR * aR * normVelDiff) / (2.0 * aR * aR);
  wss = waveSpeed * waveStrength;
  diss[0] += wss * mfR;
  diss[1] += wss * (roe[1] + aR * n[0]);
  diss[2] += wss * (roe[2] + aR * n[1]);
  diss[3] += wss * (roe[3] + aR * n[2]);
  diss[4] += wss * (hR + aR * velNormR);
  for (int e = NF; e < NEQ; ++e) diss[e] += wss * roe[e];
  /* turbulence waves, inviscidFlux.hpp:363-372 */
  waveSpeed = fabs(velNormR);
  for (int e = NF; e < NEQ; ++e) {
    waveStrength = rhoR * delta[e] + roe[e] * delta[0] - delta[4] * roe[e] / (aR * aR);
    wss = waveSpeed * waveStrength;
    diss[e] += wss * 1.0;
  }
  double fl[NEQM], fr[NEQM];
  phys_flux(c, l, n, fl);
  phys_flux(c, r, n, fr);
  /* inviscidFlux::RoeFlux src/inviscidFlux.cpp:26-32 */
  for (int e = 0; e < NEQ; ++e) {
    fl[e] += fr[e] - diss[e];
    flux[e] = fl[e] * 0.5;
  }
}

static inline int sign_d(double v) { return (0.0 < v) - (v < 0.0); }

/* AUSMFlux (AUSMPW+) inviscidFlux.hpp:396-481 and member :162-209 */
static void ausm_flux(const ora_ctx *c, const double *l, const double *r,
                      const double *n, double *f) {
  const double velNormL = dot3(l + 1, n);
  const double velNormR = dot3(r + 1, n);
  const double sosL = sos(c, l), sosR = sos(c, r);
  const double sosStar = sqrt(sosL * sosR);
  const double vel = 0.5 * (velNormL + velNormR);
  double s = sosStar;
  if (vel < 0.0) {
    s = sosStar * sosStar / (velNormR > sosStar ? velNormR : sosStar);
  } else if (vel > 0.0) {
    s = sosStar * sosStar / (velNormL > sosStar ? velNormL : sosStar);
  }
  const double ml = velNormL / s, mr = velNormR / s;
  const double mPlusL = fabs(ml) <= 1.0 ? 0.25 * pow(ml + 1.0, 2.0)
                                        : 0.5 * (ml + fabs(ml));
  const double mMinusR = fabs(mr) <= 1.0 ? -0.25 * pow(mr - 1.0, 2.0)
                                         : 0.5 * (mr - fabs(mr));
  const double pPlus = fabs(ml) <= 1.0
                           ? 0.25 * pow(ml + 1.0, 2.0) * (2.0 - ml)
                           : 0.5 * (1.0 + sign_d(ml));
  const double pMinus = fabs(mr) <= 1.0
                            ? 0.25 * pow(mr - 1.0, 2.0) * (2.0 + mr)
                            : 0.5 * (1.0 - sign_d(mr));
  const double ps = pPlus * l[4] + pMinus * r[4];
  const double pr1 = l[4] / r[4], pr2 = r[4] / l[4];
  const double w = 1.0 - pow(pr1 < pr2 ? pr1 : pr2, 3.0); /* std::min(a,b) */
  const double fl = fabs(ml) < 1.0 ? l[4] / ps - 1.0 : 0.0;
  const double fr = fabs(mr) < 1.0 ? r[4] / ps - 1.0 : 0.0;
  const double mavg = mPlusL + mMinusR;
  const double mPlusLBar =
      mavg >= 0.0 ? mPlusL + mMinusR * ((1.0 - w) * (1.0 + fr) - fl)
                  : mPlusL * w * (1.0 + fl);
  const double mMinusRBar =
      mavg >= 0.0 ? mMinusR * w * (1.0 + fr)
                  : mMinusR + mPlusL * ((1.0 - w) * (1.0 + fl) - fr);
  const double vl = mPlusLBar * s;
  f[0] = l[0] * vl;
  const double rhoL = l[0];
  f[1] = rhoL * vl * l[1] + pPlus * l[4] * n[0];
  f[2] = rhoL * vl * l[2] + pPlus * l[4] * n[1];
  f[3] = rhoL * vl * l[3] + pPlus * l[4] * n[2];
  f[4] = rhoL * vl * enthalpy(c, l);
  for (int e = NF; e < NEQ; ++e) f[e] = rhoL * vl * l[e];
  const double vr = mMinusRBar * s;
  f[0] += r[0] * vr;
  const double rhoR = r[0];
  f[1] += rhoR * vr * r[1] + pMinus * r[4] * n[0];
  f[2] += rhoR * vr * r[2] + pMinus * r[4] * n[1];
  f[3] += rhoR * vr * r[3] + pMinus * r[4] * n[2];
  f[4] += rhoR * vr * enthalpy(c, r);
  for (int e = NF; e < NEQ; ++e) f[e] += rhoR * vr * r[e];
}

/* InvCellSpectralRadius spectralRadius.hpp:44-64 */
static double inv_cell_spec_rad(const ora_ctx *c, const double *s,
                                const double *al, const double *au) {
  double v[3] = {0.5 * (al[0] + au[0]), 0.5 * (al[1] + au[1]),
                 0.5 * (al[2] + au[2])};
  const double m = mag3(v);
  double nv[3] = {v[0] / m, v[1] / m, v[2] / m};
  const double fMag = 0.5 * (al[3] + au[3]);
  return (fabs(dot3(s + 1, nv)) + sos(c, s)) * fMag;
}
static double turb_prandtl(const ora_ctx *c);
/* ViscCellSpectralRadius spectralRadius.hpp:94-124 */
static double visc_term_t(const ora_ctx *c, double mu, double mut) {
  return c->scaling * (mu / c->prandtl + mut / turb_prandtl(c));
}
static double visc_cell_spec_rad(const ora_ctx *c, const double *s,
                                 const double *al, const double *au,
                                 double vol, double mu, double mut) {
  const double fMag = 0.5 * (al[3] + au[3]);
  const double a = 4.0 / (3.0 * s[0]);
  const double b = c->gamma / s[0];
  const double maxTerm = a > b ? a : b;        /* max(a, b) */
  return maxTerm * visc_term_t(c, mu, mut) * fMag * fMag / vol;
}

/* ------------------------------------------------------------------------ */
/* boundary conditions                                                       */
/* boundaryConditions::GetBCSurface boundaryConditions.cpp:109-170 */
static const agx_bc_surface *get_bc_surface(const blk_t *b, int i, int j,
                                            int k, int surf) {
  int s0, s1;
  if (surf <= 2) {
    s0 = 0; s1 = b->nsurf_i;
    for (int n = s0; n < s1; ++n) {
      const agx_bc_surface *q = &b->surf[n];
      if (i >= q->imin && i <= q->imax && j >= q->jmin && j < q->jmax &&
          k >= q->kmin && k < q->kmax) return q;
    }
  } else if (surf <= 4) {
    s0 = b->nsurf_i; s1 = s0 + b->nsurf_j;
    for (int n = s0; n < s1; ++n) {
      const agx_bc_surface *q = &b->surf[n];
      if (i >= q->imin && i < q->imax && j >= q->jmin && j <= q->jmax &&
          k >= q->kmin && k < q->kmax) return q;
    }
  } else {
    s0 = b->nsurf_i + b->nsurf_j; s1 = s0 + b->nsurf_k;
    for (int n = s0; n < s1; ++n) {
      const agx_bc_surface *q = &b->surf[n];
      if (i >= q->imin && i < q->imax && j >= q->jmin && j < q->jmax &&
          k >= q->kmin && k <= q->kmax) return q;
    }
  }
  return NULL;
}
static int surf_type(const agx_bc_surface *q) {
  /* boundarySurface::SurfaceType boundaryConditions.cpp:2424-2456 */
  if (q->imin == q->imax) return q->imax == 0 ? 1 : 2;
  if (q->jmin == q->jmax) return q->jmax == 0 ? 3 : 4;
  return q->kmax == 0 ? 5 : 6;
}
static int bc_is_connection(const blk_t *b, int i, int j, int k, int surf) {
  const agx_bc_surface *q = get_bc_surface(b, i, j, k, surf);
  return q && (q->bc_type == AGX_BC_INTERBLOCK || q->bc_type == AGX_BC_PERIODIC);
}

/* ExtrapolateHoldMixture ghostStates.cpp:691-708 */
static void extrap_hold(const double *bnd, double factor, const double *in,
                        double *out) {
  const double bndRho = bnd[0];
  const double bndMf = bnd[0] / bnd[0];
  const double ghostRho = factor * bndRho - in[0];
  if (ghostRho <= 0.0) {
    for (int e = 0; e < NEQ; ++e) out[e] = bnd[e];
    return;
  }
  double g[NEQM];
  for (int e = 0; e < NEQ; ++e) g[e] = factor * bnd[e] - in[e];
  const double v = ghostRho * bndMf;
  g[0] = v > 0.0 ? v : 0.0;
  for (int e = 0; e < NEQ; ++e) out[e] = g[e];
}

/* GetGhostState ghostStates.cpp:62-689 (laminar, low-Re walls, reflecting
 * inlet/outlet) */
static int is_wilcox_fwd(const ora_ctx *c) { return c->cfg.turbulence_model == AGX_TURB_KW_WILCOX2006; }
static double turb_prandtl_fwd(const ora_ctx *c) { return is_wilcox_fwd(c) ? 8.0 / 9.0 : 0.9; }
/* ---- wall functions: wallLaw::AdiabaticBCs wallLaw.cpp:30-77 with its helpers
 * (:182-289) and FindRoot (Ridder, utility.hpp:130-184).  The function whose root is
 * sought has side effects: what is kept afterwards belongs to its LAST evaluation. */
typedef struct {
  const ora_ctx *c;
  const double *state;
  double vonKarmen, wallDist, yplus0, beta, gamma, q, phi, yplusWhite, uStar, uplus, tW, rhoW,
         muW, mutW, kW, recoveryFactor, velTanMag, heatFlux, yplus_last;
  int mode;        /* 0 adiabatic, 1 constant heat flux, 2 isothermal (wallLaw.cpp:31, :89, :147) */
  double tInt;     /* temperature of the interior state */
} wall_law;
/* wallLaw::SetWallVars wallLaw.cpp:239-246 */
static void wl_set_wall_vars(wall_law *w, double tW) {
  const ora_ctx *c = w->c;
  w->tW = tW;
  w->rhoW = w->state[4] / ((0.0 + 1.0 * c->cfg.gas.gas_constant) * tW);
  w->muW = viscosity(c, tW) * c->scaling;
  w->kW = conductivity(c, tW) * c->scaling;
}
static double sign_of(double v) { return (double)((0.0 < v) - (v < 0.0)); }
static double wl_func(wall_law *w, double yplus) {
  const ora_ctx *c = w->c;
  /* CalcVelocities, UpdateGamma, UpdateConstants, CalcYplusWhite, CalcYplusRoot */
  w->uplus = (w->wallDist * w->rhoW * w->velTanMag) / (w->muW * yplus);
  w->uStar = w->velTanMag / w->uplus;
  if (w->mode == 1) {
    /* HeatFluxBCs :113-124: wall temperature from Crocco-Busemann with the wall properties
     * of the PREVIOUS evaluation (CalcWallTemperature :231-237), then SetWallVars */
    const double tNew = w->tInt + w->recoveryFactor * w->uStar * w->uStar * w->uplus * w->uplus /
                                      (2.0 * c->cp + w->heatFlux * w->muW / (w->rhoW * w->kW * w->uStar));
    wl_set_wall_vars(w, tNew);
  }
  w->gamma = w->recoveryFactor * w->uStar * w->uStar / (2.0 * c->cp * w->tW);
  if (w->mode == 2) {
    /* IsothermalBCs :170-172, CalcHeatFlux :223-229 */
    const double tmp = (w->tInt / w->tW - 1.0 + w->gamma * w->uplus * w->uplus) / w->uplus;
    w->heatFlux = tmp * (w->rhoW * w->tW * w->kW * w->uStar) / w->muW;
  }
  w->beta = w->heatFlux * w->muW / (w->rhoW * w->tW * w->kW * w->uStar);
  w->q = sqrt(w->beta * w->beta + 4.0 * w->gamma);
  w->phi = asin(-w->beta / w->q);
  w->yplusWhite = exp((w->vonKarmen / sqrt(w->gamma)) *
                      (asin((2.0 * w->gamma * w->uplus - w->beta) / w->q) - w->phi)) * w->yplus0;
  w->yplus_last = yplus;
  const double ku = w->vonKarmen * w->uplus;
  return yplus - (w->uplus + w->yplusWhite -
                  w->yplus0 * (1.0 + ku + 0.5 * ku * ku + (1.0 / 6.0) * pow(ku, 3.0)));
}
static void wl_find_root(wall_law *w, double x1, double x2, double tol) {
  double f1 = wl_func(w, x1), f2 = wl_func(w, x2);
  if (sign_of(f1) == sign_of(f2) && sign_of(f1) != 0.0) return;   /* (reference: message, midpoint) */
  for (int ii = 0; ii < 100; ++ii) {
    const double x3 = 0.5 * (x1 + x2);
    const double f3 = wl_func(w, x3);
    if (f3 == 0.0) return;
    const double denom = sqrt(fabs(f3 * f3 - f1 * f2));
    if (denom == 0.0) return;
    const double fac = sign_of(f1 - f2);
    const double x4 = x3 + (x3 - x1) * (fac * f3) / denom;
    const double f4 = wl_func(w, x4);
    if (f4 == 0.0) return;
    if (sign_of(f4) != sign_of(f3)) { x1 = x3; f1 = f3; x2 = x4; f2 = f4; }
    else if (sign_of(f4) != sign_of(f1)) { x2 = x4; f2 = f4; }
    else { x1 = x4; f1 = f4; }
    if (fabs(x2 - x1) <= tol) return;
  }
}
static void wall_law_solve(const ora_ctx *c, const double *state, double wallDist,
                           const double *normArea, const double *velWall, int isLower,
                           double vonKarmen, double wallConst, int mode, double wallValue,
                           wall_vars *wv);
static void wall_law_adiabatic(const ora_ctx *c, const double *state, double wallDist,
                               const double *normArea, const double *velWall, int isLower,
                               double vonKarmen, double wallConst, wall_vars *wv) {
  wall_law_solve(c, state, wallDist, normArea, velWall, isLower, vonKarmen, wallConst, 0, 0.0, wv);
}
/* mode 0: wallLaw::AdiabaticBCs :31-87; 1: HeatFluxBCs :89-145 (wallValue = q_w);
 * 2: IsothermalBCs :147-200 (wallValue = T_w) */
static void wall_law_solve(const ora_ctx *c, const double *state, double wallDist,
                           const double *normArea, const double *velWall, int isLower,
                           double vonKarmen, double wallConst, int mode, double wallValue,
                           wall_vars *wv) {
  wall_law w;
  memset(&w, 0, sizeof w);
  w.mode = mode;
  w.c = c; w.state = state; w.vonKarmen = vonKarmen; w.wallDist = wallDist;
  w.yplus0 = exp(-vonKarmen * wallConst);
  w.heatFlux = 0.0;
  const double vel[3] = {state[1] - velWall[0], state[2] - velWall[1], state[3] - velWall[2]};
  const double vn = dot3(vel, normArea);
  const double velTan[3] = {vel[0] - vn * normArea[0], vel[1] - vn * normArea[1],
                            vel[2] - vn * normArea[2]};
  w.velTanMag = mag3(velTan);
  const double t = temperature(c, state);
  w.tInt = t;
  w.recoveryFactor = pow(c->prandtl, 1.0 / 3.0);
  if (mode == 0) {         /* wall temperature from Crocco-Busemann, adiabatic */
    wl_set_wall_vars(&w, t + 0.5 * w.recoveryFactor * w.velTanMag * w.velTanMag / c->cp);
  } else if (mode == 1) {  /* guess: wall temperature equals interior temperature */
    w.heatFlux = wallValue;
    wl_set_wall_vars(&w, t);
  } else {
    wl_set_wall_vars(&w, wallValue);
  }
  wl_find_root(&w, 1.0e1, 1.0e4, 1.0e-8);
  wv->yplus = w.yplus_last;
  wv->heat_flux = w.heatFlux;          /* 0, q_w, or the last evaluation's Crocco-Busemann flux */
  /* CalcTurbVars with EddyVisc, wallLaw.cpp:243-279 */
  {
    const double dYplusWhite =
        2.0 * w.yplusWhite * w.vonKarmen * sqrt(w.gamma) / w.q *
        sqrt(fmax(1.0 - pow(2.0 * w.gamma * w.uplus - w.beta, 2.0) / (w.q * w.q), 0.0));
    const double ku = w.vonKarmen * w.uplus;
    w.mutW = w.muW * (1.0 + dYplusWhite - w.vonKarmen * w.yplus0 * (1.0 + ku + 0.5 * ku * ku)) -
             viscosity(c, t) * c->scaling;
    w.mutW = fmax(w.mutW, 0.0);
    const double wallBeta = is_wilcox_fwd(c) ? 0.0708 : 0.075;
    double wi = 6.0 * w.muW / (wallBeta * w.rhoW * wallDist * wallDist);
    wi *= c->scaling;
    double wo = w.uStar / (sqrt(0.09) * w.vonKarmen * wallDist);
    wo *= c->scaling;
    wv->sdr = sqrt(wi * wi + wo * wo);
    wv->tke = wv->sdr * w.mutW / state[0] * (1.0 / c->scaling);
  }
  wv->density = w.rhoW;
  wv->temperature = w.tW;
  wv->viscosity = w.muW;
  wv->turb_eddy_visc = w.mutW;
  wv->friction_velocity = w.uStar;
  const double ssm = w.uStar * w.uStar * w.rhoW;
  for (int q = 0; q < 3; ++q) {
    wv->shear[q] = ssm * velTan[q] / w.velTanMag;
    if (!isLower) wv->shear[q] *= -1.0;
  }
}

/* what the nonreflecting (LODI) branches of GetGhostState read besides the interior
 * state: dt and the state at time n of the adjacent cell, its pressure and velocity
 * gradients, Mach mean / maximum over the surface (procBlock.cpp:6233-6262) */
typedef struct { double dt, sn[NEQM], pg[3], vg[9], avg_mach, max_mach; } nr_data;
/* primitive::ApplyFarfieldTurbBC primitive.cpp:83-98 (k and omega from a turbulence
 * intensity and an eddy-viscosity ratio, then LimitTurb) */
static void apply_farfield_turb(const ora_ctx *c, double *s, const double *vel,
                                double turbInten, double viscRatio) {
  s[5] = 1.5 * pow(turbInten * mag3(vel), 2.0);
  s[6] = s[0] * s[5] / (viscRatio * viscosity(c, temperature(c, s)));
  for (int e = NF; e < NEQ; ++e)
    if (!(s[e] > TURB_MIN)) s[e] = TURB_MIN;
}
static int ghost_state_nr(const ora_ctx *c, const double *interior, int bc,
                          const double *areaUnit, int surf, const agx_bc_state *d,
                          int layer, double wallDist, double nuW, const nr_data *nr,
                          wall_vars *wv, double *ghost);
/* nuW: kinematic viscosity of the wall-adjacent cell (rans viscous walls,
 * procBlock.cpp:2814-2820) */
static int ghost_state(const ora_ctx *c, const double *interior, int bc,
                       const double *areaVec, int surf,
                       const agx_bc_state *d, int layer, double wallDist, double nuW,
                       double *ghost) {
  return ghost_state_nr(c, interior, bc, areaVec, surf, d, layer, wallDist, nuW, NULL, NULL, ghost);
}
static int ghost_state_nr(const ora_ctx *c, const double *interior, int bc,
                          const double *areaVec, int surf,
                          const agx_bc_state *d, int layer, double wallDist, double nuW,
                          const nr_data *nr, wall_vars *wv, double *ghost) {
  const int rans = NEQ > NF;
  for (int e = 0; e < NEQ; ++e) ghost[e] = interior[e];
  const int isLower = surf % 2 == 1;
  double n[3];
  for (int q = 0; q < 3; ++q) n[q] = isLower ? -1.0 * areaVec[q] : areaVec[q];
  if (bc == AGX_BC_SLIPWALL) {
    const double vn = dot3(interior + 1, n);
    for (int q = 0; q < 3; ++q)
      ghost[1 + q] = interior[1 + q] - 2.0 * n[q] * vn;
  } else if (bc == AGX_BC_VISCOUSWALL) {
    for (int q = 0; q < 3; ++q)
      ghost[1 + q] = 2.0 * d->velocity[q] - interior[1 + q];
    /* ghostStates.cpp:144-260: per thermal wall type the low-Re ghost density, or -- with
     * wall functions -- the wall law first, then the ghost density it implies; k and omega
     * at the wall from the wall law unless y+ < 10 switches the face back to low-Re */
    const double R = 0.0 + 1.0 * c->cfg.gas.gas_constant;
    int lowRe = 1;
    wall_vars loc;
    wall_vars *w = wv ? wv : &loc;
    if (d->is_isothermal) {
      const double tWall = d->wall_temperature;
      double tGhost = 2.0 * tWall - temperature(c, interior);
      if (d->is_wall_law) {
        wall_law_solve(c, interior, wallDist, n, d->velocity, isLower, d->von_karman,
                       d->wall_constant, 2, tWall, w);
        lowRe = w->yplus < 10.0;                     /* wallVars::SwitchToLowRe */
        if (!lowRe) {
          /* the wall law's heat flux with the turbulent conductivity (the eddy viscosity is
           * not zero at the wall), 2 x wall distance as gradient length :161-172 */
          const double kappa = conductivity(c, w->temperature) +
                               w->turb_eddy_visc * c->cp / turb_prandtl_fwd(c);
          tGhost = tWall - w->heat_flux / kappa * 2.0 * wallDist;
        }
      }
      /* idealGas::DensityTP eos.cpp:111-115, MixtureGasConstant */
      const double rho = ghost[4] / (R * tGhost);
      ghost[0] = rho * (interior[0] / interior[0]);
    } else if (d->is_heat_flux) {
      /* low-Re constant heat flux wall, ghostStates.cpp:228-242 */
      const double t = temperature(c, interior);
      const double kappa = conductivity(c, t);
      double tGhost = t - d->wall_heat_flux / kappa * 2.0 * wallDist;
      if (d->is_wall_law) {
        wall_law_solve(c, interior, wallDist, n, d->velocity, isLower, d->von_karman,
                       d->wall_constant, 1, d->wall_heat_flux, w);
        lowRe = w->yplus < 10.0;
        if (!lowRe) tGhost = 2.0 * w->temperature - t;      /* :213-219 */
      }
      const double rho = ghost[4] / (R * tGhost);
      ghost[0] = rho * (interior[0] / interior[0]);
    } else if (d->is_wall_law) {
      wall_law_adiabatic(c, interior, wallDist, n, d->velocity, isLower, d->von_karman,
                         d->wall_constant, w);
      lowRe = w->yplus < 10.0;
    }
    if (d->is_wall_law && rans && !lowRe) {
      ghost[5] = 2.0 * w->tke - interior[5];
      ghost[6] = 2.0 * w->sdr - interior[6];
      if (layer > 1) {
        ghost[5] = layer * ghost[5] - w->tke;
        ghost[6] = layer * ghost[6] - w->sdr;
      }
    }
    if (rans && lowRe) {
      /* low-Re wall, ghostStates.cpp:261-279: k = 0 at the face, omega of Menter's
       * wall value (WallBeta = beta1, turbulence.hpp:577) */
      ghost[5] = -1.0 * interior[5];
      const double wallBeta = c->cfg.turbulence_model == AGX_TURB_KW_WILCOX2006 ? 0.0708 : 0.075;
      const double wWall = c->scaling * c->scaling * 60.0 * nuW / (wallDist * wallDist * wallBeta);
      ghost[6] = 2.0 * wWall - interior[6];
      if (layer > 1) ghost[6] = layer * ghost[6] - wWall;
    }
  } else if (bc == AGX_BC_CHARACTERISTIC) {
    double fs[NEQM] = {d->density * 1.0, d->velocity[0], d->velocity[1],
                      d->velocity[2], d->pressure};
    const double velIntNorm = dot3(interior + 1, n);
    const double SoSInt = sos(c, interior);
    const double machInt = fabs(velIntNorm) / SoSInt;
    if (machInt >= 1.0 && velIntNorm < 0.0) {
      for (int e = 0; e < NEQ; ++e) ghost[e] = fs[e];
      if (rans) apply_farfield_turb(c, ghost, fs + 1, d->turb_intensity, d->eddy_visc_ratio);
    } else if (machInt >= 1.0 && velIntNorm >= 0.0) {
      /* supersonic outflow: interior */
    } else if (machInt < 1.0 && velIntNorm < 0.0) {
      const double rhoSoSInt = interior[0] * SoSInt;
      double velDiff[3] = {fs[1] - interior[1], fs[2] - interior[2],
                           fs[3] - interior[3]};
      ghost[4] = 0.5 * (fs[4] + interior[4] - rhoSoSInt * dot3(n, velDiff));
      const double dP = fs[4] - ghost[4];
      const double rho = fs[0] - dP / (SoSInt * SoSInt);
      ghost[0] = rho * (fs[0] / fs[0]);
      ghost[1] = fs[1] - n[0] * dP / rhoSoSInt;
      ghost[2] = fs[2] - n[1] * dP / rhoSoSInt;
      ghost[3] = fs[3] - n[2] * dP / rhoSoSInt;
      if (rans) apply_farfield_turb(c, ghost, fs + 1, d->turb_intensity, d->eddy_visc_ratio);
    } else if (machInt < 1.0 && velIntNorm >= 0.0) {
      const double rhoSoSInt = interior[0] * SoSInt;
      const double dP = interior[4] - fs[4];
      const double rho = interior[0] - dP / (SoSInt * SoSInt);
      ghost[0] = rho * (interior[0] / interior[0]);
      ghost[1] = interior[1] + n[0] * dP / rhoSoSInt;
      ghost[2] = interior[2] + n[1] * dP / rhoSoSInt;
      ghost[3] = interior[3] + n[2] * dP / rhoSoSInt;
      ghost[4] = fs[4];
    } else {
      return fail("characteristic BC: flow condition not recognized");
    }
    double tmp[NEQM];
    extrap_hold(ghost, 2.0, interior, tmp);
    memcpy(ghost, tmp, sizeof(double) * NEQ);
    if (layer > 1) {
      extrap_hold(ghost, (double)layer, interior, tmp);
      memcpy(ghost, tmp, sizeof(double) * NEQ);
      /* (whatever the flow direction, ghostStates.cpp:381-387) */
      if (rans) apply_farfield_turb(c, ghost, fs + 1, d->turb_intensity, d->eddy_visc_ratio);
    }
  } else if (bc == AGX_BC_INLET) {
    if (d->is_nonreflecting && !nr)
      return fail("nonreflecting inlet needs the state at time n (StoreOldSolution)");
    double fs[NEQM] = {d->density * 1.0, d->velocity[0], d->velocity[1],
                      d->velocity[2], d->pressure};
    const double velIntNorm = dot3(interior + 1, n);
    const double SoSInt = sos(c, interior);
    const double machInt = fabs(velIntNorm) / SoSInt;
    if (machInt >= 1.0) {
      for (int e = 0; e < NEQ; ++e) ghost[e] = fs[e];
      /* ghostStates.cpp:419-423 */
      if (rans) apply_farfield_turb(c, ghost, fs + 1, d->turb_intensity, d->eddy_visc_ratio);
    } else {
      const double rhoSoSInt = interior[0] * SoSInt;
      double velDiff[3] = {fs[1] - interior[1], fs[2] - interior[2],
                           fs[3] - interior[3]};
      ghost[4] = 0.5 * (fs[4] + interior[4] - rhoSoSInt * dot3(n, velDiff));
      if (d->is_nonreflecting) {
        /* LODI terms, ghostStates.cpp:435-462 */
        const double sigma = 0.25;
        const double rhoN = nr->sn[0], sosN = sos(c, nr->sn), rhoSoSN = rhoN * sosN;
        const double deltaPressure = ghost[4] - nr->sn[4];
        const double alpha = sigma * sosN / d->length_scale;
        const double rhoNp1 = (rhoN + nr->dt * alpha * fs[0] + deltaPressure / (sosN * sosN)) /
                              (1.0 + nr->dt * alpha);
        ghost[0] = rhoNp1 * 1.0;
        const double k = alpha * (1.0 - nr->max_mach * nr->max_mach);
        for (int q = 0; q < 3; ++q)
          ghost[1 + q] = (nr->sn[1 + q] + nr->dt * k * fs[1 + q] - n[q] * deltaPressure / rhoSoSN) /
                         (1.0 + nr->dt * k);
      } else {
      const double dP = fs[4] - ghost[4];
      const double rho = fs[0] - dP / (SoSInt * SoSInt);
      ghost[0] = rho * (fs[0] / fs[0]);
      ghost[1] = fs[1] - n[0] * dP / rhoSoSInt;
      ghost[2] = fs[2] - n[1] * dP / rhoSoSInt;
      ghost[3] = fs[3] - n[2] * dP / rhoSoSInt;
      }
      /* ghostStates.cpp:475-479 (reflecting and nonreflecting alike) */
      if (rans) apply_farfield_turb(c, ghost, fs + 1, d->turb_intensity, d->eddy_visc_ratio);
      double tmp[NEQM];
      extrap_hold(ghost, 2.0, interior, tmp);
      memcpy(ghost, tmp, sizeof(double) * NEQ);
      if (layer > 1) {
        extrap_hold(ghost, (double)layer, interior, tmp);
        memcpy(ghost, tmp, sizeof(double) * NEQ);
      }
    }
  } else if (bc == AGX_BC_SUPERSONIC_INFLOW) {
    ghost[0] = 0.0;
    ghost[0] = d->density * 1.0;
    ghost[1] = d->velocity[0];
    ghost[2] = d->velocity[1];
    ghost[3] = d->velocity[2];
    ghost[4] = d->pressure;
    /* ghostStates.cpp:513-517 */
    if (rans) apply_farfield_turb(c, ghost, d->velocity, d->turb_intensity, d->eddy_visc_ratio);
  } else if (bc == AGX_BC_SUPERSONIC_OUTFLOW) {
    if (layer > 1)
      for (int e = 0; e < NEQ; ++e) ghost[e] = layer * ghost[e] - interior[e];
  } else if (bc == AGX_BC_STAGNATION_INLET) {
    const double g = c->gamma - 1.0;
    const double sosI = sos(c, interior);
    const double vn = dot3(interior + 1, n);
    const double rNeg = vn - 2.0 * sosI / g;
    const double cosTheta = -1.0 * vn / mag3(interior + 1);
    const double stagSoSsq =
        pow(sosI, 2.0) + 0.5 * g * dot3(interior + 1, interior + 1);
    const double sosB =
        -1.0 * rNeg * g / (g * cosTheta * cosTheta + 2.0) *
        (1.0 + cosTheta * sqrt((g * cosTheta * cosTheta + 2.0) * stagSoSsq /
                                   (g * rNeg * rNeg) -
                               0.5 * g));
    const double tb = d->stagnation_temperature * (sosB * sosB / stagSoSsq);
    const double pb = d->stagnation_pressure *
                      pow(sosB * sosB / stagSoSsq, c->gamma / g);
    const double vbMag = sqrt(2.0 / g * (d->stagnation_temperature - tb));
    const double R = 0.0 + 1.0 * c->cfg.gas.gas_constant;
    const double rhoGhost = pb / (R * tb);
    ghost[0] = rhoGhost * 1.0;
    ghost[1] = vbMag * d->direction[0];
    ghost[2] = vbMag * d->direction[1];
    ghost[3] = vbMag * d->direction[2];
    ghost[4] = pb;
    /* farfield turbulence from the ghost velocity, ghostStates.cpp:581-585, :593-598 */
    if (rans) apply_farfield_turb(c, ghost, ghost + 1, d->turb_intensity, d->eddy_visc_ratio);
    double tmp[NEQM];
    extrap_hold(ghost, 2.0, interior, tmp);
    memcpy(ghost, tmp, sizeof(double) * NEQ);
    if (layer > 1) {
      extrap_hold(ghost, (double)layer, interior, tmp);
      memcpy(ghost, tmp, sizeof(double) * NEQ);
      if (rans) apply_farfield_turb(c, ghost, ghost + 1, d->turb_intensity, d->eddy_visc_ratio);
    }
  } else if (bc == AGX_BC_PRESSURE_OUTLET) {
    if (d->is_nonreflecting && !nr)
      return fail("nonreflecting outlet needs the state at time n (StoreOldSolution)");
    const double pb = d->pressure;
    const double SoSInt = sos(c, interior);
    const double rhoSoSInt = interior[0] * SoSInt;
    ghost[4] = pb;
    if (d->is_nonreflecting) {
      /* LODI + transverse terms, ghostStates.cpp:614-643 */
      const double *sn = nr->sn;
      const double dvel[3] = {interior[1] - sn[1], interior[2] - sn[2], interior[3] - sn[3]};
      const double deltaVel = dot3(dvel, n);
      const double sigma = 0.25;
      const double rhoN = sn[0], sosN = sos(c, sn), rhoSoSN = rhoN * sosN;
      const double k = sigma * sosN * (1.0 - nr->max_mach * nr->max_mach) / d->length_scale;
      const double beta = nr->avg_mach;
      const double pgn = dot3(nr->pg, n), vnn = dot3(sn + 1, n);
      double pGradT[3], velT[3], vgt[9], dVelN[3] = {0.0, 0.0, 0.0};
      for (int q = 0; q < 3; ++q) {
        pGradT[q] = nr->pg[q] - pgn * n[q];
        velT[q] = sn[1 + q] - vnn * n[q];
      }
      double sum = 0.0;
      for (int r = 0; r < 3; ++r) {          /* tensor::RemoveComponent, rows */
        const double rn = dot3(nr->vg + 3 * r, n);
        for (int q = 0; q < 3; ++q) vgt[3 * r + q] = nr->vg[3 * r + q] - rn * n[q];
      }
      for (int q = 0; q < 9; ++q) sum += vgt[q];                       /* tensor::Sum */
      for (int r = 0; r < 3; ++r)                                     /* LinearCombination */
        for (int q = 0; q < 3; ++q) dVelN[q] += vgt[3 * r + q] * n[r];
      const double dVelT = sum - (dVelN[0] + dVelN[1] + dVelN[2]);
      double tv[3];
      for (int q = 0; q < 3; ++q) tv[q] = pGradT[q] - rhoSoSN * dVelN[q];
      const double trans = -0.5 * (dot3(velT, tv) + c->gamma * sn[4] * dVelT);
      ghost[4] = (sn[4] + rhoSoSN * deltaVel + nr->dt * k * pb - nr->dt * beta * trans) /
                 (1.0 + nr->dt * k);
    }
    const double dP = interior[4] - ghost[4];
    const double rho = interior[0] - dP / (SoSInt * SoSInt);
    ghost[0] = rho * (interior[0] / interior[0]);
    ghost[1] = interior[1] + n[0] * dP / rhoSoSInt;
    ghost[2] = interior[2] + n[1] * dP / rhoSoSInt;
    ghost[3] = interior[3] + n[2] * dP / rhoSoSInt;
    if (dot3(ghost + 1, n) / sos(c, ghost) >= 1.0)
      for (int e = 0; e < NEQ; ++e) ghost[e] = interior[e];
    for (int e = 0; e < NEQ; ++e) ghost[e] = 2.0 * ghost[e] - interior[e];
    if (layer > 1)
      for (int e = 0; e < NEQ; ++e) ghost[e] = layer * ghost[e] - interior[e];
  } else if (bc == AGX_BC_INTERBLOCK || bc == AGX_BC_PERIODIC) {
    /* nothing */
  } else {
    return fail("ghost state for BC type %d is not supported", bc);
  }
  return 0;
}

/* cell index from (direction-3 index, dir-1 index, dir-2 index) of a surface:
 * dir3 = i: (d1, d2) = (j, k); dir3 = j: (k, i); dir3 = k: (i, j)
 * (boundaryConditions.cpp:2531-2577) */
static inline void surf_ijk(int d3, int a3, int a1, int a2, int *i, int *j,
                            int *k) {
  if (d3 == 0) { *i = a3; *j = a1; *k = a2; }
  else if (d3 == 1) { *j = a3; *k = a1; *i = a2; }
  else { *k = a3; *i = a1; *j = a2; }
}
static void surf_ranges(const agx_bc_surface *q, int d3, int *r1s, int *r1e,
                        int *r2s, int *r2e, int *r3s) {
  const int lo[3] = {q->imin, q->jmin, q->kmin};
  const int hi[3] = {q->imax, q->jmax, q->kmax};
  const int d1 = (d3 + 1) % 3, d2 = (d3 + 2) % 3;
  *r1s = lo[d1]; *r1e = hi[d1];
  *r2s = lo[d2]; *r2e = hi[d2];
  *r3s = lo[d3];
}

/* procBlock::AssignInviscidGhostCells procBlock.cpp:2449-2532 and
 * AssignViscousGhostCells :2760-2838 (viscous != 0: only viscousWall
 * surfaces are rewritten, with the viscousWall rule) */
static int assign_ghost_faces(ora_ctx *c, blk_t *b, int viscous) {
  const int nn[3] = {b->ni, b->nj, b->nk};
  for (int layer = 1; layer <= b->ng; ++layer) {
    for (int sn = 0; sn < b->nsurf; ++sn) {
      const agx_bc_surface *q = &b->surf[sn];
      const int st = surf_type(q);
      const int d3 = (st - 1) / 2;
      int r1s, r1e, r2s, r2e, r3s;
      surf_ranges(q, d3, &r1s, &r1e, &r2s, &r2e, &r3s);
      int gCell, iCell, aCell, bnd;
      if (st % 2 == 0) {
        gCell = r3s + layer - 1;
        iCell = r3s - layer;
        aCell = r3s - 1;
        if (iCell < 0) iCell = 0;
        bnd = r3s;
      } else {
        gCell = r3s - layer;
        iCell = r3s + layer - 1;
        aCell = r3s;
        if (iCell >= nn[d3]) iCell = nn[d3] - 1;
        bnd = r3s;
      }
      int bc = q->bc_type;
      if (bc == AGX_BC_INTERBLOCK || bc == AGX_BC_PERIODIC) continue;
      if (viscous) {
        if (bc != AGX_BC_VISCOUSWALL) continue;
      } else if (bc == AGX_BC_VISCOUSWALL) {
        bc = AGX_BC_SLIPWALL;
      }
      /* slipWall (and viscous-pass viscousWall) reflect the layer-th
       * interior cell, the others extrapolate from the adjacent cell */
      const int srcCell = (bc == AGX_BC_SLIPWALL || viscous) ? iCell : aCell;
      /* nonreflecting inlet / outlet: Mach mean and maximum over the surface
       * (procBlock.cpp:6233-6262; local patch only, as the reference) */
      const int is_nr = !viscous && q->state.is_nonreflecting &&
                        (bc == AGX_BC_INLET || bc == AGX_BC_PRESSURE_OUTLET);
      nr_data nr;
      memset(&nr, 0, sizeof nr);
      if (is_nr) {
        if (!c->have_time_n)
          return fail("nonreflecting boundary: the state at time n has not been stored");
        double sum = 0.0, mx = -1.7976931348623157e308;
        long cnt = 0;
        for (int a2 = r2s; a2 < r2e; ++a2)
          for (int a1 = r1s; a1 < r1e; ++a1) {
            int i, j, k, fi, fj, fk;
            surf_ijk(d3, aCell, a1, a2, &i, &j, &k);
            surf_ijk(d3, bnd, a1, a2, &fi, &fj, &fk);
            const double *ar = b->fa[d3] + 4 * FI(b, d3, fi, fj, fk);
            const double sg = st % 2 == 1 ? -1.0 : 1.0;
            const double nn3[3] = {sg * ar[0], sg * ar[1], sg * ar[2]};
            const double *sb = b->state + NEQ * CI(b, i, j, k);
            const double mach = dot3(sb + 1, nn3) / sos(c, sb);
            if (mach > mx) mx = mach;
            sum += mach;
            ++cnt;
          }
        nr.avg_mach = sum / (double)cnt;
        nr.max_mach = mx;
      }
      for (int a2 = r2s; a2 < r2e; ++a2) {
        for (int a1 = r1s; a1 < r1e; ++a1) {
          int i, j, k, gi, gj, gk, fi, fj, fk;
          surf_ijk(d3, srcCell, a1, a2, &i, &j, &k);
          surf_ijk(d3, gCell, a1, a2, &gi, &gj, &gk);
          surf_ijk(d3, bnd, a1, a2, &fi, &fj, &fk);
          const double *area = b->fa[d3] + 4 * FI(b, d3, fi, fj, fk);
          double g[NEQM];
          int wi, wj, wk;                    /* procBlock.cpp:2813: aCell */
          surf_ijk(d3, aCell, a1, a2, &wi, &wj, &wk);
          if (is_nr) {
            const long pa = PI(b, wi, wj, wk);
            nr.dt = b->dt[pa];
            cons_to_prim(c, b->consn + NEQ * pa, nr.sn);
            memcpy(nr.vg, b->grad18 + 18 * pa, sizeof nr.vg);
            memcpy(nr.pg, b->grad18 + 18 * pa + 15, sizeof nr.pg);
          }
          /* rans viscous walls: nu of the wall-adjacent cell from the viscosity_ of the
           * last UpdateAuxillaryVariables (procBlock.cpp:2814-2820) */
          const long qa = CI(b, wi, wj, wk);
          const double nuW = (NEQ > NF && viscous) ? b->visc[qa] / b->state[NEQ * qa] : 0.0;
          /* wallData_ is written by the first viscous ghost layer (procBlock.cpp:6289-6292) */
          wall_vars *wv = NULL;
          if (viscous && layer == 1 && bc == AGX_BC_VISCOUSWALL && q->state.is_wall_law &&
              b->wall_off && b->wall_off[sn] >= 0)
            wv = b->wallv + b->wall_off[sn] + (long)(a2 - r2s) * (r1e - r1s) + (a1 - r1s);
          if (ghost_state_nr(c, b->state + NEQ * CI(b, i, j, k), bc, area, st,
                             &q->state, layer, b->wdist ? b->wdist[qa] : 0.0, nuW,
                             is_nr ? &nr : NULL, wv, g))
            return 1;
          memcpy(b->state + NEQ * CI(b, gi, gj, gk), g, sizeof(double) * NEQ);
        }
      }
    }
  }
  return 0;
}

/* procBlock::AssignInviscidGhostCellsEdge procBlock.cpp:2565-2708 and
 * AssignViscousGhostCellsEdge :2874-3029 */
static int assign_ghost_edges(ora_ctx *c, blk_t *b, int viscous) {
  const int nn[3] = {b->ni, b->nj, b->nk};
  for (int dd = 0; dd < 3; ++dd) {
    const int d2 = (dd + 1) % 3, d3 = (dd + 2) % 3;
    const int max1 = nn[dd], max2 = nn[d2], max3 = nn[d3];
    const int surfStart2 = 2 * d2 + 1, surfStart3 = 2 * d3 + 1;
    for (int layer3 = 1; layer3 <= b->ng; ++layer3) {
      for (int layer2 = 1; layer2 <= b->ng; ++layer2) {
        for (int cc = 0; cc < 4; ++cc) {
          const int upper2 = cc > 1, upper3 = cc % 2 == 1;
          const int pCellD2 = upper2 ? max2 + layer2 - 2 : 1 - layer2;
          const int gCellD2 = upper2 ? pCellD2 + 1 : pCellD2 - 1;
          const int pCellD3 = upper3 ? max3 + layer3 - 2 : 1 - layer3;
          const int gCellD3 = upper3 ? pCellD3 + 1 : pCellD3 - 1;
          const int surf2 = upper2 ? surfStart2 + 1 : surfStart2;
          const int surf3 = upper3 ? surfStart3 + 1 : surfStart3;
          const int cFaceD2_2 = upper2 ? max2 : 0;
          const int cFaceD2_3 = upper3 ? max3 - 1 : 0;
          const int cFaceD3_2 = upper2 ? max2 - 1 : 0;
          const int cFaceD3_3 = upper3 ? max3 : 0;
          for (int d1 = 0; d1 < max1; ++d1) {
            /* operator()(dir, d1, d2, d3): dir=i -> (d1,d2,d3); j -> (d3,d1,d2);
             * k -> (d2,d3,d1)  (multiArray3d.hpp:215-241) */
            int idx[3];
#define PERM(A1, A2, A3) (idx[dd] = (A1), idx[d2] = (A2), idx[d3] = (A3))
            PERM(d1, cFaceD2_2, cFaceD2_3);
            const agx_bc_surface *s2 =
                get_bc_surface(b, idx[0], idx[1], idx[2], surf2);
            PERM(d1, cFaceD3_2, cFaceD3_3);
            const agx_bc_surface *s3 =
                get_bc_surface(b, idx[0], idx[1], idx[2], surf3);
            PERM(d1, cFaceD2_2, gCellD3);
            const double *fArea2 =
                b->fa[d2] + 4 * FI(b, d2, idx[0], idx[1], idx[2]);
            PERM(d1, gCellD2, cFaceD3_3);
            const double *fArea3 =
                b->fa[d3] + 4 * FI(b, d3, idx[0], idx[1], idx[2]);
            int bc2 = s2 ? s2->bc_type : -1;
            int bc3 = s3 ? s3->bc_type : -1;
            if (!viscous) {
              if (bc2 == AGX_BC_VISCOUSWALL) bc2 = AGX_BC_SLIPWALL;
              if (bc3 == AGX_BC_VISCOUSWALL) bc3 = AGX_BC_SLIPWALL;
            }
            PERM(d1, pCellD2, gCellD3);
            double *sP2 = b->state + NEQ * CI(b, idx[0], idx[1], idx[2]);
            PERM(d1, gCellD2, pCellD3);
            double *sP3 = b->state + NEQ * CI(b, idx[0], idx[1], idx[2]);
            PERM(d1, gCellD2, gCellD3);
            double *sG = b->state + NEQ * CI(b, idx[0], idx[1], idx[2]);
#undef PERM
            double g[NEQM];
            if (bc2 == AGX_BC_SLIPWALL && bc3 != AGX_BC_SLIPWALL) {
              if (ghost_state(c, sP2, bc2, fArea2, surf2, &s2->state, layer2, 0.0, 0.0, g))
                return 1;
              memcpy(sG, g, sizeof(double) * NEQ);
            } else if (bc2 != AGX_BC_SLIPWALL && bc3 == AGX_BC_SLIPWALL) {
              if (ghost_state(c, sP3, bc3, fArea3, surf3, &s3->state, layer3, 0.0, 0.0, g))
                return 1;
              memcpy(sG, g, sizeof(double) * NEQ);
            } else if (!viscous || (bc2 == AGX_BC_VISCOUSWALL &&
                                    bc3 == AGX_BC_VISCOUSWALL)) {
              if (layer2 == layer3) {
                for (int e = 0; e < NEQ; ++e) g[e] = 0.5 * (sP2[e] + sP3[e]);
                memcpy(sG, g, sizeof(double) * NEQ);
              } else if (layer2 > layer3) {
                memcpy(g, sP3, sizeof(double) * NEQ);
                memcpy(sG, g, sizeof(double) * NEQ);
              } else {
                memcpy(g, sP2, sizeof(double) * NEQ);
                memcpy(sG, g, sizeof(double) * NEQ);
              }
            }
          }
        }
      }
    }
  }
  return 0;
}

/* ------------------------------------------------------------------------ */
/* halo exchange index maps: GetSwapLoc boundaryConditions.cpp:3006-3181,
 * connection::First/SecondSliceIndices :1016-1150, AdjustForSlice :833-860,
 * InsertSlice multiArray3d.hpp:868-918                                      */
static void dirs_of(int boundary, int *d1, int *d2, int *d3) {
  *d3 = (boundary - 1) / 2;
  *d1 = (*d3 + 1) % 3;
  *d2 = (*d3 + 2) % 3;
}
static void build_side_map(const agx_connection *cc, int recv, const blk_t *br,
                           const blk_t *bs, long **dst_out, long **src_out,
                           long *n_out) {
  const int snd = 1 - recv;
  const int ng = br->ng;
  int orient = cc->orientation;
  if (recv == 1) { /* connection::SwapOrder */
    if (orient == 4) orient = 5; else if (orient == 5) orient = 4;
  }
  int rd1, rd2, rd3, sd1, sd2, sd3;
  dirs_of(cc->boundary[recv], &rd1, &rd2, &rd3);
  dirs_of(cc->boundary[snd], &sd1, &sd2, &sd3);
  /* adjusted receiver ranges */
  const int r_d1s = cc->d1_start[recv] - ng, r_d1e = cc->d1_end[recv] + ng;
  const int r_d2s = cc->d2_start[recv] - ng, r_d2e = cc->d2_end[recv] + ng;
  const int r_upper = cc->boundary[recv] % 2 == 0;
  const int blkStart = r_upper ? cc->const_surf[recv] : -ng;
  /* sender slice placement */
  const int s_upper = cc->boundary[snd] % 2 == 0;
  const int s_d3s = cc->const_surf[snd] + (s_upper ? -ng : 0);
  const int s_d1s = cc->d1_start[snd] - ng, s_d2s = cc->d2_start[snd] - ng;
  const int s_len1 = cc->d1_end[snd] - cc->d1_start[snd] + 2 * ng;
  const int s_len2 = cc->d2_end[snd] - cc->d2_start[snd] + 2 * ng;
  const int len1 = r_d1e - r_d1s, len2 = r_d2e - r_d2s;
  const int *pb = cc->patch_border + (recv == 0 ? 0 : 4);
  const int adjS1 = pb[0] ? ng : 0, adjE1 = pb[1] ? ng : 0;
  const int adjS2 = pb[2] ? ng : 0, adjE2 = pb[3] ? ng : 0;
  const int llu = (cc->boundary[0] + cc->boundary[1]) % 2 == 0;
  long cap = (long)ng * len1 * len2, n = 0;
  long *dst = (long *)malloc(sizeof(long) * (cap > 0 ? cap : 1));
  long *src = (long *)malloc(sizeof(long) * (cap > 0 ? cap : 1));
  for (int l3 = 0; l3 < ng; ++l3)
    for (int l2 = adjS2; l2 < len2 - adjE2; ++l2)
      for (int l1 = adjS1; l1 < len1 - adjE1; ++l1) {
        int a[3], s[3];
        a[rd1] = r_d1s + l1;
        a[rd2] = r_d2s + l2;
        a[rd3] = blkStart + l3;
        int q1, q2; /* slice-local in-plane indices along sender d1/d2 */
        if (orient == 2 || orient == 4 || orient == 5 || orient == 7) {
          q2 = (orient == 5 || orient == 7) ? s_len2 - 1 - l1 : l1;
          q1 = (orient == 4 || orient == 7) ? s_len1 - 1 - l2 : l2;
        } else if (sd3 == 0) { /* i-patch rule, cpp:3064-3073 */
          q1 = (orient == 6 || orient == 8) ? s_len1 - 1 - l1 : l1;
          q2 = (orient == 3 || orient == 8) ? s_len2 - 1 - l2 : l2;
        } else {
          q1 = (orient == 3 || orient == 8) ? s_len1 - 1 - l1 : l1;
          q2 = (orient == 6 || orient == 8) ? s_len2 - 1 - l2 : l2;
        }
        const int q3 = llu ? ng - l3 - 1 : l3;
        s[sd1] = s_d1s + q1;
        s[sd2] = s_d2s + q2;
        s[sd3] = s_d3s + q3;
        dst[n] = CI(br, a[0], a[1], a[2]);
        src[n] = CI(bs, s[0], s[1], s[2]);
        ++n;
      }
  *dst_out = dst; *src_out = src; *n_out = n;
}

/* what a halo exchange moves per cell: `ncopy` doubles from base[stride * cell + off];
 * a slab always carries NEQ slots per cell.  velocityGrad_ (9 per cell, swapped after
 * the residual for the off-diagonal terms of the block-matrix solvers,
 * gridLevel.cpp:343-368) goes in two halves. */
typedef struct { double *base; int stride, off, ncopy; } halo_view;
static halo_view halo_array(blk_t *b, int what) {
  halo_view v = {b->x, NEQ, 0, NEQ};
  if (what == AGX_HALO_STATE) v.base = b->state;
  else if (what == AGX_HALO_VELGRAD_A) { v.base = b->velgrad; v.stride = 9; v.off = 0; v.ncopy = 5; }
  else if (what == AGX_HALO_VELGRAD_B) { v.base = b->velgrad; v.stride = 9; v.off = 5; v.ncopy = 4; }
  else if (what == AGX_HALO_TURB) { v.base = b->turb3; v.stride = 3; v.off = 0; v.ncopy = 3; }
  return v;
}
static void halo_get(const halo_view *v, long cell, double *slot) {
  for (int e = 0; e < v->ncopy; ++e) slot[e] = v->base[(long)v->stride * cell + v->off + e];
}
static void halo_put(const halo_view *v, long cell, const double *slot) {
  for (int e = 0; e < v->ncopy; ++e) v->base[(long)v->stride * cell + v->off + e] = slot[e];
}

/* ------------------------------------------------------------------------ */
/* residual                                                                  */
static void face_states(const ora_ctx *c, const blk_t *b, int d, int i, int j,
                        int k, double *fl, double *fr) {
  /* procBlock::CalcInvFluxI/J/K reconstruction part, procBlock.cpp:393-431 */
  const int o[3] = {d == 0, d == 1, d == 2};
#define S(m) (b->state + NEQ * CI(b, i + (m) * o[0], j + (m) * o[1], k + (m) * o[2]))
#define W(m) (b->wid[d][CI(b, i + (m) * o[0], j + (m) * o[1], k + (m) * o[2])])
  if (c->cfg.recon == AGX_RECON_CONSTANT) {
    memcpy(fl, S(-1), NEQ * sizeof(double));
    memcpy(fr, S(0), NEQ * sizeof(double));
  } else if (c->cfg.recon == AGX_RECON_MUSCL) {
    recon_muscl(c, S(-2), S(-1), S(0), W(-2), W(-1), W(0), fl);
    recon_muscl(c, S(1), S(0), S(-1), W(1), W(0), W(-1), fr);
  } else {
    const int z = c->cfg.recon == AGX_RECON_WENOZ;
    recon_weno(S(-3), S(-2), S(-1), S(0), S(1), W(-3), W(-2), W(-1), W(0),
               W(1), z, fl);
    recon_weno(S(2), S(1), S(0), S(-1), S(-2), W(2), W(1), W(0), W(-1), W(-2),
               z, fr);
  }
#undef S
#undef W
}

/* procBlock::CalcInvFluxI/J/K procBlock.cpp:384-795 (scalar diagonal) */

static double proj_c2c(const blk_t *b, int d, int i, int j, int k);
/* ------------------------------------------------------------------------ */
/* block-matrix solvers (blusgs / bdplur): 5 x 5 flow Jacobians, row major    */
static int is_block(const ora_ctx *c) {    /* input::IsBlockMatrix input.cpp:713 */
  return c->cfg.matrix_solver == AGX_SOLVER_BLUSGS || c->cfg.matrix_solver == AGX_SOLVER_BDPLUR;
}
static int is_lusgs(const ora_ctx *c) {    /* input.cpp:847 */
  return c->cfg.matrix_solver == AGX_SOLVER_LUSGS || c->cfg.matrix_solver == AGX_SOLVER_BLUSGS;
}
#define NJ (NF * NF)
/* fluxJacobian::InvFluxJacobian fluxJacobian.hpp:483-560, one species (mf = 1) */
static void inv_flux_jacobian(const ora_ctx *c, const double *s, const double *area, double *J) {
  const double *n = area;
  const double velNorm = dot3(s + 1, n);
  const double gamma = c->gamma, gm1 = gamma - 1.0;
  const double phi = 0.5 * gm1 * dot3(s + 1, s + 1);
  double u[NEQM];
  prim_to_cons(c, s, u);
  const double a1 = gamma * (u[4] / s[0]) - phi;     /* primitive::Energy */
  const double a3 = gamma - 2.0;
  for (int q = 0; q < NJ; ++q) J[q] = 0.0;
#define JJ(r, cc) J[NF * (r) + (cc)]
  JJ(0, 0) = velNorm * (1.0 - 1.0);
  for (int q = 0; q < 3; ++q) {
    JJ(0, 1 + q) = 1.0 * n[q];
    JJ(1 + q, 0) = phi * n[q] - s[1 + q] * velNorm;
  }
  JJ(4, 0) = velNorm * (phi - a1);
  for (int cc = 0; cc < 3; ++cc) {          /* columns of the momentum equations */
    for (int r = 0; r < 3; ++r)
      JJ(1 + r, 1 + cc) = r == cc ? velNorm - a3 * n[cc] * s[1 + cc]
                                  : s[1 + r] * n[cc] - gm1 * s[1 + cc] * n[r];
    JJ(4, 1 + cc) = a1 * n[cc] - gm1 * s[1 + cc] * velNorm;
    JJ(1 + cc, 4) = gm1 * n[cc];
  }
  JJ(4, 4) = gamma * velNorm;
  for (int q = 0; q < NJ; ++q) J[q] *= 0.5 * area[3];
}
/* fluxJacobian::RusanovFluxJacobian fluxJacobian.hpp:446-479, InvFaceSpectralRadius
 * spectralRadius.hpp:67-80 */
static void rusanov_flux_jacobian(const ora_ctx *c, const double *s, const double *area,
                                  int positive, double *J) {
  const double specRad = 0.5 * area[3] * (fabs(dot3(s + 1, area)) + sos(c, s));
  inv_flux_jacobian(c, s, area, J);
  for (int e = 0; e < NF; ++e) JJ(e, e) = positive ? JJ(e, e) + specRad : JJ(e, e) - specRad;
}
/* fluxJacobian::ApproxTSLJacobian fluxJacobian.hpp:660-758 with
 * DelprimitiveDelConservative :613-656 and TauNormal utility.cpp:426-436; laminar,
 * one species */
static void tsl_jacobian(const ora_ctx *c, const double *s, double lamVisc, double turbVisc,
                         const double *area, double dist, int left, const double *vGrad,
                         double *J) {
  const double t = temperature(c, s);
  const double mu = c->scaling * lamVisc, mut = c->scaling * turbVisc;
  const double *n = area;
  const double velNorm = dot3(s + 1, n);
  const double rho = s[0];
  const double k = conductivity(c, t) * c->scaling;
  const double kt = mut * c->cp / 0.9;
  const double lambda = 0.0 - (2.0 / 3.0) * (mu + mut);
  const double trace = vGrad[0] + vGrad[4] + vGrad[8];
  double tauNorm[3];
  for (int r = 0; r < 3; ++r) {
    double mm = 0.0;
    for (int q = 0; q < 3; ++q) mm += (vGrad[3 * r + q] + vGrad[3 * q + r]) * n[q];
    tauNorm[r] = lambda * trace * n[r] + (mu + mut) * mm;
  }
  const double fac = left ? -1.0 : 1.0;
  const double third = 1.0 / 3.0;
  double T[NJ], P[NJ];
  for (int q = 0; q < NJ; ++q) { T[q] = 0.0; P[q] = 0.0; }
#define TT_(r, cc) T[NF * (r) + (cc)]
#define PP_(r, cc) P[NF * (r) + (cc)]
  TT_(0, 0) = 0.0;                                   /* DiffCoeff * (1 - mf) / ... */
  TT_(4, 0) = -(k + kt) * t / ((mu + mut) * rho) + 0.0;
  for (int cc = 0; cc < 3; ++cc) {
    for (int r = 0; r < 3; ++r) TT_(1 + r, 1 + cc) = third * n[cc] * n[r] + (r == cc ? 1.0 : 0.0);
    TT_(4, 1 + cc) = fac * 0.5 * dist / (mu + mut) * tauNorm[cc] + third * n[cc] * velNorm + s[1 + cc];
  }
  TT_(4, 4) = (k + kt) / ((mu + mut) * rho);
  for (int q = 0; q < NJ; ++q) T[q] *= area[3] * (mu + mut) / dist;
  const double gm1 = c->gamma - 1.0, invRho = 1.0 / rho;
  PP_(0, 0) = 1.0;
  for (int q = 0; q < 3; ++q) {
    PP_(1 + q, 0) = -invRho * s[1 + q];
    PP_(1 + q, 1 + q) = invRho;
    PP_(4, 1 + q) = -gm1 * s[1 + q];
  }
  PP_(4, 0) = 0.5 * gm1 * dot3(s + 1, s + 1);
  PP_(4, 4) = gm1;
  for (int q = 0; q < NJ; ++q) J[q] = 0.0;
  for (int cc = 0; cc < NF; ++cc)                   /* MatrixMultiply matrix.cpp:193-207 */
    for (int rr = 0; rr < NF; ++rr)
      for (int ii = 0; ii < NF; ++ii) JJ(rr, ii) += TT_(rr, cc) * PP_(cc, ii);
}
/* MatrixInverse matrix.cpp:57-103 (Gauss-Jordan, partial pivoting) */
static int matrix_inverse(double *m, int size) {
  double I[NJ];
  for (int r = 0; r < size; ++r)
    for (int q = 0; q < size; ++q) I[r * size + q] = r == q ? 1.0 : 0.0;
  for (int cPivot = 0, r = 0; r < size; ++r, ++cPivot) {
    double maxVal = 0.0;
    int rPivot = 0;                                  /* FindMaxInColumn(mat, size, r, cPivot, size-1) */
    for (int ii = cPivot; ii <= size - 1; ++ii)
      if (fabs(m[ii * size + r]) > maxVal) { maxVal = fabs(m[ii * size + r]); rPivot = ii; }
    if (r != rPivot)
      for (int q = 0; q < size; ++q) {
        double t = m[r * size + q]; m[r * size + q] = m[rPivot * size + q]; m[rPivot * size + q] = t;
        t = I[r * size + q]; I[r * size + q] = I[rPivot * size + q]; I[rPivot * size + q] = t;
      }
    if (r != 0)
      for (int ii = 0; ii < cPivot; ++ii) {
        const double factor = m[r * size + ii] / m[ii * size + ii];
        for (int q = 0; q < size; ++q) {
          m[r * size + q] = m[r * size + q] - factor * m[ii * size + q];
          I[r * size + q] = I[r * size + q] - factor * I[ii * size + q];
        }
      }
    if (m[r * size + cPivot] == 0.0) return fail("Singular matrix in Gauss-Jordan elimination!");
    const double normFactor = 1.0 / m[r * size + cPivot];
    for (int q = cPivot; q < size; ++q) m[r * size + q] *= normFactor;
    for (int q = 0; q < size; ++q) I[r * size + q] *= normFactor;
  }
  for (int cPivot = size - 2, r = size - 2; r >= 0; --r, --cPivot)
    for (int ii = size - 1; ii > cPivot; --ii) {
      const double factor = m[r * size + ii];
      for (int q = 0; q < size; ++q) {
        m[r * size + q] = m[r * size + q] - factor * m[ii * size + q];
        I[r * size + q] = I[r * size + q] - factor * I[ii * size + q];
      }
    }
  for (int q = 0; q < size * size; ++q) m[q] = I[q];
  return 0;
}
/* ArrayMultiplication fluxJacobian.hpp:50-87 (block branch) */
static void mat_vec(const double *m, const double *v, double *out) {
  for (int rr = 0; rr < NF; ++rr) {
    out[rr] = 0.0;
    for (int cc = 0; cc < NF; ++cc) out[rr] += m[NF * rr + cc] * v[cc];
  }
}

/* rans with a block-matrix solver: the 2 x 2 turbulence block of every Jacobian is
 * diagonal -- turbModel::InvJac turbulence.cpp:117-160 (0.5 (conv +- diss), the same
 * for k and omega), turbKWSst::ViscJac :772-795 (sigma_k / sigma_w), TurbSrcJac
 * :749-770 -- so it is kept as two numbers per cell, am_t / aminv_t [cells][2] */
static double turb_inv_jac(const double *s, const double *area, int positive) {
  const double velNorm = dot3(s + 1, area);
  return positive ? 0.5 * (velNorm * area[3] + fabs(velNorm) * area[3])
                  : 0.5 * (velNorm * area[3] - fabs(velNorm) * area[3]);
}
static void turb_visc_jac(const ora_ctx *c, const double *s, const double *area, double mu,
                          double dist, double mut, double f1, double *jk, double *jw);

static void calc_inv_flux(ora_ctx *c, blk_t *b, int d) {
  const int nn[3] = {b->ni, b->nj, b->nk};
  const int o[3] = {d == 0, d == 1, d == 2};
  const int implicit = c->cfg.time_integration >= AGX_TIME_IMPLICIT_EULER;
  /* threads (cpu_baseline on all host cores) take whole k-planes -- j-rows for the
   * k-faces --, so the two cells a face adds to belong to one thread and every cell
   * receives its contributions in the serial order: the result does not depend on
   * the number of threads */
  const int n_outer = d == 2 ? b->nj : b->nk, n_inner = d == 2 ? b->nk + 1 : b->nj + o[1];
#pragma omp parallel for schedule(static) if (b->ncell >= OMP_MIN_CELLS)
  for (int outer = 0; outer < n_outer; ++outer)
    for (int inner = 0; inner < n_inner; ++inner)
      for (int i = 0; i < b->ni + o[0]; ++i) {
        const int k = d == 2 ? inner : outer, j = d == 2 ? outer : inner;
        double fl[NEQM], fr[NEQM], flux[NEQM];
        face_states(c, b, d, i, j, k, fl, fr);
        const double *area = b->fa[d] + 4 * FI(b, d, i, j, k);
        if (c->cfg.inviscid_flux == AGX_FLUX_ROE)
          roe_flux(c, fl, fr, area, flux);
        else
          ausm_flux(c, fl, fr, area, flux);
        const int idx[3] = {i, j, k};
        if (idx[d] > 0) {
          const long pl = PI(b, i - o[0], j - o[1], k - o[2]);
          double *r = b->resid + NEQ * pl;
          for (int e = 0; e < NEQ; ++e) r[e] += flux[e] * area[3];
          if (implicit && is_block(c)) {             /* procBlock.cpp:452-457 */
            double J[NJ];
            rusanov_flux_jacobian(c, fl, area, 1, J);
            for (int q = 0; q < NJ; ++q) b->am[NJ * pl + q] += J[q];
            if (NEQ > NF) {
              const double tj = turb_inv_jac(fl, area, 1);
              b->am_t[2 * pl] += tj; b->am_t[2 * pl + 1] += tj;
            }
          }
        }
        if (idx[d] < nn[d]) {
          const long p = PI(b, i, j, k);
          double *r = b->resid + NEQ * p;
          for (int e = 0; e < NEQ; ++e) r[e] -= flux[e] * area[3];
          if (implicit && is_block(c)) {             /* procBlock.cpp:481-486 */
            double J[NJ];
            rusanov_flux_jacobian(c, fr, area, 0, J);
            for (int q = 0; q < NJ; ++q) b->am[NJ * p + q] -= J[q];
            if (NEQ > NF) {
              const double tj = turb_inv_jac(fr, area, 0);
              b->am_t[2 * p] -= tj; b->am_t[2 * p + 1] -= tj;
            }
          }
          const double *au =
              b->fa[d] + 4 * FI(b, d, i + o[0], j + o[1], k + o[2]);
          const double sr = inv_cell_spec_rad(
              c, b->state + NEQ * CI(b, i, j, k), area, au);
          b->specrad[p] += sr;
          if (implicit) b->a[p] += sr;
          if (NEQ > NF) {
            /* turbModel::InviscidCellSpectralRadius turbulence.cpp:162-172 */
            const double *sc = b->state + NEQ * CI(b, i, j, k);
            double v[3] = {0.5 * (area[0] + au[0]), 0.5 * (area[1] + au[1]),
                           0.5 * (area[2] + au[2])};
            const double m = mag3(v);
            const double nv[3] = {v[0] / m, v[1] / m, v[2] / m};
            const double tsr = fabs(dot3(sc + 1, nv)) * (0.5 * (area[3] + au[3]));
            b->specrad_t[p] += tsr;
            if (implicit) b->a_t[p] += tsr;
          }
        }
      }
}

/* procBlock::UpdateAuxillaryVariables procBlock.cpp:6171-6189 */
static void update_aux(ora_ctx *c, blk_t *b) {
#pragma omp parallel for schedule(static) if (b->ncell >= OMP_MIN_CELLS)
  for (int k = -b->ng; k < b->nk + b->ng; ++k)
    for (int j = -b->ng; j < b->nj + b->ng; ++j)
      for (int i = -b->ng; i < b->ni + b->ng; ++i) {
        if (at_corner(b, i, j, k)) continue;
        const long q = CI(b, i, j, k);
        b->temp[q] = temperature(c, b->state + NEQ * q);
        if (c->cfg.is_viscous) b->visc[q] = viscosity(c, b->temp[q]);
      }
}

/* Green-Gauss gradients at a face: procBlock::CalcGradsI/J/K
 * procBlock.cpp:5173-5786, VectorGradGG / ScalarGradGG utility.cpp:59-188.
 * d = face direction; t1, t2 = the two transverse directions in the order the
 * reference passes them to the GG functions (i, j, k order).                */
static void area_vec(const double *a, double *v) {
  v[0] = a[0] * a[3]; v[1] = a[1] * a[3]; v[2] = a[2] * a[3];
}
static void calc_grads(const blk_t *b, int d, int i, int j, int k,
                       double *velGrad, double *tGrad, double *kGrad, double *wGrad) {
  int o[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
  const int *od = o[d];
  double al[3][3], au[3][3]; /* [direction][component] */
  /* face direction d: 0.5*(A(face) + A(face+1)), 0.5*(A(face) + A(face-1)) */
  {
    double a0[3], a1[3], a2[3];
    area_vec(b->fa[d] + 4 * FI(b, d, i, j, k), a0);
    area_vec(b->fa[d] + 4 * FI(b, d, i + od[0], j + od[1], k + od[2]), a1);
    area_vec(b->fa[d] + 4 * FI(b, d, i - od[0], j - od[1], k - od[2]), a2);
    for (int q = 0; q < 3; ++q) {
      au[d][q] = 0.5 * (a0[q] + a1[q]);
      al[d][q] = 0.5 * (a0[q] + a2[q]);
    }
  }
  for (int t = 0; t < 3; ++t) {
    if (t == d) continue;
    const int *ot = o[t];
    double a0[3], a1[3];
    area_vec(b->fa[t] + 4 * FI(b, t, i + ot[0], j + ot[1], k + ot[2]), a0);
    area_vec(b->fa[t] + 4 * FI(b, t, i + ot[0] - od[0], j + ot[1] - od[1],
                               k + ot[2] - od[2]), a1);
    for (int q = 0; q < 3; ++q) au[t][q] = 0.5 * (a0[q] + a1[q]);
    area_vec(b->fa[t] + 4 * FI(b, t, i, j, k), a0);
    area_vec(b->fa[t] + 4 * FI(b, t, i - od[0], j - od[1], k - od[2]), a1);
    for (int q = 0; q < 3; ++q) al[t][q] = 0.5 * (a0[q] + a1[q]);
  }
  const long cL = CI(b, i - od[0], j - od[1], k - od[2]);
  const long cU = CI(b, i, j, k);
  const double vol = 0.5 * (b->vol[cL] + b->vol[cU]);
  const double invVol = 1.0 / vol;
  /* values on the six faces of the alternate control volume, 4 fields:
   * u, v, w, T */
  /* (rans: + k, omega -- ScalarGradGG of the turbulence variables) */
  const int nf = kGrad ? 6 : 4;
  double vl[3][6], vu[3][6];
  for (int f = 0; f < nf; ++f) {
#define VAL(cell) (f < 3 ? b->state[NEQ * (cell) + 1 + f] : (f == 3 ? b->temp[(cell)] : b->state[NEQ * (cell) + f + 1]))
    vl[d][f] = VAL(cL);
    vu[d][f] = VAL(cU);
    for (int t = 0; t < 3; ++t) {
      if (t == d) continue;
      const int *ot = o[t];
      const long cUu = CI(b, i + ot[0], j + ot[1], k + ot[2]);
      const long cLu = CI(b, i + ot[0] - od[0], j + ot[1] - od[1],
                          k + ot[2] - od[2]);
      const long cUl = CI(b, i - ot[0], j - ot[1], k - ot[2]);
      const long cLl = CI(b, i - ot[0] - od[0], j - ot[1] - od[1],
                          k - ot[2] - od[2]);
      vu[t][f] = 0.25 * (VAL(cL) + VAL(cU) + VAL(cUu) + VAL(cLu));
      vl[t][f] = 0.25 * (VAL(cL) + VAL(cU) + VAL(cUl) + VAL(cLl));
    }
#undef VAL
  }
  /* tensor data_[3*r + c]: row r = derivative direction, c = velocity comp */
  for (int r = 0; r < 3; ++r) {
    for (int f = 0; f < nf; ++f) {
      const double v =
          vu[0][f] * au[0][r] - vl[0][f] * al[0][r] + vu[1][f] * au[1][r] -
          vl[1][f] * al[1][r] + vu[2][f] * au[2][r] - vl[2][f] * al[2][r];
      if (f < 3) velGrad[3 * r + f] = v * invVol;
      else if (f == 3) tGrad[r] = v * invVol;
      else if (f == 4) kGrad[r] = v * invVol;
      else wGrad[r] = v * invVol;
    }
  }
}

/* Cell-centre gradients as the reference accumulates them: one sixth of each of
 * the six face gradients of the cell (procBlock.cpp:1397-1449 in the viscous flux
 * routines, CalcGradsI/J/K :5950-5954 otherwise); six fields: u, v, w, T, rho, p.
 * Formed on demand from the current state.  out: [cell][18] = velGrad[9] (3 r + c),
 * tGrad[3], rhoGrad[3], pGrad[3]; physical cells, i fastest. */
static double grad_field(const ora_ctx *c, const blk_t *b, long cell, int f) {
  const double *s = b->state + NEQ * cell;
  if (f < 3) return s[1 + f];
  if (f == 3) return temperature(c, s);
  if (f >= 6) return s[f - 1];               /* rans: k (6), omega (7) */
  return f == 4 ? s[0] : s[4];
}
#define NGF_MAX 8   /* u, v, w, T, rho, p [, k, omega] */
static void face_gradn(const ora_ctx *c, const blk_t *b, int d, int i, int j, int k,
                       int nf, double g6[3][NGF_MAX]) {
  static const int o[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
  const int *od = o[d];
  double au[3][3], al[3][3];
  {
    double a0[3], a1[3], a2[3];
    area_vec(b->fa[d] + 4 * FI(b, d, i, j, k), a0);
    area_vec(b->fa[d] + 4 * FI(b, d, i + od[0], j + od[1], k + od[2]), a1);
    area_vec(b->fa[d] + 4 * FI(b, d, i - od[0], j - od[1], k - od[2]), a2);
    for (int q = 0; q < 3; ++q) { au[d][q] = 0.5 * (a0[q] + a1[q]); al[d][q] = 0.5 * (a0[q] + a2[q]); }
  }
  for (int t = 0; t < 3; ++t) {
    if (t == d) continue;
    const int *ot = o[t];
    double a0[3], a1[3];
    area_vec(b->fa[t] + 4 * FI(b, t, i + ot[0], j + ot[1], k + ot[2]), a0);
    area_vec(b->fa[t] + 4 * FI(b, t, i + ot[0] - od[0], j + ot[1] - od[1], k + ot[2] - od[2]), a1);
    for (int q = 0; q < 3; ++q) au[t][q] = 0.5 * (a0[q] + a1[q]);
    area_vec(b->fa[t] + 4 * FI(b, t, i, j, k), a0);
    area_vec(b->fa[t] + 4 * FI(b, t, i - od[0], j - od[1], k - od[2]), a1);
    for (int q = 0; q < 3; ++q) al[t][q] = 0.5 * (a0[q] + a1[q]);
  }
  const long cL = CI(b, i - od[0], j - od[1], k - od[2]);
  const long cU = CI(b, i, j, k);
  const double invVol = 1.0 / (0.5 * (b->vol[cL] + b->vol[cU]));
  for (int f = 0; f < nf; ++f) {
    double vl[3], vu[3];
    const double fL = grad_field(c, b, cL, f), fU = grad_field(c, b, cU, f);
    vl[d] = fL;
    vu[d] = fU;
    for (int t = 0; t < 3; ++t) {
      if (t == d) continue;
      const int *ot = o[t];
      const long cUu = CI(b, i + ot[0], j + ot[1], k + ot[2]);
      const long cLu = CI(b, i + ot[0] - od[0], j + ot[1] - od[1], k + ot[2] - od[2]);
      const long cUl = CI(b, i - ot[0], j - ot[1], k - ot[2]);
      const long cLl = CI(b, i - ot[0] - od[0], j - ot[1] - od[1], k - ot[2] - od[2]);
      vu[t] = 0.25 * (fL + fU + grad_field(c, b, cUu, f) + grad_field(c, b, cLu, f));
      vl[t] = 0.25 * (fL + fU + grad_field(c, b, cUl, f) + grad_field(c, b, cLl, f));
    }
    for (int r = 0; r < 3; ++r)
      g6[r][f] = (vu[0] * au[0][r] - vl[0] * al[0][r] + vu[1] * au[1][r] - vl[1] * al[1][r] +
                  vu[2] * au[2][r] - vl[2] * al[2][r]) * invVol;
  }
}
/* out: [cell][3 nf]; per field f >= 3 the three derivatives at 9 + 3 (f - 3) */
static void cell_gradients_n(const ora_ctx *c, const blk_t *b, int nf, double *out) {
  const double sixth = 1.0 / 6.0;
  const int ng3 = 3 * nf;
  for (int k = 0; k < b->nk; ++k)
    for (int j = 0; j < b->nj; ++j)
      for (int i = 0; i < b->ni; ++i) {
        double acc[3 * NGF_MAX];
        for (int q = 0; q < ng3; ++q) acc[q] = 0.0;
        for (int d = 0; d < 3; ++d)
          for (int up = 0; up < 2; ++up) {
            double g6[3][NGF_MAX];
            face_gradn(c, b, d, i + (up && d == 0), j + (up && d == 1), k + (up && d == 2), nf, g6);
            for (int r = 0; r < 3; ++r) {
              for (int f = 0; f < 3; ++f) acc[3 * r + f] += sixth * g6[r][f];
              for (int f = 3; f < nf; ++f) acc[9 + 3 * (f - 3) + r] += sixth * g6[r][f];
            }
          }
        memcpy(out + ng3 * PI(b, i, j, k), acc, sizeof(double) * ng3);
      }
}
static void cell_gradients(const ora_ctx *c, const blk_t *b, double *out) {
  cell_gradients_n(c, b, 6, out);
}

/* FaceReconCentral reconstruction.hpp:315-328 with LagrangeCoeff(.,1,0,0) */
static void central_coeffs(double wU, double wD, double *cf) {
  const double w[2] = {wU, wD};
  lagrange_coeff(w, 1, 0, 0, cf);
}

/* viscousFlux::CalcFlux / CalcWallFlux viscousFlux.cpp:58-211, TauNormal
 * utility.cpp:426-437 (laminar, single species) */
/* k-omega SST 2003, turbulence.hpp:489-606 / turbulence.cpp:573-840 */
#define SST_BETA_STAR 0.09
#define SST_SIGMA_K1 0.85
#define SST_SIGMA_K2 1.0
#define SST_SIGMA_W1 0.5
#define SST_SIGMA_W2 0.856
#define SST_BETA1 0.075
#define SST_BETA2 0.0828
#define SST_GAMMA1 (5.0 / 9.0)
#define SST_GAMMA2 0.44
#define SST_A1 0.31
#define SST_KPROD2DEST 10.0
#define EPS_REF 1.0e-30                   /* macros.hpp.in:21 */
static double sst_blend(double c1, double c2, double f1) { return f1 * c1 + (1.0 - f1) * c2; }
static double sst_cdkw(const double *s, const double *kGrad, const double *wGrad) {
  const double v = 2.0 * s[0] * SST_SIGMA_W2 / s[6] * dot3(kGrad, wGrad);
  return v > 1.0e-10 ? v : 1.0e-10;       /* std::max(v, 1.0e-10) */
}
/* turbKWSst::EddyViscAndBlending turbulence.cpp:695-727 with Alpha1-3 :614-635,
 * F1 / F2 :592-603 and EddyVisc :573-589 */
static void sst_eddy_visc_blending(const ora_ctx *c, const double *s, const double *velGrad,
                                   const double *kGrad, const double *wGrad, double mu,
                                   double wallDist, double *mut, double *f1, double *f2) {
  const double wd = wallDist + EPS_REF;
  const double alpha1 = c->scaling * sqrt(s[5]) / (SST_BETA_STAR * s[6] * wd);
  const double alpha2 = c->scaling * c->scaling * 500.0 * mu / (wd * wd * s[0] * s[6]);
  const double cdkw = sst_cdkw(s, kGrad, wGrad);
  const double alpha3 = 4.0 * s[0] * SST_SIGMA_W2 * s[5] / (cdkw * wd * wd);
  const double m12 = alpha1 > alpha2 ? alpha1 : alpha2;
  const double arg1 = m12 < alpha3 ? m12 : alpha3;
  *f1 = tanh(pow(arg1, 4.0));
  const double arg2 = 2.0 * alpha1 > alpha2 ? 2.0 * alpha1 : alpha2;
  *f2 = tanh(arg2 * arg2);
  double ss = 0.0;                        /* strainRate.DoubleDotTrans(strainRate) */
  for (int r = 0; r < 3; ++r)
    for (int q = 0; q < 3; ++q) {
      const double srq = 0.5 * (velGrad[3 * r + q] + velGrad[3 * q + r]);
      const double sqr = 0.5 * (velGrad[3 * q + r] + velGrad[3 * r + q]);
      ss += srq * sqr;
    }
  const double meanStrainRate = sqrt(2.0 * ss);
  const double d1 = SST_A1 * s[6], d2 = c->scaling * meanStrainRate * *f2;
  *mut = s[0] * SST_A1 * s[5] / (d1 > d2 ? d1 : d2);
}

/* k-omega Wilcox 2006, turbulence.hpp:390-470 / turbulence.cpp:266-560 */
#define KW_GAMMA 0.52
#define KW_BETA_STAR 0.09
#define KW_SIGMA 0.5
#define KW_SIGMA_STAR 0.6
#define KW_SIGMA_D0 0.125
#define KW_BETA0 0.0708
#define KW_CLIM 0.875
static int is_wilcox(const ora_ctx *c) { return c->cfg.turbulence_model == AGX_TURB_KW_WILCOX2006; }
/* turbSstDes (turbulence.hpp:616-656): turbKWSst whose k destruction is scaled by phi */
static int is_sstdes(const ora_ctx *c) { return c->cfg.turbulence_model == AGX_TURB_SST_DES; }
/* TurbPrandtlNumber: 8/9 (Wilcox), 0.9 (SST) */
static double turb_prandtl(const ora_ctx *c) { return is_wilcox(c) ? 8.0 / 9.0 : 0.9; }
static double sigma_k(const ora_ctx *c, double f1) {
  return is_wilcox(c) ? KW_SIGMA_STAR : sst_blend(SST_SIGMA_K1, SST_SIGMA_K2, f1);
}
static double sigma_w(const ora_ctx *c, double f1) {
  return is_wilcox(c) ? KW_SIGMA : sst_blend(SST_SIGMA_W1, SST_SIGMA_W2, f1);
}
/* the eddy viscosity of the k / omega diffusion and of the turbulence spectral radii:
 * the limited one (SST) or EddyViscosityNoLim = rho k / omega (Wilcox,
 * UseUnlimitedEddyVisc turbulence.hpp:439, turbulence.cpp:495-548) */
static double turb_diff_visc(const ora_ctx *c, const double *s, double mut) {
  return is_wilcox(c) ? s[0] * s[5] / s[6] : mut;
}
/* turbKWWilcox::EddyViscAndBlending turbulence.cpp:412-430 with OmegaTilda :339-351 */
static void kw_eddy_visc_blending(const ora_ctx *c, const double *s, const double *velGrad,
                                  double *mut, double *f1, double *f2) {
  const double trace = velGrad[0] + velGrad[4] + velGrad[8];
  double ss = 0.0;
  for (int r = 0; r < 3; ++r)
    for (int q = 0; q < 3; ++q) {
      const double id = r == q ? 1.0 : 0.0;
      const double a = 0.5 * (velGrad[3 * r + q] + velGrad[3 * q + r]) - 1.0 / 3.0 * trace * id;
      const double bq = 0.5 * (velGrad[3 * q + r] + velGrad[3 * r + q]) - 1.0 / 3.0 * trace * id;
      ss += a * bq;
    }
  const double lim = c->scaling * KW_CLIM * sqrt(2.0 * ss / KW_BETA_STAR);
  const double omegaTilda = s[6] > lim ? s[6] : lim;
  *f1 = 1.0;
  *f2 = 0.0;
  *mut = s[0] * s[5] / omegaTilda;
}
/* turbKWWilcox::Beta / FBeta / Xw / StrainKI turbulence.cpp:291-337 */
static double kw_beta(const ora_ctx *c, const double *s, const double *vg) {
  double W[9], K[9], WW[9];
  const double trace = vg[0] + vg[4] + vg[8];
  for (int r = 0; r < 3; ++r)
    for (int q = 0; q < 3; ++q) {
      W[3 * r + q] = 0.5 * (vg[3 * r + q] - vg[3 * q + r]);
      K[3 * r + q] = 0.5 * (vg[3 * r + q] + vg[3 * q + r] - trace * (r == q ? 1.0 : 0.0));
    }
  for (int r = 0; r < 3; ++r)
    for (int q = 0; q < 3; ++q) {
      WW[3 * r + q] = 0.0;
      for (int m = 0; m < 3; ++m) WW[3 * r + q] += W[3 * r + m] * W[3 * m + q];
    }
  double ddot = 0.0;                       /* DoubleDotTrans: sum_ij A_ij B_ji */
  for (int r = 0; r < 3; ++r)
    for (int q = 0; q < 3; ++q) ddot += WW[3 * r + q] * K[3 * q + r];
  const double xw = fabs(ddot / pow(KW_BETA_STAR * s[6], 3.0)) * pow(c->scaling, 3.0);
  return KW_BETA0 * ((1.0 + 85.0 * xw) / (1.0 + 100.0 * xw));
}

/* viscousFlux::CalcFlux viscousFlux.cpp:58-135 (one species; turbVisc = 0 and no
 * turbulence entries in laminar runs) */
static void visc_flux(const ora_ctx *c, const double *velGrad,
                      const double *tGrad, const double *n, const double *s,
                      double lamVisc, double turbVisc, double f1, const double *kGrad,
                      const double *wGrad, double *f) {
  const double mu = c->scaling * lamVisc;
  const double mut = c->scaling * turbVisc;
  const double lambda = 0.0 - (2.0 / 3.0) * (mu + mut); /* sutherland::Lambda */
  const double trace = velGrad[0] + velGrad[4] + velGrad[8];
  double sym[9];
  for (int r = 0; r < 3; ++r)
    for (int q = 0; q < 3; ++q) sym[3 * r + q] = velGrad[3 * r + q] + velGrad[3 * q + r];
  double mm[3];
  for (int r = 0; r < 3; ++r)
    mm[r] = sym[3 * r] * n[0] + sym[3 * r + 1] * n[1] + sym[3 * r + 2] * n[2];
  double tau[3];
  for (int r = 0; r < 3; ++r)
    tau[r] = lambda * trace * n[r] + (mu + mut) * mm[r];
  f[0] = 0.0;
  f[1] = tau[0];
  f[2] = tau[1];
  f[3] = tau[2];
  const double t = temperature(c, s);
  const double kk = conductivity(c, t) * c->scaling;
  const double kt = mut * c->cp / turb_prandtl(c);
  f[4] = dot3(tau, s + 1) + (kk + kt) * dot3(tGrad, n) + 0.0;
  if (NEQ > NF) {
    const double tkeCoeff = sigma_k(c, f1);
    const double omgCoeff = sigma_w(c, f1);
    /* UseUnlimitedEddyVisc (Wilcox): NondimScaling * EddyViscNoLim(state) */
    const double mutt = is_wilcox(c) ? c->scaling * (s[0] * s[5] / s[6]) : mut;
    f[5] = (mu + tkeCoeff * mutt) * dot3(kGrad, n);
    f[6] = (mu + omgCoeff * mutt) * dot3(wGrad, n);
  }
}

/* turbKWSst::ViscJac turbulence.cpp:772-795 */
static void turb_visc_jac(const ora_ctx *c, const double *s, const double *area, double mu,
                          double dist, double mut, double f1, double *jk, double *jw) {
  const double length = area[3] / dist;
  *jk = c->scaling * length / s[0] * (mu + sigma_k(c, f1) * turb_diff_visc(c, s, mut));
  *jw = c->scaling * length / s[0] * (mu + sigma_w(c, f1) * turb_diff_visc(c, s, mut));
}

/* procBlock::CalcViscFluxI/J/K procBlock.cpp:1233-2135 (laminar, central) */
static void calc_visc_flux(ora_ctx *c, blk_t *b, int d) {
  const int nn[3] = {b->ni, b->nj, b->nk};
  const int o[3] = {d == 0, d == 1, d == 2};
  const int implicit = c->cfg.time_integration >= AGX_TIME_IMPLICIT_EULER;
  const double viscCoeff = c->cfg.viscous_cfl_coeff;
  const double sixth = 1.0 / 6.0;
  /* threads (cpu_baseline on all host cores) take whole k-planes -- j-rows for the
   * k-faces --, so the two cells a face adds to belong to one thread and every cell
   * receives its contributions in the serial order: the result does not depend on
   * the number of threads */
  const int n_outer = d == 2 ? b->nj : b->nk, n_inner = d == 2 ? b->nk + 1 : b->nj + o[1];
#pragma omp parallel for schedule(static) if (b->ncell >= OMP_MIN_CELLS)
  for (int outer = 0; outer < n_outer; ++outer)
    for (int inner = 0; inner < n_inner; ++inner)
      for (int i = 0; i < b->ni + o[0]; ++i) {
        const int k = d == 2 ? inner : outer, j = d == 2 ? outer : inner;
        const int rans = NEQ > NF;
        double velGrad[9], tGrad[3], kGrad[3] = {0, 0, 0}, wGrad[3] = {0, 0, 0};
        calc_grads(b, d, i, j, k, velGrad, tGrad, rans ? kGrad : NULL, rans ? wGrad : NULL);
        const long cL = CI(b, i - o[0], j - o[1], k - o[2]);
        const long cU = CI(b, i, j, k);
        double st[NEQM], mu;
        if (c->cfg.viscous_recon == AGX_VISC_RECON_CENTRAL_4TH) {
          /* FaceReconCentral4th reconstruction.hpp:335-379 with
           * LagrangeCoeff(cellWidth, 3, 1, 1); procBlock.cpp:1325-1346 */
          const long cU2 = CI(b, i - 2 * o[0], j - 2 * o[1], k - 2 * o[2]);
          const long cD2 = CI(b, i + o[0], j + o[1], k + o[2]);
          const double w4[4] = {b->wid[d][cU2], b->wid[d][cL], b->wid[d][cU], b->wid[d][cD2]};
          double cf4[4];
          lagrange_coeff(w4, 3, 1, 1, cf4);
          for (int e = 0; e < NEQ; ++e)
            st[e] = cf4[0] * b->state[NEQ * cU2 + e] + cf4[1] * b->state[NEQ * cL + e] +
                    cf4[2] * b->state[NEQ * cU + e] + cf4[3] * b->state[NEQ * cD2 + e];
          mu = cf4[0] * b->visc[cU2] + cf4[1] * b->visc[cL] + cf4[2] * b->visc[cU] +
               cf4[3] * b->visc[cD2];
        } else {
          double cf[2];
          central_coeffs(b->wid[d][cL], b->wid[d][cU], cf);
          for (int e = 0; e < NEQ; ++e)
            st[e] = cf[0] * b->state[NEQ * cU + e] + cf[1] * b->state[NEQ * cL + e];
          mu = cf[0] * b->visc[cU] + cf[1] * b->visc[cL];
        }
        const double *area = b->fa[d] + 4 * FI(b, d, i, j, k);
        double f[NEQM], mut = 0.0, f1 = 0.0, f2 = 0.0;
        /* wall-law boundary face (procBlock.cpp:1259-1299): the stored wall data give the
         * state, viscosities and the flux itself, unless y+ < 10 switched the face to
         * the low-Re treatment */
        const wall_vars *wl = NULL;
        const agx_bc_surface *wsurf = NULL;
        if (rans && b->wall_off && ((d == 0 ? i : d == 1 ? j : k) == 0 ||
                                    (d == 0 ? i : d == 1 ? j : k) == nn[d])) {
          const int stype = 2 * d + ((d == 0 ? i : d == 1 ? j : k) == 0 ? 1 : 2);
          wsurf = get_bc_surface(b, i, j, k, stype);
          if (wsurf && wsurf->bc_type == AGX_BC_VISCOUSWALL && wsurf->state.is_wall_law) {
            const long sn = wsurf - b->surf;
            int r1s, r1e, r2s, r2e, r3s;
            surf_ranges(wsurf, d, &r1s, &r1e, &r2s, &r2e, &r3s);
            const int c3[3] = {i, j, k};
            const int a1 = c3[(d + 1) % 3], a2 = c3[(d + 2) % 3];
            const wall_vars *w = b->wallv + b->wall_off[sn] + (long)(a2 - r2s) * (r1e - r1s) +
                                 (a1 - r1s);
            if (!(w->yplus < 10.0)) wl = w;
          }
        }
        if (wl) {
          const double invSc = 1.0 / c->scaling;
          f1 = 1.0; f2 = 1.0;
          mu = wl->viscosity * invSc;
          mut = wl->turb_eddy_visc * invSc;
          /* wallData::WallState wallData.cpp:299-313 */
          st[0] = wl->density;
          for (int q3 = 0; q3 < 3; ++q3) st[1 + q3] = wsurf->state.velocity[q3];
          st[4] = wl->density * (0.0 + 1.0 * c->cfg.gas.gas_constant) * wl->temperature;
          st[5] = wl->tke; st[6] = wl->sdr;
          /* viscousFlux::CalcWallLawFlux viscousFlux.cpp:214-247 (WallSigmaK / W: sigma_k1 /
           * sigma_w1 of SST, 0 of the base class -- Wilcox has no override) */
          const double wsk = is_wilcox(c) ? 0.0 : SST_SIGMA_K1, wsw = is_wilcox(c) ? 0.0 : SST_SIGMA_W1;
          f[0] = 0.0;
          for (int q3 = 0; q3 < 3; ++q3) f[1 + q3] = wl->shear[q3];
          f[4] = dot3(wl->shear, wsurf->state.velocity) + wl->heat_flux;
          f[5] = (wl->viscosity + wsk * wl->turb_eddy_visc) * dot3(kGrad, area);
          f[6] = (wl->viscosity + wsw * wl->turb_eddy_visc) * dot3(wGrad, area);
        } else
        if (rans) {
          /* state.LimitTurb, wall distance at the face, eddy viscosity and blending
           * (procBlock.cpp:1303-1355; the wall distance always by the two-cell rule) */
          for (int e = NF; e < NEQ; ++e)
            if (!(st[e] > TURB_MIN)) st[e] = TURB_MIN;
          double cf[2];
          central_coeffs(b->wid[d][cL], b->wid[d][cU], cf);
          double wDist = cf[0] * b->wdist[cU] + cf[1] * b->wdist[cL];
          if (wDist < 0.0 && wDist > -1.0e-10) wDist = 0.0;      /* WALL_DIST_NEG_TOL */
          if (is_wilcox(c)) kw_eddy_visc_blending(c, st, velGrad, &mut, &f1, &f2);
          else sst_eddy_visc_blending(c, st, velGrad, kGrad, wGrad, mu, wDist, &mut, &f1, &f2);
        }
        if (!wl) visc_flux(c, velGrad, tGrad, area, st, mu, mut, f1, kGrad, wGrad, f);
        const int idx[3] = {i, j, k};
        if (idx[d] > 0) {
          const long p = PI(b, i - o[0], j - o[1], k - o[2]);
          for (int e = 0; e < NEQ; ++e) b->resid[NEQ * p + e] -= f[e] * area[3];
          for (int q = 0; q < 9; ++q) b->velgrad[9 * cL + q] += sixth * velGrad[q];
          if (rans) {                                  /* procBlock.cpp:1401-1409 */
            b->turb3[3 * cL] += sixth * mut;
            b->turb3[3 * cL + 1] += sixth * f1;
            b->turb3[3 * cL + 2] += sixth * f2;
            for (int q = 0; q < 3; ++q) {
              b->kgrad[3 * p + q] += sixth * kGrad[q];
              b->wgrad[3 * p + q] += sixth * wGrad[q];
            }
          }
          if (implicit && is_block(c)) {             /* procBlock.cpp:1417-1424 */
            double J[NJ];
            tsl_jacobian(c, st, mu, mut, area, proj_c2c(b, d, i, j, k), 1, velGrad, J);
            for (int q = 0; q < NJ; ++q) b->am[NJ * p + q] -= J[q];
            if (rans) {             /* fac = -1 for the left cell, then Subtract */
              double jk, jw;
              turb_visc_jac(c, st, area, mu, proj_c2c(b, d, i, j, k), mut, f1, &jk, &jw);
              b->am_t[2 * p] -= -1.0 * jk; b->am_t[2 * p + 1] -= -1.0 * jw;
            }
          }
        }
        if (idx[d] < nn[d]) {
          const long p = PI(b, i, j, k);
          for (int e = 0; e < NEQ; ++e) b->resid[NEQ * p + e] += f[e] * area[3];
          for (int q = 0; q < 9; ++q) b->velgrad[9 * cU + q] += sixth * velGrad[q];
          if (rans) {                                  /* procBlock.cpp:1436-1444 */
            b->turb3[3 * cU] += sixth * mut;
            b->turb3[3 * cU + 1] += sixth * f1;
            b->turb3[3 * cU + 2] += sixth * f2;
            for (int q = 0; q < 3; ++q) {
              b->kgrad[3 * p + q] += sixth * kGrad[q];
              b->wgrad[3 * p + q] += sixth * wGrad[q];
            }
          }
          const double *au =
              b->fa[d] + 4 * FI(b, d, i + o[0], j + o[1], k + o[2]);
          const double vsr = visc_cell_spec_rad(c, b->state + NEQ * cU, area,
                                                au, b->vol[cU], b->visc[cU], mut);
          b->specrad[p] += vsr * viscCoeff;
          if (implicit) b->a[p] += 2.0 * vsr;
          if (rans) {
            /* turbKWSst::ViscousCellSpectralRadius turbulence.cpp:797-815 with the
             * face's mut and f1 (procBlock.cpp:1459-1466) */
            const double fMag = 0.5 * (area[3] + au[3]);
            const double tvsr = c->scaling * (fMag * fMag / b->vol[cU]) / b->state[NEQ * cU] *
                                (b->visc[cU] + sigma_k(c, f1) *
                                 turb_diff_visc(c, b->state + NEQ * cU, mut));
            b->specrad_t[p] += tvsr * viscCoeff;
            if (implicit) b->a_t[p] += 2.0 * tvsr;
          }
          if (implicit && is_block(c)) {             /* procBlock.cpp:1468-1475 */
            double J[NJ];
            tsl_jacobian(c, st, mu, mut, area, proj_c2c(b, d, i, j, k), 0, velGrad, J);
            for (int q = 0; q < NJ; ++q) b->am[NJ * p + q] += J[q];
            if (rans) {
              double jk, jw;
              turb_visc_jac(c, st, area, mu, proj_c2c(b, d, i, j, k), mut, f1, &jk, &jw);
              b->am_t[2 * p] += jk; b->am_t[2 * p + 1] += jw;
            }
          }
        }
      }
}

/* procBlock::CalcSrcTerms procBlock.cpp:5956-6027 with source::CalcTurbSrc
 * source.cpp:60-92 and turbKWSst::CalcTurbSrc turbulence.cpp:637-690 (own physical
 * cells only, so it may run before the turbulence variables are swapped) */
static void calc_src_terms(ora_ctx *c, blk_t *b) {
  const int implicit = c->cfg.time_integration >= AGX_TIME_IMPLICIT_EULER;
  const double invScaling = 1.0 / c->scaling;
#pragma omp parallel for schedule(static) if (b->ncell >= OMP_MIN_CELLS)
  for (int k = 0; k < b->nk; ++k)
    for (int j = 0; j < b->nj; ++j)
      for (int i = 0; i < b->ni; ++i) {
        const long p = PI(b, i, j, k), q = CI(b, i, j, k);
        const double *s = b->state + NEQ * q;
        const double *vg = b->velgrad + 9 * q;
        const double *kg = b->kgrad + 3 * p, *wg = b->wgrad + 3 * p;
        const double mut = b->turb3[3 * q], f1 = b->turb3[3 * q + 1];
        const double vol = b->vol[q];
        /* BoussinesqReynoldsStress turbulence.cpp:57-69, DoubleDotTrans with velGrad */
        const double lambda = 0.0 - (2.0 / 3.0) * mut;
        const double trace = vg[0] + vg[4] + vg[8];
        double ddot = 0.0;
        for (int r = 0; r < 3; ++r)
          for (int cc = 0; cc < 3; ++cc) {
            const double id = r == cc ? 1.0 : 0.0;
            const double tau = lambda * trace * id + mut * (vg[3 * r + cc] + vg[3 * cc + r]) -
                               2.0 / 3.0 * s[0] * s[5] * id;
            ddot += tau * vg[3 * cc + r];
          }
        /* turbSstDes::CalcTurbSrc turbulence.cpp:866-922: phi = max((1 - f2) Lt / (cdes width), 1)
         * with width = MaxCellWidth (procBlock.cpp:5993-5995) scales the k destruction */
        double phi = 1.0, width = 1.0;
        if (is_sstdes(c)) {
          const double f2 = b->turb3[3 * q + 2];
          width = fmax(fmax(b->wid[0][q], b->wid[1][q]), b->wid[2][q]);
          const double cdes = sst_blend(0.78, 0.61, f1);
          const double lt = sqrt(s[5]) / (SST_BETA_STAR * s[6]) * c->scaling;
          phi = fmax((1.0 - f2) * lt / (cdes * width), 1.0);
        }
        const double tkeDest = invScaling * SST_BETA_STAR * (s[0] * s[5] * s[6] * phi);
        double beta, src5, src6;
        if (is_wilcox(c)) {
          /* turbKWWilcox::CalcTurbSrc turbulence.cpp:359-407 */
          beta = kw_beta(c, s, vg);
          const double omgDest = invScaling * beta * (s[0] * s[6] * s[6]);
          double tkeProd = c->scaling * ddot;
          if (tkeProd < 0.0) tkeProd = 0.0;
          double omgProd = KW_GAMMA * s[6] / s[5] * tkeProd;
          if (omgProd < 0.0) omgProd = 0.0;
          const double kw = dot3(kg, wg);
          const double omgCd = c->scaling * (kw <= 0.0 ? 0.0 : KW_SIGMA_D0) * (s[0] / s[6] * kw);
          src5 = tkeProd - tkeDest;
          src6 = omgProd - omgDest + omgCd;
        } else {
          const double cdkw = sst_cdkw(s, kg, wg);
          const double gamma = sst_blend(SST_GAMMA1, SST_GAMMA2, f1);
          beta = sst_blend(SST_BETA1, SST_BETA2, f1);
          const double omgDest = invScaling * beta * (s[0] * s[6] * s[6]);
          double tkeProd = c->scaling * ddot;
          if (SST_KPROD2DEST * tkeDest < tkeProd) tkeProd = SST_KPROD2DEST * tkeDest;
          if (tkeProd < 0.0) tkeProd = 0.0;
          double omgProd = gamma * s[0] / mut * tkeProd;
          if (omgProd < 0.0) omgProd = 0.0;
          const double omgCd = c->scaling * (1.0 - f1) * cdkw;
          src5 = tkeProd - tkeDest;
          src6 = omgProd - omgDest + omgCd;
        }
        /* turbKWSst::SrcSpecRad turbulence.cpp:739-747; turbSstDes::SrcSpecRad :925-935 takes
         * the larger diagonal entry of TurbSrcJac with beta2 -- and receives the cell WIDTH
         * in the place of phi (procBlock.cpp:5993-6004 hands the same variable to both) */
        double turbSpecRad = -2.0 * SST_BETA_STAR * s[6] * vol * invScaling;
        if (is_sstdes(c)) {
          const double j00 = -2.0 * SST_BETA_STAR * s[6] * width * vol * invScaling;
          const double j11 = -2.0 * SST_BETA2 * s[6] * vol * invScaling;
          turbSpecRad = -1.0 * fmax(fabs(j00), fabs(j11));
        }
        b->specrad_t[p] -= turbSpecRad;
        if (implicit) b->a_t[p] -= turbSpecRad;
        if (implicit && is_block(c)) {       /* SubtractFromTurb(TurbSrcJac), :749-770 */
          b->am_t[2 * p] -= -2.0 * SST_BETA_STAR * s[6] * phi * vol * invScaling;
          b->am_t[2 * p + 1] -= -2.0 * beta * s[6] * vol * invScaling;
        }
        b->resid[NEQ * p + 5] -= src5 * vol;
        b->resid[NEQ * p + 6] -= src6 * vol;
      }
}

/* procBlock::CalcResidualNoSource procBlock.cpp:6111-6147 (the inviscid
 * branch's cell gradients, :6142-6145, feed only output and nonreflecting
 * BCs and are not formed here) */
static int calc_residual(ora_ctx *c, blk_t *b) {
  memset(b->resid, 0, sizeof(double) * NEQ * b->ncell);
  memset(b->specrad, 0, sizeof(double) * b->ncell);
  memset(b->velgrad, 0, sizeof(double) * 9 * b->ncell_g);
  if (NEQ > NF) {                       /* ResetTurbVars, ResetGradients */
    memset(b->specrad_t, 0, sizeof(double) * b->ncell);
    memset(b->turb3, 0, sizeof(double) * 3 * b->ncell_g);
    memset(b->kgrad, 0, sizeof(double) * 3 * b->ncell);
    memset(b->wgrad, 0, sizeof(double) * 3 * b->ncell);
  }
  for (int d = 0; d < 3; ++d) calc_inv_flux(c, b, d);
  if (c->cfg.is_viscous) {
    if (assign_ghost_faces(c, b, 1)) return 1;
    if (assign_ghost_edges(c, b, 1)) return 1;
    update_aux(c, b);
    for (int d = 0; d < 3; ++d) calc_visc_flux(c, b, d);
  } else {
    update_aux(c, b);
  }
  if (NEQ > NF) calc_src_terms(c, b);
  /* the cell gradients of this residual feed the nonreflecting ghost states of the
   * next ghost fill (pressureGrad_, velocityGrad_: procBlock.cpp:1397-1449 / :6143) */
  for (int sn = 0; sn < b->nsurf; ++sn)
    if (b->surf[sn].state.is_nonreflecting) { cell_gradients(c, b, b->grad18); break; }
  return 0;
}

/* procBlock::CalcBlockTimeStep / CalcCellDt procBlock.cpp:782-822 */
static int calc_dt(ora_ctx *c, blk_t *b, double cfl) {
  for (long p = 0; p < b->ncell; ++p) {
    if (c->cfg.dt_nondim > 0.0) {
      b->dt[p] = c->cfg.dt_nondim;
    } else if (cfl > 0.0) {
      int i = (int)(p % b->ni), j = (int)((p / b->ni) % b->nj),
          k = (int)(p / ((long)b->ni * b->nj));
      double sr = b->specrad[p] > 0.0 ? b->specrad[p] : 0.0; /* uncoupledScalar::Max() */
      if (NEQ > NF && b->specrad_t[p] > sr) sr = b->specrad_t[p];
      b->dt[p] = cfl * (b->vol[CI(b, i, j, k)] / sr);
    } else {
      return fail("Neither dt or cfl was specified!");
    }
  }
  return 0;
}

/* procBlock::UpdateBlock procBlock.cpp:826-872 with ExplicitEulerTimeAdvance
 * :882-899, RK4TimeAdvance :927-947, ImplicitTimeAdvance :902-916 */
static void update_block(ora_ctx *c, blk_t *b, int rr, double *l2,
                         agx_linf *linf) {
  const double alpha[4] = {0.25, 1.0 / 3.0, 0.5, 1.0};
  /* per k-plane partial norms, folded in plane order below (the same result for any
   * number of threads) */
  double *pl2 = (double *)calloc((size_t)b->nk * NEQM, sizeof(double));
  agx_linf *plinf = (agx_linf *)malloc((size_t)b->nk * sizeof(agx_linf));
#pragma omp parallel for schedule(static) if (b->ncell >= OMP_MIN_CELLS)
  for (int k = 0; k < b->nk; ++k) {
    double *l2k = pl2 + (size_t)k * NEQM;
    agx_linf *lk = plinf + k;
    lk->linf = -1.7976931348623157e308; lk->block = lk->i = lk->j = lk->k = lk->eqn = 0;
    for (int j = 0; j < b->nj; ++j)
      for (int i = 0; i < b->ni; ++i) {
        const long p = PI(b, i, j, k), q = CI(b, i, j, k);
        double *s = b->state + NEQ * q;
        const double *r = b->resid + NEQ * p;
        double u[NEQM], ns[NEQM];
        if (c->cfg.time_integration == AGX_TIME_EXPLICIT_EULER) {
          prim_to_cons(c, s, u);
          const double fac = b->dt[p] / b->vol[q];
          for (int e = 0; e < NEQ; ++e) u[e] -= fac * r[e];
          cons_to_prim(c, u, ns);
        } else if (c->cfg.time_integration == AGX_TIME_RK4) {
          const double fac = b->dt[p] / b->vol[q] * alpha[rr];
          for (int e = 0; e < NEQ; ++e)
            u[e] = b->consn[NEQ * p + e] - fac * r[e];
          cons_to_prim(c, u, ns);
        } else {
          update_prim_with_cons(c, s, b->x + NEQ * q, ns);
        }
        memcpy(s, ns, sizeof(double) * NEQ);
        for (int e = 0; e < NEQ; ++e) l2k[e] += r[e] * r[e];
        for (int e = 0; e < NEQ; ++e) {
          if (r[e] > lk->linf) {
            lk->linf = r[e];
            lk->block = b->parent;
            lk->i = i; lk->j = j; lk->k = k;
            lk->eqn = e + 1;
          }
        }
      }
  }
  for (int k = 0; k < b->nk; ++k) {
    for (int e = 0; e < NEQ; ++e) l2[e] += pl2[(size_t)k * NEQM + e];
    if (plinf[k].linf > linf->linf) {
      const int32_t pad = linf->pad_;
      *linf = plinf[k];
      linf->pad_ = pad;
    }
  }
  free(pl2);
  free(plinf);
}


/* ------------------------------------------------------------------------ */
/* implicit                                                                  */
/* procBlock::SolDeltaNCoeff / SolDeltaMmN / SolDeltaNm1 procBlock.cpp:1010-1035
 * and the 'b' term of linearSolver.cpp:370-374 */
/* (forcing: the multigrid forcing term is part of b in the LU-SGS sweeps,
 * linearSolver.cpp:377, :422; DPLUR adds it to b, :503; AXmB has none, :72-75) */
static void rhs_b_f(const ora_ctx *c, const blk_t *b, int i, int j, int k, int forcing,
                    double *out);
static void rhs_b(const ora_ctx *c, const blk_t *b, int i, int j, int k,
                  double *out) {
  rhs_b_f(c, b, i, j, k, 0, out);
}
static void rhs_b_f(const ora_ctx *c, const blk_t *b, int i, int j, int k, int forcing,
                    double *out) {
  const long p = PI(b, i, j, k), q = CI(b, i, j, k);
  const double thetaInv = 1.0 / c->cfg.theta;
  const double coeffN =
      (b->vol[q] * (1.0 + c->cfg.zeta)) / (b->dt[p] * c->cfg.theta);
  double u[NEQM];
  prim_to_cons(c, b->state + NEQ * q, u);
  const int multi = c->cfg.time_integration == AGX_TIME_BDF2;
  const double coeffNm1 = (b->vol[q] * c->cfg.zeta) / (b->dt[p] * c->cfg.theta);
  for (int e = 0; e < NEQ; ++e) {
    const double mmn = coeffN * (u[e] - b->consn[NEQ * p + e]);
    const double nm1 =
        multi ? coeffNm1 * (b->consn[NEQ * p + e] - b->consnm1[NEQ * p + e])
              : 0.0;
    const double f = forcing && b->forcing ? b->forcing[NEQ * p + e] : 0.0;
    out[e] = -thetaInv * b->resid[NEQ * p + e] + f + nm1 - mmn;
  }
}

/* RusanovScalarOffDiagonal fluxJacobian.cpp:122-162, FaceSpectralRadius
 * spectralRadius.hpp:182-203, ConvectiveFluxUpdate inviscidFlux.hpp:544-562 */
static void off_diagonal(const ora_ctx *c, const double *state, const double *diag,
                         const double *update, const double *fArea, double mu,
                         double dist, int positive, const double *vGrad, double mut, double f1,
                         double *out) {
  if (is_block(c)) {
    /* RusanovBlockOffDiagonal fluxJacobian.cpp:164-194 */
    double J[NJ];
    rusanov_flux_jacobian(c, state, fArea, positive, J);
    if (c->cfg.is_viscous) {
      double V[NJ];
      tsl_jacobian(c, state, mu, mut, fArea, dist, positive, vGrad, V);
      for (int q = 0; q < NJ; ++q) J[q] = positive ? J[q] - V[q] : J[q] + V[q];
    }
    mat_vec(J, update, out);
    if (NEQ > NF) {
      /* diagonal turbulence block: InvJac -+ fac * ViscJac (fluxJacobian.cpp:183-191,
       * fluxJacobian.hpp:749-757: fac = -1 when `left` = positive) */
      double tj = turb_inv_jac(state, fArea, positive), jk = 0.0, jw = 0.0;
      if (c->cfg.is_viscous) turb_visc_jac(c, state, fArea, mu, dist, mut, f1, &jk, &jw);
      const double fac = positive ? -1.0 : 1.0;
      const double dk = positive ? tj - fac * jk : tj + fac * jk;
      const double dw = positive ? tj - fac * jw : tj + fac * jw;
      out[5] = dk * update[5];
      out[6] = dw * update[6];
    }
    return;
  }
  double su[NEQM], fo[NEQM], fn[NEQM];
  update_prim_with_cons(c, state, update, su);
  if (c->cfg.inv_flux_jacobian == AGX_JACOBIAN_APPROX_ROE) {
    /* RoeOffDiagonal fluxJacobian.cpp:240-291 (inviscid: the viscous branch of the
     * reference receives dist and f1 in swapped order, :232-234 vs :240-245, and
     * divides by f1 = 0 in laminar runs; it is not restated) */
    roe_flux(c, state, diag, fArea, fo);
    if (positive) roe_flux(c, su, diag, fArea, fn);
    else roe_flux(c, diag, su, fArea, fn);
    for (int e = 0; e < NEQ; ++e) out[e] = fArea[3] * (fn[e] - fo[e]);
    return;
  }
  phys_flux(c, state, fArea, fo);
  phys_flux(c, su, fArea, fn);
  double sr = 0.5 * fArea[3] * (fabs(dot3(state + 1, fArea)) + sos(c, state));
  if (c->cfg.is_viscous) {
    const double a = 4.0 / (3.0 * state[0]);
    const double bq = c->gamma / state[0];
    const double maxTerm = a > bq ? a : bq;
    sr += fArea[3] / dist * maxTerm * visc_term_t(c, mu, mut);
  }
  for (int e = 0; e < NF; ++e) {
    const double fc = 0.5 * fArea[3] * (fn[e] - fo[e]);
    out[e] = positive ? fc + update[e] * sr : fc - update[e] * sr;
  }
  if (NEQ > NF) {
    /* turbulence entries: the flux change is dropped (fluxJacobian.cpp:145-148), the
     * spectral radius is turbModel::FaceSpectralRadius turbulence.hpp:309-330 =
     * InviscidFaceSpectralRadius (turbulence.cpp:174-186) + turbKWSst::
     * ViscousFaceSpectralRadius (:817-831) */
    const double velNorm = dot3(state + 1, fArea);
    double tsr = positive ? 0.5 * fArea[3] * fabs(velNorm + fabs(velNorm))
                          : 0.5 * fArea[3] * fabs(velNorm - fabs(velNorm));
    tsr += c->scaling * (fArea[3] / dist) / state[0] *
           (mu + sigma_k(c, f1) * turb_diff_visc(c, state, mut));
    for (int e = NF; e < NEQ; ++e) {
      const double fc = 0.0;
      out[e] = positive ? fc + update[e] * tsr : fc - update[e] * tsr;
    }
  }
}
/* procBlock::ProjC2CDist procBlock.cpp:6316-6342 */
static double proj_c2c(const blk_t *b, int d, int i, int j, int k) {
  const int o[3] = {d == 0, d == 1, d == 2};
  const double *cu = b->center + 3 * CI(b, i, j, k);
  const double *cl = b->center + 3 * CI(b, i - o[0], j - o[1], k - o[2]);
  double v[3] = {cu[0] - cl[0], cu[1] - cl[1], cu[2] - cl[2]};
  return dot3(v, b->fa[d] + 4 * FI(b, d, i, j, k));
}
/* procBlock::ImplicitLower / ImplicitUpper procBlock.cpp:1056-1163 */
static void implicit_lower(const ora_ctx *c, const blk_t *b, int i, int j,
                           int k, const double *x, double *L) {
  for (int e = 0; e < NEQ; ++e) L[e] = 0.0;
  for (int d = 0; d < 3; ++d) {
    const int o[3] = {d == 0, d == 1, d == 2};
    const int ii = i - o[0], jj = j - o[1], kk = k - o[2];
    if (is_physical(b, ii, jj, kk) || bc_is_connection(b, i, j, k, 2 * d + 1)) {
      const double dist = proj_c2c(b, d, i, j, k);
      const long q = CI(b, ii, jj, kk);
      double od[NEQM];
      off_diagonal(c, b->state + NEQ * q, b->state + NEQ * CI(b, i, j, k), x + NEQ * q,
                   b->fa[d] + 4 * FI(b, d, i, j, k),
                   c->cfg.is_viscous ? b->visc[q] : 0.0, dist, 1, b->velgrad + 9 * q,
                   NEQ > NF ? b->turb3[3 * q] : 0.0, NEQ > NF ? b->turb3[3 * q + 1] : 0.0, od);
      for (int e = 0; e < NEQ; ++e) L[e] += od[e];
    }
  }
}
static void implicit_upper(const ora_ctx *c, const blk_t *b, int i, int j,
                           int k, const double *x, double *U) {
  for (int e = 0; e < NEQ; ++e) U[e] = 0.0;
  for (int d = 0; d < 3; ++d) {
    const int o[3] = {d == 0, d == 1, d == 2};
    const int ii = i + o[0], jj = j + o[1], kk = k + o[2];
    if (is_physical(b, ii, jj, kk) ||
        bc_is_connection(b, ii, jj, kk, 2 * d + 2)) {
      const double dist = proj_c2c(b, d, ii, jj, kk);
      const long q = CI(b, ii, jj, kk);
      double od[NEQM];
      off_diagonal(c, b->state + NEQ * q, b->state + NEQ * CI(b, i, j, k), x + NEQ * q,
                   b->fa[d] + 4 * FI(b, d, ii, jj, kk),
                   c->cfg.is_viscous ? b->visc[q] : 0.0, dist, 0, b->velgrad + 9 * q,
                   NEQ > NF ? b->turb3[3 * q] : 0.0, NEQ > NF ? b->turb3[3 * q + 1] : 0.0, od);
      for (int e = 0; e < NEQ; ++e) U[e] += od[e];
    }
  }
}

static int requires_init(const ora_ctx *c) {
  /* input::MatrixRequiresInitialization input.cpp:1120-1125 */
  return c->cfg.matrix_solver == AGX_SOLVER_DPLUR || c->cfg.matrix_solver == AGX_SOLVER_BDPLUR ||
         c->cfg.matrix_sweeps > 1;
}

/* gridLevel::InvertDiagonal -> linearSolver::AddDiagonalTerms
 * linearSolver.cpp:146-175, Invert :177-188; InitializeMatrixUpdate :111-144 */
/* aInv.ArrayMult(i, j, k, v) (matMultiArray3d.hpp:141-160): scalar or block */
static void apply_ainv(const ora_ctx *c, const blk_t *b, long p, const double *v, double *out) {
  if (is_block(c)) {
    mat_vec(b->aminv + NJ * p, v, out);
    for (int e = NF; e < NEQ; ++e) out[e] = b->aminv_t[2 * p + e - NF] * v[e];
  } else {
    for (int e = 0; e < NF; ++e) out[e] = v[e] * b->ainv[p];
    for (int e = NF; e < NEQ; ++e) out[e] = v[e] * b->ainv_t[p];   /* turbulence part */
  }
}
static int implicit_begin_x(ora_ctx *c, blk_t *b, int init_x);
static int implicit_begin(ora_ctx *c, blk_t *b) { return implicit_begin_x(c, b, 1); }
static int implicit_begin_x(ora_ctx *c, blk_t *b, int init_x) {
  int singular = 0;
#pragma omp parallel for schedule(static) if (b->ncell >= OMP_MIN_CELLS) reduction(|:singular)
  for (int k = 0; k < b->nk; ++k)
    for (int j = 0; j < b->nj; ++j)
      for (int i = 0; i < b->ni; ++i) {
        const long p = PI(b, i, j, k), q = CI(b, i, j, k);
        double diagVolTime =
            (b->vol[q] * (1.0 + c->cfg.zeta)) / (b->dt[p] * c->cfg.theta);
        if (c->cfg.dual_time_cfl > 0.0) {
          double sr = b->specrad[p] > 0.0 ? b->specrad[p] : 0.0;
          if (NEQ > NF && b->specrad_t[p] > sr) sr = b->specrad_t[p];
          diagVolTime += sr / c->cfg.dual_time_cfl;
        }
        b->a[p] *= c->cfg.matrix_relaxation;
        b->a[p] += diagVolTime;
        b->ainv[p] = 1.0 / b->a[p];
        if (NEQ > NF) {
          b->a_t[p] *= c->cfg.matrix_relaxation;
          b->a_t[p] += diagVolTime;
          b->ainv_t[p] = 1.0 / b->a_t[p];
        }
        if (is_block(c)) {     /* MultiplyOnDiagonal / AddOnDiagonal / Inverse */
          double *m = b->am + NJ * p, *mi = b->aminv + NJ * p;
          for (int e = 0; e < NF; ++e) {
            m[NF * e + e] *= c->cfg.matrix_relaxation;
            m[NF * e + e] += diagVolTime;
          }
          memcpy(mi, m, sizeof(double) * NJ);
          if (matrix_inverse(mi, NF)) { singular |= 1; continue; }
          for (int e = 0; e < NEQ - NF; ++e) {
            b->am_t[2 * p + e] *= c->cfg.matrix_relaxation;
            b->am_t[2 * p + e] += diagVolTime;
            if (b->am_t[2 * p + e] == 0.0) { singular |= 1; continue; }
            b->aminv_t[2 * p + e] = 1.0 / b->am_t[2 * p + e];
          }
        }
      }
  if (singular) return fail("Singular matrix in Gauss-Jordan elimination!");
  if (!init_x) return 0;      /* (a coarse multigrid level: x is the restricted one) */
  if (requires_init(c)) {
#pragma omp parallel for schedule(static) if (b->ncell >= OMP_MIN_CELLS)
    for (int k = 0; k < b->nk; ++k)
      for (int j = 0; j < b->nj; ++j)
        for (int i = 0; i < b->ni; ++i) {
          const long p = PI(b, i, j, k), q = CI(b, i, j, k);
          double rb[NEQM];
          rhs_b(c, b, i, j, k, rb);
          apply_ainv(c, b, p, rb, b->x + NEQ * q);
        }
  } else {
    memset(b->x, 0, sizeof(double) * NEQ * b->ncell_g);
  }
  return 0;
}

/* lusgs::LUSGS_Forward linearSolver.cpp:341-383; hyperplane order
 * HyperplaneReorder utility.cpp:377-398 generated on the fly */
static void lusgs_forward(ora_ctx *c, blk_t *b, int sweep) {
  const int nplanes = b->ni + b->nj + b->nk - 2;
  for (int pp = 0; pp < nplanes; ++pp)
#pragma omp parallel for schedule(static) if (b->ncell >= OMP_MIN_CELLS)   /* the cells of a hyperplane are independent */
    for (int k = 0; k < b->nk; ++k)
      for (int j = 0; j < b->nj; ++j) {
        const int i = pp - j - k;
        if (i < 0 || i >= b->ni) continue;
        double off[NEQM], U[NEQM], rb[NEQM];
        implicit_lower(c, b, i, j, k, b->x, off);
        if (sweep > 0 || requires_init(c)) {
          implicit_upper(c, b, i, j, k, b->x, U);
          for (int e = 0; e < NEQ; ++e) off[e] -= U[e];
        }
        rhs_b_f(c, b, i, j, k, 1, rb);
        const long p = PI(b, i, j, k), q = CI(b, i, j, k);
        double v[NEQM];
        for (int e = 0; e < NEQ; ++e) v[e] = rb[e] + off[e];
        apply_ainv(c, b, p, v, b->x + NEQ * q);
      }
}
/* lusgs::LUSGS_Backward linearSolver.cpp:385-428 */
static void lusgs_backward(ora_ctx *c, blk_t *b, int sweep) {
  const int nplanes = b->ni + b->nj + b->nk - 2;
  for (int pp = nplanes - 1; pp >= 0; --pp)
#pragma omp parallel for schedule(static) if (b->ncell >= OMP_MIN_CELLS)
    for (int k = b->nk - 1; k >= 0; --k)
      for (int j = b->nj - 1; j >= 0; --j) {
        const int i = pp - j - k;
        if (i < 0 || i >= b->ni) continue;
        double U[NEQM], L[NEQM], rb[NEQM];
        implicit_upper(c, b, i, j, k, b->x, U);
        const long p = PI(b, i, j, k), q = CI(b, i, j, k);
        if (sweep > 0 || requires_init(c)) {
          implicit_lower(c, b, i, j, k, b->x, L);
          rhs_b_f(c, b, i, j, k, 1, rb);
          double v[NEQM];
          for (int e = 0; e < NEQ; ++e) v[e] = rb[e] + L[e] - U[e];
          apply_ainv(c, b, p, v, b->x + NEQ * q);
        } else {
          double v[NEQM];
          apply_ainv(c, b, p, U, v);
          for (int e = 0; e < NEQ; ++e) b->x[NEQ * q + e] = b->x[NEQ * q + e] - v[e];
        }
      }
}
/* dplur::DPLUR linearSolver.cpp:473-507 */
static void dplur_sweep(ora_ctx *c, blk_t *b) {
  memcpy(b->xold, b->x, sizeof(double) * NEQ * b->ncell_g);
#pragma omp parallel for schedule(static) if (b->ncell >= OMP_MIN_CELLS)
  for (int k = 0; k < b->nk; ++k)
    for (int j = 0; j < b->nj; ++j)
      for (int i = 0; i < b->ni; ++i) {
        double off[NEQM], U[NEQM], rb[NEQM];
        implicit_lower(c, b, i, j, k, b->xold, off);
        implicit_upper(c, b, i, j, k, b->xold, U);
        for (int e = 0; e < NEQ; ++e) off[e] -= U[e];
        rhs_b(c, b, i, j, k, rb);
        const long p = PI(b, i, j, k), q = CI(b, i, j, k);
        double v[NEQM];
        for (int e = 0; e < NEQ; ++e)
          v[e] = rb[e] + (b->forcing ? b->forcing[NEQ * p + e] : 0.0) + off[e];
        apply_ainv(c, b, p, v, b->x + NEQ * q);
      }
}
/* linearSolver::AXmB :58-90 and Residual :92-109, squared and summed as in
 * mgSolution::CycleAtLevel mgSolution.cpp:198-206 */
static void matrix_residual(ora_ctx *c, blk_t *b, double *sumsq, long *size) {
  double *part = (double *)calloc((size_t)b->nk, sizeof(double));
  double *opart = (double *)calloc((size_t)b->nk, sizeof(double));
#pragma omp parallel for schedule(static) if (b->ncell >= OMP_MIN_CELLS)
  for (int k = 0; k < b->nk; ++k)
    for (int j = 0; j < b->nj; ++j)
      for (int i = 0; i < b->ni; ++i) {
        double off[NEQM], U[NEQM], rb[NEQM];
        implicit_lower(c, b, i, j, k, b->x, off);
        implicit_upper(c, b, i, j, k, b->x, U);
        for (int e = 0; e < NEQ; ++e) off[e] -= U[e];
        rhs_b(c, b, i, j, k, rb);
        const long p = PI(b, i, j, k), q = CI(b, i, j, k);
        double ax[NEQM];
        if (is_block(c)) {
          mat_vec(b->am + NJ * p, b->x + NEQ * q, ax);
          for (int e = NF; e < NEQ; ++e) ax[e] = b->am_t[2 * p + e - NF] * b->x[NEQ * q + e];
        } else {
          for (int e = 0; e < NF; ++e) ax[e] = b->x[NEQ * q + e] * b->a[p];
          for (int e = NF; e < NEQ; ++e) ax[e] = b->x[NEQ * q + e] * b->a_t[p];
        }
        for (int e = 0; e < NEQ; ++e) {
          const double axmb = ax[e] - off[e] - rb[e];
          const double r = (b->forcing ? b->forcing[NEQ * p + e] : 0.0) - axmb;
          if (b->mres) b->mres[NEQ * p + e] = r;
          part[k] += r * r;
          opart[k] += ax[e] * ax[e] + off[e] * off[e] + rb[e] * rb[e];
        }
      }
  for (int k = 0; k < b->nk; ++k) *sumsq += part[k];   /* folded in plane order */
  for (int k = 0; k < b->nk; ++k) c->mres_opsq += opart[k];
  free(part);
  free(opart);
  *size += NEQ * b->ncell_g;
}

/* ------------------------------------------------------------------------ */
/* API                                                                       */
const char *ora_last_error(void) { return g_err; }
const char *ora_version(void) { return "aither-oracle 0.1 (CPU restatement)"; }

int ora_ctx_create(int device, int rank, ora_ctx **out) {
  if (getenv("ORA_OMP_MIN_CELLS")) g_omp_min_cells = atol(getenv("ORA_OMP_MIN_CELLS"));
  (void)device;
  ora_ctx *c = (ora_ctx *)calloc(1, sizeof *c);
  if (!c) return fail("out of memory");
  c->rank = rank;
  *out = c;
  return 0;
}
static void free_blk(blk_t *b) {
  free(b->forcing); free(b->mres); free(b->xsave);
  double **ptrs[] = {&b->state, &b->fa[0], &b->fa[1], &b->fa[2], &b->vol,
                     &b->center, &b->wid[0], &b->wid[1], &b->wid[2],
                     &b->wdist, &b->temp, &b->visc, &b->velgrad, &b->grad18, &b->resid,
                     &b->specrad, &b->dt, &b->consn, &b->consnm1, &b->x,
                     &b->xold, &b->a, &b->ainv, &b->am, &b->aminv, &b->specrad_t,
                     &b->a_t, &b->ainv_t, &b->turb3, &b->kgrad, &b->wgrad, &b->am_t, &b->aminv_t};
  for (size_t n = 0; n < sizeof ptrs / sizeof *ptrs; ++n) {
    free(*ptrs[n]);
    *ptrs[n] = NULL;
  }
  free(b->surf);
  b->surf = NULL;
  free(b->wall_off); b->wall_off = NULL;
  free(b->wallv); b->wallv = NULL;
}
void ora_ctx_destroy(ora_ctx *c) {
  if (c && c->have_cfg) --g_live_cfg;
  if (!c) return;
  for (int n = 0; n < c->nblk; ++n) free_blk(&c->blk[n]);
  for (int n = 0; n < c->nconn; ++n)
    for (int s = 0; s < 2; ++s) {
      free(c->conn[n].dst[s]);
      free(c->conn[n].src[s]);
    }
  free(c);
}
int ora_ctx_set_stream(ora_ctx *c, void *s) { (void)c; (void)s; return 0; }

/* test hook (tests/test_block_matrix.py): the Jacobians of the block-matrix solvers
 * for one state / face.  which = 0: RusanovFluxJacobian, 1: ApproxTSLJacobian,
 * 2: MatrixInverse of the 25 numbers in `vgrad_or_mat` */
int ora_debug_jacobian(ora_ctx *c, int which, const double *state, const double *area,
                       double mu, double dist, int flag, const double *vgrad_or_mat,
                       double *out25) {
  if (!c->have_cfg) return fail("config_set first");
  if (which == 0) rusanov_flux_jacobian(c, state, area, flag, out25);
  else if (which == 1) tsl_jacobian(c, state, mu, 0.0, area, dist, flag, vgrad_or_mat, out25);
  else {
    memcpy(out25, vgrad_or_mat, sizeof(double) * NJ);
    return matrix_inverse(out25, NF);
  }
  return 0;
}

int ora_config_set(ora_ctx *c, const agx_config *cfg) {
  if (cfg->n_eq != 5 && cfg->n_eq != 7) return fail("n_eq is 5, or 7 for rans");
  if ((cfg->n_eq == 7) != (cfg->equation_set == AGX_EQN_RANS))
    return fail("n_eq = 7 goes with equation_set rans and nothing else");
  if (cfg->n_eq == 7 && cfg->turbulence_model != AGX_TURB_SST2003 &&
      cfg->turbulence_model != AGX_TURB_KW_WILCOX2006 && cfg->turbulence_model != AGX_TURB_SST_DES)
    return fail("rans: the sst2003, sstdes and kOmegaWilcox2006 models are restated");
  if (cfg->n_eq == 7 && cfg->inv_flux_jacobian == AGX_JACOBIAN_APPROX_ROE)
    return fail("rans: approximateRoe is not restated");
  if (g_live_cfg > 0 && !c->have_cfg && cfg->n_eq != g_neq)
    return fail("oracle: one equation count per process at a time (%d live)", g_neq);
  if (!c->have_cfg) ++g_live_cfg;
  g_neq = cfg->n_eq;
  c->cfg = *cfg;
  c->have_cfg = 1;
  const agx_gas *g = &cfg->gas;
  /* thermodynamic::Cp/Cv/Gamma thermodynamic.cpp:60-81, thermodynamic.hpp:53-58 */
  c->cp = 0.0 + 1.0 * (g->gas_constant * (g->n + 1.0));
  c->cv = 0.0 + 1.0 * (g->gas_constant * g->n);
  c->gamma = c->cp / c->cv;
  c->prandtl = (4.0 * c->gamma) / (9.0 * c->gamma - 5.0);
  /* sutherland::sutherland transport.cpp:31-69 */
  c->mu_ref = g->visc_c1 * pow(g->t_ref, 1.5) / (g->t_ref + g->visc_s);
  c->k_nondim = (g->a_ref * g->a_ref * c->mu_ref) / g->t_ref;
  c->scaling = c->mu_ref / (g->rho_ref * g->a_ref * g->l_ref);
  return 0;
}

static double *dup_arr(const double *src, long n) {
  double *p = (double *)malloc(sizeof(double) * (n > 0 ? n : 1));
  if (src) memcpy(p, src, sizeof(double) * n);
  else memset(p, 0, sizeof(double) * n);
  return p;
}
int ora_block_create(ora_ctx *c, const agx_block_geom *g, int *id) {
  if (!c->have_cfg) return fail("config_set must precede block_create");
  if (c->nblk >= MAXBLK) return fail("too many blocks");
  if (g->ng != c->cfg.n_ghost) return fail("ghost layer mismatch");
  blk_t *b = &c->blk[c->nblk];
  memset(b, 0, sizeof *b);
  b->ni = g->ni; b->nj = g->nj; b->nk = g->nk; b->ng = g->ng;
  b->parent = g->parent_block; b->gpos = g->global_pos;
  b->ci = b->ni + 2 * b->ng; b->cj = b->nj + 2 * b->ng; b->ck = b->nk + 2 * b->ng;
  b->ncell = (long)b->ni * b->nj * b->nk;
  b->ncell_g = (long)b->ci * b->cj * b->ck;
  const long nf[3] = {(long)(b->ci + 1) * b->cj * b->ck,
                      (long)b->ci * (b->cj + 1) * b->ck,
                      (long)b->ci * b->cj * (b->ck + 1)};
  b->fa[0] = dup_arr(g->farea_i, 4 * nf[0]);
  b->fa[1] = dup_arr(g->farea_j, 4 * nf[1]);
  b->fa[2] = dup_arr(g->farea_k, 4 * nf[2]);
  b->vol = dup_arr(g->vol, b->ncell_g);
  b->center = dup_arr(g->center, 3 * b->ncell_g);
  b->wid[0] = dup_arr(g->width_i, b->ncell_g);
  b->wid[1] = dup_arr(g->width_j, b->ncell_g);
  b->wid[2] = dup_arr(g->width_k, b->ncell_g);
  b->wdist = dup_arr(g->wall_dist, b->ncell_g);
  b->state = dup_arr(NULL, NEQ * b->ncell_g);
  b->temp = dup_arr(NULL, b->ncell_g);
  b->visc = dup_arr(NULL, b->ncell_g);
  b->velgrad = dup_arr(NULL, 9 * b->ncell_g);
  b->grad18 = dup_arr(NULL, 18 * b->ncell);
  b->resid = dup_arr(NULL, NEQ * b->ncell);
  b->specrad = dup_arr(NULL, b->ncell);
  b->dt = dup_arr(NULL, b->ncell);
  b->consn = dup_arr(NULL, NEQ * b->ncell);
  b->consnm1 = dup_arr(NULL, NEQ * b->ncell);
  b->x = dup_arr(NULL, NEQ * b->ncell_g);
  b->xold = dup_arr(NULL, NEQ * b->ncell_g);
  b->a = dup_arr(NULL, b->ncell);
  b->ainv = dup_arr(NULL, b->ncell);
  b->am = dup_arr(NULL, NJ * b->ncell);
  b->aminv = dup_arr(NULL, NJ * b->ncell);
  b->specrad_t = dup_arr(NULL, b->ncell);
  b->a_t = dup_arr(NULL, b->ncell);
  b->ainv_t = dup_arr(NULL, b->ncell);
  b->turb3 = dup_arr(NULL, 3 * b->ncell_g);
  b->kgrad = dup_arr(NULL, 3 * b->ncell);
  b->wgrad = dup_arr(NULL, 3 * b->ncell);
  b->am_t = dup_arr(NULL, 2 * b->ncell);
  b->aminv_t = dup_arr(NULL, 2 * b->ncell);
  *id = c->nblk++;
  return 0;
}
int ora_block_set_bcs(ora_ctx *c, int id, int n, const agx_bc_surface *s) {
  if (id < 0 || id >= c->nblk) return fail("bad block id");
  blk_t *b = &c->blk[id];
  free(b->surf);
  b->surf = (agx_bc_surface *)malloc(sizeof *s * (n > 0 ? n : 1));
  memcpy(b->surf, s, sizeof *s * n);
  b->nsurf = n;
  b->nsurf_i = b->nsurf_j = b->nsurf_k = 0;
  for (int q = 0; q < n; ++q) {
    const int st = surf_type(&s[q]);
    if (st <= 2) b->nsurf_i++; else if (st <= 4) b->nsurf_j++; else b->nsurf_k++;
  }
  /* wallData_ (procBlock.hpp:99): one wallVars per face of every wall-law surface */
  free(b->wall_off); free(b->wallv);
  b->wall_off = (long *)malloc(sizeof(long) * (n > 0 ? n : 1));
  long total = 0;
  for (int q = 0; q < n; ++q) {
    b->wall_off[q] = -1;
    if (s[q].bc_type != AGX_BC_VISCOUSWALL || !s[q].state.is_wall_law) continue;
    const int d3 = (surf_type(&s[q]) - 1) / 2;
    int r1s, r1e, r2s, r2e, r3s;
    surf_ranges(&s[q], d3, &r1s, &r1e, &r2s, &r2e, &r3s);
    b->wall_off[q] = total;
    total += (long)(r1e - r1s) * (r2e - r2s);
  }
  b->wallv = (wall_vars *)calloc(total > 0 ? total : 1, sizeof(wall_vars));
  return 0;
}
int ora_conn_create(ora_ctx *c, const agx_connection *cc, int *id) {
  if (c->nconn >= MAXCONN) return fail("too many connections");
  conn_t *k = &c->conn[c->nconn];
  memset(k, 0, sizeof *k);
  k->c = *cc;
  *id = c->nconn++;
  return 0;
}
int ora_setup_finalize(ora_ctx *c) {
  for (int n = 0; n < c->nconn; ++n) {
    conn_t *k = &c->conn[n];
    if (k->c.rank[0] == c->rank && k->c.rank[1] == c->rank) {
      blk_t *b0 = &c->blk[k->c.local_block[0]], *b1 = &c->blk[k->c.local_block[1]];
      build_side_map(&k->c, 0, b0, b1, &k->dst[0], &k->src[0], &k->n[0]);
      build_side_map(&k->c, 1, b1, b0, &k->dst[1], &k->src[1], &k->n[1]);
    } else {
      /* remote partner: only dims of the partner are needed for src indices;
       * the partner is assumed to have the matching slab shape, so the map is
       * built against a slab-local numbering (see ora_halo_pack/unpack) */
      for (int s = 0; s < 2; ++s) {
        if (k->c.rank[s] != c->rank) continue;
        blk_t *me = &c->blk[k->c.local_block[s]];
        /* fake partner block holding just the slab: dims chosen so that the
         * slab [d3: ng cells] x [d1 len + 2ng] x [d2 len + 2ng] starts at the
         * indices build_side_map computes */
        blk_t slab;
        memset(&slab, 0, sizeof slab);
        slab.ng = me->ng;
        int d1, d2, d3;
        const int o = 1 - s;
        dirs_of(k->c.boundary[o], &d1, &d2, &d3);
        int dims[3];
        dims[d1] = k->c.d1_end[o];
        dims[d2] = k->c.d2_end[o];
        dims[d3] = k->c.const_surf[o] > 0 ? k->c.const_surf[o] : me->ng;
        slab.ni = dims[0]; slab.nj = dims[1]; slab.nk = dims[2];
        slab.ci = slab.ni + 2 * slab.ng; slab.cj = slab.nj + 2 * slab.ng;
        slab.ck = slab.nk + 2 * slab.ng;
        build_side_map(&k->c, s, me, &slab, &k->dst[s], &k->src[s], &k->n[s]);
        /* what we send: our cells the partner's ghost cells read.  Build the
         * partner's receive map against our real block to get those cells.  */
        blk_t pslab = slab;
        build_side_map(&k->c, o, &pslab, me, &k->dst[o], &k->src[o], &k->n[o]);
      }
    }
  }
  return 0;
}

int ora_state_upload(ora_ctx *c, int id, const double *s) {
  if (id < 0 || id >= c->nblk) return fail("bad block id");
  blk_t *b = &c->blk[id];
  memcpy(b->state, s, sizeof(double) * NEQ * b->ncell_g);
  /* gridLevel::AuxillaryAndWidths main.cpp:169: temperature_ / viscosity_ of the
   * physical cells before the first iteration (the rans wall ghost states read the
   * viscosity_ of the LAST UpdateAuxillaryVariables, procBlock.cpp:2814) */
  if (NEQ > NF)
    for (int k = 0; k < b->nk; ++k)
      for (int j = 0; j < b->nj; ++j)
        for (int i = 0; i < b->ni; ++i) {
          const long q = CI(b, i, j, k);
          b->temp[q] = temperature(c, b->state + NEQ * q);
          b->visc[q] = viscosity(c, b->temp[q]);
        }
  return 0;
}
static double *field_ptr(blk_t *b, int field, long *n) {
  switch (field) {
    case AGX_FIELD_STATE: *n = NEQ * b->ncell_g; return b->state;
    case AGX_FIELD_RESIDUAL: *n = NEQ * b->ncell; return b->resid;
    case AGX_FIELD_DT: *n = b->ncell; return b->dt;
    case AGX_FIELD_SPEC_RADIUS: *n = b->ncell; return b->specrad;
    case AGX_FIELD_CONS_N: *n = NEQ * b->ncell; return b->consn;
    case AGX_FIELD_UPDATE: *n = NEQ * b->ncell_g; return b->x;
    case AGX_FIELD_DIAGONAL: *n = b->ncell; return b->a;
    case AGX_FIELD_TEMPERATURE: *n = b->ncell_g; return b->temp;
    case AGX_FIELD_VISCOSITY: *n = b->ncell_g; return b->visc;
    case AGX_FIELD_CONS_NM1: *n = NEQ * b->ncell; return b->consnm1;
  }
  return NULL;
}
/* The gradients an output step asks for reach into the ghost cells: they are formed with
 * the ghost cells the NEXT residual would see -- the inviscid fill of the state as it is
 * now, then the viscous-wall fill (gridLevel.cpp:287-319, procBlock.cpp:6131-6136); ghost
 * cells of connections to other ranks stay as last exchanged. */
int ora_phase_bc_faces(ora_ctx *c);
int ora_phase_bc_edges(ora_ctx *c);
int ora_halo_swap_local(ora_ctx *c, int what);
static int ghosts_for_output(ora_ctx *c) {
  if (ora_phase_bc_faces(c)) return 1;
  if (ora_halo_swap_local(c, AGX_HALO_STATE)) return 1;
  if (ora_phase_bc_edges(c)) return 1;
  if (c->cfg.is_viscous)
    for (int n = 0; n < c->nblk; ++n) {
      if (assign_ghost_faces(c, &c->blk[n], 1)) return 1;
      if (assign_ghost_edges(c, &c->blk[n], 1)) return 1;
    }
  return 0;
}
int ora_field_download(ora_ctx *c, int id, int field, double *out) {
  if (id < 0 || id >= c->nblk) return fail("bad block id");
  if (field >= AGX_FIELD_VEL_GRAD && field <= AGX_FIELD_PRESS_GRAD) {
    blk_t *b = &c->blk[id];
    if (ghosts_for_output(c)) return 1;
    double *all = (double *)malloc(sizeof(double) * 18 * b->ncell);
    cell_gradients(c, b, all);
    const int off = field == AGX_FIELD_VEL_GRAD ? 0 : 9 + 3 * (field - AGX_FIELD_TEMP_GRAD);
    const int nc = field == AGX_FIELD_VEL_GRAD ? 9 : 3;
    for (long q = 0; q < b->ncell; ++q) memcpy(out + nc * q, all + 18 * q + off, sizeof(double) * nc);
    free(all);
    return 0;
  }
  long n;
  double *p = field_ptr(&c->blk[id], field, &n);
  if (!p) return fail("unknown field %d", field);
  memcpy(out, p, sizeof(double) * n);
  return 0;
}
/* WriteFunFile output.cpp:209-437: variables of one block, variable by variable, physical
 * cells, dimensional (the scales of :235-407) */
int ora_output_pack(ora_ctx *c, int id, int nvar, const int32_t *vars, double *out) {
  if (id < 0 || id >= c->nblk) return fail("bad block id");
  blk_t *b = &c->blk[id];
  const agx_gas *gs = &c->cfg.gas;
  const double rR = gs->rho_ref, aR = gs->a_ref, lR = gs->l_ref, tR = gs->t_ref, muR = c->mu_ref;
  const int nf = 6 + (NEQ - NF);
  double *gr = NULL;
  for (int v = 0; v < nvar; ++v) {
    if (vars[v] < 0 || vars[v] >= AGX_OUT_COUNT) return fail("unknown output variable %d", vars[v]);
    if (vars[v] == AGX_OUT_VISCOSITY && !c->cfg.is_viscous)
      return fail("viscosity_ is only kept for viscous runs (procBlock.cpp:6171)");
    if (vars[v] >= AGX_OUT_VELGRAD && vars[v] < AGX_OUT_RESID && !gr) {
      if (ghosts_for_output(c)) return 1;
      gr = (double *)malloc(sizeof(double) * 3 * nf * b->ncell);
      cell_gradients_n(c, b, nf, gr);
    }
  }
  for (int v = 0; v < nvar; ++v) {
    const int var = vars[v];
    double *o = out + (long)v * b->ncell;
    for (int k = 0; k < b->nk; ++k)
      for (int j = 0; j < b->nj; ++j)
        for (int i = 0; i < b->ni; ++i) {
          const long p = PI(b, i, j, k), q = CI(b, i, j, k);
          const double *s = b->state + NEQ * q;
          const double t = temperature(c, s);
          const double vmag = sqrt(dot3(s + 1, s + 1));
          double val = 0.0;
          switch (var) {
            case AGX_OUT_DENSITY: val = s[0] * rR; break;
            case AGX_OUT_VEL_X: val = s[1] * aR; break;
            case AGX_OUT_VEL_Y: val = s[2] * aR; break;
            case AGX_OUT_VEL_Z: val = s[3] * aR; break;
            case AGX_OUT_PRESSURE: val = s[4] * rR * aR * aR; break;
            case AGX_OUT_MACH: val = vmag / sos(c, s); break;
            case AGX_OUT_SOS: val = sos(c, s) * aR; break;
            case AGX_OUT_DT: val = b->dt[p] / (aR * lR); break;
            case AGX_OUT_TEMPERATURE: val = t * tR; break;
            case AGX_OUT_ENERGY: val = energy(c, s) * aR * aR; break;
            case AGX_OUT_ENTHALPY: val = enthalpy(c, s) * aR * aR; break;
            case AGX_OUT_CP: val = gs->gas_constant * (gs->n + 1.0) * aR * aR / tR; break;
            case AGX_OUT_CV: val = gs->gas_constant * gs->n * aR * aR / tR; break;
            case AGX_OUT_RANK: val = (double)c->rank; break;
            case AGX_OUT_GLOBAL_POSITION: val = (double)b->gpos; break;
            case AGX_OUT_VISCOSITY_RATIO:
              val = NEQ > NF ? b->turb3[3 * q] / viscosity(c, t) : 0.0; break;
            case AGX_OUT_TURB_VISCOSITY: val = NEQ > NF ? b->turb3[3 * q] * muR : 0.0; break;
            case AGX_OUT_VISCOSITY: val = viscosity(c, t) * muR; break;
            case AGX_OUT_TKE: val = NEQ > NF ? s[5] * aR * aR : 0.0; break;
            case AGX_OUT_SDR: val = NEQ > NF ? s[6] * aR * aR * rR / muR : 0.0; break;
            case AGX_OUT_F1: val = NEQ > NF ? b->turb3[3 * q + 1] : 0.0; break;
            case AGX_OUT_F2: val = NEQ > NF ? b->turb3[3 * q + 2] : 0.0; break;
            case AGX_OUT_WALL_DISTANCE: val = (b->wdist ? b->wdist[q] : 0.0) * lR; break;
            default:
              if (var >= AGX_OUT_RESID) {
                const int e = var - AGX_OUT_RESID;
                const double sc[7] = {rR * aR * lR * lR, rR * aR * aR * lR * lR,
                                      rR * aR * aR * lR * lR, rR * aR * aR * lR * lR,
                                      rR * pow(aR, 3.0) * lR * lR, rR * pow(aR, 3.0) * lR * lR,
                                      rR * rR * pow(aR, 4.0) * lR * lR / muR};
                val = e < NEQ ? b->resid[NEQ * p + e] * sc[e] : 0.0;
              } else {
                const int gidx = var - AGX_OUT_VELGRAD;      /* 0 .. 23 */
                const double sc = gidx < 9 ? aR / lR
                                  : gidx < 12 ? tR / lR
                                  : gidx < 15 ? rR / lR
                                  : gidx < 18 ? rR * aR * aR / lR
                                  : gidx < 21 ? aR * aR / lR
                                              : aR * aR * rR / (muR * lR);
                val = gidx < 3 * nf ? gr[3 * nf * p + gidx] * sc : 0.0;
              }
          }
          o[p] = val;
        }
  }
  free(gr);
  return 0;
}
/* WriteRestart output.cpp:651-752, the payload of one block */
int ora_restart_pack(ora_ctx *c, int id, int which, double *out) {
  if (id < 0 || id >= c->nblk) return fail("bad block id");
  if (which != 0 && which != 1) return fail("which is 0 (state) or 1 (consVarsNm1)");
  blk_t *b = &c->blk[id];
  const agx_gas *gs = &c->cfg.gas;
  const double rR = gs->rho_ref, aR = gs->a_ref, muR = c->mu_ref;
  const double sp[7] = {rR, aR, aR, aR, rR * aR * aR, aR * aR, aR * aR * rR / muR};
  const double sc[7] = {rR, aR * rR, aR * rR, aR * rR, aR * aR * rR, aR * aR * rR,
                        aR * aR * rR * rR / muR};
  const int nv = NEQ + 1;
  for (int k = 0; k < b->nk; ++k)
    for (int j = 0; j < b->nj; ++j)
      for (int i = 0; i < b->ni; ++i) {
        const long p = PI(b, i, j, k), q = CI(b, i, j, k);
        double *o = out + nv * p;
        for (int e = 0; e < NEQ; ++e)
          o[e] = which == 0 ? b->state[NEQ * q + e] * sp[e] : b->consnm1[NEQ * p + e] * sc[e];
        o[NEQ] = 1.0;        /* mass fraction of the single species */
      }
  return 0;
}
/* ---- set-up: plot3dBlock metrics (plot3d.cpp:35-360) ------------------------ */
static void v_sub(const double *a, const double *b, double *o) {
  o[0] = a[0] - b[0]; o[1] = a[1] - b[1]; o[2] = a[2] - b[2];
}
static void v_cross(const double *a, const double *b, double *o) {   /* vector3d.hpp:330-339 */
  o[0] = a[1] * b[2] - a[2] * b[1];
  o[1] = -1.0 * (a[0] * b[2] - a[2] * b[0]);
  o[2] = a[0] * b[1] - a[1] * b[0];
}
static double pyramid_volume(const double *p, const double *a, const double *b,
                             const double *cc, const double *d) {   /* plot3d.cpp:490-498 */
  double xp[3], xac[3], xbd[3], cr[3], t[3];
  for (int q = 0; q < 3; ++q) {
    t[q] = (a[q] - p[q]) + (b[q] - p[q]);
    t[q] = t[q] + (cc[q] - p[q]);
    t[q] = t[q] + (d[q] - p[q]);
    xp[q] = 0.25 * t[q];
  }
  v_sub(cc, a, xac);
  v_sub(d, b, xbd);
  v_cross(xac, xbd, cr);
  return 1.0 / 6.0 * dot3(xp, cr);
}
static void face_area_center(const double *n00, const double *n10, const double *n01,
                             const double *n11, const double *xac, const double *xbd,
                             double *fa, double *fc) {
  double cr[3];
  v_cross(xbd, xac, cr);
  const double h[3] = {0.5 * cr[0], 0.5 * cr[1], 0.5 * cr[2]};
  const double mag = sqrt(dot3(h, h));
  if (fa) { fa[0] = h[0] / mag; fa[1] = h[1] / mag; fa[2] = h[2] / mag; fa[3] = mag; }
  if (fc)
    for (int q = 0; q < 3; ++q) fc[q] = 0.25 * (((n00[q] + n10[q]) + n01[q]) + n11[q]);
}
int ora_plot3d_metrics(ora_ctx *c, int ni, int nj, int nk, const double *x, double *vol,
                       double *center, double *fai, double *faj, double *fak, double *fci,
                       double *fcj, double *fck) {
  (void)c;
  if (ni < 1 || nj < 1 || nk < 1) return fail("empty block");
#define ND(i, j, k) (x + 3 * (((long)(k) * (nj + 1) + (j)) * (ni + 1) + (i)))
  for (int k = 0; k <= nk; ++k)
    for (int j = 0; j <= nj; ++j)
      for (int i = 0; i <= ni; ++i) {
        if (i < ni && j < nj && k < nk && (vol || center)) {
          const double *c000 = ND(i, j, k), *c100 = ND(i + 1, j, k), *c010 = ND(i, j + 1, k),
                       *c110 = ND(i + 1, j + 1, k), *c001 = ND(i, j, k + 1),
                       *c101 = ND(i + 1, j, k + 1), *c011 = ND(i, j + 1, k + 1),
                       *c111 = ND(i + 1, j + 1, k + 1);
          double cen[3];
          for (int q = 0; q < 3; ++q)      /* plot3d.cpp:35-43 */
            cen[q] = 0.125 * (((((((c000[q] + c100[q]) + c010[q]) + c110[q]) + c001[q]) +
                                c101[q]) + c011[q]) + c111[q]);
          const long p = ((long)k * nj + j) * ni + i;
          if (center) memcpy(center + 3 * p, cen, sizeof cen);
          if (vol) {                        /* plot3d.cpp:60-113, the same pyramid order */
            double v = pyramid_volume(cen, c000, c001, c011, c010);
            v = v + pyramid_volume(cen, c100, c110, c111, c101);
            v = v + pyramid_volume(cen, c000, c100, c101, c001);
            v = v + pyramid_volume(cen, c010, c011, c111, c110);
            v = v + pyramid_volume(cen, c000, c010, c110, c100);
            v = v + pyramid_volume(cen, c001, c101, c111, c011);
            if (v <= 0.0) return fail("negative volume in PLOT3D block");
            vol[p] = v;
          }
        }
        double xac[3], xbd[3];
        if (j < nj && k < nk && (fai || fci)) {      /* i-faces, plot3d.cpp:150-181 */
          const double *n00 = ND(i, j, k), *n10 = ND(i, j + 1, k), *n01 = ND(i, j, k + 1),
                       *n11 = ND(i, j + 1, k + 1);
          v_sub(n11, n00, xac); v_sub(n10, n01, xbd);
          const long f = ((long)k * nj + j) * (ni + 1) + i;
          face_area_center(n00, n10, n01, n11, xac, xbd, fai ? fai + 4 * f : NULL,
                           fci ? fci + 3 * f : NULL);
        }
        if (i < ni && k < nk && (faj || fcj)) {      /* j-faces, plot3d.cpp:224-255 */
          const double *n00 = ND(i, j, k), *n10 = ND(i + 1, j, k), *n01 = ND(i, j, k + 1),
                       *n11 = ND(i + 1, j, k + 1);
          v_sub(n01, n10, xac); v_sub(n00, n11, xbd);
          const long f = ((long)k * (nj + 1) + j) * ni + i;
          face_area_center(n00, n10, n01, n11, xac, xbd, faj ? faj + 4 * f : NULL,
                           fcj ? fcj + 3 * f : NULL);
        }
        if (i < ni && j < nj && (fak || fck)) {      /* k-faces, plot3d.cpp:300-331 */
          const double *n00 = ND(i, j, k), *n10 = ND(i + 1, j, k), *n01 = ND(i, j + 1, k),
                       *n11 = ND(i + 1, j + 1, k);
          v_sub(n01, n10, xac); v_sub(n11, n00, xbd);
          const long f = ((long)k * nj + j) * ni + i;
          face_area_center(n00, n10, n01, n11, xac, xbd, fak ? fak + 4 * f : NULL,
                           fck ? fck + 3 * f : NULL);
        }
      }
#undef ND
  return 0;
}
/* kdtree::NearestNeighbor (kdtree.cpp:123-225) as CalcWallDistance uses it: the distance
 * to the nearest point of the set -- here by exhaustive search (what the tree computes) */
int ora_nearest_wall_distance(ora_ctx *c, int64_t ncell, const double *cen, int64_t nwall,
                              const double *wall, double *dist) {
  (void)c;
  if (nwall < 1) return fail("no wall points");
#pragma omp parallel for schedule(static) if (ncell * nwall > 4000000)
  for (int64_t q = 0; q < ncell; ++q) {
    double best = 1.7976931348623157e308;
    for (int64_t p = 0; p < nwall; ++p) {
      const double d[3] = {cen[3 * q] - wall[3 * p], cen[3 * q + 1] - wall[3 * p + 1],
                           cen[3 * q + 2] - wall[3 * p + 2]};
      const double d2 = dot3(d, d);
      if (d2 < best) best = d2;
    }
    dist[q] = sqrt(best);
  }
  return 0;
}
int ora_field_upload(ora_ctx *c, int id, int field, const double *in) {
  if (id < 0 || id >= c->nblk) return fail("bad block id");
  long n;
  double *p = field_ptr(&c->blk[id], field, &n);
  if (!p) return fail("unknown field %d", field);
  memcpy(p, in, sizeof(double) * n);
  return 0;
}

/* procBlock::AssignSolToTimeN / AssignSolToTimeNm1 procBlock.cpp:1037-1054 */
/* ---- geometric multigrid ---------------------------------------------------------------- */
static int mg_blocks(ora_ctx *f, ora_ctx *cz, int blk, blk_t **bf, blk_t **bc) {
  if (!f || !cz || blk < 0 || blk >= f->nblk || blk >= cz->nblk) return fail("mg: bad block %d", blk);
  *bf = &f->blk[blk];
  *bc = &cz->blk[blk];
  return 0;
}
/* BlockRestriction procBlock.hpp:636-690: the coarse array zeroed, then every fine cell in
 * k, j, i order added to its coarse cell */
int ora_mg_restrict(ora_ctx *f, ora_ctx *cz, int blk, int what, const int32_t *tc,
                    const double *vf) {
  blk_t *bf, *bc;
  if (mg_blocks(f, cz, blk, &bf, &bc)) return 1;
  if (what == AGX_MG_STATE) {
    if (!vf) return fail("mg_restrict: volume weights missing");
    memset(bc->state, 0, sizeof(double) * NEQ * bc->ncell_g);
    for (int k = 0; k < bf->nk; ++k)
      for (int j = 0; j < bf->nj; ++j)
        for (int i = 0; i < bf->ni; ++i) {
          const long p = PI(bf, i, j, k);
          const long qc = CI(bc, tc[3 * p], tc[3 * p + 1], tc[3 * p + 2]);
          const long qf = CI(bf, i, j, k);
          for (int e = 0; e < NEQ; ++e)
            bc->state[NEQ * qc + e] = bc->state[NEQ * qc + e] + vf[p] * bf->state[NEQ * qf + e];
        }
    return 0;
  }
  if (what == AGX_MG_UPDATE) {
    if (!vf) return fail("mg_restrict: volume weights missing");
    memset(bc->x, 0, sizeof(double) * NEQ * bc->ncell_g);
    for (int k = 0; k < bf->nk; ++k)
      for (int j = 0; j < bf->nj; ++j)
        for (int i = 0; i < bf->ni; ++i) {
          const long p = PI(bf, i, j, k);
          const long qc = CI(bc, tc[3 * p], tc[3 * p + 1], tc[3 * p + 2]);
          const long qf = CI(bf, i, j, k);
          for (int e = 0; e < NEQ; ++e)
            bc->x[NEQ * qc + e] = bc->x[NEQ * qc + e] + vf[p] * bf->x[NEQ * qf + e];
        }
    return 0;
  }
  if (what != AGX_MG_FORCING) return fail("mg_restrict: bad selector %d", what);
  if (!bf->mres) return fail("mg_restrict: the fine level has no matrix residual yet");
  if (!bc->forcing) bc->forcing = (double *)calloc((size_t)NEQ * bc->ncell, sizeof(double));
  double *fo = bc->forcing;
  memset(fo, 0, sizeof(double) * NEQ * bc->ncell);
  for (int k = 0; k < bf->nk; ++k)
    for (int j = 0; j < bf->nj; ++j)
      for (int i = 0; i < bf->ni; ++i) {
        const long p = PI(bf, i, j, k);
        const long pc = PI(bc, tc[3 * p], tc[3 * p + 1], tc[3 * p + 2]);
        for (int e = 0; e < NEQ; ++e) fo[NEQ * pc + e] = fo[NEQ * pc + e] + bf->mres[NEQ * p + e];
      }
  /* + A x - b of the coarse level (linearSolver::AXmB :58-90; gridLevel.cpp:579-589) */
  for (int k = 0; k < bc->nk; ++k)
    for (int j = 0; j < bc->nj; ++j)
      for (int i = 0; i < bc->ni; ++i) {
        double off[NEQM], U[NEQM], rb[NEQM], ax[NEQM];
        implicit_lower(cz, bc, i, j, k, bc->x, off);
        implicit_upper(cz, bc, i, j, k, bc->x, U);
        for (int e = 0; e < NEQ; ++e) off[e] -= U[e];
        rhs_b(cz, bc, i, j, k, rb);
        const long p = PI(bc, i, j, k), q = CI(bc, i, j, k);
        if (is_block(cz)) {
          mat_vec(bc->am + NJ * p, bc->x + NEQ * q, ax);
          for (int e = NF; e < NEQ; ++e) ax[e] = bc->am_t[2 * p + e - NF] * bc->x[NEQ * q + e];
        } else {
          for (int e = 0; e < NF; ++e) ax[e] = bc->x[NEQ * q + e] * bc->a[p];
          for (int e = NF; e < NEQ; ++e) ax[e] = bc->x[NEQ * q + e] * bc->a_t[p];
        }
        for (int e = 0; e < NEQ; ++e) fo[NEQ * p + e] = (ax[e] - off[e] - rb[e]) + fo[NEQ * p + e];
      }
  return 0;
}
int ora_mg_matrix_residual(ora_ctx *c, double *mr) {
  double sumsq = 0.0;
  long size = 0;
  c->mres_opsq = 0.0;
  for (int n = 0; n < c->nblk; ++n) {
    blk_t *b = &c->blk[n];
    if (!b->mres) b->mres = (double *)calloc((size_t)NEQ * b->ncell, sizeof(double));
    matrix_residual(c, b, &sumsq, &size);
  }
  c->mres_sumsq = sumsq;
  *mr = size > 0 ? sumsq / (double)size : 0.0;
  return 0;
}
int ora_mg_invert_diagonal(ora_ctx *c) {
  for (int n = 0; n < c->nblk; ++n)
    if (implicit_begin_x(c, &c->blk[n], 0)) return 1;
  return 0;
}
/* gridLevel::ResetDiagonal gridLevel.cpp:408-412 (mgSolution::ImplicitUpdate resets every
 * level when the iteration ends, mgSolution.cpp:236-239; the finest one inside
 * ora_phase_implicit_update) */
int ora_mg_reset_diagonal(ora_ctx *c) {
  for (int n = 0; n < c->nblk; ++n) {
    blk_t *b = &c->blk[n];
    memset(b->a, 0, sizeof(double) * b->ncell);
    memset(b->a_t, 0, sizeof(double) * b->ncell);
    memset(b->am, 0, sizeof(double) * NJ * b->ncell);
    memset(b->am_t, 0, sizeof(double) * 2 * b->ncell);
  }
  return 0;
}
int ora_mg_save_update(ora_ctx *c) {
  for (int n = 0; n < c->nblk; ++n) {
    blk_t *b = &c->blk[n];
    if (!b->xsave) b->xsave = (double *)calloc((size_t)NEQ * b->ncell_g, sizeof(double));
    memcpy(b->xsave, b->x, sizeof(double) * NEQ * b->ncell_g);
  }
  return 0;
}
/* SubtractFromUpdate (linearSolver.cpp:120-126), BlockProlongation (gridLevel.hpp:159-214)
 * with ConvertCellToNode(coarse, ignoreEdge, ignoreGhosts) (utility.hpp:186-330),
 * AddToUpdate (linearSolver.cpp:128-134) */
int ora_mg_prolong(ora_ctx *cz, ora_ctx *f, int blk, const int32_t *tc, const double *cf) {
  blk_t *bf, *bc;
  if (mg_blocks(f, cz, blk, &bf, &bc)) return 1;
  if (!bc->xsave) return fail("mg_prolong: no saved update on the coarse level");
  for (long q = 0; q < NEQ * bc->ncell_g; ++q) bc->x[q] -= bc->xsave[q];
  const int ni = bc->ni, nj = bc->nj, nk = bc->nk;
  const long nn = (long)(ni + 1) * (nj + 1) * (nk + 1);
  double *nd = (double *)calloc((size_t)NEQ * nn, sizeof(double));
#define ND(i, j, k) (nd + NEQ * (((long)(k) * (nj + 1) + (j)) * (ni + 1) + (i)))
  for (int k = 0; k < nk; ++k)
    for (int j = 0; j < nj; ++j)
      for (int i = 0; i < ni; ++i) {
        const double *v = bc->x + NEQ * CI(bc, i, j, k);
        /* the order of utility.hpp:286-311 */
        static const int o[8][3] = {{0, 0, 0}, {0, 1, 0}, {0, 1, 1}, {0, 0, 1},
                                    {1, 0, 0}, {1, 1, 0}, {1, 1, 1}, {1, 0, 1}};
        for (int m = 0; m < 8; ++m) {
          double *t = ND(i + o[m][0], j + o[m][1], k + o[m][2]);
          for (int e = 0; e < NEQ; ++e) t[e] += v[e];
        }
      }
  for (int k = 0; k <= nk; ++k)
    for (int j = 0; j <= nj; ++j)
      for (int i = 0; i <= ni; ++i) {
        const int xi = i == 0 || i == ni, xj = j == 0 || j == nj, xk = k == 0 || k == nk;
        /* AtInteriorCorner / AtInteriorEdge of the node array (multiArray3d.hpp:1595-1683);
         * no ghost cells: corner 1, edge 1/2, the rest 1/8 */
        const double fac = (xi && xj && xk) ? 1.0 : ((xj && xk) || (xi && xk) || (xi && xj)) ? 0.5 : 0.125;
        double *t = ND(i, j, k);
        for (int e = 0; e < NEQ; ++e) t[e] *= fac;
      }
  for (int k = 0; k < bf->nk; ++k)
    for (int j = 0; j < bf->nj; ++j)
      for (int i = 0; i < bf->ni; ++i) {
        const long p = PI(bf, i, j, k);
        const int ci = tc[3 * p], cj = tc[3 * p + 1], ck = tc[3 * p + 2];
        const double *w = cf + 7 * p;
        const double *d0 = ND(ci, cj, ck), *d1 = ND(ci + 1, cj, ck), *d2 = ND(ci, cj + 1, ck),
                     *d3 = ND(ci + 1, cj + 1, ck), *d4 = ND(ci, cj, ck + 1),
                     *d5 = ND(ci + 1, cj, ck + 1), *d6 = ND(ci, cj + 1, ck + 1),
                     *d7 = ND(ci + 1, cj + 1, ck + 1);
        double *xf = bf->x + NEQ * CI(bf, i, j, k);
        for (int e = 0; e < NEQ; ++e) {
          /* TrilinearInterp utility.hpp:356-372, LinearInterp :341-344 */
          const double d04 = (1.0 - w[0]) * d0[e] + w[0] * d4[e];
          const double d15 = (1.0 - w[1]) * d1[e] + w[1] * d5[e];
          const double d26 = (1.0 - w[2]) * d2[e] + w[2] * d6[e];
          const double d37 = (1.0 - w[3]) * d3[e] + w[3] * d7[e];
          const double d0415 = (1.0 - w[4]) * d04 + w[4] * d15;
          const double d2637 = (1.0 - w[5]) * d26 + w[5] * d37;
          xf[e] += (1.0 - w[6]) * d0415 + w[6] * d2637;
        }
      }
#undef ND
  free(nd);
  return 0;
}

int ora_store_time_n(ora_ctx *c, int also_nm1) {
  c->have_time_n = 1;
  for (int n = 0; n < c->nblk; ++n) {
    blk_t *b = &c->blk[n];
    for (int k = 0; k < b->nk; ++k)
      for (int j = 0; j < b->nj; ++j)
        for (int i = 0; i < b->ni; ++i)
          prim_to_cons(c, b->state + NEQ * CI(b, i, j, k),
                       b->consn + NEQ * PI(b, i, j, k));
    if (also_nm1)
      memcpy(b->consnm1, b->consn, sizeof(double) * NEQ * b->ncell);
  }
  return 0;
}

int ora_phase_bc_faces(ora_ctx *c) {
  for (int n = 0; n < c->nblk; ++n)
    if (assign_ghost_faces(c, &c->blk[n], 0)) return 1;
  return 0;
}
int ora_phase_bc_edges(ora_ctx *c) {
  for (int n = 0; n < c->nblk; ++n)
    if (assign_ghost_edges(c, &c->blk[n], 0)) return 1;
  return 0;
}
int ora_phase_residual(ora_ctx *c, int mm, double cfl) {
  (void)mm;
  for (int n = 0; n < c->nblk; ++n) {
    if (calc_residual(c, &c->blk[n])) return 1;
  }
  for (int n = 0; n < c->nblk; ++n)
    if (calc_dt(c, &c->blk[n], cfl)) return 1;
  return 0;
}
int ora_phase_explicit_update(ora_ctx *c, int mm, double *l2, agx_linf *linf) {
  for (int n = 0; n < c->nblk; ++n) update_block(c, &c->blk[n], mm, l2, linf);
  return 0;
}
int ora_phase_implicit_begin(ora_ctx *c) {
  for (int n = 0; n < c->nblk; ++n)
    if (implicit_begin(c, &c->blk[n])) return 1;
  return 0;
}
int ora_phase_relax_forward(ora_ctx *c, int sweep) {
  for (int n = 0; n < c->nblk; ++n) {
    if (is_lusgs(c)) lusgs_forward(c, &c->blk[n], sweep);
    else dplur_sweep(c, &c->blk[n]);
  }
  return 0;
}
int ora_phase_relax_backward(ora_ctx *c, int sweep) {
  if (!is_lusgs(c)) return 0;
  for (int n = 0; n < c->nblk; ++n) lusgs_backward(c, &c->blk[n], sweep);
  return 0;
}
/* test hook: sum of the squares of what the last matrix residual was the difference of (A x,
 * the off-diagonal terms, b), for the cancellation factor the parity tests derive their
 * tolerance of the matrix residual from */
int ora_debug_matrix_operands(ora_ctx *c, double *opsq, double *ressq) {
  *opsq = c->mres_opsq;
  *ressq = c->mres_sumsq;
  return 0;
}
int ora_phase_matrix_residual(ora_ctx *c, double *mr) {
  c->mres_opsq = 0.0;
  double sumsq = 0.0;
  long size = 0;
  for (int n = 0; n < c->nblk; ++n) matrix_residual(c, &c->blk[n], &sumsq, &size);
  c->mres_sumsq = sumsq;
  *mr = size > 0 ? sumsq / (double)size : 0.0;
  return 0;
}
int ora_phase_implicit_update(ora_ctx *c, int mm, double *l2, agx_linf *linf) {
  for (int n = 0; n < c->nblk; ++n) {
    blk_t *b = &c->blk[n];
    update_block(c, b, mm, l2, linf);
    /* gridLevel::UpdateBlocks gridLevel.cpp:425-428 */
    if (c->cfg.time_integration == AGX_TIME_BDF2 &&
        mm == c->cfg.nonlinear_iterations - 1)
      memcpy(b->consnm1, b->consn, sizeof(double) * NEQ * b->ncell);
    /* gridLevel::ResetDiagonal gridLevel.cpp:408-412 */
    memset(b->a, 0, sizeof(double) * b->ncell);
    memset(b->a_t, 0, sizeof(double) * b->ncell);
    memset(b->am, 0, sizeof(double) * NJ * b->ncell);
    memset(b->am_t, 0, sizeof(double) * 2 * b->ncell);
  }
  return 0;
}

int ora_halo_swap_local(ora_ctx *c, int what) {
  for (int n = 0; n < c->nconn; ++n) {
    conn_t *k = &c->conn[n];
    if (!(k->c.rank[0] == c->rank && k->c.rank[1] == c->rank)) continue;
    blk_t *b0 = &c->blk[k->c.local_block[0]], *b1 = &c->blk[k->c.local_block[1]];
    const int nc = NEQ;
    const halo_view a0 = halo_array(b0, what), a1 = halo_array(b1, what);
    /* both slices are taken before either insert (multiArray3d.hpp:810-821) */
    double *s1 = (double *)calloc(nc * (k->n[0] > 0 ? k->n[0] : 1), sizeof(double));
    double *s0 = (double *)calloc(nc * (k->n[1] > 0 ? k->n[1] : 1), sizeof(double));
    for (long q = 0; q < k->n[0]; ++q) halo_get(&a1, k->src[0][q], s1 + nc * q);
    for (long q = 0; q < k->n[1]; ++q) halo_get(&a0, k->src[1][q], s0 + nc * q);
    for (long q = 0; q < k->n[0]; ++q) halo_put(&a0, k->dst[0][q], s1 + nc * q);
    for (long q = 0; q < k->n[1]; ++q) halo_put(&a1, k->dst[1][q], s0 + nc * q);
    free(s0);
    free(s1);
  }
  return 0;
}
static int my_side(const ora_ctx *c, const conn_t *k) {
  if (k->c.rank[0] == c->rank && k->c.rank[1] != c->rank) return 0;
  if (k->c.rank[1] == c->rank && k->c.rank[0] != c->rank) return 1;
  return -1;
}
int64_t ora_halo_count(ora_ctx *c, int id, int what) {
  (void)what;
  if (id < 0 || id >= c->nconn) return -1;
  const conn_t *k = &c->conn[id];
  const int s = my_side(c, k);
  if (s < 0) return 0;
  /* we send the cells the partner inserts: n[1-s]; we receive n[s] */
  return (int64_t)NEQ * (k->n[1 - s] > k->n[s] ? k->n[1 - s] : k->n[s]);
}
int ora_halo_pack(ora_ctx *c, int id, int what, double *buf) {
  if (id < 0 || id >= c->nconn) return fail("bad connection id");
  conn_t *k = &c->conn[id];
  const int s = my_side(c, k);
  if (s < 0) return fail("connection %d is not remote", id);
  blk_t *b = &c->blk[k->c.local_block[s]];
  const halo_view a = halo_array(b, what);
  for (long q = 0; q < k->n[1 - s]; ++q) halo_get(&a, k->src[1 - s][q], buf + NEQ * q);
  return 0;
}
int ora_halo_unpack(ora_ctx *c, int id, int what, const double *buf) {
  if (id < 0 || id >= c->nconn) return fail("bad connection id");
  conn_t *k = &c->conn[id];
  const int s = my_side(c, k);
  if (s < 0) return fail("connection %d is not remote", id);
  blk_t *b = &c->blk[k->c.local_block[s]];
  const halo_view a = halo_array(b, what);
  for (long q = 0; q < k->n[s]; ++q) halo_put(&a, k->dst[s][q], buf + NEQ * q);
  return 0;
}

/* ---- multi-rank: the same exchange table as the product library (always host
 * buffers here), see include/aither_gfx950.h ---------------------------------- */
typedef struct { double l2[8]; double mres, linf; int32_t block, i, j, k, eqn, pad;
                 double fill[3]; } norm_record;     /* 128 bytes, as in agx_api.hip */
int ora_set_exchange(ora_ctx *c, const agx_exchange *ex) {
  if (!ex || !ex->swap || !ex->allgather || ex->nranks < 1)
    return fail("set_exchange: swap, allgather and nranks are required");
  c->ex = *ex;
  c->have_ex = 1;
  return 0;
}
int ora_rccl_unique_id(void *id128) { (void)id128; return fail("the CPU oracle has no RCCL"); }
int ora_rccl_exchange_create(ora_ctx *c, const void *id, int n, int r) {
  (void)c; (void)id; (void)n; (void)r;
  return fail("the CPU oracle has no RCCL");
}
int ora_halo_exchange(ora_ctx *c, int what) {
  if (ora_halo_swap_local(c, what)) return 1;
  int nrem = 0;
  for (int n = 0; n < c->nconn; ++n) nrem += my_side(c, &c->conn[n]) >= 0;
  if (!nrem) return 0;
  if (!c->have_ex) return fail("connections to other ranks need an exchange");
  agx_slab *sl = (agx_slab *)malloc(sizeof(agx_slab) * nrem);
  int *cid = (int *)malloc(sizeof(int) * nrem);
  int q = 0;
  for (int n = 0; n < c->nconn; ++n) {
    const int s = my_side(c, &c->conn[n]);
    if (s < 0) continue;
    sl[q].peer = c->conn[n].c.rank[1 - s];
    sl[q].tag = 0;      /* ordinal among the connections with this peer */
    for (int m = 0; m < n; ++m) {
      const int sm = my_side(c, &c->conn[m]);
      if (sm >= 0 && c->conn[m].c.rank[1 - sm] == sl[q].peer) sl[q].tag++;
    }
    cid[q] = n;
    sl[q].count = ora_halo_count(c, n, what);
    sl[q].send = (double *)calloc(sl[q].count > 0 ? sl[q].count : 1, sizeof(double));
    sl[q].recv = (double *)calloc(sl[q].count > 0 ? sl[q].count : 1, sizeof(double));
    ora_halo_pack(c, n, what, sl[q].send);
    ++q;
  }
  const int rc = c->ex.swap(c->ex.user, nrem, sl, NULL);
  for (q = 0; q < nrem; ++q) {
    if (!rc) ora_halo_unpack(c, cid[q], what, sl[q].recv);
    free(sl[q].send);
    free(sl[q].recv);
  }
  free(sl);
  free(cid);
  return rc ? fail("the exchange's swap operation failed") : 0;
}

/* mgSolution::Iterate mgSolution.cpp:246-269, gridLevel::GetBoundaryConditions
 * gridLevel.cpp:287-319, mgSolution::ImplicitUpdate :209-244, lusgs::Relax
 * linearSolver.cpp:430-470, dplur::Relax :509-535; with an exchange installed the
 * norms are reduced over the ranks as main.cpp:254-264 does */
int ora_iterate(ora_ctx *c, int mm, double cfl, double *l2, agx_linf *linf,
                double *matrix_resid) {
  double l2_in[NEQM];
  for (int e = 0; e < NEQ; ++e) l2_in[e] = l2[e];
  if (!c->have_ex)
    for (int n = 0; n < c->nconn; ++n)
      if (my_side(c, &c->conn[n]) >= 0)
        return fail("iterate: connections to other ranks need an exchange or the phase API");
  if (ora_phase_bc_faces(c)) return 1;
  if (ora_halo_exchange(c, AGX_HALO_STATE)) return 1;
  if (ora_phase_bc_edges(c)) return 1;
  if (ora_phase_residual(c, mm, cfl)) return 1;
  *matrix_resid = 0.0;
  if (c->cfg.time_integration >= AGX_TIME_IMPLICIT_EULER) {
    /* gridLevel::SwapEddyViscAndGradients gridLevel.cpp:386-388 (read by the
     * off-diagonal terms of the block-matrix solvers only) */
    if (is_block(c) && c->cfg.is_viscous) {
      if (ora_halo_exchange(c, AGX_HALO_VELGRAD_A)) return 1;
      if (ora_halo_exchange(c, AGX_HALO_VELGRAD_B)) return 1;
    }
    /* ... and of eddyViscosity_, f1_, f2_ (SwapTurbVars :389-392) */
    if (NEQ > NF && ora_halo_exchange(c, AGX_HALO_TURB)) return 1;
    if (ora_phase_implicit_begin(c)) return 1;
    for (int s = 0; s < c->cfg.matrix_sweeps; ++s) {
      if (ora_halo_exchange(c, AGX_HALO_UPDATE)) return 1;
      ora_phase_relax_forward(c, s);
      if (is_lusgs(c)) {
        if (ora_halo_exchange(c, AGX_HALO_UPDATE)) return 1;
        ora_phase_relax_backward(c, s);
      }
    }
    if (ora_halo_exchange(c, AGX_HALO_UPDATE)) return 1;
    ora_phase_matrix_residual(c, matrix_resid);
    ora_phase_implicit_update(c, mm, l2, linf);
  } else {
    ora_phase_explicit_update(c, mm, l2, linf);
  }
  if (c->have_ex && c->ex.nranks > 1) {
    const int nr = c->ex.nranks;
    norm_record mine, *all = (norm_record *)calloc(nr, sizeof(norm_record));
    memset(&mine, 0, sizeof mine);
    for (int e = 0; e < NEQ; ++e) mine.l2[e] = l2[e] - l2_in[e];
    mine.mres = *matrix_resid;
    mine.linf = linf->linf; mine.block = linf->block; mine.i = linf->i; mine.j = linf->j;
    mine.k = linf->k; mine.eqn = linf->eqn;
    if (c->ex.allgather(c->ex.user, &mine, all, (int64_t)sizeof mine, NULL)) {
      free(all);
      return fail("the exchange's allgather operation failed");
    }
    double mres = 0.0;
    for (int e = 0; e < NEQ; ++e) l2[e] = l2_in[e];
    for (int r = 0; r < nr; ++r) {
      for (int e = 0; e < NEQ; ++e) l2[e] += all[r].l2[e];
      mres += all[r].mres;
      if (all[r].linf > linf->linf) {
        linf->linf = all[r].linf; linf->block = all[r].block; linf->i = all[r].i;
        linf->j = all[r].j; linf->k = all[r].k; linf->eqn = all[r].eqn;
      }
    }
    *matrix_resid = mres;
    free(all);
  }
  return 0;
}

int ora_timing_enable(ora_ctx *c, int on) { (void)c; (void)on; return 0; }
int ora_timing_get(ora_ctx *c, int g, double *ms, int64_t *n) {
  (void)c; (void)g; *ms = 0.0; *n = 0; return 0;
}
int ora_timing_reset(ora_ctx *c) { (void)c; return 0; }
int ora_sync(ora_ctx *c) { (void)c; return 0; }
