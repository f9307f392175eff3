// agx_kernels.hpp -- HIP kernels of the hot path (gfx950, wave64, fp64).
//
// Data layout in HBM: structure of arrays.  Every per-cell and per-face
// quantity of a block is a separate plane of doubles with ONE common padded
// indexing
//     q(i,j,k) = (k+ng)*sxy + (j+ng)*sx + (i+ioff)
// where ioff >= ng is chosen so that physical cell i = 0 starts a 128-byte
// line, sx is a multiple of 16 doubles and face (i,j,k) of direction d is the
// LOWER d-face of cell (i,j,k).  Consecutive lanes of a wavefront walk
// consecutive i => every load/store of a plane is a fully coalesced 512-byte
// request per wave.  (The reference's AoS multiArray3d, multiArray3d.hpp:104-113,
// is converted at upload/download only.)
#pragma once
#include "agx_device.hpp"
#include "agx_lusgs.hpp"

namespace agx {

struct BlockDev {
  int ni, nj, nk, ng;
  int ioff;
  long sx, sxy, nplane;       // strides and plane length
  int parent;
  double* state[AGX_NEQ];     // state_                  procBlock.hpp:65
  double* state2[AGX_NEQ];    // second state buffer (fused explicit update)
  double* fa[3][4];           // fAreaI/J/K_ {nx,ny,nz,|A|}   :71-73
  double* vol;                // vol_                    :81
  double* cen[3];             // center_                 :82
  double* wid[3];             // cellWidthI/J/K_         :84-86
  double* resid[AGX_NEQ];     // residual_               :68
  double* specrad;            // specRadius_ (flow)      :74
  double* dt;                 // dt_                     :76
  double* consn[AGX_NEQ];     // consVarsN_              :66
  double* consnm1[AGX_NEQ];   // consVarsNm1_            :67
  double* x[AGX_NEQ];         // linearSolver::x_        linearSolver.hpp:38
  double* xold[AGX_NEQ];      // dplur copy              linearSolver.cpp:487
  double* a;                  // linearSolver::a_ (scalar flow part)
  double* ainv;               // linearSolver::aInv_
  // block-matrix solvers (null otherwise): 25 planes each of a_ / aInv_ (entry e of
  // cell q at [e * nplane + q], row major) and 9 planes of velocityGrad_
  double* am;
  double* aminv;
  double* vg;
  // 7-equation (rans) build, null otherwise: turbulence part of specRadius_ / a_ /
  // aInv_ (uncoupledScalar), viscosity_ of the last UpdateAuxillaryVariables (the
  // wall ghost states read it one iteration late, procBlock.cpp:2814), and
  // eddyViscosity_, f1_, f2_ of the cells
  double* specrad_t;
  double* a_t;
  double* ainv_t;
  double* viscp;
  double* turb3[3];
  // rans + block-matrix solvers: diagonal of the 2 x 2 turbulence block of a_ / aInv_
  // (entry e of cell q at [e * nplane + q])
  double* am_t;
  double* aminv_t;
  double* wdist;              // wallDist_                procBlock.hpp:88
  // hyperplane-by-hyperplane sweeps (null otherwise): cell-major records of what a sweep
  // reads of a cell and its neighbours (see k_sweep_records) and of the right-hand side
  double* sw_geo;
  double* sw_dyn;
  double* sw_rhs;
  // multigrid (null on a level without): mgForcing_, the matrix residual of the last
  // agx_mg_matrix_residual, coarseDu -- AGX_NEQ planes each, entry e of cell q at
  // [e * nplane + q]
  double* mg_forcing;
  double* mg_mres;
  double* mg_xsave;
  D2Dev d2;                   // diagonal-ordered arrays of the LU-SGS path (agx_lusgs.hpp)
  const agx_bc_surface* surf; // boundaryConditions      boundaryConditions.hpp:231
  int nsurf, nsurf_i, nsurf_j, nsurf_k;
  // nonreflecting inlet / outlet surfaces (null when the block has none): per
  // surface the offset of its cells in nr_grad (-1: not such a surface),
  // {pGrad[3], velGrad[9]} of the adjacent cell from the last residual, and the
  // surface's Mach {mean, max}
  const int* nr_off;
  double* nr_grad;
  // wall-law surfaces (rans): offset of the surface's faces in wallv (-1: not one)
  const int* wall_off;
  WallVars* wallv;
  double* nr_mach;
  // per side (surface type 1..6): 0 no connection BC on it, 1 all of it is
  // interblock / periodic, 2 mixed (look the cell up in `surf`)
  int side_conn[6];
  __host__ __device__ long idx(int i, int j, int k) const {
    return (long)(k + ng) * sxy + (long)(j + ng) * sx + (i + ioff);
  }
  __host__ __device__ long stride(int d) const {
    return d == 0 ? 1 : (d == 1 ? sx : sxy);
  }
};

struct NormPartial { double l2[AGX_NEQ]; double vmax; long long lin; };

// All planes of a block live in ONE slab, plane p at base + p * nplane, so a
// kernel needs a single base pointer (2 SGPRs) instead of ~60 plane pointers.
// (AGX_NEQ = 5; the 7-equation rans build of the same sources adds the planes of the
// turbulence part of the spectral radius / diagonal, the lagged viscosity_ and
// eddyViscosity_, f1_, f2_)
enum {
  PL_STATE_A = 0, PL_STATE_B = AGX_NEQ, PL_RESID = 2 * AGX_NEQ, PL_CONSN = 3 * AGX_NEQ,
  PL_CONSNM1 = 4 * AGX_NEQ, PL_X = 5 * AGX_NEQ, PL_XOLD = 6 * AGX_NEQ,
  PL_FA = 7 * AGX_NEQ /* + 4*d + c */, PL_VOL = PL_FA + 12, PL_CEN = PL_VOL + 1,
  PL_WID = PL_CEN + 3, PL_SPECRAD = PL_WID + 3, PL_DT = PL_SPECRAD + 1, PL_A = PL_DT + 1,
  PL_AINV = PL_A + 1, PL_WDIST = PL_AINV + 1,
#if AGX_NEQ == 7
  PL_SPECRAD_T = PL_WDIST + 1, PL_A_T, PL_AINV_T, PL_VISC, PL_TURB3 /* mut, f1, f2 */,
  PL_COUNT = PL_TURB3 + 3
#else
  PL_COUNT = PL_WDIST + 1
#endif
};
// the LDS-tiled / diagonal-ordered production kernels are written for the
// 5-equation set; the 7-equation build runs on the one-thread-per-cell kernels
#define AGX_FAST (AGX_NEQ == 5)
struct SlabDev {               // compact view used by the marching kernel
  double* base;
  long nplane, sx, sxy;
  int ni, nj, nk, ng, ioff;
  int st, sn;                  // plane ids of the current / next state buffer
  __device__ __forceinline__ double* pl(int id) const { return base + (long)id * nplane; }
  __device__ __forceinline__ const double* state(int e) const { return pl(st + e); }
  __device__ __forceinline__ double* snew(int e) const { return pl(sn + e); }
  __device__ __forceinline__ const double* wid(int d) const { return pl(PL_WID + d); }
  __device__ __forceinline__ const double* fa(int d, int c) const { return pl(PL_FA + 4 * d + c); }
  __device__ __forceinline__ long idx(int i, int j, int k) const {
    return (long)(k + ng) * sxy + (long)(j + ng) * sx + (i + ioff);
  }
  __device__ __forceinline__ void area(int d, long q, double* a) const {
#pragma unroll
    for (int c = 0; c < 4; ++c) a[c] = fa(d, c)[q];
  }
  // 32-bit BYTE offsets inside a plane (host guarantees nplane * 8 < 2^32): the
  // address is a wave-uniform plane base plus one VGPR, so the loads use the
  // SGPR-base addressing mode instead of a 64-bit VALU add per access
  __device__ __forceinline__ double ldb(int id, unsigned qb) const {
    return *reinterpret_cast<const double*>(reinterpret_cast<const char*>(pl(id)) + qb);
  }
  __device__ __forceinline__ void stb(int id, unsigned qb, double v) const {
    *reinterpret_cast<double*>(reinterpret_cast<char*>(pl(id)) + qb) = v;
  }
  __device__ __forceinline__ void areab(int d, unsigned qb, double* a) const {
#pragma unroll
    for (int c = 0; c < 4; ++c) a[c] = ldb(PL_FA + 4 * d + c, qb);
  }
};

struct SolverDev {   // scalar run-time parameters of agx_config
  double kappa, theta, zeta, relax, dual_time_cfl, dt_fixed, visc_cfl_coeff;
  int viscous, implicit, bdf2, requires_init, time_integration;
  // a coarse multigrid level: the inviscid residual ADDS its spectral radii to the main
  // diagonal, as the reference does everywhere (CalcInvFluxI/J/K; ResetDiagonal when an
  // iteration ends) -- a level restricted to twice in a W cycle keeps its first visit's
  int diag_add;
  int roe_jacobian;    // inviscidFluxJacobian: approximateRoe (RoeOffDiagonal)
  // the state has not changed since AssignSolToTimeN: U - U_n of the implicit
  // right-hand side (procBlock.cpp:1037, linearSolver.cpp:370) is exactly zero and
  // consVarsN need not be read
  int un_is_u;
  int block;           // input::IsBlockMatrix (blusgs / bdplur)
};

__device__ __forceinline__ void load5(double* const* p, long q, double* s) {
#pragma unroll
  for (int e = 0; e < AGX_NEQ; ++e) s[e] = p[e][q];
}
__device__ __forceinline__ void store5(double* const* p, long q, const double* s) {
#pragma unroll
  for (int e = 0; e < AGX_NEQ; ++e) p[e][q] = s[e];
}
__device__ __forceinline__ void load_area(const BlockDev& b, int d, long q, double* a) {
#pragma unroll
  for (int c = 0; c < 4; ++c) a[c] = b.fa[d][c][q];
}

// ---------------------------------------------------------------------------
// Inviscid residual, one thread per cell (gather form): the six face fluxes
// are added to the cell in the reference's order (-I_lower, +I_upper,
// -J_lower, +J_upper, -K_lower, +K_upper; procBlock.cpp:447-463,6121-6123),
// so no atomics and a reproducible sum.  Also forms the inviscid cell
// spectral radius, the scalar implicit diagonal and (inviscid runs) dt.
// Counterpart of procBlock::CalcInvFluxI/J/K procBlock.cpp:384-795.
template <int RECON, int LIM>
__device__ __forceinline__ void face_states_1d(double kappa, const double (*st)[AGX_NEQ],
                                               const double* w, int c, double* l, double* r) {
  // st[m], w[m]: 1-D stencil; the face lies between entries c-1 and c
  if (RECON == AGX_RECON_CONSTANT) {
#pragma unroll
    for (int e = 0; e < AGX_NEQ; ++e) { l[e] = st[c - 1][e]; r[e] = st[c][e]; }
  } else if (RECON == AGX_RECON_MUSCL) {
    // FaceReconMUSCL reconstruction.hpp:110-154
    const double dPl = (w[c - 1] + w[c - 1]) / (w[c - 1] + w[c]);
    const double dMl = (w[c - 1] + w[c - 1]) / (w[c - 1] + w[c - 2]);
    const double dPr = (w[c] + w[c]) / (w[c] + w[c - 1]);
    const double dMr = (w[c] + w[c]) / (w[c] + w[c + 1]);
#pragma unroll
    for (int e = 0; e < AGX_NEQ; ++e) {
      l[e] = muscl<LIM>(st[c - 2][e], st[c - 1][e], st[c][e], dPl, dMl, kappa);
      r[e] = muscl<LIM>(st[c + 1][e], st[c][e], st[c - 1][e], dPr, dMr, kappa);
    }
  } else {
    // FaceReconWENO reconstruction.hpp:244-310
    const double cwl[5] = {w[c - 3], w[c - 2], w[c - 1], w[c], w[c + 1]};
    const double cwr[5] = {w[c + 2], w[c + 1], w[c], w[c - 1], w[c - 2]};
    WenoCoeffs kl, kr;
    weno_coeffs(cwl, kl);
    weno_coeffs(cwr, kr);
#pragma unroll
    for (int e = 0; e < AGX_NEQ; ++e) {
      l[e] = weno<RECON == AGX_RECON_WENOZ>(kl, st[c - 3][e], st[c - 2][e],
                                            st[c - 1][e], st[c][e], st[c + 1][e]);
      r[e] = weno<RECON == AGX_RECON_WENOZ>(kr, st[c + 2][e], st[c + 1][e],
                                            st[c][e], st[c - 1][e], st[c - 2][e]);
    }
  }
}
template <int RECON, int LIM, int FLUX>
__device__ __forceinline__ void face_flux_1d(const GasDev& g, double kappa,
                                             const double (*st)[AGX_NEQ],
                                             const double* w, int c,
                                             const double* area, double* f) {
  double l[AGX_NEQ], r[AGX_NEQ];
  face_states_1d<RECON, LIM>(kappa, st, w, c, l, r);
  inviscid_flux<FLUX>(g, l, r, area, f);
}

// Block-matrix solvers: inviscid part of the main diagonal of a cell,
// + RusanovFluxJacobian(left face state) of its upper faces, - RusanovFluxJacobian(
// right face state) of its lower faces (procBlock.cpp:452-457, :481-486 and the j / k
// twins); the face states are reconstructed again here, one thread per cell.
template <int RECON, int LIM>
__global__ void __launch_bounds__(256)
k_block_diag_inv(BlockDev b, GasDev g, SolverDev sp) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int j = blockIdx.y * blockDim.y + threadIdx.y;
  const int k = blockIdx.z;
  if (i >= b.ni || j >= b.nj) return;
  constexpr int H = RECON == AGX_RECON_CONSTANT ? 1
                    : (RECON == AGX_RECON_MUSCL ? 2 : 3);
  constexpr int NS = 2 * H + 1;
  const long q = b.idx(i, j, k);
  double D[AGX_NJ], dt_ = 0.0;
#pragma unroll
  for (int e = 0; e < AGX_NJ; ++e) D[e] = 0.0;
  for (int d = 0; d < 3; ++d) {
    const long s = b.stride(d);
    double st[NS][AGX_NEQ], w[NS];
#pragma unroll
    for (int m = 0; m < NS; ++m) {
      const long qq = q + (m - H) * s;
      load5(b.state, qq, st[m]);
      w[m] = b.wid[d][qq];
    }
    double al[4], au[4], l[AGX_NEQ], r[AGX_NEQ], J[AGX_NJ];
    load_area(b, d, q, al);
    load_area(b, d, q + s, au);
    // the reference adds the upper neighbour's face first (face loop order i, i+1);
    // the lower face of this cell is face `i`, visited before face `i + 1`
    face_states_1d<RECON, LIM>(sp.kappa, st, w, H, l, r);
    rusanov_flux_jacobian(g, r, al, false, J);
#pragma unroll
    for (int e = 0; e < AGX_NJ; ++e) D[e] -= J[e];
    if (AGX_NEQ > 5) dt_ -= turb_inv_jac(r, al, false);
    face_states_1d<RECON, LIM>(sp.kappa, st, w, H + 1, l, r);
    rusanov_flux_jacobian(g, l, au, true, J);
#pragma unroll
    for (int e = 0; e < AGX_NJ; ++e) D[e] += J[e];
    if (AGX_NEQ > 5) dt_ += turb_inv_jac(l, au, true);
  }
#pragma unroll
  for (int e = 0; e < AGX_NJ; ++e) {
    double* am = b.am + (long)e * b.nplane + q;
    *am = sp.diag_add ? *am + D[e] : D[e];       // (a coarse multigrid level, see SolverDev)
  }
  if (AGX_NEQ > 5) {       // the same number for k and omega (turbModel::InvJac)
    b.am_t[q] = sp.diag_add ? b.am_t[q] + dt_ : dt_;
    b.am_t[b.nplane + q] = sp.diag_add ? b.am_t[b.nplane + q] + dt_ : dt_;
  }
}

template <int RECON, int LIM, int FLUX>
__global__ void __launch_bounds__(256)
k_inv_residual(BlockDev b, GasDev g, SolverDev sp, double cfl) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int j = blockIdx.y * blockDim.y + threadIdx.y;
  const int k = blockIdx.z;
  if (i >= b.ni || j >= b.nj) return;
  constexpr int H = RECON == AGX_RECON_CONSTANT ? 1
                    : (RECON == AGX_RECON_MUSCL ? 2 : 3);
  constexpr int NS = 2 * H + 1;
  const long q = b.idx(i, j, k);
  double res[AGX_NEQ] = {0.0, 0.0, 0.0, 0.0, 0.0};
  double sr = 0.0, srt = 0.0;
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    const long s = b.stride(d);
    double st[NS][AGX_NEQ], w[NS];
#pragma unroll
    for (int m = 0; m < NS; ++m) {
      const long qq = q + (m - H) * s;
      load5(b.state, qq, st[m]);
      w[m] = b.wid[d][qq];
    }
    double al[4], au[4], f[AGX_NEQ];
    load_area(b, d, q, al);
    load_area(b, d, q + s, au);
    face_flux_1d<RECON, LIM, FLUX>(g, sp.kappa, st, w, H, al, f);
#pragma unroll
    for (int e = 0; e < AGX_NEQ; ++e) res[e] -= f[e] * al[3];
    face_flux_1d<RECON, LIM, FLUX>(g, sp.kappa, st, w, H + 1, au, f);
#pragma unroll
    for (int e = 0; e < AGX_NEQ; ++e) res[e] += f[e] * au[3];
    sr += inv_cell_spec_rad(g, st[H], al, au);
    if (AGX_NEQ > 5) {
      // turbModel::InviscidCellSpectralRadius turbulence.cpp:162-172
      const double v[3] = {0.5 * (al[0] + au[0]), 0.5 * (al[1] + au[1]), 0.5 * (al[2] + au[2])};
      srt += fabs(dot3(st[H] + 1, v)) * fast_rsqrt(dot3(v, v)) * (0.5 * (al[3] + au[3]));
    }
  }
  store5(b.resid, q, res);
  b.specrad[q] = sr;
  if (sp.implicit) b.a[q] = sp.diag_add ? b.a[q] + sr : sr;
  if (AGX_NEQ > 5) {
    b.specrad_t[q] = srt;
    if (sp.implicit) b.a_t[q] = sp.diag_add ? b.a_t[q] + srt : srt;
  }
  if (!sp.viscous)
    b.dt[q] = sp.dt_fixed > 0.0 ? sp.dt_fixed : cfl * (b.vol[q] / fmax(sr, 0.0));
}

// ---------------------------------------------------------------------------
// Inviscid residual, face-once "marching" form (the production kernel).
//
// Work decomposition for 64-wide wavefronts, one 1024-thread workgroup
// (16 waves, 4 per SIMD, <= 128 VGPRs) per tile of 64 (i) x MARCH_TJ (j)
// cells, marching through a chunk of k planes:
//   * waves 0..TJ-1   : one wave = one i-row of 64 cells.  Each lane owns one
//     cell column and computes exactly ONE flux per direction per step: the
//     lower i-face, the lower j-face and the upper k-face of its cell.
//   * i hand-off     : the upper i-face flux is the neighbour lane's lower
//     flux, fetched with a wavefront shuffle (no memory);
//   * j hand-off     : lower j-face fluxes go through LDS (double buffered,
//     one barrier per k-step); each row reads the lower flux of the row above
//     as its own upper flux;
//   * k hand-off     : marching -- the upper k-face flux of step k stays in
//     registers and is the lower k-face flux of step k+1 (the k-stencil cells
//     are re-read; they are L2-resident from the previous steps);
//   * wave TJ        : halo row, computes only the j-flux of the tile's top face;
//   * wave TJ+1      : halo column, lane l computes only the i-flux of the
//     tile's right face for row l.
// Every flux is therefore evaluated once per face (+ 2/(3 TJ) halo overhead +
// 1/chunk for the first k-face of a chunk) instead of twice as in the gather
// form, and all plane accesses of a wave are 512-byte coalesced rows.
// The residual is still summed per cell in the reference's order
// (-I_lo +I_up -J_lo +J_up -K_lo +K_up; procBlock.cpp:447-463,6121-6123).
// With FUSE the kernel also performs procBlock::UpdateBlock (explicit Euler /
// RK stage, procBlock.cpp:826-947) into the second state buffer and reduces
// the residual norms, so one launch is one whole mgSolution::Iterate stage
// and moves the algorithmic 296 B / cell.
struct MarchArgs {
  int kchunk;              // k planes per workgroup
  int mode;                // FUSE: 0 explicit Euler, 1 RK stage
  double alpha;            // RK stage coefficient
  int store_consn;         // first residual of a time step: also write cons(state) to consVarsN
  int ablate;              // diagnostic (AGX_ABLATE): 1 no flux math, 2 no
                           // reconstruction, 4 no stores, 8 no cons->prim,
                           // 16 no spectral radius, 32 idle halo waves,
                           // 64 no tile publish, 128 no prefetch loads
  NormPartial* partials;   // FUSE: one per workgroup
};

// Left/right states at the lower d-face of cell qc, reconstructed variable by
// variable straight from the SoA planes (only the 2x5 face values stay live).
template <int RECON, int LIM>
__device__ __forceinline__ void recon_face(const SlabDev& b, int d, long qc,
                                           long s, double kappa, double* l,
                                           double* r) {
  if (RECON == AGX_RECON_CONSTANT) {
#pragma unroll
    for (int e = 0; e < AGX_NEQ; ++e) { l[e] = b.state(e)[qc - s]; r[e] = b.state(e)[qc]; }
  } else if (RECON == AGX_RECON_MUSCL) {
    // FaceReconMUSCL reconstruction.hpp:110-154
    const double* wd = b.wid(d);
    const double w2 = wd[qc - 2 * s], w1 = wd[qc - s], w0 = wd[qc], wp = wd[qc + s];
    const double r10 = fast_rcp(w1 + w0);
    const double dPl = (w1 + w1) * r10, dMl = (w1 + w1) * fast_rcp(w1 + w2);
    const double dPr = (w0 + w0) * r10, dMr = (w0 + w0) * fast_rcp(w0 + wp);
#pragma unroll
    for (int e = 0; e < AGX_NEQ; ++e) {
      const double* p = b.state(e);
      const double u2 = p[qc - 2 * s], u1 = p[qc - s], u0 = p[qc], up = p[qc + s];
      l[e] = muscl<LIM>(u2, u1, u0, dPl, dMl, kappa);
      r[e] = muscl<LIM>(up, u0, u1, dPr, dMr, kappa);
    }
  } else {
    // FaceReconWENO reconstruction.hpp:244-310
    double w[6];
#pragma unroll
    for (int m = 0; m < 6; ++m) w[m] = b.wid(d)[qc + (m - 3) * s];
    const double cwl[5] = {w[0], w[1], w[2], w[3], w[4]};
    const double cwr[5] = {w[5], w[4], w[3], w[2], w[1]};
    WenoCoeffs kl, kr;
    weno_coeffs(cwl, kl);
    weno_coeffs(cwr, kr);
#pragma unroll
    for (int e = 0; e < AGX_NEQ; ++e) {
      const double* p = b.state(e);
      double u[6];
#pragma unroll
      for (int m = 0; m < 6; ++m) u[m] = p[qc + (m - 3) * s];
      l[e] = weno<RECON == AGX_RECON_WENOZ>(kl, u[0], u[1], u[2], u[3], u[4]);
      r[e] = weno<RECON == AGX_RECON_WENOZ>(kr, u[5], u[4], u[3], u[2], u[1]);
    }
  }
}
// flux through the lower d-face of cell qc, already multiplied by |A|
template <int RECON, int LIM, int FLUX>
__device__ __forceinline__ void face_flux_area(const SlabDev& b, const GasDev& g,
                                               double kappa, int d, long qc,
                                               long s, const double* area,
                                               double* f) {
  double l[AGX_NEQ], r[AGX_NEQ];
  recon_face<RECON, LIM>(b, d, qc, s, kappa, l, r);
  inviscid_flux<FLUX>(g, l, r, area, f);
#pragma unroll
  for (int e = 0; e < AGX_NEQ; ++e) f[e] *= area[3];
}

template <int RECON, int LIM, int FLUX, bool FUSE, int TJ>
__global__ void __launch_bounds__(64 * (TJ + 2))
k_residual_march(SlabDev b, GasDev g, SolverDev sp, double cfl, MarchArgs ma) {
  __shared__ double sFj[2][TJ + 1][AGX_NEQ][64];
  __shared__ double sFi[2][TJ][AGX_NEQ];
  const int lane = threadIdx.x, wv = threadIdx.y;
  const int i0 = blockIdx.x * 64, j0 = blockIdx.y * TJ;
  const int k0 = blockIdx.z * ma.kchunk;
  const int k1 = min(k0 + ma.kchunk, b.nk);
  const int itop = min(i0 + 64, b.ni), jtop = min(j0 + TJ, b.nj);
  // roles
  const bool cell_wave = wv < TJ;
  int i, j;
  bool cell = false, do_fi = false, do_fj = false;
  if (cell_wave) {
    i = i0 + lane; j = j0 + wv;
    cell = i < itop && j < jtop;
    do_fi = cell; do_fj = cell;
  } else if (wv == TJ) {            // halo row: top j-face of the tile
    i = i0 + lane; j = jtop;
    do_fj = i < itop;
  } else {                          // halo column: right i-face of the tile
    i = itop; j = j0 + lane;
    do_fi = lane < TJ && j < jtop;
  }
  // clamp so that idle lanes still address valid memory
  const int ic = min(i, b.ni), jc = min(j, b.nj);
  const long s_i = 1, s_j = b.sx, s_k = b.sxy;
  long q = b.idx(ic, jc, k0);
  double fk_lo[AGX_NEQ] = {0, 0, 0, 0, 0}, ak_lo[4] = {0, 0, 0, 0};
  if (cell) {   // first k-face of the chunk (prologue)
    b.area(2, q, ak_lo);
    face_flux_area<RECON, LIM, FLUX>(b, g, sp.kappa, 2, q, s_k, ak_lo, fk_lo);
  }
  double l2[AGX_NEQ] = {0, 0, 0, 0, 0};
  double vmax = -1.0e300;
  long long vlin = 0x7fffffffffffffffLL;
  int buf = 0;
  for (int k = k0; k < k1; ++k, q += s_k, buf ^= 1) {
    double fi[AGX_NEQ] = {0, 0, 0, 0, 0}, fj[AGX_NEQ] = {0, 0, 0, 0, 0};
    double fk_up[AGX_NEQ], ai_lo[4], aj_lo[4], ak_up[4];
    if (do_fi) {
      b.area(0, q, ai_lo);
      face_flux_area<RECON, LIM, FLUX>(b, g, sp.kappa, 0, q, s_i, ai_lo, fi);
    }
    if (do_fj) {
      b.area(1, q, aj_lo);
      face_flux_area<RECON, LIM, FLUX>(b, g, sp.kappa, 1, q, s_j, aj_lo, fj);
    }
    if (cell) {
      b.area(2, q + s_k, ak_up);
      face_flux_area<RECON, LIM, FLUX>(b, g, sp.kappa, 2, q + s_k, s_k, ak_up, fk_up);
    }
    // publish the fluxes the neighbours need
    if (wv <= TJ) {
      const int row = cell_wave ? wv : (jtop - j0);
      if (do_fj) {
#pragma unroll
        for (int e = 0; e < AGX_NEQ; ++e) sFj[buf][row][e][lane] = fj[e];
      }
    } else if (do_fi) {
#pragma unroll
      for (int e = 0; e < AGX_NEQ; ++e) sFi[buf][lane][e] = fi[e];
    }
    __syncthreads();
    if (cell_wave) {
      double fi_up[AGX_NEQ];
#pragma unroll
      for (int e = 0; e < AGX_NEQ; ++e) fi_up[e] = __shfl_down(fi[e], 1, 64);
      if (cell) {
        if (i == itop - 1) {
#pragma unroll
          for (int e = 0; e < AGX_NEQ; ++e) fi_up[e] = sFi[buf][wv][e];
        }
        double res[AGX_NEQ];
#pragma unroll
        for (int e = 0; e < AGX_NEQ; ++e)
          res[e] = ((((-fi[e] + fi_up[e]) - fj[e]) + sFj[buf][wv + 1][e][lane]) -
                    fk_lo[e]) + fk_up[e];
        // InvCellSpectralRadius spectralRadius.hpp:44-64, three directions
        double sc[AGX_NEQ];
#pragma unroll
        for (int e = 0; e < AGX_NEQ; ++e) sc[e] = b.state(e)[q];
        const double cs = sound_speed(g, sc);
        double sr = 0.0;
        {
          double au[4];
          b.area(0, q + s_i, au);
          double v[3] = {0.5 * (ai_lo[0] + au[0]), 0.5 * (ai_lo[1] + au[1]), 0.5 * (ai_lo[2] + au[2])};
          sr += (fabs(dot3(sc + 1, v)) * rsqrt(dot3(v, v)) + cs) * (0.5 * (ai_lo[3] + au[3]));
          b.area(1, q + s_j, au);
          v[0] = 0.5 * (aj_lo[0] + au[0]); v[1] = 0.5 * (aj_lo[1] + au[1]); v[2] = 0.5 * (aj_lo[2] + au[2]);
          sr += (fabs(dot3(sc + 1, v)) * rsqrt(dot3(v, v)) + cs) * (0.5 * (aj_lo[3] + au[3]));
          v[0] = 0.5 * (ak_lo[0] + ak_up[0]); v[1] = 0.5 * (ak_lo[1] + ak_up[1]); v[2] = 0.5 * (ak_lo[2] + ak_up[2]);
          sr += (fabs(dot3(sc + 1, v)) * rsqrt(dot3(v, v)) + cs) * (0.5 * (ak_lo[3] + ak_up[3]));
        }
        const double vol = b.pl(PL_VOL)[q];
        const double dt = sp.dt_fixed > 0.0 ? sp.dt_fixed : cfl * (vol / fmax(sr, 0.0));
#pragma unroll
        for (int e = 0; e < AGX_NEQ; ++e) b.pl(PL_RESID + e)[q] = res[e];
        b.pl(PL_SPECRAD)[q] = sr;
        if (sp.implicit) b.pl(PL_A)[q] = sp.diag_add ? b.pl(PL_A)[q] + sr : sr;
        if (!sp.viscous) b.pl(PL_DT)[q] = dt;
        if (FUSE) {
          double u[AGX_NEQ], ns[AGX_NEQ];
          double fac = dt / vol;
          if (ma.mode == 0) {
            prim_to_cons(g, sc, u);
          } else {
#pragma unroll
            for (int e = 0; e < AGX_NEQ; ++e) u[e] = b.pl(PL_CONSN + e)[q];
            fac *= ma.alpha;
          }
#pragma unroll
          for (int e = 0; e < AGX_NEQ; ++e) u[e] -= fac * res[e];
          cons_to_prim(g, u, ns);
#pragma unroll
          for (int e = 0; e < AGX_NEQ; ++e) b.snew(e)[q] = ns[e];
          const long lin0 = (((long)k * b.nj + j) * b.ni + i) * AGX_NEQ;
#pragma unroll
          for (int e = 0; e < AGX_NEQ; ++e) {
            l2[e] += res[e] * res[e];
            if (res[e] > vmax) { vmax = res[e]; vlin = lin0 + e; }
          }
        }
#pragma unroll
        for (int e = 0; e < AGX_NEQ; ++e) fk_lo[e] = fk_up[e];
#pragma unroll
        for (int c = 0; c < 4; ++c) ak_lo[c] = ak_up[c];
      }
    }
  }
  if (FUSE) {
    // workgroup reduction of the norm partials (TJ + 2 waves)
    constexpr int NW = TJ + 2;
    __shared__ double shv[AGX_NEQ + 1][NW];
    __shared__ long long shl[NW];
    for (int off = 32; off > 0; off >>= 1) {
#pragma unroll
      for (int e = 0; e < AGX_NEQ; ++e) l2[e] += __shfl_down(l2[e], off, 64);
      const double ov = __shfl_down(vmax, off, 64);
      const long long ol = __shfl_down(vlin, off, 64);
      if (ov > vmax || (ov == vmax && ol < vlin)) { vmax = ov; vlin = ol; }
    }
    if (lane == 0) {
#pragma unroll
      for (int e = 0; e < AGX_NEQ; ++e) shv[e][wv] = l2[e];
      shv[AGX_NEQ][wv] = vmax;
      shl[wv] = vlin;
    }
    __syncthreads();
    if (lane == 0 && wv == 0) {
      NormPartial p;
#pragma unroll
      for (int e = 0; e < AGX_NEQ; ++e) {
        double acc = 0.0;
        for (int w = 0; w < NW; ++w) acc += shv[e][w];
        p.l2[e] = acc;
      }
      p.vmax = shv[AGX_NEQ][0];
      p.lin = shl[0];
      for (int w = 1; w < NW; ++w)
        if (shv[AGX_NEQ][w] > p.vmax || (shv[AGX_NEQ][w] == p.vmax && shl[w] < p.lin)) {
          p.vmax = shv[AGX_NEQ][w];
          p.lin = shl[w];
        }
      const long bid = ((long)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
      ma.partials[bid] = p;
    }
  }
}

#if AGX_FAST
// ---------------------------------------------------------------------------
// Inviscid residual, tiled marching form with LDS-staged planes (production).
//
// Same face-once decomposition as k_residual_march (one wave per i-row of 64
// cells, shuffle hand-off in i, LDS hand-off in j, register hand-off in k,
// two halo waves), plus the classic 2.5-D blocking so that every state value
// is fetched from HBM once per workgroup:
//   * each lane keeps the k-stencil of its own column in a rolling register
//     window; the cell entering the window is loaded ONE STEP AHEAD, so its
//     HBM latency overlaps a whole step of flux arithmetic;
//   * the current k-plane of the tile (5 state variables + the i/j cell widths,
//     with a halo of H cells in i and j) lives in LDS, double buffered: the
//     interior comes from the register windows, the halo ring is prefetched
//     from global memory one step ahead; all i/j stencil reads are LDS reads;
//   * one workgroup barrier per k-step publishes both the j-fluxes and the
//     next plane.
// diagnostic ablation switches (MarchArgs::ablate) exist only in builds made
// with -DAGX_ABLATION; the production build compiles them away
#ifdef AGX_ABLATION
#define AGX_AB(bit) ((ma.ablate & (bit)) != 0)
#else
#define AGX_AB(bit) (false)
#endif
// scheduling barriers of the tile kernel: between the two sides of a WENO face and between
// the faces of a step (they bound the live ranges; AGX_SB=0 builds without them)
#ifndef AGX_SB
#define AGX_SB 3
#endif
#define AGX_SB_SIDES do { if (AGX_SB & 1) __builtin_amdgcn_sched_barrier(0); } while (0)
#define AGX_SB_FACES do { if (AGX_SB & 2) __builtin_amdgcn_sched_barrier(0); } while (0)
template <int RECON, int LIM, class Get, class GetW>
__device__ __forceinline__ void recon_generic(Get get, GetW getw, double kappa,
                                              double* l, double* r) {
  // get(e, m): variable e at stencil offset m from the face (m = -1: left
  // cell, m = 0: right cell); getw(m): cell width at that offset
  if (RECON == AGX_RECON_CONSTANT) {
#pragma unroll
    for (int e = 0; e < AGX_NEQ; ++e) { l[e] = get(e, -1); r[e] = get(e, 0); }
  } else if (RECON == AGX_RECON_MUSCL) {
    const double w2 = getw(-2), w1 = getw(-1), w0 = getw(0), wp = getw(1);
    const double r10 = fast_rcp(w1 + w0);
    const double dPl = (w1 + w1) * r10, dMl = (w1 + w1) * fast_rcp(w1 + w2);
    const double dPr = (w0 + w0) * r10, dMr = (w0 + w0) * fast_rcp(w0 + wp);
#pragma unroll
    for (int e = 0; e < AGX_NEQ; ++e) {
      const double u2 = get(e, -2), u1 = get(e, -1), u0 = get(e, 0), up = get(e, 1);
      l[e] = muscl<LIM>(u2, u1, u0, dPl, dMl, kappa);
      r[e] = muscl<LIM>(up, u0, u1, dPr, dMr, kappa);
    }
  } else {
    // one side at a time: the width-only coefficient set of a side is 16 doubles
    // (agx_device.hpp: WenoCoeffs), and holding both sets across the variable loop
    // costs more registers than the shared reciprocals save instructions
    {
      double cw[5];
#pragma unroll
      for (int m = 0; m < 5; ++m) cw[m] = getw(m - 3);
      WenoCoeffs kc;
      weno_coeffs(cw, kc);
#pragma unroll
      for (int e = 0; e < AGX_NEQ; ++e)
        l[e] = weno<RECON == AGX_RECON_WENOZ>(kc, get(e, -3), get(e, -2), get(e, -1),
                                              get(e, 0), get(e, 1));
    }
    AGX_SB_SIDES;
    {
      double cw[5];
#pragma unroll
      for (int m = 0; m < 5; ++m) cw[m] = getw(2 - m);
      WenoCoeffs kc;
      weno_coeffs(cw, kc);
#pragma unroll
      for (int e = 0; e < AGX_NEQ; ++e)
        r[e] = weno<RECON == AGX_RECON_WENOZ>(kc, get(e, 2), get(e, 1), get(e, 0),
                                              get(e, -1), get(e, -2));
    }
  }
}

// FUSE: 0 residual only, 1 + explicit stage update, 2 + AssignSolToTimeN (stage 0)
template <int RECON, int LIM, int FLUX, int FUSE, int TJ>
__global__ void __launch_bounds__(64 * (TJ + 2), 2)
k_residual_tile(SlabDev b, GasDev g, SolverDev sp, double cfl, MarchArgs ma) {
  constexpr int H = RECON == AGX_RECON_CONSTANT ? 1
                    : (RECON == AGX_RECON_MUSCL ? 2 : 3);
  constexpr bool BAL = TJ >= 4;   // SIMD load balancing, see the j-face block
  constexpr int NV = AGX_NEQ + 2;                 // state + wid_i + wid_j
  constexpr int TW = 64 + 2 * H, TR = TJ + 2 * H;
  constexpr int PLANE = TR * TW;                  // doubles per variable
  __shared__ double tile[2][NV][TR][TW];
  __shared__ double sFj[2][TJ + 1][AGX_NEQ][64];  // lower j-face fluxes
  __shared__ double sFi[2][TJ][AGX_NEQ];          // right-face i-fluxes (halo wave)
  // per-thread slots of the cell waves: norm accumulators, and (MUSCL and
  // below, where LDS allows) the lower k-face flux / area carried between steps
  constexpr bool PARK = H <= 2;
  __shared__ double sNorm[AGX_NEQ + 1][TJ][64];
  __shared__ long long sLin[TJ][64];
  __shared__ double sFk[PARK ? TJ : 1][AGX_NEQ][64];
  __shared__ double sAk[PARK ? TJ : 1][4][64];
  const int lane = threadIdx.x, wv = threadIdx.y;
  const long s_k = b.sxy;
  if (FUSE && wv < TJ) {
#pragma unroll
    for (int e = 0; e < AGX_NEQ; ++e) sNorm[e][wv][lane] = 0.0;
    sNorm[AGX_NEQ][wv][lane] = -1.0e300;
    sLin[wv][lane] = 0x7fffffffffffffffLL;
  }
  // Persistent workgroups (one per CU): the (column tile, k) steps of the block
  // are one linear sequence cut into gridDim.x equal ranges, so every CU
  // finishes together whatever the block shape; a range that crosses into the
  // next column re-primes its window there.  Workgroup n runs on XCD n % 8:
  // ranges are dealt so that each XCD's L2 sees neighbouring columns.
  const int gx = (b.ni + 63) / 64, gy = (b.nj + TJ - 1) / TJ;
  const long S = (long)gx * gy * b.nk;
  const int P = gridDim.x;
  const int rr = P % 8 == 0 ? (int)(blockIdx.x % 8) * (P / 8) + (int)(blockIdx.x / 8)
                            : (int)blockIdx.x;
  long s_pos = S * rr / P;
  const long s_end = S * (rr + 1) / P;
  while (s_pos < s_end) {
  const int col = (int)(s_pos / b.nk);
  const int k0 = (int)(s_pos - (long)col * b.nk);
  const int k1 = (int)min((long)b.nk, k0 + (s_end - s_pos));
  s_pos += k1 - k0;
  const int i0 = (col % gx) * 64, j0 = (col / gx) * TJ;
  const int itop = min(i0 + 64, b.ni), jtop = min(j0 + TJ, b.nj);
  if (wv < TJ) {
    // ======================= cell waves =====================================
    const int i = i0 + lane, j = j0 + wv;
    const bool cell = i < itop && j < jtop;
    const int ic = min(i, b.ni + b.ng - 1), jc = min(j, b.nj + b.ng - 1);
    long q = b.idx(ic, jc, k0);
    const int to = (wv + H) * TW + lane + H;      // own slot in a tile plane
    const int pw = PARK ? wv : 0;
    double W[2 * H][AGX_NEQ], wk[2 * H];
    double fk_reg[AGX_NEQ] = {0, 0, 0, 0, 0}, ak_reg[4];   // !PARK only
    {
      double W0[AGX_NEQ], w0, ak_lo[4], fk_lo[AGX_NEQ] = {0, 0, 0, 0, 0};
      b.area(2, q, ak_lo);
#pragma unroll
      for (int e = 0; e < AGX_NEQ; ++e) W0[e] = b.state(e)[q - H * s_k];
      w0 = b.wid(2)[q - H * s_k];
#pragma unroll
      for (int m = 0; m < 2 * H; ++m) {
#pragma unroll
        for (int e = 0; e < AGX_NEQ; ++e) W[m][e] = b.state(e)[q + (m - H + 1) * s_k];
        wk[m] = b.wid(2)[q + (m - H + 1) * s_k];
      }
      if (cell) {
        double l[AGX_NEQ], r[AGX_NEQ];
        recon_generic<RECON, LIM>(
            [&](int e, int m) { return m + H == 0 ? W0[e] : W[m + H - 1][e]; },
            [&](int m) { return m + H == 0 ? w0 : wk[m + H - 1]; }, sp.kappa, l, r);
        inviscid_flux<FLUX>(g, l, r, ak_lo, fk_lo);
#pragma unroll
        for (int e = 0; e < AGX_NEQ; ++e) fk_lo[e] *= ak_lo[3];
      }
      if (PARK) {
#pragma unroll
        for (int e = 0; e < AGX_NEQ; ++e) sFk[pw][e][lane] = fk_lo[e];
#pragma unroll
        for (int e = 0; e < 4; ++e) sAk[pw][e][lane] = ak_lo[e];
      } else {
#pragma unroll
        for (int e = 0; e < AGX_NEQ; ++e) fk_reg[e] = fk_lo[e];
#pragma unroll
        for (int e = 0; e < 4; ++e) ak_reg[e] = ak_lo[e];
      }
      double* tl = &tile[0][0][0][0];
#pragma unroll
      for (int e = 0; e < AGX_NEQ; ++e) tl[e * PLANE + to] = W[H - 1][e];
      tl[AGX_NEQ * PLANE + to] = b.wid(0)[q];
      tl[(AGX_NEQ + 1) * PLANE + to] = b.wid(1)[q];
    }
    __syncthreads();
    int cur = 0;
    unsigned qb = (unsigned)(q * 8);
    const unsigned skb = (unsigned)(s_k * 8), sxb = (unsigned)(b.sx * 8);
    for (int k = k0; k < k1; ++k, q += s_k, qb += skb, cur ^= 1) {
      const double* tc = &tile[cur][0][0][0];
      double* tn = &tile[cur ^ 1][0][0][0];
      // the load needed first is issued first (vmcnt retires in order), the
      // next-step prefetch after it
      double ak_up[4];
      b.areab(2, qb + skb, ak_up);
      double nxt[AGX_NEQ], nwk, nwi, nwj;
      auto prefetch = [&]() {
        if AGX_AB(128) {
#pragma unroll
          for (int e = 0; e < AGX_NEQ; ++e) nxt[e] = W[2 * H - 1][e];
          nwk = wk[0]; nwi = wk[0]; nwj = wk[0];
        } else {
#pragma unroll
          for (int e = 0; e < AGX_NEQ; ++e) nxt[e] = b.ldb(b.st + e, qb + (H + 1) * skb);
          nwk = b.ldb(PL_WID + 2, qb + (H + 1) * skb);
          nwi = b.ldb(PL_WID + 0, qb + skb);
          nwj = b.ldb(PL_WID + 1, qb + skb);
        }
      };
      // MUSCL and below: requested first, a whole step ahead of its use.  WENO: the
      // k- and i-face with their coefficient sets leave no room for eight more live
      // values (256 VGPRs + scratch), so there it is requested after the i-face.
#ifndef AGX_WENO_LATE
#define AGX_WENO_LATE 1
#endif
      constexpr bool LATE = AGX_WENO_LATE && H >= 3;
      if (!LATE) prefetch();
      // InvCellSpectralRadius spectralRadius.hpp:44-64, one direction per block
      const double* sc = W[H - 1];
      const double cs = sound_speed(g, sc);
      auto specrad = [&](const double* al, const double* au) {
        const double v[3] = {0.5 * (al[0] + au[0]), 0.5 * (al[1] + au[1]),
                             0.5 * (al[2] + au[2])};
        return (fabs(dot3(sc + 1, v)) * fast_rsqrt(dot3(v, v)) + cs) * (0.5 * (al[3] + au[3]));
      };
      double res[AGX_NEQ] = {0, 0, 0, 0, 0};
      double sr_i = 0.0, sr_j = 0.0, sr_k = 0.0;
      // ---- k face (upper): stencil from the register window ----
      if (cell) {
        double l[AGX_NEQ], r[AGX_NEQ], f[AGX_NEQ];
        if AGX_AB(2) {
#pragma unroll
          for (int e = 0; e < AGX_NEQ; ++e) { l[e] = W[H - 1][e]; r[e] = W[H][e]; }
        } else {
          recon_generic<RECON, LIM>([&](int e, int m) { return W[m + H][e]; },
                                    [&](int m) { return wk[m + H]; }, sp.kappa, l, r);
        }
        if AGX_AB(1) {
#pragma unroll
          for (int e = 0; e < AGX_NEQ; ++e) f[e] = l[e] + r[e] * ak_up[0];
        } else {
          inviscid_flux<FLUX>(g, l, r, ak_up, f);
        }
        double ak_lo[4];
#pragma unroll
        for (int e = 0; e < AGX_NEQ; ++e) {
          f[e] *= ak_up[3];
          if (PARK) {
            res[e] = f[e] - sFk[pw][e][lane];
            sFk[pw][e][lane] = f[e];
          } else {
            res[e] = f[e] - fk_reg[e];
            fk_reg[e] = f[e];
          }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          if (PARK) {
            ak_lo[e] = sAk[pw][e][lane];
            sAk[pw][e][lane] = ak_up[e];
          } else {
            ak_lo[e] = ak_reg[e];
            ak_reg[e] = ak_up[e];
          }
        }
        sr_k = AGX_AB(16) ? 1.0 : specrad(ak_lo, ak_up);
      }
      AGX_SB_FACES;
      // ---- i face (lower): stencil from the LDS plane; the upper face comes
      // from lane + 1 (the tile's right face from the halo wave, below) ----
      {
        double fi[AGX_NEQ] = {0, 0, 0, 0, 0};
        if (cell) {
          double l[AGX_NEQ], r[AGX_NEQ], ai_lo[4], ai_up[4];
          b.areab(0, qb, ai_lo);
          b.areab(0, qb + 8, ai_up);
          const double* base = tc + to;
          if AGX_AB(2) {
#pragma unroll
            for (int e = 0; e < AGX_NEQ; ++e) { l[e] = base[e * PLANE - 1]; r[e] = base[e * PLANE]; }
          } else {
            recon_generic<RECON, LIM>(
                [&](int e, int m) { return base[e * PLANE + m]; },
                [&](int m) { return base[AGX_NEQ * PLANE + m]; }, sp.kappa, l, r);
          }
          if AGX_AB(1) {
#pragma unroll
            for (int e = 0; e < AGX_NEQ; ++e) fi[e] = l[e] + r[e] * ai_lo[0];
          } else {
            inviscid_flux<FLUX>(g, l, r, ai_lo, fi);
          }
#pragma unroll
          for (int e = 0; e < AGX_NEQ; ++e) fi[e] *= ai_lo[3];
          sr_i = AGX_AB(16) ? 1.0 : specrad(ai_lo, ai_up);
        }
        const bool edge = i == itop - 1;
#pragma unroll
        for (int e = 0; e < AGX_NEQ; ++e) {
          const double up = __shfl_down(fi[e], 1, 64);
          res[e] += (edge ? 0.0 : up) - fi[e];
        }
      }
      AGX_SB_FACES;
      if (LATE) prefetch();
      // ---- j face (lower): handed to the row below through LDS.  Rows 0 and 1
      // leave this flux to the two halo waves (BAL): the cell waves of those rows
      // share a SIMD with another cell wave, the halo waves with one only, and
      // this evens the four SIMDs out at five flux evaluations each. ----
      if (cell && BAL && wv < 2) {
        double aj_lo[4], aj_up[4];
        b.areab(1, qb, aj_lo);
        b.areab(1, qb + sxb, aj_up);
        sr_j = AGX_AB(16) ? 1.0 : specrad(aj_lo, aj_up);
      } else if (cell) {
        double l[AGX_NEQ], r[AGX_NEQ], f[AGX_NEQ], aj_lo[4], aj_up[4];
        b.areab(1, qb, aj_lo);
        b.areab(1, qb + sxb, aj_up);
        const double* base = tc + to;
        if AGX_AB(2) {
#pragma unroll
          for (int e = 0; e < AGX_NEQ; ++e) { l[e] = base[e * PLANE - TW]; r[e] = base[e * PLANE]; }
        } else {
          recon_generic<RECON, LIM>(
              [&](int e, int m) { return base[e * PLANE + m * TW]; },
              [&](int m) { return base[(AGX_NEQ + 1) * PLANE + m * TW]; }, sp.kappa, l, r);
        }
        if AGX_AB(1) {
#pragma unroll
          for (int e = 0; e < AGX_NEQ; ++e) f[e] = l[e] + r[e] * aj_lo[0];
        } else {
          inviscid_flux<FLUX>(g, l, r, aj_lo, f);
        }
#pragma unroll
        for (int e = 0; e < AGX_NEQ; ++e) {
          f[e] *= aj_lo[3];
          sFj[cur][wv][e][lane] = f[e];
          res[e] -= f[e];
        }
        sr_j = AGX_AB(16) ? 1.0 : specrad(aj_lo, aj_up);
      }
      AGX_SB_FACES;
      // publish the own entry of the next plane; the loads of the update fly
      // across the barrier
      if (!AGX_AB(64)) {
#pragma unroll
        for (int e = 0; e < AGX_NEQ; ++e) tn[e * PLANE + to] = W[H][e];
        tn[AGX_NEQ * PLANE + to] = nwi;
        tn[(AGX_NEQ + 1) * PLANE + to] = nwj;
      }
      const double vol = b.ldb(PL_VOL, qb);
      double cn[AGX_NEQ];
#pragma unroll
      for (int e = 0; e < AGX_NEQ; ++e)
        cn[e] = (FUSE == 1 && ma.mode != 0) ? b.ldb(PL_CONSN + e, qb) : 0.0;
      __syncthreads();
      if (cell) {
        if (BAL && wv < 2) {
#pragma unroll
          for (int e = 0; e < AGX_NEQ; ++e) res[e] -= sFj[cur][wv][e][lane];
        }
#pragma unroll
        for (int e = 0; e < AGX_NEQ; ++e) res[e] += sFj[cur][wv + 1][e][lane];
        if (i == itop - 1) {
#pragma unroll
          for (int e = 0; e < AGX_NEQ; ++e) res[e] += sFi[cur][wv][e];
        }
        const double sr = (sr_i + sr_j) + sr_k;
        const double dt = sp.dt_fixed > 0.0 ? sp.dt_fixed
                                            : cfl * (vol * fast_rcp(fmax(sr, 0.0)));
        const bool st_ok = !AGX_AB(4) || res[0] == 12345.678;
        if (st_ok) {
#pragma unroll
          for (int e = 0; e < AGX_NEQ; ++e) b.stb(PL_RESID + e, qb, res[e]);
          b.stb(PL_SPECRAD, qb, sr);
          if (sp.implicit) b.stb(PL_A, qb, sp.diag_add ? b.ldb(PL_A, qb) + sr : sr);
          if (!sp.viscous) b.stb(PL_DT, qb, dt);
          if (FUSE == 0 && ma.store_consn) {   // AssignSolToTimeN (procBlock.cpp:1037) rides along
            double u0[AGX_NEQ];
            prim_to_cons(g, sc, u0);
#pragma unroll
            for (int e = 0; e < AGX_NEQ; ++e) b.stb(PL_CONSN + e, qb, u0[e]);
          }
        }
        if (FUSE) {
          double u[AGX_NEQ], ns[AGX_NEQ];
          double fac = dt * fast_rcp(vol);
          if (ma.mode == 0 || FUSE == 2) {
            prim_to_cons(g, sc, u);            // U_n of this step
            if (FUSE == 2 && st_ok) {          // AssignSolToTimeN procBlock.cpp:1037
#pragma unroll
              for (int e = 0; e < AGX_NEQ; ++e) b.stb(PL_CONSN + e, qb, u[e]);
            }
          } else {
#pragma unroll
            for (int e = 0; e < AGX_NEQ; ++e) u[e] = cn[e];
          }
          if (ma.mode != 0) fac *= ma.alpha;
#pragma unroll
          for (int e = 0; e < AGX_NEQ; ++e) u[e] -= fac * res[e];
          if AGX_AB(8) {
#pragma unroll
            for (int e = 0; e < AGX_NEQ; ++e) ns[e] = u[e];
          } else {
            cons_to_prim(g, u, ns);
          }
          if (st_ok) {
#pragma unroll
            for (int e = 0; e < AGX_NEQ; ++e) b.stb(b.sn + e, qb, ns[e]);
          }
          const long lin0 = (((long)k * b.nj + j) * b.ni + i) * AGX_NEQ;
          double vm = sNorm[AGX_NEQ][wv][lane];
          long long vl = sLin[wv][lane];
#pragma unroll
          for (int e = 0; e < AGX_NEQ; ++e) {
            sNorm[e][wv][lane] += res[e] * res[e];
            if (res[e] > vm) { vm = res[e]; vl = lin0 + e; }
          }
          sNorm[AGX_NEQ][wv][lane] = vm;
          sLin[wv][lane] = vl;
        }
      }
      // advance the k window
#pragma unroll
      for (int m = 0; m < 2 * H - 1; ++m) {
#pragma unroll
        for (int e = 0; e < AGX_NEQ; ++e) W[m][e] = W[m + 1][e];
        wk[m] = wk[m + 1];
      }
#pragma unroll
      for (int e = 0; e < AGX_NEQ; ++e) W[2 * H - 1][e] = nxt[e];
      wk[2 * H - 1] = nwk;
    }
  } else {
    // ======================= halo waves =======================================
    // wave TJ  : j-flux of the tile's top face + upper halo rows of the tile
    // wave TJ+1: i-flux of the tile's right face + lower halo rows + halo columns
    const bool top = wv == TJ;
    const int gi = min(i0 + lane, b.ni + b.ng - 1);
    long qrow[H];            // global offsets (k = 0) of this wave's halo rows
    int lrow[H];
#pragma unroll
    for (int m = 0; m < H; ++m) {
      const int row = top ? TJ + H + m : m;
      const int gj = min(max(j0 - H + row, -b.ng), b.nj + b.ng - 1);
      qrow[m] = b.idx(gi, gj, 0);
      lrow[m] = row * TW + lane + H;
    }
    // halo columns (wave TJ+1 only): lane -> (row, column)
    const bool hc = !top && lane < 2 * H * TJ;
    const int cc = lane % (2 * H), crow = H + min(lane / (2 * H), TJ - 1);
    const int ccol = cc < H ? cc : 64 + cc;
    const long qcol = b.idx(min(max(i0 - H + ccol, -b.ng), b.ni + b.ng - 1),
                            min(j0 - H + crow, b.nj + b.ng - 1), 0);
    const int lcol = crow * TW + ccol;
    // flux duty
    bool do_f;
    long qf;
    int fo;                  // stencil origin in the tile plane
    if (top) {
      do_f = i0 + lane < itop;
      qf = b.idx(min(i0 + lane, b.ni), jtop, 0);
      fo = (jtop - j0 + H) * TW + lane + H;
    } else {
      do_f = lane < TJ && j0 + lane < jtop;
      qf = b.idx(itop, min(j0 + lane, b.nj), 0);
      fo = (min(lane, TJ - 1) + H) * TW + (itop - i0) + H;
    }
    {
      double* tl = &tile[0][0][0][0];
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        const double* p = b.pl(v < AGX_NEQ ? b.st + v : PL_WID + (v - AGX_NEQ));
#pragma unroll
        for (int m = 0; m < H; ++m) tl[v * PLANE + lrow[m]] = p[qrow[m] + (long)k0 * s_k];
        if (hc) tl[v * PLANE + lcol] = p[qcol + (long)k0 * s_k];
      }
    }
    __syncthreads();
    int cur = 0;
    // byte offsets inside a plane, see SlabDev::ldb
    const unsigned skb = (unsigned)(s_k * 8);
    unsigned qrowb[H];
#pragma unroll
    for (int m = 0; m < H; ++m) qrowb[m] = (unsigned)((qrow[m] + (long)(k0 + 1) * s_k) * 8);
    unsigned qcolb = (unsigned)((qcol + (long)(k0 + 1) * s_k) * 8);
    unsigned qfb = (unsigned)((qf + (long)k0 * s_k) * 8);
    unsigned qob = (unsigned)(b.idx(min(i0 + lane, b.ni - 1), min(j0 + (top ? 0 : 1), b.nj - 1), k0) * 8);
    for (int k = k0; k < k1; ++k, cur ^= 1, qcolb += skb, qfb += skb, qob += skb) {
      const double* tc = &tile[cur][0][0][0];
      double* tn = &tile[cur ^ 1][0][0][0];
      double af[4] = {0, 0, 0, 1};
      if AGX_AB(32) { __syncthreads(); continue; }
      if (do_f) b.areab(top ? 1 : 0, qfb, af);
      // prefetch the halo ring of plane k+1
      double hv[NV][H], hcv[NV];
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        const int pid = v < AGX_NEQ ? b.st + v : PL_WID + (v - AGX_NEQ);
#pragma unroll
        for (int m = 0; m < H; ++m) hv[v][m] = b.ldb(pid, qrowb[m]);
        hcv[v] = hc ? b.ldb(pid, qcolb) : 0.0;
      }
#pragma unroll
      for (int m = 0; m < H; ++m) qrowb[m] += skb;
      if (do_f) {
        double l[AGX_NEQ], r[AGX_NEQ], f[AGX_NEQ];
        const double* base = tc + fo;
        if (top) {
          recon_generic<RECON, LIM>(
              [&](int e, int m) { return base[e * PLANE + m * TW]; },
              [&](int m) { return base[(AGX_NEQ + 1) * PLANE + m * TW]; }, sp.kappa, l, r);
          inviscid_flux<FLUX>(g, l, r, af, f);
#pragma unroll
          for (int e = 0; e < AGX_NEQ; ++e) sFj[cur][jtop - j0][e][lane] = f[e] * af[3];
        } else {
          recon_generic<RECON, LIM>(
              [&](int e, int m) { return base[e * PLANE + m]; },
              [&](int m) { return base[AGX_NEQ * PLANE + m]; }, sp.kappa, l, r);
          inviscid_flux<FLUX>(g, l, r, af, f);
#pragma unroll
          for (int e = 0; e < AGX_NEQ; ++e) sFi[cur][lane][e] = f[e] * af[3];
        }
      }
      if (BAL) {
        // lower j-face flux of cell row 0 (top wave) / 1 (right wave)
        const int orow = top ? 0 : 1;
        if (i0 + lane < itop && j0 + orow < jtop) {
          double ao[4], l[AGX_NEQ], r[AGX_NEQ], f[AGX_NEQ];
          b.areab(1, qob, ao);
          const double* base = tc + (orow + H) * TW + lane + H;
          recon_generic<RECON, LIM>(
              [&](int e, int m) { return base[e * PLANE + m * TW]; },
              [&](int m) { return base[(AGX_NEQ + 1) * PLANE + m * TW]; }, sp.kappa, l, r);
          inviscid_flux<FLUX>(g, l, r, ao, f);
#pragma unroll
          for (int e = 0; e < AGX_NEQ; ++e) sFj[cur][orow][e][lane] = f[e] * ao[3];
        }
      }
#pragma unroll
      for (int v = 0; v < NV; ++v) {
#pragma unroll
        for (int m = 0; m < H; ++m) tn[v * PLANE + lrow[m]] = hv[v][m];
        if (hc) tn[v * PLANE + lcol] = hcv[v];
      }
      __syncthreads();
    }
  }
  __syncthreads();   // segment boundary: the LDS buffers are reused
  }
  if (FUSE) {
    __syncthreads();
    // fold the per-thread accumulators: wave 0 sums over the waves, then lanes
    if (wv == 0) {
      double l2[AGX_NEQ];
      double vmax = -1.0e300;
      long long vlin = 0x7fffffffffffffffLL;
#pragma unroll
      for (int e = 0; e < AGX_NEQ; ++e) l2[e] = 0.0;
      for (int w = 0; w < TJ; ++w) {
#pragma unroll
        for (int e = 0; e < AGX_NEQ; ++e) l2[e] += sNorm[e][w][lane];
        const double ov = sNorm[AGX_NEQ][w][lane];
        const long long ol = sLin[w][lane];
        if (ov > vmax || (ov == vmax && ol < vlin)) { vmax = ov; vlin = ol; }
      }
      for (int off = 32; off > 0; off >>= 1) {
#pragma unroll
        for (int e = 0; e < AGX_NEQ; ++e) l2[e] += __shfl_down(l2[e], off, 64);
        const double ov = __shfl_down(vmax, off, 64);
        const long long ol = __shfl_down(vlin, off, 64);
        if (ov > vmax || (ov == vmax && ol < vlin)) { vmax = ov; vlin = ol; }
      }
      if (lane == 0) {
        NormPartial p;
#pragma unroll
        for (int e = 0; e < AGX_NEQ; ++e) p.l2[e] = l2[e];
        p.vmax = vmax;
        p.lin = vlin;
        const long bid = ((long)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
        ma.partials[bid] = p;
      }
    }
  }
}

#endif  // AGX_FAST
// ---------------------------------------------------------------------------
// Viscous residual, one thread per cell, six faces.  Counterpart of
// procBlock::CalcViscFluxI/J/K procBlock.cpp:1233-2135 (laminar, central
// reconstruction) with CalcGradsI/J/K :5173-5786, VectorGradGG/ScalarGradGG
// utility.cpp:59-188, viscousFlux::CalcFlux viscousFlux.cpp:58-135 and
// TauNormal utility.cpp:426-437.  Temperature and viscosity are recomputed
// from the state (UpdateAuxillaryVariables procBlock.cpp:6171) -- no arrays.
__device__ __forceinline__ void area_vec(const BlockDev& b, int d, long q, double* v) {
  const double m = b.fa[d][3][q];
  v[0] = b.fa[d][0][q] * m; v[1] = b.fa[d][1][q] * m; v[2] = b.fa[d][2][q] * m;
}
__device__ __forceinline__ void uvwt(const BlockDev& b, const GasDev& g, long q, double* v) {
  const double rho = b.state[0][q];
  v[0] = b.state[1][q]; v[1] = b.state[2][q]; v[2] = b.state[3][q];
  v[3] = b.state[4][q] / (rho * g.R);
}
// state and laminar viscosity at the lower d-face of cell index qU (cells qL|qU), the second
// half of what the viscous flux and the thin-shear-layer Jacobian need
__device__ __forceinline__ void visc_face_state(const BlockDev& b, const GasDev& g, int d,
                                                long qU, bool fourth, double* sf, double& muf) {
  const long sd = b.stride(d);
  const long qL = qU - sd;
  // FaceReconCentral reconstruction.hpp:315-328: coeffs = LagrangeCoeff(
  // {wU, wD}, 1, 0, 0) = {wD, wU} / (wU + wD) and the reference forms
  // coeffs[0] * varD + coeffs[1] * varU (the wider cell gets the larger weight)
  if (fourth) {
    // FaceReconCentral4th reconstruction.hpp:335-379, LagrangeCoeff(w, 3, 1, 1);
    // state and viscosity of the four cells around the face (procBlock.cpp:1325-1346)
    // In Newton form on the divided differences of the four cell values (as the WENO
    // stencils, agx_device.hpp): with the face at the upper edge of cell 1,
    //   f = u1 + w1 G12 - w1 w2 D3a - w1 w2 (w0 + w1) D4,
    //   Gab = (ub - ua) / (wa + wb),  D3a = (G12 - G01) / (w0 + w1 + w2),
    //   D3b = (G23 - G12) / (w1 + w2 + w3),  D4 = (D3b - D3a) / (w0 + w1 + w2 + w3)
    // (uniform widths: -1/12, 7/12, 7/12, -1/12).  Formula 2.20's generic loops unrolled to
    // 3.3 KB of scratch per lane in this kernel.
    const long qs[4] = {qL - sd, qL, qU, qU + sd};
    const double w0 = b.wid[d][qs[0]], w1 = b.wid[d][qs[1]], w2 = b.wid[d][qs[2]],
                 w3 = b.wid[d][qs[3]];
    const double r01 = 1.0 / (w0 + w1), r12 = 1.0 / (w1 + w2), r23 = 1.0 / (w2 + w3);
    const double t012 = 1.0 / (w0 + w1 + w2), t123 = 1.0 / (w1 + w2 + w3);
    const double q4 = 1.0 / ((w0 + w1) + (w2 + w3));
    const double k3 = w1 * w2, k4 = k3 * (w0 + w1);
    auto c4 = [&](double u0, double u1, double u2, double u3) {
      const double g01 = (u1 - u0) * r01, g12 = (u2 - u1) * r12, g23 = (u3 - u2) * r23;
      const double d3a = (g12 - g01) * t012, d3b = (g23 - g12) * t123;
      return u1 + w1 * g12 - k3 * d3a - k4 * ((d3b - d3a) * q4);
    };
    double s4[4][AGX_NEQ], mu4[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      load5(b.state, qs[m], s4[m]);
      mu4[m] = viscosity(g, temperature(g, s4[m]));
    }
#pragma unroll
    for (int e = 0; e < AGX_NEQ; ++e) sf[e] = c4(s4[0][e], s4[1][e], s4[2][e], s4[3][e]);
    muf = c4(mu4[0], mu4[1], mu4[2], mu4[3]);
  } else {
    const double wU = b.wid[d][qL], wD = b.wid[d][qU];
    const double cD = wD / (wU + wD), cU = wU / (wU + wD);
    double sL[AGX_NEQ], sU[AGX_NEQ];
    load5(b.state, qL, sL);
    load5(b.state, qU, sU);
#pragma unroll
    for (int e = 0; e < AGX_NEQ; ++e) sf[e] = cD * sU[e] + cU * sL[e];
    muf = cD * viscosity(g, sU[4] / (sU[0] * g.R)) + cU * viscosity(g, sL[4] / (sL[0] * g.R));
  }
}
// what the viscous flux and the thin-shear-layer Jacobian need at the lower d-face of
// cell index qU (cells qL|qU): Green-Gauss gradients, face state and viscosity
__device__ __forceinline__ void visc_face_terms(const BlockDev& b, const GasDev& g, int d,
                                                long qU, bool fourth, double (*grad)[4],
                                                double* sf, double& muf) {
  const long sd = b.stride(d);
  const long qL = qU - sd;
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 4; ++c) grad[r][c] = 0.0;
  double vL[4], vU[4];
  uvwt(b, g, qL, vL);
  uvwt(b, g, qU, vU);
  {
    double a0[3], a1[3], a2[3];
    area_vec(b, d, qU, a0);
    area_vec(b, d, qU + sd, a1);
    area_vec(b, d, qU - sd, a2);
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      const double au = 0.5 * (a0[r] + a1[r]), al = 0.5 * (a0[r] + a2[r]);
#pragma unroll
      for (int c = 0; c < 4; ++c) grad[r][c] += vU[c] * au - vL[c] * al;
    }
  }
#pragma unroll
  for (int t = 0; t < 3; ++t) {
    if (t == d) continue;
    const long stt = b.stride(t);
    double a0[3], a1[3], vu[4], vl[4], t0[4], t1[4];
    area_vec(b, t, qU + stt, a0);
    area_vec(b, t, qL + stt, a1);
    uvwt(b, g, qU + stt, t0);
    uvwt(b, g, qL + stt, t1);
#pragma unroll
    for (int c = 0; c < 4; ++c) vu[c] = 0.25 * (vL[c] + vU[c] + t0[c] + t1[c]);
    double b0[3], b1[3];
    area_vec(b, t, qU, b0);
    area_vec(b, t, qL, b1);
    uvwt(b, g, qU - stt, t0);
    uvwt(b, g, qL - stt, t1);
#pragma unroll
    for (int c = 0; c < 4; ++c) vl[c] = 0.25 * (vL[c] + vU[c] + t0[c] + t1[c]);
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      const double au = 0.5 * (a0[r] + a1[r]), al = 0.5 * (b0[r] + b1[r]);
#pragma unroll
      for (int c = 0; c < 4; ++c) grad[r][c] += vu[c] * au - vl[c] * al;
    }
  }
  const double inv_vol = 1.0 / (0.5 * (b.vol[qL] + b.vol[qU]));
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 4; ++c) grad[r][c] *= inv_vol;
  visc_face_state(b, g, d, qU, fourth, sf, muf);
}
// viscous flux * |A| through the lower d-face of cell index qU (cells qL|qU)
__device__ __forceinline__ void visc_face(const BlockDev& b, const GasDev& g,
                                          int d, long qU, double* f, bool fourth = false) {
  double grad[3][4];          // [derivative direction][u, v, w, T]
  double sf[AGX_NEQ], muf;
  visc_face_terms(b, g, d, qU, fourth, grad, sf, muf);
  double n[4];
  load_area(b, d, qU, n);
  const double mu = g.scaling * muf;
  const double lambda = -(2.0 / 3.0) * mu;
  const double trace = grad[0][0] + grad[1][1] + grad[2][2];
  double tau[3];
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    const double mm = (grad[r][0] + grad[0][r]) * n[0] +
                      (grad[r][1] + grad[1][r]) * n[1] +
                      (grad[r][2] + grad[2][r]) * n[2];
    tau[r] = lambda * trace * n[r] + mu * mm;
  }
  const double kk = conductivity(g, temperature(g, sf)) * g.scaling;
  const double tg = grad[0][3] * n[0] + grad[1][3] * n[1] + grad[2][3] * n[2];
  f[0] = 0.0;
  f[1] = tau[0] * n[3];
  f[2] = tau[1] * n[3];
  f[3] = tau[2] * n[3];
  f[4] = (dot3(tau, sf + 1) + kk * tg) * n[3];
}

// Block-matrix solvers: thin-shear-layer part of the main diagonal of a cell (the
// face Jacobian with left = false is added for its lower faces, the one with
// left = true subtracted for its upper faces, procBlock.cpp:1417-1424, :1468-1475)
// and the cell's velocity gradient, one sixth of each face gradient
// (:1397, :1432), which the off-diagonal terms of the neighbours read.
__global__ void __launch_bounds__(256)
k_block_diag_visc(BlockDev b, GasDev g, SolverDev sp, int fourth) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int j = blockIdx.y * blockDim.y + threadIdx.y;
  const int k = blockIdx.z;
  if (i >= b.ni || j >= b.nj) return;
  const long q = b.idx(i, j, k);
  double D[AGX_NJ], vg[9];
#pragma unroll
  for (int e = 0; e < AGX_NJ; ++e) D[e] = b.am[(long)e * b.nplane + q];
#pragma unroll
  for (int e = 0; e < 9; ++e) vg[e] = 0.0;
  for (int d = 0; d < 3; ++d) {
    const long s = b.stride(d);
    for (int up = 0; up < 2; ++up) {
      const long qU = q + (up ? s : 0), qL = qU - s;
      double grad[3][4], sf[AGX_NEQ], muf, area[4], G[9], J[AGX_NJ];
      visc_face_terms(b, g, d, qU, fourth != 0, grad, sf, muf);
      load_area(b, d, qU, area);
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          G[3 * r + c] = grad[r][c];
          vg[3 * r + c] += (1.0 / 6.0) * grad[r][c];
        }
      const double v[3] = {b.cen[0][qU] - b.cen[0][qL], b.cen[1][qU] - b.cen[1][qL],
                           b.cen[2][qU] - b.cen[2][qL]};
      const double dist = dot3(v, area);     // ProjC2CDist procBlock.cpp:6316-6342
      // this cell is the right cell of its lower face, the left cell of its upper face
      tsl_jacobian(g, sf, muf, area, dist, up != 0, G, J);
#pragma unroll
      for (int e = 0; e < AGX_NJ; ++e) D[e] += up ? -J[e] : J[e];
    }
  }
#pragma unroll
  for (int e = 0; e < AGX_NJ; ++e) b.am[(long)e * b.nplane + q] = D[e];
#pragma unroll
  for (int e = 0; e < 9; ++e) b.vg[(long)e * b.nplane + q] = vg[e];
}

__global__ void __launch_bounds__(256)
k_visc_residual(BlockDev b, GasDev g, SolverDev sp, double cfl, int fourth) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int j = blockIdx.y * blockDim.y + threadIdx.y;
  const int k = blockIdx.z;
  if (i >= b.ni || j >= b.nj) return;
  const long q = b.idx(i, j, k);
  double res[AGX_NEQ];
  load5(b.resid, q, res);
  double sr = b.specrad[q];
  double diag = sp.implicit ? b.a[q] : 0.0;
  double sc[AGX_NEQ];
  load5(b.state, q, sc);
  const double muc = viscosity(g, temperature(g, sc));
  const double vol = b.vol[q];
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    const long s = b.stride(d);
    double f[AGX_NEQ];
    visc_face(b, g, d, q, f, fourth != 0);
#pragma unroll
    for (int e = 0; e < AGX_NEQ; ++e) res[e] += f[e];
    visc_face(b, g, d, q + s, f, fourth != 0);
#pragma unroll
    for (int e = 0; e < AGX_NEQ; ++e) res[e] -= f[e];
    // ViscCellSpectralRadius spectralRadius.hpp:94-124
    const double fmag = 0.5 * (b.fa[d][3][q] + b.fa[d][3][q + s]);
    const double vsr = visc_max_term(g, sc[0]) * visc_term(g, muc) * fmag * fmag / vol;
    sr += vsr * sp.visc_cfl_coeff;
    diag += 2.0 * vsr;
  }
  store5(b.resid, q, res);
  b.specrad[q] = sr;
  if (sp.implicit) b.a[q] = diag;
  b.dt[q] = sp.dt_fixed > 0.0 ? sp.dt_fixed : cfl * (vol / fmax(sr, 0.0));
}

#if AGX_NEQ == 7
// ---------------------------------------------------------------------------
// rans (k-omega SST 2003): viscous residual with eddy viscosity, the diffusion of k
// and omega, the turbulence source terms and the turbulence part of the spectral
// radius / scalar diagonal, one thread per cell.  procBlock::CalcViscFluxI/J/K
// (procBlock.cpp:1233-2135), viscousFlux::CalcFlux (viscousFlux.cpp:58-135),
// turbKWSst (turbulence.cpp:573-840, turbulence.hpp:489-606), CalcSrcTerms
// (procBlock.cpp:5956-6027).
constexpr double SST_BETA_STAR = 0.09, SST_SIGMA_K1 = 0.85, SST_SIGMA_K2 = 1.0,
                 SST_SIGMA_W1 = 0.5, SST_SIGMA_W2 = 0.856, SST_BETA1 = 0.075,
                 SST_BETA2 = 0.0828, SST_GAMMA1 = 5.0 / 9.0, SST_GAMMA2 = 0.44, SST_A1 = 0.31,
                 SST_KPROD2DEST = 10.0;
__device__ __forceinline__ double sst_blend(double c1, double c2, double f1) {
  return f1 * c1 + (1.0 - f1) * c2;
}
__device__ __forceinline__ double sst_cdkw(const double* s, const double* kg, const double* wg) {
  return fmax(2.0 * s[0] * SST_SIGMA_W2 / s[6] * dot3(kg, wg), 1.0e-10);
}
// turbKWSst::EddyViscAndBlending turbulence.cpp:695-727; vg[3 r + c]
__device__ inline void sst_eddy_visc_blending(const GasDev& g, const double* s, const double* vg,
                                              const double* kg, const double* wg, double mu,
                                              double wall_dist, double& mut, double& f1,
                                              double& f2) {
  const double wd = wall_dist + AGX_EPS;
  const double alpha1 = g.scaling * sqrt(s[5]) / (SST_BETA_STAR * s[6] * wd);
  const double alpha2 = g.scaling * g.scaling * 500.0 * mu / (wd * wd * s[0] * s[6]);
  const double cdkw = sst_cdkw(s, kg, wg);
  const double alpha3 = 4.0 * s[0] * SST_SIGMA_W2 * s[5] / (cdkw * wd * wd);
  const double arg1 = fmin(fmax(alpha1, alpha2), alpha3);
  f1 = tanh(pow(arg1, 4.0));
  const double arg2 = fmax(2.0 * alpha1, alpha2);
  f2 = tanh(arg2 * arg2);
  double ss = 0.0;
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const double sy = 0.5 * (vg[3 * r + c] + vg[3 * c + r]);
      ss += sy * sy;
    }
  const double strain = sqrt(2.0 * ss);
  mut = s[0] * SST_A1 * s[5] / fmax(SST_A1 * s[6], g.scaling * strain * f2);
}
// turbKWWilcox::EddyViscAndBlending turbulence.cpp:412-430 with OmegaTilda :339-351
__device__ inline void kw_eddy_visc_blending(const GasDev& g, const double* s, const double* vg,
                                             double& mut, double& f1, double& f2) {
  const double trace = vg[0] + vg[4] + vg[8];
  double ss = 0.0;
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const double a = 0.5 * (vg[3 * r + c] + vg[3 * c + r]) - (r == c ? 1.0 / 3.0 * trace : 0.0);
      ss += a * a;
    }
  const double lim = g.scaling * 0.875 * sqrt(2.0 * ss / 0.09);
  f1 = 1.0;
  f2 = 0.0;
  mut = s[0] * s[5] / fmax(s[6], lim);
}
// turbKWWilcox::Beta / FBeta / Xw / StrainKI turbulence.cpp:291-337
__device__ inline double kw_beta(const GasDev& g, const double* s, const double* vg) {
  double W[9], K[9];
  const double trace = vg[0] + vg[4] + vg[8];
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      W[3 * r + c] = 0.5 * (vg[3 * r + c] - vg[3 * c + r]);
      K[3 * r + c] = 0.5 * (vg[3 * r + c] + vg[3 * c + r] - (r == c ? trace : 0.0));
    }
  double ddot = 0.0;
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      double ww = 0.0;
#pragma unroll
      for (int m = 0; m < 3; ++m) ww += W[3 * r + m] * W[3 * m + c];
      ddot += ww * K[3 * c + r];
    }
  const double bw = 0.09 * s[6];
  const double xw = fabs(ddot / (bw * bw * bw)) * (g.scaling * g.scaling * g.scaling);
  return 0.0708 * ((1.0 + 85.0 * xw) / (1.0 + 100.0 * xw));
}
// Green-Gauss gradients of k and omega at the lower d-face of cell qU
// (ScalarGradGG utility.cpp:138-188 on the face-centred dual volume)
__device__ inline void turb_face_grads(const BlockDev& b, int d, long qU, double* kg,
                                       double* wg) {
  const long sd = b.stride(d), qL = qU - sd;
  double au[3][3], al[3][3];
  {
    double a0[3], a1[3], a2[3];
    area_vec(b, d, qU, a0); area_vec(b, d, qU + sd, a1); area_vec(b, d, qU - sd, a2);
    for (int r = 0; r < 3; ++r) { au[d][r] = 0.5 * (a0[r] + a1[r]); al[d][r] = 0.5 * (a0[r] + a2[r]); }
  }
  for (int t = 0; t < 3; ++t) {
    if (t == d) continue;
    const long st = b.stride(t);
    double a0[3], a1[3];
    area_vec(b, t, qU + st, a0); area_vec(b, t, qL + st, a1);
    for (int r = 0; r < 3; ++r) au[t][r] = 0.5 * (a0[r] + a1[r]);
    area_vec(b, t, qU, a0); area_vec(b, t, qL, a1);
    for (int r = 0; r < 3; ++r) al[t][r] = 0.5 * (a0[r] + a1[r]);
  }
  const double inv_vol = 1.0 / (0.5 * (b.vol[qL] + b.vol[qU]));
  for (int f = 0; f < 2; ++f) {
    const double* pl = b.state[5 + f];
    double vu[3], vl[3];
    const double fL = pl[qL], fU = pl[qU];
    vl[d] = fL; vu[d] = fU;
    for (int t = 0; t < 3; ++t) {
      if (t == d) continue;
      const long st = b.stride(t);
      vu[t] = 0.25 * (fL + fU + pl[qU + st] + pl[qL + st]);
      vl[t] = 0.25 * (fL + fU + pl[qU - st] + pl[qL - st]);
    }
    double* out = f == 0 ? kg : wg;
    for (int r = 0; r < 3; ++r)
      out[r] = (vu[0] * au[0][r] - vl[0] * al[0][r] + vu[1] * au[1][r] - vl[1] * al[1][r] +
                vu[2] * au[2][r] - vl[2] * al[2][r]) * inv_vol;
  }
}

template <class B>
__device__ inline const agx_bc_surface* get_bc_surface(const B& b, int i, int j, int k, int surf);

// ---- one face of the rans viscous residual -----------------------------------------------
// What the lower d-face of the cell (fi, fj, fk) (cells qL | qU; fi, fj, fk may be one past
// the block in direction d) contributes: its flux times |A|, and the face quantities whose
// mean over a cell's six faces the source terms are evaluated with (procBlock.cpp:1397-1432,
// :1462-1475 add a sixth of each to both cells).
struct RansFace {
  double f[AGX_NEQ];            // viscous flux * |A|
  double G[9], kg[3], wg[3];    // velocity, k and omega gradients
  double mut, f1, f2;           // eddy viscosity and blending functions
};
constexpr int RANS_REC = (AGX_NEQ - 1) + 18;   // doubles of a stored face (f[0] = 0 is not)
// wall-law boundary face (procBlock.cpp:1259-1299): the wall data stored by the viscous ghost
// fill give the state, the viscosities and the flux itself, unless y+ < 10 switched the face
// to the low-Re treatment
__device__ inline const WallVars* rans_wall_face(const BlockDev& b, int d, int fi, int fj,
                                                 int fk, const agx_bc_surface*& ws) {
  const int fc[3] = {fi, fj, fk};
  const int nn[3] = {b.ni, b.nj, b.nk};
  ws = nullptr;
  if (!b.wall_off || !(fc[d] == 0 || fc[d] == nn[d])) return nullptr;
  ws = get_bc_surface(b, fc[0], fc[1], fc[2], 2 * d + (fc[d] == 0 ? 1 : 2));
  if (!ws || ws->bc_type != AGX_BC_VISCOUSWALL || !ws->state.is_wall_law) return nullptr;
  const int d1 = (d + 1) % 3, d2 = (d + 2) % 3;
  const int lo[3] = {ws->imin, ws->jmin, ws->kmin}, hi[3] = {ws->imax, ws->jmax, ws->kmax};
  const WallVars* w = b.wallv + b.wall_off[ws - b.surf] +
                      (long)(fc[d2] - lo[d2]) * (hi[d1] - lo[d1]) + (fc[d1] - lo[d1]);
  return w->yplus < 10.0 ? nullptr : w;
}
// the state and laminar viscosity the flux and the Jacobians of the face are evaluated with
// (wallData::WallState wallData.cpp:299-313 at a wall-law face; state.LimitTurb elsewhere)
__device__ inline void rans_face_state(const GasDev& g, const WallVars* wl,
                                       const agx_bc_surface* ws, double* sf, double& muf) {
  if (wl) {
    muf = wl->viscosity * (1.0 / g.scaling);
    sf[0] = wl->density;
    for (int r = 0; r < 3; ++r) sf[1 + r] = ws->state.velocity[r];
    sf[4] = wl->density * g.R * wl->temperature;
    sf[5] = wl->tke; sf[6] = wl->sdr;
  } else {
    sf[5] = fmax(sf[5], AGX_TURB_MIN);
    sf[6] = fmax(sf[6], AGX_TURB_MIN);
  }
}
__device__ inline void rans_face(const BlockDev& b, const GasDev& g, int d, int fi, int fj,
                                 int fk, bool fourth, RansFace& o, double* sf, double& muf,
                                 double* n) {
  const long qU = b.idx(fi, fj, fk), qL = qU - b.stride(d);
  double grad[3][4];
  visc_face_terms(b, g, d, qU, fourth, grad, sf, muf);
  turb_face_grads(b, d, qU, o.kg, o.wg);
  load_area(b, d, qU, n);
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) o.G[3 * r + c] = grad[r][c];
  const agx_bc_surface* ws;
  const WallVars* wl = rans_wall_face(b, d, fi, fj, fk, ws);
  rans_face_state(g, wl, ws, sf, muf);
  double f[AGX_NEQ];
  if (wl) {
    const double inv_sc = 1.0 / g.scaling;
    o.f1 = 1.0; o.f2 = 1.0;
    o.mut = wl->turb_eddy_visc * inv_sc;
    // viscousFlux::CalcWallLawFlux viscousFlux.cpp:214-247 (WallSigmaK / W: sigma_k1 /
    // sigma_w1 of SST, 0 of the base class)
    const double wsk = g.wilcox ? 0.0 : SST_SIGMA_K1, wsw = g.wilcox ? 0.0 : SST_SIGMA_W1;
    f[0] = 0.0;
    for (int r = 0; r < 3; ++r) f[1 + r] = wl->shear[r];
    f[4] = dot3(wl->shear, ws->state.velocity) + wl->heat_flux;
    f[5] = (wl->viscosity + wsk * wl->turb_eddy_visc) * dot3(o.kg, n);
    f[6] = (wl->viscosity + wsw * wl->turb_eddy_visc) * dot3(o.wg, n);
  } else {
    // wall distance at the face by the two-cell rule
    const double wU = b.wid[d][qL], wD = b.wid[d][qU];
    const double cD = wD / (wU + wD), cU = wU / (wU + wD);
    double wdist = cD * b.wdist[qU] + cU * b.wdist[qL];
    if (wdist < 0.0 && wdist > -1.0e-10) wdist = 0.0;
    if (g.wilcox) kw_eddy_visc_blending(g, sf, o.G, o.mut, o.f1, o.f2);
    else sst_eddy_visc_blending(g, sf, o.G, o.kg, o.wg, muf, wdist, o.mut, o.f1, o.f2);
    // viscousFlux::CalcFlux
    const double mu = g.scaling * muf, mt = g.scaling * o.mut;
    const double lambda = -(2.0 / 3.0) * (mu + mt);
    const double trace = grad[0][0] + grad[1][1] + grad[2][2];
    double tau[3];
    for (int r = 0; r < 3; ++r) {
      const double mm = (grad[r][0] + grad[0][r]) * n[0] + (grad[r][1] + grad[1][r]) * n[1] +
                        (grad[r][2] + grad[2][r]) * n[2];
      tau[r] = lambda * trace * n[r] + (mu + mt) * mm;
    }
    const double kk = conductivity(g, temperature(g, sf)) * g.scaling;
    const double kt = mt * g.cp / g.turb_prandtl;
    const double tg = grad[0][3] * n[0] + grad[1][3] * n[1] + grad[2][3] * n[2];
    // UseUnlimitedEddyVisc (Wilcox): the k / omega diffusion takes rho k / omega
    const double mtt = g.scaling * turb_diff_visc(g, sf, o.mut);
    f[0] = 0.0;
    f[1] = tau[0]; f[2] = tau[1]; f[3] = tau[2];
    f[4] = dot3(tau, sf + 1) + (kk + kt) * tg;
    f[5] = (mu + turb_sigma_k(g, o.f1) * mtt) * dot3(o.kg, n);
    f[6] = (mu + turb_sigma_w(g, o.f1) * mtt) * dot3(o.wg, n);
  }
  for (int e = 0; e < AGX_NEQ; ++e) o.f[e] = f[e] * n[3];
}

// ---- one cell: six faces in, residual / spectral radii / diagonal / sources out ----------
struct RansCell {
  double res[AGX_NEQ], sc[AGX_NEQ];
  double sr, srt, diag, diag_t, muc, vol;
  double vgc[9], kgc[3], wgc[3], mutc, f1c, f2c;
  double mut_lo, f1_lo;
};
__device__ inline void rans_cell_begin(const BlockDev& b, const GasDev& g, const SolverDev& sp,
                                       long q, RansCell& c) {
  load5(b.resid, q, c.res);
  c.sr = b.specrad[q]; c.srt = b.specrad_t[q];
  c.diag = sp.implicit ? b.a[q] : 0.0; c.diag_t = sp.implicit ? b.a_t[q] : 0.0;
  load5(b.state, q, c.sc);
  c.muc = viscosity(g, temperature(g, c.sc));
  c.vol = b.vol[q];
  for (int e = 0; e < 9; ++e) c.vgc[e] = 0.0;
  for (int r = 0; r < 3; ++r) { c.kgc[r] = 0.0; c.wgc[r] = 0.0; }
  c.mutc = 0.0; c.f1c = 0.0; c.f2c = 0.0; c.mut_lo = 0.0; c.f1_lo = 0.0;
}
// a face joins its cell: the right cell of its lower face (+), the left of its upper (-)
__device__ inline void rans_cell_add_face(RansCell& c, const RansFace& o, bool up) {
  for (int e = 0; e < AGX_NEQ; ++e) c.res[e] += up ? -o.f[e] : o.f[e];
  for (int e = 0; e < 9; ++e) c.vgc[e] += (1.0 / 6.0) * o.G[e];
  for (int r = 0; r < 3; ++r) { c.kgc[r] += (1.0 / 6.0) * o.kg[r]; c.wgc[r] += (1.0 / 6.0) * o.wg[r]; }
  c.mutc += (1.0 / 6.0) * o.mut; c.f1c += (1.0 / 6.0) * o.f1; c.f2c += (1.0 / 6.0) * o.f2;
  if (!up) { c.mut_lo = o.mut; c.f1_lo = o.f1; }
}
// thin-shear-layer Jacobian with the eddy viscosity (flow block) and turbKWSst::ViscJac
// (turbulence block): + for both cells of a face (procBlock.cpp:1417-1424, :1468-1475; fac of
// fluxJacobian.hpp:749-757)
// (D, Dt: the cell's main-diagonal blocks, held by the caller from rans_cell_jac_begin to
// rans_cell_finish -- one load and one store per entry instead of one per face)
__device__ inline void rans_cell_face_jacobians(const BlockDev& b, const GasDev& g, long qU,
                                                long qL, bool up, const double* sf, double muf,
                                                const double* n, const RansFace& o, double* D,
                                                double* Dt) {
  const double v[3] = {b.cen[0][qU] - b.cen[0][qL], b.cen[1][qU] - b.cen[1][qL],
                       b.cen[2][qU] - b.cen[2][qL]};
  const double dist = dot3(v, n);
  double J[AGX_NJ], jk, jw;
  tsl_jacobian(g, sf, muf, n, dist, up, o.G, J, o.mut);
  for (int e = 0; e < AGX_NJ; ++e) D[e] += up ? -J[e] : J[e];
  turb_visc_jac(g, sf, n, muf, dist, o.mut, o.f1, jk, jw);
  Dt[0] += jk;
  Dt[1] += jw;
}
__device__ inline void rans_cell_jac_begin(const BlockDev& b, const SolverDev& sp, long q,
                                           double* D, double* Dt) {
  if (!(sp.implicit && sp.block)) return;
  for (int e = 0; e < AGX_NJ; ++e) D[e] = b.am[(long)e * b.nplane + q];
  Dt[0] = b.am_t[q];
  Dt[1] = b.am_t[b.nplane + q];
}
// ViscCellSpectralRadius spectralRadius.hpp:94-124 and turbKWSst::ViscousCellSpectralRadius
// turbulence.cpp:797-815 with the LOWER face's mut, f1
__device__ inline void rans_cell_direction(const BlockDev& b, const GasDev& g,
                                           const SolverDev& sp, int d, long q, RansCell& c) {
  const double fmag = 0.5 * (b.fa[d][3][q] + b.fa[d][3][q + b.stride(d)]);
  const double vsr = visc_max_term(g, c.sc[0]) *
                     (g.scaling * (c.muc * g.inv_prandtl + c.mut_lo / g.turb_prandtl)) * fmag * fmag / c.vol;
  c.sr += vsr * sp.visc_cfl_coeff;
  c.diag += 2.0 * vsr;
  const double tvsr = g.scaling * (fmag * fmag / c.vol) / c.sc[0] *
                      (c.muc + turb_sigma_k(g, c.f1_lo) * turb_diff_visc(g, c.sc, c.mut_lo));
  c.srt += tvsr * sp.visc_cfl_coeff;
  c.diag_t += 2.0 * tvsr;
}
// source terms of the cell (turbKWSst::CalcTurbSrc turbulence.cpp:637-690), then everything
// the cell stores
__device__ inline void rans_cell_finish(const BlockDev& b, const GasDev& g, const SolverDev& sp,
                                        double cfl, long q, RansCell& c, double* D, double* Dt) {
  const double* sc = c.sc;
  const double* vgc = c.vgc;
  const double vol = c.vol, mutc = c.mutc, f1c = c.f1c, f2c = c.f2c;
  {
    const double inv_sc = 1.0 / g.scaling;
    // turbSstDes::CalcTurbSrc turbulence.cpp:866-922: phi = max((1 - f2) Lt / (cdes width), 1)
    // with width = MaxCellWidth (procBlock.cpp:5993-5995) scales the k destruction
    double phi = 1.0, width = 1.0;
    if (g.sstdes) {
      width = fmax(fmax(b.wid[0][q], b.wid[1][q]), b.wid[2][q]);
      const double cdes = sst_blend(0.78, 0.61, f1c);
      const double lt = sqrt(sc[5]) / (SST_BETA_STAR * sc[6]) * g.scaling;
      phi = fmax((1.0 - f2c) * lt / (cdes * width), 1.0);
    }
    const double tke_dest = inv_sc * SST_BETA_STAR * (sc[0] * sc[5] * sc[6] * phi);
    const double lambda = -(2.0 / 3.0) * mutc;
    const double trace = vgc[0] + vgc[4] + vgc[8];
    double ddot = 0.0;
    for (int r = 0; r < 3; ++r)
      for (int cc = 0; cc < 3; ++cc) {
        const double id = r == cc ? 1.0 : 0.0;
        const double tau = lambda * trace * id + mutc * (vgc[3 * r + cc] + vgc[3 * cc + r]) -
                           2.0 / 3.0 * sc[0] * sc[5] * id;
        ddot += tau * vgc[3 * cc + r];
      }
    double beta;
    if (g.wilcox) {      // turbKWWilcox::CalcTurbSrc turbulence.cpp:359-407
      beta = kw_beta(g, sc, vgc);
      const double omg_dest = inv_sc * beta * (sc[0] * sc[6] * sc[6]);
      const double tke_prod = fmax(g.scaling * ddot, 0.0);
      const double omg_prod = fmax(0.52 * sc[6] / sc[5] * tke_prod, 0.0);
      const double kw = dot3(c.kgc, c.wgc);
      const double omg_cd = g.scaling * (kw <= 0.0 ? 0.0 : 0.125) * (sc[0] / sc[6] * kw);
      c.res[5] -= (tke_prod - tke_dest) * vol;
      c.res[6] -= (omg_prod - omg_dest + omg_cd) * vol;
    } else {
      const double cdkw = sst_cdkw(sc, c.kgc, c.wgc);
      const double gam = sst_blend(SST_GAMMA1, SST_GAMMA2, f1c);
      beta = sst_blend(SST_BETA1, SST_BETA2, f1c);
      const double omg_dest = inv_sc * beta * (sc[0] * sc[6] * sc[6]);
      const double tke_prod = fmax(fmin(g.scaling * ddot, SST_KPROD2DEST * tke_dest), 0.0);
      const double omg_prod = fmax(gam * sc[0] / mutc * tke_prod, 0.0);
      const double omg_cd = g.scaling * (1.0 - f1c) * cdkw;
      c.res[5] -= (tke_prod - tke_dest) * vol;
      c.res[6] -= (omg_prod - omg_dest + omg_cd) * vol;
    }
    double src_sr = -2.0 * SST_BETA_STAR * sc[6] * vol * inv_sc;   // SrcSpecRad :739-747
    if (g.sstdes) {
      // turbSstDes::SrcSpecRad :925-935: the larger diagonal entry of TurbSrcJac with beta2;
      // it receives the cell WIDTH in the place of phi (procBlock.cpp:5993-6004)
      const double j00 = -2.0 * SST_BETA_STAR * sc[6] * width * vol * inv_sc;
      const double j11 = -2.0 * SST_BETA2 * sc[6] * vol * inv_sc;
      src_sr = -1.0 * fmax(fabs(j00), fabs(j11));
    }
    c.srt -= src_sr;
    c.diag_t -= src_sr;
    if (sp.implicit && sp.block) {     // SubtractFromTurb(TurbSrcJac), turbulence.cpp:749-770
      Dt[0] -= -2.0 * SST_BETA_STAR * sc[6] * phi * vol * inv_sc;
      Dt[1] -= -2.0 * beta * sc[6] * vol * inv_sc;
      for (int e = 0; e < AGX_NJ; ++e) b.am[(long)e * b.nplane + q] = D[e];
      b.am_t[q] = Dt[0];
      b.am_t[b.nplane + q] = Dt[1];
    }
  }
  store5(b.resid, q, c.res);
  b.specrad[q] = c.sr;
  b.specrad_t[q] = c.srt;
  if (sp.implicit) { b.a[q] = c.diag; b.a_t[q] = c.diag_t; }
  b.dt[q] = sp.dt_fixed > 0.0 ? sp.dt_fixed : cfl * (vol / fmax(fmax(c.sr, c.srt), 0.0));
  b.turb3[0][q] = mutc; b.turb3[1][q] = f1c; b.turb3[2][q] = f2c;
  b.viscp[q] = c.muc;     // viscosity_ of this UpdateAuxillaryVariables, read next iteration
  if (sp.implicit && sp.block)     // velocityGrad_ of the cell, read by the off-diagonal terms
    for (int e = 0; e < 9; ++e) b.vg[(long)e * b.nplane + q] = vgc[e];
}

// Gather form: one thread per cell evaluates its six faces (every face twice).  Kept as the
// form the face-once kernels below are tested against (AGX_RANS_VISC=gather).
__global__ void __launch_bounds__(256)
k_visc_residual_rans(BlockDev b, GasDev g, SolverDev sp, double cfl, int fourth) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int j = blockIdx.y * blockDim.y + threadIdx.y;
  const int k = blockIdx.z;
  if (i >= b.ni || j >= b.nj) return;
  const long q = b.idx(i, j, k);
  RansCell c;
  double D[AGX_NJ], Dt[2];
  rans_cell_begin(b, g, sp, q, c);
  rans_cell_jac_begin(b, sp, q, D, Dt);
  for (int d = 0; d < 3; ++d) {
    const long s = b.stride(d);
    for (int up = 0; up < 2; ++up) {
      int fc[3] = {i, j, k};
      fc[d] += up;
      RansFace o;
      double sf[AGX_NEQ], muf, n[4];
      rans_face(b, g, d, fc[0], fc[1], fc[2], fourth != 0, o, sf, muf, n);
      rans_cell_add_face(c, o, up != 0);
      if (sp.implicit && sp.block) {
        const long qU = q + (up ? s : 0);
        rans_cell_face_jacobians(b, g, qU, qU - s, up != 0, sf, muf, n, o, D, Dt);
      }
    }
    rans_cell_direction(b, g, sp, d, q, c);
  }
  rans_cell_finish(b, g, sp, cfl, q, c, D, Dt);
}

// Face-once form (production).  k_rans_faces<D> evaluates every d-face of the block once --
// one thread per face, the ten-cell gradient stencils, pow / tanh of the blending functions
// and the flux -- and stores RANS_REC doubles per face on the lattice of (ni+1)(nj+1)(nk+1)
// points, one plane per quantity; k_rans_cells reads the six records of a cell and does
// what is left: the sums, the Jacobians of the block solvers (from the stored gradients and
// eddy viscosity; the face state is formed again, two cells), spectral radii, sources.  The
// gather form evaluates each face twice inside one 7-equation thread with six faces' live
// ranges; this one trades that for 3 x RANS_REC doubles per cell through memory.
struct RansRec {
  double* p;      // 3 x RANS_REC planes of nf doubles
  long nf;        // lattice points
  int ni1, nj1;   // lattice extents in i, j
  __device__ long at(int i, int j, int k) const { return ((long)k * nj1 + j) * ni1 + i; }
  __device__ double* plane(int d, int v) const { return p + ((long)d * RANS_REC + v) * nf; }
};
#ifndef RANS_FACES_WAVES
#define RANS_FACES_WAVES 1
#endif
#ifndef RANS_CELLS_WAVES
#define RANS_CELLS_WAVES 1
#endif
template <int D>
__global__ void __launch_bounds__(256, RANS_FACES_WAVES)
k_rans_faces(BlockDev b, GasDev g, int fourth, RansRec rec) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int j = blockIdx.y * blockDim.y + threadIdx.y;
  const int k = blockIdx.z;
  if (i >= b.ni + (D == 0) || j >= b.nj + (D == 1) || k >= b.nk + (D == 2)) return;
  RansFace o;
  double sf[AGX_NEQ], muf, n[4];
  rans_face(b, g, D, i, j, k, fourth != 0, o, sf, muf, n);
  const long at = rec.at(i, j, k);
  int v = 0;
  for (int e = 1; e < AGX_NEQ; ++e) rec.plane(D, v++)[at] = o.f[e];
  for (int e = 0; e < 9; ++e) rec.plane(D, v++)[at] = o.G[e];
  for (int r = 0; r < 3; ++r) rec.plane(D, v++)[at] = o.kg[r];
  for (int r = 0; r < 3; ++r) rec.plane(D, v++)[at] = o.wg[r];
  rec.plane(D, v++)[at] = o.mut;
  rec.plane(D, v++)[at] = o.f1;
  rec.plane(D, v++)[at] = o.f2;
}
__global__ void __launch_bounds__(256, RANS_CELLS_WAVES)
k_rans_cells(BlockDev b, GasDev g, SolverDev sp, double cfl, int fourth, RansRec rec) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int j = blockIdx.y * blockDim.y + threadIdx.y;
  const int k = blockIdx.z;
  if (i >= b.ni || j >= b.nj) return;
  const long q = b.idx(i, j, k);
  RansCell c;
  double D[AGX_NJ], Dt[2];
  rans_cell_begin(b, g, sp, q, c);
  rans_cell_jac_begin(b, sp, q, D, Dt);
  for (int d = 0; d < 3; ++d) {
    const long s = b.stride(d);
    for (int up = 0; up < 2; ++up) {
      int fc[3] = {i, j, k};
      fc[d] += up;
      const long at = rec.at(fc[0], fc[1], fc[2]);
      RansFace o;
      int v = 0;
      o.f[0] = 0.0;
      for (int e = 1; e < AGX_NEQ; ++e) o.f[e] = rec.plane(d, v++)[at];
      for (int e = 0; e < 9; ++e) o.G[e] = rec.plane(d, v++)[at];
      for (int r = 0; r < 3; ++r) o.kg[r] = rec.plane(d, v++)[at];
      for (int r = 0; r < 3; ++r) o.wg[r] = rec.plane(d, v++)[at];
      o.mut = rec.plane(d, v++)[at];
      o.f1 = rec.plane(d, v++)[at];
      o.f2 = rec.plane(d, v++)[at];
      rans_cell_add_face(c, o, up != 0);
      if (sp.implicit && sp.block) {
        const long qU = q + (up ? s : 0);
        double sf[AGX_NEQ], muf, n[4];
        visc_face_state(b, g, d, qU, fourth != 0, sf, muf);
        const agx_bc_surface* ws;
        const WallVars* wl = rans_wall_face(b, d, fc[0], fc[1], fc[2], ws);
        rans_face_state(g, wl, ws, sf, muf);
        load_area(b, d, qU, n);
        rans_cell_face_jacobians(b, g, qU, qU - s, up != 0, sf, muf, n, o, D, Dt);
      }
    }
    rans_cell_direction(b, g, sp, d, q, c);
  }
  rans_cell_finish(b, g, sp, cfl, q, c, D, Dt);
}
#endif  // AGX_NEQ == 7

// Face-once form of k_visc_residual (production): a workgroup of 64 x 8 threads
// owns a 63 x 7 column of cells and marches it along k.  Every thread evaluates
// the viscous flux of its lower i-, lower j- and upper k-face; the upper i-face
// comes from lane + 1 (wave shuffle), the upper j-face from the row above (LDS),
// the lower k-face is the upper one of the previous step (registers).  Lane 63 and
// row 7 are helpers that only produce the faces their neighbours need, so each
// face flux (gradient stencil of ten cells) is formed once instead of twice.
constexpr int VTI = 63, VTJ = 7;
__global__ void __launch_bounds__(512)
k_visc_march(BlockDev b, GasDev g, SolverDev sp, double cfl, int kchunk) {
  __shared__ double sFj[2][VTJ + 1][AGX_NEQ][64];
  const int lane = threadIdx.x, ty = threadIdx.y;
  const int i = blockIdx.x * VTI + lane, j = blockIdx.y * VTJ + ty;
  const int k0 = blockIdx.z * kchunk, k1 = min(k0 + kchunk, b.nk);
  const bool own = lane < VTI && ty < VTJ && i < b.ni && j < b.nj;
  const bool do_i = i <= b.ni && j < b.nj && ty < VTJ;      // lower i-face exists
  const bool do_j = j <= b.nj && i < b.ni && lane < VTI;    // lower j-face exists
  const int ic = min(i, b.ni), jc = min(j, b.nj);
  long q = b.idx(ic, jc, k0);
  const long sk = b.sxy;
  double fk_lo[AGX_NEQ] = {0, 0, 0, 0, 0};
  if (own) visc_face(b, g, 2, q, fk_lo);
  int cur = 0;
  for (int k = k0; k < k1; ++k, q += sk, cur ^= 1) {
    double fi[AGX_NEQ] = {0, 0, 0, 0, 0}, fj[AGX_NEQ] = {0, 0, 0, 0, 0};
    double fk_up[AGX_NEQ] = {0, 0, 0, 0, 0};
    if (do_i) visc_face(b, g, 0, q, fi);
    if (do_j) visc_face(b, g, 1, q, fj);
    if (own) visc_face(b, g, 2, q + sk, fk_up);
#pragma unroll
    for (int e = 0; e < AGX_NEQ; ++e) sFj[cur][ty][e][lane] = fj[e];
    double fi_up[AGX_NEQ];
#pragma unroll
    for (int e = 0; e < AGX_NEQ; ++e) fi_up[e] = __shfl_down(fi[e], 1, 64);
    __syncthreads();
    if (own) {
      double res[AGX_NEQ];
      load5(b.resid, q, res);
      // same accumulation order as the gather form: +lower -upper per direction
#pragma unroll
      for (int e = 0; e < AGX_NEQ; ++e) {
        res[e] += fi[e];
        res[e] -= fi_up[e];
        res[e] += fj[e];
        res[e] -= sFj[cur][ty + 1][e][lane];
        res[e] += fk_lo[e];
        res[e] -= fk_up[e];
        fk_lo[e] = fk_up[e];
      }
      double sc[AGX_NEQ];
      load5(b.state, q, sc);
      const double muc = viscosity(g, temperature(g, sc));
      const double vol = b.vol[q];
      double sr = b.specrad[q];
      double diag = sp.implicit ? b.a[q] : 0.0;
#pragma unroll
      for (int d = 0; d < 3; ++d) {
        // ViscCellSpectralRadius spectralRadius.hpp:94-124
        const double fmag = 0.5 * (b.fa[d][3][q] + b.fa[d][3][q + b.stride(d)]);
        const double vsr = visc_max_term(g, sc[0]) * visc_term(g, muc) * fmag * fmag / vol;
        sr += vsr * sp.visc_cfl_coeff;
        diag += 2.0 * vsr;
      }
      store5(b.resid, q, res);
      b.specrad[q] = sr;
      if (sp.implicit) b.a[q] = diag;
      b.dt[q] = sp.dt_fixed > 0.0 ? sp.dt_fixed : cfl * (vol / fmax(sr, 0.0));
    }
  }
}

}  // namespace agx
#if AGX_FAST
#include "agx_visc_tile.hpp"
#endif
namespace agx {

// ---------------------------------------------------------------------------
// boundary conditions
// boundaryConditions::GetBCSurface boundaryConditions.cpp:109-170
template <class B>
__device__ inline const agx_bc_surface* get_bc_surface(const B& b, int i,
                                                       int j, int k, int surf) {
  if (surf <= 2) {
    for (int n = 0; n < b.nsurf_i; ++n) {
      const agx_bc_surface* s = b.surf + n;
      if (i >= s->imin && i <= s->imax && j >= s->jmin && j < s->jmax &&
          k >= s->kmin && k < s->kmax) return s;
    }
  } else if (surf <= 4) {
    for (int n = b.nsurf_i; n < b.nsurf_i + b.nsurf_j; ++n) {
      const agx_bc_surface* s = b.surf + n;
      if (i >= s->imin && i < s->imax && j >= s->jmin && j <= s->jmax &&
          k >= s->kmin && k < s->kmax) return s;
    }
  } else {
    for (int n = b.nsurf_i + b.nsurf_j; n < b.nsurf; ++n) {
      const agx_bc_surface* s = b.surf + n;
      if (i >= s->imin && i < s->imax && j >= s->jmin && j < s->jmax &&
          k >= s->kmin && k <= s->kmax) return s;
    }
  }
  return nullptr;
}
template <class B>
__device__ __forceinline__ bool bc_is_connection(const B& b, int i, int j,
                                                 int k, int surf) {
  // the common cases (a side without any / made only of connection surfaces) are
  // answered from the kernel arguments: the surface list sits in global memory
  // and walking it is a chain of dependent loads
  const int sc = b.side_conn[surf - 1];
  if (sc != 2) return sc == 1;
  const agx_bc_surface* s = get_bc_surface(b, i, j, k, surf);
  return s && (s->bc_type == AGX_BC_INTERBLOCK || s->bc_type == AGX_BC_PERIODIC);
}
__host__ __device__ inline int surface_type(const agx_bc_surface& s) {
  if (s.imin == s.imax) return s.imax == 0 ? 1 : 2;
  if (s.jmin == s.jmax) return s.jmax == 0 ? 3 : 4;
  return s.kmax == 0 ? 5 : 6;
}

// Face ghost cells: procBlock::AssignInviscidGhostCells procBlock.cpp:2449-2532
// (viscous = 0) / AssignViscousGhostCells :2760-2838 (viscous = 1, viscousWall
// surfaces only).  All layers and surfaces are independent (they read physical
// cells only), so one launch covers every surface of the block: blockIdx.y is
// the surface, blockIdx.x * blockDim.x + threadIdx.x the ghost cell on it.
__global__ void __launch_bounds__(256)
k_bc_faces(BlockDev b, GasDev g, int viscous, int* err) {
  // blockIdx.y: the surface, blockIdx.x * blockDim.x + threadIdx.x: the cell on it; one
  // thread fills all ghost layers of its surface cell (on i-surfaces the layers and the
  // interior cells they mirror share cache lines; a thread per layer measured 104 against
  // 80 us).  Indices are formed from the three strides -- small arrays indexed by the
  // surface's direction end up in scratch memory.
  const int sn = blockIdx.y;
  const agx_bc_surface sf = b.surf[sn];
  const int st = surface_type(sf);
  const int d3 = (st - 1) / 2, d1 = (d3 + 1) % 3, d2 = (d3 + 2) % 3;
  auto pick = [](int d, int x, int y, int z) { return d == 0 ? x : (d == 1 ? y : z); };
  const int lo1 = pick(d1, sf.imin, sf.jmin, sf.kmin), hi1 = pick(d1, sf.imax, sf.jmax, sf.kmax);
  const int lo2 = pick(d2, sf.imin, sf.jmin, sf.kmin), hi2 = pick(d2, sf.imax, sf.jmax, sf.kmax);
  const int r3 = pick(d3, sf.imin, sf.jmin, sf.kmin), nn3 = pick(d3, b.ni, b.nj, b.nk);
  const int n1 = hi1 - lo1, n2 = hi2 - lo2;
  const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (long)n1 * n2) return;
  int bc = sf.bc_type;
  if (bc == AGX_BC_INTERBLOCK || bc == AGX_BC_PERIODIC) return;
  if (viscous) { if (bc != AGX_BC_VISCOUSWALL) return; }
  else if (bc == AGX_BC_VISCOUSWALL) bc = AGX_BC_SLIPWALL;
  const int rem = (int)t;
  // the fastest-varying surface direction gets consecutive lanes
  int a1, a2;
  if (d1 < d2) { a1 = lo1 + rem % n1; a2 = lo2 + rem / n1; }
  else { a2 = lo2 + rem % n2; a1 = lo1 + rem / n2; }
  const long s3 = b.stride(d3);
  const long col = b.idx(0, 0, 0) + (long)a1 * b.stride(d1) + (long)a2 * b.stride(d2);
  double area[4];
  load_area(b, d3, col + (long)r3 * s3, area);
  const int adj = st % 2 == 0 ? r3 - 1 : r3;          // the cell next to the surface
  // wall distance of the wall-adjacent cell (procBlock.cpp:2813), heat-flux walls only
  double wd = 0.0, nu_w = 0.0;
  if (bc == AGX_BC_VISCOUSWALL && (sf.state.is_heat_flux || AGX_NEQ > 5)) {
    const long qa = col + (long)adj * s3;
    wd = b.wdist[qa];
    if (AGX_NEQ > 5) nu_w = b.viscp[qa] / b.state[0][qa];
  }
  NrDev nr;
  const bool is_nr = !viscous && sf.state.is_nonreflecting && b.nr_off && b.nr_off[sn] >= 0 &&
                     (bc == AGX_BC_INLET || bc == AGX_BC_PRESSURE_OUTLET);
  if (is_nr) {
    const long qa = col + (long)adj * s3;
    nr.dt = b.dt[qa];
    double un[AGX_NEQ];
    load5(b.consn, qa, un);
    cons_to_prim(g, un, nr.sn);
    const double* gr = b.nr_grad + 12 * ((long)b.nr_off[sn] + rem);
    for (int q = 0; q < 3; ++q) nr.pg[q] = gr[q];
    for (int q = 0; q < 9; ++q) nr.vg[q] = gr[3 + q];
    nr.avg_mach = b.nr_mach[2 * sn];
    nr.max_mach = b.nr_mach[2 * sn + 1];
  }
  for (int layer = 1; layer <= b.ng; ++layer) {
    int gCell, iCell;
    if (st % 2 == 0) { gCell = r3 + layer - 1; iCell = max(r3 - layer, 0); }
    else { gCell = r3 - layer; iCell = min(r3 + layer - 1, nn3 - 1); }
    const int src = (bc == AGX_BC_SLIPWALL || viscous) ? iCell : adj;
    const long qs = col + (long)src * s3, qg = col + (long)gCell * s3;
    double in[AGX_NEQ], gh[AGX_NEQ];
    load5(b.state, qs, in);
    // wall functions: the wall data of the face belong to the first layer's call
    WallVars* wv = nullptr;
    if (AGX_NEQ > 5 && viscous && layer == 1 && sf.state.is_wall_law && b.wall_off &&
        b.wall_off[sn] >= 0)
      wv = b.wallv + b.wall_off[sn] + (long)(a2 - lo2) * n1 + (a1 - lo1);
    if (!ghost_state(g, in, bc, area, st, sf.state, layer, wd, gh, is_nr ? &nr : nullptr, nu_w,
                     wv)) {
      *err = 1;
      return;
    }
    store5(b.state, qg, gh);
  }
}

// Nonreflecting surfaces, before a ghost fill: Mach mean / maximum of the adjacent
// cells (GetGhostStates procBlock.cpp:6233-6262); one workgroup per surface
__global__ void __launch_bounds__(256) k_nr_mach(BlockDev b, GasDev g) {
  __shared__ double ssum[256], smax[256];
  const int sn = blockIdx.x;
  if (b.nr_off[sn] < 0) return;
  const agx_bc_surface sf = b.surf[sn];
  const int st = surface_type(sf);
  const int d3 = (st - 1) / 2, d1 = (d3 + 1) % 3, d2 = (d3 + 2) % 3;
  const int lo[3] = {sf.imin, sf.jmin, sf.kmin}, hi[3] = {sf.imax, sf.jmax, sf.kmax};
  const int n1 = hi[d1] - lo[d1], n2 = hi[d2] - lo[d2];
  double sum = 0.0, mx = -1.7976931348623157e308;
  for (int t = threadIdx.x; t < n1 * n2; t += 256) {
    int c[3];
    if (d1 < d2) { c[d1] = lo[d1] + t % n1; c[d2] = lo[d2] + t / n1; }
    else { c[d2] = lo[d2] + t % n2; c[d1] = lo[d1] + t / n2; }
    c[d3] = lo[d3];
    double area[4], s[AGX_NEQ];
    load_area(b, d3, b.idx(c[0], c[1], c[2]), area);
    c[d3] = st % 2 == 0 ? lo[d3] - 1 : lo[d3];
    load5(b.state, b.idx(c[0], c[1], c[2]), s);
    const double sg = st % 2 == 1 ? -1.0 : 1.0;
    const double mach = sg * dot3(s + 1, area) / sound_speed(g, s);
    sum += mach;
    mx = fmax(mx, mach);
  }
  ssum[threadIdx.x] = sum;
  smax[threadIdx.x] = mx;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if ((int)threadIdx.x < off) {
      ssum[threadIdx.x] += ssum[threadIdx.x + off];
      smax[threadIdx.x] = fmax(smax[threadIdx.x], smax[threadIdx.x + off]);
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    b.nr_mach[2 * sn] = ssum[0] / (double)(n1 * n2);
    b.nr_mach[2 * sn + 1] = smax[0];
  }
}

// Edge ghost cells: procBlock::AssignInviscidGhostCellsEdge procBlock.cpp:2565-2708
// and AssignViscousGhostCellsEdge :2874-3029.  One thread per (edge, d1); the
// ng x ng layer recursion is serial inside the thread (each layer reads the
// previous one of the same d1 only).
__global__ void k_bc_edges(BlockDev b, GasDev g, int viscous, int* err) {
  const int nn[3] = {b.ni, b.nj, b.nk};
  long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  int dd = -1, cc = 0, d1 = 0;
  for (int q = 0; q < 3; ++q) {
    if (t < 4L * nn[q]) { dd = q; cc = (int)(t / nn[q]); d1 = (int)(t % nn[q]); break; }
    t -= 4L * nn[q];
  }
  if (dd < 0) return;
  const int e2 = (dd + 1) % 3, e3 = (dd + 2) % 3;
  const int max2 = nn[e2], max3 = nn[e3];
  const bool upper2 = cc > 1, upper3 = cc % 2 == 1;
  const int surf2 = 2 * e2 + 1 + (upper2 ? 1 : 0), surf3 = 2 * e3 + 1 + (upper3 ? 1 : 0);
  const int cF22 = upper2 ? max2 : 0, cF23 = upper3 ? max3 - 1 : 0;
  const int cF32 = upper2 ? max2 - 1 : 0, cF33 = upper3 ? max3 : 0;
  int c[3];
  c[dd] = d1; c[e2] = cF22; c[e3] = cF23;
  const agx_bc_surface* s2 = get_bc_surface(b, c[0], c[1], c[2], surf2);
  c[e2] = cF32; c[e3] = cF33;
  const agx_bc_surface* s3 = get_bc_surface(b, c[0], c[1], c[2], surf3);
  int bc2 = s2 ? s2->bc_type : -1, bc3 = s3 ? s3->bc_type : -1;
  if (!viscous) {
    if (bc2 == AGX_BC_VISCOUSWALL) bc2 = AGX_BC_SLIPWALL;
    if (bc3 == AGX_BC_VISCOUSWALL) bc3 = AGX_BC_SLIPWALL;
  }
  for (int layer3 = 1; layer3 <= b.ng; ++layer3)
    for (int layer2 = 1; layer2 <= b.ng; ++layer2) {
      const int p2 = upper2 ? max2 + layer2 - 2 : 1 - layer2;
      const int g2 = upper2 ? p2 + 1 : p2 - 1;
      const int p3 = upper3 ? max3 + layer3 - 2 : 1 - layer3;
      const int g3 = upper3 ? p3 + 1 : p3 - 1;
      c[e2] = p2; c[e3] = g3; const long qP2 = b.idx(c[0], c[1], c[2]);
      c[e2] = g2; c[e3] = p3; const long qP3 = b.idx(c[0], c[1], c[2]);
      c[e2] = g2; c[e3] = g3; const long qG = b.idx(c[0], c[1], c[2]);
      double gh[AGX_NEQ], in[AGX_NEQ], area[4];
      if (bc2 == AGX_BC_SLIPWALL && bc3 != AGX_BC_SLIPWALL) {
        c[e2] = cF22; c[e3] = g3;
        load_area(b, e2, b.idx(c[0], c[1], c[2]), area);
        load5(b.state, qP2, in);
        if (!ghost_state(g, in, bc2, area, surf2, s2->state, layer2, 0.0, gh)) { *err = 1; return; }
        store5(b.state, qG, gh);
      } else if (bc2 != AGX_BC_SLIPWALL && bc3 == AGX_BC_SLIPWALL) {
        c[e2] = g2; c[e3] = cF33;
        load_area(b, e3, b.idx(c[0], c[1], c[2]), area);
        load5(b.state, qP3, in);
        if (!ghost_state(g, in, bc3, area, surf3, s3->state, layer3, 0.0, gh)) { *err = 1; return; }
        store5(b.state, qG, gh);
      } else if (!viscous || (bc2 == AGX_BC_VISCOUSWALL && bc3 == AGX_BC_VISCOUSWALL)) {
        if (layer2 == layer3) {
          double a[AGX_NEQ];
          load5(b.state, qP2, in);
          load5(b.state, qP3, a);
          for (int e = 0; e < AGX_NEQ; ++e) gh[e] = 0.5 * (in[e] + a[e]);
        } else if (layer2 > layer3) {
          load5(b.state, qP3, gh);
        } else {
          load5(b.state, qP2, gh);
        }
        store5(b.state, qG, gh);
      }
    }
}

// ---------------------------------------------------------------------------
// halo exchange: multiArray3d.hpp:790-918 (SwapSliceLocal / InsertSlice) with
// index maps precomputed on the host from GetSwapLoc
// (boundaryConditions.cpp:3006-3181).  buf is [n][ncomp].
// element q of plane e: p[e][q * stride]; rec (scatter only, may be null): the cell-major
// copy of x the plane-by-plane sweeps read (k_sweep_records), entry e of cell q at
// rec[q * SW_DYN + SW_X + e]
struct Planes5 { double* p[AGX_NEQ]; long stride; double* rec; };
constexpr int SW_GEO = 16, SW_DYN = 32, SW_RHS = 8, SW_X = 24;
// both sides of one local connection in one launch (blockIdx.y = side)
struct HaloSide { Planes5 a; const long* map; long n; double* buf; };
__global__ void k_halo_gather2(HaloSide s0, HaloSide s1) {
  const HaloSide& s = blockIdx.y == 0 ? s0 : s1;
  const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= s.n) return;
  const long q = s.map[t];
#pragma unroll
  for (int e = 0; e < AGX_NEQ; ++e) s.buf[e * s.n + t] = s.a.p[e][q * s.a.stride];
}
__global__ void k_halo_scatter2(HaloSide s0, HaloSide s1) {
  const HaloSide& s = blockIdx.y == 0 ? s0 : s1;
  const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= s.n) return;
  const long q = s.map[t];
#pragma unroll
  for (int e = 0; e < AGX_NEQ; ++e) {
    const double v = s.buf[e * s.n + t];
    s.a.p[e][q * s.a.stride] = v;
    if (s.a.rec) s.a.rec[q * SW_DYN + SW_X + e] = v;
  }
}
// all local connections of a rank in one launch per direction (blockIdx.y: side of the
// table; the table sits in device memory and is wave-uniform: scalar loads).  A 2 x 2 x 2 cube
// of blocks has twelve connections and exchanges six times per DPLUR iteration: 72 launch
// pairs of ~7.5 us each were a tenth of the iteration.
__global__ void k_halo_gather_all(const HaloSide* __restrict__ tab) {
  const HaloSide& s = tab[blockIdx.y];
  const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= s.n) return;
  const long q = s.map[t];
#pragma unroll
  for (int e = 0; e < AGX_NEQ; ++e) s.buf[e * s.n + t] = s.a.p[e][q * s.a.stride];
}
__global__ void k_halo_scatter_all(const HaloSide* __restrict__ tab) {
  const HaloSide& s = tab[blockIdx.y];
  const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= s.n) return;
  const long q = s.map[t];
#pragma unroll
  for (int e = 0; e < AGX_NEQ; ++e) {
    const double v = s.buf[e * s.n + t];
    s.a.p[e][q * s.a.stride] = v;
    if (s.a.rec) s.a.rec[q * SW_DYN + SW_X + e] = v;
  }
}
__global__ void k_halo_gather(Planes5 a, const long* __restrict__ src, long n,
                              double* __restrict__ buf) {
  const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  const long q = src[t];
#pragma unroll
  for (int e = 0; e < AGX_NEQ; ++e) buf[e * n + t] = a.p[e][q * a.stride];
}
__global__ void k_halo_scatter(Planes5 a, const long* __restrict__ dst, long n,
                               const double* __restrict__ buf) {
  const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  const long q = dst[t];
#pragma unroll
  for (int e = 0; e < AGX_NEQ; ++e) {
    const double v = buf[e * n + t];
    a.p[e][q * a.stride] = v;
    if (a.rec) a.rec[q * SW_DYN + SW_X + e] = v;
  }
}

// ---------------------------------------------------------------------------
// time advance + residual norms.  Counterpart of procBlock::UpdateBlock
// procBlock.cpp:826-872 (ExplicitEulerTimeAdvance :882, RK4TimeAdvance :927,
// ImplicitTimeAdvance :902).  Norm partials: per-block sums of r^2 per
// equation and the signed max residual with its first (k,j,i,eqn) location;
// a second kernel folds the partials in a fixed order (reproducible).

// workgroup fold of per-thread partial norms (sum of r^2 per equation, signed
// max with its first location); thread 0 writes the result
__device__ __forceinline__ void norm_block_fold(double* v, double vmax, long long lin,
                                                NormPartial* out) {
  __shared__ double sh[AGX_NEQ + 1][16];
  __shared__ long long shl[16];
  for (int off = 32; off > 0; off >>= 1) {
#pragma unroll
    for (int e = 0; e < AGX_NEQ; ++e) v[e] += __shfl_down(v[e], off, 64);
    const double ov = __shfl_down(vmax, off, 64);
    const long long ol = __shfl_down(lin, off, 64);
    if (ov > vmax || (ov == vmax && ol < lin)) { vmax = ov; lin = ol; }
  }
  const int tid = threadIdx.y * blockDim.x + threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  if (lane == 0) {
#pragma unroll
    for (int e = 0; e < AGX_NEQ; ++e) sh[e][wave] = v[e];
    sh[AGX_NEQ][wave] = vmax;
    shl[wave] = lin;
  }
  __syncthreads();
  if (tid == 0) {
    const int nw = (blockDim.x * blockDim.y + 63) >> 6;
    NormPartial p;
#pragma unroll
    for (int e = 0; e < AGX_NEQ; ++e) {
      double s = 0.0;
      for (int w = 0; w < nw; ++w) s += sh[e][w];
      p.l2[e] = s;
    }
    p.vmax = sh[AGX_NEQ][0];
    p.lin = shl[0];
    for (int w = 1; w < nw; ++w)
      if (sh[AGX_NEQ][w] > p.vmax || (sh[AGX_NEQ][w] == p.vmax && shl[w] < p.lin)) {
        p.vmax = sh[AGX_NEQ][w];
        p.lin = shl[w];
      }
    *out = p;
  }
}
__device__ __forceinline__ void norm_block_reduce(const double* r, long lin0,
                                                  bool active,
                                                  NormPartial* out) {
  double v[AGX_NEQ];
  double vmax = -1.0e300;
  long long lin = 0x7fffffffffffffffLL;
#pragma unroll
  for (int e = 0; e < AGX_NEQ; ++e) {
    v[e] = active ? r[e] * r[e] : 0.0;
    if (active && r[e] > vmax) { vmax = r[e]; lin = lin0 + e; }
  }
  norm_block_fold(v, vmax, lin, out);
}

__global__ void __launch_bounds__(256)
k_update(BlockDev b, GasDev g, SolverDev sp, int mode, double alpha, int last_mm,
         NormPartial* partials) {
  // mode: 0 explicit Euler, 1 RK stage, 2 implicit
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int j = blockIdx.y * blockDim.y + threadIdx.y;
  const int k = blockIdx.z;
  const bool active = i < b.ni && j < b.nj;
  double r[AGX_NEQ] = {0, 0, 0, 0, 0};
  if (active) {
    const long q = b.idx(i, j, k);
    load5(b.resid, q, r);
    double s[AGX_NEQ], u[AGX_NEQ], ns[AGX_NEQ];
    load5(b.state, q, s);
    if (mode == 0) {
      prim_to_cons(g, s, u);
      const double fac = b.dt[q] / b.vol[q];
#pragma unroll
      for (int e = 0; e < AGX_NEQ; ++e) u[e] -= fac * r[e];
      cons_to_prim(g, u, ns);
    } else if (mode == 1) {
      const double fac = b.dt[q] / b.vol[q] * alpha;
      load5(b.consn, q, u);
#pragma unroll
      for (int e = 0; e < AGX_NEQ; ++e) u[e] -= fac * r[e];
      cons_to_prim(g, u, ns);
    } else {
      double du[AGX_NEQ];
      load5(b.x, q, du);
      update_prim_with_cons(g, s, du, ns);
      b.a[q] = 0.0;                       // gridLevel::ResetDiagonal
      if (sp.bdf2 && last_mm) {           // gridLevel.cpp:425-428
        load5(b.consn, q, u);
        store5(b.consnm1, q, u);
      }
    }
    store5(b.state, q, ns);
  }
  const long lin0 = (((long)k * b.nj + j) * b.ni + i) * AGX_NEQ;
  const long bid = ((long)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
  norm_block_reduce(r, lin0, active, partials + bid);
}

// Workgroup g of G folds the contiguous slice g of the partials into out[g]
// (G = 1: the whole array); fixed order, hence run-to-run deterministic.
__global__ void k_norm_final(const NormPartial* partials, long n, NormPartial* out) {
  __shared__ NormPartial sh[256];
  NormPartial p;
  for (int e = 0; e < AGX_NEQ; ++e) p.l2[e] = 0.0;
  p.vmax = -1.0e300;
  p.lin = 0x7fffffffffffffffLL;
  const long per = (n + gridDim.x - 1) / gridDim.x;
  const long t0 = (long)blockIdx.x * per, t1 = min(n, t0 + per);
  for (long t = t0 + threadIdx.x; t < t1; t += blockDim.x) {
    const NormPartial o = partials[t];
    for (int e = 0; e < AGX_NEQ; ++e) p.l2[e] += o.l2[e];
    if (o.vmax > p.vmax || (o.vmax == p.vmax && o.lin < p.lin)) { p.vmax = o.vmax; p.lin = o.lin; }
  }
  sh[threadIdx.x] = p;
  __syncthreads();
  for (int off = blockDim.x / 2; off > 0; off >>= 1) {
    if ((int)threadIdx.x < off) {
      NormPartial a = sh[threadIdx.x], o = sh[threadIdx.x + off];
      for (int e = 0; e < AGX_NEQ; ++e) a.l2[e] += o.l2[e];
      if (o.vmax > a.vmax || (o.vmax == a.vmax && o.lin < a.lin)) { a.vmax = o.vmax; a.lin = o.lin; }
      sh[threadIdx.x] = a;
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) out[blockIdx.x] = sh[0];
}

// procBlock::AssignSolToTimeN / AssignSolToTimeNm1 procBlock.cpp:1037-1054
__global__ void __launch_bounds__(256)
k_store_time_n(BlockDev b, GasDev g, int also_nm1) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int j = blockIdx.y * blockDim.y + threadIdx.y;
  const int k = blockIdx.z;
  if (i >= b.ni || j >= b.nj) return;
  const long q = b.idx(i, j, k);
  double s[AGX_NEQ], u[AGX_NEQ];
  load5(b.state, q, s);
  prim_to_cons(g, s, u);
  store5(b.consn, q, u);
  if (also_nm1) store5(b.consnm1, q, u);
}

// ---------------------------------------------------------------------------
// implicit: scalar LU-SGS / DPLUR.
// b-term of linearSolver.cpp:370-374 with procBlock::SolDeltaMmN / SolDeltaNm1
// procBlock.cpp:1010-1035
__device__ __forceinline__ void rhs_b(const BlockDev& b, const GasDev& g,
                                      const SolverDev& sp, long q, double* out) {
  double r[AGX_NEQ];
  load5(b.resid, q, r);
  const double thetaInv = 1.0 / sp.theta;
  if (sp.un_is_u && !sp.bdf2) {
#pragma unroll
    for (int e = 0; e < AGX_NEQ; ++e) out[e] = -thetaInv * r[e] - 0.0;
    return;
  }
  double s[AGX_NEQ], u[AGX_NEQ], un[AGX_NEQ];
  load5(b.state, q, s);
  load5(b.consn, q, un);
  prim_to_cons(g, s, u);
  const double vdt = b.vol[q] / (b.dt[q] * sp.theta);
  const double cN = vdt * (1.0 + sp.zeta);
#pragma unroll
  for (int e = 0; e < AGX_NEQ; ++e)
    out[e] = -thetaInv * r[e] - (sp.un_is_u ? 0.0 : cN * (u[e] - un[e]));
  if (sp.bdf2) {
    double um[AGX_NEQ];
    load5(b.consnm1, q, um);
    const double cM = vdt * sp.zeta;
#pragma unroll
    for (int e = 0; e < AGX_NEQ; ++e) out[e] += cM * (un[e] - um[e]);
  }
}

// b of the LU-SGS sweeps on a multigrid level: the forcing term is part of it
// (linearSolver.cpp:377, :422; DPLUR adds it beside b, :503, and A x - b has none, :72-75)
__device__ __forceinline__ void rhs_bf(const BlockDev& b, const GasDev& g,
                                       const SolverDev& sp, long q, double* out) {
  rhs_b(b, g, sp, q, out);
  if (b.mg_forcing) {
#pragma unroll
    for (int e = 0; e < AGX_NEQ; ++e) out[e] += b.mg_forcing[(long)e * b.nplane + q];
  }
}

// procBlock::ImplicitLower / ImplicitUpper procBlock.cpp:1056-1163 with
// ProjC2CDist :6316-6342; accumulates (L - U) or parts of it.
__device__ __forceinline__ void add_off_diag(const BlockDev& b, const GasDev& g,
                                             const SolverDev& sp,
                                             double* const* x, int i, int j,
                                             int k, long q, bool lower,
                                             double sign, double* acc) {
  const int c[3] = {i, j, k};
  const int nn[3] = {b.ni, b.nj, b.nk};
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    const long s = b.stride(d);
    bool use;
    long qn, qf;
    if (lower) {
      use = c[d] > 0 || bc_is_connection(b, i, j, k, 2 * d + 1);
      qn = q - s; qf = q;
    } else {
      const int o[3] = {d == 0, d == 1, d == 2};
      use = c[d] < nn[d] - 1 ||
            bc_is_connection(b, i + o[0], j + o[1], k + o[2], 2 * d + 2);
      qn = q + s; qf = q + s;
    }
    if (!use) continue;
    double area[4], sn[AGX_NEQ], du[AGX_NEQ], od[AGX_NEQ];
    load_area(b, d, qf, area);
    load5(b.state, qn, sn);
    load5(x, qn, du);
    double dist = 1.0, mu = 0.0;
    if (sp.viscous) {
      const long ql = qf - s;   // cells across face qf: ql | qf
      const double v[3] = {b.cen[0][qf] - b.cen[0][ql], b.cen[1][qf] - b.cen[1][ql],
                           b.cen[2][qf] - b.cen[2][ql]};
      dist = dot3(v, area);
      mu = viscosity(g, temperature(g, sn));
    }
    double sd[AGX_NEQ];
    if (sp.block) {
      double vgn[9];
#pragma unroll
      for (int e = 0; e < 9; ++e) vgn[e] = sp.viscous ? b.vg[(long)e * b.nplane + qn] : 0.0;
      block_off_diagonal(g, sp.viscous != 0, sn, du, area, mu, dist, lower, vgn, od,
                         AGX_NEQ > 5 ? b.turb3[0][qn] : 0.0, AGX_NEQ > 5 ? b.turb3[1][qn] : 0.0);
    } else {
      if (sp.roe_jacobian) load5(b.state, q, sd);
      off_diagonal(g, sp.viscous, sn, du, area, mu, dist, lower, od, sp.roe_jacobian ? sd : nullptr,
                   AGX_NEQ > 5 ? b.turb3[0][qn] : 0.0, AGX_NEQ > 5 ? b.turb3[1][qn] : 0.0);
    }
#pragma unroll
    for (int e = 0; e < AGX_NEQ; ++e) acc[e] += sign * od[e];
  }
}

// The cells of a hyperplane i + j + k = p lie in different grid rows: read from the
// plane-major arrays, every load of a sweep thread is a cache line of its own (~150 per
// cell).  Before the sweeps of an iteration, what they read of a cell or of its
// neighbours is gathered once, coalesced, into cell-major records:
//   geo [16]: centre (3), the lower faces' unit normal and |A| (3 x 4)
//   dyn [32]: state (<= 7, from 0), velocityGrad_ (9, from 8), eddy viscosity and f1 (17, 18),
//             x (<= 7, from 24): written here, by the scatter of every exchange of x
//             (Planes5::rec) and by the sweeps themselves beside the plane-major x
//   rhs  [8]: the right-hand side b of the cell (k_implicit_begin)
// so that a neighbour costs three lines instead of forty-two.
// (with_geo: the geometry records are static, written by the first launch of a block only)
__global__ void __launch_bounds__(256) k_sweep_records(BlockDev b, SolverDev sp, int with_geo) {
  // 256 consecutive cells per workgroup: plane-major reads (coalesced), the records of
  // the 256 cells written as one contiguous run through LDS (row stride 33: the
  // write-out walks consecutive words)
  __shared__ double sh[256][SW_DYN + 1];
  const int tid = threadIdx.x;
  const long t0 = (long)blockIdx.x * 256, t = t0 + tid;
  const long ncell = min(256L, b.nplane - t0);
  if (with_geo) {       // (kernel argument: uniform)
    if (t < b.nplane) {
#pragma unroll
      for (int r = 0; r < 3; ++r) sh[tid][r] = b.cen[r][t];
#pragma unroll
      for (int d = 0; d < 3; ++d)
#pragma unroll
        for (int c = 0; c < 4; ++c) sh[tid][3 + 4 * d + c] = b.fa[d][c][t];
      sh[tid][15] = 0.0;
    }
    __syncthreads();
    {
      double* out = b.sw_geo + t0 * SW_GEO;
#pragma unroll
      for (int m = 0; m < SW_GEO; ++m) {
        const int idx = m * 256 + tid;
        if (idx < ncell * SW_GEO) out[idx] = sh[idx / SW_GEO][idx % SW_GEO];
      }
    }
    __syncthreads();
  }
  if (t < b.nplane) {
#pragma unroll
    for (int e = 0; e < SW_DYN; ++e) sh[tid][e] = 0.0;
#pragma unroll
    for (int e = 0; e < AGX_NEQ; ++e) sh[tid][e] = b.state[e][t];
    if (b.vg && sp.viscous) {
#pragma unroll
      for (int e = 0; e < 9; ++e) sh[tid][8 + e] = b.vg[(long)e * b.nplane + t];
    }
#if AGX_NEQ > 5
    sh[tid][17] = b.turb3[0][t];
    sh[tid][18] = b.turb3[1][t];
#endif
#pragma unroll
    for (int e = 0; e < AGX_NEQ; ++e) sh[tid][SW_X + e] = b.x[e][t];
  }
  __syncthreads();
  {
    double* out = b.sw_dyn + t0 * SW_DYN;
#pragma unroll
    for (int m = 0; m < SW_DYN; ++m) {
      const int idx = m * 256 + tid;
      if (idx < ncell * SW_DYN) out[idx] = sh[idx / SW_DYN][idx % SW_DYN];
    }
  }
}
// one direction of add_off_diag, from the records (identical arithmetic): the
// off-diagonal term of the lower / upper d-neighbour of cell q into od; false: the
// neighbour does not count (physical boundary)
// (COH: x is read with agent-scope loads that pass the caches -- the pipelined form, where
// it was written by another workgroup of the same launch)
template <bool COH = false>
__device__ __forceinline__ bool off_diag_rec_dir(const BlockDev& b, const GasDev& g,
                                                 const SolverDev& sp, int i, int j, int k,
                                                 long q, bool lower, int d, double* od) {
  const int c[3] = {i, j, k};
  const int nn[3] = {b.ni, b.nj, b.nk};
  const double* gq = b.sw_geo + q * SW_GEO;
  const long s = b.stride(d);
  bool use;
  if (lower) {
    use = c[d] > 0 || bc_is_connection(b, i, j, k, 2 * d + 1);
  } else {
    const int o[3] = {d == 0, d == 1, d == 2};
    use = c[d] < nn[d] - 1 || bc_is_connection(b, i + o[0], j + o[1], k + o[2], 2 * d + 2);
  }
  if (!use) return false;
  const long qn = lower ? q - s : q + s;
  const double* gn = b.sw_geo + qn * SW_GEO;
  const double* dn = b.sw_dyn + qn * SW_DYN;
  const double* gf = lower ? gq : gn;           // the face belongs to the upper cell
  double area[4], sn[AGX_NEQ], du[AGX_NEQ];
#pragma unroll
  for (int e = 0; e < 4; ++e) area[e] = gf[3 + 4 * d + e];
#pragma unroll
  for (int e = 0; e < AGX_NEQ; ++e) {
    sn[e] = dn[e];
    du[e] = COH ? __hip_atomic_load(dn + SW_X + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                : dn[SW_X + e];
  }
  double dist = 1.0, mu = 0.0;
  if (sp.viscous) {
    const double v[3] = {lower ? gq[0] - gn[0] : gn[0] - gq[0],
                         lower ? gq[1] - gn[1] : gn[1] - gq[1],
                         lower ? gq[2] - gn[2] : gn[2] - gq[2]};
    dist = dot3(v, area);
    mu = viscosity(g, temperature(g, sn));
  }
  const double mut_n = AGX_NEQ > 5 ? dn[17] : 0.0, f1_n = AGX_NEQ > 5 ? dn[18] : 0.0;
  if (sp.block) {
    double vgn[9];
#pragma unroll
    for (int e = 0; e < 9; ++e) vgn[e] = sp.viscous ? dn[8 + e] : 0.0;
    block_off_diagonal(g, sp.viscous != 0, sn, du, area, mu, dist, lower, vgn, od, mut_n, f1_n);
  } else {
    double sd[AGX_NEQ];
    if (sp.roe_jacobian) {
      const double* dq = b.sw_dyn + q * SW_DYN;
#pragma unroll
      for (int e = 0; e < AGX_NEQ; ++e) sd[e] = dq[e];
    }
    off_diagonal(g, sp.viscous, sn, du, area, mu, dist, lower, od,
                 sp.roe_jacobian ? sd : nullptr, mut_n, f1_n);
  }
  return true;
}
__device__ __forceinline__ void add_off_diag_rec(const BlockDev& b, const GasDev& g,
                                                 const SolverDev& sp, int i,
                                                 int j, int k, long q, bool lower, double sign,
                                                 double* acc) {
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    double od[AGX_NEQ];
    if (!off_diag_rec_dir(b, g, sp, i, j, k, q, lower, d, od)) continue;
#pragma unroll
    for (int e = 0; e < AGX_NEQ; ++e) acc[e] += sign * od[e];
  }
}

// linearSolver::AddDiagonalTerms :146-175, Invert :177-188,
// InitializeMatrixUpdate :111-144
__global__ void __launch_bounds__(256)
k_implicit_begin(BlockDev b, GasDev g, SolverDev sp, int* err, int write_x) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int j = blockIdx.y * blockDim.y + threadIdx.y;
  const int k = blockIdx.z;
  if (i >= b.ni || j >= b.nj) return;
  const long q = b.idx(i, j, k);
  if (b.sw_rhs) {       // the right-hand side, cell-major, for the plane-by-plane sweeps
    double rb[AGX_NEQ];
    rhs_bf(b, g, sp, q, rb);
#pragma unroll
    for (int e = 0; e < AGX_NEQ; ++e) b.sw_rhs[q * SW_RHS + e] = rb[e];
  }
  double dvt = (b.vol[q] * (1.0 + sp.zeta)) / (b.dt[q] * sp.theta);
  if (sp.dual_time_cfl > 0.0)    // specRadius_.Max(): the larger of the flow and turbulence parts
    dvt += fmax(fmax(b.specrad[q], AGX_NEQ > 5 ? b.specrad_t[q] : 0.0), 0.0) / sp.dual_time_cfl;
  const double a = b.a[q] * sp.relax + dvt;
  const double ainv = 1.0 / a;
  b.a[q] = a;
  b.ainv[q] = ainv;
  double ainv_t = 0.0;
  if (AGX_NEQ > 5) {
    // turbulence part of the scalar diagonal (uncoupledScalar); the dual-time term
    // uses specRadius_.Max()
    double dvt_t = (b.vol[q] * (1.0 + sp.zeta)) / (b.dt[q] * sp.theta);
    if (sp.dual_time_cfl > 0.0)
      dvt_t += fmax(fmax(b.specrad[q], b.specrad_t[q]), 0.0) / sp.dual_time_cfl;
    const double at = b.a_t[q] * sp.relax + dvt_t;
    b.a_t[q] = at;
    ainv_t = 1.0 / at;
    b.ainv_t[q] = ainv_t;
  }
  double x0[AGX_NEQ] = {0, 0, 0, 0, 0};
  if (sp.block) {
    // MultiplyOnDiagonal / AddOnDiagonal / Inverse, matMultiArray3d.hpp:96-111
    double m[AGX_NJ];
#pragma unroll
    for (int e = 0; e < AGX_NJ; ++e) m[e] = b.am[(long)e * b.nplane + q];
#pragma unroll
    for (int e = 0; e < AGX_NF; ++e) m[AGX_NF * e + e] = m[AGX_NF * e + e] * sp.relax + dvt;
#pragma unroll
    for (int e = 0; e < AGX_NF; ++e) b.am[(long)(AGX_NF * e + e) * b.nplane + q] = m[AGX_NF * e + e];
    if (!matrix_inverse5(m)) *err = 3;
#pragma unroll
    for (int e = 0; e < AGX_NJ; ++e) b.aminv[q * AGX_NJ + e] = m[e];   // (cell-major, see apply_ainv)
    double it[2] = {0.0, 0.0};
    if (AGX_NEQ > 5) {
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const double at = b.am_t[(long)e * b.nplane + q] * sp.relax + dvt;
        b.am_t[(long)e * b.nplane + q] = at;
        if (at == 0.0) *err = 3;
        it[e] = 1.0 / at;
        b.aminv_t[2 * q + e] = it[e];
      }
    }
    if (sp.requires_init) {
      double rb[AGX_NEQ];
      rhs_b(b, g, sp, q, rb);
      mat_vec5(m, rb, x0);
#pragma unroll
      for (int e = 5; e < AGX_NEQ; ++e) x0[e] = it[e - 5] * rb[e];
    }
    if (write_x) store5(b.x, q, x0);
    return;
  }
  if (!write_x) return;       // (a coarse multigrid level: x is the restricted one)
  if (sp.requires_init) {
    rhs_b(b, g, sp, q, x0);
#pragma unroll
    for (int e = 0; e < AGX_NEQ; ++e) x0[e] *= e < 5 ? ainv : ainv_t;
  }
  store5(b.x, q, x0);
}
// aInv.ArrayMult (matMultiArray3d.hpp:141-160): scalar or block
__device__ __forceinline__ void apply_ainv(const BlockDev& b, const SolverDev& sp, long q,
                                           const double* v, double* out) {
  if (sp.block) {
    // the inverse is stored cell-major (25 + 2 contiguous doubles): the cells of a
    // hyperplane lie in different rows, so every plane-major load of the sweep kernel
    // is a cache line per lane
    double m[AGX_NJ];
#pragma unroll
    for (int e = 0; e < AGX_NJ; ++e) m[e] = b.aminv[q * AGX_NJ + e];
    mat_vec5(m, v, out);
#pragma unroll
    for (int e = 5; e < AGX_NEQ; ++e) out[e] = b.aminv_t[2 * q + (e - 5)] * v[e];
  } else {
    const double ainv = b.ainv[q];
    const double ainv_t = AGX_NEQ > 5 ? b.ainv_t[q] : 0.0;
#pragma unroll
    for (int e = 0; e < AGX_NEQ; ++e) out[e] = v[e] * (e < 5 ? ainv : ainv_t);
  }
}
__global__ void k_zero5(Planes5 a, long n) {
  const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
#pragma unroll
  for (int e = 0; e < AGX_NEQ; ++e) a.p[e][t] = 0.0;
}
__global__ void k_copy5(Planes5 dst, Planes5 src, long n) {
  const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
#pragma unroll
  for (int e = 0; e < AGX_NEQ; ++e) dst.p[e][t] = src.p[e][t];
}

// One hyperplane i+j+k = p of lusgs::LUSGS_Forward linearSolver.cpp:341-383 /
// LUSGS_Backward :385-428.  Cells of a plane are mutually independent
// (HyperplaneReorder utility.cpp:377-398); planes are launched in order.
template <bool FORWARD>
__device__ __forceinline__ void lusgs_plane_cell(const BlockDev& b, const GasDev& g,
                                                 const SolverDev& sp, int plane, int full) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  const int k = blockIdx.y * blockDim.y + threadIdx.y;
  if (j >= b.nj || k >= b.nk) return;
  const int i = plane - j - k;
  if (i < 0 || i >= b.ni) return;
  const long q = b.idx(i, j, k);
  double acc[AGX_NEQ] = {0, 0, 0, 0, 0}, out[AGX_NEQ];
  const bool rec = b.sw_geo != nullptr;
  auto off = [&](bool lower, double sign) {
    if (rec) add_off_diag_rec(b, g, sp, i, j, k, q, lower, sign, acc);
    else add_off_diag(b, g, sp, b.x, i, j, k, q, lower, sign, acc);
  };
  auto rhs = [&](double* rb) {
    if (rec) {
#pragma unroll
      for (int e = 0; e < AGX_NEQ; ++e) rb[e] = b.sw_rhs[q * SW_RHS + e];
    } else {
      rhs_bf(b, g, sp, q, rb);
    }
  };
  if (FORWARD) {
    off(true, 1.0);
    if (full) off(false, -1.0);
    double rb[AGX_NEQ];
    rhs(rb);
#pragma unroll
    for (int e = 0; e < AGX_NEQ; ++e) acc[e] = rb[e] + acc[e];
    apply_ainv(b, sp, q, acc, out);
  } else {
    off(false, -1.0);
    if (full) {
      off(true, 1.0);
      double rb[AGX_NEQ];
      rhs(rb);
#pragma unroll
      for (int e = 0; e < AGX_NEQ; ++e) acc[e] = rb[e] + acc[e];
      apply_ainv(b, sp, q, acc, out);
    } else {
      double xo[AGX_NEQ];
      if (rec) {
#pragma unroll
        for (int e = 0; e < AGX_NEQ; ++e) xo[e] = b.sw_dyn[q * SW_DYN + SW_X + e];
      } else {
        load5(b.x, q, xo);
      }
      apply_ainv(b, sp, q, acc, out);     // acc = -U
#pragma unroll
      for (int e = 0; e < AGX_NEQ; ++e) out[e] = xo[e] + out[e];
    }
  }
  if (rec) {
#pragma unroll
    for (int e = 0; e < AGX_NEQ; ++e) b.sw_dyn[q * SW_DYN + SW_X + e] = out[e];
  }
  store5(b.x, q, out);
}
template <bool FORWARD>
__global__ void __launch_bounds__(256) k_lusgs_plane(BlockDev b, GasDev g, SolverDev sp, int plane, int full) {
  lusgs_plane_cell<FORWARD>(b, g, sp, plane, full);
}
// The same step with THREE lanes per cell (records only): lane 3c + d of a wave forms the
// off-diagonal terms of direction d of cell c -- the long part of the step, three of them
// in a row per thread otherwise -- the lane of d = 0 adds them up in the order of the
// one-lane form (lower i, j, k, then upper i, j, k), applies the inverse and stores.
// 21 cells per wave row (lane 63 idles): grid.x = ceil(nj / 21).
constexpr int PL3_CELLS = 21;
// lane d of the three lanes of cell (i, j, k); inactive lanes only take part in the shuffles
template <bool FORWARD, bool COH>
__device__ __forceinline__ void lusgs_cell3(const BlockDev& b, const GasDev& g,
                                            const SolverDev& sp, int i, int j, int k,
                                            bool active, int d, int full) {
  const long q = active ? b.idx(i, j, k) : 0;
  // first the side the sweep comes from, then (both triangles) the other one
  double v1[AGX_NEQ], v2[AGX_NEQ];
#pragma unroll
  for (int e = 0; e < AGX_NEQ; ++e) { v1[e] = 0.0; v2[e] = 0.0; }
  if (active) {
    double od[AGX_NEQ];
    if (off_diag_rec_dir<COH>(b, g, sp, i, j, k, q, FORWARD, d, od)) {
#pragma unroll
      for (int e = 0; e < AGX_NEQ; ++e) v1[e] = (FORWARD ? 1.0 : -1.0) * od[e];
    }
    if (full && off_diag_rec_dir<COH>(b, g, sp, i, j, k, q, !FORWARD, d, od)) {
#pragma unroll
      for (int e = 0; e < AGX_NEQ; ++e) v2[e] = (FORWARD ? -1.0 : 1.0) * od[e];
    }
  }
  double acc[AGX_NEQ];
#pragma unroll
  for (int e = 0; e < AGX_NEQ; ++e) {
    const double a1 = __shfl_down(v1[e], 1, 64), a2 = __shfl_down(v1[e], 2, 64);
    acc[e] = ((0.0 + v1[e]) + a1) + a2;
  }
  if (full) {
#pragma unroll
    for (int e = 0; e < AGX_NEQ; ++e) {
      const double a1 = __shfl_down(v2[e], 1, 64), a2 = __shfl_down(v2[e], 2, 64);
      acc[e] = ((acc[e] + v2[e]) + a1) + a2;
    }
  }
  if (!active || d != 0) return;
  double out[AGX_NEQ];
  double* xq = b.sw_dyn + q * SW_DYN + SW_X;
  if (FORWARD || full) {
#pragma unroll
    for (int e = 0; e < AGX_NEQ; ++e) acc[e] = b.sw_rhs[q * SW_RHS + e] + acc[e];
    apply_ainv(b, sp, q, acc, out);
  } else {
    double xo[AGX_NEQ];
#pragma unroll
    for (int e = 0; e < AGX_NEQ; ++e)
      xo[e] = COH ? __hip_atomic_load(xq + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : xq[e];
    apply_ainv(b, sp, q, acc, out);     // acc = -U
#pragma unroll
    for (int e = 0; e < AGX_NEQ; ++e) out[e] = xo[e] + out[e];
  }
#pragma unroll
  for (int e = 0; e < AGX_NEQ; ++e) {
    if (COH) __hip_atomic_store(xq + e, out[e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else xq[e] = out[e];
  }
  store5(b.x, q, out);
}
template <bool FORWARD>
__device__ __forceinline__ void lusgs_plane_cell3(const BlockDev& b, const GasDev& g,
                                                  const SolverDev& sp, int plane, int full) {
  const int lane = threadIdx.x;
  const int c = lane / 3, d = lane - 3 * c;
  const int j = blockIdx.x * PL3_CELLS + c;
  const int k = blockIdx.y * blockDim.y + threadIdx.y;
  const int i = plane - j - k;
  const bool active = c < PL3_CELLS && j < b.nj && k < b.nk && i >= 0 && i < b.ni;
  lusgs_cell3<FORWARD, false>(b, g, sp, i, j, k, active, d, full);
}
template <bool FORWARD>
__global__ void __launch_bounds__(256) k_lusgs_plane3(BlockDev b, GasDev g, SolverDev sp, int plane, int full) {
  lusgs_plane_cell3<FORWARD>(b, g, sp, plane, full);
}
template <bool FORWARD>
__global__ void __launch_bounds__(256)
k_lusgs_plane_all3(const BlockDev* tab, GasDev g, SolverDev sp, int t, int full) {
  const BlockDev& b = tab[blockIdx.z];
  const int nplanes = b.ni + b.nj + b.nk - 2;
  if (t >= nplanes) return;
  lusgs_plane_cell3<FORWARD>(b, g, sp, FORWARD ? t : nplanes - 1 - t, full);
}

// step t of the half sweeps of ALL blocks of the rank in one launch (blockIdx.z: block;
// the blocks of a half sweep are independent, ghost x comes from the exchange before it):
// block n is at its hyperplane t, or nplanes_n - 1 - t going back, and idles once it is done
template <bool FORWARD>
__global__ void __launch_bounds__(256)
k_lusgs_plane_all(const BlockDev* tab, GasDev g, SolverDev sp, int t, int full) {
  const BlockDev& b = tab[blockIdx.z];
  const int nplanes = b.ni + b.nj + b.nk - 2;
  if (t >= nplanes) return;
  lusgs_plane_cell<FORWARD>(b, g, sp, FORWARD ? t : nplanes - 1 - t, full);
}


// ---------------------------------------------------------------------------
// The same half sweep as ONE launch: a workgroup per k-plane of every block, marching the
// anti-diagonals s = i + j of its plane -- cell (i, j, k) is step s of plane k, which is
// hyperplane order: its in-plane neighbours are the workgroup's own previous step, the third
// one is step s of the plane below (above, going back), finished by that plane's workgroup.
// The planes of a block therefore run as a pipeline one step apart; a rank's blocks side by
// side (4 x 64 planes = 256 workgroups for BASELINE configs[4]).  Every cell gets the x of
// exactly the neighbours the plane-per-launch forms give it: bit-identical results
// (test_plane_sweep_forms_agree_bitwise); one launch instead of 318 per half sweep.
//
// What a step costs (trace: -DAGX_PIPE_TRACE, tools/pipe_trace_summary.py; rans4, BLU-SGS):
// the hand-off 0.5 us, the cells 4 - 5 us while one or two waves have any and 12 us with all
// seven -- ~ 1000 dependent fp64 instructions per lane (the flux and thin-shear-layer
// Jacobians of the neighbour, 34 divisions) behind two or three memory round trips, on one
// or two waves per SIMD: the half sweep takes what 318 launches took, 3.9 against 4.1 ms.
// Measured and dropped in round 3: the six off-diagonal matrices of a cell formed once per
// iteration by a parallel kernel and multiplied here (7 GB written and read again per
// iteration: 0.65 ms per block to form them and a sweep bound by the 256-byte records, 4.6
// ms); the neighbours' records requested a step ahead into registers, four lanes per cell
// (the step is bound by the chain of dependent instructions, not by the loads: 5.1 ms);
// touching the next step's lines (+ 0.3 ms).
//   * x is written with agent-scope (write-through) stores and read with agent-scope loads;
//     what else a cell reads does not change during the launch.
//   * hand-off between planes: a workgroup waits for its stores' acknowledgement, passes a
//     barrier and publishes "step t done" in its plane's progress word (a 128-byte line of
//     its own); the plane above polls that word.  Progress values and tickets only grow
//     (launch serial x steps), so nothing is reset between launches.
//   * planes are handed out by ticket in pipeline order: a plane's predecessor has a lower
//     ticket, so it is running or done whatever the dispatch order; a wait longer than
//     AGX_SPIN_LIMIT polls raises the error flag and every workgroup leaves.
struct PipeJob { int block, k; };
struct PipeArgs {
  const PipeJob* jobs;      // pipeline order of this direction
  const int* slot0;         // per block: progress slot of its plane 0
  long long* progress;      // 16 words per slot
  unsigned long long* ticket;
  long long base;           // progress of step t of this launch: base + t + 1
  unsigned long long tbase; // first ticket of this launch
  int njobs, spin_limit;
  int* err;
  long long* trace;         // -DAGX_PIPE_TRACE builds: 5 timestamps per step of job njobs / 2
};
#ifdef AGX_PIPE_TRACE
#define PIPE_STAMP(n) do { if (lane == 0 && wv == 0 && pa.trace && job == pa.njobs / 2) \
    pa.trace[5 * t + (n)] = wall_clock64(); } while (0)
#else
#define PIPE_STAMP(n) do {} while (0)
#endif
template <bool FORWARD>
__global__ void __launch_bounds__(512)
k_lusgs_pipe(const BlockDev* tab, GasDev g, SolverDev sp, int full, PipeArgs pa) {
  __shared__ int s_job, s_bad;
  const int lane = threadIdx.x, wv = threadIdx.y, nw = blockDim.y;
  if (lane == 0 && wv == 0) {
    s_job = (int)(atomicAdd(pa.ticket, 1ULL) - pa.tbase);
    s_bad = 0;
  }
  __syncthreads();
  // (the job is the same for the whole workgroup: said so, the block's descriptor is read
  // with scalar loads)
  const int job = __builtin_amdgcn_readfirstlane(s_job);
  if (job < 0 || job >= pa.njobs) return;
  const PipeJob jb = pa.jobs[job];
  const BlockDev& b = tab[__builtin_amdgcn_readfirstlane(jb.block)];
  const int k = __builtin_amdgcn_readfirstlane(jb.k), kp = FORWARD ? k - 1 : k + 1;
  const bool has_pred = kp >= 0 && kp < b.nk;
  const long long* ppred = pa.progress + (long)(pa.slot0[jb.block] + (has_pred ? kp : k)) * 16;
  long long* pown = pa.progress + (long)(pa.slot0[jb.block] + k) * 16;
  const int nsteps = b.ni + b.nj - 1;
  const int c = lane / 3, d = lane - 3 * c;
  for (int t = 0; t < nsteps; ++t) {
    const int s = FORWARD ? t : nsteps - 1 - t;
    bool ok = true;
    PIPE_STAMP(0);
    if (has_pred) {
      int spins = 0;
      while (__hip_atomic_load(ppred, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < pa.base + t + 1) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > pa.spin_limit ||
            ((spins & 63) == 0 &&
             __hip_atomic_load(pa.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) {
          ok = false;
          break;
        }
      }
    }
    PIPE_STAMP(1);
    if (ok) {
      const int ilo = max(0, s - (b.nj - 1)), ncell = min(b.ni - 1, s) - ilo + 1;
      for (int c0 = wv * PL3_CELLS; c0 < ncell; c0 += nw * PL3_CELLS) {
        const int i = ilo + c0 + c;
        lusgs_cell3<FORWARD, true>(b, g, sp, i, s - i, k, c < PL3_CELLS && c0 + c < ncell, d, full);
      }
    } else {
      s_bad = 1;
      __hip_atomic_store(pa.err, 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    PIPE_STAMP(2);
    __builtin_amdgcn_s_waitcnt(0x0F70);     // this wave's stores of x have been acknowledged
    PIPE_STAMP(3);
    __syncthreads();
    PIPE_STAMP(4);
    if (s_bad) return;
    if (lane == 0 && wv == 0)
      __hip_atomic_store(pown, pa.base + t + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

}  // namespace agx
#if AGX_FAST
#include "agx_lusgs_kernels.hpp"
#endif
namespace agx {

// MODE of the kernels below: the solver's run-time switches as compile-time constants for the
// two common cases (0: scalar diagonal, inviscid, rusanov; 1: scalar, viscous, rusanov) --
// the code of the other paths (block matrices, thin-shear-layer terms, approximateRoe) drops
// out and with it most of the registers; 2: as configured.
template <int MODE>
__device__ __forceinline__ SolverDev solver_mode(SolverDev sp) {
  if (MODE < 2) { sp.block = 0; sp.roe_jacobian = 0; sp.viscous = MODE; }
  return sp;
}
inline int solver_mode_of(const SolverDev& sp) {
  return (!sp.block && !sp.roe_jacobian) ? (sp.viscous ? 1 : 0) : 2;
}
// dplur::DPLUR linearSolver.cpp:473-507 (point Jacobi on the copied xold)
__device__ __forceinline__ void dplur_cell(const BlockDev& b, const GasDev& g,
                                           const SolverDev& sp, int i, int j, int k) {
  const long q = b.idx(i, j, k);
  double acc[AGX_NEQ] = {0, 0, 0, 0, 0}, rb[AGX_NEQ];
  add_off_diag(b, g, sp, b.xold, i, j, k, q, true, 1.0, acc);
  add_off_diag(b, g, sp, b.xold, i, j, k, q, false, -1.0, acc);
  rhs_b(b, g, sp, q, rb);
  double out[AGX_NEQ];
  // b + forcing + offDiagonal (linearSolver.cpp:503; the forcing term of a coarse multigrid
  // level)
#pragma unroll
  for (int e = 0; e < AGX_NEQ; ++e)
    acc[e] = (b.mg_forcing ? rb[e] + b.mg_forcing[(long)e * b.nplane + q] : rb[e]) + acc[e];
  apply_ainv(b, sp, q, acc, out);
  store5(b.x, q, out);
}
// wait_mask (bit surface type - 1; 1, 2: i lower / upper, 3, 4: j, 5, 6: k): the block faces
// whose ghost cells are still travelling -- connections to other ranks.  k_dplur leaves out the
// cells next to them (0: every cell), k_dplur_shell relaxes those afterwards.
__device__ __forceinline__ bool dplur_waits(const BlockDev& b, int m, int i, int j, int k) {
  return ((m & 1) && i == 0) || ((m & 2) && i == b.ni - 1) || ((m & 4) && j == 0) ||
         ((m & 8) && j == b.nj - 1) || ((m & 16) && k == 0) || ((m & 32) && k == b.nk - 1);
}
template <int MODE>
__global__ void __launch_bounds__(256)
k_dplur(BlockDev b, GasDev g, SolverDev sp_, int wait_mask) {
  const SolverDev sp = solver_mode<MODE>(sp_);
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int j = blockIdx.y * blockDim.y + threadIdx.y;
  const int k = blockIdx.z;
  if (i >= b.ni || j >= b.nj) return;
  if (wait_mask && dplur_waits(b, wait_mask, i, j, k)) return;
  dplur_cell(b, g, sp, i, j, k);
}
// blockIdx.y: surface type - 1; blockIdx.x * 256 + threadIdx.x: cell of that face (i fastest
// where i runs).  A cell on several waiting faces belongs to the one with the highest type.
template <int MODE>
__global__ void __launch_bounds__(256)
k_dplur_shell(BlockDev b, GasDev g, SolverDev sp_, int wait_mask) {
  const SolverDev sp = solver_mode<MODE>(sp_);
  const int f = blockIdx.y;
  if (!((wait_mask >> f) & 1)) return;
  const long t = (long)blockIdx.x * 256 + threadIdx.x;
  int i, j, k;
  if (f >= 4) {
    if (t >= (long)b.ni * b.nj) return;
    i = (int)(t % b.ni); j = (int)(t / b.ni); k = f == 5 ? b.nk - 1 : 0;
  } else if (f >= 2) {
    if (t >= (long)b.ni * b.nk) return;
    i = (int)(t % b.ni); k = (int)(t / b.ni); j = f == 3 ? b.nj - 1 : 0;
  } else {
    if (t >= (long)b.nj * b.nk) return;
    j = (int)(t % b.nj); k = (int)(t / b.nj); i = f == 1 ? b.ni - 1 : 0;
  }
  // (faces of a higher type that wait as well and hold this cell relax it)
  if (dplur_waits(b, wait_mask & ~((2 << f) - 1), i, j, k)) return;
  dplur_cell(b, g, sp, i, j, k);
}

// linearSolver::AXmB :58-90 / Residual :92-109 as a pure reduction
template <int MODE>
__global__ void __launch_bounds__(256)
k_matrix_resid(BlockDev b, GasDev g, SolverDev sp_, NormPartial* partials) {
  const SolverDev sp = solver_mode<MODE>(sp_);
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int j = blockIdx.y * blockDim.y + threadIdx.y;
  const int k = blockIdx.z;
  const bool active = i < b.ni && j < b.nj;
  double r[AGX_NEQ] = {0, 0, 0, 0, 0};
  if (active) {
    const long q = b.idx(i, j, k);
    double acc[AGX_NEQ] = {0, 0, 0, 0, 0}, rb[AGX_NEQ], xc[AGX_NEQ];
    add_off_diag(b, g, sp, b.x, i, j, k, q, true, 1.0, acc);
    add_off_diag(b, g, sp, b.x, i, j, k, q, false, -1.0, acc);
    rhs_b(b, g, sp, q, rb);
    load5(b.x, q, xc);
    double ax[AGX_NEQ];
    if (sp.block) {
      double m[AGX_NJ];
#pragma unroll
      for (int e = 0; e < AGX_NJ; ++e) m[e] = b.am[(long)e * b.nplane + q];
      mat_vec5(m, xc, ax);
#pragma unroll
      for (int e = 5; e < AGX_NEQ; ++e) ax[e] = b.am_t[(long)(e - 5) * b.nplane + q] * xc[e];
    } else {
      // (a block on the diagonal-ordered path: the finished diagonal is in the aInv_ plane,
      // k_lusgs_prepare)
      const double a = b.d2.base ? b.ainv[q] : b.a[q];
      const double at = AGX_NEQ > 5 ? b.a_t[q] : 0.0;
#pragma unroll
      for (int e = 0; e < AGX_NEQ; ++e) ax[e] = xc[e] * (e < 5 ? a : at);
    }
    // matrix residual = f - (A x - b), linearSolver.cpp:92-109
#pragma unroll
    for (int e = 0; e < AGX_NEQ; ++e) {
      const double axmb = ax[e] - acc[e] - rb[e];
      r[e] = b.mg_forcing ? b.mg_forcing[(long)e * b.nplane + q] - axmb : -axmb;
      if (b.mg_mres) b.mg_mres[(long)e * b.nplane + q] = r[e];
    }
  }
  const long bid = ((long)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
  norm_block_reduce(r, 0, active, partials + bid);
}

// ---- geometric multigrid: transfers between a level and the next coarser one ---------------
// (gridLevel.cpp:538-611; a level is a context of its own, include/aither_gfx950.h)
struct MgMap {
  const int* tc;        // fine cell -> coarse cell, [nk][nj][ni][3]
  const int* start[3];  // per direction: first fine index of coarse cell c, [n_coarse + 1]
  const double* vf;     // volume weights [nk][nj][ni]
  const double* cf;     // trilinear coefficients [nk][nj][ni][7]
};
// BlockRestriction procBlock.hpp:636-690: a coarse cell adds up its fine cells in k, j, i
// order (the order in which the reference's loop over the fine cells reaches them).
// what 0: state, volume weighted; 1: x, volume weighted; 2: matrix residual -> forcing, summed
__global__ void __launch_bounds__(256)
k_mg_restrict(BlockDev f, BlockDev c, MgMap m, int what) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int j = blockIdx.y * blockDim.y + threadIdx.y;
  const int k = blockIdx.z;
  if (i >= c.ni || j >= c.nj) return;
  double acc[AGX_NEQ];
#pragma unroll
  for (int e = 0; e < AGX_NEQ; ++e) acc[e] = 0.0;
  for (int fk = m.start[2][k]; fk < m.start[2][k + 1]; ++fk)
    for (int fj = m.start[1][j]; fj < m.start[1][j + 1]; ++fj)
      for (int fi = m.start[0][i]; fi < m.start[0][i + 1]; ++fi) {
        const long q = f.idx(fi, fj, fk);
        const long p = ((long)fk * f.nj + fj) * f.ni + fi;
        const double w = what == 2 ? 1.0 : m.vf[p];
#pragma unroll
        for (int e = 0; e < AGX_NEQ; ++e) {
          const double v = what == 0 ? f.state[e][q]
                                     : (what == 1 ? f.x[e][q] : f.mg_mres[(long)e * f.nplane + q]);
          acc[e] = what == 2 ? acc[e] + v : acc[e] + w * v;
        }
      }
  const long qc = c.idx(i, j, k);
#pragma unroll
  for (int e = 0; e < AGX_NEQ; ++e) {
    if (what == 0) c.state[e][qc] = acc[e];
    else if (what == 1) c.x[e][qc] = acc[e];
    else c.mg_forcing[(long)e * c.nplane + qc] = acc[e];
  }
}
// forcing += A x - b of the level (linearSolver::AXmB :58-90, gridLevel.cpp:579-589)
__global__ void __launch_bounds__(256)
k_mg_axmb(BlockDev b, GasDev g, SolverDev sp) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int j = blockIdx.y * blockDim.y + threadIdx.y;
  const int k = blockIdx.z;
  if (i >= b.ni || j >= b.nj) return;
  const long q = b.idx(i, j, k);
  double acc[AGX_NEQ] = {0, 0, 0, 0, 0}, rb[AGX_NEQ], xc[AGX_NEQ];
  add_off_diag(b, g, sp, b.x, i, j, k, q, true, 1.0, acc);
  add_off_diag(b, g, sp, b.x, i, j, k, q, false, -1.0, acc);
  rhs_b(b, g, sp, q, rb);
  load5(b.x, q, xc);
  double ax[AGX_NEQ];
  if (sp.block) {
    double m[AGX_NJ];
#pragma unroll
    for (int e = 0; e < AGX_NJ; ++e) m[e] = b.am[(long)e * b.nplane + q];
    mat_vec5(m, xc, ax);
#pragma unroll
    for (int e = 5; e < AGX_NEQ; ++e) ax[e] = b.am_t[(long)(e - 5) * b.nplane + q] * xc[e];
  } else {
    const double a = b.d2.base ? b.ainv[q] : b.a[q];
    const double at = AGX_NEQ > 5 ? b.a_t[q] : 0.0;
#pragma unroll
    for (int e = 0; e < AGX_NEQ; ++e) ax[e] = xc[e] * (e < 5 ? a : at);
  }
#pragma unroll
  for (int e = 0; e < AGX_NEQ; ++e) {
    double* fo = b.mg_forcing + (long)e * b.nplane + q;
    *fo = (ax[e] - acc[e] - rb[e]) + *fo;
  }
  if (b.sw_rhs) {       // (the record sweeps read b from their records)
    rhs_b(b, g, sp, q, rb);
#pragma unroll
    for (int e = 0; e < AGX_NEQ; ++e)
      b.sw_rhs[q * SW_RHS + e] = rb[e] + b.mg_forcing[(long)e * b.nplane + q];
  }
}
// the records' copy of x after something else than a sweep or an exchange wrote the planes
__global__ void k_mg_x_records(BlockDev b) {
  const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= b.nplane) return;
#pragma unroll
  for (int e = 0; e < AGX_NEQ; ++e) b.sw_dyn[t * SW_DYN + SW_X + e] = b.x[e][t];
}
// x -= coarseDu, ghost cells included (linearSolver::SubtractFromUpdate :120-126);
// save: coarseDu = x
__global__ void k_mg_axpy(BlockDev b, int save) {
  const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= b.nplane) return;
#pragma unroll
  for (int e = 0; e < AGX_NEQ; ++e) {
    double* s = b.mg_xsave + (long)e * b.nplane + t;
    if (save) *s = b.x[e][t];
    else b.x[e][t] -= *s;
  }
}
// ConvertCellToNode(coarse x, ignoreEdge, ignoreGhosts) utility.hpp:186-330: a node adds up
// the physical cells around it in the order the reference's cell loop reaches them (k, j, i
// ascending), times 1 at the block's corners, 1/2 on its edges, 1/8 elsewhere
__global__ void __launch_bounds__(256)
k_mg_nodes(BlockDev b, double* nodes) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int j = blockIdx.y * blockDim.y + threadIdx.y;
  const int k = blockIdx.z;
  if (i > b.ni || j > b.nj) return;
  double acc[AGX_NEQ];
#pragma unroll
  for (int e = 0; e < AGX_NEQ; ++e) acc[e] = 0.0;
  for (int dk = -1; dk <= 0; ++dk)
    for (int dj = -1; dj <= 0; ++dj)
      for (int di = -1; di <= 0; ++di) {
        const int ci = i + di, cj = j + dj, ck = k + dk;
        if (ci < 0 || ci >= b.ni || cj < 0 || cj >= b.nj || ck < 0 || ck >= b.nk) continue;
        const long q = b.idx(ci, cj, ck);
#pragma unroll
        for (int e = 0; e < AGX_NEQ; ++e) acc[e] += b.x[e][q];
      }
  const bool xi = i == 0 || i == b.ni, xj = j == 0 || j == b.nj, xk = k == 0 || k == b.nk;
  const double fac = (xi && xj && xk) ? 1.0 : ((xj && xk) || (xi && xk) || (xi && xj)) ? 0.5 : 0.125;
  const long nn = (long)(b.ni + 1) * (b.nj + 1) * (b.nk + 1);
  const long at = ((long)k * (b.nj + 1) + j) * (b.ni + 1) + i;
#pragma unroll
  for (int e = 0; e < AGX_NEQ; ++e) nodes[(long)e * nn + at] = acc[e] * fac;
}
// BlockProlongation gridLevel.hpp:159-214 + AddToUpdate: fine x += TrilinearInterp of the
// coarse nodes (utility.hpp:341-372)
__global__ void __launch_bounds__(256)
k_mg_prolong(BlockDev f, MgMap m, const double* nodes, int cni, int cnj, int cnk) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int j = blockIdx.y * blockDim.y + threadIdx.y;
  const int k = blockIdx.z;
  if (i >= f.ni || j >= f.nj) return;
  const long p = ((long)k * f.nj + j) * f.ni + i;
  const int ci = m.tc[3 * p], cj = m.tc[3 * p + 1], ck = m.tc[3 * p + 2];
  const double* w = m.cf + 7 * p;
  const long nn = (long)(cni + 1) * (cnj + 1) * (cnk + 1);
  auto nd = [&](int a, int b2, int c2, int e) {
    return nodes[(long)e * nn + ((long)(ck + c2) * (cnj + 1) + (cj + b2)) * (cni + 1) + ci + a];
  };
  const long q = f.idx(i, j, k);
#pragma unroll
  for (int e = 0; e < AGX_NEQ; ++e) {
    const double d04 = (1.0 - w[0]) * nd(0, 0, 0, e) + w[0] * nd(0, 0, 1, e);
    const double d15 = (1.0 - w[1]) * nd(1, 0, 0, e) + w[1] * nd(1, 0, 1, e);
    const double d26 = (1.0 - w[2]) * nd(0, 1, 0, e) + w[2] * nd(0, 1, 1, e);
    const double d37 = (1.0 - w[3]) * nd(1, 1, 0, e) + w[3] * nd(1, 1, 1, e);
    const double d0415 = (1.0 - w[4]) * d04 + w[4] * d15;
    const double d2637 = (1.0 - w[5]) * d26 + w[5] * d37;
    f.x[e][q] += (1.0 - w[6]) * d0415 + w[6] * d2637;
  }
}

// Cell-centre gradients for output (velocityGrad_, temperatureGrad_, densityGrad_,
// pressureGrad_): one sixth of each of the six Green-Gauss face gradients of the
// cell (procBlock.cpp:1397-1449, CalcGradsI/J/K :5173-5786), six fields u, v, w,
// T, rho, p.  Formed on demand; one thread per cell straight from the planes (an
// output step, not the iteration).  out: [cell][18], physical cells, i fastest.
// fields: u, v, w, T, rho, p and -- rans -- k, omega (tkeGrad_, omegaGrad_)
constexpr int NGF = 6 + (AGX_NEQ - 5);
__device__ __forceinline__ double grad_field(const BlockDev& b, const GasDev& g, long q, int f) {
  if (f < 3) return b.state[1 + f][q];
  if (f == 4) return b.state[0][q];
  if (f == 5) return b.state[4][q];
  if (f >= 6) return b.state[AGX_NEQ > 5 ? f - 1 : 0][q];
  return b.state[4][q] / (b.state[0][q] * g.R);
}
__device__ inline void face_grad6(const BlockDev& b, const GasDev& g, int d, long qU,
                                  double (*g6)[NGF]) {
  const long sd = b.stride(d), qL = qU - sd;
  double au[3][3], al[3][3];
  {
    double a0[3], a1[3], a2[3];
    area_vec(b, d, qU, a0); area_vec(b, d, qU + sd, a1); area_vec(b, d, qU - sd, a2);
    for (int r = 0; r < 3; ++r) { au[d][r] = 0.5 * (a0[r] + a1[r]); al[d][r] = 0.5 * (a0[r] + a2[r]); }
  }
  for (int t = 0; t < 3; ++t) {
    if (t == d) continue;
    const long st = b.stride(t);
    double a0[3], a1[3];
    area_vec(b, t, qU + st, a0); area_vec(b, t, qL + st, a1);
    for (int r = 0; r < 3; ++r) au[t][r] = 0.5 * (a0[r] + a1[r]);
    area_vec(b, t, qU, a0); area_vec(b, t, qL, a1);
    for (int r = 0; r < 3; ++r) al[t][r] = 0.5 * (a0[r] + a1[r]);
  }
  const double inv_vol = 1.0 / (0.5 * (b.vol[qL] + b.vol[qU]));
  for (int f = 0; f < NGF; ++f) {
    double vu[3], vl[3];
    const double fL = grad_field(b, g, qL, f), fU = grad_field(b, g, qU, f);
    vl[d] = fL; vu[d] = fU;
    for (int t = 0; t < 3; ++t) {
      if (t == d) continue;
      const long st = b.stride(t);
      vu[t] = 0.25 * (fL + fU + grad_field(b, g, qU + st, f) + grad_field(b, g, qL + st, f));
      vl[t] = 0.25 * (fL + fU + grad_field(b, g, qU - st, f) + grad_field(b, g, qL - st, f));
    }
    for (int r = 0; r < 3; ++r)
      g6[r][f] = (vu[0] * au[0][r] - vl[0] * al[0][r] + vu[1] * au[1][r] - vl[1] * al[1][r] +
                  vu[2] * au[2][r] - vl[2] * al[2][r]) * inv_vol;
  }
}
// acc: [3 NGF]; velocity gradient [3 r + c] first, then per field f >= 3 its three
// derivatives at 9 + 3 (f - 3)
__device__ inline void cell_grads18(const BlockDev& b, const GasDev& g, long q, double* acc) {
  for (int n = 0; n < 3 * NGF; ++n) acc[n] = 0.0;
  for (int d = 0; d < 3; ++d)
    for (int up = 0; up < 2; ++up) {
      double g6[3][NGF];
      face_grad6(b, g, d, q + (up ? b.stride(d) : 0), g6);
      for (int r = 0; r < 3; ++r) {
        for (int f = 0; f < 3; ++f) acc[3 * r + f] += (1.0 / 6.0) * g6[r][f];
        for (int f = 3; f < NGF; ++f) acc[9 + 3 * (f - 3) + r] += (1.0 / 6.0) * g6[r][f];
      }
    }
}
__global__ void __launch_bounds__(256) k_cell_grads(BlockDev b, GasDev g, double* out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int j = blockIdx.y * blockDim.y + threadIdx.y;
  const int k = blockIdx.z;
  if (i >= b.ni || j >= b.nj) return;
  double acc[3 * NGF];
  cell_grads18(b, g, b.idx(i, j, k), acc);
  double* o = out + 3 * NGF * (((long)k * b.nj + j) * b.ni + i);
  for (int n = 0; n < 3 * NGF; ++n) o[n] = acc[n];
}

// ---------------------------------------------------------------------------
// Output: the per-cell variables of a function file, WriteFunFile (output.cpp:235-407),
// formed and re-dimensionalised on the device; one thread per physical cell, variable by
// variable into out[v * ncell + cell] (each store of a wave is a contiguous row).  grads:
// k_cell_grads' output, or null when no gradient is asked for.
struct OutSpec {
  int nvar;
  int var[AGX_OUT_COUNT];
  double rho_ref, a_ref, l_ref, t_ref, mu_ref;
  int rank, global_pos;
};
__global__ void __launch_bounds__(256)
k_output_pack(BlockDev b, GasDev g, OutSpec sp, const double* __restrict__ grads,
              double* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int j = blockIdx.y * blockDim.y + threadIdx.y;
  const int k = blockIdx.z;
  if (i >= b.ni || j >= b.nj) return;
  const long q = b.idx(i, j, k);
  const long ncell = (long)b.ni * b.nj * b.nk, p = ((long)k * b.nj + j) * b.ni + i;
  double s[AGX_NEQ];
  load5(b.state, q, s);
  const double rR = sp.rho_ref, aR = sp.a_ref, lR = sp.l_ref, tR = sp.t_ref, muR = sp.mu_ref;
  // plain division / sqrt: an output path, the values go to a file
  const double t = s[4] / (s[0] * g.R);
  const double cs = sqrt(g.gamma * s[4] / s[0]);
  const double v2 = dot3(s + 1, s + 1);
  const double en = g.hf + g.n * s[4] / s[0] + 0.5 * v2;       // Energy: e(T) + |v|^2 / 2
  for (int v = 0; v < sp.nvar; ++v) {
    const int var = sp.var[v];
    double val = 0.0;
    switch (var) {
      case AGX_OUT_DENSITY: val = s[0] * rR; break;
      case AGX_OUT_VEL_X: val = s[1] * aR; break;
      case AGX_OUT_VEL_Y: val = s[2] * aR; break;
      case AGX_OUT_VEL_Z: val = s[3] * aR; break;
      case AGX_OUT_PRESSURE: val = s[4] * rR * aR * aR; break;
      case AGX_OUT_MACH: val = sqrt(v2) / cs; break;
      case AGX_OUT_SOS: val = cs * aR; break;
      case AGX_OUT_DT: val = b.dt[q] / (aR * lR); break;
      case AGX_OUT_TEMPERATURE: val = t * tR; break;
      case AGX_OUT_ENERGY: val = en * aR * aR; break;
      case AGX_OUT_ENTHALPY: val = (en + s[4] / s[0]) * aR * aR; break;
      case AGX_OUT_CP: val = g.cp * aR * aR / tR; break;
      case AGX_OUT_CV: val = g.cv * aR * aR / tR; break;
      case AGX_OUT_RANK: val = (double)sp.rank; break;
      case AGX_OUT_GLOBAL_POSITION: val = (double)sp.global_pos; break;
      case AGX_OUT_VISCOSITY: {
        const double temp = t * g.t_ref;
        val = (g.visc_c1 * temp * sqrt(temp)) / ((temp + g.visc_s) * g.mu_ref) * muR;
        break;
      }
      case AGX_OUT_WALL_DISTANCE: val = b.wdist[q] * lR; break;
#if AGX_NEQ == 7
      case AGX_OUT_VISCOSITY_RATIO: {
        const double temp = t * g.t_ref;
        val = b.turb3[0][q] / ((g.visc_c1 * temp * sqrt(temp)) / ((temp + g.visc_s) * g.mu_ref));
        break;
      }
      case AGX_OUT_TURB_VISCOSITY: val = b.turb3[0][q] * muR; break;
      case AGX_OUT_TKE: val = s[5] * aR * aR; break;
      case AGX_OUT_SDR: val = s[6] * aR * aR * rR / muR; break;
      case AGX_OUT_F1: val = b.turb3[1][q]; break;
      case AGX_OUT_F2: val = b.turb3[2][q]; break;
#endif
      default:
        if (var >= AGX_OUT_RESID) {
          const int e = var - AGX_OUT_RESID;
          if (e < AGX_NEQ) {
            const double l2 = lR * lR;
            const double sc = e == 0 ? rR * aR * l2
                              : e < 4 ? rR * aR * aR * l2
                              : e < 6 ? rR * aR * aR * aR * l2
                                      : rR * rR * aR * aR * aR * aR * l2 / muR;
            val = b.resid[e][q] * sc;
          }
        } else if (var >= AGX_OUT_VELGRAD) {
          const int gidx = var - AGX_OUT_VELGRAD;
          const double sc = gidx < 9 ? aR / lR
                            : gidx < 12 ? tR / lR
                            : gidx < 15 ? rR / lR
                            : gidx < 18 ? rR * aR * aR / lR
                            : gidx < 21 ? aR * aR / lR
                                        : aR * aR * rR / (muR * lR);
          val = gidx < 3 * NGF ? grads[3 * NGF * p + gidx] * sc : 0.0;
        }
    }
    out[(long)v * ncell + p] = val;
  }
}
// WriteRestart (output.cpp:651-752): n_eq + 1 dimensional values per cell, cell by cell;
// which = 0: the state, 1: consVarsNm1.  The payload is cell-major, so a wave's stores of
// one variable are (n_eq + 1) * 8 bytes apart -- the restart interval is thousands of
// iterations, the kernel moves 6 (8) values per cell once.
__global__ void __launch_bounds__(256)
k_restart_pack(BlockDev b, OutSpec sp, int which, double* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int j = blockIdx.y * blockDim.y + threadIdx.y;
  const int k = blockIdx.z;
  if (i >= b.ni || j >= b.nj) return;
  const long q = b.idx(i, j, k);
  const long p = ((long)k * b.nj + j) * b.ni + i;
  const double rR = sp.rho_ref, aR = sp.a_ref, muR = sp.mu_ref;
  const double scp[7] = {rR, aR, aR, aR, rR * aR * aR, aR * aR, aR * aR * rR / muR};
  const double scc[7] = {rR, aR * rR, aR * rR, aR * rR, aR * aR * rR, aR * aR * rR,
                         aR * aR * rR * rR / muR};
  double* o = out + (AGX_NEQ + 1) * p;
#pragma unroll
  for (int e = 0; e < AGX_NEQ; ++e)
    o[e] = which == 0 ? b.state[e][q] * scp[e] : b.consnm1[e][q] * scc[e];
  o[AGX_NEQ] = 1.0;          // mass fraction of the single species
}

// Nonreflecting surfaces, after a residual: pressure and velocity gradient of the
// cells next to the surface (the slice AssignInviscidGhostCells takes of
// pressureGrad_ / velocityGrad_, procBlock.cpp:2514-2515), kept for the ghost fills
// up to the next residual.  blockIdx.y is the surface.
__global__ void __launch_bounds__(256) k_nr_grads(BlockDev b, GasDev g) {
  const int sn = blockIdx.y;
  if (b.nr_off[sn] < 0) return;
  const agx_bc_surface sf = b.surf[sn];
  const int st = surface_type(sf);
  const int d3 = (st - 1) / 2, d1 = (d3 + 1) % 3, d2 = (d3 + 2) % 3;
  const int lo[3] = {sf.imin, sf.jmin, sf.kmin}, hi[3] = {sf.imax, sf.jmax, sf.kmax};
  const int n1 = hi[d1] - lo[d1], n2 = hi[d2] - lo[d2];
  const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (long)n1 * n2) return;
  const int rem = (int)t;
  int c[3];
  if (d1 < d2) { c[d1] = lo[d1] + rem % n1; c[d2] = lo[d2] + rem / n1; }
  else { c[d2] = lo[d2] + rem % n2; c[d1] = lo[d1] + rem / n2; }
  c[d3] = st % 2 == 0 ? lo[d3] - 1 : lo[d3];
  double acc[3 * NGF];
  cell_grads18(b, g, b.idx(c[0], c[1], c[2]), acc);
  double* o = b.nr_grad + 12 * ((long)b.nr_off[sn] + rem);
  for (int q = 0; q < 3; ++q) o[q] = acc[15 + q];
  for (int q = 0; q < 9; ++q) o[3 + q] = acc[q];
}

// procBlock::UpdateAuxillaryVariables (procBlock.cpp:6171): temperature_ and
// viscosity_ are never stored on the device (every kernel recomputes them from the
// state in registers); an output step that asks for them gets them formed into a
// scratch plane.  which: 0 temperature, 1 laminar viscosity
__global__ void __launch_bounds__(256) k_aux_field(BlockDev b, GasDev g, int which, double* out) {
  const long q = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= b.nplane) return;
  double s[AGX_NEQ];
  load5(b.state, q, s);
  double v = 0.0;
  if (s[0] > 0.0) {
    v = temperature(g, s);
    if (which == 1) v = viscosity(g, v);
  }
  out[q] = v;
}

// ---------------------------------------------------------------------------
// Set-up.  plot3dBlock::Volume / Centroid / FaceAreaI,J,K / FaceCenterI,J,K
// (plot3d.cpp:35-360) from the node coordinates x[nk+1][nj+1][ni+1][3]: one thread per
// node, which forms the cell and the three faces that have this node as their lowest
// corner.  Outputs in the reference's layout (agx_plot3d_metrics); null = not wanted.
struct MetricsOut { double *vol, *center, *fa[3], *fc[3]; };
__device__ __forceinline__ void m_sub(const double* a, const double* b, double* o) {
  o[0] = a[0] - b[0]; o[1] = a[1] - b[1]; o[2] = a[2] - b[2];
}
__device__ __forceinline__ void m_cross(const double* a, const double* b, double* o) {
  o[0] = a[1] * b[2] - a[2] * b[1];                 // vector3d::CrossProd vector3d.hpp:330-339
  o[1] = -1.0 * (a[0] * b[2] - a[2] * b[0]);
  o[2] = a[0] * b[1] - a[1] * b[0];
}
__device__ __forceinline__ double m_pyramid(const double* p, const double* a, const double* b,
                                            const double* c, const double* d) {
  double xp[3], xac[3], xbd[3], cr[3];            // PyramidVolume plot3d.cpp:490-498
#pragma unroll
  for (int q = 0; q < 3; ++q)
    xp[q] = 0.25 * ((((a[q] - p[q]) + (b[q] - p[q])) + (c[q] - p[q])) + (d[q] - p[q]));
  m_sub(c, a, xac);
  m_sub(d, b, xbd);
  m_cross(xac, xbd, cr);
  return 1.0 / 6.0 * dot3(xp, cr);
}
__device__ __forceinline__ void m_face(const double* n00, const double* n10, const double* n01,
                                       const double* n11, const double* xac, const double* xbd,
                                       double* fa, double* fc) {
  double cr[3];
  m_cross(xbd, xac, cr);
  const double h[3] = {0.5 * cr[0], 0.5 * cr[1], 0.5 * cr[2]};
  const double mag = sqrt(dot3(h, h));
  if (fa) { fa[0] = h[0] / mag; fa[1] = h[1] / mag; fa[2] = h[2] / mag; fa[3] = mag; }
  if (fc) {
#pragma unroll
    for (int q = 0; q < 3; ++q) fc[q] = 0.25 * (((n00[q] + n10[q]) + n01[q]) + n11[q]);
  }
}
__global__ void __launch_bounds__(256)
k_plot3d_metrics(int ni, int nj, int nk, const double* __restrict__ x, MetricsOut o, int* err) {
  const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long nn = (long)(ni + 1) * (nj + 1) * (nk + 1);
  if (t >= nn) return;
  const int i = (int)(t % (ni + 1)), j = (int)((t / (ni + 1)) % (nj + 1)),
            k = (int)(t / ((long)(ni + 1) * (nj + 1)));
  auto nd = [&](int a, int b, int c, double* v) {
    const double* p = x + 3 * (((long)c * (nj + 1) + b) * (ni + 1) + a);
    v[0] = p[0]; v[1] = p[1]; v[2] = p[2];
  };
  if (i < ni && j < nj && k < nk && (o.vol || o.center)) {
    double c000[3], c100[3], c010[3], c110[3], c001[3], c101[3], c011[3], c111[3], cen[3];
    nd(i, j, k, c000); nd(i + 1, j, k, c100); nd(i, j + 1, k, c010); nd(i + 1, j + 1, k, c110);
    nd(i, j, k + 1, c001); nd(i + 1, j, k + 1, c101); nd(i, j + 1, k + 1, c011);
    nd(i + 1, j + 1, k + 1, c111);
#pragma unroll
    for (int q = 0; q < 3; ++q)
      cen[q] = 0.125 * (((((((c000[q] + c100[q]) + c010[q]) + c110[q]) + c001[q]) + c101[q]) +
                         c011[q]) + c111[q]);
    const long p = ((long)k * nj + j) * ni + i;
    if (o.center) { o.center[3 * p] = cen[0]; o.center[3 * p + 1] = cen[1]; o.center[3 * p + 2] = cen[2]; }
    if (o.vol) {
      double v = m_pyramid(cen, c000, c001, c011, c010);
      v = v + m_pyramid(cen, c100, c110, c111, c101);
      v = v + m_pyramid(cen, c000, c100, c101, c001);
      v = v + m_pyramid(cen, c010, c011, c111, c110);
      v = v + m_pyramid(cen, c000, c010, c110, c100);
      v = v + m_pyramid(cen, c001, c101, c111, c011);
      if (!(v > 0.0)) *err = 4;          // "negative volume in PLOT3D block"
      o.vol[p] = v;
    }
  }
  double n00[3], n10[3], n01[3], n11[3], xac[3], xbd[3];
  if (j < nj && k < nk && (o.fa[0] || o.fc[0])) {          // i-face, plot3d.cpp:150-181
    nd(i, j, k, n00); nd(i, j + 1, k, n10); nd(i, j, k + 1, n01); nd(i, j + 1, k + 1, n11);
    m_sub(n11, n00, xac); m_sub(n10, n01, xbd);
    const long f = ((long)k * nj + j) * (ni + 1) + i;
    m_face(n00, n10, n01, n11, xac, xbd, o.fa[0] ? o.fa[0] + 4 * f : nullptr,
           o.fc[0] ? o.fc[0] + 3 * f : nullptr);
  }
  if (i < ni && k < nk && (o.fa[1] || o.fc[1])) {          // j-face, plot3d.cpp:224-255
    nd(i, j, k, n00); nd(i + 1, j, k, n10); nd(i, j, k + 1, n01); nd(i + 1, j, k + 1, n11);
    m_sub(n01, n10, xac); m_sub(n00, n11, xbd);
    const long f = ((long)k * (nj + 1) + j) * ni + i;
    m_face(n00, n10, n01, n11, xac, xbd, o.fa[1] ? o.fa[1] + 4 * f : nullptr,
           o.fc[1] ? o.fc[1] + 3 * f : nullptr);
  }
  if (i < ni && j < nj && (o.fa[2] || o.fc[2])) {          // k-face, plot3d.cpp:300-331
    nd(i, j, k, n00); nd(i + 1, j, k, n10); nd(i, j + 1, k, n01); nd(i + 1, j + 1, k, n11);
    m_sub(n01, n10, xac); m_sub(n11, n00, xbd);
    const long f = ((long)k * nj + j) * ni + i;
    m_face(n00, n10, n01, n11, xac, xbd, o.fa[2] ? o.fa[2] + 4 * f : nullptr,
           o.fc[2] ? o.fc[2] + 3 * f : nullptr);
  }
}
// kdtree::NearestNeighbor (kdtree.cpp:123-225) over the wall face centres, as an exhaustive
// search: a workgroup keeps 256 cells in registers and streams the wall points through LDS
// 256 at a time (every thread reads every point of a tile: LDS broadcast reads).
__global__ void __launch_bounds__(256)
k_nearest_wall(long ncell, const double* __restrict__ cen, long nwall,
               const double* __restrict__ wall, double* __restrict__ dist) {
  __shared__ double sw[256][3];
  const long q = (long)blockIdx.x * 256 + threadIdx.x;
  const long qc = min(q, ncell - 1);
  const double c[3] = {cen[3 * qc], cen[3 * qc + 1], cen[3 * qc + 2]};
  double best = 1.7976931348623157e308;
  for (long p0 = 0; p0 < nwall; p0 += 256) {
    const long p = min(p0 + threadIdx.x, nwall - 1);       // (the last point again: harmless)
    __syncthreads();
    sw[threadIdx.x][0] = wall[3 * p]; sw[threadIdx.x][1] = wall[3 * p + 1];
    sw[threadIdx.x][2] = wall[3 * p + 2];
    __syncthreads();
#pragma unroll 8
    for (int m = 0; m < 256; ++m) {
      const double d[3] = {c[0] - sw[m][0], c[1] - sw[m][1], c[2] - sw[m][2]};
      best = fmin(best, dot3(d, d));
    }
  }
  if (q < ncell) dist[q] = sqrt(best);
}

// ---------------------------------------------------------------------------
// AoS (reference host layout) <-> SoA conversion at the boundary
__global__ void k_aos_to_soa(const double* __restrict__ aos, Planes5 soa, int ncomp,
                             int ci, int cj, int ck, int gsrc, BlockDev b) {
  // aos dims (ci, cj, ck) with gsrc ghost layers, i fastest, ncomp per cell
  const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long n = (long)ci * cj * ck;
  if (t >= n) return;
  const int i = (int)(t % ci) - gsrc, j = (int)((t / ci) % cj) - gsrc,
            k = (int)(t / ((long)ci * cj)) - gsrc;
  const long q = b.idx(i, j, k);
  for (int e = 0; e < ncomp; ++e) soa.p[e][q] = aos[t * ncomp + e];
}
__global__ void k_soa_to_aos(double* __restrict__ aos, Planes5 soa, int ncomp,
                             int ci, int cj, int ck, int gsrc, BlockDev b) {
  const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long n = (long)ci * cj * ck;
  if (t >= n) return;
  const int i = (int)(t % ci) - gsrc, j = (int)((t / ci) % cj) - gsrc,
            k = (int)(t / ((long)ci * cj)) - gsrc;
  const long q = b.idx(i, j, k);
  for (int e = 0; e < ncomp; ++e) aos[t * ncomp + e] = soa.p[e][q];
}

}  // namespace agx
