// agx_visc_tile.hpp -- viscous residual, staged through registers / shuffles / LDS.
//
// Counterpart of procBlock::CalcViscFluxI/J/K (procBlock.cpp:1233-2135, laminar,
// central reconstruction) with CalcGradsI/J/K (:5173-5786), VectorGradGG /
// ScalarGradGG (utility.cpp:59-188), viscousFlux::CalcFlux (viscousFlux.cpp:58-135),
// TauNormal (utility.cpp:426-437) and the viscous part of the time step / scalar
// diagonal (ViscCellSpectralRadius spectralRadius.hpp:94-124).
//
// A face needs a ten-cell stencil of (u, v, w, T) and eleven face-area vectors;
// read from global memory per face (k_visc_march) that is ~300 vector-memory
// instructions per thread and k-step and 4.2 x the compulsory HBM traffic.  Here
// a workgroup of 64 x 8 threads sits on 64 x 8 cells of a k-plane and marches
// along k; each thread
//   * keeps the k-stencil of ITS column in registers (three planes of u,v,w,T,
//     the area vectors of two / three planes), loading one new plane per step,
//   * publishes its column once per step in LDS, where the i- and j-neighbours
//     read it (only the finished i-flux is handed over by a wave shuffle),
// and evaluates the lower i-, lower j- and upper k-face of its cell once.  Lane 0
// / row 0 only supply data, lane 63 / row 7 only the face their neighbour lacks:
// 62 x 6 cells are owned per workgroup.  Per thread and step: ~45 vector-memory
// instructions, every plane of the block read once per workgroup.
#pragma once

namespace agx {

constexpr int VT_L = 64, VT_R = 8;           // threads: lanes (i) x rows (j)
constexpr int VT_OI = VT_L - 2, VT_OJ = VT_R - 2;   // owned cells per workgroup
// LDS: what a thread publishes for its neighbours, [var][row][lane]
enum { VS_S0 = 0, VS_SM = 4, VS_SP = 8, VS_AI0 = 12, VS_AJ0 = 15, VS_AJ1 = 18, VS_AK0 = 21,
       VS_AK1 = 24, VS_VOL = 27, VS_WJ = 28, VS_RHO = 29, VS_MU = 30, VS_AI1 = 31, VS_WI = 34,
       VS_FJ = 35, VS_COUNT = 39 };

// VT_LDSAREA: the area vectors of the faces at i+1 / j+1 are the neighbour lane's / row's
// own lower-face vectors, already in its LDS slots; only lane 63 / row 7 (whose
// neighbour belongs to the next tile) fetch them from memory (0: every thread loads them:
// eight more loads per thread and step, of 29)
#ifndef VT_LDSAREA
#define VT_LDSAREA 1
#endif
#ifndef VT_EDGE_BARRIER
#define VT_EDGE_BARRIER __builtin_amdgcn_sched_barrier(0)
#endif
// operand of a face: a slot of the LDS window, (var, row, lane)
struct VRef { int var, row, lane; };
// [row][lane][var]: the eight (row, lane) bases a thread uses live in registers, the
// variable is an immediate offset; 39 doubles per slot (odd) => conflict-free
using VShared = double (*)[VT_L][VS_COUNT];
__device__ __forceinline__ void vt_ld4(VShared sh, const VRef& r, double* o) {
#pragma unroll
  for (int c = 0; c < 4; ++c) o[c] = sh[r.row][r.lane][r.var + c];
}
__device__ __forceinline__ void vt_ld3(VShared sh, const VRef& r, double* o) {
#pragma unroll
  for (int c = 0; c < 3; ++c) o[c] = sh[r.row][r.lane][r.var + c];
}
// grad[r][c] += sg * val[c] * a[r]
__device__ __forceinline__ void vt_acc(double (*grad)[4], const double* val, const double* a,
                                       double sg) {
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    const double w = sg * a[r];
#pragma unroll
    for (int c = 0; c < 4; ++c) grad[r][c] = fma(val[c], w, grad[r][c]);
  }
}
// one closing face of the dual volume at a transverse edge: the mean of the four
// cells around the edge (vL, vU and two more) times the mean of two area vectors
struct VtEdge { VRef c0, c1, a0, a1; };
__device__ __forceinline__ void vt_edge(VShared sh, double (*grad)[4], const double* vsum,
                                        const VtEdge& e, double sg) {
  double c0[4], c1[4], a0[3], a1[3], v[4], a[3];
  vt_ld4(sh, e.c0, c0); vt_ld4(sh, e.c1, c1);
  vt_ld3(sh, e.a0, a0); vt_ld3(sh, e.a1, a1);
#pragma unroll
  for (int c = 0; c < 4; ++c) v[c] = 0.25 * (vsum[c] + c0[c] + c1[c]);
#pragma unroll
  for (int r = 0; r < 3; ++r) a[r] = 0.5 * (a0[r] + a1[r]);
  vt_acc(grad, v, a, sg);
  VT_EDGE_BARRIER;   // keep the operands of the next edge out of flight
}
// One face.  L | U: the cells across it; aF the face's area vector, aFm / aFp the
// area vectors of the same-direction faces below / above (registers or LDS);
// per transverse direction the upper / lower edge.  Operands are fetched from the
// LDS window just in time: held all at once they do not fit the register file.
// F4 (viscousReconstruction centralFourth, FaceReconCentral4th reconstruction.hpp:335-379):
// the face state and viscosity come from the four cells around the face instead of two;
// `wide` hands over the two outer ones -- (u, v, w, T, rho, mu) each and their widths -- once
// the gradient is complete (they would not fit beside its operands).
struct VtNoWide {
  __device__ __forceinline__ void operator()(double*, double*, double&, double&) const {}
};
template <bool F4, class Wide>
__device__ __forceinline__ void vt_face(VShared sh, const GasDev& g, const VRef& rL,
                                        const VRef& rU, const VRef& rhoL, const VRef& rhoU,
                                        double rhoU_reg, double muU_reg, bool u_in_reg,
                                        const double* uReg, const double* aF, const double* aFm,
                                        const double* aFp, const VtEdge& t1u, const VtEdge& t1l,
                                        const VtEdge& t2u, const VtEdge& t2l, double volL,
                                        double volU, double wL, double wU, double* f,
                                        const Wide& wide) {
  double grad[3][4];
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 4; ++c) grad[r][c] = 0.0;
  double vL[4], vU[4], vsum[4];
  vt_ld4(sh, rL, vL);
  if (u_in_reg) {
#pragma unroll
    for (int c = 0; c < 4; ++c) vU[c] = uReg[c];
  } else {
    vt_ld4(sh, rU, vU);
  }
#pragma unroll
  for (int c = 0; c < 4; ++c) vsum[c] = vL[c] + vU[c];
  {
    double au[3], al[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) { au[r] = 0.5 * (aF[r] + aFp[r]); al[r] = 0.5 * (aF[r] + aFm[r]); }
    vt_acc(grad, vU, au, 1.0);
    vt_acc(grad, vL, al, -1.0);
  }
  __builtin_amdgcn_sched_barrier(0);
  vt_edge(sh, grad, vsum, t1u, 1.0);
  vt_edge(sh, grad, vsum, t1l, -1.0);
  vt_edge(sh, grad, vsum, t2u, 1.0);
  vt_edge(sh, grad, vsum, t2l, -1.0);
  const double inv_vol = fast_rcp(0.5 * (volL + volU));
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 4; ++c) grad[r][c] *= inv_vol;
  const double rL_ = sh[rhoL.row][rhoL.lane][rhoL.var];
  const double mL_ = sh[rhoL.row][rhoL.lane][rhoL.var + 1];          // VS_MU = VS_RHO + 1
  const double rU_ = u_in_reg ? rhoU_reg : sh[rhoU.row][rhoU.lane][rhoU.var];
  const double mU_ = u_in_reg ? muU_reg : sh[rhoU.row][rhoU.lane][rhoU.var + 1];
  double vf[3], tf, mu;
  if constexpr (F4) {
    // Newton form on the divided differences of the four cell values (visc_face_state,
    // agx_kernels.hpp); the reference reconstructs (rho, u, v, w, p) and mu
    double c0[6], c3[6], w0 = 1.0, w3 = 1.0;
    wide(c0, c3, w0, w3);
    const double w1 = wL, w2 = wU;
    const double r01 = fast_rcp(w0 + w1), r12 = fast_rcp(w1 + w2), r23 = fast_rcp(w2 + w3);
    const double t012 = fast_rcp(w0 + w1 + w2), t123 = fast_rcp(w1 + w2 + w3);
    const double q4 = fast_rcp((w0 + w1) + (w2 + w3));
    const double k3 = w1 * w2, k4 = k3 * (w0 + w1);
    auto c4 = [&](double u0, double u1, double u2, double u3) {
      const double g01 = (u1 - u0) * r01, g12 = (u2 - u1) * r12, g23 = (u3 - u2) * r23;
      const double d3a = (g12 - g01) * t012, d3b = (g23 - g12) * t123;
      return u1 + w1 * g12 - k3 * d3a - k4 * ((d3b - d3a) * q4);
    };
#pragma unroll
    for (int c = 0; c < 3; ++c) vf[c] = c4(c0[c], vL[c], vU[c], c3[c]);
    const double rf = c4(c0[4], rL_, rU_, c3[4]);
    const double pf = c4(c0[4] * c0[3], rL_ * vL[3], rU_ * vU[3], c3[4] * c3[3]);   // p / R
    tf = pf * fast_rcp(rf);
    mu = g.scaling * c4(c0[5], mL_, mU_, c3[5]);
  } else {
    // FaceReconCentral reconstruction.hpp:315-328: the reference forms
    // coeffs[0] * varD + coeffs[1] * varU with coeffs = {wD, wU} / (wU + wD)
    // (wU: the cell below the face, wD: the cell above)
    const double iw = fast_rcp(wL + wU);
    const double cD = wU * iw, cU = wL * iw;
#pragma unroll
    for (int c = 0; c < 3; ++c) vf[c] = cD * vU[c] + cU * vL[c];
    // T of the face-averaged state: p_f / (rho_f R) with p = rho R T
    const double rf = cD * rU_ + cU * rL_;
    tf = (cD * rU_ * vU[3] + cU * rL_ * vL[3]) * fast_rcp(rf);
    mu = g.scaling * (cD * mU_ + cU * mL_);
  }
  const double lambda = -(2.0 / 3.0) * mu;
  const double trace = grad[0][0] + grad[1][1] + grad[2][2];
  // tau . A  (viscousFlux.cpp:58-135 with the area vector instead of n |A|)
  double tau[3];
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    const double mm = (grad[r][0] + grad[0][r]) * aF[0] + (grad[r][1] + grad[1][r]) * aF[1] +
                      (grad[r][2] + grad[2][r]) * aF[2];
    tau[r] = lambda * trace * aF[r] + mu * mm;
  }
  const double kk = conductivity(g, tf) * g.scaling;
  const double tg = grad[0][3] * aF[0] + grad[1][3] * aF[1] + grad[2][3] * aF[2];
  f[0] = tau[0];
  f[1] = tau[1];
  f[2] = tau[2];
  f[3] = dot3(tau, vf) + kk * tg;
}

// Persistent: one workgroup per CU; the (column tile, k) steps of the block form one linear
// sequence cut into gridDim.x equal ranges (a range that crosses into the next column
// re-primes its window there), so all CUs finish together whatever the block shape --
// with a grid of (tiles x k-chunks) workgroups at one workgroup per CU the last round ran
// a fifth full.  Workgroup n runs on XCD n % 8: ranges are dealt so that each XCD's L2
// sees neighbouring columns.
//
// F4 (centralFourth): a face's state reaches two cells to either side.  Along i the window
// simply owns two lanes less (lanes 2 .. 61; 0, 1 and 63 only supply data); along j rows 1 and 7
// -- one wave each -- fetch the cell beyond the window (row -1 / row 8) into registers with the
// plane they prefetch; along k the own column's plane kk+2 is requested one step earlier and
// rho, mu and the k-width of plane kk-1 stay in registers.
template <bool F4>
__global__ void __launch_bounds__(VT_L * VT_R)
k_visc_tile(SlabDev b, GasDev g, SolverDev sp, double cfl, int gx, int gy) {
  __shared__ double sh[VT_R][VT_L][VS_COUNT];
  constexpr int LO = F4 ? 2 : 1;                        // first owning lane
  constexpr int OI = F4 ? VT_L - 4 : VT_OI;             // owned cells along i
  const int l = threadIdx.x, ty = threadIdx.y;
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
  const int ci = (col % gx) * OI - LO + l, cj = (col / gx) * VT_OJ - 1 + ty;
  const bool inner = l >= LO && l < LO + OI && ty >= 1 && ty <= VT_OJ;
  const bool own = inner && ci < b.ni && cj < b.nj;
  // lower i-face: own cell or the owned cell to the left; lower j-face likewise
  const bool do_i = l >= LO && l <= LO + OI && ty >= 1 && ty <= VT_OJ && ci <= b.ni && cj < b.nj;
  const bool do_j = ty >= 1 && l >= LO && l < LO + OI && cj <= b.nj && ci < b.ni;
  // overhanging threads stay in bounds (F4: the lane / row after the face at ni / nj
  // supplies cell ni+1 / nj+1)
  const int ic = min(ci, F4 ? b.ni + 1 : b.ni), jc = min(cj, F4 ? b.nj + 1 : b.nj);
  // one 32-bit byte offset per lane (column base, k = 0) + wave-uniform plane
  // offsets; every array is "slab plane base (SGPRs) + offset" (SlabDev::ldb).
  // Unsigned wrap-around makes the k = -1 plane come out right.
  const unsigned qc = (unsigned)(b.idx(ic, jc, 0) * 8);
  const unsigned sk = (unsigned)(b.sxy * 8), sj = (unsigned)(b.sx * 8);
  const int kcmax = b.nk + b.ng - 1, kfmax = b.nk + b.ng; // last valid cell / face plane
  auto ld_state = [&](int k, double* s4, double& rho, double& mu) {
    const unsigned q = qc + (unsigned)min(k, kcmax) * sk;
    rho = b.ldb(b.st + 0, q);
    s4[0] = b.ldb(b.st + 1, q); s4[1] = b.ldb(b.st + 2, q); s4[2] = b.ldb(b.st + 3, q);
    s4[3] = rho > 0.0 ? b.ldb(b.st + 4, q) * fast_rcp(rho * g.R) : 0.0;
    mu = rho > 0.0 ? viscosity(g, s4[3]) : 0.0;
  };
  // the same in two halves: the request (nothing computed on the values, so no wait is placed
  // in front of the barrier that follows) and the conversion where the plane enters the window
  auto rq_state = [&](int k, double* raw) {
    const unsigned q = qc + (unsigned)min(k, kcmax) * sk;
#pragma unroll
    for (int e = 0; e < 5; ++e) raw[e] = b.ldb(b.st + e, q);
  };
  auto cv_state = [&](const double* raw, double* s4, double& rho, double& mu) {
    rho = raw[0];
    s4[0] = raw[1]; s4[1] = raw[2]; s4[2] = raw[3];
    s4[3] = rho > 0.0 ? raw[4] * fast_rcp(rho * g.R) : 0.0;
    mu = rho > 0.0 ? viscosity(g, s4[3]) : 0.0;
  };
  auto ld_avec = [&](int d, unsigned q, double* a3) {
    const double mag = b.ldb(PL_FA + 4 * d + 3, q);
    a3[0] = b.ldb(PL_FA + 4 * d + 0, q) * mag; a3[1] = b.ldb(PL_FA + 4 * d + 1, q) * mag;
    a3[2] = b.ldb(PL_FA + 4 * d + 2, q) * mag;
  };
  auto rq_avec = [&](int d, unsigned q, double* raw) {
#pragma unroll
    for (int c = 0; c < 4; ++c) raw[c] = b.ldb(PL_FA + 4 * d + c, q);
  };
  auto cv_avec = [&](const double* raw, double* a3) {
    a3[0] = raw[0] * raw[3]; a3[1] = raw[1] * raw[3]; a3[2] = raw[2] * raw[3];
  };
  auto PUT4 = [&](int var, const double* v) {
#pragma unroll
    for (int c = 0; c < 4; ++c) sh[ty][l][var + c] = v[c];
  };
  auto PUT3 = [&](int var, const double* v) {
#pragma unroll
    for (int c = 0; c < 3; ++c) sh[ty][l][var + c] = v[c];
  };
  auto LD4 = [&](int var, int row, int lane, double* o) {
#pragma unroll
    for (int c = 0; c < 4; ++c) o[c] = sh[row][lane][var + c];
  };
  auto LD3 = [&](int var, int row, int lane, double* o) {
#pragma unroll
    for (int c = 0; c < 3; ++c) o[c] = sh[row][lane][var + c];
  };
  // ---- the window of the own column (planes kk-1, kk, kk+1) lives in the
  // thread's own LDS slots; registers hold only what nobody else reads.  kk starts
  // one plane early: the first pass only forms the k-face below the chunk ----
  int kk = k0 - 1;
  double R1, MU1, V1, WK0, WK1, AK2[3], AIp[3] = {0, 0, 0}, AJp[3] = {0, 0, 0};
  // F4 only: plane kk+2 of the own column, rho / mu / k-width of plane kk-1, and the cell
  // beyond the window's rows (u, v, w, T, rho, mu, j-width; rows 1 and 7)
  double S2[4] = {0, 0, 0, 0}, R2 = 0, MU2 = 0, WK2 = 0, RM = 0, MUM = 0, WKM = 0;
  double EJ[7] = {0, 0, 0, 0, 0, 0, 0};
  const bool ej_row = F4 && (ty == 1 || ty == VT_R - 1);
  // (row 1: j-2, row 7: j+1 -- wanted only where the row's own face is, cj <= nj)
  const unsigned qej = (unsigned)(b.idx(ic, ty == 1 ? jc - 2 : min(jc + 1, b.nj + 1), 0) * 8);
  {
    double t4[4], t3[3], r, m;
    const unsigned qk = qc + (unsigned)kk * sk;
    ld_state(kk, t4, r, m);
    PUT4(VS_S0, t4);
    if (!F4) PUT4(VS_SM, t4);
    sh[ty][l][VS_RHO] = r; sh[ty][l][VS_MU] = m;
    ld_state(kk + 1, t4, R1, MU1);
    PUT4(VS_SP, t4);
    if (F4) {
      ld_state(kk - 1, t4, RM, MUM);          // (k0 - 2 >= -2: the second ghost layer)
      PUT4(VS_SM, t4);
      ld_state(kk + 2, S2, R2, MU2);
      WKM = b.ldb(PL_WID + 2, qk - sk);
      WK2 = b.ldb(PL_WID + 2, qc + (unsigned)min(kk + 2, kcmax) * sk);
    }
    ld_avec(0, qk, t3); PUT3(VS_AI0, t3);
    ld_avec(0, qk + sk, t3); PUT3(VS_AI1, t3);
    ld_avec(1, qk, t3); PUT3(VS_AJ0, t3);
    ld_avec(1, qk + sk, t3); PUT3(VS_AJ1, t3);
    ld_avec(2, qk, t3); PUT3(VS_AK0, t3);
    ld_avec(2, qk + sk, t3); PUT3(VS_AK1, t3);
    ld_avec(2, qc + (unsigned)min(kk + 2, kfmax) * sk, AK2);
    if (!VT_LDSAREA || l == VT_L - 1) ld_avec(0, qk + 8, AIp);
    if (!VT_LDSAREA || ty == VT_R - 1) ld_avec(1, qk + sj, AJp);
    sh[ty][l][VS_VOL] = b.ldb(PL_VOL, qk); V1 = b.ldb(PL_VOL, qk + sk);
    WK0 = b.ldb(PL_WID + 2, qk); WK1 = b.ldb(PL_WID + 2, qk + sk);
    sh[ty][l][VS_WI] = b.ldb(PL_WID + 0, qk); sh[ty][l][VS_WJ] = b.ldb(PL_WID + 1, qk);
  }
  double fk_lo[4] = {0, 0, 0, 0};
  const int tu = min(ty + 1, VT_R - 1), td = max(ty - 1, 0);
  const int lr = min(l + 1, VT_L - 1), ll = max(l - 1, 0);
  for (; kk < k1; ++kk) {
    const bool pre = kk < k0;                  // only the k-face of this pass is used
    __syncthreads();                           // the windows of all columns are in place
    // neighbours are read where they are used (i: lanes l -+ 1 of the own row, j:
    // rows ty -+ 1); clamped indices only ever serve threads whose face is not needed
    double fi[4] = {0, 0, 0, 0}, fj[4] = {0, 0, 0, 0}, fk_up[4] = {0, 0, 0, 0};
    const VRef me0{VS_S0, ty, l}, meP{VS_SP, ty, l}, meM{VS_SM, ty, l};
    const VRef meR{VS_RHO, ty, l};
    // ---- lower i-face: cells (i-1,j) | (i,j) ----
    if (do_i && !pre) {
      double aF[3], aFm[3];
      vt_ld3(sh, VRef{VS_AI0, ty, l}, aF); vt_ld3(sh, VRef{VS_AI0, ty, ll}, aFm);
      if (VT_LDSAREA && l < VT_L - 1) vt_ld3(sh, VRef{VS_AI0, ty, lr}, AIp);
      const VtEdge ju{{VS_S0, tu, l}, {VS_S0, tu, ll}, {VS_AJ0, tu, l}, {VS_AJ0, tu, ll}};
      const VtEdge jl{{VS_S0, td, l}, {VS_S0, td, ll}, {VS_AJ0, ty, l}, {VS_AJ0, ty, ll}};
      const VtEdge ku{meP, {VS_SP, ty, ll}, {VS_AK1, ty, l}, {VS_AK1, ty, ll}};
      const VtEdge kl{meM, {VS_SM, ty, ll}, {VS_AK0, ty, l}, {VS_AK0, ty, ll}};
      auto wide = [&](double* c0, double* c3, double& w0, double& w3) {
        const int l0 = max(l - 2, 0);
        vt_ld4(sh, VRef{VS_S0, ty, l0}, c0); vt_ld4(sh, VRef{VS_S0, ty, lr}, c3);
        c0[4] = sh[ty][l0][VS_RHO]; c0[5] = sh[ty][l0][VS_MU]; w0 = sh[ty][l0][VS_WI];
        c3[4] = sh[ty][lr][VS_RHO]; c3[5] = sh[ty][lr][VS_MU]; w3 = sh[ty][lr][VS_WI];
      };
      vt_face<F4>(sh, g, VRef{VS_S0, ty, ll}, me0, VRef{VS_RHO, ty, ll}, meR, 0.0, 0.0, false,
                  nullptr, aF, aFm, AIp, ju, jl, ku, kl, sh[ty][ll][VS_VOL], sh[ty][l][VS_VOL],
                  sh[ty][ll][VS_WI], sh[ty][l][VS_WI], fi, wide);
    }
    // hand the i-flux to the lane on the left, the j-flux to the row below as soon
    // as they exist: racc = +lower -upper (i) +lower (j)
    double racc[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) racc[c] = fi[c] - __shfl_down(fi[c], 1, 64);
    __builtin_amdgcn_sched_barrier(0);
    // ---- lower j-face: cells (i,j-1) | (i,j) ----
    if (do_j && !pre) {
      double aF[3], aFm[3];
      vt_ld3(sh, VRef{VS_AJ0, ty, l}, aF); vt_ld3(sh, VRef{VS_AJ0, td, l}, aFm);
      if (VT_LDSAREA && ty < VT_R - 1) vt_ld3(sh, VRef{VS_AJ0, tu, l}, AJp);
      const VtEdge iu{{VS_S0, ty, lr}, {VS_S0, td, lr}, {VS_AI0, ty, lr}, {VS_AI0, td, lr}};
      const VtEdge il{{VS_S0, ty, ll}, {VS_S0, td, ll}, {VS_AI0, ty, l}, {VS_AI0, td, l}};
      const VtEdge ku{meP, {VS_SP, td, l}, {VS_AK1, ty, l}, {VS_AK1, td, l}};
      const VtEdge kl{meM, {VS_SM, td, l}, {VS_AK0, ty, l}, {VS_AK0, td, l}};
      auto wide = [&](double* c0, double* c3, double& w0, double& w3) {
        if (ty == 1) {                         // (wave-uniform: a row is a wave)
#pragma unroll
          for (int c = 0; c < 6; ++c) c0[c] = EJ[c];
          w0 = EJ[6];
        } else {
          const int t0 = max(ty - 2, 0);
          vt_ld4(sh, VRef{VS_S0, t0, l}, c0);
          c0[4] = sh[t0][l][VS_RHO]; c0[5] = sh[t0][l][VS_MU]; w0 = sh[t0][l][VS_WJ];
        }
        if (ty == VT_R - 1) {
#pragma unroll
          for (int c = 0; c < 6; ++c) c3[c] = EJ[c];
          w3 = EJ[6];
        } else {
          vt_ld4(sh, VRef{VS_S0, tu, l}, c3);
          c3[4] = sh[tu][l][VS_RHO]; c3[5] = sh[tu][l][VS_MU]; w3 = sh[tu][l][VS_WJ];
        }
      };
      vt_face<F4>(sh, g, VRef{VS_S0, td, l}, me0, VRef{VS_RHO, td, l}, meR, 0.0, 0.0, false,
                  nullptr, aF, aFm, AJp, iu, il, ku, kl, sh[td][l][VS_VOL], sh[ty][l][VS_VOL],
                  sh[td][l][VS_WJ], sh[ty][l][VS_WJ], fj, wide);
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) { sh[ty][l][VS_FJ + c] = fj[c]; racc[c] += fj[c]; }
    __builtin_amdgcn_sched_barrier(0);
    // ---- upper k-face: cells (i,j,k) | (i,j,k+1) ----
    const double R0 = sh[ty][l][VS_RHO], MU0 = sh[ty][l][VS_MU], V0 = sh[ty][l][VS_VOL];
    if (own) {
      double aF[3], aFm[3], SPr[4];
      vt_ld3(sh, VRef{VS_AK1, ty, l}, aF); vt_ld3(sh, VRef{VS_AK0, ty, l}, aFm);
      vt_ld4(sh, meP, SPr);
      const VtEdge iu{{VS_SP, ty, lr}, {VS_S0, ty, lr}, {VS_AI1, ty, lr}, {VS_AI0, ty, lr}};
      const VtEdge il{{VS_SP, ty, ll}, {VS_S0, ty, ll}, {VS_AI1, ty, l}, {VS_AI0, ty, l}};
      const VtEdge ju{{VS_SP, tu, l}, {VS_S0, tu, l}, {VS_AJ1, tu, l}, {VS_AJ0, tu, l}};
      const VtEdge jl{{VS_SP, td, l}, {VS_S0, td, l}, {VS_AJ1, ty, l}, {VS_AJ0, ty, l}};
      auto wide = [&](double* c0, double* c3, double& w0, double& w3) {
        vt_ld4(sh, meM, c0);
        c0[4] = RM; c0[5] = MUM; w0 = WKM;
#pragma unroll
        for (int c = 0; c < 4; ++c) c3[c] = S2[c];
        c3[4] = R2; c3[5] = MU2; w3 = WK2;
      };
      vt_face<F4>(sh, g, me0, meP, meR, meR, R1, MU1, true, SPr, aF, aFm, AK2, iu, il, ju, jl, V0,
                  V1, WK0, WK1, fk_up, wide);
    }
    __builtin_amdgcn_sched_barrier(0);
    // ---- request the plane that enters the window next step (after the faces; in flight
    // during the barrier and the residual update).  RAW values: nothing is computed on them
    // here -- with T, mu and the area vectors formed right behind the loads every wave waited
    // for its requests in front of the barrier, a round trip to memory per step (R pass
    // 5.45 -> 5.0 ms); they are converted where the plane enters the window.  Ahead of the
    // faces the requests cost 60 registers and gain nothing (5.25 ms) ----
    double nRaw[5], rAI[4], rAJ[4], rAK[4], rAIp[4] = {0, 0, 0, 0}, rAJp[4] = {0, 0, 0, 0};
    const unsigned qn = qc + (unsigned)(kk + 1) * sk;            // plane kk+1 (always valid)
    const unsigned qn2 = qc + (unsigned)min(kk + 2, kcmax) * sk; // cells of plane kk+2
    rq_state(F4 ? kk + 3 : kk + 2, nRaw);          // (F4: plane kk+2 is already in S2)
    rq_avec(0, qn2, rAI);
    rq_avec(1, qn2, rAJ);
    rq_avec(2, qc + (unsigned)min(kk + 3, kfmax) * sk, rAK);
    if (!VT_LDSAREA || l == VT_L - 1) rq_avec(0, qn + 8, rAIp);
    if (!VT_LDSAREA || ty == VT_R - 1) rq_avec(1, qn + sj, rAJp);
    const double nV = b.ldb(PL_VOL, qn2);
    const double nWK = b.ldb(PL_WID + 2, F4 ? qc + (unsigned)min(kk + 3, kcmax) * sk : qn2);
    double nEJ[6] = {0, 0, 0, 0, 0, 0}, nEw = 0;
    if (ej_row) {                              // the cell beyond the window's rows, plane kk+1
      const unsigned qe = qej + (unsigned)(kk + 1) * sk;
#pragma unroll
      for (int e = 0; e < 5; ++e) nEJ[e] = b.ldb(b.st + e, qe);
      nEw = b.ldb(PL_WID + 1, qe);
    }
    const double nwi = b.ldb(PL_WID + 0, qn), nwj = b.ldb(PL_WID + 1, qn);
    __syncthreads();                           // all faces done: the windows may rotate
    if (own && !pre) {
      const unsigned q = qc + (unsigned)kk * sk;
      double res[AGX_NEQ];
#pragma unroll
      for (int e = 1; e < AGX_NEQ; ++e) res[e] = b.ldb(PL_RESID + e, q);
      // +lower -upper per direction
#pragma unroll
      for (int c = 0; c < 4; ++c)
        res[1 + c] += ((racc[c] - sh[ty + 1][l][VS_FJ + c]) + fk_lo[c]) - fk_up[c];
      // ViscCellSpectralRadius spectralRadius.hpp:94-124
      const double ivol = fast_rcp(V0);
      const double vfac = visc_max_term(g, R0) * visc_term(g, MU0);
      double sr = b.ldb(PL_SPECRAD, q);
      double diag = sp.implicit ? b.ldb(PL_A, q) : 0.0;
      // |A| of the six faces from the area vectors (n |A| was formed at load time)
      double ai0[3], aj0[3], ak0[3], ak1[3];
      LD3(VS_AI0, ty, l, ai0); LD3(VS_AJ0, ty, l, aj0);
      LD3(VS_AK0, ty, l, ak0); LD3(VS_AK1, ty, l, ak1);
      const double fm[3] = {0.5 * (fast_sqrt(dot3(ai0, ai0)) + fast_sqrt(dot3(AIp, AIp))),
                            0.5 * (fast_sqrt(dot3(aj0, aj0)) + fast_sqrt(dot3(AJp, AJp))),
                            0.5 * (fast_sqrt(dot3(ak0, ak0)) + fast_sqrt(dot3(ak1, ak1)))};
#pragma unroll
      for (int d = 0; d < 3; ++d) {
        const double vsr = vfac * fm[d] * fm[d] * ivol;
        sr += vsr * sp.visc_cfl_coeff;
        diag += 2.0 * vsr;
      }
#pragma unroll
      for (int e = 1; e < AGX_NEQ; ++e) b.stb(PL_RESID + e, q, res[e]);   // mass: no viscous flux
      b.stb(PL_SPECRAD, q, sr);
      if (sp.implicit) b.stb(PL_A, q, diag);
      b.stb(PL_DT, q, sp.dt_fixed > 0.0 ? sp.dt_fixed : cfl * (V0 * fast_rcp(fmax(sr, 0.0))));
    }
    // ---- rotate the window in the own LDS slots ----
    {
      double t3[3], t4[4];
      double nS[4], nR, nMU, nAI[3], nAJ[3], nAK[3], nAIp[3], nAJp[3];
      cv_state(nRaw, nS, nR, nMU);
      cv_avec(rAI, nAI); cv_avec(rAJ, nAJ); cv_avec(rAK, nAK);
      cv_avec(rAIp, nAIp); cv_avec(rAJp, nAJp);
      LD4(VS_S0, ty, l, t4); PUT4(VS_SM, t4);
      LD4(VS_SP, ty, l, t4); PUT4(VS_S0, t4);
      if (F4) {
        PUT4(VS_SP, S2);
        RM = sh[ty][l][VS_RHO]; MUM = sh[ty][l][VS_MU];
        sh[ty][l][VS_RHO] = R1; R1 = R2; R2 = nR;
        sh[ty][l][VS_MU] = MU1; MU1 = MU2; MU2 = nMU;
#pragma unroll
        for (int c = 0; c < 4; ++c) S2[c] = nS[c];
        if (ej_row) {
          const double rho = nEJ[0];
          EJ[0] = nEJ[1]; EJ[1] = nEJ[2]; EJ[2] = nEJ[3];
          EJ[3] = rho > 0.0 ? nEJ[4] * fast_rcp(rho * g.R) : 0.0;
          EJ[4] = rho;
          EJ[5] = rho > 0.0 ? viscosity(g, EJ[3]) : 0.0;
          EJ[6] = nEw;
        }
      } else {
        PUT4(VS_SP, nS);
        sh[ty][l][VS_RHO] = R1; R1 = nR;
        sh[ty][l][VS_MU] = MU1; MU1 = nMU;
      }
      LD3(VS_AI1, ty, l, t3); PUT3(VS_AI0, t3); PUT3(VS_AI1, nAI);
      LD3(VS_AJ1, ty, l, t3); PUT3(VS_AJ0, t3); PUT3(VS_AJ1, nAJ);
      LD3(VS_AK1, ty, l, t3); PUT3(VS_AK0, t3); PUT3(VS_AK1, AK2);
      sh[ty][l][VS_VOL] = V1; V1 = nV;
      sh[ty][l][VS_WI] = nwi; sh[ty][l][VS_WJ] = nwj;
      if (F4) { WKM = WK0; WK0 = WK1; WK1 = WK2; WK2 = nWK; }
      else { WK0 = WK1; WK1 = nWK; }
#pragma unroll
      for (int c = 0; c < 4; ++c) fk_lo[c] = fk_up[c];
#pragma unroll
      for (int c = 0; c < 3; ++c) { AK2[c] = nAK[c]; AIp[c] = nAIp[c]; AJp[c] = nAJp[c]; }
    }
  }
  __syncthreads();   // segment boundary: the windows are primed afresh
  }
}

}  // namespace agx
