// agx_lusgs_kernels.hpp -- kernels of the D2 LU-SGS path (see agx_lusgs.hpp).
// Included from agx_kernels.hpp after the shared helpers (rhs_b,
// bc_is_connection, norm_block_reduce).
#pragma once

namespace agx {

// ---------------------------------------------------------------------------
// 32 x 32 tile <-> D2 transposition helpers.  Cells of a tile diagonal
// li + lj = dd are contiguous in the D2 plane, so a tile is read/written on the
// D2 side in "tile-diagonal order" t = 0..1023 and on the SoA side row by row.
constexpr int TT = 32;          // tile edge
constexpr int TRS = TT + 2;     // LDS row stride: (TRS - 1) odd => a diagonal
                                // walks 32 different 8-byte banks
__device__ __forceinline__ int tile_diag_index(int li, int lj) {
  const int dd = li + lj;
  if (dd < TT) return dd * (dd + 1) / 2 + lj;
  const int r = 2 * TT - 2 - dd;              // diagonals after this one
  return TT * TT - (r + 1) * (r + 2) / 2 + (lj - (dd - (TT - 1)));
}
// s_tab[t] = li | lj << 8 of position t in tile-diagonal order
__device__ __forceinline__ void tile_table(unsigned short* s_tab) {
  const int tid = threadIdx.x;
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    const int li = tid & (TT - 1), lj = (tid >> 5) + 8 * m;
    s_tab[tile_diag_index(li, lj)] = (unsigned short)(li | (lj << 8));
  }
}

// tag of a writer launch in the two low mantissa bits of an x value (see KpArgs)
__device__ __forceinline__ double kp_tagged(double v, unsigned tag) {
  const unsigned long long u = (unsigned long long)__double_as_longlong(v);
  return __longlong_as_double((long long)((u & ~3ull) | tag));
}
__device__ __forceinline__ bool kp_has_tag(double v, unsigned tag) {
  return ((unsigned)__double_as_longlong(v) & 3u) == tag;
}
// viscous factor of a cell: visc_max_term * visc_term of ViscFaceSpectralRadius
// (spectralRadius.hpp:94-160) with the laminar viscosity of
// UpdateAuxillaryVariables (procBlock.cpp:6171); ghost corners may hold zeros
__device__ __forceinline__ double cell_visc_factor(const GasDev& g, const double* s) {
  if (!(s[0] > 0.0)) return 0.0;
  return visc_max_term(g, s[0]) * visc_term(g, viscosity(g, temperature(g, s)));
}

// static part of the D2 arrays: per lower face of every padded cell the unit
// normal, |A| and |A| / dist with ProjC2CDist (procBlock.cpp:6316-6342) =
// (centre - centre of the lower neighbour) . n  (one-time gather at block
// creation; the stores are scattered)
__global__ void __launch_bounds__(256) k_d2_geo(BlockDev b) {
  const D2Dev& z = b.d2;
  const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long n = (long)z.Pi * z.Pj * (b.nk + 2 * b.ng);
  if (t >= n) return;
  const int e3[3] = {(int)(t % z.Pi), (int)((t / z.Pi) % z.Pj), (int)(t / ((long)z.Pi * z.Pj))};
  const long q = b.idx(e3[0] - b.ng, e3[1] - b.ng, e3[2] - b.ng);
  const long p = (long)e3[2] * z.ps + z.pos2(e3[0], e3[1]);
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    double a[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) a[c] = b.fa[d][c][q];
    double ad = 0.0;
    if (e3[d] > 0) {
      const long ql = q - b.stride(d);
      const double v[3] = {b.cen[0][q] - b.cen[0][ql], b.cen[1][q] - b.cen[1][ql],
                           b.cen[2][q] - b.cen[2][ql]};
      const double dist = dot3(v, a);
      ad = dist != 0.0 ? a[3] / dist : 0.0;
    }
    z.pa(PA_F + 2 * d + 0)[p] = make_double2(a[0], a[1]);
    z.pa(PA_F + 2 * d + 1)[p] = make_double2(a[2], a[3]);
    z.ad(d)[p] = ad;
  }
}
// x between the SoA planes and the D2 array (field download / upload only)
__global__ void __launch_bounds__(256) k_d2_x_copy(BlockDev b, int to_d2, unsigned tag) {
  const D2Dev& z = b.d2;
  const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long n = (long)z.Pi * z.Pj * (b.nk + 2 * b.ng);
  if (t >= n) return;
  const int ie = (int)(t % z.Pi), je = (int)((t / z.Pi) % z.Pj),
            ke = (int)(t / ((long)z.Pi * z.Pj));
  const long q = b.idx(ie - b.ng, je - b.ng, ke - b.ng);
  const long p = (long)ke * z.ps + z.pos2(ie, je);
#pragma unroll
  for (int e = 0; e < AGX_NEQ; ++e) {
    double* xp = reinterpret_cast<double*>(z.pa(PA_X + (e >> 1))) + 2 * p + (e & 1);
    if (to_d2) *xp = kp_tagged(b.x[e][q], tag);   // every writer of x tags it (KpArgs)
    else b.x[e][q] = *xp;
  }
}

// ---------------------------------------------------------------------------
// linearSolver::AddDiagonalTerms :146-175, Invert :177-188 and
// InitializeMatrixUpdate :111-144 fused with the SoA -> D2 conversion of what
// the sweeps read: state (all padded cells: ghost cells feed the off-diagonals
// across connection boundaries) with speed of sound and viscous factor,
// right-hand side b, 1/a, initial x.  write_x = 0: nobody reads the initial x (one
// sweep: the forward sweep writes every physical x before it is read, and
// without connection surfaces no ghost x is ever a neighbour) -- skip 48 B/cell.
// Grid: (Pi / 32, Pj / 32, Pk) tiles of the padded box, 256 threads.
__global__ void __launch_bounds__(256)
k_lusgs_prepare(BlockDev b, GasDev g, SolverDev sp, int write_x, unsigned tag, int again) {
  // (again: a second launch on the same residual -- the forcing term of a multigrid level now
  // exists -- takes the finished diagonal of the first one)
  __shared__ double sv[6][TT][TRS];
  __shared__ unsigned short s_tab[TT * TT];
  const D2Dev& z = b.d2;
  const int tid = threadIdx.x;
  tile_table(s_tab);
  const int i0 = blockIdx.x * TT, j0 = blockIdx.y * TT, ke = blockIdx.z;
  const int k = ke - b.ng;
  const int li = tid & (TT - 1);
  double bb[4][AGX_NEQ], x0[4][AGX_NEQ], ainv[4], vf[4];
  // ---- round 1: state + speed of sound ----
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    const int lj = (tid >> 5) + 8 * m;
    const int ie = i0 + li, je = j0 + lj;
    const int i = ie - b.ng, j = je - b.ng;
#pragma unroll
    for (int e = 0; e < AGX_NEQ; ++e) { bb[m][e] = 0.0; x0[m][e] = 0.0; }
    ainv[m] = 1.0;
    vf[m] = 0.0;
    double cs = 0.0;
    if (ie < z.Pi && je < z.Pj) {
      const long q = b.idx(i, j, k);
      double s[AGX_NEQ];
#pragma unroll
      for (int e = 0; e < AGX_NEQ; ++e) { s[e] = b.state[e][q]; sv[e][lj][li] = s[e]; }
      if (s[0] > 0.0) cs = sound_speed(g, s);
      if (sp.viscous) vf[m] = cell_visc_factor(g, s);
      const bool phys = i >= 0 && i < b.ni && j >= 0 && j < b.nj && k >= 0 && k < b.nk;
      if (phys) {
        double dvt = (b.vol[q] * (1.0 + sp.zeta)) / (b.dt[q] * sp.theta);
        if (sp.dual_time_cfl > 0.0) dvt += fmax(b.specrad[q], 0.0) / sp.dual_time_cfl;
        // (the plane-major a_ itself is not written back on the finest level: the next
        // residual assigns it afresh.  The finished diagonal goes to the D2 arrays as 1/a
        // and, for the multigrid calls that work on the planes, into the aInv_ plane,
        // which nothing else uses on this path.)
        const double a = again ? b.ainv[q] : b.a[q] * sp.relax + dvt;
        b.ainv[q] = a;
        // (a coarse multigrid level: the reference's a_ holds the finished diagonal from
        // here on, and a second restriction to the level within the iteration adds its
        // spectral radii to THAT -- SolverDev::diag_add)
        if (sp.diag_add) b.a[q] = a;
        ainv[m] = 1.0 / a;
        rhs_bf(b, g, sp, q, bb[m]);
        if (sp.requires_init) {
#pragma unroll
          for (int e = 0; e < AGX_NEQ; ++e) x0[m][e] = bb[m][e] * ainv[m];
        }
      }
    }
    sv[5][lj][li] = cs;
  }
  __syncthreads();
  long pos[4];
  int tli[4], tlj[4];
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    const unsigned tl = s_tab[tid + 256 * m];
    tli[m] = tl & 0xff; tlj[m] = tl >> 8;
    const int ie = i0 + tli[m], je = j0 + tlj[m];
    pos[m] = -1;
    if (ie < z.Pi && je < z.Pj) {
      pos[m] = (long)ke * z.ps + z.pos2(ie, je);
#pragma unroll
      for (int h = 0; h < 3; ++h)
        z.pa(PA_S + h)[pos[m]] = make_double2(sv[2 * h][tlj[m]][tli[m]], sv[2 * h + 1][tlj[m]][tli[m]]);
    }
  }
  __syncthreads();
  // ---- round 2: right-hand side b + 1/a ----
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    const int lj = (tid >> 5) + 8 * m;
#pragma unroll
    for (int e = 0; e < AGX_NEQ; ++e) sv[e][lj][li] = bb[m][e];
    sv[5][lj][li] = ainv[m];
  }
  __syncthreads();
  double ai2[4];
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    ai2[m] = sv[5][tlj[m]][tli[m]];
    if (pos[m] >= 0) {
#pragma unroll
      for (int h = 0; h < 3; ++h)
        z.pa(PA_B + h)[pos[m]] = make_double2(sv[2 * h][tlj[m]][tli[m]], sv[2 * h + 1][tlj[m]][tli[m]]);
    }
  }
  __syncthreads();
  // ---- round 3: initial x (zero unless the solver needs b / a) + viscous factor ----
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    const int lj = (tid >> 5) + 8 * m;
#pragma unroll
    for (int e = 0; e < AGX_NEQ; ++e) sv[e][lj][li] = x0[m][e];
    sv[5][lj][li] = vf[m];
  }
  __syncthreads();
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    if (pos[m] >= 0) {
      if (write_x) {
        // (every writer of x tags it: KpArgs)
        z.pa(PA_X + 0)[pos[m]] = make_double2(kp_tagged(sv[0][tlj[m]][tli[m]], tag),
                                              kp_tagged(sv[1][tlj[m]][tli[m]], tag));
        z.pa(PA_X + 1)[pos[m]] = make_double2(kp_tagged(sv[2][tlj[m]][tli[m]], tag),
                                              kp_tagged(sv[3][tlj[m]][tli[m]], tag));
        z.pa(PA_X + 2)[pos[m]] = make_double2(kp_tagged(sv[4][tlj[m]][tli[m]], tag), ai2[m]);
      }
      if (sp.viscous) z.vf()[pos[m]] = sv[5][tlj[m]][tli[m]];
    }
  }
}

// ---------------------------------------------------------------------------
// The record of a finished cell: everything a successor needs from it that does
// not depend on the face between them.
struct KpRec {
  double dx[AGX_NEQ];          // its update dU
  double r1, v1[3], p1, h1;    // prim(U + dU): rho, velocity, p, rho H
  double r0, v0[3], p0, h0;    // prim(U)
  double cs, vf;               // speed of sound, viscous factor of U
};
constexpr int KP_NV = 19;
template <class G>
__device__ __forceinline__ void kp_build_rec(const G& g, const double* s, double cs,
                                             double vf, const double* dx, KpRec& r) {
  // UpdatePrimWithCons primitive.hpp:206-231 (prim_to_cons, + dU, cons_to_prim)
  const double ke = g.hf + 0.5 * dot3(s + 1, s + 1);
  const double e0 = s[0] * ke + g.n * s[4];               // rho E
  r.r0 = s[0]; r.p0 = s[4]; r.h0 = e0 + s[4]; r.cs = cs; r.vf = vf;
  const double r1 = s[0] + dx[0];
  const double ir = fast_rcp(r1);
  double v2 = 0.0;
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    r.v0[q] = s[1 + q];
    r.v1[q] = fma(s[0], s[1 + q], dx[1 + q]) * ir;
    v2 = fma(r.v1[q], r.v1[q], v2);
  }
  const double e1 = e0 + dx[4];
  r.r1 = r1;
  r.p1 = (e1 - r1 * (g.hf + 0.5 * v2)) * g.inv_n;
  r.h1 = e1 + r.p1;
#pragma unroll
  for (int e = 0; e < AGX_NEQ; ++e) r.dx[e] = dx[e];
}
// one neighbour's term of ImplicitLower / ImplicitUpper (procBlock.cpp:1056-1163)
// = RusanovScalarOffDiagonal (fluxJacobian.cpp:122-162) through the face with
// unit normal n, area fa, fa / dist = fad; `lower` is the side of the neighbour
__device__ __forceinline__ void kp_term(const KpRec& r, const double* n, double fa, double fad,
                                        bool viscous, bool lower, double* acc) {
  const double vn0 = dot3(r.v0, n), vn1 = dot3(r.v1, n);
  double lam = 0.5 * fa * (fabs(vn0) + r.cs);            // FaceSpectralRadius :182-203
  if (viscous) lam = fma(fad, r.vf, lam);
  const double ha = lower ? 0.5 * fa : -0.5 * fa;
  const double m0 = r.r0 * vn0, m1 = r.r1 * vn1, dp = r.p1 - r.p0;
  acc[0] += ha * (m1 - m0) + lam * r.dx[0];
#pragma unroll
  for (int q = 0; q < 3; ++q)
    acc[1 + q] += ha * (fma(m1, r.v1[q], fma(-m0, r.v0[q], dp * n[q]))) + lam * r.dx[1 + q];
  acc[4] += ha * fma(vn1, r.h1, -vn0 * r.h0) + lam * r.dx[4];
}
// face statics of the lower d-face of the cell at D2 position p
struct KpFace { double n[3], a, ad; };
template <class Z>
__device__ __forceinline__ void kp_load_face(const Z& z, int d, long p, bool viscous,
                                             KpFace& f) {
  const double2 t0 = z.ld_pair(PA_F + 2 * d, p), t1 = z.ld_pair(PA_F + 2 * d + 1, p);
  f.n[0] = t0.x; f.n[1] = t0.y; f.n[2] = t1.x; f.a = t1.y;
  f.ad = viscous ? z.ld_ad(d, p) : 0.0;
}
// a neighbour read straight from the D2 arrays (ghost cells across connection
// boundaries, the far-side triangle of multi-sweep runs, the matrix residual)
template <class Z, class G>
__device__ __forceinline__ void kp_term_mem(const Z& z, const G& g, bool viscous,
                                            long pn, const KpFace& f, bool lower, double* acc) {
  const double2 s0 = z.ld_pair(PA_S, pn), s1 = z.ld_pair(PA_S + 1, pn), s2 = z.ld_pair(PA_S + 2, pn);
  const double2 x0 = z.ld_pair(PA_X, pn), x1 = z.ld_pair(PA_X + 1, pn), x2 = z.ld_pair(PA_X + 2, pn);
  const double s[AGX_NEQ] = {s0.x, s0.y, s1.x, s1.y, s2.x};
  const double dx[AGX_NEQ] = {x0.x, x0.y, x1.x, x1.y, x2.x};
  KpRec r;
  kp_build_rec(g, s, s2.y, viscous ? z.ld_vf(pn) : 0.0, dx, r);
  kp_term(r, f.n, f.a, f.ad, viscous, lower, acc);
}

// Hand-off between the planes: DATA-TAGGED values, no flag.  Every x a sweep launch stores
// carries the launch's tag in the two low mantissa bits of each double (<= 3 ulp, 7e-16
// relative: five orders inside the parity budget; the tagged value is also the one handed
// to the in-plane successors, so every consumer sees the same number).  The plane above
// polls the doubles it needs themselves (sc1 loads) and goes on the moment they carry this
// launch's tag -- what was there before carries the tag of an earlier writer launch
// (prepare, the previous half sweep, an upload), all of which tag what they write and
// advance the block's epoch.  Against flag + payload (store, drain, barrier, flag store,
// poll, barrier, load) this takes the store acknowledgement, the flag's own trip and one
// barrier out of every plane-to-plane hop.
struct KpArgs {
  int* ticket;      // next k-plane to hand out
  int* err;
  int spin_limit;
  unsigned tag;     // epoch & 3 of this launch
  long long* trace;  // -DAGX_KP_TRACE builds: 6 timestamps per step of the middle plane
};
#ifdef AGX_KP_TRACE
#define KP_STAMP(n) do { if (tid == 0 && kp.trace && k == b.nk / 2) \
    kp.trace[(size_t)(t + 1) * 6 + (n)] = wall_clock64(); } while (0)
#else
#define KP_STAMP(n) do {} while (0)
#endif

struct KpCell {   // everything of one cell that does not depend on this launch
  double2 o[3];           // forward: (b0,b1) (b2,b3) (b4,1/a); backward: own (x0,x1) (x2,x3) (x4,1/a)
  double2 s[3], qs[3];    // state + c: own, sweep-side k-neighbour
  double vf, qvf;
  KpFace f[3];            // faces towards the sweep side
  int i, j, pos, use;     // use bit d: sweep-side neighbour d counts
  bool act;
};

// FULL: both triangles count (matrixSweeps > 1 or an initialised x); CONN: the
// block has interblock / periodic surfaces, i.e. ghost cells can be neighbours.
// The common case (one sweep, physical boundaries only) compiles without either
// path: the step loop then has a single divergent branch (lane has a cell or not).
#ifndef KP_SLEEP
#define KP_SLEEP 2
#endif
template <bool FWD, bool FULL, bool CONN, int CH>
__global__ void __launch_bounds__(256, 1)
k_lusgs_kp(KpBlk b, KpGas g, int viscous, int nsl, KpArgs kp) {
  extern __shared__ double kp_lds[];          // [2][KP_NV][nsl]
  __shared__ int s_kt, s_ok;
  const KpBlk& z = b;
  const int tid = threadIdx.x;
  const int ng = b.ng;
  const int nsteps = b.ni + b.nj - 1;
  const bool visc = viscous != 0;
  if (tid == 0) s_ok = 1;
  // records of cells that do not exist are read (and weighted with a zero face
  // area): keep every slot finite from the start
  for (int n = tid; n < 2 * KP_NV * nsl; n += 256) kp_lds[n] = 0.0;
  for (;;) {
    __syncthreads();
    if (tid == 0) s_kt = atomicAdd(kp.ticket, 1);
    __syncthreads();
    const int kt = __builtin_amdgcn_readfirstlane(s_kt);   // wave-uniform by construction
    if (kt >= b.nk) break;
    const int k = FWD ? kt : b.nk - 1 - kt;
    const int kq = FWD ? k - 1 : k + 1;       // plane on the sweep side
    const bool has_pre = kt > 0;              // ... is swept by this launch
    // does the k-neighbour count at all (wave-uniform)?  Without connection
    // surfaces only inside the block.
    const bool k_any = has_pre || (CONN && b.side_conn[FWD ? 4 : 5] != 0);
    const long kbase = (long)(k + ng) * z.ps, qbase = (long)(kq + ng) * z.ps;
    // geometry of step t: diagonal d, first cell jmin, count, plane position of
    // the first cell, and "pn": plane position of in-plane neighbour (ie -+ 1, je)
    // is pn + je
    auto step_geom = [&](int t, int& d, int& jmin, int& cnt, int& p0, int& pn) {
      d = FWD ? t : nsteps - 1 - t;
      jmin = max(0, d - (b.ni - 1));
      cnt = min(b.nj - 1, d) - jmin + 1;
      const int de = d + 2 * ng;
      p0 = z.dstart_cf(de) + (jmin + ng) - z.jlo(de);
      const int dq = FWD ? de - 1 : de + 1;
      pn = z.dstart_cf(dq) - z.jlo(dq);
    };
    auto fetch = [&](int t, KpCell* cc) {
      int d, jmin, cnt, p0, pn;
      step_geom(t, d, jmin, cnt, p0, pn);
#pragma unroll
      for (int m = 0; m < CH; ++m) {
        KpCell& c = cc[m];
        // a lane without a cell loads the diagonal's first cell (and ignores it): no
        // branch around the loads -- at a join of paths with and without loads in
        // flight the compiler's wait for anything older turns into a wait for everything
        const int nraw = tid + 256 * m;
        c.act = nraw < cnt;
        const int n = c.act ? nraw : 0;
        c.j = jmin + n;
        c.i = d - c.j;
        c.pos = p0 + n;
        // wave-uniform array bases (SGPRs) + one 32-bit lane offset
        const unsigned vo = (unsigned)c.pos * 16u;
#pragma unroll
        for (int h = 0; h < 3; ++h) {
          c.o[h] = ld16(z.pab((FWD ? PA_B : PA_X) + h) + kbase * 16, vo);
          c.s[h] = ld16(z.pab(PA_S + h) + kbase * 16, vo);
          c.qs[h] = ld16(z.pab(PA_S + h) + qbase * 16, vo);
        }
        // (all loads unconditional: a select around a load costs a wait; the
        // viscous-factor array is zero in inviscid runs, plane kq always exists)
        c.vf = ld8(z.vfb() + kbase * 8, (unsigned)c.pos * 8u);
        c.qvf = ld8(z.vfb() + qbase * 8, (unsigned)c.pos * 8u);
        // the faces towards the sweep side: own lower faces going forward, the
        // upper neighbours' lower faces going back
        const long fb[3] = {FWD ? kbase : kbase + pn, FWD ? kbase : kbase + pn + 1,
                            FWD ? kbase : kbase + z.ps};
        const unsigned fo[3] = {FWD ? vo : (unsigned)(c.j + ng) * 16u,
                                FWD ? vo : (unsigned)(c.j + ng) * 16u, vo};
#pragma unroll
        for (int q = 0; q < 3; ++q) {
          const double2 t0 = ld16(z.pab(PA_F + 2 * q) + fb[q] * 16, fo[q]);
          const double2 t1 = ld16(z.pab(PA_F + 2 * q + 1) + fb[q] * 16, fo[q]);
          c.f[q].n[0] = t0.x; c.f[q].n[1] = t0.y; c.f[q].n[2] = t1.x; c.f[q].a = t1.y;
          c.f[q].ad = ld8(z.adb(q) + fb[q] * 8, fo[q] >> 1);
        }
        // which sweep-side neighbours count (ImplicitLower / ImplicitUpper: physical
        // or across a connection surface); a face that does not count gets area 0
        const bool in_i = FWD ? c.i > 0 : c.i < b.ni - 1;
        const bool in_j = FWD ? c.j > 0 : c.j < b.nj - 1;
        int use = (in_i ? 1 : 0) | (in_j ? 2 : 0) | (has_pre ? 4 : 0);
        if (CONN) {
          if (FWD) {
            if (!in_i && bc_is_connection(b, c.i, c.j, k, 1)) use |= 1 | 8;
            if (!in_j && bc_is_connection(b, c.i, c.j, k, 3)) use |= 2 | 16;
            if (!has_pre && bc_is_connection(b, c.i, c.j, k, 5)) use |= 4;
          } else {
            if (!in_i && bc_is_connection(b, c.i + 1, c.j, k, 2)) use |= 1 | 8;
            if (!in_j && bc_is_connection(b, c.i, c.j + 1, k, 4)) use |= 2 | 16;
            if (!has_pre && bc_is_connection(b, c.i, c.j, k + 1, 6)) use |= 4;
          }
        }
        c.use = use;            // bits 3, 4: in-plane neighbour q is a ghost cell
      }
    };
    // One step = one diagonal.  At its top the x of the k-neighbours (plane kq, same
    // positions) and the data of step t+1 are requested; the two in-plane terms (records
    // of the previous diagonal, LDS) are formed meanwhile; then the k-neighbour's values
    // are checked for this launch's tag and asked for again until they carry it (see
    // KpArgs); the k-term, x = ..., the tagged stores and the record for the next
    // diagonal follow.  ONE barrier per step (the LDS records); nothing waits for a store.
    KpCell cur[CH], nxt[CH];
    double qx[CH][AGX_NEQ];
    // sc1 loads: the producer (another CU) stored these write-through
    auto qx_request = [&]() {
#pragma unroll
      for (int m = 0; m < CH; ++m) {
#pragma unroll
        for (int e = 0; e < AGX_NEQ; ++e) {       // (lanes without a cell: see fetch)
          const double* p = reinterpret_cast<const double*>(
              z.pab(PA_X + (e >> 1)) + qbase * 16 + (unsigned)cur[m].pos * 16u) + (e & 1);
          qx[m][e] = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
    };
    // (branch-free: one mask of all the low words against the tag)
    auto qx_landed = [&]() {
      unsigned bad = 0;
#pragma unroll
      for (int m = 0; m < CH; ++m) {
#pragma unroll
        for (int e = 0; e < AGX_NEQ; ++e)
          bad |= cur[m].act ? ((unsigned)__double_as_longlong(qx[m][e]) ^ kp.tag) : 0u;
      }
      return (bad & 3u) == 0u;
    };
#pragma unroll
    for (int m = 0; m < CH; ++m) {
      cur[m].act = false;
#pragma unroll
      for (int e = 0; e < AGX_NEQ; ++e) qx[m][e] = 0.0;
    }
    int jmin_prev = 0;
    // the data of step 0, complete before the loop: every path into the step then has
    // nothing of the running register set in flight (a load the compiler believes pending
    // at a join is waited for together with everything older -- the stores included)
    fetch(0, cur);
    __builtin_amdgcn_s_waitcnt(0x0F70);
    for (int t = 0; t < nsteps; ++t) {
      KP_STAMP(0);
      if (k_any) qx_request();
      fetch(min(t + 1, nsteps - 1), nxt);       // (unconditional: see fetch)
      KP_STAMP(1);
      {
        int d, jmin, cnt, p0, pn;
      step_geom(t, d, jmin, cnt, p0, pn);
      double* lw = kp_lds + (size_t)(t & 1) * KP_NV * nsl;          // this step's records
      const double* lr = kp_lds + (size_t)((t & 1) ^ 1) * KP_NV * nsl;  // previous diagonal
      double accs[CH][AGX_NEQ];
#pragma unroll
      for (int m = 0; m < CH; ++m) {
        const KpCell& c = cur[m];
        double* acc = accs[m];
#pragma unroll
        for (int e = 0; e < AGX_NEQ; ++e) acc[e] = 0.0;
        // (lanes without a cell run along on the diagonal's first cell and store nothing:
        // no branch around loads or stores, see fetch)
        // in-plane neighbours: slot of cell j' in the previous diagonal's records
        // is j' - jmin_prev + 1 (clamped: a neighbour outside the block has a zero
        // face area and any finite record will do)
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const int jn = q == 0 ? c.j : (FWD ? c.j - 1 : c.j + 1);
          const double* rp = lr + min(max(jn - jmin_prev + 1, 0), nsl - 1);
          KpRec r;
#pragma unroll
          for (int e = 0; e < AGX_NEQ; ++e) r.dx[e] = rp[e * nsl];
          r.r1 = rp[5 * nsl]; r.v1[0] = rp[6 * nsl]; r.v1[1] = rp[7 * nsl]; r.v1[2] = rp[8 * nsl];
          r.p1 = rp[9 * nsl]; r.h1 = rp[10 * nsl];
          r.r0 = rp[11 * nsl]; r.v0[0] = rp[12 * nsl]; r.v0[1] = rp[13 * nsl]; r.v0[2] = rp[14 * nsl];
          r.p0 = rp[15 * nsl]; r.h0 = rp[16 * nsl]; r.cs = rp[17 * nsl]; r.vf = rp[18 * nsl];
          if (CONN && (c.use & (8 << q))) {
            // ghost cell across a connection boundary: from the arrays
            const long pg = kbase + pn + (c.j + ng) + (q == 0 ? 0 : (FWD ? -1 : 1));
            kp_term_mem(z, g, visc, pg, c.f[q], FWD, acc);
          } else {
            const bool on = (c.use & (1 << q)) != 0;
            kp_term(r, c.f[q].n, on ? c.f[q].a : 0.0, on ? c.f[q].ad : 0.0, true, FWD, acc);
          }
        }
      }
      KP_STAMP(2);
      if (has_pre) {
        // the k-neighbours' x of THIS launch (wave by wave: no barrier, no flag)
        int spins = 0;
        while (!__all(qx_landed())) {
          __builtin_amdgcn_s_sleep(KP_SLEEP);
          if (++spins > kp.spin_limit ||
              ((spins & 63) == 0 &&
               __hip_atomic_load(kp.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) {
            s_ok = 0;
            __hip_atomic_store(kp.err, 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            break;
          }
          qx_request();
          // (waited for here, inside the loop: a request left pending at the loop's exit
          // makes the compiler's scoreboard wait for it -- and for everything older, the
          // prefetch included -- wherever its registers are written next)
          __builtin_amdgcn_s_waitcnt(0x0F70);
        }
      }
      KP_STAMP(3);
      // the prefetch of the next diagonal, requested at the top of the step together with
      // the k-neighbours' x that has just been seen: it has landed or is about to.  Waited
      // for HERE, before this step's stores are issued (see the stores)
      __builtin_amdgcn_s_waitcnt(0x0F70);
#pragma unroll
      for (int m = 0; m < CH; ++m) {
        const KpCell& c = cur[m];
        double* acc = accs[m];
        const long own = kbase + c.pos;
        // the k-neighbour last: its x was requested at the top of this step
        if (k_any) {
          const double qs5[AGX_NEQ] = {c.qs[0].x, c.qs[0].y, c.qs[1].x, c.qs[1].y, c.qs[2].x};
          KpRec r;
          kp_build_rec(g, qs5, c.qs[2].y, c.qvf, qx[m], r);
          const bool on = CONN ? (c.use & 4) != 0 : true;
          kp_term(r, c.f[2].n, on ? c.f[2].a : 0.0, on ? c.f[2].ad : 0.0, true, FWD, acc);
        }
        const double ainv = c.o[2].y;
        const double ov[AGX_NEQ] = {c.o[0].x, c.o[0].y, c.o[1].x, c.o[1].y, c.o[2].x};
        double xn[AGX_NEQ];
        if (FULL) {
          // the other triangle with the values of the previous sweep, read in place
          const int de = d + 2 * ng;
          const int dqo = FWD ? de + 1 : de - 1;
          const long po = kbase + z.dstart_cf(dqo) - z.jlo(dqo) + (c.j + ng);
          KpFace ff;
          if (FWD) {
            if (c.i < b.ni - 1 || bc_is_connection(b, c.i + 1, c.j, k, 2)) {
              kp_load_face(z, 0, po, visc, ff);
              kp_term_mem(z, g, visc, po, ff, false, acc);
            }
            if (c.j < b.nj - 1 || bc_is_connection(b, c.i, c.j + 1, k, 4)) {
              kp_load_face(z, 1, po + 1, visc, ff);
              kp_term_mem(z, g, visc, po + 1, ff, false, acc);
            }
            if (k < b.nk - 1 || bc_is_connection(b, c.i, c.j, k + 1, 6)) {
              kp_load_face(z, 2, own + z.ps, visc, ff);
              kp_term_mem(z, g, visc, own + z.ps, ff, false, acc);
            }
          } else {
            if (c.i > 0 || bc_is_connection(b, c.i, c.j, k, 1)) {
              kp_load_face(z, 0, own, visc, ff);
              kp_term_mem(z, g, visc, po, ff, true, acc);
            }
            if (c.j > 0 || bc_is_connection(b, c.i, c.j, k, 3)) {
              kp_load_face(z, 1, own, visc, ff);
              kp_term_mem(z, g, visc, po - 1, ff, true, acc);
            }
            if (k > 0 || bc_is_connection(b, c.i, c.j, k, 5)) {
              kp_load_face(z, 2, own, visc, ff);
              kp_term_mem(z, g, visc, own - z.ps, ff, true, acc);
            }
          }
          double bv[AGX_NEQ] = {ov[0], ov[1], ov[2], ov[3], ov[4]};
          if (!FWD) {   // going back the own record holds x: fetch b
            const double2 b0 = z.ld_pair(PA_B, own), b1 = z.ld_pair(PA_B + 1, own),
                          b2 = z.ld_pair(PA_B + 2, own);
            bv[0] = b0.x; bv[1] = b0.y; bv[2] = b1.x; bv[3] = b1.y; bv[4] = b2.x;
          }
#pragma unroll
          for (int e = 0; e < AGX_NEQ; ++e) xn[e] = (bv[e] + acc[e]) * ainv;
        } else if (FWD) {
#pragma unroll
          for (int e = 0; e < AGX_NEQ; ++e) xn[e] = (ov[e] + acc[e]) * ainv;
        } else {
#pragma unroll
          for (int e = 0; e < AGX_NEQ; ++e) xn[e] = ov[e] + acc[e] * ainv;
        }
        // the launch's tag goes into every double; one 16-byte write-through (sc1) store
        // per pair: what an agent-scope atomic store lowers to, at twice the width HIP's
        // atomics offer; on its way while the record is formed
#pragma unroll
        for (int e = 0; e < AGX_NEQ; ++e) xn[e] = kp_tagged(xn[e], kp.tag);
        if (c.act) {
          // Inline asm on purpose: the compiler's wait-count pass must NOT see these stores.
          // With loads and stores pending together it stops counting and waits for
          // everything (vmcnt(0)) -- the stores' acknowledgements included, which is a
          // trip to memory and back.  Unseen, they cost a wait only where an OLDER
          // operation is waited for, and the step is arranged so that there is none: the
          // prefetch of the next diagonal is drained just above (it has had the whole step
          // to land), and the next wait is the one for the next step's k-neighbours,
          // a step from now.
          typedef double v2d __attribute__((ext_vector_type(2)));
          const unsigned po = (unsigned)c.pos * 16u;
          const v2d v0 = {xn[0], xn[1]}, v1 = {xn[2], xn[3]}, v2 = {xn[4], ainv};
          // (s_nop 4: the base may have been written by a v_readlane just before --
          // VALU-writes-SGPR -> VMEM needs 5 wait states, and the compiler's hazard
          // recogniser does not look inside inline asm)
          asm volatile("s_nop 4\n\tglobal_store_dwordx4 %0, %1, %2 sc1"
                       :: "v"(po), "v"(v0), "s"(z.pab(PA_X + 0) + kbase * 16) : "memory");
          asm volatile("s_nop 4\n\tglobal_store_dwordx4 %0, %1, %2 sc1"
                       :: "v"(po), "v"(v1), "s"(z.pab(PA_X + 1) + kbase * 16) : "memory");
          asm volatile("s_nop 4\n\tglobal_store_dwordx4 %0, %1, %2 sc1"
                       :: "v"(po), "v"(v2), "s"(z.pab(PA_X + 2) + kbase * 16) : "memory");
        }
        // hand-over to the next diagonal through LDS
        const double s5[AGX_NEQ] = {c.s[0].x, c.s[0].y, c.s[1].x, c.s[1].y, c.s[2].x};
        KpRec r;
        kp_build_rec(g, s5, c.s[2].y, c.vf, xn, r);
        if (c.act) {
        double* wp = lw + (c.j - jmin + 1);
#pragma unroll
        for (int e = 0; e < AGX_NEQ; ++e) wp[e * nsl] = xn[e];
        wp[5 * nsl] = r.r1; wp[6 * nsl] = r.v1[0]; wp[7 * nsl] = r.v1[1]; wp[8 * nsl] = r.v1[2];
        wp[9 * nsl] = r.p1; wp[10 * nsl] = r.h1;
        wp[11 * nsl] = r.r0; wp[12 * nsl] = r.v0[0]; wp[13 * nsl] = r.v0[1]; wp[14 * nsl] = r.v0[2];
        wp[15 * nsl] = r.p0; wp[16 * nsl] = r.h0; wp[17 * nsl] = r.cs; wp[18 * nsl] = r.vf;
        }
      }
      jmin_prev = jmin;
      }
      KP_STAMP(4);
      __syncthreads();          // the records of this diagonal are in place
      if (!s_ok) return;
#pragma unroll
      for (int m = 0; m < CH; ++m) cur[m] = nxt[m];
      KP_STAMP(5);
    }
  }
}

// ---------------------------------------------------------------------------
// linearSolver::AXmB :58-90 / Residual :92-109 as a pure reduction in D2 index
// space: one thread per plane position, all six neighbours are coalesced rows of
// the neighbouring diagonals / planes.
__global__ void __launch_bounds__(256)
k_matrix_resid_d2(BlockDev b, GasDev g, SolverDev sp, int nsplit, NormPartial* partials) {
  const D2Dev& z = b.d2;
  // 1-D grid of (position chunk, k) pairs.  Workgroup n runs on XCD n % 8, each
  // with its own L2.  A cell's in-plane neighbours sit one diagonal (~ one chunk)
  // away, its k-neighbours one plane away, so every XCD gets a contiguous RANGE of
  // chunks (a band of diagonals) and walks it plane by plane: planes k-1 and k of
  // the band are still in its L2 from the previous two rounds, and only the two
  // chunks bordering the band are fetched by two XCDs (measured at 256^3: 13.7 GB
  // of L2 misses per launch with the chunks dealt round-robin, 1.77 -> 1.40 ms).
  // nsplit > 1 walks that many narrower bands per XCD one after the other.
  const int nchunk = (z.Pi * z.Pj + 255) / 256;
  const int per = (nchunk + 8 * nsplit - 1) / (8 * nsplit);   // chunks per band
  const int xcd = blockIdx.x % 8, m = blockIdx.x / 8;
  const int band = (m / (per * b.nk)) * 8 + xcd, rem = m % (per * b.nk);
  const int k = rem / per, chunk = band * per + rem % per;
  const int t = chunk * 256 + threadIdx.x;
  const bool visc = sp.viscous != 0;
  double r[AGX_NEQ] = {0, 0, 0, 0, 0};
  bool active = false;
  if (chunk < nchunk && t < z.Pi * z.Pj) {
    const int ij = z.ij_of_pos[t];
    const int ie = ij & 0xffff, je = ij >> 16;
    const int i = ie - b.ng, j = je - b.ng;
    active = i >= 0 && i < b.ni && j >= 0 && j < b.nj;
    if (active) {
      const long kbase = (long)(k + b.ng) * z.ps;
      const long own = kbase + t;
      const int de = ie + je;
      const long plo = kbase + z.dstart[de - 1] - z.jlo(de - 1) + je;   // (ie-1, je)
      const long pup = kbase + z.dstart[de + 1] - z.jlo(de + 1) + je;   // (ie+1, je)
      double acc[AGX_NEQ] = {0, 0, 0, 0, 0};
      const long nlo[3] = {plo, plo - 1, own - z.ps}, nup[3] = {pup, pup + 1, own + z.ps};
      const int c3[3] = {i, j, k}, n3[3] = {b.ni, b.nj, b.nk};
      KpFace ff;
#pragma unroll
      for (int d = 0; d < 3; ++d) {
        if (c3[d] > 0 || bc_is_connection(b, i, j, k, 2 * d + 1)) {
          kp_load_face(z, d, own, visc, ff);
          kp_term_mem(z, g, visc, nlo[d], ff, true, acc);
        }
      }
#pragma unroll
      for (int d = 0; d < 3; ++d) {
        if (c3[d] < n3[d] - 1 ||
            bc_is_connection(b, i + (d == 0), j + (d == 1), k + (d == 2), 2 * d + 2)) {
          kp_load_face(z, d, nup[d], visc, ff);
          kp_term_mem(z, g, visc, nup[d], ff, false, acc);
        }
      }
      const double2 x0 = z.pa(PA_X)[own], x1 = z.pa(PA_X + 1)[own], x2 = z.pa(PA_X + 2)[own];
      const double2 b0 = z.pa(PA_B)[own], b1 = z.pa(PA_B + 1)[own], b2 = z.pa(PA_B + 2)[own];
      const double a = 1.0 / b2.y;
      const double xv[AGX_NEQ] = {x0.x, x0.y, x1.x, x1.y, x2.x};
      const double bv[AGX_NEQ] = {b0.x, b0.y, b1.x, b1.y, b2.x};
#pragma unroll
      for (int e = 0; e < AGX_NEQ; ++e) r[e] = -(xv[e] * a - acc[e] - bv[e]);
    }
  }
  norm_block_reduce(r, 0, active, partials + blockIdx.x);
}

// ---------------------------------------------------------------------------
// The same reduction MARCHING along k: a thread keeps state + c, x and the viscous factor of
// ITS plane position for the planes k-1, k, k+1 in registers (and the k-face statics of
// k and k+1), so the two k-neighbours -- in the form above two more reads of S and X per
// cell, 8.6 GB moved for 4.7 GB algorithmic at 256^3 -- cost nothing; only the four
// in-plane neighbours come from the arrays (rows of the adjacent diagonals, which the
// neighbouring chunks of the same XCD band are reading as their own at the same time).
// Grid: (position chunk, k-chunk) pairs, a contiguous band of chunks per XCD.
#ifndef AGX_MR_WAVES
#define AGX_MR_WAVES 1
#endif
__global__ void __launch_bounds__(256, AGX_MR_WAVES)
k_matrix_resid_d2m(BlockDev b, GasDev g, SolverDev sp, int kc, NormPartial* partials) {
  const D2Dev& z = b.d2;
  const int nchunk = (z.Pi * z.Pj + 255) / 256;
  const int per = (nchunk + 7) / 8;                 // chunks per XCD band
  const int xcd = blockIdx.x % 8, m = blockIdx.x / 8;
  const int chunk = xcd * per + m % per, kch = m / per;
  const int k0 = kch * kc, k1 = min(k0 + kc, b.nk);
  const int t = chunk * 256 + threadIdx.x;
  const bool visc = sp.viscous != 0;
  double l2[AGX_NEQ] = {0, 0, 0, 0, 0};
  bool active = false;
  int i = 0, j = 0, je = 0, de = 0;
  if (chunk < nchunk && t < z.Pi * z.Pj) {
    const int ij = z.ij_of_pos[t];
    const int ie = ij & 0xffff;
    je = ij >> 16;
    i = ie - b.ng; j = je - b.ng; de = ie + je;
    active = i >= 0 && i < b.ni && j >= 0 && j < b.nj;
  }
  if (active) {
    // plane-relative positions of the in-plane neighbours (ie -+ 1, je); (ie, je -+ 1) sit
    // one position below / above them
    const long qlo = z.dstart[de - 1] - z.jlo(de - 1) + je;
    const long qup = z.dstart[de + 1] - z.jlo(de + 1) + je;
    struct Col { double s[AGX_NEQ], cs, x[AGX_NEQ], vf; };
    auto ld_col = [&](long p, Col& c) {
      const double2 s0 = z.ld_pair(PA_S, p), s1 = z.ld_pair(PA_S + 1, p), s2 = z.ld_pair(PA_S + 2, p);
      const double2 x0 = z.ld_pair(PA_X, p), x1 = z.ld_pair(PA_X + 1, p), x2 = z.ld_pair(PA_X + 2, p);
      c.s[0] = s0.x; c.s[1] = s0.y; c.s[2] = s1.x; c.s[3] = s1.y; c.s[4] = s2.x; c.cs = s2.y;
      c.x[0] = x0.x; c.x[1] = x0.y; c.x[2] = x1.x; c.x[3] = x1.y; c.x[4] = x2.x;
      c.vf = visc ? z.ld_vf(p) : 0.0;
    };
    auto col_term = [&](const Col& c, const KpFace& f, bool lower, double* acc) {
      KpRec r;
      kp_build_rec(g, c.s, c.cs, c.vf, c.x, r);
      kp_term(r, f.n, f.a, f.ad, visc, lower, acc);
    };
    Col cm, c0, cp;
    KpFace fk0, fk1;
    long own = (long)(k0 + b.ng) * z.ps + t;
    ld_col(own - z.ps, cm);
    ld_col(own, c0);
    kp_load_face(z, 2, own, visc, fk0);
    for (int k = k0; k < k1; ++k, own += z.ps) {
      const long kbase = own - t;
      ld_col(own + z.ps, cp);                       // plane k+1 (a ghost plane at the top)
      kp_load_face(z, 2, own + z.ps, visc, fk1);
      const double2 b0 = z.pa(PA_B)[own], b1 = z.pa(PA_B + 1)[own], b2 = z.pa(PA_B + 2)[own];
      double acc[AGX_NEQ] = {0, 0, 0, 0, 0};
      KpFace ff;
      // the order of k_matrix_resid_d2: lower i, j, k, then upper i, j, k
      if (i > 0 || bc_is_connection(b, i, j, k, 1)) {
        kp_load_face(z, 0, own, visc, ff);
        kp_term_mem(z, g, visc, kbase + qlo, ff, true, acc);
      }
      if (j > 0 || bc_is_connection(b, i, j, k, 3)) {
        kp_load_face(z, 1, own, visc, ff);
        kp_term_mem(z, g, visc, kbase + qlo - 1, ff, true, acc);
      }
      if (k > 0 || bc_is_connection(b, i, j, k, 5)) col_term(cm, fk0, true, acc);
      if (i < b.ni - 1 || bc_is_connection(b, i + 1, j, k, 2)) {
        kp_load_face(z, 0, kbase + qup, visc, ff);
        kp_term_mem(z, g, visc, kbase + qup, ff, false, acc);
      }
      if (j < b.nj - 1 || bc_is_connection(b, i, j + 1, k, 4)) {
        kp_load_face(z, 1, kbase + qup + 1, visc, ff);
        kp_term_mem(z, g, visc, kbase + qup + 1, ff, false, acc);
      }
      if (k < b.nk - 1 || bc_is_connection(b, i, j, k + 1, 6)) col_term(cp, fk1, false, acc);
      const double a = 1.0 / b2.y;
      const double bv[AGX_NEQ] = {b0.x, b0.y, b1.x, b1.y, b2.x};
#pragma unroll
      for (int e = 0; e < AGX_NEQ; ++e) {
        const double r = -(c0.x[e] * a - acc[e] - bv[e]);
        l2[e] += r * r;
      }
      cm = c0; c0 = cp; fk0 = fk1;
    }
  }
  norm_block_fold(l2, -1.0e300, 0x7fffffffffffffffLL, partials + blockIdx.x);
}

// ---------------------------------------------------------------------------
// procBlock::UpdateBlock / ImplicitTimeAdvance (procBlock.cpp:826-872, :902)
// with x read from the D2 array through the tile transposition; norms as in
// k_update.  Grid: (ni / 32, nj / 32, nk) tiles of the physical cells.
__global__ void __launch_bounds__(256)
k_update_d2(BlockDev b, GasDev g, SolverDev sp, int last_mm, NormPartial* partials) {
  __shared__ double sv[AGX_NEQ][TT][TRS];
  __shared__ unsigned short s_tab[TT * TT];
  const D2Dev& z = b.d2;
  const int tid = threadIdx.x;
  tile_table(s_tab);
  __syncthreads();
  const int i0 = blockIdx.x * TT, j0 = blockIdx.y * TT, k = blockIdx.z;
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    const unsigned tl = s_tab[tid + 256 * m];
    const int tli = tl & 0xff, tlj = tl >> 8;
    const int i = i0 + tli, j = j0 + tlj;
    if (i < b.ni && j < b.nj) {
      const long p = (long)(k + b.ng) * z.ps + z.pos2(i + b.ng, j + b.ng);
      const double2 x0 = z.pa(PA_X)[p], x1 = z.pa(PA_X + 1)[p], x2 = z.pa(PA_X + 2)[p];
      sv[0][tlj][tli] = x0.x; sv[1][tlj][tli] = x0.y; sv[2][tlj][tli] = x1.x;
      sv[3][tlj][tli] = x1.y; sv[4][tlj][tli] = x2.x;
    }
  }
  __syncthreads();
  const int li = tid & (TT - 1);
  double l2[AGX_NEQ] = {0, 0, 0, 0, 0};
  double vmax = -1.0e300;
  long long vlin = 0x7fffffffffffffffLL;
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    const int lj = (tid >> 5) + 8 * m;
    const int i = i0 + li, j = j0 + lj;
    if (i < b.ni && j < b.nj) {
      const long q = b.idx(i, j, k);
      double r[AGX_NEQ], s[AGX_NEQ], du[AGX_NEQ], ns[AGX_NEQ];
      load5(b.resid, q, r);
      load5(b.state, q, s);
#pragma unroll
      for (int e = 0; e < AGX_NEQ; ++e) du[e] = sv[e][lj][li];
      update_prim_with_cons(g, s, du, ns);
      store5(b.state, q, ns);
      // (gridLevel::ResetDiagonal: the inviscid residual kernels ASSIGN the scalar
      // diagonal, so there is nothing to reset on this path)
      if (sp.bdf2 && last_mm) {           // gridLevel.cpp:425-428
        double u[AGX_NEQ];
        load5(b.consn, q, u);
        store5(b.consnm1, q, u);
      }
      const long lin0 = (((long)k * b.nj + j) * b.ni + i) * AGX_NEQ;
#pragma unroll
      for (int e = 0; e < AGX_NEQ; ++e) {
        l2[e] += r[e] * r[e];
        if (r[e] > vmax || (r[e] == vmax && lin0 + e < vlin)) { vmax = r[e]; vlin = lin0 + e; }
      }
    }
  }
  const long bid = ((long)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
  norm_block_fold(l2, vmax, vlin, partials + bid);
}

}  // namespace agx
