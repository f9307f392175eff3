// agx_lusgs.hpp -- scalar LU-SGS on the diagonal-ordered ("D2") layout.
//
// lusgs::LUSGS_Forward / LUSGS_Backward (src/linearSolver.cpp:341-428) sweep
// the cells in hyperplane order (HyperplaneReorder src/utility.cpp:377-398):
// cell (i,j,k) needs x of (i-1,j,k), (i,j-1,k), (i,j,k-1).  Mapping for
// gfx950:
//
//   * ONE WORKGROUP PER k-PLANE.  The workgroup marches the anti-diagonals
//     d = i + j of its plane in order; the cells of a diagonal are mutually
//     independent, the two in-plane neighbours lie on the previous diagonal
//     (handed over through LDS), the third neighbour is the same (i,j) of the
//     plane below, which the workgroup of that plane finished a few steps
//     earlier.  The k-planes therefore run as a pipeline synchronised by one
//     progress counter per plane (256 planes <-> 256 CUs for a 256^3 block);
//     any topological order of the dependency graph gives the reference's x,
//     so results are unchanged.
//   * D2 LAYOUT.  Everything the sweeps touch is stored per k-plane in
//     diagonal order: position(i,j) = dstart[i+j] + (j - jlo(i+j)).  The cells
//     of a diagonal are contiguous, so every load/store of a step is a
//     coalesced row although consecutive lanes sit in different grid rows;
//     the k-neighbour is the same position one plane stride away.  Values are
//     stored in PAIRS (double2 arrays, one 16-byte load per lane: the step is
//     bound by the number of vector-memory instructions a lone workgroup can
//     issue, not by bytes).  The static part -- per lower face the unit normal and
//     |A| (pairs), |A| / (centre-to-centre distance) (single) -- is formed once; the dynamic
//     part (state, sound speed, viscous factor, right-hand side b, 1/diagonal)
//     by k_lusgs_prepare, which transposes 32 x 32 tiles through LDS (tile
//     diagonals are contiguous segments of the D2 plane) while it forms
//     AddDiagonalTerms / Invert / InitializeMatrixUpdate
//     (linearSolver.cpp:111-188).
//   * ONE NONLINEAR EVALUATION PER CELL.  RusanovScalarOffDiagonal
//     (fluxJacobian.cpp:122-162) needs F(U_n + dU_n) - F(U_n) of a neighbour n
//     through the shared face.  Everything that does not depend on the face --
//     prim(U_n + dU_n), rho H, sound speed, the viscous factor -- is formed ONCE
//     when the cell is finished and handed to its successors as a record (LDS);
//     a face then costs ~45 fp64 instructions instead of ~140.
//   * x never leaves D2 during an implicit iteration: the matrix residual
//     (linearSolver::Residual :92-109) runs in D2 index space, the update
//     kernel reads x through the same tile transposition.
#pragma once
#include "agx_device.hpp"

namespace agx {

// pair arrays (double2 per cell) of the D2 slab of a block, then single arrays
enum {
  PA_S = 0,    // (rho,u) (v,w) (p,c): state + speed of sound
  PA_B = 3,    // (b0,b1) (b2,b3) (b4,1/a): right-hand side, linearSolver.cpp:370-374
  PA_X = 6,    // (x0,x1) (x2,x3) (x4,1/a): update x_
  PA_F = 9,    // per lower face d: (nx,ny) (nz,|A|) at 2 * d + m; static
  PA_COUNT = 15,
  D1_VF = 2 * PA_COUNT,   // single: viscous factor of the cell (spectralRadius.hpp:94-160)
  D1_AD = D1_VF + 1,      // singles, per lower face d: |A| / dist; static
  D2_DOUBLES = 2 * PA_COUNT + 4
};

struct D2Dev {
  double* base;          // D2_DOUBLES arrays' worth of nd2 doubles
  long nd2, ps;          // cells per array, plane stride (padded to 16 cells)
  const int* dstart;     // [Pi + Pj + 1] start of diagonal de = ie + je in a plane
  const int* ij_of_pos;  // [Pi * Pj] ie | je << 16 of a plane position
  int Pi, Pj;            // padded plane dims (n + 2 ng)
  __device__ __forceinline__ double2* pa(int id) const {
    return reinterpret_cast<double2*>(base + 2L * id * nd2);
  }
  __device__ __forceinline__ double* vf() const { return base + (long)D1_VF * nd2; }
  __device__ __forceinline__ double* ad(int d) const { return base + (long)(D1_AD + d) * nd2; }
  __device__ __forceinline__ double ld_ad(int d, long cell) const { return ad(d)[cell]; }
  __device__ __forceinline__ double2 ld_pair(int id, long cell) const { return pa(id)[cell]; }
  __device__ __forceinline__ double ld_vf(long cell) const { return vf()[cell]; }
  __device__ __forceinline__ int jlo(int de) const { return max(0, de - (Pi - 1)); }
  // position of padded cell (ie, je) inside a plane
  __device__ __forceinline__ int pos2(int ie, int je) const {
    const int de = ie + je;
    return dstart[de] + je - jlo(de);
  }
};

// What the sweep kernel needs of a block and of the gas model.  Passing the whole
// BlockDev / GasDev (about 100 kernel-argument SGPRs) made the compiler park
// scalars in VGPR lanes: ~400 v_readlane / v_writelane in the step loop.
struct KpBlk {
  double* base;
  const int* dstart;
  const agx_bc_surface* surf;
  long nd2, ps;
  int Pi, Pj, ni, nj, nk, ng;
  int nsurf, nsurf_i, nsurf_j, nsurf_k;
  int side_conn[6];
  // start of diagonal de in a plane, closed form of the dstart table (scalar
  // arithmetic: a table lookup is a memory access in the step loop)
  __device__ __forceinline__ int dstart_cf(int de) const {
    const int a = min(Pi, Pj), bm = max(Pi, Pj);
    if (de <= a) return de * (de + 1) / 2;
    if (de <= bm) return a * (a + 1) / 2 + (de - a) * a;
    const int r = Pi + Pj - 1 - de;
    return Pi * Pj - r * (r + 1) / 2;
  }
  // array `id` (pair arrays: 16 bytes per cell) as a byte pointer, wave-uniform
  __device__ __forceinline__ const char* pab(int id) const {
    return reinterpret_cast<const char*>(base) + (size_t)id * (size_t)nd2 * 16;
  }
  __device__ __forceinline__ const char* vfb() const {
    return reinterpret_cast<const char*>(base) + (size_t)D1_VF * (size_t)nd2 * 8;
  }
  __device__ __forceinline__ const char* adb(int d) const {
    return reinterpret_cast<const char*>(base) + (size_t)(D1_AD + d) * (size_t)nd2 * 8;
  }
  __device__ __forceinline__ double ld_ad(int d, long cell) const {
    return *reinterpret_cast<const double*>(adb(d) + cell * 8);
  }
  __device__ __forceinline__ double2 ld_pair(int id, long cell) const {
    return *reinterpret_cast<const double2*>(pab(id) + cell * 16);
  }
  __device__ __forceinline__ double ld_vf(long cell) const {
    return *reinterpret_cast<const double*>(vfb() + cell * 8);
  }
  __device__ __forceinline__ int jlo(int de) const { return max(0, de - (Pi - 1)); }
};
struct KpGas { double hf, n, inv_n; };
// uniform base + 32-bit lane offset: the SGPR-base addressing mode, no 64-bit
// vector address arithmetic
__device__ __forceinline__ double2 ld16(const char* ubase, unsigned voff) {
  return *reinterpret_cast<const double2*>(ubase + voff);
}
__device__ __forceinline__ double ld8(const char* ubase, unsigned voff) {
  return *reinterpret_cast<const double*>(ubase + voff);
}

}  // namespace agx
