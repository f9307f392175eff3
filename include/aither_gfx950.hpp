// aither_gfx950.hpp -- C++14 host layer over the C-ABI of aither_gfx950.h.
//
// The reference is C++ and its driver (src/main.cpp) talks to the path through
// mgSolution::{StoreOldSolution, Iterate} and the residual / resid value types.
// This header is what a C++ host links against: the same member names, the same
// argument meaning and the same error behaviour (message on std::cerr followed
// by exit(EXIT_FAILURE), main.cpp / procBlock.cpp convention), on top of
// nothing but the extern "C" entry points.  Header-only, no torch, no Python.
//
//   reference                                   here
//   ---------------------------------------------------------------------------
//   class resid (include/resid.hpp:24-75)       aither_gfx950::resid
//   class residual (include/varArray.hpp)       aither_gfx950::residual
//   mgSolution::StoreOldSolution (:103-114)     hotPath::StoreOldSolution
//   mgSolution::Iterate (:246-269)              hotPath::Iterate
//   GetFinestGridLevel pack (procBlock.cpp:4491) hotPath::Download
//   mgSolution with several levels (:128-262)   multigrid::{StoreOldSolution, Iterate,
//                                               CycleAtLevel, Restriction, Relax, Prolongation}
//
// AGX_SYMBOL_PREFIX lets the test suite build the same driver against the CPU
// oracle (prefix ora_), which exports the identical entry points.
#ifndef AITHER_GFX950_HPP
#define AITHER_GFX950_HPP
#include <algorithm>
#include <cstdlib>
#include <iostream>
#include <memory>
#include <string>
#include <vector>
#include "aither_gfx950.h"

#ifndef AGX_SYMBOL_PREFIX
#define AGX_SYMBOL_PREFIX agx_
#endif
#define AGX_CAT2(a, b) a##b
#define AGX_CAT(a, b) AGX_CAT2(a, b)
#define AGX_SYM(name) AGX_CAT(AGX_SYMBOL_PREFIX, name)

namespace aither_gfx950 {

// L-infinity residual with its location, resid.hpp:24-75
class resid {
  double linf_ = 0.0;
  int blk_ = 0, i_ = 0, j_ = 0, k_ = 0, eqn_ = 0;

 public:
  resid() = default;
  resid(double a, int b, int c, int d, int e, int f)
      : linf_(a), blk_(b), i_(c), j_(d), k_(e), eqn_(f) {}
  double Linf() const { return linf_; }
  int Block() const { return blk_; }
  int ILoc() const { return i_; }
  int JLoc() const { return j_; }
  int KLoc() const { return k_; }
  int Eqn() const { return eqn_; }
  void UpdateMax(double a, int b, int c, int d, int e, int f) {
    if (a > linf_) *this = resid(a, b, c, d, e, f);
  }
  void Zero() { *this = resid(); }
};

// per-equation sum of squared residuals (what procBlock.cpp:858 accumulates)
class residual {
  std::vector<double> data_;

 public:
  explicit residual(int numEqns = 5) : data_(numEqns, 0.0) {}
  int Size() const { return static_cast<int>(data_.size()); }
  double &operator[](int r) { return data_[r]; }
  const double &operator[](int r) const { return data_[r]; }
  double *data() { return data_.data(); }
  void Zero() { data_.assign(data_.size(), 0.0); }
};

// The accelerated part of one rank's gridLevel: blocks, their boundary
// conditions and connections, and the per-iteration calls.
class hotPath {
  AGX_CAT(AGX_SYMBOL_PREFIX, ctx) *ctx_ = nullptr;
  int numEqns_ = 5;

  static void Check(int rc, const char *what) {
    if (rc != 0) {   // reference convention: report and stop
      const char *m = AGX_SYM(last_error)();
      std::cerr << "ERROR: Error in " << what << ": " << (m ? m : "") << std::endl;
      exit(EXIT_FAILURE);
    }
  }

 public:
  hotPath(int device, int rank) {
    Check(AGX_SYM(ctx_create)(device, rank, &ctx_), "hotPath::hotPath (ctx_create)");
  }
  ~hotPath() { if (ctx_) AGX_SYM(ctx_destroy)(ctx_); }
  hotPath(const hotPath &) = delete;
  hotPath &operator=(const hotPath &) = delete;

  void Configure(const agx_config &cfg) {
    numEqns_ = cfg.n_eq;
    Check(AGX_SYM(config_set)(ctx_, &cfg), "hotPath::Configure");
  }
  // geometry of one procBlock (ghost-inclusive arrays in the reference's own
  // layout), its boundarySurfaces and the initial state; returns the block id
  int AddBlock(const agx_block_geom &geom, const std::vector<agx_bc_surface> &surfs) {
    int id = -1;
    Check(AGX_SYM(block_create)(ctx_, &geom, &id), "hotPath::AddBlock");
    Check(AGX_SYM(block_set_bcs)(ctx_, id, static_cast<int>(surfs.size()), surfs.data()),
          "hotPath::AddBlock (boundary conditions)");
    return id;
  }
  int AddConnection(const agx_connection &conn) {
    int id = -1;
    Check(AGX_SYM(conn_create)(ctx_, &conn, &id), "hotPath::AddConnection");
    return id;
  }
  // multi-rank runs: the transport that stands where the reference has
  // MPI_Sendrecv / MPI_Reduce (multiArray3d.hpp:830-866, main.cpp:254-264).  With
  // one installed, Iterate exchanges the ghost slabs of connections to other ranks
  // itself and returns norms already reduced over the ranks.
  void SetExchange(const agx_exchange &ex) {
    Check(AGX_SYM(set_exchange)(ctx_, &ex), "hotPath::SetExchange");
  }
  // the library's own RCCL transport; id128 from RcclUniqueId() on one rank,
  // handed to the others by the host (MPI_Bcast)
  static void RcclUniqueId(void *id128) {
    Check(AGX_SYM(rccl_unique_id)(id128), "hotPath::RcclUniqueId");
  }
  void UseRccl(const void *id128, int numRanks, int rank) {
    Check(AGX_SYM(rccl_exchange_create)(ctx_, id128, numRanks, rank), "hotPath::UseRccl");
  }
  void Finalize() { Check(AGX_SYM(setup_finalize)(ctx_), "hotPath::Finalize"); }
  void UploadState(int block, const double *stateAos) {
    Check(AGX_SYM(state_upload)(ctx_, block, stateAos), "hotPath::UploadState");
  }
  void Download(int block, int field, double *aos) const {
    Check(AGX_SYM(field_download)(ctx_, block, field, aos), "hotPath::Download");
  }

  // WriteFunFile (output.cpp:209-437): the block's payload of a function file -- the
  // listed variables (AGX_OUT_*), physical cells, variable by variable, dimensional --
  // formed on the device; what comes back is what the caller writes to the file
  void OutputPack(int block, const std::vector<int32_t> &vars, double *out) const {
    Check(AGX_SYM(output_pack)(ctx_, block, static_cast<int>(vars.size()), vars.data(), out),
          "hotPath::OutputPack");
  }
  // WriteRestart (output.cpp:651-752): the block's payload, numEqns + 1 values per cell;
  // secondSolution: consVarsNm1 (multilevel time integration)
  void RestartPack(int block, bool secondSolution, double *out) const {
    Check(AGX_SYM(restart_pack)(ctx_, block, secondSolution ? 1 : 0, out),
          "hotPath::RestartPack");
  }

  // mgSolution::StoreOldSolution: consVarsN <- cons(state) (and N-1 on the first
  // step of a multilevel-in-time scheme)
  void StoreOldSolution(bool alsoNm1) {
    Check(AGX_SYM(store_time_n)(ctx_, alsoNm1 ? 1 : 0), "hotPath::StoreOldSolution");
  }

  // mgSolution::Iterate for one rank: boundary conditions, residual, time step,
  // explicit update or implicit solve.  residL2 is accumulated into, residLinf
  // only replaced by a larger value; returns sum(matrix residual^2) / size, which
  // the caller reduces over ranks and square-roots (main.cpp:271).
  double Iterate(int mm, double cfl, residual &residL2, resid &residLinf) {
    agx_linf linf;
    linf.linf = residLinf.Linf();
    linf.block = residLinf.Block();
    linf.i = residLinf.ILoc();
    linf.j = residLinf.JLoc();
    linf.k = residLinf.KLoc();
    linf.eqn = residLinf.Eqn();
    linf.pad_ = 0;
    double matrixResid = 0.0;
    Check(AGX_SYM(iterate)(ctx_, mm, cfl, residL2.data(), &linf, &matrixResid),
          "hotPath::Iterate");
    residLinf.UpdateMax(linf.linf, linf.block, linf.i, linf.j, linf.k, linf.eqn);
    return matrixResid;
  }

  friend class multigrid;
};

// mgSolution with more than one grid level (mgSolution.cpp:128-262) on one rank: a hotPath
// per gridLevel (finest first; every level configured, given its coarsened blocks and
// surfaces -- procBlock::GetCoarseMeshAndBCs -- and finalized by the caller) and, per level
// but the coarsest and per block, the arrays gridLevel::Coarsen builds: toCoarse_ (int32
// triples), volWeightFactor_, prolongCoeffs_ (7 doubles), all per physical fine cell in
// k-j-i order.  The cycle is the reference's; what crosses between two levels runs in the
// library (agx_mg_*).  Runs of more than one rank: every level's hotPath gets the rank's
// blocks of that level and its own SetExchange / UseRccl; the levels' connections to other
// ranks then go through agx_halo_exchange like the finest one's (restriction and prolongation
// never leave a block), and the norms Iterate returns are this rank's.
class multigrid {
  struct transfer {
    std::vector<int32_t> toCoarse;
    std::vector<double> volFac, coeffs;
  };
  std::vector<std::unique_ptr<hotPath>> solution_;
  std::vector<std::vector<transfer>> transfer_;   // [fine level][block]
  int mgCycleIndex_ = 1;                          // V: 1, W: 2
  int matrixSweeps_ = 1;
  bool lusgs_ = false;

  static void Check(int rc, const char *what) { hotPath::Check(rc, what); }
  AGX_CAT(AGX_SYMBOL_PREFIX, ctx) *Ctx(int ll) const { return solution_[ll]->ctx_; }

  // gridLevel::GetBoundaryConditions, CalcResidual, CalcTimeStep
  void BoundaryAndResidual(int ll, int mm, double cfl) {
    Check(AGX_SYM(phase_bc_faces)(Ctx(ll)), "multigrid (phase_bc_faces)");
    Check(AGX_SYM(halo_exchange)(Ctx(ll), AGX_HALO_STATE), "multigrid (halo_exchange)");
    Check(AGX_SYM(phase_bc_edges)(Ctx(ll)), "multigrid (phase_bc_edges)");
    Check(AGX_SYM(phase_residual)(Ctx(ll), mm, cfl), "multigrid (phase_residual)");
  }

 public:
  multigrid(int cycleIndex, const agx_config &cfg)
      : mgCycleIndex_(cycleIndex), matrixSweeps_(cfg.matrix_sweeps),
        lusgs_(cfg.matrix_solver == AGX_SOLVER_LUSGS || cfg.matrix_solver == AGX_SOLVER_BLUSGS) {}
  int NumGridLevels() const { return static_cast<int>(solution_.size()); }
  hotPath &Level(int ll) { return *solution_[ll]; }
  // levels are added finest first
  void AddLevel(std::unique_ptr<hotPath> level) {
    solution_.push_back(std::move(level));
    transfer_.emplace_back();
  }
  // the transfer arrays of block `blk` of level `fl` towards level fl + 1 (blocks in order)
  void AddTransfer(int fl, std::vector<int32_t> toCoarse, std::vector<double> volFac,
                   std::vector<double> coeffs) {
    transfer_[fl].push_back(transfer{std::move(toCoarse), std::move(volFac), std::move(coeffs)});
  }

  // mgSolution::StoreOldSolution (:103-114): every level
  void StoreOldSolution(bool alsoNm1) {
    for (auto &sol : solution_) sol->StoreOldSolution(alsoNm1);
  }
  // linearSolver::Relax (linearSolver.cpp:430-470, :509-536); returns sum(r^2) / size
  double Relax(int ll, int sweeps) {
    for (int ii = 0; ii < sweeps; ++ii) {
      Check(AGX_SYM(halo_exchange)(Ctx(ll), AGX_HALO_UPDATE), "multigrid::Relax (SwapUpdate)");
      Check(AGX_SYM(phase_relax_forward)(Ctx(ll), ii), "multigrid::Relax");
      if (lusgs_) {
        Check(AGX_SYM(halo_exchange)(Ctx(ll), AGX_HALO_UPDATE), "multigrid::Relax (SwapUpdate)");
        Check(AGX_SYM(phase_relax_backward)(Ctx(ll), ii), "multigrid::Relax");
      }
    }
    Check(AGX_SYM(halo_exchange)(Ctx(ll), AGX_HALO_UPDATE), "multigrid::Relax (SwapUpdate)");
    double l2 = 0.0;
    Check(AGX_SYM(mg_matrix_residual)(Ctx(ll), &l2), "multigrid::Relax (Residual)");
    return l2;
  }
  // gridLevel::Restriction (gridLevel.cpp:538-595)
  void Restriction(int fi, int mm, double cfl) {
    const int ci = fi + 1;
    const auto &tr = transfer_[fi];
    for (size_t bb = 0; bb < tr.size(); ++bb) {
      Check(AGX_SYM(mg_restrict)(Ctx(fi), Ctx(ci), static_cast<int>(bb), AGX_MG_STATE,
                                 tr[bb].toCoarse.data(), tr[bb].volFac.data()),
            "multigrid::Restriction (state)");
      if (mm == 0) solution_[ci]->StoreOldSolution(false);
    }
    BoundaryAndResidual(ci, mm, cfl);
    Check(AGX_SYM(mg_invert_diagonal)(Ctx(ci)), "multigrid::Restriction (InvertDiagonal)");
    for (size_t bb = 0; bb < tr.size(); ++bb)
      Check(AGX_SYM(mg_restrict)(Ctx(fi), Ctx(ci), static_cast<int>(bb), AGX_MG_UPDATE,
                                 tr[bb].toCoarse.data(), tr[bb].volFac.data()),
            "multigrid::Restriction (update)");
    Check(AGX_SYM(halo_exchange)(Ctx(ci), AGX_HALO_UPDATE), "multigrid::Restriction (SwapUpdate)");
    for (size_t bb = 0; bb < tr.size(); ++bb)
      Check(AGX_SYM(mg_restrict)(Ctx(fi), Ctx(ci), static_cast<int>(bb), AGX_MG_FORCING,
                                 tr[bb].toCoarse.data(), nullptr),
            "multigrid::Restriction (forcing)");
  }
  // SubtractFromUpdate + Prolongation (mgSolution.cpp:189-192)
  void Prolongation(int ci) {
    const auto &tr = transfer_[ci - 1];
    for (size_t bb = 0; bb < tr.size(); ++bb)
      Check(AGX_SYM(mg_prolong)(Ctx(ci), Ctx(ci - 1), static_cast<int>(bb),
                                tr[bb].toCoarse.data(), tr[bb].coeffs.data()),
            "multigrid::Prolongation");
  }
  // mgSolution::CycleAtLevel (:160-205)
  double CycleAtLevel(int fl, int mm, double cfl) {
    if (fl == NumGridLevels() - 1) return Relax(fl, matrixSweeps_);   // recursive base case
    const int sweeps = std::max(matrixSweeps_ / 2, 1);
    Relax(fl, sweeps);
    const int cl = fl + 1;
    Restriction(fl, mm, cfl);
    Check(AGX_SYM(mg_save_update)(Ctx(cl)), "multigrid::CycleAtLevel (coarseDu)");
    for (int ii = 0; ii < mgCycleIndex_; ++ii) CycleAtLevel(cl, mm, cfl);
    Prolongation(cl);
    return Relax(fl, sweeps);
  }
  // mgSolution::Iterate / ImplicitUpdate (:207-262); residL2 accumulated into, residLinf only
  // replaced by a larger value, the return value as hotPath::Iterate
  double Iterate(int mm, double cfl, residual &residL2, resid &residLinf) {
    BoundaryAndResidual(0, mm, cfl);
    Check(AGX_SYM(phase_implicit_begin)(Ctx(0)), "multigrid::Iterate (InvertDiagonal)");
    const double matrixResid = CycleAtLevel(0, mm, cfl);
    agx_linf linf;
    linf.linf = residLinf.Linf();
    linf.block = residLinf.Block();
    linf.i = residLinf.ILoc();
    linf.j = residLinf.JLoc();
    linf.k = residLinf.KLoc();
    linf.eqn = residLinf.Eqn();
    linf.pad_ = 0;
    Check(AGX_SYM(phase_implicit_update)(Ctx(0), mm, residL2.data(), &linf),
          "multigrid::Iterate (UpdateBlocks)");
    residLinf.UpdateMax(linf.linf, linf.block, linf.i, linf.j, linf.k, linf.eqn);
    for (int ll = 1; ll < NumGridLevels(); ++ll)      // ResetDiagonal on every level
      Check(AGX_SYM(mg_reset_diagonal)(Ctx(ll)), "multigrid::Iterate (ResetDiagonal)");
    return matrixResid;
  }
};

}  // namespace aither_gfx950
#endif
