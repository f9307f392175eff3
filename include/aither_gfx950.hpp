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
//
// AGX_SYMBOL_PREFIX lets the test suite build the same driver against the CPU
// oracle (prefix ora_), which exports the identical entry points.
#ifndef AITHER_GFX950_HPP
#define AITHER_GFX950_HPP
#include <cstdlib>
#include <iostream>
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
};

}  // namespace aither_gfx950
#endif
