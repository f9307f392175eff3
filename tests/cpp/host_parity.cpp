// host_parity.cpp -- drives the hot path from C++ (the reference's language)
// through include/aither_gfx950.hpp.  Built twice by tests/cpp/Makefile: against
// libaither_gfx950.so (prefix agx_) and against the CPU oracle (prefix ora_);
// tests/test_cpp_host.py writes the case file, runs both and compares.
//
//   host_parity <case.bin> <out.bin>
//
// case.bin (little endian), written by tests/test_cpp_host.py:
//   int32 magic, n_blocks, n_conns, nonlinear_iterations, n_steps, store_time_n,
//         multilevel ; double cfl[n_steps] ; agx_config
//   per block : int32 ni,nj,nk,ng,parent,global_pos ; farea_i, farea_j, farea_k,
//               vol, center, width_i, width_j, width_k, wall_dist ;
//               int32 n_surfaces ; agx_bc_surface[] ; state
//   per connection : agx_connection (local_block already set)
// out.bin: per iteration l2[n_eq], linf, matrix residual ; then every block's state.
#include <cstdint>
#include <cstdio>
#include <vector>
#ifdef HOST_PARITY_ORACLE
#include "../../oracle/oracle.h"
#define AGX_SYMBOL_PREFIX ora_
#endif
#include "../../include/aither_gfx950.hpp"

using aither_gfx950::hotPath;
using aither_gfx950::resid;
using aither_gfx950::residual;

namespace {
template <class T> T get(FILE *f) {
  T v;
  if (fread(&v, sizeof(T), 1, f) != 1) { std::cerr << "ERROR: short case file\n"; exit(EXIT_FAILURE); }
  return v;
}
std::vector<double> getv(FILE *f, size_t n) {
  std::vector<double> v(n);
  if (n && fread(v.data(), sizeof(double), n, f) != n) {
    std::cerr << "ERROR: short case file\n";
    exit(EXIT_FAILURE);
  }
  return v;
}
}  // namespace

int main(int argc, char **argv) {
  if (argc != 3) { std::cerr << "usage: host_parity <case.bin> <out.bin>\n"; return 2; }
  FILE *f = fopen(argv[1], "rb");
  if (!f) { std::cerr << "ERROR: cannot open " << argv[1] << "\n"; return 2; }
  if (get<int32_t>(f) != 0x31584741) { std::cerr << "ERROR: bad magic\n"; return 2; }
  const int nBlocks = get<int32_t>(f), nConns = get<int32_t>(f);
  const int nonlin = get<int32_t>(f), nSteps = get<int32_t>(f);
  const int storeN = get<int32_t>(f), multilevel = get<int32_t>(f);
  const std::vector<double> cfl = getv(f, nSteps);
  const agx_config cfg = get<agx_config>(f);

  hotPath path(0, 0);
  path.Configure(cfg);
  std::vector<std::vector<double>> keep;           // geometry must outlive AddBlock
  std::vector<std::vector<double>> states;
  std::vector<size_t> stateSize;
  for (int b = 0; b < nBlocks; ++b) {
    agx_block_geom g{};
    g.ni = get<int32_t>(f); g.nj = get<int32_t>(f); g.nk = get<int32_t>(f);
    g.ng = get<int32_t>(f); g.parent_block = get<int32_t>(f); g.global_pos = get<int32_t>(f);
    const size_t G = 2 * g.ng, ci = g.ni + G, cj = g.nj + G, ck = g.nk + G;
    const size_t sizes[9] = {(ci + 1) * cj * ck * 4, ci * (cj + 1) * ck * 4, ci * cj * (ck + 1) * 4,
                             ci * cj * ck, ci * cj * ck * 3, ci * cj * ck, ci * cj * ck,
                             ci * cj * ck, ci * cj * ck};
    const double **dst[9] = {&g.farea_i, &g.farea_j, &g.farea_k, &g.vol, &g.center,
                             &g.width_i, &g.width_j, &g.width_k, &g.wall_dist};
    for (int a = 0; a < 9; ++a) {
      keep.push_back(getv(f, sizes[a]));
      *dst[a] = keep.back().data();
    }
    const int nSurf = get<int32_t>(f);
    std::vector<agx_bc_surface> surfs(nSurf);
    for (auto &s : surfs) s = get<agx_bc_surface>(f);
    const int id = path.AddBlock(g, surfs);
    if (id != b) { std::cerr << "ERROR: unexpected block id\n"; return 2; }
    stateSize.push_back(ci * cj * ck * cfg.n_eq);
    states.push_back(getv(f, stateSize.back()));
  }
  for (int c = 0; c < nConns; ++c) path.AddConnection(get<agx_connection>(f));
  fclose(f);
  path.Finalize();
  for (int b = 0; b < nBlocks; ++b) path.UploadState(b, states[b].data());

  FILE *o = fopen(argv[2], "wb");
  if (!o) { std::cerr << "ERROR: cannot open " << argv[2] << "\n"; return 2; }
  // the time loop of main.cpp:232-275 reduced to the calls on the path
  for (int nn = 0; nn < nSteps; ++nn) {
    if (storeN) path.StoreOldSolution(multilevel && nn == 0);
    for (int mm = 0; mm < nonlin; ++mm) {
      residual residL2(cfg.n_eq);
      resid residLinf;
      const double matrixResid = path.Iterate(mm, cfl[nn], residL2, residLinf);
      fwrite(residL2.data(), sizeof(double), cfg.n_eq, o);
      const double tail[2] = {residLinf.Linf(), matrixResid};
      fwrite(tail, sizeof(double), 2, o);
      const int32_t loc[5] = {residLinf.Block(), residLinf.ILoc(), residLinf.JLoc(),
                              residLinf.KLoc(), residLinf.Eqn()};
      fwrite(loc, sizeof(int32_t), 5, o);
    }
  }
  for (int b = 0; b < nBlocks; ++b) {
    std::vector<double> s(stateSize[b]);
    path.Download(b, AGX_FIELD_STATE, s.data());
    fwrite(s.data(), sizeof(double), s.size(), o);
  }
  fclose(o);
  return 0;
}
