// host_multigrid.cpp -- the multigrid cycle driven from C++ (the reference's language) through
// aither_gfx950::multigrid (include/aither_gfx950.hpp).  Built twice by tests/cpp/Makefile
// (product library / CPU oracle); tests/test_cpp_host.py writes the files and compares with
// the Python driver.
//
//   host_multigrid <cycle index> <transfers.bin> <out.bin> <level0.bin> [<level1.bin> ...]
//
// level*.bin: the case files of host_parity.cpp, one per grid level, finest first (one rank).
// transfers.bin: int32 magic, n_levels - 1; per fine level: int32 n_blocks; per block:
//   int64 n_cells, int32 to_coarse[3 n], double vol_fac[n], double coeffs[7 n].
// out.bin: per iteration l2[n_eq], linf, matrix residual, location; then the finest level's
// states.
#include <cstdint>
#include <cstdio>
#include <vector>
#ifdef HOST_PARITY_ORACLE
#include "../../oracle/oracle.h"
#define AGX_SYMBOL_PREFIX ora_
#endif
#include "../../include/aither_gfx950.hpp"

using aither_gfx950::hotPath;
using aither_gfx950::multigrid;
using aither_gfx950::resid;
using aither_gfx950::residual;

namespace {
template <class T> T get(FILE *f) {
  T v;
  if (fread(&v, sizeof(T), 1, f) != 1) { std::cerr << "ERROR: short file\n"; exit(EXIT_FAILURE); }
  return v;
}
template <class T> std::vector<T> getv(FILE *f, size_t n) {
  std::vector<T> v(n);
  if (n && fread(v.data(), sizeof(T), n, f) != n) { std::cerr << "ERROR: short file\n"; exit(EXIT_FAILURE); }
  return v;
}
struct levelInfo {
  int nonlin = 1, nSteps = 0, storeN = 0, multilevel = 0;
  std::vector<double> cfl;
  agx_config cfg;
  std::vector<size_t> stateSize;
};
// one level: the case file of host_parity.cpp, all blocks on this rank
std::unique_ptr<hotPath> LoadLevel(const char *path, levelInfo &info) {
  FILE *f = fopen(path, "rb");
  if (!f) { std::cerr << "ERROR: cannot open " << path << "\n"; exit(EXIT_FAILURE); }
  if (get<int32_t>(f) != 0x32584741) { std::cerr << "ERROR: bad magic\n"; exit(EXIT_FAILURE); }
  const int nBlocks = get<int32_t>(f), nConns = get<int32_t>(f);
  info.nonlin = get<int32_t>(f); info.nSteps = get<int32_t>(f);
  info.storeN = get<int32_t>(f); info.multilevel = get<int32_t>(f);
  info.cfl = getv<double>(f, info.nSteps);
  info.cfg = get<agx_config>(f);
  std::unique_ptr<hotPath> level(new hotPath(0, 0));
  level->Configure(info.cfg);
  std::vector<std::vector<double>> keep, states;
  for (int b = 0; b < nBlocks; ++b) {
    agx_block_geom g{};
    g.ni = get<int32_t>(f); g.nj = get<int32_t>(f); g.nk = get<int32_t>(f);
    g.ng = get<int32_t>(f); g.parent_block = get<int32_t>(f); g.global_pos = get<int32_t>(f);
    get<int32_t>(f);                                    // rank: one rank here
    const size_t G = 2 * g.ng, ci = g.ni + G, cj = g.nj + G, ck = g.nk + G;
    const size_t sizes[9] = {(ci + 1) * cj * ck * 4, ci * (cj + 1) * ck * 4, ci * cj * (ck + 1) * 4,
                             ci * cj * ck, ci * cj * ck * 3, ci * cj * ck, ci * cj * ck,
                             ci * cj * ck, ci * cj * ck};
    const double **dst[9] = {&g.farea_i, &g.farea_j, &g.farea_k, &g.vol, &g.center,
                             &g.width_i, &g.width_j, &g.width_k, &g.wall_dist};
    for (int a = 0; a < 9; ++a) {
      keep.push_back(getv<double>(f, sizes[a]));
      *dst[a] = keep.back().data();
    }
    const int nSurf = get<int32_t>(f);
    std::vector<agx_bc_surface> surfs(nSurf);
    for (auto &s : surfs) s = get<agx_bc_surface>(f);
    states.push_back(getv<double>(f, ci * cj * ck * info.cfg.n_eq));
    info.stateSize.push_back(states.back().size());
    level->AddBlock(g, surfs);
  }
  for (int c = 0; c < nConns; ++c) level->AddConnection(get<agx_connection>(f));
  fclose(f);
  level->Finalize();
  for (size_t b = 0; b < states.size(); ++b) level->UploadState(static_cast<int>(b), states[b].data());
  return level;
}
}  // namespace

int main(int argc, char **argv) {
  if (argc < 5) {
    std::cerr << "usage: host_multigrid <cycle index> <transfers.bin> <out.bin> <level0.bin> ...\n";
    return 2;
  }
  const int cycleIndex = atoi(argv[1]), nLevels = argc - 4;
  std::vector<levelInfo> info(nLevels);
  std::vector<std::unique_ptr<hotPath>> levels;
  for (int ll = 0; ll < nLevels; ++ll) levels.push_back(LoadLevel(argv[4 + ll], info[ll]));
  multigrid mg(cycleIndex, info[0].cfg);
  for (auto &l : levels) mg.AddLevel(std::move(l));
  FILE *t = fopen(argv[2], "rb");
  if (!t) { std::cerr << "ERROR: cannot open " << argv[2] << "\n"; return 2; }
  if (get<int32_t>(t) != 0x3247474d || get<int32_t>(t) != nLevels - 1) {
    std::cerr << "ERROR: transfers do not match the levels\n";
    return 2;
  }
  for (int fl = 0; fl + 1 < nLevels; ++fl) {
    const int nb = get<int32_t>(t);
    for (int b = 0; b < nb; ++b) {
      const size_t n = static_cast<size_t>(get<int64_t>(t));
      auto tc = getv<int32_t>(t, 3 * n);
      auto vf = getv<double>(t, n);
      auto cf = getv<double>(t, 7 * n);
      mg.AddTransfer(fl, std::move(tc), std::move(vf), std::move(cf));
    }
  }
  fclose(t);

  FILE *o = fopen(argv[3], "wb");
  if (!o) { std::cerr << "ERROR: cannot open " << argv[3] << "\n"; return 2; }
  const levelInfo &top = info[0];
  // the time loop of main.cpp:232-275 reduced to the calls on the path
  for (int nn = 0; nn < top.nSteps; ++nn) {
    if (top.storeN) mg.StoreOldSolution(top.multilevel && nn == 0);
    for (int mm = 0; mm < top.nonlin; ++mm) {
      residual residL2(top.cfg.n_eq);
      resid residLinf;
      const double matrixResid = mg.Iterate(mm, top.cfl[nn], residL2, residLinf);
      fwrite(residL2.data(), sizeof(double), top.cfg.n_eq, o);
      const double tail[2] = {residLinf.Linf(), matrixResid};
      fwrite(tail, sizeof(double), 2, o);
      const int32_t loc[5] = {residLinf.Block(), residLinf.ILoc(), residLinf.JLoc(),
                              residLinf.KLoc(), residLinf.Eqn()};
      fwrite(loc, sizeof(int32_t), 5, o);
    }
  }
  for (size_t b = 0; b < top.stateSize.size(); ++b) {
    std::vector<double> s(top.stateSize[b]);
    mg.Level(0).Download(static_cast<int>(b), AGX_FIELD_STATE, s.data());
    fwrite(s.data(), sizeof(double), s.size(), o);
  }
  fclose(o);
  return 0;
}
