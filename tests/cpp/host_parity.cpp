// host_parity.cpp -- drives the hot path from C++ (the reference's language)
// through include/aither_gfx950.hpp.  Built twice by tests/cpp/Makefile: against
// libaither_gfx950.so (prefix agx_) and against the CPU oracle (prefix ora_);
// tests/test_cpp_host.py writes the case file, runs both and compares.
//
//   host_parity <case.bin> <out.bin> [nranks]
//
// With nranks > 1 the process forks one child per rank; each child builds only
// the blocks of its rank, installs a transport over AF_UNIX socket pairs (the
// stand-in for MPI_Sendrecv / MPI_Allgather on host buffers) through
// hotPath::SetExchange and writes <out.bin>.<rank>.
//
// case.bin (little endian), written by tests/test_cpp_host.py:
//   int32 magic, n_blocks, n_conns, nonlinear_iterations, n_steps, store_time_n,
//         multilevel ; double cfl[n_steps] ; agx_config
//   per block : int32 ni,nj,nk,ng,parent,global_pos,rank ; farea_i, farea_j, farea_k,
//               vol, center, width_i, width_j, width_k, wall_dist ;
//               int32 n_surfaces ; agx_bc_surface[] ; state
//   per connection : agx_connection (local_block already set)
// out.bin: per iteration l2[n_eq], linf, matrix residual ; then every block's state.
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <map>
#include <vector>
#include <sys/socket.h>
#include <sys/wait.h>
#include <unistd.h>
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
// transport of the multi-rank test: blocking socket pairs, host buffers
struct SocketExchange {
  int rank = 0, nranks = 1;
  std::vector<int> fd;                       // fd[p]: socket to rank p
  static bool wr(int fd, const void *p, size_t n) {
    const char *c = static_cast<const char *>(p);
    while (n) { const ssize_t k = write(fd, c, n); if (k <= 0) return false; c += k; n -= k; }
    return true;
  }
  static bool rd(int fd, void *p, size_t n) {
    char *c = static_cast<char *>(p);
    while (n) { const ssize_t k = read(fd, c, n); if (k <= 0) return false; c += k; n -= k; }
    return true;
  }
  // the lower rank of a pair sends first, the higher receives first; both walk
  // their slabs in connection-creation order
  static int Swap(void *u, int n, const agx_slab *s, void *) {
    SocketExchange *x = static_cast<SocketExchange *>(u);
    for (int q = 0; q < n; ++q) {
      const size_t bytes = sizeof(double) * static_cast<size_t>(s[q].count);
      const int fd = x->fd[s[q].peer];
      const bool ok = x->rank < s[q].peer ? (wr(fd, s[q].send, bytes) && rd(fd, s[q].recv, bytes))
                                          : (rd(fd, s[q].recv, bytes) && wr(fd, s[q].send, bytes));
      if (!ok) return 1;
    }
    return 0;
  }
  static int Allgather(void *u, const void *send, void *recv, int64_t bytes, void *) {
    SocketExchange *x = static_cast<SocketExchange *>(u);
    char *out = static_cast<char *>(recv);
    memcpy(out + x->rank * bytes, send, bytes);
    for (int p = 0; p < x->nranks; ++p)
      if (p != x->rank && !wr(x->fd[p], send, bytes)) return 1;
    for (int p = 0; p < x->nranks; ++p)
      if (p != x->rank && !rd(x->fd[p], out + p * bytes, bytes)) return 1;
    return 0;
  }
};
}  // namespace

static int run_rank(const char *casePath, const std::string &outPath, int myRank,
                    SocketExchange *sx);

int main(int argc, char **argv) {
  if (argc != 3 && argc != 4) {
    std::cerr << "usage: host_parity <case.bin> <out.bin> [nranks]\n";
    return 2;
  }
  const int nranks = argc == 4 ? atoi(argv[3]) : 1;
  if (nranks <= 1) return run_rank(argv[1], argv[2], 0, nullptr);
  // one socket pair per pair of ranks, made before the fork
  std::vector<std::vector<int>> fds(nranks, std::vector<int>(nranks, -1));
  for (int a = 0; a < nranks; ++a)
    for (int b = a + 1; b < nranks; ++b) {
      int sv[2];
      if (socketpair(AF_UNIX, SOCK_STREAM, 0, sv)) { std::cerr << "ERROR: socketpair\n"; return 2; }
      fds[a][b] = sv[0];
      fds[b][a] = sv[1];
    }
  std::vector<pid_t> kids;
  for (int r = 0; r < nranks; ++r) {
    const pid_t pid = fork();
    if (pid == 0) {
      SocketExchange sx;
      sx.rank = r; sx.nranks = nranks; sx.fd = fds[r];
      _exit(run_rank(argv[1], std::string(argv[2]) + "." + std::to_string(r), r, &sx));
    }
    kids.push_back(pid);
  }
  int bad = 0;
  for (pid_t k : kids) {
    int st = 0;
    waitpid(k, &st, 0);
    if (!WIFEXITED(st) || WEXITSTATUS(st) != 0) bad = 1;
  }
  return bad;
}

static int run_rank(const char *casePath, const std::string &outPath, int myRank,
                    SocketExchange *sx) {
  FILE *f = fopen(casePath, "rb");
  if (!f) { std::cerr << "ERROR: cannot open " << casePath << "\n"; return 2; }
  if (get<int32_t>(f) != 0x32584741) { std::cerr << "ERROR: bad magic\n"; return 2; }
  const int nBlocks = get<int32_t>(f), nConns = get<int32_t>(f);
  const int nonlin = get<int32_t>(f), nSteps = get<int32_t>(f);
  const int storeN = get<int32_t>(f), multilevel = get<int32_t>(f);
  const std::vector<double> cfl = getv(f, nSteps);
  const agx_config cfg = get<agx_config>(f);

  hotPath path(0, myRank);
  path.Configure(cfg);
  if (sx) {
    agx_exchange ex{};
    ex.user = sx; ex.swap = SocketExchange::Swap; ex.allgather = SocketExchange::Allgather;
    ex.nranks = sx->nranks; ex.host_buffers = 1;
    path.SetExchange(ex);
  }
  std::map<int, int> localId;                      // global block -> id on this rank
  std::vector<std::vector<double>> keep;           // geometry must outlive AddBlock
  std::vector<std::vector<double>> states;
  std::vector<size_t> stateSize;
  for (int b = 0; b < nBlocks; ++b) {
    agx_block_geom g{};
    g.ni = get<int32_t>(f); g.nj = get<int32_t>(f); g.nk = get<int32_t>(f);
    g.ng = get<int32_t>(f); g.parent_block = get<int32_t>(f); g.global_pos = get<int32_t>(f);
    const int blkRank = get<int32_t>(f);
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
    std::vector<double> st = getv(f, ci * cj * ck * cfg.n_eq);
    if (blkRank != myRank) {                     // another rank's block: not built here
      keep.resize(keep.size() - 9);
      continue;
    }
    localId[b] = path.AddBlock(g, surfs);
    stateSize.push_back(st.size());
    states.push_back(std::move(st));
  }
  for (int c = 0; c < nConns; ++c) {
    agx_connection cc = get<agx_connection>(f);
    if (cc.rank[0] != myRank && cc.rank[1] != myRank) continue;
    for (int s = 0; s < 2; ++s)
      cc.local_block[s] = cc.rank[s] == myRank ? localId[cc.block[s]] : -1;
    path.AddConnection(cc);
  }
  fclose(f);
  path.Finalize();
  const int nLocal = static_cast<int>(states.size());
  for (int b = 0; b < nLocal; ++b) path.UploadState(b, states[b].data());

  FILE *o = fopen(outPath.c_str(), "wb");
  if (!o) { std::cerr << "ERROR: cannot open " << outPath << "\n"; return 2; }
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
  for (int b = 0; b < nLocal; ++b) {
    std::vector<double> s(stateSize[b]);
    path.Download(b, AGX_FIELD_STATE, s.data());
    fwrite(s.data(), sizeof(double), s.size(), o);
  }
  fclose(o);
  return 0;
}
