// agx_api.hip -- C-ABI of libaither_gfx950.so (include/aither_gfx950.h).
//
// Host side of the library: owns device-resident SoA mirrors of every block,
// converts the reference's AoS host arrays at upload/download, precomputes the
// halo index maps and launches the kernels of agx_kernels.hpp in the order of
// mgSolution::Iterate (src/mgSolution.cpp:246-269).  There is no CPU compute
// path here: every entry point fails loudly if HIP is unavailable.
#include "agx_kernels.hpp"
#include <rccl/rccl.h>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <cmath>
#include <vector>
#include <string>
#include <algorithm>

using namespace agx;

static char g_err[512] = "";
static int fail(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof g_err, fmt, ap);
  va_end(ap);
  return 1;
}
#define HIPCHK(call)                                                       \
  do {                                                                     \
    hipError_t e_ = (call);                                                \
    if (e_ != hipSuccess)                                                  \
      return fail("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_),   \
                  __FILE__, __LINE__);                                     \
  } while (0)
// a kernel templated on the solver mode (solver_mode_of, agx_kernels.hpp)
#define AGX_BY_MODE(sp, kernel, grid, block, stream, ...)                                  \
  do {                                                                                     \
    switch (agx::solver_mode_of(sp)) {                                                     \
      case 0: hipLaunchKernelGGL(kernel<0>, grid, block, 0, stream, __VA_ARGS__); break;   \
      case 1: hipLaunchKernelGGL(kernel<1>, grid, block, 0, stream, __VA_ARGS__); break;   \
      default: hipLaunchKernelGGL(kernel<2>, grid, block, 0, stream, __VA_ARGS__); break;  \
    }                                                                                      \
  } while (0)

namespace {

struct Block {
  BlockDev d;
  int global_pos = 0;
  double* slab = nullptr;
  double* d2 = nullptr;       // D2 arrays of the LU-SGS path (agx_lusgs.hpp)
  double* blockmat = nullptr; // block-matrix solvers: a_ | aInv_ | velocityGrad_ planes
  double* sweep_rec = nullptr; // plane-by-plane sweeps: geo | dyn | rhs records (k_sweep_records)
  bool sweep_geo_built = false; // the (static) geometry records exist
  bool x_planes_current = false; // D2 path: the planes of x equal the D2 arrays (multigrid calls)
  // multigrid: forcing | matrix residual | saved update planes (each allocated on first
  // use); the transfer maps of this block as the FINE side (device copies, keyed by the
  // host pointers they were uploaded from); node values of this block as the coarse side
  double *mg_forcing = nullptr, *mg_mres = nullptr, *mg_xsave = nullptr, *mg_nodes = nullptr;
  int* mg_tc = nullptr;
  int* mg_start = nullptr;
  double *mg_vf = nullptr, *mg_cf = nullptr;
  const void *mg_tc_key = nullptr, *mg_vf_key = nullptr, *mg_cf_key = nullptr;
  int* d2_tab = nullptr;      // device: dstart[Pi + Pj] | ij_of_pos[Pi * Pj]
  std::vector<int> dstart;    // host copy (halo index maps)
  // hyperplane-per-launch sweeps captured as graphs: [forward][both triangles][un_is_u]
  hipGraphExec_t sweep_graph[2][2][2] = {};
  int* kp_mem = nullptr;      // k_lusgs_kp: ticket counter
  unsigned kp_epoch = 0;      // writer launches of the D2 x so far (its low bits tag the values)
  // D2 index of padded cell (i, j, k), host side
  long d2idx(int i, int j, int k) const {
    const int ie = i + d.ng, je = j + d.ng, de = ie + je;
    return (long)(k + d.ng) * d.d2.ps + dstart[de] + je - std::max(0, de - (d.d2.Pi - 1));
  }
  bool state_is_a = true;
  agx_bc_surface* surf_dev = nullptr;
  std::vector<agx_bc_surface> surf_host;
  // nonreflecting inlet / outlet surfaces: offsets | gradients | Mach (BlockDev)
  int* nr_off_dev = nullptr;
  double* nr_mem = nullptr;
  long nr_max = 0;            // cells of the largest such surface (0: none)
  // wall-law surfaces (rans): offsets | wallData_ of their faces (BlockDev)
  int* wall_off_dev = nullptr;
  double* wall_mem = nullptr;
};

struct ConnSide {          // what side s receives / sends
  std::vector<long> h_dst, h_src;   // host copies (SoA index space) until agx_setup_finalize is done
  long n = 0;              // cells inserted into side s
  long* dst = nullptr;     // device: ghost cells of side s (this rank's block)
  long* src = nullptr;     // device: partner cells that fill them
  long* dst2 = nullptr;    // the same cells in D2 index space (x of the LU-SGS path)
  long* src2 = nullptr;
};
struct Conn {
  agx_connection c;
  ConnSide side[2];
  // for remote connections: cells of MY block that the partner's ghosts read
  long n_send = 0;
  long* send_src = nullptr;
  long* send_src2 = nullptr;
};

// timing groups of agx_timing_get (include/aither_gfx950.h)
enum { G_RESID = 0, G_UPDATE = 1, G_BC = 2, G_SWEEP = 3, G_VISC = 4, G_PREPARE = 5,
       G_MRESID = 6, G_NGROUP = 7 };

}  // namespace

struct agx_ctx {
  int device = 0, rank = 0;
  hipStream_t stream = nullptr;
  bool have_cfg = false, finalized = false;
  agx_config cfg;
  GasDev gas;
  SolverDev sp;
  std::vector<Block> blocks;
  std::vector<Conn> conns;
  // multi-rank: the installed exchange, one persistent slab pair per remote
  // connection, and the norm records of all ranks
  agx_exchange ex = {};
  bool have_ex = false;
  struct Remote { int cid; long count; double *send = nullptr, *recv = nullptr;
                  double *hsend = nullptr, *hrecv = nullptr; };
  std::vector<Remote> remote;
  std::vector<agx_slab> slabs;
  struct NormRecord { double l2[8]; double mres, linf; int32_t block, i, j, k, eqn, status;
                      double fill[3]; };     // 128 bytes
  NormRecord* rec_dev = nullptr;        // [1 + nranks] device (RCCL)
  NormRecord* rec_host = nullptr;       // [1 + nranks] pinned
  ncclComm_t nccl = nullptr;
  NormPartial* partials = nullptr;
  long n_partials = 0;
  NormPartial* norm_out = nullptr;      // device, one per block
  NormPartial* norm_host = nullptr;     // pinned
  int* err_dev = nullptr;
  int* err_host = nullptr;              // pinned
  double* halo_buf = nullptr;
  long halo_cap = 0;
  double* stage_buf = nullptr;   // AoS staging of uploads / downloads (stage_buffer)
  size_t stage_cap = 0;
  double* rans_rec = nullptr;    // face records of the rans viscous residual (k_rans_faces),
  size_t rans_rec_cap = 0;       // sized for the largest block, shared by all of them
  bool use_gather = false;   // AGX_KERNEL=gather: one-thread-per-cell gather kernel
  bool use_tile = true;      // AGX_KERNEL=tile (default) | march
  int num_cu = 256;          // persistent workgroups of the tile kernel
  bool eager_ghosts = true;  // AGX_EAGER_GHOSTS=0: fill ghost cells at the start of agx_iterate
  // AGX_OVERLAP=0: the slabs of x travel before a DPLUR sweep starts (default: while its
  // interior cells are relaxed, on ov_stream)
  bool overlap = true;
  hipStream_t ov_stream = nullptr;
  hipEvent_t ov_ready = nullptr, ov_done = nullptr;
  bool visc_gather = false;  // AGX_VISC=gather: one-thread-per-cell viscous kernel
  bool visc_march = false;   // AGX_VISC=march: face-once form without LDS staging
  int lusgs_mode = 1;        // AGX_LUSGS=plane (0: launch per hyperplane on the SoA
                             // planes, comparison form) | kp (1, default: agx_lusgs.hpp)
  int spin_limit = 4000000;  // AGX_SPIN_LIMIT: polls before a waiting plane gives up
  bool allow_fuse = true;    // AGX_NO_FUSE=1: separate update kernel
  bool fused_pending = false;
  bool use_graphs = true;    // AGX_GRAPHS=0: launch the hyperplane sweeps one by one
  bool sweep_records = true; // AGX_SWEEP_RECORDS=0: the hyperplane sweeps read the plane-major arrays
  bool sweep_three = true;   // AGX_SWEEP_THREE=0: one lane per cell in the hyperplane sweeps
  hipStream_t cap_stream = nullptr;   // stream the sweep graphs are recorded on
  // several blocks swept hyperplane by hyperplane: the blocks of a half sweep are
  // independent (ghost x comes from the exchange before it), so their chains of
  // launches run side by side on up to 8 branch streams
  std::vector<hipStream_t> branch_streams;
  std::vector<hipEvent_t> branch_events;
  // ... or, hyperplane step by step, all blocks in one launch (k_lusgs_plane_all) from a
  // table of their BlockDev in device memory, one graph per direction / triangle set /
  // un_is_u (AGX_SWEEP_ALL=0: the branch streams)
  bool sweep_all_launch = true;
  BlockDev* blocks_tab = nullptr;        // device
  BlockDev* blocks_tab_host = nullptr;   // pinned
  size_t blocks_tab_n = 0;
  hipGraphExec_t sweep_graph_all[2][2][2] = {};
  // the pipelined half sweep (k_lusgs_pipe): a workgroup per k-plane of every block
  bool mg_coarse = false;                // a coarse multigrid level (agx_mg_restrict made it one)
  bool mres_plane_form = false;          // agx_mg_matrix_residual: k_matrix_resid on every block
  bool sweep_pipe = true;                // AGX_SWEEP_PIPE=0: one launch per hyperplane
  PipeJob* pipe_jobs[2] = {nullptr, nullptr};   // device: pipeline order back / forward
  int* pipe_slot0 = nullptr;             // device
  long long* pipe_progress = nullptr;    // device, 16 words per plane
  unsigned long long* pipe_ticket = nullptr;
  size_t pipe_nblocks = 0;
  int pipe_njobs = 0, pipe_maxsteps = 0, pipe_waves = 4;
  long long pipe_launches = 0;
  // local connections exchanged in one gather + one scatter launch (AGX_HALO_BATCH=0: a
  // launch pair per connection).  Legal when no slice reads a cell another connection's
  // insert writes (checked at agx_setup_finalize); tables per halo selector in device memory
  bool halo_batch = true;
  bool halo_batch_required = false;
  int halo_batch_sides = 0;
  long halo_batch_nmax = 0;
  // levels of the batched exchange (halo_batch_plan): the local connections in table order,
  // and per level its first table entry, its entries and its longest side
  std::vector<int> halo_conn_order;
  struct HaloLevel { int first, sides; long nmax; };
  std::vector<HaloLevel> halo_levels;
  // the state (fused explicit stages) and x (DPLUR) alternate between two sets of planes:
  // one table pair per set, built once each
  HaloSide* halo_tab_dev[5][2][2] = {};   // [what][plane set][gather | scatter]
  bool halo_tab_valid[5][2] = {};
  int halo_set[5] = {};
  HaloSide* halo_tab_host = nullptr;      // pinned staging, 2 * sides entries
  int mresid_split = 1;      // bands of diagonals per XCD in k_matrix_resid_d2 (AGX_MRESID_SPLIT)
  bool mresid_march = true;  // AGX_MRESID=plane: one plane position per thread (comparison form)
  bool have_time_n = false;  // agx_store_time_n has run (nonreflecting BCs read consVarsN)
  // agx_iterate fills the ghost cells for the NEXT call right after the update,
  // behind the norm read-back the host waits for, so that the GPU does not idle
  // while the host turns the iteration around
  bool in_iterate = false, ghosts_prefilled = false;
  // inside agx_iterate the matrix residual is not read back by a host synchronisation of
  // its own (the update would wait for the host): it rides with the update's norms
  bool mres_deferred = false;
  hipEvent_t norm_event = nullptr;
  bool consn_pending = false;   // AssignSolToTimeN deferred into the first residual launch of the step
  bool state_is_time_n = false; // nothing has changed the state since agx_store_time_n
  long fused_parts = 0;
  // timing: hipEvent pairs recorded on the library's stream around each
  // kernel group, resolved lazily in agx_timing_get (no sync while running)
  bool timing = false;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_pool;
  std::vector<std::pair<int, int>> ev_used;   // (group, pool index)
  double t_ms[G_NGROUP] = {};
  long t_n[G_NGROUP] = {};
};

static int my_side(const agx_ctx* c, const Conn& k);

namespace {

void resolve_timing(agx_ctx* c);

struct Timer {   // hipEvents on the library's stream around one kernel group
  agx_ctx* c;
  int slot = -1;
  Timer(agx_ctx* ctx, int g) : c(ctx) {
    if (!c->timing) return;
    if (c->ev_used.size() >= 8192) resolve_timing(c);
    slot = (int)c->ev_used.size();
    if (slot >= (int)c->ev_pool.size()) {
      hipEvent_t a, b;
      if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) {
        slot = -1;
        return;
      }
      c->ev_pool.emplace_back(a, b);
    }
    c->ev_used.emplace_back(g, slot);
    hipEventRecord(c->ev_pool[slot].first, c->stream);
  }
  ~Timer() {
    if (slot >= 0) hipEventRecord(c->ev_pool[slot].second, c->stream);
  }
};

void resolve_timing(agx_ctx* c) {
  if (c->ev_used.empty()) return;
  hipStreamSynchronize(c->stream);
  for (auto& u : c->ev_used) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, c->ev_pool[u.second].first,
                            c->ev_pool[u.second].second) == hipSuccess) {
      c->t_ms[u.first] += ms;
      c->t_n[u.first] += 1;
    }
  }
  c->ev_used.clear();
}

int derive_gas(const agx_config& cfg, GasDev& g) {
  const agx_gas& a = cfg.gas;
  g.R = a.gas_constant;
  g.n = a.n;
  g.inv_n = 1.0 / a.n;
  g.hf = a.heat_of_formation;
  g.cp = a.gas_constant * (a.n + 1.0);   // thermodynamic.hpp:108-113
  g.cv = a.gas_constant * a.n;
  g.gamma = g.cp / g.cv;
  g.prandtl = (4.0 * g.gamma) / (9.0 * g.gamma - 5.0);
  g.inv_prandtl = 1.0 / g.prandtl;
  g.visc_c1 = a.visc_c1; g.visc_s = a.visc_s;
  g.cond_c1 = a.cond_c1; g.cond_s = a.cond_s;
  g.t_ref = a.t_ref;
  g.mu_ref = a.visc_c1 * pow(a.t_ref, 1.5) / (a.t_ref + a.visc_s);  // transport.cpp:58-59
  g.k_nondim = (a.a_ref * a.a_ref * g.mu_ref) / a.t_ref;            // :67
  g.scaling = g.mu_ref / (a.rho_ref * a.a_ref * a.l_ref);            // transport.hpp:43-46
  return 0;
}

dim3 cell_grid(const BlockDev& b, dim3 blk) {
  return dim3((b.ni + blk.x - 1) / blk.x, (b.nj + blk.y - 1) / blk.y, b.nk);
}
const dim3 CELL_BLOCK(64, 4, 1);

Planes5 planes(double* const* p, int n = AGX_NEQ) {
  Planes5 r;
  r.rec = nullptr;
  for (int e = 0; e < AGX_NEQ; ++e) r.p[e] = e < n ? p[e] : nullptr;
  r.stride = 1;
  return r;
}

// upload an AoS host array (dims (ci,cj,ck) incl. gsrc ghosts, ncomp per cell)
// device staging of the AoS <-> SoA conversions: one buffer per context that only grows
// (an output step downloads a dozen fields per block; no hipMalloc / hipFree per call)
static int stage_buffer(agx_ctx* c, size_t doubles, double** out) {
  if (doubles > c->stage_cap) {
    HIPCHK(hipStreamSynchronize(c->stream));
    if (c->stage_buf) HIPCHK(hipFree(c->stage_buf));
    c->stage_buf = nullptr; c->stage_cap = 0;
    HIPCHK(hipMalloc((void**)&c->stage_buf, sizeof(double) * doubles));
    c->stage_cap = doubles;
  }
  *out = c->stage_buf;
  return 0;
}
int upload_aos(agx_ctx* c, Block& b, const double* host, double* const* dst,
               int ncomp, int ci, int cj, int ck, int gsrc) {
  const long n = (long)ci * cj * ck;
  double* tmp = nullptr;
  if (stage_buffer(c, (size_t)n * ncomp, &tmp)) return 1;
  HIPCHK(hipMemcpyAsync(tmp, host, sizeof(double) * n * ncomp,
                        hipMemcpyHostToDevice, c->stream));
  hipLaunchKernelGGL(k_aos_to_soa, dim3((n + 255) / 256), dim3(256), 0,
                     c->stream, tmp, planes(dst, ncomp), ncomp, ci, cj, ck, gsrc, b.d);
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(c->stream));
  return 0;
}
int download_aos(agx_ctx* c, Block& b, double* host, double* const* src,
                 int ncomp, int ci, int cj, int ck, int gsrc) {
  const long n = (long)ci * cj * ck;
  double* tmp = nullptr;
  if (stage_buffer(c, (size_t)n * ncomp, &tmp)) return 1;
  hipLaunchKernelGGL(k_soa_to_aos, dim3((n + 255) / 256), dim3(256), 0,
                     c->stream, tmp, planes(src, ncomp), ncomp, ci, cj, ck, gsrc, b.d);
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpyAsync(host, tmp, sizeof(double) * n * ncomp,
                        hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  return 0;
}

// ---- halo index maps: restates connection::First/SecondSliceIndices
// (boundaryConditions.cpp:1016-1150), AdjustForSlice (:833-860), InsertSlice
// (multiArray3d.hpp:868-918) and GetSwapLoc (boundaryConditions.cpp:3006-3181)
// directly in device-index space.
struct Dims { int n[3]; };
struct MapOut { std::vector<long> dst, src; };

void dirs_of(int boundary, int& d1, int& d2, int& d3) {
  d3 = (boundary - 1) / 2;
  d1 = (d3 + 1) % 3;
  d2 = (d3 + 2) % 3;
}
// receiver = side `recv`; idxR/idxS give device indices for (i,j,k) in the
// receiving / sending block
template <class FR, class FS>
void build_side_map(const agx_connection& cc, int recv, int ng, FR idxR, FS idxS,
                    MapOut& out) {
  const int snd = 1 - recv;
  int orient = cc.orientation;
  if (recv == 1) {                       // connection::SwapOrder :341-364
    if (orient == 4) orient = 5; else if (orient == 5) orient = 4;
  }
  int rd1, rd2, rd3, sd1, sd2, sd3;
  dirs_of(cc.boundary[recv], rd1, rd2, rd3);
  dirs_of(cc.boundary[snd], sd1, sd2, sd3);
  const int r_d1s = cc.d1_start[recv] - ng, r_d1e = cc.d1_end[recv] + ng;
  const int r_d2s = cc.d2_start[recv] - ng, r_d2e = cc.d2_end[recv] + ng;
  const bool r_upper = cc.boundary[recv] % 2 == 0;
  const int blk_start = r_upper ? cc.const_surf[recv] : -ng;
  const bool s_upper = cc.boundary[snd] % 2 == 0;
  const int s_d3s = cc.const_surf[snd] + (s_upper ? -ng : 0);
  const int s_d1s = cc.d1_start[snd] - ng, s_d2s = cc.d2_start[snd] - ng;
  const int s_len1 = cc.d1_end[snd] - cc.d1_start[snd] + 2 * ng;
  const int s_len2 = cc.d2_end[snd] - cc.d2_start[snd] + 2 * ng;
  const int len1 = r_d1e - r_d1s, len2 = r_d2e - r_d2s;
  const int32_t* pb = cc.patch_border + (recv == 0 ? 0 : 4);
  const int aS1 = pb[0] ? ng : 0, aE1 = pb[1] ? ng : 0;
  const int aS2 = pb[2] ? ng : 0, aE2 = pb[3] ? ng : 0;
  const bool llu = (cc.boundary[0] + cc.boundary[1]) % 2 == 0;
  for (int l3 = 0; l3 < ng; ++l3)
    for (int l2 = aS2; l2 < len2 - aE2; ++l2)
      for (int l1 = aS1; l1 < len1 - aE1; ++l1) {
        int a[3], s[3], q1, q2;
        a[rd1] = r_d1s + l1;
        a[rd2] = r_d2s + l2;
        a[rd3] = blk_start + l3;
        if (orient == 2 || orient == 4 || orient == 5 || orient == 7) {
          q2 = (orient == 5 || orient == 7) ? s_len2 - 1 - l1 : l1;
          q1 = (orient == 4 || orient == 7) ? s_len1 - 1 - l2 : l2;
        } else if (sd3 == 0) {           // i-patch rule, cpp:3064-3073
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
        out.dst.push_back(idxR(a[0], a[1], a[2]));
        out.src.push_back(idxS(s[0], s[1], s[2]));
      }
}

int to_device(const std::vector<long>& v, long** out) {
  *out = nullptr;
  if (v.empty()) return 0;
  HIPCHK(hipMalloc((void**)out, sizeof(long) * v.size()));
  HIPCHK(hipMemcpy(*out, v.data(), sizeof(long) * v.size(), hipMemcpyHostToDevice));
  return 0;
}

int ensure_halo_buf(agx_ctx* c, long ndoubles) {
  if (ndoubles <= c->halo_cap) return 0;
  if (c->halo_buf) HIPCHK(hipFree(c->halo_buf));
  HIPCHK(hipMalloc((void**)&c->halo_buf, sizeof(double) * ndoubles));
  c->halo_cap = ndoubles;
  for (auto& v : c->halo_tab_valid) v[0] = v[1] = false;   // (the tables hold slices of this buffer)
  return 0;
}

// what a halo exchange moves: the state planes, or x -- which the D2 LU-SGS path
// keeps in its own arrays and index space (maps dst2 / src2)
bool halo_in_d2(const Block& b, int what) { return what == AGX_HALO_UPDATE && b.d.d2.base; }
// something is about to write x where it lives (the D2 arrays on the diagonal-ordered path)
void x_changed(agx_ctx* c) {
  for (auto& blk : c->blocks) blk.x_planes_current = false;
}
Planes5 halo_planes(Block& b, int what) {
  Planes5 r;
  // (the scatter of an exchange of x also refreshes the sweep records' copy)
  r.rec = what == AGX_HALO_UPDATE ? b.d.sw_dyn : nullptr;
  const bool z2 = halo_in_d2(b, what);
  r.stride = z2 ? 2 : 1;     // x of the D2 path sits in pair arrays (x0,x1) (x2,x3) (x4,-)
#if AGX_NEQ == 7
  if (what == AGX_HALO_TURB) {     // eddyViscosity_, f1_, f2_ in the first three slots
    for (int e = 0; e < AGX_NEQ; ++e) r.p[e] = b.d.turb3[e < 3 ? e : 2];
    return r;
  }
#endif
  if (what == AGX_HALO_VELGRAD_A || what == AGX_HALO_VELGRAD_B) {
    // velocityGrad_ (9 planes) in two halves of five slots; the fifth slot of the
    // second half repeats component 8
    for (int e = 0; e < AGX_NEQ; ++e) {
      const int comp = what == AGX_HALO_VELGRAD_A ? e : std::min(5 + e, 8);
      r.p[e] = b.d.vg + (long)comp * b.d.nplane;
    }
    return r;
  }
  for (int e = 0; e < AGX_NEQ; ++e)
    r.p[e] = what == AGX_HALO_STATE ? b.d.state[e]
             : (z2 ? b.d.d2.base + 2L * (PA_X + (e >> 1)) * b.d.d2.nd2 + (e & 1) : b.d.x[e]);
  return r;
}

int check_device_error(agx_ctx* c) {
  if (*c->err_host) {
    const int code = *c->err_host;
    *c->err_host = 0;
    hipMemsetAsync(c->err_dev, 0, sizeof(int), c->stream);
    if (code == 3)
      return fail("Singular matrix in Gauss-Jordan elimination!");   // matrix.cpp:81
    if (code == 2) {
      c->pipe_nblocks = 0;   // (the progress words of an abandoned launch: set up afresh)
      return fail("LU-SGS pipeline: a k-plane waited beyond the spin limit for its "
                  "predecessor (AGX_LUSGS=plane / AGX_SWEEP_PIPE=0 select the "
                  "launch-per-hyperplane forms)");
    }
    return fail("a boundary-condition variant outside this build's coverage was requested");
  }
  return 0;
}

struct MarchPlan { dim3 grid; int kchunk; long nparts; };
int g_march_tj = 6;   // cell rows per workgroup (512 threads)
// the tile kernel addresses a plane with 32-bit byte offsets (SlabDev::ldb)
bool tile_ok(const agx_ctx* c, const BlockDev& b) {
  return AGX_FAST && c->use_tile && !c->use_gather && (double)b.nplane * 8.0 < 4294967296.0;
}
bool all_tile_ok(const agx_ctx* c) {
  for (const auto& blk : c->blocks)
    if (!tile_ok(c, blk.d)) return false;
  return true;
}
MarchPlan march_plan(const agx_ctx* c, const BlockDev& b) {
  MarchPlan p;
  const int gx = (b.ni + 63) / 64, gy = (b.nj + g_march_tj - 1) / g_march_tj;
  if (tile_ok(c, b)) {
    // persistent workgroups, one per CU (LDS allows no more); small blocks get
    // fewer so that a range is at least ~8 steps long
    const long steps = (long)gx * gy * b.nk;
    long np = std::min<long>(c->num_cu, std::max<long>(1, steps / 8));
    if (np >= 8) np -= np % 8;
    p.kchunk = 0;
    p.grid = dim3((unsigned)np);
    p.nparts = np;
    return p;
  }
  // aim at >= ~2048 workgroups so that all 256 CUs stay busy to the end
  int nz = std::max(1, (int)std::lround(2048.0 / (gx * gy)));
  nz = std::min(nz, std::max(1, b.nk / 4));
  p.kchunk = (b.nk + nz - 1) / nz;
  nz = (b.nk + p.kchunk - 1) / p.kchunk;
  p.grid = dim3(gx, gy, nz);
  p.nparts = (long)gx * gy * nz;
  return p;
}

SlabDev make_slab(const BlockDev& b) {
  SlabDev sd;
  sd.base = b.vol - (long)PL_VOL * b.nplane;
  sd.nplane = b.nplane; sd.sx = b.sx; sd.sxy = b.sxy;
  sd.ni = b.ni; sd.nj = b.nj; sd.nk = b.nk; sd.ng = b.ng; sd.ioff = b.ioff;
  sd.st = (int)((b.state[0] - sd.base) / b.nplane);
  sd.sn = (int)((b.state2[0] - sd.base) / b.nplane);
  return sd;
}
template <int RECON, int LIM, int FLUX>
void launch_inv_kernel(agx_ctx* c, const BlockDev& b, double cfl, bool fuse,
                       const MarchArgs& ma, const MarchPlan& mp) {
  const SlabDev sd = make_slab(b);
  if (c->use_gather || AGX_NEQ != 5) {   // (the 7-equation build runs the gather form)
    hipLaunchKernelGGL((k_inv_residual<RECON, LIM, FLUX>), cell_grid(b, CELL_BLOCK),
                       CELL_BLOCK, 0, c->stream, b, c->gas, c->sp, cfl);
    return;
  }
  const dim3 tb(64, g_march_tj + 2);
#if AGX_FAST
  if (tile_ok(c, b)) {
    if (fuse && ma.store_consn)
      hipLaunchKernelGGL((k_residual_tile<RECON, LIM, FLUX, 2, 6>), mp.grid, tb, 0,
                         c->stream, sd, c->gas, c->sp, cfl, ma);
    else if (fuse)
      hipLaunchKernelGGL((k_residual_tile<RECON, LIM, FLUX, 1, 6>), mp.grid, tb, 0,
                         c->stream, sd, c->gas, c->sp, cfl, ma);
    else
      hipLaunchKernelGGL((k_residual_tile<RECON, LIM, FLUX, 0, 6>), mp.grid, tb, 0,
                         c->stream, sd, c->gas, c->sp, cfl, ma);
  } else
#endif
  {
    if (fuse)
      hipLaunchKernelGGL((k_residual_march<RECON, LIM, FLUX, true, 6>), mp.grid, tb, 0,
                         c->stream, sd, c->gas, c->sp, cfl, ma);
    else
      hipLaunchKernelGGL((k_residual_march<RECON, LIM, FLUX, false, 6>), mp.grid, tb, 0,
                         c->stream, sd, c->gas, c->sp, cfl, ma);
  }
}
template <int RECON, int LIM>
void launch_inv_flux(agx_ctx* c, const BlockDev& b, double cfl, bool fuse,
                     const MarchArgs& ma, const MarchPlan& mp) {
  if (c->cfg.inviscid_flux == AGX_FLUX_ROE)
    launch_inv_kernel<RECON, LIM, AGX_FLUX_ROE>(c, b, cfl, fuse, ma, mp);
  else
    launch_inv_kernel<RECON, LIM, AGX_FLUX_AUSM>(c, b, cfl, fuse, ma, mp);
}
template <int RECON>
void launch_inv_lim(agx_ctx* c, const BlockDev& b, double cfl, bool fuse,
                    const MarchArgs& ma, const MarchPlan& mp) {
  switch (c->cfg.limiter) {
    case AGX_LIMITER_VANALBADA: launch_inv_flux<RECON, AGX_LIMITER_VANALBADA>(c, b, cfl, fuse, ma, mp); break;
    case AGX_LIMITER_MINMOD: launch_inv_flux<RECON, AGX_LIMITER_MINMOD>(c, b, cfl, fuse, ma, mp); break;
    default: launch_inv_flux<RECON, AGX_LIMITER_NONE>(c, b, cfl, fuse, ma, mp); break;
  }
}
void launch_inv(agx_ctx* c, const BlockDev& b, double cfl, bool fuse,
                const MarchArgs& ma, const MarchPlan& mp) {
  switch (c->cfg.recon) {
    case AGX_RECON_CONSTANT: launch_inv_flux<AGX_RECON_CONSTANT, AGX_LIMITER_NONE>(c, b, cfl, fuse, ma, mp); break;
    case AGX_RECON_MUSCL: launch_inv_lim<AGX_RECON_MUSCL>(c, b, cfl, fuse, ma, mp); break;
    case AGX_RECON_WENO: launch_inv_flux<AGX_RECON_WENO, AGX_LIMITER_NONE>(c, b, cfl, fuse, ma, mp); break;
    default: launch_inv_flux<AGX_RECON_WENOZ, AGX_LIMITER_NONE>(c, b, cfl, fuse, ma, mp); break;
  }
}

// block-matrix solvers: the inviscid part of the main diagonal (k_block_diag_inv)
template <int RECON>
void launch_block_diag_lim(agx_ctx* c, const BlockDev& b) {
  const dim3 grid = cell_grid(b, CELL_BLOCK);
  switch (RECON == AGX_RECON_MUSCL ? c->cfg.limiter : AGX_LIMITER_NONE) {
    case AGX_LIMITER_VANALBADA:
      hipLaunchKernelGGL((k_block_diag_inv<RECON, AGX_LIMITER_VANALBADA>), grid, CELL_BLOCK, 0,
                         c->stream, b, c->gas, c->sp);
      break;
    case AGX_LIMITER_MINMOD:
      hipLaunchKernelGGL((k_block_diag_inv<RECON, AGX_LIMITER_MINMOD>), grid, CELL_BLOCK, 0,
                         c->stream, b, c->gas, c->sp);
      break;
    default:
      hipLaunchKernelGGL((k_block_diag_inv<RECON, AGX_LIMITER_NONE>), grid, CELL_BLOCK, 0,
                         c->stream, b, c->gas, c->sp);
      break;
  }
}
void launch_block_diag(agx_ctx* c, const BlockDev& b) {
  switch (c->cfg.recon) {
    case AGX_RECON_CONSTANT: launch_block_diag_lim<AGX_RECON_CONSTANT>(c, b); break;
    case AGX_RECON_MUSCL: launch_block_diag_lim<AGX_RECON_MUSCL>(c, b); break;
    case AGX_RECON_WENO: launch_block_diag_lim<AGX_RECON_WENO>(c, b); break;
    default: launch_block_diag_lim<AGX_RECON_WENOZ>(c, b); break;
  }
}

bool can_fuse(const agx_ctx* c) {
  // (nonreflecting surfaces read the gradients of the residual's own state after it)
  for (const auto& blk : c->blocks) if (blk.nr_max > 0) return false;
  return c->allow_fuse && !c->use_gather && !c->sp.implicit && !c->sp.viscous;
}

// the D2 LU-SGS path (agx_lusgs.hpp) serves scalar LU-SGS unless AGX_LUSGS=plane
bool is_block_solver(const agx_ctx* c) {   // input::IsBlockMatrix input.cpp:713
  return c->cfg.matrix_solver == AGX_SOLVER_BLUSGS || c->cfg.matrix_solver == AGX_SOLVER_BDPLUR;
}
bool is_lusgs_solver(const agx_ctx* c) {   // input.cpp:847
  return c->cfg.matrix_solver == AGX_SOLVER_LUSGS || c->cfg.matrix_solver == AGX_SOLVER_BLUSGS;
}
bool use_d2(const agx_ctx* c) {
  // (the Roe off-diagonal needs the state on both sides of a face: served by the
  // hyperplane-per-launch form on the SoA planes)
  return AGX_FAST && c->sp.implicit && c->cfg.matrix_solver == AGX_SOLVER_LUSGS && c->lusgs_mode == 1 &&
         c->cfg.inv_flux_jacobian == AGX_JACOBIAN_RUSANOV;
}
#if AGX_FAST
// One LU-SGS half sweep over a block, one launch: a workgroup per k-plane marches
// the plane's diagonals, the planes follow each other one step apart (k_lusgs_kp).
template <bool FWD, bool FULL, bool CONN, int CH>
int lusgs_kp_launch(agx_ctx* c, Block& blk) {
  const BlockDev& b = blk.d;
  if (!blk.kp_mem) {                      // the ticket counter
    HIPCHK(hipMalloc((void**)&blk.kp_mem, sizeof(int) * 32));
    HIPCHK(hipMemsetAsync(blk.kp_mem, 0, sizeof(int) * 32, c->stream));
  }
  KpArgs kp;
  kp.ticket = blk.kp_mem;
  kp.err = c->err_dev;
  // the launch's tag of the values it stores (agx_lusgs_kernels.hpp: KpArgs); every
  // writer of x advances the block's epoch
  kp.tag = (unsigned)(++blk.kp_epoch) & 3u;
  kp.spin_limit = c->spin_limit;
  kp.trace = nullptr;
#ifdef AGX_KP_TRACE
  static long long* trace_dev = nullptr;
  const size_t trace_n = 6 * (size_t)(b.ni + b.nj + 2);
  if (getenv("AGX_KP_TRACE")) {
    if (!trace_dev) hipMalloc((void**)&trace_dev, sizeof(long long) * 6 * 8192);
    hipMemsetAsync(trace_dev, 0, sizeof(long long) * trace_n, c->stream);
    kp.trace = trace_dev;
  }
#endif
  HIPCHK(hipMemsetAsync(kp.ticket, 0, sizeof(int), c->stream));
  // LDS: two buffers of 14-double records for the cells of a diagonal (+ 2 slots)
  int nsl = std::min(b.ni, b.nj) + 2;
  nsl += nsl & 1;
  const size_t lds = sizeof(double) * 2 * KP_NV * (size_t)nsl;
  const void* fn = reinterpret_cast<const void*>(&k_lusgs_kp<FWD, FULL, CONN, CH>);
  if (lds > 48 * 1024)
    HIPCHK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  // persistent: never more workgroups than are resident at once (a plane waits
  // for its predecessor, which must therefore be running or finished)
  int per_cu = 1;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, 256, lds) != hipSuccess ||
      per_cu < 1)
    per_cu = 1;
  const int wgs = std::min(per_cu * c->num_cu, b.nk);
  KpBlk kb;
  kb.base = b.d2.base; kb.dstart = b.d2.dstart; kb.surf = b.surf;
  kb.nd2 = b.d2.nd2; kb.ps = b.d2.ps; kb.Pi = b.d2.Pi; kb.Pj = b.d2.Pj;
  kb.ni = b.ni; kb.nj = b.nj; kb.nk = b.nk; kb.ng = b.ng;
  kb.nsurf = b.nsurf; kb.nsurf_i = b.nsurf_i; kb.nsurf_j = b.nsurf_j; kb.nsurf_k = b.nsurf_k;
  for (int q = 0; q < 6; ++q) kb.side_conn[q] = b.side_conn[q];
  const KpGas kg{c->gas.hf, c->gas.n, c->gas.inv_n};
  hipLaunchKernelGGL((k_lusgs_kp<FWD, FULL, CONN, CH>), dim3(wgs), dim3(256), lds, c->stream,
                     kb, kg, c->sp.viscous, nsl, kp);
#ifdef AGX_KP_TRACE
  if (kp.trace) {   // diagnostic build only: dump the step timestamps of this launch
    std::vector<long long> h(trace_n);
    hipStreamSynchronize(c->stream);
    hipMemcpy(h.data(), kp.trace, sizeof(long long) * trace_n, hipMemcpyDeviceToHost);
    FILE* f = fopen(getenv("AGX_KP_TRACE"), "a");
    if (f) {
      fprintf(f, "# %s ni %d nj %d nk %d wgs %d\n", FWD ? "fwd" : "bwd", b.ni, b.nj, b.nk, wgs);
      for (size_t t = 0; t + 6 <= trace_n; t += 6)
        fprintf(f, "%lld %lld %lld %lld %lld %lld\n", h[t], h[t + 1], h[t + 2], h[t + 3],
                h[t + 4], h[t + 5]);
      fclose(f);
    }
  }
#endif
  return 0;
}
// diagonals longer than 256 cells are walked in CH chunks per step
constexpr int KP_MAX_DIAG = 512;   // LDS: 2 * 19 * 8 * (n + 2) bytes <= 160 KiB
template <bool FWD, bool FULL, bool CONN>
int lusgs_kp_chunks(agx_ctx* c, Block& blk) {
  const int n = std::min(blk.d.ni, blk.d.nj);
  if (n <= 256) return lusgs_kp_launch<FWD, FULL, CONN, 1>(c, blk);
  return lusgs_kp_launch<FWD, FULL, CONN, 2>(c, blk);
}
template <bool FWD>
int lusgs_kp_variant(agx_ctx* c, Block& blk, int full) {
  bool conn = false;
  for (int q = 0; q < 6; ++q) conn = conn || blk.d.side_conn[q] != 0;
  if (full) return conn ? lusgs_kp_chunks<FWD, true, true>(c, blk)
                        : lusgs_kp_chunks<FWD, true, false>(c, blk);
  return conn ? lusgs_kp_chunks<FWD, false, true>(c, blk)
              : lusgs_kp_chunks<FWD, false, false>(c, blk);
}

#else
constexpr int KP_MAX_DIAG = 1 << 30;
#endif

static void launch_plane_sweep(agx_ctx* c, const BlockDev& b, bool forward, int full,
                               hipStream_t st) {
  // with the cell-major records: three lanes per cell (k_lusgs_plane3), 21 cells per wave row
  const bool three = b.sw_geo != nullptr && c->sweep_three;
  const dim3 tb(64, 4), grid(three ? (b.nj + PL3_CELLS - 1) / PL3_CELLS : (b.nj + 63) / 64,
                             (b.nk + 3) / 4);
  const int nplanes = b.ni + b.nj + b.nk - 2;
  for (int t = 0; t < nplanes; ++t) {
    const int p = forward ? t : nplanes - 1 - t;
    if (three) {
      if (forward)
        hipLaunchKernelGGL((k_lusgs_plane3<true>), grid, tb, 0, st, b, c->gas, c->sp, p, full);
      else
        hipLaunchKernelGGL((k_lusgs_plane3<false>), grid, tb, 0, st, b, c->gas, c->sp, p, full);
    } else if (forward) {
      hipLaunchKernelGGL((k_lusgs_plane<true>), grid, tb, 0, st, b, c->gas, c->sp, p, full);
    } else {
      hipLaunchKernelGGL((k_lusgs_plane<false>), grid, tb, 0, st, b, c->gas, c->sp, p, full);
    }
  }
}

// every block of this rank side by side (see branch_streams); false: not applicable
// (one block, a diagonal-ordered block, graphs switched off)
static bool plane_sweep_all_applicable(const agx_ctx* c) {
  if (!c->use_graphs || c->blocks.size() < 2) return false;
  for (auto& blk : c->blocks)
    if (blk.d.d2.base || blk.d.ni + blk.d.nj + blk.d.nk - 2 < 8) return false;
  return true;
}
int lusgs_sweep(agx_ctx* c, Block& blk, bool forward, int full, hipStream_t st = nullptr);
static void drop_sweep_graphs_all(agx_ctx* c) {
  for (auto& g1 : c->sweep_graph_all) for (auto& g2 : g1) for (auto& g3 : g2)
    if (g3) { hipGraphExecDestroy(g3); g3 = nullptr; }
}
// the device copy of the blocks' descriptors, for kernels that serve all blocks in one launch
static int sync_blocks_tab(agx_ctx* c) {
  const size_t nb = c->blocks.size();
  if (c->blocks_tab && c->blocks_tab_n != nb) {      // (blocks were added since)
    HIPCHK(hipStreamSynchronize(c->stream));
    HIPCHK(hipFree(c->blocks_tab));
    HIPCHK(hipHostFree(c->blocks_tab_host));
    c->blocks_tab = nullptr; c->blocks_tab_host = nullptr;
    drop_sweep_graphs_all(c);                        // (their grids cover nb blocks)
  }
  if (!c->blocks_tab) {
    HIPCHK(hipMalloc((void**)&c->blocks_tab, sizeof(BlockDev) * nb));
    HIPCHK(hipHostMalloc((void**)&c->blocks_tab_host, sizeof(BlockDev) * nb));
    memset(c->blocks_tab_host, 0, sizeof(BlockDev) * nb);
    c->blocks_tab_n = nb;
  }
  // the table follows the blocks (pointers that change roles, new surfaces): if it
  // differs from what the device holds, upload it
  bool same = true;
  for (size_t n = 0; n < nb; ++n)
    same = same && memcmp(&c->blocks_tab_host[n], &c->blocks[n].d, sizeof(BlockDev)) == 0;
  if (!same) {
    HIPCHK(hipStreamSynchronize(c->stream));      // (nobody reads the pinned copy any more)
    for (size_t n = 0; n < nb; ++n) memcpy(&c->blocks_tab_host[n], &c->blocks[n].d, sizeof(BlockDev));
    HIPCHK(hipMemcpyAsync(c->blocks_tab, c->blocks_tab_host, sizeof(BlockDev) * nb,
                          hipMemcpyHostToDevice, c->stream));
  }
  return 0;
}
// Pipelined half sweep of all blocks (k_lusgs_pipe): applicable to blocks on the cell-major
// records (the 7-equation and block-matrix builds; the 5-equation scalar solver has its own
// diagonal-ordered pipeline, k_lusgs_kp)
static bool pipe_sweep_applicable(const agx_ctx* c) {
  if (!c->sweep_pipe || c->blocks.empty()) return false;
  for (auto& blk : c->blocks)
    if (blk.d.d2.base || !blk.d.sw_geo) return false;
  return true;
}
static void pipe_free(agx_ctx* c) {
  for (auto& p : c->pipe_jobs) { if (p) hipFree(p); p = nullptr; }
  if (c->pipe_slot0) hipFree(c->pipe_slot0);
  if (c->pipe_progress) hipFree(c->pipe_progress);
  if (c->pipe_ticket) hipFree(c->pipe_ticket);
  c->pipe_slot0 = nullptr; c->pipe_progress = nullptr; c->pipe_ticket = nullptr;
  c->pipe_nblocks = 0; c->pipe_njobs = 0; c->pipe_launches = 0;
}
static int pipe_setup(agx_ctx* c) {
  const size_t nb = c->blocks.size();
  HIPCHK(hipStreamSynchronize(c->stream));
  pipe_free(c);
  // pipeline order: plane k of every block before plane k + 1 of any (going back: from the top)
  std::vector<PipeJob> fwd, bwd;
  std::vector<int> slot0(nb);
  int nk_max = 0, slots = 0, diag = 1;
  c->pipe_maxsteps = 0;
  for (size_t n = 0; n < nb; ++n) {
    const BlockDev& b = c->blocks[n].d;
    slot0[n] = slots;
    slots += b.nk;
    nk_max = std::max(nk_max, b.nk);
    c->pipe_maxsteps = std::max(c->pipe_maxsteps, b.ni + b.nj - 1);
    diag = std::max(diag, std::min(b.ni, b.nj));
  }
  for (int k = 0; k < nk_max; ++k)
    for (size_t n = 0; n < nb; ++n) {
      const int nk = c->blocks[n].d.nk;
      if (k < nk) { fwd.push_back({(int)n, k}); bwd.push_back({(int)n, nk - 1 - k}); }
    }
  c->pipe_njobs = (int)fwd.size();
  c->pipe_waves = std::max(1, std::min(8, (diag + PL3_CELLS - 1) / PL3_CELLS));
  for (int dir = 0; dir < 2; ++dir) {
    HIPCHK(hipMalloc((void**)&c->pipe_jobs[dir], sizeof(PipeJob) * fwd.size()));
    HIPCHK(hipMemcpy(c->pipe_jobs[dir], (dir ? fwd : bwd).data(), sizeof(PipeJob) * fwd.size(),
                     hipMemcpyHostToDevice));
  }
  HIPCHK(hipMalloc((void**)&c->pipe_slot0, sizeof(int) * nb));
  HIPCHK(hipMemcpy(c->pipe_slot0, slot0.data(), sizeof(int) * nb, hipMemcpyHostToDevice));
  HIPCHK(hipMalloc((void**)&c->pipe_progress, sizeof(long long) * 16 * slots));
  HIPCHK(hipMemset(c->pipe_progress, 0, sizeof(long long) * 16 * slots));
  HIPCHK(hipMalloc((void**)&c->pipe_ticket, 128));
  HIPCHK(hipMemset(c->pipe_ticket, 0, 128));
  c->pipe_nblocks = nb;
  return 0;
}
static int lusgs_sweep_pipe(agx_ctx* c, bool forward, int full) {
  const size_t nb = c->blocks.size();
  bool same = c->pipe_nblocks == nb;
  if (same) {
    int slots = 0, steps = 0;
    for (auto& blk : c->blocks) { slots += blk.d.nk; steps = std::max(steps, blk.d.ni + blk.d.nj - 1); }
    same = slots == c->pipe_njobs && steps == c->pipe_maxsteps;
  }
  if (!same && pipe_setup(c)) return 1;
  if (sync_blocks_tab(c)) return 1;
  PipeArgs pa;
  pa.jobs = c->pipe_jobs[forward ? 1 : 0];
  pa.slot0 = c->pipe_slot0;
  pa.progress = c->pipe_progress;
  pa.ticket = c->pipe_ticket;
  pa.base = c->pipe_launches * (long long)(c->pipe_maxsteps + 1);
  pa.tbase = (unsigned long long)c->pipe_launches * (unsigned long long)c->pipe_njobs;
  pa.njobs = c->pipe_njobs;
  pa.spin_limit = c->spin_limit;
  pa.err = c->err_dev;
  pa.trace = nullptr;
#ifdef AGX_PIPE_TRACE
  static long long* trace_dev = nullptr;
  const size_t trace_n = 5 * (size_t)(c->pipe_maxsteps + 1);
  if (getenv("AGX_PIPE_TRACE")) {
    if (!trace_dev) hipMalloc((void**)&trace_dev, sizeof(long long) * 5 * 65536);
    hipMemsetAsync(trace_dev, 0, sizeof(long long) * trace_n, c->stream);
    pa.trace = trace_dev;
  }
#endif
  ++c->pipe_launches;
  const dim3 grid((unsigned)c->pipe_njobs);
  const dim3 tb(64, c->pipe_waves);
  if (forward)
    hipLaunchKernelGGL((k_lusgs_pipe<true>), grid, tb, 0, c->stream, c->blocks_tab, c->gas, c->sp, full, pa);
  else
    hipLaunchKernelGGL((k_lusgs_pipe<false>), grid, tb, 0, c->stream, c->blocks_tab, c->gas, c->sp, full, pa);
  HIPCHK(hipGetLastError());
#ifdef AGX_PIPE_TRACE
  if (pa.trace) {   // diagnostic build only: dump the step timestamps of this launch
    std::vector<long long> h(trace_n);
    hipStreamSynchronize(c->stream);
    hipMemcpy(h.data(), pa.trace, sizeof(long long) * trace_n, hipMemcpyDeviceToHost);
    if (FILE* f = fopen(getenv("AGX_PIPE_TRACE"), "a")) {
      fprintf(f, "# %s jobs %d\n", forward ? "fwd" : "bwd", c->pipe_njobs);
      for (size_t t = 0; t + 5 <= trace_n; t += 5)
        fprintf(f, "%lld %lld %lld %lld %lld\n", h[t], h[t + 1], h[t + 2], h[t + 3], h[t + 4]);
      fclose(f);
    }
  }
#endif
  return 0;
}
static int lusgs_sweep_all_one_launch(agx_ctx* c, bool forward, int full) {
  const size_t nb = c->blocks.size();
  if (sync_blocks_tab(c)) return 1;
  int steps = 0;
  unsigned gx = 1, gy = 1;
  bool three = c->sweep_three;
  for (auto& blk : c->blocks) three = three && blk.d.sw_geo != nullptr;
  for (auto& blk : c->blocks) {
    steps = std::max(steps, blk.d.ni + blk.d.nj + blk.d.nk - 2);
    gx = std::max(gx, three ? (unsigned)(blk.d.nj + PL3_CELLS - 1) / PL3_CELLS
                            : (unsigned)(blk.d.nj + 63) / 64);
    gy = std::max(gy, (unsigned)(blk.d.nk + 3) / 4);
  }
  const dim3 tb(64, 4), grid(gx, gy, (unsigned)nb);
  auto launch_all = [&](hipStream_t st) {
    for (int t = 0; t < steps; ++t) {
      if (three) {
        if (forward)
          hipLaunchKernelGGL((k_lusgs_plane_all3<true>), grid, tb, 0, st, c->blocks_tab, c->gas, c->sp, t, full);
        else
          hipLaunchKernelGGL((k_lusgs_plane_all3<false>), grid, tb, 0, st, c->blocks_tab, c->gas, c->sp, t, full);
        continue;
      }
      if (forward)
        hipLaunchKernelGGL((k_lusgs_plane_all<true>), grid, tb, 0, st, c->blocks_tab, c->gas, c->sp, t, full);
      else
        hipLaunchKernelGGL((k_lusgs_plane_all<false>), grid, tb, 0, st, c->blocks_tab, c->gas, c->sp, t, full);
    }
  };
  hipGraphExec_t& ge = c->sweep_graph_all[forward ? 1 : 0][full ? 1 : 0][c->sp.un_is_u ? 1 : 0];
  if (!ge) {
    if (!c->cap_stream) HIPCHK(hipStreamCreateWithFlags(&c->cap_stream, hipStreamNonBlocking));
    hipGraph_t graph = nullptr;
    HIPCHK(hipStreamBeginCapture(c->cap_stream, hipStreamCaptureModeThreadLocal));
    launch_all(c->cap_stream);
    HIPCHK(hipStreamEndCapture(c->cap_stream, &graph));
    HIPCHK(hipGraphInstantiate(&ge, graph, nullptr, nullptr, 0));
    HIPCHK(hipGraphDestroy(graph));
  }
  HIPCHK(hipGraphLaunch(ge, c->stream));
  return 0;
}
static int lusgs_sweep_all(agx_ctx* c, bool forward, int full) {
  if (c->sweep_all_launch) return lusgs_sweep_all_one_launch(c, forward, full);
  const size_t ns = std::min<size_t>(c->blocks.size(), 8);
  while (c->branch_streams.size() < ns) {
    hipStream_t st;
    HIPCHK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    c->branch_streams.push_back(st);
  }
  while (c->branch_events.size() < ns + 1) {
    hipEvent_t ev;
    HIPCHK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    c->branch_events.push_back(ev);
  }
  // fork from the library's stream, one chain of hyperplane launches (the block's own
  // graph) per branch stream, join
  HIPCHK(hipEventRecord(c->branch_events[ns], c->stream));
  for (size_t q = 0; q < ns; ++q)
    HIPCHK(hipStreamWaitEvent(c->branch_streams[q], c->branch_events[ns], 0));
  for (size_t n = 0; n < c->blocks.size(); ++n)
    if (lusgs_sweep(c, c->blocks[n], forward, full, c->branch_streams[n % ns])) return 1;
  for (size_t q = 0; q < ns; ++q) {
    HIPCHK(hipEventRecord(c->branch_events[q], c->branch_streams[q]));
    HIPCHK(hipStreamWaitEvent(c->stream, c->branch_events[q], 0));
  }
  return 0;
}

int lusgs_sweep(agx_ctx* c, Block& blk, bool forward, int full, hipStream_t st) {
  const BlockDev& b = blk.d;
  if (!st) st = c->stream;
  if (!b.d2.base) {
    // One launch per hyperplane i + j + k = p: ni + nj + nk - 2 short launches per half
    // sweep, bound by launch latency.  The sequence is the same every iteration, so it is
    // captured once into a hipGraph (per direction / triangle set / value of un_is_u, the
    // one solver parameter that changes between iterations) and replayed.
    const int nplanes = b.ni + b.nj + b.nk - 2;
    auto launch_all = [&](hipStream_t st) { launch_plane_sweep(c, b, forward, full, st); };
    if (!c->use_graphs || nplanes < 8) {
      launch_all(st);
      return 0;
    }
    hipGraphExec_t& ge = blk.sweep_graph[forward ? 1 : 0][full ? 1 : 0][c->sp.un_is_u ? 1 : 0];
    if (!ge) {
      // (recorded on a stream of its own: the library's stream may be the legacy default
      // stream, which cannot capture; a graph is launched on any stream)
      if (!c->cap_stream) HIPCHK(hipStreamCreateWithFlags(&c->cap_stream, hipStreamNonBlocking));
      hipGraph_t graph = nullptr;
      HIPCHK(hipStreamBeginCapture(c->cap_stream, hipStreamCaptureModeThreadLocal));
      launch_all(c->cap_stream);
      HIPCHK(hipStreamEndCapture(c->cap_stream, &graph));
      HIPCHK(hipGraphInstantiate(&ge, graph, nullptr, nullptr, 0));
      HIPCHK(hipGraphDestroy(graph));
    }
    HIPCHK(hipGraphLaunch(ge, st));
    return 0;
  }
#if AGX_FAST
  return forward ? lusgs_kp_variant<true>(c, blk, full) : lusgs_kp_variant<false>(c, blk, full);
#else
  return fail("no diagonal-ordered sweep in this build");
#endif
}

// consVarsN = cons(state) that agx_store_time_n deferred (see there)
int flush_consn(agx_ctx* c) {
  if (!c->consn_pending) return 0;
  c->consn_pending = false;
  for (auto& blk : c->blocks)
    hipLaunchKernelGGL(k_store_time_n, cell_grid(blk.d, CELL_BLOCK), CELL_BLOCK,
                       0, c->stream, blk.d, c->gas, 0);
  HIPCHK(hipGetLastError());
  return 0;
}


int bc_pass(agx_ctx* c, bool faces, int viscous) {
  for (auto& blk : c->blocks) {
    const BlockDev& b = blk.d;
    if (faces) {
      long nmax = 0;
      for (int sn = 0; sn < b.nsurf; ++sn) {
        const agx_bc_surface& s = blk.surf_host[sn];
        if (s.bc_type == AGX_BC_INTERBLOCK || s.bc_type == AGX_BC_PERIODIC) continue;
        if (viscous && s.bc_type != AGX_BC_VISCOUSWALL) continue;
        const int st = surface_type(s);
        const int d3 = (st - 1) / 2, d1 = (d3 + 1) % 3, d2 = (d3 + 2) % 3;
        const int lo[3] = {s.imin, s.jmin, s.kmin}, hi[3] = {s.imax, s.jmax, s.kmax};
        nmax = std::max(nmax, (long)(hi[d1] - lo[d1]) * (hi[d2] - lo[d2]));
      }
      if (!viscous && blk.nr_max > 0) {
        if (!c->have_time_n)
          return fail("nonreflecting boundary: the state at time n has not been stored");
        if (flush_consn(c)) return 1;   // (its ghost states read consVarsN)
        hipLaunchKernelGGL(k_nr_mach, dim3(b.nsurf), dim3(256), 0, c->stream, b, c->gas);
      }
      if (nmax > 0)
        hipLaunchKernelGGL(k_bc_faces, dim3((nmax + 255) / 256, b.nsurf), dim3(256), 0,
                           c->stream, b, c->gas, viscous, c->err_dev);
    } else {
      const long n = 4L * (b.ni + b.nj + b.nk);
      hipLaunchKernelGGL(k_bc_edges, dim3((n + 127) / 128), dim3(128), 0,
                         c->stream, b, c->gas, viscous, c->err_dev);
    }
  }
  HIPCHK(hipGetLastError());
  return 0;
}

int reduce_norms(agx_ctx* c, size_t blk_index, long nparts, NormPartial* out = nullptr) {
  if (!out) out = c->norm_out + blk_index;
  if (nparts > 4096) {
    // two levels: 64 workgroups fold slices into the scratch behind norm_out
    NormPartial* tmp = c->norm_out + c->blocks.size();
    hipLaunchKernelGGL(k_norm_final, dim3(64), dim3(256), 0, c->stream, c->partials, nparts, tmp);
    hipLaunchKernelGGL(k_norm_final, dim3(1), dim3(256), 0, c->stream, tmp, 64L, out);
  } else {
    hipLaunchKernelGGL(k_norm_final, dim3(1), dim3(256), 0, c->stream,
                       c->partials, nparts, out);
  }
  HIPCHK(hipGetLastError());
  return 0;
}

// gridLevel::GetBoundaryConditions gridLevel.cpp:287-319 for the blocks of this rank
int fill_ghosts(agx_ctx* c) {
  if (agx_phase_bc_faces(c)) return 1;
  if (agx_halo_exchange(c, AGX_HALO_STATE)) return 1;
  return agx_phase_bc_edges(c);
}
int update_pass(agx_ctx* c, int mode, int mm, double* l2, agx_linf* linf) {
  c->state_is_time_n = false;
  const double alpha[4] = {0.25, 1.0 / 3.0, 0.5, 1.0};   // procBlock.cpp:938
  if (mode != 2 && c->fused_pending) {
    // the marching kernel already advanced the state into the second buffer
    // and left one norm partial per workgroup: fold them and swap buffers
    Timer t(c, G_UPDATE);
    long off = 0;
    for (size_t n = 0; n < c->blocks.size(); ++n) {
      BlockDev& b = c->blocks[n].d;
      const MarchPlan mp = march_plan(c, b);
      hipLaunchKernelGGL(k_norm_final, dim3(1), dim3(256), 0, c->stream,
                         c->partials + off, mp.nparts, c->norm_out + n);
      off += mp.nparts;
      for (int e = 0; e < AGX_NEQ; ++e) std::swap(b.state[e], b.state2[e]);
    }
    c->halo_set[AGX_HALO_STATE] ^= 1;              // (the tables hold plane pointers)
    c->fused_pending = false;
    HIPCHK(hipGetLastError());
  } else {
    Timer t(c, G_UPDATE);
    for (size_t n = 0; n < c->blocks.size(); ++n) {
      const BlockDev& b = c->blocks[n].d;
      const int last = mm == c->cfg.nonlinear_iterations - 1;
#if AGX_FAST
      if (mode == 2 && b.d2.base) {
        const dim3 tg((b.ni + TT - 1) / TT, (b.nj + TT - 1) / TT, b.nk);
        hipLaunchKernelGGL(k_update_d2, tg, dim3(256), 0, c->stream, b, c->gas, c->sp, last,
                           c->partials);
        if (reduce_norms(c, n, (long)tg.x * tg.y * tg.z)) return 1;
        continue;
      }
#endif
      const dim3 grid = cell_grid(b, CELL_BLOCK);
      hipLaunchKernelGGL(k_update, grid, CELL_BLOCK, 0, c->stream, b, c->gas,
                         c->sp, mode, mode == 1 ? alpha[mm & 3] : 1.0, last,
                         c->partials);
      if (reduce_norms(c, n, (long)grid.x * grid.y * grid.z)) return 1;
    }
  }
  HIPCHK(hipMemcpyAsync(c->norm_host, c->norm_out,
                        sizeof(NormPartial) * c->blocks.size(),
                        hipMemcpyDeviceToHost, c->stream));
  if (c->mres_deferred)
    HIPCHK(hipMemcpyAsync(c->norm_host + c->blocks.size(), c->norm_out + c->blocks.size() + 64,
                          sizeof(NormPartial) * c->blocks.size(), hipMemcpyDeviceToHost,
                          c->stream));
  HIPCHK(hipMemcpyAsync(c->err_host, c->err_dev, sizeof(int),
                        hipMemcpyDeviceToHost, c->stream));
  if (c->in_iterate && c->eager_ghosts) {
    // the host waits for the norms only; the ghost fill of the next iteration is
    // already queued behind them
    if (!c->norm_event) HIPCHK(hipEventCreateWithFlags(&c->norm_event, hipEventDisableTiming));
    HIPCHK(hipEventRecord(c->norm_event, c->stream));
    if (fill_ghosts(c)) return 1;
    c->ghosts_prefilled = true;
    HIPCHK(hipEventSynchronize(c->norm_event));
  } else {
    HIPCHK(hipStreamSynchronize(c->stream));   // the one sync per iteration
  }
  if (check_device_error(c)) return 1;
  for (size_t n = 0; n < c->blocks.size(); ++n) {
    const NormPartial& p = c->norm_host[n];
    const BlockDev& b = c->blocks[n].d;
    for (int e = 0; e < AGX_NEQ; ++e) l2[e] += p.l2[e];
    if (p.vmax > linf->linf) {           // procBlock.cpp:863-866
      long cell = p.lin / AGX_NEQ;
      linf->linf = p.vmax;
      linf->eqn = (int)(p.lin % AGX_NEQ) + 1;
      linf->i = (int)(cell % b.ni);
      linf->j = (int)((cell / b.ni) % b.nj);
      linf->k = (int)(cell / ((long)b.ni * b.nj));
      linf->block = b.parent;
    }
  }
  return 0;
}

}  // namespace

// ===========================================================================
extern "C" {

const char* agx_last_error(void) { return g_err; }
const char* agx_version(void) { return "aither_gfx950 0.1 (HIP, gfx950)"; }

int agx_ctx_create(int device, int rank, agx_ctx** out) {
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev <= 0)
    return fail("no HIP device available (%s): the gfx950 library has no CPU "
                "path", hipGetErrorString(e));
  if (device < 0 || device >= ndev) return fail("device %d out of range", device);
  HIPCHK(hipSetDevice(device));
  agx_ctx* c = new agx_ctx();
  c->device = device;
  c->rank = rank;
  if (const char* kn = getenv("AGX_KERNEL")) {
    c->use_gather = !strcmp(kn, "gather");
    c->use_tile = !strcmp(kn, "tile");
  }
  c->allow_fuse = !(getenv("AGX_NO_FUSE") && atoi(getenv("AGX_NO_FUSE")) != 0);
  {
    int ncu = 0;
    HIPCHK(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, device));
    if (ncu > 0) c->num_cu = ncu;
    if (const char* w = getenv("AGX_WORKGROUPS")) c->num_cu = std::max(1, atoi(w));
    if (const char* w = getenv("AGX_VISC")) {
      c->visc_gather = !strcmp(w, "gather");
      c->visc_march = !strcmp(w, "march");
    }
    if (const char* w = getenv("AGX_EAGER_GHOSTS")) c->eager_ghosts = atoi(w) != 0;
    if (const char* w = getenv("AGX_OVERLAP")) c->overlap = atoi(w) != 0;
    if (const char* w = getenv("AGX_LUSGS")) c->lusgs_mode = !strcmp(w, "plane") ? 0 : 1;
    if (const char* w = getenv("AGX_SPIN_LIMIT")) c->spin_limit = std::max(1, atoi(w));
    if (const char* w = getenv("AGX_GRAPHS")) c->use_graphs = atoi(w) != 0;
    if (const char* w = getenv("AGX_SWEEP_ALL")) c->sweep_all_launch = atoi(w) != 0;
    if (const char* w = getenv("AGX_SWEEP_PIPE")) c->sweep_pipe = atoi(w) != 0;
    if (const char* w = getenv("AGX_SWEEP_THREE")) c->sweep_three = atoi(w) != 0;
    if (const char* w = getenv("AGX_SWEEP_RECORDS")) c->sweep_records = atoi(w) != 0;
    if (const char* w = getenv("AGX_MRESID_SPLIT")) c->mresid_split = std::min(64, std::max(1, atoi(w)));
    if (const char* w = getenv("AGX_MRESID")) c->mresid_march = strcmp(w, "plane") != 0;
    if (const char* w = getenv("AGX_HALO_BATCH")) {   // 0: a launch pair per connection; require:
      c->halo_batch = strcmp(w, "0") != 0;            // finalize fails where no batch forms
      c->halo_batch_required = !strcmp(w, "require");
    }
  }
  HIPCHK(hipMalloc((void**)&c->err_dev, sizeof(int)));
  HIPCHK(hipMemset(c->err_dev, 0, sizeof(int)));
  HIPCHK(hipHostMalloc((void**)&c->err_host, sizeof(int)));
  *c->err_host = 0;
  *out = c;
  return 0;
}

void agx_ctx_destroy(agx_ctx* c) {
  if (!c) return;
  hipSetDevice(c->device);
  hipStreamSynchronize(c->stream);
  for (auto& b : c->blocks) {
    if (b.slab) hipFree(b.slab);
    if (b.d2) hipFree(b.d2);
    if (b.blockmat) hipFree(b.blockmat);
    if (b.sweep_rec) hipFree(b.sweep_rec);
    for (void* p : {(void*)b.mg_forcing, (void*)b.mg_mres, (void*)b.mg_xsave, (void*)b.mg_nodes,
                    (void*)b.mg_tc, (void*)b.mg_start, (void*)b.mg_vf, (void*)b.mg_cf})
      if (p) hipFree(p);
    if (b.d2_tab) hipFree(b.d2_tab);
    if (b.kp_mem) hipFree(b.kp_mem);
    for (auto& g1 : b.sweep_graph) for (auto& g2 : g1) for (auto& g3 : g2)
      if (g3) hipGraphExecDestroy(g3);
    if (b.surf_dev) hipFree(b.surf_dev);
    if (b.nr_off_dev) hipFree(b.nr_off_dev);
    if (b.nr_mem) hipFree(b.nr_mem);
    if (b.wall_off_dev) hipFree(b.wall_off_dev);
    if (b.wall_mem) hipFree(b.wall_mem);
  }
  for (auto& k : c->conns) {
    for (int s = 0; s < 2; ++s) {
      if (k.side[s].dst) hipFree(k.side[s].dst);
      if (k.side[s].src) hipFree(k.side[s].src);
      if (k.side[s].dst2) hipFree(k.side[s].dst2);
      if (k.side[s].src2) hipFree(k.side[s].src2);
    }
    if (k.send_src) hipFree(k.send_src);
    if (k.send_src2) hipFree(k.send_src2);
  }
  if (c->partials) hipFree(c->partials);
  if (c->norm_out) hipFree(c->norm_out);
  if (c->norm_host) hipHostFree(c->norm_host);
  if (c->err_dev) hipFree(c->err_dev);
  if (c->err_host) hipHostFree(c->err_host);
  if (c->halo_buf) hipFree(c->halo_buf);
  if (c->stage_buf) hipFree(c->stage_buf);
  if (c->rans_rec) hipFree(c->rans_rec);
  pipe_free(c);
  for (auto& r : c->remote) {
    if (r.send) hipFree(r.send);
    if (r.recv) hipFree(r.recv);
    if (r.hsend) hipHostFree(r.hsend);
    if (r.hrecv) hipHostFree(r.hrecv);
  }
  if (c->rec_dev) hipFree(c->rec_dev);
  if (c->rec_host) hipHostFree(c->rec_host);
  if (c->nccl) ncclCommDestroy(c->nccl);
  for (auto& e : c->ev_pool) { hipEventDestroy(e.first); hipEventDestroy(e.second); }
  if (c->cap_stream) hipStreamDestroy(c->cap_stream);
  if (c->ov_stream) {
    hipStreamSynchronize(c->ov_stream);
    hipStreamDestroy(c->ov_stream);
    hipEventDestroy(c->ov_ready);
    hipEventDestroy(c->ov_done);
  }
  drop_sweep_graphs_all(c);
  if (c->blocks_tab) hipFree(c->blocks_tab);
  if (c->blocks_tab_host) hipHostFree(c->blocks_tab_host);
  if (c->halo_tab_host) hipHostFree(c->halo_tab_host);
  for (int w = 0; w < 5; ++w)
    for (int st = 0; st < 2; ++st)
      for (int q = 0; q < 2; ++q)
        if (c->halo_tab_dev[w][st][q]) hipFree(c->halo_tab_dev[w][st][q]);
  for (auto ev : c->branch_events) hipEventDestroy(ev);
  for (auto st : c->branch_streams) hipStreamDestroy(st);
  delete c;
}

int agx_ctx_set_stream(agx_ctx* c, void* s) {
  // at any time: what was queued on the old stream (set-up memsets, a previous iteration)
  // is finished before work is issued on the new one
  HIPCHK(hipSetDevice(c->device));
  HIPCHK(hipStreamSynchronize(c->stream));
  c->stream = (hipStream_t)s;
  return 0;
}

int agx_config_set(agx_ctx* c, const agx_config* cfg) {
  // allocations (block-matrix planes, sweep records, D2 arrays) and the captured sweep
  // graphs are decided from the configuration a block was created under
  if (!c->blocks.empty())
    return fail("agx_config_set: the configuration is fixed after the first "
                "agx_block_create (make a new context for another scheme)");
  if (cfg->n_eq != AGX_NEQ)
    return fail("n_eq = %d: this library is built for %d equations (5: euler / "
                "navierStokes, libaither_gfx950.so; 7: rans, libaither_gfx950_rans.so)",
                cfg->n_eq, AGX_NEQ);
  if (cfg->n_ghost < 1 || cfg->n_ghost > 3) return fail("n_ghost out of range");
  // Never substitute: a scheme this build does not implement is an error here,
  // not a different scheme silently (mgSolution.hpp:112-115 must mean the same).
#if AGX_NEQ == 7
  if (cfg->equation_set != AGX_EQN_RANS || !cfg->is_viscous)
    return fail("the 7-equation library serves equation_set rans (viscous) only");
  if (cfg->turbulence_model != AGX_TURB_SST2003 && cfg->turbulence_model != AGX_TURB_KW_WILCOX2006 &&
      cfg->turbulence_model != AGX_TURB_SST_DES)
    return fail("turbulence_model %d: sst2003, sstdes and kOmegaWilcox2006 are built",
                cfg->turbulence_model);
  if (cfg->inv_flux_jacobian == AGX_JACOBIAN_APPROX_ROE)
    return fail("rans: approximateRoe is not built");
#else
  if (cfg->equation_set != AGX_EQN_EULER && cfg->equation_set != AGX_EQN_NAVIER_STOKES)
    return fail("equation_set %d: this library covers euler and navierStokes "
                "(rans: libaither_gfx950_rans.so)", cfg->equation_set);
  if ((cfg->equation_set == AGX_EQN_NAVIER_STOKES) != (cfg->is_viscous != 0))
    return fail("equation_set %d contradicts is_viscous %d", cfg->equation_set, cfg->is_viscous);
  if (cfg->turbulence_model != AGX_TURB_NONE)
    return fail("turbulence_model %d is not built (laminar / inviscid only)",
                cfg->turbulence_model);
#endif
  if (cfg->inv_flux_jacobian != AGX_JACOBIAN_RUSANOV &&
      cfg->inv_flux_jacobian != AGX_JACOBIAN_APPROX_ROE)
    return fail("inv_flux_jacobian %d is not one of rusanov / approximateRoe",
                cfg->inv_flux_jacobian);
  if (cfg->inv_flux_jacobian == AGX_JACOBIAN_APPROX_ROE && cfg->is_viscous)
    return fail("approximateRoe with viscous terms is not built: the reference hands dist and "
                "f1 to RoeOffDiagonal in swapped order (fluxJacobian.cpp:232 vs :240) and "
                "divides by f1 = 0 in laminar runs");
  if (cfg->viscous_recon != AGX_VISC_RECON_CENTRAL &&
      cfg->viscous_recon != AGX_VISC_RECON_CENTRAL_4TH)
    return fail("viscous_recon %d is not one of central / centralFourth", cfg->viscous_recon);
  if (cfg->viscous_recon == AGX_VISC_RECON_CENTRAL_4TH && cfg->n_ghost < 2)
    return fail("centralFourth needs two ghost layers (input::NumberGhostLayers, "
                "input.cpp:1127-1143)");
  if (cfg->matrix_solver < AGX_SOLVER_LUSGS || cfg->matrix_solver > AGX_SOLVER_BDPLUR)
    return fail("matrix_solver %d is not one of lusgs / dplur / blusgs / bdplur",
                cfg->matrix_solver);
  if ((cfg->matrix_solver == AGX_SOLVER_BLUSGS || cfg->matrix_solver == AGX_SOLVER_BDPLUR) &&
      cfg->inv_flux_jacobian == AGX_JACOBIAN_APPROX_ROE)
    return fail("block-matrix solvers are built for inviscidFluxJacobian rusanov only");
  c->cfg = *cfg;
  derive_gas(*cfg, c->gas);
  c->gas.wilcox = cfg->turbulence_model == AGX_TURB_KW_WILCOX2006 ? 1 : 0;
  c->gas.sstdes = cfg->turbulence_model == AGX_TURB_SST_DES ? 1 : 0;
  c->gas.turb_prandtl = c->gas.wilcox ? 8.0 / 9.0 : 0.9;
  SolverDev& sp = c->sp;
  sp.diag_add = 0;
  sp.kappa = cfg->kappa;
  sp.theta = cfg->theta;
  sp.zeta = cfg->zeta;
  sp.relax = cfg->matrix_relaxation;
  sp.dual_time_cfl = cfg->dual_time_cfl;
  sp.dt_fixed = cfg->dt_nondim;
  sp.visc_cfl_coeff = cfg->viscous_cfl_coeff;
  sp.viscous = cfg->is_viscous;
  sp.implicit = cfg->time_integration >= AGX_TIME_IMPLICIT_EULER;
  sp.bdf2 = cfg->time_integration == AGX_TIME_BDF2;
  // input::MatrixRequiresInitialization input.cpp:1120-1125
  sp.requires_init = cfg->matrix_solver == AGX_SOLVER_DPLUR ||
                     cfg->matrix_solver == AGX_SOLVER_BDPLUR || cfg->matrix_sweeps > 1;
  sp.block = (cfg->matrix_solver == AGX_SOLVER_BLUSGS || cfg->matrix_solver == AGX_SOLVER_BDPLUR) ? 1 : 0;
  sp.time_integration = cfg->time_integration;
  sp.roe_jacobian = cfg->inv_flux_jacobian == AGX_JACOBIAN_APPROX_ROE;
  c->have_cfg = true;
  return 0;
}

int agx_block_create(agx_ctx* c, const agx_block_geom* g, int* block_id) {
  if (!c->have_cfg) return fail("agx_config_set must precede agx_block_create");
  if (g->ng != c->cfg.n_ghost) return fail("block ghost layers != config n_ghost");
  if (g->ni < 1 || g->nj < 1 || g->nk < 1) return fail("empty block");
  HIPCHK(hipSetDevice(c->device));
  c->blocks.emplace_back();
  Block& b = c->blocks.back();
  BlockDev& d = b.d;
  memset(&d, 0, sizeof d);
  d.ni = g->ni; d.nj = g->nj; d.nk = g->nk; d.ng = g->ng;
  d.parent = g->parent_block;
  b.global_pos = g->global_pos;
  d.ioff = 16;                               // physical i = 0 on a 128-B line
  const long row = d.ioff + d.ni + d.ng + 1; // +1: upper i-face
  d.sx = (row + 15) / 16 * 16;
  d.sxy = d.sx * (d.nj + 2 * d.ng + 1);
  d.nplane = d.sxy * (d.nk + 2 * d.ng + 1);
  // one slab for all planes of the block (see SlabDev in agx_kernels.hpp)
  HIPCHK(hipMalloc((void**)&b.slab, sizeof(double) * d.nplane * PL_COUNT));
  HIPCHK(hipMemsetAsync(b.slab, 0, sizeof(double) * d.nplane * PL_COUNT, c->stream));
  auto pl = [&](int id) { return b.slab + (long)id * d.nplane; };
  d.vol = pl(PL_VOL); d.specrad = pl(PL_SPECRAD); d.dt = pl(PL_DT);
  d.a = pl(PL_A); d.ainv = pl(PL_AINV); d.wdist = pl(PL_WDIST);
  for (int e = 0; e < AGX_NEQ; ++e) {
    d.state[e] = pl(PL_STATE_A + e); d.state2[e] = pl(PL_STATE_B + e);
    d.resid[e] = pl(PL_RESID + e); d.consn[e] = pl(PL_CONSN + e);
    d.consnm1[e] = pl(PL_CONSNM1 + e); d.x[e] = pl(PL_X + e);
    d.xold[e] = pl(PL_XOLD + e);
  }
  for (int q = 0; q < 3; ++q) {
    d.cen[q] = pl(PL_CEN + q);
    d.wid[q] = pl(PL_WID + q);
    for (int cc = 0; cc < 4; ++cc) d.fa[q][cc] = pl(PL_FA + 4 * q + cc);
  }
  b.state_is_a = true;
#if AGX_NEQ == 7
  d.specrad_t = pl(PL_SPECRAD_T); d.a_t = pl(PL_A_T); d.ainv_t = pl(PL_AINV_T);
  d.viscp = pl(PL_VISC);
  for (int q = 0; q < 3; ++q) d.turb3[q] = pl(PL_TURB3 + q);
#endif
  if (c->sp.implicit && is_block_solver(c)) {
    // a_, aInv_ (25 planes each) and velocityGrad_ (9) of the block-matrix solvers
    // (+ 2 x 2 planes: diagonal of the turbulence block in the rans build)
    const size_t n = (size_t)d.nplane * (2 * AGX_NJ + 9 + 4);
    HIPCHK(hipMalloc((void**)&b.blockmat, sizeof(double) * n));
    HIPCHK(hipMemsetAsync(b.blockmat, 0, sizeof(double) * n, c->stream));
    d.am = b.blockmat;
    d.aminv = b.blockmat + (size_t)d.nplane * AGX_NJ;
    d.vg = b.blockmat + (size_t)d.nplane * 2 * AGX_NJ;
    d.am_t = d.vg + (size_t)d.nplane * 9;
    d.aminv_t = d.am_t + (size_t)d.nplane * 2;
  }
  d.sw_geo = d.sw_dyn = d.sw_rhs = nullptr;
  d.mg_forcing = d.mg_mres = d.mg_xsave = nullptr;
  if (c->sp.implicit && is_lusgs_solver(c) && !use_d2(c) && c->sweep_records) {
    const size_t n = (size_t)d.nplane * (SW_GEO + SW_DYN + SW_RHS);
    HIPCHK(hipMalloc((void**)&b.sweep_rec, sizeof(double) * n));
    HIPCHK(hipMemsetAsync(b.sweep_rec, 0, sizeof(double) * n, c->stream));
    d.sw_geo = b.sweep_rec;
    d.sw_dyn = d.sw_geo + (size_t)d.nplane * SW_GEO;
    d.sw_rhs = d.sw_dyn + (size_t)d.nplane * SW_DYN;
  }
  if (use_d2(c) && std::min(d.ni, d.nj) > KP_MAX_DIAG)
    return fail("LU-SGS: a block with min(ni, nj) = %d exceeds the %d cells per diagonal "
                "the sweep kernel holds in LDS (AGX_LUSGS=plane has no such limit)",
                std::min(d.ni, d.nj), KP_MAX_DIAG);
  if (use_d2(c)) {
    // diagonal-ordered arrays of the LU-SGS path (agx_lusgs.hpp)
    D2Dev& z = d.d2;
    z.Pi = d.ni + 2 * d.ng; z.Pj = d.nj + 2 * d.ng;
    z.ps = ((long)z.Pi * z.Pj + 15) / 16 * 16;
    z.nd2 = z.ps * (d.nk + 2 * d.ng);
    b.dstart.assign(z.Pi + z.Pj + 1, 0);
    std::vector<int> tab(z.Pi + z.Pj + 1 + (size_t)z.Pi * z.Pj);
    for (int de = 0; de < z.Pi + z.Pj - 1; ++de) {
      const int jl = std::max(0, de - (z.Pi - 1)), jh = std::min(z.Pj - 1, de);
      b.dstart[de + 1] = b.dstart[de] + (jh - jl + 1);
      for (int je = jl; je <= jh; ++je)
        tab[z.Pi + z.Pj + 1 + b.dstart[de] + (je - jl)] = (de - je) | (je << 16);
    }
    b.dstart[z.Pi + z.Pj] = b.dstart[z.Pi + z.Pj - 1];
    std::copy(b.dstart.begin(), b.dstart.end(), tab.begin());
    HIPCHK(hipMalloc((void**)&b.d2_tab, sizeof(int) * tab.size()));
    HIPCHK(hipMemcpy(b.d2_tab, tab.data(), sizeof(int) * tab.size(), hipMemcpyHostToDevice));
    z.dstart = b.d2_tab;
    z.ij_of_pos = b.d2_tab + z.Pi + z.Pj + 1;
    HIPCHK(hipMalloc((void**)&b.d2, sizeof(double) * z.nd2 * D2_DOUBLES));
    HIPCHK(hipMemsetAsync(b.d2, 0, sizeof(double) * z.nd2 * D2_DOUBLES, c->stream));
    z.base = b.d2;
  }
  const int ci = d.ni + 2 * d.ng, cj = d.nj + 2 * d.ng, ck = d.nk + 2 * d.ng;
  if (upload_aos(c, b, g->farea_i, d.fa[0], 4, ci + 1, cj, ck, d.ng)) return 1;
  if (upload_aos(c, b, g->farea_j, d.fa[1], 4, ci, cj + 1, ck, d.ng)) return 1;
  if (upload_aos(c, b, g->farea_k, d.fa[2], 4, ci, cj, ck + 1, d.ng)) return 1;
  double* volp[1] = {d.vol};
  if (upload_aos(c, b, g->vol, volp, 1, ci, cj, ck, d.ng)) return 1;
  if (upload_aos(c, b, g->center, d.cen, 3, ci, cj, ck, d.ng)) return 1;
  const double* wsrc[3] = {g->width_i, g->width_j, g->width_k};
  for (int q = 0; q < 3; ++q) {
    double* wp[1] = {d.wid[q]};
    if (upload_aos(c, b, wsrc[q], wp, 1, ci, cj, ck, d.ng)) return 1;
  }
#if AGX_FAST
  if (d.d2.base) {
    const long n = (long)d.d2.Pi * d.d2.Pj * (d.nk + 2 * d.ng);
    hipLaunchKernelGGL(k_d2_geo, dim3((n + 255) / 256), dim3(256), 0, c->stream, d);
  }
#endif
  HIPCHK(hipGetLastError());
  if (g->wall_dist) {
    double* wp[1] = {d.wdist};
    if (upload_aos(c, b, g->wall_dist, wp, 1, ci, cj, ck, d.ng)) return 1;
  }
  *block_id = (int)c->blocks.size() - 1;
  return 0;
}

int agx_block_set_bcs(agx_ctx* c, int id, int n, const agx_bc_surface* s) {
  if (id < 0 || id >= (int)c->blocks.size()) return fail("bad block id %d", id);
  Block& b = c->blocks[id];
  for (auto& g1 : b.sweep_graph) for (auto& g2 : g1) for (auto& g3 : g2)
    if (g3) { hipGraphExecDestroy(g3); g3 = nullptr; }     // (they hold the old BlockDev)
  b.surf_host.assign(s, s + n);
  if (b.surf_dev) HIPCHK(hipFree(b.surf_dev));
  HIPCHK(hipMalloc((void**)&b.surf_dev, sizeof(agx_bc_surface) * (n > 0 ? n : 1)));
  HIPCHK(hipMemcpy(b.surf_dev, s, sizeof(agx_bc_surface) * n, hipMemcpyHostToDevice));
  b.d.surf = b.surf_dev;
  b.d.nsurf = n;
  b.d.nsurf_i = b.d.nsurf_j = b.d.nsurf_k = 0;
  int n_conn[6] = {0, 0, 0, 0, 0, 0}, n_other[6] = {0, 0, 0, 0, 0, 0};
  for (int q = 0; q < n; ++q) {
    const int st = surface_type(s[q]);
    if (st <= 2) b.d.nsurf_i++; else if (st <= 4) b.d.nsurf_j++; else b.d.nsurf_k++;
    const int t = s[q].bc_type;
    if (t < AGX_BC_SLIPWALL || t > AGX_BC_PERIODIC) return fail("unknown bc type %d", t);
    if (t == AGX_BC_INTERBLOCK || t == AGX_BC_PERIODIC) n_conn[st - 1]++; else n_other[st - 1]++;
    if (t == AGX_BC_VISCOUSWALL && s[q].state.is_wall_law) {
      // wall functions: the 7-equation library (wallLaw::AdiabaticBCs / HeatFluxBCs / IsothermalBCs)
      if (AGX_NEQ == 5) return fail("wallTreatment=wallLaw needs the rans library");
    }
  }
  for (int q = 0; q < 6; ++q)
    b.d.side_conn[q] = n_conn[q] == 0 ? 0 : (n_other[q] == 0 ? 1 : 2);
  // nonreflecting inlet / outlet surfaces keep the gradients of their adjacent cells
  // and their Mach mean / maximum between a residual and the next ghost fills
  if (b.nr_off_dev) HIPCHK(hipFree(b.nr_off_dev));
  if (b.nr_mem) HIPCHK(hipFree(b.nr_mem));
  b.nr_off_dev = nullptr; b.nr_mem = nullptr; b.nr_max = 0;
  b.d.nr_off = nullptr; b.d.nr_grad = nullptr; b.d.nr_mach = nullptr;
  std::vector<int> off(n > 0 ? n : 1, -1);
  long total = 0;
  for (int q = 0; q < n; ++q) {
    const int t = s[q].bc_type;
    if (!s[q].state.is_nonreflecting || (t != AGX_BC_INLET && t != AGX_BC_PRESSURE_OUTLET)) continue;
    const int st = surface_type(s[q]);
    const int d3 = (st - 1) / 2, d1 = (d3 + 1) % 3, d2 = (d3 + 2) % 3;
    const int lo[3] = {s[q].imin, s[q].jmin, s[q].kmin}, hi[3] = {s[q].imax, s[q].jmax, s[q].kmax};
    const long cells = (long)(hi[d1] - lo[d1]) * (hi[d2] - lo[d2]);
    off[q] = (int)total;
    total += cells;
    b.nr_max = std::max(b.nr_max, cells);
  }
  if (total > 0) {
    HIPCHK(hipMalloc((void**)&b.nr_off_dev, sizeof(int) * n));
    HIPCHK(hipMemcpy(b.nr_off_dev, off.data(), sizeof(int) * n, hipMemcpyHostToDevice));
    const size_t nd = (size_t)12 * total + 2 * (size_t)n;
    HIPCHK(hipMalloc((void**)&b.nr_mem, sizeof(double) * nd));
    HIPCHK(hipMemset(b.nr_mem, 0, sizeof(double) * nd));   // gradients before the first residual
    b.d.nr_off = b.nr_off_dev;
    b.d.nr_grad = b.nr_mem;
    b.d.nr_mach = b.nr_mem + 12 * total;
  }
  // wall-law surfaces keep the wall data of their faces (wallData_, wallData.hpp:33-62)
  // from the viscous ghost fill to the viscous fluxes of the same residual
  if (b.wall_off_dev) HIPCHK(hipFree(b.wall_off_dev));
  if (b.wall_mem) HIPCHK(hipFree(b.wall_mem));
  b.wall_off_dev = nullptr; b.wall_mem = nullptr;
  b.d.wall_off = nullptr; b.d.wallv = nullptr;
  std::fill(off.begin(), off.end(), -1);
  total = 0;
  for (int q = 0; q < n; ++q) {
    if (s[q].bc_type != AGX_BC_VISCOUSWALL || !s[q].state.is_wall_law) continue;
    const int st = surface_type(s[q]);
    const int d3 = (st - 1) / 2, d1 = (d3 + 1) % 3, d2 = (d3 + 2) % 3;
    const int lo[3] = {s[q].imin, s[q].jmin, s[q].kmin}, hi[3] = {s[q].imax, s[q].jmax, s[q].kmax};
    off[q] = (int)total;
    total += (long)(hi[d1] - lo[d1]) * (hi[d2] - lo[d2]);
  }
  if (total > 0) {
    HIPCHK(hipMalloc((void**)&b.wall_off_dev, sizeof(int) * n));
    HIPCHK(hipMemcpy(b.wall_off_dev, off.data(), sizeof(int) * n, hipMemcpyHostToDevice));
    const size_t nd = sizeof(WallVars) * (size_t)total;
    HIPCHK(hipMalloc((void**)&b.wall_mem, nd));
    HIPCHK(hipMemset(b.wall_mem, 0, nd));        // y+ = 0: low-Re until the first ghost fill
    b.d.wall_off = b.wall_off_dev;
    b.d.wallv = (WallVars*)b.wall_mem;
  }
  return 0;
}

int agx_conn_create(agx_ctx* c, const agx_connection* cc, int* conn_id) {
  c->conns.emplace_back();
  c->conns.back().c = *cc;
  *conn_id = (int)c->conns.size() - 1;
  return 0;
}

// workgroups of k_matrix_resid_d2: 8 XCDs x mresid_split bands x chunks per band x nk
#ifndef AGX_MRESID_KC
#define AGX_MRESID_KC 32
#endif
constexpr int MRESID_KC = AGX_MRESID_KC;   // planes a workgroup of k_matrix_resid_d2m marches
long mresid_wgs(const agx_ctx* c, const BlockDev& b) {
  const long nchunk = ((long)b.d2.Pi * b.d2.Pj + 255) / 256;
  if (c->mresid_march) return 8L * ((nchunk + 7) / 8) * ((b.nk + MRESID_KC - 1) / MRESID_KC);
  const long bands = 8L * c->mresid_split;
  return bands * ((nchunk + bands - 1) / bands) * b.nk;
}

namespace {
// The local connections exchanged in as few launches as their order allows.  The reference
// takes them one after the other (multiArray3d.hpp:790-828: slice both sides, insert both):
// where patches border other connections a later slice reads ghost cells an earlier insert
// wrote, and two inserts write the same edge ghost cells.  Connections are therefore put
// into levels -- "all slices of the level, then all its inserts", one gather and one scatter
// launch per level -- such that a connection comes after (a higher level than) every earlier
// one whose insert it reads or overwrites, and not before (the same level will do: slices
// precede inserts) an earlier one whose slice reads what it inserts; checked cell by cell.
int halo_batch_plan(agx_ctx* c, long* max_halo) {
  int sides = 0;
  long total = 0;
  std::vector<std::vector<unsigned char>> wlev(c->blocks.size()), rlev(c->blocks.size());
  auto flags = [&](std::vector<std::vector<unsigned char>>& v, int blk) -> std::vector<unsigned char>& {
    if (v[blk].empty()) v[blk].assign((size_t)c->blocks[blk].d.nplane, 0);
    return v[blk];
  };
  std::vector<std::pair<int, int>> order;        // (level, connection)
  int nlev = 0;
  for (size_t n = 0; n < c->conns.size(); ++n) {
    auto& k = c->conns[n];
    const agx_connection& cc = k.c;
    if (!(cc.rank[0] == c->rank && cc.rank[1] == c->rank)) continue;
    int lev = 1;
    for (int sd = 0; sd < 2; ++sd) {
      // side sd is filled from cells of the partner block
      const auto& wr = flags(wlev, cc.local_block[1 - sd]);
      for (long q : k.side[sd].h_src) lev = std::max(lev, wr[(size_t)q] + 1);
      const auto& ww = flags(wlev, cc.local_block[sd]);
      const auto& rr = flags(rlev, cc.local_block[sd]);
      for (long q : k.side[sd].h_dst)
        lev = std::max(lev, std::max<int>(ww[(size_t)q] + 1, rr[(size_t)q]));
    }
    lev = std::min(lev, 250);
    for (int sd = 0; sd < 2; ++sd) {
      auto& rr = flags(rlev, cc.local_block[1 - sd]);
      for (long q : k.side[sd].h_src) rr[(size_t)q] = std::max<unsigned char>(rr[(size_t)q], lev);
      auto& ww = flags(wlev, cc.local_block[sd]);
      for (long q : k.side[sd].h_dst) ww[(size_t)q] = (unsigned char)lev;
      total += k.side[sd].n;
    }
    order.emplace_back(lev, (int)n);
    nlev = std::max(nlev, lev);
    sides += 2;
  }
  for (auto& k : c->conns)
    for (int sd = 0; sd < 2; ++sd) {
      std::vector<long>().swap(k.side[sd].h_dst);
      std::vector<long>().swap(k.side[sd].h_src);
    }
  // (250 levels: a chain this long gains nothing over the pairwise launches)
  c->halo_batch = c->halo_batch && sides >= 4 && nlev < 250 && 2 * nlev < sides;
  if (!c->halo_batch && c->halo_batch_required)
    return fail("AGX_HALO_BATCH=require: %d local connections in %d levels", sides / 2, nlev);
  if (!c->halo_batch) return 0;
  std::stable_sort(order.begin(), order.end(),
                   [](const std::pair<int, int>& x, const std::pair<int, int>& y) { return x.first < y.first; });
  c->halo_conn_order.clear();
  c->halo_levels.assign(nlev, agx_ctx::HaloLevel{0, 0, 0});
  long nmax_all = 0;
  for (auto& o : order) {
    auto& L = c->halo_levels[o.first - 1];
    if (L.sides == 0) L.first = 2 * (int)c->halo_conn_order.size();
    L.sides += 2;
    const auto& k = c->conns[o.second];
    L.nmax = std::max(L.nmax, std::max(k.side[0].n, k.side[1].n));
    nmax_all = std::max(nmax_all, L.nmax);
    c->halo_conn_order.push_back(o.second);
  }
  c->halo_batch_sides = sides;
  c->halo_batch_nmax = nmax_all;
  *max_halo = std::max(*max_halo, (total * AGX_NEQ + 1) / 2);     // (the buffer is 2 * max_halo)
  HIPCHK(hipHostMalloc((void**)&c->halo_tab_host, sizeof(HaloSide) * 2 * sides));
  return 0;
}
// the gather / scatter tables of one halo selector (plane pointers of every side)
int halo_batch_tables(agx_ctx* c, int what) {
  const int set = c->halo_set[what];
  if (c->halo_tab_valid[what][set]) return 0;
  const int sides = c->halo_batch_sides;
  for (int q = 0; q < 2; ++q)
    if (!c->halo_tab_dev[what][set][q])
      HIPCHK(hipMalloc((void**)&c->halo_tab_dev[what][set][q], sizeof(HaloSide) * sides));
  // (the staging entries may still be read by an earlier copy)
  HIPCHK(hipStreamSynchronize(c->stream));
  HaloSide* g = c->halo_tab_host;
  HaloSide* p = c->halo_tab_host + sides;
  long off = 0;
  int n = 0;
  for (int cid : c->halo_conn_order) {            // (level by level)
    auto& k = c->conns[cid];
    const agx_connection& cc = k.c;
    Block& b0 = c->blocks[cc.local_block[0]];
    Block& b1 = c->blocks[cc.local_block[1]];
    const bool z2 = halo_in_d2(b0, what);
    const long n0 = k.side[0].n, n1 = k.side[1].n;
    double* buf0 = c->halo_buf + off;
    double* buf1 = buf0 + n0 * AGX_NEQ;
    off += (n0 + n1) * AGX_NEQ;
    g[n] = HaloSide{halo_planes(b1, what), z2 ? k.side[0].src2 : k.side[0].src, n0, buf0};
    g[n + 1] = HaloSide{halo_planes(b0, what), z2 ? k.side[1].src2 : k.side[1].src, n1, buf1};
    p[n] = HaloSide{halo_planes(b0, what), z2 ? k.side[0].dst2 : k.side[0].dst, n0, buf0};
    p[n + 1] = HaloSide{halo_planes(b1, what), z2 ? k.side[1].dst2 : k.side[1].dst, n1, buf1};
    n += 2;
  }
  HIPCHK(hipMemcpyAsync(c->halo_tab_dev[what][set][0], g, sizeof(HaloSide) * sides,
                        hipMemcpyHostToDevice, c->stream));
  HIPCHK(hipMemcpyAsync(c->halo_tab_dev[what][set][1], p, sizeof(HaloSide) * sides,
                        hipMemcpyHostToDevice, c->stream));
  c->halo_tab_valid[what][set] = true;
  return 0;
}
}  // namespace

int agx_setup_finalize(agx_ctx* c) {
  HIPCHK(hipSetDevice(c->device));
  const int ng = c->cfg.n_ghost;
  long max_parts = 1, max_halo = 1;
  long march_parts = 0;
  for (auto& blk : c->blocks) {
    const dim3 g = cell_grid(blk.d, CELL_BLOCK);
    max_parts = std::max(max_parts, (long)g.x * g.y * g.z);
    march_parts += march_plan(c, blk.d).nparts;
    if (blk.d.d2.base)    // k_matrix_resid_d2: one partial per 256 plane positions
      max_parts = std::max(max_parts, mresid_wgs(c, blk.d));
  }
  max_parts = std::max(max_parts, march_parts);
  for (auto& k : c->conns) {
    const agx_connection& cc = k.c;
    const bool l0 = cc.rank[0] == c->rank, l1 = cc.rank[1] == c->rank;
    if (!l0 && !l1) continue;
    for (int s = 0; s < 2; ++s) {
      const bool mine = s == 0 ? l0 : l1, partner_mine = s == 0 ? l1 : l0;
      if (!mine) continue;
      const int lb = cc.local_block[s];
      if (lb < 0 || lb >= (int)c->blocks.size()) return fail("connection refers to unknown block");
      const BlockDev& br = c->blocks[lb].d;
      auto idxR = [&](int i, int j, int kk) { return br.idx(i, j, kk); };
      MapOut m;
      if (partner_mine) {
        const BlockDev& bs = c->blocks[cc.local_block[1 - s]].d;
        auto idxS = [&](int i, int j, int kk) { return bs.idx(i, j, kk); };
        build_side_map(cc, s, ng, idxR, idxS, m);
      } else {
        auto idxS = [&](int, int, int) { return 0L; };
        build_side_map(cc, s, ng, idxR, idxS, m);
        // cells of my block the partner's ghost cells read, in the partner's
        // insertion order (what agx_halo_pack sends)
        MapOut ms;
        auto idxRp = [&](int, int, int) { return 0L; };
        build_side_map(cc, 1 - s, ng, idxRp, idxR, ms);
        k.n_send = (long)ms.src.size();
        if (to_device(ms.src, &k.send_src)) return 1;
        max_halo = std::max(max_halo, (long)ms.src.size() * AGX_NEQ);
      }
      k.side[s].n = (long)m.dst.size();
      k.side[s].h_dst = m.dst;
      if (partner_mine) k.side[s].h_src = m.src;
      if (to_device(m.dst, &k.side[s].dst)) return 1;
      if (partner_mine && to_device(m.src, &k.side[s].src)) return 1;
      max_halo = std::max(max_halo, (long)m.dst.size() * AGX_NEQ);
      if (use_d2(c)) {
        // the same maps in D2 index space: x of the LU-SGS path lives there
        const Block& Br = c->blocks[lb];
        auto idxR2 = [&](int i, int j, int kk) { return Br.d2idx(i, j, kk); };
        MapOut m2;
        if (partner_mine) {
          const Block& Bs = c->blocks[cc.local_block[1 - s]];
          auto idxS2 = [&](int i, int j, int kk) { return Bs.d2idx(i, j, kk); };
          build_side_map(cc, s, ng, idxR2, idxS2, m2);
          if (to_device(m2.src, &k.side[s].src2)) return 1;
        } else {
          auto idxS0 = [&](int, int, int) { return 0L; };
          build_side_map(cc, s, ng, idxR2, idxS0, m2);
          MapOut ms2;
          build_side_map(cc, 1 - s, ng, idxS0, idxR2, ms2);
          if (to_device(ms2.src, &k.send_src2)) return 1;
        }
        if (to_device(m2.dst, &k.side[s].dst2)) return 1;
      }
    }
  }
  if (halo_batch_plan(c, &max_halo)) return 1;
  if (ensure_halo_buf(c, 2 * max_halo)) return 1;
  // persistent slab pairs of the connections to other ranks (sorted by connection
  // id on every rank: the order RCCL's p2p matching relies on)
  std::vector<std::pair<int, int>> per_peer;   // (peer, connections seen so far)
  for (size_t n = 0; n < c->conns.size(); ++n) {
    if (my_side(c, c->conns[n]) < 0) continue;
    agx_ctx::Remote r;
    r.cid = (int)n;
    r.count = (long)agx_halo_count(c, (int)n, AGX_HALO_STATE);
    HIPCHK(hipMalloc((void**)&r.send, sizeof(double) * std::max<long>(r.count, 1)));
    HIPCHK(hipMalloc((void**)&r.recv, sizeof(double) * std::max<long>(r.count, 1)));
    if (c->have_ex && c->ex.host_buffers) {
      HIPCHK(hipHostMalloc((void**)&r.hsend, sizeof(double) * std::max<long>(r.count, 1)));
      HIPCHK(hipHostMalloc((void**)&r.hrecv, sizeof(double) * std::max<long>(r.count, 1)));
    }
    c->remote.push_back(r);
    const int s = my_side(c, c->conns[n]);
    agx_slab sl;
    sl.peer = c->conns[n].c.rank[1 - s];
    // tag: ordinal among the connections with this peer (hosts create connections
    // in the global order on every rank, so both sides count alike)
    sl.tag = 0;
    bool seen = false;
    for (auto& pp : per_peer)
      if (pp.first == sl.peer) { sl.tag = pp.second++; seen = true; }
    if (!seen) per_peer.emplace_back(sl.peer, 1);
    sl.count = r.count;
    const bool hb = c->have_ex && c->ex.host_buffers;
    sl.send = hb ? r.hsend : r.send;
    sl.recv = hb ? r.hrecv : r.recv;
    c->slabs.push_back(sl);
  }
  if (c->have_ex) {
    const size_t nrec = 1 + (size_t)c->ex.nranks;
    HIPCHK(hipMalloc((void**)&c->rec_dev, sizeof(agx_ctx::NormRecord) * nrec));
    HIPCHK(hipHostMalloc((void**)&c->rec_host, sizeof(agx_ctx::NormRecord) * nrec));
  }
  c->n_partials = max_parts;
  HIPCHK(hipMalloc((void**)&c->partials, sizeof(NormPartial) * max_parts));
  const size_t nb = std::max<size_t>(c->blocks.size(), 1);
  // [0, nb): norms of the update; [nb, nb + 64): scratch of the two-level fold;
  // [nb + 64, 2 nb + 64): the matrix residual (read back with the update's norms)
  HIPCHK(hipMalloc((void**)&c->norm_out, sizeof(NormPartial) * (2 * nb + 64)));
  HIPCHK(hipHostMalloc((void**)&c->norm_host, sizeof(NormPartial) * 2 * nb));
  c->finalized = true;
  return 0;
}

int agx_state_upload(agx_ctx* c, int id, const double* state) {
  c->ghosts_prefilled = false;
  c->state_is_time_n = false;
  if (flush_consn(c)) return 1;
  if (id < 0 || id >= (int)c->blocks.size()) return fail("bad block id %d", id);
  Block& b = c->blocks[id];
  const BlockDev& d = b.d;
  if (upload_aos(c, b, state, d.state, AGX_NEQ, d.ni + 2 * d.ng,
                 d.nj + 2 * d.ng, d.nk + 2 * d.ng, d.ng))
    return 1;
#if AGX_NEQ == 7
  // gridLevel::AuxillaryAndWidths main.cpp:169: viscosity_ before the first iteration
  // (the rans wall ghost states read the viscosity_ of the LAST UpdateAuxillaryVariables)
  hipLaunchKernelGGL(k_aux_field, dim3((d.nplane + 255) / 256), dim3(256), 0, c->stream, d,
                     c->gas, 1, d.viscp);
  HIPCHK(hipGetLastError());
#endif
  return 0;
}

static int field_info(Block& b, int field, double* const** p, int* ncomp, int* ghost) {
  BlockDev& d = b.d;
  static thread_local double* one[1];
  switch (field) {
    case AGX_FIELD_STATE: *p = d.state; *ncomp = AGX_NEQ; *ghost = 1; return 0;
    case AGX_FIELD_RESIDUAL: *p = d.resid; *ncomp = AGX_NEQ; *ghost = 0; return 0;
    case AGX_FIELD_CONS_N: *p = d.consn; *ncomp = AGX_NEQ; *ghost = 0; return 0;
    case AGX_FIELD_CONS_NM1: *p = d.consnm1; *ncomp = AGX_NEQ; *ghost = 0; return 0;
    case AGX_FIELD_UPDATE: *p = d.x; *ncomp = AGX_NEQ; *ghost = 1; return 0;
    case AGX_FIELD_DT: one[0] = d.dt; *p = one; *ncomp = 1; *ghost = 0; return 0;
    case AGX_FIELD_SPEC_RADIUS: one[0] = d.specrad; *p = one; *ncomp = 1; *ghost = 0; return 0;
    case AGX_FIELD_DIAGONAL: one[0] = d.a; *p = one; *ncomp = 1; *ghost = 0; return 0;
  }
  return 1;
}

// x of the D2 LU-SGS path <-> the SoA planes the field transfers use
static int d2_x_copy(agx_ctx* c, Block& b, int to_d2) {
#if AGX_FAST
  b.x_planes_current = true;
  const long n = (long)b.d.d2.Pi * b.d.d2.Pj * (b.d.nk + 2 * b.d.ng);
  hipLaunchKernelGGL(k_d2_x_copy, dim3((n + 255) / 256), dim3(256), 0, c->stream, b.d, to_d2,
                     to_d2 ? (unsigned)(++b.kp_epoch) & 3u : 0u);
  HIPCHK(hipGetLastError());
#endif
  return 0;
}

// The gradients an output step asks for reach into the ghost cells: they are formed with
// the ghost cells the NEXT residual would see -- the inviscid fill of the state as it is now
// (already queued behind the last update unless something touched the state since), then
// the viscous-wall fill (gridLevel.cpp:287-319, procBlock.cpp:6131-6136).  Ghost cells of
// connections to other ranks stay as last exchanged (an output step is not a collective).
static int ghosts_for_output(agx_ctx* c) {
  if (!c->ghosts_prefilled) {
    if (agx_phase_bc_faces(c)) return 1;
    if (agx_halo_swap_local(c, AGX_HALO_STATE)) return 1;
    if (agx_phase_bc_edges(c)) return 1;
  }
  if (c->cfg.is_viscous) {
    if (bc_pass(c, true, 1)) return 1;
    if (bc_pass(c, false, 1)) return 1;
  }
  c->ghosts_prefilled = false;   // (the next iteration starts from its own inviscid fill)
  return 0;
}

int agx_field_download(agx_ctx* c, int id, int field, double* out) {
  if (flush_consn(c)) return 1;
  if (id < 0 || id >= (int)c->blocks.size()) return fail("bad block id %d", id);
  Block& b = c->blocks[id];
  if (field == AGX_FIELD_UPDATE && b.d.d2.base && d2_x_copy(c, b, 0)) return 1;
  if (field >= AGX_FIELD_VEL_GRAD && field <= AGX_FIELD_PRESS_GRAD) {
    // cell-centre gradients: formed on demand into a temporary (an output path)
    const long ncell = (long)b.d.ni * b.d.nj * b.d.nk;
    double* tmp = nullptr;
    if (ghosts_for_output(c)) return 1;
    if (stage_buffer(c, (size_t)3 * NGF * ncell, &tmp)) return 1;
    hipLaunchKernelGGL(k_cell_grads, cell_grid(b.d, CELL_BLOCK), CELL_BLOCK, 0, c->stream, b.d,
                       c->gas, tmp);
    HIPCHK(hipGetLastError());
    const int off = field == AGX_FIELD_VEL_GRAD ? 0 : 9 + 3 * (field - AGX_FIELD_TEMP_GRAD);
    const int nc = field == AGX_FIELD_VEL_GRAD ? 9 : 3;
    // strided device -> host copy of the requested columns
    HIPCHK(hipMemcpy2DAsync(out, sizeof(double) * nc, tmp + off, sizeof(double) * 3 * NGF,
                            sizeof(double) * nc, ncell, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return 0;
  }
  if (field == AGX_FIELD_TEMPERATURE || field == AGX_FIELD_VISCOSITY) {
    // formed on demand from the current state into a scratch plane (the x_old
    // plane of DPLUR: rebuilt at the start of every implicit iteration)
    if (field == AGX_FIELD_VISCOSITY && !c->sp.viscous)
      return fail("viscosity_ is only kept for viscous runs (procBlock.cpp:6171)");
    double* scratch = b.d.xold[0];
    hipLaunchKernelGGL(k_aux_field, dim3((b.d.nplane + 255) / 256), dim3(256), 0, c->stream,
                       b.d, c->gas, field == AGX_FIELD_VISCOSITY ? 1 : 0, scratch);
    HIPCHK(hipGetLastError());
    double* one[1] = {scratch};
    const int g = b.d.ng;
    return download_aos(c, b, out, one, 1, b.d.ni + 2 * g, b.d.nj + 2 * g, b.d.nk + 2 * g, g);
  }
  double* const* p; int nc, gh;
  if (field_info(b, field, &p, &nc, &gh))
    return fail("unknown field %d", field);
  const int g = gh ? b.d.ng : 0;
  return download_aos(c, b, out, p, nc, b.d.ni + 2 * g, b.d.nj + 2 * g, b.d.nk + 2 * g, g);
}
// ---- output: WriteFunFile / WriteRestart payloads packed on the device ------
namespace {
OutSpec out_spec(const agx_ctx* c, const Block& b) {
  OutSpec sp;
  memset(&sp, 0, sizeof sp);
  const agx_gas& a = c->cfg.gas;
  sp.rho_ref = a.rho_ref; sp.a_ref = a.a_ref; sp.l_ref = a.l_ref; sp.t_ref = a.t_ref;
  sp.mu_ref = c->gas.mu_ref;                  // transport::MuRef, transport.cpp:58-66
  sp.rank = c->rank;
  sp.global_pos = b.global_pos;
  return sp;
}
}  // namespace

int agx_output_pack(agx_ctx* c, int id, int nvar, const int32_t* vars, double* out) {
  if (flush_consn(c)) return 1;
  if (id < 0 || id >= (int)c->blocks.size()) return fail("bad block id %d", id);
  if (nvar < 1 || nvar > AGX_OUT_COUNT) return fail("agx_output_pack: nvar %d out of range", nvar);
  Block& b = c->blocks[id];
  OutSpec sp = out_spec(c, b);
  sp.nvar = nvar;
  bool need_grads = false;
  for (int v = 0; v < nvar; ++v) {
    if (vars[v] < 0 || vars[v] >= AGX_OUT_COUNT) return fail("unknown output variable %d", vars[v]);
    if (vars[v] == AGX_OUT_VISCOSITY && !c->sp.viscous)
      return fail("viscosity_ is only kept for viscous runs (procBlock.cpp:6171)");
    sp.var[v] = vars[v];
    need_grads = need_grads || (vars[v] >= AGX_OUT_VELGRAD && vars[v] < AGX_OUT_RESID);
  }
  const long ncell = (long)b.d.ni * b.d.nj * b.d.nk;
  double* tmp = nullptr;
  const size_t gdoubles = need_grads ? (size_t)3 * NGF * ncell : 0;
  if (stage_buffer(c, gdoubles + (size_t)nvar * ncell, &tmp)) return 1;
  if (need_grads) {
    if (ghosts_for_output(c)) return 1;
    hipLaunchKernelGGL(k_cell_grads, cell_grid(b.d, CELL_BLOCK), CELL_BLOCK, 0, c->stream, b.d,
                       c->gas, tmp);
  }
  double* packed = tmp + gdoubles;
  hipLaunchKernelGGL(k_output_pack, cell_grid(b.d, CELL_BLOCK), CELL_BLOCK, 0, c->stream, b.d,
                     c->gas, sp, need_grads ? tmp : nullptr, packed);
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpyAsync(out, packed, sizeof(double) * nvar * ncell, hipMemcpyDeviceToHost,
                        c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  return 0;
}

int agx_restart_pack(agx_ctx* c, int id, int which, double* out) {
  if (flush_consn(c)) return 1;
  if (id < 0 || id >= (int)c->blocks.size()) return fail("bad block id %d", id);
  if (which != 0 && which != 1) return fail("agx_restart_pack: which is 0 (state) or 1 (consVarsNm1)");
  Block& b = c->blocks[id];
  const long ncell = (long)b.d.ni * b.d.nj * b.d.nk;
  double* tmp = nullptr;
  if (stage_buffer(c, (size_t)(AGX_NEQ + 1) * ncell, &tmp)) return 1;
  hipLaunchKernelGGL(k_restart_pack, cell_grid(b.d, CELL_BLOCK), CELL_BLOCK, 0, c->stream, b.d,
                     out_spec(c, b), which, tmp);
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpyAsync(out, tmp, sizeof(double) * (AGX_NEQ + 1) * ncell, hipMemcpyDeviceToHost,
                        c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  return 0;
}

// ---- set-up helpers -------------------------------------------------
int agx_plot3d_metrics(agx_ctx* c, int ni, int nj, int nk, const double* nodes, double* vol,
                       double* center, double* fai, double* faj, double* fak, double* fci,
                       double* fcj, double* fck) {
  if (ni < 1 || nj < 1 || nk < 1) return fail("empty block");
  HIPCHK(hipSetDevice(c->device));
  const long nn = (long)(ni + 1) * (nj + 1) * (nk + 1), ncell = (long)ni * nj * nk;
  const long nf[3] = {(long)(ni + 1) * nj * nk, (long)ni * (nj + 1) * nk, (long)ni * nj * (nk + 1)};
  double* host_out[8] = {vol, center, fai, faj, fak, fci, fcj, fck};
  const long count[8] = {ncell, 3 * ncell, 4 * nf[0], 4 * nf[1], 4 * nf[2], 3 * nf[0], 3 * nf[1],
                         3 * nf[2]};
  size_t total = (size_t)3 * nn;
  for (int q = 0; q < 8; ++q) if (host_out[q]) total += (size_t)count[q];
  double* buf = nullptr;
  if (stage_buffer(c, total, &buf)) return 1;
  HIPCHK(hipMemcpyAsync(buf, nodes, sizeof(double) * 3 * nn, hipMemcpyHostToDevice, c->stream));
  double* dev_out[8];
  double* cur = buf + 3 * nn;
  for (int q = 0; q < 8; ++q) { dev_out[q] = host_out[q] ? cur : nullptr; if (host_out[q]) cur += count[q]; }
  MetricsOut o;
  o.vol = dev_out[0]; o.center = dev_out[1];
  for (int d = 0; d < 3; ++d) { o.fa[d] = dev_out[2 + d]; o.fc[d] = dev_out[5 + d]; }
  HIPCHK(hipMemsetAsync(c->err_dev, 0, sizeof(int), c->stream));
  hipLaunchKernelGGL(k_plot3d_metrics, dim3((unsigned)((nn + 255) / 256)), dim3(256), 0, c->stream,
                     ni, nj, nk, buf, o, c->err_dev);
  HIPCHK(hipGetLastError());
  for (int q = 0; q < 8; ++q)
    if (host_out[q])
      HIPCHK(hipMemcpyAsync(host_out[q], dev_out[q], sizeof(double) * count[q],
                            hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipMemcpyAsync(c->err_host, c->err_dev, sizeof(int), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  if (*c->err_host == 4) {
    *c->err_host = 0;
    hipMemsetAsync(c->err_dev, 0, sizeof(int), c->stream);
    return fail("negative volume in PLOT3D block");
  }
  return 0;
}

int agx_nearest_wall_distance(agx_ctx* c, int64_t ncell, const double* cen, int64_t nwall,
                              const double* wall, double* dist) {
  if (ncell < 1) return 0;
  if (nwall < 1) return fail("no wall points");
  HIPCHK(hipSetDevice(c->device));
  double* buf = nullptr;
  if (stage_buffer(c, (size_t)(4 * ncell + 3 * nwall), &buf)) return 1;
  double *dc = buf, *dw = buf + 3 * ncell, *dd = dw + 3 * nwall;
  HIPCHK(hipMemcpyAsync(dc, cen, sizeof(double) * 3 * ncell, hipMemcpyHostToDevice, c->stream));
  HIPCHK(hipMemcpyAsync(dw, wall, sizeof(double) * 3 * nwall, hipMemcpyHostToDevice, c->stream));
  hipLaunchKernelGGL(k_nearest_wall, dim3((unsigned)((ncell + 255) / 256)), dim3(256), 0, c->stream,
                     (long)ncell, dc, (long)nwall, dw, dd);
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpyAsync(dist, dd, sizeof(double) * ncell, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  return 0;
}

int agx_field_upload(agx_ctx* c, int id, int field, const double* in) {
  x_changed(c);
  c->ghosts_prefilled = false;
  c->state_is_time_n = false;
  if (flush_consn(c)) return 1;
  if (id < 0 || id >= (int)c->blocks.size()) return fail("bad block id %d", id);
  Block& b = c->blocks[id];
  double* const* p; int nc, gh;
  if (field_info(b, field, &p, &nc, &gh)) return fail("field %d cannot be uploaded", field);
  const int g = gh ? b.d.ng : 0;
  if (upload_aos(c, b, in, p, nc, b.d.ni + 2 * g, b.d.nj + 2 * g, b.d.nk + 2 * g, g)) return 1;
  if (field == AGX_FIELD_UPDATE && b.d.d2.base) return d2_x_copy(c, b, 1);
  return 0;
}

// ---- geometric multigrid (include/aither_gfx950.h) -------------------------------------------
static int implicit_begin(agx_ctx* c, int write_x);
int agx_phase_matrix_residual(agx_ctx* c, double* mr);
static int d2_x_copy(agx_ctx* c, Block& b, int to_d2);
#if AGX_FAST
static void d2_prepare(agx_ctx* c, Block& blk, int write_x, int again);
#endif
namespace {
int mg_solver_ok(agx_ctx* c) {
  if (!c->sp.implicit) return fail("multigrid: implicit time integration only");
  return 0;
}
// On the diagonal-ordered LU-SGS path x lives in the D2 arrays; the multigrid kernels work on
// the planes: current planes before they read, the D2 arrays (tagged) after they wrote.
// (x_planes_current: nothing wrote the D2 x since the last copy -- sweeps, exchanges, the
// prepare kernel and uploads clear it, x_changed)
int mg_x_to_planes(agx_ctx* c, Block& b) {
  return b.d.d2.base && !b.x_planes_current ? d2_x_copy(c, b, 0) : 0;
}
int mg_x_from_planes(agx_ctx* c, Block& b) { return b.d.d2.base ? d2_x_copy(c, b, 1) : 0; }
// x of a block's planes changed outside sweeps and exchanges: the records' copy follows
void mg_x_records(agx_ctx* c, Block& b) {
  if (b.d.sw_dyn)
    hipLaunchKernelGGL(k_mg_x_records, dim3((unsigned)((b.d.nplane + 255) / 256)), dim3(256), 0,
                       c->stream, b.d);
}
int mg_check(agx_ctx* f, agx_ctx* cz, int blk) {
  if (!f || !cz) return fail("mg: null context");
  if (blk < 0 || blk >= (int)f->blocks.size() || blk >= (int)cz->blocks.size())
    return fail("mg: bad block %d", blk);
  if (f->device != cz->device) return fail("mg: the two levels live on different devices");
  for (agx_ctx* c : {f, cz})
    if (mg_solver_ok(c)) return 1;
  return 0;
}
int mg_planes(agx_ctx* c, Block& b, double** p) {     // AGX_NEQ zeroed planes, once
  if (*p) return 0;
  HIPCHK(hipMalloc((void**)p, sizeof(double) * AGX_NEQ * b.d.nplane));
  HIPCHK(hipMemsetAsync(*p, 0, sizeof(double) * AGX_NEQ * b.d.nplane, c->stream));
  return 0;
}
// the device copies of a fine block's transfer maps (uploaded when the host pointer changes)
int mg_maps(agx_ctx* f, Block& b, const int32_t* tc, const double* vf, const double* cf,
            int cni, int cnj, int cnk, MgMap* m) {
  const BlockDev& d = b.d;
  const size_t ncell = (size_t)d.ni * d.nj * d.nk;
  if (tc && b.mg_tc_key != tc) {
    HIPCHK(hipStreamSynchronize(f->stream));
    if (!b.mg_tc) HIPCHK(hipMalloc((void**)&b.mg_tc, sizeof(int) * 3 * ncell));
    HIPCHK(hipMemcpy(b.mg_tc, tc, sizeof(int) * 3 * ncell, hipMemcpyHostToDevice));
    // first fine index of every coarse cell, per direction (the map is a product of three
    // monotone 1-D maps: procBlock.cpp:6556-6581)
    std::vector<int> st((size_t)cni + cnj + cnk + 3, 0);
    int* s0 = st.data();
    int* s1 = s0 + cni + 1;
    int* s2 = s1 + cnj + 1;
    const int n[3] = {d.ni, d.nj, d.nk}, cn[3] = {cni, cnj, cnk};
    int* ss[3] = {s0, s1, s2};
    for (int dir = 0; dir < 3; ++dir) {
      int cc = 0;
      ss[dir][0] = 0;
      for (int q = 0; q < n[dir]; ++q) {
        const size_t cell = dir == 0 ? (size_t)q : dir == 1 ? (size_t)q * d.ni : (size_t)q * d.ni * d.nj;
        const int to = tc[3 * cell + dir];
        if (to < 0 || to >= cn[dir] || to < cc) return fail("mg: to_coarse is not a monotone map onto the coarse block");
        while (cc < to) ss[dir][++cc] = q;
      }
      while (cc < cn[dir]) ss[dir][++cc] = n[dir];
    }
    if (b.mg_start) HIPCHK(hipFree(b.mg_start));
    HIPCHK(hipMalloc((void**)&b.mg_start, sizeof(int) * st.size()));
    HIPCHK(hipMemcpy(b.mg_start, st.data(), sizeof(int) * st.size(), hipMemcpyHostToDevice));
    b.mg_tc_key = tc;
  }
  if (vf && b.mg_vf_key != vf) {
    HIPCHK(hipStreamSynchronize(f->stream));
    if (!b.mg_vf) HIPCHK(hipMalloc((void**)&b.mg_vf, sizeof(double) * ncell));
    HIPCHK(hipMemcpy(b.mg_vf, vf, sizeof(double) * ncell, hipMemcpyHostToDevice));
    b.mg_vf_key = vf;
  }
  if (cf && b.mg_cf_key != cf) {
    HIPCHK(hipStreamSynchronize(f->stream));
    if (!b.mg_cf) HIPCHK(hipMalloc((void**)&b.mg_cf, sizeof(double) * 7 * ncell));
    HIPCHK(hipMemcpy(b.mg_cf, cf, sizeof(double) * 7 * ncell, hipMemcpyHostToDevice));
    b.mg_cf_key = cf;
  }
  if (!b.mg_tc) return fail("mg: no fine-to-coarse map for this block yet");
  m->tc = b.mg_tc;
  m->start[0] = b.mg_start;
  m->start[1] = b.mg_start + cni + 1;
  m->start[2] = m->start[1] + cnj + 1;
  m->vf = b.mg_vf;
  m->cf = b.mg_cf;
  return 0;
}
// the other level's stream has finished what it was given (the two contexts may run on
// different streams)
int mg_order(agx_ctx* producer, agx_ctx* consumer) {
  if (producer->stream != consumer->stream) HIPCHK(hipStreamSynchronize(producer->stream));
  return 0;
}
}  // namespace

int agx_mg_restrict(agx_ctx* f, agx_ctx* cz, int blk, int what, const int32_t* tc,
                    const double* vf) {
  if (mg_check(f, cz, blk)) return 1;
  Block &bf = f->blocks[blk], &bc = cz->blocks[blk];
  if (what < AGX_MG_STATE || what > AGX_MG_FORCING) return fail("mg_restrict: bad selector %d", what);
  if (what != AGX_MG_FORCING && !vf) return fail("mg_restrict: volume weights missing");
  if (flush_consn(f) || flush_consn(cz)) return 1;
  MgMap m;
  if (mg_maps(f, bf, tc, vf, nullptr, bc.d.ni, bc.d.nj, bc.d.nk, &m)) return 1;
  if (what == AGX_MG_UPDATE && mg_x_to_planes(f, bf)) return 1;
  if (what == AGX_MG_FORCING && mg_x_to_planes(cz, bc)) return 1;   // (ghosts of the exchange)
  if (mg_order(f, cz)) return 1;
  if (what == AGX_MG_STATE) {
    // (coarse.Zero(): ghost cells included, procBlock.hpp:641)
    for (int e = 0; e < AGX_NEQ; ++e)
      HIPCHK(hipMemsetAsync(bc.d.state[e], 0, sizeof(double) * bc.d.nplane, cz->stream));
    cz->ghosts_prefilled = false;
    cz->state_is_time_n = false;
    cz->mg_coarse = true;
  } else if (what == AGX_MG_UPDATE) {
    for (int e = 0; e < AGX_NEQ; ++e)
      HIPCHK(hipMemsetAsync(bc.d.x[e], 0, sizeof(double) * bc.d.nplane, cz->stream));
  } else {
    if (!bf.d.mg_mres) return fail("mg_restrict: the fine level has no matrix residual yet");
    if (mg_planes(cz, bc, &bc.mg_forcing)) return 1;
    bc.d.mg_forcing = bc.mg_forcing;
  }
  hipLaunchKernelGGL(k_mg_restrict, cell_grid(bc.d, CELL_BLOCK), CELL_BLOCK, 0, cz->stream, bf.d,
                     bc.d, m, what);
  if (what == AGX_MG_FORCING)
    hipLaunchKernelGGL(k_mg_axmb, cell_grid(bc.d, CELL_BLOCK), CELL_BLOCK, 0, cz->stream, bc.d,
                       cz->gas, cz->sp);
  if (what == AGX_MG_UPDATE) {
    mg_x_records(cz, bc);
    if (mg_x_from_planes(cz, bc)) return 1;
  }
#if AGX_FAST
  // (b of the diagonal-ordered sweeps: formed again, now with the forcing term)
  if (what == AGX_MG_FORCING && bc.d.d2.base) d2_prepare(cz, bc, 0, 1);
#endif
  HIPCHK(hipGetLastError());
  return 0;
}

int agx_mg_matrix_residual(agx_ctx* c, double* mr) {
  if (mg_solver_ok(c)) return 1;
  for (auto& blk : c->blocks) {
    if (mg_planes(c, blk, &blk.mg_mres)) return 1;
    blk.d.mg_mres = blk.mg_mres;
    if (mg_x_to_planes(c, blk)) return 1;
  }
  // (the plane form: the diagonal-ordered reductions form the norm only)
  c->mres_plane_form = true;
  const int rc = agx_phase_matrix_residual(c, mr);
  c->mres_plane_form = false;
  return rc;
}

int agx_mg_invert_diagonal(agx_ctx* c) { return implicit_begin(c, 0); }

// gridLevel::ResetDiagonal (gridLevel.cpp:408-412) of a coarse level, whose residual adds to the
// diagonal (SolverDev::diag_add); the finest level's kernels overwrite theirs
int agx_mg_reset_diagonal(agx_ctx* c) {
  for (auto& blk : c->blocks) {
    HIPCHK(hipMemsetAsync(blk.d.a, 0, sizeof(double) * blk.d.nplane, c->stream));
    if (blk.d.am)
      HIPCHK(hipMemsetAsync(blk.d.am, 0, sizeof(double) * AGX_NJ * blk.d.nplane, c->stream));
    if (AGX_NEQ > 5) {
      HIPCHK(hipMemsetAsync(blk.d.a_t, 0, sizeof(double) * blk.d.nplane, c->stream));
      if (blk.d.am_t)
        HIPCHK(hipMemsetAsync(blk.d.am_t, 0, sizeof(double) * 2 * blk.d.nplane, c->stream));
    }
  }
  return 0;
}

int agx_mg_save_update(agx_ctx* c) {
  for (auto& blk : c->blocks) {
    if (mg_planes(c, blk, &blk.mg_xsave)) return 1;
    blk.d.mg_xsave = blk.mg_xsave;
    if (mg_x_to_planes(c, blk)) return 1;
    hipLaunchKernelGGL(k_mg_axpy, dim3((unsigned)((blk.d.nplane + 255) / 256)), dim3(256), 0,
                       c->stream, blk.d, 1);
  }
  HIPCHK(hipGetLastError());
  return 0;
}

int agx_mg_prolong(agx_ctx* cz, agx_ctx* f, int blk, const int32_t* tc, const double* cf) {
  if (mg_check(f, cz, blk)) return 1;
  Block &bf = f->blocks[blk], &bc = cz->blocks[blk];
  if (!bc.d.mg_xsave) return fail("mg_prolong: no saved update on the coarse level");
  if (!cf) return fail("mg_prolong: interpolation coefficients missing");
  MgMap m;
  if (mg_maps(f, bf, tc, nullptr, cf, bc.d.ni, bc.d.nj, bc.d.nk, &m)) return 1;
  const long nn = (long)(bc.d.ni + 1) * (bc.d.nj + 1) * (bc.d.nk + 1);
  if (!bc.mg_nodes) HIPCHK(hipMalloc((void**)&bc.mg_nodes, sizeof(double) * AGX_NEQ * nn));
  if (mg_x_to_planes(cz, bc) || mg_x_to_planes(f, bf)) return 1;
  hipLaunchKernelGGL(k_mg_axpy, dim3((unsigned)((bc.d.nplane + 255) / 256)), dim3(256), 0,
                     cz->stream, bc.d, 0);
  const dim3 tb = CELL_BLOCK;
  hipLaunchKernelGGL(k_mg_nodes, dim3((bc.d.ni + tb.x) / tb.x, (bc.d.nj + tb.y) / tb.y, bc.d.nk + 1),
                     tb, 0, cz->stream, bc.d, bc.mg_nodes);
  HIPCHK(hipGetLastError());
  if (mg_order(cz, f)) return 1;
  hipLaunchKernelGGL(k_mg_prolong, cell_grid(bf.d, tb), tb, 0, f->stream, bf.d, m,
                     (const double*)bc.mg_nodes, bc.d.ni, bc.d.nj, bc.d.nk);
  mg_x_records(cz, bc);
  mg_x_records(f, bf);
  if (mg_x_from_planes(cz, bc) || mg_x_from_planes(f, bf)) return 1;
  HIPCHK(hipGetLastError());
  return 0;
}

int agx_store_time_n(agx_ctx* c, int also_nm1) {
  if (flush_consn(c)) return 1;
  c->have_time_n = true;
  // nonreflecting ghost states read consVarsN (ghostStates.cpp:435-462): a ghost fill
  // queued behind the last update saw the old time level
  for (auto& blk : c->blocks)
    if (blk.nr_max > 0) c->ghosts_prefilled = false;
  // Explicit fused path: the stage-0 launch of k_residual_tile forms
  // cons(state) anyway and writes it to consVarsN itself (5 stores instead of a
  // separate 5-load/5-store pass); anything else that touches consVarsN or the
  // state first calls flush_consn().
  static const bool lazy = !(getenv("AGX_NO_LAZY_CONSN") && atoi(getenv("AGX_NO_LAZY_CONSN")));
  c->state_is_time_n = true;
  if (lazy && !also_nm1 && !c->use_gather && all_tile_ok(c)) { c->consn_pending = true; return 0; }
  for (auto& blk : c->blocks)
    hipLaunchKernelGGL(k_store_time_n, cell_grid(blk.d, CELL_BLOCK), CELL_BLOCK,
                       0, c->stream, blk.d, c->gas, also_nm1);
  HIPCHK(hipGetLastError());
  return 0;
}

// ---- phases ---------------------------------------------------------------
int agx_phase_bc_faces(agx_ctx* c) { Timer t(c, G_BC); return bc_pass(c, true, 0); }
int agx_phase_bc_edges(agx_ctx* c) { Timer t(c, G_BC); return bc_pass(c, false, 0); }

int agx_phase_residual(agx_ctx* c, int mm, double cfl) {
  if (c->cfg.dt_nondim <= 0.0 && cfl <= 0.0)
    return fail("Neither dt or cfl was specified!");   // procBlock.cpp:813-816
  c->sp.diag_add = c->mg_coarse ? 1 : 0;
  const bool fuse = can_fuse(c);
  const int store_consn = !c->use_gather && all_tile_ok(c) && c->consn_pending && mm == 0;
  if (store_consn) c->consn_pending = false;
  else if (flush_consn(c)) return 1;
  {
    Timer t(c, G_RESID);
    long off = 0;
    for (auto& blk : c->blocks) {
      const MarchPlan mp = march_plan(c, blk.d);
      MarchArgs ma;
      memset(&ma, 0, sizeof ma);
      ma.kchunk = mp.kchunk;
      ma.mode = c->cfg.time_integration == AGX_TIME_RK4 ? 1 : 0;
      ma.store_consn = store_consn;
      const double rk_alpha[4] = {0.25, 1.0 / 3.0, 0.5, 1.0};   // procBlock.cpp:938
      ma.alpha = rk_alpha[mm & 3];
      ma.partials = c->partials + off;
      ma.ablate = getenv("AGX_ABLATE") ? atoi(getenv("AGX_ABLATE")) : 0;
      launch_inv(c, blk.d, cfl, fuse, ma, mp);
      off += mp.nparts;
      if (c->sp.implicit && c->sp.block) launch_block_diag(c, blk.d);
    }
  }
  HIPCHK(hipGetLastError());
  c->fused_pending = fuse;
  if (c->cfg.is_viscous) {
    {
      Timer t(c, G_BC);
      if (bc_pass(c, true, 1)) return 1;
      if (bc_pass(c, false, 1)) return 1;
    }
    Timer t(c, G_VISC);
    for (auto& blk : c->blocks)
#if AGX_NEQ == 7
      {
        const int fourth = c->cfg.viscous_recon == AGX_VISC_RECON_CENTRAL_4TH ? 1 : 0;
        const BlockDev& vb = blk.d;
        if (c->visc_gather) {
          hipLaunchKernelGGL(k_visc_residual_rans, cell_grid(vb, CELL_BLOCK), CELL_BLOCK,
                             0, c->stream, vb, c->gas, c->sp, cfl, fourth);
          continue;
        }
        // face-once form: every face evaluated by one thread, records through memory (the
        // blocks run one after the other on the stream and share the record buffer)
        RansRec rec;
        rec.ni1 = vb.ni + 1; rec.nj1 = vb.nj + 1;
        rec.nf = (long)rec.ni1 * rec.nj1 * (vb.nk + 1);
        const size_t need = (size_t)3 * RANS_REC * rec.nf;
        if (need > c->rans_rec_cap) {
          HIPCHK(hipStreamSynchronize(c->stream));
          if (c->rans_rec) HIPCHK(hipFree(c->rans_rec));
          c->rans_rec = nullptr; c->rans_rec_cap = 0;
          HIPCHK(hipMalloc((void**)&c->rans_rec, sizeof(double) * need));
          c->rans_rec_cap = need;
        }
        rec.p = c->rans_rec;
        const dim3 tb = CELL_BLOCK;
        auto fgrid = [&](int di, int dj, int dk) {
          return dim3((vb.ni + di + tb.x - 1) / tb.x, (vb.nj + dj + tb.y - 1) / tb.y, vb.nk + dk);
        };
        hipLaunchKernelGGL(k_rans_faces<0>, fgrid(1, 0, 0), tb, 0, c->stream, vb, c->gas, fourth, rec);
        hipLaunchKernelGGL(k_rans_faces<1>, fgrid(0, 1, 0), tb, 0, c->stream, vb, c->gas, fourth, rec);
        hipLaunchKernelGGL(k_rans_faces<2>, fgrid(0, 0, 1), tb, 0, c->stream, vb, c->gas, fourth, rec);
        hipLaunchKernelGGL(k_rans_cells, cell_grid(vb, tb), tb, 0, c->stream, vb, c->gas, c->sp,
                           cfl, fourth, rec);
      }
#else
    {
      const bool fourth = c->cfg.viscous_recon == AGX_VISC_RECON_CENTRAL_4TH;
      const bool wide_plane = (double)blk.d.nplane * 8.0 >= 4294967296.0;
      if (!AGX_FAST || c->visc_gather || (fourth && (c->visc_march || wide_plane))) {
        // (one thread per cell, the stencil straight from the planes: AGX_VISC=gather, and
        // centralFourth where the tile kernel's 32-bit plane offsets do not reach)
        hipLaunchKernelGGL(k_visc_residual, cell_grid(blk.d, CELL_BLOCK), CELL_BLOCK,
                           0, c->stream, blk.d, c->gas, c->sp, cfl, fourth ? 1 : 0);
      } else {
        const BlockDev& vb = blk.d;
        if (c->visc_march || wide_plane) {
          const int gx = (vb.ni + VTI - 1) / VTI, gy = (vb.nj + VTJ - 1) / VTJ;
          int nz = std::max(1, std::min(vb.nk / 8, (int)std::lround(2048.0 / (gx * gy))));
          const int kchunk = (vb.nk + nz - 1) / nz;
          nz = (vb.nk + kchunk - 1) / kchunk;
          hipLaunchKernelGGL(k_visc_march, dim3(gx, gy, nz), dim3(64, VTJ + 1), 0, c->stream,
                             vb, c->gas, c->sp, cfl, kchunk);
          continue;
        }
#if AGX_FAST
        // 62 x 6 owned cells per workgroup and k-step (centralFourth: 60 x 6); persistent
        // workgroups, one per CU (the kernel's LDS windows fill a CU), each marching an equal
        // share of the (column tile, k) steps
        const int oi = fourth ? VT_L - 4 : VT_OI;
        const int gx = (vb.ni + oi - 1) / oi, gy = (vb.nj + VT_OJ - 1) / VT_OJ;
        const long steps = (long)gx * gy * vb.nk;
        const int nwg = (int)std::min<long>(c->num_cu, std::max<long>(1, steps / 8));
        if (fourth)
          hipLaunchKernelGGL(k_visc_tile<true>, dim3(nwg), dim3(VT_L, VT_R), 0, c->stream,
                             make_slab(vb), c->gas, c->sp, cfl, gx, gy);
        else
          hipLaunchKernelGGL(k_visc_tile<false>, dim3(nwg), dim3(VT_L, VT_R), 0, c->stream,
                             make_slab(vb), c->gas, c->sp, cfl, gx, gy);
#endif
      }
    }
#endif  // AGX_NEQ == 7
#if AGX_NEQ != 7     // (the rans viscous kernel accumulates its Jacobians itself)
    if (c->sp.implicit && c->sp.block)
      for (auto& blk : c->blocks)
        hipLaunchKernelGGL(k_block_diag_visc, cell_grid(blk.d, CELL_BLOCK), CELL_BLOCK, 0,
                           c->stream, blk.d, c->gas, c->sp,
                           c->cfg.viscous_recon == AGX_VISC_RECON_CENTRAL_4TH ? 1 : 0);
#endif
    HIPCHK(hipGetLastError());
  }
  // the gradients of this residual's state feed the nonreflecting ghost states of
  // the ghost fills up to the next residual (CalcGrads*, procBlock.cpp:6143)
  for (auto& blk : c->blocks)
    if (blk.nr_max > 0) {
      Timer t(c, G_BC);
      hipLaunchKernelGGL(k_nr_grads, dim3((blk.nr_max + 255) / 256, blk.d.nsurf), dim3(256), 0,
                         c->stream, blk.d, c->gas);
      HIPCHK(hipGetLastError());
    }
  return 0;
}

int agx_phase_explicit_update(agx_ctx* c, int mm, double* l2, agx_linf* linf) {
  const int mode = c->cfg.time_integration == AGX_TIME_RK4 ? 1 : 0;
  return update_pass(c, mode, mm, l2, linf);
}

static int implicit_begin(agx_ctx* c, int write_x);
#if AGX_FAST
// k_lusgs_prepare of one block (write_x: a launch that writes x is a writer launch: it tags
// the values and advances the epoch)
static void d2_prepare(agx_ctx* c, Block& blk, int write_x, int again) {
  const BlockDev& b = blk.d;
  const dim3 grid((b.d2.Pi + TT - 1) / TT, (b.d2.Pj + TT - 1) / TT, b.nk + 2 * b.ng);
  hipLaunchKernelGGL(k_lusgs_prepare, grid, dim3(256), 0, c->stream, b, c->gas, c->sp,
                     write_x, write_x ? (unsigned)(++blk.kp_epoch) & 3u : 0u, again);
}
#endif
int agx_phase_implicit_begin(agx_ctx* c) { return implicit_begin(c, 1); }
// (write_x = 0: gridLevel::InvertDiagonal alone, for a coarse multigrid level)
static int implicit_begin(agx_ctx* c, int write_x) {
  x_changed(c);
  Timer t(c, G_PREPARE);
  c->sp.un_is_u = c->state_is_time_n ? 1 : 0;   // (read by every rhs_b of this iteration)
  for (auto& blk : c->blocks) {
    const BlockDev& b = blk.d;
#if AGX_FAST
    if (b.d2.base) {
      // diagonal terms, b and x0 straight into the D2 arrays of the sweeps (a coarse
      // multigrid level, write_x = 0: x is the restricted one)
      bool conn = false;
      for (int q = 0; q < 6; ++q) conn = conn || b.side_conn[q] != 0;
      d2_prepare(c, blk, write_x && (c->sp.requires_init || conn) ? 1 : 0, 0);
      continue;
    }
#endif
    if (!c->sp.requires_init && write_x)   // x_[bb].Zero() incl. ghosts, linearSolver.cpp:141
      hipLaunchKernelGGL(k_zero5, dim3((b.nplane + 255) / 256), dim3(256), 0,
                         c->stream, planes(b.x), b.nplane);
    hipLaunchKernelGGL(k_implicit_begin, cell_grid(b, CELL_BLOCK), CELL_BLOCK, 0,
                       c->stream, b, c->gas, c->sp, c->err_dev, write_x);
    if (b.sw_geo) {
      hipLaunchKernelGGL(k_sweep_records, dim3((unsigned)((b.nplane + 255) / 256)), dim3(256), 0,
                         c->stream, b, c->sp, blk.sweep_geo_built ? 0 : 1);
      blk.sweep_geo_built = true;
    }
  }
  HIPCHK(hipGetLastError());
  return 0;
}

int agx_phase_relax_forward(agx_ctx* c, int sweep) {
  x_changed(c);
  Timer t(c, G_SWEEP);
  const int full = sweep > 0 || c->sp.requires_init;
  bool swept = false;
  if (is_lusgs_solver(c) && pipe_sweep_applicable(c)) return lusgs_sweep_pipe(c, true, full);
  if (is_lusgs_solver(c) && plane_sweep_all_applicable(c)) return lusgs_sweep_all(c, true, full);
  for (auto& blk : c->blocks) {
    BlockDev& b = blk.d;
    if (is_lusgs_solver(c)) {
      if (lusgs_sweep(c, blk, true, full)) return 1;
      swept = true;
    } else {
      // dplur::DPLUR copies x to xold (linearSolver.cpp:487) and relaxes from the
      // copy; here the two sets of planes change roles instead.  Ghost cells of
      // the new x that nobody rewrites (physical boundaries) are never read
      // (ImplicitLower/Upper only cross physical or connection faces).
      for (int e = 0; e < AGX_NEQ; ++e) std::swap(b.x[e], b.xold[e]);
      AGX_BY_MODE(c->sp, k_dplur, cell_grid(b, CELL_BLOCK), CELL_BLOCK, c->stream, b, c->gas,
                  c->sp, 0);
    }
  }
  if (!is_lusgs_solver(c)) c->halo_set[AGX_HALO_UPDATE] ^= 1;   // (x and xold changed roles)
  (void)swept;
  HIPCHK(hipGetLastError());
  return 0;
}

int agx_phase_relax_backward(agx_ctx* c, int sweep) {
  x_changed(c);
  if (!is_lusgs_solver(c)) return 0;
  Timer t(c, G_SWEEP);
  const int full = sweep > 0 || c->sp.requires_init;
  if (pipe_sweep_applicable(c)) return lusgs_sweep_pipe(c, false, full);
  if (plane_sweep_all_applicable(c)) return lusgs_sweep_all(c, false, full);
  for (auto& blk : c->blocks) {
    if (lusgs_sweep(c, blk, false, full)) return 1;
  }
  HIPCHK(hipGetLastError());
  return 0;
}

namespace {
void matrix_residual_fold(agx_ctx* c, const NormPartial* host, double* mr) {
  double sumsq = 0.0;
  long size = 0;
  for (size_t n = 0; n < c->blocks.size(); ++n) {
    const BlockDev& b = c->blocks[n].d;
    for (int e = 0; e < AGX_NEQ; ++e) sumsq += host[n].l2[e];
    // mr.Size(): ghost-inclusive (linearSolver.cpp:66-68, mgSolution.cpp:203)
    size += (long)AGX_NEQ * (b.ni + 2 * b.ng) * (b.nj + 2 * b.ng) * (b.nk + 2 * b.ng);
  }
  *mr = size > 0 ? sumsq / (double)size : 0.0;
}
}  // namespace

int agx_phase_matrix_residual(agx_ctx* c, double* mr) {
  const bool defer = c->in_iterate;      // agx_iterate reads it back with the update's norms
  NormPartial* out = c->norm_out + (defer ? c->blocks.size() + 64 : 0);
  {
    Timer t(c, G_MRESID);
    for (size_t n = 0; n < c->blocks.size(); ++n) {
      const BlockDev& b = c->blocks[n].d;
#if AGX_FAST
      if (b.d2.base && !c->mres_plane_form) {
        // (position chunk, k) pairs dealt to the XCDs band by band, see the kernels
        const long nwg = mresid_wgs(c, b);
        if (c->mresid_march)
          hipLaunchKernelGGL(k_matrix_resid_d2m, dim3((unsigned)nwg), dim3(256), 0, c->stream, b,
                             c->gas, c->sp, MRESID_KC, c->partials);
        else
          hipLaunchKernelGGL(k_matrix_resid_d2, dim3((unsigned)nwg), dim3(256), 0, c->stream, b,
                             c->gas, c->sp, c->mresid_split, c->partials);
        if (reduce_norms(c, n, nwg, out + n)) return 1;
        continue;
      }
#endif
      const dim3 grid = cell_grid(b, CELL_BLOCK);
      AGX_BY_MODE(c->sp, k_matrix_resid, grid, CELL_BLOCK, c->stream, b, c->gas, c->sp,
                  c->partials);
      if (reduce_norms(c, n, (long)grid.x * grid.y * grid.z, out + n)) return 1;
    }
  }
  if (defer) {
    c->mres_deferred = true;
    *mr = 0.0;
    return 0;
  }
  HIPCHK(hipMemcpyAsync(c->norm_host, c->norm_out,
                        sizeof(NormPartial) * c->blocks.size(),
                        hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  matrix_residual_fold(c, c->norm_host, mr);
  return 0;
}

int agx_phase_implicit_update(agx_ctx* c, int mm, double* l2, agx_linf* linf) {
  return update_pass(c, 2, mm, l2, linf);
}

// ---- halo -----------------------------------------------------------------
int agx_halo_swap_local(agx_ctx* c, int what) {
  if (what == AGX_HALO_UPDATE) x_changed(c);
  if (c->conns.empty()) return 0;
  Timer t(c, G_BC);
  if (c->halo_batch && c->halo_batch_sides > 0) {
    // level by level: every slice of the level, then every insert (halo_batch_plan)
    if (halo_batch_tables(c, what)) return 1;
    HaloSide* const* tab = c->halo_tab_dev[what][c->halo_set[what]];
    for (const auto& L : c->halo_levels) {
      const dim3 grid((unsigned)((L.nmax + 255) / 256), (unsigned)L.sides);
      hipLaunchKernelGGL(k_halo_gather_all, grid, dim3(256), 0, c->stream, tab[0] + L.first);
      hipLaunchKernelGGL(k_halo_scatter_all, grid, dim3(256), 0, c->stream, tab[1] + L.first);
    }
    HIPCHK(hipGetLastError());
    return 0;
  }
  for (auto& k : c->conns) {
    const agx_connection& cc = k.c;
    if (!(cc.rank[0] == c->rank && cc.rank[1] == c->rank)) continue;
    Block& b0 = c->blocks[cc.local_block[0]];
    Block& b1 = c->blocks[cc.local_block[1]];
    const long n0 = k.side[0].n, n1 = k.side[1].n;
    if (ensure_halo_buf(c, (n0 + n1) * AGX_NEQ)) return 1;
    double* buf0 = c->halo_buf;                 // what side 0 receives
    double* buf1 = c->halo_buf + n0 * AGX_NEQ;  // what side 1 receives
    // both slices are taken before either insert (multiArray3d.hpp:810-821)
    const long nmax = std::max(n0, n1);
    if (nmax > 0) {
      const bool z2 = halo_in_d2(b0, what);
      HaloSide g0{halo_planes(b1, what), z2 ? k.side[0].src2 : k.side[0].src, n0, buf0};
      HaloSide g1{halo_planes(b0, what), z2 ? k.side[1].src2 : k.side[1].src, n1, buf1};
      hipLaunchKernelGGL(k_halo_gather2, dim3((nmax + 255) / 256, 2), dim3(256), 0, c->stream,
                         g0, g1);
      HaloSide p0{halo_planes(b0, what), z2 ? k.side[0].dst2 : k.side[0].dst, n0, buf0};
      HaloSide p1{halo_planes(b1, what), z2 ? k.side[1].dst2 : k.side[1].dst, n1, buf1};
      hipLaunchKernelGGL(k_halo_scatter2, dim3((nmax + 255) / 256, 2), dim3(256), 0, c->stream,
                         p0, p1);
    }
  }
  HIPCHK(hipGetLastError());
  return 0;
}

static int my_side(const agx_ctx* c, const Conn& k) {
  if (k.c.rank[0] == c->rank && k.c.rank[1] != c->rank) return 0;
  if (k.c.rank[1] == c->rank && k.c.rank[0] != c->rank) return 1;
  return -1;
}
int64_t agx_halo_count(agx_ctx* c, int id, int what) {
  (void)what;
  if (id < 0 || id >= (int)c->conns.size()) return -1;
  const Conn& k = c->conns[id];
  const int s = my_side(c, k);
  if (s < 0) return 0;
  return (int64_t)AGX_NEQ * std::max(k.n_send, k.side[s].n);
}
int agx_halo_pack(agx_ctx* c, int id, int what, double* dev_buf) {
  if (id < 0 || id >= (int)c->conns.size()) return fail("bad connection id");
  Conn& k = c->conns[id];
  const int s = my_side(c, k);
  if (s < 0) return fail("connection %d is not remote", id);
  Block& b = c->blocks[k.c.local_block[s]];
  const long n = k.n_send;
  if (n) hipLaunchKernelGGL(k_halo_gather, dim3((n + 255) / 256), dim3(256), 0,
                            c->stream, halo_planes(b, what),
                            halo_in_d2(b, what) ? k.send_src2 : k.send_src, n, dev_buf);
  HIPCHK(hipGetLastError());
  return 0;
}
int agx_halo_unpack(agx_ctx* c, int id, int what, const double* dev_buf) {
  if (what == AGX_HALO_UPDATE) x_changed(c);
  c->ghosts_prefilled = false;
  if (id < 0 || id >= (int)c->conns.size()) return fail("bad connection id");
  Conn& k = c->conns[id];
  const int s = my_side(c, k);
  if (s < 0) return fail("connection %d is not remote", id);
  Block& b = c->blocks[k.c.local_block[s]];
  const long n = k.side[s].n;
  if (n) hipLaunchKernelGGL(k_halo_scatter, dim3((n + 255) / 256), dim3(256), 0,
                            c->stream, halo_planes(b, what),
                            halo_in_d2(b, what) ? k.side[s].dst2 : k.side[s].dst, n, dev_buf);
  HIPCHK(hipGetLastError());
  return 0;
}

// ---- multi-rank -------------------------------------------------------------
int agx_set_exchange(agx_ctx* c, const agx_exchange* ex) {
  if (!ex || !ex->swap || !ex->allgather || ex->nranks < 1)
    return fail("agx_set_exchange: swap, allgather and nranks are required");
  if (c->finalized) return fail("agx_set_exchange must precede agx_setup_finalize");
  c->ex = *ex;
  c->have_ex = true;
  return 0;
}

namespace {
// built-in transport: RCCL on the library's stream
int rccl_swap(void* user, int n, const agx_slab* slabs, void* stream) {
  agx_ctx* c = static_cast<agx_ctx*>(user);
  ncclResult_t r = ncclGroupStart();
  for (int q = 0; q < n && r == ncclSuccess; ++q) {
    r = ncclSend(slabs[q].send, (size_t)slabs[q].count, ncclDouble, slabs[q].peer, c->nccl,
                 (hipStream_t)stream);
    if (r == ncclSuccess)
      r = ncclRecv(slabs[q].recv, (size_t)slabs[q].count, ncclDouble, slabs[q].peer, c->nccl,
                   (hipStream_t)stream);
  }
  const ncclResult_t e = ncclGroupEnd();
  if (r == ncclSuccess) r = e;
  return r == ncclSuccess ? 0 : fail("RCCL halo exchange failed: %s", ncclGetErrorString(r));
}
int rccl_allgather(void* user, const void* send, void* recv, int64_t bytes, void* stream) {
  agx_ctx* c = static_cast<agx_ctx*>(user);
  const ncclResult_t r = ncclAllGather(send, recv, (size_t)bytes, ncclChar, c->nccl,
                                       (hipStream_t)stream);
  return r == ncclSuccess ? 0 : fail("RCCL all-gather failed: %s", ncclGetErrorString(r));
}
}  // namespace

int agx_rccl_unique_id(void* id128) {
  static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
  const ncclResult_t r = ncclGetUniqueId(static_cast<ncclUniqueId*>(id128));
  return r == ncclSuccess ? 0 : fail("ncclGetUniqueId: %s", ncclGetErrorString(r));
}
int agx_rccl_exchange_create(agx_ctx* c, const void* id128, int nranks, int rank) {
  if (c->finalized) return fail("agx_rccl_exchange_create must precede agx_setup_finalize");
  HIPCHK(hipSetDevice(c->device));
  ncclUniqueId id;
  memcpy(&id, id128, sizeof id);
  const ncclResult_t r = ncclCommInitRank(&c->nccl, nranks, id, rank);
  if (r != ncclSuccess) return fail("ncclCommInitRank: %s", ncclGetErrorString(r));
  agx_exchange ex;
  ex.user = c; ex.swap = rccl_swap; ex.allgather = rccl_allgather;
  ex.nranks = nranks; ex.host_buffers = 0;
  c->ex = ex;
  c->have_ex = true;
  return 0;
}

static int halo_exchange_remote(agx_ctx* c, int what);
// gridLevel::GetBoundaryConditions (state) / lusgs::Relax, dplur::Relax (update):
// local connections, then the slabs of connections to other ranks
int agx_halo_exchange(agx_ctx* c, int what) {
  if (what < AGX_HALO_STATE || what > AGX_HALO_TURB) return fail("bad halo selector %d", what);
  if (what == AGX_HALO_TURB && AGX_NEQ != 7) return fail("AGX_HALO_TURB: rans library only");
  if (what >= AGX_HALO_VELGRAD_A && what <= AGX_HALO_VELGRAD_B && !(c->sp.implicit && c->sp.block))
    return fail("velocity gradients are kept (and exchanged) for the block-matrix solvers only");
  if (agx_halo_swap_local(c, what)) return 1;
  return halo_exchange_remote(c, what);
}

// the slabs of connections to other ranks: pack, the transport's swap, unpack -- all on
// c->stream (dplur_sweep_overlapped points it at its second stream meanwhile)
static int halo_exchange_remote(agx_ctx* c, int what) {
  if (c->remote.empty()) return 0;
  if (!c->have_ex) return fail("connections to other ranks need an exchange (agx_set_exchange)");
  Timer t(c, G_BC);
  for (auto& r : c->remote)
    if (agx_halo_pack(c, r.cid, what, r.send)) return 1;
  if (c->ex.host_buffers) {
    for (auto& r : c->remote)
      HIPCHK(hipMemcpyAsync(r.hsend, r.send, sizeof(double) * r.count, hipMemcpyDeviceToHost,
                            c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
  }
  if (c->ex.swap(c->ex.user, (int)c->slabs.size(), c->slabs.data(), c->stream))
    return g_err[0] ? 1 : fail("the exchange's swap operation failed");
  if (c->ex.host_buffers)
    for (auto& r : c->remote)
      HIPCHK(hipMemcpyAsync(r.recv, r.hrecv, sizeof(double) * r.count, hipMemcpyHostToDevice,
                            c->stream));
  for (auto& r : c->remote)
    if (agx_halo_unpack(c, r.cid, what, r.recv)) return 1;
  return 0;
}

// One DPLUR / BDPLUR sweep of a rank with neighbours (dplur::Relax linearSolver.cpp:509-535:
// SwapUpdate, then DPLUR on every block) with the interior / boundary split: point Jacobi
// reads the previous x of a cell's six neighbours, so only the cells next to a block face with
// a connection to another rank see a ghost cell that is still travelling.  All other cells are
// relaxed on the context's stream while the slabs of the previous x are packed, swapped and
// unpacked on a second stream; the cells of those faces follow when both are done.  Bitwise
// the sequential form (the same cell function on two disjoint sets of cells).
static bool overlap_applicable(const agx_ctx* c) {
  return c->overlap && c->sp.implicit && !is_lusgs_solver(c) && c->have_ex && !c->remote.empty();
}
static int dplur_sweep_overlapped(agx_ctx* c) {
  if (!c->ov_stream) {
    HIPCHK(hipStreamCreateWithFlags(&c->ov_stream, hipStreamNonBlocking));
    HIPCHK(hipEventCreateWithFlags(&c->ov_ready, hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&c->ov_done, hipEventDisableTiming));
  }
  // connections inside the rank first: their ghost cells are operands of the shell as well
  if (agx_halo_swap_local(c, AGX_HALO_UPDATE)) return 1;
  HIPCHK(hipEventRecord(c->ov_ready, c->stream));
  x_changed(c);
  std::vector<int> waits(c->blocks.size(), 0);     // faces with a connection to another rank
  for (auto& r : c->remote) {
    const Conn& k = c->conns[r.cid];
    const int sd = my_side(c, k);
    waits[k.c.local_block[sd]] |= 1 << (k.c.boundary[sd] - 1);
  }
  {
    Timer t(c, G_SWEEP);
    for (size_t n = 0; n < c->blocks.size(); ++n) {
      BlockDev bs = c->blocks[n].d;      // x and xold in the roles they take in this sweep
      for (int e = 0; e < AGX_NEQ; ++e) std::swap(bs.x[e], bs.xold[e]);
      AGX_BY_MODE(c->sp, k_dplur, cell_grid(bs, CELL_BLOCK), CELL_BLOCK, c->stream, bs, c->gas,
                  c->sp, waits[n]);
    }
    HIPCHK(hipGetLastError());
  }
  {
    struct OnStream {                    // pack / swap / unpack issue on c->stream
      agx_ctx* c; hipStream_t keep;
      OnStream(agx_ctx* c_, hipStream_t s) : c(c_), keep(c_->stream) { c->stream = s; }
      ~OnStream() { c->stream = keep; }
    } on(c, c->ov_stream);
    HIPCHK(hipStreamWaitEvent(c->ov_stream, c->ov_ready, 0));
    if (halo_exchange_remote(c, AGX_HALO_UPDATE)) return 1;
    HIPCHK(hipEventRecord(c->ov_done, c->ov_stream));
  }
  HIPCHK(hipStreamWaitEvent(c->stream, c->ov_done, 0));
  {
    Timer t(c, G_SWEEP);
    for (size_t n = 0; n < c->blocks.size(); ++n) {
      BlockDev& b = c->blocks[n].d;
      for (int e = 0; e < AGX_NEQ; ++e) std::swap(b.x[e], b.xold[e]);
      if (!waits[n]) continue;
      const long nface = std::max({(long)b.ni * b.nj, (long)b.ni * b.nk, (long)b.nj * b.nk});
      AGX_BY_MODE(c->sp, k_dplur_shell, dim3((unsigned)((nface + 255) / 256), 6), dim3(256),
                  c->stream, b, c->gas, c->sp, waits[n]);
    }
    HIPCHK(hipGetLastError());
  }
  c->halo_set[AGX_HALO_UPDATE] ^= 1;     // (x and xold changed roles)
  return 0;
}

namespace {
// main.cpp:254-264 over the exchange: every rank ends up with the global norms
// `status`: this rank's local result of the iteration; a failure anywhere makes every
// rank return failure (the collectives are reached by all ranks either way)
int reduce_over_ranks(agx_ctx* c, double* l2, agx_linf* linf, double* matrix_resid,
                      const double* l2_in, int status) {
  typedef agx_ctx::NormRecord Rec;
  const int nr = c->ex.nranks;
  Rec& mine = c->rec_host[0];
  memset(&mine, 0, sizeof mine);
  for (int e = 0; e < AGX_NEQ; ++e) mine.l2[e] = l2[e] - l2_in[e];   // this call's local sums
  mine.mres = *matrix_resid;
  mine.linf = linf->linf; mine.block = linf->block; mine.i = linf->i; mine.j = linf->j;
  mine.k = linf->k; mine.eqn = linf->eqn;
  mine.status = status;
  Rec* all = c->rec_host + 1;
  if (c->ex.host_buffers) {
    if (c->ex.allgather(c->ex.user, &mine, all, (int64_t)sizeof(Rec), c->stream))
      return g_err[0] ? 1 : fail("the exchange's allgather operation failed");
  } else {
    HIPCHK(hipMemcpyAsync(c->rec_dev, &mine, sizeof(Rec), hipMemcpyHostToDevice, c->stream));
    if (c->ex.allgather(c->ex.user, c->rec_dev, c->rec_dev + 1, (int64_t)sizeof(Rec), c->stream))
      return 1;
    HIPCHK(hipMemcpyAsync(all, c->rec_dev + 1, sizeof(Rec) * nr, hipMemcpyDeviceToHost,
                          c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
  }
  for (int r = 0; r < nr; ++r)
    if (all[r].status != 0) {
      if (status != 0) return status;      // g_err holds this rank's own message
      return fail("agx_iterate: rank %d failed (status %d); its agx_last_error has the cause",
                  r, all[r].status);
    }
  // fold in rank order: the same result on every rank
  double mres = 0.0;
  for (int e = 0; e < AGX_NEQ; ++e) l2[e] = l2_in[e];
  for (int r = 0; r < nr; ++r) {
    for (int e = 0; e < AGX_NEQ; ++e) l2[e] += all[r].l2[e];
    mres += all[r].mres;
    if (all[r].linf > linf->linf) {
      linf->linf = all[r].linf; linf->block = all[r].block; linf->i = all[r].i;
      linf->j = all[r].j; linf->k = all[r].k; linf->eqn = all[r].eqn;
    }
  }
  *matrix_resid = mres;
  return 0;
}
}  // namespace

// ---- mgSolution::Iterate (mgSolution.cpp:246-269) ---------------------------
int agx_iterate(agx_ctx* c, int mm, double cfl, double* l2, agx_linf* linf,
                double* matrix_resid) {
  if (!c->finalized) return fail("agx_setup_finalize has not been called");
  if (!c->remote.empty() && !c->have_ex)
    return fail("agx_iterate: connections to other ranks need an exchange "
                "(agx_set_exchange / agx_rccl_exchange_create) or the phase API");
  double l2_in[AGX_NEQ];
  for (int e = 0; e < AGX_NEQ; ++e) l2_in[e] = l2[e];
  // A local failure (a singular block, the sweep's spin limit, a refused phase) must not
  // leave the other ranks waiting in a collective: from the first failure on the compute
  // phases are skipped, the exchanges and the final all-gather are still reached, and the
  // status word of the norm record fails the call on every rank.
  int st = 0;
  char first_err[sizeof g_err] = "";
  const bool collective = c->have_ex && c->ex.nranks > 1;
  auto phase = [&](int r) {
    if (r && !st) { st = r; memcpy(first_err, g_err, sizeof g_err); }
  };
#define AGX_PHASE(call) do { if (!st) phase(call); } while (0)
  // an exchange runs even after a local failure when other ranks take part in it; a
  // failure of the exchange itself cannot be hidden from the peers and ends the call
#define AGX_XCHG(what)                                              \
  do {                                                              \
    if (!st || collective) {                                        \
      const int r_ = agx_halo_exchange(c, what);                    \
      if (r_ && st) { memcpy(g_err, first_err, sizeof g_err); return st; } \
      if (r_) return r_;                                            \
    }                                                               \
  } while (0)
  // gridLevel::GetBoundaryConditions gridLevel.cpp:287-319 (already done behind the
  // previous call's norm read-back unless something touched the state since)
  if (!c->ghosts_prefilled && fill_ghosts(c)) return 1;
  c->ghosts_prefilled = false;
  struct Scope { agx_ctx* c; ~Scope() { c->in_iterate = false; } } scope{c};
  c->in_iterate = true;
  // gridLevel::CalcResidual :372-400 + CalcTimeStep :240-247
  AGX_PHASE(agx_phase_residual(c, mm, cfl));
  *matrix_resid = 0.0;
  if (c->sp.implicit) {
    // gridLevel::SwapEddyViscAndGradients gridLevel.cpp:386-388 (read by the
    // off-diagonal terms of the block-matrix solvers only)
    if (c->sp.block && c->sp.viscous && !c->conns.empty()) {
      AGX_XCHG(AGX_HALO_VELGRAD_A);
      AGX_XCHG(AGX_HALO_VELGRAD_B);
    }
    // ... and of eddyViscosity_, f1_, f2_ (SwapTurbVars :389-392), rans
    if (AGX_NEQ == 7 && !c->conns.empty()) AGX_XCHG(AGX_HALO_TURB);
    // mgSolution::ImplicitUpdate :209-244; lusgs::Relax linearSolver.cpp:430-470;
    // dplur::Relax :509-535
    AGX_PHASE(agx_phase_implicit_begin(c));
    for (int s = 0; s < c->cfg.matrix_sweeps; ++s) {
      if (!st && overlap_applicable(c)) {
        // (the exchange is part of it: its failure ends the call like AGX_XCHG's)
        const int r_ = dplur_sweep_overlapped(c);
        if (r_) return r_;
        continue;
      }
      AGX_XCHG(AGX_HALO_UPDATE);
      AGX_PHASE(agx_phase_relax_forward(c, s));
      if (is_lusgs_solver(c)) {
        AGX_XCHG(AGX_HALO_UPDATE);
        AGX_PHASE(agx_phase_relax_backward(c, s));
      }
    }
    AGX_XCHG(AGX_HALO_UPDATE);
    AGX_PHASE(agx_phase_matrix_residual(c, matrix_resid));
    AGX_PHASE(agx_phase_implicit_update(c, mm, l2, linf));
    if (!st && c->mres_deferred)      // came back with the update's norms
      matrix_residual_fold(c, c->norm_host + c->blocks.size(), matrix_resid);
    c->mres_deferred = false;
  } else {
    AGX_PHASE(agx_phase_explicit_update(c, mm, l2, linf));
  }
#undef AGX_PHASE
#undef AGX_XCHG
  // the update queues the next call's ghost fill -- an exchange -- behind its norms
  if (st && collective && c->eager_ghosts) (void)fill_ghosts(c);
  if (st) memcpy(g_err, first_err, sizeof g_err);
  if (c->have_ex) return reduce_over_ranks(c, l2, linf, matrix_resid, l2_in, st);
  return st;
}

// ---- measurement ------------------------------------------------------------
int agx_timing_enable(agx_ctx* c, int on) {
  if (!on) resolve_timing(c);
  c->timing = on != 0;
  return 0;
}
int agx_timing_get(agx_ctx* c, int group, double* avg_ms, int64_t* launches) {
  if (group < 0 || group >= G_NGROUP) return fail("bad timing group");
  resolve_timing(c);
  *launches = c->t_n[group];
  *avg_ms = c->t_n[group] ? c->t_ms[group] / c->t_n[group] : 0.0;
  return 0;
}
int agx_timing_reset(agx_ctx* c) {
  resolve_timing(c);
  for (int g = 0; g < G_NGROUP; ++g) { c->t_ms[g] = 0.0; c->t_n[g] = 0; }
  return 0;
}
int agx_sync(agx_ctx* c) {
  HIPCHK(hipStreamSynchronize(c->stream));
  return 0;
}

}  // extern "C"
