/* aither_gfx950.h -- C-ABI of libaither_gfx950.so (5 equations: euler,
 * navierStokes) and of libaither_gfx950_rans.so (7 equations: rans with k-omega
 * SST 2003, SST-DES or k-omega Wilcox 2006, low-Re walls or wall functions; the same sources built with -DAGX_NEQ=7, the same entry points).
 *
 * MI355X (gfx950) implementation of AITHER's per-iteration hot path:
 * ghost-cell fill, face reconstruction, inviscid/viscous fluxes, time step,
 * explicit update and the LU-SGS / DPLUR implicit sweeps.
 *
 * The library stands behind the calls that the reference's
 *   mgSolution::Iterate            (src/mgSolution.cpp:246-269)
 * makes on its finest gridLevel, plus the three data-movement points around
 * it (setup after main.cpp:163-203, StoreOldSolution main.cpp:239, and
 * GetFinestGridLevel main.cpp:282).  All arrays crossing this boundary use the
 * reference's own host layout (multiArray3d, include/multiArray3d.hpp:104-113):
 * AoS, block-size doubles per cell, i fastest, ghost-inclusive dimensions
 * (n + 2*ng).  The library owns device-resident SoA copies and never aliases
 * host memory.
 *
 * Error convention: every entry point returns 0 on success, non-zero on
 * failure; agx_last_error() returns a message (the reference prints to cerr
 * and calls exit(EXIT_FAILURE); the adapter in INTEGRATION.md keeps that).
 * Not thread-safe: one context per process per device (the reference is
 * single-threaded per MPI rank).
 */
#ifndef AITHER_GFX950_H
#define AITHER_GFX950_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
/* The libraries are built with -fvisibility=hidden: only the agx_* entry points below are
 * exported.  libaither_gfx950.so (5 equations) and libaither_gfx950_rans.so (7 equations)
 * are the SAME sources compiled for two state layouts, so their internals must never bind
 * to each other; with the internals hidden the two can share a process (each dlopen'ed
 * RTLD_LOCAL and called through dlsym, since both export the same agx_* names). */
#if defined(__GNUC__) || defined(__clang__)
#pragma GCC visibility push(default)
#endif

/* ---- enumerations (resolved once from the reference's std::string input
 *      options, include/input.hpp:48-291) -------------------------------- */
enum { AGX_RECON_CONSTANT = 0, AGX_RECON_MUSCL = 1, AGX_RECON_WENO = 2,
       AGX_RECON_WENOZ = 3 };                 /* input.cpp:276-303 */
enum { AGX_LIMITER_NONE = 0, AGX_LIMITER_VANALBADA = 1,
       AGX_LIMITER_MINMOD = 2 };              /* limiter.cpp:24-54 */
enum { AGX_FLUX_ROE = 0, AGX_FLUX_AUSM = 1 }; /* inviscidFlux.hpp:484-507 */
enum { AGX_TIME_EXPLICIT_EULER = 0, AGX_TIME_RK4 = 1,
       AGX_TIME_IMPLICIT_EULER = 2, AGX_TIME_CRANK_NICHOLSON = 3,
       AGX_TIME_BDF2 = 4 };                   /* input.cpp:259-275 */
enum { AGX_SOLVER_LUSGS = 0, AGX_SOLVER_DPLUR = 1,
       AGX_SOLVER_BLUSGS = 2, AGX_SOLVER_BDPLUR = 3 }; /* input.cpp:843-858 */
enum { AGX_EQN_EULER = 0, AGX_EQN_NAVIER_STOKES = 1,
       AGX_EQN_RANS = 2 };                    /* input::equationSet_, input.cpp:1033-1060 */
enum { AGX_JACOBIAN_RUSANOV = 0,
       AGX_JACOBIAN_APPROX_ROE = 1 };         /* input::InvFluxJac, fluxJacobian.cpp:196-238 */
enum { AGX_VISC_RECON_CENTRAL = 0,
       AGX_VISC_RECON_CENTRAL_4TH = 1 };      /* reconstruction.hpp:315-379 */
enum { AGX_TURB_NONE = 0, AGX_TURB_SST2003 = 1, AGX_TURB_KW_WILCOX2006 = 2,
       AGX_TURB_SST_DES = 3,
       AGX_TURB_WALE = 4 /* named for completeness: agx_config_set REFUSES it (not built) */
     };                                       /* turbulence.hpp */

/* boundary condition types, ghostStates.cpp:62-689 */
enum { AGX_BC_SLIPWALL = 0, AGX_BC_VISCOUSWALL = 1, AGX_BC_CHARACTERISTIC = 2,
       AGX_BC_INLET = 3, AGX_BC_SUPERSONIC_INFLOW = 4,
       AGX_BC_SUPERSONIC_OUTFLOW = 5, AGX_BC_STAGNATION_INLET = 6,
       AGX_BC_PRESSURE_OUTLET = 7, AGX_BC_INTERBLOCK = 8,
       AGX_BC_PERIODIC = 9 };

/* fields that can be downloaded (procBlock members, include/procBlock.hpp:60-110) */
enum { AGX_FIELD_STATE = 0,      /* state_      nEq, with ghosts           */
       AGX_FIELD_RESIDUAL = 1,   /* residual_   nEq, no ghosts             */
       AGX_FIELD_DT = 2,         /* dt_         1,   no ghosts             */
       AGX_FIELD_SPEC_RADIUS = 3,/* specRadius_ 1,   no ghosts (flow part) */
       AGX_FIELD_CONS_N = 4,     /* consVarsN_  nEq, no ghosts             */
       AGX_FIELD_UPDATE = 5,     /* linearSolver x_ nEq, with ghosts       */
       AGX_FIELD_DIAGONAL = 6,   /* linearSolver a_ (scalar) 1, no ghosts  */
       AGX_FIELD_TEMPERATURE = 7,/* temperature_ 1,  with ghosts           */
       AGX_FIELD_VISCOSITY = 8,  /* viscosity_  1,   with ghosts           */
       AGX_FIELD_CONS_NM1 = 9,   /* consVarsNm1_ nEq, no ghosts            */
       /* cell-centre gradients (velocityGrad_, temperatureGrad_, densityGrad_,
        * pressureGrad_; procBlock.cpp:1397-1449, :5950-5954): the mean of the six
        * Green-Gauss face gradients of the cell, formed ON DEMAND from the state
        * the device holds (they are not kept between iterations) with the ghost
        * cells the next residual would see: the inviscid fill of that state, then
        * the viscous-wall fill (ghost cells of connections to other ranks as last
        * exchanged); no ghosts.
        * VEL_GRAD: 9 per cell, [3 r + c] = d(velocity c)/d(x_r) (tensor.hpp) */
       AGX_FIELD_VEL_GRAD = 10, AGX_FIELD_TEMP_GRAD = 11,
       AGX_FIELD_DENS_GRAD = 12, AGX_FIELD_PRESS_GRAD = 13
};

/* variables of a function file, WriteFunFile (output.cpp:235-407): formed and
 * re-dimensionalised ON THE DEVICE by agx_output_pack, so an output step moves what the
 * file holds instead of the whole state (GetFinestGridLevel, main.cpp:282).  Gradients:
 * d/dx, d/dy, d/dz; VELGRAD in the reference's order ux vx wx uy vy wy uz vz wz
 * (tensor XX XY XZ YX ...).  The gradients are those of the state the device holds when
 * the call is made, with the ghost cells the next residual would see (AGX_FIELD_VEL_GRAD;
 * the reference writes the ones its last residual accumulated, procBlock.cpp:1397-1449,
 * i.e. of the state before the last update); eddy viscosity, f1, f2 and the residuals
 * are the last residual's, as in the reference. */
enum { AGX_OUT_DENSITY = 0, AGX_OUT_VEL_X, AGX_OUT_VEL_Y, AGX_OUT_VEL_Z, AGX_OUT_PRESSURE,
       AGX_OUT_MACH, AGX_OUT_SOS, AGX_OUT_DT, AGX_OUT_TEMPERATURE, AGX_OUT_ENERGY,
       AGX_OUT_ENTHALPY, AGX_OUT_CP, AGX_OUT_CV, AGX_OUT_RANK, AGX_OUT_GLOBAL_POSITION,
       AGX_OUT_VISCOSITY_RATIO, AGX_OUT_TURB_VISCOSITY, AGX_OUT_VISCOSITY, AGX_OUT_TKE,
       AGX_OUT_SDR, AGX_OUT_F1, AGX_OUT_F2, AGX_OUT_WALL_DISTANCE,
       AGX_OUT_VELGRAD = 23,      /* .. 31: nine entries                          */
       AGX_OUT_TEMPGRAD = 32,     /* .. 34                                        */
       AGX_OUT_DENSGRAD = 35, AGX_OUT_PRESSGRAD = 38,
       AGX_OUT_TKEGRAD = 41, AGX_OUT_OMEGAGRAD = 44,   /* rans library             */
       AGX_OUT_RESID = 47,        /* .. 53: mass, mom_x, mom_y, mom_z, energy, tke, sdr */
       AGX_OUT_COUNT = 54 };

/* what a halo exchange carries (gridLevel.cpp:299-313, utility.cpp:400-423) */
enum { AGX_HALO_STATE = 0, AGX_HALO_UPDATE = 1,
       /* velocityGrad_ of the cells across connection surfaces, swapped after the
        * residual (gridLevel.cpp:343-368, :386-388) and read by the off-diagonal
        * terms of the block-matrix solvers with viscous terms; two halves of the
        * nine components, each in slabs of the usual five slots per cell */
       AGX_HALO_VELGRAD_A = 2, AGX_HALO_VELGRAD_B = 3,
       /* rans: eddyViscosity_, f1_, f2_ of the cells across connection surfaces
        * (SwapEddyViscAndGradients / SwapTurbVars, gridLevel.cpp:386-392); read by the
        * off-diagonal terms.  libaither_gfx950_rans.so (and the oracle); the 5-equation
        * library refuses it. */
       AGX_HALO_TURB = 4 };

/* ---- plain-old-data descriptors --------------------------------------- */

/* single-species calorically-perfect ideal gas + Sutherland transport.
 * Values are the already-nondimensional ones the reference holds after
 * input::NondimensionalizeFluid (fluid.cpp:83-97, eos.cpp:26-36,
 * thermodynamic.cpp:27-41, transport.cpp:31-69). */
typedef struct agx_gas {
  double gas_constant;   /* idealGas::gasConst_[0] (nondimensional R)      */
  double n;              /* fluid::N(): cv = n R, cp = (n+1) R             */
  double heat_of_formation; /* caloricallyPerfect::hf_[0] (nondimensional) */
  double visc_c1, visc_s;   /* Sutherland viscosity C1 [kg/(m s K^.5)], S [K] */
  double cond_c1, cond_s;   /* Sutherland conductivity C1, S               */
  double t_ref;          /* referenceTemperature [K]                       */
  double rho_ref;        /* referenceDensity [kg/m^3]                      */
  double l_ref;          /* referenceLength [m]                            */
  double a_ref;          /* reference speed of sound [m/s] (input.cpp:608-613) */
} agx_gas;

/* solver configuration; one per context */
typedef struct agx_config {
  int32_t n_eq;              /* 5, or 7 = [rho, u, v, w, p, k, omega] in the rans library */
  int32_t n_ghost;           /* input::NumberGhostLayers (input.cpp:1127)  */
  int32_t recon;             /* AGX_RECON_*                                */
  int32_t limiter;           /* AGX_LIMITER_*                              */
  int32_t inviscid_flux;     /* AGX_FLUX_*                                 */
  int32_t is_viscous;        /* input::IsViscous                           */
  int32_t time_integration;  /* AGX_TIME_*                                 */
  int32_t matrix_solver;     /* AGX_SOLVER_*                               */
  int32_t matrix_sweeps;     /* input::MatrixSweeps                        */
  int32_t nonlinear_iterations; /* input::NonlinearIterations              */
  /* the rest of what the path depends on.  agx_config_set REFUSES every value
   * this build does not implement (it never substitutes another scheme): */
  int32_t equation_set;      /* AGX_EQN_*; must agree with is_viscous / n_eq */
  int32_t inv_flux_jacobian; /* AGX_JACOBIAN_* (input::InvFluxJac)          */
  int32_t viscous_recon;     /* AGX_VISC_RECON_* (input::ViscousFaceReconstruction) */
  int32_t turbulence_model;  /* AGX_TURB_* (input::TurbulenceModel)         */
  double kappa;              /* MUSCL kappa (input.cpp:277-292)            */
  double theta, zeta;        /* Beam-Warming (input.cpp:261-270)           */
  double matrix_relaxation;  /* input::MatrixRelaxation                    */
  double dual_time_cfl;      /* input::DualTimeCFL, <= 0: off              */
  double dt_nondim;          /* Dt * aRef / lRef (procBlock.cpp:808); <= 0:
                                local time stepping from CFL               */
  double viscous_cfl_coeff;  /* input::ViscousCFLCoefficient (input.cpp:1110) */
  agx_gas gas;
} agx_config;

/* geometry of one block, host AoS arrays with ghosts (procBlock.hpp:65-90).
 * Face-area arrays hold unitVec3dMag = {nx, ny, nz, |A|} per face
 * (vector3d.hpp:125-190).  Dimensions with G = 2*ng:
 *   farea_i: (ni+1+G, nj+G,   nk+G)   x 4
 *   farea_j: (ni+G,   nj+1+G, nk+G)   x 4
 *   farea_k: (ni+G,   nj+G,   nk+1+G) x 4
 *   vol, width_i/j/k, wall_dist: (ni+G, nj+G, nk+G) x 1
 *   center:  (ni+G, nj+G, nk+G) x 3                                        */
typedef struct agx_block_geom {
  int32_t ni, nj, nk;        /* physical cells                             */
  int32_t ng;                /* ghost layers                               */
  int32_t parent_block;      /* procBlock::ParentBlock (for L-inf report)  */
  int32_t global_pos;        /* procBlock::GlobalPos                       */
  const double *farea_i, *farea_j, *farea_k;
  const double *vol;
  const double *center;
  const double *width_i, *width_j, *width_k;   /* cellWidthI/J/K_          */
  const double *wall_dist;   /* may be NULL for inviscid                   */
} agx_block_geom;

/* boundary-state data for one surface: the union of the fields the
 * reference's inputState subclasses hold (include/inputStates.hpp:112-420),
 * already nondimensional. */
typedef struct agx_bc_state {
  double pressure, density;
  double velocity[3];
  double stagnation_pressure, stagnation_temperature;
  double direction[3];
  double wall_temperature;   /* viscousWall isothermal                     */
  double wall_heat_flux;     /* viscousWall constant heat flux             */
  double length_scale;       /* nonreflecting inlet / outlet               */
  int32_t is_isothermal, is_heat_flux, is_nonreflecting, pad_;
  /* farfield turbulence of the state (ApplyFarfieldTurbBC, primitive.cpp:83-98):
   * turbulenceIntensity and eddyViscosityRatio; read by rans runs only */
  double turb_intensity, eddy_visc_ratio;
  /* viscousWall(wallTreatment=wallLaw): von Karman constant and wall constant
   * (inputStates.hpp:343-345); rans library; adiabatic, isothermal or heat-flux wall */
  double von_karman, wall_constant;
  int32_t is_wall_law, pad2_;
} agx_bc_state;

/* one boundarySurface (boundaryConditions.hpp:55-150): index ranges are the
 * reference's node-style ranges exactly as read from the .inp file. */
typedef struct agx_bc_surface {
  int32_t bc_type;           /* AGX_BC_*                                   */
  int32_t imin, imax, jmin, jmax, kmin, kmax;
  int32_t tag;
  agx_bc_state state;
} agx_bc_surface;

/* POD mirror of class connection (boundaryConditions.hpp:371-433) */
typedef struct agx_connection {
  int32_t rank[2];
  int32_t block[2];          /* global block ids                           */
  int32_t local_block[2];    /* ids returned by agx_block_create on that rank */
  int32_t boundary[2];       /* surface type 1..6                          */
  int32_t d1_start[2], d1_end[2], d2_start[2], d2_end[2];
  int32_t const_surf[2];
  int32_t patch_border[8];
  int32_t orientation;       /* 1..8 (boundaryConditions.cpp:653-727)      */
  int32_t is_interblock;     /* 0 for periodic                             */
} agx_connection;

/* L-infinity residual record, class resid (include/resid.hpp) */
typedef struct agx_linf {
  double linf;
  int32_t block, i, j, k, eqn;   /* eqn is 1-based (procBlock.cpp:864)     */
  int32_t pad_;
} agx_linf;

typedef struct agx_ctx agx_ctx;

/* ---- context ----------------------------------------------------------- */
const char *agx_last_error(void);
const char *agx_version(void);
/* rank: this process's rank in the job (matches agx_connection.rank[]) */
int agx_ctx_create(int device, int rank, agx_ctx **out);
void agx_ctx_destroy(agx_ctx *ctx);
/* run all library work on this hipStream_t (NULL = default stream); may be called at any
 * time: the stream used so far is drained first */
int agx_ctx_set_stream(agx_ctx *ctx, void *hip_stream);
int agx_config_set(agx_ctx *ctx, const agx_config *cfg);

/* ---- setup: after SendFinestGridLevel / AuxillaryAndWidths / SwapWallDist
 *      (main.cpp:163-203) ------------------------------------------------ */
int agx_block_create(agx_ctx *ctx, const agx_block_geom *geom, int *block_id);
int agx_block_set_bcs(agx_ctx *ctx, int block_id, int n_surfaces,
                      const agx_bc_surface *surfaces);
int agx_conn_create(agx_ctx *ctx, const agx_connection *conn, int *conn_id);
/* finish setup: build index maps, hyperplane order, allocate work arrays */
int agx_setup_finalize(agx_ctx *ctx);

/* ---- state movement ---------------------------------------------------- */
/* replaces gridLevel ctor state init / ReadRestart (gridLevel.cpp:55,63-65) */
int agx_state_upload(agx_ctx *ctx, int block_id, const double *state_aos);
/* replaces the pack in GetFinestGridLevel (procBlock.cpp:4491-4660) */
int agx_field_download(agx_ctx *ctx, int block_id, int field, double *out_aos);
int agx_field_upload(agx_ctx *ctx, int block_id, int field, const double *in_aos);

/* ---- per time step: mgSolution::StoreOldSolution (mgSolution.cpp:103-114):
 *      consVarsN_ <- ConsVars(state_), and consVarsNm1_ <- consVarsN_ when
 *      also_nm1 != 0 (first bdf2 step) ----------------------------------- */
int agx_store_time_n(agx_ctx *ctx, int also_nm1);

/* ---- per nonlinear iteration: mgSolution::Iterate (mgSolution.cpp:246-269).
 *  mm   -- nonlinear iteration index (RK stage for rk4)
 *  cfl  -- input::CFL() after input::CalcCFL(nn) (input.cpp:637-639)
 *  l2   -- n_eq doubles, ACCUMULATED into (sum of residual^2, procBlock.cpp:858)
 *  linf -- updated only when a larger signed residual is found (:863-866)
 *  matrix_resid -- returns sum(matrixResid^2)/count (mgSolution.cpp:198-206)
 * With connections to other ranks: install an exchange first (agx_set_exchange /
 * agx_rccl_exchange_create, below), or drive the phases and exchange the slabs
 * yourself. */
int agx_iterate(agx_ctx *ctx, int mm, double cfl, double *l2, agx_linf *linf,
                double *matrix_resid);

/* ---- phases of one iteration, for multi-process runs --------------------
 * Order (explicit):  bc_faces, [halo STATE], bc_edges, residual, explicit_update
 * Order (implicit):  bc_faces, [halo STATE], bc_edges, residual, implicit_begin,
 *                    sweeps x { [halo UPDATE], relax_forward, [halo UPDATE],
 *                    relax_backward }   (LU-SGS, linearSolver.cpp:430-470)
 *                    or sweeps x { [halo UPDATE], relax_forward } (DPLUR :509-535)
 *                    [halo UPDATE], matrix_residual, implicit_update          */
int agx_phase_bc_faces(agx_ctx *ctx);          /* AssignInviscidGhostCells  procBlock.cpp:2449 */
int agx_phase_bc_edges(agx_ctx *ctx);          /* AssignInviscidGhostCellsEdge procBlock.cpp:2565 */
/* mm: nonlinear iteration (RK stage) -- explicit inviscid runs fuse the stage
 * update into the residual kernel, explicit_update then only swaps buffers */
int agx_phase_residual(agx_ctx *ctx, int mm, double cfl); /* CalcResidualNoSource :6111 + CalcBlockTimeStep :798 */
int agx_phase_explicit_update(agx_ctx *ctx, int mm, double *l2, agx_linf *linf); /* UpdateBlock :826 */
int agx_phase_implicit_begin(agx_ctx *ctx);    /* InvertDiagonal + InitializeMatrixUpdate mgSolution.cpp:225-229 */
int agx_phase_relax_forward(agx_ctx *ctx, int sweep);  /* LUSGS_Forward :341 / DPLUR :473 */
int agx_phase_relax_backward(agx_ctx *ctx, int sweep); /* LUSGS_Backward :385 */
int agx_phase_matrix_residual(agx_ctx *ctx, double *matrix_resid); /* linearSolver::Residual :92 */
int agx_phase_implicit_update(agx_ctx *ctx, int mm, double *l2, agx_linf *linf); /* UpdateBlocks gridLevel.cpp:418 */

/* ---- output: the payloads of the reference's files, packed on the device ----
 * WriteFunFile (output.cpp:209-437): `nvar` variables (AGX_OUT_*, in the caller's order --
 * the reference writes its std::set alphabetically) of one block, physical cells, i
 * fastest, variable by variable: out[v * ni*nj*nk + cell], dimensional as the reference
 * writes them (reference quantities from agx_config.gas).  Variables a build does not hold
 * (tke, sdr, eddy viscosity, f1, f2 and their gradients in the 5-equation library) come
 * out as the reference's laminar values (0), viscosity in inviscid runs is refused. */
int agx_output_pack(agx_ctx *ctx, int block, int nvar, const int32_t *vars, double *out);
/* WriteRestart (output.cpp:651-752), the payload of one block: cell by cell (i fastest)
 * n_eq + 1 dimensional values -- density, velocity, pressure, [tke, sdr,] mass fraction of
 * the single species (1).  which = 0: the state; which = 1: consVarsNm1 (the second
 * solution of multilevel time integration: conserved variables, :700-750). */
int agx_restart_pack(agx_ctx *ctx, int block, int which, double *out);

/* ---- set-up: the volume-sized parts of the reference's grid set-up -------
 * plot3dBlock::Volume / Centroid / FaceAreaI,J,K / FaceCenterI,J,K (plot3d.cpp:35-360) of
 * one block from its node coordinates nodes[nk+1][nj+1][ni+1][3] (i fastest, as
 * plot3dBlock holds them).  Outputs in the reference's layout, physical cells / faces only
 * (ghost geometry -- PadWithGhosts, AssignGhostCellsGeom, SwapGeomSlice: surface-sized
 * work -- stays with the caller): vol[nk][nj][ni], center[nk][nj][ni][3],
 * farea_i[nk][nj][ni+1][4] = {unit normal, |A|}, fcenter_i[nk][nj][ni+1][3], the j- and
 * k-face arrays with the extra entry in their own direction.  Any output may be NULL.
 * Host pointers; needs neither a configuration nor blocks. */
int agx_plot3d_metrics(agx_ctx *ctx, int ni, int nj, int nk, const double *nodes,
                       double *vol, double *center, double *farea_i, double *farea_j,
                       double *farea_k, double *fcenter_i, double *fcenter_j,
                       double *fcenter_k);
/* kdtree::NearestNeighbor over the viscous-wall face centres (kdtree.cpp:123-225 as
 * main.cpp:191-203 / procBlock::CalcWallDistance procBlock.cpp:6030-6042 use it): for
 * each of ncell points (x, y, z) the distance to the nearest of nwall points.  On the
 * device an exhaustive search tiled through LDS (no tree: 16.7 M cells x 65 k wall faces is
 * a fraction of a second).  The ghost-cell rule of CalcWallDistance (:6044-6107) and
 * SwapWallDist are surface-sized and stay with the caller.  Host pointers. */
int agx_nearest_wall_distance(agx_ctx *ctx, int64_t ncell, const double *cell_centres,
                              int64_t nwall, const double *wall_points, double *dist);

/* ---- geometric multigrid ---------------------------------------------
 * A grid level is a context of its own (gridLevel): same configuration, the coarsened
 * blocks and surfaces of procBlock::GetCoarseMeshAndBCs (procBlock.cpp:6471-6603; host
 * set-up, aither_amd/case/multigrid.py).  These calls move data between block `blk` of a
 * fine context and the same block of the next coarser one, and give a level what
 * mgSolution::CycleAtLevel (mgSolution.cpp:160-205) needs beyond the phases above.  Both
 * contexts live on the same device.  The forcing term is carried by all four relaxations of
 * both libraries (DPLUR, BDPLUR, BLU-SGS, scalar LU-SGS -- in the 5-equation library on its
 * diagonal-ordered production path and in its hyperplane form); the turbulence equations of
 * the 7-equation library are restricted, forced and prolonged like the flow equations, as
 * the reference's transfers on varArray do.  `to_coarse` [nk][nj][ni][3] (int32, host):
 * the coarse cell (i, j, k) of every physical cell of the fine block. */
enum { AGX_MG_STATE = 0, AGX_MG_UPDATE = 1, AGX_MG_FORCING = 2 };
/* BlockRestriction (procBlock.hpp:636-690) of
 *   AGX_MG_STATE    the primitive state, volume weighted (procBlock::Restriction :6847);
 *                   the coarse block's state is zeroed first, ghost cells included
 *   AGX_MG_UPDATE   the linear solver's x, volume weighted (linearSolver.cpp:212-222; the
 *                   caller swaps the coarse x across connections afterwards)
 *   AGX_MG_FORCING  the fine level's matrix residual (the array of its last
 *                   agx_mg_matrix_residual), summed, plus A x - b of the coarse level
 *                   (gridLevel.cpp:568-590): the coarse level's forcing term
 * vol_fac [nk][nj][ni] (host): a fine cell's volume over that of its coarse cell; NULL for
 * AGX_MG_FORCING. */
int agx_mg_restrict(agx_ctx *fine, agx_ctx *coarse, int blk, int what,
                    const int32_t *to_coarse, const double *vol_fac);
/* linearSolver::Residual (:92-109): forcing - (A x - b) of every cell, kept as an array
 * (what AGX_MG_FORCING restricts), and its mean square as mgSolution::CycleAtLevel forms
 * it (:196-204).  The caller has swapped x across connections. */
int agx_mg_matrix_residual(agx_ctx *ctx, double *mean_square);
/* gridLevel::InvertDiagonal (gridLevel.cpp:402-406) WITHOUT InitializeMatrixUpdate: a
 * coarse level's x is the restricted one */
int agx_mg_invert_diagonal(agx_ctx *ctx);
/* gridLevel::ResetDiagonal (gridLevel.cpp:408-412) of a coarse level when an iteration
 * ends (mgSolution.cpp:236-239).  The residual of a coarse level ADDS its spectral radii to
 * the diagonal, as the reference does: a level restricted to twice within a W cycle keeps
 * what its first visit left (the reference's truth contains that). */
int agx_mg_reset_diagonal(agx_ctx *ctx);
/* coarseDu = x (mgSolution.cpp:183), kept in the context for agx_mg_prolong */
int agx_mg_save_update(agx_ctx *ctx);
/* SubtractFromUpdate + Prolongation (mgSolution.cpp:189-192, gridLevel.cpp:597-611): the
 * coarse level's correction x - coarseDu at the nodes of its cells (ConvertCellToNode
 * utility.hpp:186-330, edges and corners by their own factors, no ghost cells), interpolated
 * to the fine cells with their trilinear coefficients (coeffs [nk][nj][ni][7], host;
 * TrilinearInterp utility.hpp:356-372) and added to the fine level's x.  The coarse x is
 * left as the correction. */
int agx_mg_prolong(agx_ctx *coarse, agx_ctx *fine, int blk, const int32_t *to_coarse,
                   const double *coeffs);

/* halo exchange (multiArray3d.hpp:790-873 SwapSliceLocal / SwapSliceParallel).
 * local: both sides on this rank. */
int agx_halo_swap_local(agx_ctx *ctx, int what);
/* remote: number of doubles in the packed slab of connection conn_id */
int64_t agx_halo_count(agx_ctx *ctx, int conn_id, int what);
/* pack this rank's side of the connection into dev_buf (device pointer) */
int agx_halo_pack(agx_ctx *ctx, int conn_id, int what, double *dev_buf);
/* unpack the partner's slab from dev_buf into this rank's ghost cells */
int agx_halo_unpack(agx_ctx *ctx, int conn_id, int what, const double *dev_buf);

/* ---- multi-rank runs: agx_iterate drives remote connections itself ---------
 * In the reference a rank exchanges ghost slabs with SwapSliceParallel /
 * PackSwapUnpackMPI (multiArray3d.hpp:830-866, utility.cpp:400-423: pack,
 * MPI_Sendrecv, unpack) and reduces the norms with GlobalReduceMPI / MPI_Reduce
 * (main.cpp:254-264).  Here the library packs and unpacks on the device and
 * leaves the wire to a table of two operations.  With an exchange installed
 * agx_iterate (and agx_halo_exchange) handle connections whose partner is on
 * another rank, and the norms agx_iterate returns are the GLOBAL ones on every
 * rank: l2 summed, linf the largest with its location, matrix_resid the sum of
 * the ranks' values (what MPI_SUM of main.cpp:259 forms) -- the adapter drops
 * its own reductions.
 *   host_buffers = 0: slab pointers are device memory and the operations are
 *     stream-ordered on hip_stream (the built-in RCCL transport);
 *   host_buffers = 1: the library stages the slabs through pinned host memory
 *     and calls the operations with host pointers after synchronising (an MPI,
 *     socket or shared-memory transport: MPI_Sendrecv / MPI_Allgather fit).  */
typedef struct agx_slab {
  int32_t peer;              /* rank of the partner                          */
  int32_t tag;               /* ordinal among the connections with this peer
                              * (creation order): the same on both sides     */
  int64_t count;             /* doubles to send and to receive               */
  double *send, *recv;
} agx_slab;
typedef struct agx_exchange {
  void *user;
  /* complete (or enqueue on hip_stream) send[i] -> peer[i], recv[i] <- peer[i]
   * for all n slabs of this rank; slabs come in connection-creation order */
  int (*swap)(void *user, int n, const agx_slab *slabs, void *hip_stream);
  /* every rank contributes `bytes` bytes at send; recv gets nranks * bytes,
   * in rank order */
  int (*allgather)(void *user, const void *send, void *recv, int64_t bytes,
                   void *hip_stream);
  int32_t nranks;
  int32_t host_buffers;
} agx_exchange;
int agx_set_exchange(agx_ctx *ctx, const agx_exchange *ex);
/* built-in transport: RCCL over xGMI, grouped ncclSend / ncclRecv per slab and
 * one ncclAllGather for the norms, all on the library's stream (no host sync
 * between pack and unpack).  id128: 128 bytes made by agx_rccl_unique_id on one
 * rank and handed to the others by the host (MPI_Bcast in the reference's
 * driver).  Call after agx_ctx_create, before agx_setup_finalize. */
int agx_rccl_unique_id(void *id128);
int agx_rccl_exchange_create(agx_ctx *ctx, const void *id128, int nranks, int rank);
/* one ghost exchange as gridLevel::GetBoundaryConditions / lusgs::Relax place it:
 * local connections + (with an exchange installed) remote ones */
int agx_halo_exchange(agx_ctx *ctx, int what);

/* ---- measurement helpers ----------------------------------------------- */
/* average duration [ms] of the named kernel group since the last reset,
 * measured with hipEvents on the library's stream; group: 0 = inviscid residual
 * (+ fused explicit stage), 1 = update + norms, 2 = ghost cells / halo,
 * 3 = LU-SGS / DPLUR sweeps, 4 = viscous residual, 5 = implicit begin
 * (diagonal, right-hand side), 6 = matrix residual */
int agx_timing_enable(agx_ctx *ctx, int on);
int agx_timing_get(agx_ctx *ctx, int group, double *avg_ms, int64_t *launches);
int agx_timing_reset(agx_ctx *ctx);
int agx_sync(agx_ctx *ctx);

#if defined(__GNUC__) || defined(__clang__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif /* AITHER_GFX950_H */
