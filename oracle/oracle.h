/* oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C, scalar, AoS exactly like the reference) of
 * AITHER's per-iteration residual + implicit-sweep path.  It exports the same
 * entry points as include/aither_gfx950.h with the prefix "ora_" so the parity
 * tests can drive both through identical calls.  Only tests/, bench.py's
 * cpu_baseline leg and __graft_entry__.smoke() may load it; the product
 * library never links, loads or falls back to it.
 *
 * Parity pinning: the reference cannot be built here by the allowed means (it
 * needs CMake-generated macros.hpp and an external MPI library), so this
 * restatement is pinned against the reference's own regression truths
 * (testCases/regressionTests.py:231-559) -- see tests/test_oracle_golden.py.
 */
#ifndef AITHER_ORACLE_H
#define AITHER_ORACLE_H
#include "../include/aither_gfx950.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ora_ctx ora_ctx;

const char *ora_last_error(void);
const char *ora_version(void);
int ora_ctx_create(int device, int rank, ora_ctx **out);
void ora_ctx_destroy(ora_ctx *ctx);
int ora_ctx_set_stream(ora_ctx *ctx, void *stream);
int ora_config_set(ora_ctx *ctx, const agx_config *cfg);
int ora_block_create(ora_ctx *ctx, const agx_block_geom *geom, int *block_id);
int ora_block_set_bcs(ora_ctx *ctx, int block_id, int n, const agx_bc_surface *s);
int ora_conn_create(ora_ctx *ctx, const agx_connection *conn, int *conn_id);
int ora_setup_finalize(ora_ctx *ctx);
int ora_state_upload(ora_ctx *ctx, int block_id, const double *state_aos);
int ora_field_download(ora_ctx *ctx, int block_id, int field, double *out);
int ora_field_upload(ora_ctx *ctx, int block_id, int field, const double *in);
int ora_output_pack(ora_ctx *ctx, int block_id, int nvar, const int32_t *vars, double *out);
int ora_restart_pack(ora_ctx *ctx, int block_id, int which, double *out);
int ora_plot3d_metrics(ora_ctx *ctx, int ni, int nj, int nk, const double *nodes,
                       double *vol, double *center, double *farea_i, double *farea_j,
                       double *farea_k, double *fcenter_i, double *fcenter_j,
                       double *fcenter_k);
int ora_nearest_wall_distance(ora_ctx *ctx, int64_t ncell, const double *cell_centres,
                              int64_t nwall, const double *wall_points, double *dist);
int ora_mg_restrict(ora_ctx *fine, ora_ctx *coarse, int blk, int what,
                    const int32_t *to_coarse, const double *vol_fac);
int ora_mg_matrix_residual(ora_ctx *ctx, double *mean_square);
int ora_mg_invert_diagonal(ora_ctx *ctx);
int ora_mg_save_update(ora_ctx *ctx);
int ora_mg_reset_diagonal(ora_ctx *ctx);
int ora_mg_prolong(ora_ctx *coarse, ora_ctx *fine, int blk, const int32_t *to_coarse,
                   const double *coeffs);
int ora_store_time_n(ora_ctx *ctx, int also_nm1);
int ora_iterate(ora_ctx *ctx, int mm, double cfl, double *l2, agx_linf *linf,
                double *matrix_resid);
int ora_phase_bc_faces(ora_ctx *ctx);
int ora_phase_bc_edges(ora_ctx *ctx);
int ora_phase_residual(ora_ctx *ctx, int mm, double cfl);
int ora_phase_explicit_update(ora_ctx *ctx, int mm, double *l2, agx_linf *linf);
int ora_phase_implicit_begin(ora_ctx *ctx);
int ora_phase_relax_forward(ora_ctx *ctx, int sweep);
int ora_phase_relax_backward(ora_ctx *ctx, int sweep);
int ora_phase_matrix_residual(ora_ctx *ctx, double *matrix_resid);
int ora_phase_implicit_update(ora_ctx *ctx, int mm, double *l2, agx_linf *linf);
int ora_halo_swap_local(ora_ctx *ctx, int what);
int64_t ora_halo_count(ora_ctx *ctx, int conn_id, int what);
int ora_halo_pack(ora_ctx *ctx, int conn_id, int what, double *buf);
int ora_halo_unpack(ora_ctx *ctx, int conn_id, int what, const double *buf);
int ora_set_exchange(ora_ctx *ctx, const agx_exchange *ex);
int ora_rccl_unique_id(void *id128);
int ora_rccl_exchange_create(ora_ctx *ctx, const void *id128, int nranks, int rank);
int ora_halo_exchange(ora_ctx *ctx, int what);
int ora_timing_enable(ora_ctx *ctx, int on);
int ora_timing_get(ora_ctx *ctx, int group, double *avg_ms, int64_t *launches);
int ora_timing_reset(ora_ctx *ctx);
int ora_sync(ora_ctx *ctx);

#ifdef __cplusplus
}
#endif
#endif
