/*
 * base9_host.h -- C surface of libbase9host.so: the C++ host side above the hot path's C ABI (base9_hip.h) --
 * the walker-parallel adaptive-Metropolis driver and its RCCL exchange (SURVEY.md section 8 rows e, f-1) plus the
 * file parsers (rows f-2, f-3) -- for callers that are not C++ (bench.py and the tests bind it with ctypes; a
 * reference-side binding would use the C++ classes in base_amd/host/ directly).
 *
 * The reference has no counterpart of the multi-GPU part: it runs one adaptive chain on CPU threads [RECALL]; walkers,
 * their sharding over GPUs and the RCCL all-gather are constructs of BASELINE.json's north_star.
 *
 * Every function returns 0 on success and -1 on failure (b9h_last_error() gives the text) unless stated otherwise.
 * Nothing here evaluates a likelihood: every number comes from libbase9hip.so on the GPU, or -- in the callback
 * variants, a seam for testing this library's own logic on a machine without a GPU -- from the caller.
 */
#ifndef BASE9_HOST_H
#define BASE9_HOST_H

#include "base9_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

const char *b9h_last_error(void);

/* ---- ranks --------------------------------------------------------------------------------------------------- */
/* RANK / WORLD_SIZE / LOCAL_RANK of this process as the launchers export them (B9_RANK ... first); 0 / 1 / 0 if absent */
void b9h_rank_from_env(int *rank, int *world, int *local_rank);
int b9h_device_synchronize(void);                       /* hipDeviceSynchronize of the current device */

/* ---- exchange of block summary rows between ranks --------------------------------------------------------------- */
typedef int (*b9h_gather_fn)(void *user, const double *mine, size_t count, double *all);
int b9h_exchange_local(void **out);                     /* one rank */
/* RCCL over xGMI, one process per GPU; dir NULL = the launch's default bootstrap directory (b9dist.hpp) */
int b9h_exchange_rccl(int rank, int world, const char *dir, int device, void **out);
int b9h_exchange_callback(b9h_gather_fn gather, void *user, int rank, int world, void **out);   /* tests */
void b9h_exchange_free(void *exchange);
int b9h_exchange_barrier(void *exchange);
int b9h_exchange_max(void *exchange, double value, double *max_over_ranks);
int b9h_exchange_world(void *exchange);                 /* returns the number of ranks */
const char *b9h_exchange_name(void *exchange);
/* What the communicator itself reports: its rank count (ncclCommCount; 0 for an exchange without a communicator) ...  */
int b9h_exchange_comm_ranks(void *exchange);
/* ... and the PCI bus ids of the ranks' GPUs in rank order, comma-separated, all-gathered through the communicator
 * ("" without one).  Returns 0, or -1 when `cap` is too small. */
int b9h_exchange_devices(void *exchange, char *out, int cap);
/* 0 when a communicator of `comm_ranks` ranks on the GPUs `devices_csv` (as b9h_exchange_devices returns them) is what a launch
 * of `world` ranks must have -- that many ranks on that many DISTINCT devices -- else 1 with the reason in msg[cap].  The RCCL
 * exchange refuses such a group itself (b9h_exchange_rccl fails); launchers check their own report with it. */
int b9h_group_check(int world, int comm_ranks, const char *devices_csv, char *msg, int cap);
/* 1 when this process is a rank of a --forceRanks / --force-ranks launch (B9_FORCE_RANKS): one GPU, full multi-rank route */
int b9h_forced_ranks(void);
/* Test hook of the launchers' deadlines: parks the calling rank for ever when B9_TEST_STALL == "<where>:<rank>". */
void b9h_test_stall(const char *where, int rank);

/* ---- the sampler ------------------------------------------------------------------------------------------------ */
typedef int (*b9h_block_fn)(void *user, const double *params_in, const double *logpost_in, const int32_t *walker_ids, int n_local,
                            const int32_t *free_idx, int d, const double *chol, uint64_t seed, int64_t step0, int n_steps,
                            double *params_out, double *logpost_out, double *samples, double *lps, int64_t *n_accept);
typedef int (*b9h_logpost_fn)(void *user, const double *params, int n, double *out);
/* n_walkers: all ranks together; free_idx / step: [d]; mode: the B9_MODE_* of the context's options */
int b9h_sampler_create(b9_ctx *ctx, int mode, int n_walkers, const int32_t *free_idx, const double *step, int d,
                       uint64_t seed, int block, void *exchange, void **out);
int b9h_sampler_create_callback(b9h_block_fn run, b9h_logpost_fn eval, void *user, int n_walkers, const int32_t *free_idx,
                                const double *step, int d, uint64_t seed, int block, void *exchange, void **out);   /* tests */
void b9h_sampler_free(void *sampler);
int b9h_sampler_initialise(void *sampler, const double *start /* [n_walkers][B9_NPARAM], the same on every rank */);
/* n_steps in blocks; adapt = 0 freezes the proposal.  samples [n_steps][n_local][d] / lps [n_steps][n_local]: nullable */
int b9h_sampler_run(void *sampler, int64_t n_steps, int adapt, double *samples, double *lps);
int b9h_sampler_n_local(void *sampler);                 /* returns the number of local walkers */
/* any output may be NULL; chol [d*d]; all_params [n_walkers][B9_NPARAM] / all_logpost [n_walkers] as of the last exchange */
int b9h_sampler_state(void *sampler, int64_t *steps, int64_t *accepted_local, double *scale, double *chol,
                      double *all_params, double *all_logpost);
/* the host statement of the device's block summary rows (b9_mcmc_block::rows) */
int b9h_summary_rows(const double *samples, const double *params_end, const double *logpost_end, int n_steps, int n_local, int d,
                     const double *origin, double *rows);

/* ---- file parsers (docs/FORMATS.md) ---------------------------------------------------------------------------------- */
int b9h_load_pack(const char *dir, const char *ms_model, const char *wd_model, const char *filters_csv, void **handle, b9_pack *view);
void b9h_free_pack(void *handle);
int b9h_read_phot(const char *path, double min_mag, double max_mag, int index, void **handle, b9_stars *view,
                  char *filters_out, int filters_cap);
void b9h_free_phot(void *handle);
int b9h_settings_dump(int argc, char **argv, char *out, int cap);
/* rank 0's merge of <final_path>.part<r> into <final_path> after a --gpus N run (b9h::merge_result_parts; exposed for tests) */
int b9h_merge_parts(const char *final_path, int world, int walkers_per_rank, long rows_per_part);

#ifdef __cplusplus
}
#endif
#endif /* BASE9_HOST_H */
