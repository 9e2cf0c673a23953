// b9dist.hpp -- one rank of a walker-parallel run: an RCCL communicator over the GPUs of one node (xGMI) and the
// all-gather of per-walker block summary rows the adaptive proposal step needs (BASELINE.json north_star: "partition
// independent walkers/chains across the 8 GPUs of one node with an RCCL all-gather of log-posteriors over xGMI for
// the adaptive proposal step").  No torch, no MPI: ranks are processes started by the launcher (singlePopMcmc --gpus N,
// bench.py --gpus N, or torchrun), and the 128-byte RCCL unique id travels through a file in a directory they share.
//
// There is no reference counterpart: the reference runs one chain on CPU threads [RECALL].
#pragma once
#include <cstddef>
#include <memory>
#include <string>

namespace b9h {

// How summary rows travel between ranks.  `slot` (0 / 1) names one of two exchanges that may be in flight at once.
class Exchange {
  public:
    virtual ~Exchange() = default;
    virtual int rank() const = 0;
    virtual int world() const = 0;
    // Rows already in HBM (b9_mcmc_block::d_rows): the collective is enqueued on the exchange's own stream behind
    // `ready_event` (a hipEvent_t) and reads them in place; nothing waits on the host.  false = not supported by this
    // exchange (the caller then passes host rows to start_host).
    virtual bool start_device(int slot, const double *d_rows, void *ready_event, size_t count) { (void)slot; (void)d_rows; (void)ready_event; (void)count; return false; }
    virtual bool reads_device_rows() const { return false; }      // true: start_device works (the runner then records the rows' event)
    // Rows on the host.
    virtual void start_host(int slot, const double *rows, size_t count) = 0;
    // Blocks until the exchange of `slot` is complete; returns world() * count doubles in rank order (valid until the
    // slot is started again).
    virtual const double *finish(int slot) = 0;
    // max over ranks (timing) and a barrier; defaults are the one-rank answers
    virtual double all_reduce_max(double v) { return v; }
    virtual void barrier() {}
    virtual const char *name() const = 0;
};

// One rank: every "exchange" is a copy.
std::unique_ptr<Exchange> make_local_exchange();

// RCCL over xGMI.  Collective among `world` processes, one per GPU: rank 0 creates the unique id and publishes it as
// <dir>/rccl_id (written under a temporary name, then renamed); the others wait for the file (up to timeout_s), and
// every rank calls ncclCommInitRank on `device`.  The exchange owns a non-blocking HIP stream of the HIGHEST priority
// (the sampler's context stream has the lowest: the gather gets in between two of its kernels), device and pinned
// host buffers for two slots, and reads device rows in place.  Throws std::runtime_error.
std::unique_ptr<Exchange> make_rccl_exchange(int rank, int world, const std::string &dir, int device, double timeout_s = 120.0);

// The directory the ranks of THIS launch share for the id: $B9_DIST_DIR when the launcher made one, else a name derived
// from the parent process and MASTER_PORT (torchrun workers share both).
std::string default_bootstrap_dir();

// RANK / WORLD_SIZE / LOCAL_RANK of this process as the launchers export them (B9_RANK ... take precedence); 0 / 1 / 0
// when absent.
void rank_from_env(int &rank, int &world, int &local_rank);

}  // namespace b9h
