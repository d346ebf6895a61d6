// ONE ensemble sharded over the GPUs of a node, the whole step loop in C (BASELINE.json config C4).
//
// The reference spreads emcee's per-walker lnprob calls over a process pool (alabi/core.py:2300, :2322).  Here the active
// half of every half step is partitioned over the ranks (one process per GPU); each rank runs the half-step kernel on its
// slice [begin, end) of that half's list, then the new rows are exchanged with ONE all-gather per half step, enqueued on
// the run stream -- no host read-back between half steps.  The draws are counter-based (seed, step, walker id), so every
// rank builds identical lists without communication and the chain does not depend on the number of ranks.
//
// Two enqueues per half step, nothing else: the walker rows live in a HISTORY indexed by (half step, rank, slot) -- the
// half-step kernel writes its slice's new rows (coords, logp, accepted) straight into its own segment of the half step's
// block, the all-gather runs IN PLACE on that block (send = the rank's segment of recv), and the next kernels read the two
// rows of a proposal from wherever the draws say they were produced (shard_link_kernel: position of a walker in the step
// that produced its current row -> rank and slot).  There is no pack kernel, no unpack kernel and no per-step chain kernel:
// the chain, the acceptance counters and the final state are gathered from the history once per chunk
// (shard_chain_kernel).  With RCCL (or no communicator traffic at all: one rank) the enqueues of a chunk are captured in a
// hipGraph and replayed.
//
// The collective is RCCL (ncclAllGather over xGMI), resolved with dlopen at run time so that libalabi_hip.so has no
// link-time dependency on it; a communicator can also carry a caller-supplied host function (test rig: two ranks on one
// GPU, which RCCL refuses).  UNMEASURED on multi-GPU hardware: one-rank RCCL rehearsal and gloo rigs only.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <cstdlib>
#include <new>

#include "gp_device.hpp"

struct alabi_comm {
    int rank = 0, nranks = 1;
    ncclComm_t nccl = nullptr;
    alabi_allgather_fn fn = nullptr;      // test rig
    void* user = nullptr;
    double* shist = nullptr;              // [2 chunk][nranks * per][d + 2] rows by (half step, rank, slot)
    size_t shist_cap = 0;
    hipGraphExec_t graph = nullptr;       // the enqueues of one chunk
    // what the captured launches carry: the handle (by serial: a new handle at an old address is another one), its settings
    // generation (y map, affine map, prior, block size: every setter bumps it), the buffers, the GP generation, the partition
    struct Key {
        long long ens_serial = 0, settings_gen = 0, gp_gen = 0;
        void *coords = nullptr, *logp = nullptr, *draws = nullptr, *shist = nullptr;
        double a = 0.0;
        int K = 0, nranks = 0, rank = 0, per = 0;
        bool operator==(const Key& o) const {
            return ens_serial == o.ens_serial && settings_gen == o.settings_gen && gp_gen == o.gp_gen && coords == o.coords &&
                   logp == o.logp && draws == o.draws && shist == o.shist && a == o.a && K == o.K && nranks == o.nranks &&
                   rank == o.rank && per == o.per;
        }
    } key{};
    long long n_replays = 0, n_eager = 0, n_captures = 0;   // full chunks replayed from the graph / chunks enqueued eagerly / captures
    int failed = 0;                       // a chunk could not be enqueued: peers may be inside the collective, the communicator is dead
};

namespace alabi {

namespace {
struct Rccl {
    void* lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    bool ok = false;
};
Rccl& rccl() {
    static Rccl r;
    if (!r.lib) {
        for (const char* name : {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"}) {
            r.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (r.lib) break;
        }
        if (r.lib) {
            r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(dlsym(r.lib, "ncclGetUniqueId"));
            r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(dlsym(r.lib, "ncclCommInitRank"));
            r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(dlsym(r.lib, "ncclCommDestroy"));
            r.AllGather = reinterpret_cast<decltype(r.AllGather)>(dlsym(r.lib, "ncclAllGather"));
            r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(dlsym(r.lib, "ncclGetErrorString"));
            r.ok = r.GetUniqueId && r.CommInitRank && r.CommDestroy && r.AllGather && r.GetErrorString;
        }
    }
    return r;
}
int rccl_fail(ncclResult_t e, const char* what) {
    char buf[384];
    snprintf(buf, sizeof(buf), "%s failed: %s", what, rccl().GetErrorString ? rccl().GetErrorString(e) : "?");
    g_last_error = buf;
    return ALABI_HIP_ERROR;
}
inline void slice_bounds(int n, int world, int rank, int* b, int* e) {
    const int base = n / world, rem = n % world;
    *b = rank * base + (rank < rem ? rank : rem);
    *e = *b + base + (rank < rem ? 1 : 0);
}
}  // namespace

// list position ph of a half of nS walkers -> slot of the (rank, slot) layout: shares differ by at most one, larger first
__device__ inline int shard_slot(int ph, int nS, int nranks, int per) {
    const int base = nS / nranks, rem = nS % nranks;
    const int r = (ph < rem * (base + 1)) ? ph / (base + 1) : rem + (base > 0 ? (ph - rem * (base + 1)) / base : 0);
    return r * per + (ph - (r * base + (r < rem ? r : rem)));
}

// link[2 pos] = word offsets into the history of the two rows the proposal at list position pos reads (own | partner << 32;
// -1: the row is the state the chunk started from).  Version v of walker x was produced by the proposal x made in step v - 1.
__global__ void __launch_bounds__(256)
shard_link_kernel(DrawBuffers b, int W, int n0, int nranks, int per, int row) {
    const int t = blockIdx.x;
    const size_t base = (size_t)t * W;
    const int nslots = nranks * per;
    for (int pos = threadIdx.x; pos < W; pos += 256) {
        const int wl = b.order[base + pos], cl = b.cw[base + pos];
        const int split = pos >= n0;
        int off[2];
        for (int which = 0; which < 2; ++which) {
            const int version = which ? t + split : t, xl = which ? cl : wl;
            off[which] = -1;
            if (version > 0) {
                const int pp = b.pos_of[(size_t)(version - 1) * W + xl];
                const int set = pp >= n0, hp = 2 * (version - 1) + set, ph = pp - set * n0;
                off[which] = (hp * nslots + shard_slot(ph, set ? W - n0 : n0, nranks, per)) * row;
            }
        }
        b.link[2 * (base + pos)] = (unsigned long long)(unsigned)off[0] | ((unsigned long long)(unsigned)off[1] << 32);
    }
}

// After the K steps of a chunk: the (thinned) chain, the acceptance counters and the final state, gathered from the history.
// One thread per (step, walker): the walker's row of that step sits where its position in the step's lists says.
__global__ void __launch_bounds__(256)
shard_chain_kernel(const int* __restrict__ pos_of, const double* __restrict__ shist, int K, int W, int d, int n0, int nranks, int per,
                   int thin_by, long long done0, double* __restrict__ chain, double* __restrict__ chain_logp,
                   unsigned long long* __restrict__ n_accept, double* __restrict__ coords, double* __restrict__ logp) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long long)K * W) return;
    const int t = (int)(i / W), w = (int)(i % W), row = d + 2, nslots = nranks * per;
    const int pp = pos_of[(size_t)t * W + w];
    const int set = pp >= n0, ph = pp - set * n0;
    const double* r = shist + ((size_t)(2 * t + set) * nslots + shard_slot(ph, set ? W - n0 : n0, nranks, per)) * row;
    if (n_accept && r[d + 1] != 0.0) atomicAdd(n_accept + w, 1ull);
    const long long done = done0 + t + 1;
    if (done % thin_by == 0) {
        const size_t slot = (size_t)(done / thin_by - 1);
        if (chain) for (int k = 0; k < d; ++k) chain[(slot * W + w) * d + k] = r[k];
        if (chain_logp) chain_logp[slot * W + w] = r[d];
    }
    if (t == K - 1) {
        for (int k = 0; k < d; ++k) coords[(size_t)w * d + k] = r[k];
        logp[w] = r[d];
    }
}

static int all_gather(alabi_comm* c, const double* send, double* recv, size_t count, hipStream_t s) {
    if (c->nranks == 1 && !c->nccl) {
        if (recv != send) ALABI_HIP_CHECK(hipMemcpyAsync(recv, send, count * sizeof(double), hipMemcpyDeviceToDevice, s));
        return ALABI_OK;
    }
    if (c->fn) return c->fn(send, recv, (long long)count, c->user, reinterpret_cast<void*>(s)) == 0 ? ALABI_OK : ALABI_HIP_ERROR;
    const ncclResult_t e = rccl().AllGather(send, recv, count, ncclDouble, c->nccl, s);
    return e == ncclSuccess ? ALABI_OK : rccl_fail(e, "ncclAllGather");
}

}  // namespace alabi

using namespace alabi;

extern "C" {

int alabi_dist_unique_id(void* id_out) {
    if (!id_out) return ALABI_BAD_ARGUMENT;
    if (!rccl().ok) { g_last_error = "librccl.so could not be loaded"; return ALABI_HIP_ERROR; }
    ncclUniqueId id;
    const ncclResult_t e = rccl().GetUniqueId(&id);
    if (e != ncclSuccess) return rccl_fail(e, "ncclGetUniqueId");
    memcpy(id_out, id.internal, NCCL_UNIQUE_ID_BYTES);
    return ALABI_OK;
}

int alabi_dist_comm_create(const void* id, int rank, int nranks, alabi_comm** out) {
    if (!out || nranks < 1 || rank < 0 || rank >= nranks || (nranks > 1 && !id)) return ALABI_BAD_ARGUMENT;
    alabi_comm* c = new (std::nothrow) alabi_comm();
    if (!c) return ALABI_BAD_ARGUMENT;
    c->rank = rank; c->nranks = nranks;
    if (nranks > 1 || id) {                      // an id with ONE rank: a real one-rank RCCL communicator (rehearsal on a one-GPU box)
        if (!rccl().ok) { delete c; g_last_error = "librccl.so could not be loaded"; return ALABI_HIP_ERROR; }
        ncclUniqueId uid;
        memcpy(uid.internal, id, NCCL_UNIQUE_ID_BYTES);
        const ncclResult_t e = rccl().CommInitRank(&c->nccl, nranks, uid, rank);
        if (e != ncclSuccess) { delete c; return rccl_fail(e, "ncclCommInitRank"); }
    }
    *out = c;
    return ALABI_OK;
}

int alabi_dist_comm_create_callback(alabi_allgather_fn fn, void* user, int rank, int nranks, alabi_comm** out) {
    if (!out || !fn || nranks < 1 || rank < 0 || rank >= nranks) return ALABI_BAD_ARGUMENT;
    alabi_comm* c = new (std::nothrow) alabi_comm();
    if (!c) return ALABI_BAD_ARGUMENT;
    c->rank = rank; c->nranks = nranks; c->fn = fn; c->user = user;
    *out = c;
    return ALABI_OK;
}

int alabi_dist_comm_stats(alabi_comm* c, long long* out) {
    if (!c || !out) return ALABI_BAD_ARGUMENT;
    out[0] = c->n_replays; out[1] = c->n_eager; out[2] = c->n_captures; out[3] = c->failed;
    return ALABI_OK;
}

int alabi_dist_comm_destroy(alabi_comm* c) {
    if (!c) return ALABI_OK;
    if (c->nccl && rccl().ok) (void)rccl().CommDestroy(c->nccl);
    if (c->graph) (void)hipGraphExecDestroy(c->graph);
    if (c->shist) (void)hipFree(c->shist);
    delete c;
    return ALABI_OK;
}

// the enqueues of one chunk of K steps whose records are in the draw buffers: link records, 2 K x (half-step kernel on this
// rank's slice + in-place all-gather), the gather of chain / counters / final state
static int enqueue_sharded_chunk(alabi_ens* e, alabi_comm* c, double* coords, double* logp, int K, int thin_by, long long done0,
                                 double* chain, double* chain_logp, long long* n_accept, hipStream_t s) {
    const int W = e->W, d = e->d, n0 = (W + 1) / 2, row = d + 2;
    const int per = (n0 + c->nranks - 1) / c->nranks;
    const size_t block = (size_t)c->nranks * per * row;                 // doubles per half step
    hipLaunchKernelGGL(shard_link_kernel, dim3(K), dim3(256), 0, s, e->draws, W, n0, c->nranks, per, row);
    int st;
    for (int t = 0; t < K; ++t)
        for (int split = 0; split < 2; ++split) {
            const int nS = split == 0 ? n0 : W - n0;
            if (nS == 0) continue;
            int b, en;
            slice_bounds(nS, c->nranks, c->rank, &b, &en);
            double* blk = c->shist + (size_t)(2 * t + split) * block;
            double* mine = blk + (size_t)c->rank * per * row;
            if (en > b && (st = alabi_ens_half_step_hist(e, coords, logp, t, split, b, en, c->shist, mine, s)) != ALABI_OK) return st;
            if ((c->nranks > 1 || c->nccl) && (st = all_gather(c, mine, blk, (size_t)per * row, s)) != ALABI_OK) return st;
        }
    hipLaunchKernelGGL(shard_chain_kernel, dim3((unsigned)(((long long)K * W + 255) / 256)), dim3(256), 0, s, e->draws.pos_of, c->shist, K, W,
                       d, n0, c->nranks, per, thin_by, done0, chain, chain_logp,
                       reinterpret_cast<unsigned long long*>(n_accept), coords, logp);
    ALABI_LAUNCH_CHECK();
    return ALABI_OK;
}

static int run_sharded(alabi_ens* e, alabi_comm* c, double* coords, double* logp, long long step0, long long nsteps,
                       int thin_by, double a, double* chain, double* chain_logp, long long* n_accept, void* stream) {
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const int W = e->W, d = e->d, n0 = (W + 1) / 2, row = d + 2;
    const int per = (n0 + c->nranks - 1) / c->nranks;                    // equal slots (the larger half decides)
    int st = ALABI_OK;
    const size_t need = (size_t)2 * e->chunk_cap * c->nranks * per * row;
    if ((size_t)2 * e->chunk_cap * c->nranks * per * row > 0x7fffffffull) return ALABI_BAD_ARGUMENT;   // 32-bit word offsets in the link records
    if (c->shist_cap < need) {
        if (c->shist) { ALABI_HIP_CHECK(hipStreamSynchronize(s)); (void)hipFree(c->shist); c->shist = nullptr; c->shist_cap = 0; }
        if (c->graph) { (void)hipGraphExecDestroy(c->graph); c->graph = nullptr; }
        ALABI_HIP_CHECK(hipMalloc(&c->shist, need * sizeof(double)));
        ALABI_HIP_CHECK(hipMemsetAsync(c->shist, 0, need * sizeof(double), s));   // slots beyond a rank's share travel in the all-gather
        c->shist_cap = need;
    }
    // A graph of one full chunk, where everything it enqueues can be captured (RCCL, or no collective at all; the host callback of the
    // test rig cannot).  OPT-IN (ALABI_ENS_SHARD_GRAPH=1) since round 4, when the replay first actually ran (before, the run was handed
    // the null stream, which cannot capture): replaying the ~4000 nodes of a chunk is SLOWER than enqueueing them -- one rank, C3
    // 6.27 vs 5.83 us per half step, C4 11.7 vs 10.5-11.2 (profiles/r04_sharded_one_rank.txt): the half-step kernel runs 5-10 us, so
    // the host is never the bound and the graph only adds its per-node cost.
    const char* genv = getenv("ALABI_ENS_SHARD_GRAPH");
    const bool want_graph = s != nullptr && !c->fn && genv && genv[0] == '1';
    if ((st = ens_sync_consts(e, s)) != ALABI_OK) return st;      // (a host copy + synchronisation: not inside a capture)
    long long done = 0;
    while (done < nsteps) {
        const int K = (int)((nsteps - done) < e->chunk_cap ? (nsteps - done) : e->chunk_cap);
        if ((st = alabi_ens_draw(e, step0 + done, K, a, stream)) != ALABI_OK) return st;   // records of the chunk, identical on every rank
        // a full chunk replays the captured graph (link records, half steps, all-gathers, final state); the chain rows and the
        // counters, whose place depends on how far the run is, are gathered by one more launch behind it
        if (want_graph && K == e->chunk_cap) {
            alabi_comm::Key key;
            key.ens_serial = e->serial; key.settings_gen = e->settings_gen; key.gp_gen = e->gp->gen;
            key.coords = coords; key.logp = logp; key.draws = e->draws.packed; key.shist = c->shist;
            key.a = a; key.K = K; key.nranks = c->nranks; key.rank = c->rank; key.per = per;
            const bool same = c->graph && key == c->key;
            if (!same) {
                if (c->graph) { (void)hipGraphExecDestroy(c->graph); c->graph = nullptr; }
                hipGraph_t g = nullptr;
                ALABI_HIP_CHECK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
                st = enqueue_sharded_chunk(e, c, coords, logp, K, 1, 0, nullptr, nullptr, nullptr, s);
                const hipError_t ce = hipStreamEndCapture(s, &g);
                if (st != ALABI_OK || ce != hipSuccess) {
                    if (g) (void)hipGraphDestroy(g);
                    (void)hipGetLastError();
                    if (st != ALABI_OK) return st;
                } else {
                    const hipError_t ie = hipGraphInstantiate(&c->graph, g, nullptr, nullptr, 0);
                    (void)hipGraphDestroy(g);
                    if (ie != hipSuccess) { (void)hipGetLastError(); c->graph = nullptr; }
                    else { c->key = key; c->n_captures++; }
                }
            }
            if (c->graph) {
                ALABI_HIP_CHECK(hipGraphLaunch(c->graph, s));
                if (chain || chain_logp || n_accept) {
                    hipLaunchKernelGGL(shard_chain_kernel, dim3((unsigned)(((long long)K * W + 255) / 256)), dim3(256), 0, s, e->draws.pos_of,
                                       c->shist, K, W, d, n0, c->nranks, per, thin_by, done, chain, chain_logp,
                                       reinterpret_cast<unsigned long long*>(n_accept), coords, logp);
                    ALABI_LAUNCH_CHECK();
                }
                c->n_replays++;
                done += K;
                continue;
            }
        }
        if ((st = enqueue_sharded_chunk(e, c, coords, logp, K, thin_by, done, chain, chain_logp, n_accept, s)) != ALABI_OK) return st;
        c->n_eager++;
        done += K;
    }
    return st;
}

int alabi_ens_run_sharded(alabi_ens* e, alabi_comm* c, double* coords, double* logp, long long step0, long long nsteps,
                          int thin_by, double a, double* chain, double* chain_logp, long long* n_accept, void* stream) {
    if (!e || !c || !coords || !logp || nsteps < 0 || thin_by < 1 || !(a > 1.0) || e->E != 1) return ALABI_BAD_ARGUMENT;
    if (!e->gp->computed || !e->gp->has_alpha) return ALABI_NOT_COMPUTED;
    if (c->failed) { g_last_error = "this communicator failed in an earlier sharded run (a peer may still be inside the collective)"; return ALABI_HIP_ERROR; }
    if (nsteps == 0) return ALABI_OK;
    // A failure on ONE rank from here on leaves its peers inside (or in front of) an all-gather that will never complete: the
    // communicator is marked dead, and the caller (alabi_amd/dist.py: ShardedRun.run) ends the process with a non-zero status
    // instead of returning to code that would go on to other collectives.
    const int st = run_sharded(e, c, coords, logp, step0, nsteps, thin_by, a, chain, chain_logp, n_accept, stream);
    if (st != ALABI_OK && c->nranks > 1) c->failed = 1;
    return st;
}

}  // extern "C"
