// ONE ensemble sharded over the GPUs of a node, the whole step loop in C (BASELINE.json config C4).
//
// The reference spreads emcee's per-walker lnprob calls over a process pool (alabi/core.py:2300, :2322).  Here the active
// half of every half step is partitioned over the ranks (one process per GPU); each rank runs the half-step kernel on its
// slice [begin, end) of that half's list, then the updated (coords, logp) rows are exchanged with ONE all-gather per half
// step, enqueued on the run stream -- no host read-back between half steps.  The draws are counter-based (seed, step,
// walker id), so every rank builds identical lists without communication and the chain does not depend on the number of
// ranks.  The collective is RCCL (ncclAllGather over xGMI), resolved with dlopen at run time so that libalabi_hip.so has
// no link-time dependency on it; a communicator can also carry a caller-supplied host function (test rig: two ranks on
// one GPU, which RCCL refuses).
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
    double *send = nullptr, *recv = nullptr;          // [per * (d+1)], [nranks * per * (d+1)]
    size_t send_cap = 0, recv_cap = 0;
    long long *nacc_local = nullptr, *nacc_all = nullptr;   // [W], [nranks * W]
    size_t nacc_cap = 0;
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

// rows of this rank's slice [begin, end) of the half's list -> send[(i - begin)][0..d] = (coords, logp)
__global__ void __launch_bounds__(256)
shard_pack_kernel(const int* __restrict__ order, int begin, int end, int d, const double* __restrict__ coords,
                  const double* __restrict__ logp, double* __restrict__ send) {
    const int i = blockIdx.x * 256 + threadIdx.x, row = d + 1;
    if (i >= (end - begin) * row) return;
    const int r = i / row, k = i % row, w = order[begin + r];
    send[i] = (k < d) ? coords[(size_t)w * d + k] : logp[w];
}

// rows of every OTHER rank's slice out of recv[rank][per][d+1] into coords / logp
__global__ void __launch_bounds__(256)
shard_unpack_kernel(const int* __restrict__ order, int nS, int nranks, int me, int per, int d, const double* __restrict__ recv,
                    double* __restrict__ coords, double* __restrict__ logp) {
    const int i = blockIdx.x * 256 + threadIdx.x, row = d + 1;
    if (i >= nS * row) return;
    const int pos = i / row, k = i % row;
    const int base = nS / nranks, rem = nS % nranks;
    // owner of list position pos (shares differ by at most one, larger shares first)
    int r = (pos < rem * (base + 1)) ? pos / (base + 1) : rem + (base > 0 ? (pos - rem * (base + 1)) / base : 0);
    if (r == me) return;
    const int b = r * base + (r < rem ? r : rem);
    const double v = recv[((size_t)r * per + (pos - b)) * row + k];
    const int w = order[pos];
    if (k < d) coords[(size_t)w * d + k] = v; else logp[w] = v;
}

__global__ void __launch_bounds__(256)
shard_store_kernel(const double* __restrict__ coords, const double* __restrict__ logp, int W, int d, double* __restrict__ chain_row,
                   double* __restrict__ lp_row) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (chain_row && i < W * d) chain_row[i] = coords[i];
    if (lp_row && i < W) lp_row[i] = logp[i];
}

__global__ void __launch_bounds__(256)
shard_nacc_kernel(const long long* __restrict__ all, int nranks, int W, long long* __restrict__ n_accept) {
    const int w = blockIdx.x * 256 + threadIdx.x;
    if (w >= W) return;
    long long s = 0;
    for (int r = 0; r < nranks; ++r) s += all[(size_t)r * W + w];
    n_accept[w] += s;
}

static int all_gather(alabi_comm* c, const double* send, double* recv, size_t count, hipStream_t s) {
    if (c->nranks == 1 && !c->nccl) {
        ALABI_HIP_CHECK(hipMemcpyAsync(recv, send, count * sizeof(double), hipMemcpyDeviceToDevice, s));
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

int alabi_dist_comm_destroy(alabi_comm* c) {
    if (!c) return ALABI_OK;
    if (c->nccl && rccl().ok) (void)rccl().CommDestroy(c->nccl);
    if (c->send) (void)hipFree(c->send);
    if (c->recv) (void)hipFree(c->recv);
    if (c->nacc_local) (void)hipFree(c->nacc_local);
    if (c->nacc_all) (void)hipFree(c->nacc_all);
    delete c;
    return ALABI_OK;
}

int alabi_ens_run_sharded(alabi_ens* e, alabi_comm* c, double* coords, double* logp, long long step0, long long nsteps,
                          int thin_by, double a, double* chain, double* chain_logp, long long* n_accept, void* stream) {
    if (!e || !c || !coords || !logp || nsteps < 0 || thin_by < 1 || !(a > 1.0) || e->E != 1) return ALABI_BAD_ARGUMENT;
    if (!e->gp->computed || !e->gp->has_alpha) return ALABI_NOT_COMPUTED;
    if (nsteps == 0) return ALABI_OK;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const int W = e->W, d = e->d, n0 = (W + 1) / 2, row = d + 1;
    const int per = (n0 + c->nranks - 1) / c->nranks;                    // equal slots (the larger half decides)
    const size_t send_n = (size_t)per * row, recv_n = send_n * c->nranks;
    if (c->send_cap < send_n) {
        if (c->send) (void)hipFree(c->send);
        ALABI_HIP_CHECK(hipMalloc(&c->send, send_n * sizeof(double))); c->send_cap = send_n;
        ALABI_HIP_CHECK(hipMemsetAsync(c->send, 0, send_n * sizeof(double), s));
    }
    if (c->recv_cap < recv_n) {
        if (c->recv) (void)hipFree(c->recv);
        ALABI_HIP_CHECK(hipMalloc(&c->recv, recv_n * sizeof(double))); c->recv_cap = recv_n;
    }
    if (c->nacc_cap < (size_t)W) {
        if (c->nacc_local) (void)hipFree(c->nacc_local);
        if (c->nacc_all) (void)hipFree(c->nacc_all);
        ALABI_HIP_CHECK(hipMalloc(&c->nacc_local, (size_t)W * sizeof(long long)));
        ALABI_HIP_CHECK(hipMalloc(&c->nacc_all, (size_t)W * c->nranks * sizeof(long long)));
        c->nacc_cap = (size_t)W;
    }
    ALABI_HIP_CHECK(hipMemsetAsync(c->nacc_local, 0, (size_t)W * sizeof(long long), s));
    int st;
    long long done = 0;
    while (done < nsteps) {
        const int K = (int)((nsteps - done) < e->chunk_cap ? (nsteps - done) : e->chunk_cap);
        if ((st = alabi_ens_draw(e, step0 + done, K, a, stream)) != ALABI_OK) return st;   // records of the chunk, identical on every rank
        for (int t = 0; t < K; ++t) {
            const int* order = e->draws.order + (size_t)t * W;
            for (int split = 0; split < 2; ++split) {
                const int nS = split == 0 ? n0 : W - n0;
                if (nS == 0) continue;
                int b, en;
                slice_bounds(nS, c->nranks, c->rank, &b, &en);
                if (en > b && (st = alabi_ens_half_step(e, coords, logp, t, split, b, en, c->nacc_local, stream)) != ALABI_OK) return st;
                if (c->nranks > 1 || c->nccl) {
                    const int* list = order + (split ? n0 : 0);
                    if (en > b)
                        hipLaunchKernelGGL(shard_pack_kernel, dim3(((en - b) * row + 255) / 256), dim3(256), 0, s, list, b, en, d, coords,
                                           logp, c->send);
                    if ((st = all_gather(c, c->send, c->recv, send_n, s)) != ALABI_OK) return st;
                    hipLaunchKernelGGL(shard_unpack_kernel, dim3((nS * row + 255) / 256), dim3(256), 0, s, list, nS, c->nranks, c->rank,
                                       per, d, c->recv, coords, logp);
                }
            }
            const long long k = done + t + 1;
            if ((chain || chain_logp) && k % thin_by == 0) {
                const size_t slot = (size_t)(k / thin_by - 1);
                hipLaunchKernelGGL(shard_store_kernel, dim3((W * d + 255) / 256), dim3(256), 0, s, coords, logp, W, d,
                                   chain ? chain + slot * W * d : nullptr, chain_logp ? chain_logp + slot * W : nullptr);
            }
        }
        ALABI_LAUNCH_CHECK();
        done += K;
    }
    if (n_accept) {   // every walker was counted by exactly one rank
        if ((st = all_gather(c, reinterpret_cast<const double*>(c->nacc_local), reinterpret_cast<double*>(c->nacc_all), (size_t)W, s)) != ALABI_OK)
            return st;
        hipLaunchKernelGGL(shard_nacc_kernel, dim3((W + 255) / 256), dim3(256), 0, s, c->nacc_all, c->nranks, W, n_accept);
        ALABI_LAUNCH_CHECK();
    }
    return ALABI_OK;
}

}  // extern "C"
