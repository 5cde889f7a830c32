// dist.hip -- sample-block sharding of the FIR across the GPUs of one node.
//
// One process per GPU (launched by torch.distributed.run or any launcher that sets
// RANK / WORLD_SIZE / LOCAL_RANK); each rank owns a contiguous shard of the signal.
// The only data-path exchange the FIR needs is the Ntaps-1 input samples that precede
// the shard: rank r sends its LAST Ntaps-1 samples to rank r+1 over
// xGMI with RCCL point-to-point (8184 B for 1024 taps of complex64 -- latency-bound,
// one hop, no ring collective), rank 0 zero-fills (zero initial state == lfilter).
// The halo lands in the headroom directly in front of the shard, so the filter
// kernels simply see n_hist = Ntaps-1 valid samples before x[0].
//
// The IIR shards by the same blocks; its exchange is one all-gather of every rank's end state
// (2 x sections doubles), from which each rank folds its own true initial state (sharding.py).
//
// RCCL is bound lazily (dlopen) so that single-GPU users never load it.
#include "skdsp_internal.hpp"
#include <rccl/rccl.h>
#include <dlfcn.h>
#include <cstring>
#include <cstdlib>

namespace skdsp {

struct Rccl {
    void *lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int *) = nullptr;
    ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1;
    double *scratch = nullptr;  // 1 device double for barrier / all-reduce
};

static Rccl &rc() { static Rccl r; return r; }

static int rccl_load()
{
    Rccl &r = rc();
    if (r.lib) return SKDSP_OK;
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char *nm : names) {
        r.lib = dlopen(nm, RTLD_NOW | RTLD_GLOBAL);
        if (r.lib) break;
    }
    SK_CHECK(r.lib, SKDSP_ERR_RCCL, "cannot dlopen librccl.so: %s", dlerror());
#define SK_SYM(field, name)                                                      \
    r.field = reinterpret_cast<decltype(r.field)>(dlsym(r.lib, name));           \
    SK_CHECK(r.field, SKDSP_ERR_RCCL, "librccl.so lacks symbol %s", name)
    SK_SYM(GetUniqueId, "ncclGetUniqueId");
    SK_SYM(CommInitRank, "ncclCommInitRank");
    SK_SYM(CommDestroy, "ncclCommDestroy");
    SK_SYM(CommCount, "ncclCommCount");
    SK_SYM(Send, "ncclSend");
    SK_SYM(Recv, "ncclRecv");
    SK_SYM(GroupStart, "ncclGroupStart");
    SK_SYM(GroupEnd, "ncclGroupEnd");
    SK_SYM(AllReduce, "ncclAllReduce");
    SK_SYM(AllGather, "ncclAllGather");
    SK_SYM(GetErrorString, "ncclGetErrorString");
#undef SK_SYM
    return SKDSP_OK;
}

#define SK_NCCL(call)                                                                           \
    do {                                                                                        \
        ncclResult_t _r = (call);                                                               \
        if (_r != ncclSuccess) {                                                                \
            set_error("RCCL error %d (%s) in %s", (int)_r, rc().GetErrorString(_r), #call);     \
            return SKDSP_ERR_RCCL;                                                              \
        }                                                                                       \
    } while (0)

// one grouped point-to-point step on the compute stream: send `bytes` to rank dst (if >= 0),
// receive `bytes` from rank src (if >= 0)
static int sendrecv_locked(const void *sendbuf, int dst, void *recvbuf, int src, size_t bytes, hipStream_t s = nullptr)
{
    Rccl &r = rc();
    SK_CHECK(r.comm, SKDSP_ERR_RCCL, "dist: no communicator (call skdsp_dist_init first)");
    if (!s) s = ctx().stream;
    SK_NCCL(r.GroupStart());
    if (dst >= 0) SK_NCCL(r.Send(sendbuf, bytes, ncclUint8, dst, r.comm, s));
    if (src >= 0) SK_NCCL(r.Recv(recvbuf, bytes, ncclUint8, src, r.comm, s));
    SK_NCCL(r.GroupEnd());
    return SKDSP_OK;
}

// zero_first = false: the caller treats the missing history of rank 0 as zeros itself (n_hist = 0) -- no fill launches
static int halo_exchange_locked(void *x_dev, int64_t n, int64_t n_halo, int dtype, hipStream_t s = nullptr, bool zero_first = true)
{
    Rccl &r = rc();
    if (!s) s = ctx().stream;
    const size_t esz = dtype_size(dtype);
    SK_CHECK(n_halo >= 0 && n_halo <= n, SKDSP_ERR_BADARG,
             "halo_exchange: halo of %lld samples needs a shard of at least that many (got %lld)", (long long)n_halo,
             (long long)n);
    if (n_halo == 0) return SKDSP_OK;
    char *x0 = (char *)x_dev;
    void *halo = x0 - (size_t)n_halo * esz;
    if ((r.world <= 1 && !opt().shard_self_halo) || !r.comm) {
        if (zero_first) SK_HIP(hipMemsetAsync(halo, 0, (size_t)n_halo * esz, s));
        return SKDSP_OK;
    }
    const size_t bytes = (size_t)n_halo * esz;
    if (r.world == 1 && opt().shard_self_halo)  // test hook: the one rank is its own left neighbour
        return sendrecv_locked(x0 + (size_t)(n - n_halo) * esz, 0, halo, 0, bytes, s);
    if (r.rank == 0 && zero_first) SK_HIP(hipMemsetAsync(halo, 0, bytes, s));  // zero initial state
    return sendrecv_locked(x0 + (size_t)(n - n_halo) * esz, r.rank + 1 < r.world ? r.rank + 1 : -1, halo,
                           r.rank > 0 ? r.rank - 1 : -1, bytes, s);
}

static int allreduce_locked(double *value, ncclRedOp_t op)
{
    Rccl &r = rc();
    hipStream_t s = ctx().stream;
    if (!r.comm) return SKDSP_OK;
    SK_HIP(hipMemcpyAsync(r.scratch, value, 8, hipMemcpyHostToDevice, s));
    SK_NCCL(r.AllReduce(r.scratch, r.scratch, 1, ncclFloat64, op, r.comm, s));
    SK_HIP(hipMemcpyAsync(value, r.scratch, 8, hipMemcpyDeviceToHost, s));
    SK_HIP(hipStreamSynchronize(s));
    return SKDSP_OK;
}

}  // namespace skdsp

using namespace skdsp;

#define API_BEGIN                        \
    {                                    \
        int _rc = ensure_init();         \
        if (_rc) return _rc;             \
    }                                    \
    std::lock_guard<std::mutex> _ctxlk(ctx().mu)

extern "C" {

int skdsp_dist_unique_id(void *id128)
{
    API_BEGIN;
    SK_CHECK(id128, SKDSP_ERR_BADARG, "dist_unique_id: null buffer");
    int r = rccl_load();
    if (r) return r;
    ncclUniqueId id;
    SK_NCCL(rc().GetUniqueId(&id));
    static_assert(sizeof(id) == 128, "ncclUniqueId is 128 bytes");
    memcpy(id128, &id, 128);
    return SKDSP_OK;
}

int skdsp_dist_init(int rank, int world, const void *id128)
{
    API_BEGIN;
    SK_CHECK(world >= 1 && rank >= 0 && rank < world, SKDSP_ERR_BADARG, "dist_init: bad rank %d / world %d", rank, world);
    Rccl &r = rc();
    SK_CHECK(!r.comm, SKDSP_ERR_BADARG, "dist_init: already initialised");
    r.rank = rank;
    r.world = world;
    // a 1-rank job needs no communicator; SKDSP_DIST_FORCE_COMM=1 builds one anyway so the
    // RCCL plumbing (dlopen, id, init, p2p to self, all-reduce) can be exercised on one GPU
    if (world == 1 && !(opt().dist_force_comm && id128)) return SKDSP_OK;
    SK_CHECK(id128, SKDSP_ERR_BADARG, "dist_init: null unique id");
    int rr = rccl_load();
    if (rr) return rr;
    ncclUniqueId id;
    memcpy(&id, id128, 128);
    SK_NCCL(r.CommInitRank(&r.comm, world, id, rank));
    SK_HIP(hipMalloc((void **)&r.scratch, 8));
    return SKDSP_OK;
}

int skdsp_dist_shutdown(void)
{
    Rccl &r = rc();
    if (!ctx().ready) return SKDSP_OK;
    std::lock_guard<std::mutex> lk(ctx().mu);
    (void)hipStreamSynchronize(ctx().stream);
    if (r.comm) {
        (void)r.CommDestroy(r.comm);
        r.comm = nullptr;
    }
    if (r.scratch) {
        (void)hipFree(r.scratch);
        r.scratch = nullptr;
    }
    r.rank = 0;
    r.world = 1;
    return SKDSP_OK;
}

int skdsp_dist_comm_count(int *nranks)
{
    API_BEGIN;
    SK_CHECK(nranks, SKDSP_ERR_BADARG, "dist_comm_count: null argument");
    Rccl &r = rc();
    *nranks = 0;  // no communicator: a 1-rank job (or dist_init not called)
    if (r.comm) SK_NCCL(r.CommCount(r.comm, nranks));
    return SKDSP_OK;
}

int skdsp_dist_barrier(void)
{
    API_BEGIN;
    double v = 0.0;
    int r = allreduce_locked(&v, ncclSum);
    if (r) return r;
    SK_HIP(hipStreamSynchronize(ctx().stream));
    return SKDSP_OK;
}

int skdsp_dist_allreduce_max(double *value)
{
    API_BEGIN;
    SK_CHECK(value, SKDSP_ERR_BADARG, "allreduce: null value");
    return allreduce_locked(value, ncclMax);
}

int skdsp_dist_allreduce_sum(double *value)
{
    API_BEGIN;
    SK_CHECK(value, SKDSP_ERR_BADARG, "allreduce: null value");
    return allreduce_locked(value, ncclSum);
}

int skdsp_dist_sendrecv(const void *send_dev, int dst, void *recv_dev, int src, int64_t bytes)
{
    API_BEGIN;
    SK_CHECK(bytes >= 0, SKDSP_ERR_BADARG, "sendrecv: negative size");
    Rccl &r = rc();
    SK_CHECK(dst < r.world && src < r.world, SKDSP_ERR_BADARG, "sendrecv: peer out of range");
    if (bytes == 0) return SKDSP_OK;
    return sendrecv_locked(send_dev, dst, recv_dev, src, (size_t)bytes);
}

int skdsp_dist_allgather(const void *send_dev, void *recv_dev, int64_t bytes)
{
    API_BEGIN;
    SK_CHECK(bytes >= 0 && (bytes == 0 || (send_dev && recv_dev)), SKDSP_ERR_BADARG, "allgather: bad buffers");
    if (bytes == 0) return SKDSP_OK;
    Rccl &r = rc();
    hipStream_t s = ctx().stream;
    if (!r.comm) {  // one rank: the gather is a copy
        SK_HIP(hipMemcpyAsync(recv_dev, send_dev, (size_t)bytes, hipMemcpyDeviceToDevice, s));
        return SKDSP_OK;
    }
    SK_NCCL(r.AllGather(send_dev, recv_dev, (size_t)bytes, ncclUint8, r.comm, s));
    return SKDSP_OK;
}

int skdsp_dist_halo_exchange(void *x_dev, int64_t n, int64_t n_halo, int dtype)
{
    API_BEGIN;
    SK_CHECK(dtype_valid(dtype), SKDSP_ERR_BADARG, "halo_exchange: bad dtype %d", dtype);
    return halo_exchange_locked(x_dev, n, n_halo, dtype);
}

int skdsp_fir_filter_shard_dev(skdsp_handle hh, void *x_dev, int64_t n_local, void *y_dev)
{
    FirHandle *h = nullptr;
    int64_t halo = 0;
    {
        API_BEGIN;
        HandleBase *b = reinterpret_cast<HandleBase *>(hh);
        SK_CHECK(b && b->kind == H_FIR, SKDSP_ERR_BADARG, "fir_filter_shard: not a FIR handle");
        h = static_cast<FirHandle *>(b);
        halo = h->ntaps - 1;
        // the first shard has no history: its kernels read zeros for x[-k] (n_hist = 0) instead of a halo that
        // would have to be cleared by two fill launches per call (~10 us of a 0.24 ms step)
        const bool self = rc().world == 1 && rc().comm && opt().shard_self_halo;  // (test hook, see Options)
        const bool first = !self && (rc().world <= 1 || !rc().comm || rc().rank == 0);
        // Overlap-save shards: only tile 0 reads the halo.  Tiles 1.. are the same problem started
        // V samples in (their history is local), so they run on the compute stream while the
        // 8 KB halo crosses xGMI on a second stream; tile 0 follows once it has landed.
        int V = 0;
        if (h->dtype == SKDSP_C64 && rc().comm && fir_algo_for(h, n_local) == SKDSP_FIR_OLS &&
            true) {
            int r0 = fir_ols_tile_outputs(h, &V);
            if (r0) return r0;
        }
        if (V > 0 && n_local >= 4 * (int64_t)V && V >= halo) {
            Context &c = ctx();
            if (!c.comm_stream) {
                SK_HIP(hipStreamCreateWithFlags(&c.comm_stream, hipStreamNonBlocking));
                SK_HIP(hipEventCreateWithFlags(&c.ev_in, hipEventDisableTiming));
                SK_HIP(hipEventCreateWithFlags(&c.ev_halo, hipEventDisableTiming));
                SK_HIP(hipMalloc((void **)&c.halo_flag, 256));
                SK_HIP(hipMemsetAsync(c.halo_flag, 0, 256, c.stream));
                c.halo_seq = 0;
            }
            unsigned *err_dev = async_err_dev(kAsyncErrHalo);
            SK_CHECK(err_dev, SKDSP_ERR_HIP, "fir_filter_shard: no host-mapped error word");
            if (c.async_err[kAsyncErrHalo] != 0) {
                // An earlier step's persistent launch gave up waiting for its halo (async_err_check reports it at the next
                // synchronising call; seen here first when the caller queues steps back to back): drain, switch to the
                // two-launch form, and refuse this step -- nothing is enqueued, the caller repeats it on every rank.
                SK_HIP(hipStreamSynchronize(c.stream));
                SK_HIP(hipStreamSynchronize(c.comm_stream));
                return async_err_check(c);
            }
            std::lock_guard<std::mutex> lk(h->mu);
            const size_t esz = dtype_size(h->dtype);
            SK_HIP(hipEventRecord(c.ev_in, c.stream));            // x (and its tail, which is sent) is ready
            SK_HIP(hipStreamWaitEvent(c.comm_stream, c.ev_in, 0));
            int r1 = halo_exchange_locked(x_dev, n_local, halo, h->dtype, c.comm_stream, false);
            if (r1) return r1;
            // The filter is ONE persistent launch that fills every CU (2 x 76 KiB of LDS) for the whole step: a few
            // workgroup slots stay free, so that the send/recv kernel of the halo can start beside it instead of behind
            // it (8 of 512 workgroups; fir_ols_launch leaves them free by default anyway).  Only tile 0 reads the halo:
            // the launch walks it last, behind a device flag that a one-thread kernel on the halo stream sets once the
            // RCCL receive is complete -- no second launch, no 1-workgroup tail (option shard_two_launches restores
            // that form: interior tiles, stream wait on the halo event, tile 0 as its own launch).
            const int reserve = 8;   // workgroup slots the persistent launch leaves to the RCCL send / recv kernel
            // Whether the RCCL kernel really starts beside a persistent launch is a property of the system (free workgroup slots, queue
            // priorities, the RCCL build) that no 1-GPU box can show.  So the overlapped form is EARNED, per process:
            //   step 1   two launches (cannot wait on anything inside a kernel),
            //   step 2   the overlapped launch ON PROBATION -- the poll gives up after a few milliseconds instead of seconds, tile 0 is
            //            computed once more behind the halo event (so the step is valid whatever the poll did), the step is synchronised and
            //            the verdict read: passed -> overlapped from now on; gave up -> two launches for good, nothing to repeat,
            //   step 3.. the proven form.  A first-ever multi-GPU run therefore cannot lose seconds (or a step) to the bounded poll.
            const int probe = opt().shard_probe;
            const bool two = opt().shard_two_launches || (probe && c.halo_state == 0);
            if (two) {
                SK_HIP(hipEventRecord(c.ev_halo, c.comm_stream));
                r1 = fir_ols_launch(h, (char *)x_dev + (size_t)V * esz, n_local - V, V, (char *)y_dev + (size_t)V * esz, c.stream, 1, reserve);
                if (r1) return r1;
                SK_HIP(hipStreamWaitEvent(c.stream, c.ev_halo, 0));
                r1 = fir_ols_launch(h, x_dev, V, first ? 0 : halo, y_dev, c.stream);
                if (r1) return r1;
                if (c.halo_state == 0) opt().shard_halo_state = c.halo_state = 1;
                return SKDSP_OK;
            }
            if (probe && c.halo_state == 1) {
                const unsigned seq = ++c.halo_seq;
                if (!first && probe != 2) {
                    r1 = fir_ols_publish_halo(c.halo_flag, seq, c.comm_stream);
                    if (r1) return r1;
                }
                SK_HIP(hipEventRecord(c.ev_halo, c.comm_stream));
                r1 = fir_ols_launch(h, x_dev, n_local, first ? 0 : halo, y_dev, c.stream, 1, reserve, first ? nullptr : c.halo_flag, seq, err_dev,
                                    4096 /* ~5 ms */);
                if (r1) return r1;
                SK_HIP(hipStreamWaitEvent(c.stream, c.ev_halo, 0));
                r1 = fir_ols_launch(h, x_dev, V, first ? 0 : halo, y_dev, c.stream);   // tile 0 again, from a halo that HAS landed
                if (r1) return r1;
                SK_HIP(hipStreamSynchronize(c.stream));
                if (c.async_err[kAsyncErrHalo] != 0) {   // the receive did not run beside the launch here: not an error, a finding
                    c.async_err[kAsyncErrHalo] = 0;
                    opt().shard_two_launches = 1;
                    opt().shard_halo_state = c.halo_state = 3;
                } else {
                    opt().shard_halo_state = c.halo_state = 2;
                }
                return SKDSP_OK;
            }
            const unsigned seq = ++c.halo_seq;
            if (!first) {
                r1 = fir_ols_publish_halo(c.halo_flag, seq, c.comm_stream);
                if (r1) return r1;
            }
            SK_HIP(hipEventRecord(c.ev_halo, c.comm_stream));
            r1 = fir_ols_launch(h, x_dev, n_local, first ? 0 : halo, y_dev, c.stream, 1, reserve, first ? nullptr : c.halo_flag, seq, err_dev);
            if (r1) return r1;
            // whatever the caller queues next on the compute stream (it may overwrite x, whose tail is being sent) is
            // ordered behind the exchange as well
            SK_HIP(hipStreamWaitEvent(c.stream, c.ev_halo, 0));
            return SKDSP_OK;
        }
        int r = halo_exchange_locked(x_dev, n_local, halo, h->dtype, nullptr, false);
        if (r) return r;
        if (first) halo = 0;
    }
    return skdsp_fir_filter_dev(hh, x_dev, n_local, halo, y_dev);
}

}  // extern "C"
