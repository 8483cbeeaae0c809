// Multi-GPU exchange of libeasyrag_hip.so: one all-gather of the fused top-k over RCCL (erh_comm_*, erh_pack_topk / erh_unpack_topk,
// erh_allgather_topk; include/easyrag_hip.h).
#include "../../include/easyrag_hip.h"
#include "handle.h"

// ---- multi-GPU: all-gather of the fused top-k over RCCL ---------------------------------------------------------------
// The corpus is replicated and the query batch sharded contiguously over the ranks (north_star),
// so the only exchange is one all-gather of [B_local x k] (score, id, len) rows.  RCCL is bound at run time with
// dlopen("librccl.so.1") -- the instance torch already mapped when the caller uses torch, the ROCm one otherwise --
// so the library has no link-time dependency on it and single-GPU users never load it.
namespace {

typedef struct { char internal[128]; } rccl_unique_id;                 // = ncclUniqueId (rccl.h)
struct Rccl {
    void *dl = nullptr;
    int (*GetUniqueId)(rccl_unique_id *) = nullptr;
    int (*CommInitRank)(void **, int, rccl_unique_id, int) = nullptr;
    int (*AllGather)(const void *, void *, size_t, int, void *, hipStream_t) = nullptr;
    int (*CommDestroy)(void *) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    bool ok = false;
};

Rccl &rccl() {
    static Rccl r;
    if (r.dl) return r;
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char *n : names) {
        r.dl = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        if (r.dl) break;
    }
    if (!r.dl) return r;
    r.GetUniqueId = (int (*)(rccl_unique_id *))dlsym(r.dl, "ncclGetUniqueId");
    r.CommInitRank = (int (*)(void **, int, rccl_unique_id, int))dlsym(r.dl, "ncclCommInitRank");
    r.AllGather = (int (*)(const void *, void *, size_t, int, void *, hipStream_t))dlsym(r.dl, "ncclAllGather");
    r.CommDestroy = (int (*)(void *))dlsym(r.dl, "ncclCommDestroy");
    r.GetErrorString = (const char *(*)(int))dlsym(r.dl, "ncclGetErrorString");
    r.ok = r.GetUniqueId && r.CommInitRank && r.AllGather && r.CommDestroy;
    return r;
}

// a timed-out erh_comm_init whose helper has meanwhile returned: destroy the communicator nobody will use
void comm_reap_pending(erh_handle *h) {
    if (!h->comm_pending || !h->comm_pending->finished.load(std::memory_order_acquire)) return;
    if (h->comm_pending->rc == 0 && h->comm_pending->comm) {
        (void)hipSetDevice(h->device);
        (void)rccl().CommDestroy(h->comm_pending->comm);
    }
    h->comm_pending.reset();
}

int rccl_fail(erh_handle *h, const char *what, int rc) {
    char buf[256];
    Rccl &r = rccl();
    snprintf(buf, sizeof buf, "%s: RCCL error %d (%s)", what, rc, r.GetErrorString ? r.GetErrorString(rc) : "?");
    return h->fail(ERH_ERR_HIP, buf);
}

}  // namespace

extern "C" {

int erh_comm_unique_id(void *out128) {
    if (!out128) return ERH_ERR_INVALID;
    Rccl &r = rccl();
    if (!r.ok) return ERH_ERR_UNSUPPORTED;
    rccl_unique_id id;
    if (r.GetUniqueId(&id) != 0) return ERH_ERR_HIP;
    memcpy(out128, &id, sizeof id);
    return ERH_OK;
}

int erh_comm_init(erh_handle *h, int rank, int world, const void *id128) {
    if (!h) return ERH_ERR_INVALID;
    if (!id128 || world < 1 || rank < 0 || rank >= world) return h->fail(ERH_ERR_INVALID, "erh_comm_init: bad rank / world / id");
    if (h->comm) return h->fail(ERH_ERR_STATE, "erh_comm_init: communicator already initialised");
    Rccl &r = rccl();
    if (!r.ok) return h->fail(ERH_ERR_UNSUPPORTED, "erh_comm_init: librccl.so.1 not found or incomplete");
    HIPCHK(h, hipSetDevice(h->device));
    rccl_unique_id id;
    memcpy(&id, id128, sizeof id);
    // ncclCommInitRank blocks until every rank has joined.  A rank that never arrives (crashed, wrong id) must not hang
    // the others for ever: the call runs on a helper thread and this one waits at most comm_timeout_s seconds (option,
    // default 120).  After a timeout the helper is abandoned (it may still be blocked inside RCCL) and the handle stays
    // without a communicator -- callers fall back to the torch.distributed gather (easyrag_amd.dist.QueryShards).
    // A communicator that arrives late is destroyed by the next erh_comm_init / erh_comm_destroy / erh_destroy that finds the
    // helper finished; a helper still inside RCCL at process exit is the caller's problem -- after a timeout the process
    // should exit (the peers hold a communicator this rank never joined).
    comm_reap_pending(h);
    auto state = std::make_shared<CommInitState>();
    auto done = std::make_shared<std::promise<void>>();
    std::future<void> fut = done->get_future();
    const int dev = h->device;
    auto init_fn = r.CommInitRank;
    std::thread([state, done, init_fn, id, world, rank, dev]() {
        (void)hipSetDevice(dev);
        state->rc = init_fn(&state->comm, world, id, rank);
        state->finished.store(1, std::memory_order_release);
        done->set_value();
    }).detach();
    if (fut.wait_for(std::chrono::seconds(h->opt_comm_timeout_s)) != std::future_status::ready) {
        h->comm_pending = state;
        return h->fail(ERH_ERR_HIP, "erh_comm_init: ncclCommInitRank did not return within comm_timeout_s (a rank is missing?)");
    }
    const int rc = state->rc;
    void *c = state->comm;
    if (rc != 0) return rccl_fail(h, "ncclCommInitRank", rc);
    h->comm = c;
    h->comm_rank = rank;
    h->comm_world = world;
    return ERH_OK;
}

int erh_comm_destroy(erh_handle *h) {
    if (!h) return ERH_ERR_INVALID;
    comm_reap_pending(h);
    if (h->comm) {
        (void)hipSetDevice(h->device);
        (void)rccl().CommDestroy(h->comm);
        h->comm = nullptr;
    }
    h->comm_rank = 0;
    h->comm_world = 1;
    return ERH_OK;
}

int erh_topk_row_bytes(int k) { return k > 0 ? erh::topk_row_bytes(k) : 0; }

int erh_pack_topk(erh_handle *h, const int32_t *ids, const double *scores, const int32_t *lens, int b_local, int k,
                  int rows, void *out_rows, void *stream) {
    if (!h) return ERH_ERR_INVALID;
    if (!ids || !scores || !lens || !out_rows || b_local < 0 || k <= 0 || rows < b_local)
        return h->fail(ERH_ERR_INVALID, "erh_pack_topk: null pointer or bad sizes");
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, erh::launch_pack_topk(ids, scores, lens, b_local, k, rows, out_rows, (hipStream_t)stream));
    return ERH_OK;
}

int erh_unpack_topk(erh_handle *h, const void *gathered_rows, int n_queries, int world, int k,
                    int32_t *out_ids, double *out_scores, int32_t *out_len, void *stream) {
    if (!h) return ERH_ERR_INVALID;
    if (!gathered_rows || !out_ids || !out_scores || !out_len || n_queries <= 0 || world <= 0 || k <= 0)
        return h->fail(ERH_ERR_INVALID, "erh_unpack_topk: null pointer or bad sizes");
    HIPCHK(h, hipSetDevice(h->device));
    const int m = (n_queries + world - 1) / world;
    HIPCHK(h, erh::launch_unpack_topk(gathered_rows, n_queries, world, k, m, out_ids, out_scores, out_len,
                                      (hipStream_t)stream));
    return ERH_OK;
}

int erh_allgather_topk(erh_handle *h, const int32_t *ids, const double *scores, const int32_t *lens, int b_local, int k,
                       int n_queries, int32_t *out_ids, double *out_scores, int32_t *out_len, void *stream) {
    if (!h) return ERH_ERR_INVALID;
    if (!ids || !scores || !lens || !out_ids || !out_scores || !out_len || k <= 0 || n_queries <= 0 || b_local < 0)
        return h->fail(ERH_ERR_INVALID, "erh_allgather_topk: null pointer or bad sizes");
    const int world = h->comm_world, rank = h->comm_rank;
    const int base = n_queries / world, rem = n_queries % world;
    if (b_local != base + (rank < rem ? 1 : 0))
        return h->fail(ERH_ERR_INVALID, "erh_allgather_topk: b_local is not this rank's contiguous shard of n_queries");
    if (world > 1 && !h->comm) return h->fail(ERH_ERR_STATE, "erh_allgather_topk before erh_comm_init");
    HIPCHK(h, hipSetDevice(h->device));
    hipStream_t st = (hipStream_t)stream;
    const int m = (n_queries + world - 1) / world;
    const size_t row = (size_t)erh::topk_row_bytes(k);
    HIPCHK(h, h->gather_send.ensure(row * m));
    HIPCHK(h, h->gather_recv.ensure(row * m * world));
    HIPCHK(h, erh::launch_pack_topk(ids, scores, lens, b_local, k, m, h->gather_send.p, st));
    if (world > 1) {
        const int rc = rccl().AllGather(h->gather_send.p, h->gather_recv.p, row * m, /*ncclChar*/ 0, h->comm, st);
        if (rc != 0) return rccl_fail(h, "ncclAllGather", rc);
    } else {
        HIPCHK(h, hipMemcpyAsync(h->gather_recv.p, h->gather_send.p, row * m, hipMemcpyDeviceToDevice, st));
    }
    HIPCHK(h, erh::launch_unpack_topk(h->gather_recv.p, n_queries, world, k, m, out_ids, out_scores, out_len, st));
    return ERH_OK;
}

}  // extern "C"

