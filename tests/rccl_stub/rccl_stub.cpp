// TEST INFRASTRUCTURE ONLY -- a stand-in for librccl.so.1, never shipped under sert_amd/.
//
// The product dlopen()s "librccl.so.1" by its bare name (sert_amd/csrc/sert_hip.hip: rccl_load) and RCCL refuses
// two ranks on one device, so the asynchronous data-parallel schedule -- grouped ncclSend / ncclRecv all-to-alls,
// ncclReduceScatter / ncclAllGather slabs and the small ncclAllReduce on the communication stream, ordered
// against the compute stream by events -- can only run with peers where there are several GPUs.  The GPU box
// behind the tests has one.  This library implements the eleven entry points the product binds with the SAME
// stream semantics (every call only ENQUEUES work on the caller's stream and returns; the data is exchanged
// when the stream gets there), between processes that share one GPU:
//
//     D2H copy into pinned staging  ->  hipLaunchHostFunc: publish my payload in a POSIX shared-memory outbox,
//     wait for the peers' payloads, combine (sums in RANK ORDER: deterministic)  ->  H2D copy of the result.
//
// Placed ahead on LD_LIBRARY_PATH by tests/test_gpu_rccl_stub.py only.  Every wait has a deadline (a peer that
// died must fail the test, not hang the box).
#include <hip/hip_runtime.h>

#include <atomic>
#include <cerrno>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

namespace {

constexpr int kMaxRanks = 64;
constexpr size_t kOutboxBytes = (size_t)3 << 30;   // virtual size of a rank's outbox (pages are committed on touch)
constexpr size_t kHeaderBytes = 4096;
constexpr int ncclSuccess = 0, ncclUnhandledCudaError = 1, ncclSystemError = 2, ncclInternalError = 3, ncclInvalidArgument = 4;
constexpr int ncclFloat32 = 7, ncclSum = 0;

double deadline_seconds() {
    const char* e = getenv("RCCL_STUB_TIMEOUT");
    return e ? atof(e) : 120.0;
}

struct Header {                       // first page of every outbox
    std::atomic<uint64_t> ready;      // sequence number of the payload that is complete in this outbox
    std::atomic<uint64_t> consumed;   // last sequence number whose peers' payloads THIS rank has finished reading
    std::atomic<uint64_t> alive;      // 1 once the owner mapped it, 2 after ncclCommDestroy
    int64_t seg_off[kMaxRanks];       // grouped send/recv: where the floats meant for rank q start, and how many
    int64_t seg_cnt[kMaxRanks];
};
static_assert(sizeof(Header) <= kHeaderBytes, "header page");

enum OpKind { OP_ALLREDUCE, OP_REDUCESCATTER, OP_ALLGATHER, OP_GROUP };

struct Comm {
    int rank = 0, world = 1;
    std::string base;
    std::vector<char*> box;            // mapped outboxes, one per rank
    uint64_t seq = 0;                  // collectives issued so far (identical on every rank, as NCCL requires)
    float* stage_send = nullptr;       // pinned staging
    float* stage_recv = nullptr;
    size_t stage_send_cap = 0, stage_recv_cap = 0;
    hipEvent_t last_done = nullptr;    // completion of the previous collective's H2D copy (collectives of one
    bool any_issued = false;           // communicator are serialised, whichever stream they are enqueued on)
    std::atomic<int> failed{0};
    Header* hdr(int r) const { return reinterpret_cast<Header*>(box[(size_t)r]); }
    float* data(int r) const { return reinterpret_cast<float*>(box[(size_t)r] + kHeaderBytes); }
};

struct Segment { void* dev; size_t count; int peer; };

struct Op {
    Comm* c;
    OpKind kind;
    uint64_t seq;
    size_t send_count = 0, recv_count = 0;      // floats staged out / in
    std::vector<int64_t> soff, scnt, roff, rcnt;   // OP_GROUP: per peer, in floats, into the staging buffers
};

struct GroupState {
    int depth = 0;
    Comm* comm = nullptr;
    hipStream_t stream = nullptr;
    std::vector<Segment> sends, recvs;
};
thread_local GroupState g_group;
thread_local std::string g_err;

int fail(int code, const std::string& what) {
    g_err = what;
    fprintf(stderr, "[rccl_stub] %s\n", what.c_str());
    return code;
}

template <typename F>
bool wait_until(F cond, const char* what, Comm* c) {
    const auto t0 = std::chrono::steady_clock::now();
    const double limit = deadline_seconds();
    unsigned spins = 0;
    while (!cond()) {
        if (++spins > 2000) std::this_thread::sleep_for(std::chrono::microseconds(50));
        if ((spins & 0x3ff) == 0) {
            const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            if (dt > limit) {
                fprintf(stderr, "[rccl_stub] rank %d: timed out after %.0f s waiting for %s -- aborting the process\n",
                        c->rank, dt, what);
                c->failed.store(1);
                _exit(97);      // a hung collective must kill the test, not the box
            }
        }
    }
    return true;
}

// The part of a collective that runs when the stream reaches it.
void exchange(void* user) {
    Op* op = static_cast<Op*>(user);
    Comm* c = op->c;
    const int W = c->world, me = c->rank;
    const uint64_t s = op->seq;
    // 1. my outbox is free once every peer has finished reading my previous payload
    for (int q = 0; q < W; ++q)
        if (q != me) wait_until([&] { return c->hdr(q)->consumed.load(std::memory_order_acquire) + 1 >= s; }, "a peer to finish reading", c);
    Header* mine = c->hdr(me);
    if (op->kind == OP_GROUP) {
        for (int q = 0; q < W; ++q) { mine->seg_off[q] = op->soff[(size_t)q]; mine->seg_cnt[q] = op->scnt[(size_t)q]; }
    }
    if (op->send_count) memcpy(c->data(me), c->stage_send, op->send_count * sizeof(float));
    mine->ready.store(s, std::memory_order_release);
    // 2. the peers' payloads
    for (int q = 0; q < W; ++q)
        if (q != me) wait_until([&] { return c->hdr(q)->ready.load(std::memory_order_acquire) >= s; }, "a peer's payload", c);
    float* out = c->stage_recv;
    switch (op->kind) {
    case OP_ALLREDUCE: {        // sum over ranks, in rank order
        const size_t n = op->recv_count;
        memcpy(out, c->data(0), n * sizeof(float));
        for (int q = 1; q < W; ++q) { const float* p = c->data(q); for (size_t i = 0; i < n; ++i) out[i] += p[i]; }
        break;
    }
    case OP_REDUCESCATTER: {    // my piece of every rank's buffer, summed in rank order
        const size_t n = op->recv_count;
        memcpy(out, c->data(0) + (size_t)me * n, n * sizeof(float));
        for (int q = 1; q < W; ++q) { const float* p = c->data(q) + (size_t)me * n; for (size_t i = 0; i < n; ++i) out[i] += p[i]; }
        break;
    }
    case OP_ALLGATHER: {
        const size_t n = op->send_count;
        for (int q = 0; q < W; ++q) memcpy(out + (size_t)q * n, c->data(q), n * sizeof(float));
        break;
    }
    case OP_GROUP: {
        for (int q = 0; q < W; ++q) {
            const int64_t want = op->rcnt[(size_t)q];
            if (!want) continue;
            const Header* h = c->hdr(q);
            if (h->seg_cnt[me] != want) {
                fprintf(stderr, "[rccl_stub] rank %d: expects %lld floats from rank %d, which sends %lld (collective %llu)\n", me,
                        (long long)want, q, (long long)h->seg_cnt[me], (unsigned long long)s);
                c->failed.store(1);
                _exit(98);
            }
            memcpy(out + op->roff[(size_t)q], c->data(q) + h->seg_off[me], (size_t)want * sizeof(float));
        }
        break;
    }
    }
    mine->consumed.store(s, std::memory_order_release);
    delete op;
}

int reserve(Comm* c, size_t send, size_t recv) {
    if (send <= c->stage_send_cap && recv <= c->stage_recv_cap) return ncclSuccess;
    // growing the staging buffers: nothing of an earlier collective may still be using them
    if (c->any_issued && hipEventSynchronize(c->last_done) != hipSuccess) return fail(ncclUnhandledCudaError, "hipEventSynchronize");
    if (send > c->stage_send_cap) {
        if (c->stage_send) (void)hipHostFree(c->stage_send);
        c->stage_send_cap = send + send / 4 + 1024;
        if (hipHostMalloc((void**)&c->stage_send, c->stage_send_cap * sizeof(float), hipHostMallocDefault) != hipSuccess)
            return fail(ncclUnhandledCudaError, "hipHostMalloc (send staging)");
    }
    if (recv > c->stage_recv_cap) {
        if (c->stage_recv) (void)hipHostFree(c->stage_recv);
        c->stage_recv_cap = recv + recv / 4 + 1024;
        if (hipHostMalloc((void**)&c->stage_recv, c->stage_recv_cap * sizeof(float), hipHostMallocDefault) != hipSuccess)
            return fail(ncclUnhandledCudaError, "hipHostMalloc (recv staging)");
    }
    return ncclSuccess;
}

#define STUB_HIP(expr)                                                                    \
    do {                                                                                  \
        hipError_t _e = (expr);                                                           \
        if (_e != hipSuccess) return fail(ncclUnhandledCudaError, std::string(#expr) + ": " + hipGetErrorString(_e)); \
    } while (0)

// Enqueue one collective: [wait for the previous one] D2H segments -> host exchange -> H2D segments [record].
int enqueue(Comm* c, Op* op, const std::vector<Segment>& in, const std::vector<size_t>& in_off,
            const std::vector<Segment>& outv, const std::vector<size_t>& out_off, hipStream_t st) {
    if ((op->send_count + 0) * sizeof(float) + kHeaderBytes > kOutboxBytes) { delete op; return fail(ncclInvalidArgument, "payload larger than the stub's outbox"); }
    int rc = reserve(c, op->send_count, op->recv_count);
    if (rc != ncclSuccess) { delete op; return rc; }
    if (c->any_issued) STUB_HIP(hipStreamWaitEvent(st, c->last_done, 0));
    for (size_t i = 0; i < in.size(); ++i)
        if (in[i].count) STUB_HIP(hipMemcpyAsync(c->stage_send + in_off[i], in[i].dev, in[i].count * sizeof(float), hipMemcpyDeviceToHost, st));
    STUB_HIP(hipLaunchHostFunc(st, exchange, op));
    for (size_t i = 0; i < outv.size(); ++i)
        if (outv[i].count) STUB_HIP(hipMemcpyAsync(outv[i].dev, c->stage_recv + out_off[i], outv[i].count * sizeof(float), hipMemcpyHostToDevice, st));
    STUB_HIP(hipEventRecord(c->last_done, st));
    c->any_issued = true;
    return ncclSuccess;
}

int flush_group() {
    GroupState& g = g_group;
    Comm* c = g.comm;
    if (!c) { g = GroupState(); return ncclSuccess; }       // an empty group
    const size_t W = (size_t)c->world;
    Op* op = new Op();
    op->c = c; op->kind = OP_GROUP; op->seq = ++c->seq;
    op->soff.assign(W, 0); op->scnt.assign(W, 0); op->roff.assign(W, 0); op->rcnt.assign(W, 0);
    std::vector<size_t> in_off(g.sends.size()), out_off(g.recvs.size());
    size_t so = 0, ro = 0;
    for (size_t i = 0; i < g.sends.size(); ++i) {
        const Segment& sg = g.sends[i];
        if (op->scnt[(size_t)sg.peer]) { delete op; g = GroupState(); return fail(ncclInvalidArgument, "the stub takes one send per peer and group"); }
        op->soff[(size_t)sg.peer] = (int64_t)so; op->scnt[(size_t)sg.peer] = (int64_t)sg.count;
        in_off[i] = so; so += sg.count;
    }
    for (size_t i = 0; i < g.recvs.size(); ++i) {
        const Segment& sg = g.recvs[i];
        if (op->rcnt[(size_t)sg.peer]) { delete op; g = GroupState(); return fail(ncclInvalidArgument, "the stub takes one recv per peer and group"); }
        op->roff[(size_t)sg.peer] = (int64_t)ro; op->rcnt[(size_t)sg.peer] = (int64_t)sg.count;
        out_off[i] = ro; ro += sg.count;
    }
    op->send_count = so; op->recv_count = ro;
    std::vector<Segment> sends = g.sends, recvs = g.recvs;
    hipStream_t st = g.stream;
    g = GroupState();
    return enqueue(c, op, sends, in_off, recvs, out_off, st);
}

int p2p(bool is_send, void* buf, size_t count, int datatype, int peer, Comm* c, hipStream_t st) {
    if (datatype != ncclFloat32) return fail(ncclInvalidArgument, "the stub moves float32 only");
    if (peer < 0 || peer >= c->world || peer == c->rank) return fail(ncclInvalidArgument, "bad peer");
    GroupState& g = g_group;
    const bool lone = g.depth == 0;
    if (lone) g.depth = 1;
    if (g.comm && (g.comm != c || g.stream != st)) return fail(ncclInvalidArgument, "one communicator and one stream per group");
    g.comm = c; g.stream = st;
    (is_send ? g.sends : g.recvs).push_back(Segment{buf, count, peer});
    if (lone) { g.depth = 0; return flush_group(); }
    return ncclSuccess;
}

// Fault injection (tests/test_gpu_rccl_stub.py::test_a_failing_collective_is_a_clean_error): RCCL_STUB_FAIL_SEQ=N makes the
// N-th collective of the communicator (1-based, counted as the product issues them: all-reduce, reduce-scatter, all-gather,
// a group of sends / receives) RETURN ncclSystemError instead of enqueueing anything -- on the rank RCCL_STUB_FAIL_RANK names,
// or on every rank when that is unset.  What a node sees when a link or a peer goes away under a collective.
bool inject_failure(Comm* c, uint64_t seq) {
    static const long fail_seq = getenv("RCCL_STUB_FAIL_SEQ") ? atol(getenv("RCCL_STUB_FAIL_SEQ")) : 0;
    static const int fail_rank = getenv("RCCL_STUB_FAIL_RANK") ? atoi(getenv("RCCL_STUB_FAIL_RANK")) : -1;
    return fail_seq > 0 && (long)seq >= fail_seq && (fail_rank < 0 || fail_rank == c->rank);
}

}  // namespace

extern "C" {

int ncclGetUniqueId(void* id) {       // 128 bytes: the base name of this communicator's shared-memory objects
    char* p = static_cast<char*>(id);
    memset(p, 0, 128);
    unsigned long long salt = (unsigned long long)std::chrono::steady_clock::now().time_since_epoch().count();
    snprintf(p, 128, "/sert_rccl_stub_%d_%llx", (int)getpid(), salt);
    return ncclSuccess;
}

struct StubUniqueId { char internal[128]; };

int ncclCommInitRank(void** comm, int nranks, StubUniqueId id, int rank) {
    if (nranks < 1 || nranks > kMaxRanks || rank < 0 || rank >= nranks) return fail(ncclInvalidArgument, "bad rank / world");
    Comm* c = new Comm();
    c->rank = rank; c->world = nranks;
    id.internal[127] = 0;
    c->base = id.internal;
    c->box.assign((size_t)nranks, nullptr);
    auto open_box = [&](int r, bool create) -> char* {
        const std::string name = c->base + "_" + std::to_string(r);
        int fd = -1;
        const auto t0 = std::chrono::steady_clock::now();
        for (;;) {
            fd = shm_open(name.c_str(), create ? (O_CREAT | O_RDWR) : O_RDWR, 0600);
            if (fd >= 0) {
                struct stat sb;
                if (create) { if (ftruncate(fd, (off_t)kOutboxBytes) != 0) { close(fd); return nullptr; } break; }
                if (fstat(fd, &sb) == 0 && (size_t)sb.st_size >= kOutboxBytes) break;      // the owner has sized it
                close(fd); fd = -1;
            }
            if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > deadline_seconds()) return nullptr;
            std::this_thread::sleep_for(std::chrono::milliseconds(2));
        }
        void* p = mmap(nullptr, kOutboxBytes, PROT_READ | PROT_WRITE, MAP_SHARED | MAP_NORESERVE, fd, 0);
        close(fd);
        return p == MAP_FAILED ? nullptr : static_cast<char*>(p);
    };
    c->box[(size_t)rank] = open_box(rank, true);
    if (!c->box[(size_t)rank]) { delete c; return fail(ncclSystemError, std::string("cannot create the outbox: ") + strerror(errno)); }
    c->hdr(rank)->alive.store(1, std::memory_order_release);
    for (int r = 0; r < nranks; ++r) {
        if (r == rank) continue;
        c->box[(size_t)r] = open_box(r, false);
        if (!c->box[(size_t)r]) { delete c; return fail(ncclSystemError, "a peer's outbox did not appear"); }
        wait_until([&] { return c->hdr(r)->alive.load(std::memory_order_acquire) >= 1; }, "a peer to map its outbox", c);
    }
    if (hipEventCreateWithFlags(&c->last_done, hipEventDisableTiming) != hipSuccess) { delete c; return fail(ncclUnhandledCudaError, "hipEventCreate"); }
    *comm = c;
    return ncclSuccess;
}

int ncclCommDestroy(void* comm) {
    Comm* c = static_cast<Comm*>(comm);
    if (!c) return ncclSuccess;
    if (c->any_issued) (void)hipEventSynchronize(c->last_done);
    // every rank unlinks its own object; the mappings of the peers stay valid until they unmap
    c->hdr(c->rank)->alive.store(2, std::memory_order_release);
    shm_unlink((c->base + "_" + std::to_string(c->rank)).c_str());
    for (char* p : c->box) if (p) munmap(p, kOutboxBytes);
    if (c->stage_send) (void)hipHostFree(c->stage_send);
    if (c->stage_recv) (void)hipHostFree(c->stage_recv);
    if (c->last_done) (void)hipEventDestroy(c->last_done);
    delete c;
    return ncclSuccess;
}

int ncclAllReduce(const void* send, void* recv, size_t count, int datatype, int op, void* comm, hipStream_t st) {
    Comm* c = static_cast<Comm*>(comm);
    if (datatype != ncclFloat32 || op != ncclSum) return fail(ncclInvalidArgument, "the stub sums float32 only");
    if (inject_failure(c, c->seq + 1)) return fail(ncclSystemError, "ncclAllReduce: injected failure (RCCL_STUB_FAIL_SEQ)");
    Op* o = new Op();
    o->c = c; o->kind = OP_ALLREDUCE; o->seq = ++c->seq; o->send_count = count; o->recv_count = count;
    return enqueue(c, o, {Segment{const_cast<void*>(send), count, 0}}, {0}, {Segment{recv, count, 0}}, {0}, st);
}

int ncclReduceScatter(const void* send, void* recv, size_t recvcount, int datatype, int op, void* comm, hipStream_t st) {
    Comm* c = static_cast<Comm*>(comm);
    if (datatype != ncclFloat32 || op != ncclSum) return fail(ncclInvalidArgument, "the stub sums float32 only");
    if (inject_failure(c, c->seq + 1)) return fail(ncclSystemError, "ncclReduceScatter: injected failure (RCCL_STUB_FAIL_SEQ)");
    Op* o = new Op();
    o->c = c; o->kind = OP_REDUCESCATTER; o->seq = ++c->seq; o->send_count = recvcount * (size_t)c->world; o->recv_count = recvcount;
    return enqueue(c, o, {Segment{const_cast<void*>(send), o->send_count, 0}}, {0}, {Segment{recv, recvcount, 0}}, {0}, st);
}

int ncclAllGather(const void* send, void* recv, size_t sendcount, int datatype, void* comm, hipStream_t st) {
    Comm* c = static_cast<Comm*>(comm);
    if (datatype != ncclFloat32) return fail(ncclInvalidArgument, "the stub moves float32 only");
    if (inject_failure(c, c->seq + 1)) return fail(ncclSystemError, "ncclAllGather: injected failure (RCCL_STUB_FAIL_SEQ)");
    Op* o = new Op();
    o->c = c; o->kind = OP_ALLGATHER; o->seq = ++c->seq; o->send_count = sendcount; o->recv_count = sendcount * (size_t)c->world;
    return enqueue(c, o, {Segment{const_cast<void*>(send), sendcount, 0}}, {0}, {Segment{recv, o->recv_count, 0}}, {0}, st);
}

int ncclSend(const void* buf, size_t count, int datatype, int peer, void* comm, hipStream_t st) {
    return p2p(true, const_cast<void*>(buf), count, datatype, peer, static_cast<Comm*>(comm), st);
}

int ncclRecv(void* buf, size_t count, int datatype, int peer, void* comm, hipStream_t st) {
    return p2p(false, buf, count, datatype, peer, static_cast<Comm*>(comm), st);
}

int ncclGroupStart() { g_group.depth += 1; return ncclSuccess; }

int ncclGroupEnd() {
    if (g_group.depth <= 0) return fail(ncclInvalidArgument, "ncclGroupEnd without ncclGroupStart");
    if (--g_group.depth > 0) return ncclSuccess;
    if (g_group.comm && inject_failure(g_group.comm, g_group.comm->seq + 1)) {
        g_group = GroupState();
        return fail(ncclSystemError, "ncclGroupEnd: injected failure (RCCL_STUB_FAIL_SEQ)");
    }
    return flush_group();
}

const char* ncclGetErrorString(int code) {
    static thread_local std::string s;
    s = "rccl_stub error " + std::to_string(code) + (g_err.empty() ? "" : (": " + g_err));
    return s.c_str();
}

// lets a test prove that THIS library is the one the product loaded
int rccl_stub_marker() { return 0x5e47; }

}  // extern "C"
