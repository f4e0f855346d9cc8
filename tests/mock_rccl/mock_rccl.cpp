// mock_rccl.cpp -- TEST INFRASTRUCTURE ONLY: an in-process stand-in for the handful of RCCL entry points that
// cuda-raytracing_amd/csrc/rt_comm.hip resolves with dlsym, so that the N-rank code of the product (rt_comm_init_all,
// rt_render_tiled_all: scratch sizing, offsets of the gathered blocks, the group of gathers, the un-stripe on the root)
// can run with N > 1 "ranks" on the ONE GPU a test box has -- real RCCL refuses two ranks on one device.  All ranks live in
// one process and one device; a gather or a send/receive pair is a device-to-device copy performed when the outermost group
// closes.  Second mode, for bench.py's rehearsal: ONE RANK PER PROCESS (ncclCommInitRank with n > 1), the processes sharing
// the box's GPU; data then travels device -> a POSIX shared-memory segment -> device, with a barrier of all ranks in
// between, synchronously inside the call (so the product's N-rank process-per-GPU code -- rt_gather, rt_all_to_all,
// rt_render_tiled, bench.py's pipeline on its streams -- runs for real, only the transport is faked).  Nothing in the
// product links or loads this file: tests point rt_comm at it with RT_RCCL_LIBRARY (tests/test_gpu_tiling.py).
//   hipcc -shared -fPIC -o librccl_mock.so mock_rccl.cpp
#include <hip/hip_runtime.h>
#include <fcntl.h>
#include <sched.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

extern "C" {

typedef enum { ncclSuccess = 0, ncclUnhandledCudaError = 1, ncclInvalidArgument = 4, ncclInvalidUsage = 5 } ncclResult_t;
typedef enum { ncclUint8 = 1 } ncclDataType_t;
typedef struct { char internal[128]; } ncclUniqueId;
// ---- shared segment of the process-per-rank mode ----
constexpr int kMaxRanks = 8;
struct ShmHeader {
    std::atomic<int> ready, arrived, generation, failed;
    int n;
    size_t region_bytes;                                         // every rank owns one region: its outgoing messages of the current group
    size_t off[kMaxRanks][kMaxRanks], len[kMaxRanks][kMaxRanks]; // message src -> dst inside src's region (len 0: none)
};
struct MockComm { int rank, nranks; ShmHeader* shm; char* data; char name[64]; };
typedef MockComm* ncclComm_t;

static bool shm_barrier(MockComm* c)
{
    ShmHeader* h = c->shm;
    const int gen = h->generation.load();
    if (h->arrived.fetch_add(1) + 1 == h->n) { h->arrived.store(0); h->generation.fetch_add(1); return h->failed.load() == 0; }
    const auto t0 = std::chrono::steady_clock::now();
    while (h->generation.load() == gen) {
        sched_yield();
        if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(120)) { h->failed.store(1); return false; }   // a rank that never arrives must not hang the test
    }
    return h->failed.load() == 0;
}

struct PendingGather { const void* send; void* recv; size_t count; int root, rank; hipStream_t stream; };
struct PendingP2p { void* buf; size_t count; int rank, peer; bool send; hipStream_t stream; bool done; };
static std::vector<PendingGather> g_pending;
static std::vector<PendingP2p> g_p2p;
static int g_group_depth = 0;

static ncclResult_t flush_p2p()
{
    // the k-th send of rank s to rank d pairs with the k-th receive of rank d from rank s (NCCL's matching rule)
    if (g_p2p.empty()) return ncclSuccess;
    for (auto& c : g_p2p) if (hipStreamSynchronize(c.stream) != hipSuccess) return ncclUnhandledCudaError;        // (per stream, like the real library)
    ncclResult_t rc = ncclSuccess;
    for (auto& s : g_p2p) {
        if (!s.send) continue;
        PendingP2p* r = nullptr;
        for (auto& c : g_p2p) if (!c.send && !c.done && c.rank == s.peer && c.peer == s.rank) { r = &c; break; }
        if (!r || r->count != s.count) { rc = ncclInvalidUsage; break; }      // real RCCL would hang or corrupt: counts must agree pairwise
        if (hipMemcpyAsync(r->buf, s.buf, s.count, hipMemcpyDeviceToDevice, r->stream) != hipSuccess) { rc = ncclUnhandledCudaError; break; }
        r->done = s.done = true;
    }
    for (auto& c : g_p2p) if (!c.done && rc == ncclSuccess) rc = ncclInvalidUsage;                // a receive nobody sends to
    for (auto& c : g_p2p) if (hipStreamSynchronize(c.stream) != hipSuccess) rc = ncclUnhandledCudaError;
    g_p2p.clear();
    return rc;
}

static ncclResult_t flush()
{
    // every rank's block to the root's buffer, on the root's stream, after what each rank queued on ITS stream
    if (g_pending.empty()) return ncclSuccess;
    void* root_recv = nullptr; hipStream_t root_stream = nullptr; int nroot = 0;
    for (auto& g : g_pending) if (g.rank == g.root) { root_recv = g.recv; root_stream = g.stream; nroot++; }
    if (nroot != 1 || !root_recv) { g_pending.clear(); return ncclInvalidUsage; }
    for (auto& g : g_pending) if (hipStreamSynchronize(g.stream) != hipSuccess) return ncclUnhandledCudaError;    // (per stream, like the real library)
    for (auto& g : g_pending)
        if (hipMemcpyAsync((char*)root_recv + (size_t)g.rank * g.count, g.send, g.count, hipMemcpyDeviceToDevice, root_stream) != hipSuccess) return ncclUnhandledCudaError;
    g_pending.clear();
    return ncclSuccess;
}

ncclResult_t ncclGetVersion(int* v) { *v = 99999; return ncclSuccess; }
const char* ncclGetErrorString(ncclResult_t r) { return r == ncclSuccess ? "ok" : "mock rccl error"; }
ncclResult_t ncclGetUniqueId(ncclUniqueId* id)
{
    static int counter = 0;
    memset(id, 0, sizeof *id);
    snprintf(id->internal, sizeof id->internal, "/rtmock_%d_%d", (int)getpid(), counter++);
    return ncclSuccess;
}
ncclResult_t ncclCommInitRank(ncclComm_t* out, int n, ncclUniqueId id, int rank)
{
    if (n < 1 || n > kMaxRanks || rank < 0 || rank >= n) return ncclInvalidArgument;
    MockComm* c = new MockComm{rank, n, nullptr, nullptr, {0}};
    if (n == 1) { *out = c; return ncclSuccess; }
    size_t region = 64u << 20;
    if (const char* e = getenv("RT_MOCK_REGION_MB")) region = (size_t)atoi(e) << 20;
    const size_t total = sizeof(ShmHeader) + (size_t)n * region;
    id.internal[sizeof id.internal - 1] = 0;
    snprintf(c->name, sizeof c->name, "%s", id.internal);
    int fd = -1;
    if (rank == 0) {
        fd = shm_open(c->name, O_CREAT | O_EXCL | O_RDWR, 0600);
        if (fd < 0 || ftruncate(fd, (off_t)total) != 0) { delete c; return ncclUnhandledCudaError; }
    } else {
        for (int tries = 0; tries < 60000 && fd < 0; tries++) {                  // rank 0 may not have created it yet
            fd = shm_open(c->name, O_RDWR, 0600);
            struct stat sb;
            if (fd >= 0 && (fstat(fd, &sb) != 0 || (size_t)sb.st_size < total)) { close(fd); fd = -1; }
            if (fd < 0) usleep(1000);
        }
        if (fd < 0) { delete c; return ncclUnhandledCudaError; }
    }
    void* m = mmap(nullptr, total, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (m == MAP_FAILED) { delete c; return ncclUnhandledCudaError; }
    c->shm = (ShmHeader*)m;
    c->data = (char*)m + sizeof(ShmHeader);
    if (rank == 0) {                                             // (a fresh segment is zero-filled: counters start at 0)
        c->shm->n = n; c->shm->region_bytes = region;
        c->shm->ready.store(1);
    } else {
        for (int tries = 0; tries < 60000 && c->shm->ready.load() == 0; tries++) usleep(1000);
        if (c->shm->ready.load() == 0) { delete c; return ncclUnhandledCudaError; }
    }
    if (!shm_barrier(c)) { delete c; return ncclUnhandledCudaError; }            // like the real call: returns when every rank has joined
    if (rank == 0) shm_unlink(c->name);                                         // the mapping stays; the name goes once everyone has it
    *out = c;
    return ncclSuccess;
}
ncclResult_t ncclCommInitAll(ncclComm_t* comms, int n, const int*) { for (int i = 0; i < n; i++) comms[i] = new MockComm{i, n, nullptr, nullptr, {0}}; return ncclSuccess; }
ncclResult_t ncclCommCount(const ncclComm_t c, int* n) { *n = c->nranks; return ncclSuccess; }
ncclResult_t ncclCommUserRank(const ncclComm_t c, int* r) { *r = c->rank; return ncclSuccess; }
ncclResult_t ncclCommDestroy(ncclComm_t c)
{
    if (c && c->shm) munmap((void*)c->shm, sizeof(ShmHeader) + (size_t)c->shm->n * c->shm->region_bytes);
    delete c;
    return ncclSuccess;
}

// ---- process-per-rank mode: the operations of one group of ONE rank; every rank closes the same groups in the same order ----
struct ShmOp { void* buf; size_t count; int peer; bool send; hipStream_t stream; };
static std::vector<ShmOp> g_shm_ops;
static MockComm* g_shm_comm = nullptr;

static ncclResult_t flush_shm()
{
    if (!g_shm_comm) return ncclSuccess;
    MockComm* c = g_shm_comm;
    g_shm_comm = nullptr;
    std::vector<ShmOp> ops;
    ops.swap(g_shm_ops);
    ShmHeader* h = c->shm;
    const int me = c->rank;
    ncclResult_t rc = ncclSuccess;
    // Ordering is per STREAM, as with the real library: only work queued on the stream an operation was given is waited for
    // (a device-wide synchronise here would hide a missing dependency between the caller's compute and exchange streams).
    char* mine = c->data + (size_t)me * h->region_bytes;
    size_t used = 0;
    for (int d = 0; d < h->n; d++) h->len[me][d] = 0;
    for (auto& o : ops) {
        if (!o.send || rc != ncclSuccess) continue;
        if (h->len[me][o.peer] != 0 || used + o.count > h->region_bytes || o.count == 0) { rc = ncclInvalidUsage; break; }   // one message per pair and group
        if (hipMemcpyAsync(mine + used, o.buf, o.count, hipMemcpyDeviceToHost, o.stream) != hipSuccess || hipStreamSynchronize(o.stream) != hipSuccess) { rc = ncclUnhandledCudaError; break; }
        h->off[me][o.peer] = used; h->len[me][o.peer] = o.count;
        used += (o.count + 255) & ~(size_t)255;
    }
    if (rc != ncclSuccess) h->failed.store(1);
    if (!shm_barrier(c)) return rc != ncclSuccess ? rc : ncclInvalidUsage;       // every rank's messages are in its region
    for (auto& o : ops) {
        if (o.send) continue;
        if (h->len[o.peer][me] != o.count) { h->failed.store(1); rc = ncclInvalidUsage; break; }     // counts must agree pairwise (real RCCL hangs or corrupts)
        const char* src = c->data + (size_t)o.peer * h->region_bytes + h->off[o.peer][me];
        if (hipMemcpyAsync(o.buf, src, o.count, hipMemcpyHostToDevice, o.stream) != hipSuccess || hipStreamSynchronize(o.stream) != hipSuccess) { h->failed.store(1); rc = ncclUnhandledCudaError; break; }
    }
    if (!shm_barrier(c)) return rc != ncclSuccess ? rc : ncclInvalidUsage;       // regions may be overwritten by the next group
    return rc;
}
static ncclResult_t shm_op(MockComm* c, void* buf, size_t count, int peer, bool send, hipStream_t s)
{
    if (g_shm_comm && g_shm_comm != c) return ncclInvalidUsage;                  // one communicator per process in this mode
    g_shm_comm = c;
    g_shm_ops.push_back({buf, count, peer, send, s});
    return g_group_depth == 0 ? flush_shm() : ncclSuccess;
}
ncclResult_t ncclGroupStart() { g_group_depth++; return ncclSuccess; }
ncclResult_t ncclGroupEnd()
{
    if (g_group_depth <= 0) return ncclInvalidUsage;
    if (--g_group_depth) return ncclSuccess;
    const ncclResult_t a = flush(), b = flush_p2p(), c = flush_shm();
    return a != ncclSuccess ? a : (b != ncclSuccess ? b : c);
}
ncclResult_t ncclGather(const void* send, void* recv, size_t count, ncclDataType_t, int root, ncclComm_t c, hipStream_t s)
{
    if (!c || root < 0 || root >= c->nranks) return ncclInvalidArgument;
    if (c->shm) {                                                // process per rank: a send to the root, and on the root a receive from everyone
        const bool was_open = g_group_depth > 0;
        g_group_depth++;
        ncclResult_t rc = ncclSuccess;
        if (c->rank != root) rc = shm_op(c, (void*)send, count, root, true, s);
        else {
            for (int r = 0; r < c->nranks && rc == ncclSuccess; r++) {
                if (r == root) { if (hipMemcpyAsync((char*)recv + (size_t)r * count, send, count, hipMemcpyDeviceToDevice, s) != hipSuccess) rc = ncclUnhandledCudaError; }
                else rc = shm_op(c, (char*)recv + (size_t)r * count, count, r, false, s);
            }
            g_shm_comm = c;                                      // (a root with no peers' data still joins the barriers)
        }
        g_group_depth--;
        if (rc != ncclSuccess) return rc;
        return was_open ? ncclSuccess : flush_shm();
    }
    g_pending.push_back({send, recv, count, root, c->rank, s});
    if (g_group_depth == 0) return c->nranks == 1 ? flush() : ncclInvalidUsage;    // several ranks in one thread need a group
    return ncclSuccess;
}
static ncclResult_t p2p(void* buf, size_t count, int peer, ncclComm_t c, hipStream_t s, bool send)
{
    if (!c || peer < 0 || peer >= c->nranks || !buf) return ncclInvalidArgument;
    if (c->shm) return shm_op(c, buf, count, peer, send, s);
    if (g_group_depth == 0) return ncclInvalidUsage;                         // one thread, several ranks: only inside a group
    g_p2p.push_back({buf, count, c->rank, peer, send, s, false});
    return ncclSuccess;
}
ncclResult_t ncclSend(const void* b, size_t n, ncclDataType_t, int peer, ncclComm_t c, hipStream_t s) { return p2p((void*)b, n, peer, c, s, true); }
ncclResult_t ncclRecv(void* b, size_t n, ncclDataType_t, int peer, ncclComm_t c, hipStream_t s) { return p2p(b, n, peer, c, s, false); }

}  // extern "C"
