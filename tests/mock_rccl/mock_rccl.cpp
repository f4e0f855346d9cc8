// mock_rccl.cpp -- TEST INFRASTRUCTURE ONLY: an in-process stand-in for the handful of RCCL entry points that
// cuda-raytracing_amd/csrc/rt_comm.hip resolves with dlsym, so that the N-rank code of the product (rt_comm_init_all,
// rt_render_tiled_all: scratch sizing, offsets of the gathered blocks, the group of gathers, the un-stripe on the root)
// can run with N > 1 "ranks" on the ONE GPU a test box has -- real RCCL refuses two ranks on one device.  All ranks live in
// one process and one device; a gather or a send/receive pair is a device-to-device copy performed when the outermost group closes.  Nothing in the
// product links or loads this file: tests point rt_comm at it with RT_RCCL_LIBRARY (tests/test_gpu_tiling.py).
//   hipcc -shared -fPIC -o librccl_mock.so mock_rccl.cpp
#include <hip/hip_runtime.h>
#include <cstring>
#include <vector>

extern "C" {

typedef enum { ncclSuccess = 0, ncclUnhandledCudaError = 1, ncclInvalidArgument = 4, ncclInvalidUsage = 5 } ncclResult_t;
typedef enum { ncclUint8 = 1 } ncclDataType_t;
typedef struct { char internal[128]; } ncclUniqueId;
struct MockComm { int rank, nranks; };
typedef MockComm* ncclComm_t;

struct PendingGather { const void* send; void* recv; size_t count; int root, rank; hipStream_t stream; };
struct PendingP2p { void* buf; size_t count; int rank, peer; bool send; hipStream_t stream; bool done; };
static std::vector<PendingGather> g_pending;
static std::vector<PendingP2p> g_p2p;
static int g_group_depth = 0;

static ncclResult_t flush_p2p()
{
    // the k-th send of rank s to rank d pairs with the k-th receive of rank d from rank s (NCCL's matching rule)
    if (g_p2p.empty()) return ncclSuccess;
    if (hipDeviceSynchronize() != hipSuccess) return ncclUnhandledCudaError;
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
    g_p2p.clear();
    if (hipDeviceSynchronize() != hipSuccess) return ncclUnhandledCudaError;
    return rc;
}

static ncclResult_t flush()
{
    // every rank's block to the root's buffer; ordered after everything the ranks queued before (one device: a device
    // synchronise is the simplest correct ordering), then on the root's stream
    if (g_pending.empty()) return ncclSuccess;
    void* root_recv = nullptr; hipStream_t root_stream = nullptr; int nroot = 0;
    for (auto& g : g_pending) if (g.rank == g.root) { root_recv = g.recv; root_stream = g.stream; nroot++; }
    if (nroot != 1 || !root_recv) { g_pending.clear(); return ncclInvalidUsage; }
    if (hipDeviceSynchronize() != hipSuccess) return ncclUnhandledCudaError;
    for (auto& g : g_pending)
        if (hipMemcpyAsync((char*)root_recv + (size_t)g.rank * g.count, g.send, g.count, hipMemcpyDeviceToDevice, root_stream) != hipSuccess) return ncclUnhandledCudaError;
    g_pending.clear();
    return ncclSuccess;
}

ncclResult_t ncclGetVersion(int* v) { *v = 99999; return ncclSuccess; }
const char* ncclGetErrorString(ncclResult_t r) { return r == ncclSuccess ? "ok" : "mock rccl error"; }
ncclResult_t ncclGetUniqueId(ncclUniqueId* id) { memset(id, 7, sizeof *id); return ncclSuccess; }
ncclResult_t ncclCommInitRank(ncclComm_t* c, int n, ncclUniqueId, int rank) { if (n != 1) return ncclInvalidUsage; *c = new MockComm{rank, n}; return ncclSuccess; }
ncclResult_t ncclCommInitAll(ncclComm_t* comms, int n, const int*) { for (int i = 0; i < n; i++) comms[i] = new MockComm{i, n}; return ncclSuccess; }
ncclResult_t ncclCommDestroy(ncclComm_t c) { delete c; return ncclSuccess; }
ncclResult_t ncclGroupStart() { g_group_depth++; return ncclSuccess; }
ncclResult_t ncclGroupEnd()
{
    if (g_group_depth <= 0) return ncclInvalidUsage;
    if (--g_group_depth) return ncclSuccess;
    const ncclResult_t a = flush(), b = flush_p2p();
    return a != ncclSuccess ? a : b;
}
ncclResult_t ncclGather(const void* send, void* recv, size_t count, ncclDataType_t, int root, ncclComm_t c, hipStream_t s)
{
    if (!c || root < 0 || root >= c->nranks) return ncclInvalidArgument;
    g_pending.push_back({send, recv, count, root, c->rank, s});
    if (g_group_depth == 0) return c->nranks == 1 ? flush() : ncclInvalidUsage;    // several ranks in one thread need a group
    return ncclSuccess;
}
static ncclResult_t p2p(void* buf, size_t count, int peer, ncclComm_t c, hipStream_t s, bool send)
{
    if (!c || peer < 0 || peer >= c->nranks || !buf) return ncclInvalidArgument;
    if (g_group_depth == 0) return ncclInvalidUsage;                         // one thread, several ranks: only inside a group
    g_p2p.push_back({buf, count, c->rank, peer, send, s, false});
    return ncclSuccess;
}
ncclResult_t ncclSend(const void* b, size_t n, ncclDataType_t, int peer, ncclComm_t c, hipStream_t s) { return p2p((void*)b, n, peer, c, s, true); }
ncclResult_t ncclRecv(void* b, size_t n, ncclDataType_t, int peer, ncclComm_t c, hipStream_t s) { return p2p(b, n, peer, c, s, false); }

}  // extern "C"
