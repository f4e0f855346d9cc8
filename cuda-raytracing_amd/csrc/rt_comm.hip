// rt_comm.hip -- frame tiling across the GPUs of one node: RCCL communicator, gather / all-to-all of stripe buffers,
// and the one-call tiled render (include/rt_hip.h, "frame tiling across GPUs").
//
// The reference is single-GPU (one render<<<>>> per frame on the default device, Camera.cu:18-41); BASELINE.json's
// north_star asks for the frame tiled over the 8 GPUs of a node "with a final RCCL gather over xGMI".  Rays are
// independent, so the only exchange step is that gather: every rank's stripes of a frame go to the rank that assembles
// it (rt_gather), or -- for a stream of frames -- every rank assembles 1/N of the frames of a group and the N gathers of
// the group travel as one all-to-all (rt_all_to_all), so that all 56 point-to-point xGMI links carry the same load
// instead of seven of them converging on one root.
//
// RCCL is resolved at run time (dlopen of librccl.so.1: the copy a host process already loaded -- PyTorch ships one --
// or ROCm's), so librt_hip.so has no link-time dependency on it and single-GPU users never load it; the dozen entry points
// it uses are declared below, so building this file needs no RCCL headers either.  Two ways to form
// the group: one process per GPU (rt_comm_unique_id on one rank, broadcast by the host's own means, rt_comm_init_rank
// on every rank), or one process driving all devices (rt_comm_init_all + the *_all calls, which wrap the per-device
// calls in ncclGroupStart / ncclGroupEnd).
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "../../include/rt_hip.h"

// ---- the part of RCCL's C interface this file uses (rccl.h of ROCm 7: the NCCL 2 ABI) ----
typedef struct ncclComm* ncclComm_t;
#define NCCL_UNIQUE_ID_BYTES 128
typedef struct { char internal[NCCL_UNIQUE_ID_BYTES]; } ncclUniqueId;
typedef enum { ncclSuccess = 0 } ncclResult_t;
typedef enum { ncclUint8 = 1 } ncclDataType_t;

namespace {

struct Rccl {
    void* lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Gather)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GetVersion)(int*) = nullptr;
    // optional (rt_comm_info reports what the library itself says when they exist)
    ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
    ncclResult_t (*CommUserRank)(const ncclComm_t, int*) = nullptr;
    std::string load_error;
};

thread_local std::string g_comm_error;
// the same text for readers on OTHER threads (a watchdog that reports where a stuck main thread last failed): the most recent
// RT_E_COMM of any thread, behind a mutex; rt_comm_last_error_any() copies it into the caller's thread-local buffer
std::mutex g_comm_error_any_mutex;
std::string g_comm_error_any;
thread_local std::string g_comm_error_any_copy;
void set_comm_error(const std::string& text)
{
    g_comm_error = text;
    std::lock_guard<std::mutex> lock(g_comm_error_any_mutex);
    g_comm_error_any = text;
}

Rccl* rccl()
{
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        // RT_RCCL_LIBRARY=<path>: load that library and no other (the tests' in-process mock, tests/mock_rccl; an RCCL build
        // elsewhere).  If it cannot be loaded the communicator is unavailable -- never a silent switch to the system's library.
        const char* forced = getenv("RT_RCCL_LIBRARY");
        if (forced && *forced) {
            r.lib = dlopen(forced, RTLD_NOW | RTLD_LOCAL);
            if (!r.lib) { const char* de = dlerror(); r.load_error = std::string("RT_RCCL_LIBRARY=") + forced + " could not be loaded: " + (de ? de : "?"); return; }
        }
        const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char* n : names) {
            if (r.lib) break;
            r.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        }
        if (!r.lib) { r.load_error = "librccl.so.1 could not be loaded"; return; }
        bool ok = true;
        std::string missing;
        auto sym = [&](const char* n) { void* p = dlsym(r.lib, n); if (!p) { ok = false; missing += std::string(missing.empty() ? "" : ", ") + n; } return p; };
        r.GetUniqueId = (decltype(r.GetUniqueId))sym("ncclGetUniqueId");
        r.CommInitRank = (decltype(r.CommInitRank))sym("ncclCommInitRank");
        r.CommInitAll = (decltype(r.CommInitAll))sym("ncclCommInitAll");
        r.CommDestroy = (decltype(r.CommDestroy))sym("ncclCommDestroy");
        r.GetErrorString = (decltype(r.GetErrorString))sym("ncclGetErrorString");
        r.GroupStart = (decltype(r.GroupStart))sym("ncclGroupStart");
        r.GroupEnd = (decltype(r.GroupEnd))sym("ncclGroupEnd");
        r.Gather = (decltype(r.Gather))sym("ncclGather");
        r.Send = (decltype(r.Send))sym("ncclSend");
        r.Recv = (decltype(r.Recv))sym("ncclRecv");
        r.GetVersion = (decltype(r.GetVersion))sym("ncclGetVersion");
        if (!ok) {
            dlclose(r.lib); r.lib = nullptr;
            r.load_error = std::string(forced && *forced ? forced : "librccl.so.1") + " lacks " + missing;
            return;
        }
        r.CommCount = (decltype(r.CommCount))dlsym(r.lib, "ncclCommCount");
        r.CommUserRank = (decltype(r.CommUserRank))dlsym(r.lib, "ncclCommUserRank");
    });
    if (!r.lib) { set_comm_error(r.load_error); return nullptr; }
    return &r;
}

int comm_fail(Rccl* r, ncclResult_t e, const char* what)
{
    set_comm_error(std::string(what) + ": " + (r ? r->GetErrorString(e) : "?"));
    return RT_E_COMM;
}

#define RT_NCCL(r, expr, what) do { ncclResult_t e_ = (expr); if (e_ != ncclSuccess) return comm_fail(r, e_, what); } while (0)
#define RT_HIP(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) return (int)e_; } while (0)

}  // namespace

struct RtComm {
    ncclComm_t comm = nullptr;
    int rank = 0, num_ranks = 1, device = 0;
    // scratch of rt_render_tiled: this rank's stripe buffer and, on a root, the rank-major gathered buffer (grow-only)
    uint8_t* d_local = nullptr; size_t local_bytes = 0;
    uint8_t* d_gathered = nullptr; size_t gathered_bytes = 0;
    // One pair of scratch buffers serves every call, so a call must not start before the previous call's last use of them
    // (its gather, or on the root its un-stripe pass) has finished.  On one stream that is stream order; a call on ANOTHER
    // stream waits for this event first (frames alternated between two streams would otherwise overwrite d_local under a
    // gather still reading it).
    hipEvent_t scratch_done = nullptr;
    hipStream_t scratch_stream = nullptr;
    bool scratch_used = false;
};

namespace {

int grow(uint8_t** p, size_t* have, size_t need)
{
    if (*have >= need) return RT_OK;
    (void)hipFree(*p);                                          // (synchronises with work in flight on the old buffer)
    *p = nullptr; *have = 0;
    RT_HIP(hipMalloc((void**)p, need));
    *have = need;
    return RT_OK;
}

// before a call touches the scratch on `stream`: order it behind the previous call if that ran on another stream
int scratch_acquire(RtComm* c, hipStream_t stream)
{
    if (c->scratch_used && c->scratch_stream != stream) RT_HIP(hipStreamWaitEvent(stream, c->scratch_done, 0));
    return RT_OK;
}
// after a call's last use of the scratch has been issued on `stream`
int scratch_release(RtComm* c, hipStream_t stream)
{
    if (!c->scratch_done) RT_HIP(hipEventCreateWithFlags(&c->scratch_done, hipEventDisableTiming));
    RT_HIP(hipEventRecord(c->scratch_done, stream));
    c->scratch_stream = stream; c->scratch_used = true;
    return RT_OK;
}

struct Tiling {                                                 // stripe bookkeeping of one frame size
    int32_t max_rows = 0;
    size_t local_pitch = 0, rank_bytes = 0;
};

int tiling_of(const RtCameraParams* cam, int32_t stripe_rows, int32_t num_ranks, Tiling& t)
{
    if (!cam || cam->width <= 0 || cam->height <= 0 || stripe_rows <= 0 || num_ranks <= 0) return RT_E_INVALID;
    t.max_rows = 0;
    for (int r = 0; r < num_ranks; r++) {
        int32_t rows = 0;
        int rc = rt_stripe_rows(cam->height, stripe_rows, r, num_ranks, &rows);
        if (rc) return rc;
        t.max_rows = std::max(t.max_rows, rows);
    }
    t.local_pitch = ((size_t)cam->width * 3 + 15) / 16 * 16;    // 16-byte rows: the un-stripe pass copies uint4s
    t.rank_bytes = (size_t)t.max_rows * t.local_pitch;
    return RT_OK;
}

int render_local(RtScene* scene, RtComm* c, const RtCameraParams* cam, const RtRenderOptions* opts, const Tiling& t,
                 int32_t stripe_rows, hipStream_t stream)
{
    int rc = scratch_acquire(c, stream);
    if (rc) return rc;
    if ((rc = grow(&c->d_local, &c->local_bytes, std::max<size_t>(t.rank_bytes, 16)))) return rc;
    if (opts && (opts->spp != 1 || opts->bounces != 0 || opts->lighting != 0))
        return rt_render_ex_stripes(scene, cam, opts, c->d_local, t.local_pitch, stripe_rows, c->rank, c->num_ranks, stream, 0);
    return rt_render_stripes(scene, cam, c->d_local, t.local_pitch, stripe_rows, c->rank, c->num_ranks, stream, 0);
}

}  // namespace

extern "C" {

const char* rt_comm_last_error(void) { return g_comm_error.c_str(); }
const char* rt_comm_last_error_any(void)
{
    std::lock_guard<std::mutex> lock(g_comm_error_any_mutex);
    g_comm_error_any_copy = g_comm_error_any;
    return g_comm_error_any_copy.c_str();
}

int rt_comm_available(int32_t* rccl_version)
{
    Rccl* r = rccl();
    if (!r) return RT_E_COMM;
    int v = 0;
    RT_NCCL(r, r->GetVersion(&v), "ncclGetVersion");
    if (rccl_version) *rccl_version = v;
    return RT_OK;
}

int rt_comm_unique_id(uint8_t* id)
{
    static_assert(RT_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "id size");
    Rccl* r = rccl();
    if (!r) return RT_E_COMM;
    if (!id) return RT_E_INVALID;
    ncclUniqueId u;
    RT_NCCL(r, r->GetUniqueId(&u), "ncclGetUniqueId");
    memcpy(id, u.internal, RT_COMM_ID_BYTES);
    return RT_OK;
}

int rt_comm_init_rank(const uint8_t* id, int32_t rank, int32_t num_ranks, RtComm** out)
{
    Rccl* r = rccl();
    if (!r) return RT_E_COMM;
    if (!id || !out || num_ranks < 1 || rank < 0 || rank >= num_ranks) return RT_E_INVALID;
    *out = nullptr;
    RtComm* c = new (std::nothrow) RtComm;
    if (!c) return RT_E_NOMEM;
    c->rank = rank; c->num_ranks = num_ranks;
    hipError_t he = hipGetDevice(&c->device);
    if (he != hipSuccess) { delete c; return he == hipErrorNoDevice ? RT_E_NODEVICE : (int)he; }
    ncclUniqueId u;
    memcpy(u.internal, id, RT_COMM_ID_BYTES);
    ncclResult_t e = r->CommInitRank(&c->comm, num_ranks, u, rank);
    if (e != ncclSuccess) { delete c; return comm_fail(r, e, "ncclCommInitRank"); }
    *out = c;
    return RT_OK;
}

int rt_comm_init_all(const int32_t* devices, int32_t num_devices, RtComm** out)
{
    Rccl* r = rccl();
    if (!r) return RT_E_COMM;
    if (!out || num_devices < 1) return RT_E_INVALID;
    std::vector<int> devs((size_t)num_devices);
    for (int i = 0; i < num_devices; i++) devs[i] = devices ? devices[i] : i;
    std::vector<ncclComm_t> comms((size_t)num_devices, nullptr);
    RT_NCCL(r, r->CommInitAll(comms.data(), num_devices, devs.data()), "ncclCommInitAll");
    for (int i = 0; i < num_devices; i++) {
        RtComm* c = new (std::nothrow) RtComm;
        if (!c) { for (int k = 0; k < i; k++) delete out[k]; for (auto cm : comms) (void)r->CommDestroy(cm); return RT_E_NOMEM; }
        c->comm = comms[i]; c->rank = i; c->num_ranks = num_devices; c->device = devs[i];
        out[i] = c;
    }
    return RT_OK;
}

int rt_comm_info(const RtComm* c, int32_t* rank, int32_t* num_ranks, int32_t* device)
{
    if (!c) return RT_E_INVALID;
    int rk = c->rank, n = c->num_ranks;
    // what the communicator itself says (ncclCommUserRank / ncclCommCount), when the library has the two queries: a
    // caller can check that RCCL formed the group it asked for
    Rccl* r = rccl();
    if (r && c->comm && r->CommCount && r->CommUserRank) {
        RT_NCCL(r, r->CommCount(c->comm, &n), "ncclCommCount");
        RT_NCCL(r, r->CommUserRank(c->comm, &rk), "ncclCommUserRank");
    }
    if (rank) *rank = rk;
    if (num_ranks) *num_ranks = n;
    if (device) *device = c->device;
    return RT_OK;
}

int rt_comm_destroy(RtComm* c)
{
    if (!c) return RT_OK;
    int prev = 0;
    const bool switched = hipGetDevice(&prev) == hipSuccess && prev != c->device && hipSetDevice(c->device) == hipSuccess;
    (void)hipFree(c->d_local); (void)hipFree(c->d_gathered);
    if (c->scratch_done) (void)hipEventDestroy(c->scratch_done);
    Rccl* r = rccl();
    if (r && c->comm) (void)r->CommDestroy(c->comm);
    if (switched) (void)hipSetDevice(prev);
    delete c;
    return RT_OK;
}

int rt_group_start(void) { Rccl* r = rccl(); if (!r) return RT_E_COMM; RT_NCCL(r, r->GroupStart(), "ncclGroupStart"); return RT_OK; }
int rt_group_end(void) { Rccl* r = rccl(); if (!r) return RT_E_COMM; RT_NCCL(r, r->GroupEnd(), "ncclGroupEnd"); return RT_OK; }

int rt_gather(RtComm* c, const void* d_send, size_t bytes, void* d_recv, int32_t root, void* stream)
{
    Rccl* r = rccl();
    if (!r) return RT_E_COMM;
    if (!c || !d_send || root < 0 || root >= c->num_ranks || (c->rank == root && !d_recv)) return RT_E_INVALID;
    RT_NCCL(r, r->Gather(d_send, d_recv, bytes, ncclUint8, root, c->comm, (hipStream_t)stream), "ncclGather");
    return RT_OK;
}

int rt_all_to_all(RtComm* c, const void* d_send, const size_t* send_bytes, const size_t* send_offsets,
                  void* d_recv, const size_t* recv_bytes, const size_t* recv_offsets, void* stream)
{
    Rccl* r = rccl();
    if (!r) return RT_E_COMM;
    if (!c || !d_send || !d_recv || !send_bytes || !send_offsets || !recv_bytes || !recv_offsets) return RT_E_INVALID;
    if (send_bytes[c->rank] != recv_bytes[c->rank]) return RT_E_INVALID;
    // this rank's own block never leaves the device: a plain copy on the same stream
    if (send_bytes[c->rank])
        RT_HIP(hipMemcpyAsync((uint8_t*)d_recv + recv_offsets[c->rank], (const uint8_t*)d_send + send_offsets[c->rank], send_bytes[c->rank],
                              hipMemcpyDeviceToDevice, (hipStream_t)stream));
    if (c->num_ranks == 1) return RT_OK;
    // one fused group of point-to-point transfers: every pair of ranks exchanges over its own xGMI link
    RT_NCCL(r, r->GroupStart(), "ncclGroupStart");
    ncclResult_t e = ncclSuccess;
    for (int p = 0; p < c->num_ranks && e == ncclSuccess; p++) {
        if (p == c->rank) continue;
        if (send_bytes[p]) e = r->Send((const uint8_t*)d_send + send_offsets[p], send_bytes[p], ncclUint8, p, c->comm, (hipStream_t)stream);
        if (e == ncclSuccess && recv_bytes[p]) e = r->Recv((uint8_t*)d_recv + recv_offsets[p], recv_bytes[p], ncclUint8, p, c->comm, (hipStream_t)stream);
    }
    ncclResult_t e2 = r->GroupEnd();
    if (e != ncclSuccess) return comm_fail(r, e, "ncclSend/ncclRecv");
    RT_NCCL(r, e2, "ncclGroupEnd");
    return RT_OK;
}

int rt_render_tiled(RtScene* scene, RtComm* c, const RtCameraParams* cam, const RtRenderOptions* opts, uint8_t* d_img, size_t pitch,
                    int32_t stripe_rows, int32_t root, void* stream, int synchronize)
{
    if (!scene || !c || !cam || root < 0 || root >= c->num_ranks) return RT_E_INVALID;
    const bool is_root = c->rank == root;
    if (is_root && (!d_img || pitch < (size_t)cam->width * 3)) return RT_E_INVALID;
    hipStream_t st = (hipStream_t)stream;
    int rc;
    if (c->num_ranks == 1) {                                    // nothing to exchange: render straight into the frame
        if (opts && (opts->spp != 1 || opts->bounces != 0 || opts->lighting != 0))
            return rt_render_ex(scene, cam, opts, d_img, pitch, nullptr, stream, synchronize);
        return rt_render(scene, cam, d_img, pitch, stream, synchronize);
    }
    Tiling t;
    if ((rc = tiling_of(cam, stripe_rows, c->num_ranks, t))) return rc;
    if ((rc = render_local(scene, c, cam, opts, t, stripe_rows, st))) return rc;
    if (is_root && (rc = grow(&c->d_gathered, &c->gathered_bytes, t.rank_bytes * (size_t)c->num_ranks))) return rc;
    if ((rc = rt_gather(c, c->d_local, t.rank_bytes, is_root ? c->d_gathered : nullptr, root, stream))) return rc;
    if (is_root && (rc = rt_unstripe(c->d_gathered, t.local_pitch, t.rank_bytes, d_img, pitch, cam->width, cam->height, stripe_rows,
                                     c->num_ranks, stream))) return rc;
    if ((rc = scratch_release(c, st))) return rc;
    if (synchronize) RT_HIP(hipStreamSynchronize(st));
    return RT_OK;
}

int rt_render_tiled_all(RtScene* const* scenes, RtComm* const* comms, int32_t num_ranks, const RtCameraParams* cam,
                        const RtRenderOptions* opts, uint8_t* d_img, size_t pitch, int32_t stripe_rows, int32_t root,
                        void* const* streams, int synchronize)
{
    if (!scenes || !comms || !cam || num_ranks < 1 || root < 0 || root >= num_ranks || !d_img || pitch < (size_t)cam->width * 3) return RT_E_INVALID;
    for (int r = 0; r < num_ranks; r++)
        if (!scenes[r] || !comms[r] || comms[r]->rank != r || comms[r]->num_ranks != num_ranks) return RT_E_INVALID;
    int prev = 0;
    RT_HIP(hipGetDevice(&prev));
    auto stream_of = [&](int r) { return streams ? streams[r] : nullptr; };
    int rc = RT_OK;
    if (num_ranks == 1) {
        RT_HIP(hipSetDevice(comms[0]->device));
        rc = rt_render_tiled(scenes[0], comms[0], cam, opts, d_img, pitch, stripe_rows, 0, stream_of(0), synchronize);
        (void)hipSetDevice(prev);
        return rc;
    }
    Tiling t;
    if ((rc = tiling_of(cam, stripe_rows, num_ranks, t))) return rc;
    // 1. every device renders its stripes (asynchronous: the devices run concurrently)
    for (int r = 0; r < num_ranks && rc == RT_OK; r++) {
        if (hipSetDevice(comms[r]->device) != hipSuccess) { rc = RT_E_INVALID; break; }
        rc = render_local(scenes[r], comms[r], cam, opts, t, stripe_rows, (hipStream_t)stream_of(r));
        if (rc == RT_OK && r == root) rc = grow(&comms[r]->d_gathered, &comms[r]->gathered_bytes, t.rank_bytes * (size_t)num_ranks);
    }
    // 2. the gather, one call per device inside one group (a single thread drives all communicators)
    if (rc == RT_OK) {
        rc = rt_group_start();
        for (int r = 0; r < num_ranks && rc == RT_OK; r++) {
            if (hipSetDevice(comms[r]->device) != hipSuccess) { rc = RT_E_INVALID; break; }
            rc = rt_gather(comms[r], comms[r]->d_local, t.rank_bytes, r == root ? comms[r]->d_gathered : nullptr, root, stream_of(r));
        }
        const int rc2 = rt_group_end();
        if (rc == RT_OK) rc = rc2;
    }
    // 3. the root puts the rows back into frame order
    if (rc == RT_OK && hipSetDevice(comms[root]->device) == hipSuccess) {
        rc = rt_unstripe(comms[root]->d_gathered, t.local_pitch, t.rank_bytes, d_img, pitch, cam->width, cam->height, stripe_rows,
                         num_ranks, stream_of(root));
    }
    for (int r = 0; r < num_ranks && rc == RT_OK; r++) {            // every rank's scratch is busy until its gather (the root: its un-stripe) is done
        if (hipSetDevice(comms[r]->device) != hipSuccess) { rc = RT_E_INVALID; break; }
        rc = scratch_release(comms[r], (hipStream_t)stream_of(r));
    }
    if (rc == RT_OK && synchronize && hipSetDevice(comms[root]->device) == hipSuccess) rc = (int)hipStreamSynchronize((hipStream_t)stream_of(root));
    (void)hipSetDevice(prev);
    return rc;
}

}  // extern "C"
