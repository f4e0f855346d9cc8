// rt_scene_internal.h -- the scene handle of the C-ABI (RtScene of include/rt_hip.h) as the library's own translation units see it:
// rt_kernels.hip (upload, render, refit) and rt_bvh_build.hip (device-resident rebuild).  Not part of the interface.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <mutex>
#include <vector>

#include "rt_device_types.h"

using rt::DevInstance;
using rt::DevMaterial;

struct RtScene {
    // Every entry point that reads or replaces the scene's device arrays holds this from its first look at them until its last kernel
    // is queued (never across a wait for the device, except where a call's contract is to block: growth of the view pool, scratch
    // re-allocation): calls on one scene from several host threads are serialised on the HOST side only -- a launch is a few
    // microseconds -- and their GPU work overlaps as their streams allow.  Recursive: rt_render -> rt_render_batch, rt_render_tiled -> ...
    std::recursive_mutex call_mu;
    int device = 0;
    float4* d_records = nullptr;                 // interior-node and triangle records, one index space
    float* d_tri_uv = nullptr;
    int32_t* d_tri_id = nullptr;
    int32_t* d_leaf_count = nullptr;
    int32_t* d_mesh_flags = nullptr;             // per mesh, rt::kBoxUnordered: written by every writer of interior records
    DevInstance* d_instances = nullptr;
    DevMaterial* d_materials = nullptr;
    std::vector<uint8_t*> d_textures;
    std::vector<DevInstance> instances;          // host mirror (for update_instance)
    std::vector<int32_t> mesh_root_ref;          // per mesh
    std::vector<int32_t> mesh_exact_uv;
    // per mesh, for rt_scene_refit_mesh: its slot range, triangle count, and its interior records grouped by tree level
    // Layout of a mesh's part of the record array: int_cap interior records from node_base (the tree's interior nodes in
    // pre-order, then unused ones), slot_cap triangle records from slot_base = node_base + int_cap.  The capacities leave room
    // for ANY tree over the mesh's triangles (at most n - 1 interior nodes, n slots), so that rt_scene_rebuild_mesh_device can
    // write a new tree in place.
    struct MeshRefit {
        int32_t node_base = 0, int_cap = 0, slot_cap = 0, levels = 1;
        int32_t slot_base = 0, num_slots = 0, num_triangles = 0;
        std::vector<int32_t> sched;              // interior record indices, deepest level first
        std::vector<int32_t> level_end;          // sched[level_end[k-1] .. level_end[k]) is one level
        int32_t* d_sched = nullptr;              // int_cap entries
    };
    std::vector<MeshRefit> mesh_refit;
    float* d_refit_scratch = nullptr;            // vertices + normals of the mesh being refitted (grow-only)
    size_t refit_scratch_bytes = 0;
    int32_t num_materials = 0;
    int32_t max_stack = 1;
    size_t device_bytes = 0;
    float4* d_ex_scratch = nullptr;              // extension renders: running sums + one chunk of samples (grow-only)
    size_t ex_scratch_bytes = 0;
    // heavy-first dispatch of single-frame launches (render_kernel<.., ORDERED>, tile_sort_kernel): one order state per frame
    // size, a few of them cached -- two cameras of different sizes, or whole frames next to a rank's stripes, alternate on one
    // scene without ever meeting each other's state (a single state would be torn down and rebuilt, with a device-wide
    // synchronise in hipFree, on every change of size)
    struct TileOrder {
        int tiles_x = 0, tiles_y = 0, ntiles = 0;
        int32_t *d_cost = nullptr, *d_keys = nullptr, *d_order[2] = {nullptr, nullptr};
        int cur = -1;                            // order buffer renders read (-1: none sorted yet -> natural order)
        bool pending = false;                    // a sort into d_order[target] is in flight on sort_stream
        int target = 0;
        hipEvent_t sort_done = nullptr;
        uint64_t last_used = 0;
        uint64_t launches = 0, sorted_at = 0;    // ordered launches of this size so far / at the last sort issued
        // the last ordered launch on each stream that uses this state (a sort waits for all of them)
        // (done = recorded behind the stream's launches when a sort is issued or the slot is wanted for another stream; recorded = it
        // has been, and is worth waiting for; dirty = the stream has launched since)
        struct Seen { hipStream_t stream = nullptr; hipEvent_t done = nullptr; bool used = false, dirty = false, recorded = false; uint64_t tick = 0; } seen[4];
    };
    // rt_render_overlapped (Camera::render_scene's asynchronous default-stream form): two library-owned BLOCKING streams that
    // consecutive frames alternate between, so that the next frame's costly tiles fill the chip while the previous frame's
    // last workgroups drain (the reference's own loop issues two renders per synchronise, kernel.cu:277-279).  Blocking
    // streams keep the default stream's ordering with everything the caller does on the null stream; two launches that write
    // overlapping image memory are ordered here: written[k] = the images of launches on stream k that the OTHER stream
    // has not been ordered after.
    struct Overlap {
        std::mutex m;
        hipStream_t stream[2] = {nullptr, nullptr};
        hipEvent_t done[2] = {nullptr, nullptr};             // recorded on stream k when the other stream has to wait for it
        struct Range { uintptr_t lo, hi; };
        std::vector<Range> written[2];
        int next = 0;
        uint64_t launches = 0, waits = 0;                    // (diagnostics: rt_render_overlapped_stats)
    } overlap;
    // View records (render_kernel<.., VIEW>): primary rays of one frame share their origin, so `box - origin` of an interior
    // record (twelve subtractions per node visit and lane) depends on the frame and the instance only.  A small kernel writes
    // them once per frame -- the frame's "view" of the interior records -- and the traversal reads those instead.  The views
    // live BEHIND the records in the same allocation (d_records is re-allocated with the pool as its tail the first time a
    // launch qualifies, and when a launch brings more frames than a slot holds), so that a lane's fetch stays one 32-bit
    // offset from one base whether it reads a triangle or a view record.  A slot = the views of the frames of one launch;
    // a slot is reused by launches on the stream it was last used on (stream order) or once its event has completed.
    // Round 6: the pool is sized by rt_scene_reserve_views (an application that batches says so once, after upload) or, without that,
    // grows to the batch sizes actually seen (4, 8, 16, 32 frames per slot) inside the first launch that needs more -- a call that then
    // blocks until the device is idle (documented at rt_render_batch).  One to three slots, as many as fit the budget
    // (RT_VIEW_MAX_BYTES, default 1 GiB).  The block it replaces is freed at once: call_mu keeps other launches of the scene out and
    // the device has been drained, so nothing can still read it.
    struct ViewPool {
        std::mutex m;                            // (statistics only: the pool itself is protected by RtScene::call_mu)
        bool decided = false, usable = false;    // static eligibility of the scene (instances, sizes), decided at the first launch
        int32_t frame_records = 0;               // interior-record capacity of all instances = records of one frame's view
        std::vector<int32_t> inst_first;         // per instance: first view record within a frame's view
        size_t base_bytes = 0;                   // byte offset of the pool from d_records
        int32_t slot_frames = 0;                 // frames a slot holds (0 = no pool yet)
        int32_t slots = 0;                       // slots of the pool (1..3)
        bool reserved = false;                   // sized by rt_scene_reserve_views: launches never grow it
        struct Slot { hipStream_t stream = nullptr; hipEvent_t done = nullptr; bool used = false; } slot[3];
        uint64_t launches = 0, fallbacks = 0, grows = 0;
    } view;
    size_t records_bytes = 0;                    // size of the record array proper (the head of the d_records allocation)
    size_t records_alloc_bytes = 0;              // size of the d_records allocation: records, padding, view pool
    struct TileOrderCache {
        std::mutex m;
        hipStream_t sort_stream = nullptr;
        uint64_t tick = 0;
        TileOrder entry[4];
    } order;
};

