// rt_device_types.h -- device-side scene layout of librt_hip.so (see DESIGN.md "Data layout in HBM").
#pragma once
#include <cstdint>
#include "rt_math.h"

namespace rt {

// Traversal-stack entry / child reference, 32 bits:
//   >= 0                      interior node: index of its record (scene-wide record array)
//   bit 31 set                leaf: bits 0..25 = index of its first triangle record in the same array,
//                             bits 26..30 = triangle count 0..30, or 31 = read leaf_count[slot]
constexpr int32_t kLeafFlag = (int32_t)0x80000000u;
constexpr int kSlotBits = 26;
constexpr int32_t kSlotMask = (1 << kSlotBits) - 1;
constexpr int kMaxStack = 64;
constexpr int32_t kBoxUnordered = 1;    // mesh_flags bit 0: the octant-specialised slab test must not be used on this mesh

// the twelve box words of an interior record: true when both boxes have min <= max on every axis (false for any NaN)
RT_HD bool boxes_ordered(const float* w)
{
    return w[0] <= w[3] && w[1] <= w[4] && w[2] <= w[5] && w[6] <= w[9] && w[7] <= w[10] && w[8] <= w[11];
}

// Interior node: 64 B = 4 x float4, holds BOTH children's boxes so one pop costs one 64-B
// record instead of the reference's three 48-B d_BVHTree loads (raycast.cu:62,69,70).
//   q0 = a.min.x a.min.y a.min.z a.max.x
//   q1 = a.max.y a.max.z b.min.x b.min.y
//   q2 = b.min.z b.max.x b.max.y b.max.z
//   q3 = ref_a   ref_b   (2 spare words)
//
// Triangle record: 64 B = 4 x float4 in leaf ("slot") order, so a leaf is a contiguous range
// and the per-leaf index lists of BVHTree.hpp:97-111 disappear from the traversal.
//   t0 = v0.x v0.y v0.z n.x
//   t1 = n.y  n.z  e0.x e0.y          e0 = v2 - v0, e1 = v1 - v0   (TrianglePrimitive.hpp:154-155)
//   t2 = e0.z e1.x e1.y e1.z
//   t3 = dot00 dot01 dot11 invDenom   (TrianglePrimitive.hpp:158-164: ray-independent, so
//                                      precomputed on the host with the same fp32 operations)

struct DevInstance {            // 144 B
    Q4 q_rot;                   // euler2quat(rotation)          raycast.cu:33
    Q4 q_pose;                  // euler2quat(pose ypr)          raycast.cu:40
    Q4 q_inv_pose;              // euler2quat(inv_pose ypr)      raycast.cu:102
    Q4 q_inv_rot;               // euler2quat(inv_rotation)      raycast.cu:115
    float pose_xyz[3];
    float inv_pose_xyz[3];
    float scale[3];
    float inv_scale[3];
    int32_t root_ref;           // stack entry of the mesh root
    int32_t material_index;
    int32_t mesh_index;
    int32_t exact_uv;           // mesh has uv values that could make uv.x == FLT_MAX (raycast.cu:96)
    int32_t identity_inv;       // scale == 1, inv_pose == 0, q_inv_pose == (1,0,0,0): mesh -> world is the identity
    int32_t unit_inv;           // scale == 1, q_inv_pose == (1,0,0,0): mesh -> world is a translation (or the identity)
    int32_t pad_[2];
};

struct DevMaterial {            // Material.hpp:6-16
    float albedo[3];
    int32_t texture_width;
    int32_t texture_height;
    uint32_t texture_pitch;
    const uint8_t* texture;
    float roughness, metallic;  // read only by the extension kernel (dead in the reference)
};

// wavefront path queues (ex_wave_kernel): a segment = the 64 slots one wave may push to, a group = exq_k segments
constexpr int kExMaxGroup = 32;
constexpr int kExPlanesA = 5, kExPlanesS = 7;

constexpr int kMaxViewInstances = 8;    // scenes of more instances render without view records (RenderParams::view_inst_off)
constexpr int kMaxBatch = 32;   // frames per launch (rt_render_batch); 32 x 96 B of per-frame parameters keep the kernel
                                // arguments under the 4 KB limit

struct FrameParams {            // what differs between the frames of one batched launch
    float kinv[9];
    float D[4];
    float origin[3];
    Q4 q_cam;                   // euler2quat(inv_camera_pose ypr), raycast.cu:185
    uint8_t* img;
    // stripes: whose stripes this frame's launch renders and how many rows they have.  Both are the launch's own (RenderParams::rank,
    // local_rows) unless the stripe owner ROTATES with the frame index (rt_render_stripes_batch_rotating): a rank then renders every
    // stripe class in turn, so the ranks' shares of a group of frames are equal whatever the frame shows
    int32_t rank, local_rows;
};

struct RenderParams {
    int32_t width, height;
    int32_t num_frames;         // grid.y
    FrameParams frames[kMaxBatch];
    const float4* records;      // interior-node and triangle records (64 B each), one index space
    const float* tri_uv;        // [slot][3][2]
    const int32_t* tri_id;      // [slot] -> caller's triangle index within its mesh
    const int32_t* leaf_count;  // [slot] count of the leaf that starts at slot (only read for count > 30)
    const int32_t* mesh_flags;  // [mesh] bit 0: some interior record of the mesh holds a box with min > max or a NaN (kBoxUnordered)
    const DevInstance* instances;
    const DevMaterial* materials;
    int32_t num_instances;
    int32_t stack_depth;        // LDS stack entries per lane
    uint64_t pitch;
    // stripes: local row ly is frame row ((ly / stripe_rows) * num_ranks + rank) * stripe_rows + ly % stripe_rows
    int32_t local_rows, stripe_rows, rank, num_ranks;   // (local_rows = the most rows any frame of the launch has: the grid's height)
    int32_t tiles_x, tiles_y;   // 16x16-pixel workgroup tiles over width x local_rows
    unsigned long long* trace;  // diagnostics: per-wave {start, end, hw id, tile} stamps, or null
    // single-frame launches (render_kernel<.., ORDERED>): heavy-tiles-first dispatch from the previous frame's costs
    const int32_t* tile_order;  // workgroup b renders tile tile_order[b] (null = b)
    int32_t* tile_cost;         // [tile] loop iterations of the tile's longest lane (null = not recorded)
    // extension kernel (rt_render_ex): samples per pixel, specular bounces, sun + shadow pass, optional pops plane
    int32_t spp, bounces, lighting;
    int32_t sample_base;        // this launch renders sample indices sample_base + blockIdx.y
    float4* ex_samples;         // [chunk][local_rows * width] radiance xyz + node pops (int bits) of one sample
    float4* ex_acc;             // [local_rows * width] running sums between chunks
    int32_t* total_pops;
    // render_ex_kernel<.., PX>: the samples of a pixel share a wave (64 lanes = px_pw x px_ph pixels x px_n sample slots)
    int32_t px_n, px_count;     // sample slots per pixel in a wave (4..64, a power of two), samples of this launch (<= px_n)
    int32_t px_pw, px_ph;       // pixels of a wave; a workgroup is 2 x 2 waves
    int32_t px_first, px_last;  // first / last chunk of the frame's samples (running sums wait in ex_acc in between)
    // render_ex_kernel<.., PHASE 1 / 2>: the camera ray's hit per lane, two planes of ex_rec_lanes float4 -- (slot, instance, u, v) and
    // (location, pops) -- for the workgroups wg_base .. wg_base + gridDim.x of the frame's grid
    float4* ex_rec;
    uint32_t ex_rec_lanes;
    int32_t wg_base;
    // wavefront form of the extension renderer (ex_wave_kernel): path queues between casts, see rt_kernels.hip
    float4* exq_s;              // shadow-ray items: kExPlanesS planes of exq_nseg * 64 float4
    float4* exq_a[3];           // bounce-ray items: kExPlanesA planes each; three arrays in rotation (exq_in1 / in2 / out)
    int32_t* exq_cnt_s;         // [segment] items in the segment (0..64)
    int32_t* exq_cnt_a[3];
    int32_t exq_nseg;           // segments per queue array = waves of the primary launch
    int32_t exq_k;              // segments per group (<= 32): a group's paths stay in the group through every stage
    int32_t exq_in1, exq_in2;   // A(depth) = the segments of exq_a[in1] (pushed by the cast launch of depth - 1) then those of exq_a[in2]
                                // (pushed by its shadow launch; -1 = none)
    int32_t exq_out;            // the exq_a array this launch pushes A(depth + 1) items to
    int32_t depth;              // depth of the rays this launch casts
    int32_t gen_samples;        // primary launch: sample indices in this chunk
    int32_t gen_spw;            // primary launch: sample indices per workgroup (4, 2 or 1: one, two or four 8x8 quads)
    // render_kernel<.., VIEW>: the view record of interior record `e` of instance i in frame f is at byte offset
    // view_base + f * view_frame_stride + view_inst_off[i] + e * 64 from `records` (32-bit arithmetic; RtScene::ViewPool)
    uint32_t view_base, view_frame_stride;
    int32_t view_inst_off[kMaxViewInstances];
    unsigned long long* loop_stats;     // render_kernel<.., STATS> (rt_scene_loop_stats): RT_LOOP_WORDS counters, null in every other launch
    // parity planes (tight [height][width], frame coordinates), any may be null
    int32_t *hit_instance, *hit_triangle, *node_pops, *aabb_tests, *tri_tests, *inside_hits;
};

static_assert(sizeof(RenderParams) <= 4096, "kernel arguments must fit in 4 KB");

}  // namespace rt
