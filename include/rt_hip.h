/*
 * rt_hip.h -- C-ABI of librt_hip.so: the MI355X (gfx950) raycast hot path.
 *
 * This is the drop-in boundary for the reference's per-pixel raycast.  The reference has no
 * FFI layer; its boundary is the host C++ that uploads the scene and launches the one
 * kernel.  Each entry point below names the reference interface it replaces (paths relative
 * to AFIDclan/cuda-raytracing CudaRaytracer/).  Plain pointers and sizes only; no C++ or
 * torch types; every function returns 0 on success, a positive hipError_t, or a negative
 * RT_E_* code, and never throws.  The caller owns image buffers; the library owns scene
 * buffers.  All `stream` arguments are a hipStream_t passed as void* (NULL = default stream).
 *
 * Environment (diagnostics and tests only): RT_TRACE_FILE=<path> makes every render launch synchronise and write
 * per-wave start / end stamps there (tools/trace_one.py), with RT_TRACE_PROF=1 through a stamped copy of the kernel;
 * RT_EX_SCRATCH_BYTES=<n> overrides the scratch budget of rt_render_ex (forces the chunked path);
 * RT_EX_SPLIT=1 renders the camera ray of a path with bounces or lighting in a launch of its own (bit-identical, measured slower);
 * RT_EX_WAVEFRONT=1 renders bounces / lighting with one cast per launch and path queues in between (bit-identical, slower
 * on the measured workloads: DESIGN.md section 3), RT_EX_GROUP=<4..32> sets its queue group size;
 * RT_TILE_ORDER=0 turns the heavy-first dispatch order of single-frame launches off; RT_BVH_LIBRARY_SCAN=1 makes
 * rt_bvh_build use the partition path of meshes above 1 M triangles; RT_BVH_DEBUG=1 prints its phase timings;
 * RT_BVH_SMALL=k (0..64) lowers the size of the subtrees one wave finishes on its own (0: level loop only; tests);
 * RT_RCCL_LIBRARY=<path> makes rt_comm_* load that library instead of librccl.so.1 (tests: an in-process mock); if it cannot
 * be loaded or lacks an entry point, rt_comm_* fail with RT_E_COMM (rt_comm_last_error() has the loader's message);
 * RT_RENDER_OVERLAP=0 makes rt_render_overlapped a plain default-stream launch; RT_TILE_SORT_INTERVAL=<n> sorts a new heavy-first
 * order every n-th single-frame launch (default 4: the events that order a sort behind the launches it must wait for are recorded only then); while RT_TEST_FAIL_UPLOAD=1 is set every rt_scene_upload fails with
 * RT_E_NOMEM before it touches anything (tests of the callers' error paths).
 * Round 6: RT_OVERLAP_PRIORITY=0 gives rt_render_overlapped two streams of equal priority (see there); RT_TILE_ORDER_STRIPES=0 keeps a
 * rank's thin striped batches in natural tile order; RT_VIEW_MAX_BYTES=<n> bounds a scene's view pool (default 1 GiB, see
 * rt_scene_reserve_views); RT_EX_SPLIT_BYTES=<n> bounds the records of the two-launch bounce form (tests: forces chunks).
 */
#ifndef RT_HIP_H
#define RT_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RT_ABI_VERSION 2

enum {
    RT_OK = 0,
    RT_E_INVALID = -1,      /* bad argument (null pointer, size, index out of range) */
    RT_E_NOMEM = -2,
    RT_E_DEPTH = -3,        /* BVH deeper than 64 levels (the reference's traversal stack holds 32 entries, raycast.cu:54) */
    RT_E_NODEVICE = -4,
    RT_E_COMM = -5          /* RCCL could not be loaded or returned an error: see rt_comm_last_error() */
};

/* One mesh as the reference uploads it: MeshPrimitive::to_device (MeshPrimitive.cpp:17-36) sends
 * the TrianglePrimitive array (TrianglePrimitive.hpp:8-11) and BVHTree::compile_tree
 * (BVHTree.hpp:364-383) the d_BVHTree array (BVHTree.hpp:18-26) plus one index list per leaf
 * (BVHTree.hpp:97-111).  Here the same data arrives flattened; the library re-lays it out for
 * the GPU (see DESIGN.md "Data layout in HBM").  All pointers are HOST pointers. */
typedef struct RtMeshDesc {
    int32_t num_triangles;
    const float *vertices;          /* [num_triangles][3][3]  TrianglePrimitive::vertices      */
    const float *normals;           /* [num_triangles][3]     TrianglePrimitive::normal        */
    const float *uvs;               /* [num_triangles][3][2]  TrianglePrimitive::uv_coords     */
    int32_t num_nodes;              /* node 0 is the root (MeshPrimitive.cpp:51).  0 = no tree given (the node and leaf
                                     * arrays are ignored): rt_scene_upload builds the reference's tree on the device,
                                     * in place -- the shortest way from triangles to a renderable scene            */
    const float *node_bounds;       /* [num_nodes][6]  min.xyz, max.xyz (d_BVHTree::min/max)   */
    const int32_t *node_children;   /* [num_nodes][2]  child_index_a/b; -1,-1 for a leaf       */
    const int32_t *node_leaf_first; /* [num_nodes]     offset of the leaf's list in leaf_indices */
    const int32_t *node_leaf_count; /* [num_nodes]     d_BVHTree::count_triangles (0 if interior) */
    int32_t num_leaf_indices;
    const int32_t *leaf_indices;    /* concatenated d_BVHTree::triangle_indices lists          */
} RtMeshDesc;

/* Material (Material.hpp:6-16).  texture = host BGR bytes (as cv::imread gives them,
 * Material.hpp:32) or NULL; texture_width == 0 selects the albedo path (raycast.cu:224). */
typedef struct RtMaterialDesc {
    float roughness;
    float albedo[3];
    float metallic;
    float illumination;
    const uint8_t *texture;
    int32_t texture_width, texture_height;
    size_t texture_pitch;
} RtMaterialDesc;

/* MeshInstance (MeshInstance.hpp:6-18), same field order and meaning; the inverse fields are
 * what MeshInstance::build_inv (MeshInstance.hpp:39-46) produces. */
typedef struct RtInstanceDesc {
    int32_t mesh_index;
    int32_t material_index;
    float pose[6];           /* lre: x y z yaw pitch roll (transforms.hpp:10-14) */
    float inv_pose[6];
    float rotation[3];
    float inv_rotation[3];
    float scale[3];
    float inv_scale[3];
} RtInstanceDesc;

typedef struct RtSceneDesc {
    int32_t num_meshes;     const RtMeshDesc *meshes;
    int32_t num_materials;  const RtMaterialDesc *materials;
    int32_t num_instances;  const RtInstanceDesc *instances;
} RtSceneDesc;

/* The by-value arguments of render<<<>>> (raycast.h:13, Camera.cu:23-36). */
typedef struct RtCameraParams {
    int32_t width, height;
    float K_inv[9];          /* row-major float3x3, invert_intrinsic(K) (utils.hpp:142) */
    float D[4];
    float camera_pose[6];    /* lre */
    float inv_camera_pose[6];/* invert_lre(camera_pose), Camera.cu:21 */
} RtCameraParams;

/* Optional per-pixel parity planes, tight [height][width] int32 DEVICE buffers, any may be NULL.
 * hit_* are -1 on a miss; counts follow raycast.cu:61 (pops), :69-70 (aabb tests), :86
 * (triangle tests), :96 (inside hits). */
typedef struct RtDebugPlanes {
    int32_t *hit_instance, *hit_triangle, *node_pops, *aabb_tests, *tri_tests, *inside_hits;
} RtDebugPlanes;

typedef struct RtScene RtScene;

/* ---- device / memory plumbing (replaces the cudaMallocPitch / cudaMemcpy / cudaFree /
 *      cudaDeviceSynchronize calls of kernel.cu:247-253,279,299) ------------------------------ */
int rt_abi_version(void);
/* "RT_CODE_HASH=<16 hex digits>": the hash of the kernel sources and compiler flags this library was built from, as
 * cuda-raytracing_amd/_build.py kernel_code_hash() computes it ("built-without-it" for a build that did not pass
 * -DRT_CODE_HASH).  A profile or a roofline fraction describes one build of the kernels: bench.py prices its line with the hash
 * of the library that RAN, and the Python loader refuses a library that was not built from the sources next to it. */
const char *rt_build_info(void);
int rt_device_count(int *count);
int rt_set_device(int device);
int rt_malloc(void **dptr, size_t bytes);
int rt_malloc_pitch(void **dptr, size_t *pitch, size_t width_bytes, size_t height);
int rt_free(void *dptr);
int rt_memcpy_d2h(void *dst, const void *src, size_t bytes, void *stream);
int rt_memcpy_h2d(void *dst, const void *src, size_t bytes, void *stream);
int rt_memcpy2d_d2h(void *dst, size_t dpitch, const void *src, size_t spitch, size_t width_bytes, size_t height, void *stream);
int rt_stream_synchronize(void *stream);
int rt_device_synchronize(void);
const char *rt_error_string(int code);

/* ---- scene (replaces Scene::upload_to_device, Scene.cpp:25-65, and everything it calls) ---- */
int rt_scene_upload(const RtSceneDesc *desc, RtScene **out);
/* Scene::update_mesh_instance (Scene.cpp:67-74): re-upload one instance */
int rt_scene_update_instance(RtScene *scene, int32_t index, const RtInstanceDesc *instance);
/* the same update ordered on `stream` instead of synchronising with the device: renders issued on that stream before the
 * call see the old instance, renders issued after it the new one -- an animated instance (the teapot of kernel.cu:272-273)
 * then costs no host wait per frame.  Renders in flight on OTHER streams are not ordered against it. */
int rt_scene_update_instance_async(RtScene *scene, int32_t index, const RtInstanceDesc *instance, void *stream);
/* Refit of a deforming mesh: the same triangles (same count, same order as at upload) at new positions.  vertices
 * [n][3][3] and normals [n][3] are HOST arrays in the RtMeshDesc layout; uvs and the tree's topology stay.  Every node
 * gets the exact bounds of its triangles (what BVHTree::fill's bounds pass, BVHTree.hpp:206-209, would compute for that
 * node), triangle records are recomputed as at upload.  Ordered on `stream`: renders issued on it before the call see the
 * old mesh, renders after it the new one; the host arrays must stay valid until the stream has passed the call
 * (rt_stream_synchronize, or any later synchronising call).  No counterpart in the reference. */
int rt_scene_refit_mesh(RtScene *scene, int32_t mesh_index, const float *vertices, const float *normals,
                        int32_t num_triangles, void *stream);
/* the same with DEVICE arrays (a deformation computed on the GPU: skinning, simulation): no copy, the arrays are read by
 * kernels ordered on `stream` and must stay untouched until the stream has passed them */
int rt_scene_refit_mesh_device(RtScene *scene, int32_t mesh_index, const float *d_vertices, const float *d_normals,
                               int32_t num_triangles, void *stream);
int rt_scene_destroy(RtScene *scene);
/* bytes of device memory the scene holds, and the traversal-stack depth it needs */
int rt_scene_info(const RtScene *scene, size_t *device_bytes, int32_t *max_stack);
/* per-mesh state bits kept on the device (a small synchronous copy).  Bit 0 (1): some interior record of the mesh holds a child box
 * with min > max or a NaN (an empty leaf of a degenerate split, non-finite vertices): the mesh is traversed with the generic slab
 * test instead of the octant-specialised loops (4-7 % slower on c2, same results).  Decided at upload and by every rebuild and
 * refit of the mesh. */
int rt_scene_mesh_flags(RtScene *scene, int32_t mesh_index, int32_t *flags);
/* the largest triangle count rt_scene_rebuild_mesh_device accepts for this mesh (= the count it was uploaded with: its part of
 * the record arrays has room for any tree over that many triangles) */
int rt_scene_mesh_capacity(const RtScene *scene, int32_t mesh_index, int32_t *max_triangles);

/* Device-resident rebuild of one mesh of an uploaded scene: a NEW tree (the reference's, node for node, as rt_bvh_build gives
 * it) over the n triangles in DEVICE arrays d_vertices [n][3][3], d_normals [n][3] and d_uvs [n][3][2] (NULL = all zero), n at
 * most the triangle count the mesh was uploaded with.  The build kernels and an emit pass write straight into the scene's
 * record arrays -- what rt_bvh_build + Scene::upload_to_device + rt_scene_upload produce for the same triangles, bit for bit,
 * without a host copy of vertices, tree or records: for a mesh whose motion has outgrown refitting, or whose triangles change.
 * Ordered on `stream` like rt_scene_refit_mesh_device; the call returns when the new tree is in place (it reads a few words
 * of build state back while it runs).  Replaces MeshPrimitive::build_bvh + Scene::upload_to_device (MeshPrimitive.cpp:38-56,
 * Scene.cpp:25-65) for that mesh.
 * Errors: RT_E_INVALID for a bad index or more triangles than rt_scene_mesh_capacity() -- nothing has been touched then.  A
 * HIP error after the rebuild has started (the mesh's records are cleared first) returns once `stream` has drained, and
 * leaves the mesh WITHOUT a valid tree: do not render it until a later rebuild (or a fresh rt_scene_upload) has succeeded. */
int rt_scene_rebuild_mesh_device(RtScene *scene, int32_t mesh_index, const float *d_vertices, const float *d_normals,
                                 const float *d_uvs, int32_t num_triangles, void *stream);
/* tests: copies one of the scene's device arrays to the host (which: 0 records [float4], 1 tri_uv, 2 tri_id, 3 leaf_count,
 * 4 instances); *bytes receives its size, nothing is copied when capacity is smaller */
int rt_scene_debug_read(RtScene *scene, int32_t which, void *host_dst, size_t capacity, size_t *bytes);

/* ---- GPU build of the reference's BVH (replaces MeshPrimitive::build_bvh -> BVHTree::fill(1, 32), MeshPrimitive.cpp:38-56,
 *      BVHTree.hpp:203-292): identical topology, bounds and pre-order node numbering, built level by level on the device.
 *      vertices: HOST [n][3][3].  Outputs are HOST arrays sized for 2n nodes (n leaf indices) in the RtMeshDesc layout;
 *      *num_nodes receives the node count, *num_levels (optional) the tree depth in levels. ------------------------------ */
int rt_bvh_build(const float *vertices, int32_t n, int32_t max_depth, float *node_bounds, int32_t *node_children,
                 int32_t *node_leaf_first, int32_t *node_leaf_count, int32_t *leaf_indices, int32_t *num_nodes,
                 int32_t *num_levels);

/* ---- render (replaces Camera::render_scene -> render<<<grid, block>>>, Camera.cu:18-41,
 *      raycast.cu:146-297).  d_img is a caller-owned DEVICE buffer of `height` rows of `pitch`
 *      bytes, 3 bytes per pixel in uchar3 .x .y .z order (raycast.cu:292-294).  Asynchronous on
 *      `stream` unless synchronize != 0 (Camera.cu:38-39). ------------------------------------- */
int rt_render(RtScene *scene, const RtCameraParams *cam, uint8_t *d_img, size_t pitch, void *stream, int synchronize);
/* Camera::render_scene(scene, img, pitch) with synchronize = false, as the reference's frame loop calls it: twice, into two
 * images, before one cudaDeviceSynchronize (kernel.cu:277-279).  Ordered like a launch on the DEFAULT stream against
 * everything the caller does on the default stream or device-wide -- copies and memsets of the image, rt_memcpy_*,
 * rt_scene_update_instance[_async] / refit / rebuild with a NULL stream, rt_render* with a NULL stream, rt_device_synchronize:
 * what was issued before is seen by the frame, what is issued after sees the frame -- but two consecutive calls that write
 * DIFFERENT images may overlap: they alternate between two blocking streams the scene owns, so that one frame's costly tiles
 * fill the chip while the previous frame's last workgroups drain (c2: 0.146 -> 0.13 ms per frame in the reference's loop).
 * Calls whose images share memory run in call order (the later frame wins).  Not implied, unlike a real default-stream launch:
 * ordering against work on OTHER blocking streams of the application; a caller with such streams passes its stream to
 * rt_render instead.  RT_RENDER_OVERLAP=0 makes this call rt_render(.., NULL, 0).
 * Round 6: the first of the two streams has the HIGHER stream priority, and the frame issued when it is idle -- the first frame after
 * a synchronise -- goes there: two equal streams share the chip, both frames of a pair run at half speed and end TOGETHER, so neither
 * tail is hidden (245 us for the pair, 125 us for a frame alone); with priorities the first frame takes the chip and the second fills
 * the slots its tail leaves free (the reference's loop on c2: 0.136 -> 0.131 ms per frame).  Side effect, measured on this runtime:
 * once a high-priority stream exists in the process, two NORMAL streams that alternate single frames run 11 % slower (0.112 -> 0.125 ms
 * per frame) -- an application that does both uses RT_OVERLAP_PRIORITY=0 (two equal streams, round 5's behaviour). */
int rt_render_overlapped(RtScene *scene, const RtCameraParams *cam, uint8_t *d_img, size_t pitch);
/* how many frames went through rt_render_overlapped on this scene and how many of them had to wait for the other stream
 * (an image overlapping one written there and one written here); either pointer may be NULL */
int rt_render_overlapped_stats(const RtScene *scene, uint64_t *launches, uint64_t *cross_stream_waits);
/* View records (round 5; no counterpart in the reference, whose kernel subtracts the ray origin from every box at every visit,
 * BVHTree.hpp:40-54 through raycast.cu:69-70): launches of at least four frames whose frames bring enough rays first write, per
 * frame and instance, the interior records with `box - origin` in place of the boxes (the same fp32 subtraction, done once), and
 * the traversal reads those.  Nothing in a frame depends on it; RT_VIEW_RECORDS=0 turns it off.
 * MEMORY AND BLOCKING (round 6).  The views live behind the scene's records in ONE allocation (a lane's fetch stays one 32-bit offset
 * from one base): a pool of one to three slots (launches in flight on different streams), each of `frames` views of
 * 64 B x (interior-record capacity of the largest mesh) x instances -- for the 70k-triangle c2 scene 4.2 MB per frame, 403 MB for three
 * slots of 32 frames beside 8.7 MB of records; for a 260k-triangle mesh 16.7 MB per frame.  The pool never exceeds RT_VIEW_MAX_BYTES
 * (default 1 GiB; slots are dropped, three -> two -> one, before frames are); a launch that finds no room or no free slot renders
 * without views -- same pixels (`fallbacks` below).
 *   - An application that batches says so once: rt_scene_reserve_views(scene, frames_per_launch) right after rt_scene_upload sizes
 *     the pool for launches of up to that many frames (it re-allocates the record array: call it while nothing renders the scene).
 *     Launches never grow a reserved pool: no render call blocks.  frames_per_launch = 0 hands sizing back to the launches.
 *     RT_E_NOMEM: not within the budget; RT_E_INVALID: the scene cannot use views (more than eight instances, RT_VIEW_RECORDS=0).
 *   - Without a reservation the pool grows inside the first qualifying launch that brings more frames than a slot holds, to the
 *     next of 4 / 8 / 16 / 32 frames: THAT CALL BLOCKS until the device is idle (hipDeviceSynchronize: every stream of the process),
 *     allocates, copies the records and frees the block they were in -- at most four times in a scene's life.  Render calls are
 *     otherwise asynchronous.
 * rt_scene_view_stats reports how many launches qualified, how many of those rendered without views after all, how often the pool
 * was (re)sized, and the frames a slot holds now; rt_scene_memory the bytes: the record array proper, the pool behind it, everything
 * the scene holds on the device, and the pool's shape.  Any pointer may be NULL. */
int rt_scene_reserve_views(RtScene *scene, int32_t frames_per_launch);
int rt_scene_memory(RtScene *scene, size_t *records_bytes, size_t *view_pool_bytes, size_t *device_bytes, int32_t *view_slots,
                    int32_t *view_slot_frames);
int rt_scene_view_stats(RtScene *scene, uint64_t *launches, uint64_t *fallbacks, uint64_t *grows, int32_t *slot_frames);
/* Which traversal loop ran (round 6; diagnostics, no counterpart in the reference).  The primary kernel picks, per wave and
 * instance, the hand-written gfx950 loop (any instance without the exact-uv mode, when the wave's rays share a sign octant and
 * the mesh's boxes are ordered), the compiler's octant-specialised loop, or the compiler's generic loop; a tree deeper than the LDS
 * part of the stack is traversed optimistically and the lanes that outgrow it are traced again on the general stack.  Parity tests
 * pass on any of them, so a change that pushes a scene off the fast loop would only show as a slower frame: this call renders the
 * batch exactly as rt_render_batch would (same launch decisions: stack form, view records) through an INSTRUMENTED copy of the
 * kernel -- never the timed one -- waits for it, and fills stats[RT_LOOP_WORDS] (host): counts of waves x instances per loop.
 * The frames are written as by rt_render_batch. */
enum {
    RT_LOOP_WAVES = 0,          /* waves that rendered at least one pixel */
    RT_LOOP_ASM = 1,            /* wave x instance casts on the hand-written loop ... */
    RT_LOOP_ASM_POSED = 2,      /* ... of which: instances that scale or rotate (the candidate block's out-of-line transform) */
    RT_LOOP_CPP_OCTANT = 3,     /* casts on the compiler's octant-specialised loop (exact-uv meshes) */
    RT_LOOP_CPP_GENERIC = 4,    /* casts on the compiler's generic loop (mixed octants in the wave, zero / non-finite direction inverses, unordered boxes) */
    RT_LOOP_DEEP = 5,           /* casts traced again on the general stack (some lane outgrew the LDS part) */
    RT_LOOP_RETRACED_LANES = 6, /* rays (lanes) traced again */
    RT_LOOP_WORDS = 8
};
int rt_scene_loop_stats(RtScene *scene, const RtCameraParams *cams, uint8_t *const *d_imgs, size_t pitch, int32_t count,
                        void *stream, uint64_t *stats);
/* `count` (1..RT_MAX_BATCH) frames of the same size in ONE launch: cams[i] is rendered into d_imgs[i].  A frame
 * stream rendered this way keeps the GPU full while the last long rays of one frame finish (the reference's own
 * loop issues two renders before it synchronises, kernel.cu:277-279).  Asynchronous on `stream`, with one exception: a launch of four
 * or more frames that has to grow the scene's view pool blocks until the device is idle (see rt_scene_reserve_views, which avoids it).
 * Host threads: calls on ONE scene are serialised while they prepare and queue their launch (a few microseconds; the GPU work of
 * different streams still overlaps); different scenes do not meet. */
#define RT_MAX_BATCH 32
int rt_render_batch(RtScene *scene, const RtCameraParams *cams, uint8_t *const *d_imgs, size_t pitch, int32_t count,
                    void *stream, int synchronize);
/* the same production kernel as rt_render, additionally storing the accepted hit of every pixel (raycast.cu:107-127):
 * tight [height][width] int32 DEVICE planes, -1 on a miss, either may be NULL.  For parity tests of the kernel that is
 * actually timed (rt_render_debug runs an instrumented copy). */
int rt_render_ids(RtScene *scene, const RtCameraParams *cam, uint8_t *d_img, size_t pitch, int32_t *d_hit_instance,
                  int32_t *d_hit_triangle, void *stream, int synchronize);
/* same frame plus the parity planes (instrumented copy of the kernel: counts every visit) */
int rt_render_debug(RtScene *scene, const RtCameraParams *cam, uint8_t *d_img, size_t pitch,
                    const RtDebugPlanes *planes, void *stream, int synchronize);

/* ---- extension (no counterpart in the reference snapshot, whose shadow pass is commented out
 *      at raycast.cu:262-287 and which has no spp / bounce loop).  Semantics: DESIGN.md "Extension".  With
 *      spp = 1, bounces = 0, lighting = 0 the frame equals rt_render's bit for bit.  d_total_pops: optional tight
 *      [height][width] int32 device plane receiving the node pops of all rays of each pixel.  The per-sample
 *      scratch lives in the scene handle: extension renders on one scene must not overlap on different streams. -- */
typedef struct RtRenderOptions {
    int32_t spp;        /* >= 1; sample 0 is the reference's un-jittered ray, sample s > 0 is jittered from its own XORWOW
                           stream (seed = the reference's per-pixel seed + s) */
    int32_t bounces;    /* specular bounces weighted by Material::metallic, perturbed by Material::roughness */
    int32_t lighting;   /* 1 = sun + shadow ray of raycast.cu:249-287, 0 = illumination 1.0 (raycast.cu:282) */
} RtRenderOptions;
int rt_render_ex(RtScene *scene, const RtCameraParams *cam, const RtRenderOptions *opts, uint8_t *d_img, size_t pitch,
                 int32_t *d_total_pops, void *stream, int synchronize);

/* this rank's stripes of an extension frame (see rt_render_stripes below for the stripe layout) */
int rt_render_ex_stripes(RtScene *scene, const RtCameraParams *cam, const RtRenderOptions *opts, uint8_t *d_local,
                         size_t local_pitch, int32_t stripe_rows, int32_t rank, int32_t num_ranks, void *stream, int synchronize);

/* ---- frame tiling across GPUs (no counterpart in the reference: it is single-GPU).  The frame
 *      is cut into stripes of `stripe_rows` rows; stripe s belongs to rank s % num_ranks.  A rank
 *      renders its stripes into a tight local buffer (rows packed in stripe order, pitch =
 *      local_pitch); rt_stripe_rows() gives that buffer's row count for any rank. ------------- */
int rt_stripe_rows(int32_t height, int32_t stripe_rows, int32_t rank, int32_t num_ranks, int32_t *rows);
int rt_render_stripes(RtScene *scene, const RtCameraParams *cam, uint8_t *d_local, size_t local_pitch,
                      int32_t stripe_rows, int32_t rank, int32_t num_ranks, void *stream, int synchronize);
int rt_render_stripes_batch(RtScene *scene, const RtCameraParams *cams, uint8_t *const *d_locals, size_t local_pitch,
                            int32_t count, int32_t stripe_rows, int32_t rank, int32_t num_ranks, void *stream, int synchronize);
/* The same with the stripe owner ROTATING over the frames: frame i of the launch renders the stripes of owner
 * (rank + first_frame + i) % num_ranks (first_frame >= 0: the index of cams[0] within its group of frames).  The owners' shares
 * of a frame differ -- 1080 rows are 67.5 stripes of 16, so at 8 ranks three own 144 rows, four 128, and stripes near the
 * object cost more than sky -- and the slowest rank sets the pace of every exchange; with rotation every rank renders every
 * owner's share once per num_ranks frames, so the ranks' shares of a group are equal whatever the frames show (c2 at 8 ranks:
 * max / mean of the render time 1.04 -> 1.00).  Frame i's rows are packed in d_locals[i] as that owner's; the buffer must hold
 * the rows of the largest share (rt_stripe_rows of rank 0).  rt_unstripe_batch_rotating is the matching un-stripe pass. */
int rt_render_stripes_batch_rotating(RtScene *scene, const RtCameraParams *cams, uint8_t *const *d_locals, size_t local_pitch,
                                     int32_t count, int32_t stripe_rows, int32_t rank, int32_t num_ranks, int32_t first_frame,
                                     void *stream, int synchronize);
/* after a gather of every rank's local buffer (rank r's rows start at d_gathered + r * rank_stride bytes, rows
 * local_pitch bytes apart) place the rows back into frame order */
int rt_unstripe(const uint8_t *d_gathered, size_t local_pitch, size_t rank_stride,
                uint8_t *d_img, size_t pitch, int32_t width, int32_t height,
                int32_t stripe_rows, int32_t num_ranks, void *stream);

/* the same for `count` frames at once: frame f reads from d_gathered + f * src_frame_stride and writes to
 * d_imgs + f * dst_frame_stride (the layout of a gathered rt_render_stripes_batch group) */
int rt_unstripe_batch(const uint8_t *d_gathered, size_t local_pitch, size_t rank_stride, size_t src_frame_stride,
                      uint8_t *d_imgs, size_t pitch, size_t dst_frame_stride, int32_t count,
                      int32_t width, int32_t height, int32_t stripe_rows, int32_t num_ranks, void *stream);

/* rt_unstripe_batch for frames rendered by rt_render_stripes_batch_rotating: frame f of the batch has index first_frame + f in its
 * group, so the block of rank r (at d_gathered + r * rank_stride) holds the rows of owner (r + first_frame + f) % num_ranks */
int rt_unstripe_batch_rotating(const uint8_t *d_gathered, size_t local_pitch, size_t rank_stride, size_t src_frame_stride,
                               uint8_t *d_imgs, size_t pitch, size_t dst_frame_stride, int32_t count,
                               int32_t width, int32_t height, int32_t stripe_rows, int32_t num_ranks, int32_t first_frame, void *stream);

/* ---- the exchange step of frame tiling: an RCCL communicator over the GPUs of one node (no counterpart in the
 *      reference; BASELINE.json north_star: "the frame is tiled across the 8 GPUs of one node with a final RCCL gather
 *      over xGMI").  RCCL (librccl.so.1) is loaded on first use; without it these calls return RT_E_COMM.
 *      One process per GPU: rank 0 calls rt_comm_unique_id, the host application passes the 128 bytes to the other
 *      ranks by its own means (MPI, a socket, torch.distributed ...), every rank selects its device (rt_set_device)
 *      and calls rt_comm_init_rank.  One process for all GPUs: rt_comm_init_all fills comms[0..num_devices) (rank i on
 *      devices[i], or on device i when devices is NULL) and the *_all calls drive them.
 *      Collectives are asynchronous on `stream` like every other call here. ------------------------------------- */
typedef struct RtComm RtComm;
#define RT_COMM_ID_BYTES 128
int rt_comm_available(int32_t *rccl_version);            /* RT_OK when RCCL could be loaded */
const char *rt_comm_last_error(void);                    /* text of the calling thread's last RT_E_COMM */
/* text of the most recent RT_E_COMM of ANY thread of the process (for a watchdog thread that reports on a main thread stuck
 * inside a collective); the pointer is the calling thread's own copy, valid until its next call */
const char *rt_comm_last_error_any(void);
int rt_comm_unique_id(uint8_t *id /* [RT_COMM_ID_BYTES] */);
int rt_comm_init_rank(const uint8_t *id, int32_t rank, int32_t num_ranks, RtComm **out);   /* on the current device */
int rt_comm_init_all(const int32_t *devices, int32_t num_devices, RtComm **comms);
/* rank and size as the RCCL communicator itself reports them (ncclCommUserRank / ncclCommCount; the values given at creation
 * if the library lacks the two queries), and the device it lives on */
int rt_comm_info(const RtComm *comm, int32_t *rank, int32_t *num_ranks, int32_t *device);
int rt_comm_destroy(RtComm *comm);
int rt_group_start(void);                                /* ncclGroupStart / ncclGroupEnd for single-process callers */
int rt_group_end(void);
/* every rank's `bytes` bytes at d_send to rank `root`, which receives rank r's block at d_recv + r * bytes
 * (d_recv may be NULL elsewhere): the per-frame gather of SURVEY.md 8(e) */
int rt_gather(RtComm *comm, const void *d_send, size_t bytes, void *d_recv, int32_t root, void *stream);
/* rank p gets send_bytes[p] bytes from d_send + send_offsets[p]; recv_bytes[p] bytes from rank p land at
 * d_recv + recv_offsets[p] (zero-byte pairs are skipped; the counts must agree pairwise).  One fused group of
 * point-to-point transfers: the gathers of a group of frames whose root rotates over the ranks. */
int rt_all_to_all(RtComm *comm, const void *d_send, const size_t *send_bytes, const size_t *send_offsets,
                  void *d_recv, const size_t *recv_bytes, const size_t *recv_offsets, void *stream);
/* One tiled frame, the whole of SURVEY.md 8(e) in one call made by every rank: render this rank's stripes (rt_render_stripes,
 * or rt_render_ex_stripes when opts is non-NULL and not the default 1 / 0 / 0), gather them to `root`, and on the root
 * put the rows back into frame order in d_img (may be NULL on other ranks).  Scratch buffers live in the communicator:
 * consecutive calls on one RtComm may use different streams (frames alternated between two streams) -- each call waits, on its
 * stream, for the previous call's last use of the scratch (an event, no host synchronisation); calls on one RtComm from
 * several host threads at once are not supported.  With one rank this is rt_render / rt_render_ex. */
int rt_render_tiled(RtScene *scene, RtComm *comm, const RtCameraParams *cam, const RtRenderOptions *opts, uint8_t *d_img,
                    size_t pitch, int32_t stripe_rows, int32_t root, void *stream, int synchronize);
/* the same from ONE process that holds a scene replica and a communicator per device (rt_comm_init_all):
 * scenes[r] / comms[r] / streams[r] (streams may be NULL) belong to rank r; d_img is on the root's device */
int rt_render_tiled_all(RtScene *const *scenes, RtComm *const *comms, int32_t num_ranks, const RtCameraParams *cam,
                        const RtRenderOptions *opts, uint8_t *d_img, size_t pitch, int32_t stripe_rows, int32_t root,
                        void *const *streams, int synchronize);

/* ---- timing on the stream the kernels run on (hipEvent) ---------------------------------- */
typedef struct RtTimer RtTimer;
int rt_timer_create(RtTimer **t);
int rt_timer_start(RtTimer *t, void *stream);
int rt_timer_stop(RtTimer *t, void *stream);
int rt_timer_elapsed_ms(RtTimer *t, float *ms);   /* synchronises on the stop event */
int rt_timer_destroy(RtTimer *t);

#ifdef __cplusplus
}
#endif
#endif /* RT_HIP_H */
