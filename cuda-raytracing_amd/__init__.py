"""cuda-raytracing_amd -- MI355X-native raycast hot path behind the reference's host API.

Python here is plumbing only: ctypes bindings of the two in-tree libraries
(``librt_hip.so`` = HIP kernels + C-ABI ``include/rt_hip.h``; ``librt_host.so`` = host C++
API mirror + C facade ``include/rt_host.h``) and small helpers used by ``bench.py`` and the
tests.  There is no CPU fallback: if the libraries are missing or no GPU is present, device
calls fail loudly.

The directory name is not a Python identifier; import it with
``importlib.import_module("cuda-raytracing_amd")``.
"""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
HIP_SO = os.path.join(HERE, "librt_hip.so")
HOST_SO = os.path.join(HERE, "librt_host.so")

_f = C.POINTER(C.c_float)
_i = C.POINTER(C.c_int32)
_vp = C.c_void_p


class RtError(RuntimeError):
    pass


class RtCameraParams(C.Structure):          # include/rt_hip.h
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("K_inv", C.c_float * 9), ("D", C.c_float * 4),
                ("camera_pose", C.c_float * 6), ("inv_camera_pose", C.c_float * 6)]


class RtDebugPlanes(C.Structure):
    _fields_ = [(n, _vp) for n in ("hit_instance", "hit_triangle", "node_pops", "aabb_tests", "tri_tests", "inside_hits")]


class RtMeshDesc(C.Structure):              # include/rt_hip.h
    _fields_ = [("num_triangles", C.c_int32), ("vertices", _f), ("normals", _f), ("uvs", _f), ("num_nodes", C.c_int32),
                ("node_bounds", _f), ("node_children", _i), ("node_leaf_first", _i), ("node_leaf_count", _i),
                ("num_leaf_indices", C.c_int32), ("leaf_indices", _i)]


class RtMaterialDesc(C.Structure):
    _fields_ = [("roughness", C.c_float), ("albedo", C.c_float * 3), ("metallic", C.c_float), ("illumination", C.c_float),
                ("texture", _vp), ("texture_width", C.c_int32), ("texture_height", C.c_int32), ("texture_pitch", C.c_size_t)]


class RtInstanceDesc(C.Structure):
    _fields_ = [("mesh_index", C.c_int32), ("material_index", C.c_int32), ("pose", C.c_float * 6), ("inv_pose", C.c_float * 6),
                ("rotation", C.c_float * 3), ("inv_rotation", C.c_float * 3), ("scale", C.c_float * 3), ("inv_scale", C.c_float * 3)]


class RtSceneDesc(C.Structure):
    _fields_ = [("num_meshes", C.c_int32), ("meshes", C.POINTER(RtMeshDesc)), ("num_materials", C.c_int32),
                ("materials", C.POINTER(RtMaterialDesc)), ("num_instances", C.c_int32), ("instances", C.POINTER(RtInstanceDesc))]


# every exported symbol of include/rt_hip.h and include/rt_host.h (tests check the libraries export them all)
RT_HIP_SYMBOLS = [
    "rt_abi_version", "rt_build_info", "rt_device_count", "rt_set_device", "rt_malloc", "rt_malloc_pitch", "rt_free", "rt_memcpy_d2h",
    "rt_memcpy_h2d", "rt_memcpy2d_d2h", "rt_stream_synchronize", "rt_device_synchronize", "rt_error_string",
    "rt_bvh_build", "rt_scene_upload", "rt_scene_update_instance", "rt_scene_update_instance_async", "rt_scene_refit_mesh", "rt_scene_refit_mesh_device", "rt_scene_rebuild_mesh_device", "rt_scene_debug_read", "rt_scene_destroy", "rt_scene_info", "rt_scene_mesh_capacity", "rt_scene_mesh_flags", "rt_render", "rt_render_overlapped", "rt_render_overlapped_stats", "rt_scene_view_stats", "rt_scene_reserve_views", "rt_scene_memory", "rt_scene_loop_stats", "rt_render_batch",
    "rt_render_debug", "rt_render_ids", "rt_render_ex", "rt_render_ex_stripes", "rt_stripe_rows", "rt_render_stripes", "rt_render_stripes_batch", "rt_render_stripes_batch_rotating", "rt_unstripe", "rt_unstripe_batch", "rt_unstripe_batch_rotating",
    "rt_comm_available", "rt_comm_last_error", "rt_comm_last_error_any", "rt_comm_unique_id", "rt_comm_init_rank", "rt_comm_init_all", "rt_comm_info", "rt_comm_destroy",
    "rt_group_start", "rt_group_end", "rt_gather", "rt_all_to_all", "rt_render_tiled", "rt_render_tiled_all", "rt_timer_create", "rt_timer_start", "rt_timer_stop",
    "rt_timer_elapsed_ms", "rt_timer_destroy"]
RT_HOST_SYMBOLS = [
    "rth_obj_load", "rth_obj_parse", "rth_scan_float", "rth_obj_load_for_device", "rth_mesh_from_triangles_for_device", "rth_obj_load_lenient", "rth_obj_load_gpu", "rth_mesh_from_triangles", "rth_mesh_from_triangles_gpu", "rth_mesh_single_triangle", "rth_mesh_free", "rth_mesh_num_triangles",
    "rth_mesh_num_nodes", "rth_mesh_max_level", "rth_mesh_get_triangles", "rth_mesh_get_nodes", "rth_mesh_get_leaf_indices",
    "rth_mesh_print_stats", "rth_scene_create", "rth_scene_free", "rth_scene_add_material", "rth_scene_add_material_ppm",
    "rth_scene_set_material_params", "rth_scene_add_mesh", "rth_scene_add_mesh_instance", "rth_scene_upload_to_device", "rth_scene_update_mesh_instance", "rth_scene_update_mesh_instance_async", "rth_scene_refit_mesh", "rth_scene_rebuild_mesh",
    "rth_scene_num_mesh_instances", "rth_scene_device_handle", "rth_instance_build", "rth_camera_create", "rth_camera_free",
    "rth_camera_set_pose", "rth_camera_set_stream", "rth_camera_render_scene", "rth_camera_render_scene_stripes",
    "rth_camera_render_scene_tiled", "rth_camera_render_scene_batch", "rth_camera_render_scene_stripes_batch", "rth_camera_render_scene_stripes_batch_rotating", "rth_camera_set_options",
    "rth_camera_render_scene_ex", "rth_xorwow", "rth_save_png", "rth_write_png_bgr",
    "rth_read_image_bgr", "rth_zlib_inflate", "rth_overlay_text_bgr", "rth_display_image", "rth_on_mouse", "rth_on_key",
    "rth_camera_params", "rth_q_rsqrt", "rth_atanf", "rth_normalize", "rth_invert_lre", "rth_apply_lre", "rth_euler2quat",
    "rth_apply_quat", "rth_invert_intrinsic", "rth_last_error"]

_hip = None
_host = None


def build(force=False, verbose=False):
    from . import _build as _b
    return _b.build(force=force, verbose=verbose)


def libs():
    """(librt_hip, librt_host) as ctypes CDLLs; raises RtError if they have not been built."""
    global _hip, _host
    if _hip is None:
        for p in (HIP_SO, HOST_SO):
            if not os.path.exists(p):
                raise RtError("%s is missing: run __graft_entry__.build() (no CPU fallback exists)" % p)
        # When PyTorch is installed it must be loaded FIRST: it ships its own libamdhip64 / librccl under the same sonames,
        # and a process must not end up with two HIP runtimes or two RCCLs (librt_hip.so dlopens "librccl.so.1" on first use
        # of rt_comm_*: after torch that resolves to torch's copy, before it to ROCm's -- and torch would then be handed
        # ROCm's copy in place of the one it was built against; seen as a double free at process exit).
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        # The library must be the build of the sources next to it: profiles, roofline fractions and parity claims are about
        # one build of the kernels, and file times prove nothing (a variant copied over the shipped library is newer than
        # every source).  A mismatch is rebuilt when the compiler is here, else refused; RT_ALLOW_VARIANT_LIB=1
        # (tools/ab_variants.sh) loads the variant as it is -- bench.py's line then carries the variant's own hash.
        from . import _build as _b
        try:
            why = _b.library_mismatch(HIP_SO)
        except _b.BuildError as e:                               # (the sources next to the library cannot be read: nothing to check it against)
            raise RtError(str(e))
        if why is not None and not _b.variant_allowed():
            try:
                _b.build()
            except Exception as e:
                raise RtError("%s; rebuilding failed: %s" % (why, e))
            why = _b.library_mismatch(HIP_SO)
            if why is not None:
                raise RtError(why + " (after a rebuild)")
        hip = C.CDLL(HIP_SO, mode=C.RTLD_GLOBAL)
        host = C.CDLL(HOST_SO)
        _declare(hip, host)
        loaded = library_hash(hip)
        if loaded != _b.library_code_hash(HIP_SO) or (loaded != _b.kernel_code_hash() and not _b.variant_allowed()):
            raise RtError("the loaded librt_hip.so reports kernel code hash %s; the file holds %s and the sources hash to %s"
                          % (loaded, _b.library_code_hash(HIP_SO), _b.kernel_code_hash()))
        _hip, _host = hip, host
    return _hip, _host


def library_hash(hip=None):
    """The kernel code hash the LOADED librt_hip.so was compiled with (rt_build_info)."""
    h = hip if hip is not None else libs()[0]
    text = h.rt_build_info().decode()
    return text.split("=", 1)[1] if "=" in text else text


def _declare(h, s):
    h.rt_error_string.restype = C.c_char_p
    h.rt_build_info.restype = C.c_char_p
    h.rt_build_info.argtypes = []
    h.rt_error_string.argtypes = [C.c_int]
    h.rt_device_count.argtypes = [_i]
    h.rt_malloc.argtypes = [C.POINTER(_vp), C.c_size_t]
    h.rt_malloc_pitch.argtypes = [C.POINTER(_vp), C.POINTER(C.c_size_t), C.c_size_t, C.c_size_t]
    h.rt_free.argtypes = [_vp]
    h.rt_memcpy_d2h.argtypes = [_vp, _vp, C.c_size_t, _vp]
    h.rt_memcpy_h2d.argtypes = [_vp, _vp, C.c_size_t, _vp]
    h.rt_memcpy2d_d2h.argtypes = [_vp, C.c_size_t, _vp, C.c_size_t, C.c_size_t, C.c_size_t, _vp]
    h.rt_stream_synchronize.argtypes = [_vp]
    h.rt_scene_upload.argtypes = [C.POINTER(RtSceneDesc), C.POINTER(_vp)]
    h.rt_scene_info.argtypes = [_vp, C.POINTER(C.c_size_t), _i]
    h.rt_scene_mesh_capacity.argtypes = [_vp, C.c_int32, _i]
    h.rt_scene_mesh_flags.argtypes = [_vp, C.c_int32, _i]
    h.rt_scene_update_instance.argtypes = [_vp, C.c_int32, _vp]
    h.rt_scene_update_instance_async.argtypes = [_vp, C.c_int32, _vp, _vp]
    h.rt_scene_refit_mesh.argtypes = [_vp, C.c_int32, _f, _f, C.c_int32, _vp]
    h.rt_scene_refit_mesh_device.argtypes = [_vp, C.c_int32, _vp, _vp, C.c_int32, _vp]
    h.rt_scene_rebuild_mesh_device.argtypes = [_vp, C.c_int32, _vp, _vp, _vp, C.c_int32, _vp]
    h.rt_scene_debug_read.argtypes = [_vp, C.c_int32, _vp, C.c_size_t, C.POINTER(C.c_size_t)]
    h.rt_scene_destroy.argtypes = [_vp]
    h.rt_render.argtypes = [_vp, C.POINTER(RtCameraParams), _vp, C.c_size_t, _vp, C.c_int]
    h.rt_render_overlapped.argtypes = [_vp, C.POINTER(RtCameraParams), _vp, C.c_size_t]
    h.rt_render_overlapped_stats.argtypes = [_vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    h.rt_scene_view_stats.argtypes = [_vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_int32)]
    h.rt_scene_reserve_views.argtypes = [_vp, C.c_int32]
    h.rt_scene_memory.argtypes = [_vp, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t), C.POINTER(C.c_size_t), _i, _i]
    h.rt_scene_loop_stats.argtypes = [_vp, C.POINTER(RtCameraParams), C.POINTER(_vp), C.c_size_t, C.c_int32, _vp, C.POINTER(C.c_uint64)]
    h.rt_render_debug.argtypes = [_vp, C.POINTER(RtCameraParams), _vp, C.c_size_t, C.POINTER(RtDebugPlanes), _vp, C.c_int]
    h.rt_render_ids.argtypes = [_vp, C.POINTER(RtCameraParams), _vp, C.c_size_t, _vp, _vp, _vp, C.c_int]
    h.rt_stripe_rows.argtypes = [C.c_int32, C.c_int32, C.c_int32, C.c_int32, _i]
    h.rt_render_stripes.argtypes = [_vp, C.POINTER(RtCameraParams), _vp, C.c_size_t, C.c_int32, C.c_int32, C.c_int32, _vp, C.c_int]
    h.rt_unstripe.argtypes = [_vp, C.c_size_t, C.c_size_t, _vp, C.c_size_t, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _vp]
    h.rt_unstripe_batch.argtypes = [_vp, C.c_size_t, C.c_size_t, C.c_size_t, _vp, C.c_size_t, C.c_size_t, C.c_int32, C.c_int32, C.c_int32,
                                    C.c_int32, C.c_int32, _vp]
    h.rt_unstripe_batch_rotating.argtypes = [_vp, C.c_size_t, C.c_size_t, C.c_size_t, _vp, C.c_size_t, C.c_size_t, C.c_int32, C.c_int32, C.c_int32,
                                             C.c_int32, C.c_int32, C.c_int32, _vp]
    _sz = C.POINTER(C.c_size_t)
    h.rt_comm_last_error.restype = C.c_char_p
    h.rt_comm_available.argtypes = [_i]
    h.rt_comm_unique_id.argtypes = [_vp]
    h.rt_comm_init_rank.argtypes = [_vp, C.c_int32, C.c_int32, C.POINTER(_vp)]
    h.rt_comm_init_all.argtypes = [_i, C.c_int32, C.POINTER(_vp)]
    h.rt_comm_info.argtypes = [_vp, _i, _i, _i]
    h.rt_comm_destroy.argtypes = [_vp]
    h.rt_gather.argtypes = [_vp, _vp, C.c_size_t, _vp, C.c_int32, _vp]
    h.rt_all_to_all.argtypes = [_vp, _vp, _sz, _sz, _vp, _sz, _sz, _vp]
    h.rt_render_tiled.argtypes = [_vp, _vp, C.POINTER(RtCameraParams), _vp, _vp, C.c_size_t, C.c_int32, C.c_int32, _vp, C.c_int]
    h.rt_render_tiled_all.argtypes = [C.POINTER(_vp), C.POINTER(_vp), C.c_int32, C.POINTER(RtCameraParams), _vp, _vp, C.c_size_t,
                                      C.c_int32, C.c_int32, C.POINTER(_vp), C.c_int]
    h.rt_timer_create.argtypes = [C.POINTER(_vp)]
    h.rt_timer_start.argtypes = [_vp, _vp]
    h.rt_timer_stop.argtypes = [_vp, _vp]
    h.rt_timer_elapsed_ms.argtypes = [_vp, _f]
    h.rt_timer_destroy.argtypes = [_vp]

    s.rth_last_error.restype = C.c_char_p
    for n in ("rth_obj_load", "rth_obj_load_for_device", "rth_mesh_from_triangles_for_device", "rth_obj_load_lenient", "rth_obj_load_gpu", "rth_mesh_from_triangles", "rth_mesh_from_triangles_gpu", "rth_mesh_single_triangle", "rth_scene_create", "rth_camera_create",
              "rth_scene_device_handle"):
        getattr(s, n).restype = _vp
    s.rth_obj_load.argtypes = [C.c_char_p]
    s.rth_obj_load_lenient.argtypes = [C.c_char_p]
    s.rth_obj_load_gpu.argtypes = [C.c_char_p]
    s.rth_obj_load_for_device.argtypes = [C.c_char_p]
    s.rth_mesh_from_triangles_for_device.argtypes = [_f, C.c_int32]
    s.rth_obj_parse.restype = C.c_int32
    s.rth_obj_parse.argtypes = [C.c_char_p, C.c_int32, _vp, C.c_int32]
    s.rth_scan_float.restype = C.c_int
    s.rth_scan_float.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(C.c_float)]
    s.rth_mesh_from_triangles_gpu.argtypes = [_f, C.c_int32]
    s.rth_mesh_from_triangles.argtypes = [_f, C.c_int32]
    s.rth_mesh_single_triangle.argtypes = [_f]
    for n in ("rth_mesh_free", "rth_mesh_num_triangles", "rth_mesh_num_nodes", "rth_mesh_max_level", "rth_mesh_print_stats",
              "rth_scene_free", "rth_scene_upload_to_device", "rth_scene_num_mesh_instances", "rth_scene_device_handle",
              "rth_camera_free"):
        getattr(s, n).argtypes = [_vp]
    s.rth_mesh_free.restype = None
    s.rth_scene_free.restype = None
    s.rth_camera_free.restype = None
    s.rth_mesh_get_triangles.argtypes = [_vp, _f]
    s.rth_mesh_get_nodes.argtypes = [_vp, _f, _i, _i]
    s.rth_mesh_get_leaf_indices.argtypes = [_vp, _i]
    s.rth_scene_add_material.argtypes = [_vp, _f, _vp, C.c_int32, C.c_int32, C.c_size_t]
    s.rth_scene_add_material_ppm.argtypes = [_vp, _f, C.c_char_p]
    s.rth_scene_add_mesh.argtypes = [_vp, _vp]
    s.rth_scene_add_mesh_instance.argtypes = [_vp, C.c_int32, C.c_int32, _f, _f]
    s.rth_scene_update_mesh_instance.argtypes = [_vp, C.c_int32, C.c_int32, C.c_int32, _f, _f]
    s.rth_scene_update_mesh_instance_async.argtypes = [_vp, C.c_int32, C.c_int32, C.c_int32, _f, _f, _vp]
    s.rth_scene_refit_mesh.argtypes = [_vp, C.c_int32, _f, C.c_int32, _vp]
    s.rth_scene_rebuild_mesh.argtypes = [_vp, C.c_int32, _f, C.c_int32, _vp]
    s.rth_instance_build.argtypes = [_f, _f, _f]
    s.rth_camera_create.argtypes = [C.c_int32, C.c_int32, _f, _f]
    s.rth_camera_set_pose.argtypes = [_vp, _f]
    s.rth_camera_set_stream.argtypes = [_vp, _vp]
    s.rth_camera_render_scene.argtypes = [_vp, _vp, _vp, C.c_size_t, C.c_int]
    s.rth_camera_render_scene_stripes.argtypes = [_vp, _vp, _vp, C.c_size_t, C.c_int32, C.c_int32, C.c_int32, C.c_int]
    s.rth_camera_render_scene_tiled.argtypes = [_vp, _vp, _vp, _vp, C.c_size_t, C.c_int32, C.c_int32, C.c_int]
    s.rth_camera_render_scene_batch.argtypes = [_vp, _vp, _f, C.POINTER(_vp), C.c_size_t, C.c_int32, C.c_int]
    s.rth_camera_render_scene_stripes_batch.argtypes = [_vp, _vp, _f, C.POINTER(_vp), C.c_size_t, C.c_int32, C.c_int32, C.c_int32,
                                                        C.c_int32, C.c_int]
    s.rth_camera_render_scene_stripes_batch_rotating.argtypes = [_vp, _vp, _f, C.POINTER(_vp), C.c_size_t, C.c_int32, C.c_int32, C.c_int32,
                                                                 C.c_int32, C.c_int32, C.c_int]
    s.rth_camera_params.argtypes = [_vp, _vp]
    s.rth_scene_set_material_params.argtypes = [_vp, C.c_int32, C.c_float, C.c_float, C.c_float]
    s.rth_camera_set_options.argtypes = [_vp, C.c_int32, C.c_int32, C.c_int32]
    s.rth_camera_render_scene_ex.argtypes = [_vp, _vp, _vp, C.c_size_t, _vp, C.c_int]
    s.rth_save_png.argtypes = [C.c_char_p, _vp, C.c_int32, C.c_int32, C.c_size_t]
    s.rth_write_png_bgr.argtypes = [C.c_char_p, _vp, C.c_int32, C.c_int32, C.c_size_t]
    s.rth_read_image_bgr.argtypes = [C.c_char_p, _vp, C.c_size_t, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    s.rth_zlib_inflate.argtypes = [_vp, C.c_size_t, _vp, C.c_size_t, C.POINTER(C.c_size_t)]
    s.rth_overlay_text_bgr.argtypes = [_vp, C.c_int32, C.c_int32, C.c_size_t, C.c_char_p, C.c_int32, C.c_int32, C.c_int32,
                                       C.c_uint8, C.c_uint8, C.c_uint8]
    s.rth_overlay_text_bgr.restype = None
    s.rth_display_image.argtypes = [_vp, C.c_int32, C.c_int32, C.c_size_t, C.c_double, C.c_char_p]
    s.rth_on_mouse.argtypes = [C.POINTER(C.c_float), C.POINTER(C.c_int32), C.c_int32, C.c_int32, C.c_int32]
    s.rth_on_mouse.restype = None
    s.rth_on_key.argtypes = [C.POINTER(C.c_float), C.c_int32]
    s.rth_xorwow.restype = C.c_uint32
    s.rth_xorwow.argtypes = [C.c_uint64, C.c_int32, _vp, _vp]
    s.rth_q_rsqrt.restype = C.c_float
    s.rth_q_rsqrt.argtypes = [C.c_float]
    s.rth_atanf.restype = C.c_float
    s.rth_atanf.argtypes = [C.c_float]


def _fa(a):
    return np.ascontiguousarray(a, np.float32)


def _fp(a):
    return a.ctypes.data_as(_f)


def check(rc, what="rt call"):
    if rc != 0:
        h, s = libs()
        msg = h.rt_error_string(rc).decode() if rc > 0 or rc >= -5 else "?"
        extra = s.rth_last_error().decode()
        if rc == -5:
            extra = h.rt_comm_last_error().decode()
        raise RtError("%s failed: %d (%s) %s" % (what, rc, msg, extra))


def device_count():
    h, _ = libs()
    n = C.c_int32(0)
    rc = h.rt_device_count(C.byref(n))
    return n.value if rc == 0 else 0


# ------------------------------------------------------------------------------------------------
# Thin object wrappers over the C facade.  Names and call order follow the reference's kernel.cu
# main(): load meshes, add materials / meshes / instances, upload_to_device, camera.render_scene.
# ------------------------------------------------------------------------------------------------

class Mesh:
    """MeshPrimitive (host triangles + BVH)."""

    def __init__(self, handle):
        if not handle:
            raise RtError("mesh creation failed: " + libs()[1].rth_last_error().decode())
        self.h = handle

    @classmethod
    def load_obj(cls, path, lenient=False, gpu_build=False, for_device=False):    # OBJLoader::load / load_lenient; BVH on host or GPU
        """for_device: OBJLoader::load_for_device -- no host tree, the GPU builds it inside the scene at Scene.upload_to_device."""
        if for_device:
            return cls(libs()[1].rth_obj_load_for_device(os.fsencode(path)))
        fn = libs()[1].rth_obj_load_gpu if gpu_build else (libs()[1].rth_obj_load_lenient if lenient else libs()[1].rth_obj_load)
        return cls(fn(os.fsencode(path)))

    @classmethod
    def from_triangles(cls, tris18, gpu_build=False, for_device=False):  # MeshPrimitive(std::vector<TrianglePrimitive>[, build_on_device])
        t = _fa(tris18).reshape(-1, 18)
        fn = libs()[1].rth_mesh_from_triangles_for_device if for_device else (libs()[1].rth_mesh_from_triangles_gpu if gpu_build else libs()[1].rth_mesh_from_triangles)
        return cls(fn(_fp(t), t.shape[0]))

    @classmethod
    def single_triangle(cls, abc9):                   # TrianglePrimitive(a, b, c)
        return cls(libs()[1].rth_mesh_single_triangle(_fp(_fa(abc9))))

    @property
    def num_triangles(self):
        return libs()[1].rth_mesh_num_triangles(self.h)

    @property
    def num_nodes(self):
        return libs()[1].rth_mesh_num_nodes(self.h)

    @property
    def max_level(self):
        return libs()[1].rth_mesh_max_level(self.h)

    def dump(self):
        s = libs()[1]
        nt, nn = self.num_triangles, self.num_nodes
        tris = np.zeros((nt, 18), np.float32)
        s.rth_mesh_get_triangles(self.h, _fp(tris))
        boxes = np.zeros((nn, 6), np.float32)
        child = np.zeros((nn, 2), np.int32)
        lc = np.zeros(nn, np.int32)
        total = s.rth_mesh_get_nodes(self.h, _fp(boxes), child.ctypes.data_as(_i), lc.ctypes.data_as(_i))
        li = np.zeros(max(total, 1), np.int32)
        s.rth_mesh_get_leaf_indices(self.h, li.ctypes.data_as(_i))
        return dict(tris=tris, boxes=boxes, child=child, leaf_count=lc, leaf_idx=li[:total])

    def close(self):
        if self.h:
            libs()[1].rth_mesh_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Scene:
    def __init__(self):
        self.h = libs()[1].rth_scene_create()
        if not self.h:
            raise RtError("scene creation failed")

    def add_material(self, albedo, texture_bgr=None, ppm=None, roughness=0.0, metallic=0.0, illumination=0.0, texture_path=None):
        """texture_path (or its older name ppm): a PNG, baseline-JPEG or binary-PPM file for Material::upload_texture;
        texture_bgr: [h, w, 3] uint8 B,G,R pixels."""
        s = libs()[1]
        a = _fa(albedo)
        path = texture_path if texture_path is not None else ppm
        if path is not None:
            check(s.rth_scene_add_material_ppm(self.h, _fp(a), os.fsencode(path)), "add_material(texture file)")
        elif texture_bgr is not None:
            t = np.ascontiguousarray(texture_bgr, np.uint8)
            check(s.rth_scene_add_material(self.h, _fp(a), t.ctypes.data, t.shape[1], t.shape[0], t.strides[0]), "add_material")
        else:
            check(s.rth_scene_add_material(self.h, _fp(a), None, 0, 0, 0), "add_material")
        self._nmat = getattr(self, "_nmat", 0) + 1
        check(s.rth_scene_set_material_params(self.h, self._nmat - 1, roughness, metallic, illumination), "set_material_params")

    def add_mesh(self, mesh):
        check(libs()[1].rth_scene_add_mesh(self.h, mesh.h), "add_mesh")

    def add_mesh_instance(self, mesh, material, pose=(0, 0, 0, 0, 0, 0), scale=(1, 1, 1)):
        check(libs()[1].rth_scene_add_mesh_instance(self.h, mesh, material, _fp(_fa(pose)), _fp(_fa(scale))), "add_mesh_instance")

    def upload_to_device(self):
        check(libs()[1].rth_scene_upload_to_device(self.h), "Scene::upload_to_device")

    def update_mesh_instance(self, index, mesh, material, pose, scale=(1, 1, 1), stream=False):
        """stream=False: synchronising update (the reference's cudaMemcpy, Scene.cpp:67-74); a stream handle (or None for
        the default stream): ordered on that stream, no host wait."""
        if stream is False:
            check(libs()[1].rth_scene_update_mesh_instance(self.h, index, mesh, material, _fp(_fa(pose)), _fp(_fa(scale))),
                  "Scene::update_mesh_instance")
        else:
            check(libs()[1].rth_scene_update_mesh_instance_async(self.h, index, mesh, material, _fp(_fa(pose)), _fp(_fa(scale)), stream),
                  "Scene::update_mesh_instance(stream)")

    def refit_mesh(self, mesh_index, tris18, stream=None):
        """Scene::refit_mesh: the mesh deformed (same triangle count and order): new records, refitted bounds, no rebuild."""
        t = _fa(tris18).reshape(-1, 18)
        check(libs()[1].rth_scene_refit_mesh(self.h, mesh_index, _fp(t), t.shape[0], stream), "Scene::refit_mesh")

    def rebuild_mesh(self, mesh_index, tris18, stream=None):
        """Scene::rebuild_mesh: new triangles (at most as many as at upload): a new tree, built on the GPU in place."""
        t = _fa(tris18).reshape(-1, 18)
        check(libs()[1].rth_scene_rebuild_mesh(self.h, mesh_index, _fp(t), t.shape[0], stream), "Scene::rebuild_mesh")

    def debug_read(self, which, dtype):
        """tests: one of the device arrays of the uploaded scene (rt_scene_debug_read) as a numpy array of `dtype`."""
        n = C.c_size_t(0)
        check(libs()[0].rt_scene_debug_read(self.device_handle, which, None, 0, C.byref(n)), "rt_scene_debug_read")
        out = np.zeros(n.value, np.uint8)
        check(libs()[0].rt_scene_debug_read(self.device_handle, which, out.ctypes.data, n.value, C.byref(n)), "rt_scene_debug_read")
        return out.view(dtype)

    @property
    def device_handle(self):
        return libs()[1].rth_scene_device_handle(self.h)

    def mesh_flags(self, mesh_index):
        """rt_scene_mesh_flags: bit 0 = the mesh is traversed with the generic slab loop (an unordered or NaN child box)"""
        v = C.c_int32(0)
        check(libs()[0].rt_scene_mesh_flags(self.device_handle, mesh_index, C.byref(v)), "rt_scene_mesh_flags")
        return v.value

    def overlap_stats(self):
        """(frames that went through rt_render_overlapped, those that had to wait for the other stream)"""
        a, b = C.c_uint64(0), C.c_uint64(0)
        check(libs()[0].rt_render_overlapped_stats(self.device_handle, C.byref(a), C.byref(b)), "rt_render_overlapped_stats")
        return a.value, b.value

    def view_stats(self):
        """rt_scene_view_stats: dict(launches=, fallbacks=, grows=, slot_frames=) of the scene's view records"""
        a, b, c, d = C.c_uint64(0), C.c_uint64(0), C.c_uint64(0), C.c_int32(0)
        check(libs()[0].rt_scene_view_stats(self.device_handle, C.byref(a), C.byref(b), C.byref(c), C.byref(d)), "rt_scene_view_stats")
        return dict(launches=a.value, fallbacks=b.value, grows=c.value, slot_frames=d.value)

    def reserve_views(self, frames_per_launch):
        """rt_scene_reserve_views: size the view pool for launches of up to that many frames (0: back to growing on demand)"""
        check(libs()[0].rt_scene_reserve_views(self.device_handle, frames_per_launch), "rt_scene_reserve_views")

    def memory(self):
        """rt_scene_memory: dict(records_bytes=, view_pool_bytes=, device_bytes=, view_slots=, view_slot_frames=)"""
        a, b, c, d, e = C.c_size_t(0), C.c_size_t(0), C.c_size_t(0), C.c_int32(0), C.c_int32(0)
        check(libs()[0].rt_scene_memory(self.device_handle, C.byref(a), C.byref(b), C.byref(c), C.byref(d), C.byref(e)), "rt_scene_memory")
        return dict(records_bytes=a.value, view_pool_bytes=b.value, device_bytes=c.value, view_slots=d.value, view_slot_frames=e.value)

    LOOP_STATS = ("waves", "asm", "asm_posed", "cpp_octant", "cpp_generic", "deep", "retraced_lanes")     # RT_LOOP_* of include/rt_hip.h

    def loop_stats(self, camera, poses, d_imgs, pitch, stream=None):
        """rt_scene_loop_stats: renders the batch (poses[i] -> d_imgs[i]) through the instrumented copy of the production kernel and says
        which traversal loop the waves ran: dict of the RT_LOOP_* counts plus asm_loop_frac = casts on the hand-written loop / all casts."""
        n = len(poses)
        cams = (RtCameraParams * n)()
        for i, pose in enumerate(poses):
            camera.set_pose(pose)
            libs()[1].rth_camera_params(camera.h, C.addressof(cams[i]))
        ptrs = (_vp * n)(*[int(x) if not isinstance(x, _vp) else x.value for x in d_imgs])
        out = (C.c_uint64 * 8)()
        check(libs()[0].rt_scene_loop_stats(self.device_handle, cams, ptrs, pitch, n, stream, out), "rt_scene_loop_stats")
        d = {k: int(out[i]) for i, k in enumerate(self.LOOP_STATS)}
        casts = d["asm"] + d["cpp_octant"] + d["cpp_generic"] + d["deep"]
        d["asm_loop_frac"] = round(d["asm"] / casts, 5) if casts else None
        return d

    def info(self):
        b = C.c_size_t(0)
        d = C.c_int32(0)
        check(libs()[0].rt_scene_info(self.device_handle, C.byref(b), C.byref(d)), "rt_scene_info")
        return dict(device_bytes=b.value, max_stack=d.value)

    def close(self):
        if self.h:
            libs()[1].rth_scene_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Camera:
    def __init__(self, width, height, K, D):
        self.width, self.height = int(width), int(height)
        self.h = libs()[1].rth_camera_create(self.width, self.height, _fp(_fa(K)), _fp(_fa(D)))
        if not self.h:
            raise RtError("camera creation failed")

    def set_pose(self, pose):
        libs()[1].rth_camera_set_pose(self.h, _fp(_fa(pose)))

    def set_stream(self, stream):
        libs()[1].rth_camera_set_stream(self.h, stream)

    def params(self):
        p = RtCameraParams()
        libs()[1].rth_camera_params(self.h, C.addressof(p))
        return p

    def render_scene(self, scene, d_img, pitch, synchronize=False):
        check(libs()[1].rth_camera_render_scene(self.h, scene.h, d_img, pitch, 1 if synchronize else 0), "Camera::render_scene")

    def prepared_render(self, scene, pose, d_img, pitch):
        """A zero-argument callable for `camera.pose = pose; camera.render_scene(scene, d_img, pitch)` -- the reference's own
        per-frame calls (kernel.cu:275-278) -- with the ctypes arguments built once."""
        host, cam_h, scene_h = libs()[1], self.h, scene.h
        P = _fa(pose)
        Pp, ptr = _fp(P), _vp(int(d_img))
        set_pose, render = host.rth_camera_set_pose, host.rth_camera_render_scene

        def call():
            set_pose(cam_h, Pp)
            rc = render(cam_h, scene_h, ptr, pitch, 0)
            if rc:
                check(rc, "Camera::render_scene")
        call._keep = (P,)
        return call

    def render_scene_stripes(self, scene, d_local, local_pitch, stripe_rows, rank, num_ranks, synchronize=False):
        check(libs()[1].rth_camera_render_scene_stripes(self.h, scene.h, d_local, local_pitch, stripe_rows, rank, num_ranks,
                                                        1 if synchronize else 0), "Camera::render_scene_stripes")

    def render_scene_tiled(self, scene, comm, d_img, pitch, synchronize=False, stripe_rows=16, root=0):
        """Camera::render_scene_tiled: every rank of `comm` calls this; the frame arrives in d_img on `root`."""
        check(libs()[1].rth_camera_render_scene_tiled(self.h, scene.h, comm.h, d_img, pitch, stripe_rows, root, 1 if synchronize else 0),
              "Camera::render_scene_tiled")

    def set_options(self, spp=1, bounces=0, lighting=0):
        libs()[1].rth_camera_set_options(self.h, spp, bounces, 1 if lighting else 0)

    def render_scene_ex(self, scene, d_img, pitch, d_total_pops=None, synchronize=False):
        check(libs()[1].rth_camera_render_scene_ex(self.h, scene.h, d_img, pitch, d_total_pops, 1 if synchronize else 0),
              "Camera::render_scene_ex")

    def render_scene_batch(self, scene, poses, d_imgs, pitch, synchronize=False):
        """frames along a camera path in one launch: poses[i] -> d_imgs[i] (device pointers)"""
        n = len(poses)
        P = _fa(np.asarray(poses, np.float32).reshape(n, 6))
        ptrs = (_vp * n)(*[int(x) if not isinstance(x, _vp) else x.value for x in d_imgs])
        check(libs()[1].rth_camera_render_scene_batch(self.h, scene.h, _fp(P), ptrs, pitch, n, 1 if synchronize else 0),
              "Camera::render_scene_batch")

    def prepared_batch(self, scene, poses, d_ptrs, pitch, stripes=None):
        """A zero-argument callable that issues one batched launch with pre-built ctypes arguments (the per-call Python
        overhead matters when a rank's share of a frame takes tens of microseconds).  stripes = (stripe_rows, rank,
        num_ranks) renders this rank's stripes, (stripe_rows, rank, num_ranks, first_frame) with the stripe owner rotating over
        the frames (frame i renders owner (rank + first_frame + i) % num_ranks), None renders whole frames."""
        n = len(poses)
        P = _fa(np.asarray(poses, np.float32).reshape(n, 6))
        ptrs = (_vp * n)(*[int(x) if not isinstance(x, _vp) else x.value for x in d_ptrs])
        host, cam_h, scene_h, Pp = libs()[1], self.h, scene.h, _fp(P)
        if stripes is None:
            fn = host.rth_camera_render_scene_batch

            def call():
                rc = fn(cam_h, scene_h, Pp, ptrs, pitch, n, 0)
                if rc:
                    check(rc, "Camera::render_scene_batch")
        elif len(stripes) == 4:
            fn = host.rth_camera_render_scene_stripes_batch_rotating
            sr, rk, nr, first = stripes

            def call():
                rc = fn(cam_h, scene_h, Pp, ptrs, pitch, n, sr, rk, nr, first, 0)
                if rc:
                    check(rc, "Camera::render_scene_stripes_batch (rotating owner)")
        else:
            fn = host.rth_camera_render_scene_stripes_batch
            sr, rk, nr = stripes

            def call():
                rc = fn(cam_h, scene_h, Pp, ptrs, pitch, n, sr, rk, nr, 0)
                if rc:
                    check(rc, "Camera::render_scene_stripes_batch")
        call._keep = (P, ptrs)
        return call

    def render_scene_stripes_batch(self, scene, poses, d_locals, local_pitch, stripe_rows, rank, num_ranks, synchronize=False, rotate_first=None):
        n = len(poses)
        P = _fa(np.asarray(poses, np.float32).reshape(n, 6))
        ptrs = (_vp * n)(*[int(x) if not isinstance(x, _vp) else x.value for x in d_locals])
        if rotate_first is not None:
            check(libs()[1].rth_camera_render_scene_stripes_batch_rotating(self.h, scene.h, _fp(P), ptrs, local_pitch, n, stripe_rows, rank,
                                                                           num_ranks, rotate_first, 1 if synchronize else 0),
                  "Camera::render_scene_stripes_batch (rotating owner)")
            return
        check(libs()[1].rth_camera_render_scene_stripes_batch(self.h, scene.h, _fp(P), ptrs, local_pitch, n, stripe_rows, rank,
                                                              num_ranks, 1 if synchronize else 0), "Camera::render_scene_stripes_batch")

    def close(self):
        if self.h:
            libs()[1].rth_camera_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Comm:
    """RtComm: the RCCL communicator behind rt_gather / rt_all_to_all / rt_render_tiled (one process per GPU).
    Comm.unique_id() on one rank -> the 128 bytes travel to the others by the host's own means -> Comm(id, rank, n)."""

    @staticmethod
    def unique_id():
        buf = (C.c_uint8 * 128)()
        check(libs()[0].rt_comm_unique_id(buf), "rt_comm_unique_id")
        return bytes(buf)

    def __init__(self, unique_id, rank, num_ranks):
        self.h = _vp()
        self.rank, self.num_ranks = rank, num_ranks
        buf = (C.c_uint8 * 128).from_buffer_copy(unique_id)
        check(libs()[0].rt_comm_init_rank(buf, rank, num_ranks, C.byref(self.h)), "rt_comm_init_rank")

    @classmethod
    def init_all(cls, devices):
        """rt_comm_init_all: one process drives every device; returns one Comm per entry of `devices` (rank i on devices[i]).
        Collectives of several of them issued from one thread must sit between group_start() and group_end()."""
        n = len(devices)
        handles = (_vp * n)()
        check(libs()[0].rt_comm_init_all((C.c_int32 * n)(*devices), n, handles), "rt_comm_init_all")
        out = []
        for r in range(n):
            c = cls.__new__(cls)
            c.h, c.rank, c.num_ranks = _vp(handles[r]), r, n
            out.append(c)
        return out

    def info(self):
        """(rank, num_ranks, device) as the RCCL communicator itself reports them (rt_comm_info)."""
        r, n, d = C.c_int32(-1), C.c_int32(-1), C.c_int32(-1)
        check(libs()[0].rt_comm_info(self.h, C.byref(r), C.byref(n), C.byref(d)), "rt_comm_info")
        return r.value, n.value, d.value

    @staticmethod
    def last_error():
        libs()[0].rt_comm_last_error.restype = C.c_char_p
        e = libs()[0].rt_comm_last_error()
        return e.decode(errors="replace") if e else ""

    @staticmethod
    def last_error_any():
        """The most recent RT_E_COMM text of ANY thread (rt_comm_last_error() is the calling thread's own): what a watchdog
        thread reports about a main thread stuck in a collective."""
        libs()[0].rt_comm_last_error_any.restype = C.c_char_p
        e = libs()[0].rt_comm_last_error_any()
        return e.decode(errors="replace") if e else ""

    @staticmethod
    def group_start():
        check(libs()[0].rt_group_start(), "rt_group_start")

    @staticmethod
    def group_end():
        check(libs()[0].rt_group_end(), "rt_group_end")

    def gather(self, d_send, nbytes, d_recv, root=0, stream=None):
        check(libs()[0].rt_gather(self.h, d_send, nbytes, d_recv, root, stream), "rt_gather")

    def all_to_all_plan(self, send_bytes, send_offsets, recv_bytes, recv_offsets):
        """The four size arrays of rt_all_to_all as ctypes arrays, for calls repeated with the same layout."""
        arr = lambda v: (C.c_size_t * self.num_ranks)(*[int(x) for x in v])
        return arr(send_bytes), arr(send_offsets), arr(recv_bytes), arr(recv_offsets)

    def all_to_all_planned(self, d_send, d_recv, plan, stream=None):
        rc = libs()[0].rt_all_to_all(self.h, d_send, plan[0], plan[1], d_recv, plan[2], plan[3], stream)
        if rc:
            check(rc, "rt_all_to_all")

    def all_to_all(self, d_send, send_bytes, send_offsets, d_recv, recv_bytes, recv_offsets, stream=None):
        self.all_to_all_planned(d_send, d_recv, self.all_to_all_plan(send_bytes, send_offsets, recv_bytes, recv_offsets), stream)

    def close(self):
        if self.h:
            libs()[0].rt_comm_destroy(self.h)
            self.h = _vp()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class DeviceBuffer:
    """A device allocation made through the C-ABI (rt_malloc / rt_malloc_pitch)."""

    def __init__(self, nbytes=None, width_bytes=None, height=None):
        h = libs()[0]
        self.ptr = _vp()
        if nbytes is not None:
            check(h.rt_malloc(C.byref(self.ptr), nbytes), "rt_malloc")
            self.pitch, self.nbytes = None, nbytes
        else:
            pitch = C.c_size_t(0)
            check(h.rt_malloc_pitch(C.byref(self.ptr), C.byref(pitch), width_bytes, height), "rt_malloc_pitch")
            self.pitch, self.nbytes = pitch.value, pitch.value * height
        self.width_bytes, self.height = width_bytes, height

    def to_host(self, dtype=np.uint8):
        h = libs()[0]
        if self.pitch is None:
            out = np.zeros(self.nbytes // np.dtype(dtype).itemsize, dtype)
            check(h.rt_memcpy_d2h(out.ctypes.data, self.ptr, self.nbytes, None), "rt_memcpy_d2h")
            return out
        out = np.zeros((self.height, self.width_bytes), np.uint8)
        check(h.rt_memcpy2d_d2h(out.ctypes.data, self.width_bytes, self.ptr, self.pitch, self.width_bytes, self.height, None),
              "rt_memcpy2d_d2h")
        return out

    def free(self):
        if self.ptr:
            libs()[0].rt_free(self.ptr)
            self.ptr = _vp()

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def render_debug(scene, camera):
    """One frame through rt_render_debug -> dict(img[h,w,3], hit_inst, hit_tri, pops, aabb, tris, inside)."""
    h = libs()[0]
    W, H = camera.width, camera.height
    img = DeviceBuffer(width_bytes=W * 3, height=H)
    names = ("hit_inst", "hit_tri", "pops", "aabb", "tris", "inside")
    bufs = [DeviceBuffer(nbytes=W * H * 4) for _ in names]
    planes = RtDebugPlanes(*[b.ptr for b in bufs])
    p = camera.params()
    check(h.rt_render_debug(scene.device_handle, C.byref(p), img.ptr, img.pitch, C.byref(planes), None, 1), "rt_render_debug")
    out = dict(img=img.to_host().reshape(H, W, 3))
    for n, b in zip(names, bufs):
        out[n] = b.to_host(np.int32).reshape(H, W)
        b.free()
    img.free()
    return out


def render_ids(scene, camera):
    """One frame through rt_render_ids (the PRODUCTION kernel plus its hit-id planes) -> dict(img, hit_inst, hit_tri)."""
    h = libs()[0]
    W, H = camera.width, camera.height
    img = DeviceBuffer(width_bytes=W * 3, height=H)
    inst, tri = DeviceBuffer(nbytes=W * H * 4), DeviceBuffer(nbytes=W * H * 4)
    p = camera.params()
    check(h.rt_render_ids(scene.device_handle, C.byref(p), img.ptr, img.pitch, inst.ptr, tri.ptr, None, 1), "rt_render_ids")
    out = dict(img=img.to_host().reshape(H, W, 3), hit_inst=inst.to_host(np.int32).reshape(H, W),
               hit_tri=tri.to_host(np.int32).reshape(H, W))
    for b in (img, inst, tri):
        b.free()
    return out


def render(scene, camera):
    """One frame through Camera::render_scene -> img[h,w,3] uint8 (uchar3 .x .y .z order)."""
    W, H = camera.width, camera.height
    img = DeviceBuffer(width_bytes=W * 3, height=H)
    camera.render_scene(scene, img.ptr, img.pitch, synchronize=True)
    out = img.to_host().reshape(H, W, 3)
    img.free()
    return out


def render_ex(scene, camera):
    """One extension frame (camera.set_options) -> dict(img[h,w,3], total_pops[h,w])."""
    W, H = camera.width, camera.height
    img = DeviceBuffer(width_bytes=W * 3, height=H)
    pops = DeviceBuffer(nbytes=W * H * 4)
    camera.render_scene_ex(scene, img.ptr, img.pitch, pops.ptr, synchronize=True)
    out = dict(img=img.to_host().reshape(H, W, 3), total_pops=pops.to_host(np.int32).reshape(H, W))
    img.free()
    pops.free()
    return out


def read_image(path):
    """PNG / baseline JPEG / binary PPM file -> [h, w, 3] uint8 B,G,R (the decoders behind Material::upload_texture)."""
    w, h = C.c_int32(0), C.c_int32(0)
    host = libs()[1]
    rc = host.rth_read_image_bgr(os.fsencode(path), None, 0, C.byref(w), C.byref(h))
    if rc:
        raise RtError("read_image(%s): %s" % (path, host.rth_last_error().decode()))
    out = np.empty((h.value, w.value, 3), np.uint8)
    check(host.rth_read_image_bgr(os.fsencode(path), out.ctypes.data, out.nbytes, C.byref(w), C.byref(h)), "read_image")
    return out


def write_png(path, img_bgr):
    """[h, w, 3] uint8 B,G,R image -> RGB PNG (host side of display_image's out.png)."""
    a = np.ascontiguousarray(img_bgr, np.uint8)
    check(libs()[1].rth_write_png_bgr(os.fsencode(path), a.ctypes.data, a.shape[1], a.shape[0], a.strides[0]), "write_png")


class Timer:
    """hipEvent pair on a given stream (rt_timer_*)."""

    def __init__(self):
        self.h = _vp()
        check(libs()[0].rt_timer_create(C.byref(self.h)), "rt_timer_create")

    def start(self, stream=None):
        check(libs()[0].rt_timer_start(self.h, stream), "rt_timer_start")

    def stop(self, stream=None):
        check(libs()[0].rt_timer_stop(self.h, stream), "rt_timer_stop")

    def elapsed_ms(self):
        ms = C.c_float(0)
        check(libs()[0].rt_timer_elapsed_ms(self.h, C.byref(ms)), "rt_timer_elapsed_ms")
        return ms.value

    def close(self):
        if self.h:
            libs()[0].rt_timer_destroy(self.h)
            self.h = _vp()
