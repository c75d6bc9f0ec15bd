"""Loads the two product libraries. There is NO fallback: a missing libactinon_hip.so is an ImportError, and a
render call without a GPU fails with ACN_ERR_DEVICE from the library itself."""
import ctypes as C
import os

from . import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
# ACN_LIBDIR: an alternative build of the two libraries (kernel variants under test); default: actinon_amd/lib
LIBDIR = os.environ.get("ACN_LIBDIR") or os.path.join(_HERE, "lib")


def _load(name):
    path = os.path.join(LIBDIR, name)
    if not os.path.exists(path):
        raise ImportError(f"{path} is missing: build it with `make` (or __graft_entry__.build()); "
                          "actinon_amd has no pure-Python or CPU rendering path")
    return C.CDLL(path, mode=C.RTLD_GLOBAL)


hip = _load("libactinon_hip.so")
host = _load("libactinon_host.so")

# symbols declared by include/actinon_hip.h
HIP_SYMBOLS = ["acn_device_count", "acn_scene_upload", "acn_scene_free", "acn_render_positions",
               "acn_render_positions_dev", "acn_render_main_pass_dev", "acn_resolve_dev", "acn_last_kernel_ms", "acn_last_stage_ms", "acn_last_counters",
               "acn_estimate_envelope", "acn_detmath_eval", "acn_last_error", "acn_shard_tile_count", "acn_shard_tile_padded",
               "acn_shard_tile_index", "acn_render_main_pass_shard_dev", "acn_shard_unpack_dev"]
# symbols declared by include/acn_scene.h
HOST_SYMBOLS = ["acn_rotx", "acn_roty", "acn_rotz", "acn_obj_plane_s_create", "acn_obj_sphere_s_create",
                "acn_obj_squaroid_s_create_squaroid", "acn_obj_squaroid_s_create_ellipsoid",
                "acn_obj_squaroid_s_create_hyperboloid1", "acn_obj_squaroid_s_create_hyperboloid2",
                "acn_obj_squaroid_s_create_cone", "acn_obj_squaroid_s_create_cylinder", "acn_obj_torus_create", "acn_obj_distance_s_create", "acn_obj_set_distance_function",
                "acn_obj_pair_inside_s_create_pair", "acn_obj_pair_outside_s_create_pair", "acn_obj_neg_s_create_neg",
                "acn_obj_scale_s_create_scale", "acn_create_inside_composite", "acn_create_outside_composite",
                "acn_obj_clone", "acn_obj_discard", "acn_obj_type", "acn_obj_move", "acn_obj_rotate", "acn_obj_scale",
                "acn_obj_set_color", "acn_obj_set_transparency", "acn_obj_set_refractive_index", "acn_obj_set_radiance",
                "acn_obj_set_fresnel_reflectivity", "acn_obj_set_chromatic_reflectivity",
                "acn_obj_set_diffuse_reflectivity", "acn_obj_set_sigma", "acn_obj_set_surface_roughness",
                "acn_obj_set_material", "acn_obj_set_texture_field_plain", "acn_obj_set_texture_field_chess",
                "acn_obj_clear_texture_field", "acn_obj_set_envelope", "acn_obj_set_auto_envelope", "acn_obj_radiance",
                "acn_obj_get_envelope", "acn_compound_s_create", "acn_compound_s_push", "acn_compound_s_get_size",
                "acn_compound_s_clear", "acn_compound_s_set_sphere_envelopes", "acn_scene_s_create",
                "acn_scene_s_discard", "acn_scene_s_clear", "acn_scene_s_push", "acn_scene_s_objects",
                "acn_scene_s_flatten", "acn_flat_scene_free", "acn_obj_flatten", "acn_lum_machine_s_run",
                "acn_scene_s_create_image_file", "acn_write_pnm", "acn_cps_from_cl", "acn_scene_primitives",
                "acn_scene_wine_glass", "acn_scene_diamond", "acn_scene_many_spheres", "acn_obj_get_pos",
                "acn_obj_sphere_s_get_radius", "acn_obj_get_field", "acn_obj_set_field", "acn_set_envelope_estimator"]
# symbols declared by include/acn_interp.h
INTERP_SYMBOLS = ["acn_interpret_file", "acn_interpret_string", "acn_interp_last_error", "acn_scene_from_script",
                  "acn_scene_s_clone"]

P = C.POINTER
vp = C.c_void_p

hip.acn_device_count.restype = C.c_int
hip.acn_scene_upload.argtypes = [P(abi.FlatScene), C.c_int, P(vp)]
hip.acn_scene_free.argtypes = [vp]
hip.acn_scene_free.restype = None
hip.acn_render_positions.argtypes = [vp, vp, C.c_size_t, vp, P(abi.RenderOpts)]
hip.acn_render_positions_dev.argtypes = [vp, vp, C.c_size_t, vp, P(abi.RenderOpts)]
hip.acn_render_main_pass_dev.argtypes = [vp, C.c_size_t, C.c_size_t, vp, P(abi.RenderOpts)]
hip.acn_resolve_dev.argtypes = [vp, vp, C.c_size_t, vp, vp, P(abi.RenderOpts)]
hip.acn_last_kernel_ms.argtypes = [vp, P(C.c_double)]
hip.acn_last_stage_ms.argtypes = [vp, P(C.c_double), C.c_int]
hip.acn_last_counters.argtypes = [vp, P(C.c_uint64), C.c_int]
hip.acn_estimate_envelope.argtypes = [vp, C.c_int32, C.c_uint64, C.c_uint32, C.c_double, P(C.c_double)]
hip.acn_detmath_eval.argtypes = [C.c_int, C.c_int, vp, vp, vp, C.c_size_t]
hip.acn_last_error.restype = C.c_char_p
hip.acn_shard_tile_count.argtypes = [C.c_size_t, C.c_uint32, C.c_uint32]
hip.acn_shard_tile_count.restype = C.c_size_t
hip.acn_shard_tile_padded.argtypes = [C.c_size_t, C.c_uint32]
hip.acn_shard_tile_padded.restype = C.c_size_t
hip.acn_shard_tile_index.argtypes = [C.c_size_t, C.c_uint32, C.c_uint32, C.c_size_t]
hip.acn_shard_tile_index.restype = C.c_size_t
hip.acn_render_main_pass_shard_dev.argtypes = [vp, C.c_size_t, C.c_size_t, C.c_uint32, C.c_uint32, vp, P(abi.RenderOpts)]
hip.acn_shard_unpack_dev.argtypes = [vp, vp, C.c_size_t, C.c_uint32, vp, P(abi.RenderOpts)]

for _n in ["acn_rotx", "acn_roty", "acn_rotz"]:
    getattr(host, _n).argtypes = [C.c_double]
    getattr(host, _n).restype = abi.M3
for _n in ["acn_obj_plane_s_create", "acn_compound_s_create", "acn_scene_primitives", "acn_scene_wine_glass",
           "acn_scene_diamond"]:
    getattr(host, _n).restype = vp
host.acn_scene_s_create.restype = P(abi.SceneStruct)
host.acn_scene_many_spheres.argtypes = [C.c_int, C.c_int]
host.acn_scene_many_spheres.restype = vp
host.acn_obj_sphere_s_create.argtypes = [C.c_double]
host.acn_obj_sphere_s_create.restype = vp
host.acn_obj_squaroid_s_create_squaroid.argtypes = [C.c_double] * 4
host.acn_obj_squaroid_s_create_squaroid.restype = vp
for _n in ["ellipsoid", "hyperboloid1", "hyperboloid2", "cone"]:
    f = getattr(host, "acn_obj_squaroid_s_create_" + _n)
    f.argtypes = [C.c_double] * 3
    f.restype = vp
host.acn_obj_squaroid_s_create_cylinder.argtypes = [C.c_double] * 2
host.acn_obj_squaroid_s_create_cylinder.restype = vp
host.acn_obj_torus_create.argtypes = [C.c_double] * 2
host.acn_obj_torus_create.restype = vp
host.acn_obj_distance_s_create.argtypes = []
host.acn_obj_distance_s_create.restype = vp
host.acn_obj_set_distance_function.argtypes = [vp, C.c_int, C.c_double]
host.acn_obj_set_distance_function.restype = C.c_int
for _n in ["acn_obj_pair_inside_s_create_pair", "acn_obj_pair_outside_s_create_pair"]:
    getattr(host, _n).argtypes = [vp, vp]
    getattr(host, _n).restype = vp
host.acn_obj_neg_s_create_neg.argtypes = [vp]
host.acn_obj_neg_s_create_neg.restype = vp
host.acn_obj_scale_s_create_scale.argtypes = [vp, abi.V3]
host.acn_obj_scale_s_create_scale.restype = vp
for _n in ["acn_create_inside_composite", "acn_create_outside_composite"]:
    getattr(host, _n).argtypes = [P(vp), C.c_size_t]
    getattr(host, _n).restype = vp
host.acn_obj_clone.argtypes = [vp]
host.acn_obj_clone.restype = vp
host.acn_obj_discard.argtypes = [vp]
host.acn_obj_discard.restype = None
host.acn_obj_type.argtypes = [vp]
host.acn_obj_move.argtypes = [vp, abi.V3]
host.acn_obj_move.restype = None
host.acn_obj_rotate.argtypes = [vp, P(abi.M3)]
host.acn_obj_rotate.restype = None
host.acn_obj_scale.argtypes = [vp, C.c_double]
host.acn_obj_scale.restype = None
for _n in ["acn_obj_set_color", "acn_obj_set_transparency"]:
    getattr(host, _n).argtypes = [vp, abi.V3]
    getattr(host, _n).restype = None
for _n in ["refractive_index", "radiance", "fresnel_reflectivity", "chromatic_reflectivity", "diffuse_reflectivity",
           "sigma", "surface_roughness"]:
    f = getattr(host, "acn_obj_set_" + _n)
    f.argtypes = [vp, C.c_double]
    f.restype = None
host.acn_obj_set_texture_field_plain.argtypes = [vp, abi.V3]
host.acn_obj_set_texture_field_plain.restype = None
host.acn_obj_set_texture_field_chess.argtypes = [vp, abi.V3, abi.V3, C.c_double]
host.acn_obj_set_texture_field_chess.restype = None
host.acn_obj_clear_texture_field.argtypes = [vp]
host.acn_obj_clear_texture_field.restype = None
host.acn_obj_set_material.argtypes = [vp, C.c_char_p]
host.acn_obj_set_envelope.argtypes = [vp, abi.V3, C.c_double]
host.acn_obj_set_envelope.restype = None
host.acn_obj_set_auto_envelope.argtypes = [vp]
host.acn_obj_radiance.argtypes = [vp]
host.acn_obj_radiance.restype = C.c_double
host.acn_obj_get_envelope.argtypes = [vp, P(C.c_double)]
host.acn_compound_s_push.argtypes = [vp, vp]
host.acn_compound_s_push.restype = None
host.acn_compound_s_get_size.argtypes = [vp]
host.acn_compound_s_get_size.restype = C.c_size_t
host.acn_compound_s_clear.argtypes = [vp]
host.acn_compound_s_clear.restype = None
host.acn_compound_s_set_sphere_envelopes.argtypes = [vp, C.c_double]
host.acn_compound_s_set_sphere_envelopes.restype = None
host.acn_scene_s_discard.argtypes = [vp]
host.acn_scene_s_discard.restype = None
host.acn_scene_s_clear.argtypes = [vp]
host.acn_scene_s_clear.restype = None
host.acn_scene_s_push.argtypes = [vp, vp]
host.acn_scene_s_push.restype = C.c_size_t
host.acn_scene_s_objects.argtypes = [vp]
host.acn_scene_s_objects.restype = C.c_size_t
host.acn_scene_s_flatten.argtypes = [vp, P(abi.FlatScene)]
host.acn_obj_get_pos.argtypes = [vp, P(C.c_double)]
host.acn_obj_get_pos.restype = None
host.acn_obj_sphere_s_get_radius.argtypes = [vp]
host.acn_obj_sphere_s_get_radius.restype = C.c_double
host.acn_obj_get_field.argtypes = [vp, C.c_char_p, P(C.c_double)]
host.acn_obj_set_field.argtypes = [vp, C.c_char_p, C.c_double]
host.acn_set_envelope_estimator.argtypes = [vp]
host.acn_set_envelope_estimator.restype = None
CREATE_IMAGE_FN = C.CFUNCTYPE(C.c_int, vp, vp, C.c_char_p)


class InterpOpts(C.Structure):
    _fields_ = [("on_create_image", CREATE_IMAGE_FN), ("ctx", vp), ("auto_envelope", C.c_int), ("readonly_fs", C.c_int),
                ("argc", C.c_int),
                ("argv", P(C.c_char_p))]


host.acn_interpret_file.argtypes = [C.c_char_p, P(InterpOpts)]
host.acn_interpret_string.argtypes = [C.c_char_p, C.c_char_p, P(InterpOpts)]
host.acn_interp_last_error.restype = C.c_char_p
host.acn_scene_from_script.argtypes = [C.c_char_p, C.c_int]
host.acn_scene_from_script.restype = vp
host.acn_scene_s_clone.argtypes = [vp]
host.acn_scene_s_clone.restype = vp
host.acn_flat_scene_free.argtypes = [P(abi.FlatScene)]
host.acn_flat_scene_free.restype = None
host.acn_obj_flatten.argtypes = [vp, P(abi.FlatScene), P(C.c_int32)]
host.acn_lum_machine_s_run.argtypes = [vp, vp, C.c_size_t]
host.acn_scene_s_create_image_file.argtypes = [vp, C.c_char_p]
host.acn_write_pnm.argtypes = [C.c_char_p, vp, C.c_size_t, C.c_size_t]
host.acn_cps_from_cl.argtypes = [P(C.c_double)]
host.acn_cps_from_cl.restype = C.c_uint32


class AcnError(RuntimeError):
    def __init__(self, status, where):
        msg = hip.acn_last_error()
        super().__init__(f"{where}: status {status} ({msg.decode() if msg else ''})")
        self.status = status


def check(status, where):
    if status != abi.ACN_OK:
        raise AcnError(status, where)
