"""ctypes mirror of include/actinon_hip.h and include/acn_scene.h (layout only, no logic)."""
import ctypes as C

ACN_ABI_VERSION = 2
ACN_OPT_LINEAR_OUT = 1
ACN_OPT_COUNT_WORK = 2
ACN_OPT_STAGE_TIMING = 4
ACN_SHARD_NONE, ACN_SHARD_SAMPLES = 0, 1
ACN_SHARD_TILE = 256

ACN_OK, ACN_ERR_ARG, ACN_ERR_UNSUPPORTED, ACN_ERR_NO_FOV, ACN_ERR_DEVICE, ACN_ERR_CANCELLED = 0, -1, -2, -3, -4, -5

NODE_TYPES = {1: "plane", 2: "sphere", 3: "squaroid", 4: "distance", 5: "pair_inside", 6: "pair_outside",
              7: "neg", 8: "scale", 9: "compound"}
ACN_PLANE, ACN_SPHERE, ACN_SQUAROID, ACN_DISTANCE, ACN_PAIR_INSIDE, ACN_PAIR_OUTSIDE, ACN_NEG, ACN_SCALE, ACN_COMPOUND = range(1, 10)
ACN_SDF_SPHERE, ACN_SDF_TORUS = 0, 1   # enum acn_sdf_kind
ACN_NODE_HAS_ENVELOPE = 1


class Node(C.Structure):
    _fields_ = [("type", C.c_int32), ("flags", C.c_uint32), ("child0", C.c_int32), ("child1", C.c_int32),
                ("sdf_kind", C.c_int32), ("cycles", C.c_int32), ("texture", C.c_int32), ("reserved", C.c_int32),
                ("pos", C.c_double * 3), ("rax", C.c_double * 9), ("env_pos", C.c_double * 3), ("env_radius", C.c_double),
                ("prm", C.c_double * 4), ("color", C.c_double * 3), ("radiance", C.c_double),
                ("refractive_index", C.c_double), ("fresnel_reflectivity", C.c_double),
                ("chromatic_reflectivity", C.c_double), ("diffuse_reflectivity", C.c_double), ("sigma", C.c_double),
                ("surface_roughness", C.c_double), ("transparency", C.c_double * 3), ("pad_", C.c_double)]


class Params(C.Structure):
    _fields_ = [("image_width", C.c_uint64), ("image_height", C.c_uint64), ("gamma", C.c_double),
                ("background_color", C.c_double * 3), ("camera_position", C.c_double * 3),
                ("camera_view_direction", C.c_double * 3), ("camera_top_direction", C.c_double * 3),
                ("camera_focal_length", C.c_double), ("trace_depth", C.c_uint64), ("trace_min_intensity", C.c_double),
                ("direct_samples", C.c_uint64), ("path_samples", C.c_uint64), ("max_path_length", C.c_double),
                ("experimental_level", C.c_int64)]


class Texture(C.Structure):
    _fields_ = [("kind", C.c_int32), ("reserved", C.c_int32), ("color1", C.c_double * 3), ("color2", C.c_double * 3),
                ("scale", C.c_double)]


class FlatScene(C.Structure):
    _fields_ = [("abi_version", C.c_uint32), ("n_nodes", C.c_uint32), ("n_elems", C.c_uint32),
                ("light_root", C.c_int32), ("matter_root", C.c_int32), ("reserved", C.c_uint32),
                ("nodes", C.POINTER(Node)), ("elems", C.POINTER(C.c_int32)), ("params", Params),
                ("n_textures", C.c_uint32), ("reserved2", C.c_uint32), ("textures", C.POINTER(Texture))]


class RenderOpts(C.Structure):
    _fields_ = [("flags", C.c_uint32), ("struct_size", C.c_uint32), ("cancel", C.POINTER(C.c_int)), ("stream", C.c_void_p),
                ("shard_mode", C.c_uint32), ("shard_rank", C.c_uint32), ("shard_world", C.c_uint32), ("reserved2", C.c_uint32)]


class V3(C.Structure):
    _fields_ = [("x", C.c_double), ("y", C.c_double), ("z", C.c_double)]


class M3(C.Structure):
    _fields_ = [("x", V3), ("y", V3), ("z", V3)]


class SceneStruct(C.Structure):
    """struct acn_scene (include/acn_scene.h)"""
    _fields_ = [("threads", C.c_uint64), ("gradient_threshold", C.c_double), ("gradient_samples", C.c_uint64),
                ("gradient_cycles", C.c_uint64), ("prm", Params), ("light", C.c_void_p), ("matter", C.c_void_p),
                ("device", C.c_int)]


assert C.sizeof(Node) == 304, C.sizeof(Node)
