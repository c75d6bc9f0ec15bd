"""Python face of the host-side scene model (libactinon_host.so) and of the render seam (libactinon_hip.so).

Everything here is plumbing over the C ABI: scene assembly happens in C (actinon_amd/host/*.c), rendering in the
HIP library.  Mirrors, for this path, what the reference's script interpreter does with scene_s
(/root/reference/src/scene.c:293-331): set fields, push objects, create_image."""
import ctypes as C

import numpy as np

from . import abi
from ._lib import hip, host, check, AcnError


def v3(x, y=None, z=None):
    if y is None:
        x, y, z = x
    return abi.V3(float(x), float(y), float(z))


class Flat:
    """An acn_flat_scene owned by Python (arrays allocated by libactinon_host, freed on __del__)."""

    def __init__(self):
        self.c = abi.FlatScene()
        self._owned = False

    def __del__(self):
        if getattr(self, "_owned", False):
            host.acn_flat_scene_free(C.byref(self.c))
            self._owned = False

    @property
    def n_nodes(self):
        return self.c.n_nodes

    @property
    def params(self):
        return self.c.params

    def node(self, i):
        return self.c.nodes[i]

    def elems_of(self, compound_index):
        n = self.c.nodes[compound_index]
        return [self.c.elems[n.child0 + k] for k in range(n.child1)]

    def nodes_bytes(self):
        return C.string_at(self.c.nodes, C.sizeof(abi.Node) * self.c.n_nodes)

    def save(self, path, driver=(0.1, 10, 1)):
        """Writes the flat scene as a compressed .npz (nodes / elems / params / textures as raw ABI bytes)."""
        tex = C.string_at(self.c.textures, C.sizeof(abi.Texture) * self.c.n_textures) if self.c.n_textures else b""
        np.savez_compressed(
            path, abi_version=self.c.abi_version, nodes=np.frombuffer(self.nodes_bytes(), dtype=np.uint8),
            elems=np.array(self.c.elems[:self.c.n_elems], dtype=np.int32),
            params=np.frombuffer(C.string_at(C.addressof(self.c.params), C.sizeof(abi.Params)), dtype=np.uint8),
            textures=np.frombuffer(tex, dtype=np.uint8),
            roots=np.array([self.c.light_root, self.c.matter_root], dtype=np.int32), driver=np.array(driver, dtype=np.float64))

    @classmethod
    def load(cls, path, **overrides):
        """Inverse of save(); arrays are owned by Python.  overrides set acn_params fields."""
        z = np.load(path)
        if int(z["abi_version"]) != abi.ACN_ABI_VERSION:
            raise AcnError(abi.ACN_ERR_ARG, f"{path}: flat scene of ABI {int(z['abi_version'])}, library is {abi.ACN_ABI_VERSION}")
        f = cls()
        nb = z["nodes"].tobytes()
        n = len(nb) // C.sizeof(abi.Node)
        f._nodes = (abi.Node * max(1, n)).from_buffer_copy(nb.ljust(C.sizeof(abi.Node), b"\0"))
        el = z["elems"].astype(np.int32)
        f._elems = (C.c_int32 * max(1, len(el)))(*[int(v) for v in el])
        tb = z["textures"].tobytes() if "textures" in z.files else b""
        nt = len(tb) // C.sizeof(abi.Texture)
        f._tex = (abi.Texture * max(1, nt)).from_buffer_copy(tb.ljust(C.sizeof(abi.Texture), b"\0"))
        f.c.abi_version = abi.ACN_ABI_VERSION
        f.c.n_nodes, f.c.n_elems, f.c.n_textures = n, len(el), nt
        f.c.light_root, f.c.matter_root = int(z["roots"][0]), int(z["roots"][1])
        f.c.nodes = C.cast(f._nodes, C.POINTER(abi.Node))
        f.c.elems = C.cast(f._elems, C.POINTER(C.c_int32))
        f.c.textures = C.cast(f._tex, C.POINTER(abi.Texture))
        C.memmove(C.addressof(f.c.params), z["params"].tobytes(), C.sizeof(abi.Params))
        f.driver = tuple(float(v) for v in z["driver"]) if "driver" in z.files else (0.1, 10, 1)
        for k, v in overrides.items():
            cur = getattr(f.c.params, k)
            if hasattr(cur, "__len__"):
                for i in range(len(cur)):
                    cur[i] = float(v[i])
            else:
                setattr(f.c.params, k, v)
        return f


class Scene:
    """Wraps an acn_scene* (scene_s counterpart)."""

    BUILDERS = {"primitives": "acn_scene_primitives", "wine_glass": "acn_scene_wine_glass",
                "diamond": "acn_scene_diamond"}

    def __init__(self, ptr=None):
        if ptr is None:
            ptr = C.cast(host.acn_scene_s_create(), C.c_void_p).value
        if not ptr:
            raise AcnError(abi.ACN_ERR_ARG, "scene construction failed")
        self.ptr = ptr
        self.s = C.cast(ptr, C.POINTER(abi.SceneStruct)).contents

    def __del__(self):
        if getattr(self, "ptr", None):
            host.acn_scene_s_discard(self.ptr)
            self.ptr = None

    @classmethod
    def build(cls, name, **overrides):
        """name: primitives | wine_glass | diamond | many_spheres[:levels[:exact]]; overrides set acn_params fields."""
        if name.startswith("many_spheres"):
            parts = name.split(":")
            levels = int(parts[1]) if len(parts) > 1 else 5
            exact = int(parts[2]) if len(parts) > 2 else 0
            ptr = host.acn_scene_many_spheres(levels, exact)
        else:
            ptr = getattr(host, cls.BUILDERS[name])()
        sc = cls(ptr)
        sc.set(**overrides)
        return sc

    AUTOENV_GPU, AUTOENV_SKIP = 0, 1

    @classmethod
    def from_script(cls, path, auto_envelope=0, **overrides):
        """Interprets an .acn script (include/acn_interp.h) without rendering; returns the scene as it was at the
        script's first create_image call."""
        ptr = host.acn_scene_from_script(str(path).encode(), auto_envelope)
        if not ptr:
            raise AcnError(abi.ACN_ERR_ARG, host.acn_interp_last_error().decode())
        sc = cls(ptr)
        sc.set(**overrides)
        return sc

    def set(self, **kw):
        for k, v in kw.items():
            if hasattr(self.s.prm, k):
                cur = getattr(self.s.prm, k)
                if hasattr(cur, "__len__"):
                    for i in range(len(cur)):
                        cur[i] = float(v[i])
                else:
                    setattr(self.s.prm, k, v)
            elif hasattr(self.s, k):
                setattr(self.s, k, v)
            else:
                raise AttributeError(f"scene_s has no member '{k}'")
        return self

    @property
    def prm(self):
        return self.s.prm

    def objects(self):
        return host.acn_scene_s_objects(self.ptr)

    def clear(self):
        host.acn_scene_s_clear(self.ptr)

    def push(self, obj_ptr):
        return host.acn_scene_s_push(self.ptr, obj_ptr)

    def flatten(self):
        f = Flat()
        check(host.acn_scene_s_flatten(self.ptr or self._borrowed, C.byref(f.c)), "acn_scene_s_flatten")
        f._owned = True
        return f

    def create_image_file(self, path, overwrite=True):
        C.c_int.in_dll(host, "acn_scene_s_overwrite_output_files_g").value = 1 if overwrite else 0
        check(host.acn_scene_s_create_image_file(self.ptr, path.encode()), "acn_scene_s_create_image_file")


def run_script(path, on_create_image=None, auto_envelope=0, readonly_fs=False, args=(), overwrite=True):
    """Interprets an .acn script (acn_interpret_file).  on_create_image( scene: Scene, file: str ) -> None replaces
    the render driver for `scene.create_image( file )`; None renders on the GPU and writes the PNM like the
    reference's actinon binary does."""
    from ._lib import InterpOpts, CREATE_IMAGE_FN
    C.c_int.in_dll(host, "acn_scene_s_overwrite_output_files_g").value = 1 if overwrite else 0
    raised = []

    def hook(ctx, scene_ptr, file):
        try:
            view = Scene.__new__(Scene)
            view.ptr = None                      # borrowed: the interpreter owns the scene
            view.s = C.cast(scene_ptr, C.POINTER(abi.SceneStruct)).contents
            view._borrowed = scene_ptr
            on_create_image(view, file.decode())
            return abi.ACN_OK
        except Exception as ex:                  # noqa: BLE001 - surfaced after the C call returns
            raised.append(ex)
            return abi.ACN_ERR_ARG

    opts = InterpOpts()
    cb = CREATE_IMAGE_FN(hook) if on_create_image else CREATE_IMAGE_FN()
    opts.on_create_image = cb
    opts.auto_envelope = auto_envelope
    opts.readonly_fs = 1 if readonly_fs else 0
    argv = (C.c_char_p * max(1, len(args)))(*[a.encode() for a in args])
    opts.argc = len(args)
    opts.argv = argv
    st = host.acn_interpret_file(str(path).encode(), C.byref(opts))
    if raised:
        raise raised[0]
    if st != abi.ACN_OK:
        raise AcnError(st, host.acn_interp_last_error().decode())


class Handle:
    """A flattened scene resident on one GPU (acn_scene_handle)."""

    def __init__(self, flat, device=0, count_work=False):
        self.flat = flat
        self.device = device
        self.count_work = count_work   # instrumented kernels: last_counters() is only meaningful when set
        self.stage_timing = False      # per-launch HIP events: last_stages() then carries walk / shade / hard ms
        self.cancel = None             # optional ctypes.c_int polled by the library (acn_render_opts.cancel)
        self.sample_shard = None       # (rank, world): ACN_SHARD_SAMPLES for the following render calls (linear output)
        self.h = C.c_void_p()
        check(hip.acn_scene_upload(C.byref(flat.c), device, C.byref(self.h)), "acn_scene_upload")

    def close(self):
        if getattr(self, "h", None) and self.h.value:
            hip.acn_scene_free(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        self.close()

    def _opts(self, linear, stream):
        o = abi.RenderOpts()
        o.struct_size = C.sizeof(abi.RenderOpts)
        o.flags = ((abi.ACN_OPT_LINEAR_OUT if linear else 0) | (abi.ACN_OPT_COUNT_WORK if self.count_work else 0)
                   | (abi.ACN_OPT_STAGE_TIMING if self.stage_timing else 0))
        o.stream = stream
        if self.cancel is not None:
            o.cancel = C.pointer(self.cancel)
        if self.sample_shard is not None:
            o.shard_mode = abi.ACN_SHARD_SAMPLES
            o.shard_rank, o.shard_world = self.sample_shard
        return o

    def render_positions(self, pos_xy, linear=False):
        """lum_machine_s_run on host arrays: pos_xy [n,2] float64 -> rgb [n,3] float64."""
        pos = np.ascontiguousarray(pos_xy, dtype=np.float64).reshape(-1, 2)
        out = np.empty((pos.shape[0], 3), dtype=np.float64)
        o = self._opts(linear, None)
        check(hip.acn_render_positions(self.h, pos.ctypes.data, pos.shape[0], out.ctypes.data, C.byref(o)),
              "acn_render_positions")
        return out

    def render_positions_dev(self, d_pos_ptr, n, d_out_ptr, linear=False, stream=None):
        o = self._opts(linear, stream)
        check(hip.acn_render_positions_dev(self.h, d_pos_ptr, n, d_out_ptr, C.byref(o)), "acn_render_positions_dev")

    def render_main_pass_dev(self, first, count, d_out_ptr, linear=False, stream=None):
        o = self._opts(linear, stream)
        check(hip.acn_render_main_pass_dev(self.h, first, count, d_out_ptr, C.byref(o)), "acn_render_main_pass_dev")

    def render_main_pass_shard_dev(self, first, count, rank, world, d_part_ptr, linear=True, stream=None):
        """This rank's tiles of the main pass (acn_shard_tile_*), into a part of acn_shard_tile_padded(count, world) rows."""
        o = self._opts(linear, stream)
        check(hip.acn_render_main_pass_shard_dev(self.h, first, count, rank, world, d_part_ptr, C.byref(o)),
              "acn_render_main_pass_shard_dev")

    def shard_unpack_dev(self, d_gathered_ptr, count, world, d_frame_ptr, stream=None):
        o = self._opts(True, stream)
        check(hip.acn_shard_unpack_dev(self.h, d_gathered_ptr, count, world, d_frame_ptr, C.byref(o)), "acn_shard_unpack_dev")

    def resolve_dev(self, d_linear_ptr, n, d_out_rgb_ptr=None, d_out_rgb8_ptr=None, stream=None):
        """cl_s_sat + 8-bit pack on a device-resident linear radiance buffer (after accumulation / all-reduce)."""
        o = self._opts(False, stream)
        check(hip.acn_resolve_dev(self.h, d_linear_ptr, n, d_out_rgb_ptr, d_out_rgb8_ptr, C.byref(o)), "acn_resolve_dev")

    def last_kernel_ms(self):
        ms = C.c_double()
        check(hip.acn_last_kernel_ms(self.h, C.byref(ms)), "acn_last_kernel_ms")
        return ms.value

    def last_stages(self):
        """Per-stage device time (ms) and pipeline statistics of the last render call."""
        names = ["walk_ms", "shade_ms", "finalize_ms", "total_ms", "walk_launches", "shade_launches", "finalize_launches",
                 "chunks", "retries", "levels", "peak_tasks", "peak_children", "queue_cap", "hard_ms", "hard_launches",
                 "hard_rays", "walk_rays", "path_hits", "host_syncs", "walk_steps", "flags", "private_rays", "probe_rays",
                 "workspace_bytes", "workspace_allocs"]
        buf = (C.c_double * 25)()
        check(hip.acn_last_stage_ms(self.h, buf, 25), "acn_last_stage_ms")
        return dict(zip(names, [float(v) for v in buf]))

    def last_counters(self):
        names = ["trans_rays", "shadow_rays", "obj_hits", "lum_calls", "cap_samples", "side_calls", "sdf_evals",
                 "overflows", "flop", "transcendentals"]
        buf = (C.c_uint64 * 10)()
        check(hip.acn_last_counters(self.h, buf, 10), "acn_last_counters")
        return dict(zip(names, [int(v) for v in buf]))

    PHASES = ["other", "light", "root_leaf", "prune", "m_leaf", "m_pair", "m_frame", "m_side", "shade", "compound", "fetch",
              "tail"]

    def last_counters_raw(self, n):
        buf = (C.c_uint64 * n)()
        check(hip.acn_last_counters(self.h, buf, n), "acn_last_counters")
        return [int(v) for v in buf]

    def last_phase_ticks(self):
        """{kernel: {phase: shader-clock ticks}} of a library built with -DACN_PHASE_TIMERS (all zero otherwise)"""
        buf = (C.c_uint64 * 74)()
        check(hip.acn_last_counters(self.h, buf, 74), "acn_last_counters")
        out = {}
        for k, kernel in enumerate(["walk", "hard_shadow", "hard_path", "shade"]):
            out[kernel] = {p: int(buf[10 + 16 * k + i]) for i, p in enumerate(self.PHASES)}
        return out

    def estimate_envelope(self, node, samples=1000, rseed=123, radius_factor=1.1):
        out = (C.c_double * 4)()
        check(hip.acn_estimate_envelope(self.h, node, samples, rseed, radius_factor, out), "acn_estimate_envelope")
        return list(out)


def main_pass_positions(width, height, first=0, count=None):
    """Pixel centres of the main pass, row-major (scene.c:1110-1119)."""
    n = width * height if count is None else count
    idx = np.arange(first, first + n)
    pos = np.empty((n, 2), dtype=np.float64)
    pos[:, 0] = (idx % width) + 0.5
    pos[:, 1] = (idx // width) + 0.5
    return pos


def cps_from_cl(rgb):
    """8-bit quantisation of scene.c:76-82 on an [...,3] array."""
    rgb = np.asarray(rgb)
    q = np.where(rgb > 0.0, np.where(rgb < 1.0, (rgb * 256).astype(np.int64), 255), 0)
    return q.astype(np.uint8)


def device_count():
    return hip.acn_device_count()


def detmath_eval(op, x, y=None, device=0):
    ops = {"sin": 0, "cos": 1, "tan": 2, "acos": 3, "log": 4, "exp": 5, "pow": 6, "sqrt": 7, "div": 8, "u64_to_f64": 9,
           "frexp_mant": 10}
    x = np.ascontiguousarray(x, dtype=np.float64)
    out = np.empty_like(x)
    yp = None
    if y is not None:
        y = np.ascontiguousarray(y, dtype=np.float64)
        yp = y.ctypes.data
    check(hip.acn_detmath_eval(device, ops[op], x.ctypes.data, yp, out.ctypes.data, x.size), "acn_detmath_eval")
    return out
