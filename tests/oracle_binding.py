"""ctypes binding of the CPU oracle (oracle/libacn_oracle*.so).  TEST INFRASTRUCTURE: only tests/, bench.py's
cpu_baseline leg and __graft_entry__.smoke() import this."""
import ctypes as C
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

COUNTER_NAMES = ["lum", "trans_ray", "shadow_ray", "obj_hit", "env_test", "plane_hit", "sphere_hit", "squaroid_hit",
                 "sdf_ray", "sdf_eval", "pair_hit", "side", "cap_sample", "oren_nayar", "fresnel", "node_visit", "flop", "transc"]


class Oracle:
    def __init__(self, libm=False, path=None):
        name = "libacn_oracle_libm.so" if libm else "libacn_oracle.so"
        path = path or os.path.join(ROOT, "oracle", name)
        if not os.path.exists(path):
            raise RuntimeError(f"{path} missing: run `make oracle`")
        self.lib = C.CDLL(path)
        L = self.lib
        vp = C.c_void_p
        L.acn_oracle_render_positions.argtypes = [vp, vp, C.c_size_t, vp, C.c_uint32, C.c_int, vp]
        L.acn_oracle_render_positions_shard.argtypes = [vp, vp, C.c_size_t, vp, C.c_uint32, C.c_int, vp, C.c_uint32, C.c_uint32]
        L.acn_oracle_estimate_envelope.argtypes = [vp, C.c_int32, C.c_uint64, C.c_uint32, C.c_double, vp]
        L.acn_oracle_sphere_ray_hit.argtypes = [vp, C.c_double, vp, vp, vp]
        L.acn_oracle_sphere_ray_hit.restype = C.c_double
        L.acn_oracle_plane_ray_hit.argtypes = [vp, vp, vp, vp]
        L.acn_oracle_plane_ray_hit.restype = C.c_double
        L.acn_oracle_fresnel_reflection.argtypes = [vp, vp, C.c_double, vp]
        L.acn_oracle_fresnel_reflection.restype = C.c_double
        L.acn_oracle_fresnel_refraction.argtypes = [vp, vp, C.c_double, vp]
        L.acn_oracle_fresnel_refraction.restype = None
        L.acn_oracle_obj_ray_hit.argtypes = [vp, C.c_int32, vp, vp, vp]
        L.acn_oracle_obj_ray_hit.restype = C.c_double
        L.acn_oracle_obj_side.argtypes = [vp, C.c_int32, vp]
        L.acn_oracle_trans_hit.argtypes = [vp, vp, vp, vp, vp, vp]
        L.acn_oracle_trans_hit.restype = C.c_double
        L.acn_oracle_random_seed.argtypes = [vp, C.c_uint64]
        L.acn_oracle_random_seed.restype = C.c_uint64
        L.acn_oracle_sphere_cap.argtypes = [vp, C.c_double, vp]
        L.acn_oracle_sphere_cap.restype = None

    def math_mode(self):
        return self.lib.acn_oracle_math_mode()

    def render_positions(self, flat, pos_xy, linear=False, threads=None, counters=False, shard=None):
        """shard = (rank, world): this rank's share of a sample-sharded call (ACN_SHARD_SAMPLES); linear only."""
        pos = np.ascontiguousarray(pos_xy, dtype=np.float64).reshape(-1, 2)
        out = np.empty((pos.shape[0], 3), dtype=np.float64)
        cnt = (C.c_uint64 * len(COUNTER_NAMES))() if counters else None
        threads = threads or os.cpu_count() or 1
        rank, world = shard if shard else (0, 1)
        st = self.lib.acn_oracle_render_positions_shard(C.addressof(flat.c), pos.ctypes.data, pos.shape[0], out.ctypes.data,
                                                        1 if linear else 0, threads, cnt, rank, world)
        if st != 0:
            raise RuntimeError(f"oracle status {st}")
        if counters:
            return out, dict(zip(COUNTER_NAMES, [int(v) for v in cnt]))
        return out

    def estimate_envelope(self, flat, node, samples=1000, rseed=123, radius_factor=1.1):
        out = (C.c_double * 4)()
        st = self.lib.acn_oracle_estimate_envelope(C.addressof(flat.c), node, samples, rseed, radius_factor, out)
        if st != 0:
            raise RuntimeError(f"oracle status {st}")
        return list(out)

    @staticmethod
    def _a(v):
        return np.ascontiguousarray(v, dtype=np.float64)

    def sphere_ray_hit(self, pos, r, rp, rd):
        pos, rp, rd, nor = self._a(pos), self._a(rp), self._a(rd), np.zeros(3)
        a = self.lib.acn_oracle_sphere_ray_hit(pos.ctypes.data, r, rp.ctypes.data, rd.ctypes.data, nor.ctypes.data)
        return a, nor

    def plane_ray_hit(self, pos, nor, rp, rd):
        pos, nor, rp, rd = self._a(pos), self._a(nor), self._a(rp), self._a(rd)
        return self.lib.acn_oracle_plane_ray_hit(pos.ctypes.data, nor.ctypes.data, rp.ctypes.data, rd.ctypes.data)

    def fresnel_reflection(self, d, n, trix):
        d, n, out = self._a(d), self._a(n), np.zeros(3)
        r = self.lib.acn_oracle_fresnel_reflection(d.ctypes.data, n.ctypes.data, trix, out.ctypes.data)
        return r, out

    def fresnel_refraction(self, d, n, trix):
        d, n, out = self._a(d), self._a(n), np.zeros(3)
        self.lib.acn_oracle_fresnel_refraction(d.ctypes.data, n.ctypes.data, trix, out.ctypes.data)
        return out

    def obj_ray_hit(self, flat, node, rp, rd):
        rp, rd, nor = self._a(rp), self._a(rd), np.zeros(3)
        a = self.lib.acn_oracle_obj_ray_hit(C.addressof(flat.c), node, rp.ctypes.data, rd.ctypes.data, nor.ctypes.data)
        return a, nor

    def obj_side(self, flat, node, pos):
        pos = self._a(pos)
        return self.lib.acn_oracle_obj_side(C.addressof(flat.c), node, pos.ctypes.data)

    def trans_hit(self, flat, rp, rd):
        rp, rd, nor = self._a(rp), self._a(rd), np.zeros(3)
        ex, en = C.c_int32(-1), C.c_int32(-1)
        a = self.lib.acn_oracle_trans_hit(C.addressof(flat.c), rp.ctypes.data, rd.ctypes.data, nor.ctypes.data,
                                          C.byref(ex), C.byref(en))
        return a, nor, ex.value, en.value

    def random_seed(self, v, rv):
        v = self._a(v)
        return self.lib.acn_oracle_random_seed(v.ctypes.data, rv)

    def sphere_cap(self, rv, h):
        s = C.c_uint64(rv)
        out = np.zeros(3)
        self.lib.acn_oracle_sphere_cap(C.byref(s), h, out.ctypes.data)
        return s.value, out
