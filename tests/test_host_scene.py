"""Host side: the C-ABI libraries load and export every declared symbol; scene assembly reproduces the reference's
construction rules; errors surface as status codes, never as a CPU fallback."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import actinon_amd as A
from actinon_amd import abi
from actinon_amd._lib import hip, host, HIP_SYMBOLS, HOST_SYMBOLS

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions(header):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(acn_[a-z0-9_]+)\s*\(", text)))


def test_every_declared_symbol_is_exported():
    for header, lib, listed in (("actinon_hip.h", hip, HIP_SYMBOLS), ("acn_scene.h", host, HOST_SYMBOLS)):
        names = declared_functions(header)
        assert set(names) == set(listed), set(names) ^ set(listed)
        for n in names:
            assert hasattr(lib, n), f"{n} declared in include/{header} but not exported"


def test_no_gpu_means_loud_failure():
    """No CPU fallback: without a device the upload fails with ACN_ERR_DEVICE and says so."""
    if A.device_count() > 0:
        pytest.skip("a GPU is present")
    flat = A.Scene.build("primitives").flatten()
    with pytest.raises(A.AcnError) as e:
        A.Handle(flat)
    assert e.value.status == abi.ACN_ERR_DEVICE and "no CPU fallback" in str(e.value)


def test_validation_mirrors_reference_aborts():
    # unsupported experimental level: bcore_err_fa in scene.c:1004-1007 -> status code
    sc = A.Scene.build("primitives", experimental_level=1)
    with pytest.raises(A.AcnError) as e:
        A.Handle(sc.flatten())
    assert e.value.status == abi.ACN_ERR_UNSUPPORTED
    # a light without fov function (objects.c:254-258): a radiant ellipsoid
    sc = A.Scene()
    el = host.acn_obj_squaroid_s_create_ellipsoid(1, 1, 1)
    host.acn_obj_set_radiance(el, 5.0)
    sc.push(el)
    host.acn_obj_discard(el)
    with pytest.raises(A.AcnError) as e:
        A.Handle(sc.flatten())
    assert e.value.status == abi.ACN_ERR_NO_FOV
    # chess texture on an object type without projection function (objects.c:240-245 aborts at shading time)
    sc = A.Scene()
    el = host.acn_obj_squaroid_s_create_ellipsoid(1, 1, 1)
    host.acn_obj_set_texture_field_chess(el, A.v3(1, 0, 0), A.v3(0, 0, 1), 1.0)
    sc.push(el)
    host.acn_obj_discard(el)
    with pytest.raises(A.AcnError) as e:
        A.Handle(sc.flatten())
    assert e.value.status == abi.ACN_ERR_UNSUPPORTED
    # corrupt flat scene
    flat = A.Scene.build("primitives").flatten()
    flat.c.abi_version = 99
    with pytest.raises(A.AcnError) as e:
        A.Handle(flat)
    assert e.value.status == abi.ACN_ERR_ARG


def test_defaults_and_materials():
    sc = A.Scene()
    p = sc.prm
    # scene.c:185-213
    assert (p.image_width, p.image_height, p.gamma, p.trace_depth, p.direct_samples, p.path_samples) == (800, 600, 1.0, 11, 100, 0)
    assert p.max_path_length == 1e30 and p.trace_min_intensity == 0 and sc.s.gradient_cycles == 1
    s = host.acn_obj_sphere_s_create(2.0)
    sc.push(s)
    n = sc.flatten()
    node = n.node(n.elems_of(n.c.matter_root)[0])
    # properties_s_init_a zeroes everything but pos/rax/color (objects.c:167-177)
    assert list(node.color) == [0.7, 0.7, 0.7] and node.refractive_index == 0 and node.diffuse_reflectivity == 0
    assert list(node.rax) == [1, 0, 0, 0, 1, 0, 0, 0, 1] and node.prm[0] == 2.0 and node.texture == -1
    assert host.acn_obj_set_material(s, b"diffuse_polished") == 0
    assert host.acn_obj_set_material(s, b"unobtainium") == abi.ACN_ERR_ARG
    host.acn_obj_set_refractive_index(s, 1.0)      # objects.c:436-448
    sc.clear()
    sc.push(s)
    node = sc.flatten()
    node = node.node(node.elems_of(node.c.matter_root)[0])
    assert node.fresnel_reflectivity == 0.0 and node.sigma == 0.29 and node.diffuse_reflectivity == 1.0
    host.acn_obj_discard(s)


def test_light_matter_split_and_compound_rules():
    sc = A.Scene()
    light = host.acn_obj_sphere_s_create(0.5)
    host.acn_obj_set_radiance(light, 30.0)
    plane = host.acn_obj_plane_s_create()
    assert sc.push(light) == 1 and sc.push(plane) == 1
    # compound without envelope is inlined, with envelope it is kept nested (compound.c:166-182)
    cmp = host.acn_compound_s_create()
    a = host.acn_obj_sphere_s_create(1.0)
    b = host.acn_obj_sphere_s_create(1.0)
    host.acn_obj_move(b, A.v3(3, 0, 0))
    host.acn_compound_s_push(cmp, a)
    host.acn_compound_s_push(cmp, b)
    assert sc.push(cmp) == 0
    flat = sc.flatten()
    assert len(flat.elems_of(flat.c.light_root)) == 1 and len(flat.elems_of(flat.c.matter_root)) == 3
    host.acn_obj_set_envelope(cmp, A.v3(1.5, 0, 0), 2.6)
    sc.push(cmp)
    flat = sc.flatten()
    els = flat.elems_of(flat.c.matter_root)
    assert len(els) == 4 and flat.node(els[3]).type == abi.ACN_COMPOUND and flat.node(els[3]).flags & 1
    assert sc.objects() == 5
    for o in (light, plane, cmp, a, b):
        host.acn_obj_discard(o)


def test_compound_envelope_merge_on_push():
    """compound.c:149-164: parent envelope = envelope_of_pair over pushed objects; an object without envelope drops it."""
    cmp = host.acn_compound_s_create()
    a = host.acn_obj_sphere_s_create(1.0)
    host.acn_obj_set_envelope(a, A.v3(0, 0, 0), 1.0)
    b = host.acn_obj_clone(a)
    host.acn_obj_move(b, A.v3(4, 0, 0))
    host.acn_compound_s_push(cmp, a)
    host.acn_compound_s_push(cmp, b)
    env = (C.c_double * 4)()
    assert host.acn_obj_get_envelope(cmp, env) == 1
    assert np.allclose(list(env), [2, 0, 0, 3])        # objects.c:105-136
    c = host.acn_obj_plane_s_create()
    host.acn_compound_s_push(cmp, c)
    assert host.acn_obj_get_envelope(cmp, env) == 0
    for o in (cmp, a, b, c):
        host.acn_obj_discard(o)


def test_transform_rules():
    # squaroid scale: r *= f^2 (objects.c:831); distance: inv_scale /= f (objects.c:970); torus ctor (closures.c:568-591)
    t = host.acn_obj_torus_create(0.35, 0.15)
    flat = A.Flat()
    node = C.c_int32()
    A.check(host.acn_obj_flatten(t, C.byref(flat.c), C.byref(node)), "flatten")
    flat._owned = True
    n = flat.node(node.value)
    assert n.type == abi.ACN_DISTANCE and n.cycles == 200 and n.sdf_kind == 1
    assert n.prm[0] == 1.0 * (1.0 / 0.35) and n.prm[1] == 0.15 / 0.35 and n.env_radius == (0.35 + 0.15) * 1.01
    cyl = host.acn_obj_squaroid_s_create_cylinder(1, 1)
    host.acn_obj_scale(cyl, 0.08)
    f2 = A.Flat()
    A.check(host.acn_obj_flatten(cyl, C.byref(f2.c), C.byref(node)), "flatten")
    f2._owned = True
    assert list(f2.node(node.value).prm) == [1.0, 1.0, 0.0, -1 * (0.08 * 0.08)]
    # rotation acts on the rows of rax and on pos (objects.c:185-190)
    pl = host.acn_obj_plane_s_create()
    host.acn_obj_move(pl, A.v3(0, 0, 1))
    m = host.acn_rotx(90)
    host.acn_obj_rotate(pl, C.byref(m))
    f3 = A.Flat()
    A.check(host.acn_obj_flatten(pl, C.byref(f3.c), C.byref(node)), "flatten")
    f3._owned = True
    nz = list(f3.node(node.value).rax)[6:9]
    assert np.allclose(nz, [0, -1, 0], atol=1e-15) and np.allclose(list(f3.node(node.value).pos), [0, -1, 0], atol=1e-15)
    for o in (t, cyl, pl):
        host.acn_obj_discard(o)


def test_distance_object_by_hand_equals_the_torus_constructor():
    """obj_distance_s with its def-string defaults (objects.c:853-861) + set_distance_function (objects.c:1691-1710) + scale +
    envelope is what create_torus does (closures.c:568-591); the members a script reaches by reflection."""
    d = host.acn_obj_distance_s_create()
    v = C.c_double()
    assert host.acn_obj_get_field(d, b"cycles", C.byref(v)) == 1 and v.value == 200.0
    assert host.acn_obj_get_field(d, b"inv_scale", C.byref(v)) == 1 and v.value == 1.0
    assert host.acn_obj_get_field(d, b"radius", C.byref(v)) == 0
    assert host.acn_obj_set_distance_function(d, abi.ACN_SDF_TORUS, 0.15 / 0.35) == abi.ACN_OK
    assert host.acn_obj_set_distance_function(d, 7, 0.0) == abi.ACN_ERR_ARG
    sph = host.acn_obj_sphere_s_create(1.0)
    assert host.acn_obj_set_distance_function(sph, abi.ACN_SDF_SPHERE, 0.0) == abi.ACN_ERR_ARG
    host.acn_obj_scale(d, 0.35)
    host.acn_obj_set_envelope(d, A.v3(0, 0, 0), (0.35 + 0.15) * 1.01)
    t = host.acn_obj_torus_create(0.35, 0.15)
    nodes = []
    for o in (d, t):
        flat = A.Flat()
        node = C.c_int32()
        A.check(host.acn_obj_flatten(o, C.byref(flat.c), C.byref(node)), "flatten")
        flat._owned = True
        n = flat.node(node.value)
        nodes.append((n.type, n.sdf_kind, n.cycles, list(n.prm), n.env_radius, list(n.env_pos), n.flags))
    assert nodes[0] == nodes[1]
    assert host.acn_obj_set_field(d, b"cycles", 64.9) == 1 and host.acn_obj_get_field(d, b"cycles", C.byref(v)) == 1 and v.value == 64.0
    for o in (d, t, sph):
        host.acn_obj_discard(o)


def test_pair_copies_properties_and_drops_envelope():
    a = host.acn_obj_sphere_s_create(1.0)
    host.acn_obj_set_envelope(a, A.v3(0, 0, 0), 1.01)
    host.acn_obj_set_color(a, A.v3(1, 0.5, 0.3))
    b = host.acn_obj_plane_s_create()
    env = (C.c_double * 4)()
    pin = host.acn_obj_pair_inside_s_create_pair(a, b)
    pout = host.acn_obj_pair_outside_s_create_pair(a, b)
    assert host.acn_obj_get_envelope(pin, env) == 1       # objects.c:1011-1018 keeps the copied envelope
    assert host.acn_obj_get_envelope(pout, env) == 0      # objects.c:1169-1173 drops it
    sc = A.Scene()
    sc.push(pin)
    flat = sc.flatten()
    top = flat.node(flat.elems_of(flat.c.matter_root)[0])
    assert top.type == abi.ACN_PAIR_INSIDE and list(top.color) == [1, 0.5, 0.3]
    assert flat.node(top.child0).type == abi.ACN_SPHERE and flat.node(top.child1).type == abi.ACN_PLANE
    for o in (a, b, pin, pout):
        host.acn_obj_discard(o)


def test_balanced_composite_shape():
    # container.c:376-392: 56 planes -> balanced binary tree of depth 6
    planes = [host.acn_obj_plane_s_create() for _ in range(56)]
    arr = (C.c_void_p * 56)(*planes)
    comp = host.acn_create_inside_composite(arr, 56)
    flat = A.Flat()
    node = C.c_int32()
    A.check(host.acn_obj_flatten(comp, C.byref(flat.c), C.byref(node)), "flatten")
    flat._owned = True

    def depth(i):
        n = flat.node(i)
        return 0 if n.type != abi.ACN_PAIR_INSIDE else 1 + max(depth(n.child0), depth(n.child1))
    assert depth(node.value) == 6
    assert sum(1 for i in range(flat.n_nodes) if flat.node(i).type == abi.ACN_PLANE) == 56
    for o in planes + [comp]:
        host.acn_obj_discard(o)


def test_baseline_scene_inventories():
    """object inventories of the BASELINE.json scenes."""
    f = A.Scene.build("primitives").flatten()
    types = [abi.NODE_TYPES[f.node(i).type] for i in f.elems_of(f.c.matter_root)]
    assert types == ["plane", "sphere", "squaroid", "distance", "squaroid", "squaroid", "squaroid", "squaroid"]
    f = A.Scene.build("wine_glass").flatten()
    els = f.elems_of(f.c.matter_root)
    assert [abi.NODE_TYPES[f.node(i).type] for i in els] == ["plane", "pair_outside", "pair_inside"]
    glass, liquid = f.node(els[1]), f.node(els[2])
    assert glass.refractive_index == 1.46 and liquid.refractive_index == 1.32
    assert np.allclose(list(glass.env_pos), [0, 0, 1.5 - 0.999]) and glass.env_radius == 1.7
    assert list(liquid.transparency) == [0.177, 9.61E-6, 9.54E-7]
    f = A.Scene.build("diamond").flatten()
    els = f.elems_of(f.c.matter_root)
    assert len(els) == 9          # floor, plate, cloth, gem, ring, 4 bars
    gem = f.node(els[3])
    assert gem.refractive_index == 2.42 and gem.flags & 1 and abs(gem.env_radius - 1.01 * 0.0472) < 1e-15
    assert sum(1 for i in range(f.n_nodes) if f.node(i).type == abi.ACN_PLANE) > 60
    f = A.Scene.build("many_spheres:2:1").flatten()
    spheres = sum(1 for i in range(f.n_nodes) if f.node(i).type == abi.ACN_SPHERE)
    assert spheres == 1 + 64      # the light + 8^2 leaves


def test_pnm_writer_and_quantisation(tmp_path):
    rgb = np.array([[0.0, 0.5, 1.0], [-1.0, 0.999, 2.0]], dtype=np.float64)
    q = A.cps_from_cl(rgb)
    assert q.tolist() == [[0, 128, 255], [0, 255, 255]]        # scene.c:76-82
    v = host.acn_cps_from_cl((C.c_double * 3)(0.0, 0.5, 1.0))
    assert v == (0 | 128 << 8 | 255 << 16)
    path = str(tmp_path / "x.pnm")
    assert host.acn_write_pnm(path.encode(), rgb.ctypes.data, 2, 1) == 0
    data = open(path, "rb").read()
    assert data == b"P6\n2 1\n255\n" + bytes([0, 128, 255, 0, 255, 255])   # scene.c:122-137
