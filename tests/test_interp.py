"""Script front-end: the .acn interpreter in libactinon_host.so.

Checked here, without a GPU: (1) the evaluation rules of the language on a script of our own whose expected
numbers follow from the reference's evaluator (src/interpreter.c:1412-1730); (2) object operators and container
rules through the flattened scene; (3) on the reference's shipped scripts, when /root/reference is present, the
interpreter reproduces node for node the scenes that actinon_amd/host/acn_scenes.c builds by direct API calls
(two independent transcriptions of the same scripts), and every shipped script interprets."""
import ctypes as C
import glob
import math
import os

import numpy as np
import pytest

import actinon_amd as A
from actinon_amd import abi
from actinon_amd._lib import host, INTERP_SYMBOLS
from test_host_scene import declared_functions

HERE = os.path.dirname(os.path.abspath(__file__))
SCRIPTS = os.path.join(HERE, "scripts")
REF = "/root/reference/src_acn"
SKIP = A.Scene.AUTOENV_SKIP

needs_reference = pytest.mark.skipif(not os.path.isdir(REF), reason="reference scripts not present on this machine")


def test_interp_symbols_exported():
    names = declared_functions("acn_interp.h")
    assert set(names) == set(INTERP_SYMBOLS), set(names) ^ set(INTERP_SYMBOLS)
    for n in names:
        assert hasattr(host, n), n


def matter_nodes(flat):
    return [flat.node(i) for i in flat.elems_of(flat.c.matter_root)]


def test_language_rules():
    flat = A.Scene.from_script(os.path.join(SCRIPTS, "language.acn"), SKIP).flatten()
    got = [n.prm[0] for n in matter_nodes(flat)]
    expect = [5, 14, 10, 0.25, 3, -10, 0.1 * 3, 150, 32, math.sin((math.pi / 180.0) * 90), 3, 5, 6, 6, 4, 7, 8, 6, 1, 3, 13,
              (3 * (1.0 / 4)) * 8, -1, 42]
    assert len(got) == len(expect)
    for i, (g, x) in enumerate(zip(got, expect)):
        assert g == x, f"emit #{i}: got {g!r}, expected {x!r}"
    # the literal rule is the reference's digit loop (interpreter.c:247-281), not strtod
    assert got[6] != 0.3 and got[6] == 0.30000000000000004


def test_csg_operators_and_containers():
    sc = A.Scene.from_script(os.path.join(SCRIPTS, "csg.acn"), SKIP)
    assert (sc.prm.image_width, sc.prm.image_height, sc.prm.direct_samples, sc.prm.path_samples) == (64, 48, 4, 0)
    assert list(sc.prm.camera_view_direction) == [0, 8, -2]
    flat = sc.flatten()
    m = matter_nodes(flat)
    T = abi
    assert [n.type for n in m] == [T.ACN_PAIR_INSIDE, T.ACN_PAIR_INSIDE, T.ACN_PAIR_OUTSIDE, T.ACN_SCALE, T.ACN_SQUAROID,
                                   T.ACN_PAIR_INSIDE, T.ACN_COMPOUND]
    slab, lens, blob, egg, rod, barrel, pile = m
    # slab = ( plane + z/2 ) & ( !plane - z/2 )
    a, b = flat.node(slab.child0), flat.node(slab.child1)
    assert a.type == T.ACN_PLANE and list(a.pos) == [0, 0, 0.5]
    assert b.type == T.ACN_NEG and list(flat.node(b.child0).pos) == [0, 0, -0.5]
    assert slab.refractive_index == 1.46 and list(slab.transparency) == [0.8, 0.9, 0.9]      # "glass"
    assert list(flat.node(lens.child1).pos) == [1, 0, 0] and lens.diffuse_reflectivity == 1 and lens.sigma == 0.29
    assert list(blob.color) == [0.9, 0.1, 0.1] and flat.node(blob.child1).prm[0] == 0.5
    assert egg.chromatic_reflectivity == 1 and flat.node(egg.child0).type == T.ACN_SPHERE
    # rod: cylinder * rotx( 90 ) * 2 + vec( 3, 0, 0 )
    assert list(rod.pos) == [3, 0, 0]
    assert np.allclose(list(rod.rax), [1, 0, 0, 0, 0, 1, 0, -1, 0], atol=1e-15)  # rows are the rotated axes
    # balanced composite of 3: pair( e0, pair( e1, e2 ) )  (container.c:368-386)
    assert flat.node(barrel.child0).type == T.ACN_SPHERE and flat.node(barrel.child1).type == T.ACN_PAIR_INSIDE
    # a compound with an envelope stays nested (compound.c:167-183)
    assert pile.flags & T.ACN_NODE_HAS_ENVELOPE and list(pile.env_pos) == [5.5, 0, 0] and pile.env_radius == 1
    assert pile.child1 == 2
    # the lamp went to scene.light (scene.c:241-249)
    lights = [flat.node(i) for i in flat.elems_of(flat.c.light_root)]
    assert len(lights) == 1 and lights[0].radiance == 20 and list(lights[0].pos) == [-3, -3, 5]


def test_errors_carry_file_and_line(tmp_path):
    p = tmp_path / "bad.acn"
    p.write_text("def a = 1;\n\na = b + 1;\n")
    with pytest.raises(A.AcnError) as e:
        A.Scene.from_script(p, SKIP)
    assert "bad.acn:3" in str(e.value) and "Unknown name 'b'" in str(e.value)
    p.write_text("def a = 1;\ndef a = 2;\n")
    with pytest.raises(A.AcnError) as e:
        A.Scene.from_script(p, SKIP)
    assert "already defined" in str(e.value)
    p.write_text("def s = scene_s; s.push( create_sphere( 1 ) & 3 );")
    with pytest.raises(A.AcnError) as e:
        A.Scene.from_script(p, SKIP)
    assert "Cannot evaluate 'object' AND 'int'" in str(e.value)
    with pytest.raises(A.AcnError):
        A.Scene.from_script(tmp_path / "missing.acn", SKIP)
    # string_fa never truncates silently: what does not fit its buffers is an error
    p.write_text('def t = "' + "x" * 300 + '";\ndef u = string_fa( "#<sc_t>", t );\n')
    with pytest.raises(A.AcnError) as e:
        A.Scene.from_script(p, SKIP)
    assert "bad.acn:2" in str(e.value) and "longer than" in str(e.value)
    p.write_text('def u = string_fa( "' + "y" * 900 + '#<s3_t*>", 7 );\n')
    with pytest.raises(A.AcnError) as e:
        A.Scene.from_script(p, SKIP)
    assert "longer than" in str(e.value)


def test_create_image_hook_and_readonly_fs(tmp_path):
    """run_script hands every create_image call to the hook; with readonly_fs the script cannot touch files."""
    p = tmp_path / "two.acn"
    marker = tmp_path / "touched"
    p.write_text(f'''def scene = scene_s;
scene.push( create_sphere( 1 ) );
file_touch( "{marker}" );
scene.create_image( "a.pnm" );
scene.push( create_sphere( 2 ) + vecx( 3 ) );
scene.create_image( #source_file_name + ".b.pnm" );
''')
    seen = []
    A.run_script(p, on_create_image=lambda sc, f: seen.append((f, sc.objects() if sc.ptr else len(matter_nodes(sc.flatten())))),
                 auto_envelope=SKIP, readonly_fs=True)
    assert [s[0] for s in seen] == ["a.pnm", str(p) + ".b.pnm"] and [s[1] for s in seen] == [1, 2]
    assert not marker.exists()


def nodes_close(a, b, rtol=1e-12):
    assert a.n_nodes == b.n_nodes and a.c.n_elems == b.c.n_elems
    assert list(a.c.elems[:a.c.n_elems]) == list(b.c.elems[:b.c.n_elems])
    for i in range(a.n_nodes):
        na, nb = a.node(i), b.node(i)
        for fname, _ in abi.Node._fields_:
            va, vb = getattr(na, fname), getattr(nb, fname)
            if hasattr(va, "__len__"):
                assert np.allclose(list(va), list(vb), rtol=rtol, atol=1e-15), (i, fname)
            elif isinstance(va, float):
                assert math.isclose(va, vb, rel_tol=rtol, abs_tol=1e-15), (i, fname)
            else:
                assert va == vb, (i, fname)


def test_texture_maps_and_distance_functions_from_scripts(tmp_path):
    """beth_object() for the types set_texture_field / set_distance_function take (closures.c:446-456, objects.c:1510-1517,
    1691-1710, textures.c:82-138, distance.c:30-92): tests/scripts/textured.acn is, node for node and texture for texture,
    the scene tests/scenes_util.py::build_textured makes through the C API -- its torus assembled by hand from an
    obj_distance_s, a distance_torus_s, scale and envelope where the builder calls create_torus."""
    import scenes_util as S
    a = A.Scene.from_script(os.path.join(SCRIPTS, "textured.acn"), SKIP).flatten()
    b = S.build_textured().flatten()
    nodes_close(a, b)
    assert a.c.n_textures == b.c.n_textures == 5
    for i in range(a.c.n_textures):
        ta, tb = a.c.textures[i], b.c.textures[i]
        for fname, _ in type(ta)._fields_:
            va, vb = getattr(ta, fname), getattr(tb, fname)
            # (numbers differ by the script language's literal rule, as in nodes_close)
            assert np.allclose(list(va), list(vb), rtol=1e-12, atol=0) if hasattr(va, "__len__") else (va == vb or math.isclose(va, vb, rel_tol=1e-12)), (i, fname)
    # reflected members read back; defaults of the def strings; members of the distance object
    p = tmp_path / "t.acn"
    p.write_text("""def scene = scene_s;
def t = beth_object( "txm_chess_s" );
def d = beth_object( "distance_torus_s" );
def o = beth_object( "obj_distance_s" );
def r = [] : t.scale : d.ex_radius : o.cycles : o.inv_scale;
t.scale = 3; d.ex_radius = 0.25; o.cycles = 50; o.scale( 4 ); t.color2 = color( 0.5, 0.25, 1 );
r = r : t.scale : d.ex_radius : o.cycles : o.inv_scale : t.color2.y : t.color1.x;
for x ( in r ) scene.push( create_sphere( x ) );
o.set_distance_function( beth_object( "distance_sphere_s" ) );
scene.push( o );
scene.create_image( "t.pnm" );
""")
    f = A.Scene.from_script(p, SKIP).flatten()
    radii = [f.node(i).prm[0] for i in f.elems_of(f.c.matter_root)]
    assert radii[:10] == [1.0, 0.5, 200.0, 1.0, 3.0, 0.25, 50.0, 0.25, 0.25, 0.0]
    last = f.node(f.elems_of(f.c.matter_root)[-1])
    assert (last.type, last.sdf_kind, last.cycles, last.prm[0]) == (abi.ACN_DISTANCE, abi.ACN_SDF_SPHERE, 50, 0.25)
    for text, message in (
            ('def s = create_sphere( 1 ); s.set_texture_field( 3 );', "Texture map expected."),
            ('def s = create_sphere( 1 ); s.set_texture_field( beth_object( "distance_sphere_s" ) );', "Texture map expected."),
            ('def s = create_sphere( 1 ); s.set_distance_function( beth_object( "distance_sphere_s" ) );', "must be 'obj_distance_s'"),
            ('def o = beth_object( "obj_distance_s" ); o.set_distance_function( beth_object( "txm_plain_s" ) );', "'txm_plain_s' cannot be used as distance function"),
            ('def t = beth_object( "txm_plain_s" ); t.color1 = color( 1, 1, 1 );', "'txm_plain_s' has no element named 'color1'"),
            ('def t = beth_object( "txm_plain_s" ); t.color = 1;', "Color expected."),
            ('def t = beth_object( "bcore_arr_s" );', "registry is not available")):
        p.write_text(text + "\n")
        with pytest.raises(A.AcnError) as e:
            A.Scene.from_script(p, SKIP)
        assert message in str(e.value), (text, str(e.value))


@needs_reference
@pytest.mark.parametrize("name", ["primitives", "wine_glass", "diamond"])
def test_shipped_scripts_match_direct_builders(name):
    """Two independent routes to the same scene: the interpreter on the reference's script, and the C builder
    written from that script by hand.  Structure must be identical; numbers differ only by the literal rule and
    by a / b being a * ( 1 / b ) in the script language (<= a few ulp)."""
    a = A.Scene.from_script(f"{REF}/{name}.acn", SKIP)
    b = A.Scene.build(name)
    nodes_close(a.flatten(), b.flatten())
    for fname, _ in abi.Params._fields_:
        va, vb = getattr(a.prm, fname), getattr(b.prm, fname)
        if hasattr(va, "__len__"):
            assert np.allclose(list(va), list(vb), rtol=1e-12)
        else:
            assert va == vb or math.isclose(va, vb, rel_tol=1e-12), fname
    assert (a.s.gradient_cycles, a.s.gradient_samples) == (b.s.gradient_cycles, b.s.gradient_samples)


@needs_reference
def test_every_shipped_script_interprets():
    files = sorted(glob.glob(f"{REF}/*.acn")) + [f"{REF}/{d}/{d}.acn" for d in
                                                ("hanging_lamp", "hanging_lamps_in_row", "paraffin_lamp", "paraffin_lamp_on_ledge")]
    assert len(files) == 12
    before = sorted(os.listdir(REF))
    sizes = {}
    for f in files:
        sc = A.Scene.from_script(f, SKIP)
        fl = sc.flatten()
        sizes[os.path.basename(f)] = (sc.objects(), fl.n_nodes, sc.prm.image_width, sc.prm.image_height)
    assert sorted(os.listdir(REF)) == before          # readonly_fs: nothing written next to the scripts
    assert sizes["hanging_lamp.acn"][2:] == (600, 800) and sizes["hanging_lamp.acn"][1] > 2000
    assert sizes["pyramid.acn"][0] == 37 and sizes["ruby_heart.acn"][0] == 45
    assert sizes["many_spheres.acn"][0] == 8 ** 5 + 2  # without envelopes the nested compounds dissolve (compound.c:184-190)
