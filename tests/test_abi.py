"""The C-ABI library loads, exports every symbol include/hpt.h declares, agrees with the Python
mirror on struct layouts, round-trips scene blobs, and refuses to compute without a HIP device
(no CPU fallback).  No compute calls here: runs without a GPU."""
import ctypes as C
import importlib
import os
import re

import numpy as np
import pytest

from tests.util import ROOT, abi

hpt = importlib.import_module("pbrt-v2_amd.hpt")


def header_symbols():
    src = open(os.path.join(ROOT, "include", "hpt.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(hpt_[a-z_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    L = hpt.lib()
    syms = header_symbols()
    assert set(hpt.EXPORTS) == set(syms), (sorted(set(syms) ^ set(hpt.EXPORTS)))
    for s in syms:
        assert hasattr(L, s), s


def test_library_exports_nothing_but_the_c_abi():
    """The export list is generated from include/hpt.h (pbrt-v2_amd/Makefile, build/hpt.map): no hpt:: internal, launcher or kernel stub is
    part of the dynamic surface — a host that binds the library sees the C ABI and nothing else."""
    import subprocess
    path = os.environ.get("HPT_LIB") or os.path.join(ROOT, "pbrt-v2_amd", "libhpt.so")
    out = subprocess.run(["nm", "-D", "--defined-only", path], capture_output=True, text=True, check=True).stdout
    defined = sorted(l.split()[-1] for l in out.splitlines() if len(l.split()) >= 3 and l.split()[-2] in "TDBRWVi")
    assert defined == header_symbols(), sorted(set(defined) ^ set(header_symbols()))


def test_struct_layouts_match_compiled_library():
    sizes = (C.c_int32 * 10)()
    hpt.lib().hpt_abi_sizes(sizes)
    assert list(sizes) == abi.ABI_SIZES


def test_blob_roundtrip_between_c_and_python(cases, tmp_path):
    s = cases["env"]
    p = str(tmp_path / "x.hpts")
    d = s.desc
    rc = hpt.lib().hpt_blob_save(p.encode(), C.byref(d), C.byref(s.camera), C.byref(s.render))
    assert rc == 0, hpt.last_error()
    t = abi.Scene.load(p)
    assert np.array_equal(t.fpool, s.fpool) and np.array_equal(t.ipool, s.ipool)
    assert bytes(t.camera) == bytes(s.camera) and bytes(t.render) == bytes(s.render)
    assert bytes(t.meshes) == bytes(s.meshes) and bytes(t.lights) == bytes(s.lights)
    p2 = str(tmp_path / "y.hpts.gz")
    t.save(p2)
    u = abi.Scene.load(p2)
    assert np.array_equal(u.fpool, s.fpool) and bytes(u.materials) == bytes(s.materials)


@pytest.mark.parametrize("name", ["tex.hpts.gz", "metal.hpts.gz", "texmap.hpts.gz", "killeroo_cfg1.hpts.gz"])
def test_c_loader_upgrades_older_blobs_like_the_python_loader(name, tmp_path):
    """ABI 9: the texture records of version-6 .. 8 blobs grow by the 2D mapping (uv) in hpt_blob_load exactly as in abi.Scene.load; a version-9 blob
    (texmap: spherical / cylindrical / planar) passes through."""
    import gzip
    import os
    from tests.util import GOLDEN
    raw = gzip.open(os.path.join(GOLDEN, name), "rb").read()
    p = str(tmp_path / "x.hpts")
    open(p, "wb").write(raw)
    L = hpt.lib()
    L.hpt_blob_load.restype = C.c_void_p
    L.hpt_blob_scene.restype = C.POINTER(abi.SceneDesc)
    L.hpt_blob_scene.argtypes = [C.c_void_p]
    L.hpt_blob_free.argtypes = [C.c_void_p]
    b = L.hpt_blob_load(p.encode())
    assert b, hpt.last_error()
    try:
        d = L.hpt_blob_scene(b).contents
        s = abi.Scene.load(p)
        assert d.n_textures == len(s.textures) and d.n_meshes == len(s.meshes)
        if d.n_textures:
            assert C.string_at(d.textures, C.sizeof(abi.Texture) * d.n_textures) == bytes(s.textures)
        version = int(np.frombuffer(raw[4:8], dtype=np.uint32)[0])
        mapped = [t.mapping for t in s.textures if t.kind == abi.HPT_TEX_IMAGEMAP]
        assert (version == abi.HPT_VERSION and sorted(set(mapped)) == [1, 2, 3]) if name.startswith("texmap") else (version < 9 and not any(mapped))
    finally:
        L.hpt_blob_free(b)


def test_invalid_descriptors_are_rejected(cases, tmp_path):
    s = abi.Scene.load(os.path.join(ROOT, "tests", "golden", "env_soup.hpts.gz"))
    s.meshes[0].material = 99
    d = s.desc
    rc = hpt.lib().hpt_blob_save(str(tmp_path / "bad.hpts").encode(), C.byref(d), C.byref(s.camera), C.byref(s.render))
    assert rc == -2 and "mesh 0" in hpt.last_error()
    s = abi.Scene.load(os.path.join(ROOT, "tests", "golden", "env_soup.hpts.gz"))
    s.ipool[5] = 10 ** 6
    d = s.desc
    assert hpt.lib().hpt_blob_save(str(tmp_path / "bad.hpts").encode(), C.byref(d), None, None) == -2
    s = abi.Scene.load(os.path.join(ROOT, "tests", "golden", "env_soup.hpts.gz"))
    s.materials[0].kind = 77
    d = s.desc
    assert hpt.lib().hpt_blob_save(str(tmp_path / "bad.hpts").encode(), C.byref(d), None, None) == -3
    s = abi.Scene.load(os.path.join(ROOT, "tests", "golden", "texmap.hpts.gz"))       # ABI 9: a 2D mapping the library does not know
    next(t for t in s.textures if t.kind == abi.HPT_TEX_IMAGEMAP).mapping = 4
    d = s.desc
    assert hpt.lib().hpt_blob_save(str(tmp_path / "bad.hpts").encode(), C.byref(d), None, None) == -2 and "mapping" in hpt.last_error()


def test_texture_tables_nested_deeper_than_the_general_evaluators_stack_are_refused():
    """hpt_scene_create bounds the operand nesting of scale / mix textures (HPT_TEX_MAX_DEPTH 12, csrc/hpt_device.h) before it looks for a device: a table
    13 deep is refused by name; one 12 deep gets as far as the device check."""
    from tests.util import load_case, nest_textures
    s = load_case("tex")
    nest_textures(s, 10)                  # the wall's Kd was scale(imagemap, colour): 1 + 10 = 11 ... the deepest Kd of the scene decides
    with pytest.raises(hpt.HptError) as e:
        hpt.DeviceScene(s)
    deep_ok = "nested deeper" not in str(e.value)
    s = load_case("tex")
    nest_textures(s, 13)
    with pytest.raises(hpt.HptError) as e:
        hpt.DeviceScene(s)
    assert "nested deeper than 12" in str(e.value) and deep_ok


def test_unsampled_emitter_records_are_validated(tmp_path):
    """include/hpt.h, HPT_LIGHT_UNSAMPLED (the area light of a shape inside an object instance, core/api.cpp:1046-1049): such records stand behind Scene::lights in the
    table, and only instanced meshes may name them — a world mesh that emits belongs to a sampled light's shape set."""
    from tests.util import load_case
    save = lambda s: hpt.lib().hpt_blob_save(str(tmp_path / "x.hpts").encode(), C.byref(s.desc), None, None)
    s = load_case("oemit")
    assert [abi.light_unsampled(l) for l in s.lights] == [False, False, True, True] and save(s) == 0, hpt.last_error()
    more = abi._arr(abi.Light, 5)                                                                 # a sampled light (a copy of the point light) behind the unsampled records
    for i in range(5):
        more[i] = abi.copy_struct(s.lights[i if i < 4 else 0])
    s.lights = more
    assert save(s) == -2 and "precede" in hpt.last_error()
    s = load_case("oemit")
    next(m for m in s.meshes if m.instance < 0).arealight = 2                                     # a mesh of the world naming an unsampled emitter
    assert save(s) == -2 and ("not inside an instance" in hpt.last_error() or "unsampled" in hpt.last_error())


def test_animated_quadric_records_are_validated(tmp_path):
    """ABI 8, hpt_instance.quadric1 (an animated sphere / disk: TransformedPrimitive over a bare GeometricPrimitive, core/api.cpp:1032-1042):
    the record it names must exist, carry the identity ObjectToWorld and no area light (api.cpp:1014-1021), and belong to one instance;
    version-7 blobs — the same records with the field as padding — still load."""
    good = os.path.join(ROOT, "tests", "golden", "aquad.hpts.gz")
    save = lambda s: hpt.lib().hpt_blob_save(str(tmp_path / "q.hpts").encode(), C.byref(s.desc), None, None)
    s = abi.Scene.load(good)
    assert sorted(i.quadric1 for i in s.instances) == [0, 1, 2] and save(s) == 0, hpt.last_error()
    s = abi.Scene.load(good); s.instances[0].quadric1 = len(s.quadrics) + 1
    assert save(s) == -2 and "quadric1" in hpt.last_error()
    s = abi.Scene.load(good); s.quadrics[s.instances[0].quadric1 - 1].o2w[3] = 0.5
    assert save(s) == -2 and "identity" in hpt.last_error()
    s = abi.Scene.load(good); s.instances[1].quadric1 = s.instances[0].quadric1
    assert save(s) == -2 and "already belongs" in hpt.last_error()
    s = abi.Scene.load(good); s.instances[2].quadric1 = s.instances[0].quadric1; s.instances[0].quadric1 = 0
    assert save(s) == -2 and "both a quadric and mesh" in hpt.last_error()
    old = abi.Scene.load(os.path.join(ROOT, "tests", "golden", "acam.hpts.gz"))       # a version-7 blob with an (ordinary) animated instance
    assert len(old.instances) == 1 and old.instances[0].quadric1 == 0 and save(old) == 0


def test_shared_instance_records_are_validated(tmp_path):
    """ABI 8, object instancing (hpt_instance.quadric1 < 0: the instance shares the primitive of instance -quadric1 - 1): the owner is an earlier
    instance that owns a primitive itself (no chains), and a sharing instance owns no mesh."""
    good = os.path.join(ROOT, "tests", "golden", "oinst.hpts.gz")
    save = lambda s: hpt.lib().hpt_blob_save(str(tmp_path / "o.hpts").encode(), C.byref(s.desc), None, None)
    s = abi.Scene.load(good)
    assert [i.quadric1 for i in s.instances] == [0, -1, -1, -1, 0, -5] and save(s) == 0, hpt.last_error()
    s = abi.Scene.load(good); s.instances[1].quadric1 = -3            # owner 2 is itself a sharing instance
    assert save(s) == -2 and "earlier owner" in hpt.last_error()
    s = abi.Scene.load(good); s.instances[1].quadric1 = -6            # owner 5 comes later
    assert save(s) == -2 and "earlier owner" in hpt.last_error()
    s = abi.Scene.load(good)
    m = [i for i, me in enumerate(s.meshes) if me.instance == 4][0]; s.meshes[m].instance = 5
    assert save(s) == -2 and "owns mesh" in hpt.last_error()
    # a sphere instanced twice (ObjectInstance of a one-sphere object): refused, not rendered without it
    q = abi.Scene.load(os.path.join(ROOT, "tests", "golden", "aquad.hpts.gz"))
    owner = [k for k, i in enumerate(q.instances) if i.quadric1 > 0][0]
    later = [k for k in range(owner + 1, len(q.instances)) if q.instances[k].quadric1 > 0][0]
    q.instances[later].quadric1 = -(owner + 1)
    assert hpt.lib().hpt_blob_save(str(tmp_path / "q.hpts").encode(), C.byref(q.desc), None, None) == -3 and "only aggregates" in hpt.last_error()


def test_spot_light_records_are_validated(tmp_path):
    """ABI 8, HPT_LIGHT_SPOT: cosTotalWidth (`area`) <= cosFalloffStart (`marg_int`), both cosines; HPT_LIGHT_DISTANT needs nothing beyond its kind."""
    good = os.path.join(ROOT, "tests", "golden", "lts.hpts.gz")
    save = lambda s: hpt.lib().hpt_blob_save(str(tmp_path / "l.hpts").encode(), C.byref(s.desc), None, None)
    s = abi.Scene.load(good)
    assert sorted(l.kind for l in s.lights) == [abi.HPT_LIGHT_POINT, abi.HPT_LIGHT_SPOT, abi.HPT_LIGHT_SPOT, abi.HPT_LIGHT_DISTANT] and save(s) == 0
    spot = [i for i, l in enumerate(s.lights) if l.kind == abi.HPT_LIGHT_SPOT][0]
    s.lights[spot].area, s.lights[spot].marg_int = s.lights[spot].marg_int + 0.01, s.lights[spot].marg_int
    assert save(s) == -2 and "spot light" in hpt.last_error()
    s = abi.Scene.load(good); s.lights[spot].kind = 9
    assert save(s) in (-2, -3)


def test_warmup_validates_its_argument_before_any_runtime_work():
    """hpt_warmup (include/hpt.h): a negative device is refused on the calling thread — no thread is started, no HIP call made."""
    assert hpt.lib().hpt_warmup(-1) == -2
    assert "hpt_warmup" in hpt.last_error()


def test_no_cpu_fallback_without_device(cases):
    if hpt.device_count() > 0:
        pytest.skip("a HIP device is present")
    with pytest.raises(hpt.HptError, match="no HIP device"):
        hpt.DeviceScene(cases["env"])
    with pytest.raises(hpt.HptError):
        hpt.sampler(cases["env"].render, 0, 0)


def test_every_device_object_is_gated_and_the_extension_units_stay_off_the_greedy_allocator():
    """Round 5 (profiles/r05_ab.md, r05_isaemu_root_cause.md): under clang 22's greedy register allocator builds of the two 0.5-1 MB extension units had one kernel
    instantiation that was wrong as a whole — a live-range copy placed above an EXEC restore — and the units shipped on -vgpr-regalloc=basic (10-20 % slower code).
    Round 6: the defect's shape is scanned for in EVERY device object the Makefile compiles (an object that has it is deleted and the build fails) — and a greedy build of
    the extension units that PASSED that scan rendered bad samples on the GPU (profiles/r06_ab.md, run I): the scan covers one shape, the units stay on the basic allocator, and
    build() executes kernels of them in the interpreter.  The rules live in the Makefile, where a clean-up would lose them silently: this test is the note on the door."""
    mk = open(os.path.join(ROOT, "pbrt-v2_amd", "Makefile")).read()
    rules = re.findall(r"^build/[^\n:]*\.o: csrc/%\.hip[^\n]*\n((?:\t[^\n]*\n)+)", mk, flags=re.M)
    assert len(rules) >= 3 and all("$(gate)" in r for r in rules), rules          # libhpt.so's objects, `variant`, `flavor` (debug / shadow)
    assert re.search(r"^GATE \?= python3 \.\./scripts/check_exec_restore\.py$", mk, flags=re.M)
    for unit in ("hpt_kernels_ext", "hpt_kernels_ext_i"):     # (round 6: a greedy build of these units passed the static gate and was wrong on the GPU and in the interpreter — they stay on the basic allocator)
        m = re.search(r"^FLAGS_%s\s*:=(.*)$" % unit, mk, flags=re.M)
        assert m and "-vgpr-regalloc=basic" in m.group(1), unit
        for part in re.findall(r"hpt_kernels_%s_p\d" % unit[len("hpt_kernels_"):], mk):
            assert re.search(r"FLAGS_hpt_kernels_\$\(u\)_p\d := \$\$\(FLAGS_hpt_kernels_\$\(u\)\)", mk), part      # the parts inherit the unit's flags
    m = re.search(r"^FLAGS_hpt_kernels_lean\s*:=(.*)$", mk, flags=re.M)
    assert m and "iterative-ilp" not in m.group(1)      # (the scheduler whose output LLVM's machine verifier rejects and on which the allocator segfaults)
    entry = open(os.path.join(ROOT, "__graft_entry__.py")).read()
    assert "check_exec_restore" in entry and "isaemu_gate" in entry                # build(): the static gate again over the whole build + the bench kernels in the interpreter
