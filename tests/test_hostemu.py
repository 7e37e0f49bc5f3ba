"""CPU-side check of the DEVICE code against the oracle, without a GPU: tests/hostemu compiles
pbrt-v2_amd/csrc/hpt_device.h + hpt_path.h (the very headers the gfx950 kernels are built from)
with g++ and runs the per-lane state machine lane by lane over the same BVH2 / triangle records.
On the same machine (same libm, -ffp-contract=off) the results must match the oracle's LD_HASH
mode essentially bit for bit; the only legitimate differences come from the different BVH (exact
ties / box-edge grazes) and from float sums of the rare samples that spill into a neighbour pixel."""
import importlib

import numpy as np
import pytest

from tests.util import ADAPTIVE_CASES, BESTCANDIDATE_CASES, CASES, DL_CASES, FILTER_CASES, HALTON_CASES, RANDOM_CASES, STRATIFIED_CASES, abi, bsdf_inputs, hash_rd, load_case, load_ref, random_rays, sample_table

film = importlib.import_module("pbrt-v2_amd.film")
from oracle import orc
from tests.hostemu import emu


@pytest.fixture(scope="module")
def pairs(cases):
    return {n: (orc.OracleScene(cases[n]), emu.EmuScene(cases[n])) for n in CASES}


def test_sampler_bit_identical(cases):
    rd = hash_rd(cases["cfg1"], seed=11, spp=16)
    for (x, y) in [(0, 0), (3, 200), (255, 17)]:
        assert np.array_equal(orc.sampler(rd, x, y), emu.sampler(rd, x, y))


@pytest.mark.parametrize("name", ["cfg1", "b8", "env", "anim"])
def test_intersect_matches_oracle(cases, pairs, name):
    o, e = pairs[name]
    rays = random_rays(cases[name], 20000, seed=5)
    ho, po = o.intersect(rays)
    he, pe = e.intersect(rays)
    same = po == pe
    # the two BVHs differ: allow a handful of exact-tie / box-edge disagreements
    assert same.mean() > 0.9995, same.mean()
    assert (po >= 0).mean() > 0.2
    assert np.array_equal(ho[same], he[same])  # t, b1, b2, rayEpsilon bit-identical
    so, _ = o.intersect(rays, anyhit=True)
    se, _ = e.intersect(rays, anyhit=True)
    _, ao = o.intersect(rays, anyhit=True)
    _, ae = e.intersect(rays, anyhit=True)
    assert (ao == ae).mean() > 0.9995


@pytest.mark.parametrize("name", ["aquad", "oinst", "aquad+33"])
def test_instance_kinds_of_abi8_find_the_oracles_hits(name):
    """ABI 8's instance kinds under aggregatetest-style rays (time 0: the start transforms), closest hit and any hit, on the BVH2 walk (traverse) and on
    the four-wide walk the path kernel runs (trav_node4): `aquad` — a sphere and a disk that are instances' primitives (trav_begin<QI>: tested when
    the walk enters the instance, skipped among the world's quadrics) beside a mesh instance and a world sphere; `oinst` — six instances over two
    shared aggregates (inst_root / inst_root4 of a sharing instance are its owner's), one of them mirrored; `aquad+33` — aquad behind 33 more world spheres, so
    that the quadrics its instances own have indices past the 31 bits of DScene::inst_quadric_mask (the instance-table lookup of trav_begin)."""
    from tests.util import load_case, with_quadric_padding
    s = with_quadric_padding(load_case("aquad"), 33) if name == "aquad+33" else load_case(name)
    o, e = orc.OracleScene(s), emu.EmuScene(s, max_leaf=2)
    rays = random_rays(s, 40000, seed=9)
    ho, po = o.intersect(rays)
    _, ao = o.intersect(rays, anyhit=True)
    assert (po >= 0).mean() > 0.15
    he, pe = e.intersect(rays)
    same = po == pe
    assert same.mean() > 0.9995 and np.array_equal(ho[same][:, :3], he[same][:, :3])
    _, ae = e.intersect(rays, anyhit=True)
    assert (ao == ae).mean() > 0.9995
    for cap in (-1, 0):
        h4, p4, _ = e.intersect4(rays, cap=cap)
        same = po == p4
        assert same.mean() > 0.9995 and np.array_equal(ho[same][:, :3], h4[same][:, :3])
        _, a4, _ = e.intersect4(rays, anyhit=True, cap=cap)
        assert (ao == a4).mean() > 0.9995


@pytest.mark.parametrize("name,copies,time", [("anim", 0, 0.0), ("anim", 0, 0.37), ("aquad", 0, 0.0), ("aquad", 0, 0.6), ("oinst", 0, 0.0), ("oinst", 58, 0.45), ("b8", 0, 0.0), ("cfg1", 0, 0.0)])
def test_top_level_tree_walk_finds_the_linear_walks_hits(name, copies, time):
    """Round 4: the walk from the TOP-LEVEL tree (hpt_device.h traverse_top / top_special_leaf; the path kernel's
    traverse_steal runs the same steps with 64 lanes) — animated instances are leaves of a tree over their motion bounds, entered nearest
    first with the world ray parked on the stack, and the world's spheres / disks are primitives of the world's tree (pseudo-triangle
    records, trav_leaf) — against the reference order of the walk (traverse: world tree, then every instance in index order) on the same
    rays at the same time: same primitive, same instance, t / b1 / b2 bit-identical, any-hit answers equal; also with NO stack rows for
    ordinary entries (cap 0: every level a masked entry — the bound kernel_residency computes must hold with the parked ray's 7 rows).
    `oinst` + 58 copies: 64 instances (the top-level tree is a real tree, not one node)."""
    from tests.util import load_case, with_instance_copies
    s = load_case(name)
    if copies:
        s = with_instance_copies(s, 2, copies, start=(-12.0, 0.0, -6.0), step=(0.45, 0.02, 0.31))
    e = emu.EmuScene(s, max_leaf=2)
    rays = random_rays(s, 40000, seed=13)
    hl, pl = e.intersect_at(rays, time)
    _, al = e.intersect_at(rays, time, anyhit=True)
    assert (pl >= 0).mean() > 0.15
    info = e.info()
    for cap in (-1, 0):
        ht, pt, it, deepest = e.intersect_top(rays, cap=cap, time=time)
        same = pl == pt
        assert same.mean() > 0.999, same.mean()
        assert np.array_equal(hl[same][:, :3], ht[same][:, :3])
        # where the walks name different primitives the hit DISTANCE is the same: coplanar faces of overlapping objects (the order of the walk
        # decides an exact tie; a mesh seen through two instances' transforms ties to an ulp)
        assert np.allclose(hl[~same][:, 0], ht[~same][:, 0], rtol=1e-6, atol=0) and ((pl[~same] >= 0) & (pt[~same] >= 0)).all()
        _, at, _, _ = e.intersect_top(rays, anyhit=True, cap=cap, time=time)
        assert (al == at).mean() > 0.9995
        if cap == 0:
            assert deepest <= 2 + info["top_depth4"], (deepest, info)
        else:
            assert deepest <= info["top_stack_bound4"] + 1, (deepest, info)
    if len(s.instances):
        assert (it >= 0).any()                                # some hits lie inside instances
    if copies:
        assert info["n_nodes4_top"] >= 5                      # 64 instances: several levels above the instances


@pytest.mark.parametrize("name", ["metal", "metalg", "tex", "on", "alpha", "tang", "qtex", "lens"])
def test_lean_extension_set_renders_the_full_sets_film(name, monkeypatch):
    """The lean extension set (MATS_LEAN, csrc/hpt_kernels_lean.hip: the extension kernels without specular lobes / the direct-lighting recursion,
    the regular half-angle BRDF, shape-set area lights, spot / distant lights and the measured BRDF — opt-in on the device, HPT_LEAN_EXT=1) on every
    fixture that reaches none of those: the same film as the full set, bit for bit, and the oracle's rays."""
    from tests.util import load_case
    s = load_case(name)
    rd = hash_rd(s, seed=3)
    e = emu.EmuScene(s)
    full, sf = e.render(s.camera, rd)
    monkeypatch.setenv("HPT_EMU_LEAN", "1")
    lean, sl = e.render(s.camera, rd)
    monkeypatch.delenv("HPT_EMU_LEAN")
    fo, so = orc.OracleScene(s).render(s.camera, rd)
    assert np.array_equal(full, lean) and list(sf[:3]) == list(sl[:3])
    assert sl[0] == so[0] and abs(int(sl[1]) - int(so[1])) <= 4 and abs(int(sl[2]) - int(so[2])) <= 4
    assert film.rmse(film.xyzw_to_rgb(lean), film.xyzw_to_rgb(fo)) < 1e-6


def test_bvh_depth_is_bounded(pairs):
    for n, (_, e) in pairs.items():
        info = e.info()
        assert info["max_depth"] <= 25 and info["n_nodes"] > 0


# Animated spheres / disks under textures and bump maps: the device inverts the interpolated transform analytically (hpt_device.h, anim_interpolate:
# equal to the reference's Gauss-Jordan inverse up to rounding), and dpdu / dpdv / dndu / dndv go through that inverse into the bump frame and the
# EWA footprint, which carry a last-bit difference to ~1e-5 of a pixel's value (a few pixels per frame: 1e-4 .. 3e-3, where one of the 8 samples takes a discrete decision — a filter level, a grazing shadow ray — the other way).  Same rays, same film weights; bounded instead of bit-compared.
INVERSE_ROUNDING_CASES = ("aquad", "aquaddl", "abi8dl")


def close_enough(name, io, ie):
    if name in INVERSE_ROUNDING_CASES:
        d = np.abs(io - ie).max(axis=2)
        return differing_pixels(io, ie) < 1e-2 and float((d / np.maximum(np.abs(io).max(axis=2), 1e-3)).max()) < 1e-2 and film.rmse(io, ie) < 5e-6
    return differing_pixels(io, ie) < 1e-3 and film.rmse(io, ie) < 1e-6


def differing_pixels(io, ie):
    """Share of pixels where the emulated device image differs from the oracle's.  Bit for bit — except that the value of a measured
    BRDF (bunny scenes) agrees with the reference's to the rounding of a short sum (grid order instead of kd-tree order, hpt_device.h),
    which moves a pixel by a few ulps: those count as equal."""
    return (np.abs(io - ie).max(axis=2) > 2e-6 * np.maximum(np.abs(io).max(axis=2), 1e-3)).mean()


@pytest.mark.parametrize("name,material", [("cfg1", 1), ("cfg1", 2), ("cfg1", 3), ("b8", 1), ("env", 0), ("ms", 0), ("ms", 1), ("ms", 2)])
def test_bsdf_bit_identical(cases, pairs, name, material):
    o, e = pairs[name]
    inp = bsdf_inputs(4000 if name != "b8" else 800)
    a, b = o.bsdf(material, inp), e.bsdf(material, inp)
    assert np.isfinite(a[:, :4]).all()
    if name == "b8":
        # measured BRDF (IrregIsotropicBRDF::f): the device finds the reference's final radius and the samples inside it through
        # a uniform grid instead of the kd-tree and sums them in grid order — the same terms in another order: equal to float
        # rounding of a sum of a handful of products (and bit-identical wherever a single order is the only one possible)
        assert np.allclose(a, b, rtol=1e-5, atol=1e-9, equal_nan=True), np.abs(a - b).max()
        assert (a == b).all(axis=1).mean() > 0.15
        return
    assert np.array_equal(a, b, equal_nan=True)


def test_measured_brdf_grid_finds_the_reference_radius(pairs):
    """Queries all over the measured BRDF's domain, including the sparse corners where the radius grows many times and the
    dense specular region: the grid walk must use exactly the samples the kd-tree query of the reference uses."""
    o, e = pairs["b8"]
    inp = bsdf_inputs(6000, seed=11)
    a, b = o.bsdf(1, inp), e.bsdf(1, inp)
    assert np.allclose(a[:, :3], b[:, :3], rtol=1e-5, atol=1e-9), np.abs(a - b).max()


@pytest.mark.parametrize("name", list(DL_CASES))
def test_direct_lighting_render_matches_oracle(name):
    """DirectLightingIntegrator (strategy all / one; measured BRDF; animated instances): the device lane, emulated
    on the host, against the oracle at the production sampler — same rays, same film up to BVH-dependent ties."""
    s = load_case(name)
    o, e = orc.OracleScene(s), emu.EmuScene(s)
    rd = hash_rd(s, seed=5)
    fo, so = o.render(s.camera, rd)
    fe, se = e.render(s.camera, rd)
    assert so[0] == se[0] == rd.x_count * rd.y_count * rd.spp
    assert abs(int(so[1]) - int(se[1])) <= 4 and abs(int(so[2]) - int(se[2])) <= 4   # closest / shadow rays
    io, ie = film.xyzw_to_rgb(fo), film.xyzw_to_rgb(fe)
    assert differing_pixels(io, ie) < 1e-3
    assert film.rmse(io, ie) < 1e-6


@pytest.mark.parametrize("name", ["on", "spec", "specdl", "trilight", "trildl", "merl", "tex", "mirtex", "alpha", "metal", "lens", "metalg", "tang", "qtex", "aquad", "aquaddl", "lts", "ltsdl", "oinst", "abi8dl", "texmap", "texmapdl", "texdeep", "oemit", "oemitdl"])
def test_round2_features_render_matches_oracle(name):
    """The MATS_EXT device code (Oren-Nayar, glass / mirror with the path integrator's specular bounces, triangle-mesh emitters,
    RegularHalfangleBRDF, image textures with EWA / trilinear lookups + ray differentials + Material::Bump, alpha-textured triangles,
    scenes/metal.pbrt as shipped), emulated on the host, against the oracle at the production sampler: same rays, same film."""
    from tests.util import load_case
    s = load_case(name)
    o, e = orc.OracleScene(s), emu.EmuScene(s)
    rd = hash_rd(s, seed=3)
    fo, so = o.render(s.camera, rd)
    fe, se = e.render(s.camera, rd)
    assert so[0] == se[0] == rd.x_count * rd.y_count * rd.spp
    assert abs(int(so[1]) - int(se[1])) <= 4 and abs(int(so[2]) - int(se[2])) <= 4
    assert np.array_equal(fo[..., 3], fe[..., 3])
    io, ie = film.xyzw_to_rgb(fo), film.xyzw_to_rgb(fe)
    assert close_enough(name, io, ie)     # (specdl / mirtex: the recursion's products are taken in another order)


@pytest.mark.parametrize("name,material", [("on", 1), ("on", 2), ("spec", 1), ("spec", 2), ("spec", 4), ("merl", 0), ("tex", 2), ("tex", 3)])
def test_round2_bsdfs_bit_identical(name, material):
    """Oren-Nayar, glass (two specular lobes: pdf 1/2, Fresnel-weighted values), mirror, RegularHalfangleBRDF, and materials whose
    parameters are textures (looked up at (u, v) = (u1, u2) of the input row): device functions compiled for the host vs the oracle."""
    from tests.util import load_case
    s = load_case(name)
    o, e = orc.OracleScene(s), emu.EmuScene(s)
    inp = bsdf_inputs(3000, seed=17)
    a, b = o.bsdf(material, inp), e.bsdf(material, inp)
    assert np.array_equal(a, b, equal_nan=True), np.abs(a - b).max()
    if name == "spec":
        assert (a[:, 11] != 0).any() and (a[:, 10] > 0).any()      # specular lobes are sampled


@pytest.mark.parametrize("name", CASES)
def test_render_matches_oracle(cases, pairs, name):
    s = cases[name]
    o, e = pairs[name]
    rd = hash_rd(s, seed=5)
    fo, so = o.render(s.camera, rd)
    fe, se = e.render(s.camera, rd)
    assert so[0] == se[0] == rd.x_count * rd.y_count * rd.spp
    io, ie = film.xyzw_to_rgb(fo), film.xyzw_to_rgb(fe)
    differing = differing_pixels(io, ie)
    assert differing < 1e-3, differing
    assert film.rmse(io, ie) < 1e-4
    assert np.array_equal(fo[..., 3], fe[..., 3])          # weights: same samples in same pixels
    assert abs(int(so[1]) - int(se[1])) <= 4 and abs(int(so[2]) - int(se[2])) <= 4  # ray counts


def test_shards_partition_the_image(cases, pairs):
    s = cases["k8"]
    _, e = pairs["k8"]
    rd = hash_rd(s, seed=2, spp=2)
    full, _ = e.render(s.camera, rd)
    for count in (2, 3):
        acc = np.zeros_like(full)
        wsum = np.zeros(full.shape[:2])
        for r in range(count):
            rd.shard_rank, rd.shard_count = r, count
            f, _ = e.render(s.camera, rd)
            own = f[..., 3] >= rd.spp
            wsum += own
            acc += f
        assert wsum.max() == 1 and wsum.min() == 1          # every pixel owned by exactly one shard
        assert np.allclose(acc, full, rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("name", CASES)
def test_replay_mode_reproduces_reference_image(cases, pairs, name):
    """The device state machine + MT_REPLAY sampler source (hpt_replay.h), emulated on the CPU,
    against the image the REFERENCE BINARY rendered (golden fixture).  Same libm, same stream:
    identical except where the different BVH resolves an exact tie / box-edge graze differently."""
    s = cases[name]
    _, e = pairs[name]
    rd = abi.copy_struct(s.render)
    rd.sampler_mode = abi.HPT_SAMPLER_MT_REPLAY
    f, st = e.render(s.camera, rd)
    img, ref = film.xyzw_to_rgb(f), load_ref(name)
    assert st[0] == rd.x_count * rd.y_count * rd.spp
    differing = differing_pixels(img, ref)
    assert differing < 1e-3, differing
    assert film.rmse(img, ref) < 1e-3


@pytest.mark.parametrize("name", ["cfg1", "b8", "env", "anim"])
def test_four_wide_trees_find_the_same_hits(cases, name):
    """Round 3: the BVH4 (collapse_bvh4, csrc/hpt_bvh.cpp) and its node step (trav_node4, csrc/hpt_device.h) — what the path kernel's
    lock-step + stealing walk runs on — against the oracle's BVHAccel::Intersect / IntersectP on aggregatetest-style rays: same primitive,
    t / barycentrics bit-identical; with every stack row taking ordinary entries, with three, and with none (one masked entry per level:
    the bounded-stack slow path), where the stack never holds more than one entry per level."""
    s = cases[name]
    e = emu.EmuScene(s, max_leaf=2)
    info = e.info()
    assert info["n_nodes4"] > 0 and info["depth4"] < info["max_depth"] and info["stack_bound4"] <= 3 * info["depth4"]
    rays = random_rays(s, 60000, seed=5)
    o = orc.OracleScene(s)
    ho, po = o.intersect(rays)
    _, ao = o.intersect(rays, anyhit=True)
    for cap in (-1, 3, 0):
        he, pe, deepest = e.intersect4(rays, cap=cap)
        same = po == pe
        assert same.mean() > 0.9995 and np.array_equal(ho[same][:, :3], he[same][:, :3])
        _, ae, _ = e.intersect4(rays, anyhit=True, cap=cap)
        assert (ao == ae).mean() > 0.9995
        assert deepest <= (info["stack_bound4"] if cap < 0 else cap + 2 + info["depth4"])
        if cap == 0:
            assert deepest <= info["depth4"]


@pytest.mark.parametrize("name", ["b8", "anim"])
def test_forked_tree_build_equals_the_serial_build(cases, name, monkeypatch):
    """Round 3: the host SAH builder (csrc/hpt_bvh.cpp) builds the subtrees of its top four levels on threads of their own and splices them
    back in depth-first order.  The result must be the array the serial build produces, byte for byte (BVH2 nodes, the collapsed BVH4, the
    leaf-ordered triangle records) — HPT_BVH_THREADS pins the thread count the fork depth is derived from."""
    hashes = {}
    for threads in ("1", "2", "16"):
        monkeypatch.setenv("HPT_BVH_THREADS", threads)
        e = emu.EmuScene(cases[name], max_leaf=2)
        info = e.info()
        hashes[threads] = (info["tree_hash"], info["n_nodes"], info["max_depth"])
        e.close()
    assert hashes["1"] == hashes["2"] == hashes["16"], hashes


@pytest.mark.parametrize("name", ["on", "spec", "trilight", "merl", "tex", "alpha", "metal", "metalg", "lens", "tang", "qtex", "aquad", "lts", "oinst", "texmap", "texdeep", "oemit"])
def test_replay_mode_reproduces_reference_images_of_the_extension_set(name):
    """Round 3: the MT_REPLAY sampler source over the FULL material set (Lane<MtReplaySrc, true, MATS_FULL>) — Oren-Nayar, glass / mirror,
    triangle-mesh emitters, the regular half-angle BRDF, EWA / trilinear image textures with camera-ray differentials, bump mapping, alpha
    cut-outs, the thin lens, metal.pbrt as shipped under both environment maps — against the images the REFERENCE BINARY wrote for the same
    scene files: the device state machine consumes the reference's random stream draw for draw on these paths too, so round-2 features are
    compared with the reference directly, not only through the oracle's two sampler modes."""
    from tests.util import load_case
    s = load_case(name)
    rd = abi.copy_struct(s.render)
    assert rd.integrator == abi.HPT_INTEGRATOR_PATH
    rd.sampler_mode = abi.HPT_SAMPLER_MT_REPLAY
    f, st = emu.EmuScene(s).render(s.camera, rd)
    img, ref = film.xyzw_to_rgb(f), load_ref(name)
    assert st[0] == rd.x_count * rd.y_count * rd.spp and st[5] == 0
    assert close_enough(name, ref, img)


def test_moving_camera_matches_the_reference_and_the_oracle():
    """The device's camera_ray / camera_ray_differentials with an AnimatedTransform CameraToWorld: in MT_REPLAY mode against the image the
    reference binary wrote, with the production sampler against the oracle."""
    from tests.util import load_case
    s = load_case("acam")
    rd = abi.copy_struct(s.render)
    rd.sampler_mode = abi.HPT_SAMPLER_MT_REPLAY
    e = emu.EmuScene(s)
    f, st = e.render(s.camera, rd, cam_motion=s.camera_motion)
    img, ref = film.xyzw_to_rgb(f), load_ref("acam")
    assert differing_pixels(img, ref) < 1e-3 and film.rmse(img, ref) < 1e-6
    rd = hash_rd(s, seed=3)
    fo, _ = orc.OracleScene(s).render(s.camera, rd, cam_motion=s.camera_motion)
    fe, _ = e.render(s.camera, rd, cam_motion=s.camera_motion)
    assert np.array_equal(fo[..., 3], fe[..., 3]) and film.rmse(film.xyzw_to_rgb(fo), film.xyzw_to_rgb(fe)) < 1e-6


def test_sample_chunking_above_64_spp(cases, pairs):
    """spp > 64 splits a pixel into several work items (64 samples each) that flush separately:
    same samples, sums regrouped."""
    s = cases["env"]
    o, e = pairs["env"]
    rd = hash_rd(s, seed=9, spp=256)
    rd.x_count, rd.y_count, rd.xres, rd.yres = 40, 24, s.render.xres, s.render.yres
    fo, so = o.render(s.camera, rd)
    fe, se = e.render(s.camera, rd)
    assert so[0] == se[0] == 40 * 24 * 256
    assert np.array_equal(fo[..., 3], fe[..., 3])
    assert np.allclose(fo, fe, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("name", list(FILTER_CASES))
def test_filtered_render_matches_oracle(name):
    """the device's table splat (film_splat_table in hpt_path.h) and its sample extent against the
    oracle's ImageFilm::AddSample, on the cases the oracle is pinned to the reference binary with.  Same samples, same
    weights; only the order of the float sums differs (one lane per pixel here, raster order there)."""
    s = load_case(name)
    o, e = orc.OracleScene(s), emu.EmuScene(s)
    rd = hash_rd(s, seed=5)
    fo, so = o.render(s.camera, rd, flt=s.filter)
    fe, se = e.render(s.camera, rd, flt=s.filter)
    xs, xe, ys, ye = abi.sample_extent(rd, s.filter)
    assert so[0] == se[0] == (xe - xs) * (ye - ys) * rd.spp
    assert abs(int(so[1]) - int(se[1])) <= 4 and abs(int(so[2]) - int(se[2])) <= 4
    np.testing.assert_allclose(fe[..., 3], fo[..., 3], rtol=2e-5, atol=1e-5)     # weight sums
    scale = float(np.abs(fo[..., :3]).max())
    close = (np.abs(fe - fo)[..., :3].max(axis=2) <= 2e-5 * scale).mean()
    assert close > 0.999, close
    assert film.rmse(film.xyzw_to_rgb(fo), film.xyzw_to_rgb(fe)) < 1e-4


def test_filtered_replay_reproduces_the_reference_image():
    """MT_REPLAY through the device lane + table splat against the image the reference binary wrote with
    PixelFilter "gaussian": tiles are cut from the SAMPLE extent (Sampler::ComputeSubWindow over
    ImageFilm::GetSampleExtent), so every tile's generator sees the reference's pixels."""
    s = load_case("fgauss")
    e = emu.EmuScene(s)
    rd = abi.copy_struct(s.render)
    rd.sampler_mode = abi.HPT_SAMPLER_MT_REPLAY
    f, st = e.render(s.camera, rd, flt=s.filter)
    img, ref = film.xyzw_to_rgb(f), load_ref("fgauss")
    assert film.rmse(img, ref) < 1e-3
    assert np.isclose(img, ref, rtol=1e-4, atol=1e-6).all(axis=2).mean() > 0.99


def test_box_filter_through_the_table_equals_the_fast_path(cases, pairs):
    """A box of width 0.5 given as a table (all ones) takes the generic splat; the default takes the fast path that
    keeps a pixel's sum in registers.  Same weights, same pixels; sums regrouped."""
    s = cases["k8"]
    _, e = pairs["k8"]
    rd = hash_rd(s, seed=4, spp=4)
    f0, s0 = e.render(s.camera, rd)
    f1, s1 = e.render(s.camera, rd, flt=abi.make_filter("box"))
    assert s0[0] == s1[0]
    assert np.array_equal(f0[..., 3], f1[..., 3])
    assert np.allclose(f0, f1, rtol=1e-6, atol=1e-7)


def test_wide_filter_shards_sum_to_the_frame():
    """Multi-GPU under a wide filter: tiles of the SAMPLE extent are dealt round-robin, every shard's film covers the
    whole frame with partial sums, and the frame is their sum (a reduce instead of the box filter's gather)."""
    s = load_case("fmitch")
    e = emu.EmuScene(s)
    rd = hash_rd(s, seed=3)
    full, st = e.render(s.camera, rd, flt=s.filter)
    for count in (2, 3):
        acc = np.zeros_like(full)
        n = 0
        for r in range(count):
            rd.shard_rank, rd.shard_count = r, count
            f, sr = e.render(s.camera, rd, flt=s.filter)
            acc += f
            n += int(sr[0])
        assert n == int(st[0])
        assert np.allclose(acc, full, rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("name", list(FILTER_CASES))
def test_two_pass_film_matches_oracle_and_is_reproducible(name):
    """The device's default film under a table filter: the lane parks {X, Y, Z, 1} + {imageX, imageY} per camera sample,
    film_gather_pixel (hpt_path.h) then sums every film pixel's samples in a fixed order.  Same film as the oracle's
    AddSample to float rounding, bit-identical run to run, and shards sum to the frame."""
    s = load_case(name)
    o, e = orc.OracleScene(s), emu.EmuScene(s)
    rd = hash_rd(s, seed=5)
    fo, so = o.render(s.camera, rd, flt=s.filter)
    fg, sg = e.render(s.camera, rd, flt=s.filter, two_pass=True)
    fg2, _ = e.render(s.camera, rd, flt=s.filter, two_pass=True)
    assert sg[0] == so[0]
    assert np.array_equal(fg, fg2)
    np.testing.assert_allclose(fg[..., 3], fo[..., 3], rtol=2e-5, atol=1e-5)
    scale = float(np.abs(fo[..., :3]).max())
    assert (np.abs(fg - fo)[..., :3].max(axis=2) <= 2e-5 * scale).mean() > 0.999
    acc = np.zeros_like(fg)
    for r in range(3):
        rd.shard_rank, rd.shard_count = r, 3
        f, _ = e.render(s.camera, rd, flt=s.filter, two_pass=True)
        acc += f
    assert np.allclose(acc, fg, rtol=1e-5, atol=2e-5 * scale)


@pytest.mark.parametrize("name", list(RANDOM_CASES))
def test_random_sampler_render_matches_oracle(name):
    """SURVEY.md §8f-4, `Sampler "random"` as HPT_SAMPLER_RANDOM_HASH: the device lane's sampler getters in their
    independent-uniform mode (hpt_device.h LdHash::rnd), spp that are not powers of two (6, 3, 5), light sample counts that
    are not rounded (5) — against the oracle, which is pinned to the reference binary on these scenes in its replay mode."""
    s = load_case(name)
    o, e = orc.OracleScene(s), emu.EmuScene(s)
    rd = abi.copy_struct(s.render)
    assert rd.sampler_mode == abi.HPT_SAMPLER_RANDOM_HASH
    rd.seed = 5
    fo, so = o.render(s.camera, rd)
    fe, se = e.render(s.camera, rd)
    assert so[0] == se[0] == rd.x_count * rd.y_count * rd.spp
    assert abs(int(so[1]) - int(se[1])) <= 4 and abs(int(so[2]) - int(se[2])) <= 4
    assert np.array_equal(fo[..., 3], fe[..., 3])
    io, ie = film.xyzw_to_rgb(fo), film.xyzw_to_rgb(fe)
    assert differing_pixels(io, ie) < 1e-3 and film.rmse(io, ie) < 1e-4


def test_random_sampler_values_and_chunking(cases):
    """Bit-identical sample values (oracle vs device getters); uniform in [0, 1) per dimension; 100 spp = a full chunk of
    64 and a short one of 36."""
    s = load_case("rk")
    rd = abi.copy_struct(s.render)
    rd.seed, rd.spp = 11, 64
    a, b = orc.sampler(rd, 3, 7), emu.sampler(rd, 3, 7)
    assert np.array_equal(a, b)
    v = a[:, 5:]
    assert ((v >= 0) & (v < 1)).all() and 0.4 < float(v.mean()) < 0.6
    assert np.all((a[:, 0] >= 3) & (a[:, 0] < 4) & (a[:, 1] >= 7) & (a[:, 1] < 8))
    rd = abi.copy_struct(s.render)
    rd.spp, rd.x_start, rd.y_start, rd.x_count, rd.y_count = 100, 40, 40, 16, 16
    fo, so = orc.OracleScene(s).render(s.camera, rd)
    fe, se = emu.EmuScene(s).render(s.camera, rd)
    assert so[0] == se[0] == 16 * 16 * 100
    assert np.array_equal(fo[..., 3], fe[..., 3]) and np.allclose(fo, fe, rtol=1e-5, atol=1e-4)


@pytest.mark.parametrize("name", list(STRATIFIED_CASES))
def test_stratified_sampler_render_matches_oracle(name):
    """SURVEY.md §8f-4, `Sampler "stratified"` as HPT_SAMPLER_STRATIFIED_HASH: 3 x 2 jittered (path), 2 x 2 with a Latin hypercube
    over 5 light samples (direct lighting), 2 x 3 unjittered on the animated scene — the device lane's getters against the oracle,
    which is pinned to the reference binary on these scenes in its replay mode."""
    s = load_case(name)
    o, e = orc.OracleScene(s), emu.EmuScene(s)
    rd = abi.copy_struct(s.render)
    assert abi.sampler_kind(rd.sampler_mode) == abi.HPT_SAMPLER_STRATIFIED_HASH
    rd.seed = 5
    fo, so = o.render(s.camera, rd)
    fe, se = e.render(s.camera, rd)
    assert so[0] == se[0] == rd.x_count * rd.y_count * rd.spp
    assert abs(int(so[1]) - int(se[1])) <= 4 and abs(int(so[2]) - int(se[2])) <= 4
    assert np.array_equal(fo[..., 3], fe[..., 3])
    io, ie = film.xyzw_to_rgb(fo), film.xyzw_to_rgb(fe)
    assert differing_pixels(io, ie) < 1e-3 and film.rmse(io, ie) < 1e-4


@pytest.mark.parametrize("name", list(HALTON_CASES))
def test_halton_sampler_render_matches_oracle(name):
    """SURVEY.md §8f-4's tail, `Sampler "halton"` as HPT_SAMPLER_HALTON_HASH: work items that are sample numbers of a 32 x 32 window
    (item_to_halton), the double-precision radical inverses of the device headers, rejection at the sample extent's edge (96 = 3 x 32:
    none; 64 + the gaussian filter's margin and 100 x 60: partial windows), the time sample, the Latin hypercube over 5 light samples —
    the device lane against the oracle, which is pinned to the reference binary on these scenes in its replay mode."""
    s = load_case(name)
    flt = getattr(s, "filter", None)
    o, e = orc.OracleScene(s), emu.EmuScene(s)
    rd = abi.copy_struct(s.render)
    assert rd.sampler_mode == abi.HPT_SAMPLER_HALTON_HASH
    rd.seed = 5
    fo, so = o.render(s.camera, rd, flt=flt)
    fe, se = e.render(s.camera, rd, flt=flt)
    xs, xe, ys, ye = abi.sample_extent(rd, flt) if flt is not None else (rd.x_start, rd.x_start + rd.x_count, rd.y_start, rd.y_start + rd.y_count)
    assert so[0] == se[0] and 0.9 * (xe - xs) * (ye - ys) * rd.spp <= so[0] <= 1.1 * (xe - xs) * (ye - ys) * rd.spp
    if (xe - xs) % 32 == 0 and (ye - ys) % 32 == 0:
        assert so[0] == (xe - xs) * (ye - ys) * rd.spp           # whole windows: every sample number lands inside
    assert abs(int(so[1]) - int(se[1])) <= 4 and abs(int(so[2]) - int(se[2])) <= 4
    if flt is None:
        assert np.array_equal(fo[..., 3], fe[..., 3])            # box filter: the weights are the per-pixel sample counts
        assert fo[..., 3].min() >= 0 and abs(float(fo[..., 3].mean()) - rd.spp) < 0.1 * rd.spp and fo[..., 3].std() > 0   # ... which vary from pixel to pixel
    io, ie = film.xyzw_to_rgb(fo), film.xyzw_to_rgb(fe)
    assert differing_pixels(io, ie) < 1e-3 and film.rmse(io, ie) < 1e-4


def test_halton_sampler_shards_partition_the_windows():
    """Round-robin super-tile shards under Sampler "halton": a window's samples stay in its own 32 x 32 pixels, so the shard films add up
    to the unsharded film exactly (weights) / to rounding (sums in another order)."""
    s = load_case("hk")
    rd = abi.copy_struct(s.render)
    rd.seed = 9
    e = emu.EmuScene(s)
    full, st = e.render(s.camera, rd)
    acc, n = np.zeros_like(full), 0
    for r in range(3):
        rd.shard_rank, rd.shard_count = r, 3
        f, sr = e.render(s.camera, rd)
        acc += f; n += int(sr[0])
    assert n == int(st[0]) and np.array_equal(acc[..., 3], full[..., 3]) and np.allclose(acc, full, rtol=1e-5, atol=1e-5)


def test_halton_crop_window_renders_the_full_frames_samples():
    """The windows of HPT_SAMPLER_HALTON_HASH are cells of the GLOBAL 32 x 32 raster grid: a crop window off that grid (and one reaching into
    negative raster coordinates under a filter's margin) renders exactly the samples of the full frame that fall inside it — oracle and device
    lane alike (what lets bench.py verify a timed frame on crops)."""
    s = load_case("hk")
    rd = abi.copy_struct(s.render)
    rd.seed = 2
    o, e = orc.OracleScene(s), emu.EmuScene(s)
    full, _ = o.render(s.camera, rd)
    crd = abi.copy_struct(rd)
    crd.x_start, crd.y_start, crd.x_count, crd.y_count = 21, 38, 50, 37
    for r in (o, e):
        f, st = r.render(s.camera, crd)
        ref = full[38:38 + 37, 21:21 + 50]
        assert st[0] <= ref[..., 3].sum() <= 1.03 * st[0]     # (a Halton point on an integer raster coordinate counts in two pixels, film/image.cpp:82-89)
        # (the last column / row aside: a point exactly on the crop's upper edge is outside the crop's extent, yet reaches the edge pixel in the full frame)
        assert np.array_equal(f[:-1, :-1, 3], ref[:-1, :-1, 3]) and np.allclose(f[:-1, :-1], ref[:-1, :-1], rtol=1e-5, atol=1e-6)
        assert (f[..., 3] <= ref[..., 3]).all()
    # a wide filter: the sample extent starts at -2 — cells with negative grid coordinates
    flt = abi.make_filter("gaussian")
    fo, so = o.render(s.camera, rd, flt=flt)
    fe, se = e.render(s.camera, rd, flt=flt)
    assert so[0] == se[0] and np.allclose(fo, fe, rtol=1e-4, atol=1e-5)
    crd.x_start, crd.y_start, crd.x_count, crd.y_count = 40, 40, 16, 16
    fc, _ = o.render(s.camera, crd, flt=flt)
    assert np.allclose(fc, fo[40:56, 40:56], rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("name", list(ADAPTIVE_CASES))
def test_adaptive_sampler_render_matches_oracle(name):
    """SURVEY.md §8f-4's tail, `Sampler "adaptive"` (method contrast) as HPT_SAMPLER_ADAPTIVE_HASH: the device lane parks a pixel's first
    batch, takes ReportResults' decision from the batch's luminances in the reference's float order, drops the batch and starts the pixel
    again with maxSamples or sends it to the film in sample order — against the oracle, which is pinned to the reference binary on these
    scenes in its replay mode.  The weights say which pixels were supersampled: they must agree exactly."""
    s = load_case(name)
    o, e = orc.OracleScene(s), emu.EmuScene(s)
    rd = abi.copy_struct(s.render)
    assert abi.sampler_kind(rd.sampler_mode) == abi.HPT_SAMPLER_ADAPTIVE_HASH
    lo = (rd.sampler_mode >> 8) & 0xfff
    rd.seed = 5
    fo, so = o.render(s.camera, rd)
    fe, se = e.render(s.camera, rd)
    w = fo[..., 3]
    assert (w == lo).sum() > 0 and (w == rd.spp).sum() > 0
    differ = (fo[..., 3] != fe[..., 3]).mean()
    assert differ < 2e-3          # a decision that sits on the threshold within float rounding of the two sides' radiances may fall the other way
    assert abs(int(so[0]) - int(se[0])) <= 2e-3 * so[0] and so[5] == se[5] == 0
    io, ie = film.xyzw_to_rgb(fo), film.xyzw_to_rgb(fe)
    assert differing_pixels(io, ie) < 3e-3 and film.rmse(io, ie) < 2e-3


def test_adaptive_sampler_under_a_table_filter_and_in_shards():
    """Under a 2 x 2 gaussian filter (atomic splat and the two-pass film: a dropped batch leaves no record) and in three shards."""
    s = load_case("ak")
    rd = abi.copy_struct(s.render)
    rd.seed = 7
    flt = abi.make_filter("gaussian")
    o, e = orc.OracleScene(s), emu.EmuScene(s)
    fo, so = o.render(s.camera, rd, flt=flt)
    for two_pass in (False, True):
        fe, se = e.render(s.camera, rd, flt=flt, two_pass=two_pass)
        assert abs(int(so[0]) - int(se[0])) <= 2e-3 * so[0]
        assert film.rmse(film.xyzw_to_rgb(fo), film.xyzw_to_rgb(fe)) < 2e-3
    full, st = e.render(s.camera, rd)
    acc = np.zeros_like(full)
    for r in range(3):
        rd.shard_rank, rd.shard_count = r, 3
        f, _ = e.render(s.camera, rd)
        acc += f
    assert np.array_equal(acc[..., 3], full[..., 3]) and np.allclose(acc, full, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("name", list(BESTCANDIDATE_CASES))
def test_bestcandidate_sampler_render_matches_oracle(name):
    """SURVEY.md §8f-4's tail, `Sampler "bestcandidate"` as HPT_SAMPLER_BESTCANDIDATE_HASH: the work items are the entries of the reference's
    sample table in the table tiles of the render (item_to_bc), the tiles' shifts come from the host-side MT19937 of hpt_bc.h, the camera
    getters read table and shifts through the tag the sampler source carries, the arrays are LD_HASH's for one pixel sample under the tile's
    key — the device lane against the oracle (pinned to the reference binary on these scenes in its replay mode).  The camera samples of
    the production mode are EXACTLY the reference's: the film weights of oracle-hash, device and reference image agree."""
    s = load_case(name)
    flt, tbl = getattr(s, "filter", None), sample_table()
    o, e = orc.OracleScene(s), emu.EmuScene(s)
    rd = abi.copy_struct(s.render)
    assert rd.sampler_mode == abi.HPT_SAMPLER_BESTCANDIDATE_HASH
    rd.seed = 5
    fo, so = o.render(s.camera, rd, flt=flt, sample_table=tbl)
    fe, se = e.render(s.camera, rd, flt=flt, sample_table=tbl)
    assert so[0] == se[0] > 0 and so[5] == se[5] == 0
    assert abs(int(so[1]) - int(se[1])) <= 4 and abs(int(so[2]) - int(se[2])) <= 4
    if flt is None:
        assert np.array_equal(fo[..., 3], fe[..., 3])
    io, ie = film.xyzw_to_rgb(fo), film.xyzw_to_rgb(fe)
    assert differing_pixels(io, ie) < 1e-3 and film.rmse(io, ie) < 1e-4
    rd.sampler_mode = abi.HPT_SAMPLER_BESTCANDIDATE_MT_REPLAY              # the same camera samples as the reference's windows hand out
    fr, sr = o.render(s.camera, rd, nthreads=1, flt=flt, sample_table=tbl)
    assert sr[0] == so[0]
    if flt is None:
        assert np.array_equal(fr[..., 3], fo[..., 3])


def test_bestcandidate_tile_shifts_and_shards():
    """The host-side tile shifts (hpt_bc.h: first three outputs of MT19937 seeded xTile + (yTile << 8)) against the oracle's generator through a
    render with a lens and a shutter; shards partition the table tiles."""
    s = load_case("banim")
    tbl = sample_table()
    rd = abi.copy_struct(s.render)
    cam = abi.copy_struct(s.camera)
    cam.lens_radius, cam.focal_distance = 0.05, 10.0               # lens samples: columns 3, 4 of the table + shifts 1, 2
    o, e = orc.OracleScene(s), emu.EmuScene(s)
    fo, so = o.render(cam, rd, sample_table=tbl)
    full, st = e.render(cam, rd, sample_table=tbl)
    assert so[0] == st[0] and film.rmse(film.xyzw_to_rgb(fo), film.xyzw_to_rgb(full)) < 1e-4
    acc, n = np.zeros_like(full), 0
    for r in range(3):
        rd.shard_rank, rd.shard_count = r, 3
        f, sr = e.render(cam, rd, sample_table=tbl)
        acc += f; n += int(sr[0])
    assert n == int(st[0]) and np.array_equal(acc[..., 3], full[..., 3]) and np.allclose(acc, full, rtol=1e-5, atol=1e-5)


def test_stratified_sampler_values_stratify():
    """Bit-identical values oracle / device for grids up to the 4095-sample limit (the device divides through float, exactly);
    one image sample and one lens sample per stratum, one time sample per 1-D stratum; unjittered = stratum centres."""
    base = load_case("sk").render
    for (n, xs) in [(6, 3), (35, 7), (4095, 65), (4094, 2), (4092, 1023)]:
        rd = abi.copy_struct(base)
        rd.seed, rd.spp, rd.sampler_mode = 3, n, abi.stratified_mode(abi.HPT_SAMPLER_STRATIFIED_HASH, xs, True)
        a, b = orc.sampler(rd, 2, 5), emu.sampler(rd, 2, 5)
        assert np.array_equal(a, b)
        ys = n // xs
        assert len(set(zip(np.floor((a[:, 0] - 2) * xs).astype(int), np.floor((a[:, 1] - 5) * ys).astype(int)))) == n
        assert len(set(zip(np.floor(a[:, 2] * xs).astype(int), np.floor(a[:, 3] * ys).astype(int)))) == n
        t = np.sort(a[:, 4].astype(np.float64) * n)            # one per 1-D stratum, up to float rounding at a stratum's upper edge
        assert np.all(t >= np.arange(n) - 1e-3) and np.all(t <= np.arange(n) + 1 + 1e-3)
    rd = abi.copy_struct(base)
    rd.spp, rd.sampler_mode = 6, abi.stratified_mode(abi.HPT_SAMPLER_STRATIFIED_HASH, 3, False)
    a = emu.sampler(rd, 0, 0)
    assert np.allclose(np.sort(a[:, 0]), np.repeat([1 / 6, 3 / 6, 5 / 6], 2)) and np.allclose(np.sort(a[:, 1]), np.repeat([.25, .75], 3))


def test_exr_environment_map_render_matches_oracle():
    """SURVEY.md §8f-3, first step: an importance-sampled 32x16 HDR environment map (Distribution2D sampling, pdf, bilinear
    lookup) through the device lane against the oracle, which is pinned to the EXR-enabled reference on this scene."""
    s = load_case("envmap")
    o, e = orc.OracleScene(s), emu.EmuScene(s)
    rd = hash_rd(s, seed=5)
    fo, so = o.render(s.camera, rd)
    fe, se = e.render(s.camera, rd)
    assert so[0] == se[0] == rd.x_count * rd.y_count * rd.spp
    assert np.array_equal(fo[..., 3], fe[..., 3])
    io, ie = film.xyzw_to_rgb(fo), film.xyzw_to_rgb(fe)
    assert differing_pixels(io, ie) < 2e-3 and film.rmse(io, ie) < 1e-4


def test_everything_at_once_matches_oracle():
    """Direct lighting (3 light samples) + Sampler "stratified" 3 x 2 + PixelFilter "mitchell" 2.5 x 1.5 + crop window through the
    device lane with the two-pass film, against the oracle (pinned to the reference binary on this scene)."""
    s = load_case("fcombo")
    o, e = orc.OracleScene(s), emu.EmuScene(s)
    rd = abi.copy_struct(s.render)
    rd.seed = 5
    fo, so = o.render(s.camera, rd, flt=s.filter)
    fe, se = e.render(s.camera, rd, flt=s.filter, two_pass=True)
    assert so[0] == se[0] and abs(int(so[2]) - int(se[2])) <= 4
    np.testing.assert_allclose(fe[..., 3], fo[..., 3], rtol=2e-5, atol=1e-5)
    scale = float(np.abs(fo[..., :3]).max())
    assert (np.abs(fe - fo)[..., :3].max(axis=2) <= 2e-5 * scale).mean() > 0.999


def test_exr_environment_map_under_direct_lighting_matches_oracle():
    s = load_case("envmap_dl")
    o, e = orc.OracleScene(s), emu.EmuScene(s)
    rd = abi.copy_struct(s.render)
    rd.seed = 5
    fo, so = o.render(s.camera, rd)
    fe, se = e.render(s.camera, rd)
    assert so[0] == se[0] and abs(int(so[2]) - int(se[2])) <= 4
    assert np.array_equal(fo[..., 3], fe[..., 3])
    io, ie = film.xyzw_to_rgb(fo), film.xyzw_to_rgb(fe)
    assert differing_pixels(io, ie) < 2e-3 and film.rmse(io, ie) < 1e-4
