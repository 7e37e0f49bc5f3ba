"""Shared test helpers.  `-m "not gpu"`: oracle pin against the golden fixtures, host logic, ABI
surface, the test-only CPU emulation of the device code.  `-m gpu`: parity tests proper, through
the C ABI of libhpt.so on cuda:0 (an MI355X)."""
import importlib
import gzip
import os
import sys

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")

abi = importlib.import_module("pbrt-v2_amd.abi")


def load_ref(name):
    import gzip
    with gzip.open(os.path.join(GOLDEN, name + ".ref.npy.gz"), "rb") as f:
        return np.load(f)


def load_case(name):
    """-> abi.Scene with the camera / render descriptor the golden image was rendered with."""
    if name == "cfg1":
        return abi.Scene.load(os.path.join(GOLDEN, "killeroo_cfg1.hpts.gz"))
    if name == "k8":
        s = abi.Scene.load(os.path.join(GOLDEN, "killeroo_cfg1.hpts.gz"))
        v = np.load(os.path.join(GOLDEN, "k8.view.npz"))
        s.camera = abi.Camera.from_buffer_copy(v["camera"].tobytes())
        s.render = abi.RenderDesc.from_buffer_copy(v["render"].tobytes())
        return s
    if name == "b8":
        return abi.Scene.load(os.path.join(GOLDEN, "bunny_b8.hpts.gz"))
    if name == "anim":
        return abi.Scene.load(os.path.join(GOLDEN, "anim_killeroos.hpts.gz"))
    if name == "ms":
        return abi.Scene.load(os.path.join(GOLDEN, "ms_soup.hpts.gz"))
    if name == "env":
        return abi.Scene.load(os.path.join(GOLDEN, "env_soup.hpts.gz"))
    if name == "envmap":   # SURVEY.md §8f-3, first step: infinite light with a 32x16 .exr map (tests/golden/make_golden_envmap.py)
        return abi.Scene.load(os.path.join(GOLDEN, "envmap_soup.hpts.gz"))
    if name in DL_CASES:   # direct-lighting cases: committed geometry + their own camera / render descriptor / light records
        s = abi.Scene.load(os.path.join(GOLDEN, DL_CASES[name]))
        v = np.load(os.path.join(GOLDEN, name + ".view.npz"))
        s.camera = abi.Camera.from_buffer_copy(v["camera"].tobytes())
        s.render = abi.RenderDesc.from_buffer_copy(v["render"].tobytes())
        lights = abi.lights_from_bytes(v["lights"].tobytes(), len(s.lights))
        assert len(bytes(lights)) == len(bytes(s.lights))
        s.lights = lights
        return s
    if name == "envmap_dl":   # the envmap scene under direct lighting (map sampled 4x per camera sample) with Sampler "random" 3 spp
        s = abi.Scene.load(os.path.join(GOLDEN, "envmap_soup.hpts.gz"))
        v = np.load(os.path.join(GOLDEN, "envmap_dl.view.npz"))
        s.camera = abi.Camera.from_buffer_copy(v["camera"].tobytes())
        s.render = abi.RenderDesc.from_buffer_copy(v["render"].tobytes())
        s.lights = abi.lights_from_bytes(v["lights"].tobytes(), len(s.lights))
        return s
    if name in RANDOM_CASES or name in STRATIFIED_CASES:   # Sampler "random" / "stratified" cases: committed geometry + camera / render descriptor (sampler mode, spp) / lights
        s = abi.Scene.load(os.path.join(GOLDEN, (RANDOM_CASES.get(name) or STRATIFIED_CASES[name])))
        v = np.load(os.path.join(GOLDEN, name + ".view.npz"))
        s.camera = abi.Camera.from_buffer_copy(v["camera"].tobytes())
        s.render = abi.RenderDesc.from_buffer_copy(v["render"].tobytes())
        s.lights = abi.lights_from_bytes(v["lights"].tobytes(), len(s.lights))
        return s
    if name == "acam":     # round 3: a moving camera — the blob + the camera's AnimatedTransform record (tests/golden/make_golden_r3.py)
        s = abi.Scene.load(os.path.join(GOLDEN, "acam.hpts.gz"))
        s.camera_motion = abi.Instance.from_buffer_copy(np.load(os.path.join(GOLDEN, "acam.view.npz"))["camera_motion"].tobytes())
        return s
    if name in R2_CASES:
        s = abi.Scene.load(os.path.join(GOLDEN, R2_CASES[name]))
        if name == "merl":       # the 17.5 MB half-angle table is rebuilt from its formula instead of being committed
            assert all(m.rh_off == s.fpool.size for m in s.materials if m.kind == abi.HPT_MAT_MEASURED_REGULAR)
            s.fpool = np.concatenate([s.fpool, merl_table()])
        return s
    if name in R2_VIEW_CASES:
        s = abi.Scene.load(os.path.join(GOLDEN, R2_VIEW_CASES[name]))
        v = np.load(os.path.join(GOLDEN, name + ".view.npz"))
        s.camera = abi.Camera.from_buffer_copy(v["camera"].tobytes())
        s.render = abi.RenderDesc.from_buffer_copy(v["render"].tobytes())
        s.lights = abi.lights_from_bytes(v["lights"].tobytes(), len(s.lights))
        return s
    if name in ADAPTIVE_CASES or name in BESTCANDIDATE_CASES:   # Sampler "adaptive" / "bestcandidate" cases (tests/golden/make_golden_adaptive.py, make_golden_bestcandidate.py)
        s = abi.Scene.load(os.path.join(GOLDEN, ADAPTIVE_CASES.get(name) or BESTCANDIDATE_CASES[name]))
        v = np.load(os.path.join(GOLDEN, name + ".view.npz"))
        s.camera = abi.Camera.from_buffer_copy(v["camera"].tobytes())
        s.render = abi.RenderDesc.from_buffer_copy(v["render"].tobytes())
        s.lights = abi.lights_from_bytes(v["lights"].tobytes(), len(s.lights))
        if "filter" in v.files:
            s.filter = abi.filter_from_array(v["filter"])
        return s
    if name in HALTON_CASES:   # Sampler "halton" cases (tests/golden/make_golden_halton.py): as the random / stratified ones; `hgauss` with the film's filter
        s = abi.Scene.load(os.path.join(GOLDEN, HALTON_CASES[name]))
        v = np.load(os.path.join(GOLDEN, name + ".view.npz"))
        s.camera = abi.Camera.from_buffer_copy(v["camera"].tobytes())
        s.render = abi.RenderDesc.from_buffer_copy(v["render"].tobytes())
        s.lights = abi.lights_from_bytes(v["lights"].tobytes(), len(s.lights))
        if "filter" in v.files:
            s.filter = abi.filter_from_array(v["filter"])
        return s
    if name in FILTER_CASES or name in COMBO_CASES:   # reconstruction-filter cases: as above + the film's filter (scene.filter)
        s = abi.Scene.load(os.path.join(GOLDEN, (FILTER_CASES.get(name) or COMBO_CASES[name])))
        v = np.load(os.path.join(GOLDEN, name + ".view.npz"))
        s.camera = abi.Camera.from_buffer_copy(v["camera"].tobytes())
        s.render = abi.RenderDesc.from_buffer_copy(v["render"].tobytes())
        s.lights = abi.lights_from_bytes(v["lights"].tobytes(), len(s.lights))
        s.filter = abi.filter_from_array(v["filter"])
        return s
    raise KeyError(name)


CASES = ["cfg1", "k8", "b8", "env", "anim", "ms"]
# SURVEY.md §8f-1 (tests/golden/make_golden_dl.py): DirectLightingIntegrator, strategy all / one
DL_CASES = {"dl1": "killeroo_cfg1.hpts.gz", "dlone": "killeroo_cfg1.hpts.gz", "dlb": "bunny_b8.hpts.gz", "dlbone": "bunny_b8.hpts.gz",
            "dlanim": "anim_killeroos.hpts.gz"}


# SURVEY.md §8f-4 (tests/golden/make_golden_filter.py): PixelFilter gaussian / mitchell / triangle (+ crop window) / sinc
FILTER_CASES = {"fgauss": "killeroo_cfg1.hpts.gz", "fmitch": "bunny_b8.hpts.gz", "ftri": "killeroo_cfg1.hpts.gz",
                "fsinc": "anim_killeroos.hpts.gz"}
# everything at once (CPU tests): direct lighting with 3 light samples, Sampler "stratified" 3 x 2, PixelFilter "mitchell" 2.5 x 1.5, crop window
COMBO_CASES = {"fcombo": "killeroo_cfg1.hpts.gz"}


# SURVEY.md §8f-4 (tests/golden/make_golden_random.py): Sampler "random" — path 6 spp, direct lighting with 5 light samples
# at 3 spp, bunny path 4 spp, animated scene direct lighting 5 spp
RANDOM_CASES = {"rk": "killeroo_cfg1.hpts.gz", "rdl": "killeroo_cfg1.hpts.gz", "rb": "bunny_b8.hpts.gz", "ranim": "anim_killeroos.hpts.gz"}
# Sampler "stratified" (same generator): 3 x 2 jittered, path; 2 x 2 jittered, direct lighting with 5 light samples; 2 x 3 unjittered,
# path on the animated scene
STRATIFIED_CASES = {"sk": "killeroo_cfg1.hpts.gz", "sdl": "killeroo_cfg1.hpts.gz", "sanim": "anim_killeroos.hpts.gz"}
# Sampler "halton" (tests/golden/make_golden_halton.py): 3 spp, path; 2 spp, direct lighting with 5 light samples; 4 spp on the animated scene (time
# sample, windows that are not square); 2 spp under PixelFilter "gaussian" (windows cut from the sample extent)
# Sampler "bestcandidate" (tests/golden/make_golden_bestcandidate.py): 4 spp path; 3 spp direct lighting, 5 -> 8 light samples; 2 spp on the animated scene;
# 2 spp under PixelFilter "gaussian" (table tiles with negative coordinates)
BESTCANDIDATE_CASES = {"bk": "killeroo_cfg1.hpts.gz", "bdl": "killeroo_cfg1.hpts.gz", "banim": "anim_killeroos.hpts.gz", "bgauss": "killeroo_cfg1.hpts.gz"}


def sample_table():
    """BestCandidateSampler::sampleTable of the reference build (4096 x 5 float32), dumped by the host plugin (make_golden_bestcandidate.py)"""
    with gzip.open(os.path.join(GOLDEN, "bestcandidate_table.npy.gz"), "rb") as f:
        return np.load(f)


# Sampler "adaptive", method "contrast" (tests/golden/make_golden_adaptive.py): 2 .. 8 samples, path; 4 .. 16, direct lighting; 2 .. 4 on the animated scene
ADAPTIVE_CASES = {"ak": "killeroo_cfg1.hpts.gz", "adl": "killeroo_cfg1.hpts.gz", "aanim": "anim_killeroos.hpts.gz"}
HALTON_CASES = {"hk": "killeroo_cfg1.hpts.gz", "hdl": "killeroo_cfg1.hpts.gz", "hanim": "anim_killeroos.hpts.gz", "hgauss": "killeroo_cfg1.hpts.gz"}


# ---- round 2 cases (tests/golden/make_golden_r2.py): Oren-Nayar, specular, triangle emitters, regular half-angle BRDF, textures, alpha
R2_CASES = {"on": "on.hpts.gz", "spec": "spec.hpts.gz", "trilight": "trilight.hpts.gz", "merl": "merl.hpts.gz", "tex": "tex.hpts.gz",
            "alpha": "alpha.hpts.gz", "metal": "metal.hpts.gz", "mirtex": "mirtex.hpts.gz",
            # round 3 (tests/golden/make_golden_r3.py): metal.pbrt as shipped under the environment map SURVEY.md §8d names, textures/grace_latlong.exr (1000 x 500)
            "metalg": "metalg.hpts.gz",
            # round 3: TriangleMesh "vector S" — explicit tangents under anisotropic substrates and a metal octahedron (row a13)
            "tang": "tang.hpts.gz",
            # round 3: image textures, a roughness texture and bump mapping on spheres and a disk (row a14)
            "qtex": "qtex.hpts.gz",
            # round 3 (tests/golden/make_golden_aquad.py): ANIMATED spheres / disks — TransformedPrimitive over a bare GeometricPrimitive (ABI 8,
            # hpt_instance.quadric1): textured + bump-mapped partial sphere, tilting annulus, a moving octahedron beside them, a mirror wall (path);
            # scenes/anim-moving-reflection.pbrt as shipped at 100 x 100 / 4 spp under the 32 x 16 stand-in map (direct lighting, its default)
            "aquad": "aquad.hpts.gz", "aquaddl": "aquaddl.hpts.gz",
            # round 3 (tests/golden/make_golden_lights.py): SpotLight (one under a rotated, non-uniformly scaled CTM; one hard-edged) and DistantLight
            # beside a point light — delta lights, unbounded shadow rays — under the path integrator and under direct lighting "all" (ABI 8)
            "lts": "lts.hpts.gz", "ltsdl": "ltsdl.hpts.gz",
            # round 3 (tests/golden/make_golden_oinst.py): OBJECT INSTANCING — two objects (meshes with their own ObjectToWorld, vertex normals, explicit
            # tangents) instanced four + two times (static, scaled, mirrored, animated): TransformedPrimitives over shared aggregates (ABI 8, quadric1 < 0)
            "oinst": "oinst.hpts.gz",
            # round 3 (tests/golden/make_golden_abi8dl.py): everything ABI 8 added, together, under direct lighting "one" with the specular recursion
            # (run Z: on the device RMSE 1.7e-6 against the oracle, identical ray counts)
            "abi8dl": "abi8dl.hpts.gz",
            # round 6 (tests/golden/make_golden_texmap.py): image maps through "spherical" (under a texture-space transform), "cylindrical" and "planar"
            # TextureMapping2Ds — Kd, roughness, bump and alpha textures (ABI 9, hpt_texture.mapping / map_m); `texmapdl`: the same through a mirror
            # under direct lighting (finite differences over the specular rays' dpdx / dpdy)
            "texmap": "texmap.hpts.gz",
            # round 6: scale / mix textures nested seven deep over uv maps (the general evaluator's explicit stack instead of the three template levels)
            "texdeep": "texdeep.hpts.gz",
            # round 6 (tests/golden/make_golden_oemit.py): EMITTERS INSIDE OBJECT INSTANCES — area lights the reference leaves out of Scene::lights (core/api.cpp:1046-1049) but
            # whose shapes still emit at camera rays and specular bounces: unsampled light records behind the scene's lights (include/hpt.h, HPT_LIGHT_UNSAMPLED)
            "oemit": "oemit.hpts.gz"}
R2_VIEW_CASES = {"specdl": "spec.hpts.gz", "trildl": "trilight.hpts.gz", "lens": "tex.hpts.gz", "texmapdl": "texmap.hpts.gz", "oemitdl": "oemit.hpts.gz"}     # same geometry, own camera / render descriptor / lights


def nest_textures(s, levels):
    """`s` (a scene with an image map behind some material's Kd) with that Kd wrapped in `levels` more scale textures (x 0.98 a level): the depth of
    the table is levels + the depth Kd had.  -> the deepest texture's index"""
    m = next(m for m in s.materials if m.tex[abi.TEXSLOT_KD] >= 0)
    tex = list(s.textures)
    c = abi.Texture(); c.kind, c.channels = abi.HPT_TEX_CONSTANT, 3
    c.value[0] = c.value[1] = c.value[2] = 0.98
    c.tex1 = c.tex2 = c.amount = -1
    tex.append(c)
    const, top = len(tex) - 1, m.tex[abi.TEXSLOT_KD]
    for _ in range(levels):
        t = abi.Texture(); t.kind, t.channels, t.tex1, t.tex2, t.amount = abi.HPT_TEX_SCALE, 3, top, const, -1
        tex.append(t); top = len(tex) - 1
    s.textures = abi._arr(abi.Texture, len(tex))
    for i, t in enumerate(tex):
        s.textures[i] = t
    m.tex[abi.TEXSLOT_KD] = top
    return top


def merl_table_doubles():
    """The synthetic MERL-layout table of the `merl` case: (3, 90 * 90 * 180) float64, channel-major like a .binary file
    (materials/measured.cpp:137-184).  A smooth lobe in the half angle, modulated in the difference angles; values are what a
    file would hold (the reader multiplies by 1/1500, 1.15/1500, 1.66/1500)."""
    ih, idd, ip = np.meshgrid(np.arange(90, dtype=np.float64), np.arange(90, dtype=np.float64), np.arange(180, dtype=np.float64), indexing="ij")
    base = 40.0 + 3000.0 * np.exp(-ih / 6.0) * (1.0 - 0.5 * idd / 90.0) + 200.0 * (1.0 + np.cos(ip * np.pi / 90.0)) * (idd / 90.0)
    out = np.stack([base * k for k in (1.0, 0.8, 0.55)]).reshape(3, -1)
    out[1, ::977] = -5.0          # negative entries: the reader clamps them to 0 (measured.cpp:177)
    return np.ascontiguousarray(out)


def merl_table():
    """... as MeasuredMaterial stores it: float32, interleaved RGB, max(0., value * scale) evaluated in double."""
    t = merl_table_doubles()
    f32 = np.float32
    scales = [f32(1.0) / f32(1500.0), f32(1.15) / f32(1500.0), f32(1.66) / f32(1500.0)]
    out = np.zeros((t.shape[1], 3), np.float32)
    for c in range(3):
        out[:, c] = np.maximum(0.0, t[c] * np.float64(scales[c])).astype(np.float32)
    return out.reshape(-1)


def hash_rd(scene, seed=7, spp=None):
    rd = abi.copy_struct(scene.render)
    rd.sampler_mode = abi.HPT_SAMPLER_LD_HASH
    rd.seed = seed
    if spp:
        rd.spp = spp
    return rd


def random_rays(scene, n, seed=1):
    """aggregatetest-style rays (renderers/aggregatetest.cpp:74-89): origins in an expanded world
    box, random / axis-aligned directions, a share of finite segments."""
    rng = np.random.default_rng(seed)
    P = []
    for m in scene.meshes:
        P.append(scene.fpool[m.p_off:m.p_off + 3 * m.nverts].reshape(-1, 3))
    P = np.concatenate(P) if P else np.zeros((1, 3), np.float32)
    lo, hi = P.min(0), P.max(0)
    # ignore huge ground quads when sizing the box
    lo, hi = np.maximum(lo, np.percentile(P, 2, axis=0) - 1), np.minimum(hi, np.percentile(P, 98, axis=0) + 1)
    ext = hi - lo
    o = rng.uniform(lo - 0.5 * ext, hi + 0.5 * ext, size=(n, 3))
    tgt = rng.uniform(lo, hi, size=(n, 3))
    d = tgt - o
    axis = rng.integers(0, 8, size=n)
    for a in range(3):  # some exactly axis-aligned directions (zero components -> inf inverse dirs)
        sel = axis == a
        d[sel] = 0
        d[sel, a] = np.where(rng.random(sel.sum()) < 0.5, 1.0, -1.0) * ext[a]
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    rays = np.zeros((n, 8), np.float32)
    rays[:, :3], rays[:, 3:6] = o, d
    rays[:, 6] = np.where(rng.random(n) < 0.3, rng.uniform(0, 0.2, n) * np.linalg.norm(ext), 0.0)
    rays[:, 7] = np.where(rng.random(n) < 0.3, rng.uniform(0.5, 2.0, n) * np.linalg.norm(ext), np.inf)
    return rays


def bsdf_inputs(n, seed=3):
    rng = np.random.default_rng(seed)

    def unit(k):
        v = rng.normal(size=(k, 3))
        return v / np.linalg.norm(v, axis=1, keepdims=True)
    inp = np.zeros((n, 16), np.float32)
    nn = unit(n)
    inp[:, 0:3], inp[:, 3:6] = unit(n), unit(n)
    inp[:, 6:9] = rng.random((n, 3)) * 0.999
    inp[:, 9:12] = nn
    t = np.cross(nn, unit(n))
    inp[:, 12:15] = t / np.linalg.norm(t, axis=1, keepdims=True) * rng.uniform(0.5, 2.0, (n, 1))
    inp[:, 15] = np.where(rng.random(n) < 0.1, -1.0, 1.0)
    return inp


def sub_windows(rd):
    """Sampler::ComputeSubWindow (core/sampler.cpp:55-74) for every task: list of (x0, x1, y0, y1)
    relative to the film extent."""
    f32 = np.float32
    xs, xe, ys, ye = rd.x_start, rd.x_start + rd.x_count, rd.y_start, rd.y_start + rd.y_count
    dx, dy = xe - xs, ye - ys
    nx, ny = rd.ntasks, 1
    while (nx & 1) == 0 and 2 * dx * ny < dy * nx:
        nx >>= 1
        ny <<= 1
    out = []
    for num in range(rd.ntasks):
        xo, yo = num % nx, num // nx
        tx0, tx1 = f32(xo) / f32(nx), f32(xo + 1) / f32(nx)
        ty0, ty1 = f32(yo) / f32(ny), f32(yo + 1) / f32(ny)
        lerp = lambda t, a, b: f32(f32(f32(1) - t) * f32(a)) + f32(t * f32(b))
        x0, x1 = int(np.floor(lerp(tx0, xs, xe))), int(np.floor(lerp(tx1, xs, xe)))
        y0, y1 = int(np.floor(lerp(ty0, ys, ye))), int(np.floor(lerp(ty1, ys, ye)))
        out.append((x0 - xs, x1 - xs, y0 - ys, y1 - ys))
    return out


# ---- parity at BASELINE size: crop windows of a full frame against the oracle (tests/test_gpu_fullsize.py, bench.py) ----
CROP = 64


def crop_windows(xres, yres, n=CROP):
    """-> list of (name, x0, y0): corners, centre, and a window across the border of XCD bands 0 | 1"""
    n_stx, n_sty = (xres + 31) // 32, (yres + 31) // 32
    tiles = n_stx * n_sty
    t1 = tiles * 1 // 8                       # first tile of band 1
    bx, by = (t1 % n_stx) * 32, (t1 // n_stx) * 32
    wins = [("top-left", 0, 0), ("top-right", xres - n, 0), ("bottom-left", 0, yres - n), ("bottom-right", xres - n, yres - n),
            ("centre", (xres - n) // 2, (yres - n) // 2),
            ("xcd-band-border", min(max(bx - n // 2, 0), xres - n), min(max(by - n // 2, 0), yres - n))]
    return wins


def content_windows(frame, k=24, n=CROP):
    """-> the k windows of n x n pixels of `frame` (the device's film, (H, W, 4) XYZ + weight) with the most going on: highest variance of
    the pixel luminance Y / w inside the window — silhouettes, shadow edges, texture, noise of deep paths — on the grid of
    non-overlapping windows.  The frame corners are mostly sky / floor in the shipped scenes; these are where the shading is."""
    H, W = frame.shape[:2]
    y = frame[..., 1].astype(np.float64) / np.maximum(frame[..., 3].astype(np.float64), 1e-30)
    y = np.log1p(np.maximum(y, 0.0))                # emitters (L = 2000 in killeroo-simple) must not drown everything else
    gy, gx = H // n, W // n
    v = y[:gy * n, :gx * n].reshape(gy, n, gx, n).transpose(0, 2, 1, 3).reshape(gy, gx, n * n).var(axis=2)
    order = np.argsort(v.ravel())[::-1][:k]
    return [("content-%d-%d" % (i % gx, i // gx), int(i % gx) * n, int(i // gx) * n) for i in order]


def compare_crops(scene, oracle, frame, rd, flt=None, tol=1e-3, n=CROP, content=0, stats=None, crop_tol=None):
    """frame: the device's full film for `rd`.  Crops: corners, centre, XCD-band border + the `content` highest-variance windows.
    Film weights must be identical in every crop.  The per-pixel RMSE (sqrt(sum d^2 / (3 W H))) is taken over ALL verified
    pixels together and must stay below `tol`; a single crop may reach crop_tol (default = tol: every crop on its own; the HDR scene passes
    a looser one — there one camera sample in half a million whose discrete decision flips on an ulp of the device's libm moves a pixel by
    more than the whole tolerance).  Returns (rmse over all crops, worst crop); stats (optional list of 6): oracle counters summed."""
    worst, sq, npx = 0.0, 0.0, 0
    import importlib
    film = importlib.import_module("pbrt-v2_amd.film")
    wins = crop_windows(rd.x_count, rd.y_count, n) + (content_windows(frame, content, n) if content else [])
    for name, x0, y0 in wins:
        ax0, ay0 = max(x0 - 1, 0), max(y0 - 1, 0)
        ax1, ay1 = min(x0 + n + 1, rd.x_count), min(y0 + n + 1, rd.y_count)
        crd = abi.copy_struct(rd)
        crd.x_start, crd.y_start, crd.x_count, crd.y_count = ax0, ay0, ax1 - ax0, ay1 - ay0
        crd.count_work = 0
        fo, so = oracle.render(scene.camera, crd, flt=flt)
        if stats is not None:
            for i in range(6):
                stats[i] += int(so[i])
        fo = fo[y0 - ay0:y0 - ay0 + n, x0 - ax0:x0 - ax0 + n]
        fd = frame[y0:y0 + n, x0:x0 + n]
        assert np.array_equal(fo[..., 3], fd[..., 3]), "%s: film weights differ" % name
        a, b = film.xyzw_to_rgb(fo).astype(np.float64), film.xyzw_to_rgb(fd).astype(np.float64)
        d2 = float(((a - b) ** 2).sum())
        err = (d2 / a.size) ** 0.5
        assert err < (crop_tol or tol), (name, err)
        worst, sq, npx = max(worst, err), sq + d2, npx + a.size
    total = (sq / npx) ** 0.5
    assert total < tol, total
    return total, worst


def with_instance_copies(scene, src, n_copies, start=(20.0, 0.0, 0.0), step=(3.0, 0.0, 0.0)):
    """`scene` with n_copies more instances: copies of instance `src` (object instancing — they share its aggregate, hpt_instance.quadric1 < 0)
    moved by start + j * step in world space.  PrimitiveToWorld' = Translate(v) PrimitiveToWorld, so WorldToPrimitive' = WorldToPrimitive
    Translate(-v): the translation column of both end matrices (and of their decomposition, T) loses M3 v, the inverses gain v, the motion
    bounds move by v; rotation and scale stay.  Round 4: scenes of MANY instances for the top-level tree."""
    import ctypes as C
    inst = list(scene.instances)
    owner = src if inst[src].quadric1 == 0 else -inst[src].quadric1 - 1
    assert inst[owner].quadric1 == 0, "copies of a mesh instance only"
    out = [abi.Instance.from_buffer_copy(bytes(i)) for i in inst]
    for j in range(n_copies):
        v = np.asarray(start, dtype=np.float64) + j * np.asarray(step, dtype=np.float64)
        c = abi.Instance.from_buffer_copy(bytes(inst[src]))
        c.quadric1 = -(owner + 1)
        for k in range(3):
            c.bounds[k] = np.float32(c.bounds[k] + v[k]); c.bounds[3 + k] = np.float32(c.bounds[3 + k] + v[k])
        for e in range(2):
            m = np.array(list(c.w2p_m[e]), dtype=np.float64).reshape(4, 4)
            mi = np.array(list(c.w2p_minv[e]), dtype=np.float64).reshape(4, 4)
            d = -(m[:3, :3] @ v)
            for k in range(3):
                c.w2p_m[e][4 * k + 3] = np.float32(m[k, 3] + d[k])
                c.T[e][k] = np.float32(c.T[e][k] + d[k])
                c.w2p_minv[e][4 * k + 3] = np.float32(mi[k, 3] + v[k])
        out.append(c)
    s2 = abi.Scene(meshes=list(scene.meshes), quadrics=list(scene.quadrics), materials=list(scene.materials), lights=list(scene.lights),
                   fpool=scene.fpool, ipool=scene.ipool, camera=scene.camera, render=scene.render, instances=out, textures=list(scene.textures))
    for attr in ("filter", "camera_motion", "meta"):
        if hasattr(scene, attr):
            setattr(s2, attr, getattr(scene, attr))
    return s2


def with_quadric_padding(scene, n_pad, at=(100.0, 100.0, 100.0), step=(3.0, 0.0, 0.0)):
    """`scene` with n_pad small world spheres (copies of its last world quadric without its emission, far outside the scene) placed BEFORE its own quadrics, so that
    the quadrics instances own get indices >= n_pad: hpt_instance.quadric1 and hpt_light.quadric move with them.  Round 4: the device keeps a bit per owned quadric
    (DScene::inst_quadric_mask) for indices < 31 and looks the rest up in the instance table."""
    owned = {i.quadric1 - 1 for i in scene.instances if i.quadric1 > 0}
    src = max(k for k in range(len(scene.quadrics)) if k not in owned)
    pads = []
    for j in range(n_pad):
        q = abi.Quadric.from_buffer_copy(bytes(scene.quadrics[src]))
        q.arealight = -1
        v = np.asarray(at, dtype=np.float64) + j * np.asarray(step, dtype=np.float64)
        m = np.eye(4); m[:3, 3] = v
        mi = np.eye(4); mi[:3, 3] = -v
        for k in range(16):
            q.o2w[k] = np.float32(m.flat[k]); q.o2w_inv[k] = np.float32(mi.flat[k])
        pads.append(q)
    inst = [abi.Instance.from_buffer_copy(bytes(i)) for i in scene.instances]
    for i in inst:
        if i.quadric1 > 0:
            i.quadric1 += n_pad
    lights = [type(l).from_buffer_copy(bytes(l)) for l in scene.lights]
    for l in lights:
        if l.kind == abi.HPT_LIGHT_DIFFUSE_AREA and l.quadric >= 0:
            l.quadric += n_pad
    s2 = abi.Scene(meshes=list(scene.meshes), quadrics=pads + list(scene.quadrics), materials=list(scene.materials), lights=lights,
                   fpool=scene.fpool, ipool=scene.ipool, camera=scene.camera, render=scene.render, instances=inst, textures=list(scene.textures))
    for attr in ("filter", "camera_motion", "meta"):
        if hasattr(scene, attr):
            setattr(s2, attr, getattr(scene, attr))
    return s2
