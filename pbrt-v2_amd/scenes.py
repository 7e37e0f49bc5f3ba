"""Host-side scene construction for the C ABI without the pbrt parser: the synthetic
triangle-soup workload of BASELINE.json configs[2] and a .pbrt exporter so the
very same scene can be fed to the reference binary (oracle/_ref/pbrt) in this container.

Camera / light set-up follows the reference formulas:
  LookAt                       core/transform.cpp:223-243
  Perspective                  core/transform.cpp:416-427
  ProjectiveCamera ctor        core/camera.cpp:83-102
  CreatePerspectiveCamera      cameras/perspective.cpp:141-175 (screen window from aspect)
  InfiniteAreaLight ctor       lights/infinite.cpp:68-105  (1x1 constant map)
  Distribution1D/2D            core/montecarlo.h:54-167, montecarlo.cpp
The matrices are computed in float64 and rounded once to float32, so they agree with pbrt's
own float32 pipeline to ~1e-7 relative, not bit-for-bit; tests that need the reference's exact
matrices use blobs dumped by host/hip_renderer.cpp instead.
"""
import math

import numpy as np

from . import abi

SYNTH_SEED = 0x5EED0001


def _m16(a):
    m = abi.M16()
    flat = np.asarray(a, dtype=np.float32).reshape(16)
    for i in range(16):
        m[i] = float(flat[i])
    return m


def look_at(pos, look, up):
    """world-to-camera, as pbrt's LookAt (returns (world_to_camera, camera_to_world), float64)."""
    pos, look, up = (np.asarray(v, dtype=np.float64) for v in (pos, look, up))
    d = look - pos
    d /= np.linalg.norm(d)
    left = np.cross(up / np.linalg.norm(up), d)
    left /= np.linalg.norm(left)
    newup = np.cross(d, left)
    c2w = np.eye(4)
    c2w[:3, 0], c2w[:3, 1], c2w[:3, 2], c2w[:3, 3] = left, newup, d, pos
    return np.linalg.inv(c2w), c2w


def perspective_camera(xres, yres, fov_deg, c2w, znear=1e-2, zfar=1000.0):
    frame = xres / yres
    if frame > 1.0:
        screen = (-frame, frame, -1.0, 1.0)
    else:
        screen = (-1.0, 1.0, -1.0 / frame, 1.0 / frame)
    persp = np.array([[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, zfar / (zfar - znear), -zfar * znear / (zfar - znear)],
                      [0, 0, 1, 0]], dtype=np.float64)
    inv_tan = 1.0 / math.tan(math.radians(fov_deg) / 2.0)
    cam_to_screen = np.diag([inv_tan, inv_tan, 1.0, 1.0]) @ persp
    scale1 = np.diag([float(xres), float(yres), 1.0, 1.0])
    scale2 = np.diag([1.0 / (screen[1] - screen[0]), 1.0 / (screen[2] - screen[3]), 1.0, 1.0])
    trans = np.eye(4)
    trans[0, 3], trans[1, 3] = -screen[0], -screen[3]
    screen_to_raster = scale1 @ scale2 @ trans
    raster_to_camera = np.linalg.inv(cam_to_screen) @ np.linalg.inv(screen_to_raster)
    cam = abi.Camera()
    cam.raster_to_camera = _m16(raster_to_camera)
    cam.camera_to_world = _m16(c2w)
    cam.lens_radius, cam.focal_distance = 0.0, 1e30
    cam.shutter_open, cam.shutter_close = 0.0, 1.0
    return cam


def recover_fov(cam, xres, yres):
    """fov (degrees) of a PerspectiveCamera from its RasterToCamera matrix."""
    m = np.array(list(cam.raster_to_camera), dtype=np.float64).reshape(4, 4)
    if xres >= yres:
        p = m @ np.array([xres / 2.0, 0.0, 0.0, 1.0])
        t = abs(p[1] / p[2])
    else:
        p = m @ np.array([0.0, yres / 2.0, 0.0, 1.0])
        t = abs(p[0] / p[2])
    return math.degrees(2.0 * math.atan(t))


def render_desc(xres, yres, spp, maxdepth, seed=0, ncores_for_ntasks=8):
    rd = abi.RenderDesc()
    rd.xres, rd.yres = xres, yres
    rd.x_start, rd.x_count, rd.y_start, rd.y_count = 0, xres, 0, yres
    rd.spp, rd.maxdepth = spp, maxdepth
    rd.sampler_mode, rd.seed = abi.HPT_SAMPLER_LD_HASH, seed
    n = max(32 * ncores_for_ntasks, (xres * yres) // 256)   # samplerrenderer.cpp:203-205
    rd.ntasks = 1 << (n - 1).bit_length()
    rd.shard_rank, rd.shard_count, rd.count_work = 0, 1, 0
    return rd


def constant_infinite_light(L, fpool_list):
    """hpt_light for `LightSource "infinite" "color L" [..]` with no map: 1x1 texel and the
    degenerate Distribution2D the reference builds for it (lights/infinite.cpp:83-104)."""
    f32 = np.float32
    L = np.asarray(L, dtype=np.float32)
    y = f32(0.212671) * L[0] + f32(0.715160) * L[1] + f32(0.072169) * L[2]
    sin_theta = f32(math.sin(float(f32(3.14159265358979323846) * f32(0.5) / f32(1.0))))
    func = f32(y * sin_theta)
    # Distribution1D(f, n=1): cdf = [0, func/1]; funcInt = cdf[1]; cdf[1] /= funcInt (or 1 if 0)
    func_int = f32(func / f32(1.0))
    cdf1 = f32(1.0) if func_int == 0 else f32(func_int / func_int)
    marg_func = func_int
    marg_int = f32(marg_func / f32(1.0))
    marg_cdf1 = f32(1.0) if marg_int == 0 else f32(marg_int / marg_int)
    li = abi.Light()
    li.kind, li.quadric = abi.HPT_LIGHT_INFINITE, -1
    li.env_w = li.env_h = 1

    def push(vals):
        off = sum(len(a) for a in fpool_list)
        fpool_list.append(np.asarray(vals, dtype=np.float32).reshape(-1))
        return off
    li.tex_off = push(L)
    li.cond_func_off = push([func])
    li.cond_cdf_off = push([0.0, cdf1])
    li.cond_int_off = push([func_int])
    li.marg_func_off = push([marg_func])
    li.marg_cdf_off = push([0.0, marg_cdf1])
    li.marg_int = float(marg_int)
    eye = np.eye(4)
    li.l2w, li.l2w_inv = _m16(eye), _m16(eye)
    return li


def matte(kd):
    m = abi.Material()
    m.kind = abi.HPT_MAT_MATTE
    for i in range(3):
        m.kd[i] = float(kd[i])
    m.kd_split_off = m.kd_bits_off = m.kd_data_off = -1
    return m


def metal(eta, k, roughness):
    m = abi.Material()
    m.kind, m.roughness = abi.HPT_MAT_METAL, roughness
    for i in range(3):
        m.eta[i], m.k[i] = float(eta[i]), float(k[i])
    m.kd_split_off = m.kd_bits_off = m.kd_data_off = -1
    return m


def substrate(kd, ks, nu, nv):
    m = abi.Material()
    m.kind, m.nu, m.nv = abi.HPT_MAT_SUBSTRATE, nu, nv
    for i in range(3):
        m.kd[i], m.ks[i] = float(kd[i]), float(ks[i])
    m.kd_split_off = m.kd_bits_off = m.kd_data_off = -1
    return m


def materials_soup(n_tris=3000, xres=160, yres=90, spp=8, maxdepth=5, seed=7):
    """Small test scene for the config-5 BxDFs: three interleaved triangle soups with a copper-like
    metal (Microfacet + FresnelConductor + Blinn), an anisotropic substrate (FresnelBlend +
    Anisotropic, uroughness != vroughness) and an isotropic substrate, under a constant infinite
    light plus a point light."""
    base = synthetic_soup(n_tris=n_tris, xres=xres, yres=yres, spp=spp, maxdepth=maxdepth, seed=seed, extent=0.1)
    P = base.fpool[:9 * n_tris].copy()
    cuts = [0, n_tris // 3, 2 * n_tris // 3, n_tris]
    mats = [metal([0.2, 0.92, 1.1], [3.9, 2.45, 2.14], 0.05), substrate([0.3, 0.4, 0.5], [0.4, 0.3, 0.2], 0.05, 0.2),
            substrate([0.5, 0.5, 0.5], [0.5, 0.5, 0.5], 0.1, 0.1)]
    meshes, idx_parts = [], []
    for i in range(3):
        nt = cuts[i + 1] - cuts[i]
        me = abi.Mesh()
        me.p_off, me.n_off, me.uv_off = 9 * cuts[i], -1, -1
        me.idx_off = 3 * cuts[i]
        me.ntris, me.nverts = nt, 3 * nt
        me.material, me.arealight, me.instance = i, -1, -1
        me.o2w, me.o2w_inv = _m16(np.eye(4)), _m16(np.eye(4))
        meshes.append(me)
        idx_parts.append(np.arange(3 * nt, dtype=np.int32))
    fparts = [P]
    env = constant_infinite_light([0.6, 0.7, 0.9], fparts)
    pt = abi.Light()
    pt.kind, pt.quadric = abi.HPT_LIGHT_POINT, -1
    for i, v in enumerate([0.0, 2.5, 3.0]):
        pt.pos[i] = v
    for i in range(3):
        pt.intensity[i] = 6.0
    l2w = np.eye(4); l2w[:3, 3] = [0.0, 2.5, 3.0]
    pt.l2w, pt.l2w_inv = _m16(l2w), _m16(np.linalg.inv(l2w))
    s = abi.Scene(meshes=meshes, materials=mats, lights=[env, pt], fpool=np.concatenate(fparts),
                  ipool=np.concatenate(idx_parts), camera=base.camera, render=base.render)
    s.meta = base.meta
    return s


def synthetic_soup(n_tris=1_000_000, xres=1920, yres=1080, spp=256, maxdepth=8, seed=SYNTH_SEED,
                   extent=0.01):
    """BASELINE.json configs[2] / SURVEY.md §8d: n_tris random triangles, vertex i of triangle k =
    c_k + s*r, c_k ~ U([-1,1]^3), r ~ U([-1,1]^3), s = 0.01; unshared vertices, no N/uv; matte
    Kd .5; one constant infinite light L=1; LookAt 0 0 3.5 / 0 0 0 / 0 1 0, fov 45.
    PRNG: numpy PCG64 seeded 0x5EED0001 (centres first, then offsets)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    c = rng.uniform(-1.0, 1.0, size=(n_tris, 1, 3))
    r = rng.uniform(-1.0, 1.0, size=(n_tris, 3, 3))
    P = (c + extent * r).astype(np.float32).reshape(-1)
    idx = np.arange(3 * n_tris, dtype=np.int32)
    fpool_parts = [P]
    mesh = abi.Mesh()
    mesh.p_off, mesh.n_off, mesh.uv_off, mesh.idx_off = 0, -1, -1, 0
    mesh.ntris, mesh.nverts = n_tris, 3 * n_tris
    mesh.material, mesh.arealight, mesh.instance = 0, -1, -1
    mesh.reverse_orientation = mesh.swaps_handedness = 0
    mesh.o2w, mesh.o2w_inv = _m16(np.eye(4)), _m16(np.eye(4))
    light = constant_infinite_light([1.0, 1.0, 1.0], fpool_parts)
    _, c2w = look_at([0, 0, 3.5], [0, 0, 0], [0, 1, 0])
    cam = perspective_camera(xres, yres, 45.0, c2w)
    s = abi.Scene(meshes=[mesh], materials=[matte([0.5, 0.5, 0.5])], lights=[light],
                  fpool=np.concatenate(fpool_parts), ipool=idx, camera=cam,
                  render=render_desc(xres, yres, spp, maxdepth))
    s.meta = {"fov": 45.0, "look_at": ([0, 0, 3.5], [0, 0, 0], [0, 1, 0])}
    return s


# ---- .pbrt export (for feeding the same scene to the reference binary) ---------------------------
def _fmt(a):
    return " ".join("%.9g" % float(v) for v in np.asarray(a).reshape(-1))


def export_pbrt(scene, path, film_path, spp=None, maxdepth=None, xres=None, yres=None, renderer=None, pixel_filter=None, sampler=None):
    """Write `scene` as a pbrt-v2 scene file.  Supported: triangle meshes (P world space; N are
    exported as world-space normals with an identity object transform), matte / plastic materials,
    point and constant infinite lights, sphere / disk emitters."""
    rd = scene.render
    xres, yres = xres or rd.xres, yres or rd.yres
    spp, maxdepth = spp or rd.spp, (rd.maxdepth if maxdepth is None else maxdepth)
    fov = getattr(scene, "meta", {}).get("fov") or recover_fov(scene.camera, rd.xres, rd.yres)
    c2w = np.array(list(scene.camera.camera_to_world), dtype=np.float64).reshape(4, 4)
    w2c = np.linalg.inv(c2w)
    out = []
    out.append("Transform [%s]" % _fmt(w2c.T))  # pbrt reads column-major (core/api.cpp:733-738)
    out.append('Camera "perspective" "float fov" [%.9g]' % fov)
    if pixel_filter:
        out.append(pixel_filter)                 # e.g. 'PixelFilter "gaussian"'
    out.append('Film "image" "integer xresolution" [%d] "integer yresolution" [%d] "string filename" "%s"'
               % (xres, yres, film_path))
    out.append(sampler or 'Sampler "lowdiscrepancy" "integer pixelsamples" [%d]' % spp)   # sampler: a full Sampler line
    out.append('SurfaceIntegrator "path" "integer maxdepth" [%d]' % maxdepth)
    if renderer:
        out.append('Renderer "%s"' % renderer)
    out.append("WorldBegin")

    def material_lines(m):
        if m.kind == abi.HPT_MAT_MATTE:
            return ['Material "matte" "color Kd" [%s]' % _fmt(list(m.kd))]
        if m.kind == abi.HPT_MAT_PLASTIC:
            return ['Material "plastic" "color Kd" [%s] "color Ks" [%s] "float roughness" [%.9g]'
                    % (_fmt(list(m.kd)), _fmt(list(m.ks)), m.roughness)]
        if m.kind == abi.HPT_MAT_METAL:
            return ['Material "metal" "color eta" [%s] "color k" [%s] "float roughness" [%.9g]'
                    % (_fmt(list(m.eta)), _fmt(list(m.k)), m.roughness)]
        if m.kind == abi.HPT_MAT_SUBSTRATE:
            return ['Material "substrate" "color Kd" [%s] "color Ks" [%s] "float uroughness" [%.9g] "float vroughness" [%.9g]'
                    % (_fmt(list(m.kd)), _fmt(list(m.ks)), m.nu, m.nv)]
        raise ValueError("export_pbrt: material kind %d not exportable" % m.kind)

    for li, l in enumerate(scene.lights):
        if l.kind == abi.HPT_LIGHT_POINT:
            out.append('AttributeBegin\nLightSource "point" "color I" [%s] "point from" [%s]\nAttributeEnd'
                       % (_fmt(list(l.intensity)), _fmt(list(l.pos))))
        elif l.kind == abi.HPT_LIGHT_INFINITE:
            if l.env_w != 1 or l.env_h != 1:
                raise ValueError("export_pbrt: only constant infinite lights are exportable")
            L = scene.fpool[l.tex_off:l.tex_off + 3]
            out.append('AttributeBegin\nLightSource "infinite" "color L" [%s]\nAttributeEnd' % _fmt(L))
    for q in scene.quadrics:
        o2w = np.array(list(q.o2w), dtype=np.float64).reshape(4, 4)
        out.append("AttributeBegin")
        out += material_lines(scene.materials[q.material])
        out.append("ConcatTransform [%s]" % _fmt(o2w.T))
        if q.arealight >= 0:
            out.append('AreaLightSource "area" "color L" [%s]' % _fmt(list(scene.lights[q.arealight].intensity)))
        if q.reverse_orientation:
            out.append("ReverseOrientation")
        if q.kind == abi.HPT_QUADRIC_SPHERE:
            out.append('Shape "sphere" "float radius" [%.9g]' % q.radius)
        else:
            out.append('Shape "disk" "float radius" [%.9g] "float height" [%.9g]' % (q.radius, q.height))
        out.append("AttributeEnd")
    with open(path, "w") as f:
        f.write("\n".join(out) + "\n")
        for m in scene.meshes:
            f.write("AttributeBegin\n")
            f.write("\n".join(material_lines(scene.materials[m.material])) + "\n")
            P = scene.fpool[m.p_off:m.p_off + 3 * m.nverts]
            idx = scene.ipool[m.idx_off:m.idx_off + 3 * m.ntris]
            f.write('Shape "trianglemesh" "integer indices" [')
            f.write(" ".join(map(str, idx.tolist())))
            f.write('] "point P" [')
            f.write(" ".join("%.9g" % v for v in P.tolist()))
            f.write("]")
            if m.n_off >= 0:
                N = scene.fpool[m.n_off:m.n_off + 3 * m.nverts].reshape(-1, 3).astype(np.float64)
                minv = np.array(list(m.o2w_inv), dtype=np.float64).reshape(4, 4)[:3, :3]
                Nw = N @ minv  # n' = mInv^T n  (core/transform.h:230-236)
                f.write(' "normal N" [')
                f.write(" ".join("%.9g" % v for v in Nw.reshape(-1).tolist()))
                f.write("]")
            if m.uv_off >= 0:
                f.write(' "float uv" [')
                f.write(" ".join("%.9g" % v for v in scene.fpool[m.uv_off:m.uv_off + 2 * m.nverts].tolist()))
                f.write("]")
            f.write("\nAttributeEnd\n")
        f.write("WorldEnd\n")
