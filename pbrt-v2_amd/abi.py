"""ctypes mirror of include/hpt.h (the C ABI of the MI355X path-tracing hot path) and the
scene-blob container ("HPTS" files written by hpt_blob_save / host/hip_renderer.cpp).

Pure Python + numpy: usable without the HIP library (the oracle binding in oracle/orc.py and
the CPU-side tests read blobs through this module).  Layouts are checked against the compiled
library by hpt.py (hpt_abi_sizes) so a drift between this file and hpt.h fails loudly.
"""
import ctypes as C
import numpy as np

HPT_MAGIC = 0x53545048
HPT_VERSION = 9

HPT_QUADRIC_SPHERE, HPT_QUADRIC_DISK = 1, 2
HPT_MAT_MATTE, HPT_MAT_PLASTIC, HPT_MAT_MEASURED_IRREG, HPT_MAT_METAL, HPT_MAT_SUBSTRATE = 1, 2, 3, 4, 5
HPT_MAT_GLASS, HPT_MAT_MIRROR, HPT_MAT_MEASURED_REGULAR = 6, 7, 8
HPT_TEX_CONSTANT, HPT_TEX_IMAGEMAP, HPT_TEX_SCALE, HPT_TEX_MIX = 1, 2, 3, 4
HPT_WRAP_REPEAT, HPT_WRAP_BLACK, HPT_WRAP_CLAMP = 0, 1, 2
HPT_MAP_UV, HPT_MAP_SPHERICAL, HPT_MAP_CYLINDRICAL, HPT_MAP_PLANAR = 0, 1, 2, 3
TEXSLOT_KD, TEXSLOT_KS, TEXSLOT_ROUGH, TEXSLOT_ROUGH_V, TEXSLOT_BUMP, TEXSLOT_KT, TEXSLOT_INDEX = 0, 1, 2, 3, 4, 5, 6
HPT_N_TEXSLOTS = 8
HPT_LIGHT_POINT, HPT_LIGHT_DIFFUSE_AREA, HPT_LIGHT_INFINITE, HPT_LIGHT_SPOT, HPT_LIGHT_DISTANT = 1, 2, 3, 4, 5
HPT_SAMPLER_LD_HASH, HPT_SAMPLER_MT_REPLAY, HPT_SAMPLER_RANDOM_HASH, HPT_SAMPLER_RANDOM_MT_REPLAY = 0, 1, 2, 3
HPT_SAMPLER_STRATIFIED_HASH, HPT_SAMPLER_STRATIFIED_MT_REPLAY = 4, 5
HPT_SAMPLER_HALTON_HASH, HPT_SAMPLER_HALTON_MT_REPLAY = 6, 7
HPT_SAMPLER_ADAPTIVE_HASH, HPT_SAMPLER_ADAPTIVE_MT_REPLAY = 8, 9
HPT_SAMPLER_BESTCANDIDATE_HASH, HPT_SAMPLER_BESTCANDIDATE_MT_REPLAY = 10, 11


def sampler_kind(mode):
    return mode & 0x7f


def stratified_mode(kind, xsamples, jitter=True):
    """HPT_SAMPLER_STRATIFIED(kind, xsamples, jitter): the sampler's parameters ride in sampler_mode's upper bits"""
    return kind | (0x80 if jitter else 0) | (xsamples << 8)


def adaptive_mode(kind, minsamples):
    """HPT_SAMPLER_ADAPTIVE(kind, minsamples): spp of the render descriptor = maxsamples"""
    return kind | (minsamples << 8)


HPT_PIPELINE_PERSISTENT, HPT_PIPELINE_WAVEFRONT = 0, 1
HPT_INTEGRATOR_PATH, HPT_INTEGRATOR_DIRECT_ALL, HPT_INTEGRATOR_DIRECT_ONE = 0, 1, 2
SAMPLE_FLOATS = 35  # 5 camera + 12 one-D + 9 two-D pairs

f32, i32, i64, u32, u64 = C.c_float, C.c_int32, C.c_int64, C.c_uint32, C.c_uint64
M16 = f32 * 16


_MESH_V6 = [("p_off", i64), ("n_off", i64), ("uv_off", i64), ("idx_off", i64),
            ("ntris", i32), ("nverts", i32), ("material", i32), ("arealight", i32),
            ("reverse_orientation", i32), ("swaps_handedness", i32),
            ("instance", i32), ("alpha_tex", i32),
            ("o2w", M16), ("o2w_inv", M16)]


class MeshV6(C.Structure):
    _fields_ = _MESH_V6


class Mesh(C.Structure):
    _fields_ = _MESH_V6 + [("s_off", i64)]     # version 7: TriangleMesh::s (explicit tangents), -1 = absent

    def __init__(self, *a, **kw):
        super().__init__(*a, **kw)
        if "s_off" not in kw:
            self.s_off = -1


class Instance(C.Structure):
    _fields_ = [("actually_animated", i32), ("quadric1", i32), ("start_time", f32), ("end_time", f32),
                ("bounds", f32 * 6), ("T", (f32 * 3) * 2), ("R", (f32 * 4) * 2), ("S", M16 * 2),
                ("w2p_m", M16 * 2), ("w2p_minv", M16 * 2)]


class Quadric(C.Structure):
    _fields_ = [("kind", i32), ("material", i32), ("arealight", i32),
                ("reverse_orientation", i32), ("swaps_handedness", i32),
                ("radius", f32), ("zmin", f32), ("zmax", f32), ("theta_min", f32),
                ("theta_max", f32), ("phi_max", f32), ("height", f32), ("inner_radius", f32),
                ("o2w", M16), ("o2w_inv", M16)]


_MATERIAL_V5 = [("kind", i32), ("kd", f32 * 3), ("sigma", f32), ("ks", f32 * 3),
                ("roughness", f32), ("kd_split_off", i64), ("kd_bits_off", i64),
                ("kd_data_off", i64), ("kd_nnodes", i32), ("pad", i32),
                ("eta", f32 * 3), ("k", f32 * 3), ("nu", f32), ("nv", f32)]
_LIGHT_V5 = [("kind", i32), ("quadric", i32), ("pos", f32 * 3), ("intensity", f32 * 3),
             ("area", f32), ("env_w", i32), ("env_h", i32),
             ("tex_off", i64), ("cond_func_off", i64), ("cond_cdf_off", i64),
             ("cond_int_off", i64), ("marg_func_off", i64), ("marg_cdf_off", i64),
             ("marg_int", f32), ("nsamples", i32), ("l2w", M16), ("l2w_inv", M16)]


class MaterialV5(C.Structure):
    _fields_ = _MATERIAL_V5


class LightV5(C.Structure):
    _fields_ = _LIGHT_V5


class Material(C.Structure):
    _fields_ = _MATERIAL_V5 + [("tex", i32 * HPT_N_TEXSLOTS), ("kt", f32 * 3), ("index", f32), ("rh_off", i64),
                               ("rh_n_theta_h", i32), ("rh_n_theta_d", i32), ("rh_n_phi_d", i32), ("pad2", i32)]

    def __init__(self, *a, **kw):
        super().__init__(*a, **kw)
        for k in range(HPT_N_TEXSLOTS):
            self.tex[k] = -1
        self.rh_off = -1


class Light(C.Structure):
    _fields_ = _LIGHT_V5 + [("set_off", i64), ("set_area_off", i64), ("set_n", i32), ("pad", i32)]


_TEXTURE_V8 = [("kind", i32), ("channels", i32), ("value", f32 * 3), ("tex1", i32), ("tex2", i32), ("amount", i32),
               ("pyr_off", i64), ("width", i32), ("height", i32), ("levels", i32), ("wrap", i32), ("do_trilinear", i32),
               ("max_aniso", f32), ("su", f32), ("sv", f32), ("du", f32), ("dv", f32)]


class TextureV8(C.Structure):
    """hpt_texture of versions 6 .. 8: every image map through its UVMapping2D"""
    _fields_ = _TEXTURE_V8


def light_unsampled(l):
    """HPT_LIGHT_UNSAMPLED (include/hpt.h): a DIFFUSE_AREA record without quadric and without shape set — an emitter inside an object instance"""
    return l.kind == HPT_LIGHT_DIFFUSE_AREA and l.quadric < 0 and l.set_n == 0


class Texture(C.Structure):
    """hpt_texture"""
    _fields_ = _TEXTURE_V8 + [("mapping", i32), ("pad9", i32), ("map_m", f32 * 16)]     # version 9: TextureMapping2D (0 = uv)


def _upgrade(old, new_type):
    """a version-5 record -> the version-6 record with the new fields at their 'absent' values"""
    n = new_type()
    C.memmove(C.addressof(n), C.addressof(old), C.sizeof(old))
    if new_type is Light:
        n.set_off = n.set_area_off = -1
        n.set_n = 0
    if new_type is Mesh:
        n.s_off = -1
    return n


class SceneDesc(C.Structure):
    _fields_ = [("meshes", C.POINTER(Mesh)), ("n_meshes", i32),
                ("quadrics", C.POINTER(Quadric)), ("n_quadrics", i32),
                ("materials", C.POINTER(Material)), ("n_materials", i32),
                ("lights", C.POINTER(Light)), ("n_lights", i32),
                ("instances", C.POINTER(Instance)), ("n_instances", i32),
                ("fpool", C.POINTER(f32)), ("n_f", i64),
                ("ipool", C.POINTER(i32)), ("n_i", i64),
                ("textures", C.POINTER(Texture)), ("n_textures", i32)]


class Camera(C.Structure):
    _fields_ = [("raster_to_camera", M16), ("camera_to_world", M16),
                ("lens_radius", f32), ("focal_distance", f32),
                ("shutter_open", f32), ("shutter_close", f32)]


class RenderDesc(C.Structure):
    _fields_ = [("xres", i32), ("yres", i32), ("x_start", i32), ("x_count", i32),
                ("y_start", i32), ("y_count", i32), ("spp", i32), ("maxdepth", i32),
                ("sampler_mode", i32), ("seed", u32), ("ntasks", i32),
                ("shard_rank", i32), ("shard_count", i32), ("count_work", i32),
                ("pipeline", i32), ("integrator", i32)]


class Filter(C.Structure):
    """hpt_filter: Filter::xWidth / yWidth + ImageFilm::filterTable (film/image.cpp:56-68)"""
    _fields_ = [("xwidth", f32), ("ywidth", f32), ("table", f32 * 256)]


def make_filter(kind, xwidth=None, ywidth=None, **kw):
    """hpt_filter for one of the reference's Filter plugins, with the plugin's default parameters
    (filters/{box,gaussian,mitchell,triangle,sinc}.cpp Create*Filter) — what ImageFilm's constructor tabulates
    (film/image.cpp:56-68): table[y][x] = Evaluate((x + .5) * xWidth / 16, (y + .5) * yWidth / 16).
    The pbrt host plugin fills the table from the scene's own Filter object; this mirror serves tests and bench.py."""
    import numpy as np
    defaults = {"box": 0.5, "gaussian": 2.0, "mitchell": 2.0, "triangle": 2.0, "sinc": 4.0}
    xw = np.float32(defaults[kind] if xwidth is None else xwidth)
    yw = np.float32(defaults[kind] if ywidth is None else ywidth)
    one = np.float32(1)
    fx = (np.arange(16, dtype=np.float32) + np.float32(.5)) * xw / np.float32(16)
    fy = (np.arange(16, dtype=np.float32) + np.float32(.5)) * yw / np.float32(16)
    if kind == "box":
        ex, ey = np.ones(16, np.float32), np.ones(16, np.float32)
    elif kind == "gaussian":       # filters/gaussian.h:46-58
        alpha = np.float32(kw.get("alpha", 2.0))
        g = lambda d, w: np.maximum(np.float32(0), np.exp(-alpha * d * d, dtype=np.float32) - np.exp(-alpha * w * w, dtype=np.float32))
        ex, ey = g(fx, xw), g(fy, yw)
    elif kind == "mitchell":       # filters/mitchell.h:46-62
        B = np.float32(kw.get("B", 1.0 / 3.0)); Cc = np.float32(kw.get("C", 1.0 / 3.0))
        def m1(x):
            x = np.abs(np.float32(2) * x)
            far = ((-B - 6 * Cc) * x * x * x + (6 * B + 30 * Cc) * x * x + (-12 * B - 48 * Cc) * x + (8 * B + 24 * Cc)) * np.float32(1.0 / 6.0)
            near = ((12 - 9 * B - 6 * Cc) * x * x * x + (-18 + 12 * B + 6 * Cc) * x * x + (6 - 2 * B)) * np.float32(1.0 / 6.0)
            return np.where(x > 1, far, near).astype(np.float32)
        ex, ey = m1(fx * (one / xw)), m1(fy * (one / yw))
    elif kind == "triangle":       # filters/triangle.cpp:40-43
        ex, ey = np.maximum(np.float32(0), xw - np.abs(fx)), np.maximum(np.float32(0), yw - np.abs(fy))
    elif kind == "sinc":           # filters/sinc.h:49-57 (Lanczos-windowed sinc)
        tau = np.float32(kw.get("tau", 3.0))
        def s1(x):
            x = np.abs(x)
            xs = x * np.float32(np.pi)
            sinc = np.where(x < 1e-5, one, np.sin(xs * tau) / np.where(xs * tau == 0, one, xs * tau))
            lanc = np.where(x < 1e-5, one, np.sin(xs) / np.where(xs == 0, one, xs))
            return np.where(x > 1, np.float32(0), sinc * lanc).astype(np.float32)
        ex, ey = s1(fx * (one / xw)), s1(fy * (one / yw))
    else:
        raise ValueError(kind)
    f = Filter()
    f.xwidth, f.ywidth = float(xw), float(yw)
    t = (ey[:, None].astype(np.float32) * ex[None, :].astype(np.float32)).astype(np.float32)
    C.memmove(f.table, t.ctypes.data, 1024)
    return f


def filter_from_array(a):
    """258 floats {xwidth, ywidth, table[256]} (the .filter sidecar the host plugin dumps) -> Filter"""
    import numpy as np
    a = np.ascontiguousarray(a, dtype=np.float32)
    assert a.size == 258
    f = Filter()
    C.memmove(C.addressof(f), a.ctypes.data, 258 * 4)
    return f


def sample_extent(rd, flt):
    """ImageFilm::GetSampleExtent (film/image.cpp:157-166) -> (xs, xe, ys, ye)"""
    import numpy as np
    fl = lambda v: int(np.floor(np.float32(v)))
    ce = lambda v: int(np.ceil(np.float32(v)))
    xw = np.float32(flt.xwidth if flt is not None else 0.5); yw = np.float32(flt.ywidth if flt is not None else 0.5)
    h = np.float32(0.5)
    return (fl(np.float32(rd.x_start) + h - xw), ce(np.float32(rd.x_start) - h + np.float32(rd.x_count) + xw),
            fl(np.float32(rd.y_start) + h - yw), ce(np.float32(rd.y_start) - h + np.float32(rd.y_count) + yw))


class Stats(C.Structure):
    _fields_ = [("kernel_ms", C.c_double), ("camera_samples", u64), ("closest_rays", u64),
                ("shadow_rays", u64), ("nodes_visited", u64), ("tris_tested", u64),
                ("bad_samples", u64), ("resident_waves", u32), ("grid_blocks", u32),
                ("block_threads", u32), ("vgprs", u32),
                ("tune_cfg", u32), ("scratch_bytes", u32)]


class SceneInfo(C.Structure):
    _fields_ = [("n_tris", i64), ("n_bvh_nodes", i64), ("n_quadrics", i64),
                ("bvh_bytes", i64), ("tri_bytes", i64), ("total_device_bytes", i64),
                ("bvh_max_depth", i32), ("pad", i32), ("build_ms", C.c_double),
                ("device_build_ms", C.c_double), ("device_built", i32), ("pad2", i32)]


class BlobHeader(C.Structure):
    _fields_ = [("magic", u32), ("version", u32), ("n_meshes", i32), ("n_quadrics", i32),
                ("n_materials", i32), ("n_lights", i32), ("n_instances", i32), ("pad", i32),
                ("n_f", i64), ("n_i", i64),
                ("cam", Camera), ("rd", RenderDesc),
                ("sizeof_mesh", u32), ("sizeof_quadric", u32), ("sizeof_material", u32),
                ("sizeof_light", u32), ("sizeof_instance", u32), ("n_textures", u32)]   # n_textures: version 6 (the word was padding before)


ABI_SIZES = [C.sizeof(Mesh), C.sizeof(Quadric), C.sizeof(Material), C.sizeof(Light),
             C.sizeof(Camera), C.sizeof(RenderDesc), C.sizeof(Stats), C.sizeof(BlobHeader),
             C.sizeof(Instance), C.sizeof(Texture)]


def lights_from_bytes(raw, n):
    """the `lights` array of a *.view.npz fixture -> (Light * n); round-1 fixtures hold version-5 records"""
    raw = bytes(raw)
    if len(raw) == n * C.sizeof(Light):
        return (Light * n).from_buffer_copy(raw)
    assert len(raw) == n * C.sizeof(LightV5), "light records of unknown size"
    old = (LightV5 * n).from_buffer_copy(raw)
    out = (Light * max(n, 0))()
    for i in range(n):
        out[i] = _upgrade(old[i], Light)
    return out


def _arr(ctype, n):
    return (ctype * max(int(n), 0))()


class Scene:
    """A flattened pbrt scene held in host memory: record arrays + float/int pools, plus the
    camera and the render defaults it was dumped with.  `.desc` is the hpt_scene_desc view."""

    def __init__(self, meshes=(), quadrics=(), materials=(), lights=(), fpool=None, ipool=None,
                 camera=None, render=None, instances=(), textures=()):
        self.textures = _arr(Texture, len(textures))
        for i, t in enumerate(textures):
            self.textures[i] = t
        self.meshes = _arr(Mesh, len(meshes))
        for i, m in enumerate(meshes):
            self.meshes[i] = m
        self.quadrics = _arr(Quadric, len(quadrics))
        for i, q in enumerate(quadrics):
            self.quadrics[i] = q
        self.materials = _arr(Material, len(materials))
        for i, m in enumerate(materials):
            self.materials[i] = m
        self.lights = _arr(Light, len(lights))
        for i, l in enumerate(lights):
            self.lights[i] = l
        self.instances = _arr(Instance, len(instances))
        for i, l in enumerate(instances):
            self.instances[i] = l
        self.fpool = np.ascontiguousarray(fpool if fpool is not None else np.zeros(0), dtype=np.float32)
        self.ipool = np.ascontiguousarray(ipool if ipool is not None else np.zeros(0), dtype=np.int32)
        self.camera = camera if camera is not None else Camera()
        self.render = render if render is not None else RenderDesc()

    @property
    def desc(self):
        d = SceneDesc()
        d.meshes = C.cast(self.meshes, C.POINTER(Mesh)); d.n_meshes = len(self.meshes)
        d.quadrics = C.cast(self.quadrics, C.POINTER(Quadric)); d.n_quadrics = len(self.quadrics)
        d.materials = C.cast(self.materials, C.POINTER(Material)); d.n_materials = len(self.materials)
        d.lights = C.cast(self.lights, C.POINTER(Light)); d.n_lights = len(self.lights)
        d.instances = C.cast(self.instances, C.POINTER(Instance)); d.n_instances = len(self.instances)
        d.fpool = self.fpool.ctypes.data_as(C.POINTER(f32)); d.n_f = self.fpool.size
        d.ipool = self.ipool.ctypes.data_as(C.POINTER(i32)); d.n_i = self.ipool.size
        d.textures = C.cast(self.textures, C.POINTER(Texture)); d.n_textures = len(self.textures)
        return d

    @property
    def n_tris(self):
        return sum(m.ntris for m in self.meshes)

    # ---- blob I/O (same layout as csrc/hpt_blob.cpp) ------------------------------------
    def save(self, path):
        h = BlobHeader()
        h.magic, h.version = HPT_MAGIC, HPT_VERSION
        h.n_meshes, h.n_quadrics = len(self.meshes), len(self.quadrics)
        h.n_materials, h.n_lights = len(self.materials), len(self.lights)
        h.n_instances = len(self.instances)
        h.sizeof_instance = C.sizeof(Instance)
        h.n_f, h.n_i = self.fpool.size, self.ipool.size
        h.cam, h.rd = self.camera, self.render
        h.sizeof_mesh, h.sizeof_quadric = C.sizeof(Mesh), C.sizeof(Quadric)
        h.sizeof_material, h.sizeof_light = C.sizeof(Material), C.sizeof(Light)
        h.n_textures = len(self.textures)
        with _open(path, "wb") as f:
            f.write(bytes(h))
            for a in (self.meshes, self.quadrics, self.materials, self.lights, self.instances, self.textures):
                f.write(bytes(a))
            f.write(self.fpool.tobytes())
            f.write(self.ipool.tobytes())

    @staticmethod
    def load(path):
        with _open(path, "rb") as f:
            raw = f.read()
        h = BlobHeader.from_buffer_copy(raw[:C.sizeof(BlobHeader)])
        if h.magic != HPT_MAGIC or h.version not in (5, 6, 7, 8, HPT_VERSION):
            raise ValueError(f"{path}: not an HPTS v5 .. v{HPT_VERSION} blob")
        v5 = h.version == 5          # round-1 fixtures: smaller material / light records, no texture table
        mat_t, light_t = (MaterialV5, LightV5) if v5 else (Material, Light)
        mesh_t = MeshV6 if h.version < 7 else Mesh       # versions 5 / 6: mesh records without s_off
        if (h.sizeof_mesh, h.sizeof_quadric, h.sizeof_material, h.sizeof_light, h.sizeof_instance) != \
                (C.sizeof(mesh_t), C.sizeof(Quadric), C.sizeof(mat_t), C.sizeof(light_t), C.sizeof(Instance)):
            raise ValueError(f"{path}: record sizes differ from this build of the ABI")
        off = C.sizeof(BlobHeader)

        def take(ctype, n):
            nonlocal off
            nbytes = C.sizeof(ctype) * n
            a = (ctype * n).from_buffer_copy(raw[off:off + nbytes])
            off += nbytes
            return a
        s = Scene()
        s.meshes = take(mesh_t, h.n_meshes)
        if mesh_t is MeshV6:
            up = _arr(Mesh, h.n_meshes)
            for i in range(h.n_meshes):
                up[i] = _upgrade(s.meshes[i], Mesh)
            s.meshes = up
        s.quadrics = take(Quadric, h.n_quadrics)
        s.materials = take(mat_t, h.n_materials)
        s.lights = take(light_t, h.n_lights)
        s.instances = take(Instance, h.n_instances)
        if v5:
            mats, lights = _arr(Material, h.n_materials), _arr(Light, h.n_lights)
            for i in range(h.n_materials):
                mats[i] = _upgrade(s.materials[i], Material)
            for i in range(h.n_lights):
                lights[i] = _upgrade(s.lights[i], Light)
            s.materials, s.lights = mats, lights
            for m in s.meshes:
                m.alpha_tex = 0
        elif h.version < 9:           # texture records without the 2D mapping block
            old = take(TextureV8, h.n_textures)
            s.textures = _arr(Texture, h.n_textures)
            for i in range(h.n_textures):
                s.textures[i] = _upgrade(old[i], Texture)
        else:
            s.textures = take(Texture, h.n_textures)
        s.fpool = np.frombuffer(raw, dtype=np.float32, count=h.n_f, offset=off).copy(); off += 4 * h.n_f
        s.ipool = np.frombuffer(raw, dtype=np.int32, count=h.n_i, offset=off).copy(); off += 4 * h.n_i
        s.camera, s.render = h.cam, h.rd
        return s


def _open(path, mode):
    """Blobs committed as fixtures are gzip-compressed (*.hpts.gz)."""
    if str(path).endswith(".gz"):
        import gzip
        if "w" in mode:       # no timestamp / file name in the header: regenerating a fixture reproduces its bytes
            return gzip.GzipFile(filename="", fileobj=open(path, "wb"), mode="wb", mtime=0)
        return gzip.open(path, mode)
    return open(path, mode)


def copy_struct(s):
    return type(s).from_buffer_copy(bytes(s))
