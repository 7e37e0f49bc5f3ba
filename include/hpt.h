/* hpt.h — C ABI of the MI355X path-tracing hot path ("hpt" = HIP path tracer).
 *
 * This is the ONLY boundary between pbrt-v2's host side and the device code.  It replaces,
 * behind pbrt's own Renderer plugin surface, the reference's
 *
 *     Renderer::Render(const Scene *)                 core/renderer.h:43-54
 *       = SamplerRenderer::Render                     renderers/samplerrenderer.cpp:188-222
 *       + SamplerRendererTask::Run                    renderers/samplerrenderer.cpp:60-164
 *       + SamplerRenderer::Li / PathIntegrator::Li    samplerrenderer.cpp:225-247, integrators/path.cpp:52-123
 *
 * The reference has no FFI of its own (plugins are statically linked C++ classes,
 * README.txt:55-61); the binding a maintainer adds is one Renderer subclass
 * (host/hip_renderer.cpp, shown in INTEGRATION.md) that fills the POD descriptors below from
 * the Scene/Camera/Sampler objects pbrt already built and calls hpt_scene_create / hpt_render.
 *
 * Conventions
 *  - plain C, POD only, caller owns every input buffer, the library copies during *_create;
 *  - every call returns 0 on success or a negative HPT_E_* code; hpt_last_error() gives text
 *    (the host wrapper maps it to pbrt's Error()/Severe(), core/error.h:49-52);
 *  - matrices are row-major float[16], exactly Matrix4x4::m (core/transform.h);
 *  - there is NO CPU fallback: without a HIP device every compute entry point fails with
 *    HPT_E_NODEVICE.
 */
#ifndef HPT_H
#define HPT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HPT_MAGIC   0x53545048u /* "HPTS" little endian */
#define HPT_VERSION 9   /* version 9: hpt_texture grew by the 2D mapping (spherical / cylindrical / planar); blobs of version 5 (round 1: no textures, no specular / regular-halfangle materials, no shape-set lights), 6 (no mesh tangents) and 7 (no animated spheres / disks: the same records, hpt_instance.quadric1 was padding) still load */

enum {
    HPT_OK = 0,
    HPT_E_NODEVICE = -1,   /* no HIP device / HIP runtime error */
    HPT_E_INVALID = -2,    /* malformed descriptor */
    HPT_E_UNSUPPORTED = -3,/* scene uses a feature outside the hot-path scope */
    HPT_E_IO = -4,
    HPT_E_HIP = -5,
    HPT_E_INTERNAL = -6    /* the library caught itself: the camera samples a frame's kernels completed are not the job's (every sample reaches the
                              film exactly once, renderers/samplerrenderer.cpp:60-164), or a check of the debug build failed.  The film is not to be used. */
};

/* ---- scene description (flattened pbrt Scene) ------------------------------------------ */

/* Shape kinds other than triangles.  shapes/sphere.cpp, shapes/disk.cpp */
enum { HPT_QUADRIC_SPHERE = 1, HPT_QUADRIC_DISK = 2 };

/* Material kinds: materials/matte.cpp:42, plastic.cpp:42, measured.cpp:194 (IrregIsotropicBRDF),
 * metal.cpp:51 (Microfacet + FresnelConductor + Blinn), substrate.cpp:42 (FresnelBlend + Anisotropic) */
enum { HPT_MAT_MATTE = 1, HPT_MAT_PLASTIC = 2, HPT_MAT_MEASURED_IRREG = 3, HPT_MAT_METAL = 4, HPT_MAT_SUBSTRATE = 5,
       /* materials/glass.cpp:43-61 (SpecularReflection + SpecularTransmission, FresnelDielectric(1, index)),
        * materials/mirror.cpp:43-57 (SpecularReflection, FresnelNoOp),
        * materials/measured.cpp:194-210 with a RegularHalfangleBRDF (MERL .binary data, core/reflection.cpp:275-300) */
       HPT_MAT_GLASS = 6, HPT_MAT_MIRROR = 7, HPT_MAT_MEASURED_REGULAR = 8 };

/* Textures (core/texture.h:44-96).  A material parameter is either the constant in its hpt_material field or —
 * hpt_material.tex[slot] >= 0 — a texture of this table, evaluated at the hit's DifferentialGeometry like Texture::Evaluate.
 *   CONSTANT  textures/constant.h:45-55                 value
 *   IMAGEMAP  textures/imagemap.cpp:41-101              MIPMap::Lookup (core/mipmap.h:238-366: trilinear or EWA) of the pyramid the
 *             reference built (MIPMap ctor, mipmap.h:105-190), through the texture's TextureMapping2D (version 9: all four of
 *             core/texture.cpp:88-164 — "uv", "spherical", "cylindrical", "planar", textures/imagemap.cpp:106-124)
 *   SCALE     textures/scale.h:46-60                    tex1 * tex2
 *   MIX       textures/mix.h:46-62                      (1 - amount) * tex1 + amount * tex2
 * SCALE / MIX operands precede the texture in the table; nesting up to 12 levels (hpt_scene_create refuses deeper tables).
 * Other texture plugins are outside the hot-path scope (the host wrapper refuses them). */
enum { HPT_TEX_CONSTANT = 1, HPT_TEX_IMAGEMAP = 2, HPT_TEX_SCALE = 3, HPT_TEX_MIX = 4 };
/* TextureMapping2D (core/texture.h:47-113).  UV: (su * u + du, sv * v + dv).  SPHERICAL / CYLINDRICAL: the hit point through
 * WorldToTexture (map_m, row-major 4 x 4), then (theta / pi, phi / 2 pi) resp. ((pi + atan2(y, x)) / 2 pi, z) of the normalized vector, with
 * finite-difference derivatives over dpdx / dpdy (delta .1 resp. .01, core/texture.cpp:102-149).  PLANAR: (ds + p . vs, dt + p . vt) with
 * map_m = {vs.x, vs.y, vs.z, vt.x, vt.y, vt.z, ds, dt} (core/texture.cpp:152-162). */
enum { HPT_MAP_UV = 0, HPT_MAP_SPHERICAL = 1, HPT_MAP_CYLINDRICAL = 2, HPT_MAP_PLANAR = 3 };
enum { HPT_WRAP_REPEAT = 0, HPT_WRAP_BLACK = 1, HPT_WRAP_CLAMP = 2 };   /* ImageWrap, core/mipmap.h:47-49 */
typedef struct hpt_texture {
    int32_t kind;
    int32_t channels;      /* 1: Texture<float>, 3: Texture<Spectrum> (RGB) */
    float value[3];        /* CONSTANT */
    int32_t tex1, tex2;    /* SCALE / MIX operands (texture indices) */
    int32_t amount;        /* MIX: float texture index */
    /* IMAGEMAP: the MIPMap's pyramid, level after level in fpool: level l is max(1, width >> l) x max(1, height >> l) texels of
     * `channels` floats, row-major (t * w + s), starting right after level l - 1; level 0 at pyr_off */
    int64_t pyr_off;
    int32_t width, height, levels;   /* level-0 resolution (a power of two after the reference's resampling), MIPMap::nLevels */
    int32_t wrap;          /* HPT_WRAP_* */
    int32_t do_trilinear;  /* MIPMap::doTrilinear */
    float max_aniso;       /* MIPMap::maxAnisotropy */
    float su, sv, du, dv;  /* UVMapping2D */
    /* ---- version 9 ---- */
    int32_t mapping;       /* HPT_MAP_* (0 = the UVMapping2D above: what every older blob holds) */
    int32_t pad9;
    float map_m[16];       /* SPHERICAL / CYLINDRICAL: WorldToTexture; PLANAR: vs, vt, ds, dt */
} hpt_texture;
/* hpt_material.tex[] slots */
enum { HPT_TEXSLOT_KD = 0,        /* matte / plastic / substrate Kd; metal eta                 (spectrum) */
       HPT_TEXSLOT_KS = 1,        /* plastic / substrate Ks; metal k; glass / mirror Kr        (spectrum) */
       HPT_TEXSLOT_ROUGH = 2,     /* plastic / metal roughness; matte sigma; substrate uroughness (float) */
       HPT_TEXSLOT_ROUGH_V = 3,   /* substrate vroughness                                          (float) */
       HPT_TEXSLOT_BUMP = 4,      /* Material::Bump displacement (core/material.cpp:46-85)          (float) */
       HPT_TEXSLOT_KT = 5,        /* glass Kt                                                   (spectrum) */
       HPT_TEXSLOT_INDEX = 6,     /* glass index                                                    (float) */
       HPT_N_TEXSLOTS = 8 };

/* Light kinds: lights/point.cpp:50, lights/diffuse.cpp:69, lights/infinite.cpp:68 */
enum { HPT_LIGHT_POINT = 1, HPT_LIGHT_DIFFUSE_AREA = 2, HPT_LIGHT_INFINITE = 3,
       HPT_LIGHT_SPOT = 4, HPT_LIGHT_DISTANT = 5 };   /* version 8: lights/spot.cpp, lights/distant.cpp (delta lights like POINT) */

#define HPT_LIGHT_UNSAMPLED(l) ((l).kind == HPT_LIGHT_DIFFUSE_AREA && (l).quadric < 0 && (l).set_n == 0)

/* One TriangleMesh (shapes/trianglemesh.cpp:42).  Offsets index the float / int pools of the
 * scene descriptor, in ELEMENTS; -1 = attribute absent.
 *   P   : nverts*3 floats, WORLD space (the reference transforms at construction, :70-71)
 *   N   : nverts*3 floats, OBJECT space (transformed per hit, trianglemesh.cpp:323-325)
 *   uv  : nverts*2 floats
 *   S   : nverts*3 floats, OBJECT space: TriangleMesh::s, the explicit tangents "vector S" (version 7; Triangle::GetShadingGeometry takes
 *         the shading tangent from them instead of dpdu, shapes/trianglemesh.cpp:326-329)
 *   idx : ntris*3 ints
 * All triangles of a mesh share material and area light (GeometricPrimitive::Refine,
 * core/primitive.cpp:147-157). */
typedef struct hpt_mesh {
    int64_t p_off, n_off, uv_off, idx_off;
    int32_t ntris, nverts;
    int32_t material;            /* index into materials */
    int32_t arealight;           /* index into lights, or -1; a mesh that emits is also listed in its light's shape set */
    int32_t reverse_orientation; /* Shape::ReverseOrientation */
    int32_t swaps_handedness;    /* Shape::TransformSwapsHandedness */
    int32_t instance;            /* index into instances, or -1: mesh lives directly in the world */
    int32_t alpha_tex;           /* 1 + index of the float texture TriangleMesh::alphaTexture (shapes/trianglemesh.cpp:191-195), 0 = none */
    float o2w[16];               /* ObjectToWorld->m    */
    float o2w_inv[16];           /* ObjectToWorld->mInv */
    int64_t s_off;               /* version 7: TriangleMesh::s in fpool, or -1 */
} hpt_mesh;

/* One animated instance: TransformedPrimitive(BVHAccel(refined shape), AnimatedTransform)
 * (core/primitive.h:104-125, created by pbrtShape for an animated CTM, core/api.cpp:1012-1044).
 * Its meshes are built with identity ObjectToWorld (api.cpp:1019-1021), i.e. their P are in the
 * instance's own space; rays reach them through WorldToPrimitive interpolated at the ray's time
 * (AnimatedTransform::Interpolate, core/transform.cpp:371-396).
 * Version 8: an animated SPHERE or DISK.  Such a shape CanIntersect(), so pbrtShape neither refines it nor builds a BVHAccel: the
 * TransformedPrimitive holds the bare GeometricPrimitive (api.cpp:1032-1042) — quadric1 names its hpt_quadric record (identity o2w, no area
 * light: api.cpp:1014-1021).  TransformedPrimitive::Intersect carries the whole differential geometry back to the world — p, nn, dpdu,
 * dpdv and, unlike a triangle's zeros, dndu / dndv (core/primitive.cpp:104-117) — so textures and bump maps on it see the moving frame. */
typedef struct hpt_instance {
    int32_t actually_animated;   /* AnimatedTransform::actuallyAnimated               */
    int32_t quadric1;            /* version 8: the instance's primitive is ONE sphere / disk (a shape that CanIntersect() stays a bare
                                  * GeometricPrimitive under the TransformedPrimitive, core/api.cpp:1032-1042): 1 + its index in
                                  * `quadrics`; that record has identity o2w (api.cpp:1019-1021) and no area light (api.cpp:1014-1016),
                                  * and is not a primitive of the world.  0: the meshes with hpt_mesh.instance == this index.
                                  * < 0 — object instancing (pbrtObjectInstance, core/api.cpp:1114-1147: several TransformedPrimitives over ONE
                                  * aggregate): this instance shares the primitive of instance -quadric1 - 1, which owns it (its meshes name the
                                  * owner).  An instanced mesh keeps its own ObjectToWorld (the CTM at its Shape statement): its P are in the
                                  * instance's space, and Intersection::ObjectToWorld = Inverse(WorldToObject * w2p) (primitive.cpp:104-107)        */
    float start_time, end_time;  /* RenderOptions::transformStartTime / EndTime       */
    float bounds[6];             /* TransformedPrimitive::WorldBound() (MotionBounds) */
    float T[2][3];               /* AnimatedTransform::Decompose of start / end       */
    float R[2][4];               /* quaternion (v.xyz, w)                             */
    float S[2][16];
    float w2p_m[2][16];          /* start / end WorldToPrimitive ->m                  */
    float w2p_minv[2][16];       /*                               ->mInv              */
} hpt_instance;

/* Sphere (shapes/sphere.cpp:40-49) or Disk (shapes/disk.cpp:40-47). */
typedef struct hpt_quadric {
    int32_t kind;
    int32_t material;
    int32_t arealight;
    int32_t reverse_orientation;
    int32_t swaps_handedness;
    float radius;
    float zmin, zmax, theta_min, theta_max; /* sphere */
    float phi_max;
    float height, inner_radius;             /* disk */
    float o2w[16];
    float o2w_inv[16];
} hpt_quadric;

/* Material parameter record: constants, with textures by reference (tex[]). */
typedef struct hpt_material {
    int32_t kind;
    float kd[3];      /* matte / plastic: Kd (already Clamp()ed as in matte.cpp:53) */
    float sigma;      /* matte: Oren-Nayar sigma in degrees, already Clamp()ed to [0, 90] (matte.cpp:54); 0 -> Lambertian */
    float ks[3];      /* plastic                                                      */
    float roughness;  /* plastic: Blinn exponent = 1/roughness (plastic.cpp:60)       */
    /* measured (IrregIsotropicBRDF, core/reflection.cpp:259): KdTree<IrregIsotropicBRDFSample>
     * copied node for node (core/kdtree.h:44-61):
     *   fpool[kd_split_off + i]      = nodes[i].splitPos
     *   ipool[kd_bits_off + i]       = splitAxis | hasLeftChild<<2 | rightChild<<3
     *   fpool[kd_data_off + 6*i ..]  = nodeData[i].p.xyz, nodeData[i].v.rgb           */
    int64_t kd_split_off, kd_bits_off, kd_data_off;
    int32_t kd_nnodes;
    int32_t pad;
    float eta[3], k[3]; /* metal: conductor index / absorption as RGB (SPDs are converted at load, spectrum.h:428);
                           roughness above = Blinn roughness                                                  */
    float nu, nv;       /* substrate: uroughness, vroughness (Anisotropic exponents 1/nu, 1/nv); kd, ks above */
    /* ---- version 6 ---- */
    int32_t tex[HPT_N_TEXSLOTS];   /* texture index per parameter slot (HPT_TEXSLOT_*), -1 = the constant in the field above */
    float kt[3];        /* glass: Kt (Kr in ks) */
    float index;        /* glass: index of refraction */
    /* measured, RegularHalfangleBRDF (core/reflection.cpp:275-300): regularHalfangleData as MeasuredMaterial read it
     * (materials/measured.cpp:137-184), 3 * nThetaH * nThetaD * nPhiD floats at fpool[rh_off] */
    int64_t rh_off;
    int32_t rh_n_theta_h, rh_n_theta_d, rh_n_phi_d, pad2;
} hpt_material;

typedef struct hpt_light {
    int32_t kind;
    int32_t quadric;   /* DIFFUSE_AREA: emitting quadric index (ShapeSet of one quadric), or -1: the shape set below */
    float pos[3];      /* POINT, SPOT: world-space position lightPos (point.cpp:43, spot.cpp:43); DISTANT: lightDir, normalized, world space (distant.cpp:43) */
    float intensity[3];/* POINT, SPOT: I ; DIFFUSE_AREA: Lemit ; DISTANT: L ; INFINITE: unused */
    float area;        /* DIFFUSE_AREA: ShapeSet::Area() ; SPOT: cosTotalWidth (spot.cpp:45) */
    /* INFINITE (lights/infinite.cpp:68-105): level-0 texels of the (pow2-resampled) radiance
     * MIPMap and the Distribution2D tables, copied from the reference objects:
     *   fpool[tex_off + 3*(v*w+u)]   RGB
     *   cond_func h*w, cond_cdf h*(w+1), cond_int h, marg_func h, marg_cdf h+1, marg_int 1 */
    int32_t env_w, env_h;
    int64_t tex_off, cond_func_off, cond_cdf_off, cond_int_off, marg_func_off, marg_cdf_off;
    float marg_int;    /* (SPOT: cosFalloffStart, spot.cpp:46; WorldToLight of SpotLight::Falloff is l2w_inv below) */
    int32_t nsamples;  /* Light::nSamples (core/light.h:60): light samples per camera sample of the
                        * direct-lighting integrator, strategy "all"; 0 reads as 1; ignored by the path integrator */
    float l2w[16];     /* LightToWorld->m    */
    float l2w_inv[16]; /* LightToWorld->mInv */
    /* ---- version 6: DIFFUSE_AREA over a ShapeSet of several shapes (core/light.cpp:114-171; quadric == -1) ----
     * ipool[set_off + 2 i] = shape kind (0: triangle, 1: quadric), ipool[set_off + 2 i + 1] = GLOBAL triangle number (meshes in
     * descriptor order, triangles in mesh order) or quadric index; fpool[set_area_off ..] = areas[n] (Shape::Area of every shape,
     * = Distribution1D::func), cdf[n + 1], then funcInt — ShapeSet::areas / areaDistribution as the reference built them;
     * `area` above is ShapeSet::sumArea */
    int64_t set_off, set_area_off;
    /* set_n == 0 (with quadric == -1): an UNSAMPLED emitter — the area light of a shape inside an object instance, which pbrtShape leaves out of
     * Scene::lights (core/api.cpp:1046-1049: "Area lights not supported with object instancing"): no integrator samples or counts it, but a camera ray
     * or a specular bounce that hits one of its shapes adds Intersection::Le (core/intersection.cpp:54-57) = DiffuseAreaLight::L (`intensity`).  Such
     * records stand AFTER every light of Scene::lights in the table; the meshes that emit name them in hpt_mesh.arealight. */
    int32_t set_n, pad;
} hpt_light;

typedef struct hpt_scene_desc {
    const hpt_mesh *meshes;        int32_t n_meshes;
    const hpt_quadric *quadrics;   int32_t n_quadrics;
    const hpt_material *materials; int32_t n_materials;
    const hpt_light *lights;       int32_t n_lights;   /* order = Scene::lights */
    const hpt_instance *instances; int32_t n_instances;
    const float *fpool;            int64_t n_f;
    const int32_t *ipool;          int64_t n_i;
    const hpt_texture *textures;   int32_t n_textures;   /* version 6 */
} hpt_scene_desc;

/* PerspectiveCamera (cameras/perspective.cpp:41-49,81-138; core/camera.cpp:83-102). */
typedef struct hpt_camera {
    float raster_to_camera[16]; /* RasterToCamera.m */
    float camera_to_world[16];  /* CameraToWorld start transform .m (static camera) */
    float lens_radius, focal_distance;
    float shutter_open, shutter_close;
} hpt_camera;

/* Sampler modes.
 *  HPT_SAMPLER_LD_HASH   : production.  Same (0,2)-sequence sample layout as LDSampler +
 *                          PathIntegrator::RequestSamples (path.cpp:41-49), scrambles and
 *                          per-array permutations from a stateless hash of (pixel, seed) so every
 *                          sample is O(1) computable on any lane (no sample buffer in HBM).
 *  HPT_SAMPLER_MT_REPLAY : parity tool.  Replays the reference's own random stream: one
 *                          MT19937 per image tile seeded with the task number
 *                          (samplerrenderer.cpp:73), LDPixelSample (montecarlo.cpp:200-252),
 *                          rng draws for bounces >= 3 and Russian roulette in reference order.
 *                          One lane per tile, serial inside the tile: slow, bit-for-bit sequence. */
enum { HPT_SAMPLER_LD_HASH = 0, HPT_SAMPLER_MT_REPLAY = 1,
       /* Sampler "random" (samplers/random.cpp): every sample value an independent uniform draw, any
        * spp >= 1 (RandomSampler::RoundSize is the identity, so light sample counts are not rounded either).
        * RANDOM_HASH: production — value = 24-bit uniform from a stateless hash of (pixel, seed, sample, array, index),
        * RandomFloat()'s resolution (core/rng.cpp:59-65).  RANDOM_MT_REPLAY: the reference's own stream (the tile's
        * MT19937 + the sub-sampler's constructor generator, random.cpp:39-60) — oracle only, the device refuses it. */
       HPT_SAMPLER_RANDOM_HASH = 2, HPT_SAMPLER_RANDOM_MT_REPLAY = 3,
       /* Sampler "stratified" (samplers/stratified.cpp): xsamples x ysamples jittered strata per pixel for
        * the image and lens samples (lens and time shuffled), Latin-hypercube arrays for the integrators; spp = xs * ys,
        * counts not rounded.  The sampler's parameters ride in the upper bits of sampler_mode (the descriptor's layout is
        * frozen by the scene blobs): HPT_SAMPLER_STRATIFIED(kind, xsamples, jitter).  STRATIFIED_HASH: production,
        * stateless; STRATIFIED_MT_REPLAY: the reference's stream — oracle only. */
       HPT_SAMPLER_STRATIFIED_HASH = 4, HPT_SAMPLER_STRATIFIED_MT_REPLAY = 5,
       /* Sampler "halton" (samplers/halton.cpp:54-80; SURVEY.md §8f-4 tail): the image positions of a sampler window are the Halton points
        * (radical inverses in bases 3 and 2) of sample numbers 0 .. spp * delta^2 - 1 scaled over the window's delta x delta square, points
        * outside the window rejected; lens (bases 5, 7) and time (base 11) from the sample number + 1; the integrators' arrays one Latin
        * hypercube per camera sample (counts not rounded).  Samples belong to a WINDOW, not to a pixel.  HALTON_MT_REPLAY: the reference's
        * windows (Sampler::ComputeSubWindow over ntasks) and the tile generator's stream — oracle only.  HALTON_HASH: production — the
        * windows are the 32 x 32 pixel super-tiles of the sample extent (the unit the work queue and the multi-GPU shards already use), the
        * array values the stateless hash of (tile, seed, sample number, array, index) of the stratified mode's Latin hypercubes; any spp. */
       HPT_SAMPLER_HALTON_HASH = 6, HPT_SAMPLER_HALTON_MT_REPLAY = 7,
       /* Sampler "adaptive" (samplers/adaptive.cpp; SURVEY.md §8f-4 tail), method "contrast": a pixel is rendered with minsamples LDPixelSample
        * samples; if any sample's luminance differs from the batch's mean by more than half of it (needsSupersampling, adaptive.cpp:149-160)
        * the batch is DISCARDED and the pixel rendered again with maxsamples samples (ReportResults, adaptive.cpp:111-137).  hpt_render_desc.spp
        * = maxsamples (a power of two, the sampler's samplesPerPixel: it scales the ray differentials), minsamples (a power of two, >= 2,
        * < spp) rides in the upper bits of sampler_mode: HPT_SAMPLER_ADAPTIVE(kind, minsamples).  ADAPTIVE_HASH: production — both batches
        * are the LD_HASH sampler's patterns of the pixel for minsamples / maxsamples; ADAPTIVE_MT_REPLAY: the reference's stream — oracle
        * only.  Method "shapeid" (a comparison of Intersection ids the device does not carry) is refused. */
       HPT_SAMPLER_ADAPTIVE_HASH = 8, HPT_SAMPLER_ADAPTIVE_MT_REPLAY = 9,
       /* Sampler "bestcandidate" (samplers/bestcandidate.cpp:50-91; SURVEY.md §8f-4 tail): the reference's precomputed 64 x 64 table of (image x,
        * y, time, lens u, v) points, tiled over the image in squares of 64 / sqrt(pixelsamples) pixels; per tile three shifts of time / lens
        * from a generator seeded with the tile's coordinates, points outside the sampler's window rejected; the integrators' arrays are
        * scrambled (0,2)-sequences drawn per camera sample (counts rounded up to powers of two).  The table is DATA of the reference
        * (BestCandidateSampler::sampleTable): the caller hands it over with hpt_scene_set_sample_table.  The camera samples do not depend on
        * the generator of a task — the production mode (BESTCANDIDATE_HASH) reproduces the reference's set of camera samples exactly (table
        * tiles are its work items) and replaces only the arrays' scrambles by the stateless hash of (tile, table entry, seed);
        * BESTCANDIDATE_MT_REPLAY: the reference's windows and stream — oracle only. */
       HPT_SAMPLER_BESTCANDIDATE_HASH = 10, HPT_SAMPLER_BESTCANDIDATE_MT_REPLAY = 11 };
#define HPT_SAMPLE_TABLE_SIZE 4096     /* SAMPLE_TABLE_SIZE, samplers/bestcandidate.h:43-45; entries of 5 floats */
#define HPT_SAMPLER_ADAPTIVE(kind, minsamples) ((kind) | ((minsamples) << 8))
#define HPT_SAMPLER_ADAPT_MIN(mode) (((mode) >> 8) & 0xfff)
#define HPT_SAMPLER_KIND(mode) ((mode) & 0x7f)
#define HPT_SAMPLER_STRATIFIED(kind, xsamples, jitter) ((kind) | ((jitter) ? 0x80 : 0) | ((xsamples) << 8))
#define HPT_SAMPLER_STRAT_XS(mode) (((mode) >> 8) & 0xfff)
#define HPT_SAMPLER_STRAT_JITTER(mode) (((mode) >> 7) & 1)

/* Kernel organisation of the same path state machine:
 *  HPT_PIPELINE_PERSISTENT : one persistent-threads launch per frame, path state in registers,
 *                            dead lanes regenerated in place;
 *  HPT_PIPELINE_WAVEFRONT  : advance / trace kernels over a pool of paths in HBM, active rays
 *                            compacted into a dense queue between the two (ballot + popcount). */
enum { HPT_PIPELINE_PERSISTENT = 0, HPT_PIPELINE_WAVEFRONT = 1 };

/* Surface integrator.  PATH: PathIntegrator::Li (integrators/path.cpp:52-123).
 * DIRECT_ALL / DIRECT_ONE: DirectLightingIntegrator::Li (integrators/directlighting.cpp:80-121) with strategy
 * "all" (UniformSampleAllLights, every light with its own nsamples, core/integrator.cpp:47-79) or "one"
 * (UniformSampleOneLight, :82-114).  No material on this path has a specular lobe, so SpecularReflect /
 * SpecularTransmit (directlighting.cpp:111-118) contribute nothing and maxdepth is not used. */
enum { HPT_INTEGRATOR_PATH = 0, HPT_INTEGRATOR_DIRECT_ALL = 1, HPT_INTEGRATOR_DIRECT_ONE = 2 };

typedef struct hpt_render_desc {
    int32_t xres, yres;               /* Film::xResolution, yResolution                  */
    int32_t x_start, x_count;         /* ImageFilm::xPixelStart/xPixelCount (crop window) */
    int32_t y_start, y_count;
    int32_t spp;                      /* samples per pixel: LDSampler::nPixelSamples (a power of two), RandomSampler::nSamples
                                       * (any), StratifiedSampler xs * ys (<= 4095)                                          */
    int32_t maxdepth;                 /* PathIntegrator::maxDepth                         */
    int32_t sampler_mode;
    uint32_t seed;                    /* LD_HASH seed                                     */
    int32_t ntasks;                   /* MT_REPLAY: nTasks of samplerrenderer.cpp:203-205 */
    int32_t shard_rank, shard_count;  /* pixel-tile shard of this device (0,1 = all)      */
    int32_t count_work;               /* 1: fill the traversal counters of hpt_stats      */
    int32_t pipeline;                 /* HPT_PIPELINE_*                                   */
    int32_t integrator;               /* HPT_INTEGRATOR_*                                 */
} hpt_render_desc;

typedef struct hpt_stats {
    double kernel_ms;          /* HIP-event time of the path kernel(s) on the render stream */
    uint64_t camera_samples;   /* samples integrated and accumulated                        */
    uint64_t closest_rays, shadow_rays;
    uint64_t nodes_visited;    /* 64-byte BVH2 nodes fetched                                */
    uint64_t tris_tested;      /* 48-byte triangle records fetched                          */
    uint64_t bad_samples;      /* NaN / negative / inf radiance zeroed (samplerrenderer.cpp:118-131) */
    uint32_t resident_waves, grid_blocks, block_threads, vgprs;
    uint32_t tune_cfg;         /* kernel configuration that ran: 0 = 4 waves/SIMD, 1 = 4 waves + early-exit
                                * traversal, 2 = 3 waves/SIMD, 3 / 4 = 4 / 3 waves with the wave's lanes in lock
                                * step (extension, shadow, MIS phases), 5 / 6 = 3 / 4 + idle lanes steal subtrees from the wave's long rays, 7 = 5 in its second compilation
                                * (other code-generation flags; matte / plastic scenes without animated instances — elsewhere it runs as 5); picked per scene by a probe render,
                                * HPT_TUNE=<n> pins it.  The shipped library carries 3, 5, 6 and 7 (0, 1, 2, 4 run as 3).  All configurations compute the same film. */
    uint32_t scratch_bytes;    /* private (scratch) memory per lane of the kernel that ran: its register spills (was padding before round 4) */
} hpt_stats;

typedef struct hpt_scene_info {
    int64_t n_tris, n_bvh_nodes, n_quadrics;
    int64_t bvh_bytes, tri_bytes, total_device_bytes;
    int32_t bvh_max_depth, pad;
    double build_ms;           /* flatten + BVH build, wall clock                                            */
    double device_build_ms;    /* HPT_BVH_BUILD=lbvh: HIP-event time of the device BVH builder's kernels      */
    int32_t device_built;      /* ... and how many trees (world + instances) it built (0: host binned SAH)    */
    int32_t pad2;
} hpt_scene_info;

typedef struct hpt_scene hpt_scene; /* opaque: device-resident flattened scene + BVH */

int hpt_device_count(void);
const char *hpt_last_error(void);
/* Starts the HIP runtime and `device`'s context on a thread of the library's own and returns at once; the first call that needs the
 * runtime (hpt_device_count, hpt_scene_create, ...) waits for that thread.  For hosts that know they will render before they have a
 * scene: the reference's pbrtWorldBegin (core/api.cpp:855) — its parser then reads the world while the runtime starts (50-190 ms on
 * an MI355X box).  Optional; HPT_OK, or HPT_E_INVALID for a negative device. */
int hpt_warmup(int device);

/* Build the device BVH (binned SAH, host) and upload everything to HBM of `device`. */
hpt_scene *hpt_scene_create(const hpt_scene_desc *desc, int device);
void hpt_scene_destroy(hpt_scene *scene);
int hpt_scene_get_info(const hpt_scene *scene, hpt_scene_info *info);

/* Film layout: x_count*y_count pixels, row-major, 4 floats {X, Y, Z, weightSum} — the
 * Lxyz/weightSum members of ImageFilm::Pixel (film/image.h:69-78).
 * hpt_render: renders and copies the film to host memory.
 * hpt_render_device: film stays in HBM (d_film_xyzw is a device pointer, zeroed by the call);
 *   `stream` is a hipStream_t (NULL = default).  Used by the multi-GPU path: each rank renders
 *   its pixel-tile shard, the film gather runs over RCCL on the same buffers. */
int hpt_render(hpt_scene *scene, const hpt_camera *cam, const hpt_render_desc *rd,
               float *film_xyzw_host, hpt_stats *stats);
int hpt_render_device(hpt_scene *scene, const hpt_camera *cam, const hpt_render_desc *rd,
                      void *d_film_xyzw, void *stream, hpt_stats *stats);

/* Pick the kernel configuration for this scene now (part of scene preparation, like the BVH build):
 * times each configuration on a probe render of a ninth of the image tiles.  Returns the configuration
 * (>= 0, also reported as hpt_stats.tune_cfg) or a negative HPT_E_*.  Optional: the first large
 * hpt_render* of an untuned scene does the same; HPT_TUNE=<cfg> in the environment pins it. */
int hpt_scene_tune(hpt_scene *scene, const hpt_camera *cam, const hpt_render_desc *rd);

/* Pixel reconstruction filter.  Replaces ImageFilm's `Filter *filter` + `filterTable`
 * (film/image.cpp:41-75): xwidth / ywidth are Filter::xWidth / yWidth, table[y * 16 + x] is
 * filter->Evaluate((x + .5) * xwidth / 16, (y + .5) * ywidth / 16) — the host computes it from whatever Filter
 * plugin the scene named (box, gaussian, mitchell, sinc, triangle), the device only looks weights up, exactly
 * as ImageFilm::AddSample does (film/image.cpp:77-137).  The filter also widens the SAMPLE extent
 * (ImageFilm::GetSampleExtent, film/image.cpp:157-166): camera samples are generated for pixels up to a filter
 * radius outside [x_start, x_start + x_count), the film stays x_count * y_count.
 * State of the scene handle, used by every later hpt_render* / hpt_scene_tune; NULL restores the default
 * (box, width 0.5: a sample lands in its own pixel).  With a filter wider than 0.5 the shards of a multi-GPU
 * render overlap at tile borders: each rank's film holds partial sums over the whole frame and the film
 * exchange is a sum (reduce) instead of a gather. */
#define HPT_FILTER_TABLE_SIZE 16
typedef struct hpt_filter {
    float xwidth, ywidth;
    float table[HPT_FILTER_TABLE_SIZE * HPT_FILTER_TABLE_SIZE];
} hpt_filter;
int hpt_scene_set_filter(hpt_scene *scene, const hpt_filter *filter);

/* A moving camera: PerspectiveCamera::CameraToWorld as an AnimatedTransform (core/camera.h:49; GenerateRay[Differential] ends with
 * CameraToWorld(*ray, ray), cameras/perspective.cpp:110,135 -> AnimatedTransform::operator()(Ray), core/transform.cpp:416-442).  The record
 * type of an animated instance, read as camera-to-world: actually_animated, start_time / end_time (RenderOptions::transformStartTime /
 * EndTime), T / R / S of the two decomposed end transforms, w2p_m[e] / w2p_minv[e] = the end transforms' m / mInv; bounds unused.  State of
 * the scene handle for the following renders; NULL = static camera (hpt_camera.camera_to_world, the default).  The camera samples a time
 * in [shutter_open, shutter_close] whether or not the scene has animated instances. */
int hpt_scene_set_camera_motion(hpt_scene *scene, const hpt_instance *camera_to_world);
/* Sampler "bestcandidate" (HPT_SAMPLER_BESTCANDIDATE_HASH): the reference's precomputed sample table — BestCandidateSampler::sampleTable,
 * samplers/bestcandidate.h:86 / bestcandidate.out: HPT_SAMPLE_TABLE_SIZE entries of 5 floats (image x, y, time, lens u, v, all in [0, 1)) —
 * copied to the device.  A render under that sampler without a table fails with HPT_E_INVALID.  NULL removes it. */
int hpt_scene_set_sample_table(hpt_scene *scene, const float *table, int n_entries);

/* ---- multi-GPU (SURVEY.md §8b "gpus", §8e) ------------------------------------------------------------------------------------
 * The path shards with no data-path collective: the scene is replicated per GPU, shard r renders the 32x32 pixel tiles t with
 * t % n == r (shard_rank / shard_count above), and ONE exchange of film data per frame brings the frame to shard 0 — a gather of the
 * owned tiles over RCCL send / recv (the reference's disjoint image tiles, renderers/samplerrenderer.cpp:203-212, core/sampler.cpp:55-74),
 * or a sum-reduce of the full-frame films under a reconstruction filter wider than the box (film/image.cpp:96-136).
 *
 * hpt_multi: one process, one host thread per GPU — what `Renderer "hip" "integer gpus" [N]` of the pbrt plugin drives.  `devices` may
 * name a device more than once (several shards on one GPU; their tiles then move with hipMemcpyAsync, RCCL refuses duplicate devices).
 * hpt_multi_render ignores rd->shard_rank / shard_count; `stats` receives n_devices records (may be NULL). */
typedef struct hpt_multi hpt_multi;
hpt_multi *hpt_multi_create(const hpt_scene_desc *desc, const int *devices, int n_devices);
void hpt_multi_destroy(hpt_multi *m);
int hpt_multi_set_filter(hpt_multi *m, const hpt_filter *filter);
int hpt_multi_set_camera_motion(hpt_multi *m, const hpt_instance *camera_to_world);   /* hpt_scene_set_camera_motion on every shard */
int hpt_multi_set_sample_table(hpt_multi *m, const float *table, int n_entries);         /* hpt_scene_set_sample_table on every shard */
/* Dynamic hand-out: with HPT_MULTI_CHUNKS=<k> (2 .. 16) in the environment and the box filter, hpt_multi_render cuts the frame into
 * n_devices x k round-robin sub-shards and every device pulls the next one from a host-side atomic counter when its kernel has drained; the
 * film exchange is then the sum.  out[i] = sub-shards device i rendered in the last frame (1 each under the static split). */
int hpt_multi_chunks_taken(hpt_multi *m, int *out_n_devices);
int hpt_multi_scene(hpt_multi *m, int shard, hpt_scene **out);     /* the shard's scene handle (hpt_scene_tune, hpt_scene_get_info) */
int hpt_multi_render(hpt_multi *m, const hpt_camera *cam, const hpt_render_desc *rd, float *film_xyzw_host, hpt_stats *stats);

/* hpt_comm: one PROCESS per GPU (bench.py under torch.distributed.run; any launcher that can hand 128 bytes from rank 0 to the others).
 * hpt_comm_unique_id on rank 0 (ncclGetUniqueId), hpt_comm_create everywhere (ncclCommInitRank, collective), then after every
 * hpt_render_device of the rank's shard: hpt_comm_exchange_film on the same stream (asynchronous like an RCCL call) — afterwards rank 0's
 * d_film_xyzw holds the whole frame.  wide_filter != 0: the scene has a filter set (sum-reduce instead of the tile gather).
 * Transport: RCCL (ncclSend / ncclRecv / ncclReduce over xGMI) by default.  With HPT_COMM_TRANSPORT=host in rank 0's environment
 * hpt_comm_unique_id returns an id that names POSIX shared-memory mailboxes instead, and every rank created from that id moves its packed
 * tiles (or its film, wide filter) through them with hipMemcpy — the same pack / unpack / sum kernels and tile bookkeeping; blocking.  For
 * hosts without RCCL and for running several ranks on ONE device, which RCCL refuses (round 4; tests/test_gpu_multi.py). */
typedef struct hpt_comm hpt_comm;
int hpt_comm_unique_id(void *out_128_bytes);
hpt_comm *hpt_comm_create(const void *id_128_bytes, int rank, int world, int device);
void hpt_comm_destroy(hpt_comm *c);
int hpt_comm_exchange_film(hpt_comm *c, const hpt_render_desc *rd, void *d_film_xyzw, void *stream, int wide_filter);
/* What the communicator is and what its last exchange did (round 6; any pointer may be NULL).  *ranks: ncclCommCount of the RCCL communicator (host-staged
 * transport: the world size); *transport: 0 RCCL, 1 host-staged (the one-device dry run); *peers: ranks whose tile records (wide filter: films) rank 0
 * received in the last exchange, 0 on the other ranks; *exchange_ms: the last exchange's duration on its stream (pack, transfer, unpack; the call waits
 * for it; < 0: none yet). */
int hpt_comm_info(hpt_comm *c, int *ranks, int *transport, int *peers, float *exchange_ms);

/* ---- scene blob (serialised hpt_scene_desc + camera + render defaults); host only ------ */
typedef struct hpt_blob hpt_blob;
int hpt_blob_save(const char *path, const hpt_scene_desc *desc, const hpt_camera *cam,
                  const hpt_render_desc *rd);
hpt_blob *hpt_blob_load(const char *path);
const hpt_scene_desc *hpt_blob_scene(const hpt_blob *b);
const hpt_camera *hpt_blob_camera(const hpt_blob *b);
const hpt_render_desc *hpt_blob_render(const hpt_blob *b);
void hpt_blob_free(hpt_blob *b);

/* ---- function-level parity hooks (each runs the SAME device functions the path kernel
 *      uses, over arrays of inputs; used by tests/ to compare with the oracle / golden vectors) */

/* rays: n * 8 floats {o.xyz, d.xyz, mint, maxt}.  out_hit: n * 4 floats {t, b1, b2, eps},
 * out_prim: n ints (global primitive id, -1 = miss).  anyhit != 0 -> BVHAccel::IntersectP. */
int hpt_test_intersect(hpt_scene *scene, const float *rays, int64_t n, int anyhit,
                       float *out_hit, int32_t *out_prim);

/* BSDF::f / Pdf / Sample_f at a synthetic hit: in: n * 16 floats
 *   {wo.xyz, wi.xyz, u1,u2,ucomp, nn.xyz(shading normal), dpdu.xyz, ng_sign}; material index.
 * out: n * 12 floats {f.rgb, pdf, sample_wi.xyz, sample_f.rgb, sample_pdf, sampled_type}. */
int hpt_test_bsdf(hpt_scene *scene, int material, const float *in, int64_t n, float *out);

/* Sampler: values of camera sample `s` of pixel (x,y): out 5+37... see hpt_sampler.h; n_out floats
 * per sample = 5 (imageX,imageY,lensU,lensV,time) + 14 one-D + 18 two-D. */
int hpt_test_sampler(const hpt_render_desc *rd, int x, int y, float *out /* spp*37 */);

/* ---- calibration: the achieved-peak HBM bandwidth of `device` ("verify with a stream-triad microbench and report
 * achieved-peak too").  float4 triad a = b + s * c over three arrays of bytes_per_array each (pass >= 1 GiB: far beyond the 256 MiB
 * Infinity Cache), best of `reps` launches; *gb_per_s = 3 * bytes_per_array / time.  bench.py reports it as roofline.achieved_peak. */
int hpt_calib_hbm_triad(int device, size_t bytes_per_array, int reps, double *gb_per_s);
/* the same harness as a float4 COPY (a = b; MI355X_MICROARCH.md's own calibration: 6.29 TB/s) and as a float4 READ (sum of b): round 4 */
int hpt_calib_hbm_copy(int device, size_t bytes_per_array, int reps, double *gb_per_s);
int hpt_calib_hbm_read(int device, size_t bytes_per_array, int reps, double *gb_per_s);

/* Bytes of one BVH node fetch of the walk that hpt_stats.nodes_visited counts (count_work): 64 — the BVH2 node with both children's
 * boxes — or 128 when this build's lock-step + stealing walk runs on the four-wide trees (two 64-byte lines per node). */
int hpt_kernel_node_bytes(void);

/* sizeof() of the ABI's records as THIS library was compiled: {hpt_mesh, hpt_quadric, hpt_material, hpt_light, hpt_camera, hpt_render_desc,
 * hpt_stats, hpt_blob_header, hpt_instance, hpt_texture} — a binding (ctypes, cgo, JNI) checks its own layouts against it before the first call. */
void hpt_abi_sizes(int32_t out[10]);

#ifdef __cplusplus
}
#endif
#endif /* HPT_H */
