// hip_renderer.cpp — HipPathRenderer: flattens the pbrt-v2 Scene that the reference's own
// parser / api.cpp / Create* factories built into the POD descriptors of include/hpt.h and calls
// the MI355X library.  Built with -fno-access-control so the private members of the reference
// classes can be READ in place without modifying or copying a single reference source file.
//
// What is read, and where it is defined in the reference:
//   Scene::aggregate (BVHAccel::primitives, accelerators/bvh.h:66-71) -> GeometricPrimitive
//     {shape, material, areaLight} (core/primitive.h:88-92) -> Triangle{mesh,v} /
//     TriangleMesh{p,n,uvs,vertexIndex} (shapes/trianglemesh.h:59-68), Sphere, Disk
//   Materials: MatteMaterial, PlasticMaterial, MeasuredMaterial (materials/*.h) with
//     ConstantTexture values (textures/constant.h:45-55)
//   Lights: PointLight, DiffuseAreaLight(+ShapeSet), InfiniteAreaLight(+MIPMap, Distribution2D)
//   PerspectiveCamera (RasterToCamera, CameraToWorld), ImageFilm (pixel extent, filter table),
//   LDSampler (nPixelSamples), PathIntegrator (maxDepth) or DirectLightingIntegrator (strategy; Light::nSamples)
// Anything else is outside the hot-path scope and is rejected with Severe().
#include "stdafx.h"
#include "hip_renderer.h"
#include <time.h>
#include <unistd.h>
#include <stdio.h>
#include <string.h>
#include <stdlib.h>

#include "scene.h"
#include "camera.h"
#include "film.h"
#include "sampler.h"
#include "integrator.h"
#include "volume.h"
#include "intersection.h"
#include "light.h"
#include "parallel.h"
#include "montecarlo.h"
#include "mipmap.h"
#include "reflection.h"
#include "kdtree.h"
#include "progressreporter.h"
#include "accelerators/bvh.h"
#include "cameras/perspective.h"
#include "film/image.h"
#include "filters/box.h"
#include "integrators/path.h"
#include "integrators/directlighting.h"
#include "lights/diffuse.h"
#include "lights/infinite.h"
#include "lights/point.h"
#include "lights/spot.h"
#include "lights/distant.h"
#include "materials/matte.h"
#include "materials/measured.h"
#include "materials/metal.h"
#include "materials/substrate.h"
#include "materials/plastic.h"
#include "samplers/lowdiscrepancy.h"
#include "samplers/random.h"
#include "samplers/stratified.h"
#include "samplers/halton.h"
#include "samplers/adaptive.h"
#include "samplers/bestcandidate.h"
#include "shapes/disk.h"
#include "shapes/sphere.h"
#include "shapes/trianglemesh.h"
#include "primitive.h"
#include "quaternion.h"
#include "textures/constant.h"
#include "textures/imagemap.h"
#include "textures/scale.h"
#include "textures/mix.h"
#include "materials/glass.h"
#include "materials/mirror.h"

#include <map>
#include <stdlib.h>
#include <string.h>

#include "hpt.h"

namespace {

void CopyM(const Matrix4x4 &m, float out[16]) { memcpy(out, m.m, 16 * sizeof(float)); }

template <typename T> bool ConstTex(const Reference<Texture<T> > &tex, T *v) {
    if (!tex.GetPtr()) return false;
    const ConstantTexture<T> *c = dynamic_cast<const ConstantTexture<T> *>(tex.GetPtr());
    if (!c) return false;
    *v = c->value;
    return true;
}

struct Flattener {
    std::vector<hpt_mesh> meshes;
    std::vector<hpt_quadric> quadrics;
    std::vector<hpt_material> materials;
    std::vector<hpt_light> lights;
    std::vector<hpt_instance> instances;
    std::vector<float> fpool;
    std::vector<int32_t> ipool;
    std::vector<hpt_texture> textures;
    std::map<const void *, int> textureIndex;
    std::map<const void *, int64_t> tableOffset;
    std::map<const Primitive *, int> instanceOfPrimitive;   // aggregate (or bare shape) of a TransformedPrimitive -> the instance that owns it
    std::vector<int64_t> meshTriBase;          // global number of every mesh's first triangle (meshes in descriptor order)
    std::map<const TriangleMesh *, int> meshIndex;
    std::map<const Shape *, int> quadricIndex;
    std::map<const Material *, int> materialIndex;
    std::map<const Light *, int> lightIndex;

    int64_t PushF(const float *p, size_t n) {
        int64_t off = (int64_t)fpool.size();
        fpool.insert(fpool.end(), p, p + n);
        return off;
    }
    int64_t PushI(const int *p, size_t n) {
        int64_t off = (int64_t)ipool.size();
        ipool.insert(ipool.end(), p, p + n);
        return off;
    }


    // ---- textures (core/texture.h): ConstantTexture, ImageTexture over any TextureMapping2D, ScaleTexture, MixTexture -------------
    // Operands are emitted before the texture that uses them (hpt_validate_desc relies on it).
    template <typename Tmem, typename Tret> int AddImageTexture(const ImageTexture<Tmem, Tret> *it, int channels) {
        hpt_texture r; memset(&r, 0, sizeof(r));
        r.kind = HPT_TEX_IMAGEMAP; r.channels = channels; r.tex1 = r.tex2 = r.amount = -1;
        // TextureMapping2D (core/texture.h:47-113): the four mappings textures/imagemap.cpp:106-124 can create
        if (const UVMapping2D *uv = dynamic_cast<const UVMapping2D *>(it->mapping)) {
            r.mapping = HPT_MAP_UV; r.su = uv->su; r.sv = uv->sv; r.du = uv->du; r.dv = uv->dv;
        } else if (const SphericalMapping2D *sp = dynamic_cast<const SphericalMapping2D *>(it->mapping)) {
            r.mapping = HPT_MAP_SPHERICAL; CopyM(sp->WorldToTexture.m, r.map_m);
        } else if (const CylindricalMapping2D *cy = dynamic_cast<const CylindricalMapping2D *>(it->mapping)) {
            r.mapping = HPT_MAP_CYLINDRICAL; CopyM(cy->WorldToTexture.m, r.map_m);
        } else if (const PlanarMapping2D *pl = dynamic_cast<const PlanarMapping2D *>(it->mapping)) {
            r.mapping = HPT_MAP_PLANAR;
            r.map_m[0] = pl->vs.x; r.map_m[1] = pl->vs.y; r.map_m[2] = pl->vs.z; r.map_m[3] = pl->vt.x; r.map_m[4] = pl->vt.y; r.map_m[5] = pl->vt.z;
            r.map_m[6] = pl->ds; r.map_m[7] = pl->dt;
        } else
            Severe("hip renderer: unknown TextureMapping2D");
        const MIPMap<Tmem> *mm = it->mipmap;
        r.width = (int)mm->width; r.height = (int)mm->height; r.levels = (int)mm->nLevels;
        r.wrap = mm->wrapMode == TEXTURE_REPEAT ? HPT_WRAP_REPEAT : mm->wrapMode == TEXTURE_BLACK ? HPT_WRAP_BLACK : HPT_WRAP_CLAMP;
        r.do_trilinear = mm->doTrilinear ? 1 : 0; r.max_aniso = mm->maxAnisotropy;
        r.pyr_off = (int64_t)fpool.size();
        for (uint32_t l = 0; l < mm->nLevels; ++l) {
            const BlockedArray<Tmem> &lv = *mm->pyramid[l];
            for (uint32_t t = 0; t < lv.vSize(); ++t)
                for (uint32_t sx = 0; sx < lv.uSize(); ++sx) PushTexel(lv(sx, t));
        }
        textures.push_back(r);
        return (int)textures.size() - 1;
    }
    void PushTexel(float v) { fpool.push_back(v); }
    void PushTexel(const RGBSpectrum &v) { float rgb[3]; v.ToRGB(rgb); fpool.push_back(rgb[0]); fpool.push_back(rgb[1]); fpool.push_back(rgb[2]); }
    static void Value(float v, float out[3]) { out[0] = out[1] = out[2] = v; }
    static void Value(const Spectrum &v, float out[3]) { v.ToRGB(out); }

    int AddFloatTexture(const Texture<float> *t) { return AddTextureT<float, float>(t, 1); }
    int AddSpectrumTexture(const Texture<Spectrum> *t) { return AddTextureT<Spectrum, RGBSpectrum>(t, 3); }
    template <typename T, typename Tmem> int AddTextureT(const Texture<T> *t, int channels) {
        std::map<const void *, int>::iterator it = textureIndex.find((const void *)t);
        if (it != textureIndex.end()) return it->second;
        int idx = -1;
        if (const ConstantTexture<T> *c = dynamic_cast<const ConstantTexture<T> *>(t)) {
            hpt_texture r; memset(&r, 0, sizeof(r));
            r.kind = HPT_TEX_CONSTANT; r.channels = channels; r.tex1 = r.tex2 = r.amount = -1;
            Value(c->value, r.value);
            textures.push_back(r); idx = (int)textures.size() - 1;
        } else if (const ImageTexture<Tmem, T> *im = dynamic_cast<const ImageTexture<Tmem, T> *>(t)) {
            idx = AddImageTexture(im, channels);
        } else if (const ScaleTexture<T, T> *sc = dynamic_cast<const ScaleTexture<T, T> *>(t)) {
            int a = AddTextureT<T, Tmem>(sc->tex1.GetPtr(), channels), b = AddTextureT<T, Tmem>(sc->tex2.GetPtr(), channels);
            hpt_texture r; memset(&r, 0, sizeof(r));
            r.kind = HPT_TEX_SCALE; r.channels = channels; r.tex1 = a; r.tex2 = b; r.amount = -1;
            textures.push_back(r); idx = (int)textures.size() - 1;
        } else if (const MixTexture<T> *mx = dynamic_cast<const MixTexture<T> *>(t)) {
            int a = AddTextureT<T, Tmem>(mx->tex1.GetPtr(), channels), b = AddTextureT<T, Tmem>(mx->tex2.GetPtr(), channels);
            int am = AddFloatTexture(mx->amount.GetPtr());
            hpt_texture r; memset(&r, 0, sizeof(r));
            r.kind = HPT_TEX_MIX; r.channels = channels; r.tex1 = a; r.tex2 = b; r.amount = am;
            textures.push_back(r); idx = (int)textures.size() - 1;
        } else
            Severe("hip renderer: texture type outside the hot-path scope (supported: constant, imagemap, scale, mix)");
        textureIndex[(const void *)t] = idx;
        return idx;
    }
    // A material parameter: the constant goes into the record's field, anything else into the texture table (slot index)
    int SpectrumSlot(const Reference<Texture<Spectrum> > &tex, float k[3], bool clamp) {
        Spectrum v;
        if (ConstTex(tex, &v)) { if (clamp) v = v.Clamp(); v.ToRGB(k); return -1; }
        k[0] = k[1] = k[2] = 0.f;
        return AddSpectrumTexture(tex.GetPtr());
    }
    int FloatSlot(const Reference<Texture<float> > &tex, float *k) {
        if (ConstTex(tex, k)) return -1;
        *k = 0.f;
        return AddFloatTexture(tex.GetPtr());
    }
    int BumpSlot(const Reference<Texture<float> > &tex) { return tex.GetPtr() ? AddFloatTexture(tex.GetPtr()) : -1; }

    int AddMaterial(const Material *m) {
        std::map<const Material *, int>::iterator it = materialIndex.find(m);
        if (it != materialIndex.end()) return it->second;
        hpt_material r;
        memset(&r, 0, sizeof(r));
        r.kd_split_off = r.kd_bits_off = r.kd_data_off = r.rh_off = -1;
        for (int k = 0; k < HPT_N_TEXSLOTS; ++k) r.tex[k] = -1;
        if (const MatteMaterial *mm = dynamic_cast<const MatteMaterial *>(m)) {
            r.kind = HPT_MAT_MATTE;
            r.tex[HPT_TEXSLOT_KD] = SpectrumSlot(mm->Kd, r.kd, true);                  // matte.cpp:53
            r.tex[HPT_TEXSLOT_ROUGH] = FloatSlot(mm->sigma, &r.sigma);
            if (r.tex[HPT_TEXSLOT_ROUGH] < 0) r.sigma = Clamp(r.sigma, 0.f, 90.f);    // matte.cpp:54
            r.tex[HPT_TEXSLOT_BUMP] = BumpSlot(mm->bumpMap);
        } else if (const PlasticMaterial *pm = dynamic_cast<const PlasticMaterial *>(m)) {
            r.kind = HPT_MAT_PLASTIC;
            r.tex[HPT_TEXSLOT_KD] = SpectrumSlot(pm->Kd, r.kd, true);                  // plastic.cpp:52
            r.tex[HPT_TEXSLOT_KS] = SpectrumSlot(pm->Ks, r.ks, true);                  // plastic.cpp:57
            r.tex[HPT_TEXSLOT_ROUGH] = FloatSlot(pm->roughness, &r.roughness);
            r.tex[HPT_TEXSLOT_BUMP] = BumpSlot(pm->bumpMap);
        } else if (const MeasuredMaterial *me = dynamic_cast<const MeasuredMaterial *>(m)) {
            r.tex[HPT_TEXSLOT_BUMP] = BumpSlot(me->bumpMap);
            if (me->regularHalfangleData) {                    // RegularHalfangleBRDF (MERL .binary), measured.cpp:137-184
                r.kind = HPT_MAT_MEASURED_REGULAR;
                r.rh_n_theta_h = (int)me->nThetaH; r.rh_n_theta_d = (int)me->nThetaD; r.rh_n_phi_d = (int)me->nPhiD;
                // MeasuredMaterial shares one table among the materials that name the same file (measured.cpp:139-142): so does the blob
                std::map<const void *, int64_t>::iterator ti = tableOffset.find((const void *)me->regularHalfangleData);
                if (ti != tableOffset.end()) r.rh_off = ti->second;
                else {
                    r.rh_off = PushF(me->regularHalfangleData, 3 * (size_t)me->nThetaH * me->nThetaD * me->nPhiD);
                    tableOffset[(const void *)me->regularHalfangleData] = r.rh_off;
                }
            } else if (me->thetaPhiData) {
            r.kind = HPT_MAT_MEASURED_IRREG;
            const KdTree<IrregIsotropicBRDFSample> *kd = me->thetaPhiData;
            uint32_t n = kd->nNodes;
            std::vector<float> split(n), data(6 * (size_t)n);
            std::vector<int> bits(n);
            for (uint32_t i = 0; i < n; ++i) {
                // a leaf's splitPos is never written by KdNode::initLeaf (core/kdtree.h:50-54) nor read by a lookup: it is
                // uninitialised memory in the reference; zero it so that blobs are reproducible
                split[i] = kd->nodes[i].splitAxis == 3 ? 0.f : kd->nodes[i].splitPos;
                bits[i] = (int)(kd->nodes[i].splitAxis | (kd->nodes[i].hasLeftChild << 2) |
                                (kd->nodes[i].rightChild << 3));
                const IrregIsotropicBRDFSample &s = kd->nodeData[i];
                float rgb[3];
                s.v.ToRGB(rgb);
                data[6 * i + 0] = s.p.x; data[6 * i + 1] = s.p.y; data[6 * i + 2] = s.p.z;
                data[6 * i + 3] = rgb[0]; data[6 * i + 4] = rgb[1]; data[6 * i + 5] = rgb[2];
            }
            r.kd_nnodes = (int)n;
            r.kd_split_off = PushF(&split[0], n);
            r.kd_bits_off = PushI(&bits[0], n);
            r.kd_data_off = PushF(&data[0], 6 * (size_t)n);
            } else
                Severe("hip renderer: measured material without data");
        } else if (const MetalMaterial *mt = dynamic_cast<const MetalMaterial *>(m)) {
            r.kind = HPT_MAT_METAL;
            r.tex[HPT_TEXSLOT_KD] = SpectrumSlot(mt->eta, r.eta, false);               // metal.cpp:64-65 (no Clamp)
            r.tex[HPT_TEXSLOT_KS] = SpectrumSlot(mt->k, r.k, false);
            r.tex[HPT_TEXSLOT_ROUGH] = FloatSlot(mt->roughness, &r.roughness);
            r.tex[HPT_TEXSLOT_BUMP] = BumpSlot(mt->bumpMap);
        } else if (const SubstrateMaterial *sm = dynamic_cast<const SubstrateMaterial *>(m)) {
            r.kind = HPT_MAT_SUBSTRATE;
            r.tex[HPT_TEXSLOT_KD] = SpectrumSlot(sm->Kd, r.kd, true);                  // substrate.cpp:50-51
            r.tex[HPT_TEXSLOT_KS] = SpectrumSlot(sm->Ks, r.ks, true);
            r.tex[HPT_TEXSLOT_ROUGH] = FloatSlot(sm->nu, &r.nu);
            r.tex[HPT_TEXSLOT_ROUGH_V] = FloatSlot(sm->nv, &r.nv);
            r.tex[HPT_TEXSLOT_BUMP] = BumpSlot(sm->bumpMap);
        } else if (const GlassMaterial *gm = dynamic_cast<const GlassMaterial *>(m)) {
            r.kind = HPT_MAT_GLASS;
            r.tex[HPT_TEXSLOT_KS] = SpectrumSlot(gm->Kr, r.ks, true);                  // glass.cpp:50-51
            r.tex[HPT_TEXSLOT_KT] = SpectrumSlot(gm->Kt, r.kt, true);
            r.tex[HPT_TEXSLOT_INDEX] = FloatSlot(gm->index, &r.index);
            r.tex[HPT_TEXSLOT_BUMP] = BumpSlot(gm->bumpMap);
        } else if (const MirrorMaterial *mi = dynamic_cast<const MirrorMaterial *>(m)) {
            r.kind = HPT_MAT_MIRROR;
            r.tex[HPT_TEXSLOT_KS] = SpectrumSlot(mi->Kr, r.ks, true);                  // mirror.cpp:52
            r.tex[HPT_TEXSLOT_BUMP] = BumpSlot(mi->bumpMap);
        } else
            Severe("hip renderer: material type outside the hot-path scope "
                   "(supported: matte, plastic, measured, metal, substrate, glass, mirror)");
        int idx = (int)materials.size();
        materials.push_back(r);
        materialIndex[m] = idx;
        return idx;
    }

    int LightOf(const AreaLight *a) {
        if (!a) return -1;
        std::map<const Light *, int>::iterator it = lightIndex.find(a);
        if (it != lightIndex.end()) return it->second;
        // Not one of Scene::lights: the area light of a shape inside an object instance (core/api.cpp:1046-1049 warns and leaves it out).  The reference never
        // samples it, but a camera ray or a specular bounce that hits the shape picks up Intersection::Le: an UNSAMPLED record behind the scene's lights
        // (include/hpt.h, HPT_LIGHT_UNSAMPLED: DIFFUSE_AREA, no quadric, an empty shape set)
        const DiffuseAreaLight *dl = dynamic_cast<const DiffuseAreaLight *>(a);
        if (!dl) Severe("hip renderer: primitive refers to an area light of an unknown class that is not in Scene::lights");
        hpt_light r;
        memset(&r, 0, sizeof(r));
        r.kind = HPT_LIGHT_DIFFUSE_AREA; r.quadric = -1; r.set_n = 0; r.set_off = r.set_area_off = -1;
        r.tex_off = r.cond_func_off = r.cond_cdf_off = r.cond_int_off = r.marg_func_off = r.marg_cdf_off = -1;
        CopyM(a->LightToWorld.m, r.l2w); CopyM(a->LightToWorld.mInv, r.l2w_inv);
        r.nsamples = a->nSamples;
        dl->Lemit.ToRGB(r.intensity);
        r.area = dl->area;
        lights.push_back(r);
        lightIndex[a] = (int)lights.size() - 1;
        return (int)lights.size() - 1;
    }

    void AddTriangle(const Triangle *tri, const GeometricPrimitive *gp, int instance = -1) {
        const TriangleMesh *mesh = tri->mesh.GetPtr();
        if (meshIndex.find(mesh) != meshIndex.end()) return;
        hpt_mesh r;
        memset(&r, 0, sizeof(r));
        r.ntris = mesh->ntris;
        r.nverts = mesh->nverts;
        r.p_off = PushF(&mesh->p[0].x, 3 * (size_t)mesh->nverts);
        r.n_off = mesh->n ? PushF(&mesh->n[0].x, 3 * (size_t)mesh->nverts) : -1;
        r.uv_off = mesh->uvs ? PushF(mesh->uvs, 2 * (size_t)mesh->nverts) : -1;
        r.s_off = mesh->s ? PushF(&mesh->s[0].x, 3 * (size_t)mesh->nverts) : -1;     // "vector S": explicit tangents (trianglemesh.cpp:326-329)
        r.idx_off = PushI(mesh->vertexIndex, 3 * (size_t)mesh->ntris);
        r.material = AddMaterial(gp->material.GetPtr());
        r.arealight = LightOf(gp->areaLight);
        r.alpha_tex = mesh->alphaTexture.GetPtr() ? 1 + AddFloatTexture(mesh->alphaTexture.GetPtr()) : 0;
        r.instance = instance;
        r.reverse_orientation = mesh->ReverseOrientation;
        r.swaps_handedness = mesh->TransformSwapsHandedness;
        CopyM(mesh->ObjectToWorld->m, r.o2w);
        CopyM(mesh->ObjectToWorld->mInv, r.o2w_inv);
        meshIndex[mesh] = (int)meshes.size();
        meshTriBase.push_back(meshTriBase.empty() ? 0 : meshTriBase.back() + meshes.back().ntris);
        meshes.push_back(r);
    }

    void AddQuadric(const Shape *shape, const GeometricPrimitive *gp) {
        if (quadricIndex.find(shape) != quadricIndex.end()) return;
        hpt_quadric q;
        memset(&q, 0, sizeof(q));
        if (const Sphere *s = dynamic_cast<const Sphere *>(shape)) {
            q.kind = HPT_QUADRIC_SPHERE;
            q.radius = s->radius; q.zmin = s->zmin; q.zmax = s->zmax;
            q.theta_min = s->thetaMin; q.theta_max = s->thetaMax; q.phi_max = s->phiMax;
        } else if (const Disk *d = dynamic_cast<const Disk *>(shape)) {
            q.kind = HPT_QUADRIC_DISK;
            q.radius = d->radius; q.inner_radius = d->innerRadius; q.height = d->height;
            q.phi_max = d->phiMax;
        } else
            Severe("hip renderer: shape type outside the hot-path scope "
                   "(supported: trianglemesh, loopsubdiv, sphere, disk)");
        q.material = AddMaterial(gp->material.GetPtr());
        q.arealight = LightOf(gp->areaLight);
        q.reverse_orientation = shape->ReverseOrientation;
        q.swaps_handedness = shape->TransformSwapsHandedness;
        CopyM(shape->ObjectToWorld->m, q.o2w);
        CopyM(shape->ObjectToWorld->mInv, q.o2w_inv);
        if (memcmp(shape->ObjectToWorld->mInv.m, shape->WorldToObject->m.m, 16 * sizeof(float)))
            Severe("hip renderer: WorldToObject is not the stored inverse of ObjectToWorld");
        quadricIndex[shape] = (int)quadrics.size();
        quadrics.push_back(q);
    }

    // TransformedPrimitive over BVHAccel(refined shape) (core/api.cpp:1012-1044)
    // An AnimatedTransform as the library's record: both end transforms and their decomposition (core/transform.cpp:345-369), read in place
    static void FillAnimated(const AnimatedTransform &at, hpt_instance *r) {
        memset(r, 0, sizeof(*r));
        r->actually_animated = at.actuallyAnimated;
        r->start_time = at.startTime; r->end_time = at.endTime;
        for (int k = 0; k < 2; ++k) {
            r->T[k][0] = at.T[k].x; r->T[k][1] = at.T[k].y; r->T[k][2] = at.T[k].z;
            r->R[k][0] = at.R[k].v.x; r->R[k][1] = at.R[k].v.y; r->R[k][2] = at.R[k].v.z; r->R[k][3] = at.R[k].w;
            CopyM(at.S[k], r->S[k]);
            const Transform *t = k == 0 ? at.startTransform : at.endTransform;
            CopyM(t->m, r->w2p_m[k]); CopyM(t->mInv, r->w2p_minv[k]);
        }
    }
    void AddInstance(const TransformedPrimitive *tp) {
        const AnimatedTransform &at = tp->WorldToPrimitive;
        hpt_instance r;
        FillAnimated(at, &r);
        BBox wb = tp->WorldBound();
        r.bounds[0] = wb.pMin.x; r.bounds[1] = wb.pMin.y; r.bounds[2] = wb.pMin.z;
        r.bounds[3] = wb.pMax.x; r.bounds[4] = wb.pMax.y; r.bounds[5] = wb.pMax.z;
        int idx = (int)instances.size();
        instances.push_back(r);
        // object instancing (pbrtObjectInstance, core/api.cpp:1114-1147): every use of an object is a TransformedPrimitive over the SAME aggregate;
        // the first one owns its meshes, the others share them (hpt_instance.quadric1 < 0)
        std::map<const Primitive *, int>::iterator shared = instanceOfPrimitive.find(tp->primitive.GetPtr());
        if (shared != instanceOfPrimitive.end()) { instances[idx].quadric1 = -(shared->second + 1); return; }
        instanceOfPrimitive[tp->primitive.GetPtr()] = idx;
        std::vector<const Primitive *> todo;
        todo.push_back(tp->primitive.GetPtr());
        while (!todo.empty()) {
            const Primitive *p = todo.back(); todo.pop_back();
            if (const BVHAccel *b = dynamic_cast<const BVHAccel *>(p)) {
                for (size_t i = 0; i < b->primitives.size(); ++i) todo.push_back(b->primitives[i].GetPtr());
            } else if (const GeometricPrimitive *gp = dynamic_cast<const GeometricPrimitive *>(p)) {
                const Triangle *tri = dynamic_cast<const Triangle *>(gp->shape.GetPtr());
                if (tri) { AddTriangle(tri, gp, idx); continue; }
                // an animated sphere / disk: a shape that CanIntersect() stays the bare GeometricPrimitive under the TransformedPrimitive
                // (core/api.cpp:1032-1042), with identity ObjectToWorld and no area light (api.cpp:1014-1021): hpt_instance.quadric1
                if (p != tp->primitive.GetPtr() || quadricIndex.find(gp->shape.GetPtr()) != quadricIndex.end())
                    Severe("hip renderer: an animated quadric inside an aggregate, or shared between instances, is outside the hot-path scope");
                AddQuadric(gp->shape.GetPtr(), gp);
                instances[idx].quadric1 = 1 + quadricIndex[gp->shape.GetPtr()];
            } else
                Severe("hip renderer: nested instances are outside the hot-path scope");
        }
    }

    void AddLights(const Scene *scene) {
        for (uint32_t i = 0; i < scene->lights.size(); ++i) lightIndex[scene->lights[i]] = (int)i;
        lights.resize(scene->lights.size());
        for (uint32_t i = 0; i < scene->lights.size(); ++i) {
            const Light *l = scene->lights[i];
            hpt_light r;
            memset(&r, 0, sizeof(r));
            r.quadric = -1;
            r.tex_off = r.cond_func_off = r.cond_cdf_off = r.cond_int_off = -1;
            r.marg_func_off = r.marg_cdf_off = -1;
            CopyM(l->LightToWorld.m, r.l2w);
            CopyM(l->LightToWorld.mInv, r.l2w_inv);
            r.nsamples = l->nSamples;
            if (const PointLight *pl = dynamic_cast<const PointLight *>(l)) {
                r.kind = HPT_LIGHT_POINT;
                r.pos[0] = pl->lightPos.x; r.pos[1] = pl->lightPos.y; r.pos[2] = pl->lightPos.z;
                pl->Intensity.ToRGB(r.intensity);
            } else if (const SpotLight *sl = dynamic_cast<const SpotLight *>(l)) {       // ABI 8 (lights/spot.cpp)
                r.kind = HPT_LIGHT_SPOT;
                r.pos[0] = sl->lightPos.x; r.pos[1] = sl->lightPos.y; r.pos[2] = sl->lightPos.z;
                sl->Intensity.ToRGB(r.intensity);
                r.area = sl->cosTotalWidth; r.marg_int = sl->cosFalloffStart;
            } else if (const DistantLight *dsl = dynamic_cast<const DistantLight *>(l)) { // ABI 8 (lights/distant.cpp)
                r.kind = HPT_LIGHT_DISTANT;
                r.pos[0] = dsl->lightDir.x; r.pos[1] = dsl->lightDir.y; r.pos[2] = dsl->lightDir.z;
                dsl->L.ToRGB(r.intensity);
            } else if (const DiffuseAreaLight *dl = dynamic_cast<const DiffuseAreaLight *>(l)) {
                r.kind = HPT_LIGHT_DIFFUSE_AREA;
                dl->Lemit.ToRGB(r.intensity);
                r.area = dl->area;
                // quadric index is patched in Finish() once the shapes are known
            } else if (const InfiniteAreaLight *il = dynamic_cast<const InfiniteAreaLight *>(l)) {
                r.kind = HPT_LIGHT_INFINITE;
                const MIPMap<RGBSpectrum> *mm = il->radianceMap;
                const BlockedArray<RGBSpectrum> &l0 = *mm->pyramid[0];
                int w = (int)l0.uSize(), h = (int)l0.vSize();
                r.env_w = w; r.env_h = h;
                std::vector<float> tex(3 * (size_t)w * h);
                for (int v = 0; v < h; ++v)
                    for (int u = 0; u < w; ++u) l0(u, v).ToRGB(&tex[3 * ((size_t)v * w + u)]);
                r.tex_off = PushF(&tex[0], tex.size());
                const Distribution2D *d2 = il->distribution;
                if ((int)d2->pConditionalV.size() != h || d2->pMarginal->count != h)
                    Severe("hip renderer: unexpected Distribution2D shape");
                std::vector<float> cf, cc, ci;
                for (int v = 0; v < h; ++v) {
                    const Distribution1D *c = d2->pConditionalV[v];
                    if (c->count != w) Severe("hip renderer: unexpected Distribution1D size");
                    cf.insert(cf.end(), c->func, c->func + w);
                    cc.insert(cc.end(), c->cdf, c->cdf + w + 1);
                    ci.push_back(c->funcInt);
                }
                r.cond_func_off = PushF(&cf[0], cf.size());
                r.cond_cdf_off = PushF(&cc[0], cc.size());
                r.cond_int_off = PushF(&ci[0], ci.size());
                r.marg_func_off = PushF(d2->pMarginal->func, h);
                r.marg_cdf_off = PushF(d2->pMarginal->cdf, h + 1);
                r.marg_int = d2->pMarginal->funcInt;
            } else
                Severe("hip renderer: light type outside the hot-path scope "
                       "(supported: point, spot, distant, area/diffuse, infinite)");
            lights[i] = r;
        }
    }

    void Flatten(const Scene *scene) {
        AddLights(scene);
        // the refined primitive list: of the plugin's own list aggregate (the patched MakeScene), or of the reference's BVHAccel
        const vector<Reference<Primitive> > *prims = NULL;
        if (const HipListAggregate *la = dynamic_cast<const HipListAggregate *>(scene->aggregate)) prims = &la->primitives;
        else if (const BVHAccel *bvh = dynamic_cast<const BVHAccel *>(scene->aggregate)) prims = &bvh->primitives;
        if (!prims) Severe("hip renderer: Accelerator must be \"bvh\" (the default)");
        for (size_t i = 0; i < prims->size(); ++i) {
            if (const TransformedPrimitive *tp =
                    dynamic_cast<const TransformedPrimitive *>((*prims)[i].GetPtr())) {
                AddInstance(tp);
                continue;
            }
            const GeometricPrimitive *gp =
                dynamic_cast<const GeometricPrimitive *>((*prims)[i].GetPtr());
            if (!gp) Severe("hip renderer: primitive type outside the hot-path scope");
            const Shape *shape = gp->shape.GetPtr();
            if (const Triangle *tri = dynamic_cast<const Triangle *>(shape))
                AddTriangle(tri, gp);
            else
                AddQuadric(shape, gp);
        }
        // DiffuseAreaLight -> its ShapeSet (core/light.cpp:114-139).  One quadric: the very Shape object of the GeometricPrimitive
        // (core/api.cpp:1003-1009).  Anything else — a triangle mesh refined into its own Triangle objects, several shapes — is
        // listed shape by shape, in the set's order, with the areas and the area distribution the reference built.
        for (uint32_t i = 0; i < scene->lights.size(); ++i) {
            const DiffuseAreaLight *dl = dynamic_cast<const DiffuseAreaLight *>(scene->lights[i]);
            if (!dl) continue;
            const ShapeSet *set = dl->shapeSet;
            lights[i].set_off = lights[i].set_area_off = -1; lights[i].set_n = 0;
            if (set->shapes.size() == 1) {
                std::map<const Shape *, int>::iterator it = quadricIndex.find(set->shapes[0].GetPtr());
                if (it != quadricIndex.end()) { lights[i].quadric = it->second; continue; }
            }
            std::vector<int> ids;
            for (size_t k = 0; k < set->shapes.size(); ++k) {
                const Shape *sh = set->shapes[k].GetPtr();
                if (const Triangle *tri = dynamic_cast<const Triangle *>(sh)) {
                    std::map<const TriangleMesh *, int>::iterator mi = meshIndex.find(tri->mesh.GetPtr());
                    if (mi == meshIndex.end()) Severe("hip renderer: area light triangle's mesh not found in the scene");
                    ids.push_back(0);
                    ids.push_back((int)(meshTriBase[mi->second] + (tri->v - tri->mesh->vertexIndex) / 3));
                } else {
                    std::map<const Shape *, int>::iterator qi = quadricIndex.find(sh);
                    if (qi == quadricIndex.end()) Severe("hip renderer: area light shape not found among the scene's shapes");
                    ids.push_back(1); ids.push_back(qi->second);
                }
            }
            const Distribution1D *ad = set->areaDistribution;
            if ((size_t)ad->count != set->shapes.size() || set->areas.size() != set->shapes.size())
                Severe("hip renderer: unexpected ShapeSet area distribution");
            lights[i].quadric = -1;
            lights[i].set_n = (int)set->shapes.size();
            lights[i].set_off = PushI(&ids[0], ids.size());
            lights[i].set_area_off = PushF(&set->areas[0], set->areas.size());
            PushF(ad->cdf, set->shapes.size() + 1);
            PushF(&ad->funcInt, 1);
        }
    }

    hpt_scene_desc Desc() const {
        hpt_scene_desc d;
        memset(&d, 0, sizeof(d));
        d.meshes = meshes.empty() ? NULL : &meshes[0];        d.n_meshes = (int)meshes.size();
        d.quadrics = quadrics.empty() ? NULL : &quadrics[0];  d.n_quadrics = (int)quadrics.size();
        d.materials = materials.empty() ? NULL : &materials[0]; d.n_materials = (int)materials.size();
        d.lights = lights.empty() ? NULL : &lights[0];        d.n_lights = (int)lights.size();
        d.instances = instances.empty() ? NULL : &instances[0]; d.n_instances = (int)instances.size();
        d.fpool = fpool.empty() ? NULL : &fpool[0];           d.n_f = (int64_t)fpool.size();
        d.ipool = ipool.empty() ? NULL : &ipool[0];           d.n_i = (int64_t)ipool.size();
        d.textures = textures.empty() ? NULL : &textures[0];  d.n_textures = (int)textures.size();
        return d;
    }
};

} // namespace

// ---- HipListAggregate ----------------------------------------------------------------------------------------------------------------
HipListAggregate::HipListAggregate(const vector<Reference<Primitive> > &prims) {
    for (uint32_t i = 0; i < prims.size(); ++i) prims[i]->FullyRefine(primitives);      // BVHAccel's own first step (accelerators/bvh.cpp:160-162): same list, same order
    for (uint32_t i = 0; i < primitives.size(); ++i) bounds = Union(bounds, primitives[i]->WorldBound());
}
bool HipListAggregate::Intersect(const Ray &ray, Intersection *isect) const {
    bool hit = false;
    for (uint32_t i = 0; i < primitives.size(); ++i) if (primitives[i]->Intersect(ray, isect)) hit = true;   // (GeometricPrimitive::Intersect shrinks ray.maxt)
    return hit;
}
bool HipListAggregate::IntersectP(const Ray &ray) const {
    for (uint32_t i = 0; i < primitives.size(); ++i) if (primitives[i]->IntersectP(ray)) return true;
    return false;
}
void HipRendererWarmup(const ParamSet &rendererParams) {
    if (getenv("HPT_NO_WARMUP")) return;
    (void)hpt_warmup(rendererParams.FindOneInt("device", 0));
}
Primitive *MakeHipAggregate(const vector<Reference<Primitive> > &prims) {
    if (const char *e = getenv("HPT_HOST_BVH")) if (atoi(e) != 0) return NULL;
    return new HipListAggregate(prims);
}

// wall clock of the plugin's stages (HPT_TIMING=1: one line on stderr; bench.py's end_to_end leg reads it)
static double NowS() { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + 1e-9 * ts.tv_nsec; }
static double g_render_done_s = 0.;
static void FastExit() { fflush(NULL); _exit(0); }
static double SinceProcessStartS();
static void ReportExitS() {
    fprintf(stderr, "hpt timing: Render() return to exit() %.3f s (pbrtCleanup: the scene's destructors), whole process so far %.3f s\n",
            NowS() - g_render_done_s, SinceProcessStartS());
}
static const double g_loaded_s = NowS();      // static initialisation of this object: the dynamic linker is done, main() not yet entered
// seconds since the kernel created this process (field 22 of /proc/self/stat: start time in clock ticks after boot; CLOCK_BOOTTIME now)
static double SinceProcessStartS() {
    FILE *f = fopen("/proc/self/stat", "r");
    if (!f) return -1.;
    char buf[1024]; size_t n = fread(buf, 1, sizeof(buf) - 1, f); fclose(f); buf[n] = 0;
    const char *p = strrchr(buf, ')');
    if (!p) return -1.;
    unsigned long long start = 0; int field = 2;
    for (p = p + 1; *p && field < 22; ++p) if (*p == ' ') { ++field; if (field == 22) { start = strtoull(p + 1, NULL, 10); break; } }
    struct timespec ts; clock_gettime(CLOCK_BOOTTIME, &ts);
    return ts.tv_sec + 1e-9 * ts.tv_nsec - (double)start / (double)sysconf(_SC_CLK_TCK);
}

HipPathRenderer::HipPathRenderer(Sampler *s, Camera *c, SurfaceIntegrator *si,
                                 VolumeIntegrator *vi, const ParamSet &params) {
    sampler = s;
    camera = c;
    surfaceIntegrator = si;
    volumeIntegrator = vi;
    device = params.FindOneInt("device", 0);
    gpus = params.FindOneInt("gpus", 1);
    if (const char *e = getenv("HPT_GPUS")) gpus = atoi(e);
    if (gpus < 1) gpus = 1;
    seed = (unsigned)params.FindOneInt("seed", 0);
    std::string sm = params.FindOneString("sampler", "ldhash");
    if (sm == "ldhash") samplerMode = HPT_SAMPLER_LD_HASH;
    else if (sm == "mtreplay") samplerMode = HPT_SAMPLER_MT_REPLAY;
    else { Warning("hip renderer: unknown sampler mode \"%s\"; using ldhash", sm.c_str());
           samplerMode = HPT_SAMPLER_LD_HASH; }
    if (const char *e = getenv("HPT_SAMPLER"))
        samplerMode = !strcmp(e, "mtreplay") ? HPT_SAMPLER_MT_REPLAY : HPT_SAMPLER_LD_HASH;
    dumpPath = params.FindOneString("dumpscene", "");
    if (const char *e = getenv("HPT_DUMP_SCENE")) dumpPath = e;
    // "string tunecache" ["<dir>"]: the directory the library may remember this scene's kernel configuration in (the library writes nothing
    // unasked: $HPT_TUNE_CACHE, which this sets when the environment has not)
    std::string tc = params.FindOneString("tunecache", "");
    if (tc != "") setenv("HPT_TUNE_CACHE", tc.c_str(), 0);
}

HipPathRenderer::~HipPathRenderer() {
    delete sampler;
    delete camera;
    delete surfaceIntegrator;
    delete volumeIntegrator;
}

void HipPathRenderer::Render(const Scene *scene) {
    // --- what the device path replaces must be exactly what pbrt was asked to run ---------
    const PerspectiveCamera *pc = dynamic_cast<const PerspectiveCamera *>(camera);
    if (!pc) Severe("hip renderer: Camera must be \"perspective\"");
    ImageFilm *film = dynamic_cast<ImageFilm *>(camera->film);
    if (!film) Severe("hip renderer: Film must be \"image\"");
    // PixelFilter: any Filter plugin.  The device looks weights up in ImageFilm's own 16x16 table (film/image.cpp:56-68),
    // so the floats it uses are the ones this build's filter->Evaluate produced.  The default box of width 0.5 (every
    // sample lands in its own pixel) keeps the table-free fast path.
    const BoxFilter *box = dynamic_cast<const BoxFilter *>(film->filter);
    const bool defaultBox = box && box->xWidth == 0.5f && box->yWidth == 0.5f;
    hpt_filter flt;
    memset(&flt, 0, sizeof(flt));
    flt.xwidth = film->filter->xWidth; flt.ywidth = film->filter->yWidth;
    memcpy(flt.table, film->filterTable, sizeof(flt.table));
    const LDSampler *lds = dynamic_cast<const LDSampler *>(sampler);
    const RandomSampler *rnds = dynamic_cast<const RandomSampler *>(sampler);
    const StratifiedSampler *strat = dynamic_cast<const StratifiedSampler *>(sampler);
    const HaltonSampler *halt = dynamic_cast<const HaltonSampler *>(sampler);
    const AdaptiveSampler *adapt = dynamic_cast<const AdaptiveSampler *>(sampler);
    const BestCandidateSampler *bcs = dynamic_cast<const BestCandidateSampler *>(sampler);
    if (!lds && !rnds && !strat && !halt && !adapt && !bcs) Severe("hip renderer: Sampler must be \"lowdiscrepancy\", \"random\", \"stratified\", \"halton\", \"adaptive\" or \"bestcandidate\"");
    if (adapt && adapt->method != AdaptiveSampler::ADAPTIVE_CONTRAST_THRESHOLD) Severe("hip renderer: Sampler \"adaptive\" with method \"shapeid\" is outside the scope (the device carries no Intersection ids); use \"contrast\"");
    const PathIntegrator *path = dynamic_cast<const PathIntegrator *>(surfaceIntegrator);
    const DirectLightingIntegrator *direct = dynamic_cast<const DirectLightingIntegrator *>(surfaceIntegrator);
    if (!path && !direct) Severe("hip renderer: SurfaceIntegrator must be \"path\" or \"directlighting\"");
    if (scene->volumeRegion) Severe("hip renderer: participating media are outside the scope");

    const double t_begin = NowS();
    Flattener fl;
    fl.Flatten(scene);
    hpt_scene_desc desc = fl.Desc();
    const double t_flat = NowS();

    hpt_camera cam;
    memset(&cam, 0, sizeof(cam));
    CopyM(pc->RasterToCamera.m, cam.raster_to_camera);
    CopyM(pc->CameraToWorld.startTransform->m, cam.camera_to_world);
    cam.lens_radius = pc->lensRadius;
    cam.focal_distance = pc->focalDistance;
    cam.shutter_open = pc->shutterOpen;
    cam.shutter_close = pc->shutterClose;
    // a moving camera (ActiveTransform StartTime / EndTime around the camera's CTM): CameraToWorld as the AnimatedTransform it is
    const bool movingCamera = pc->CameraToWorld.actuallyAnimated;
    hpt_instance camMotion;
    Flattener::FillAnimated(pc->CameraToWorld, &camMotion);

    hpt_render_desc rd;
    memset(&rd, 0, sizeof(rd));
    rd.xres = film->xResolution; rd.yres = film->yResolution;
    rd.x_start = film->xPixelStart; rd.x_count = film->xPixelCount;
    rd.y_start = film->yPixelStart; rd.y_count = film->yPixelCount;
    rd.spp = lds ? lds->nPixelSamples : rnds ? rnds->nSamples : halt ? halt->samplesPerPixel : adapt ? adapt->maxSamples : bcs ? bcs->samplesPerPixel
                 : strat->xPixelSamples * strat->yPixelSamples;
    rd.maxdepth = path ? path->maxDepth : direct->maxDepth;
    rd.integrator = path ? HPT_INTEGRATOR_PATH
                         : (direct->strategy == SAMPLE_ALL_UNIFORM ? HPT_INTEGRATOR_DIRECT_ALL : HPT_INTEGRATOR_DIRECT_ONE);
    rd.sampler_mode = lds ? samplerMode : (samplerMode == HPT_SAMPLER_MT_REPLAY ? HPT_SAMPLER_RANDOM_MT_REPLAY : HPT_SAMPLER_RANDOM_HASH);
    if (strat) {
        if (strat->xPixelSamples > 0xfff) Severe("hip renderer: stratified xsamples above 4095");
        rd.sampler_mode = HPT_SAMPLER_STRATIFIED(samplerMode == HPT_SAMPLER_MT_REPLAY ? HPT_SAMPLER_STRATIFIED_MT_REPLAY : HPT_SAMPLER_STRATIFIED_HASH,
                                                 strat->xPixelSamples, strat->jitterSamples);
    }
    if (adapt) rd.sampler_mode = HPT_SAMPLER_ADAPTIVE(samplerMode == HPT_SAMPLER_MT_REPLAY ? HPT_SAMPLER_ADAPTIVE_MT_REPLAY : HPT_SAMPLER_ADAPTIVE_HASH, adapt->minSamples);   // samplers/adaptive.cpp:44-83 (both counts already powers of two)
    if (bcs) rd.sampler_mode = samplerMode == HPT_SAMPLER_MT_REPLAY ? HPT_SAMPLER_BESTCANDIDATE_MT_REPLAY : HPT_SAMPLER_BESTCANDIDATE_HASH;   // samplers/bestcandidate.cpp:50-91
    if (halt) rd.sampler_mode = samplerMode == HPT_SAMPLER_MT_REPLAY ? HPT_SAMPLER_HALTON_MT_REPLAY : HPT_SAMPLER_HALTON_HASH;   // samplers/halton.cpp:54-80
    rd.seed = seed;
    // nTasks exactly as SamplerRenderer::Render computes it (samplerrenderer.cpp:203-205)
    int nPixels = film->xResolution * film->yResolution;
    rd.ntasks = (int)RoundUpPow2((uint32_t)max(32 * NumSystemCores(), nPixels / (16 * 16)));
    rd.shard_rank = 0; rd.shard_count = 1;

    if (dumpPath != "") {
        if (hpt_blob_save(dumpPath.c_str(), &desc, &cam, &rd) != HPT_OK)
            Severe("hip renderer: %s", hpt_last_error());
        if (movingCamera) {   // sidecar: the hpt_instance record of CameraToWorld
            string mpath = dumpPath + ".camera_motion";
            FILE *mf = fopen(mpath.c_str(), "wb");
            if (!mf || fwrite(&camMotion, sizeof(camMotion), 1, mf) != 1) Severe("hip renderer: cannot write %s", mpath.c_str());
            fclose(mf);
        }
        if (bcs) {           // sidecar: the sampler's table (BestCandidateSampler::sampleTable: 4096 x 5 floats)
            string tpath = dumpPath + ".sampletable";
            FILE *tf = fopen(tpath.c_str(), "wb");
            if (!tf || fwrite(&BestCandidateSampler::sampleTable[0][0], sizeof(float), 5 * SAMPLE_TABLE_SIZE, tf) != 5 * SAMPLE_TABLE_SIZE) Severe("hip renderer: cannot write %s", tpath.c_str());
            fclose(tf);
        }
        if (!defaultBox) {   // sidecar: 258 floats {xwidth, ywidth, table[256]} = hpt_filter
            string fpath = dumpPath + ".filter";
            FILE *ff = fopen(fpath.c_str(), "wb");
            if (!ff || fwrite(&flt, sizeof(flt), 1, ff) != 1) Severe("hip renderer: cannot write %s", fpath.c_str());
            fclose(ff);
        }
        Info("hip renderer: scene blob written to %s; not rendering", dumpPath.c_str());
        return;
    }

    std::vector<float> xyzw(4 * (size_t)rd.x_count * rd.y_count);
    hpt_stats st;
    memset(&st, 0, sizeof(st));
    double t_create = t_flat, t_tune = t_flat, t_render = t_flat, bvh_ms = 0.;
    ProgressReporter reporter(1, "Rendering (HIP)");
    if (gpus > 1) {
        // `gpus` — one host thread per device inside the library, pixel tiles round-robin, one film gather over RCCL
        std::vector<int> devs;
        for (int i = 0; i < gpus; ++i) devs.push_back(device + i);
        if (const char *e = getenv("HPT_GPU_LIST")) {               // explicit list, repeats allowed ("0,0": two shards on one GPU)
            devs.clear();
            for (const char *p = e; *p;) { devs.push_back(atoi(p)); while (*p && *p != ',') ++p; if (*p == ',') ++p; }
        }
        hpt_multi *hm = hpt_multi_create(&desc, &devs[0], (int)devs.size());
        if (!hm) Severe("hip renderer: %s", hpt_last_error());
        if (!defaultBox && hpt_multi_set_filter(hm, &flt) != HPT_OK) Severe("hip renderer: %s", hpt_last_error());
        if (movingCamera && hpt_multi_set_camera_motion(hm, &camMotion) != HPT_OK) Severe("hip renderer: %s", hpt_last_error());
        if (bcs && hpt_multi_set_sample_table(hm, &BestCandidateSampler::sampleTable[0][0], SAMPLE_TABLE_SIZE) != HPT_OK) Severe("hip renderer: %s", hpt_last_error());
        std::vector<hpt_stats> sts(devs.size());
        if (hpt_multi_render(hm, &cam, &rd, &xyzw[0], &sts[0]) != HPT_OK) Severe("hip renderer: %s", hpt_last_error());
        hpt_multi_destroy(hm);
        for (size_t i = 0; i < sts.size(); ++i) {
            st.camera_samples += sts[i].camera_samples; st.bad_samples += sts[i].bad_samples;
            if (sts[i].kernel_ms > st.kernel_ms) st.kernel_ms = sts[i].kernel_ms;
        }
    } else {
        hpt_scene *hs = hpt_scene_create(&desc, device);
        if (!hs) Severe("hip renderer: %s", hpt_last_error());
        t_create = NowS();
        if (!defaultBox && hpt_scene_set_filter(hs, &flt) != HPT_OK) Severe("hip renderer: %s", hpt_last_error());
        if (movingCamera && hpt_scene_set_camera_motion(hs, &camMotion) != HPT_OK) Severe("hip renderer: %s", hpt_last_error());
        if (bcs && hpt_scene_set_sample_table(hs, &BestCandidateSampler::sampleTable[0][0], SAMPLE_TABLE_SIZE) != HPT_OK) Severe("hip renderer: %s", hpt_last_error());
        // kernel configuration of a job big enough to repay it (what hpt_render would do on its own: here as a step of its own, timed):
        // probe renders, or the cached choice for this scene
        if ((int64_t)rd.x_count * rd.y_count * rd.spp >= ((int64_t)32 << 20) && hpt_scene_tune(hs, &cam, &rd) < 0)
            Severe("hip renderer: %s", hpt_last_error());
        t_tune = NowS();
        if (hpt_render(hs, &cam, &rd, &xyzw[0], &st) != HPT_OK)
            Severe("hip renderer: %s", hpt_last_error());
        t_render = NowS();
        hpt_scene_info info;
        if (hpt_scene_get_info(hs, &info) == HPT_OK) bvh_ms = info.build_ms;
        hpt_scene_destroy(hs);
    }
    reporter.Update();
    reporter.Done();
    if (st.bad_samples)
        Error("hip renderer: %llu camera samples had NaN / negative / infinite luminance and "
              "were set to black", (unsigned long long)st.bad_samples);
    Info("hip renderer: %.3f Msamples/s (%.2f ms kernel)",
         st.camera_samples / (st.kernel_ms * 1e3), st.kernel_ms);

    // Hand the film to ImageFilm so the reference's own WriteImage (film/image.cpp:178) runs.
    for (int y = 0; y < rd.y_count; ++y)
        for (int x = 0; x < rd.x_count; ++x) {
            ImageFilm::Pixel &px = (*film->pixels)(x, y);
            const float *s = &xyzw[4 * ((size_t)y * rd.x_count + x)];
            px.Lxyz[0] = s[0]; px.Lxyz[1] = s[1]; px.Lxyz[2] = s[2];
            px.weightSum = s[3];
        }
    const double t_film = NowS();
    camera->film->WriteImage();
    // HPT_FAST_EXIT=1: once main() returns, leave without the HIP runtime's teardown (unloading 23 MB of code objects, hsa_shut_down: 70-90 ms
    // of a 0.35 s job).  The image is written and every stream flushed; the exit status becomes 0, which is why this is opt-in.
    if (getenv("HPT_FAST_EXIT")) atexit(FastExit);
    if (getenv("HPT_TIMING")) {
        g_render_done_s = NowS();
        atexit(ReportExitS);       // (handlers run in reverse order of registration: after pbrtCleanup and main's return, before the HIP runtime's own teardown)
        const double now = NowS(), since = SinceProcessStartS();
        fprintf(stderr, "hpt timing: exec + dynamic linking %.3f s, pbrt parse + scene construction %.3f s (up to Render())\n",
                since - (now - g_loaded_s), t_begin - g_loaded_s);
    }
    if (getenv("HPT_TIMING"))
        fprintf(stderr, "hpt timing: flatten %.3f s, scene create %.3f s (BVH build %.1f ms), kernel configuration %.3f s, render + film download %.3f s "
                        "(kernel %.2f ms), film to ImageFilm %.3f s, WriteImage %.3f s\n", t_flat - t_begin, t_create - t_flat, bvh_ms, t_tune - t_create,
                t_render - t_tune, st.kernel_ms, t_film - t_render, NowS() - t_film);
}

// Renderer::Li / Transmittance (core/renderer.h:47-53): what the reference's integrators call back into for a ray of their own
// (SpecularReflect / SpecularTransmit, core/integrator.cpp:177-258; irradiance caching, photon mapping ...).  Nothing on the device path
// calls them — the kernel owns the whole per-sample loop — but the interface is honoured for host-side callers: the same evaluation
// SamplerRenderer::Li performs (renderers/samplerrenderer.cpp:225-247), with the plugins this renderer owns, on the CPU.
Spectrum HipPathRenderer::Li(const Scene *scene, const RayDifferential &ray, const Sample *sample, RNG &rng,
                             MemoryArena &arena, Intersection *isect, Spectrum *T) const {
    Spectrum localT;
    if (!T) T = &localT;
    Intersection localIsect;
    if (!isect) isect = &localIsect;
    Spectrum Li = 0.f;
    if (scene->Intersect(ray, isect))
        Li = surfaceIntegrator->Li(scene, this, ray, *isect, sample, rng, arena);
    else
        for (uint32_t i = 0; i < scene->lights.size(); ++i) Li += scene->lights[i]->Le(ray);
    Spectrum Lvi = volumeIntegrator->Li(scene, this, ray, sample, rng, T, arena);
    return *T * Li + Lvi;
}

Spectrum HipPathRenderer::Transmittance(const Scene *scene, const RayDifferential &ray, const Sample *sample,
                                        RNG &rng, MemoryArena &arena) const {
    return volumeIntegrator->Transmittance(scene, this, ray, sample, rng, arena);
}
