// hip_renderer.h — the reference-side binding: a pbrt-v2 Renderer plugin that hands the
// SamplerRenderer + PathIntegrator hot loop to the MI355X library through the C ABI of
// include/hpt.h.  Compiles against the reference headers where they lie (/root/reference/src);
// see INTEGRATION.md for the one-branch patch to RenderOptions::MakeRenderer (core/api.cpp:1252).
//
// Mirrors SamplerRenderer's interface (renderers/samplerrenderer.h:44-62): same constructor
// arguments, same ownership (the renderer owns and deletes sampler, camera, both integrators —
// samplerrenderer.cpp:180-185), same error convention (Error()/Severe(), core/error.h:49-52).
#ifndef PBRT_RENDERERS_HIPRENDERER_H
#define PBRT_RENDERERS_HIPRENDERER_H

#include "pbrt.h"
#include "renderer.h"
#include "paramset.h"
#include "primitive.h"

// What RenderOptions::MakeScene hands the hip renderer instead of a BVHAccel (core/api.cpp:1186-1203, patched): the fully refined
// primitive list and its bound — all the plugin reads of the reference's accelerator (BVHAccel::primitives).  The device library
// builds its own tree from the flattened triangles; pbrt's CPU BVH build (0.1-0.7 s of a sub-second job) would be thrown away.
// Intersect / IntersectP walk the list: correct for the Renderer::Li callback of host-side callers, never on the device path.
// HPT_HOST_BVH=1 keeps the reference's accelerator.
class HipListAggregate : public Aggregate {
public:
    HipListAggregate(const vector<Reference<Primitive> > &prims);
    BBox WorldBound() const { return bounds; }
    bool CanIntersect() const { return true; }
    bool Intersect(const Ray &ray, Intersection *isect) const;
    bool IntersectP(const Ray &ray) const;
    vector<Reference<Primitive> > primitives;
private:
    BBox bounds;
};
// called from pbrtWorldBegin (api_hip_renderer.patch): starts the HIP runtime on the device the renderer will use while the world is parsed
void HipRendererWarmup(const ParamSet &rendererParams);
Primitive *MakeHipAggregate(const vector<Reference<Primitive> > &prims);   // NULL: use the scene's own accelerator (HPT_HOST_BVH=1)

class HipPathRenderer : public Renderer {
public:
    HipPathRenderer(Sampler *s, Camera *c, SurfaceIntegrator *si, VolumeIntegrator *vi,
                    const ParamSet &params);
    ~HipPathRenderer();
    // Renderer interface (core/renderer.h:43-54)
    void Render(const Scene *scene);
    Spectrum Li(const Scene *scene, const RayDifferential &ray, const Sample *sample, RNG &rng,
                MemoryArena &arena, Intersection *isect = NULL, Spectrum *T = NULL) const;
    Spectrum Transmittance(const Scene *scene, const RayDifferential &ray, const Sample *sample,
                           RNG &rng, MemoryArena &arena) const;

private:
    Sampler *sampler;
    Camera *camera;
    SurfaceIntegrator *surfaceIntegrator;
    VolumeIntegrator *volumeIntegrator;
    int device;        // "integer device"  [0]
    int gpus;          // "integer gpus"    [1]: the frame's pixel tiles sharded over devices device .. device + gpus - 1 (hpt_multi, include/hpt.h)
    int samplerMode;   // "string sampler"  ["ldhash" | "mtreplay"]
    unsigned seed;     // "integer seed"    [0]
    std::string dumpPath; // "string dumpscene" [""] or env HPT_DUMP_SCENE: write blob, do not render
};

#endif
