// hpt_wavefront.h — interface of the multi-kernel wavefront pipeline (hpt_wavefront.hip).
#ifndef HPT_WAVEFRONT_H
#define HPT_WAVEFRONT_H
#include <hip/hip_runtime.h>
#include "hpt_path.h"

#ifndef HPT_WF_TRACE_WAVES
#define HPT_WF_TRACE_WAVES 6   /* __launch_bounds__ waves/SIMD of the trace kernel (<= 80 VGPRs) */
#endif

namespace hpt {

struct WfArgs {
    DScene sc;
    RenderParams rp;
    float *film;
    unsigned long long *next_item;   // global work counter (pixel x sample-chunk items)
    WorkCounters *counters;
    float4 *state;                   // [WF_STATE_VEC][P] serialized Lane state
    float4 *rays;                    // [2][P]  {o.xyz, mint} {d.xyz, maxt}
    float4 *hits;                    // [P]     {t, b1, b2, prim}  (in: {time,..} for instanced scenes)
    int *hit_inst;                   // [P]     instance of the hit (instanced scenes)
    int *queue;                      // [P]     compacted slots with a pending ray; bit 31 = any-hit
    int *qcount;                     // [2]     queue length, ping-pong by iteration parity
    int *qhead;                      // [2]     trace kernel's fetch cursor
    int64_t P;                       // path slots (multiple of HPT_BLOCK)
    int parity;
};

hipError_t wf_launch_advance(int mats, const WfArgs &a, bool count, hipStream_t s);
hipError_t wf_launch_trace(const WfArgs &a, int grid, bool count, int bvh_depth, hipStream_t s);
int wf_trace_occupancy(bool inst, int bvh_depth, int *blocks_per_cu, int *vgprs);

} // namespace hpt
#endif
