// hpt_internal.h — declarations shared by the translation units of libhpt.so.
#ifndef HPT_INTERNAL_H
#define HPT_INTERNAL_H
#include "../../include/hpt.h"

void hpt_set_error(const char *fmt, ...) __attribute__((format(printf, 1, 2)));
int hpt_validate_desc(const hpt_scene_desc *d);
extern "C" void hpt_abi_sizes(int32_t out[10]);
// hpt_render_device that ADDS to the film instead of clearing it first (hpt_multi's dynamic hand-out: a device renders several disjoint
// sub-shards into one film)
int hpt_render_device_into(hpt_scene *s, const hpt_camera *cam, const hpt_render_desc *rd, void *d_film, void *stream, hpt_stats *stats, bool clear_film);

#endif
