// hpt_internal.h — declarations shared by the translation units of libhpt.so.
#ifndef HPT_INTERNAL_H
#define HPT_INTERNAL_H
#include "../../include/hpt.h"

void hpt_set_error(const char *fmt, ...) __attribute__((format(printf, 1, 2)));
int hpt_validate_desc(const hpt_scene_desc *d);
extern "C" void hpt_abi_sizes(int32_t out[10]);

#endif
