// Internal declarations shared by the translation units of libbwamem_hip.so.
#pragma once
#include "../../include/bwamem_hip.h"
#include "fmd_dev.h"

struct bmh_index {
	fmd_dev_t dev;
	bool owns;             // arrays were hipMalloc'd by bmh_index_upload
	uint64_t n_words;
};

#ifdef __cplusplus
extern "C" {
#endif
void bmh_set_error(const char *fmt, ...) __attribute__((format(printf, 1, 2)));
#ifdef __cplusplus
}
#endif
