/* GASAL2/include/interfaces.h -- see gasal.h in this directory. */
#ifndef __GASAL_INTERFACES_H__
#define __GASAL_INTERFACES_H__
#include "gasal.h"
#include "args_parser.h"

void gasal_host_alns_resize(gasal_gpu_storage_t *gpu_storage, int new_max_alns, Parameters *params);
gasal_res_t *gasal_res_new_host(uint32_t max_n_alns, Parameters *params);
void gasal_res_destroy_host(gasal_res_t *res);
void gasal_set_device(int gpu_select = 0, bool isPrintingProp = true);
#endif
