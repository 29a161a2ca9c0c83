// Local alignment with start positions and second-best score, as the reference's mate rescue uses it (ksw_align2 with
// KSW_XSUBO | KSW_XSTART [| KSW_XBYTE], /root/reference/src/ksw.c:389-740).  The reference computes it with a striped
// SSE2 kernel whose results depend on the striping in two places: E(i+1,j) is taken from H(i,j) BEFORE the lazy-F
// correction (ksw.c:485-493 vs :499-511), and zero-score padding lanes take part in the row maximum.  To return the same
// numbers this host routine walks the same segments (16 lanes of 8 bits or 8 lanes of 16 bits) in scalar code.
#pragma once
#include <cstdint>
#include "../../include/bwamem_hip.h"

struct bmh_sw_result_t { int score, te, qe, score2, te2, tb, qb; };      // kswr_t

#define BMH_SW_XBYTE  0x10000
#define BMH_SW_XSTOP  0x20000
#define BMH_SW_XSUBO  0x40000
#define BMH_SW_XSTART 0x80000

// query / target: codes 0..4 (both are reversed in place and restored, like the reference); xtra: flags | threshold
bmh_sw_result_t bmh_local_sw(int qlen, uint8_t *query, int tlen, uint8_t *target, const bmh_ext_params_t &p, int xtra);
