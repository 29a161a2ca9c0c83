// Internals of the region tail shared by regs_post.cpp (single-end) and pair_post.cpp (paired-end).
#pragma once
#include <cstdint>
#include "bmh_internal.h"

namespace rp {

struct Reg {
	int64_t rb, re; int qb, qe, rid, score, truesc, sub, csub, sub_n, w, seedcov, secondary, n_comp, is_alt, secondary_all, alt_sc;
	float frac_rep; uint64_t hash;
};

struct Ctx { const bmh_chain_opt_t *co; const bmh_ext_params_t *ep; const bmh_post_opt_t *po; int64_t l_pac; const uint8_t *pac;
             int n_contigs; const int64_t *ctg_off; };

int text_base(const uint8_t *pac, int64_t l_pac, int64_t i);
int pos2rid(const Ctx &x, int64_t pos_f);
int sort_dedup_patch(const Ctx &x, const uint8_t *query, int n, Reg *a);     // query == nullptr: no patching (mem_matesw's call)
int mark_primary(const Ctx &x, int n, Reg *a, int64_t id);                    // returns n_pri: the hits on the primary assembly (all of them without ALT contigs)
inline bool alt_mode(const Ctx &x) { return x.po->contig_is_alt != nullptr; }
void set_is_alt(const Ctx &x, int n, Reg *a);                                // p->is_alt = bns->anns[p->rid].is_alt (src/bwamem.c:2321-2325)
int approx_mapq(const Ctx &x, const Reg &a);
uint64_t hash64(uint64_t key);
void reg_from_record(const Ctx &x, const int32_t *g, float frac_rep, Reg &p);   // {read, score, qb, qe, rb, re} record -> Reg

} // namespace rp
