// Test driver: the per-read region tail the device kernels instantiate (bwa-mem_gpu_amd/csrc/regs_core.h) compiled as plain C++,
// behind the signature of bmh_finalize_regs, so that it can be checked on a machine without a GPU against the host
// form (regs_post.cpp).  Test infrastructure only.
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>
#include "../bwa-mem_gpu_amd/csrc/regs_core.h"

extern "C" int64_t regs_core_run(const bmh_chain_opt_t *copt, const bmh_ext_params_t *ep, const bmh_post_opt_t *popt, int64_t l_pac,
                                 const uint8_t *pac, uint32_t n_reads, const uint8_t *reads, const uint64_t *read_offs,
                                 const int32_t *regs_in, const uint32_t *regs_per_read, const float *frac_rep,
                                 int n_contigs, const int64_t *contig_offset, int32_t *out, uint32_t *out_per_read, int with_dp)
{
	using namespace regs_core;
	std::vector<double> logtab(1 << 16);
	for (int k = 0; k < (int)logtab.size(); ++k) logtab[k] = log((double)k);
	std::vector<int32_t> dph(1 << 16), dpe(1 << 16);
	ctx_t x; memset(&x, 0, sizeof(x));
	x.co = *copt; x.ep = *ep; x.po = *popt; x.l_pac = l_pac; x.pac = pac; x.n_contigs = n_contigs; x.ctg_off = contig_offset;
	x.logtab = logtab.data(); x.n_log = (int)logtab.size();
	x.ctg_alt = popt->contig_is_alt;                              // (ALT contigs: the table is host memory here)
	if (with_dp) { x.dp_h = dph.data(); x.dp_e = dpe.data(); x.dp_cap = (int)dph.size(); }
	uint64_t in = 0, w = 0;
	std::vector<rec_t> a; std::vector<int32_t> z;
	for (uint32_t r = 0; r < n_reads; ++r) {
		const int n_in = (int)regs_per_read[r];
		a.assign(n_in + 1, rec_t()); z.assign(n_in + 1, 0);
		for (int i = 0; i < n_in; ++i) { memset(&a[i], 0x5a, sizeof(rec_t)); memcpy(a[i].v, regs_in + 8 * (in + i), 32); }      // [8..15]: garbage in
		const int n = finalize_read<false>(x, reads + read_offs[r], r, popt->id0 + r, frac_rep ? frac_rep[r] : 0.f, n_in, a.data(), z.data());
		if (n < 0) return n;
		for (int i = 0; i < n; ++i) memcpy(out + 16 * (w + i), a[i].v, 64);
		out_per_read[r] = (uint32_t)n;
		in += n_in; w += n;
	}
	return (int64_t)w;
}
