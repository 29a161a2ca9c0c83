// Test driver: compiles the per-read chaining core of the device job builder (bwa-mem_gpu_amd/csrc/chain_core.h)
// as plain C++ (serial form) so that its decisions can be checked on a machine without a GPU against
// bmh_build_jobs and the golden job stream of the reference's host code.  Test infrastructure only: the
// product path is the HIP kernels of csrc/chain_kernels.hip, which instantiate the same header on the device.
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "../bwa-mem_gpu_amd/csrc/chain_core.h"

struct result_t {
	uint64_t n_regs, n_jobs, q_bytes, t_bytes;
	uint32_t *regs_per_read, *qoff, *qlen, *toff, *tlen, *h0, *job_read, *job_reg, *job_side;
	uint8_t *q, *t;
	int err;
};

template <class T> static T *dup(const std::vector<T> &v) { T *p = (T *)malloc(sizeof(T) * (v.size() + 1)); if (!v.empty()) memcpy(p, v.data(), sizeof(T) * v.size()); return p; }

// compact != 0: the scratch records of the cooperative kernels' LDS (ch_compact_ty: 16-bit links and read coordinates) instead of the wide ones, in the form
// without the seed filter -- the same decisions are expected wherever that filter does not apply and no read samples 65 535 seeds or more
static result_t *run(const bmh_chain_opt_t *opt, int64_t l_pac, const uint8_t *pac, uint32_t n_reads, const uint8_t *reads,
                     const uint64_t *read_offs, const uint32_t *read_lens, const uint64_t *rbeg, const int32_t *qbeg,
                     const uint32_t *score, const uint32_t *n_ref, const uint32_t *prefix, uint64_t n_seeds, int compact)
{
	const size_t S = n_seeds + 1;
	std::vector<ch_seed_t> seeds(S); std::vector<ch_chain_t> chains(S); std::vector<uint32_t> order(S), klist(S), cidx(S);
	std::vector<int64_t> opos(S); std::vector<uint64_t> srt(S); std::vector<ch_reg_t> regs(S); std::vector<ch_est_t> est(S);
	std::vector<uint32_t> rpr(n_reads + 1), jpr(n_reads + 1); std::vector<float> frep(n_reads + 1);
	int err = 0;
	ch_ctx_t x; memset(&x, 0, sizeof(x));
	x.o = *opt; x.l_pac = l_pac; x.n_contigs = 1;
	x.rbeg = rbeg; x.qbeg = qbeg; x.score = score; x.n_ref = n_ref; x.prefix = prefix; x.read_lens = read_lens;
	x.g.S = seeds.data(); x.g.CH = chains.data(); x.g.order = order.data(); x.g.opos = opos.data(); x.g.klist = klist.data(); x.g.srt = srt.data();
	x.g.cidx = cidx.data(); x.g.E = est.data(); x.regs = regs.data(); x.regs_per_read = rpr.data(); x.jobs_per_read = jpr.data(); x.frac_rep = frep.data(); x.err = &err;
	std::vector<uint32_t> offs32(n_reads + 1);
	for (uint32_t r = 0; r < n_reads; ++r) offs32[r] = (uint32_t)read_offs[r];
	x.reads = reads; x.read_offs = offs32.data(); x.pac = pac;
	// (the form with the reference's seed filter: it runs for the reads the options make it apply to, as in the kernels launched for such options)
	if (!compact) for (uint32_t r = 0; r < n_reads; ++r) chain_core::chain_read<false, false, true>(x, r, chain_core::global_scratch(x, r));
	else {
		std::vector<ch_seed_c> cs(S); std::vector<ch_chain_c> cc(S); std::vector<ch_est_c> ce(S); std::vector<uint16_t> co(S), ck(S), ci(S);
		for (uint32_t r = 0; r < n_reads; ++r) {
			if (n_ref[r] >= 0xFFFFu) { err = 9; break; }
			const uint32_t b = prefix[r];
			ch_scr<ch_compact_ty> L;
			L.S = cs.data() + b; L.CH = cc.data() + b; L.E = ce.data() + b; L.order = co.data() + b; L.klist = ck.data() + b; L.cidx = ci.data() + b; L.opos = opos.data() + b; L.srt = srt.data() + b;
			chain_core::chain_read<false, false, false>(x, r, L);
		}
	}
	std::vector<uint32_t> qoff, qlen, toff, tlen, h0, job_read, job_reg, job_side;
	std::vector<uint8_t> q, t;
	auto text = [&](int64_t p) { const bool rev = p >= l_pac; const int64_t f = rev ? (l_pac << 1) - 1 - p : p; const int c = (pac[f >> 2] >> ((~f & 3) << 1)) & 3; return (uint8_t)(rev ? 3 - c : c); };
	uint32_t g = 0;
	for (uint32_t r = 0; r < n_reads; ++r) {
		const uint8_t *query = reads + read_offs[r];
		for (uint32_t i = 0; i < rpr[r]; ++i, ++g) {
			const ch_reg_t a = regs[prefix[r] + i];
			if (a.seed_qbeg > 0) {
				qoff.push_back((uint32_t)q.size()); toff.push_back((uint32_t)t.size()); qlen.push_back(a.seed_qbeg); tlen.push_back(a.lr); h0.push_back(a.seedlen0);
				job_read.push_back(r); job_reg.push_back(g); job_side.push_back(0);
				for (int k = 0; k < a.seed_qbeg; ++k) q.push_back(query[a.seed_qbeg - 1 - k]);
				for (int k = 0; k < a.lr; ++k) t.push_back(text(a.rmax0 + a.lr - 1 - k));
			}
			if (a.rq > 0) {
				qoff.push_back((uint32_t)q.size()); toff.push_back((uint32_t)t.size()); qlen.push_back(a.rq); tlen.push_back(a.rr); h0.push_back(a.seedlen0);
				job_read.push_back(r); job_reg.push_back(g); job_side.push_back(1);
				for (int k = 0; k < a.rq; ++k) q.push_back(query[a.seed_qbeg + a.seedlen0 + k]);
				for (int k = 0; k < a.rr; ++k) t.push_back(text(a.seed_rbeg + a.seedlen0 + k));
			}
		}
	}
	result_t *R = (result_t *)calloc(1, sizeof(result_t));
	R->n_regs = g; R->n_jobs = qlen.size(); R->q_bytes = q.size(); R->t_bytes = t.size(); R->err = err;
	rpr.resize(n_reads);
	R->regs_per_read = dup(rpr); R->qoff = dup(qoff); R->qlen = dup(qlen); R->toff = dup(toff); R->tlen = dup(tlen); R->h0 = dup(h0);
	R->job_read = dup(job_read); R->job_reg = dup(job_reg); R->job_side = dup(job_side); R->q = dup(q); R->t = dup(t);
	return R;
}

extern "C" result_t *chain_core_run(const bmh_chain_opt_t *opt, int64_t l_pac, const uint8_t *pac, uint32_t n_reads, const uint8_t *reads,
                                    const uint64_t *read_offs, const uint32_t *read_lens, const uint64_t *rbeg, const int32_t *qbeg,
                                    const uint32_t *score, const uint32_t *n_ref, const uint32_t *prefix, uint64_t n_seeds)
{
	return run(opt, l_pac, pac, n_reads, reads, read_offs, read_lens, rbeg, qbeg, score, n_ref, prefix, n_seeds, 0);
}
extern "C" result_t *chain_core_run_compact(const bmh_chain_opt_t *opt, int64_t l_pac, const uint8_t *pac, uint32_t n_reads, const uint8_t *reads,
                                            const uint64_t *read_offs, const uint32_t *read_lens, const uint64_t *rbeg, const int32_t *qbeg,
                                            const uint32_t *score, const uint32_t *n_ref, const uint32_t *prefix, uint64_t n_seeds)
{
	return run(opt, l_pac, pac, n_reads, reads, read_offs, read_lens, rbeg, qbeg, score, n_ref, prefix, n_seeds, 1);
}

extern "C" void chain_core_free(result_t *R)
{
	void *ps[] = {R->regs_per_read, R->qoff, R->qlen, R->toff, R->tlen, R->h0, R->job_read, R->job_reg, R->job_side, R->q, R->t};
	for (void *p : ps) free(p);
	free(R);
}
