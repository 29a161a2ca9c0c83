// Test driver for the GASAL2-compatible layer: uses ONLY the reference-facing API, the way
// fill_extension / mem_align1_core do (src/bwamem.c:1102-1167, 2106-2211), including the include
// path the reference uses (src/bntseq.h:35-40).  Reads jobs from a binary file, writes results.
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "../GASAL2/include/gasal.h"
#include "../GASAL2/include/args_parser.h"
#include "../GASAL2/include/host_batch.h"
#include "../GASAL2/include/gasal_align.h"
#include "../GASAL2/include/ctors.h"
#include "../GASAL2/include/interfaces.h"

int main(int argc, char **argv)
{
	if (argc < 3) return 2;
	FILE *f = fopen(argv[1], "rb");
	uint32_t n;
	if (!f || fread(&n, 4, 1, f) != 1) return 3;
	std::vector<uint32_t> qoff(n), qlen(n), toff(n), tlen(n), h0(n);
	fread(qoff.data(), 4, n, f); fread(qlen.data(), 4, n, f); fread(toff.data(), 4, n, f); fread(tlen.data(), 4, n, f); fread(h0.data(), 4, n, f);
	uint32_t nq, nt;
	fread(&nq, 4, 1, f); std::vector<uint8_t> q(nq + 1); fread(q.data(), 1, nq, f);
	fread(&nt, 4, 1, f); std::vector<uint8_t> t(nt + 1); fread(t.data(), 1, nt, f);
	fclose(f);

	gasal_subst_scores sub; sub.match = 1; sub.mismatch = 4; sub.gap_open = 6; sub.gap_extend = 1;   // fastmap.c:417-424
	gasal_copy_subst_scores(&sub);
	Parameters *args = new Parameters(0, NULL);
	args->algo = KSW; args->start_pos = WITHOUT_START;
	gasal_gpu_storage_v v = gasal_init_gpu_storage_v(2);                                            // NB_STREAMS, fastmap.c:477
	gasal_init_streams(&v, 1000 * 152, 1000 * 152, 1000 * 300, 1000 * 300, 64, 64, args);             // small on purpose: forces growth
	// two halves on the two streams, in flight together
	std::vector<int32_t> out(3 * (size_t)n);
	uint32_t half = n / 2, starts[2] = {0, half}, counts[2] = {half, n - half};
	uint32_t nqb[2], ntb[2];
	for (int s = 0; s < 2; ++s) {
		gasal_gpu_storage_t *st = &v.a[s];
		uint32_t iq = 0, it = 0, k = 0;
		st->current_n_alns = 0;
		for (uint32_t i = starts[s]; i < starts[s] + counts[s]; ++i, ++k) {
			st->current_n_alns++;
			if (st->current_n_alns > st->host_max_n_alns) gasal_host_alns_resize(st, st->host_max_n_alns * 2, args);  // bwamem.c:1107-1116
			st->host_target_batch_offsets[k] = it; st->host_query_batch_offsets[k] = iq;
			it = gasal_host_batch_fill(st, it, t.data() + toff[i], tlen[i], TARGET);
			iq = gasal_host_batch_fill(st, iq, q.data() + qoff[i], qlen[i], QUERY);
			st->host_query_batch_lens[k] = qlen[i]; st->host_target_batch_lens[k] = tlen[i];
			st->host_seed_scores[k] = h0[i];
		}
		nqb[s] = iq; ntb[s] = it;
		if (st->is_free != 1) return 4;
		gasal_aln_async(st, nqb[s], ntb[s], counts[s], args);                                       // bwamem.c:2127
	}
	for (int s = 0; s < 2; ++s) {
		gasal_gpu_storage_t *st = &v.a[s];
		if (counts[s] == 0) continue;
		while (gasal_is_aln_async_done(st) != 0) ;                                                  // bwamem.c:2181
		if (st->is_free != 1) return 5;
		for (uint32_t k = 0; k < counts[s]; ++k) {
			size_t i = starts[s] + k;
			out[3 * i] = st->host_res->aln_score[k]; out[3 * i + 1] = st->host_res->query_batch_end[k]; out[3 * i + 2] = st->host_res->target_batch_end[k];
		}
	}
	gasal_destroy_streams(&v, args);
	gasal_destroy_gpu_storage_v(&v);
	delete args;
	f = fopen(argv[2], "wb");
	fwrite(out.data(), 4, out.size(), f);
	fclose(f);
	return 0;
}
