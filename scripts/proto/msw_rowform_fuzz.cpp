// prototype of the row-parallel form of the striped byte-mode pass, fuzzed against the exact emulation (local_sw.cpp)
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <algorithm>
#include <random>
#include "local_sw.h"
struct res5 { int score, te, qe, score2, te2; };
static inline int sat0(int v) { return v < 0 ? 0 : v; }
static res5 pass_new(int qlen, const uint8_t *q, int tlen, const uint8_t *t, const bmh_ext_params_t &p, int xtra)
{
	const int lanes = 16, slen = (qlen + lanes - 1) / lanes;
	const int minsc = (xtra & BMH_SW_XSUBO) ? xtra & 0xffff : 0x10000, endsc = (xtra & BMH_SW_XSTOP) ? xtra & 0xffff : 0x10000;
	const int oe_del = p.o_del + p.e_del, oe_ins = p.o_ins + p.e_ins;
	int mn = std::min(std::min(p.a, -p.b), -1), mx = std::max(std::max(p.a, -p.b), -1);
	const int shift = (256 - (mn & 0xff)) & 0xff;
	std::vector<int> H(16 * slen, 0), E(16 * slen, 0), Hm(16 * slen, 0), qc(16 * slen);
	for (int l = 0; l < 16; ++l) for (int j = 0; j < slen; ++j) { int k = l * slen + j; qc[l * slen + j] = k < qlen ? q[k] : 5; }
	int gmax = 0, te = -1;
	std::vector<std::pair<int,int>> b;
	for (int i = 0; i < tlen; ++i) {
		const int tt = t[i];
		int hin[16], fout[16], fin[16], mxv[16];
		for (int l = 0; l < 16; ++l) hin[l] = l ? H[(l - 1) * slen + slen - 1] : 0;
		for (int l = 0; l < 16; ++l) {
			int hv = hin[l], f = 0; mxv[l] = 0;
			for (int j = 0; j < slen; ++j) {
				const int c = qc[l * slen + j];
				int S = c > 3 ? (c & 1) - 1 : (tt > 3 ? -1 : (c == tt ? p.a : -p.b));
				int h = std::max(std::max(hv + S, E[l * slen + j]), f);      // >= 0 by itself
				mxv[l] = std::max(mxv[l], h);
				hv = H[l * slen + j];
				H[l * slen + j] = h;
				E[l * slen + j] = std::max(sat0(E[l * slen + j] - p.e_del), sat0(h - oe_del));
				f = std::max(sat0(f - p.e_ins), sat0(h - oe_ins));
			}
			fout[l] = f;
		}
		// inflow by a max-plus scan
		int run = -(1 << 28);
		for (int l = 0; l < 16; ++l) {
			fin[l] = l ? sat0(run - p.e_ins * slen * (l - 1)) : 0;
			run = std::max(run, fout[l] + p.e_ins * slen * l);
		}
		for (int l = 0; l < 16; ++l) if (fin[l] > 0) for (int j = 0; j < slen; ++j) H[l * slen + j] = std::max(H[l * slen + j], fin[l] - p.e_ins * j);
		int imax = 0;
		for (int l = 0; l < 16; ++l) imax = std::max(imax, mxv[l]);
		if (imax >= minsc) {
			if (b.empty() || b.back().second + 1 != i) b.push_back({imax, i});
			else if (b.back().first < imax) b.back() = {imax, i};
		}
		if (imax > gmax) {
			gmax = imax; te = i; Hm = H;
			if (gmax + shift >= 255 || gmax >= endsc) break;
		}
	}
	res5 r; r.score = gmax + shift < 255 ? gmax : 255; r.te = te; r.qe = -1; r.score2 = -1; r.te2 = -1;
	if (r.score != 255) {
		int best = -1;
		for (int l = 0; l < 16; ++l) for (int j = 0; j < slen; ++j) { int v = Hm[l * slen + j], pos = l * slen + j; if (v > best || (v == best && pos < r.qe)) { best = v; r.qe = pos; } }
		if (!b.empty()) {
			const int d = (r.score + mx - 1) / mx, low = te - d, high = te + d;
			for (auto &x : b) if ((x.second < low || x.second > high) && x.first > r.score2) { r.score2 = x.first; r.te2 = x.second; }
		}
	}
	return r;
}
extern "C" void bmh_local_sw_c(int qlen, uint8_t *query, int tlen, uint8_t *target, const bmh_ext_params_t *p, int xtra, int32_t out[7]);
int main(int argc, char **argv)
{
	const long N = argc > 1 ? atol(argv[1]) : 100000;
	std::mt19937_64 rng(argc > 2 ? atol(argv[2]) : 1);
	long bad = 0, n255 = 0, ns2 = 0;
	const int SC[6][6] = {{1,4,6,1,6,1},{2,3,5,2,5,2},{1,3,4,2,4,2},{3,2,6,1,6,1},{1,2,3,1,3,1},{1,4,6,1,2,2}};
	for (long it = 0; it < N; ++it) {
		const int *s = SC[it % 6];
		bmh_ext_params_t p; memset(&p, 0, sizeof(p)); p.a = s[0]; p.b = s[1]; p.o_del = s[2]; p.e_del = s[3]; p.o_ins = s[4]; p.e_ins = s[5];
		int qlen = 20 + rng() % 231; if (rng() % 4 == 0) qlen = 150;
		int tlen = qlen + rng() % 400;
		std::vector<uint8_t> t(tlen), q(qlen);
		const int kind = rng() % 4;
		int per = 1 + rng() % 12;
		for (int i = 0; i < tlen; ++i) t[i] = kind == 1 ? (i < per ? rng() & 3 : t[i - per]) : rng() & 3;
		if (kind == 1) for (int k = 0; k < tlen / 8; ++k) t[rng() % tlen] = rng() & 3;
		// the query: a copy of a stretch of the target with substitutions and indels (or random)
		if (kind == 3) for (int k = 0; k < qlen; ++k) q[k] = rng() & 3;
		else {
			int pos = rng() % std::max(1, tlen - qlen / 2), k = 0;
			const int erate = 1 + rng() % 12, irate = rng() % 3 ? 40 + rng() % 100 : 8;
			while (k < qlen) {
				if (pos >= tlen) { q[k++] = rng() & 3; continue; }
				const int r = rng() % 1000;
				if (r < 1000 / irate / 2) { pos += 1 + rng() % 6; continue; }                        // deletion from the query
				if (r < 1000 / irate) { int m = 1 + rng() % 6; while (m-- && k < qlen) q[k++] = rng() & 3; continue; }
				q[k++] = (rng() % 100 < (unsigned)erate) ? rng() & 3 : t[pos]; ++pos;
			}
		}
		if (rng() % 5 == 0) for (int k = 0; k < 3; ++k) q[rng() % qlen] = 4;
		if (rng() % 5 == 0) for (int k = 0; k < 3; ++k) t[rng() % tlen] = 4;
		const int minsc = (rng() % 3 ? 19 : 1 + rng() % 60) * p.a;
		int xtra = BMH_SW_XBYTE | BMH_SW_XSUBO | minsc;
		if (rng() % 8 == 0) xtra = BMH_SW_XBYTE | BMH_SW_XSTOP | (int)(10 + rng() % 100);
		int32_t want[7];
		std::vector<uint8_t> q2(q), t2(t);
		bmh_local_sw_c(qlen, q2.data(), tlen, t2.data(), &p, xtra, want);
		const res5 g = pass_new(qlen, q.data(), tlen, t.data(), p, xtra);
		n255 += want[0] == 255; ns2 += want[3] >= 0;
		if (g.score != want[0] || g.te != want[1] || (want[0] != 255 && (g.qe != want[2] || g.score2 != want[3] || g.te2 != want[4]))) {
			if (bad < 10) printf("MISMATCH it %ld sc %d qlen %d tlen %d kind %d: got %d %d %d %d %d want %d %d %d %d %d\n", it, (int)(it % 6), qlen, tlen, kind, g.score, g.te, g.qe, g.score2, g.te2, want[0], want[1], want[2], want[3], want[4]);
			++bad;
		}
	}
	printf("%ld cases, %ld mismatches (%ld overflowed, %ld with a second-best)\n", N, bad, n255, ns2);
	return bad != 0;
}
