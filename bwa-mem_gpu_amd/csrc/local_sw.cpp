#include <algorithm>
#include <cstring>
#include <vector>
#include "local_sw.h"

namespace {

inline int sc(const bmh_ext_params_t &p, int t, int q) { return (t > 3 || q > 3) ? -1 : (t == q ? p.a : -p.b); }
inline int sat0(int v) { return v < 0 ? 0 : v; }

// one pass of the striped kernel (ksw_u8 when lanes == 16, ksw_i16 when lanes == 8) on query[0..qlen), target[0..tlen)
bmh_sw_result_t sw_pass(int lanes, int qlen, const uint8_t *query, int tlen, const uint8_t *target, const bmh_ext_params_t &p, int xtra)
{
	bmh_sw_result_t r = {0, -1, -1, -1, -1, -1, -1};
	const bool byte = lanes == 16;
	const int slen = (qlen + lanes - 1) / lanes, n = slen * lanes;
	const int minsc = (xtra & BMH_SW_XSUBO) ? xtra & 0xffff : 0x10000, endsc = (xtra & BMH_SW_XSTOP) ? xtra & 0xffff : 0x10000;
	const int oe_del = p.o_del + p.e_del, oe_ins = p.o_ins + p.e_ins;
	int mn = 127, mx = 0;                                   // shift and largest score of the 5 x 5 matrix (a, -b, -1)
	for (int t = 0; t < 5; ++t) for (int q = 0; q < 5; ++q) { const int v = sc(p, t, q); mn = std::min(mn, v); mx = std::max(mx, v); }
	const int shift = byte ? (256 - (mn & 0xff)) & 0xff : 0;
	const int cap = byte ? 255 : 32767;
	// memory order of the reference's vectors: element [j][l] is query position j + l * slen
	std::vector<int> H0(n, 0), H1(n, 0), E(n, 0), Hmax(n, 0), prof(5 * n);
	for (int t = 0; t < 5; ++t)
		for (int j = 0; j < slen; ++j)
			for (int l = 0; l < lanes; ++l) { const int k = j + l * slen; prof[(t * slen + j) * lanes + l] = k >= qlen ? 0 : sc(p, t, query[k]); }
	std::vector<int> hv(lanes), f(lanes), mxv(lanes);
	std::vector<uint64_t> b;
	int gmax = 0, te = -1;
	int *h0 = H0.data(), *h1 = H1.data();
	for (int i = 0; i < tlen; ++i) {
		const int *S = prof.data() + (size_t)target[i] * slen * lanes;
		for (int l = 0; l < lanes; ++l) { hv[l] = l ? h0[(slen - 1) * lanes + l - 1] : 0; f[l] = 0; mxv[l] = 0; }
		for (int j = 0; j < slen; ++j) {
			for (int l = 0; l < lanes; ++l) {
				int h = byte ? sat0(std::min(hv[l] + S[j * lanes + l] + shift, 255) - shift) : std::max(std::min(hv[l] + S[j * lanes + l], cap), -32768);
				int e = E[j * lanes + l];
				h = std::max(h, e); h = std::max(h, f[l]);
				mxv[l] = std::max(mxv[l], h);
				h1[j * lanes + l] = h;
				const int hu = byte ? h : (h & 0xffff);       // subs_epu16 treats the 16 bits as unsigned
				e = std::max(sat0(e - p.e_del), byte ? sat0(h - oe_del) : sat0(hu - oe_del));
				E[j * lanes + l] = e;
				f[l] = std::max(sat0(f[l] - p.e_ins), byte ? sat0(h - oe_ins) : sat0(hu - oe_ins));
				hv[l] = h0[j * lanes + l];
			}
		}
		for (int k = 0; k < 16; ++k) {                      // lazy F (ksw.c:497-511, 627-638): both widths loop 16 times
			for (int l = lanes - 1; l > 0; --l) f[l] = f[l - 1];
			f[0] = 0;
			bool done = false;
			for (int j = 0; j < slen; ++j) {
				bool any = false;
				for (int l = 0; l < lanes; ++l) {
					int h = std::max(h1[j * lanes + l], f[l]);
					h1[j * lanes + l] = h;
					h = sat0((byte ? h : (h & 0xffff)) - oe_ins);
					f[l] = sat0(f[l] - p.e_ins);
					if (f[l] > h) any = true;
				}
				if (!any) { done = true; break; }
			}
			if (done) break;
		}
		int imax = 0;
		for (int l = 0; l < lanes; ++l) imax = std::max(imax, mxv[l]);
		if (imax >= minsc) {
			if (b.empty() || (int32_t)b.back() + 1 != i) b.push_back((uint64_t)imax << 32 | (uint32_t)i);
			else if ((int)(b.back() >> 32) < imax) b.back() = (uint64_t)imax << 32 | (uint32_t)i;
		}
		if (imax > gmax) {
			gmax = imax; te = i;
			memcpy(Hmax.data(), h1, sizeof(int) * n);
			if ((byte && gmax + shift >= 255) || gmax >= endsc) break;
		}
		std::swap(h0, h1);
	}
	r.score = byte ? (gmax + shift < 255 ? gmax : 255) : gmax;
	r.te = te;
	if (!byte || r.score != 255) {
		int best = -1;
		if (!byte) r.qe = -1;
		for (int i = 0; i < n; ++i) {                       // memory index i -> query position i / lanes + i % lanes * slen
			const int v = Hmax[i], pos = i / lanes + i % lanes * slen;
			if (v > best) { best = v; r.qe = pos; }
			else if (v == best && pos < r.qe) r.qe = pos;
		}
		if (!b.empty()) {
			const int d = (r.score + mx - 1) / mx, low = te - d, high = te + d;
			for (uint64_t x : b) {
				const int e = (int32_t)x;
				if ((e < low || e > high) && (int)(x >> 32) > r.score2) { r.score2 = (int)(x >> 32); r.te2 = e; }
			}
		}
	}
	return r;
}

} // namespace

bmh_sw_result_t bmh_local_sw(int qlen, uint8_t *query, int tlen, uint8_t *target, const bmh_ext_params_t &p, int xtra)
{
	const int lanes = (xtra & BMH_SW_XBYTE) ? 16 : 8;
	bmh_sw_result_t r = sw_pass(lanes, qlen, query, tlen, target, p, xtra);
	if ((xtra & BMH_SW_XSTART) == 0 || ((xtra & BMH_SW_XSUBO) && r.score < (xtra & 0xffff))) return r;
	// start positions: the same pass over the reversed prefixes, stopped at the score (ksw.c:722-736; the target is
	// passed with its full length, as the reference does)
	std::reverse(query, query + r.qe + 1);
	std::reverse(target, target + r.te + 1);
	const bmh_sw_result_t rr = sw_pass(lanes, r.qe + 1, query, tlen, target, p, BMH_SW_XSTOP | r.score);
	std::reverse(query, query + r.qe + 1);
	std::reverse(target, target + r.te + 1);
	if (r.score == rr.score) { r.tb = r.te - rr.te; r.qb = r.qe - rr.qe; }
	return r;
}

// C entry for tests
extern "C" void bmh_local_sw_c(int qlen, uint8_t *query, int tlen, uint8_t *target, const bmh_ext_params_t *p, int xtra, int32_t out[7])
{
	const bmh_sw_result_t r = bmh_local_sw(qlen, query, tlen, target, *p, xtra);
	out[0] = r.score; out[1] = r.te; out[2] = r.qe; out[3] = r.score2; out[4] = r.te2; out[5] = r.tb; out[6] = r.qb;
}
