// klib's introsort (src/ksort.h:146-226 of the reference; klib is (c) 2008-2011 Attractive Chaos <attractor@live.co.uk>, MIT
// license -- the algorithm and its exact exchange sequence are klib's), restated as a template: it is not a stable sort, so
// wherever the reference sorts records that can compare equal the order of the ties depends on its exact steps.
#pragma once
#include <cstddef>
#include <utility>
#include <vector>

namespace klib {

// median-of-three partition of the whole range, sub-ranges of <= 16 left to a final insertion sort, comb sort when
// the depth budget runs out
template <class T, class LT> void insertion_sort(T *s, T *t, LT lt)
{
	for (T *i = s + 1; i < t; ++i)
		for (T *j = i; j > s && lt(*j, *(j - 1)); --j) std::swap(*j, *(j - 1));
}
template <class T, class LT> void comb_sort(size_t n, T *a, LT lt)
{
	const double shrink = 1.2473309501039786540366528676643;
	bool swapped; size_t gap = n;
	do {
		if (gap > 2) { gap = (size_t)(gap / shrink); if (gap == 9 || gap == 10) gap = 11; }
		swapped = false;
		for (T *i = a; i < a + n - gap; ++i) { T *j = i + gap; if (lt(*j, *i)) { std::swap(*i, *j); swapped = true; } }
	} while (swapped || gap > 2);
	if (gap != 1) insertion_sort(a, a + n, lt);
}
template <class T, class LT> void klib_introsort(size_t n, T *a, LT lt)
{
	if (n < 1) return;
	if (n == 2) { if (lt(a[1], a[0])) std::swap(a[0], a[1]); return; }
	int d;
	for (d = 2; 1ul << d < n; ++d) ;
	struct Fr { T *l, *r; int depth; };
	std::vector<Fr> stack;
	T *s = a, *t = a + (n - 1);
	d <<= 1;
	for (;;) {
		if (s < t) {
			if (--d == 0) { comb_sort((size_t)(t - s + 1), s, lt); t = s; continue; }
			T *i = s, *j = t, *k = i + ((j - i) >> 1) + 1;
			if (lt(*k, *i)) { if (lt(*k, *j)) k = j; }
			else k = lt(*j, *i) ? i : j;
			T rp = *k;
			if (k != t) std::swap(*k, *t);
			for (;;) {
				do ++i; while (lt(*i, rp));
				do --j; while (i <= j && lt(rp, *j));
				if (j <= i) break;
				std::swap(*i, *j);
			}
			std::swap(*i, *t);
			if (i - s > t - i) {
				if (i - s > 16) stack.push_back({s, i - 1, d});
				s = t - i > 16 ? i + 1 : t;
			} else {
				if (t - i > 16) stack.push_back({i + 1, t, d});
				t = i - s > 16 ? i - 1 : s;
			}
		} else {
			if (stack.empty()) { insertion_sort(a, a + n, lt); return; }
			Fr f = stack.back(); stack.pop_back();
			s = f.l; t = f.r; d = f.depth;
		}
	}
}


} // namespace klib
