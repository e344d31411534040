// Host-side check (no GPU): the replay of std::list::sort that the long-target path runs on host threads -- blocks finished one
// after the other, merges that would hand back their input skipped (tnt_pair.hip.h, list_sort / sort_block) -- against the plain
// breadth-first form of the same merge tree, on random lists of bound and unbound entries under sort_by_oligo_loc's comparison
// (not a strict weak order: the outcome depends on the merge tree).  Built and run by tests/test_host_list_sort.py.
#include "../../thermonucleotideblast_amd/csrc/tnt_pair.hip.h"
#include <cstdio>
#include <cstdlib>
#include <vector>
using namespace tnt;
template <class Less> void ref_sort(Entry *e, Entry *tmp, int n, Less less)
{
	if (n < 2) return;
	int start = 0;
	for (int bit = 30; bit >= 0; bit--) {
		const int size = 1 << bit;
		if (!(n & size)) continue;
		for (int width = 1; width < size; width <<= 1)
			for (int lo = start; lo < start + size; lo += 2 * width) merge_runs(e, tmp, lo, lo + width, lo + 2 * width, less);
		start += size;
	}
	int lo_bit = 0;
	while (!(n & (1 << lo_bit))) lo_bit++;
	int right = n - (1 << lo_bit);
	for (int bit = lo_bit + 1; bit <= 30; bit++) {
		const int size = 1 << bit;
		if (!(n & size)) continue;
		const int left = right - size;
		merge_runs(e, tmp, left, right, n, less);
		right = left;
	}
}
int main()
{
	srand(7);
	long bad = 0, cases = 0;
	for (int it = 0; it < 3000; it++) {
		const int n = it < 2800 ? rand() % 300 : 1000 + rand() % 40000;
		std::vector<Entry> a(n), b, t(n);
		const int span = 50 + rand() % 2000;
		const int pb = rand() % 100; // per cent bound
		const bool presorted = rand() % 2;
		for (int i = 0; i < n; i++) {
			a[i].tloc = uint32_t(rand() % span);
			a[i].rec = i;
			a[i].mask = 0;
			if (rand() % 100 < pb) a[i].loc5 = int(a[i].tloc) - rand() % 30 + 10, a[i].loc3 = a[i].loc5 + rand() % 40;
			else a[i].loc5 = a[i].loc3 = 0;
		}
		if (presorted) std::stable_sort(a.begin(), a.end(), [](const Entry &x, const Entry &y) { return x.tloc < y.tloc; });
		b = a;
		auto less = [](const Entry &x, const Entry &y) { return oligo_loc_less(x, y); };
		ref_sort(a.data(), t.data(), n, less);
		list_sort(b.data(), t.data(), n, less, (it & 1) != 0); // (odd cases: with the skip of blocks of unbound entries in target_loc order, lists of 4 096 entries and more)
		cases++;
		for (int i = 0; i < n; i++)
			if (a[i].rec != b[i].rec) {
				bad++;
				break;
			}
	}
	printf("%ld cases, %ld differ\n", cases, bad);
	return bad != 0;
}
