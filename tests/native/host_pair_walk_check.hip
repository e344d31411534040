// Host-side check (no GPU): pair_group's pairing loops -- amplicon()'s walk over (forward, reverse) pairs and padlock()'s over
// (upstream, downstream) arms -- with the bounded walks of round 4 (stop / start where the list's loc5 order says no pair can
// follow) against the reference's walk over every pair (the same source compiled with -DTNT_EXP_FULL_WALK): the program prints a
// digest of the hit multiset of every random list; tests/test_host_pair_walk.py builds it both ways and compares the lines.
#include "../../thermonucleotideblast_amd/csrc/tnt_pair.hip.h"
#include <algorithm>
#include <climits>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <pthread.h>
#include <vector>
using namespace tnt;

struct Job {
	PairArgs p;
	Group G;
};
static void *run(void *v)
{
	Job *j = static_cast<Job *>(v);
	pair_group<PAIR_HOST>(j->p, j->G);
	return nullptr;
}

int main()
{
	srand(29);
	std::vector<Hit> hits(1 << 18);
	for (int it = 0; it < 240; it++) {
		const int kind = it % 3 == 2 ? 1 : 0;                    // PCR, PCR with a probe, padlock / MIPS
		const bool has_probe = it % 3 == 1;
		const int n = 20 + rand() % (it < 200 ? 400 : 6000);
		const int span = n * (3 + rand() % 60);
		const int pb = 5 + rand() % 90;                          // per cent bound
		const int max_len = kind == 1 ? (it % 2 ? 0 : 10 + rand() % 60) : 60 + rand() % 400;
		std::vector<uint32_t> pos(n);
		for (int i = 0; i < n; i++) pos[i] = uint32_t(rand() % span);
		std::sort(pos.begin(), pos.end());
		std::vector<Item> items(n);
		std::vector<int32_t> result(n, -1);
		std::vector<BoundRec> pool;
		for (int i = 0; i < n; i++) {
			const uint32_t role = has_probe ? uint32_t(rand() % 3) : uint32_t(rand() & 1), minus = rand() & 1, wi = rand() % 14;
			items[i] = Item{0u, pos[i], wi | (minus ? ITEM_MINUS : 0u) | (role << ITEM_SLOT_SHIFT), 0u};
			if (rand() % 100 < pb) {
				BoundRec r;
				memset(&r, 0, sizeof(r));
				r.loc5 = int(pos[i]) - int(wi) - rand() % 3, r.loc3 = r.loc5 + 15 + rand() % 12;
				if (rand() % 50 == 0) r.loc3 = r.loc5 - rand() % 3; // (odd footprints: the slack of the bounded walk)
				r.tm = 40.0f + float(rand() % 300) / 10.0f, r.dH = -150.f, r.dS = -0.4f, r.a5 = rand() % 8, r.a3 = rand() % 8, r.mm = rand() % 3, r.text_len = 10;
				strcpy(r.text, "5' ACGT 3'");
				result[i] = int32_t(pool.size());
				pool.push_back(r);
			}
		}
		if (kind == 1) { // the hash overload's lists arrive per list in diagonal order: the order key of pair_group restores it, any order will do here
		}
		AssayDev A;
		memset(&A, 0, sizeof(A));
		A.slot_f = 0, A.slot_r = 1, A.slot_p = has_probe ? 2 : -1, A.kind = kind, A.has_probe = has_probe ? 1 : 0;
		Slot slots[3];
		memset(slots, 0, sizeof(slots));
		uint32_t unit_target[1] = {0};
		int32_t own_lo[1] = {INT32_MIN}, own_hi[1] = {INT32_MAX};
		uint32_t n_hits = 0, err = 0, zero = 0;
		Job j;
		memset(&j.p, 0, sizeof(j.p));
		j.G = Group{0u, (rand() & 1) ? GROUP_MINUS_FIRST : 0u, 0u, uint32_t(n)};
		j.p.groups = &j.G, j.p.items = items.data(), j.p.result = result.data(), j.p.pool = pool.data(), j.p.assays = &A, j.p.slots = slots;
		j.p.max_len = max_len, j.p.single_primer_pcr = rand() & 1, j.p.min_max_primer_clamp = (rand() & 1) ? -1 : 3;
		j.p.assay_format = kind == 1 ? (max_len ? 3 : 2) : 0, j.p.target_strand = 3;
		j.p.hits = hits.data(), j.p.hit_cap = uint32_t(hits.size()), j.p.hit_count = &n_hits, j.p.err = &err, j.p.n_big = &zero;
		j.p.unit_target = unit_target, j.p.unit_own_lo = own_lo, j.p.unit_own_hi = own_hi;
		pthread_attr_t attr;
		pthread_attr_init(&attr);
		pthread_attr_setstacksize(&attr, size_t(512) << 20); // pair_group<PAIR_HOST> keeps its lists on the stack
		pthread_t t;
		pthread_create(&t, &attr, run, &j);
		pthread_join(t, nullptr);
		// digest of the hit sequence (order included: the bounded walks report the same pairs in the same order)
		unsigned long long h = 1469598103934665603ull;
		for (uint32_t i = 0; i < n_hits; i++) {
			const Hit &x = hits[i];
			const int v[8] = {x.amplicon_start, x.amplicon_stop, x.probe_start, x.probe_stop, x.primer_strand, x.forward_oligo, x.reverse_oligo, int(x.forward_tm * 10)};
			for (int k = 0; k < 8; k++) h = (h ^ (unsigned long long)(unsigned)v[k]) * 1099511628211ull;
		}
		printf("case %d kind %d probe %d n %d bound %zu max_len %d: %u hits, err %u, digest %016llx\n", it, kind, int(has_probe), n, pool.size(), max_len, n_hits, err, h);
	}
	return 0;
}
