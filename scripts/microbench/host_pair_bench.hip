// Host-only timing of pair_group<PAIR_HOST> on a synthetic PCR list the size of a 250-Mb record's (no GPU):
//   hipcc -O3 -std=c++17 -ffp-contract=off --offload-arch=gfx950 -pthread scripts/microbench/host_pair_bench.hip -o /tmp/host_pair_bench && /tmp/host_pair_bench <entries> <span> <bound per mille> <repetitions> <threads>
#include "../../thermonucleotideblast_amd/csrc/tnt_pair.hip.h"
#include <chrono>
#include <cstring>
#include <climits>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <pthread.h>
using namespace tnt;
int main(int argc, char **argv)
{
	const int n = argc > 1 ? atoi(argv[1]) : 165000;
	const double span = argc > 2 ? atof(argv[2]) : 250e6;
	const int pct_bound_x10 = argc > 3 ? atoi(argv[3]) : 10; // per mille
	const int reps = argc > 4 ? atoi(argv[4]) : 3;
	srand(11);
	std::vector<uint32_t> pos(n);
	for (int i = 0; i < n; i++) pos[i] = uint32_t((double(rand()) / RAND_MAX) * span);
	std::sort(pos.begin(), pos.end());
	std::vector<Item> items(n);
	std::vector<int32_t> result(n, -1);
	std::vector<BoundRec> pool;
	for (int i = 0; i < n; i++) {
		const uint32_t role = rand() & 1, minus = rand() & 1, wi = rand() % 14;
		items[i] = Item{0u, pos[i], wi | (minus ? ITEM_MINUS : 0u) | (role << ITEM_SLOT_SHIFT), 0u};
		if (rand() % 1000 < pct_bound_x10) {
			BoundRec r;
			memset(&r, 0, sizeof(r));
			r.loc5 = int(pos[i]) - int(wi), r.loc3 = r.loc5 + 19 + rand() % 5;
			r.tm = 40.0f + float(rand() % 300) / 10.0f, r.dH = -150.f, r.dS = -0.4f, r.a5 = 5, r.a3 = 6, r.mm = rand() % 3, r.text_len = 70;
			strcpy(r.text, "5' ACGT 3'");
			result[i] = int32_t(pool.size());
			pool.push_back(r);
		}
	}
	AssayDev A;
	memset(&A, 0, sizeof(A));
	A.slot_f = 0, A.slot_r = 1, A.slot_p = -1, A.kind = 0;
	Slot slots[3];
	memset(slots, 0, sizeof(slots));
	Group G{0u, 0u | GROUP_MINUS_FIRST, 0u, uint32_t(n)};
	uint32_t unit_target[1] = {0};
	int32_t own_lo[1] = {INT32_MIN}, own_hi[1] = {INT32_MAX};
	std::vector<Hit> hits(1 << 20);
	for (int rep = 0; rep < reps; rep++) {
		uint32_t n_hits = 0, err = 0, zero = 0;
		PairArgs p;
		memset(&p, 0, sizeof(p));
		p.groups = &G, p.items = items.data(), p.result = result.data(), p.pool = pool.data(), p.assays = &A, p.slots = slots;
		p.max_len = 2000, p.single_primer_pcr = 1, p.min_max_primer_clamp = -1, p.assay_format = 0, p.target_strand = 3;
		p.hits = hits.data(), p.hit_cap = uint32_t(hits.size()), p.hit_count = &n_hits, p.err = &err, p.n_big = &zero;
		p.unit_target = unit_target, p.unit_own_lo = own_lo, p.unit_own_hi = own_hi;
		static double ms;
		static const PairArgs *pp;
		static const Group *gp;
		pp = &p, gp = &G;
		const int T = argc > 5 ? atoi(argv[5]) : 1;
		pthread_attr_t attr;
		pthread_attr_init(&attr);
		pthread_attr_setstacksize(&attr, size_t(512) << 20);
		std::vector<pthread_t> ts(T);
		const auto t0 = std::chrono::steady_clock::now();
		for (int k = 0; k < T; k++)
			pthread_create(&ts[k], &attr, [](void *) -> void * {
				pair_group<PAIR_HOST>(*pp, *gp);
				return nullptr;
			}, nullptr);
		for (int k = 0; k < T; k++) pthread_join(ts[k], nullptr);
		ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
		printf("n %d span %.0f bound %zu, %d thread(s) (fresh stacks): %.2f ms, %u hits, err %u\n", n, span, pool.size(), argc > 5 ? atoi(argv[5]) : 1, ms, n_hits, err);
	}
	return 0;
}
