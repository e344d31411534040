// tnt_long.hip.h -- the long-target path of the library: targets that do not fit one seed work unit (genomes, chromosomes;
// dense seeding with short hash words).  Part of tnt_host.hip's translation unit (included there: it uses the context, the
// planner, the seed launch and the bind phases defined above it); the device side of the list assembly and of the first cull
// is tnt_sort.hip.  Product code: nothing here includes or links anything under oracle/.
#ifndef TNT_LONG_HIP_H
#define TNT_LONG_HIP_H

// ---------------------------------------------------------------------------------------------------------------------
// Targets that do not fit one seed work unit (genomes; dense seeding with short hash words).  The reference's culls, its
// choice of the strand that is bound first and its pairing loops act on the lists of the WHOLE target
// (amplicon_search.cpp:96-674, padlock_search.cpp, probe_search.cpp), and the first cull's outcome is not local
// (cull_oligo_match counts the strands of list successors, sort_by_oligo_loc is not a strict weak order: SURVEY Q4, Q5).
// So such a target is SEEDED in windows of k-mer positions (every position once), its seeds are put into per-(target, assay)
// lists and culled once on the device (tnt_sort.hip), every seed the first cull keeps is BOUND through the same bucket ->
// pre-filter -> bind kernels as everything else, and since round 6 the bind / cull sequence and the pairing loops of the PCR
// lists and the sorts and pairing of the probe / padlock lists run on the device as well, for all lists of a pass together
// (tnt_sort.hip, long_pair_device / long_hash_device).  What is left to the host threads below -- which run the code the pair
// kernel runs, tnt_pair.hip.h being compiled for both sides --: the lists of the assays under the cache replay, the segments of
// probes beyond 128 bases, the lists of contexts with primers beyond 128 bases (assembled from segments on the host), and a PCR
// list the device hands back because
// sort_by_oligo_loc is not a strict weak order on its entries (the outcome of std::list::sort then hangs on its merge tree,
// which list_sort restates).
// The host workers of this path: a process-wide pool of threads with 256-MB stacks (pair_group<PAIR_HOST> keeps its lists
// on the stack), started at the first long-target search and parked on a condition variable between runs -- starting 64
// or 128 threads with such stacks costs 2-3 ms every time, as much as the culls and the pairing they are started for.
// run(n, f): f(0) .. f(n - 1), the workers take the indices one at a time; one run at a time (searches of several
// contexts take turns).
class HostPool
{
	std::mutex run_mu, mu, err_mu;
	std::string error; // what a worker's job threw during the current run (the first one)
	std::condition_variable cv_work, cv_done;
	std::vector<pthread_t> th;
	const std::function<void(size_t)> *fn = nullptr;
	size_t n = 0;
	std::atomic<size_t> next{0};
	unsigned busy = 0;
	uint64_t generation = 0;
	bool stop = false;
	pid_t owner = 0;

	static void *worker_main(void *p)
	{
		static_cast<HostPool *>(p)->worker();
		return nullptr;
	}
	void worker()
	{
		uint64_t seen = 0;
		std::unique_lock<std::mutex> lk(mu);
		for (;;) {
			cv_work.wait(lk, [&] { return stop || generation != seen; });
			if (stop) return;
			seen = generation;
			lk.unlock();
			for (size_t i = next.fetch_add(1); i < n; i = next.fetch_add(1)) {
				try { // (an exception that leaves a thread's start function ends the process: pair_group allocates)
					(*fn)(i);
				} catch (const std::exception &e) {
					std::lock_guard<std::mutex> g(err_mu);
					if (error.empty()) error = e.what();
				} catch (...) {
					std::lock_guard<std::mutex> g(err_mu);
					if (error.empty()) error = "unknown exception";
				}
			}
			lk.lock();
			if (--busy == 0) cv_done.notify_all();
		}
	}

public:
	static HostPool &get()
	{
		static HostPool pool;
		return pool;
	}
	~HostPool()
	{
		{
			std::lock_guard<std::mutex> lk(mu);
			stop = true;
		}
		cv_work.notify_all();
		if (owner == getpid())
			for (pthread_t t : th) pthread_join(t, nullptr);
	}
	unsigned size()
	{
		std::lock_guard<std::mutex> g(run_mu);
		start();
		return unsigned(th.size());
	}
	// false: no worker thread could be started (f needs the large stack, it must not run on the caller's), or a job threw
	// (*why says what)
	bool run(size_t count, const std::function<void(size_t)> &f, std::string *why = nullptr)
	{
		std::lock_guard<std::mutex> g(run_mu);
		start();
		if (th.empty()) {
			if (why) *why = "no worker thread with a list-sized stack could be started";
			return false;
		}
		if (count == 0) return true;
		error.clear();
		std::unique_lock<std::mutex> lk(mu);
		fn = &f;
		n = count;
		next = 0;
		busy = unsigned(th.size());
		generation++;
		cv_work.notify_all();
		cv_done.wait(lk, [&] { return busy == 0; });
		fn = nullptr;
		if (!error.empty()) {
			if (why) *why = "a pairing job failed: " + error;
			return false;
		}
		return true;
	}

private:
	void start() // (under run_mu)
	{
		if (!th.empty() && owner != getpid()) th.clear(); // a forked child has none of its parent's threads
		if (!th.empty()) return;
		owner = getpid();
		// the lists of one target are few and long (96 lists of 19 000 seeds for a 5-Mb target against config 2): as many
		// workers as the host has cores, up to 128
		const unsigned want = std::max(1u, std::min(128u, std::thread::hardware_concurrency()));
		pthread_attr_t attr;
		pthread_attr_init(&attr);
		// address space, not memory: pages are touched as far as a list reaches.  pair_group<PAIR_HOST> keeps about 190 MB of
		// lists on its stack at the largest list (a million entries: the list, its order keys, four hash-overload lists and the
		// scratch copies of cull / bind_step / list_sort): 256 MB per worker, 32 GB of address space for 128 of them
		pthread_attr_setstacksize(&attr, size_t(256) << 20);
		for (unsigned w = 0; w < want; w++) {
			pthread_t t;
			if (pthread_create(&t, &attr, worker_main, this) == 0) th.push_back(t);
		}
		pthread_attr_destroy(&attr);
	}
};

// One pass over the windows [u0, u1) of some of the long targets (whole targets: their lists are culled and paired whole).
// cull_rows != nullptr: stop behind the first cull and append its outcome as rows of tnt_first_cull.
// LONG_RESTART: the windows were rebuilt at half their length (a window's seed list overflowed): every pass starts again.
constexpr int LONG_RESTART = 1000;
static int search_long_pass(tnt_ctx *ctx, tnt_db *db, const size_t u0, const size_t u1, std::vector<tnt_hit> &out, std::vector<int32_t> *cull_rows)
{
	if (u0 >= u1) return TNT_OK;
	const bool timing = getenv("TNT_MI355X_LONG_TIMING") != nullptr; // phase times of this path on stderr
	auto t_last = std::chrono::steady_clock::now();
	auto lap = [&](const char *what) {
		if (!timing) return;
		const auto now = std::chrono::steady_clock::now();
		fprintf(stderr, "long targets: %-28s %8.2f ms\n", what, std::chrono::duration<double, std::milli>(now - t_last).count());
		t_last = now;
	};
	HIP_TRY(hipSetDevice(ctx->device), ctx->err);
	if (hipError_t stale = hipGetLastError()) { // nothing before this point may have failed
		ctx->err = std::string("before the long-target path: ") + hipGetErrorString(stale);
		return TNT_E_DEVICE;
	}
	uint32_t *cnt = ctx->S().counters.p;
	// ---- seeds of every window (the seed kernel's unit-level output: kept seeds before the cull); they stay on the device
	size_t n_raw = 0;
	for (int attempt = 0;; attempt++) {
		size_t tpb = 1;
		int rc = plan(ctx, db, tpb);
		if (rc) return rc;
		double positions = 0.0;
		for (size_t u = u0; u < u1; u++) positions += double(db->h_unit_end[u] - db->h_unit_begin[u]);
		const double words = double(size_t(1) << (2 * ctx->opt.hash_word_size));
		const size_t cap = size_t(positions * double(ctx->n_entries) / words * 1.5) + (size_t(1) << 20);
		if (cap > (size_t(1) << 31)) {
			ctx->err = "too many seeds on the long targets of this batch";
			return TNT_E_LIMIT;
		}
		if (cap > ctx->seeds_out.n) HIP_TRY(ctx->seeds_out.alloc(cap), ctx->err); // (kept for the next search)
		HIP_TRY(hipMemsetAsync(cnt, 0, sizeof(uint32_t) * C_LTEXT, ctx->stream), ctx->err); // (not the long-text arena's fill: the texts of the search so far stay)
		rc = enqueue_batch(ctx, db, u0, u1, true);
		if (rc) return rc;
		uint32_t h[C_COUNT];
		HIP_TRY(hipMemcpyAsync(h, cnt, sizeof(h), hipMemcpyDeviceToHost, ctx->stream), ctx->err);
		HIP_TRY(hipStreamSynchronize(ctx->stream), ctx->err);
		if (hipError_t bad = hipGetLastError()) {
			ctx->err = std::string("seeding the windows of long targets: ") + hipGetErrorString(bad);
			return TNT_E_DEVICE;
		}
		if ((h[C_ERR] & ERR_SEED_OVERFLOW) && attempt < 16) { // a window's list overflowed: fewer assays per pass, then shorter windows
			if (ctx->chunk_begin.size() < ctx->h_assays.size() + 1) {
				ctx->chunk_scale *= 0.5;
			} else {
				if (ctx->unit_scale < 1.0 / 64.0) { // the windows are at their shortest already
					ctx->err = err_text(ERR_SEED_OVERFLOW);
					return TNT_E_LIMIT;
				}
				// shorter windows for the targets that are seeded in windows; the whole targets (units below u0, already
				// searched and collected) stay what they are: build_units does not look at unit_scale for the split
				ctx->unit_scale *= 0.5;
				HIP_TRY(build_units(ctx, db), ctx->err);
				return LONG_RESTART; // the unit ranges changed
			}
			continue;
		}
		if (h[C_ERR]) {
			ctx->err = err_text(h[C_ERR]);
			return TNT_E_LIMIT;
		}
		if (h[C_SEEDS] > ctx->seeds_out.n) {
			ctx->err = "seed output overflow on a long target";
			return TNT_E_LIMIT;
		}
		n_raw = h[C_SEEDS];
		break;
	}
	if (n_raw == 0) return TNT_OK;
	lap("seed windows");

	// ---- per (target, assay) lists in the reference's list order, assembled on the device (tnt_sort.hip: key kernel, two
	// stable radix sorts, gather); the host gets the seeds as work items, list after list, and the list of every item
	const size_t n_assays_all = ctx->h_assays.size();
	std::vector<uint32_t> long_targets; // the targets these windows belong to, ascending
	for (size_t u = u0; u < u1; u++)
		if (long_targets.empty() || long_targets.back() != db->h_unit_target[u]) long_targets.push_back(db->h_unit_target[u]);
	std::sort(long_targets.begin(), long_targets.end());
	long_targets.erase(std::unique(long_targets.begin(), long_targets.end()), long_targets.end());
	const size_t n_buckets = long_targets.size() * n_assays_all;
	if (n_buckets >= (size_t(1) << 32)) {
		ctx->err = "too many (target, assay) lists on the long targets of this batch";
		return TNT_E_LIMIT;
	}
	// The lists are assembled AND culled on the device (tnt_sort.hip): only the seeds the first cull keeps, and one group
	// record per list, come back.  TNT_MI355X_LONG_HOST_CULL=1 keeps round 2's route (every seed to the host, cull_oligo_match
	// replayed per list on the worker threads) as the cross-check of the device cull.
	const bool host_cull = getenv("TNT_MI355X_LONG_HOST_CULL") != nullptr || ctx->long_primers; // (the segments of primers beyond 128 bases: their seeds join one list on the host, below)
	HostPool &workers = HostPool::get();
	// (all device arrays of this path come out of one allocation the context keeps: tnt_arena.h)
	DeviceArena &arena = ctx->long_arena;
	const size_t n_lists_max = std::min(n_raw, n_buckets);
	HIP_TRY(arena.reserve(8 * 256 + db->n_targets * sizeof(uint32_t) + n_raw * (2 * sizeof(Item) + sizeof(uint32_t)) +
	                      std::max(long_sort_bytes(n_raw), long_cull_bytes(n_raw, n_lists_max))),
	        ctx->err);
	uint32_t *const d_rank = arena.take<uint32_t>(db->n_targets), *const d_list = arena.take<uint32_t>(n_raw);
	Item *const d_items = arena.take<Item>(n_raw), *const d_kept = arena.take<Item>(n_raw);
	if (!d_rank || !d_list || !d_items || !d_kept) {
		ctx->err = "the device arena of the long-target path is too small";
		return TNT_E_DEVICE;
	}
	{
		std::vector<uint32_t> rank(db->n_targets, 0);
		for (size_t r = 0; r < long_targets.size(); r++) rank[long_targets[r]] = uint32_t(r);
		HIP_TRY(hipMemcpy(d_rank, rank.data(), rank.size() * sizeof(uint32_t), hipMemcpyHostToDevice), ctx->err);
		HIP_TRY(long_sort_seeds(ctx->seeds_out.p, uint32_t(n_raw), d_rank, ctx->assays.p, uint32_t(n_assays_all), uint32_t(n_buckets), d_items, d_list, arena, ctx->stream), ctx->err);
		if (timing) HIP_TRY(hipStreamSynchronize(ctx->stream), ctx->err);
	}
	lap("keys + sort");

	Item *bind_items = nullptr; // the kept seeds, list after list (pinned host copy: the pairing reads it)
	size_t n_bind = 0;
	bool dev_pair = false;       // the PCR lists are sequenced and paired on the device
	bool host_copies = false;    // the host threads need the kept seeds, their results and the bound records
	std::vector<uint8_t> take;   // [group] the device takes the list
	std::vector<Group> groups;
	std::vector<uint32_t> unit_target;
	std::vector<int32_t> own_lo, own_hi;
	if (!host_cull) {
		// ---- first cull of every list (amplicon_search.cpp:128-131) on the device
		Group *d_groups = nullptr;
		uint32_t n_kept = 0, n_groups = 0;
		HIP_TRY(long_first_cull(d_items, d_list, uint32_t(n_raw), ctx->assays.p, uint32_t(n_assays_all), ctx->opt.max_len, ctx->opt.single_primer_pcr, d_kept, &n_kept,
		                        &d_groups, &n_groups, arena, ctx->stream),
		        ctx->err);
		std::vector<Group> all(n_groups);
		if (n_groups) HIP_TRY(hipMemcpyAsync(all.data(), d_groups, size_t(n_groups) * sizeof(Group), hipMemcpyDeviceToHost, ctx->stream), ctx->err);
		HIP_TRY(hipStreamSynchronize(ctx->stream), ctx->err);
		n_bind = n_kept;
		for (const Group &g : all) {
			if (g.count == 0) continue;
			groups.push_back(g); // (a list beyond the host replay's 1 048 576 entries has to be one the device takes and keeps: below)
		}
		// (Group::target = rank of the target among the windowed ones)
		unit_target = long_targets;
		own_lo.assign(unit_target.size(), INT32_MIN);
		own_hi.assign(unit_target.size(), INT32_MAX);
		lap("first cull (device)");
		// The bind / cull sequence and the pairing of the PCR lists run on the device (tnt_sort.hip, long_pair_device) unless a switch
		// or the context says otherwise: the lists of probe / padlock assays (one pass over their bound seeds) and of the assays that
		// go through the cache replay stay with the host threads below, and so does a list the device hands back (flagged).
		dev_pair = n_bind != 0 && !cull_rows && !getenv("TNT_MI355X_LONG_HOST_PAIR");
		if (dev_pair) {
			take.assign(groups.size(), 0);
			bool any = false;
			const bool hash_lists = !getenv("TNT_MI355X_LONG_HOST_HASH_LISTS");
			for (size_t g = 0; g < groups.size(); g++) {
				const AssayDev &A = ctx->h_assays[groups[g].assay & GROUP_ASSAY_MASK];
				const bool replayed_here = ctx->cache_replay && (A.cache_replay & 1);
				// 1: a PCR list (long_pair_device); 2: a probe / padlock list (long_hash_device; not the segments of a probe beyond 128
				// bases, whose hits carry their diagonal to the merge); 0: the host threads'
				take[g] = replayed_here ? 0 : (A.kind == 0 ? 1 : ((hash_lists && !(A.cache_replay & 2)) ? 2 : 0));
				any = any || take[g];
				host_copies = host_copies || !take[g];
			}
			dev_pair = any;
		}
		if (!dev_pair) host_copies = true;
		for (size_t g = 0; g < groups.size(); g++)
			if (groups[g].count > uint32_t(PAIR_HOST) && !(dev_pair && take[g])) {
				ctx->err = "more than 1048576 seeds of one assay on one target after the first cull (a list the host threads would have to pair)";
				return TNT_E_LIMIT;
			}
		if (n_bind && host_copies) {
			HIP_TRY(ctx->pin_bind.ensure(n_bind), ctx->err);
			bind_items = ctx->pin_bind.p;
			HIP_TRY(hipMemcpyAsync(bind_items, d_kept, n_bind * sizeof(Item), hipMemcpyDeviceToHost, ctx->stream), ctx->err); // (done by the first sync of the bind loop)
		}
	} else {
		host_copies = true;
		HIP_TRY(ctx->pin_items.ensure(n_raw), ctx->err);
		HIP_TRY(ctx->pin_list.ensure(n_raw), ctx->err);
		Item *const sorted = ctx->pin_items.p;
		uint32_t *const list_of = ctx->pin_list.p;
		HIP_TRY(hipMemcpyAsync(sorted, d_items, n_raw * sizeof(Item), hipMemcpyDeviceToHost, ctx->stream), ctx->err);
		HIP_TRY(hipMemcpyAsync(list_of, d_list, n_raw * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream), ctx->err);
		HIP_TRY(hipStreamSynchronize(ctx->stream), ctx->err);
		if (ctx->long_primers) {
			// Primers, TaqMan probes and padlock arms of more than 128 bases were seeded in segments of 128 query words, one
			// virtual assay per segment (tnt_ctx_create): their seeds join the list of the assay they belong to, in the order of
			// the whole oligo's word indices (the keys of tnt_sort.hip with the segment's offset added), and of the seeds two
			// segments found on one diagonal of one (oligo, strand) only the first stays -- match_oligo_to_*_strand sorts by
			// diagonal and keeps the smallest word index (bind_oligo.cpp:98-99); inside a segment the seed kernel did that.
			struct Row {
				uint32_t list;
				unsigned long long key;
				uint32_t at;
			};
			struct Dup {
				uint32_t list, who; // who: role | strand << 2
				long long diag;
				uint32_t qloc, row;
			};
			std::vector<Row> rows(n_raw);
			std::vector<Dup> dups;
			for (size_t i = 0; i < n_raw; i++) {
				const Item &it = sorted[i];
				const uint32_t slot = it.info >> ITEM_SLOT_SHIFT, role = slot % 3u, minus = (it.info & ITEM_MINUS) ? 1u : 0u;
				const uint32_t rank = uint32_t(list_of[i] / n_assays_all), assay = uint32_t(list_of[i] % n_assays_all), base = ctx->h_assay_base[assay];
				const unsigned long long qloc = (unsigned long long)(int64_t(it.info & 127u) + int64_t(ctx->h_slots[slot].qloc_bias));
				unsigned long long key;
				if (ctx->h_assays[base].kind == 0) {
					const unsigned long long grp = role == 2 ? (minus ? 4u : 5u) : ((minus ? 0u : 2u) + role);
					key = ((unsigned long long)it.tloc << 14) | (grp << 11) | qloc;
				} else {
					const unsigned long long listid = (minus ? 0u : 2u) + (role == 1 ? 0u : 1u);
					key = (listid << 46) | ((unsigned long long)(int64_t(qloc) - int64_t(it.tloc) + (int64_t(1) << 33)) << 11) | qloc;
				}
				rows[i] = Row{uint32_t(size_t(rank) * n_assays_all + base), key, uint32_t(i)};
				if (ctx->h_segment[base] >= 0) dups.push_back(Dup{rows[i].list, role | (minus << 2), int64_t(qloc) - int64_t(it.tloc), uint32_t(qloc), uint32_t(i)});
			}
			std::vector<uint8_t> dropped(n_raw, 0);
			std::sort(dups.begin(), dups.end(), [](const Dup &a, const Dup &b) {
				if (a.list != b.list) return a.list < b.list;
				if (a.who != b.who) return a.who < b.who;
				if (a.diag != b.diag) return a.diag < b.diag;
				return a.qloc < b.qloc;
			});
			for (size_t i = 1; i < dups.size(); i++)
				if (dups[i].list == dups[i - 1].list && dups[i].who == dups[i - 1].who && dups[i].diag == dups[i - 1].diag) dropped[dups[i].row] = 1;
			std::sort(rows.begin(), rows.end(), [](const Row &a, const Row &b) { return a.list != b.list ? a.list < b.list : (a.key != b.key ? a.key < b.key : a.at < b.at); });
			std::vector<Item> merged;
			std::vector<uint32_t> merged_list;
			merged.reserve(n_raw), merged_list.reserve(n_raw);
			for (const Row &r : rows)
				if (!dropped[r.at]) merged.push_back(sorted[r.at]), merged_list.push_back(r.list);
			n_raw = merged.size();
			if (n_raw) memcpy(sorted, merged.data(), n_raw * sizeof(Item)), memcpy(list_of, merged_list.data(), n_raw * sizeof(uint32_t));
		}
		std::vector<uint32_t> seg_begin, seg_assay;
		for (size_t i = 0; i < n_raw; i++)
			if (i == 0 || list_of[i] != list_of[i - 1]) {
				seg_begin.push_back(uint32_t(i));
				seg_assay.push_back(uint32_t(list_of[i] % n_assays_all));
			}
		seg_begin.push_back(uint32_t(n_raw));
		lap("lists to the host");

		// ---- first cull of every list (amplicon_search.cpp:128-131) on the host
		std::vector<uint8_t> valid(n_raw, 0), minus_first(seg_assay.size(), 0);
		std::atomic<bool> overflow(false);
		const bool started = workers.run(seg_assay.size(), [&](size_t g) {
			const uint32_t b = seg_begin[g], e = seg_begin[g + 1];
			const AssayDev &A = ctx->h_assays[seg_assay[g]];
			if (e - b > uint32_t(PAIR_HOST)) {
				overflow = true;
				return;
			}
			if (A.kind != 0) { // probe / padlock lists are bound as a whole (hash overload, bind_oligo.cpp:124-454)
				for (uint32_t i = b; i < e; i++) valid[i] = 1;
				return;
			}
			// amplicon(): the assembled match list (already in the order the cull's stable sort by target_loc leaves it in)
			static thread_local std::unique_ptr<Entry[]> store; // the list and the working space of its sort
			if (!store) store.reset(new Entry[2 * size_t(PAIR_HOST)]);
			ListT<PAIR_HOST> l;
			l.e = store.get(), l.scratch = store.get() + PAIR_HOST;
			l.n = int(e - b);
			for (uint32_t i = b; i < e; i++) {
				Entry &en = l.e[i - b];
				const uint32_t role = (sorted[i].info >> ITEM_SLOT_SHIFT) % 3u;
				en.loc5 = en.loc3 = 0;
				en.tloc = sorted[i].tloc;
				en.rec = int32_t(i); // remember which item this is
				en.mask = uint8_t((role == 0 ? MF : (role == 1 ? MR : MP)) | ((sorted[i].info & ITEM_MINUS) ? MMINUS : MPLUS));
			}
			int cm = 0, cp = 0;
			cull(l, ctx->opt.max_len, A.has_probe != 0, ctx->opt.single_primer_pcr != 0, cm, cp); // amplicon_search.cpp:128
			for (int i = 0; i < l.n; i++) valid[size_t(l.e[i].rec)] = 1;
			minus_first[g] = cm < cp ? 1 : 0; // :131
		});
		if (!started) {
			ctx->err = "could not start a host thread for the lists of long targets";
			return TNT_E_DEVICE;
		}
		if (overflow) {
			ctx->err = "more than 1048576 seeds of one assay on one target";
			return TNT_E_LIMIT;
		}

		lap("first cull (host)");
		// ---- the kept seeds become bind work items, grouped per (target, assay)
		{
			// (two passes over the lists on the worker threads: count, then copy to the counted places)
			const size_t n_seg = seg_assay.size();
			std::vector<uint32_t> first_of(n_seg + 1, 0);
			workers.run(n_seg, [&](size_t g) {
				uint32_t c = 0;
				for (uint32_t i = seg_begin[g]; i < seg_begin[g + 1]; i++) c += valid[i] ? 1u : 0u;
				first_of[g + 1] = c;
			});
			for (size_t g = 0; g < n_seg; g++) first_of[g + 1] += first_of[g];
			n_bind = first_of[n_seg];
			HIP_TRY(ctx->pin_bind.ensure(n_bind), ctx->err);
			bind_items = ctx->pin_bind.p;
			workers.run(n_seg, [&](size_t g) {
				uint32_t o = first_of[g];
				for (uint32_t i = seg_begin[g]; i < seg_begin[g + 1]; i++)
					if (valid[i]) bind_items[o++] = sorted[i];
			});
			for (size_t g = 0; g < n_seg; g++) {
				const uint32_t count = first_of[g + 1] - first_of[g];
				if (!count) continue;
				const uint32_t t = sorted[seg_begin[g]].target;
				if (unit_target.empty() || unit_target.back() != t) unit_target.push_back(t), own_lo.push_back(INT32_MIN), own_hi.push_back(INT32_MAX);
				groups.push_back(Group{uint32_t(unit_target.size() - 1), seg_assay[g] | (minus_first[g] ? GROUP_MINUS_FIRST : 0u), first_of[g], count});
			}
		}
	}
	if (n_bind == 0) return TNT_OK;
	if (cull_rows) { // tnt_first_cull: the kept seeds of the PCR lists and the strand their group binds first
		HIP_TRY(hipStreamSynchronize(ctx->stream), ctx->err); // (the device route's copy of the kept seeds)
		for (const Group &g : groups) {
			const uint32_t assay = g.assay & GROUP_ASSAY_MASK;
			if (ctx->h_assays[assay].kind != 0) continue; // probe / padlock lists are not culled
			for (uint32_t i = g.first; i < g.first + g.count; i++) {
				const Item &it = bind_items[i];
				const uint32_t slot = it.info >> ITEM_SLOT_SHIFT;
				const int32_t row[7] = {int32_t(db->first_seq_index + it.target), int32_t(assay), int32_t(slot % 3u), (it.info & ITEM_MINUS) ? TNT_MINUS : TNT_PLUS,
				                        int32_t(it.info & 127u) + ctx->h_slots[slot].qloc_bias, int32_t(it.tloc), (g.assay & GROUP_MINUS_FIRST) ? 1 : 0};
				cull_rows->insert(cull_rows->end(), row, row + 7);
			}
		}
		return TNT_OK;
	}

	lap("work items");
	// ---- bind them on the device, a million at a time: bucket -> pre-filter -> bind, as a search phase does
	const DbView v = make_view(db, 0, db->n_targets);
	if (host_copies) HIP_TRY(ctx->pin_result.ensure(n_bind), ctx->err);
	int32_t *result = host_copies ? ctx->pin_result.p : nullptr;
	// device pairing: the results of all passes side by side (pool indices over all passes), the bound records of all passes
	DeviceArena &pair_arena = ctx->long_pair_arena;
	int32_t *d_result_all = nullptr;
	if (dev_pair) {
		if (pair_arena.reserve(std::max(long_pair_bytes(n_bind), long_hash_bytes(n_bind, n_bind / 2 + 65536, groups.size())) + n_bind * sizeof(int32_t) + 4096) != hipSuccess ||
		    !(d_result_all = pair_arena.take<int32_t>(n_bind))) {
			(void)hipGetLastError();
			dev_pair = false; // (no room: the host threads take everything)
			if (!host_copies) {
				host_copies = true;
				HIP_TRY(ctx->pin_bind.ensure(n_bind), ctx->err);
				bind_items = ctx->pin_bind.p;
				HIP_TRY(hipMemcpyAsync(bind_items, d_kept, n_bind * sizeof(Item), hipMemcpyDeviceToHost, ctx->stream), ctx->err);
				HIP_TRY(ctx->pin_result.ensure(n_bind), ctx->err);
				result = ctx->pin_result.p;
			}
		}
	}
	// (up to 16 M items per pass; a pass that binds more records than the device pool holds is repeated with half as many)
	size_t n_pool = 0;
	size_t chunk = std::min(ctx->item_cap, size_t(1) << 24);
	for (size_t c0 = 0; c0 < n_bind;) {
		const size_t nc = std::min(chunk, n_bind - c0);
		HIP_TRY(hipMemsetAsync(cnt, 0, sizeof(uint32_t) * C_LTEXT, ctx->stream), ctx->err); // (not the long-text arena's fill: the texts of the search so far stay)
		if (host_cull) HIP_TRY(hipMemcpyAsync(ctx->S().items.p, bind_items + c0, nc * sizeof(Item), hipMemcpyHostToDevice, ctx->stream), ctx->err);
		else HIP_TRY(hipMemcpyAsync(ctx->S().items.p, d_kept + c0, nc * sizeof(Item), hipMemcpyDeviceToDevice, ctx->stream), ctx->err);
		hipLaunchKernelGGL(iota_kernel, dim3(1024), dim3(256), 0, ctx->stream, ctx->S().qb.p, uint32_t(nc), cnt + C_QB); // the queue: every item
		BindArgs b = bind_args(ctx, v);
		b.groups = nullptr;
		launch_bind_phase(ctx, b, v, db->n_exceptions != 0, ctx->S().qb.p, cnt + C_QB, nullptr);
		uint32_t h[C_COUNT];
		int32_t *const res = host_copies ? result + c0 : nullptr;
		if (host_copies) HIP_TRY(hipMemcpyAsync(res, ctx->S().result.p, nc * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream), ctx->err);
		HIP_TRY(hipMemcpyAsync(h, cnt, sizeof(h), hipMemcpyDeviceToHost, ctx->stream), ctx->err);
		HIP_TRY(hipStreamSynchronize(ctx->stream), ctx->err);
		HIP_TRY(hipGetLastError(), ctx->err);
		if (h[C_ERR] == ERR_POOL_OVERFLOW && chunk > (size_t(1) << 16)) {
			chunk /= 2;
			continue;
		}
		if (h[C_ERR]) {
			ctx->err = err_text(h[C_ERR]);
			return TNT_E_LIMIT;
		}
		const size_t np = std::min<size_t>(h[C_POOL], POOL_CAP), base = n_pool;
		if (base + np >= (size_t(1) << 31)) {
			ctx->err = "more than 2^31 bound oligos on the long targets of this batch";
			return TNT_E_LIMIT;
		}
		if (host_copies) {
			HIP_TRY(ctx->pin_pool.ensure_keep(base + np, base), ctx->err);
			if (np) HIP_TRY(hipMemcpy(ctx->pin_pool.p + base, ctx->S().pool.p, np * sizeof(BoundRec), hipMemcpyDeviceToHost), ctx->err);
		}
		if (dev_pair) {
			if (base + np > ctx->long_pool.n) { // grow, keeping the records of the passes so far
				DevBuf<BoundRec> bigger;
				HIP_TRY(bigger.alloc(std::max<size_t>((base + np) * 2, size_t(1) << 16)), ctx->err);
				if (base) HIP_TRY(hipMemcpy(bigger.p, ctx->long_pool.p, base * sizeof(BoundRec), hipMemcpyDeviceToDevice), ctx->err);
				std::swap(bigger.p, ctx->long_pool.p), std::swap(bigger.n, ctx->long_pool.n);
			}
			if (np) HIP_TRY(hipMemcpyAsync(ctx->long_pool.p + base, ctx->S().pool.p, np * sizeof(BoundRec), hipMemcpyDeviceToDevice, ctx->stream), ctx->err);
			hipLaunchKernelGGL(rebase_kernel, dim3(1024), dim3(256), 0, ctx->stream, ctx->S().result.p, uint32_t(nc), int32_t(base), d_result_all + c0);
			HIP_TRY(hipStreamSynchronize(ctx->stream), ctx->err); // (the next pass overwrites the batch's arrays)
		}
		n_pool += np;
		if (base && host_copies)
			for (size_t i = 0; i < nc; i++)
				if (res[i] >= 0) res[i] = int32_t(base + size_t(res[i]));
		ctx->acc_items += nc;
		c0 += nc;
	}
	lap("bind on the device");
	// ---- the lists on the device: the PCR lists' bind / cull sequence, last sort and pairing loops (amplicon_search.cpp:131-674),
	// the probe / padlock lists' sorts and pairing (probe_search.cpp:67-230, padlock_search.cpp:62-361)
	std::vector<Hit> dev_hits;
	std::vector<uint8_t> flagged;
	std::vector<Hit> dev_hash_hits; // the probe / padlock lists' (long_hash_device)
	bool hash_done = false;
	if (dev_pair) {
		std::vector<uint8_t> take_pcr(groups.size()), take_hash(groups.size());
		bool any_hash = false;
		for (size_t g = 0; g < groups.size(); g++) take_pcr[g] = take[g] == 1, take_hash[g] = take[g] == 2, any_hash = any_hash || take[g] == 2;
		if (any_hash) {
			const int pair_max_len = ctx->opt.assay_format == TNT_ASSAY_MIPS ? ctx->opt.max_len : 0; // tntblast_local.cpp:592,606
			const hipError_t he = long_hash_device(d_kept, uint32_t(n_bind), groups.data(), uint32_t(groups.size()), take_hash.data(), d_result_all, ctx->long_pool.p, ctx->assays.p,
			                                       ctx->h_assays.data(), unit_target.data(), db->first_seq_index, pair_max_len, dev_hash_hits, pair_arena, ctx->stream);
			if (he == hipSuccess) {
				hash_done = true;
			} else if (he == hipErrorOutOfMemory) { // (more bound seeds than the reservation expects: the host threads take these lists)
				(void)hipGetLastError();
				dev_hash_hits.clear();
			} else {
				ctx->err = std::string("the probe / padlock lists of long targets on the device: ") + hipGetErrorString(he);
				return TNT_E_DEVICE;
			}
		}
		const hipError_t pe = long_pair_device(d_kept, uint32_t(n_bind), groups.data(), uint32_t(groups.size()), take_pcr.data(), d_result_all, ctx->long_pool.p, ctx->assays.p,
		                                       ctx->h_assays.data(), unit_target.data(), db->first_seq_index, ctx->opt.max_len, ctx->opt.single_primer_pcr,
		                                       ctx->opt.min_max_primer_clamp, dev_hits, flagged, pair_arena, ctx->stream);
		if (pe == hipErrorOutOfMemory) { // (more bound seeds than the reservation expects: the host threads take the lists)
			(void)hipGetLastError();
			dev_hits.clear();
			flagged.assign(groups.size(), 1);
		} else if (pe != hipSuccess) {
			ctx->err = std::string("pairing the lists of long targets on the device: ") + hipGetErrorString(pe);
			return TNT_E_DEVICE;
		}
		if (!hash_done)
			for (size_t g = 0; g < groups.size(); g++)
				if (take[g] == 2) flagged[g] = 1; // (no room for them on the device: the host threads take them)
		size_t n_back = 0;
		for (size_t g = 0; g < groups.size(); g++) {
			n_back += (take[g] && flagged[g]) ? 1 : 0;
			if (take[g] && flagged[g] && groups[g].count > uint32_t(PAIR_HOST)) {
				ctx->err = "more than 1048576 seeds of one assay on one target after the first cull, in a list the device handed back";
				return TNT_E_LIMIT;
			}
		}
		if (timing) fprintf(stderr, "long targets: %zu PCR and %zu probe / padlock lists on the device, %zu of them handed back (sort_by_oligo_loc not a strict weak order on them), %zu + %zu hits\n",
		                    size_t(std::count(take.begin(), take.end(), uint8_t(1))), size_t(std::count(take.begin(), take.end(), uint8_t(2))), n_back, dev_hits.size(), dev_hash_hits.size());
		if (n_back && !host_copies) { // the host threads need the seeds, results and records after all
			host_copies = true;
			HIP_TRY(ctx->pin_bind.ensure(n_bind), ctx->err);
			bind_items = ctx->pin_bind.p;
			HIP_TRY(ctx->pin_result.ensure(n_bind), ctx->err);
			result = ctx->pin_result.p;
			HIP_TRY(ctx->pin_pool.ensure(n_pool), ctx->err);
			// (the seeds and results of the lists that came back, at their places in the arrays; the bound records of the pass)
			for (size_t g = 0; g < groups.size(); g++) {
				if (!(take[g] && flagged[g])) continue;
				const size_t first = groups[g].first, count = groups[g].count & ~GROUP_ALIVE;
				HIP_TRY(hipMemcpyAsync(bind_items + first, d_kept + first, count * sizeof(Item), hipMemcpyDeviceToHost, ctx->stream), ctx->err);
				HIP_TRY(hipMemcpyAsync(result + first, d_result_all + first, count * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream), ctx->err);
			}
			if (n_pool) HIP_TRY(hipMemcpyAsync(ctx->pin_pool.p, ctx->long_pool.p, n_pool * sizeof(BoundRec), hipMemcpyDeviceToHost, ctx->stream), ctx->err);
			HIP_TRY(hipStreamSynchronize(ctx->stream), ctx->err);
		}
		lap("lists on the device");
	}
	auto on_device = [&](size_t g) { return dev_pair && take[g] && !flagged[g]; };
	for (size_t g = 0; g < groups.size(); g++) ctx->long_pair_stats[on_device(g) ? 0 : ((dev_pair && take[g]) ? 1 : 2)]++;
	ctx->long_pair_stats[3] += dev_hits.size() + dev_hash_hits.size();
	dev_hits.insert(dev_hits.end(), dev_hash_hits.begin(), dev_hash_hits.end()); // (a list's hits come from one side: the order below puts the lists in place)
	const BoundRec *const pool = ctx->pin_pool.p;

	// ---- row a6 on these targets: the groups of the assays that share an oligo across strand concentrations / threshold sets are
	// decided by the reference's per-target BindCache (tntblast_local.cpp:503-506, bind_oligo.cpp:506-512).  As on whole-target work
	// units (enqueue_cache_replay): every one of their seeds is bound in full by the unit-level instance, then the groups of a
	// target are replayed in assay order through pair_group<N, true> -- here on a host thread per target, the memo as a map.
	std::vector<ConflictRec> crecs;
	std::vector<uint32_t> cvals;
	std::vector<std::pair<uint32_t, uint32_t>> cruns; // per target with such groups: its records [first, end)
	std::vector<Item> citems;
	std::vector<BindFull> cfull;
	if (ctx->cache_replay) {
		for (size_t g = 0; g < groups.size(); g++) {
			const Group &G = groups[g];
			if (!(ctx->h_assays[G.assay & GROUP_ASSAY_MASK].cache_replay & 1)) continue;
			const uint32_t count = G.count & ~GROUP_ALIVE;
			if (citems.size() + count >= (size_t(1) << 31) - 2) {
				ctx->err = "more than 2^31 seeds of assays that share an oligo on the long targets of this batch";
				return TNT_E_LIMIT;
			}
			if (cruns.empty() || groups[crecs[cruns.back().first].group].target != G.target) cruns.push_back({uint32_t(crecs.size()), uint32_t(crecs.size())});
			cvals.push_back(uint32_t(crecs.size()));
			crecs.push_back(ConflictRec{uint32_t(g), uint32_t(citems.size()), count, 0u});
			cruns.back().second = uint32_t(crecs.size());
			for (uint32_t k2 = 0; k2 < count; k2++) {
				Item it = bind_items[G.first + k2];
				it.info &= ~ITEM_PHASE_B;
				citems.push_back(it);
			}
		}
		cfull.resize(citems.size());
		DevBuf<BindFull> d_full;
		const size_t step = std::min(ctx->item_cap, size_t(1) << 20);
		if (!citems.empty()) HIP_TRY(d_full.alloc(std::min(step, citems.size())), ctx->err);
		for (size_t c0 = 0; c0 < citems.size(); c0 += step) {
			const size_t nc = std::min(step, citems.size() - c0);
			HIP_TRY(hipMemsetAsync(cnt, 0, sizeof(uint32_t) * C_LTEXT, ctx->stream), ctx->err);
			HIP_TRY(hipMemcpyAsync(ctx->S().items.p, citems.data() + c0, nc * sizeof(Item), hipMemcpyHostToDevice, ctx->stream), ctx->err);
			BindArgs b = bind_args(ctx, v);
			b.groups = nullptr, b.index = nullptr;
			b.n_items = uint32_t(nc), b.n_items_dev = nullptr;
			b.full = d_full.p;
			launch_bind(ctx, b, ctx->bind_blocks, 0); // status, Tm, dH, dS, dG, anchors, counts, footprint, text -- whatever its own thresholds say
			uint32_t h[C_COUNT];
			HIP_TRY(hipMemcpyAsync(h, cnt, sizeof(h), hipMemcpyDeviceToHost, ctx->stream), ctx->err);
			HIP_TRY(hipMemcpyAsync(cfull.data() + c0, d_full.p, nc * sizeof(BindFull), hipMemcpyDeviceToHost, ctx->stream), ctx->err);
			HIP_TRY(hipStreamSynchronize(ctx->stream), ctx->err);
			HIP_TRY(hipGetLastError(), ctx->err);
			if (h[C_ERR]) {
				ctx->err = err_text(h[C_ERR]);
				return TNT_E_LIMIT;
			}
		}
		lap("cache replay: full binds");
	}
	std::vector<uint32_t> corigin(citems.size());
	std::vector<BoundRec> cpool(citems.size());
	auto replayed = [&](const Group &G) { return ctx->cache_replay && (ctx->h_assays[G.assay & GROUP_ASSAY_MASK].cache_replay & 1) != 0; };
	// ---- the bind / cull sequence and the pairing loops on the whole lists (pair_group, compiled for the host)
	size_t hit_cap = std::max<size_t>(1024, n_pool);
	std::unique_ptr<Hit[]> hits(new Hit[hit_cap]); // (not initialised: a record is 1.4 kB, pages are touched as hits arrive)
	for (;;) {
		uint32_t n_hits = 0, err = 0, zero = 0;
		PairArgs p;
		memset(&p, 0, sizeof(p));
		p.groups = groups.data();
		p.items = bind_items;
		p.result = result;
		p.pool = pool;
		p.assays = ctx->h_assays.data();
		p.slots = ctx->h_slots.data();
		p.db.first_seq_index = db->first_seq_index;
		p.max_len = ctx->opt.max_len;
		p.single_primer_pcr = ctx->opt.single_primer_pcr;
		p.min_max_primer_clamp = ctx->opt.min_max_primer_clamp;
		p.assay_format = ctx->opt.assay_format;
		p.target_strand = ctx->opt.target_strand;
		p.hits = hits.get();
		p.hit_cap = uint32_t(std::min<size_t>(hit_cap, UINT32_MAX));
		p.hit_count = &n_hits;
		p.err = &err;
		p.n_big = &zero;
		p.unit_target = unit_target.data();
		p.unit_own_lo = own_lo.data();
		p.unit_own_hi = own_hi.data();
		std::string why;
		std::vector<double> group_ms(timing ? groups.size() : 0);
		std::vector<Item> exp_items;
		std::vector<int32_t> exp_result;
		std::vector<BoundRec> exp_pool;
		if (getenv("TNT_EXP_LONG_COPY")) { // experiment: the pairing reads pageable copies instead of the page-locked arrays
			exp_items.assign(bind_items, bind_items + n_bind), exp_result.assign(result, result + n_bind), exp_pool.assign(pool, pool + n_pool);
			p.items = exp_items.data(), p.result = exp_result.data(), p.pool = exp_pool.data();
			lap("experiment: copies");
		}
		if (!workers.run(groups.size(), [&](size_t g) {
			    if (replayed(groups[g]) || on_device(g)) return; // (below / done above)
			    const auto t0 = std::chrono::steady_clock::now();
			    pair_group<PAIR_HOST>(p, groups[g]);
			    if (timing) group_ms[g] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
		    }, &why)) {
			ctx->err = "the lists of long targets: " + why;
			return TNT_E_DEVICE;
		}
		if (!cruns.empty()) {
			std::fill(corigin.begin(), corigin.end(), 0xFFFFFFFFu);
			uint32_t n_cpool = 0;
			PairArgs pc = p;
			pc.thresh = ctx->h_thresh.data();
			pc.citems = citems.data(), pc.cfull = cfull.data(), pc.corigin = corigin.data();
			pc.cpool = cpool.data(), pc.cpool_cap = uint32_t(cpool.size()), pc.cpool_count = &n_cpool;
			if (!workers.run(cruns.size(), [&](size_t t) {
				    CacheMemo memo; // the target's BindCache: (oligo, strand, window) -> the seed evaluated first
				    PairArgs mine = pc;
				    mine.cmemo = &memo;
				    const ConflictRun run{cvals.data() + cruns[t].first, crecs.data()};
				    int n_run = 0;
				    for (uint32_t j = cruns[t].first; j < cruns[t].second; j++) {
					    const ConflictRec r = run[n_run++];
					    pair_group<PAIR_HOST, true>(mine, groups[r.group], run, n_run, r.cfirst);
				    }
			    }, &why)) {
				ctx->err = "the cache replay on long targets: " + why;
				return TNT_E_DEVICE;
			}
			lap("cache replay");
		}
		if ((err & ERR_HIT_OVERFLOW) && hit_cap < UINT32_MAX) {
			hit_cap *= 4;
			hits.reset(new Hit[hit_cap]);
			continue;
		}
		if (err) {
			ctx->err = err_text(err);
			return TNT_E_LIMIT;
		}
		if (timing && !groups.empty()) {
			std::vector<double> d = group_ms;
			std::sort(d.begin(), d.end());
			double sum = 0;
			for (double x : d) sum += x;
			size_t big = 0;
			for (const Group &g : groups) big = std::max<size_t>(big, g.count & ~GROUP_ALIVE);
			fprintf(stderr, "long targets: %zu lists (longest %zu seeds, %zu items, %zu bound): per list min %.2f median %.2f max %.2f ms, sum %.1f ms\n", groups.size(), big, n_bind, n_pool,
			        d.front(), d[d.size() / 2], d.back(), sum);
		}
		lap("pairing");
		if (!dev_hits.empty()) { // the device's hits behind the host threads': a list's hits come from one side, in the order of its loops
			if (size_t(n_hits) + dev_hits.size() > hit_cap) {
				std::unique_ptr<Hit[]> bigger(new Hit[size_t(n_hits) + dev_hits.size()]);
				if (n_hits) memcpy(bigger.get(), hits.get(), size_t(n_hits) * sizeof(Hit));
				hits = std::move(bigger);
				hit_cap = size_t(n_hits) + dev_hits.size();
			}
			memcpy(hits.get() + n_hits, dev_hits.data(), dev_hits.size() * sizeof(Hit));
			n_hits += uint32_t(dev_hits.size());
		}
		// The workers took the groups as they came: put the hits back into the order one thread produces -- groups by
		// (target, assay position), the hits of a group in the order its worker wrote them (a worker's slots ascend) --, so
		// that the output does not depend on the scheduling.
		std::map<std::pair<int32_t, int32_t>, uint32_t> position; // (assay id, degen id) -> first position in the assay list
		for (size_t ai = ctx->h_assays.size(); ai-- > 0;) position[{ctx->h_assays[ai].id, ctx->h_assays[ai].degen_id}] = uint32_t(ai);
		std::vector<std::pair<uint64_t, uint32_t>> order(n_hits); // ((target, assay position), slot)
		for (uint32_t i = 0; i < n_hits; i++) {
			const tnt_hit &x = reinterpret_cast<const tnt_hit &>(hits[i]);
			order[i] = {(uint64_t(uint32_t(x.seq_index)) << 32) | position[{x.assay_id, x.degen_id}], i};
		}
		std::sort(order.begin(), order.end());
		out.resize(n_hits);
		for (uint32_t i = 0; i < n_hits; i++) memcpy(&out[i], &hits[order[i].second], sizeof(tnt_hit));
		break;
	}
	lap("hits in order");
	return TNT_OK;
}

// The targets that are seeded in windows, a few at a time: the device arrays of a pass are ~150 bytes per seed (tnt_sort.hip), so a
// database that holds a whole genome goes through in passes of whole records whose expected seeds stay below 256 M
// (TNT_MI355X_LONG_PASS_SEEDS; a single record beyond that is a pass of its own).  Hits come out in target order.
static int search_long_targets(tnt_ctx *ctx, tnt_db *db, std::vector<tnt_hit> &out, std::vector<int32_t> *cull_rows = nullptr)
{
	const size_t rows0 = cull_rows ? cull_rows->size() : 0;
	double cap = 256.0e6;
	if (const char *e = getenv("TNT_MI355X_LONG_PASS_SEEDS")) cap = std::max(1.0, atof(e));
	for (;;) {
		out.clear();
		for (int i = 0; i < 4; i++) ctx->long_pair_stats[i] = 0;
		if (cull_rows) cull_rows->resize(rows0);
		const size_t u0 = db->n_short_units, u1 = db->h_unit_target.size();
		const double per_position = double(ctx->n_entries) / double(size_t(1) << (2 * ctx->opt.hash_word_size));
		bool restart = false;
		for (size_t ua = u0; ua < u1 && !restart;) {
			size_t ub = ua;
			double seeds = 0.0;
			while (ub < u1) { // one more whole record
				size_t ue = ub;
				double s = 0.0;
				while (ue < u1 && db->h_unit_target[ue] == db->h_unit_target[ub]) s += double(db->h_unit_end[ue] - db->h_unit_begin[ue]) * per_position, ue++;
				if (ub > ua && seeds + s > cap) break;
				seeds += s, ub = ue;
			}
			std::vector<tnt_hit> part;
			const int rc = search_long_pass(ctx, db, ua, ub, part, cull_rows);
			if (rc == LONG_RESTART) restart = true;
			else if (rc) return rc;
			else out.insert(out.end(), part.begin(), part.end());
			ua = ub;
		}
		if (!restart) return TNT_OK;
	}
}

#endif
