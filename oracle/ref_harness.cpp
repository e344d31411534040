// ref_harness.cpp -- TEST INFRASTRUCTURE ONLY.
//
// A thin driver, written for this repository, that is compiled AGAINST the
// reference's own sources where they lie under /root/reference (never copied
// into this repository) and linked with the reference's own objects into
// oracle/_ref/libtntref.so.  It exposes the reference's implementation of the
// hot path through the same C structs as include/tnt_mi355x.h (prefix
// `tntref_`) so that the CPU restatement (oracle/tnt_oracle.cpp) and the HIP
// path can be diffed against the real thing.
//
// It drives exactly the functions the OpenMP loop body of
// tntblast_local.cpp:388-793 calls: DNAHash::hash, amplicon(), padlock(),
// hybrid() (tntblast.h:409-472), and for unit-level checks
// match_oligo_to_*_strand / NucCruc directly.
//
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
// load this library.  It cannot be built without /root/reference.

#define private public // NucCruc tables are private (nuc_cruc.h:527-675); read-only peeking
#include "nuc_cruc.h"
#undef private

#include "tntblast.h"
#include "hybrid_sig.h"
#include "seq_hash.h"
#include "seq.h"
#include "compress.h"
#include "degenerate_na.h"
#include "sequence_data.h"

#include <omp.h>
#include <cstring>
#include <sstream>
#include <string>
#include <new>
#include <vector>
#include <list>
#include <unordered_map>

#include "../include/tnt_mi355x.h"

using namespace std;

namespace {

string g_err;

void copy_text(char *dst, const string &s)
{
	size_t n = min(s.size(), size_t(TNT_ALIGN_TEXT - 1));
	memcpy(dst, s.data(), n);
	dst[n] = 0;
}

void setup_melt(NucCruc &melt, const tnt_options &opt)
{
	// tntblast_local.cpp:363-367
	melt.Salt(opt.salt);
	melt.dangle(opt.allow_dangle_5 != 0, opt.allow_dangle_3 != 0);
	melt.dinkelbach(opt.use_dinkelbach != 0);
}

SEQPTR make_seqptr(const uint8_t *codes, uint64_t begin, uint64_t end)
{
	const unsigned int len = (unsigned int)(end - begin);
	SEQPTR p = new unsigned char[SEQ_HEADER_SIZE + len + 1];
	*((unsigned int *)p) = len;
	memcpy(SEQ_START(p), codes + begin, len);
	return p;
}

// Load the window exactly as bind_oligo.cpp:515-592 (minus) / :1218-1290 (plus) do.
void load_window(NucCruc &melt, const unsigned char *seq, unsigned int start, unsigned int stop, bool minus)
{
	melt.clear_target();

	for (unsigned int i = start; i < stop; i++) {
		const unsigned char b = seq[i];
		if (minus) {
			static const BASE::nucleic_acid comp[16] = {
				BASE::T, BASE::G, BASE::C, BASE::A, BASE::I, BASE::K, BASE::Y, BASE::S,
				BASE::B, BASE::W, BASE::R, BASE::D, BASE::M, BASE::H, BASE::V, BASE::N};
			if (b <= DB_N) melt.push_front_target(comp[b]);
		} else {
			static const BASE::nucleic_acid same[16] = {
				BASE::A, BASE::C, BASE::G, BASE::T, BASE::I, BASE::M, BASE::R, BASE::S,
				BASE::V, BASE::W, BASE::Y, BASE::H, BASE::K, BASE::D, BASE::B, BASE::N};
			if (b <= DB_N) melt.push_back_target(same[b]);
		}
	}
}

} // namespace

// defined in the reference's amplicon_search.cpp (external linkage, declared there only)
pair<unsigned int, unsigned int> cull_oligo_match(list<oligo_info> &m_match_list, const unsigned int &m_max_amplicon_len,
	const bool &m_has_probe, const bool &m_single_primer_pcr);

extern "C" {

const char *tntref_last_error(void) { return g_err.c_str(); }

int tntref_num_threads(void) { return omp_get_max_threads(); }

// G1: tables at (temperature, salt)
int tntref_tables(const tnt_options *opt, int32_t *delta_g, float *param_H, float *param_S,
                  float *loop_S, float *bulge_S, float *hairpin_S, float *scalars)
{
	try {
		NucCruc melt(NucCruc::SANTA_LUCIA, opt->target_t);
		melt.Salt(opt->salt);
		if (delta_g) for (int i = 0; i < 2401; i++) delta_g[i] = melt.delta_g[i];
		if (param_H) memcpy(param_H, melt.param_H, 2401 * sizeof(float));
		if (param_S) memcpy(param_S, melt.param_S, 2401 * sizeof(float));
		if (loop_S) memcpy(loop_S, melt.param_loop_S, (MAX_LOOP_LENGTH + 1) * sizeof(float));
		if (bulge_S) memcpy(bulge_S, melt.param_bulge_S, (MAX_BULGE_LENGTH + 1) * sizeof(float));
		if (hairpin_S) memcpy(hairpin_S, melt.param_hairpin_S, (MAX_HAIRPIN_LENGTH + 1) * sizeof(float));
		if (scalars) {
			scalars[0] = melt.param_init_H;
			scalars[1] = melt.param_init_S;
			scalars[2] = melt.param_AT_closing_H;
			scalars[3] = melt.param_AT_closing_S;
			scalars[4] = melt.param_symmetry_S;
			scalars[5] = melt.param_SALT;
			scalars[6] = melt.param_asymmetric_loop_dS;
			scalars[7] = melt.param_bulge_AT_closing_S;
		}
		// the loop/hairpin terminal tables must be plain copies (nuc_cruc_santa_lucia.cpp:591-603)
		if (memcmp(melt.param_loop_terminal_H, melt.param_H, sizeof(melt.param_H)) ||
		    memcmp(melt.param_loop_terminal_S, melt.param_S, sizeof(melt.param_S))) {
			g_err = "loop terminal tables differ from param_H/S";
			return -1;
		}
		return 0;
	} catch (const char *e) {
		g_err = e;
		return -1;
	}
}

// special tri/tetra hairpin loops (nuc_cruc.h:571-588, nuc_cruc_santa_lucia.cpp:773-1180)
int tntref_hairpin_special(float *H, float *S, int *n)
{
	try {
		NucCruc melt(NucCruc::SANTA_LUCIA, 310.15f);
		*n = NucCruc::NUM_SPECIAL_HAIRPIN_LOOP;
		memcpy(H, melt.param_hairpin_special_H, sizeof(float) * NucCruc::NUM_SPECIAL_HAIRPIN_LOOP);
		memcpy(S, melt.param_hairpin_special_S, sizeof(float) * NucCruc::NUM_SPECIAL_HAIRPIN_LOOP);
		if (memcmp(melt.param_hairpin_terminal_H, melt.param_H, sizeof(melt.param_H)) ||
		    memcmp(melt.param_hairpin_terminal_S, melt.param_S, sizeof(melt.param_S))) {
			g_err = "hairpin terminal tables differ from param_H/S";
			return -1;
		}
		return 0;
	} catch (const char *e) {
		g_err = e;
		return -1;
	}
}

// G4: seeds of every oligo against every target, both strands, after the
// per-(oligo,strand) diagonal dedup (bind_oligo.cpp:84-122).
int tntref_seeds(const tnt_options *opt, const char *const *oligos, int n_oligos, const uint8_t *codes,
                 const uint64_t *offsets, size_t n_targets, uint32_t first_seq_index, tnt_seed **out,
                 size_t *n_out)
{
	try {
		vector<tnt_seed> all;
		DNAHash dbase(opt->hash_word_size);

		for (size_t t = 0; t < n_targets; t++) {
			SEQPTR seq = make_seqptr(codes, offsets[t], offsets[t + 1]);
			if (SEQ_SIZE(seq) >= dbase.min_sequence_size()) { // tntblast_local.cpp:513
				dbase.hash(seq, SEQ_SIZE(seq), 0, SEQ_SIZE(seq));
				for (int o = 0; o < n_oligos; o++) {
					for (int s = 0; s < 2; s++) {
						list<oligo_info> l;
						if (s == 0) match_oligo_to_minus_strand(l, dbase, oligos[o], oligo_info::F);
						else match_oligo_to_plus_strand(l, dbase, oligos[o], oligo_info::F);
						for (list<oligo_info>::const_iterator i = l.begin(); i != l.end(); ++i) {
							tnt_seed sd;
							sd.seq_index = int(first_seq_index + t);
							sd.oligo = o;
							sd.strand = (s == 0) ? TNT_MINUS : TNT_PLUS;
							sd.query_loc = int(i->query_loc);
							sd.target_loc = int(i->target_loc);
							all.push_back(sd);
						}
					}
				}
			}
			delete[] seq;
		}
		*n_out = all.size();
		*out = (tnt_seed *)malloc(sizeof(tnt_seed) * max<size_t>(1, all.size()));
		memcpy(*out, all.data(), sizeof(tnt_seed) * all.size());
		return 0;
	} catch (const char *e) {
		g_err = e;
		return -1;
	}
}

// G2: one NucCruc evaluation per request, all derived quantities reported.
int tntref_bind_batch(const tnt_options *opt, const char *const *oligos, int n_oligos, const uint8_t *codes,
                      const uint64_t *offsets, size_t n_targets, uint32_t first_seq_index,
                      const tnt_bind_req *req, size_t n, tnt_bind *out)
{
	int rc = 0;
#pragma omp parallel
	{
		try {
			NucCruc melt(NucCruc::SANTA_LUCIA, opt->target_t);
			setup_melt(melt, *opt);
			unordered_map<BindCacheKey, BindCacheValue> cache;

#pragma omp for schedule(dynamic, 64)
			for (long k = 0; k < (long)n; k++) {
				const tnt_bind_req &r = req[k];
				tnt_bind &b = out[k];
				memset(&b, 0, sizeof(b));

				const size_t t = size_t(r.seq_index) - first_seq_index;
				if (t >= n_targets || r.oligo < 0 || r.oligo >= n_oligos) {
					b.status = -1;
					continue;
				}
				const unsigned char *seq = codes + offsets[t];
				const unsigned int seq_len = (unsigned int)(offsets[t + 1] - offsets[t]);
				const string oligo = oligos[r.oligo];
				const bool minus = (r.strand == TNT_MINUS);
				const bool probe = (r.role == 1);

				const float min_tm = probe ? opt->min_probe_tm : opt->min_primer_tm;
				const float max_tm = probe ? opt->max_probe_tm : opt->max_primer_tm;
				const float min_dg = probe ? opt->min_probe_dg : opt->min_primer_dg;
				const float max_dg = probe ? opt->max_probe_dg : opt->max_primer_dg;
				const unsigned int clamp5 = probe ? opt->probe_clamp_5 : 0;
				const unsigned int clamp3 = probe ? opt->probe_clamp_3 : opt->primer_clamp;

				melt.set_query(oligo);
				melt.strand(r.ct, 0.0f);

				const unsigned int window = melt.size_query();
				const unsigned int target_length = window + 2 * NUM_FLANK_BASE;
				// bind_oligo.cpp:502-505
				unsigned int target_start = max(int(r.target_loc) - int(r.query_loc + NUM_FLANK_BASE), 0);
				unsigned int target_stop = min(target_start + target_length, seq_len);
				b.target_start = target_start;
				b.target_stop = target_stop;

				load_window(melt, seq, target_start, target_stop, minus);

				const float tm = melt.approximate_tm_heterodimer();
				b.tm = tm;
				b.dH = melt.delta_H();
				b.dS = melt.delta_S();
				b.dg = melt.delta_G();
				b.valid = melt.curr_align.valid ? 1 : 0;
				b.n_max = int(melt.max_ptr.size());
				b.dp_max = melt.max_ptr.empty() ? -1 : melt.max_ptr[0]->M;

				// threshold cascade, bind_oligo.cpp:600-714
				int status = 0;
				if (tm < min_tm || tm > max_tm) status = 1;
				else if (b.dg < min_dg || b.dg > max_dg) status = 2;

				if (b.valid) {
					// The remaining quantities are only meaningful when an alignment exists
					// (curr_align keeps stale first/last_match otherwise: SURVEY Q11).
					b.anchor5 = melt.anchor5_query();
					b.anchor3 = melt.anchor3_query();
					b.num_mm = melt.num_mismatch();
					b.num_gap = melt.num_gap();
					b.poly_degen = melt.max_contiguous_target_degen();

					if (!status) {
						if ((unsigned)b.anchor5 < clamp5) status = 3;
						else if ((unsigned)b.anchor3 < clamp3) status = 4;
						else if (b.num_mm > opt->max_mismatch) status = 5;
						else if (b.num_gap > opt->max_gap) status = 6;
						else if (b.poly_degen > opt->max_poly_degen) status = 7;
					}

					pair<unsigned int, unsigned int> qr, tr;
					melt.alignment_range(qr, tr);
					b.q_first = int(qr.first);
					b.q_last = int(qr.second);
					b.t_first = int(tr.first);
					b.t_last = int(tr.second);

					int t5 = target_start, t3 = target_start;
					if (minus) { // bind_oligo.cpp:721-731
						t5 += target_stop - target_start - 1 - (int)tr.second;
						t3 += target_stop - target_start - 1 - (int)tr.first;
						t5 -= (int)qr.first;
						t3 += (int)(window - 1) - (int)qr.second;
					} else { // bind_oligo.cpp:1424-1434
						t5 += (int)tr.first;
						t3 += (int)tr.second;
						t3 += (int)qr.first;
						t5 -= (int)(window - 1) - (int)qr.second;
					}
					b.loc5 = t5;
					b.loc3 = t3;

					stringstream ss;
					ss << melt;
					copy_text(b.align, ss.str());
				} else if (!status) {
					status = 8; // passes Tm/dG with no alignment (only possible with dG-only thresholds, Q11)
				}
				b.status = status;

				// cross-check against the reference's own wrapper
				if (b.valid) {
					list<oligo_info> l;
					const unsigned char mask = oligo_info::F | (minus ? oligo_info::MINUS_STRAND : oligo_info::PLUS_STRAND);
					l.push_back(oligo_info((unsigned)r.query_loc, (unsigned)r.target_loc, mask));
					cache.clear();
					SEQPTR sp = make_seqptr(codes, offsets[t], offsets[t + 1]);
					if (minus)
						bind_oligo_to_minus_strand(l, oligo_info::F, sp, oligo, melt, cache, min_tm, max_tm, min_dg,
						                           max_dg, clamp5, clamp3, opt->max_gap, opt->max_mismatch,
						                           opt->max_poly_degen);
					else
						bind_oligo_to_plus_strand(l, oligo_info::F, sp, oligo, melt, cache, min_tm, max_tm, min_dg,
						                          max_dg, clamp5, clamp3, opt->max_gap, opt->max_mismatch,
						                          opt->max_poly_degen);
					delete[] sp;
					const bool bound = !l.empty();
					if (bound != (status == 0) ||
					    (bound && (l.front().loc_5 != b.loc5 || l.front().loc_3 != b.loc3 ||
					               l.front().alignment.substr(0, TNT_ALIGN_TEXT - 1) != string(b.align)))) { // (b.align holds the text up to the field's size)
#pragma omp critical
						{
							g_err = "harness cascade disagrees with bind_oligo";
							rc = -2;
						}
					}
				}
			}
		} catch (const char *e) {
#pragma omp critical
			{
				g_err = e;
				rc = -1;
			}
		}
	}
	return rc;
}

// a4 at unit level: the head of amplicon() (amplicon_search.cpp:96-129) with the reference's own functions -- match list in
// assembly order, early exits, cull_oligo_match -- and what is left of the list.  Same records as tntor_first_cull; the
// seventh field (strand bound first) comes from counts that read one node past the list: see the list's block below.
int tntref_first_cull(const tnt_options *opt, const tnt_assay *assays, size_t n_assays, const uint8_t *codes, const uint64_t *offsets,
                      size_t n_targets, uint32_t first_seq_index, int32_t **out, size_t *n_out)
{
	try {
		vector<int32_t> all;
		DNAHash dbase(opt->hash_word_size);
		for (size_t t = 0; t < n_targets; t++) {
			SEQPTR seq = make_seqptr(codes, offsets[t], offsets[t + 1]);
			if (SEQ_SIZE(seq) >= dbase.min_sequence_size()) {
				dbase.hash(seq, SEQ_SIZE(seq), 0, SEQ_SIZE(seq));
				for (size_t a = 0; a < n_assays; a++) {
					const tnt_assay &A = assays[a];
					if (!(A.forward && A.reverse) || opt->assay_format != TNT_ASSAY_PCR) continue;
					// cull_oligo_match reads the strand bits of the node behind the last element when that one is valid:
					// the list header itself, i.e. whatever lies behind it in memory.  Here the header sits in a zeroed block, so
					// that read finds no strand bit ("counts nothing": the deterministic reading the oracle and the product follow);
					// in the reference binary it is stack contents.
					alignas(16) unsigned char ml_block[sizeof(list<oligo_info>) + 2 * sizeof(oligo_info) + 64];
					memset(ml_block, 0, sizeof(ml_block));
					struct ListHolder {
						list<oligo_info> *l;
						explicit ListHolder(void *at) : l(new (at) list<oligo_info>()) {}
						~ListHolder() { l->~list<oligo_info>(); }
					} ml_holder(ml_block);
					list<oligo_info> &ml = *ml_holder.l;
					match_oligo_to_minus_strand(ml, dbase, A.forward, oligo_info::F);
					match_oligo_to_minus_strand(ml, dbase, A.reverse, oligo_info::R);
					const size_t n_minus = ml.size();
					if (n_minus == 0) continue;
					match_oligo_to_plus_strand(ml, dbase, A.forward, oligo_info::F);
					match_oligo_to_plus_strand(ml, dbase, A.reverse, oligo_info::R);
					const size_t n_plus = ml.size();
					if (n_plus == n_minus) continue;
					if (A.probe) {
						match_oligo_to_minus_strand(ml, dbase, A.probe, oligo_info::P);
						match_oligo_to_plus_strand(ml, dbase, A.probe, oligo_info::P);
						if (ml.size() == n_plus) continue;
					}
					const pair<unsigned int, unsigned int> sc = cull_oligo_match(ml, (unsigned int)opt->max_len, A.probe != NULL, opt->single_primer_pcr != 0);
					for (list<oligo_info>::const_iterator i = ml.begin(); i != ml.end(); ++i) {
						const int32_t rec[7] = {int32_t(first_seq_index + t), int32_t(a),
						                        (i->mask & oligo_info::F) ? 0 : ((i->mask & oligo_info::R) ? 1 : 2),
						                        (i->mask & oligo_info::PLUS_STRAND) ? TNT_PLUS : TNT_MINUS, int32_t(i->query_loc),
						                        int32_t(i->target_loc), sc.first < sc.second ? 1 : 0};
						all.insert(all.end(), rec, rec + 7);
					}
				}
			}
			delete[] seq;
		}
		*n_out = all.size() / 7;
		*out = (int32_t *)malloc(sizeof(int32_t) * max<size_t>(1, all.size()));
		memcpy(*out, all.data(), sizeof(int32_t) * all.size());
		return 0;
	} catch (const char *e) {
		g_err = e;
		return -1;
	}
}

// G5: the whole path.  Mirrors the loop body of tntblast_local.cpp:491-706 (without the
// per-hit oligo self-structure lines :656-686, which depend only on the assay).
int tntref_search(const tnt_options *opt, const tnt_assay *assays, size_t n_assays, const uint8_t *codes,
                  const uint64_t *offsets, size_t n_targets, uint32_t first_seq_index, tnt_hit **out,
                  size_t *n_out, int n_threads)
{
	int rc = 0;
	vector<tnt_hit> all;

	// string tables, hybrid_sig.h:536-586 / input.cpp
	unordered_map<string, size_t> str_table;
	vector<hybrid_sig> sigs;
	for (size_t a = 0; a < n_assays; a++) {
		const tnt_assay &A = assays[a];
		hybrid_sig s;
		const size_t name = str_to_index("assay" + to_string(A.id) + "_" + to_string(A.degen_id), str_table);
		if (A.forward && A.reverse && A.probe)
			s = hybrid_sig(name, str_to_index(A.forward, str_table), str_to_index(A.reverse, str_table),
			               str_to_index(A.probe, str_table), A.id);
		else if (A.forward && A.reverse)
			s = hybrid_sig(name, str_to_index(A.forward, str_table), str_to_index(A.reverse, str_table), A.id);
		else if (A.probe)
			s = hybrid_sig(name, str_to_index(A.probe, str_table), A.id);
		else {
			g_err = "assay without oligos";
			return -1;
		}
		s.my_degen_id(A.degen_id);
		s.forward_degen = A.forward_degen;
		s.reverse_degen = A.reverse_degen;
		s.probe_degen = A.probe_degen;
		sigs.push_back(s);
	}
	const vector<string> index_table = ordered_keys(str_table);

	// tntblast_local.cpp:232-234
	const float forward_primer_strand = opt->asymmetric_strand_ratio * opt->primer_strand;
	const float reverse_primer_strand = opt->primer_strand;

	if (n_threads <= 0) n_threads = omp_get_max_threads();

#pragma omp parallel num_threads(n_threads)
	{
		try {
			DNAHash dbase(opt->hash_word_size);
			NucCruc melt(NucCruc::SANTA_LUCIA, opt->target_t);
			setup_melt(melt, *opt);
			unordered_map<BindCacheKey, BindCacheValue> plus_cache, minus_cache;
			unordered_map<string, size_t> local_str_table(str_table);
			vector<pair<size_t, hybrid_sig>> local; // (assay index, hit)

#pragma omp for schedule(dynamic, 16)
			for (long t = 0; t < (long)n_targets; t++) {
				// an exception may not leave a work-sharing loop: the reference's own driver dies of the same
				// throw (tntblast_local.cpp has no handler inside its target loop either); report it instead
				try {
				unordered_map<BindCacheKey, BindCacheValue>().swap(plus_cache);
				unordered_map<BindCacheKey, BindCacheValue>().swap(minus_cache);

				pair<string, SEQPTR> bio_seq;
				bio_seq.first = "seq" + to_string(first_seq_index + t);
				bio_seq.second = make_seqptr(codes, offsets[t], offsets[t + 1]);

				if (SEQ_SIZE(bio_seq.second) >= dbase.min_sequence_size()) {
					dbase.hash(bio_seq.second, SEQ_SIZE(bio_seq.second), 0, SEQ_SIZE(bio_seq.second));

					for (size_t a = 0; a < sigs.size(); a++) {
						const hybrid_sig &sig_ref = sigs[a];
						list<hybrid_sig> res;
						if (sig_ref.has_primers()) { // tntblast_local.cpp:559-611
							switch (opt->assay_format) {
							case ASSAY_PCR:
								res = amplicon(dbase, bio_seq, sig_ref, melt, plus_cache, minus_cache,
								               forward_primer_strand, reverse_primer_strand, opt->probe_strand,
								               opt->min_primer_tm, opt->max_primer_tm, opt->min_primer_dg,
								               opt->max_primer_dg, opt->min_probe_tm, opt->max_probe_tm,
								               opt->min_probe_dg, opt->max_probe_dg, opt->primer_clamp,
								               opt->min_max_primer_clamp, opt->probe_clamp_5, opt->probe_clamp_3,
								               opt->max_gap, opt->max_mismatch, opt->max_poly_degen, opt->max_len,
								               opt->single_primer_pcr != 0, NO_MASK, index_table, local_str_table);
								break;
							case ASSAY_PADLOCK:
							case ASSAY_MIPS:
								res = padlock(dbase, bio_seq, sig_ref, melt, plus_cache, minus_cache,
								              forward_primer_strand, reverse_primer_strand, opt->min_probe_tm,
								              opt->max_probe_tm, opt->min_probe_dg, opt->max_probe_dg,
								              opt->probe_clamp_5, opt->probe_clamp_3, opt->max_gap, opt->max_mismatch,
								              opt->max_poly_degen, opt->target_strand,
								              (opt->assay_format == ASSAY_MIPS) ? opt->max_len : 0, index_table,
								              local_str_table);
								break;
							default:
								break;
							}
						} else if (sig_ref.has_probe()) {
							res = hybrid(dbase, bio_seq, sig_ref, melt, opt->probe_strand, opt->min_probe_tm,
							             opt->max_probe_tm, opt->min_probe_dg, opt->max_probe_dg, opt->probe_clamp_5,
							             opt->probe_clamp_3, opt->max_gap, opt->max_mismatch, opt->max_poly_degen,
							             opt->target_strand, index_table, local_str_table);
						}
						for (list<hybrid_sig>::iterator i = res.begin(); i != res.end(); ++i) {
							i->seq_id(int(first_seq_index + t));
							local.push_back(make_pair(a, *i));
						}
					}
				}
				delete[] bio_seq.second;
				} catch (const char *e) {
#pragma omp critical
					{
						g_err = e;
						rc = -1;
					}
				}
			}

			const vector<string> local_index = ordered_keys(local_str_table);

#pragma omp critical
			{
				for (size_t k = 0; k < local.size(); k++) {
					const hybrid_sig &h = local[k].second;
					const tnt_assay &A = assays[local[k].first];
					tnt_hit o;
					memset(&o, 0, sizeof(o));
					o.assay_id = h.id;
					o.degen_id = h.degen_id;
					o.seq_index = h.seq_index;
					o.amplicon_start = h.amplicon_range.first;
					o.amplicon_stop = h.amplicon_range.second;
					o.probe_start = h.probe_range.first;
					o.probe_stop = h.probe_range.second;
					o.forward_tm = h.forward_tm;
					o.reverse_tm = h.reverse_tm;
					o.probe_tm = h.probe_tm;
					o.forward_dH = h.forward_dH;
					o.forward_dS = h.forward_dS;
					o.reverse_dH = h.reverse_dH;
					o.reverse_dS = h.reverse_dS;
					o.probe_dH = h.probe_dH;
					o.probe_dS = h.probe_dS;
					o.primer_strand = h.primer_strand;
					o.probe_strand = h.probe_strand;
					o.forward_primer_clamp = h.forward_primer_clamp;
					o.reverse_primer_clamp = h.reverse_primer_clamp;
					o.forward_mm = h.forward_mm;
					o.forward_gap = h.forward_gap;
					o.reverse_mm = h.reverse_mm;
					o.reverse_gap = h.reverse_gap;
					o.probe_mm = h.probe_mm;
					o.probe_gap = h.probe_gap;
					o.has_primers = h.has_primers() ? 1 : 0;
					o.has_probe = h.has_probe() ? 1 : 0;
					o.forward_oligo = 0;
					o.reverse_oligo = 1;
					if (o.has_primers) {
						const string f = index_to_str(h.forward_oligo_str_index, local_index);
						const string r = index_to_str(h.reverse_oligo_str_index, local_index);
						o.forward_oligo = (A.forward && f == A.forward) ? 0 : 1;
						o.reverse_oligo = (A.reverse && r == A.reverse) ? 1 : 0;
						copy_text(o.forward_align, inflate_dna_seq(index_to_str(h.forward_align_str_index, local_index)));
						copy_text(o.reverse_align, inflate_dna_seq(index_to_str(h.reverse_align_str_index, local_index)));
					}
					if (o.has_probe)
						copy_text(o.probe_align, inflate_dna_seq(index_to_str(h.probe_align_str_index, local_index)));
					all.push_back(o);
				}
			}
		} catch (const char *e) {
#pragma omp critical
			{
				g_err = e;
				rc = -1;
			}
		}
	}

	*n_out = all.size();
	*out = (tnt_hit *)malloc(sizeof(tnt_hit) * max<size_t>(1, all.size()));
	memcpy(*out, all.data(), sizeof(tnt_hit) * all.size());
	return rc;
}

// a17: hairpin / homodimer / heterodimer Tm of the assay oligos, tntblast_local.cpp:656-686
int tntref_oligo_structure(const tnt_options *opt, const char *oligo, float ct, float *hairpin_tm, float *homodimer_tm)
{
	try {
		NucCruc melt(NucCruc::SANTA_LUCIA, opt->target_t);
		setup_melt(melt, *opt);
		melt.set_duplex(oligo);
		melt.strand(ct, ct);
		*hairpin_tm = melt.approximate_tm_hairpin();
		*homodimer_tm = melt.approximate_tm_homodimer();
		return 0;
	} catch (const char *e) {
		g_err = e;
		return -1;
	}
}

int tntref_heterodimer(const tnt_options *opt, const char *a, const char *b, float ct_a, float ct_b, float *tm)
{
	try {
		NucCruc melt(NucCruc::SANTA_LUCIA, opt->target_t);
		setup_melt(melt, *opt);
		melt.set_query(a);
		melt.set_target(b);
		melt.strand(ct_a, ct_b);
		*tm = melt.approximate_tm_heterodimer();
		return 0;
	} catch (const char *e) {
		g_err = e;
		return -1;
	}
}

// degenerate expansion, degenerate_na.cpp:44-215
int tntref_expand(const char *oligo, char *buf, size_t cap, int *n_out, int *degeneracy_out)
{
	try {
		list<string> l = expand_nucleic_acid(string(oligo));
		string joined;
		for (list<string>::const_iterator i = l.begin(); i != l.end(); ++i) {
			joined += *i;
			joined += '\n';
		}
		if (joined.size() + 1 > cap) {
			g_err = "buffer too small";
			return -1;
		}
		memcpy(buf, joined.c_str(), joined.size() + 1);
		*n_out = int(l.size());
		*degeneracy_out = int(degeneracy(string(oligo)));
		return 0;
	} catch (const char *e) {
		g_err = e;
		return -1;
	}
}

// FASTA ingest through the reference's own reader: sequence_data::open -> load_fasta (sequence_data_fastx.cpp:13-79),
// read_bio_seq -> read_bio_seq_fasta_slow (:187-382).  Outputs are malloc'ed: 1-byte codes back to back,
// n + 1 offsets, deflines joined by '\n'.
int tntref_read_fasta(const char *path, uint8_t **codes, uint64_t **offsets, size_t *n_targets, char **deflines)
{
	try {
		sequence_data sd;
		sd.open(path, vector<string>(), vector<string>());
		const size_t n = sd.size();
		vector<uint8_t> all;
		vector<uint64_t> off(1, 0);
		string defs;
		for (size_t i = 0; i < n; i++) {
			pair<string, SEQPTR> seq;
			sd.read_bio_seq(seq, (unsigned int)i);
			const unsigned int len = SEQ_SIZE(seq.second);
			const SEQPTR p = SEQ_START(seq.second);
			all.insert(all.end(), (const uint8_t *)p, (const uint8_t *)p + len);
			off.push_back(all.size());
			defs += seq.first;
			defs += '\n';
			delete[] seq.second;
		}
		*codes = (uint8_t *)malloc(all.size() + 1);
		memcpy(*codes, all.data(), all.size());
		*offsets = (uint64_t *)malloc(off.size() * sizeof(uint64_t));
		memcpy(*offsets, off.data(), off.size() * sizeof(uint64_t));
		*deflines = (char *)malloc(defs.size() + 1);
		memcpy(*deflines, defs.c_str(), defs.size() + 1);
		*n_targets = n;
		return 0;
	} catch (const char *e) {
		g_err = e;
		return -1;
	}
}

void tntref_free(void *p) { free(p); }

} // extern "C"
