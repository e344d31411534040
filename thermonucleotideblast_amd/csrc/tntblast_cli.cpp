// tntblast_cli.cpp -- command-line front end over the C ABI: the reference's flags, its
// assay-file format and its "-m 0" output file, with the per-target inner loop running on the MI355X.
//
// What is restated here is host-side only (file:line in the reference):
//   option names and defaults            options.cpp:77-106,112-470, options.h:85-170, tntblast.h:19-76
//   threshold rules                      options.cpp:833-916
//   assay file                           input.cpp:43-168 (white-space columns, '#' comments, 2 / 3 / 4 columns)
//   degenerate expansion                 degenerate_na.cpp:44-215, tntblast_util.cpp:579-713
//   per-hit hairpin / dimer Tm           tntblast_local.cpp:656-686 (through tnt_oligo_structure)
//   result order                         hybrid_sig::operator<, hybrid_sig.h:328-357 (std::list::sort, stable)
//   output records                       tntblast_local.cpp:910-1280, write_alignment tntblast_util.cpp:1313-1335,
//                                        gc_content :1274-1295, primer_heuristics :917-925 + primer.cpp:44-328
//   primer / probe masking -k -K -r      mask_binding_sites, tntblast_util.cpp:97-576 (NucCruc evaluations = tnt_bind_batch)
//   --max-target-len fragmentation       tntblast_local.cpp:282-289,438-468,638-648, seq_len_increment sequence_data.cpp:739-754,
//                                        uniquify_results tntblast_util.cpp:1555-1755
// Between reading the inputs and formatting the hits there is one call per batch of targets: tnt_search().
// Not supported (exit with a message): BLAST / GenBank / EMBL databases, -D.
#include <getopt.h>

#include <algorithm>
#include <cctype>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <iterator>
#include <list>
#include <map>
#include <set>
#include <sstream>
#include <string>
#include <thread>
#include <vector>

#include "../../include/tnt_mi355x.h"

namespace {

struct Fail {
	std::string msg;
};
[[noreturn]] void fail(const std::string &m) { throw Fail{m}; }

bool parse_bool(const char *s) // options.cpp parse_bool: T / F (any case), TRUE / FALSE
{
	std::string v(s);
	for (char &c : v) c = char(toupper(c));
	if (v == "T" || v == "TRUE") return true;
	if (v == "F" || v == "FALSE") return false;
	fail("Unable to parse a boolean option (expected T or F)");
}

struct Options {
	std::string input, output, dbase, local_dbase;
	tnt_options t;
	bool have_primer_bounds = false, have_probe_bounds = false;
	bool show_alignments = true, show_sequence = true, append_name = false, ignore_probe = false, rescale_ct = true;
	bool verbose = true;
	// -m (options.cpp:672-708): exactly one of the five file formats; -a / -M add to the standard one
	enum { STANDARD = 1, FASTA = 2, NETWORK = 4, INVERSE_TARGET = 8, INVERSE_QUERY = 16 };
	int format = STANDARD;
	bool per_query = false;  // -n: one output file per query
	bool summary = false;    // -S: assay summary on stdout
	bool best_match = false; // --best-match
	bool multiplex = false;  // --plex
	bool dump_query = false; // --dump-query
	long max_target_len = 500000; // --max-target-len, DEFAULT_FRAGMENT_TARGET_LENGTH (tntblast.h:61)
	enum { MASK_PRIMERS = 1, MASK_PROBE = 2, REPLACE_PRIMERS = 4 }; // tntblast.h:115-117
	int mask = 0;                 // -k / -K / -r
	int gpus = 0;                 // --gpus: devices the database is dealt to (0: every visible device)
};

constexpr size_t INVALID_INDEX = ~size_t(0); // hybrid_sig.h:25
struct Assay { // one line of the assay file
	std::string name, forward, reverse, probe;
	int id;
	// position of the oligo strings in the reference's string table (first use, hybrid_sig.h:555-570): --plex sorts on them
	size_t fi = INVALID_INDEX, ri = INVALID_INDEX, pi = INVALID_INDEX;
};

struct StringTable { // str_to_index
	std::map<std::string, size_t> idx;
	size_t operator()(const std::string &s)
	{
		auto it = idx.find(s);
		if (it != idx.end()) return it->second;
		const size_t v = idx.size();
		idx[s] = v;
		return v;
	}
};

struct Expanded { // one non-degenerate assay
	int parent; // index into the assay list (== hybrid_sig::my_id)
	std::string forward, reverse, probe;
	int forward_degen = 1, reverse_degen = 1, probe_degen = 1;
};

// expand_nucleic_acid, degenerate_na.cpp:69-173: odometer over the IUPAC members, first position fastest
std::list<std::string> expand_oligo(const std::string &seq)
{
	std::vector<std::string> members(seq.size());
	size_t count = 1;
	for (size_t i = 0; i < seq.size(); i++) {
		const char *m = nullptr;
		switch (toupper(seq[i])) {
		case 'A': m = "A"; break;
		case 'T': m = "T"; break;
		case 'G': m = "G"; break;
		case 'C': m = "C"; break;
		case 'I': m = "I"; break;
		case 'M': m = "AC"; break;
		case 'R': m = "GA"; break;
		case 'S': m = "GC"; break;
		case 'V': m = "GCA"; break;
		case 'W': m = "AT"; break;
		case 'Y': m = "TC"; break;
		case 'H': m = "ACT"; break;
		case 'K': m = "GT"; break;
		case 'D': m = "GAT"; break;
		case 'B': m = "GTC"; break;
		case 'N': m = "ATGC"; break;
		default: std::cerr << "Found the unknown letter \"" << seq[i] << "\"" << std::endl; fail("Unknown base!");
		}
		members[i] = m;
		count *= members[i].size();
	}
	std::list<std::string> out;
	std::vector<size_t> at(seq.size(), 0);
	for (size_t n = 0; n < count; n++) {
		std::string s(seq.size(), 'A');
		for (size_t j = 0; j < seq.size(); j++) s[j] = members[j][at[j]];
		out.push_back(s);
		for (size_t j = 0; j < seq.size(); j++) {
			if (++at[j] == members[j].size()) at[j] = 0;
			else break;
		}
	}
	return out;
}

int degeneracy(const std::string &oligo) // degenerate_na.cpp:175-215
{
	int d = 1;
	for (char c : oligo) switch (toupper(c)) {
		case 'A': case 'T': case 'G': case 'C': case 'I': break;
		case 'M': case 'R': case 'S': case 'W': case 'Y': case 'K': d *= 2; break;
		case 'V': case 'H': case 'D': case 'B': d *= 3; break;
		case 'N': d *= 4; break;
		default: std::cerr << "Found the unknown letter '" << c << "'" << std::endl; fail("Unknown base!");
		}
	return d;
}

// input.cpp:43-168.  `force_probe` (-A PROBE): primer pairs become single probes named <name>_F / _R / _P.  The string
// indices follow the reference as built by g++, which evaluates the constructor arguments right to left: probe,
// reverse, forward, name.
std::vector<Assay> read_input_file(const std::string &path, bool ignore_probe, bool force_probe, StringTable &st)
{
	std::ifstream fin(path.c_str());
	if (!fin) {
		std::cerr << "Unable to open " << path << " for reading" << std::endl;
		fail("read_input_file: I/O Error");
	}
	std::vector<Assay> out;
	auto probe_only = [&](const std::string &name, const std::string &oligo) {
		Assay a;
		a.id = int(out.size());
		a.probe = oligo, a.pi = st(oligo);
		a.name = name;
		(void)st(name);
		out.push_back(a);
	};
	std::string line;
	while (std::getline(fin, line)) {
		const std::string::size_type pos = line.find('#');
		if (pos != std::string::npos) line = line.substr(0, pos);
		std::stringstream ss(line);
		std::vector<std::string> col;
		std::string entry;
		while (ss >> entry) col.push_back(entry);
		if (col.empty()) continue;
		switch (col.size()) {
		case 2:
			if (ignore_probe) fail("ignore_probe is true but only probes have been provided!");
			probe_only(col[0], col[1]);
			break;
		case 3:
		case 4: {
			const bool with_probe = col.size() == 4 && !ignore_probe;
			if (force_probe) {
				probe_only(col[0] + "_F", col[1]);
				probe_only(col[0] + "_R", col[2]);
				if (with_probe) probe_only(col[0] + "_P", col[3]);
				break;
			}
			Assay a;
			a.id = int(out.size());
			if (with_probe) a.probe = col[3], a.pi = st(col[3]);
			a.reverse = col[2], a.ri = st(col[2]);
			a.forward = col[1], a.fi = st(col[1]);
			a.name = col[0];
			(void)st(col[0]);
			out.push_back(a);
			break;
		}
		default: std::cerr << "Error reading " << path << std::endl; fail("Error: Invalid number of arguments input file");
		}
	}
	return out;
}

// multiplex_expansion, tntblast_util.cpp:715-915 (--plex T): every combination of the primers (and probes) of the input
// assays.  Redundant combinations are removed with std::sort + std::unique on the string-table indices, exactly as the
// reference does: std::sort is not stable, so which of several equal assays survives (its name) is whatever libstdc++'s
// introsort leaves first -- the same here, because the same algorithm sees the same sequence and comparisons.
std::vector<Assay> multiplex_expansion(const std::vector<Assay> &sig, int assay_format, StringTable &st)
{
	std::vector<Assay> ret;
	auto make = [&](const std::string &name, const Assay &f_from, bool f_is_forward, const Assay &r_from, bool r_is_forward) {
		Assay a;
		a.name = name;
		(void)st(name);
		a.forward = f_is_forward ? f_from.forward : f_from.reverse, a.fi = f_is_forward ? f_from.fi : f_from.ri;
		a.reverse = r_is_forward ? r_from.forward : r_from.reverse, a.ri = r_is_forward ? r_from.fi : r_from.ri;
		a.id = 0;
		return a;
	};
	if (assay_format == TNT_ASSAY_PADLOCK || assay_format == TNT_ASSAY_MIPS) {
		for (const Assay &i : sig)
			for (const Assay &j : sig) ret.push_back(make(&i == &j ? i.name : i.name + "(5')/" + j.name + "(3')", i, true, j, false));
	}
	if (assay_format == TNT_ASSAY_PCR) {
		bool has_probes = false;
		for (const Assay &i : sig) {
			if (i.fi == INVALID_INDEX) continue; // a probe-only assay is dropped (:765-769 builds it and moves on)
			if (i.pi != INVALID_INDEX) has_probes = true;
			for (const Assay &j : sig) {
				if (i.fi == j.ri) continue;
				ret.push_back(make(i.name + "(F)/" + j.name + "(R)", i, true, j, false));
			}
		}
		for (const Assay &i : sig) {
			if (i.fi == INVALID_INDEX) continue;
			for (const Assay &j : sig) {
				if (i.fi == j.fi) continue;
				ret.push_back(make(i.name + "(F)/" + j.name + "(F)", i, true, j, true));
			}
		}
		for (const Assay &i : sig) {
			if (i.fi == INVALID_INDEX) continue;
			for (const Assay &j : sig) {
				if (i.ri == j.ri) continue;
				ret.push_back(make(i.name + "(R)/" + j.name + "(R)", i, false, j, false));
			}
		}
		if (has_probes) {
			std::vector<Assay> with_probe;
			for (const Assay &i : ret)
				for (const Assay &j : sig) {
					if (j.pi == INVALID_INDEX) continue;
					Assay a = i;
					a.name = i.name + "+" + j.name + "(P)";
					(void)st(a.name);
					a.probe = j.probe, a.pi = j.pi;
					with_probe.push_back(a);
				}
			ret.swap(with_probe);
		}
	}
	if (assay_format == TNT_ASSAY_PROBE) return sig;
	auto ordered = [](const Assay &a, size_t &F, size_t &R) {
		F = a.fi, R = a.ri;
		if (F < R) std::swap(F, R);
	};
	std::sort(ret.begin(), ret.end(), [&](const Assay &a, const Assay &b) { // sort_by_seq, tntblast_util.cpp:31-62
		size_t Fa, Ra, Fb, Rb;
		ordered(a, Fa, Ra), ordered(b, Fb, Rb);
		if (Fa == Fb) {
			if (Ra == Rb) return a.pi < b.pi;
			return Ra < Rb;
		}
		return Fa < Fb;
	});
	ret.erase(std::unique(ret.begin(), ret.end(), [&](const Assay &a, const Assay &b) { // compare_by_seq, :64-91
		size_t Fa, Ra, Fb, Rb;
		ordered(a, Fa, Ra), ordered(b, Fb, Rb);
		return Fa == Fb && Ra == Rb && a.pi == b.pi;
	}), ret.end());
	int id = 0;
	for (Assay &a : ret) a.id = id++;
	std::cerr << "Multiplexing has created " << ret.size() << " assays from " << sig.size() << " input assays" << std::endl;
	return ret;
}

std::vector<Expanded> expand_assays(const std::vector<Assay> &assays, bool rescale_ct) // tntblast_util.cpp:579-713
{
	std::vector<Expanded> out;
	for (const Assay &a : assays) {
		std::list<std::pair<std::string, std::string>> primers;
		if (!a.forward.empty() && !a.reverse.empty())
			for (const std::string &f : expand_oligo(a.forward))
				for (const std::string &r : expand_oligo(a.reverse)) primers.push_back(std::make_pair(f, r));
		std::list<std::string> probes;
		if (!a.probe.empty()) probes = expand_oligo(a.probe);
		const size_t n = primers.size() * probes.size();
		if (n > 1) std::cout << "Expanded degenerate bases in " << a.name << " to make " << n << " non-degenerate assays" << std::endl;
		Expanded e;
		e.parent = a.id;
		e.forward_degen = (rescale_ct && !a.forward.empty()) ? degeneracy(a.forward) : 1;
		e.reverse_degen = (rescale_ct && !a.reverse.empty()) ? degeneracy(a.reverse) : 1;
		e.probe_degen = (rescale_ct && !a.probe.empty()) ? degeneracy(a.probe) : 1;
		if (primers.empty()) {
			for (const std::string &p : probes) {
				e.probe = p;
				out.push_back(e);
			}
		} else {
			for (const auto &pr : primers) {
				e.forward = pr.first, e.reverse = pr.second;
				if (probes.empty()) {
					e.probe.clear();
					out.push_back(e);
				} else {
					for (const std::string &p : probes) {
						e.probe = p;
						out.push_back(e);
					}
				}
			}
		}
	}
	return out;
}

float gc_content(const std::string &s) // tntblast_util.cpp:1274-1295
{
	float gc = 0.0f;
	for (char c : s)
		if (c == 'G' || c == 'g' || c == 'C' || c == 'c') gc++;
	return gc / s.size();
}

// primer_heuristics (tntblast_util.cpp:917-925): PCRPrimer(POLY_3_GC | MULTI_5_GC | NO_POLY_RUNS | NO_3_T), run 5,
// verbose reporting (primer.cpp:44-328, error text :406-470)
std::string primer_heuristics(const std::string &p)
{
	enum { POLY_3_GC = 1, MULTI_5_GC = 2, NO_POLY_RUNS = 4, NO_3_T = 8 };
	for (char c : p)
		if (!strchr("ATGCatgc", c)) return "BAD_BASE";
	int status = 0;
	if (toupper(p.back()) == 'T') status |= NO_3_T;
	unsigned tc = 0, ag = 0, g = 0, cc = 0, max_tc = 0, max_ag = 0, n_gc = 0, index = 0;
	// The reference packs the primer into a CircleBuffer of MAX_PCR_LENGTH = 128 slots (primer.h:92, primer.cpp:330-370) whose
	// push_back wraps without moving the head (circle_buffer.h:106-111): of an oligo of 128 bases or more the checks see the
	// last `length mod 128` bases only.  Reproduced, so that the report of a long primer reads as the reference's does.
	const std::string seen = p.size() >= 128 && p.size() % 128 ? p.substr(p.size() - p.size() % 128) : p;
	for (char ch : seen) {
		switch (toupper(ch)) {
		case 'A': ag++, tc = 0, g = 0, cc = 0; break;
		case 'T': tc++, ag = 0, g = 0, cc = 0; break;
		case 'G': ag++, g++, tc = 0, cc = 0, n_gc++; break;
		case 'C': tc++, cc++, ag = 0, g = 0, n_gc++; break;
		}
		max_tc = std::max(max_tc, tc);
		max_ag = std::max(max_ag, ag);
		if (index == 4) {
			if (max_tc >= 2 ? n_gc > 2 : n_gc > 3) status |= MULTI_5_GC;
		}
		index++;
	}
	if (max_tc >= 5 || max_ag >= 5) status |= NO_POLY_RUNS;
	if (cc >= 3 || g >= 3) status |= POLY_3_GC;
	if (!status) return "PCR_VALID";
	std::string out;
	auto add = [&](int bit, const char *name) {
		if (!(status & bit)) return;
		if (!out.empty()) out += ", ";
		out += name;
	};
	add(POLY_3_GC, "POLY_3_GC");
	add(MULTI_5_GC, "MULTI_5_GC");
	add(NO_POLY_RUNS, "NO_POLY_RUNS");
	add(NO_3_T, "NO_3_T");
	return out;
}

void write_alignment(std::ostream &out, const std::string &prefix, const char *alignment) // tntblast_util.cpp:1313-1335
{
	if (!alignment[0]) return;
	out << prefix;
	for (const char *c = alignment; *c; c++) {
		out << *c;
		if (*c == '\n') out << prefix;
	}
	out << std::endl;
}

struct Record { // a hit with what the writer needs beside the tnt_hit payload
	tnt_hit h;
	size_t order; // position in the device output (ties keep it: std::list::sort is stable)
	std::string amplicon, defline; // taken while the batch of targets that produced the hit is still resident
	std::string long_align[3];     // alignments that do not fit the hit's 448-byte fields (oligos beyond 64 bases), read through tnt_long_text then too
	const char *align(int which) const // 0 forward, 1 reverse, 2 probe
	{
		if (!long_align[which].empty()) return long_align[which].c_str();
		return which == 0 ? h.forward_align : (which == 1 ? h.reverse_align : h.probe_align);
	}
	float min_tm() const { return std::min(h.forward_tm, h.reverse_tm); }
	float max_tm() const { return std::max(h.forward_tm, h.reverse_tm); }
};

// an alignment field that only refers to its text (0x7F ...): fetch the text while the context still holds it
void resolve_long_alignments(tnt_ctx *ctx, Record &rec)
{
	const char *field[3] = {rec.h.forward_align, rec.h.reverse_align, rec.h.probe_align};
	for (int k = 0; k < 3; k++) {
		if (static_cast<unsigned char>(field[k][0]) != 0x7Fu) continue;
		std::vector<char> buf(1 << 14);
		const int n = tnt_long_text(ctx, field[k], buf.data(), buf.size());
		if (n < 0) fail("tnt_long_text failed");
		rec.long_align[k].assign(buf.data(), size_t(n));
	}
}

bool record_less(const Record &a, const Record &b) // hybrid_sig::operator< for equal ids, hybrid_sig.h:328-357
{
	if (a.min_tm() != b.min_tm()) return a.min_tm() > b.min_tm();
	if (a.h.probe_tm != b.h.probe_tm) return a.h.probe_tm > b.h.probe_tm;
	if (a.max_tm() != b.max_tm()) return a.max_tm() > b.max_tm();
	return a.h.seq_index < b.h.seq_index;
}

// select_best_match, tntblast_util.cpp:1482-1546: of the matches of one assay on one target, keep one.  The list
// is first sorted (stably) by target; every later match of the same target then either dies or replaces the survivor.
void select_best_match(std::vector<Record> &m)
{
	if (m.empty()) return;
	std::stable_sort(m.begin(), m.end(), [](const Record &a, const Record &b) { return a.h.seq_index < b.h.seq_index; });
	auto min_primer_tm = [](const Record &r) { return std::max(0.0f, r.min_tm()); }; // hybrid_sig.h:369-373
	std::vector<Record> keep;
	size_t best = 0;
	std::vector<char> dead(m.size(), 0);
	for (size_t curr = 1; curr < m.size(); curr++) {
		if (m[curr].h.seq_index != m[best].h.seq_index) {
			best = curr;
			continue;
		}
		bool delete_curr = false;
		if (m[curr].h.has_primers) {
			if (min_primer_tm(m[curr]) == min_primer_tm(m[best])) {
				if (m[curr].h.probe_tm < m[best].h.probe_tm) delete_curr = true;
				if (m[curr].max_tm() < m[best].max_tm()) delete_curr = true;
			} else if (min_primer_tm(m[curr]) < min_primer_tm(m[best])) {
				delete_curr = true;
			}
		} else if (m[curr].h.probe_tm < m[best].h.probe_tm) {
			delete_curr = true;
		}
		if (delete_curr) {
			dead[curr] = 1;
		} else {
			dead[best] = 1;
			best = curr;
		}
	}
	for (size_t i = 0; i < m.size(); i++)
		if (!dead[i]) keep.push_back(m[i]);
	m.swap(keep);
}


// seq_len_increment, sequence_data.cpp:739-754: length of the pieces a target of (approximate) length `len` is cut into
std::pair<unsigned int, unsigned int> seq_len_increment(unsigned int len, unsigned int max_len)
{
	if (len <= max_len) return std::make_pair(len - 1, 1u);
	unsigned int n = 2;
	while (len > n * max_len) n++;
	return std::make_pair(len / n + ((len % n != 0) ? 1u : 0u), n);
}

// top_strand, tntblast_util.cpp:1758-1780
std::string top_strand(const std::string &align)
{
	const std::string::size_type start = align.find("5' ");
	if (start == std::string::npos) fail("top_strand: Unable to parse alignment");
	const std::string::size_type stop = align.find(" 3'");
	if (stop == std::string::npos) fail("top_strand: Unable to parse alignment");
	return align.substr(start + 3, stop - (start + 3));
}

// uniquify_results, tntblast_util.cpp:1555-1755: fragments of a long target overlap, so a match can be found twice
// (sometimes once complete and once clipped); of matches that contain one another the larger survives.
// `oligo_len(r, slot)` = length of the oligo in the forward (0) / reverse (1) slot of record r.
template <class LEN> void uniquify_results(std::vector<Record> &m, LEN oligo_len)
{
	if (m.size() < 2) return;
	auto range = [](const Record &r) {
		return r.h.has_primers ? std::make_pair(r.h.amplicon_start, r.h.amplicon_stop) : std::make_pair(r.h.probe_start, r.h.probe_stop);
	};
	std::stable_sort(m.begin(), m.end(), [&](const Record &a, const Record &b) { // sort_by_loc, hybrid_sig.h:468-497 (equal ids here)
		if (a.h.seq_index != b.h.seq_index) return a.h.seq_index < b.h.seq_index;
		return range(a) < range(b);
	});
	std::vector<char> keep(m.size(), 0);
	size_t start = 0;
	while (start < m.size()) {
		size_t stop = start;
		while (stop < m.size() && m[stop].h.degen_id == m[start].h.degen_id && m[stop].h.seq_index == m[start].h.seq_index) stop++;
		std::list<size_t> valid;
		for (size_t it = start; it < stop; it++) {
			if (valid.empty()) {
				valid.push_back(it);
				continue;
			}
			const Record &A = m[it];
			const int fl = oligo_len(A, 0) / 2, rl = oligo_len(A, 1) / 2;
			int status = 0; // 0 NO_MATCH, 1 A_CONTAINS_B, 2 B_CONTAINS_A
			for (auto v = valid.begin(); v != valid.end(); ++v) {
				const Record &B = m[*v];
				int same = 0;
				if (A.h.has_primers) {
					const bool primers_overlap = std::abs(A.h.amplicon_start - B.h.amplicon_start) < fl && std::abs(A.h.amplicon_stop - B.h.amplicon_stop) < rl;
					if (primers_overlap) {
						if (A.h.amplicon_start <= B.h.amplicon_start && A.h.amplicon_stop >= B.h.amplicon_stop &&
						    top_strand(A.align(0)).find(top_strand(B.align(0))) != std::string::npos &&
						    top_strand(A.align(1)).find(top_strand(B.align(1))) != std::string::npos)
							same = 1;
						else if (B.h.amplicon_start <= A.h.amplicon_start && B.h.amplicon_stop >= A.h.amplicon_stop &&
						         top_strand(B.align(0)).find(top_strand(A.align(0))) != std::string::npos &&
						         top_strand(B.align(1)).find(top_strand(A.align(1))) != std::string::npos)
							same = 2;
						if (A.h.has_probe && B.h.has_probe && (A.h.probe_start != B.h.probe_start || A.h.probe_stop != B.h.probe_stop)) same = 0;
					}
				} else {
					if (A.h.probe_start <= B.h.probe_start && A.h.probe_stop >= B.h.probe_stop &&
					    top_strand(A.align(2)).find(top_strand(B.align(2))) != std::string::npos)
						same = 1;
					else if (B.h.probe_start <= A.h.probe_start && B.h.probe_stop >= A.h.probe_stop &&
					         top_strand(B.align(2)).find(top_strand(A.align(2))) != std::string::npos)
						same = 2;
				}
				if (same == 0) continue;
				if (same == 1) {
					*v = it; // overwrite B with A and keep searching
					status = 1;
				} else {
					status = 2; // A is not valid
					break;
				}
			}
			if (status == 0) valid.push_front(it); // valid.splice(valid.begin(), valid_update)
		}
		for (size_t v : valid) keep[v] = 1;
		start = stop;
	}
	std::vector<Record> out;
	for (size_t i = 0; i < m.size(); i++)
		if (keep[i]) out.push_back(m[i]);
	m.swap(out);
}

// ---------------------------------------------------------------------------------------------------------------------
// mask_binding_sites, tntblast_util.cpp:97-576 (called from amplicon(), amplicon_search.cpp:539,663): the primer binding
// sites of an amplicon are lower-cased (-k) or replaced by the primers themselves (-r), the probe's sites lower-cased
// (-K).  Where a site lies is decided by NucCruc on pieces of the amplicon text; those evaluations are bind requests
// (tnt_bind_batch, the unit-level kernel instance) against a scratch database whose targets are the pieces.
struct MaskJob {
	size_t record;     // index into the batch's records
	int kind;          // 0 forward primer (5' end, minus strand), 1 reverse primer (3' end, plus), 2 probe plus, 3 probe minus
	size_t pos;        // probe: position i of the sliding window
	unsigned int gap_offset, target_len;
};

// the base NucCruc is given for a character of the amplicon: `complement` for push_front(complement) loads; the "as is"
// loads of mask_primer_3 / mask_probe pass degenerate letters through the SAME table as the complementing ones
// (tntblast_util.cpp:300-331,409-441), so for them a degenerate letter stands for its complement
int mask_code(char ch, bool as_is)
{
	static const char *letters = "ACGTIMRSVWYHKDBN"; // seq.h:12-35 codes 0..15
	static const char *degen_swap = "ACGTIKYSBWRDMHVN";
	const char u = char(toupper(static_cast<unsigned char>(ch)));
	const char *p = strchr(letters, u);
	if (!p || !u) return -1;
	int code = int(p - letters);
	if (as_is) code = int(strchr(letters, degen_swap[code]) - letters);
	return code;
}

char base_complement_char(char c) // seq.h / tntblast_util: complement of an oligo letter, upper case
{
	static const char *from = "ACGTIMRSVWYHKDBN", *to = "TGCAIKYSBWRDMHVN";
	const char *p = strchr(from, char(toupper(static_cast<unsigned char>(c))));
	return p ? to[p - from] : c;
}

template <class OLIGO> void apply_masks_round(tnt_ctx *ctx, const Options &opt, std::vector<Record> &recs, size_t first, OLIGO oligo_of, bool probe_round);

// the probe's windows are cut from the amplicon as the primer pass left it (-r changes its text and its length)
template <class OLIGO> void apply_masks(tnt_ctx *ctx, const Options &opt, std::vector<Record> &recs, size_t first, OLIGO oligo_of)
{
	if (!opt.mask) return;
	apply_masks_round(ctx, opt, recs, first, oligo_of, false);
	apply_masks_round(ctx, opt, recs, first, oligo_of, true);
}

template <class OLIGO> void apply_masks_round(tnt_ctx *ctx, const Options &opt, std::vector<Record> &recs, size_t first, OLIGO oligo_of, bool probe_round)
{
	std::map<std::string, int> oligo_index;
	for (int i = 0; i < tnt_ctx_num_oligos(ctx); i++) oligo_index[tnt_ctx_oligo(ctx, i)] = i;
	std::vector<uint8_t> codes;
	std::vector<uint64_t> offs(1, 0);
	std::vector<tnt_bind_req> reqs;
	std::vector<MaskJob> jobs;
	auto add = [&](const std::vector<int> &piece, const std::string &oligo, bool minus, float ct, MaskJob job) {
		job.target_len = (unsigned int)piece.size();
		for (int c : piece) codes.push_back(uint8_t(c));
		offs.push_back(codes.size());
		tnt_bind_req r;
		r.seq_index = int(offs.size() - 2);
		r.oligo = oligo_index.at(oligo);
		r.strand = minus ? TNT_MINUS : TNT_PLUS;
		r.query_loc = 0, r.target_loc = 0; // the window is the whole piece (it is never longer than the oligo + 2)
		r.role = job.kind >= 2 ? 1 : 0;
		r.ct = ct;
		reqs.push_back(r);
		jobs.push_back(job);
	};
	const float fwd_ct = opt.t.asymmetric_strand_ratio * opt.t.primer_strand, rev_ct = opt.t.primer_strand;
	for (size_t ri = first; ri < recs.size(); ri++) {
		const Record &R = recs[ri];
		if (!R.h.has_primers) continue; // only amplicon() masks
		const std::string &amp = R.amplicon;
		const std::string fo = oligo_of(R, 0), ro = oligo_of(R, 1), po = oligo_of(R, 2);
		if (!probe_round && (opt.mask & (Options::MASK_PRIMERS | Options::REPLACE_PRIMERS))) {
			{ // mask_primer_5, :141-243: the first |oligo| characters, complemented and reversed, until a bad character
				std::vector<int> piece;
				unsigned int gaps = 0;
				for (size_t i = 0; i < fo.size() && i < amp.size(); i++) {
					if (amp[i] == '-') {
						gaps++;
						continue;
					}
					const int c = mask_code(amp[i], false);
					if (c < 0) break;
					piece.push_back(c);
				}
				add(piece, fo, true, fwd_ct, MaskJob{ri, 0, 0, gaps, 0});
			}
			{ // mask_primer_3, :245-384: the last |oligo| characters as they are; a bad character restarts the piece
				std::vector<int> piece;
				unsigned int gaps = 0;
				for (size_t i = amp.size() >= ro.size() ? amp.size() - ro.size() : 0; i < amp.size(); i++) {
					if (amp[i] == '-') {
						gaps++;
						continue;
					}
					const int c = mask_code(amp[i], true);
					if (c < 0) {
						piece.clear();
						continue;
					}
					piece.push_back(c);
				}
				add(piece, ro, false, rev_ct, MaskJob{ri, 1, 0, gaps, 0});
			}
		}
		if (probe_round && (opt.mask & Options::MASK_PROBE) && R.h.has_probe) { // mask_probe, :387-576: a window of |oligo| + 2 slides over both strands
			for (int minus = 0; minus < 2; minus++) {
				std::vector<int> window; // oldest base first, in amplicon order
				for (size_t i = 0; i < amp.size(); i++) {
					const int c = mask_code(amp[i], minus == 0);
					if (c < 0) window.clear();
					else window.push_back(c);
					add(window, po, minus != 0, opt.t.probe_strand, MaskJob{ri, 2 + minus, i, 0, 0});
					if (window.size() == po.size() + 2) window.erase(window.begin());
				}
			}
		}
	}
	if (reqs.empty()) return;
	tnt_db *mdb = nullptr;
	if (tnt_db_create(ctx, codes.data(), offs.data(), offs.size() - 1, 0, &mdb) != TNT_OK) fail(tnt_last_error(ctx));
	std::vector<tnt_bind> out(reqs.size());
	if (tnt_bind_batch(ctx, mdb, reqs.data(), reqs.size(), out.data()) != TNT_OK) fail(tnt_last_error(ctx));
	tnt_db_destroy(mdb);
	const bool do_mask = (opt.mask & Options::MASK_PRIMERS) != 0, do_replace = (opt.mask & Options::REPLACE_PRIMERS) != 0;
	// the primers first (5', then 3'), then the probe passes, per record in the order the reference applies them
	for (size_t k = 0; k < jobs.size(); k++) {
		const MaskJob &J = jobs[k];
		const tnt_bind &b = out[k];
		std::string &amp = recs[J.record].amplicon;
		const Record &R = recs[J.record];
		if (J.kind == 0) {
			const std::string oligo = oligo_of(R, 0);
			const size_t len = amp.size(), oligo_len = oligo.size();
			// alignment_range_target() = (t_first, t_last); in amplicon coordinates (:224-227)
			const long r1 = long(J.gap_offset) + long(J.target_len) - long(b.t_first) - 1, r2 = long(J.gap_offset) + long(J.target_len) - long(b.t_last) - 1;
			if (do_replace) {
				amp = oligo + (size_t(r1 + 1) <= len ? amp.substr(size_t(r1 + 1), len - size_t(r1) - 1) : std::string());
				if (do_mask)
					for (size_t j = 0; j < oligo_len && j < amp.size(); j++) amp[j] = char(tolower(static_cast<unsigned char>(amp[j])));
			} else if (do_mask) {
				for (long j = r2; j <= r1; j++)
					if (j >= 0 && size_t(j) < amp.size()) amp[size_t(j)] = char(tolower(static_cast<unsigned char>(amp[size_t(j)])));
			}
		} else if (J.kind == 1) {
			const std::string oligo = oligo_of(R, 1);
			const size_t len = amp.size(), oligo_len = oligo.size();
			const long r1 = long(b.t_first) - long(J.gap_offset), r2 = long(b.t_last) - long(J.gap_offset);
			if (do_replace) {
				std::string comp(oligo_len, 'A');
				for (size_t i = 0; i < oligo_len; i++) comp[i] = base_complement_char(oligo[oligo_len - 1 - i]);
				const long keep = long(len) - long(J.target_len) + r1;
				amp = amp.substr(0, size_t(std::max(0l, std::min(keep, long(len))))) + comp;
				if (do_mask)
					for (size_t j = amp.size() - oligo_len; j < amp.size(); j++) amp[j] = char(tolower(static_cast<unsigned char>(amp[j])));
			} else if (do_mask) {
				const long stop = long(len) + r2 + 1 - long(J.target_len);
				for (long j = long(len) - long(J.target_len) + r1; j < stop; j++)
					if (j >= 0 && size_t(j) < amp.size()) amp[size_t(j)] = char(tolower(static_cast<unsigned char>(amp[size_t(j)])));
			}
		} else if (b.valid && b.tm >= opt.t.min_probe_tm) {
			long lo, hi;
			if (J.kind == 2) { // plus strand, :466-482
				lo = long(b.t_first) + long(J.pos) - (long(J.target_len) - 1);
				hi = long(b.t_last) + long(J.pos) - (long(J.target_len) - 1);
			} else { // minus strand, :556-570
				hi = long(J.pos) - long(b.t_first);
				lo = long(J.pos) - long(b.t_last);
			}
			for (long j = lo; j <= hi; j++)
				if (j >= 0 && size_t(j) < amp.size()) amp[size_t(j)] = char(tolower(static_cast<unsigned char>(amp[size_t(j)])));
		}
	}
}

void usage()
{
	std::cerr << "tntblast-mi355x: the tntblast search on an MI355X (FASTA databases)\n"
	             "\t-i <assay file> -d <FASTA database, plain or gzip> [-o <output file>] -e <min primer Tm> -E <min probe Tm>\n"
	             "\t[-x -X -z -Z -g -G -s -t -T -y -l -A <PCR|PROBE|PADLOCK|MIPS> -W -a -M -L -p -v -m <0..4> -n -S --best-match --plex --dump-query]\n"
	             "\t[--primer-clamp --min-max-primer-clamp --probe-clamp5 --probe-clamp3 --single-primer-pcr --target-strand\n"
	             "\t --temperature --dangle5 --dangle3 --max-gap --max-mismatch --max-poly-degen --rescale-ct --hash-size]"
	          << std::endl;
}

void parse_args(int argc, char **argv, Options &o)
{
	tnt_options_default(&o.t);
	int flag = 0;
	const option longs[] = {{"help", no_argument, &flag, 1},
	                        {"primer-clamp", required_argument, &flag, 3},
	                        {"probe-clamp5", required_argument, &flag, 4},
	                        {"probe-clamp3", required_argument, &flag, 5},
	                        {"plex", required_argument, &flag, 6},
	                        {"single-primer-pcr", required_argument, &flag, 7},
	                        {"hash-size", required_argument, &flag, 8},
	                        {"target-strand", required_argument, &flag, 9},
	                        {"temperature", required_argument, &flag, 10},
	                        {"max-target-len", required_argument, &flag, 11},
	                        {"query-seg", required_argument, &flag, 12},
	                        {"dump-query", required_argument, &flag, 13},
	                        {"dangle5", required_argument, &flag, 14},
	                        {"dangle3", required_argument, &flag, 15},
	                        {"min-max-primer-clamp", required_argument, &flag, 16},
	                        {"dinkelbach", required_argument, &flag, 17},
	                        {"max-gap", required_argument, &flag, 18},
	                        {"max-mismatch", required_argument, &flag, 19},
	                        {"rescale-ct", required_argument, &flag, 20},
	                        {"best-match", no_argument, &flag, 21},
	                        {"blast-include", required_argument, &flag, 22},
	                        {"blast-exclude", required_argument, &flag, 23},
	                        {"max-poly-degen", required_argument, &flag, 24},
	                        {"gpus", required_argument, &flag, 25},
	                        {nullptr, 0, nullptr, 0}};
	if (argc == 1) {
		usage();
		exit(EXIT_FAILURE);
	}
	int c, idx = 0;
	opterr = 0;
	while ((c = getopt_long(argc, argv, "-i:o:d:D:l:e:E:z:Z:x:X:g:G:s:t:T:y:A:W:m:a:M:k:K:r:v:p:n:L:S:?h", longs, &idx)) != EOF) {
		switch (c) {
		case 0:
			switch (flag) {
			case 1: usage(); exit(EXIT_SUCCESS);
			case 3: o.t.primer_clamp = atoi(optarg); break;
			case 4: o.t.probe_clamp_5 = atoi(optarg); break;
			case 5: o.t.probe_clamp_3 = atoi(optarg); break;
			case 6: o.multiplex = parse_bool(optarg); break;
			case 7: o.t.single_primer_pcr = parse_bool(optarg) ? 1 : 0; break;
			case 8: o.t.hash_word_size = atoi(optarg); break;
			case 9: {
				std::string v(optarg);
				for (char &ch : v) ch = char(toupper(ch));
				if (v == "PLUS" || v == "SENSE" || v == "+") o.t.target_strand = TNT_STRAND_PLUS;
				else if (v == "MINUS" || v == "ANTISENSE" || v == "-") o.t.target_strand = TNT_STRAND_MINUS;
				else if (v == "BOTH") o.t.target_strand = TNT_STRAND_BOTH;
				else fail("Unknown --target-strand value");
				break;
			}
			case 10: o.t.target_t = float(atof(optarg)); break;
			case 11: // --max-target-len: longer targets are searched in overlapping fragments like the reference does
			         // (options.cpp:177-184), so that its results come out -- edge effects included
				o.max_target_len = atol(optarg);
				if (o.max_target_len <= 1) fail("Please specify a --max-target-len > 1");
				break;
			case 12: { // --query-seg: how the reference deals queries to its workers, no effect on the results
				std::string v(optarg);
				for (char &ch : v) ch = char(tolower(ch));
				if (v != "always" && v != "never" && v != "adaptive") fail("Unknown --query-seg value");
				break;
			}
			case 13: o.dump_query = parse_bool(optarg); break;
			case 14: o.t.allow_dangle_5 = parse_bool(optarg) ? 1 : 0; break;
			case 15: o.t.allow_dangle_3 = parse_bool(optarg) ? 1 : 0; break;
			case 16: o.t.min_max_primer_clamp = atoi(optarg); break;
			case 17: o.t.use_dinkelbach = parse_bool(optarg) ? 1 : 0; break; // options.cpp --dinkelbach
			case 18: o.t.max_gap = atoi(optarg); break;
			case 19: o.t.max_mismatch = atoi(optarg); break;
			case 20: o.rescale_ct = parse_bool(optarg); break;
			case 21: o.best_match = true; break;
			case 24: o.t.max_poly_degen = abs(atoi(optarg)); break;
			case 25: // --gpus: the role of `mpirun -np` for the reference's master / worker build (tntblast_master.cpp)
				o.gpus = atoi(optarg);
				if (o.gpus < 1) fail("Please specify --gpus >= 1");
				break;
			default: fail(std::string("--") + longs[idx].name + " is not supported by the MI355X front end");
			}
			break;
		case 'i': o.input = optarg; break;
		case 'o': o.output = optarg; break;
		case 'd': o.dbase = optarg; break;
		case 'l': o.t.max_len = atoi(optarg); break;
		case 'e': o.t.min_primer_tm = float(atof(optarg)), o.have_primer_bounds = true; break;
		case 'E': o.t.min_probe_tm = float(atof(optarg)), o.have_probe_bounds = true; break;
		case 'z': o.t.min_primer_dg = float(atof(optarg)), o.have_primer_bounds = true; break;
		case 'Z': o.t.min_probe_dg = float(atof(optarg)), o.have_probe_bounds = true; break;
		case 'x': o.t.max_primer_tm = float(atof(optarg)), o.have_primer_bounds = true; break;
		case 'X': o.t.max_probe_tm = float(atof(optarg)), o.have_probe_bounds = true; break;
		case 'g': o.t.max_primer_dg = float(atof(optarg)), o.have_primer_bounds = true; break;
		case 'G': o.t.max_probe_dg = float(atof(optarg)), o.have_probe_bounds = true; break;
		case 's': o.t.salt = float(atof(optarg)); break;
		case 't': o.t.primer_strand = float(atof(optarg)); break;
		case 'T': o.t.probe_strand = float(atof(optarg)); break;
		case 'y': o.t.asymmetric_strand_ratio = float(atof(optarg)); break;
		case 'A': {
			std::string v(optarg);
			for (char &ch : v) ch = char(toupper(ch));
			if (v == "PCR") o.t.assay_format = TNT_ASSAY_PCR;
			else if (v == "PROBE") o.t.assay_format = TNT_ASSAY_PROBE;
			else if (v == "PADLOCK") o.t.assay_format = TNT_ASSAY_PADLOCK;
			else if (v == "MIPS") o.t.assay_format = TNT_ASSAY_MIPS;
			else fail("Unknown assay format (-A)");
			break;
		}
		case 'W': o.t.hash_word_size = atoi(optarg); break;
		case 'm':
			switch (atoi(optarg)) {
			case 0: o.format = Options::STANDARD; break;
			case 1: o.format = Options::FASTA; break;
			case 2: o.format = Options::NETWORK; break;
			case 3: o.format = Options::INVERSE_TARGET; break;
			case 4: o.format = Options::INVERSE_QUERY; break;
			default: fail("Unknown output format. Please specify a number between 0-3");
			}
			break;
		case 'a': o.show_alignments = parse_bool(optarg); break;
		case 'M': o.show_sequence = parse_bool(optarg); break;
		case 'v': o.verbose = parse_bool(optarg); break;
		case 'p': o.ignore_probe = parse_bool(optarg); break;
		case 'L': o.append_name = parse_bool(optarg); break;
		case 'n': o.per_query = parse_bool(optarg); break;
		case 'S': o.summary = parse_bool(optarg); break;
		case 'k': // options.cpp:371-394
			o.mask = parse_bool(optarg) ? (o.mask | Options::MASK_PRIMERS) : (o.mask & ~Options::MASK_PRIMERS);
			break;
		case 'K':
			o.mask = parse_bool(optarg) ? (o.mask | Options::MASK_PROBE) : (o.mask & ~Options::MASK_PROBE);
			break;
		case 'r':
			o.mask = parse_bool(optarg) ? (o.mask | Options::REPLACE_PRIMERS) : (o.mask & ~Options::REPLACE_PRIMERS);
			break;
		case 'D': o.local_dbase = optarg; break; // a database only the master node sees (options.cpp:288, tntblast_local.cpp:186-193): read like -d here
		case 'h':
		case '?': usage(); exit(EXIT_SUCCESS);
		default: std::cerr << '"' << char(c) << "\" is not a valid option!" << std::endl; break;
		}
	}
	if (o.dbase.empty() && o.local_dbase.empty()) fail("Unable to read either dbase or local_dbase"); // options.cpp:531-537
	if (!o.dbase.empty() && !o.local_dbase.empty()) fail("Please specify either dbase or local_dbase (but not both)");
	if (o.dbase.empty()) o.dbase = o.local_dbase;
	if (o.input.empty()) fail("Please specify an assay file (-i)");
	// validate_parameters, options.cpp:529-675 (the checks that concern the supported flags)
	if (o.ignore_probe && o.t.assay_format != TNT_ASSAY_PCR) fail("Error: Ignore probes (i.e. -p T) can only be used with a PCR-based assay format");
	if (o.t.salt <= 0.0f) fail("[Na+] (i.e. \"salt\") is less than zero");
	if (o.t.salt >= 1.0f) fail("[Na+] (i.e. \"salt\") is greater than 1M");
	if (o.t.primer_strand <= 0.0f) fail("[Ct] (i.e. \"primer_strand\") is less than zero");
	if (o.t.probe_strand < 0.0f) o.t.probe_strand = o.t.primer_strand;
	if (o.t.probe_strand <= 0.0f) fail("[Ct] (i.e. \"probe_strand\") is less than zero");
	if (o.t.asymmetric_strand_ratio <= 0.0f) fail("The ratio of forward to reverse primer [Ct] is <= 0");
	if (o.t.min_primer_tm > o.t.max_primer_tm) fail("min_primer_tm > max_primer_tm. Please use consistent values!");
	if (o.t.min_probe_tm > o.t.max_probe_tm) fail("min_probe_tm > max_probe_tm. Please use consistent values!");
}

void validate_search_threshold(Options &o, const std::vector<Assay> &assays) // options.cpp:833-916
{
	switch (o.t.assay_format) {
	case TNT_ASSAY_PCR:
		for (const Assay &a : assays) {
			if (!a.forward.empty() && !o.have_primer_bounds) fail("Please specify primer search bounds in Tm and/or Delta G");
			if (!a.probe.empty() && !o.have_probe_bounds) fail("Please specify probe search bounds in Tm and/or Delta G");
		}
		break;
	default:
		if (!o.have_probe_bounds) {
			if (!o.have_primer_bounds) fail("Please specify probe search bounds in Tm and/or Delta G");
			o.t.min_probe_dg = o.t.min_primer_dg, o.t.max_probe_dg = o.t.max_primer_dg;
			o.t.min_probe_tm = o.t.min_primer_tm, o.t.max_probe_tm = o.t.max_primer_tm;
		}
		break;
	}
}

} // namespace

int main(int argc, char **argv)
{
	// One context per device (--gpus N; default: every visible device).  The streaming reader's batches of whole records are
	// dealt to the devices in turn -- targets are independent, seq_index is global -- and searched by one host thread per
	// device; the hits of a round come back through the gang (compact blobs gathered to the first device over RCCL, one copy to
	// the host; host merge where RCCL cannot serve) and are filed in batch order, so the output does not depend on N.  This is
	// the role of the reference's master / workers (tntblast_master.cpp:438-461,811-828; tntblast_worker.cpp:173-186,602).
	std::vector<tnt_ctx *> ctxs;
	tnt_ctx *ctx = nullptr; // the first context: oligo structure, masking and fragment searches of batch 0's device ...
	tnt_gang *gang = nullptr;
	std::vector<tnt_db *> dbs;
	tnt_fasta_reader *reader = nullptr;
	auto release = [&]() {
		for (tnt_db *d : dbs)
			if (d) tnt_db_destroy(d);
		dbs.clear();
		if (reader) tnt_fasta_close(reader);
		reader = nullptr;
		if (gang) tnt_gang_destroy(gang);
		gang = nullptr;
		for (tnt_ctx *c : ctxs)
			if (c) tnt_ctx_destroy(c);
		ctxs.clear();
	};
	try {
		Options opt;
		parse_args(argc, argv, opt);
		StringTable st;
		std::vector<Assay> assays = read_input_file(opt.input, opt.ignore_probe, opt.t.assay_format == TNT_ASSAY_PROBE, st);
		if (opt.multiplex) assays = multiplex_expansion(assays, opt.t.assay_format, st); // tntblast_local.cpp:137-146
		const std::vector<Expanded> expanded = expand_assays(assays, opt.rescale_ct);
		if (opt.dump_query) // Options::write_queries, options.cpp:918-941: the expanded list
			for (const Expanded &e : expanded) {
				std::cout << assays[size_t(e.parent)].name;
				if (!e.forward.empty() && !e.reverse.empty()) std::cout << '\t' << e.forward << '\t' << e.reverse;
				if (!e.probe.empty()) std::cout << '\t' << e.probe;
				std::cout << std::endl;
			}
		if (assays.empty()) fail("No assays found in the input file");
		validate_search_threshold(opt, assays);

		std::vector<tnt_assay> ta(expanded.size());
		for (size_t i = 0; i < expanded.size(); i++) {
			const Expanded &e = expanded[i];
			ta[i].id = e.parent;
			ta[i].degen_id = int(i);
			ta[i].forward = e.forward.empty() ? nullptr : e.forward.c_str();
			ta[i].reverse = e.reverse.empty() ? nullptr : e.reverse.c_str();
			ta[i].probe = e.probe.empty() ? nullptr : e.probe.c_str();
			ta[i].forward_degen = e.forward_degen, ta[i].reverse_degen = e.reverse_degen, ta[i].probe_degen = e.probe_degen;
		}
		{
			// rank -> device: 0 .. N-1, or the list in TNT_MI355X_DEVICES (several ranks may name one device: host-merged gather)
			std::vector<int> devices;
			if (const char *e = getenv("TNT_MI355X_DEVICES")) {
				std::stringstream ss(e);
				std::string tok;
				while (std::getline(ss, tok, ','))
					if (!tok.empty()) devices.push_back(atoi(tok.c_str()));
			}
			const int visible = tnt_device_count();
			if (visible < 1) fail("no HIP device");
			int n = opt.gpus ? opt.gpus : (devices.empty() ? visible : int(devices.size()));
			if (devices.empty()) {
				if (n > visible) fail("--gpus asks for more devices than are visible");
				for (int r = 0; r < n; r++) devices.push_back(r);
			}
			if (int(devices.size()) < n) fail("TNT_MI355X_DEVICES names fewer devices than --gpus");
			devices.resize(size_t(n));
			for (int r = 0; r < n; r++) {
				tnt_ctx *c = nullptr;
				if (tnt_ctx_create(&opt.t, ta.data(), ta.size(), devices[size_t(r)], &c) != TNT_OK) fail(tnt_last_error(nullptr));
				ctxs.push_back(c);
			}
			ctx = ctxs[0];
			if (tnt_gang_create(ctxs.data(), n, &gang) != TNT_OK) fail(tnt_gang_last_error(nullptr));
			// the transport: on request, and always when RCCL was wanted and the gather fell back to host copies
			const std::string transport = tnt_gang_transport(gang);
			if (getenv("TNT_MI355X_VERBOSE_GANG") || (n > 1 && transport.rfind("host (", 0) == 0))
				std::cerr << "tntblast-mi355x: " << n << " device context(s), gather over " << transport << std::endl;
			if (const char *w = tnt_ctx_warnings(ctx); w && *w) std::cerr << "tntblast-mi355x: warning: " << w << std::endl;
		}
		const size_t n_rank = ctxs.size();
		if (opt.verbose) std::cout << "Reading sequence database: " << opt.dbase << std::endl;
		// The database streams through the device in batches of whole records (TNT_MI355X_CHUNK_BYTES of text each,
		// default 1 GiB): the next batch is read and inflated while this one is searched.
		if (tnt_fasta_open(opt.dbase.c_str(), &reader) != TNT_OK) fail("Unable to open fasta sequence file for reading");
		size_t chunk = size_t(1) << 30;
		if (const char *e = getenv("TNT_MI355X_CHUNK_BYTES")) chunk = std::max<size_t>(1, strtoull(e, nullptr, 10));
		// results per assay id, in hybrid_sig order (tntblast_local.cpp:703-705,931)
		std::vector<std::vector<Record>> results(assays.size());
		std::vector<char> amp(1 << 20);
		bool fragmented = false; // some target was searched in pieces: the results are made unique at the end
		// Options::max_product_length + 2 (options.cpp:790-830, tntblast_local.cpp:174): how far a fragment reaches beyond its cut
		uint64_t max_product_length = 0;
		if (opt.t.assay_format == TNT_ASSAY_PCR) {
			for (const Expanded &e : expanded) {
				if (!e.forward.empty() && !e.reverse.empty()) {
					max_product_length = uint64_t(std::max(0, opt.t.max_len));
					break;
				}
				max_product_length = std::max<uint64_t>(max_product_length, e.probe.size());
			}
		} else if (opt.t.assay_format == TNT_ASSAY_PADLOCK) {
			for (const Expanded &e : expanded) max_product_length = std::max<uint64_t>(max_product_length, e.forward.size() + e.reverse.size());
		} else {
			for (const Expanded &e : expanded) max_product_length = std::max<uint64_t>(max_product_length, e.probe.size());
		}
		max_product_length += 2;
		size_t n_targets = 0, n_seen = 0;
		std::vector<std::string> all_deflines; // -m 3 only
		struct Batch { // one batch of whole records on its way through a device
			tnt_db *db = nullptr;
			size_t first = 0;
			std::vector<char> is_long;
			std::vector<size_t> long_targets;
			std::string error;
		};
		bool at_end = false;
		std::vector<tnt_hit> unpacked;
		while (!at_end) {
			// ---- a round: up to one batch per device, read in file order
			std::vector<Batch> round;
			for (size_t k = 0; k < n_rank && !at_end; k++) {
				tnt_db *db = nullptr;
				if (tnt_fasta_next(reader, ctxs[k], chunk, &db) != TNT_OK) fail(tnt_last_error(ctxs[k]));
				if (!db) {
					at_end = true;
					break;
				}
				dbs.push_back(db);
				Batch B;
				B.db = db;
				B.first = n_targets;
				n_targets += tnt_db_num_targets(db);
				if (opt.format & Options::INVERSE_TARGET)
					for (size_t t = 0; t < tnt_db_num_targets(db); t++) {
						const char *defline = "";
						size_t dl = 0;
						(void)tnt_db_defline(db, t, &defline, &dl);
						all_deflines.emplace_back(defline, dl);
					}
				// targets whose record is longer than --max-target-len are searched in fragments (tntblast_local.cpp:282-289,
				// 438-468): the record's size in the file -- defline and line ends included -- is what the reference goes by
				B.is_long.assign(tnt_db_num_targets(db), 0);
				for (size_t t = 0; t < tnt_db_num_targets(db); t++) {
					uint64_t approx = 0;
					if (tnt_db_target_info(db, t, nullptr, &approx) != TNT_OK) fail("tnt_db_target_info failed");
					if (approx > uint64_t(opt.max_target_len)) B.is_long[t] = 1, B.long_targets.push_back(t);
				}
				round.push_back(std::move(B));
			}
			if (round.empty()) break;
			// ---- every device searches its batch (one host thread per context; the records stay on the devices)
			auto search_batch = [&](size_t k) {
				Batch &B = round[k];
				// (records beyond --max-target-len are not searched whole: only their fragments are, below)
				if (!B.long_targets.empty() && tnt_db_skip_targets(ctxs[k], B.db, reinterpret_cast<const uint8_t *>(B.is_long.data()), B.is_long.size()) != TNT_OK) {
					B.error = tnt_last_error(ctxs[k]);
					return;
				}
				size_t n = 0;
				if (tnt_search_enqueue(ctxs[k], B.db) != TNT_OK || tnt_search_collect(ctxs[k], nullptr, &n) != TNT_OK) B.error = tnt_last_error(ctxs[k]);
			};
			{
				std::vector<std::thread> workers;
				for (size_t k = 1; k < round.size(); k++) workers.emplace_back(search_batch, k);
				search_batch(0);
				for (std::thread &w : workers) w.join();
			}
			for (const Batch &B : round)
				if (!B.error.empty()) fail(B.error);
			// ---- the hits of the round come to the host through the gang, rank by rank = batch by batch
			std::vector<uint8_t> has(n_rank, 0);
			for (size_t k = 0; k < round.size(); k++) has[k] = 1;
			std::vector<const void *> blobs(n_rank, nullptr);
			std::vector<size_t> blob_bytes(n_rank, 0), blob_hits(n_rank, 0);
			if (tnt_gang_gather(gang, has.data(), blobs.data(), blob_bytes.data(), blob_hits.data()) != TNT_OK) fail(tnt_gang_last_error(gang));
			for (size_t k = 0; k < round.size(); k++) {
				tnt_ctx *ctx = ctxs[k]; // (this batch's device: amplicon text, fragments and masking run where its targets are)
				tnt_db *db = round[k].db;
				const size_t first = round[k].first;
				const std::vector<char> &is_long = round[k].is_long;
				const std::vector<size_t> &long_targets = round[k].long_targets;
				size_t n_hits = 0;
				unpacked.resize(std::max<size_t>(1, blob_hits[k]));
				if (tnt_hits_unpack(blobs[k], blob_bytes[k], unpacked.data(), unpacked.size(), &n_hits) != TNT_OK || n_hits != blob_hits[k]) fail("malformed hit blob");
				const tnt_hit *hits = unpacked.data();
				std::vector<Record> batch; // this batch's records: masked (if asked for), then filed under their assays
				for (size_t i = 0; i < n_hits; i++) {
					if (is_long[size_t(hits[i].seq_index) - first]) continue; // (none: skipped above)
					Record rec{hits[i], n_seen++, std::string(), std::string()};
					resolve_long_alignments(ctx, rec);
					const int na = tnt_hit_sequence(ctx, db, &hits[i], amp.data(), amp.size());
					if (na < 0) fail("amplicon longer than the sequence buffer");
					rec.amplicon.assign(amp.data(), size_t(na));
					const char *defline = "";
					size_t dl = 0;
					(void)tnt_db_defline(db, size_t(hits[i].seq_index) - first, &defline, &dl);
					rec.defline.assign(defline, dl);
					batch.push_back(rec);
				}
				if (!long_targets.empty()) {
					// the fragments become the targets of a database of their own
					struct Frag {
						size_t target;
						unsigned int start, stop, max_stop, len; // [start, stop] of the cut, bases actually read
					};
					std::vector<Frag> frags;
					std::vector<uint8_t> codes;
					std::vector<uint64_t> offs(1, 0);
					std::vector<uint8_t> whole;
					for (size_t t : long_targets) {
						uint32_t n_bases = 0;
						uint64_t approx = 0;
						(void)tnt_db_target_info(db, t, &n_bases, &approx);
						whole.resize(std::max<size_t>(1, n_bases));
						if (tnt_db_target_codes(ctx, db, t, whole.data(), whole.size()) < 0) fail(tnt_last_error(ctx));
						const unsigned int max_stop = (unsigned int)(approx - 1);
						const unsigned int delta = seq_len_increment((unsigned int)approx, (unsigned int)opt.max_target_len).first;
						unsigned int start = 0, stop = delta;
						while (true) {
							// read_bio_seq(target, start, stop + max_product_length), sequence_data_fastx.cpp:296-372
							const uint64_t want_stop = uint64_t(stop) + max_product_length;
							const uint64_t last = std::min<uint64_t>(want_stop, n_bases ? n_bases - 1 : 0);
							const unsigned int len = (n_bases && start <= last) ? (unsigned int)(last - start + 1) : 0u;
							if (len >= (unsigned int)opt.t.hash_word_size) { // shorter pieces are skipped (tntblast_local.cpp:513-531)
								frags.push_back(Frag{t, start, stop, max_stop, len});
								codes.insert(codes.end(), whole.begin() + start, whole.begin() + start + len);
								offs.push_back(codes.size());
							}
							if (stop == max_stop) break;
							start = stop + 1;
							stop = std::min(stop + delta, max_stop);
							fragmented = true;
						}
					}
					if (!frags.empty()) {
						tnt_db *fdb = nullptr;
						if (tnt_db_create(ctx, codes.data(), offs.data(), frags.size(), 0, &fdb) != TNT_OK) fail(tnt_last_error(ctx));
						const tnt_hit *fh = nullptr;
						size_t nf = 0;
						if (tnt_search(ctx, fdb, &fh, &nf) != TNT_OK) fail(tnt_last_error(ctx));
						for (size_t i = 0; i < nf; i++) {
							const Frag &F = frags[size_t(fh[i].seq_index)];
							const int lo = fh[i].has_primers ? fh[i].amplicon_start : fh[i].probe_start;
							const int hi = fh[i].has_primers ? fh[i].amplicon_stop : fh[i].probe_stop;
							// matches cut off by the fragment's edges are dropped (tntblast_local.cpp:638-648)
							if (F.start != 0 && lo <= 0) continue;
							if (F.stop != F.max_stop && hi >= int(F.len) - 1) continue;
							Record rec{fh[i], n_seen++, std::string(), std::string()};
							resolve_long_alignments(ctx, rec);
							const int na = tnt_hit_sequence(ctx, fdb, &fh[i], amp.data(), amp.size());
							if (na < 0) fail("amplicon longer than the sequence buffer");
							rec.amplicon.assign(amp.data(), size_t(na));
							rec.h.seq_index = int(first + F.target);
							if (rec.h.has_primers) rec.h.amplicon_start += int(F.start), rec.h.amplicon_stop += int(F.start); // offset_ranges
							if (rec.h.has_probe) rec.h.probe_start += int(F.start), rec.h.probe_stop += int(F.start);
							const char *defline = "";
							size_t dl = 0;
							(void)tnt_db_defline(db, F.target, &defline, &dl);
							rec.defline.assign(defline, dl);
							batch.push_back(rec);
						}
						tnt_db_destroy(fdb);
					}
				}
				apply_masks(ctx, opt, batch, 0, [&](const Record &r, int slot) { // tntblast_util.cpp:97-137
					const Expanded &e = expanded[size_t(r.h.degen_id)];
					if (slot == 2) return e.probe;
					const int which = slot == 0 ? r.h.forward_oligo : r.h.reverse_oligo; // single-primer amplicons swap identities
					return which == 0 ? e.forward : e.reverse;
				});
				for (Record &rec : batch) results[size_t(rec.h.assay_id)].push_back(rec);
				tnt_db_destroy(db);
				for (tnt_db *&d : dbs)
					if (d == db) d = nullptr;
			}
			dbs.clear();
		}
		tnt_fasta_close(reader);
		reader = nullptr;
		if (opt.verbose) {
			std::cout << "Found " << n_targets << " database sequences" << std::endl;
			std::cout << "Found " << assays.size() << " query assays" << std::endl;
		}
		// ---- output, tntblast_local.cpp:70-130 (files) and :877-1370 (records, summaries)
		std::ofstream fout, fout_sif, fout_atr;
		std::ostream *out = &std::cout;
		if (!opt.output.empty()) {
			if (!opt.per_query) {
				if (opt.format & (Options::STANDARD | Options::FASTA)) {
					fout.open(opt.output.c_str());
					if (!fout) fail("Unable to open output file");
				}
				if (opt.format & Options::NETWORK) {
					fout_sif.open((opt.output + ".sif").c_str());
					if (!fout_sif) fail("Unable to open " + opt.output + ".sif");
				}
			}
			if (opt.format & Options::NETWORK) { // one attribute file per run
				fout_atr.open((opt.output + ".atr").c_str());
				if (!fout_atr) fail("Unable to open " + opt.output + ".atr");
				fout_atr << "FunctionalCatagory" << std::endl;
			}
			if (opt.format & (Options::INVERSE_TARGET | Options::INVERSE_QUERY)) {
				fout.open(opt.output.c_str());
				if (!fout) fail("Unable to open output file");
			}
			out = &fout;
		}
		const bool standard = (opt.format & Options::STANDARD) != 0;
		auto mask_white_space = [](std::string v) { // tntblast_util.cpp:1298-1311
			for (char &ch : v)
				if (isspace(static_cast<unsigned char>(ch))) ch = '_';
			return v;
		};

		if (opt.format & Options::INVERSE_QUERY) {
			// names of the queries without a match.  The reference keeps one flag per assay id but walks the *expanded*
			// list with the same index (tntblast_local.cpp:877-903); the two agree unless an assay is degenerate.
			std::set<std::string> query_set, match_set, diff;
			for (size_t i = 0; i < expanded.size(); i++) {
				const std::string &name = assays[size_t(expanded[i].parent)].name;
				query_set.insert(name);
				if (i < results.size() && !results[i].empty()) match_set.insert(name);
			}
			std::set_difference(query_set.begin(), query_set.end(), match_set.begin(), match_set.end(), std::inserter(diff, diff.begin()));
			for (const std::string &name : diff) *out << name << std::endl;
			for (auto &r : results) r.clear(); // this format keeps no matches (:701-716)
		}

		// strand concentrations of the self-structure values, tntblast_local.cpp:228-234,656-686
		const float fwd_strand = opt.t.asymmetric_strand_ratio * opt.t.primer_strand, rev_strand = opt.t.primer_strand;
		std::map<std::string, std::pair<float, float>> self_cache; // oligo + role -> (hairpin, homodimer)
		auto self_tm = [&](const std::string &oligo, float ct, char role) {
			const std::string key = oligo + role;
			auto it = self_cache.find(key);
			if (it != self_cache.end()) return it->second;
			float h = 0, d = 0, x = 0;
			if (tnt_oligo_structure(ctx, oligo.c_str(), nullptr, ct, ct, &h, &d, &x) != TNT_OK) fail(tnt_last_error(ctx));
			return self_cache[key] = std::make_pair(h, d);
		};
		std::map<std::string, float> dimer_cache;
		auto hetero_tm = [&](const std::string &f, const std::string &r) {
			const std::string key = f + "/" + r;
			auto it = dimer_cache.find(key);
			if (it != dimer_cache.end()) return it->second;
			float h = 0, d = 0, x = 0;
			if (tnt_oligo_structure(ctx, f.c_str(), r.c_str(), fwd_strand, rev_strand, &h, &d, &x) != TNT_OK) fail(tnt_last_error(ctx));
			return dimer_cache[key] = x;
		};

		// statistics of the final summary, tntblast_local.cpp:236-262
		typedef std::pair<float, float> FRange;
		typedef std::pair<unsigned int, unsigned int> URange;
		auto widen = [](FRange &r, float v) { r.first = std::min(r.first, v), r.second = std::max(r.second, v); };
		auto widen_u = [](URange &r, size_t v) { r.first = std::min(r.first, (unsigned int)v), r.second = std::max(r.second, (unsigned int)v); };
		FRange forward_tm_range(9999.0f, -1.0f), reverse_tm_range(9999.0f, -1.0f), probe_tm_range(9999.0f, -1.0f);
		FRange forward_dg_range(9999.0f, -9999.0f), reverse_dg_range(9999.0f, -9999.0f), probe_dg_range(9999.0f, -9999.0f);
		FRange forward_gc_range(9999.0f, -1.0f), reverse_gc_range(9999.0f, -1.0f), probe_gc_range(9999.0f, -1.0f);
		URange amplicon_size_range(9999, 0), forward_size_range(9999, 0), reverse_size_range(9999, 0), probe_size_range(9999, 0);
		float max_primer_hairpin = -1.0f, max_primer_homodimer = -1.0f, max_primer_heterodimer = -1.0f;
		float max_probe_homodimer = -1.0f, max_probe_hairpin = -1.0f;
		unsigned int num_primer = 0, num_probe = 0;

		std::set<std::string> total_unique_targets;
		std::vector<size_t> match_count(results.size(), 0);
		for (size_t id = 0; id < results.size(); id++) {
			std::vector<Record> &list = results[id];
			if (list.empty()) continue;
			if (opt.best_match) select_best_match(list);
			if (fragmented) // tntblast_local.cpp:923-928
				uniquify_results(list, [&](const Record &r, int slot) {
					const Expanded &e = expanded[size_t(r.h.degen_id)];
					const int which = slot == 0 ? r.h.forward_oligo : r.h.reverse_oligo; // single-primer amplicons swap identities
					return int((which == 0 ? e.forward : e.reverse).size());
				});
			std::stable_sort(list.begin(), list.end(), record_less);
			if (opt.per_query) { // :933-965
				const std::string filename = opt.output + "." + assays[id].name;
				if (opt.format & (Options::STANDARD | Options::FASTA)) {
					fout.close();
					fout.open(filename.c_str());
					if (!fout) fail("Unable to open " + filename);
				}
				if (opt.format & Options::NETWORK) {
					fout_sif.close();
					fout_sif.open((filename + ".sif").c_str());
					if (!fout_sif) fail("Unable to open " + filename + ".sif");
				}
			}
			if (standard) *out << "#####################################################################################" << std::endl;
			std::set<std::string> unique_targets;
			for (const Record &rec : list) {
				const tnt_hit &h = rec.h;
				const Expanded &e = expanded[size_t(h.degen_id)];
				if (standard) *out << "name = " << assays[id].name << std::endl;
				const std::string &amplicon = rec.amplicon;
				std::string fp, rp;
				if (h.has_primers) {
					num_primer++;
					const bool pcr = opt.t.assay_format == TNT_ASSAY_PCR;
					fp = pcr ? "forward primer" : "5' probe";
					rp = pcr ? "reverse primer" : "3' probe";
					// single-primer amplicons swap identities (amplicon_search.cpp:453-463)
					const std::string &fo = h.forward_oligo == 0 ? e.forward : e.reverse;
					const std::string &ro = h.reverse_oligo == 1 ? e.reverse : e.forward;
					const float forward_dg = h.forward_dH - opt.t.target_t * h.forward_dS;
					const float reverse_dg = h.reverse_dH - opt.t.target_t * h.reverse_dS;
					const std::pair<float, float> fs = self_tm(fo, fwd_strand, 'F'), rs = self_tm(ro, rev_strand, 'R');
					const float hetero = hetero_tm(fo, ro);
					const float fgc = 100.0f * gc_content(fo), rgc = 100.0f * gc_content(ro);
					max_primer_hairpin = std::max(max_primer_hairpin, std::max(fs.first, rs.first));
					max_primer_homodimer = std::max(max_primer_homodimer, std::max(fs.second, rs.second));
					max_primer_heterodimer = std::max(max_primer_heterodimer, hetero);
					widen_u(forward_size_range, fo.size()), widen_u(reverse_size_range, ro.size());
					widen(forward_tm_range, h.forward_tm), widen(forward_dg_range, forward_dg);
					widen(reverse_tm_range, h.reverse_tm), widen(reverse_dg_range, reverse_dg);
					widen(forward_gc_range, fgc), widen(reverse_gc_range, rgc);
					widen_u(amplicon_size_range, amplicon.size());
					if (standard) {
						*out << fp << " = 5' " << fo << " 3'" << std::endl;
						*out << rp << " = 5' " << ro << " 3'" << std::endl;
						*out << fp << " tm = " << h.forward_tm << std::endl;
						*out << rp << " tm = " << h.reverse_tm << std::endl;
						*out << fp << " hairpin tm = " << fs.first << std::endl;
						*out << rp << " hairpin tm = " << rs.first << std::endl;
						*out << fp << " homodimer tm = " << fs.second << std::endl;
						*out << rp << " homodimer tm = " << rs.second << std::endl;
						*out << "heterodimer tm = " << hetero << std::endl;
						*out << fp << " dG[" << forward_dg << "] = dH[" << h.forward_dH << "] - T*dS[" << h.forward_dS << "]" << std::endl;
						*out << rp << " dG[" << reverse_dg << "] = dH[" << h.reverse_dH << "] - T*dS[" << h.reverse_dS << "]" << std::endl;
						*out << fp << " mismatches = " << h.forward_mm << std::endl;
						*out << rp << " mismatches = " << h.reverse_mm << std::endl;
						*out << fp << " gaps = " << h.forward_gap << std::endl;
						*out << rp << " gaps = " << h.reverse_gap << std::endl;
						if (pcr) {
							*out << "min 3' clamp = " << std::min(h.forward_primer_clamp, h.reverse_primer_clamp) << std::endl;
							*out << "max 3' clamp = " << std::max(h.forward_primer_clamp, h.reverse_primer_clamp) << std::endl;
						} else {
							*out << "5' probe 3' ligation clamp = " << h.forward_primer_clamp << std::endl;
							*out << "3' probe 5' ligation clamp = " << h.reverse_primer_clamp << std::endl;
						}
						*out << fp << " %GC = " << fgc << std::endl;
						*out << rp << " %GC = " << rgc << std::endl;
						*out << fp << " heuristics = " << primer_heuristics(fo) << std::endl;
						*out << rp << " heuristics = " << primer_heuristics(ro) << std::endl;
						if (pcr) {
							*out << "amplicon range = " << h.amplicon_start << " .. " << h.amplicon_stop << std::endl;
							*out << "amplicon length = " << amplicon.size() << std::endl;
						} else {
							*out << "product range = " << h.amplicon_start << " .. " << h.amplicon_stop << std::endl;
							*out << "product length = " << amplicon.size() << std::endl;
						}
						if (h.primer_strand == TNT_PLUS) *out << "Forward primer is contained in the target plus strand (+)" << std::endl;
						else *out << "Forward primer is contained in the target minus strand (-)" << std::endl;
					}
				}
				if (h.has_probe) {
					num_probe++;
					const float probe_dg = h.probe_dH - opt.t.target_t * h.probe_dS;
					const std::pair<float, float> ps = self_tm(e.probe, opt.t.probe_strand, 'P');
					const float pgc = 100.0f * gc_content(e.probe);
					widen(probe_gc_range, pgc), widen(probe_tm_range, h.probe_tm), widen(probe_dg_range, probe_dg);
					widen_u(probe_size_range, e.probe.size());
					max_probe_hairpin = std::max(max_probe_hairpin, ps.first);
					max_probe_homodimer = std::max(max_probe_homodimer, ps.second);
					if (standard) {
						*out << "probe = 5' " << e.probe << " 3'" << std::endl;
						*out << "probe tm = " << h.probe_tm << std::endl;
						*out << "probe hairpin tm = " << ps.first << std::endl;
						*out << "probe homodimer tm = " << ps.second << std::endl;
						*out << "probe dG[" << probe_dg << "] = dH[" << h.probe_dH << "] - T*dS[" << h.probe_dS << "]" << std::endl;
						*out << "probe mismatches = " << h.probe_mm << std::endl;
						*out << "probe gaps = " << h.probe_gap << std::endl;
						*out << "probe %GC = " << pgc << std::endl;
						*out << "probe range = " << h.probe_start << " .. " << h.probe_stop << std::endl;
						if (h.probe_strand != h.primer_strand) *out << "probe contained in forward strand (+)" << std::endl;
						else *out << "probe contained in reverse strand (-)" << std::endl;
					}
				}
				if (standard && opt.show_alignments) {
					write_alignment(*out, fp + " align ", rec.align(0));
					write_alignment(*out, rp + " align ", rec.align(1));
					write_alignment(*out, "probe align ", rec.align(2));
				}
				if (opt.format & (Options::STANDARD | Options::FASTA)) {
					*out << '>' << rec.defline;
					if (opt.append_name) *out << ' ' << assays[id].name;
					*out << std::endl;
					if (opt.show_sequence) *out << amplicon << std::endl;
				}
				if (standard) *out << std::endl;
				if (opt.format & Options::NETWORK) fout_sif << mask_white_space(assays[id].name) << " binds " << mask_white_space(rec.defline) << std::endl;
				unique_targets.insert(rec.defline);
				total_unique_targets.insert(rec.defline);
			}
			match_count[id] = unique_targets.size();
		}
		if (opt.format & Options::NETWORK) { // :1282-1294
			for (const Expanded &e : expanded) fout_atr << mask_white_space(assays[size_t(e.parent)].name) << " = parent" << std::endl;
			for (const std::string &t : total_unique_targets) fout_atr << mask_white_space(t) << " = child" << std::endl;
		}
		if (opt.format & Options::INVERSE_TARGET) { // write_inverse_matches, tntblast_util.cpp:1244-1272
			unsigned int count = 0;
			for (const std::string &d : all_deflines)
				if (total_unique_targets.find(d) == total_unique_targets.end()) {
					*out << d << std::endl;
					count++;
				}
			if (opt.verbose) std::cout << "Wrote " << count << " inverse target matches (that did not match any query!)" << std::endl;
		}
		const bool inverse_query = (opt.format & Options::INVERSE_QUERY) != 0;
		if (opt.verbose && !inverse_query) std::cout << "Found " << total_unique_targets.size() << " (total) target sequence matches" << std::endl;
		if (opt.verbose && num_primer > 0 && !inverse_query) { // :1312-1333
			std::cout << "Amplicon:" << std::endl << "\t" << amplicon_size_range.first << " <= Amplicon length <= " << amplicon_size_range.second << std::endl;
			std::cout << "Forward primer:" << std::endl
			          << "\t" << forward_tm_range.first << " <= Tm (C) <= " << forward_tm_range.second << std::endl
			          << "\t" << forward_dg_range.first << " <= Delta G (Kcal/Mol) <= " << forward_dg_range.second << std::endl
			          << "\t" << forward_gc_range.first << " <= %GC <= " << forward_gc_range.second << std::endl
			          << "\t" << forward_size_range.first << " <= length <= " << forward_size_range.second << std::endl;
			std::cout << "Reverse primer:" << std::endl
			          << "\t" << reverse_tm_range.first << " <= Tm (C) <= " << reverse_tm_range.second << std::endl
			          << "\t" << reverse_dg_range.first << " <= Delta G (Kcal/Mol) <= " << reverse_dg_range.second << std::endl
			          << "\t" << reverse_gc_range.first << " <= %GC <= " << reverse_gc_range.second << std::endl
			          << "\t" << reverse_size_range.first << " <= length <= " << reverse_size_range.second << std::endl;
			std::cout << "Max primer hairpin Tm = " << max_primer_hairpin << std::endl;
			std::cout << "Max primer heterodimer Tm = " << max_primer_heterodimer << std::endl;
			std::cout << "Max primer homodimer Tm = " << max_primer_homodimer << std::endl;
		}
		if (opt.verbose && num_probe > 0 && !inverse_query) { // :1335-1345
			std::cout << "Probe:" << std::endl
			          << "\t" << probe_tm_range.first << " <= Tm (C) <= " << probe_tm_range.second << std::endl
			          << "\t" << probe_dg_range.first << " <= Delta G (Kcal/Mol) <= " << probe_dg_range.second << std::endl
			          << "\t" << probe_gc_range.first << " <= %GC <= " << probe_gc_range.second << std::endl
			          << "\t" << probe_size_range.first << " <= length <= " << probe_size_range.second << std::endl;
			std::cout << "Max probe hairpin Tm = " << max_probe_hairpin << std::endl;
			std::cout << "Max probe homodimer Tm = " << max_probe_homodimer << std::endl;
		}
		if (opt.summary && !inverse_query) { // :1347-1366, one entry per expanded assay
			std::cout << "*** Assay Summary ***" << std::endl;
			for (const Expanded &e : expanded) {
				std::cout << assays[size_t(e.parent)].name << " matched " << match_count[size_t(e.parent)] << " sequences" << std::endl;
				if (!e.forward.empty() && !e.reverse.empty()) std::cout << "\tF::R = " << e.forward << " :: " << e.reverse << std::endl;
				if (!e.probe.empty()) std::cout << "\tP = " << e.probe << std::endl;
			}
		}
		release();
		return EXIT_SUCCESS;
	} catch (const Fail &f) {
		std::cerr << "Caught the error: " << f.msg << std::endl;
		release();
		return EXIT_FAILURE;
	}
}
