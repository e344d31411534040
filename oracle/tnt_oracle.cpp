// tnt_oracle.cpp -- TEST INFRASTRUCTURE ONLY (the parity oracle).
//
// A CPU restatement, written for this repository, of the algorithm of
// ThermonucleotideBLAST's per-target inner loop (reference v2.77 under
// /root/reference; every function cites the file:line it follows).  It is the
// checker for the HIP path and the "port" CPU baseline of bench.py; it is
// pinned against the reference itself (oracle/_ref/libtntref.so, built from the
// reference's own sources) by tests/test_oracle_vs_reference.py and against the
// committed golden vectors under tests/golden/ by tests/test_oracle_golden.py.
//
// Nothing in the product (thermonucleotideblast_amd/, include/) may include,
// link or call this file.
//
// Known, deliberate fences (also listed in DESIGN.md):
//  * Q4  cull_oligo_match reads one past a list node when counting strands
//        (amplicon_search.cpp:748-755); end() counts as "no strand" here.
//  * Q7  trace_back can index query[query_len] if a row-1 cell has M >= 0
//        (nuc_cruc.cpp:1497-1541; impossible at default T/salt): row 0 is
//        treated as terminal without pushing a column.
//  * Q11 when no enumerated alignment is valid the reference keeps stale
//        first/last_match; such windows are reported as "no bind".
//  * a6  the per-target BindCache (tntblast.h:247-324) IS modelled (round 4): amplicon() and padlock() memoise every
//        evaluated (oligo string, window) per target and strand -- the partial record where the threshold cascade
//        stopped -- and re-apply the caller's thresholds to the memo on a hit (bind_oligo.cpp:508-512,750-803), whatever
//        strand concentration and threshold set made it; hybrid() binds with a cache of its own per call
//        (probe_search.cpp:96,166).  tntor_set_bind_cache(0) switches the memo off (every use evaluated on its own).

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <list>
#include <string>
#include <unordered_map>
#include <vector>
#include <omp.h>

#include "../include/tnt_mi355x.h"
#include "tnt_oracle_params.inc"

namespace tor {

// ---------------------------------------------------------------------------------------------
// Encodings: nuc_cruc.h:181-188
enum Base : uint8_t { A = 0, C, G, T, I, E, GAP, M, R, S, V, W, Y, H, K, D, B, N };
static const int NB = 7;   // NUM_BASE, nuc_cruc.h:37
static const int NBP = 49; // NUM_BASE_PAIR
static const int P_AT = A * NB + T, P_TA = T * NB + A, P_CG = C * NB + G, P_GC = G * NB + C;
static const int P_GT = G * NB + T, P_TG = T * NB + G, P_EE = E * NB + E, P__ = GAP * NB + GAP;

static inline bool is_virtual(int b) { return b == E || b == GAP; }                   // nuc_cruc.h:52
static inline bool non_virtual_pair(int p) { return (p % NB < E) && (p / NB < E); }    // nuc_cruc.h:45
static inline bool has_gap(int p) { return (p % NB == GAP) || (p / NB >= GAP); }       // nuc_cruc.h:47

static inline float f32(unsigned int bits)
{
	float f;
	memcpy(&f, &bits, 4);
	return f;
}

// nuc_cruc.cpp:14-201.  Optimistic resolution of an IUPAC code against its partner.
// Case B falls through into case N in the reference (no `break`, :163-197) -- reproduced.
static Base resolve(Base x, Base partner)
{
	switch (x) {
	case M: return partner == T ? A : (partner == G ? C : A);
	case R: return partner == T ? A : (partner == C ? G : A);
	case S: return partner == G ? C : (partner == C ? G : G);
	case V: return partner == G ? C : (partner == C ? G : (partner == T ? A : A));
	case W: return partner == A ? T : (partner == T ? A : A);
	case Y: return partner == G ? C : (partner == A ? T : T);
	case H: return partner == T ? A : (partner == G ? C : (partner == A ? T : A));
	case K: return partner == C ? G : (partner == A ? T : T);
	case D: return partner == C ? G : (partner == T ? A : (partner == A ? T : A));
	case B: // result of the B switch is overwritten by the N switch
	case N: return partner == A ? T : (partner == T ? A : (partner == G ? C : (partner == C ? G : A)));
	default: return x;
	}
}

// nuc_cruc.cpp:203-213
static inline int bbp(Base a, Base b) { return int(resolve(a, b)) * NB + int(resolve(b, a)); }

// nuc_cruc_anchor.cpp:8-139
static bool is_complementary(Base q, Base t)
{
	// bit0 A, bit1 T, bit2 G, bit3 C
	static const unsigned qm[18] = {1, 8, 4, 2, 15, 0, 0, 1 | 8, 4 | 1, 4 | 8, 4 | 8 | 1, 1 | 2, 2 | 8, 1 | 8 | 2, 4 | 2, 4 | 1 | 2, 4 | 2 | 8, 15};
	// target mapped to its complement
	static const unsigned tm[18] = {2, 4, 8, 1, 15, 0, 0, 2 | 4, 8 | 2, 8 | 4, 8 | 4 | 2, 2 | 1, 1 | 4, 2 | 4 | 1, 8 | 1, 8 | 2 | 1, 8 | 1 | 4, 15};
	return (qm[q] & tm[t]) != 0;
}

static Base char_to_base(char c) // nuc_cruc.h:190-231
{
	switch (c) {
	case 'A': case 'a': return A;
	case 'C': case 'c': return C;
	case 'G': case 'g': return G;
	case 'T': case 't': return T;
	case 'I': case 'i': return I;
	case 'M': case 'm': return M;
	case 'R': case 'r': return R;
	case 'S': case 's': return S;
	case 'V': case 'v': return V;
	case 'W': case 'w': return W;
	case 'Y': case 'y': return Y;
	case 'H': case 'h': return H;
	case 'K': case 'k': return K;
	case 'D': case 'd': return D;
	case 'B': case 'b': return B;
	case 'N': case 'n': return N;
	default: throw "char_to_base: illegal base";
	}
}

// ---------------------------------------------------------------------------------------------
// Parameters: nuc_cruc_santa_lucia.cpp (values from the generated table) + nuc_cruc.cpp:226-300
struct Params {
	float H[2401], S[2401];
	float loop_S[513], bulge_S[513], hairpin_S[513];
	float init_H, init_S, at_H, at_S, sym_S, salt, asym_S, bulge_at_S;
	bool wc[NBP];
	Params()
	{
		memset(H, 0, sizeof(H));
		memset(S, 0, sizeof(S));
		for (int k = 0; k < TOR_NN_COUNT; k++) {
			H[TOR_NN[k][0]] = f32(TOR_NN[k][1]);
			S[TOR_NN[k][0]] = f32(TOR_NN[k][2]);
		}
		for (int i = 0; i < 513; i++) {
			loop_S[i] = f32(TOR_LOOP_S[i]);
			bulge_S[i] = f32(TOR_BULGE_S[i]);
			hairpin_S[i] = f32(TOR_HAIRPIN_S[i]);
		}
		init_H = f32(TOR_INIT_H);
		init_S = f32(TOR_INIT_S);
		at_H = f32(TOR_AT_CLOSING_H);
		at_S = f32(TOR_AT_CLOSING_S);
		sym_S = f32(TOR_SYMMETRY_S);
		salt = f32(TOR_SALT);
		asym_S = f32(TOR_ASYM_LOOP_S);
		bulge_at_S = f32(TOR_BULGE_AT_S);
		// nuc_cruc.cpp:229-238
		memset(wc, 0, sizeof(wc));
		wc[P_AT] = wc[P_TA] = wc[P_CG] = wc[P_GC] = true;
		for (int b = A; b <= I; b++) wc[b * NB + I] = wc[I * NB + b] = true;
	}
};
static const Params &params()
{
	static const Params p;
	return p;
}

// nuc_cruc.cpp:340-487 (update_dp_param)
static void make_delta_g(float T, float na, int32_t *dg)
{
	const Params &P = params();
	// supplementary parameters nuc_cruc.cpp:271-300
	const float LOOP_H = -5.779f, LOOP_S = -2.330e-2f, BULGE_H = 5.247e-1f, BULGE_S = 3.318e-4f;
	const float TM_AT_H = -4.474f, TM_AT_S = -2.091e-2f, TM_GC_H = -3.000f, TM_GC_S = -1.318e-2f;
	const float TMM_H = -2.421f, TMM_S = -1.180e-2f;
	const float salt_correction = P.salt * logf(na);
	const float loop_sc = salt_correction * 3.08f;
	const float bulge_sc = salt_correction * 0.69f;
	const float term_match_sc = salt_correction * 0.56f;
	const float term_mismatch_sc = salt_correction * 1.31f;
#define SCALE(X) (int32_t((X) * 10000.0f)) /* NC_SCORE_SCALE, nuc_cruc.h:166 */
	for (int i = 0; i < 2401; i++) dg[i] = SCALE(P.H[i] - T * (P.S[i] + salt_correction));

	for (int i = A; i <= I; i++)
		for (int j = A; j <= I; j++) {
			const int curr = i * NB + j;
			for (int k = A; k <= I; k++) {
				const int prev1 = k * NB + GAP, prev2 = GAP * NB + k;
				int32_t v;
				if (P.wc[curr]) {
					if (curr == P_AT || curr == P_TA) v = SCALE(TM_AT_H - T * (TM_AT_S + term_match_sc));
					else if (curr == P_GC || curr == P_CG) v = SCALE(TM_GC_H - T * (TM_GC_S + term_match_sc));
					else v = SCALE(TM_AT_H - T * (TM_AT_S + term_match_sc)); // inosine == AT (:289-290)
				} else {
					v = SCALE(TMM_H - T * (TMM_S + term_mismatch_sc));
				}
				v = std::max(0, v);
				dg[curr * NBP + prev1] = dg[prev1 * NBP + curr] = dg[curr * NBP + prev2] = dg[prev2 * NBP + curr] = v;
			}
			for (int k = A; k <= I; k++)
				for (int l = A; l <= I; l++) {
					const int prev = k * NB + l;
					if (!P.wc[curr] && !P.wc[prev]) dg[curr * NBP + prev] = std::max(0, SCALE(LOOP_H - T * (LOOP_S + loop_sc)));
				}
		}
	for (int i = A; i <= I; i++)
		for (int j = A; j <= I; j++) {
			const int32_t v = std::max(0, SCALE(BULGE_H - T * (BULGE_S + bulge_sc)));
			dg[(i * NB + GAP) * NBP + (j * NB + GAP)] = v;
			dg[(GAP * NB + i) * NBP + (GAP * NB + j)] = v;
		}
#undef SCALE
}

// ---------------------------------------------------------------------------------------------
// NucCruc restated (hetero-dimer path only)
static const uint8_t TR_M = 1, TR_IT = 2, TR_IQ = 4, TR_INVALID = 8; // nuc_cruc.h:62-65

struct Alignment { // nuc_cruc.h:343-484
	bool valid = false;
	float dH = 0, dS = 0, tm = 0;
	std::deque<Base> q, t;
	int fm_q = 0, fm_t = 0, lm_q = 0, lm_t = 0; // first_match / last_match (.first=query, .second=target)
};

enum Mode { HOMO_DIMER = 0, HETERO_DIMER, HAIRPIN }; // NucCruc::mode, nuc_cruc.h:494

struct Cell {
	int32_t M = -1, Iq = -1, It = -1;
	uint8_t Mt = TR_INVALID, Iqt = TR_INVALID, Itt = TR_INVALID;
};

struct Branch { // trace_branch, nuc_cruc.h:279-341; keyed by (cell, state) instead of the byte address
	int key;
	uint8_t mask, cur;
	bool next()
	{
		while ((cur = uint8_t(cur << 1)) < TR_INVALID)
			if (cur & mask) return true;
		return false;
	}
};

class Melt {
public:
	float Tk, na, log_na, ct; // Tk: do not name this `T` (would shadow Base::T)
	bool dangle5, dangle3;
	bool dinkelbach = false; // --dinkelbach: Tm by iterating the alignment at its own melting temperature
	int32_t dg[2401];
	std::vector<Base> query, target;
	Alignment cur;
	std::vector<int> max_cells;
	int32_t max_score;

	Melt(float T_, float na_, bool d5, bool d3) : Tk(T_), na(na_), ct(-1), dangle5(d5), dangle3(d3)
	{
		log_na = logf(na);
		make_delta_g(Tk, na, dg);
	}
	void set_query(const std::string &s)
	{
		query.clear();
		for (char c : s) query.push_back(char_to_base(c));
	}

	void temperature(float t) // NucCruc::temperature -> update_dp_param, nuc_cruc.cpp:340-487
	{
		Tk = t;
		make_delta_g(Tk, na, dg);
	}
	// the Dinkelbach loop shared by the three approximate_tm_* (nuc_cruc.cpp:2399-2432,2459-2498,2548-2593): align at
	// 0 C, then again and again at the Tm the last alignment gave, while its free energy at that temperature is negative
	// and still rising; the last pass stands.  The initial temperature (and its DP parameters) come back afterwards.
	template <class PASS> float iterate(PASS one_pass)
	{
		if (!dinkelbach) return one_pass();
		const float init_T = Tk;
		float q = -999999.9f, last_q = q, local_tm = 0.0f;
		temperature(273.15f); // NC_ZERO_C
		do {
			local_tm = one_pass();
			last_q = q;
			q = delta_G();
			temperature(273.15f + local_tm);
		} while ((q < 0.0) && (q > last_q));
		temperature(init_T);
		return local_tm;
	}
	float approximate_tm_heterodimer() // nuc_cruc.cpp:2397-2455
	{
		return iterate([this]() {
			cur = Alignment();
			align_dimer();
			for (int c : max_cells) enumerate(c, HETERO_DIMER); // tm_dimer, nuc_cruc.cpp:2533-2537
			return cur.tm;
		});
	}
	float approximate_tm_homodimer() // nuc_cruc.cpp:2457-2515: the query against itself
	{
		target = query;
		return iterate([this]() {
			cur = Alignment();
			align_dimer();
			for (int c : max_cells) enumerate(c, HOMO_DIMER);
			return cur.tm;
		});
	}
	float approximate_tm_hairpin() // nuc_cruc.cpp:2542-2616
	{
		target = query; // trace_back(m_q, m_q), nuc_cruc.cpp:1212
		return iterate([this]() {
			cur = Alignment();
			align_hairpin();
			for (int c : max_cells) enumerate_hairpin(c);
			return cur.tm;
		});
	}
	float delta_G() const { return cur.dH - Tk * cur.dS; } // nuc_cruc.h:1217

	// nuc_cruc_anchor.cpp:143-192
	unsigned anchor5() const
	{
		const int tl = int(target.size()), ql = int(query.size());
		unsigned a = 0;
		int qi = 0, ti = cur.fm_q + cur.fm_t;
		if (!cur.t.empty() && cur.t.front() == E) return a;
		if (!cur.q.empty() && cur.q.front() == E) ti--;
		if (ti >= tl) return a;
		while (true) {
			if (qi >= ql || ti < 0) return a;
			if (!is_complementary(query[qi], target[ti])) return a;
			a++, qi++, ti--;
		}
	}
	// nuc_cruc_anchor.cpp:249-298
	unsigned anchor3() const
	{
		const int tl = int(target.size()), ql = int(query.size());
		unsigned a = 0;
		int qi = ql - 1, ti = (cur.lm_q + cur.lm_t + 1) - ql;
		if (!cur.t.empty() && cur.t.back() == E) return a;
		if (!cur.q.empty() && cur.q.back() == E) ti++;
		if (ti >= tl || ti < 0) return a;
		while (true) {
			if (qi < 0 || ti >= tl) return a;
			if (!is_complementary(query[qi], target[ti])) return a;
			a++, qi--, ti++;
		}
	}
	unsigned num_gap() const // nuc_cruc.h:389-408
	{
		unsigned g = 0;
		for (Base b : cur.q) g += (b == GAP);
		for (Base b : cur.t) g += (b == GAP);
		return g;
	}
	unsigned num_mismatch() const // nuc_cruc.h:410-440
	{
		unsigned mm = 0, aligned = 0;
		for (size_t k = 0; k < cur.q.size(); k++) {
			if (!is_virtual(cur.q[k])) {
				if (!is_virtual(cur.t[k]) && !is_complementary(cur.q[k], cur.t[k])) mm++;
				aligned++;
			}
		}
		return mm + unsigned(query.size()) - aligned;
	}
	unsigned poly_degen() const // nuc_cruc.h:465-483
	{
		unsigned best = 0, run = 0;
		for (Base b : cur.t) {
			if (b >= M && b <= N) best = std::max(best, ++run);
			else run = 0;
		}
		return best;
	}
	std::string print() const // nuc_cruc_output.cpp:74-205
	{
		static const char *bm = "ACGTI$-MRSVWYHKDBN";
		const int ql = int(query.size()), tl = int(target.size());
		const int prefix = std::max(0, std::min(cur.fm_q, tl - 1 - cur.fm_t));
		const int suffix = std::max(0, std::min(ql - 1 - cur.lm_q, cur.lm_t));
		std::string s = "5' ";
		for (int i = 0; i < prefix; i++) s += bm[query[cur.fm_q - prefix + i]];
		for (Base b : cur.q) s += bm[b];
		for (int i = 0; i < suffix; i++) s += bm[query[cur.lm_q + 1 + i]];
		s += " 3'\n   ";
		for (int i = 0; i < prefix; i++)
			s += is_complementary(query[cur.fm_q - prefix + i], target[cur.fm_t + prefix - i]) ? ':' : ' ';
		for (size_t k = 0; k < cur.t.size(); k++) s += is_complementary(cur.t[k], cur.q[k]) ? '|' : ' ';
		for (int i = 0; i < suffix; i++)
			s += is_complementary(query[cur.lm_q + 1 + i], target[cur.lm_t - i - 1]) ? ':' : ' ';
		s += "\n3' ";
		for (int i = prefix; i > 0; i--) s += bm[target[cur.fm_t + i]];
		for (Base b : cur.t) s += bm[b];
		for (int i = 1; i <= suffix; i++) s += bm[target[cur.lm_t - i]];
		s += " 5'";
		return s;
	}

private:
	std::vector<Cell> dp;
	int stride = 0;

public: // (the fill and its score table are read by the restatement of the pre-filter's DP-score tier, tntor_filter_bound)
	int g(int prev, int curp) const { return dg[prev * NBP + curp]; }

	// nuc_cruc.cpp:492-696
	void align_dimer()
	{
		const int L = int(query.size()), W = int(target.size());
		stride = W + 1;
		dp.assign(size_t(L + 1) * stride, Cell());
		max_cells.clear();
		max_score = -1;
		for (int i = 1; i <= L; i++) {
			const Base qb = query[L - i];
			const Base pq = (i == 1) ? GAP : query[L - (i - 1)];
			for (int j = 1; j <= W; j++) {
				const Base tb = target[j - 1];
				const Base pt = (j == 1) ? GAP : target[j - 2];
				const Cell &Ac = dp[(i - 1) * stride + j - 1], &Bc = dp[(i - 1) * stride + j], &Cc = dp[i * stride + j - 1];
				Cell &X = dp[i * stride + j];

				int curp = bbp(tb, qb);
				const int dg1 = std::max(0, Ac.M) - g(bbp(pt, pq), curp);
				const int dg2 = std::max(0, Ac.Iq) - g(bbp(pt, GAP), curp);
				const int dg3 = std::max(0, Ac.It) - g(bbp(GAP, pq), curp);
				if (dg1 >= dg2) {
					if (dg1 >= dg3) {
						X.M = dg1;
						X.Mt = TR_M;
						if (dg1 == dg2) X.Mt |= TR_IQ;
						if (dg1 == dg3) X.Mt |= TR_IT;
					} else {
						X.M = dg3;
						X.Mt = TR_IT;
					}
				} else {
					if (dg2 >= dg3) {
						X.M = dg2;
						X.Mt = TR_IQ;
						if (dg2 == dg3) X.Mt |= TR_IT;
					} else {
						X.M = dg3;
						X.Mt = TR_IT;
					}
				}
				// insertion in query (gap in the query strand), :599-632
				curp = bbp(tb, GAP);
				int ins = std::max(0, Cc.M) - g(bbp(pt, qb), curp);
				int ext = std::max(0, Cc.Iq) - g(bbp(pt, GAP), curp);
				if (ins >= ext) {
					X.Iq = ins;
					X.Iqt = TR_M;
					if (ins == ext) X.Iqt |= TR_IQ;
				} else {
					X.Iq = ext;
					X.Iqt = TR_IQ;
				}
				// insertion in target, :634-666
				curp = bbp(GAP, qb);
				ins = std::max(0, Bc.M) - g(bbp(tb, pq), curp);
				ext = std::max(0, Bc.It) - g(bbp(GAP, pq), curp);
				if (ins >= ext) {
					X.It = ins;
					X.Itt = TR_M;
					if (ins == ext) X.Itt |= TR_IT;
				} else {
					X.It = ext;
					X.Itt = TR_IT;
				}
				// :670-691
				if (X.M >= max_score) {
					if (X.M > max_score) {
						max_score = X.M;
						max_cells.clear();
					}
					max_cells.push_back(i * stride + j);
				}
			}
		}
	}

private:
	// nuc_cruc.cpp:1409-1618
	void trace_back(int cell, std::deque<Branch> &stack, int &zero_count, Alignment &al)
	{
		const int L = int(query.size());
		int li = cell / stride, lj = cell % stride;
		al.fm_q = L - li;
		al.fm_t = lj - 1;
		int truncate_at_zero = 0;
		bool count_zeros = false;
		if (zero_count < 0) {
			zero_count = 0;
			count_zeros = true;
		} else {
			truncate_at_zero = zero_count--;
		}
		int key = -1;        // the static first_match byte (never a split)
		uint8_t mask = TR_M; // *match_ptr
		while (true) {
			bool valid = true;
			uint8_t local;
			if (((mask & 1) + ((mask >> 1) & 1) + ((mask >> 2) & 1)) > 1) { // PATH_SPLIT
				std::deque<Branch>::iterator it = stack.begin();
				for (; it != stack.end(); ++it)
					if (it->key == key) break;
				if (it == stack.end()) {
					Branch b;
					b.key = key;
					b.mask = mask;
					b.cur = (mask & TR_M) ? TR_M : ((mask & TR_IT) ? TR_IT : TR_IQ);
					stack.push_back(b);
					local = b.cur;
				} else {
					local = it->cur;
				}
			} else {
				local = mask;
			}
			const Cell &c = dp[li * stride + lj];
			const int cidx = li * stride + lj;
			switch (local) {
			case TR_M:
				if (li > L || lj < 1 || li < 1 /* Q7 fence */) {
					valid = false;
				} else {
					if (c.M < 0) valid = false;
					else if (c.M == 0) {
						if (count_zeros) zero_count++;
						else if (--truncate_at_zero == 0) valid = false;
					}
					al.q.push_back(query[L - li]);
					al.t.push_back(target[lj - 1]);
					al.lm_q = L - li;
					al.lm_t = lj - 1;
					mask = c.Mt;
					key = cidx * 4 + 0;
					li--, lj--;
				}
				break;
			case TR_IQ: // gap_target: GAP in the query strand
				if (lj < 1) {
					valid = false;
				} else {
					if (c.Iq < 0) valid = false;
					al.q.push_back(GAP);
					al.t.push_back(target[lj - 1]);
					al.lm_q = L - li + 1;
					al.lm_t = lj - 1;
					mask = c.Iqt;
					key = cidx * 4 + 1;
					lj--;
				}
				break;
			case TR_IT: // query_gap: GAP in the target strand
				if (li > L || li < 1 /* Q7 fence */) {
					valid = false;
				} else {
					if (c.It < 0) valid = false;
					al.q.push_back(query[L - li]);
					al.t.push_back(GAP);
					al.lm_q = L - li;
					al.lm_t = lj;
					mask = c.Itt;
					key = cidx * 4 + 2;
					li--;
				}
				break;
			default:
				throw "invalid_match in trace back";
			}
			if (!valid) break;
		}
	}

	// nuc_cruc.cpp:973-1170
	void enumerate(int cell, Mode mode)
	{
		const Params &P = params();
		const int ql = int(query.size()), tl = int(target.size());
		bool first_time = true;
		std::deque<Branch> stack;
		int zero_count = -1;
		unsigned trace_count = 0;
		float best_dg = cur.dH - Tk * cur.dS;
		while (true) {
			if (!first_time && stack.empty() && zero_count <= 0) break;
			if (16 < trace_count) break; // max_dp_path_enum = 16, :332,1005
			trace_count++;
			first_time = false;
			Alignment al;
			trace_back(cell, stack, zero_count, al);
			// trim frayed ends, :1022-1054
			while (!al.q.empty() && !P.wc[bbp(al.q.back(), al.t.back())]) {
				if (!is_virtual(al.q.back())) --al.lm_q;
				if (!is_virtual(al.t.back())) ++al.lm_t;
				al.q.pop_back();
				al.t.pop_back();
			}
			while (!al.q.empty() && !P.wc[bbp(al.q.front(), al.t.front())]) {
				if (!is_virtual(al.q.front())) ++al.fm_q;
				if (!is_virtual(al.t.front())) --al.fm_t;
				al.q.pop_front();
				al.t.pop_front();
			}
			// :1058-1066
			if (zero_count == 0 && !stack.empty()) {
				while (!stack.empty() && !stack.back().next()) stack.pop_back();
				zero_count = -1;
			}
			// dangling ends, :1088-1137
			if (dangle5 && (al.fm_q != 0 || al.fm_t != tl - 1)) {
				if (al.fm_q == 0) al.q.push_front(E);
				else al.q.push_front(query[--al.fm_q]);
				if (al.fm_t == tl - 1) al.t.push_front(E);
				else al.t.push_front(target[++al.fm_t]);
			}
			if (dangle3 && (al.lm_q != ql - 1 || al.lm_t != 0)) {
				if (al.lm_q == ql - 1) al.q.push_back(E);
				else al.q.push_back(query[++al.lm_q]);
				if (al.lm_t == 0) al.t.push_back(E);
				else al.t.push_back(target[--al.lm_t]);
			}
			if (al.q.size() < 3) continue; // :1148
			if (evaluate(al, mode)) {
				const float local_dg = al.dH - Tk * al.dS;
				if (!cur.valid || local_dg < best_dg) { // :1160-1167
					cur = al;
					cur.valid = true;
					best_dg = local_dg;
				}
			}
		}
	}

	// nuc_cruc.cpp:771-971.  Stem rows i = 1..L-4 (reversed query), columns j = 0..L-4-i (forward query);
	// cell (i, j+1) pairs query[L-i] with query[j].
	void align_hairpin()
	{
		const int L = int(query.size());
		if (L == 0) throw ":NucCruc::align_hairpin: Empty query sequence";
		const int max_stem = L - 4; // steric_limit = 3 + 1
		stride = L + 1;
		dp.assign(size_t(L + 1) * stride, Cell());
		max_cells.clear();
		max_score = -1;
		for (int i = 1; i <= max_stem; i++) {
			const Base qb = query[L - i];
			const Base pq = (i == 1) ? GAP : query[L - (i - 1)];
			const int upper_j = max_stem - (i - 1);
			for (int j = 0; j < upper_j; j++) {
				const Base tb = query[j];
				const Base pt = (j == 0) ? GAP : query[j - 1];
				const Cell &Ac = dp[(i - 1) * stride + j], &Bc = dp[(i - 1) * stride + j + 1], &Cc = dp[i * stride + j];
				Cell &X = dp[i * stride + j + 1];
				int curp = bbp(tb, qb);
				const int dg1 = std::max(0, Ac.M) - g(bbp(pt, pq), curp);
				const int dg2 = std::max(0, Ac.Iq) - g(bbp(pt, GAP), curp);
				const int dg3 = std::max(0, Ac.It) - g(bbp(GAP, pq), curp);
				if (dg1 >= dg2) {
					if (dg1 >= dg3) {
						X.M = dg1;
						X.Mt = TR_M;
						if (dg1 == dg2) X.Mt |= TR_IQ;
						if (dg1 == dg3) X.Mt |= TR_IT;
					} else {
						X.M = dg3;
						X.Mt = TR_IT;
					}
				} else {
					if (dg2 >= dg3) {
						X.M = dg2;
						X.Mt = TR_IQ;
						if (dg2 == dg3) X.Mt |= TR_IT;
					} else {
						X.M = dg3;
						X.Mt = TR_IT;
					}
				}
				curp = bbp(tb, GAP);
				int ins = std::max(0, Cc.M) - g(bbp(pt, qb), curp);
				int ext = std::max(0, Cc.Iq) - g(bbp(pt, GAP), curp);
				if (ins >= ext) {
					X.Iq = ins;
					X.Iqt = TR_M;
					if (ins == ext) X.Iqt |= TR_IQ;
				} else {
					X.Iq = ext;
					X.Iqt = TR_IQ;
				}
				curp = bbp(GAP, qb);
				ins = std::max(0, Bc.M) - g(bbp(tb, pq), curp);
				ext = std::max(0, Bc.It) - g(bbp(GAP, pq), curp);
				if (ins >= ext) {
					X.It = ins;
					X.Itt = TR_M;
					if (ins == ext) X.Itt |= TR_IT;
				} else {
					X.It = ext;
					X.Itt = TR_IT;
				}
				if (X.M >= max_score) {
					if (X.M > max_score) {
						max_score = X.M;
						max_cells.clear();
					}
					max_cells.push_back(i * stride + j + 1);
				}
			}
		}
	}

	// find_loop_index, nuc_cruc.cpp:2619-2869: exact match of the 5 (tri-loop) or 6 (tetra-loop) bases starting
	// at the 5' closing base against the special-loop list; bases print through "ACGTE" (inosine never matches)
	int find_loop_index(int start, int len) const
	{
		char key[8] = {0};
		for (int k = 0; k < len; k++) {
			const int b = query[start + k];
			key[k] = b <= 4 ? "ACGTE"[b] : '?';
		}
		for (int i = 0; i < TOR_HAIRPIN_SPECIAL_COUNT; i++)
			if (!strcmp(TOR_HAIRPIN_SPECIAL[i].seq, key)) return i;
		return -1;
	}

	// evaluate_hairpin_alignment, nuc_cruc.cpp:2301-2394
	bool evaluate_hairpin(Alignment &al) const
	{
		const Params &P = params();
		const int last_3 = al.fm_q, last_5 = al.fm_t;
		const unsigned loop_len = unsigned(last_3 - last_5 - 1);
		al.dH = 0.0f;
		al.dS = 0.0f;
		al.dS += P.hairpin_S[loop_len];
		const int last_pair = bbp(query[last_5], query[last_3]);
		if (loop_len == 3) {
			const int li = find_loop_index(last_5, 5);
			if (li >= 0) {
				al.dH += f32(TOR_HAIRPIN_SPECIAL[li].h);
				al.dS += f32(TOR_HAIRPIN_SPECIAL[li].s);
			}
			if (last_pair == P_AT || last_pair == P_TA) al.dS += P.bulge_at_S;
		} else {
			if (loop_len == 4) {
				const int li = find_loop_index(last_5, 6);
				if (li >= 0) {
					al.dH += f32(TOR_HAIRPIN_SPECIAL[li].h);
					al.dS += f32(TOR_HAIRPIN_SPECIAL[li].s);
				}
			}
			const int curp = bbp(query[last_5 + 1], query[last_3 - 1]);
			al.dH += P.H[last_pair * NBP + curp]; // param_hairpin_terminal == param (nuc_cruc_santa_lucia.cpp:594-603)
			al.dS += P.S[last_pair * NBP + curp];
		}
		return evaluate(al, HAIRPIN);
	}

	// enumerate_hairpin_alignments, nuc_cruc.cpp:1172-1407
	void enumerate_hairpin(int cell)
	{
		const Params &P = params();
		const int ql = int(query.size());
		bool first_time = true;
		std::deque<Branch> stack;
		int zero_count = -1;
		unsigned trace_count = 0;
		float best_dg = cur.dH - Tk * cur.dS;
		auto consider = [&](Alignment &al) {
			if (evaluate_hairpin(al)) {
				const float local_dg = al.dH - Tk * al.dS;
				if (!cur.valid || local_dg < best_dg) {
					cur = al;
					cur.valid = true;
					best_dg = local_dg;
				}
			}
		};
		while (true) {
			if (!first_time && stack.empty() && zero_count <= 0) break;
			if (16 < trace_count) break;
			trace_count++;
			first_time = false;
			Alignment al;
			trace_back(cell, stack, zero_count, al);
			while (!al.q.empty() && !P.wc[bbp(al.q.back(), al.t.back())]) {
				if (!is_virtual(al.q.back())) --al.lm_q;
				if (!is_virtual(al.t.back())) ++al.lm_t;
				al.q.pop_back();
				al.t.pop_back();
			}
			while (!al.q.empty() && !P.wc[bbp(al.q.front(), al.t.front())]) {
				if (!is_virtual(al.q.front())) ++al.fm_q;
				if (!is_virtual(al.t.front())) --al.fm_t;
				al.q.pop_front();
				al.t.pop_front();
			}
			if (zero_count == 0 && !stack.empty()) {
				while (!stack.empty() && !stack.back().next()) stack.pop_back();
				zero_count = -1;
			}
			if (al.q.size() >= 3) consider(al); // :1265
			// dangling end at the open end of the stem, always attached (:1307-1324)
			if (al.lm_t != 0 || al.lm_q != ql - 1) {
				if (al.lm_t == 0) al.t.push_back(E);
				else al.t.push_back(query[--al.lm_t]);
				if (al.lm_q == ql - 1) al.q.push_back(E);
				else al.q.push_back(query[++al.lm_q]);
			}
			const size_t align_size = al.q.size();
			if (align_size < 3) continue;
			consider(al); // :1333
			if (align_size <= 3) continue;
			// try without an A-T closing pair (:1356-1405)
			const int lp = bbp(query[al.fm_t], query[al.fm_q]);
			if (lp == P_GC || lp == P_CG) continue;
			al.fm_q++;
			al.fm_t--;
			al.q.pop_front();
			al.t.pop_front();
			consider(al);
		}
	}

	// nuc_cruc.cpp:2888-2905
	bool has_AT_initiation(const Alignment &al, int k) const
	{
		int i = k;
		do {
			i--;
		} while (i != 0 && (al.q[i] == GAP || al.t[i] == GAP));
		const int p = bbp(al.q[i], al.t[i]);
		return p == P_AT || p == P_TA;
	}

	// nuc_cruc.cpp:1620-2299
	bool evaluate(Alignment &al, Mode mode) const
	{
		const Params &P = params();
		const float *Hh = P.H, *Ss = P.S;
		int terminal = P__, lastlast = P__, last = P__, curp = P__;
		if (mode != HAIRPIN) { // :1627-1633
			al.dH = P.init_H;
			al.dS = P.init_S + (mode == HOMO_DIMER ? P.sym_S : 0.0f);
		}
		unsigned nqg = 0, ntg = 0, nmm = 0, num_base = 0;
		bool terminal_5 = false;
		const int n = int(al.q.size());

		curp = bbp(al.q[0], al.t[0]);
		if (P.wc[curp]) {
			terminal_5 = true;
			if (curp == P_AT || curp == P_TA) {
				al.dH += P.at_H;
				al.dS += P.at_S;
			}
		}
		num_base += is_virtual(al.q[0]) ? 0 : 1;
		num_base += is_virtual(al.t[0]) ? 0 : 1;

		for (int k = 1; k < n; k++) {
			const Base qk = al.q[k], tk = al.t[k];
			lastlast = last;
			last = curp;
			curp = bbp(qk, tk);
			const bool align_start = (k == 1), align_stop = (k == n - 1);
			const bool in_loop = (qk == GAP) || (tk == GAP) || (!P.wc[last] && !P.wc[curp]);

			if (!in_loop) {
				if (align_start && !P.wc[last] && non_virtual_pair(last)) { // frayed start, :1726-1745
					const Base tq = Base(last / NB), tt = Base(last % NB);
					int tmp = bbp(tq, E);
					al.dH += Hh[tmp * NBP + curp];
					al.dS += Ss[tmp * NBP + curp];
					tmp = bbp(E, tt);
					al.dH += Hh[tmp * NBP + curp];
					al.dS += Ss[tmp * NBP + curp];
				} else if (align_stop && !P.wc[curp] && non_virtual_pair(curp)) { // frayed stop, :1747-1763
					int tmp = bbp(qk, E);
					al.dH += Hh[last * NBP + tmp];
					al.dS += Ss[last * NBP + tmp];
					tmp = bbp(E, tk);
					al.dH += Hh[last * NBP + tmp];
					al.dS += Ss[last * NBP + tmp];
				} else {
					al.dH += Hh[last * NBP + curp];
					al.dS += Ss[last * NBP + curp];
				}
				num_base += is_virtual(qk) ? 0 : 1;
				num_base += is_virtual(tk) ? 0 : 1;
			}

			if (P.wc[curp] || curp == P_EE) { // :1794
				terminal = curp;
				if (!terminal_5) {
					terminal_5 = true;
					if (curp == P_AT || curp == P_TA) {
						al.dH += P.at_H;
						al.dS += P.at_S;
					}
				}
				const unsigned max_gap = std::max(nqg, ntg);
				if (nmm > 1 || (max_gap > 0 && nmm == 1)) { // internal loop, :1824
					const unsigned gap_diff = (nqg > ntg) ? nqg - ntg : ntg - nqg;
					const unsigned loop_size = nmm * 2 + gap_diff;
					if (loop_size == 2 && (last == P_GT || last == P_TG) && (lastlast == P_GT || lastlast == P_TG)) {
						al.dH += Hh[lastlast * NBP + last];
						al.dS += Ss[lastlast * NBP + last];
						num_base += 2;
					} else {
						al.dS += P.loop_S[loop_size];
						al.dS += gap_diff * P.asym_S;
						int rhs_q = k - 1, rhs_t = k - 1;
						al.dH -= Hh[last * NBP + curp];
						al.dS -= Ss[last * NBP + curp];
						if (!has_gap(last)) { // right terminal mismatch, :1923-1992
							al.dH += Hh[last * NBP + curp]; // param_loop_terminal == param (santa_lucia.cpp:591-595)
							al.dS += Ss[last * NBP + curp];
						} else {
							int mm = P__;
							if (last / NB == GAP) {
								while (true) {
									if (!is_virtual(al.q[rhs_q])) {
										mm = bbp(al.q[rhs_q], Base(last % NB));
										break;
									}
									if (rhs_q == 0) break;
									rhs_q--;
								}
							} else {
								while (true) {
									if (!is_virtual(al.t[rhs_t])) {
										mm = bbp(Base(last / NB), al.t[rhs_t]);
										break;
									}
									if (rhs_t == 0) break;
									rhs_t--;
								}
							}
							al.dH += Hh[mm * NBP + curp];
							al.dS += Ss[mm * NBP + curp];
						}
						// left terminal mismatch, :2000-2077
						int lhs_q = k - 1, lhs_t = k - 1;
						while (true) {
							const int pm = bbp(al.q[lhs_q], al.t[lhs_t]);
							if (P.wc[pm]) {
								lhs_q++, lhs_t++;
								if (al.q[lhs_q] != GAP && al.t[lhs_t] != GAP) {
									const int mm = bbp(al.q[lhs_q], al.t[lhs_t]);
									al.dH -= Hh[pm * NBP + mm];
									al.dS -= Ss[pm * NBP + mm];
								} else {
									num_base += 2;
									while (al.q[lhs_q] == GAP) lhs_q++;
									while (al.t[lhs_t] == GAP) lhs_t++;
								}
								const int mm = bbp(al.q[lhs_q], al.t[lhs_t]);
								al.dH += Hh[pm * NBP + mm];
								al.dS += Ss[pm * NBP + mm];
								break;
							}
							if (lhs_q == 0) break;
							lhs_q--, lhs_t--;
						}
						if (rhs_q != lhs_q) num_base++;
						if (rhs_t != lhs_t) num_base++;
					}
				} else if (nqg || ntg) { // bulge, :2105-2185
					const unsigned bulge = std::max(nqg, ntg);
					if (bulge == 1) {
						al.dH += Hh[lastlast * NBP + curp];
						al.dS += Ss[lastlast * NBP + curp];
					}
					al.dS += P.bulge_S[bulge];
					if (bulge != 1 && (qk == A || qk == T)) al.dS += P.bulge_at_S;
					if (bulge != 1 && has_AT_initiation(al, k)) al.dS += P.bulge_at_S;
				}
				nqg = ntg = nmm = 0;
			} else {
				nmm += (!is_virtual(qk) && !is_virtual(tk)) ? 1 : 0;
			}
			nqg += (qk == GAP) ? 1 : 0;
			ntg += (tk == GAP) ? 1 : 0;
		}
		if (terminal == P_AT || terminal == P_TA) {
			al.dH += P.at_H;
			al.dS += P.at_S;
		}
		if (al.dH >= 0.0f) return false; // :2253
		al.dS += P.salt * (0.5f * num_base - 1) * log_na; // :2270
		const float tm = (mode == HAIRPIN) ? al.dH / al.dS - 273.15f // :2286-2289, no strand concentration
		                                   : al.dH / (1.9872e-3f * logf(ct * 1.0f) + al.dS) - 273.15f; // :2291
		al.tm = std::max(0.0f, tm);
		return true;
	}
};

// ---------------------------------------------------------------------------------------------
// Seeds: DNAHash restated (seq_hash.h)
struct Seeder {
	int k;
	std::vector<std::vector<uint32_t>> table; // word -> ascending target positions (counting sort, :524-642)
	std::vector<uint32_t> touched;
	explicit Seeder(int k_) : k(k_), table(size_t(1) << (2 * k_)) {}
	void hash(const uint8_t *seq, uint32_t len)
	{
		for (uint32_t w : touched) table[w].clear();
		touched.clear();
		const uint32_t wmask = (1u << (2 * k)) - 1;
		uint16_t word = 0;
		size_t run = 0;
		for (uint32_t i = 0; i < len; i++) {
			const uint8_t b = seq[i] & 3;             // Q1: every code is hashed as its low two bits (:571-583)
			word = uint16_t((word << 2) | b);
			run = (b <= 3) ? run + 1 : 0;
			if (run >= size_t(k)) {
				std::vector<uint32_t> &v = table[word & wmask];
				if (v.empty()) touched.push_back(word & wmask);
				v.push_back(i + 1 - k);
			}
		}
	}
	// build_word_list<std::string>, seq_hash.h:287-374
	std::vector<uint16_t> words(const std::string &oligo, bool complement) const
	{
		std::vector<uint16_t> w;
		const uint32_t wmask = (1u << (2 * k)) - 1;
		if (size_t(k) > oligo.size()) return w;
		uint16_t word = 0;
		size_t run = 0;
		const long n = long(oligo.size());
		for (long s = 0; s < n; s++) {
			const char c = complement ? oligo[n - 1 - s] : oligo[s];
			int b = -1;
			switch (c) {
			case 'A': case 'a': b = complement ? 3 : 0; break;
			case 'T': case 't': b = complement ? 0 : 3; break;
			case 'C': case 'c': b = complement ? 2 : 1; break;
			case 'G': case 'g': b = complement ? 1 : 2; break;
			default: break;
			}
			++run;
			if (b >= 0) word = uint16_t((word << 2) | b);
			else run = 0;
			if (run >= size_t(k)) w.push_back(word & wmask);
		}
		return w;
	}
};

struct Info { // oligo_info, tntblast.h:145-243
	int loc5 = 0, loc3 = 0;
	float tm = -1, dH = -1, dS = -1;
	unsigned a5 = 0, a3 = 0, mm = 0, gap = 0;
	std::string align;
	unsigned qloc = 0, tloc = 0;
	uint8_t mask = 0;
};
enum { MF = 1, MR = 2, MP = 4, MPLUS = 8, MMINUS = 16, MVALID = 32 };

static bool info_less(const Info &a, const Info &b) // oligo_info::operator<, tntblast.h:230-242
{
	if (a.loc5 != b.loc5) return a.loc5 < b.loc5;
	if (a.loc3 != b.loc3) return a.loc3 < b.loc3;
	return a.tm > b.tm;
}

// match_oligo_to_{minus,plus}_strand, bind_oligo.cpp:84-122
static void match_oligo(std::list<Info> &info, const Seeder &h, const std::string &oligo, uint8_t mask, bool plus)
{
	std::list<Info> hits;
	const std::vector<uint16_t> w = h.words(oligo, plus);
	for (size_t wi = 0; wi < w.size(); wi++)
		for (uint32_t pos : h.table[w[wi]]) {
			Info o;
			o.qloc = unsigned(wi); // offset() == index in the word list (Q2)
			o.tloc = pos;
			o.mask = uint8_t(mask | (plus ? MPLUS : MMINUS));
			hits.push_back(o);
		}
	hits.sort([](const Info &a, const Info &b) { return (int(a.qloc) - int(a.tloc)) < (int(b.qloc) - int(b.tloc)); });
	hits.unique([](const Info &a, const Info &b) { return (int(a.qloc) - int(a.tloc)) == (int(b.qloc) - int(b.tloc)); });
	info.merge(hits, info_less);
}

struct Thresh {
	float min_tm, max_tm, min_dg, max_dg;
	unsigned clamp5, clamp3, max_gap, max_mm, max_poly;
};

// window load: bind_oligo.cpp:515-592 (minus: complement + push_front) / :1218-1290 (plus)
static void load_window(Melt &m, const uint8_t *seq, unsigned start, unsigned stop, bool minus)
{
	static const Base comp[16] = {T, G, C, A, I, K, Y, S, B, W, R, D, M, H, V, N};
	static const Base same[16] = {A, C, G, T, I, M, R, S, V, W, Y, H, K, D, B, N};
	m.target.clear();
	std::deque<Base> d;
	for (unsigned i = start; i < stop; i++) {
		const uint8_t b = seq[i];
		if (b > 15) continue; // Q6: unknown / gap codes are skipped
		if (minus) d.push_front(comp[b]);
		else d.push_back(same[b]);
	}
	m.target.assign(d.begin(), d.end());
}

// BindCacheKey / BindCacheValue, tntblast.h:247-324: one map per target and strand (tntblast_local.cpp:371-372, cleared
// per target :505-506), keyed by the oligo STRING and the window -- not by the strand concentration or the thresholds of
// whoever evaluated it first.
struct CacheKey {
	std::string oligo;
	unsigned start, stop;
	bool operator==(const CacheKey &o) const { return oligo == o.oligo && start == o.start && stop == o.stop; }
};
struct CacheKeyHash {
	size_t operator()(const CacheKey &k) const { return std::hash<std::string>()(k.oligo) ^ ((size_t(k.start) << 32) | size_t(k.stop)); }
};
struct CacheVal { // what the cascade had computed when it stopped; the rest stays zero / empty (bind_oligo.cpp:600-714)
	float tm = 0, dg = 0, dH = 0, dS = 0;
	unsigned a5 = 0, a3 = 0, mm = 0, gap = 0, poly = 0;
	int t5 = 0, t3 = 0;
	std::string align;
	bool no_alignment = false; // the Q11 fence of this file: a window without a valid alignment stays "no bind" on a hit too
};
typedef std::unordered_map<CacheKey, CacheVal, CacheKeyHash> BindCache;
struct Caches {
	BindCache plus, minus;
	void clear() { BindCache().swap(plus), BindCache().swap(minus); }
};
static bool g_bind_cache = true; // tntor_set_bind_cache

// One seed through NucCruc + the threshold cascade, bind_oligo.cpp:502-748 / :1202-1446.
// Returns the cascade status (0 = bound) and fills `out` / `b`.  `cache` (with the oligo string): the per-target memo of
// the strand; a hit re-applies the thresholds to the memo (bind_oligo.cpp:396-428 / :763-803) instead of evaluating.
static int bind_one(Melt &m, const uint8_t *seq, unsigned seq_len, unsigned qloc, unsigned tloc, bool minus,
                    const Thresh &th, tnt_bind *b, Info *out, BindCache *cache = nullptr, const std::string *oligo = nullptr)
{
	const unsigned window = unsigned(m.query.size());
	const unsigned target_length = window + 2 * TNT_FLANK;
	const unsigned start = unsigned(std::max(int(tloc) - int(qloc + TNT_FLANK), 0));
	const unsigned stop = std::min(start + target_length, seq_len);
	if (!g_bind_cache) cache = nullptr;
	CacheKey key;
	if (cache) {
		key.oligo = *oligo, key.start = start, key.stop = stop;
		const BindCache::const_iterator hit = cache->find(key);
		if (hit != cache->end()) {
			const CacheVal &v = hit->second;
			if (v.tm < th.min_tm || v.tm > th.max_tm) return 1;
			if (v.dg < th.min_dg || v.dg > th.max_dg) return 2;
			if (v.no_alignment) return 8;
			if (v.a5 < th.clamp5 || v.a3 < th.clamp3) return 3;
			if (v.mm > th.max_mm) return 5;
			if (v.gap > th.max_gap) return 6;
			if (v.poly > th.max_poly) return 7;
			if (out) {
				out->loc5 = v.t5, out->loc3 = v.t3, out->tm = v.tm, out->dH = v.dH, out->dS = v.dS;
				out->a5 = v.a5, out->a3 = v.a3, out->mm = v.mm, out->gap = v.gap, out->align = v.align;
			}
			return 0;
		}
	}
	load_window(m, seq, start, stop, minus);

	const float tm = m.approximate_tm_heterodimer();
	const float dg = m.delta_G();
	int status = 0;
	if (tm < th.min_tm || tm > th.max_tm) status = 1;
	else if (dg < th.min_dg || dg > th.max_dg) status = 2;

	unsigned a5 = 0, a3 = 0, mm = 0, gap = 0, poly = 0;
	int loc5 = 0, loc3 = 0;
	std::string text;
	if (m.cur.valid) {
		a5 = m.anchor5();
		a3 = m.anchor3();
		mm = m.num_mismatch();
		gap = m.num_gap();
		poly = m.poly_degen();
		if (!status) {
			if (a5 < th.clamp5) status = 3;
			else if (a3 < th.clamp3) status = 4;
			else if (mm > th.max_mm) status = 5;
			else if (gap > th.max_gap) status = 6;
			else if (poly > th.max_poly) status = 7;
		}
		// alignment_range, nuc_cruc_anchor.cpp:379-396
		const int qf = m.cur.fm_q, ql = m.cur.lm_q, tf = m.cur.lm_t, tl = m.cur.fm_t;
		loc5 = loc3 = int(start);
		if (minus) { // bind_oligo.cpp:721-731
			loc5 += int(stop - start - 1) - tl;
			loc3 += int(stop - start - 1) - tf;
			loc5 -= qf;
			loc3 += int(window - 1) - ql;
		} else { // bind_oligo.cpp:1424-1434
			loc5 += tf;
			loc3 += tl;
			loc3 += qf;
			loc5 -= int(window - 1) - ql;
		}
		text = m.print();
		if (b) {
			b->q_first = qf, b->q_last = ql, b->t_first = tf, b->t_last = tl;
		}
	} else if (!status) {
		status = 8; // Q11 fence
	}
	if (b) {
		b->status = status;
		b->target_start = int(start), b->target_stop = int(stop);
		b->tm = tm, b->dH = m.cur.dH, b->dS = m.cur.dS, b->dg = dg;
		b->anchor5 = int(a5), b->anchor3 = int(a3), b->num_mm = int(mm), b->num_gap = int(gap), b->poly_degen = int(poly);
		b->loc5 = loc5, b->loc3 = loc3;
		b->valid = m.cur.valid;
		b->dp_max = m.max_score;
		b->n_max = int(m.max_cells.size());
		const size_t n = std::min(text.size(), size_t(TNT_ALIGN_TEXT - 1));
		memcpy(b->align, text.data(), n);
		b->align[n] = 0;
	}
	if (out && status == 0) {
		out->loc5 = loc5, out->loc3 = loc3, out->tm = tm, out->dH = m.cur.dH, out->dS = m.cur.dS;
		out->a5 = a5, out->a3 = a3, out->mm = mm, out->gap = gap, out->align = text;
	}
	if (cache) { // the memo holds what the cascade had looked at when it stopped (bind_oligo.cpp:600-714, :750-757)
		CacheVal v;
		v.tm = tm;
		if (status != 1) v.dg = dg;
		if (status == 8) v.no_alignment = true;
		if (status == 0 || (status >= 3 && status <= 7)) {
			v.a5 = a5;
			if (status != 3) v.a3 = a3;
			if (status == 0 || status >= 5) v.mm = mm;
			if (status == 0 || status >= 6) v.gap = gap;
			if (status == 0 || status == 7) v.poly = poly;
		}
		if (status == 0) v.dH = m.cur.dH, v.dS = m.cur.dS, v.t5 = loc5, v.t3 = loc3, v.align = text;
		(*cache)[key] = v;
	}
	return status;
}

// sort_by_bound_match, bind_oligo.cpp:49-82
static bool bound_less(const Info &a, const Info &b)
{
	if (a.loc5 != b.loc5) return a.loc5 < b.loc5;
	if (a.loc3 != b.loc3) return a.loc3 < b.loc3;
	if (a.tm == b.tm) {
		if (a.mm == b.mm) return a.align.size() > b.align.size();
		return a.mm > b.mm;
	}
	return a.tm > b.tm;
}

// mask overload, bind_oligo.cpp:456-827 / :1159-1530
static void bind_mask(std::list<Info> &info, uint8_t oligo_mask, const uint8_t *seq, unsigned seq_len, Melt &m,
                      bool minus, const Thresh &th, Caches *cc = nullptr, const std::string *oligo = nullptr)
{
	BindCache *cache = cc ? (minus ? &cc->minus : &cc->plus) : nullptr;
	std::list<Info> curr;
	const uint8_t want = uint8_t(oligo_mask | (minus ? MMINUS : MPLUS));
	for (std::list<Info>::iterator it = info.begin(); it != info.end();) {
		if ((it->mask & want) != want) {
			++it;
			continue;
		}
		curr.push_front(*it);
		it = info.erase(it);
		Info &o = curr.front();
		if (bind_one(m, seq, seq_len, o.qloc, o.tloc, minus, th, nullptr, &o, cache, oligo) != 0) curr.pop_front();
	}
	if (curr.empty()) return;
	curr.sort(bound_less);
	info.push_back(curr.front());
	curr.pop_front();
	while (!curr.empty()) {
		if (info.back().loc5 != curr.front().loc5 || info.back().loc3 != curr.front().loc3) info.push_back(curr.front());
		curr.pop_front();
	}
}

// hash overload, bind_oligo.cpp:124-454 / :829-1157
static void bind_hash(std::list<Info> &info, const Seeder &h, const uint8_t *seq, unsigned seq_len,
                      const std::string &oligo, Melt &m, bool minus, const Thresh &th, Caches *cc = nullptr)
{
	BindCache *cache = cc ? (minus ? &cc->minus : &cc->plus) : nullptr;
	std::list<Info> seeds, hits;
	match_oligo(seeds, h, oligo, 0, !minus); // same seeds, same diagonal order (sort_by_delta / unique_by_delta)
	for (Info &s : seeds) {
		Info o;
		if (bind_one(m, seq, seq_len, s.qloc, s.tloc, minus, th, nullptr, &o, cache, &oligo) == 0) hits.push_back(o);
	}
	info.clear();
	if (hits.empty()) return;
	hits.sort(info_less);
	info.push_back(hits.front());
	hits.pop_front();
	while (!hits.empty()) {
		if (info.back().loc5 != hits.front().loc5 || info.back().loc3 != hits.front().loc3) info.push_back(hits.front());
		hits.pop_front();
	}
}

// sort_by_oligo_loc, amplicon_search.cpp:12-26 (Q5: mixed keys)
static bool oligo_loc_less(const Info &a, const Info &b)
{
	if (!(a.loc5 + a.loc3) || !(b.loc5 + b.loc3)) return a.tloc < b.tloc;
	if (a.loc5 == b.loc5) return a.loc3 < b.loc3;
	return a.loc5 < b.loc5;
}

// cull_oligo_match, amplicon_search.cpp:679-765
static std::pair<unsigned, unsigned> cull(std::list<Info> &l, unsigned max_len, bool has_probe, bool single_primer)
{
	const unsigned threshold = max_len + 50;
	std::pair<unsigned, unsigned> ret(0, 0);
	l.sort(oligo_loc_less);
	for (Info &i : l) i.mask &= uint8_t(~MVALID);
	for (std::list<Info>::iterator f = l.begin(); f != l.end(); ++f) {
		if (f->mask & (MPLUS | MP)) continue;
		std::list<Info>::iterator r = f;
		for (++r; r != l.end(); ++r) {
			if ((r->tloc - f->tloc) > threshold) break; // unsigned arithmetic, :707
			if (r->mask & (MMINUS | MP)) continue;
			if (!single_primer && ((f->mask & (MR | MF)) == (r->mask & (MR | MF)))) continue;
			if (has_probe) {
				std::list<Info>::iterator p = f;
				for (++p; p != r; ++p)
					if (p->mask & MP) {
						p->mask |= MVALID;
						f->mask |= MVALID;
						r->mask |= MVALID;
					}
			} else {
				f->mask |= MVALID;
				r->mask |= MVALID;
			}
		}
	}
	for (std::list<Info>::iterator i = l.begin(); i != l.end();) {
		if (i->mask & MVALID) {
			++i;
			if (i != l.end()) { // Q4 fence: the reference reads i->mask here even at end()
				ret.first += (i->mask & MMINUS) ? 1 : 0;
				ret.second += (i->mask & MPLUS) ? 1 : 0;
			}
			continue;
		}
		i = l.erase(i);
	}
	return ret;
}

struct Ctx {
	tnt_options opt;
	float fwd_strand, rev_strand;
};

static void fill_align(char *dst, const std::string &s)
{
	const size_t n = std::min(s.size(), size_t(TNT_ALIGN_TEXT - 1));
	memcpy(dst, s.data(), n);
	dst[n] = 0;
}

// hybrid_sig keeps its mismatch / gap / clamp counts in int8_t (hybrid_sig.h:121-164): a count beyond 127 -- a probe of
// several hundred bases -- comes out of the reference wrapped
static inline int wrap8(unsigned v) { return int(int8_t(uint8_t(v))); }

static tnt_hit blank_hit(const tnt_assay &A, int seq_index)
{
	tnt_hit h;
	memset(&h, 0, sizeof(h));
	h.assay_id = A.id, h.degen_id = A.degen_id, h.seq_index = seq_index;
	// hybrid_sig::init(), hybrid_sig.h:54-110
	h.forward_tm = h.reverse_tm = h.probe_tm = -1.0f;
	h.forward_dH = h.reverse_dH = h.probe_dH = 100.0f;
	h.forward_primer_clamp = h.reverse_primer_clamp = -1;
	h.forward_mm = h.forward_gap = h.reverse_mm = h.reverse_gap = h.probe_mm = h.probe_gap = -1;
	h.primer_strand = h.probe_strand = TNT_PLUS;
	h.has_primers = (A.forward && A.reverse) ? 1 : 0;
	h.has_probe = A.probe ? 1 : 0;
	h.forward_oligo = 0, h.reverse_oligo = 1;
	return h;
}

// The head of amplicon() (amplicon_search.cpp:96-128): the match list in its assembly order, the early exits, the first cull.
// Returns false where amplicon() returns before binding anything; `sc` = the strand counts that pick the strand bound first.
static bool first_cull(const tnt_options &o, const tnt_assay &A, const Seeder &h, std::list<Info> &ml, std::pair<unsigned, unsigned> &sc)
{
	const bool has_probe = A.probe != nullptr;
	const std::string F = A.forward, R = A.reverse, Pp = has_probe ? A.probe : "";
	match_oligo(ml, h, F, MF, false);
	match_oligo(ml, h, R, MR, false);
	const size_t n_minus = ml.size();
	if (n_minus == 0) return false;
	match_oligo(ml, h, F, MF, true);
	match_oligo(ml, h, R, MR, true);
	const size_t n_plus = ml.size();
	if (n_plus == n_minus) return false;
	if (has_probe) {
		match_oligo(ml, h, Pp, MP, false);
		match_oligo(ml, h, Pp, MP, true);
		if (ml.size() == n_plus) return false;
	}
	sc = cull(ml, o.max_len, has_probe, o.single_primer_pcr != 0);
	return true;
}

// amplicon(), amplicon_search.cpp:58-677
static void amplicon(const Ctx &cx, const tnt_assay &A, const Seeder &h, const uint8_t *seq, unsigned seq_len,
                     int seq_index, Melt &m, std::vector<tnt_hit> &out, Caches *cc = nullptr)
{
	const tnt_options &o = cx.opt;
	const bool has_probe = A.probe != nullptr;
	const bool single = o.single_primer_pcr != 0;
	const float fs = cx.fwd_strand / A.forward_degen, rs = cx.rev_strand / A.reverse_degen,
	            ps = o.probe_strand / A.probe_degen;
	const bool apply_mm = o.min_max_primer_clamp >= 0;
	const unsigned min_max_clamp = apply_mm ? unsigned(o.min_max_primer_clamp) : 0;
	const std::string F = A.forward, R = A.reverse, Pp = has_probe ? A.probe : "";

	std::list<Info> ml;
	std::pair<unsigned, unsigned> sc;
	if (!first_cull(o, A, h, ml, sc)) return;

	const Thresh pt = {o.min_primer_tm, o.max_primer_tm, o.min_primer_dg, o.max_primer_dg, 0u, unsigned(o.primer_clamp),
	                   unsigned(o.max_gap), unsigned(o.max_mismatch), unsigned(o.max_poly_degen)};
	struct Step { bool fwd; bool minus; };
	const Step minus_first[4] = {{true, true}, {false, true}, {true, false}, {false, false}};
	const Step plus_first[4] = {{true, false}, {false, false}, {true, true}, {false, true}};
	const Step *steps = (sc.first < sc.second) ? minus_first : plus_first; // :131
	for (int s = 0; s < 4; s++) {
		m.set_query(steps[s].fwd ? F : R);
		m.ct = steps[s].fwd ? fs : rs; // strand(c, 0) = c, nuc_cruc.h:890-910
		bind_mask(ml, steps[s].fwd ? MF : MR, seq, seq_len, m, steps[s].minus, pt, cc, steps[s].fwd ? &F : &R);
		if (s < 3) {
			// a cull follows the first three binds; the list-empty early return is absent after the
			// third bind of the minus-first branch only (:199) but present in the plus-first one (:286-290)
			cull(ml, o.max_len, has_probe, single);
			if (ml.empty() && !(steps == minus_first && s == 2)) return;
		}
	}
	if (has_probe) { // :311-353
		cull(ml, o.max_len, has_probe, single);
		if (ml.empty()) return;
		const Thresh prt = {o.min_probe_tm, o.max_probe_tm, o.min_probe_dg, o.max_probe_dg, unsigned(o.probe_clamp_5),
		                    unsigned(o.probe_clamp_3), unsigned(o.max_gap), unsigned(o.max_mismatch), unsigned(o.max_poly_degen)};
		m.set_query(Pp);
		m.ct = ps;
		bind_mask(ml, MP, seq, seq_len, m, true, prt, cc, &Pp);
		bind_mask(ml, MP, seq, seq_len, m, false, prt, cc, &Pp);
	}
	ml.sort(oligo_loc_less); // :356

	for (std::list<Info>::iterator f = ml.begin(); f != ml.end(); ++f) {
		if (f->mask & (MPLUS | MP)) continue;
		std::list<Info>::iterator r = f;
		for (++r; r != ml.end(); ++r) {
			if (r->mask & (MMINUS | MP)) continue;
			if (!single && ((f->mask & (MR | MF)) == (r->mask & (MR | MF)))) continue;
			if (f->loc3 >= r->loc5) continue;
			if ((r->loc3 - f->loc5 + 1) > int(o.max_len)) continue;
			if (apply_mm && std::max(f->a3, r->a3) <= min_max_clamp) continue;

			tnt_hit hit = blank_hit(A, seq_index);
			if ((f->mask & MR) && (r->mask & MR)) hit.forward_oligo = 1; // two reverse oligos, :453-457
			if ((f->mask & MF) && (r->mask & MF)) hit.reverse_oligo = 0; // two forward oligos, :459-463
			hit.primer_strand = (f->mask & MF) ? TNT_PLUS : TNT_MINUS;
			hit.amplicon_start = f->loc5;
			hit.amplicon_stop = r->loc3;
			const Info *fo = &*f, *ro = &*r;
			if ((f->mask & MR) && (r->mask & MF)) std::swap(fo, ro); // :478-483
			hit.forward_tm = fo->tm, hit.reverse_tm = ro->tm;
			hit.forward_dH = fo->dH, hit.reverse_dH = ro->dH;
			hit.forward_dS = fo->dS, hit.reverse_dS = ro->dS;
			hit.forward_mm = wrap8(fo->mm), hit.reverse_mm = wrap8(ro->mm);
			hit.forward_gap = wrap8(fo->gap), hit.reverse_gap = wrap8(ro->gap);
			hit.forward_primer_clamp = wrap8(fo->a3), hit.reverse_primer_clamp = wrap8(ro->a3);
			fill_align(hit.forward_align, fo->align);
			fill_align(hit.reverse_align, ro->align);

			if (has_probe) {
				std::list<Info>::iterator p = f;
				for (++p; p != r; ++p) {
					if (!(p->mask & MP)) continue;
					const int amp_start = f->loc5, amp_stop = r->loc3;
					if (!(p->loc5 >= amp_start && p->loc3 <= amp_stop)) continue;
					if ((p->mask & (MPLUS | MMINUS)) == (f->mask & (MPLUS | MMINUS))) {
						if (p->loc5 <= f->loc3) continue;
					} else {
						if (p->loc3 >= r->loc5) continue;
					}
					tnt_hit hp = hit;
					hp.probe_start = p->loc5, hp.probe_stop = p->loc3;
					hp.probe_tm = p->tm, hp.probe_dH = p->dH, hp.probe_dS = p->dS;
					hp.probe_mm = wrap8(p->mm), hp.probe_gap = wrap8(p->gap);
					hp.probe_strand = (p->mask & MPLUS) ? TNT_PLUS : TNT_MINUS;
					fill_align(hp.probe_align, p->align);
					out.push_back(hp);
				}
			} else {
				out.push_back(hit);
			}
		}
	}
}

// padlock(), padlock_search.cpp:62-361
static void padlock(const Ctx &cx, const tnt_assay &A, const Seeder &h, const uint8_t *seq, unsigned seq_len,
                    int seq_index, Melt &m, int max_len, std::vector<tnt_hit> &out, Caches *cc = nullptr)
{
	const tnt_options &o = cx.opt;
	const float fs = cx.fwd_strand / A.forward_degen, rs = cx.rev_strand / A.reverse_degen;
	const std::string F = A.forward, R = A.reverse;
	const Thresh up_t = {o.min_probe_tm, o.max_probe_tm, o.min_probe_dg, o.max_probe_dg, unsigned(o.probe_clamp_5), 0u,
	                     unsigned(o.max_gap), unsigned(o.max_mismatch), unsigned(o.max_poly_degen)};
	const Thresh down_t = {o.min_probe_tm, o.max_probe_tm, o.min_probe_dg, o.max_probe_dg, 0u, unsigned(o.probe_clamp_3),
	                       unsigned(o.max_gap), unsigned(o.max_mismatch), unsigned(o.max_poly_degen)};
	for (int pass = 0; pass < 2; pass++) {
		const bool minus = (pass == 0);
		std::list<Info> up, down;
		if (o.target_strand & (minus ? TNT_STRAND_MINUS : TNT_STRAND_PLUS)) {
			m.set_query(R);
			m.ct = rs;
			bind_hash(up, h, seq, seq_len, R, m, minus, up_t, cc);
			m.set_query(F);
			m.ct = fs;
			bind_hash(down, h, seq, seq_len, F, m, minus, down_t, cc);
		}
		for (const Info &u : up)
			for (const Info &d : down) {
				const int len = minus ? (u.loc5 - d.loc3 - 1) : (d.loc5 - u.loc3 - 1); // :147, :283
				if (len < 0 || len > max_len) continue;
				tnt_hit hit = blank_hit(A, seq_index);
				hit.primer_strand = minus ? TNT_MINUS : TNT_PLUS;
				hit.amplicon_start = minus ? d.loc5 : u.loc5;
				hit.amplicon_stop = minus ? u.loc3 : d.loc3;
				hit.forward_tm = d.tm, hit.reverse_tm = u.tm;
				hit.forward_dH = d.dH, hit.reverse_dH = u.dH;
				hit.forward_dS = d.dS, hit.reverse_dS = u.dS;
				hit.forward_mm = wrap8(d.mm), hit.reverse_mm = wrap8(u.mm);
				hit.forward_gap = wrap8(d.gap), hit.reverse_gap = wrap8(u.gap);
				hit.forward_primer_clamp = wrap8(d.a3);
				hit.reverse_primer_clamp = wrap8(u.a5);
				fill_align(hit.forward_align, d.align);
				fill_align(hit.reverse_align, u.align);
				out.push_back(hit);
			}
	}
}

// hybrid(), probe_search.cpp:67-230
static void hybrid(const Ctx &cx, const tnt_assay &A, const Seeder &h, const uint8_t *seq, unsigned seq_len,
                   int seq_index, Melt &m, std::vector<tnt_hit> &out)
{
	const tnt_options &o = cx.opt;
	const std::string Pp = A.probe;
	const Thresh th = {o.min_probe_tm, o.max_probe_tm, o.min_probe_dg, o.max_probe_dg, unsigned(o.probe_clamp_5),
	                   unsigned(o.probe_clamp_3), unsigned(o.max_gap), unsigned(o.max_mismatch), unsigned(o.max_poly_degen)};
	m.ct = o.probe_strand / A.probe_degen;
	m.set_query(Pp);
	for (int pass = 0; pass < 2; pass++) {
		const bool minus = (pass == 0);
		std::list<Info> bind;
		if (o.target_strand & (minus ? TNT_STRAND_MINUS : TNT_STRAND_PLUS)) bind_hash(bind, h, seq, seq_len, Pp, m, minus, th);
		for (const Info &b : bind) {
			tnt_hit hit = blank_hit(A, seq_index);
			hit.probe_tm = b.tm, hit.probe_dH = b.dH, hit.probe_dS = b.dS;
			hit.probe_mm = wrap8(b.mm), hit.probe_gap = wrap8(b.gap);
			hit.probe_start = b.loc5, hit.probe_stop = b.loc3;
			hit.probe_strand = minus ? TNT_MINUS : TNT_PLUS;
			fill_align(hit.probe_align, b.align);
			out.push_back(hit);
		}
	}
}

static std::string g_err;

} // namespace tor

using namespace tor;

extern "C" {

const char *tntor_last_error(void) { return g_err.c_str(); }

// The reference's per-target BindCache (a6) on (default) or off: off, every use of an oligo is evaluated on its own -- what the
// library computes for contexts it accepts, where the two agree (same strand concentration and thresholds for every use).
void tntor_set_bind_cache(int on) { g_bind_cache = on != 0; }
static int g_filter_tier = 1; // tntor_set_filter_tier: which bound tntor_filter_bound restates (1: the pre-filter's automaton, 2: the second tier's)
void tntor_set_filter_tier(int tier) { g_filter_tier = (tier == 2 || tier == 3 || tier == 30) ? tier : 1; }

int tntor_tables(const tnt_options *opt, int32_t *delta_g, float *param_H, float *param_S, float *loop_S, float *bulge_S,
                 float *hairpin_S, float *scalars)
{
	const Params &P = params();
	if (delta_g) make_delta_g(opt->target_t, opt->salt, delta_g);
	if (param_H) memcpy(param_H, P.H, sizeof(P.H));
	if (param_S) memcpy(param_S, P.S, sizeof(P.S));
	if (loop_S) memcpy(loop_S, P.loop_S, sizeof(P.loop_S));
	if (bulge_S) memcpy(bulge_S, P.bulge_S, sizeof(P.bulge_S));
	if (hairpin_S)
		for (int i = 0; i < 513; i++) hairpin_S[i] = f32(TOR_HAIRPIN_S[i]);
	if (scalars) {
		const float v[8] = {P.init_H, P.init_S, P.at_H, P.at_S, P.sym_S, P.salt, P.asym_S, P.bulge_at_S};
		memcpy(scalars, v, sizeof(v));
	}
	return 0;
}

int tntor_seeds(const tnt_options *opt, const char *const *oligos, int n_oligos, const uint8_t *codes, const uint64_t *offsets,
                size_t n_targets, uint32_t first_seq_index, tnt_seed **out, size_t *n_out)
{
	std::vector<tnt_seed> all;
	Seeder h(opt->hash_word_size);
	for (size_t t = 0; t < n_targets; t++) {
		const uint32_t len = uint32_t(offsets[t + 1] - offsets[t]);
		if (len < uint32_t(opt->hash_word_size)) continue; // tntblast_local.cpp:513
		h.hash(codes + offsets[t], len);
		for (int o = 0; o < n_oligos; o++)
			for (int s = 0; s < 2; s++) {
				std::list<Info> l;
				match_oligo(l, h, oligos[o], MF, s == 1);
				for (const Info &i : l) {
					tnt_seed sd = {int(first_seq_index + t), o, (s == 0) ? TNT_MINUS : TNT_PLUS, int(i.qloc), int(i.tloc)};
					all.push_back(sd);
				}
			}
	}
	*n_out = all.size();
	*out = (tnt_seed *)malloc(sizeof(tnt_seed) * std::max<size_t>(1, all.size()));
	memcpy(*out, all.data(), sizeof(tnt_seed) * all.size());
	return 0;
}

int tntor_bind_batch(const tnt_options *opt, const char *const *oligos, int n_oligos, const uint8_t *codes,
                     const uint64_t *offsets, size_t n_targets, uint32_t first_seq_index, const tnt_bind_req *req, size_t n,
                     tnt_bind *out)
{
	int rc = 0;
#pragma omp parallel
	{
		try {
			Melt m(opt->target_t, opt->salt, opt->allow_dangle_5 != 0, opt->allow_dangle_3 != 0);
			m.dinkelbach = opt->use_dinkelbach != 0;
#pragma omp for schedule(dynamic, 64)
			for (long k = 0; k < long(n); k++) {
				const tnt_bind_req &r = req[k];
				tnt_bind &b = out[k];
				memset(&b, 0, sizeof(b));
				const size_t t = size_t(r.seq_index) - first_seq_index;
				if (t >= n_targets || r.oligo < 0 || r.oligo >= n_oligos) {
					b.status = -1;
					continue;
				}
				const bool probe = r.role == 1;
				const Thresh th = {probe ? opt->min_probe_tm : opt->min_primer_tm, probe ? opt->max_probe_tm : opt->max_primer_tm,
				                   probe ? opt->min_probe_dg : opt->min_primer_dg, probe ? opt->max_probe_dg : opt->max_primer_dg,
				                   probe ? unsigned(opt->probe_clamp_5) : 0u,
				                   probe ? unsigned(opt->probe_clamp_3) : unsigned(opt->primer_clamp), unsigned(opt->max_gap),
				                   unsigned(opt->max_mismatch), unsigned(opt->max_poly_degen)};
				m.set_query(oligos[r.oligo]);
				m.ct = r.ct;
				bind_one(m, codes + offsets[t], unsigned(offsets[t + 1] - offsets[t]), unsigned(r.query_loc),
				         unsigned(r.target_loc), r.strand == TNT_MINUS, th, &b, nullptr);
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

int tntor_search(const tnt_options *opt, const tnt_assay *assays, size_t n_assays, const uint8_t *codes, const uint64_t *offsets,
                 size_t n_targets, uint32_t first_seq_index, tnt_hit **out, size_t *n_out, int n_threads)
{
	int rc = 0;
	std::vector<tnt_hit> all;
	Ctx cx;
	cx.opt = *opt;
	cx.fwd_strand = opt->asymmetric_strand_ratio * opt->primer_strand; // tntblast_local.cpp:232-234
	cx.rev_strand = opt->primer_strand;
	if (n_threads <= 0) n_threads = omp_get_max_threads();
#pragma omp parallel num_threads(n_threads)
	{
		try {
			Seeder h(opt->hash_word_size);
			Melt m(opt->target_t, opt->salt, opt->allow_dangle_5 != 0, opt->allow_dangle_3 != 0);
			m.dinkelbach = opt->use_dinkelbach != 0;
			std::vector<tnt_hit> local;
			Caches cc; // plus / minus strand memo of the target (tntblast_local.cpp:371-372)
#pragma omp for schedule(dynamic, 16)
			for (long t = 0; t < long(n_targets); t++) {
				const uint8_t *seq = codes + offsets[t];
				const unsigned len = unsigned(offsets[t + 1] - offsets[t]);
				if (len < unsigned(opt->hash_word_size)) continue;
				h.hash(seq, len);
				cc.clear(); // tntblast_local.cpp:505-506
				for (size_t a = 0; a < n_assays; a++) {
					const tnt_assay &A = assays[a];
					const int si = int(first_seq_index + t);
					if (A.forward && A.reverse) { // dispatch, tntblast_local.cpp:559-626
						if (opt->assay_format == TNT_ASSAY_PCR) amplicon(cx, A, h, seq, len, si, m, local, &cc);
						else if (opt->assay_format == TNT_ASSAY_PADLOCK) padlock(cx, A, h, seq, len, si, m, 0, local, &cc);
						else if (opt->assay_format == TNT_ASSAY_MIPS) padlock(cx, A, h, seq, len, si, m, opt->max_len, local, &cc);
					} else if (A.probe) {
						hybrid(cx, A, h, seq, len, si, m, local);
					}
				}
			}
#pragma omp critical
			all.insert(all.end(), local.begin(), local.end());
		} catch (const char *e) {
#pragma omp critical
			{
				g_err = e;
				rc = -1;
			}
		}
	}
	*n_out = all.size();
	*out = (tnt_hit *)malloc(sizeof(tnt_hit) * std::max<size_t>(1, all.size()));
	memcpy(*out, all.data(), sizeof(tnt_hit) * all.size());
	return rc;
}

// a4 at unit level: the seeds of every (target, PCR assay) that survive amplicon()'s early exits and the first
// cull_oligo_match, and the strand that is bound first (amplicon_search.cpp:96-131).  Records of six ints:
// seq_index, assay (position in `assays`), role (0 forward, 1 reverse, 2 probe), strand, query_loc, target_loc, and a
// seventh: 1 when the minus strand is bound first.
int tntor_first_cull(const tnt_options *opt, const tnt_assay *assays, size_t n_assays, const uint8_t *codes, const uint64_t *offsets,
                     size_t n_targets, uint32_t first_seq_index, int32_t **out, size_t *n_out)
{
	std::vector<int32_t> all;
	try {
		Seeder h(opt->hash_word_size);
		for (size_t t = 0; t < n_targets; t++) {
			const uint8_t *seq = codes + offsets[t];
			const unsigned len = unsigned(offsets[t + 1] - offsets[t]);
			if (len < unsigned(opt->hash_word_size)) continue;
			h.hash(seq, len);
			for (size_t a = 0; a < n_assays; a++) {
				const tnt_assay &A = assays[a];
				if (!(A.forward && A.reverse) || opt->assay_format != TNT_ASSAY_PCR) continue;
				std::list<Info> ml;
				std::pair<unsigned, unsigned> sc;
				if (!first_cull(*opt, A, h, ml, sc)) continue;
				for (const Info &i : ml) {
					const int32_t rec[7] = {int32_t(first_seq_index + t), int32_t(a), (i.mask & MF) ? 0 : ((i.mask & MR) ? 1 : 2),
					                        (i.mask & MPLUS) ? TNT_PLUS : TNT_MINUS, int32_t(i.qloc), int32_t(i.tloc), sc.first < sc.second ? 1 : 0};
					all.insert(all.end(), rec, rec + 7);
				}
			}
		}
	} catch (const char *e) {
		g_err = e;
		return -1;
	}
	*n_out = all.size() / 7;
	*out = (int32_t *)malloc(sizeof(int32_t) * std::max<size_t>(1, all.size()));
	memcpy(*out, all.data(), sizeof(int32_t) * all.size());
	return 0;
}

// a17: hairpin / homodimer Tm of one oligo and the heterodimer Tm of two (tntblast_local.cpp:656-686)
// ---------------------------------------------------------------------------------------------------------
// Checker of the product's bind pre-filter (thermonucleotideblast_amd/csrc/tnt_filter.hip.h): the same upper bound of
// -(dH - Tk dS) over all alignments of a window, restated in the order evaluate_alignment reads an alignment
// (nuc_cruc.cpp:1620-2299; the kernel sweeps the other way), on plain windows.  score[2k] = bound, score[2k+1] =
// threshold of request k (INT32_MIN: this class is never rejected), both in 1e-3 kcal/mol (the bound saturates at 32 767); INT32_MAX / INT32_MIN when the
// window holds a skipped code (such windows are never rejected).  tests/test_filter_bound.py holds it against
// tntor_bind_batch ("accepted => bound >= threshold") and tests/test_gpu_parity.py against the kernel (equal integers).
int tntor_filter_bound(const tnt_options *opt, const char *const *oligos, int n_oligos, const uint8_t *codes, const uint64_t *offsets,
                       size_t n_targets, uint32_t first_seq_index, const tnt_bind_req *req, size_t n, int32_t *score)
{
	using namespace tor;
	const Params &P = params();
	// 16-bit table units, as the kernel holds them: 1e-3 kcal/mol rounded up; "not possible" = a price of 16.384 kcal/mol;
	// sums saturate at 32 767 (a saturated bound is never rejected)
	const int NEG = -16384, SAT = 32767;
	const double UNITS = 1000.0;
	auto units = [&](double kcal) { return int(std::min(double(SAT), std::max(double(NEG), std::ceil(kcal * UNITS)))); };
	const double log_na = double(logf(opt->salt));
	bool ok = !opt->use_dinkelbach && log_na <= 0.0;
	for (int i = 0; i < 513; i++) ok = ok && P.loop_S[i] <= 0.0f && P.bulge_S[i] <= 0.0f;
	ok = ok && P.asym_S <= 0.0f && P.bulge_at_S <= 0.0f;
	for (int x = A; x <= I; x++)
		for (int c = 0; c < NBP; c++)
			if (c / NB <= I && c % NB <= I) {
				const int g1 = x * NB + GAP, g2 = GAP * NB + x;
				ok = ok && P.H[g1 * NBP + c] == 0.0f && P.S[g1 * NBP + c] == 0.0f && P.H[g2 * NBP + c] == 0.0f && P.S[g2 * NBP + c] == 0.0f;
			}
	struct Cls {
		bool on;
		float tk, theta0, ct_scale;
		std::vector<int> fpos, close; // [prev pair * 49 + cur pair] in evaluate order
		std::vector<int> fcur, fprev; // second tier: the stack as it is (no clamp at zero) where the CURRENT / the PREVIOUS column is Watson-Crick, else "not possible"
		std::vector<int> b_step, b_gap; // tier 3 (units of 1e-4 kcal/mol, rounded up): the credit of a diagonal step / of the aligned column at either end of a gap run
	} cls[2];
	const double min_tm[2] = {double(opt->min_primer_tm), double(opt->min_probe_tm)};
	const double max_dg[2] = {double(opt->max_primer_dg), double(opt->max_probe_dg)};
	for (int k = 0; k < 2; k++) {
		Cls &C = cls[k];
		// the threshold of the cascade (bind_oligo.cpp:600-624) the bound is taken against: the minimum Tm where there is one,
		// else a negative maximum dG (the same automaton at the target temperature, no strand-concentration term)
		const bool by_tm = min_tm[k] > 0.0 && min_tm[k] < 150.0;
		const bool by_dg = !by_tm && max_dg[k] < 0.0 && opt->target_t > 273.15f && opt->target_t < 423.15f;
		const double Tk = by_tm ? min_tm[k] + 273.15 : double(opt->target_t);
		C.tk = float(Tk);
		C.ct_scale = by_tm ? 1.0f : 0.0f;
		C.on = ok && (by_tm || by_dg);
		C.fpos.assign(NBP * NBP, 0);
		C.close.assign(NBP * NBP, NEG);
		C.fcur.assign(NBP * NBP, NEG);
		C.fprev.assign(NBP * NBP, NEG);
		C.b_step.assign(NBP * NBP, 0);
		C.b_gap.assign(NBP * NBP, 0);
		if (!C.on) continue;
		const double sc = double(P.salt) * log_na;
		auto phi = [&](int prev, int cur) { return -(double(P.H[prev * NBP + cur]) - Tk * (double(P.S[prev * NBP + cur]) + sc)); };
		double pb1 = -Tk * double(P.bulge_S[1]), pb2 = 1.0e30, pl3 = 1.0e30;
		// bulges up to the longest window, loops up to 2 x 64 mismatches + the window (entry 512 of both tables is 0 in the
		// reference, nuc_cruc_santa_lucia.cpp:670,720; no 128-mer reaches it)
		// (sizes an oligo of up to 128 bases and its window can form: the pre-filter has instances up to 128 rows)
		for (int z = 2; z <= 128 + 2 * TNT_FLANK; z++) pb2 = std::min(pb2, -Tk * double(P.bulge_S[z]));
		for (int z = 3; z <= 2 * 128 + 128 + 2 * TNT_FLANK; z++) pl3 = std::min(pl3, -Tk * double(P.loop_S[z]));
		pl3 += -Tk * double(P.asym_S);
		const double gap_price = std::min(std::min(pb2, pl3), pb1);
		for (int prev = 0; prev < NBP; prev++)
			for (int cur = 0; cur < NBP; cur++) {
				if (prev / NB > I || prev % NB > I || cur / NB > I || cur % NB > I) continue;
				const double f = phi(prev, cur);
				C.fpos[prev * NBP + cur] = units((P.wc[prev] || P.wc[cur]) ? std::max(f, 0.0) : 0.0);
				if (P.wc[cur]) C.fcur[prev * NBP + cur] = units(f);
				if (P.wc[prev]) C.fprev[prev * NBP + cur] = units(f);
				C.b_step[prev * NBP + cur] = int(std::ceil(((P.wc[prev] || P.wc[cur]) ? std::max(f, 0.0) : 0.0) * 1.0e4));
				{ // an aligned column next to a gap run: its terminal stack (with the diagonally adjacent cell), or the stack across a single-base bulge
					double best = (P.wc[prev] || P.wc[cur]) ? std::max(f, 0.0) : 0.0;
					if (P.wc[cur] || P.wc[prev])
						for (int pm = 0; pm < NBP; pm++)
							if (P.wc[pm] && pm / NB <= I && pm % NB <= I) best = std::max(best, std::max(phi(pm, cur), phi(prev, pm)) - pb1);
					C.b_gap[prev * NBP + cur] = int(std::ceil(best * 1.0e4));
				}
				if (!P.wc[cur]) continue; // a loop / bulge closes on a Watson-Crick column
				double u1 = -1.0e30;
				for (int pm = 0; pm < NBP; pm++)
					if (P.wc[pm] && pm / NB <= I && pm % NB <= I) u1 = std::max(u1, phi(pm, cur));
				u1 += gap_price - pb1;
				C.close[prev * NBP + cur] = units(std::max(std::max(f, 0.0), u1) - gap_price);
			}
		const double g_init = double(P.init_H) - Tk * double(P.init_S), g_at = double(P.at_H) - Tk * double(P.at_S);
		// dangling-end options: one more column per end, worth at most the sum of the two single-dangle stacks on the closing
		// pair (evaluate_alignment's frayed end, nuc_cruc.cpp:1737-1768), and one column's salt charge less
		double dangle = 0.0;
		if (opt->allow_dangle_5 || opt->allow_dangle_3) {
			auto raw = [&](int prev, int cur) { return std::max(0.0, -(double(P.H[prev * NBP + cur]) - Tk * double(P.S[prev * NBP + cur]))); };
			double d_start = 0.0, d_stop = 0.0;
			for (int c = 0; c < NBP; c++) {
				if (!P.wc[c] || c / NB > I || c % NB > I) continue;
				double qs = 0.0, ts = 0.0, qe = 0.0, te = 0.0;
				for (int x = A; x <= I; x++) {
					qs = std::max(qs, raw(x * NB + E, c)), ts = std::max(ts, raw(E * NB + x, c));
					qe = std::max(qe, raw(c, x * NB + E)), te = std::max(te, raw(c, E * NB + x));
				}
				d_start = std::max(d_start, qs + ts), d_stop = std::max(d_stop, qe + te);
			}
			// --dangle5 adds the column in front of the alignment (the 5' end of the query, read first: the stack INTO the first
			// Watson-Crick pair), --dangle3 the one behind it; a dangling column counts one base where the salt term assumed two
			const int ends = (opt->allow_dangle_5 ? 1 : 0) + (opt->allow_dangle_3 ? 1 : 0);
			dangle = (opt->allow_dangle_5 ? d_start : 0.0) + (opt->allow_dangle_3 ? d_stop : 0.0) - 0.5 * double(ends) * Tk * sc; // (sc <= 0)
		}
		C.theta0 = float(g_init + std::min(0.0, 2.0 * g_at) - 0.02 - (by_dg ? max_dg[k] : 0.0) - dangle);
	}
#pragma omp parallel for schedule(dynamic, 256)
	for (long k = 0; k < long(n); k++) {
		const tnt_bind_req &r = req[k];
		score[2 * k] = score[2 * k + 1] = -1;
		const size_t t = size_t(r.seq_index) - first_seq_index;
		if (t >= n_targets || r.oligo < 0 || r.oligo >= n_oligos) continue;
		const std::string q = oligos[r.oligo];
		const int L = int(q.size());
		const uint8_t *seq = codes + offsets[t];
		const unsigned seq_len = unsigned(offsets[t + 1] - offsets[t]);
		const unsigned start = unsigned(std::max(int(r.target_loc) - int(r.query_loc + TNT_FLANK), 0));
		const unsigned stop = std::min(start + unsigned(L) + 2 * TNT_FLANK, seq_len);
		// IUPAC codes (5..15 = M R S V W Y H K D B N) become "any of A C G T": resolve_degenerate (nuc_cruc.cpp:14-201) turns
		// them into one of the four per aligned pair, so the maximum over the four bounds whatever it picks.  Inosine in the
		// target (code 4) takes the same column value: the maximum runs over A C G T I.  The codes the window loader skips
		// (> 15) are not modelled: bound = INT32_MAX, the window is never rejected.
		const int ANY = -1;
		bool modelled = true;
		for (unsigned p = start; p < stop; p++) modelled = modelled && seq[p] <= 15;
		if (!modelled) {
			score[2 * k] = INT32_MAX;
			score[2 * k + 1] = INT32_MIN;
			continue;
		}
		// the window as NucCruc sees it (bind_oligo.cpp:524-573 / :1227-1276) ...
		std::vector<int> tg;
		for (unsigned p = start; p < stop; p++) tg.push_back(seq[p] > 3 ? ANY : (r.strand == TNT_MINUS ? 3 - int(seq[p]) : int(seq[p])));
		if (r.strand == TNT_MINUS) std::reverse(tg.begin(), tg.end());
		// ... read in pairing order: query 5'->3' against the target 3'->5'
		std::reverse(tg.begin(), tg.end());
		const int W = int(tg.size());
		std::vector<int> qb(L);
		for (int a = 0; a < L; a++) qb[a] = int(char_to_base(q[a]));
		const Cls &C = cls[r.role == 1 ? 1 : 0];
		// X: best helix value ending in (a, b); LP: best value inside a loop / bulge that left its diagonal, gap price paid
		std::vector<int> X(size_t(L + 1) * (W + 1), 0), LP(size_t(L + 1) * (W + 1), NEG);
		const double gp_unused = 0.0;
		(void)gp_unused;
		int best = 0;
		auto at = [&](int a, int b) { return size_t(a) * (W + 1) + b; };
		// recompute the gap price in integers the way the tables hold it: close[] already carries "- gap_price", so the entry
		// into LP from a helix cell is free here and the price is paid on the way out
		// table value of (previous pair, current pair), maximised over the bases an ANY column can stand for
		auto entry = [&](const std::vector<int> &tab, int qp, int tp, int qc, int tc) {
			int m = INT32_MIN;
			for (int x = (tp == ANY ? int(A) : tp); x <= (tp == ANY ? int(I) : tp); x++)
				for (int y = (tc == ANY ? int(A) : tc); y <= (tc == ANY ? int(I) : tc); y++) m = std::max(m, tab[(qp * NB + x) * NBP + qc * NB + y]);
			return m;
		};
		if (g_filter_tier == 3 || g_filter_tier == 30) {
			// Tier 3 (tnt_sbound.hip.h): NucCruc scores the MAXIMAL-SCORE paths of align_dimer (nuc_cruc.cpp:492-696, :2533-2537,
			// :1409-1618).  credit(A) = what the first tier's argument grants an alignment along its own path (non-negative: a
			// diagonal step's stack when one of its columns is Watson-Crick; at an aligned column next to a gap run the stack with its
			// diagonally adjacent cell, or the stack across a single-base bulge).  One integer per DP state, V = score << 11 | credit
			// (score exact, credit in units of 0.02 kcal/mol rounded up): the fill's own recurrence over these picks the higher score
			// and, among equal scores, the higher credit.  The credit of the window's maximal V bounds Phi - const of whatever
			// NucCruc scores.  Bit 10 of the credit field is a guard: set in any state, the window passes.  Plain windows, oligos of
			// up to 40 bases; the restatement runs the kernel's table-driven loops.
			// Windows with IUPAC codes or inosine in the target: align_dimer resolves the target base of a cell against the query base
			// it is paired with (bbp, nuc_cruc.cpp:203-213; against "none" where the partner is a gap), so the table entry of a
			// transition is that of the RESOLVED bases -- the kernel's general instance (sbound_kernel<L, false, true>) indexes its
			// shared table that way, this restatement calls the same resolve().
			int credit3 = INT32_MAX; // (not served by this tier: passes)
			if (C.on && L <= 40 && W >= 1) {
				Melt m(opt->target_t, opt->salt, false, false);
				m.set_query(q);
				load_window(m, seq, start, stop, r.strand == TNT_MINUS);
				const int K = 1 << 11, NEG3 = -(1 << 30);
				const int NONE = 5; // target base of a column: A C G T I, 5 = none (before the first / behind the last column)
				auto cl = [&](long long v) { return int(std::max<long long>(v, NEG3)); };
				auto up50 = [&](double kcal) { return int(std::ceil(kcal * 50.0)); };
				auto dgv = [&](int prev, int cur) { return m.g(prev, cur); };
				auto term = [&](int c) { // Tables::term (tnt_host.hip fill_tables): g((x,GAP), c) for real pairs, FORB_XY for (GAP, x)
					if (c / NB == GAP) return dgv(GAP * NB + A, A * NB + GAP);
					return dgv(A * NB + GAP, c);
				};
				const int GG = GAP * NB + GAP;
				const int c_bulge = dgv(A * NB + GAP, A * NB + GAP) * K, c_zero = dgv(GG, A * NB + GAP) * K;
				const int c_forb_m = cl(-(long long)dgv(GG, A * NB + A) * K), c_forb_xy = cl(-(long long)dgv(GAP * NB + A, A * NB + GAP) * K);
				// credits in the tables' convention: pair = query base * 7 + target base, phi(first = the later cell, second = its diagonal predecessor)
				const double sc = double(P.salt) * log_na, Tk3 = double(C.tk), pb1 = -Tk3 * double(P.bulge_S[1]);
				auto phi = [&](int prev, int cur) { return -(double(P.H[prev * NBP + cur]) - Tk3 * (double(P.S[prev * NBP + cur]) + sc)); };
				auto b_step = [&](int first, int second) { return (P.wc[first] || P.wc[second]) ? std::max(phi(first, second), 0.0) : 0.0; };
				auto b_gap = [&](int first, int second) {
					double best = b_step(first, second);
					if (P.wc[first] || P.wc[second])
						for (int pm = 0; pm < NBP; pm++)
							if (P.wc[pm] && pm / NB <= I && pm % NB <= I) best = std::max(best, std::max(phi(pm, second), phi(first, pm)) - pb1);
					return best;
				};
				// entry values for (x', x, q', q): x', x in A C G T I none(5); q', q base codes 0..6
				auto planes = [&](int xp, int x, int qp, int qq, int out[4]) {
					const int xpB = xp == NONE ? int(GAP) : xp, xB = x == NONE ? int(GAP) : x;
					const bool v1 = qq <= I && x != NONE, v2 = qp <= I && xp != NONE;
					const int first = qq * NB + x, second = qp * NB + xp; // (read only where valid)
					const int cpair = xB * NB + qq, ppair = xpB * NB + qp;
					const int cs = (v1 && v2) ? up50(b_step(first, second)) : 0;
					const int cg = (v1 && v2) ? up50(b_gap(first, second)) : (v1 ? up50(b_gap(first, first)) : 0);
					const int cc = v2 ? ((v1) ? up50(b_gap(first, second)) : up50(b_gap(second, second))) : 0;
					out[0] = x == NONE ? NEG3 : cl(-(long long)dgv(ppair, cpair) * K + cs);
					out[1] = x == NONE ? NEG3 : cl(-(long long)term(cpair) * K + cg);
					out[2] = xp == NONE ? c_forb_m : out[1];
					out[3] = cl(-(long long)term(xpB * NB + qp) * K + cc);
				};
				std::vector<int> oM(L, 0), oQ(L, 0), oT(L, 0), qrow(L + 2, int(GAP));
				for (int i = 1; i <= L; i++) qrow[i] = int(m.query[L - i]); // row i = query[L - i]; row 0 and row L + 1: none
				const int Wt = int(m.target.size()); // (codes the window loader skips are not in it: such windows are not served)
				bool served = Wt == W;
				int vmax = 0, sticky = 0;
				// the base of column j as the cell of row i sees it
				auto rx = [&](int j, int i) { return (j >= 1 && j <= Wt) ? int(resolve(m.target[j - 1], Base(qrow[i]))) : NONE; };
				for (int j = 1; j <= Wt && served; j++) {
					const bool firstcol = j == 1;
					const int v5 = firstcol ? c_zero : c_bulge;
					int dM = 0, dQ = 0, dT = 0, uM = 0, uT = 0;
					for (int i = 1; i <= L; i++) {
						int pa[4], pn[4], p4[4];
						planes(rx(j - 1, i - 1), rx(j, i), qrow[i - 1], qrow[i], pa);
						planes(rx(j - 1, i), rx(j, i + 1), qrow[i], qrow[i + 1], pn); // the next row's entry
						planes(rx(j, i - 1), rx(j + 1, i), qrow[i - 1], qrow[i], p4);
						const int cM = oM[i - 1], cQ = oQ[i - 1], cT = oT[i - 1];
						const int d1 = dM + pa[0], d2 = dQ + pa[2], d3 = dT + (i == 1 ? c_forb_m : pa[1]);
						const int M = std::max(d1, std::max(d2, d3));
						const int Iq = std::max(cM + pn[3], cQ - v5);
						const int It = std::max(uM + (i == 1 ? c_forb_xy : p4[3]), uT - (i == 1 ? c_zero : c_bulge));
						const int Mz = std::max(M, 0), Iqz = std::max(Iq, 0), Itz = std::max(It, 0);
						vmax = std::max(vmax, Mz);
						sticky |= Mz | Iqz | Itz;
						dM = cM, dQ = cQ, dT = cT;
						uM = Mz, uT = Itz;
						oM[i - 1] = Mz, oQ[i - 1] = Iqz, oT[i - 1] = Itz;
					}
				}
				if (served) credit3 = (sticky & (1 << 10)) ? INT32_MAX : (vmax & (K - 1)) * 20;
				if (served && getenv("TNTOR_TIER3_CHECK_SCORE")) { // development check: the fill's maximal score (positive maxima only)
					m.align_dimer();
					if (!(sticky & (1 << 10)) && std::max(m.max_score, 0) != (vmax >> 11)) fprintf(stderr, "tier 3: score %d vs fill %d\n", vmax >> 11, m.max_score);
				}
			}
			// (the first tier's bound follows; the smaller of the two stands)
			score[2 * k] = credit3;
		}
		if (g_filter_tier == 2) {
			// Second tier: the same decomposition of an alignment at its Watson-Crick columns, but a terminal
			// stack is charged as it is -- the first tier drops every unfavourable stack on a diagonal, which makes mismatches
			// free.  Two states per cell on its diagonal: H = the cell is a counted (Watson-Crick) column, S = the cell lies inside
			// a region of mismatch / gap columns that stays on (or returns to) this diagonal: entered from H with the stack
			// [H cell -> next cell] (the terminal stack at the region's first column), crossed for nothing (stacks inside a loop
			// are not credited; a region that returns to its diagonal may cost nothing at all: loop_S[2] = 0, symmetric gaps),
			// left into H with the stack [previous cell -> H cell].  LP as in the first tier, entered from H or S.
			std::vector<int> H(size_t(L + 1) * (W + 1), 0), S(size_t(L + 1) * (W + 1), 0);
			std::fill(LP.begin(), LP.end(), NEG);
			auto sat = [&](int v) { return std::max(NEG, std::min(SAT, v)); };
			for (int a = 1; a <= L; a++)
				for (int b = 1; b <= W; b++) {
					int h = 0, sv = 0, lp = NEG;
					if (a > 1 && b > 1) {
						const int y = std::max(H[at(a - 1, b - 1)], S[at(a - 1, b - 1)]);
						h = std::max(h, sat(y + entry(C.fcur, qb[a - 2], tg[b - 2], qb[a - 1], tg[b - 1])));
						h = std::max(h, sat(LP[at(a - 1, b - 1)] + entry(C.close, qb[a - 2], tg[b - 2], qb[a - 1], tg[b - 1])));
						sv = std::max(S[at(a - 1, b - 1)], sat(H[at(a - 1, b - 1)] + entry(C.fprev, qb[a - 2], tg[b - 2], qb[a - 1], tg[b - 1])));
						sv = std::max(sv, 0);
						lp = LP[at(a - 1, b - 1)];
					}
					lp = std::max(lp, std::max(LP[at(a - 1, b)], LP[at(a, b - 1)]));
					lp = std::max(lp, std::max(std::max(H[at(a - 1, b)], S[at(a - 1, b)]), std::max(H[at(a, b - 1)], S[at(a, b - 1)])));
					H[at(a, b)] = h, S[at(a, b)] = sv, LP[at(a, b)] = lp;
					best = std::max(best, h);
				}
		} else
		for (int a = 1; a <= L; a++)
			for (int b = 1; b <= W; b++) {
				int x = 0, lp = NEG;
				if (a > 1 && b > 1) {
					x = std::max(X[at(a - 1, b - 1)] + entry(C.fpos, qb[a - 2], tg[b - 2], qb[a - 1], tg[b - 1]),
					             LP[at(a - 1, b - 1)] + entry(C.close, qb[a - 2], tg[b - 2], qb[a - 1], tg[b - 1]));
					lp = LP[at(a - 1, b - 1)];
				}
				x = std::max(x, 0);
				lp = std::max(lp, std::max(LP[at(a - 1, b)], LP[at(a, b - 1)]));
				lp = std::max(lp, std::max(X[at(a - 1, b)], X[at(a, b - 1)]));
				X[at(a, b)] = x;
				LP[at(a, b)] = lp;
				best = std::max(best, x);
			}
		if (g_filter_tier == 3) best = std::min(best, score[2 * k]);
		if (g_filter_tier == 30) best = score[2 * k]; // (the tier's own figure: what the kernel's unit-level entry reports)
		score[2 * k] = std::min(best, SAT);
		if (C.on) {
			const float log_ct = logf((r.ct - 0.5f * 0.0f) * 1.0f);
			const float th = C.theta0 - C.ct_scale * (C.tk * (1.9872e-3f * log_ct));
			score[2 * k + 1] = int(floorf(th * float(UNITS))) - 1;
		} else {
			score[2 * k + 1] = INT32_MIN;
		}
	}
	return 0;
}

int tntor_oligo_structure(const tnt_options *opt, const char *oligo, float ct, float *hairpin_tm, float *homodimer_tm)
{
	try {
		Melt m(opt->target_t, opt->salt, opt->allow_dangle_5 != 0, opt->allow_dangle_3 != 0);
			m.dinkelbach = opt->use_dinkelbach != 0;
		m.set_query(oligo);
		m.ct = ct - 0.5f * ct; // strand(c, c), nuc_cruc.h:890-910
		*hairpin_tm = m.approximate_tm_hairpin();
		*homodimer_tm = m.approximate_tm_homodimer();
		return 0;
	} catch (const char *e) {
		g_err = e;
		return -1;
	}
}

int tntor_heterodimer(const tnt_options *opt, const char *a, const char *b, float ct_a, float ct_b, float *tm)
{
	try {
		Melt m(opt->target_t, opt->salt, opt->allow_dangle_5 != 0, opt->allow_dangle_3 != 0);
			m.dinkelbach = opt->use_dinkelbach != 0;
		m.set_query(a);
		m.target.clear();
		for (const char *p = b; *p; p++) m.target.push_back(char_to_base(*p)); // set_target, nuc_cruc.h:977-990
		m.ct = (ct_a > ct_b) ? ct_a - 0.5f * ct_b : ct_b - 0.5f * ct_a;
		*tm = m.approximate_tm_heterodimer();
		return 0;
	} catch (const char *e) {
		g_err = e;
		return -1;
	}
}

// ---------------------------------------------------------------------------------------------------------
// FASTA ingest.  Restates sequence_data::load_fasta (sequence_data_fastx.cpp:13-79) and
// read_bio_seq_fasta_slow (:187-382) + ascii_to_hash_base (seq.h:148-189) on an in-memory copy of the file.
static uint8_t ascii_to_code(unsigned char c)
{
	switch (toupper(c)) { // seq.h:150-187
	case 'A': return 0;
	case 'C': return 1;
	case 'G': return 2;
	case 'T':
	case 'U': return 3;
	case 'I': return 4;
	case 'M': return 5;
	case 'R': return 6;
	case 'S': return 7;
	case 'V': return 8;
	case 'W': return 9;
	case 'Y': return 10;
	case 'H': return 11;
	case 'K': return 12;
	case 'D': return 13;
	case 'B': return 14;
	case 'N': return 15;
	case '-': return 16;
	}
	return 17;
}

// file_type, annotation.cpp:296-374, as far as FASTA / FASTQ go: the first non-space character of the first 511 bytes
static bool is_fastq(const char *text, size_t n_bytes)
{
	for (size_t i = 0; i < n_bytes && i < 511; i++)
		if (!isspace((unsigned char)text[i])) return text[i] == '@';
	return false;
}

// load_fastq + read_bio_seq_fastq_slow, sequence_data_fastx.cpp:81-186,384-583
static int parse_fastq(const char *text, size_t n_bytes, std::vector<uint8_t> &all, std::vector<uint64_t> &off, std::string &defs)
{
	// record index (:103-158): an '@' that starts a line opens a record unless the line before it was the '+' line
	// (then it is the first quality value)
	std::vector<size_t> index;
	char last_header = '-';
	bool read_eol = true;
	for (size_t i = 0; i < n_bytes; i++) {
		switch (text[i]) {
		case '@':
			if (read_eol) {
				if (last_header != '+') index.push_back(i);
				last_header = '@';
			}
			read_eol = false;
			break;
		case '+':
			if (read_eol) last_header = (last_header == '+') ? '-' : '+';
			read_eol = false;
			break;
		case ' ':
		case '\t': break;
		case '\n':
		case '\r': read_eol = true; break;
		default:
			if (read_eol) last_header = '-';
			read_eol = false;
			break;
		}
	}
	const size_t n = index.size();
	index.push_back(n_bytes);
	for (size_t r = 0; r < n; r++) {
		const char *ptr = text + index[r] + 1, *last = text + index[r + 1];
		while (last > ptr && isspace((unsigned char)*ptr)) ptr++; // :452-455
		if (ptr == last) return -1;
		const char *defline_start = ptr;
		while (last > ptr && *ptr != '\n' && *ptr != '\r') ptr++;
		if (ptr == last) return -1;
		defs.append(defline_start, ptr - defline_start);
		defs += '\n';
		ptr++; // the end-of-line character (one of them: a "\r\n" file has empty sequences, :568-571)
		if (ptr == last) return -1;
		for (; ptr != last && *ptr != '\n' && *ptr != '\r'; ++ptr) // the sequence is one line
			if (!isspace((unsigned char)*ptr) && *ptr != '*' && *ptr != '-') all.push_back(ascii_to_code((unsigned char)*ptr));
		off.push_back(all.size());
	}
	return int(n);
}

int tntor_parse_fasta(const char *text, size_t n_bytes, uint8_t **codes, uint64_t **offsets, size_t *n_targets, char **deflines)
{
	std::vector<uint8_t> all;
	std::vector<uint64_t> off(1, 0);
	std::string defs;
	size_t n = 0;
	std::vector<size_t> index;
	if (is_fastq(text, n_bytes)) {
		const int rc = parse_fastq(text, n_bytes, all, off, defs);
		if (rc < 0) {
			g_err = "Truncated fastq file detected!";
			return -1;
		}
		n = size_t(rc);
	} else {
	// record index: a '>' opens a record unless one was already seen on this line (:37-52)
	bool read_fasta = false;
	for (size_t i = 0; i < n_bytes; i++) {
		if (!read_fasta && text[i] == '>') {
			read_fasta = true;
			index.push_back(i);
		} else if (text[i] == '\n') {
			read_fasta = false;
		}
	}
	n = index.size();
	index.push_back(n_bytes);
	}
	for (size_t r = 0; r < n && !index.empty(); r++) {
		const char *ptr = text + index[r] + 1, *last = text + index[r + 1]; // skip the '>' (:245)
		while (last > ptr && isspace((unsigned char)*ptr)) ptr++;           // :248-250
		if (ptr == last) {
			g_err = "Truncated fasta file detected!";
			return -1;
		}
		const char *defline_start = ptr;
		while (last > ptr && *ptr != '\n' && *ptr != '\r') ++ptr; // :265-267
		if (ptr == last) {
			g_err = "Truncated fasta file detected!";
			return -1;
		}
		defs.append(defline_start, ptr - defline_start);
		defs += '\n';
		for (; ptr != last; ++ptr) // :366-370
			if (!isspace((unsigned char)*ptr) && *ptr != '*' && *ptr != '-' && *ptr != '\r') all.push_back(ascii_to_code((unsigned char)*ptr));
		off.push_back(all.size());
	}
	*codes = (uint8_t *)malloc(all.size() + 1);
	memcpy(*codes, all.data(), all.size());
	*offsets = (uint64_t *)malloc(off.size() * sizeof(uint64_t));
	memcpy(*offsets, off.data(), off.size() * sizeof(uint64_t));
	*deflines = (char *)malloc(defs.size() + 1);
	memcpy(*deflines, defs.c_str(), defs.size() + 1);
	*n_targets = n;
	return 0;
}

void tntor_free(void *p) { free(p); }

} // extern "C"
