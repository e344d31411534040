// tnt_host.hip -- host side of libtnt_mi355x.so: the C ABI of include/tnt_mi355x.h on top of the
// gfx950 kernels (tnt_seed.hip.h, tnt_bind.hip.h, tnt_pair.hip.h).
//
// Product code.  There is NO CPU fallback: every entry point that computes needs a HIP device and
// fails with TNT_E_DEVICE otherwise.  Nothing here includes or links anything under oracle/.
//
// Host-side restatements needed to set the kernels up (tables, query words) cite the reference:
//   update_dp_param            nuc_cruc.cpp:340-487
//   resolve_degenerate / best_base_pair   nuc_cruc.cpp:14-213
//   build_word_list<string>    seq_hash.h:287-374
//   strand concentrations      tntblast_local.cpp:232-234, amplicon_search.cpp:85-87, nuc_cruc.h:890-910
//   assay dispatch             tntblast_local.cpp:559-626
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <map>
#include <deque>
#include <memory>
#include <string>
#include <thread>
#include <tuple>
#include <vector>

#include "../../include/tnt_mi355x.h"
#include "tnt_dev.h"
#include "tnt_arena.h"
#include "tnt_seed.hip.h"
#include "tnt_seed2.hip.h"
#define TNT_FINISH_KERNEL
#include "tnt_bind.hip.h"
#include "tnt_filter.hip.h"
#define TNT_REGROUP_KERNELS
#include "tnt_ufilter.hip.h"
#include "tnt_sbound.hip.h"
#include "tnt_pair.hip.h"
#include "tnt_ingest.hip.h"
#include <zlib.h>
#include <dlfcn.h>
#include <rccl/rccl.h> // types and prototypes only: the library is looked up at run time (tnt_gang_create)
#include <pthread.h>
#include <unistd.h>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <mutex>
#include "santalucia_tables.inc"

using namespace tnt;

namespace tnt { // tnt_sort.hip
size_t long_sort_bytes(size_t n);
size_t long_cull_bytes(size_t n, size_t n_lists_max);
hipError_t long_first_cull(const Item *items, const uint32_t *bucket, uint32_t n, const AssayDev *assays, uint32_t n_assays, int max_len, int single_primer,
                           Item *kept, uint32_t *n_kept, Group **groups, uint32_t *n_groups, DeviceArena &arena, hipStream_t stream);
hipError_t sort_pairs_u64_u32(void *tmp, size_t &tmp_bytes, const unsigned long long *keys_in, unsigned long long *keys_out, const uint32_t *vals_in, uint32_t *vals_out,
                              size_t n, hipStream_t stream);
hipError_t long_sort_seeds(const SeedOut *seeds, uint32_t n, const uint32_t *target_rank, const AssayDev *assays, uint32_t n_assays,
                           uint32_t n_buckets, Item *items, uint32_t *bucket, DeviceArena &arena, hipStream_t stream);
size_t long_pair_bytes(size_t n);
size_t long_hash_bytes(size_t n_kept, size_t n_bound, size_t n_lists);
hipError_t long_hash_device(const Item *kept, uint32_t n_kept, const Group *h_groups, uint32_t n_groups, const uint8_t *h_take, const int32_t *result, const BoundRec *pool,
                            const AssayDev *assays, const AssayDev *h_assays, const uint32_t *h_unit_target, uint32_t first_seq_index, int pair_max_len, std::vector<Hit> &hits_out,
                            DeviceArena &arena, hipStream_t stream);
hipError_t long_pair_device(const Item *kept, uint32_t n_kept, const Group *h_groups, uint32_t n_groups, const uint8_t *h_take, const int32_t *result, const BoundRec *pool,
                            const AssayDev *assays, const AssayDev *h_assays, const uint32_t *h_unit_target, uint32_t first_seq_index, int max_len, int single_primer,
                            int min_max_primer_clamp, std::vector<Hit> &hits_out, std::vector<uint8_t> &h_flagged, DeviceArena &arena, hipStream_t stream);
}

static_assert(sizeof(Hit) == sizeof(tnt_hit), "device Hit must mirror tnt_hit");
static_assert(MAX_OLIGO == TNT_MAX_OLIGO && ALIGN_TEXT == TNT_ALIGN_TEXT && FLANK == TNT_FLANK, "ABI constants");

namespace {

std::string g_create_error;

#define HIP_TRY(expr, ctx_err)                                                          \
	do {                                                                                \
		hipError_t e_ = (expr);                                                         \
		if (e_ != hipSuccess) {                                                         \
			(ctx_err) = std::string(#expr) + ": " + hipGetErrorString(e_);              \
			return TNT_E_DEVICE;                                                        \
		}                                                                               \
	} while (0)

inline float f32(unsigned int bits)
{
	float f;
	memcpy(&f, &bits, 4);
	return f;
}

// ---------------------------------------------------------------------------------------------
// host-side tables

int char_to_base(char c) // expanded oligos only: A C G T I (degenerate_na.cpp:81-133 expands the rest)
{
	switch (c) {
	case 'A': case 'a': return bA;
	case 'C': case 'c': return bC;
	case 'G': case 'g': return bG;
	case 'T': case 't': return bT;
	case 'I': case 'i': return bI;
	default: return -1;
	}
}

int resolve(int x, int partner) // resolve_degenerate, nuc_cruc.cpp:14-201 (B falls through into N, :163-197)
{
	switch (x) {
	case bM: return partner == bT ? bA : (partner == bG ? bC : bA);
	case bR: return partner == bT ? bA : (partner == bC ? bG : bA);
	case bS: return partner == bG ? bC : (partner == bC ? bG : bG);
	case bV: return partner == bG ? bC : (partner == bC ? bG : bA);
	case bW: return partner == bA ? bT : bA;
	case bY: return partner == bG ? bC : bT;
	case bH: return partner == bT ? bA : (partner == bG ? bC : (partner == bA ? bT : bA));
	case bK: return partner == bC ? bG : bT;
	case bD: return partner == bC ? bG : (partner == bT ? bA : (partner == bA ? bT : bA));
	case bB:
	case bN: return partner == bA ? bT : (partner == bT ? bA : (partner == bG ? bC : (partner == bC ? bG : bA)));
	default: return x;
	}
}

// returns false when delta_g lacks the class structure the bind kernel relies on (cannot happen for tables
// produced by update_dp_param; checked anyway so a future table change fails loudly instead of silently)
bool fill_tables(Tables &T, const tnt_options &opt)
{
	memset(&T, 0, sizeof(T));
	for (int k = 0; k < TNT_SL_NN_COUNT; k++) {
		T.H[TNT_SL_NN[k][0]] = f32(TNT_SL_NN[k][1]);
		T.S[TNT_SL_NN[k][0]] = f32(TNT_SL_NN[k][2]);
	}
	{ // distinct (H, S) pairs (bit patterns) -> byte map for the LDS copy of the scoring tables
		int n = 0;
		for (int i = 0; i < NBP * NBP; i++) {
			int k = 0;
			for (; k < n; k++)
				if (memcmp(&T.hs_H[k], &T.H[i], 4) == 0 && memcmp(&T.hs_S[k], &T.S[i], 4) == 0) break;
			if (k == n) {
				if (n == 256) return false;
				T.hs_H[n] = T.H[i];
				T.hs_S[n] = T.S[i];
				n++;
			}
			T.hs_map[i] = uint8_t(k);
		}
	}
	for (int i = 0; i < 513; i++) {
		T.loop_S[i] = f32(TNT_SL_LOOP_S[i]);
		T.bulge_S[i] = f32(TNT_SL_BULGE_S[i]);
		T.hairpin_S[i] = f32(TNT_SL_HAIRPIN_S[i]);
	}
	T.n_special = TNT_SL_HAIRPIN_SPECIAL_COUNT;
	for (int i = 0; i < TNT_SL_HAIRPIN_SPECIAL_COUNT && i < 136; i++) {
		const char *q = TNT_SL_HAIRPIN_SPECIAL[i].seq;
		uint32_t key = uint32_t(strlen(q)) << 24;
		for (int k = 0; q[k]; k++) key |= uint32_t(char_to_base(q[k])) << (3 * k);
		T.special_key[i] = key;
		T.special_H[i] = f32(TNT_SL_HAIRPIN_SPECIAL[i].h);
		T.special_S[i] = f32(TNT_SL_HAIRPIN_SPECIAL[i].s);
	}
	T.sym_S = f32(TNT_SL_SYMMETRY_S);
	T.init_H = f32(TNT_SL_INIT_H);
	T.init_S = f32(TNT_SL_INIT_S);
	T.at_H = f32(TNT_SL_AT_CLOSING_H);
	T.at_S = f32(TNT_SL_AT_CLOSING_S);
	T.salt = f32(TNT_SL_SALT);
	T.asym_S = f32(TNT_SL_ASYM_LOOP_S);
	T.bulge_at_S = f32(TNT_SL_BULGE_AT_S);
	T.temperature = opt.target_t;
	T.log_na = logf(opt.salt);
	T.dangle5 = opt.allow_dangle_5 ? 1 : 0;
	T.dangle3 = opt.allow_dangle_3 ? 1 : 0;
	T.word_size = opt.hash_word_size;
	for (int x = 0; x < 18; x++)
		for (int y = 0; y < 18; y++) T.bbp[x * 18 + y] = uint8_t(resolve(x, y) * NB + resolve(y, x));
	for (int x = 0; x < 18; x++) {
		T.rt[x] = 0;
		for (int q = 0; q < NB; q++) T.rt[x] |= uint32_t(resolve(x, q)) << (4 * q);
	}

	// update_dp_param, nuc_cruc.cpp:340-487; supplementary parameters nuc_cruc.cpp:271-300
	bool wc[NBP];
	memset(wc, 0, sizeof(wc));
	wc[P_AT] = wc[P_TA] = wc[P_CG] = wc[P_GC] = true;
	for (int b = bA; b <= bI; b++) wc[b * NB + bI] = wc[bI * NB + b] = true;
	const float Tk = opt.target_t;
	const float salt_correction = T.salt * logf(opt.salt);
	const float loop_sc = salt_correction * 3.08f, bulge_sc = salt_correction * 0.69f;
	const float tmatch_sc = salt_correction * 0.56f, tmis_sc = salt_correction * 1.31f;
	auto scale = [](float x) { return int32_t(x * 10000.0f); };
	for (int i = 0; i < NBP * NBP; i++) {
		T.dk_H[i] = T.H[i], T.dk_S[i] = T.S[i] + salt_correction, T.dk_F[i] = INT32_MIN;
		T.dg[i] = scale(T.H[i] - Tk * (T.S[i] + salt_correction));
	}
	// the supplementary classes as (H, S + salt term) pairs: their value at any temperature is max(0, scale(H - T * S))
	struct Cls {
		float H, S;
	};
	const Cls k_at = {-4.474f, -2.091e-2f + tmatch_sc}, k_gc = {-3.000f, -1.318e-2f + tmatch_sc}, k_mm = {-2.421f, -1.180e-2f + tmis_sc},
	          k_loop = {-5.779f, -2.330e-2f + loop_sc}, k_bulge = {5.247e-1f, 3.318e-4f + bulge_sc};
	auto put = [&](int idx, const Cls &k) {
		T.dk_H[idx] = k.H, T.dk_S[idx] = k.S, T.dk_F[idx] = 0;
		T.dg[idx] = std::max(0, scale(k.H - Tk * k.S));
	};
	const int32_t v_at = std::max(0, scale(k_at.H - Tk * k_at.S));
	const int32_t v_gc = std::max(0, scale(k_gc.H - Tk * k_gc.S));
	const int32_t v_mm = std::max(0, scale(k_mm.H - Tk * k_mm.S));
	const int32_t v_loop = std::max(0, scale(k_loop.H - Tk * k_loop.S));
	const int32_t v_bulge = std::max(0, scale(k_bulge.H - Tk * k_bulge.S));
	(void)v_loop, (void)v_bulge;
	for (int i = bA; i <= bI; i++)
		for (int j = bA; j <= bI; j++) {
			const int cur = i * NB + j;
			const Cls &kv = !wc[cur] ? k_mm : ((cur == P_GC || cur == P_CG) ? k_gc : k_at); // inosine pairs == AT (:289-290)
			for (int k = bA; k <= bI; k++) {
				const int p1 = k * NB + bGAP, p2 = bGAP * NB + k;
				put(cur * NBP + p1, kv), put(p1 * NBP + cur, kv), put(cur * NBP + p2, kv), put(p2 * NBP + cur, kv);
			}
			for (int k = bA; k <= bI; k++)
				for (int l = bA; l <= bI; l++)
					if (!wc[cur] && !wc[k * NB + l]) put(cur * NBP + k * NB + l, k_loop);
		}
	for (int i = bA; i <= bI; i++)
		for (int j = bA; j <= bI; j++) {
			put((i * NB + bGAP) * NBP + (j * NB + bGAP), k_bulge);
			put((bGAP * NB + i) * NBP + (bGAP * NB + j), k_bulge);
		}
	(void)v_at, (void)v_gc, (void)v_mm;

	// ---- class constants used by the bind kernel instead of six of the seven table reads per cell
	bool ok = true;
	auto G = [&](int prev, int cur) { return T.dg[prev * NBP + cur]; };
	const int GG = P__;
	T.const_src[0] = (bA * NB + bGAP) * NBP + bA * NB + bGAP; // BULGE
	T.const_src[1] = GG * NBP + bA * NB + bGAP;              // C0 = g(__, (x,GAP))
	T.const_src[2] = GG * NBP + bA * NB + bA;                // FORB_M = g(__, real pair)
	T.const_src[3] = (bGAP * NB + bA) * NBP + bA * NB + bGAP; // FORB_XY = g((GAP,x),(y,GAP))
	for (int k = 0; k < 4; k++) T.dg_const[k] = T.dg[T.const_src[k]];
	for (int p = 0; p < NBP; p++) T.term[p] = 0, T.term_src[p] = -1;
	for (int x = bA; x <= bI; x++)
		for (int y = bA; y <= bI; y++) {
			const int c = x * NB + y;
			const int v = G(bA * NB + bGAP, c);
			T.term[c] = v;
			T.term_src[c] = (bA * NB + bGAP) * NBP + c;
			ok &= G(GG, c) == T.dg_const[2];
			for (int k = bA; k <= bI; k++) {
				const int p1 = k * NB + bGAP, p2 = bGAP * NB + k;
				ok &= G(p1, c) == v && G(p2, c) == v && G(c, p1) == v && G(c, p2) == v;
			}
			// bulge extensions and the forbidden / gap-opening combinations
			const int xg = x * NB + bGAP, gx = bGAP * NB + x, yg = y * NB + bGAP, gy = bGAP * NB + y;
			ok &= G(xg, yg) == T.dg_const[0] && G(gx, gy) == T.dg_const[0];
			ok &= G(gx, yg) == T.dg_const[3] && G(xg, gy) == T.dg_const[3];
			ok &= G(GG, xg) == T.dg_const[1] && G(GG, gx) == T.dg_const[1];
		}
	for (int x = bA; x <= bI; x++) T.term[bGAP * NB + x] = T.dg_const[3], T.term_src[bGAP * NB + x] = T.const_src[3]; // first column: prev = bp(GAP, q)
	// the triples reproduce the table at the target temperature (what --dinkelbach recomputes per pass)
	for (int i = 0; i < NBP * NBP; i++) ok &= std::max(scale(T.dk_H[i] - Tk * T.dk_S[i]), T.dk_F[i]) == T.dg[i];
	return ok;
}

// Tables of the pre-filter (tnt_filter.hip.h): an upper bound, per pair of diagonally adjacent cells, of what
// evaluate_alignment (nuc_cruc.cpp:1620-2299) can credit to an alignment that uses them, at the temperature of the
// minimum-Tm threshold.  Returns with enabled[] = 0 wherever the derivation's premises do not hold (then nothing is
// rejected): dangling-end options, a threshold at or below 0 C, parameter tables with a stabilising loop / bulge term.
void fill_filter_tables(FilterTables &F, const Tables &T, const tnt_options &opt)
{
	memset(&F, 0, sizeof(F));
	bool wc[NBP];
	memset(wc, 0, sizeof(wc));
	wc[P_AT] = wc[P_TA] = wc[P_CG] = wc[P_GC] = true;
	for (int b = bA; b <= bI; b++) wc[b * NB + bI] = wc[bI * NB + b] = true;
	// premises: every loop / bulge / asymmetry / AT-bulge term is an entropy penalty, the salt term
	// destabilises, stacks on a gap column carry nothing, no dangling-end columns
	bool ok = !opt.use_dinkelbach && T.log_na <= 0.0f && T.salt >= 0.0f;
	for (int i = 0; i < 513; i++) ok = ok && T.loop_S[i] <= 0.0f && T.bulge_S[i] <= 0.0f;
	ok = ok && T.asym_S <= 0.0f && T.bulge_at_S <= 0.0f;
	for (int x = bA; x <= bI; x++)
		for (int c1 = bA; c1 <= bI; c1++)
			for (int c2 = bA; c2 <= bI; c2++) {
				const int c = c1 * NB + c2, g1 = x * NB + bGAP, g2 = bGAP * NB + x;
				ok = ok && T.H[g1 * NBP + c] == 0.0f && T.S[g1 * NBP + c] == 0.0f && T.H[g2 * NBP + c] == 0.0f && T.S[g2 * NBP + c] == 0.0f;
			}
	const double min_tm[FILT_CLASSES] = {double(opt.min_primer_tm), double(opt.min_probe_tm)};
	const double max_dg[FILT_CLASSES] = {double(opt.max_primer_dg), double(opt.max_probe_dg)};
	// table units: FILT_UNITS per kcal/mol, rounded up, 16 bits ("not possible" = FILT_NEG: a price of 16 kcal/mol)
	auto units = [](double kcal) { return int(std::min(double(FILT_SAT), std::max(double(FILT_NEG), std::ceil(kcal * double(FILT_UNITS))))); };
	auto pack = [](int stack, int entry) { return uint32_t(uint16_t(int16_t(stack))) | (uint32_t(uint16_t(int16_t(entry))) << 16); };
	for (int cls = 0; cls < FILT_CLASSES; cls++) {
		int v[FILT_PLANE][2];
		for (int i = 0; i < FILT_PLANE; i++) v[i][0] = 0, v[i][1] = FILT_NEG;
		// the threshold the bound is taken against (tnt_filter.hip.h, FILT_BY_*): the minimum Tm, else a negative maximum dG
		const bool by_tm = min_tm[cls] > 0.0 && min_tm[cls] < 150.0;
		const bool by_dg = !by_tm && max_dg[cls] < 0.0 && opt.target_t > 273.15 && opt.target_t < 423.15;
		const double Tk = by_tm ? min_tm[cls] + 273.15 : double(opt.target_t);
		F.tk[cls] = float(Tk);
		F.ct_scale[cls] = by_tm ? 1.0f : 0.0f;
		for (int i = 0; i < FILT_PLANE; i++) F.plane[cls][i] = pack(0, FILT_NEG);
		if (!ok || !(by_tm || by_dg)) continue;
		const double sc = double(T.salt) * double(T.log_na); // per counted column (nuc_cruc.cpp:2270: 0.5 per base)
		auto phi = [&](int prev, int cur) { return -(double(T.H[prev * NBP + cur]) - Tk * (double(T.S[prev * NBP + cur]) + sc)); };
		double pb1 = -Tk * double(T.bulge_S[1]), pb2 = 1.0e30, pl3 = 1.0e30;
		// sizes a window of this library can hold (the reference leaves the last entry of both tables at 0,
		// nuc_cruc_santa_lucia.cpp:670,720: a 512-base loop no 128-mer can form)
		for (int sz = 2; sz <= MAX_WINDOW; sz++) pb2 = std::min(pb2, -Tk * double(T.bulge_S[sz]));
		for (int sz = 3; sz <= 2 * MAX_LANE_OLIGO + MAX_WINDOW; sz++) pl3 = std::min(pl3, -Tk * double(T.loop_S[sz]));
		pl3 += -Tk * double(T.asym_S); // a loop that leaves its diagonal is asymmetric by at least one base
		const double gap_price = std::min(std::min(pb2, pl3), pb1);
		// best stack across a single-base bulge into pair c (nuc_cruc.cpp:2105-2120: the flanking stack is kept)
		double u1[NBP];
		for (int c = 0; c < NBP; c++) {
			u1[c] = -1.0e30;
			for (int pm = 0; pm < NBP; pm++)
				if (wc[pm] && pm / NB <= bI && pm % NB <= bI) u1[c] = std::max(u1[c], phi(pm, c));
			u1[c] += gap_price - pb1;
		}
		// the two table values of (previous column, current column) = ((x', q'), (x, q)), target bases as codes A C G T I.
		// The sweep runs against the order evaluate_alignment reads the columns in: cell (i, j) = (q, x) comes first there, its
		// diagonal neighbour (i-1, j-1) = (q', x') second.
		auto cell = [&](int xp, int x, int qp, int q, int k) {
			const int first = q * NB + x, second = qp * NB + xp;
			const double f = phi(first, second);
			if (k == 0) return units((wc[first] || wc[second]) ? std::max(f, 0.0) : 0.0); // a stack inside a loop is never credited
			if (!wc[second]) return int(FILT_NEG);                                          // a loop / bulge is closed by a Watson-Crick column
			return units(std::max(std::max(f, 0.0), u1[second]) - gap_price);
		};
		// A column that holds an IUPAC code (FILT_ANY) is resolved per aligned pair to one of A C G T (resolve_degenerate,
		// nuc_cruc.cpp:14-201), so the maximum over the four bases bounds whatever it picks, in either role of the column.
		// Inosine in the TARGET (seq.h DB_I) takes the same column value: the maximum runs over A C G T I.
		for (int qp = bA; qp <= bI; qp++)
			for (int q = bA; q <= bI; q++)
				for (int k = 0; k < 2; k++) {
					auto at = [&](int xp, int x) { return (xp * FILT_X + x) * NBP + qp * NB + q; };
					int all = INT32_MIN;
					for (int xp = bA; xp <= bI; xp++) {
						int row = INT32_MIN;
						for (int x = bA; x <= bI; x++) {
							const int c = cell(xp, x, qp, q, k);
							row = std::max(row, c), all = std::max(all, c);
							if (xp < 4 && x < 4) v[at(xp, x)][k] = c;
						}
						if (xp < 4) v[at(xp, FILT_ANY)][k] = row; // (x', ANY)
					}
					for (int x = bA; x <= bT; x++) { // (ANY, x)
						int col = INT32_MIN;
						for (int xp = bA; xp <= bI; xp++) col = std::max(col, cell(xp, x, qp, q, k));
						v[at(FILT_ANY, x)][k] = col;
					}
					v[at(FILT_ANY, FILT_ANY)][k] = all;
				}
		for (int i = 0; i < FILT_PLANE; i++) F.plane[cls][i] = pack(v[i][0], v[i][1]);
		const double g_init = double(T.init_H) - Tk * double(T.init_S);
		const double g_at = double(T.at_H) - Tk * double(T.at_S);
		// Dangling-end options (enumerate_dimer_alignments, nuc_cruc.cpp:1068-1137): the trimmed alignment gets one more column
		// per end -- a dangling base, or both bases of a frayed end, which evaluate_alignment credits as the SUM of the two
		// single-dangle stacks (:1737-1768).  Each end therefore adds at most D = max over the closing Watson-Crick pair of
		// (best (q,E) stack + best (E,t) stack), and a dangling column carries one base where the salt term's count assumed
		// two: one column's salt charge less.  Both are constants: they loosen the threshold.
		double dangle = 0.0;
		if (opt.allow_dangle_5 || opt.allow_dangle_3) {
			auto raw = [&](int prev, int cur) { return std::max(0.0, -(double(T.H[prev * NBP + cur]) - Tk * double(T.S[prev * NBP + cur]))); };
			double d_start = 0.0, d_stop = 0.0;
			for (int c = 0; c < NBP; c++) {
				if (!wc[c] || c / NB > bI || c % NB > bI) continue;
				double qs = 0.0, ts = 0.0, qe = 0.0, te = 0.0;
				for (int x = bA; x <= bI; x++) {
					qs = std::max(qs, raw(x * NB + bE, c)), ts = std::max(ts, raw(bE * NB + x, c));
					qe = std::max(qe, raw(c, x * NB + bE)), te = std::max(te, raw(c, bE * NB + x));
				}
				d_start = std::max(d_start, qs + ts), d_stop = std::max(d_stop, qe + te);
			}
			// --dangle5 adds the column in front of the alignment (the 5' end of the query, read first: the stack INTO the first
			// Watson-Crick pair), --dangle3 the one behind it; a dangling column counts one base where the salt term assumed two
			const int ends = (opt.allow_dangle_5 ? 1 : 0) + (opt.allow_dangle_3 ? 1 : 0);
			dangle = (opt.allow_dangle_5 ? d_start : 0.0) + (opt.allow_dangle_3 ? d_stop : 0.0) - 0.5 * double(ends) * Tk * sc; // (sc <= 0)
		}
		F.theta0[cls] = float(g_init + std::min(0.0, 2.0 * g_at) - 0.02 - (by_dg ? max_dg[cls] : 0.0) - dangle); // 0.02 kcal/mol: float32 rounding of the reference's sums, generously
		F.enabled[cls] = by_tm ? FILT_BY_TM : FILT_BY_DG;
	}
}

// Tables of the pre-filter's second tier (tnt_sbound.hip.h): per threshold class one entry per (x', x, q', q) -- the diagonal
// predecessor's and the cell's target base (A C G T, 4 = none) and query base -- holding (-delta_g) << 11 | credit for the three
// transitions of align_dimer that read the tables: diagonal step; into an aligned column from a gap state; from an aligned
// column into a gap state (read at the NEXT row's / the NEXT column's index: the credit is the closing column's stack with its
// diagonally adjacent later cell).  Credits follow fill_filter_tables (pair = query base * 7 + target
// base, phi(later cell, its diagonal predecessor)), in units of 0.02 kcal/mol rounded up.
// X = 5: target bases A C G T none (plain windows); X = 6: A C G T I none (the general instance: a column's code resolved against
// the query base, tnt_sbound.hip.h) into `entry` [FILT_CLASSES][X * X * 49][4].
template <int X> void fill_sbound_entries(int32_t (*entry)[X * X * NBP][4], const FilterTables &F, const Tables &T)
{
	constexpr int NONE = X - 1;
	bool wc[NBP];
	memset(wc, 0, sizeof(wc));
	wc[P_AT] = wc[P_TA] = wc[P_CG] = wc[P_GC] = true;
	for (int b = bA; b <= bI; b++) wc[b * NB + bI] = wc[bI * NB + b] = true;
	const long long K = 1ll << SB_SHIFT;
	auto cl = [](long long v) { return int32_t(std::max<long long>(v, SB_NEG)); };
	auto up = [](double kcal) { return int(std::ceil(kcal * double(SB_UNITS))); };
	for (int cls = 0; cls < FILT_CLASSES; cls++) {
		for (int i = 0; i < X * X * NBP; i++) entry[cls][i][0] = entry[cls][i][1] = entry[cls][i][2] = SB_NEG, entry[cls][i][3] = 0;
		if (!F.enabled[cls]) continue;
		const double Tk = double(F.tk[cls]);
		const double sc = double(T.salt) * double(T.log_na);
		auto phi = [&](int prev, int cur) { return -(double(T.H[prev * NBP + cur]) - Tk * (double(T.S[prev * NBP + cur]) + sc)); };
		const double pb1 = -Tk * double(T.bulge_S[1]);
		auto b_step = [&](int first, int second) { return (wc[first] || wc[second]) ? std::max(phi(first, second), 0.0) : 0.0; };
		auto b_gap = [&](int first, int second) {
			double best = b_step(first, second);
			if (wc[first] || wc[second])
				for (int pm = 0; pm < NBP; pm++)
					if (wc[pm] && pm / NB <= bI && pm % NB <= bI) best = std::max(best, std::max(phi(pm, second), phi(first, pm)) - pb1);
			return best;
		};
		for (int xp = 0; xp < X; xp++)
			for (int x = 0; x < X; x++)
				for (int qp = 0; qp < NB; qp++)
					for (int q = 0; q < NB; q++) {
						const int at = (xp * X + x) * NBP + qp * NB + q;
						const int xpB = xp == NONE ? int(bGAP) : xp, xB = x == NONE ? int(bGAP) : x;
						const bool v1 = q <= bI && x != NONE, v2 = qp <= bI && xp != NONE;
						const int first = q * NB + x, second = qp * NB + xp; // (read only where valid)
						const int cpair = xB * NB + q, ppair = xpB * NB + qp;
						const int cs = (v1 && v2) ? up(b_step(first, second)) : 0;
						const int cg = (v1 && v2) ? up(b_gap(first, second)) : (v1 ? up(b_gap(first, first)) : 0);
						const int cc = v2 ? (v1 ? up(b_gap(first, second)) : up(b_gap(second, second))) : 0;
						entry[cls][at][0] = x == NONE ? SB_NEG : cl(-(long long)T.dg[ppair * NBP + cpair] * K + cs);
						entry[cls][at][1] = x == NONE ? SB_NEG : cl(-(long long)T.term[cpair] * K + cg); // (first column: the kernel forbids the step out of a gap state itself)
						entry[cls][at][2] = cl(-(long long)T.term[xpB * NB + qp] * K + cc);
					}
	}
}

void fill_sbound_tables(SBoundTables &S, SBoundTablesD &D, const FilterTables &F, const Tables &T, const tnt_options &opt)
{
	memset(&S, 0, sizeof(S));
	memset(&D, 0, sizeof(D));
	const long long K = 1ll << SB_SHIFT;
	auto cl = [](long long v) { return int32_t(std::max<long long>(v, SB_NEG)); };
	S.bulge = int32_t(T.dg_const[0] * K), S.zero = int32_t(T.dg_const[1] * K);
	S.forb_m = cl(-(long long)T.dg_const[2] * K), S.forb_xy = cl(-(long long)T.dg_const[3] * K);
	// the largest gain of a column: the score of a window of SB_MAX_ROWS + 8 columns has to stay below 2^20 units
	long long gain = 0;
	for (int p = 0; p < NBP; p++)
		for (int c = 0; c < NBP; c++)
			if (p / NB <= bI && p % NB <= bI && c / NB <= bI && c % NB <= bI) gain = std::max<long long>(gain, -(long long)T.dg[p * NBP + c]);
	const bool fits = gain * (SB_MAX_ROWS + 2 * FLANK) < (1ll << 20) && std::abs((long long)T.dg_const[0]) < (1ll << 19) && std::abs((long long)T.dg_const[1]) < (1ll << 19);
	for (int cls = 0; cls < FILT_CLASSES; cls++)
		S.enabled[cls] = (F.enabled[cls] && fits && !opt.use_dinkelbach && !getenv("TNT_MI355X_NO_TIER2")) ? 1 : 0;
	fill_sbound_entries<SB_X>(S.entry, F, T);
	fill_sbound_entries<SBD_X>(D.entry, F, T);
}

// build_word_list<std::string>, seq_hash.h:287-374, in the packed stream's little-endian word convention
// (first base of the k-mer in the lowest two bits)
std::vector<uint16_t> word_list(const std::string &oligo, bool complement, int k)
{
	std::vector<uint16_t> w;
	if (size_t(k) > oligo.size()) return w;
	const uint32_t wmask = (1u << (2 * k)) - 1u;
	uint32_t word = 0;
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
		if (b >= 0) word = (word >> 2) | (uint32_t(b) << (2 * (k - 1)));
		else run = 0;
		if (run >= size_t(k)) w.push_back(uint16_t(word & wmask));
	}
	return w;
}

// minimal parallel-for on std::thread (host packing only; no OpenMP runtime dependency)
template <class F> void parallel_for(size_t n, F f)
{
	const unsigned hw = std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
	if (n < 4096 || hw == 1) {
		for (size_t i = 0; i < n; i++) f(i);
		return;
	}
	std::vector<std::thread> th;
	const size_t per = (n + hw - 1) / hw;
	for (unsigned w = 0; w < hw; w++) {
		const size_t b = w * per, e = std::min(n, b + per);
		if (b >= e) break;
		th.emplace_back([=]() {
			for (size_t i = b; i < e; i++) f(i);
		});
	}
	for (auto &t : th) t.join();
}

// the same for a few coarse jobs (no minimum count)
template <class F> void parallel_for_each(size_t n, F f)
{
	const unsigned hw = std::max(1u, std::min(32u, std::thread::hardware_concurrency()));
	std::vector<std::thread> th;
	std::atomic<size_t> next(0);
	for (unsigned w = 0; w < hw && w < n; w++)
		th.emplace_back([&]() {
			for (size_t i = next.fetch_add(1); i < n; i = next.fetch_add(1)) f(i);
		});
	for (auto &t : th) t.join();
}

template <class T> struct DevBuf {
	T *p = nullptr;
	size_t n = 0;
	hipError_t alloc(size_t count)
	{
		release();
		const hipError_t e = hipMalloc((void **)&p, std::max<size_t>(1, count) * sizeof(T));
		if (e != hipSuccess) p = nullptr; // a failed allocation leaves an empty buffer (n = 0): callers that keep buffers across
		else n = count;                  // searches and only grow them (`cap > buf.n`) allocate again instead of using a null pointer
		return e;
	}
	hipError_t upload(const std::vector<T> &v)
	{
		hipError_t e = alloc(v.size());
		if (e != hipSuccess) return e;
		if (!v.empty()) e = hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice);
		return e;
	}
	void release()
	{
		if (p) (void)hipFree(p);
		p = nullptr;
		n = 0;
	}
	~DevBuf() { release(); }
};

// page-locked host array that only grows (the long-target path moves tens of megabytes per search in both directions:
// from pageable memory those copies run at a tenth of the link's speed)
template <class T> struct PinnedBuf {
	T *p = nullptr;
	size_t cap = 0;
	hipError_t ensure(size_t n)
	{
		if (n <= cap) return hipSuccess;
		if (p) (void)hipHostFree(p);
		p = nullptr, cap = 0;
		const size_t want = n + n / 4 + 1024;
		const hipError_t e = hipHostMalloc(reinterpret_cast<void **>(&p), want * sizeof(T), hipHostMallocDefault);
		if (e == hipSuccess) cap = want;
		return e;
	}
	// the same, keeping the first `used` elements when the buffer is replaced
	hipError_t ensure_keep(size_t n, size_t used)
	{
		if (n <= cap) return hipSuccess;
		T *q = nullptr;
		const size_t want = n + n / 2 + 1024;
		const hipError_t e = hipHostMalloc(reinterpret_cast<void **>(&q), want * sizeof(T), hipHostMallocDefault);
		if (e != hipSuccess) return e;
		if (p && used) memcpy(q, p, used * sizeof(T));
		if (p) (void)hipHostFree(p);
		p = q, cap = want;
		return hipSuccess;
	}
	~PinnedBuf()
	{
		if (p) (void)hipHostFree(p);
	}
};

} // namespace

struct tnt_db {
	tnt_ctx *ctx = nullptr;
	std::vector<uint32_t> h_packed; // host copy (tnt_hit_sequence); fetched from the device on first use after a GPU ingest
	std::vector<uint64_t> h_word_off;
	std::vector<uint32_t> h_len, h_exc_off, h_exc_pos;
	std::vector<uint8_t> h_exc_code;
	bool host_copy = false;         // h_packed / h_exc_pos / h_exc_code are filled
	// seed work units: whole targets, or overlapping windows of long ones (SeedArgs, tnt_seed.hip.h)
	std::vector<uint32_t> h_unit_target, h_unit_begin, h_unit_end;
	std::vector<int32_t> h_unit_own_lo, h_unit_own_hi; // coordinates whose hits the unit reports
	DevBuf<uint32_t> unit_target, unit_begin, unit_end;
	DevBuf<uint4> unit_rec;         // whole-target units: {target, bases, first packed word (64 bits)}: one scalar load per unit
	DevBuf<int32_t> unit_own_lo, unit_own_hi;
	uint32_t max_unit_len = 0;
	uint32_t max_short_len = 0; // longest target that is searched as one work unit (the units the round-3 seed kernel sees)
	size_t n_short_units = 0;       // units [0, n_short_units) are whole targets, the rest windows of long targets
	std::vector<uint32_t> h_long_targets; // targets that are seeded in windows and paired on the host
	bool split = false;             // some target is cut into windows
	bool all_long = false;          // a whole-target work unit overflowed its seed list with one assay per pass: every target is seeded in windows
	std::vector<uint8_t> h_skip;    // per target: 1 = not searched (tnt_db_skip_targets)
	uint64_t n_exceptions = 0;
	std::string deflines;           // FASTA ingest: all deflines back to back
	std::vector<uint64_t> defline_off;
	std::vector<uint64_t> record_bytes; // FASTA / FASTQ ingest: bytes of every record, defline and line ends included
	tnt_ingest_timing ingest;
	DevBuf<uint32_t> packed, len, exc_off, exc_pos;
	DevBuf<uint64_t> word_off;
	DevBuf<uint8_t> exc_code;
	size_t n_targets = 0;
	uint64_t n_bases = 0;
	uint32_t max_len = 0;
	uint32_t first_seq_index = 0;
};

struct tnt_ctx {
	std::string err;
	int device = 0;
	tnt_options opt;
	hipStream_t stream = nullptr;
	// The length buckets of a phase are independent launches: they run on BIND_LANES side streams (each with its
	// own scratch slabs) so that the drain of one launch overlaps the ramp-up of the next.  Three by default: the runtime maps a
	// process's streams onto four hardware queues, the context's own stream holds one, and two streams on one queue run one after
	// the other -- a fourth side stream was sharing a queue with the first (profiles/r06s_timeline_config2.txt: four of config 2's
	// eight bucket chains on one queue, the other queues idle for the last quarter of the phase).
	static constexpr int BIND_LANES = 8; // at most; `lanes` of them are used (TNT_MI355X_BIND_LANES, default 3)
	int lanes = 3;
	hipStream_t lane_stream[BIND_LANES] = {};
	hipEvent_t lane_done[BIND_LANES] = {};
	bool split_filter = false;                 // all pre-filter launches on lane 0, the bind launches behind them on the other lanes (TNT_MI355X_SPLIT_FILTER)
	hipEvent_t filtered[MAX_BUCKETS] = {};      // split_filter: bucket i has met its pre-filter
	hipEvent_t phase_ready = nullptr;
	// how long the launch chain of every bucket took in phase A of the last batch (events around the chain on its side stream):
	// the next batch deals its buckets to the streams by these (launch_bind_phase)
	hipEvent_t chain_ev[MAX_BUCKETS][2] = {};
	bool chain_timed[MAX_BUCKETS] = {};
	double chain_ms[MAX_BUCKETS] = {};
	size_t lane_trace = 0, lane_cols = 0, lane_ties = 0; // scratch elements per lane

	uint64_t long_pair_stats[4] = {0, 0, 0, 0}; // tnt_long_pair_stats: the lists of windowed targets in the last search
	std::string warnings; // what a caller should know about this context's options (tnt_ctx_warnings): the Q11 regime
	std::vector<std::string> oligos;
	std::vector<AssayDev> h_assays;
	std::vector<Slot> h_slots;
	int lmax = 24; // largest bind kernel instance this context needs
	std::vector<int> bucket_lmax; // the instances in use, ascending; Slot::bucket indexes this list
	int scratch_lmax = 24; // what the scratch slabs are sized for
	size_t n_entries = 0;
	std::vector<uint32_t> chunk_begin; // assay chunks for the seed kernel (+ end)

	Tables h_tab;
	DevBuf<Tables> tab;
	// pre-filter of the bind stage (tnt_filter.hip.h)
	FilterTables h_ftab;
	DevBuf<FilterTables> ftab;
	bool filter_on = false, filter_audit = false;
	bool filter_long = true; // the 128-row pre-filter instances are worth their time (calibrate_long_filter)
	float filter_long_rate = -1.0f; // share of the calibration windows they rejected (-1: not measured)
	bool filter_status_exact = false; // set around launches whose caller reports statuses (tnt_bind_batch_routed)
	int wave_from = 0;        // buckets of this many rows or more are bound by the wavefront-per-window kernel (tnt_wbind.hip.h); 0: none (TNT_MI355X_WAVE_BIND)
	bool serial_bind = false; // profiling: every launch of the bind stage on one side stream (per-kernel durations add up to the stage time)
	DevBuf<uint8_t> olig_base;
	DevBuf<uint32_t> olig_q4;
	DevBuf<int32_t> olig_len;
	DevBuf<Slot> slots;
	// shared oligos (degenerate expansion, --plex): slot -> first slot with the same (oligo, log_ct, thresh); items
	// that repeat an earlier item of their target take its bind result (the reference's per-target BindCache, a6)
	std::vector<uint32_t> h_slot_canon, h_slot_next;
	DevBuf<uint32_t> slot_canon, slot_next;
	bool dedupe = false;
	int unit_rows = 64; // rows of the unit-level kernel instance this context's oligos need (64 or MAX_OLIGO)
	bool filter_general = false; // the general buckets (degenerate windows, merged short queues) meet the pre-filter too
	bool seed_lds = false; // the seed kernel instance with LDS-resident tables serves this context
	// round-3 seed kernel of the search path (tnt_seed2.hip.h): items / groups in regions private to a wavefront, phase-A
	// items straight into the queue of their length bucket
	bool seed2 = false;
	size_t item_cap = 0, group_cap = 0; // capacities of the work buffers (ensure_work_buffers)
	bool seed2_big = false;          // the assay set needs many chunks: the seed kernel's instances with the large lists (S2_CAP_BIG, plan())
	int seed2_wgs = TNT_S2_OCC;      // workgroups of the round-3 seed kernel per CU (persistent grid; TNT_MI355X_SEED2_WGS: fewer leave LDS and registers to the bind stage of the batch before, with TNT_MI355X_PIPELINE)
	int pipeline_batches = 1;        // batches a large search is cut into on two buffer sets (TNT_MI355X_PIPELINE; 1: one batch after the other on one stream -- measured faster)
	static constexpr uint32_t MAX_REGIONS = 8192;
	DevBuf<uint32_t> slot_pack;      // [slot] bucket | bucket_general << 8 | oligo length << 16
	// Round 5: the phase-A queue of a plain length bucket is regrouped by (oligo, threshold class) in front of its pre-filter
	// (tnt_ufilter.hip.h): slot -> rank of its (oligo, class) inside its bucket, the buckets' shares of the histogram / cursor
	// arrays and of the regrouped queue (every run is padded to whole blocks: 64 slots per bin on top of the bucket's region)
	// second tier of the pre-filter (tnt_sbound.hip.h): the credit of the fill's maximal-score paths
	SBoundTables h_stab;
	DevBuf<SBoundTables> stab;
	std::unique_ptr<SBoundTablesD> h_dstab; // ... of the general instance (a column's code resolved against the query base)
	DevBuf<SBoundTablesD> dstab;
	bool tier2 = false;
	bool defer_finish = false; // the throughput bind instances park the windows that pass Tm / dG for bind_finish_kernel (tnt_bind.hip.h)
	bool regroup = false;
	DevBuf<uint32_t> slot_bin;       // [slot][2]: inside its plain bucket, inside its general bucket
	std::vector<uint32_t> h_slot_bin, rg_bins, rg_bin_base; // [slot][2]; [bucket] bins, first bin
	uint32_t rg_total_bins = 0;
	std::vector<RegroupBucket> h_rg_desc; // the buckets of a batch that are regrouped (planned with the queue regions)
	std::vector<int> rg_desc_of;          // [bucket] -> its descriptor, or -1
	DevBuf<RegroupBucket> rg_desc;
	DevBuf<uint32_t> rg_bin_base_dev; // [bucket] first group of the bucket (the seed kernel counts the entries per group where the groups fit its LDS histogram)
	bool rg_seed_counts = false;      // ... so the regrouping runs without its counting pass
	DevBuf<uint32_t> q_base, q_cap;  // [bucket] region of every phase-A queue inside `qa`
	std::vector<uint32_t> h_q_base, h_q_cap;
	bool q_planned_exceptions = false;
	uint32_t batch_regions = 0;      // regions (wavefronts) of the batch that ran last
	bool batch_direct = false;       // ... and whether it went through seed2_kernel (per-wavefront regions) or the round-2 kernel (32 shards)
	DevBuf<Thresh> thresh;
	std::vector<Thresh> h_thresh;       // (host copy: the cache replay of windowed targets runs on host threads)
	DevBuf<AssayDev> assays;
	DevBuf<uint32_t> wt_ent, wl_off;
	DevBuf<unsigned long long> wl_seq;
	std::vector<unsigned long long> h_wl_seq; // host copy: the per-entry copies of a chunk's table are made from it (plan)
	DevBuf<uint4> wt_rec;                     // [entry] {entry, first 32 bases of its oligo, 0}: the dense pass's one load per candidate
	DevBuf<unsigned long long> wt_seq;        // [entry][2] the packed oligo of every table entry (the round-3 seed kernel reads entry and oligo with one index)
	DevBuf<uint16_t> wt_off, wlist;
	std::vector<std::vector<std::pair<uint16_t, uint32_t>>> assay_entries; // per assay: (word, entry bits)
	std::vector<uint32_t> chunk_ent_base;  // first entry of every chunk in wt_ent
	std::vector<uint32_t> planned_chunks;  // chunk_begin the device tables were built for

	// Work buffers of one internal batch.  Two sets: the seed stage of batch n + 1 runs on a stream of its own and fills one
	// set while the bind and pair stages of batch n work on the other (tnt_search_enqueue).
	struct BatchSet {
		DevBuf<Item> items;
		DevBuf<int32_t> result;
		DevBuf<Group> groups;
		DevBuf<uint32_t> active, qa, qb;
		DevBuf<uint32_t> pairable; // the groups of `active` that can report a hit (pair_select_kernel)
		DevBuf<uint32_t> qbucket; // the phase queue regrouped by oligo length, bucket after bucket
		DevBuf<uint32_t> qsurv;   // the bucket queue after the filter, bucket by bucket at the same offsets
		DevBuf<uint32_t> qsurv2;  // ... after the filter's second tier
		DevBuf<uint4> qfat;       // phase-A queues of the round-3 seed kernel: self-contained entries (FilterArgs2::fat)
		DevBuf<uint4> qsort;      // ... regrouped by (oligo, class): bucket i at q_base[i] + 64 * rg_bin_base[i] (UFilterArgs::fat)
		DevBuf<uint32_t> rg_hist, rg_cursor; // [rg_total_bins] entries per bin of the batch, next free slot of its run
		DevBuf<uint32_t> rg_fill;            // ... survivors of the pre-filter's first tier handed to the second (by group: tnt_sbound.hip.h)
		DevBuf<uint8_t> audit;    // audit mode: 1 = the filter would reject this item
		DevBuf<uint32_t> item_rep;
		DevBuf<BoundRec> pool;
		DevBuf<PendingRec> pending; // windows behind the Tm / dG tests, waiting for bind_finish_kernel
		DevBuf<uint32_t> counters;       // per-batch counters; the search-wide ones (hits, error word, totals) live in set 0's
		DevBuf<uint32_t> region_counts;  // [2 * MAX_REGIONS] items, groups of every region
		DevBuf<uint32_t> q_state;        // [MAX_REGIONS][MAX_BUCKETS][2]
		hipEvent_t seeded = nullptr, released = nullptr; // the seed stage filled the set / the pair stage is done with it
		bool used = false;
	};
	BatchSet set[2];
	int cur = 0;
	BatchSet &S() { return set[cur]; }
	uint32_t *G() { return set[0].counters.p; } // search-wide counters
	hipStream_t seed_stream = nullptr;
	bool dev_hits_valid = false; // ctx->hits still equals the host copy of the last collected search
	size_t dev_hits_n = 0;
	DevBuf<uint8_t> hit_blob;   // compact form of the hits (tnt_search_hits_compact)
	DevBuf<uint32_t> hit_sizes;
	DevBuf<Hit> hits;
	DevBuf<uint8_t> trace;
	DevBuf<uint16_t> ties;
	DevBuf<uint8_t> best;
	DevBuf<uint32_t> stack;
	DevBuf<SeedOut> seeds_out;
	DeviceArena long_arena; // device arrays of the long-target path (lists, scans, kept seeds)
	DeviceArena long_pair_arena; // ... of the device pairing of its PCR lists (tnt_sort.hip, long_pair_device)
	DevBuf<BoundRec> long_pool;  // the bound records of all bind passes of a long-target search, for the device pairing
	int bind_blocks = 0; // grid of the 256-thread instances
	int n_cu = 0;

	tnt_hit *h_hits = nullptr; // pinned host buffer the hit records are copied into (library owned)
	PinnedBuf<Item> pin_items, pin_bind; // long-target path: the sorted seeds, the work items of the kept ones
	PinnedBuf<uint32_t> pin_list;
	PinnedBuf<int32_t> pin_result;
	PinnedBuf<BoundRec> pin_pool;        // the bound oligos of all bind passes of a long-target search
	PinnedBuf<uint8_t> pin_blob;         // tnt_search_collect_compact: the hits of the last search in wire form
	// Row a6: assays that hold an oligo used under two strand concentrations / threshold sets go through the replay of the
	// reference's per-target BindCache (tnt_pair.hip.h, cache_replay_kernel) instead of the ordinary pairing
	// Oligos of 129 to 1 024 bases (probes of probe-only assays): the 7-bit word index of a work item reaches 128 query words, so
	// such a probe is seeded in segments of 128 words -- every segment behind the first a VIRTUAL assay of its own (same id /
	// degen_id, same oligo, Slot::qloc_bias = 128 * segment) appended behind the caller's assays.  Two seeds of one diagonal in
	// two segments bind the same window; the hits of the segments are merged when a search is collected (merge_long_probe_hits).
	std::vector<int32_t> h_segment;      // per device assay: segment of its probe's word list (-1: the whole list)
	std::vector<uint32_t> h_assay_base;  // per device assay: the assay whose lists its seeds join (itself; the first segment's assay for the segments of long primers)
	bool long_primers = false;           // a primer, TaqMan probe or padlock arm of more than 128 bases: every target is seeded in windows and its lists are assembled, culled and paired whole on the host (tnt_long.hip.h)
	size_t n_input_assays = 0;
	bool long_probes = false;
	DevBuf<char> ltext;                  // alignment texts beyond TNT_ALIGN_TEXT (tnt_long_text)
	std::vector<char> h_ltext;           // ... on the host, as of the last collected search / bind call
	bool cache_replay = false;
	uint32_t n_cache_assays = 0;
	DevBuf<unsigned long long> ck_in, ck_out; // sort keys of their groups (target << 32 | assay)
	DevBuf<uint32_t> cv_in, cv_out, c_origin, c_counts; // c_counts: {groups, items, pool records}
	DevBuf<ConflictRec> c_recs;
	DevBuf<Item> c_items;
	DevBuf<BindFull> c_full;
	DevBuf<BoundRec> c_pool;
	DevBuf<uint8_t> c_sort_tmp;
	bool compact_in_search = false;      // the compaction kernels run behind the pair stage of every search (set by the first request for the compact form)
	size_t blob_cap_hits = 0;            // records the blob of the pending search was sized for (0: not packed with the search)
	size_t blob_hits = 0, blob_bytes = 0; // what ctx->hit_blob holds: valid while blob_hits == dev_hits_n and blob_bytes != 0
	size_t h_hits_cap = 0;
	std::vector<tnt_seed> h_seeds;
	std::vector<int32_t> h_cull_rows; // tnt_first_cull
	tnt_timing timing;
	std::vector<hipEvent_t> events;
	size_t n_batches = 0;
	uint64_t acc_items = 0, acc_bound = 0;
	bool pending = false;
	tnt_db *pending_db = nullptr;
	std::vector<tnt_db *> dbs; // live databases of this context: detached (not freed) when the context goes first
	size_t hit_cap = size_t(1) << 20; // initial hit capacity; grown on demand (tnt_search_collect)
	int retries = 0;
	double batch_scale = 1.0;
	bool force_big = false;   // launch the crowded-group instance of the pair kernel whatever the expected seed density
	double unit_scale = 1.0;  // stride of the seed windows of long targets relative to the planned one (halved when single-assay chunks still overflow)
	double chunk_scale = 1.0; // assay-chunk size of the seed kernel relative to the planned one (halved after a seed-list overflow) // targets per internal batch relative to the planned number (halved after an overflow)

	~tnt_ctx()
	{
		for (hipEvent_t e : events) (void)hipEventDestroy(e);
		if (h_hits) (void)hipHostFree(h_hits);
		for (hipEvent_t ev : filtered)
			if (ev) (void)hipEventDestroy(ev);
		for (int i = 0; i < BIND_LANES; i++) {
			if (lane_done[i]) (void)hipEventDestroy(lane_done[i]);
			if (lane_stream[i]) (void)hipStreamDestroy(lane_stream[i]);
		}
		if (phase_ready) (void)hipEventDestroy(phase_ready);
		for (auto &pair : chain_ev)
			for (hipEvent_t ev : pair)
				if (ev) (void)hipEventDestroy(ev);
		for (BatchSet &W : set) {
			if (W.seeded) (void)hipEventDestroy(W.seeded);
			if (W.released) (void)hipEventDestroy(W.released);
		}
		if (seed_stream) (void)hipStreamDestroy(seed_stream);
		if (stream) (void)hipStreamDestroy(stream);
	}
};

namespace {

enum { C_POOL = 0, C_ACTIVE, C_QB, C_BIG /* groups of more than PAIR_MAX items */, C_HITS, C_ERR, C_DUPS /* phase-A items bound through a representative */, C_FILTERED /* windows rejected by the pre-filter (whole search) */, C_TOT_ITEMS_LO, C_TOT_ITEMS_HI, C_TOT_GROUPS_LO, C_TOT_GROUPS_HI, C_TOT_BINDS_LO, C_TOT_BINDS_HI, C_TOT_BOUND, C_SEEDS, C_PAIRABLE /* groups handed to the pair kernels */, C_PENDING /* windows parked for bind_finish_kernel (per bind phase) */,
       C_TOT_CELLS_LO = 18, C_TOT_CELLS_HI /* DP cells of the windows handed to the bind stage, sum of L x W (whole search, 64 bits) */,
       // sharded counters of the seed kernel (N_SHARD * SHARD_STRIDE words each)
       C_ITEMS = 32, C_QA = C_ITEMS + N_SHARD * SHARD_STRIDE, C_GROUPS = C_QA + N_SHARD * SHARD_STRIDE,
       // per length bucket: entries of the current phase queue
       C_BUCKET = C_GROUPS + N_SHARD * SHARD_STRIDE, C_DEBUG = C_BUCKET + MAX_BUCKETS * BUCKET_STRIDE,
       C_BLOB = C_DEBUG + 16 /* compact hit blob of the search: bytes, hits (enqueue_compaction) */,
       C_LTEXT = C_BLOB + 2 /* bytes of the long-text arena handed out */, C_COUNT = C_LTEXT + 2 };
static_assert(C_TOT_ITEMS_LO % 2 == 0 && C_TOT_GROUPS_LO % 2 == 0 && C_TOT_BINDS_LO % 2 == 0 && C_TOT_CELLS_LO % 2 == 0 && C_TOT_CELLS_HI < 32, "64-bit totals must be 8-byte aligned");

// per-batch counters -> running totals of the whole search (the per-batch ones restart every batch)
__global__ void accumulate_kernel(uint32_t *cnt, uint32_t *g) // cnt: the batch's counters, g: the search-wide ones
{
	unsigned long long *ti = reinterpret_cast<unsigned long long *>(g + C_TOT_ITEMS_LO);
	unsigned long long *tg = reinterpret_cast<unsigned long long *>(g + C_TOT_GROUPS_LO);
	unsigned long long *tb = reinterpret_cast<unsigned long long *>(g + C_TOT_BINDS_LO);
	for (uint32_t r = 0; r < N_SHARD; r++) {
		*ti += cnt[C_ITEMS + r * SHARD_STRIDE];
		*tg += cnt[C_GROUPS + r * SHARD_STRIDE];
		*tb += cnt[C_QA + r * SHARD_STRIDE];
	}
	*tb += cnt[C_QB];
	*tb -= cnt[C_DUPS];
	cnt[C_DUPS] = 0;
	g[C_TOT_BOUND] += cnt[C_POOL];
}

// q[i] = i for i < n, *count = n: the queue of a bind phase over every item of a buffer (long-target path)
// results of one bind pass of the long-target path -> the array over all passes (pool indices counted from the first pass)
__global__ void rebase_kernel(const int32_t *res, uint32_t n, int32_t base, int32_t *out)
{
	for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
		const int32_t r = res[i];
		out[i] = r >= 0 ? r + base : r;
	}
}
__global__ void iota_kernel(uint32_t *q, uint32_t n, uint32_t *count)
{
	for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) q[i] = i;
	if (blockIdx.x == 0 && threadIdx.x == 0) *count = n;
}

static size_t cap_from_env(const char *name, size_t dflt_m)
{
	const char *e = getenv(name);
	const long v = e ? atol(e) : 0;
	return size_t(v >= 16 && v <= 2048 ? size_t(v) : dflt_m) << 20;
}
static const size_t ITEM_CAP_MAX = cap_from_env("TNT_MI355X_ITEM_CAP_M", 768);  // bind work items per internal batch: the most a context allocates
static const size_t GROUP_CAP_MAX = cap_from_env("TNT_MI355X_GROUP_CAP_M", 256); // (tnt_ctx::item_cap / group_cap: what it has, sized by the databases it met)
constexpr size_t POOL_CAP = size_t(4) << 20;
constexpr size_t PACKED_SLACK = 16; // zero words behind the packed targets: a window loader reads (MAX_OLIGO + 2 * FLANK + 15) / 16 + 1 = 10 words from its first word

DbView make_view(const tnt_db *db, size_t t0, size_t t1)
{
	DbView v;
	v.packed = db->packed.p;
	v.word_off = db->word_off.p + t0;
	v.len = db->len.p + t0;
	v.exc_off = db->exc_off.p + t0;
	v.exc_pos = db->exc_pos.p;
	v.exc_code = db->exc_code.p;
	v.n_targets = uint32_t(t1 - t0);
	v.first_seq_index = db->first_seq_index + uint32_t(t0);
	return v;
}

// Template instances of the bind kernel (TNT_LMAX_LADDER, tnt_dev.h) live in tnt_bind_inst.hip so that groups
// of them compile in parallel; the unit-level instance bind_kernel<64, true> is compiled here.
int pick_lmax(size_t max_len)
{
#define X(N) if (max_len <= N) return N;
	TNT_LMAX_LADDER(X)
	TNT_LONG_LADDER(X)
#undef X
	return BUCKET_WAVE | wb_rows_per_lane(int(max_len)); // beyond 128 bases: the wavefront-per-window kernel (4, 8, 16 rows per lane)
}

// persistent grid of one bind kernel instance
// `id` = rows | BUCKET_* flags of an instance (0: the unit-level instance)
int bind_threads(int id)
{
	if (id & BUCKET_WAVE) return 64 * wb_waves(id & 0xFF);
	return bind_threads_for(id ? (id & 0xFF) : 64, (id & BUCKET_PLAIN) != 0);
}
int bind_grid(const tnt_ctx *c, int id)
{
	if (id & BUCKET_WAVE) return c->n_cu * wb_blocks_per_cu(id & 0xFF) * 2;
	return c->n_cu * bind_blocks_per_cu(id & 0xFF, (id & BUCKET_PLAIN) != 0);
}

// instance id for plain windows of oligos of length L: the exact instance where there is one, else the ladder
int pick_instance(size_t L)
{
#define X(N) if (L == N) return N | BUCKET_EXACT | BUCKET_PLAIN;
	TNT_EXACT_LADDER(X)
#undef X
	if (L > size_t(MAX_LANE_OLIGO)) return pick_lmax(L); // (one instance for every window of such an oligo)
	return pick_lmax(L) | BUCKET_PLAIN;
}

void launch_wbind(tnt_ctx *c, const BindArgs &a, int rows, bool full, hipStream_t stream, size_t n_items = 0);

void launch_bind(tnt_ctx *c, const BindArgs &a, int blocks, int lmax, hipStream_t stream = nullptr, int unit_rows = 0 /* 0: the context's */)
{
	if (!stream) stream = c->stream;
	if (c->opt.use_dinkelbach) {
		// every bind through an instance that iterates the temperature: the 64-row general lane instances, or -- oligos beyond 64
		// bases -- the wavefront-per-window kernel, whose workgroup (one wavefront) rebuilds its delta_g planes per pass
		const int rows = lmax == 0 ? (unit_rows ? unit_rows : c->unit_rows) : bucket_rows(lmax);
		if (rows > 64) {
			// one wavefront per workgroup, one trace slab per wavefront: the grid is what ensure_scratch sized a lane's slabs for
			const int R = wb_rows_per_lane(rows), wblocks = bind_grid(c, BUCKET_WAVE | R) * wb_waves(R);
			switch (R) {
#define X(N) case N: tnt::launch_wbind_dink_##N(a, wblocks, lmax == 0, stream); break;
				TNT_WBIND_LADDER(X)
#undef X
			default: break;
			}
		} else if (lmax == 0) {
			hipLaunchKernelGGL((bind_kernel<64, true, false, false, true>), dim3(blocks), dim3(bind_threads(0)), 0, stream, a);
		} else {
			hipLaunchKernelGGL((bind_kernel<64, false, false, false, true>), dim3(blocks), dim3(bind_threads(64)), 0, stream, a);
		}
		return;
	}
	if (lmax == 0) { // unit-level API: the 64-row instance, the 128-row one, or the wavefront-per-window kernel when an oligo needs it
		const int rows = unit_rows ? unit_rows : c->unit_rows;
		if (rows > MAX_LANE_OLIGO) launch_wbind(c, a, rows, true, stream);
		else if (rows > 64) hipLaunchKernelGGL((bind_kernel<MAX_LANE_OLIGO, true>), dim3(blocks), dim3(bind_threads(0)), 0, stream, a);
		else hipLaunchKernelGGL((bind_kernel<64, true>), dim3(blocks), dim3(bind_threads(0)), 0, stream, a);
		return;
	}
	if (lmax & BUCKET_WAVE) {
		launch_wbind(c, a, bucket_rows(lmax), false, stream);
		return;
	}
	switch (lmax) {
#define X(N) case N: tnt::launch_bind_##N(a, blocks, stream); break;
		TNT_LMAX_LADDER(X)
		TNT_LONG_LADDER(X)
#undef X
#define X(N) case N | BUCKET_PLAIN: tnt::launch_bind_plain_##N(a, blocks, stream); break;
		TNT_LMAX_LADDER(X)
		TNT_LONG_LADDER(X)
#undef X
#define X(N) case N | BUCKET_EXACT | BUCKET_PLAIN: tnt::launch_bind_exact_##N(a, blocks, stream); break;
		TNT_EXACT_LADDER(X)
#undef X
	default: break;
	}
}

// the wavefront-per-window form of the bind kernel (tnt_wbind.hip.h): rows per lane by the longest oligo of the launch
void launch_wbind(tnt_ctx *c, const BindArgs &a, int rows, bool full, hipStream_t stream, size_t n_items /* 0: a full queue */)
{
	const int R = wb_rows_per_lane(rows);
	int blocks = c->n_cu * wb_blocks_per_cu(R) * 2; // persistent grid: 2 x the resident workgroups
	if (n_items) blocks = int(std::min<size_t>(size_t(blocks), (n_items + size_t(wb_waves(R)) - 1) / size_t(wb_waves(R))));
	switch (R) {
#define X(N) case N: if (full) tnt::launch_wbind_full_##N(a, blocks, stream); else tnt::launch_wbind_##N(a, blocks, stream); break;
		TNT_WBIND_LADDER(X)
#undef X
	default: break;
	}
}

// is there a pre-filter instance of this many rows (degen: for windows with any target code)?
bool has_filter_instance(int rows, bool degen)
{
	if (degen) {
		switch (rows) {
#define X(N) case N: return true;
			TNT_LMAX_LADDER(X)
			TNT_LONG_LADDER(X)
#undef X
		default: return false;
		}
	}
	switch (rows) {
#define X(N) case N: return true;
		TNT_FILTER_LADDER(X)
		TNT_LONG_LADDER(X)
#undef X
	default: return false;
	}
}

void launch_filter(const FilterArgs2 &a, int blocks, int rows, hipStream_t stream, bool degen = false /* windows with any target code */)
{
	if (degen) {
		switch (rows) {
#define X(N) case N: tnt::launch_filter_degen_##N(a, blocks, stream); break;
			TNT_LMAX_LADDER(X)
			TNT_LONG_LADDER(X)
#undef X
		default: break;
		}
		return;
	}
	switch (rows) {
#define X(N) case N: tnt::launch_filter_##N(a, blocks, stream); break;
		TNT_FILTER_LADDER(X)
		TNT_LONG_LADDER(X)
#undef X
	default: break;
	}
}

// the pre-filter on a queue regrouped by oligo (tnt_ufilter.hip.h): instances of up to 64 rows
bool has_ufilter_instance(int rows, bool degen)
{
	if (degen) {
		switch (rows) {
#define X(N) case N: return true;
			TNT_LMAX_LADDER(X)
#undef X
		default: return false;
		}
	}
	switch (rows) {
#define X(N) case N: return N <= 64;
		TNT_FILTER_LADDER(X)
#undef X
	default: return false;
	}
}

void launch_ufilter(const UFilterArgs &a, int blocks, int rows, hipStream_t stream, bool degen)
{
	if (degen) {
		switch (rows) {
#define X(N) case N: tnt::launch_ufilter_degen_##N(a, blocks, stream); break;
			TNT_LMAX_LADDER(X)
#undef X
		default: break;
		}
		return;
	}
	switch (rows) {
#define X(N) case N: tnt::launch_ufilter_##N(a, blocks, stream); break;
		TNT_FILTER_LADDER(X)
#undef X
	default: break;
	}
}

bool has_sbound_instance(int rows)
{
	switch (rows) {
#define X(N) case N: return true;
		TNT_SBOUND_LADDER(X)
#undef X
	default: return false;
	}
}

void launch_sbound(const SBoundArgs &a, int blocks, int rows, hipStream_t stream)
{
	switch (rows) {
#define X(N) case N: tnt::launch_sbound_##N(a, blocks, stream); break;
		TNT_SBOUND_LADDER(X)
#undef X
	default: break;
	}
}

bool has_sbound_degen_instance(int rows)
{
	switch (rows) {
#define X(N) case N: return true;
		TNT_SBOUND_DEGEN_LADDER(X)
#undef X
	default: return false;
	}
}

void launch_sbound_degen(const SBoundDegenArgs &a, int blocks, int rows, hipStream_t stream)
{
	switch (rows) {
#define X(N) case N: tnt::launch_sbound_degen_##N(a, blocks, stream); break;
		TNT_SBOUND_DEGEN_LADDER(X)
#undef X
	default: break;
	}
}

// The regrouping of a batch's phase-A queues by (oligo, threshold class): count, scan, scatter, each ONE launch over all the
// buckets of `desc` (device array of n_desc descriptors).
void launch_regroup(tnt_ctx *c, const RegroupBucket *desc, uint32_t n_desc, const uint4 *src, uint4 *dst, uint32_t *counters, const uint32_t *slot_bin,
                    uint32_t *hist, uint32_t *cursor, hipStream_t stream, bool counted = false /* the histogram is filled already (by the seed kernel) */)
{
	RegroupArgs r;
	memset(&r, 0, sizeof(r));
	r.buckets = desc, r.n_buckets = n_desc;
	r.src = src, r.dst = dst, r.counters = counters;
	r.slot_bin = slot_bin, r.hist = hist, r.cursor = cursor;
	if (!counted) hipLaunchKernelGGL(regroup_count_kernel, dim3(c->n_cu * 8), dim3(RG_THREADS), 0, stream, r);
	hipLaunchKernelGGL(regroup_scan_kernel, dim3(n_desc), dim3(1024), 0, stream, r);
	hipLaunchKernelGGL(regroup_scatter_kernel, dim3(c->n_cu * 8), dim3(RG_THREADS), 0, stream, r);
}

template <bool LDS_TABLES> void launch_seed(const SeedArgs &s, int mode, int blocks, int threads, hipStream_t stream)
{
	if (mode == SEED_LIST) hipLaunchKernelGGL((seed_kernel<LDS_TABLES, SEED_LIST>), dim3(blocks), dim3(threads), 0, stream, s);
	else if (mode == SEED_SEARCH_SHARED) hipLaunchKernelGGL((seed_kernel<LDS_TABLES, SEED_SEARCH_SHARED>), dim3(blocks), dim3(threads), 0, stream, s);
	else hipLaunchKernelGGL((seed_kernel<LDS_TABLES, SEED_SEARCH>), dim3(blocks), dim3(threads), 0, stream, s);
}

// One phase of binding: split the queue by oligo length, then one launch per length bucket.
// `direct`: the queues of the length buckets are in place already (the round-3 seed kernel writes its phase-A items straight
// into them, region q_base[i] of c->S().qa, slots handed out in C_BUCKET + i * BUCKET_STRIDE; unused slots hold QUEUE_HOLE).
void launch_bind_phase(tnt_ctx *c, BindArgs b, const DbView &v, bool has_exceptions, const uint32_t *src, const uint32_t *src_count,
                       const uint32_t *shard_counts, int merge_mode = 0 /* 0: short queues go to the longest general instance; 1: never; 2: always; 3: queues up to 8 x as long */,
                       bool direct = false)
{
	if (c->opt.use_dinkelbach) merge_mode = 2; // the Dinkelbach passes only exist in the longest general instance
	uint32_t *cnt = c->S().counters.p, *gcnt = c->G();
	if (!direct) {
		(void)hipMemsetAsync(cnt + C_BUCKET, 0, sizeof(uint32_t) * MAX_BUCKETS * BUCKET_STRIDE, c->stream);
		BucketArgs k;
		memset(&k, 0, sizeof(k));
		k.src = src;
		k.src_count = src_count;
		k.shard_counts = shard_counts;
		k.shard_region = uint32_t(c->item_cap / N_SHARD);
		k.items = c->S().items.p;
		k.slots = b.slots;
		k.olig_len = c->olig_len.p;
		k.db = v;
		k.has_exceptions = has_exceptions ? 1 : 0;
		k.dst = c->S().qbucket.p;
		k.cap = uint32_t(c->item_cap);
		k.counters = cnt + C_BUCKET;
		k.n_buckets = int(c->bucket_lmax.size());
		k.merge_below = uint32_t(bind_grid(c, c->bucket_lmax.back()) * bind_threads(c->bucket_lmax.back()));
		if (merge_mode == 1) k.merge_below = 0;
		if (merge_mode == 2) k.merge_below = 0xFFFFFFFFu;
		// phase B of a search: few of its windows pass the pre-filter, and a launch per length class costs its start-up
		// and the serial tail of its slowest lane (0.2 - 0.3 ms each, eighteen of them on config 2) whatever it binds
		if (merge_mode == 3) k.merge_below *= 8;
		k.err = gcnt + C_ERR;
		k.item_rep = (c->dedupe && shard_counts) ? c->S().item_rep.p : nullptr; // phase A only: phase-B items are never duplicates
		k.cells = reinterpret_cast<unsigned long long *>(gcnt + C_TOT_CELLS_LO);
		hipLaunchKernelGGL(bucket_kernel<false>, dim3(1024), dim3(256), 0, c->stream, k);
		hipLaunchKernelGGL(bucket_kernel<true>, dim3(1024), dim3(256), 0, c->stream, k);
	}
	const bool deferred = c->defer_finish && c->S().pending.p;
	if (deferred) {
		(void)hipMemsetAsync(cnt + C_PENDING, 0, sizeof(uint32_t), c->stream);
		b.pending = c->S().pending.p, b.pending_cap = uint32_t(c->S().pending.n), b.n_pending = cnt + C_PENDING;
	} else {
		b.pending = nullptr, b.pending_cap = 0, b.n_pending = nullptr;
	}
	const bool regrouped = direct && c->regroup && c->filter_on && !c->h_rg_desc.empty() && c->S().qsort.p;
	if (regrouped) // the phase-A queues of the batch, windows of one oligo side by side (tnt_ufilter.hip.h)
		launch_regroup(c, c->rg_desc.p, uint32_t(c->h_rg_desc.size()), c->S().qfat.p, c->S().qsort.p, cnt + C_BUCKET, c->slot_bin.p, c->S().rg_hist.p, c->S().rg_cursor.p, c->stream, c->rg_seed_counts);
	(void)hipEventRecord(c->phase_ready, c->stream);
	auto LS = [&](int l) { return c->lane_stream[l]; };
	for (int l = 0; l < c->lanes; l++) (void)hipStreamWaitEvent(c->lane_stream[l], c->phase_ready, 0);
	int n_launch = 0;
	// Which side stream takes which bucket.  The chains of the buckets (pre-filter -> second tier -> bind) run beside each other
	// and fill each other's tails -- as long as the streams end together: the phase waits for the last one.  Round 6: the buckets
	// are dealt by expected work (oligo slots of the bucket x rows x columns; the windows per oligo are alike), longest first to
	// the least loaded stream, instead of round robin (which left one stream with four of config 2's eight buckets and the
	// others idle for the last quarter of the phase: profiles/r06s_timeline_config2.txt).
	auto launches = [&](size_t i) {
		if (!has_exceptions && !(c->bucket_lmax[i] & BUCKET_PLAIN) && i + 1 < c->bucket_lmax.size()) return false; // general instances other than the last (merge) bucket only see windows with degenerate bases
		if (direct && c->h_q_cap[i] == 0) return false; // no slot routes its items here
		return true;
	};
	std::vector<int> lane_of(c->bucket_lmax.size(), 0);
	if (!c->serial_bind && c->lanes > 1 && !getenv("TNT_MI355X_LANES_ROUND_ROBIN")) {
		std::vector<double> weight(c->bucket_lmax.size(), 0.0), load(size_t(c->lanes), 0.0);
		for (const Slot &sl : c->h_slots) {
			if (sl.oligo < 0) continue;
			const size_t bk = size_t(has_exceptions ? sl.bucket_general : sl.bucket); // (the plain buckets carry nearly all windows of a database without exceptions)
			const double rows = double(c->oligos[size_t(sl.oligo)].size());
			if (size_t(sl.bucket) < weight.size()) weight[size_t(sl.bucket)] += rows * (rows + 2.0 * FLANK);
			if (has_exceptions && bk < weight.size() && bk != size_t(sl.bucket)) weight[bk] += 0.1 * rows * (rows + 2.0 * FLANK);
		}
		// (phase A of a search whose context has met such a batch before: the chains' own times, under the same sharing of the device)
		double measured = 0.0;
		if (shard_counts)
			for (size_t i = 0; i < c->bucket_lmax.size() && i < size_t(MAX_BUCKETS); i++)
				if (launches(i)) measured += c->chain_ms[i];
		if (measured > 0.0 && !getenv("TNT_MI355X_LANES_STATIC"))
			for (size_t i = 0; i < c->bucket_lmax.size() && i < size_t(MAX_BUCKETS); i++) weight[i] = c->chain_ms[i] * 1.0e6;
		std::vector<size_t> order;
		for (size_t i = 0; i < c->bucket_lmax.size(); i++)
			if (launches(i)) order.push_back(i);
		std::stable_sort(order.begin(), order.end(), [&](size_t x, size_t y) { return weight[x] > weight[y]; });
		const int first_lane = (c->split_filter && c->lanes >= 2) ? 1 : 0;
		for (size_t i : order) {
			int best = first_lane;
			for (int l = first_lane; l < c->lanes; l++)
				if (load[size_t(l)] < load[size_t(best)]) best = l;
			lane_of[i] = best;
			load[size_t(best)] += weight[i] + 1.0;
		}
	} else {
		int n = 0;
		const bool split_rr = c->split_filter && !c->serial_bind && c->lanes >= 2;
		for (size_t k = c->bucket_lmax.size(); k-- > 0;)
			if (launches(k)) lane_of[k] = c->serial_bind ? 0 : (split_rr ? 1 + n++ % (c->lanes - 1) : n++ % c->lanes);
	}
	// longest buckets first: they take the longest per item, the short ones fill the gaps
	for (size_t k = c->bucket_lmax.size(); k-- > 0;) {
		const size_t i = k;
		if (!launches(i)) continue;
		const bool split = c->split_filter && !c->serial_bind && c->lanes >= 2;
		const int lane = lane_of[i];
		n_launch++;
		if (c->serial_bind) n_launch = 1;
		const bool time_chain = shard_counts && !c->serial_bind && !split && i < size_t(MAX_BUCKETS);
		if (time_chain) {
			for (int e = 0; e < 2; e++)
				if (!c->chain_ev[i][e]) (void)hipEventCreate(&c->chain_ev[i][e]);
			(void)hipEventRecord(c->chain_ev[i][0], LS(lane));
		}
		// (direct: the queue holds 16-byte entries whose first word is the item index)
		const uint32_t *queue = direct ? reinterpret_cast<const uint32_t *>(c->S().qfat.p + c->h_q_base[i]) : c->S().qbucket.p;
		// (survivor lists keep the regrouped queues' geometry: bucket i at q_base[i] + 64 * rg_bin_base[i], so that a bucket whose
		// survivors go by group -- in place of its regrouped entries -- does not reach into its neighbour's list)
		uint32_t *survivors = c->S().qsurv.p ? c->S().qsurv.p + (direct ? c->h_q_base[i] + uint32_t(UF_BLOCK) * (i < c->rg_bin_base.size() ? c->rg_bin_base[i] : 0u) : 0u) : nullptr;
		b.index = queue;
		b.index_stride = direct ? 4 : 1;
		b.index_base_dev = direct ? nullptr : cnt + C_BUCKET + i * BUCKET_STRIDE + 2;
		b.n_items_dev = cnt + C_BUCKET + i * BUCKET_STRIDE;
		// cap on the device count.  Direct queues: the bucket's own region -- when a region overflows the seed kernel keeps
		// counting blocks past q_cap (and raises ERR_ITEM_OVERFLOW: the batch is repeated smaller), and the launches behind it
		// in this batch must not read into the next bucket's region or past the end of the queue buffer meanwhile
		const uint32_t queue_cap = direct ? c->h_q_cap[i] : uint32_t(c->item_cap);
		b.n_items = queue_cap;
		b.audit = nullptr;
		if (c->filter_on && !(c->bucket_lmax[i] & BUCKET_WAVE) && ((c->bucket_lmax[i] & BUCKET_PLAIN) || c->filter_general) && ((c->bucket_lmax[i] & 0xFF) <= 64 || c->filter_long) &&
		    has_filter_instance(c->bucket_lmax[i] & 0xFF, !(c->bucket_lmax[i] & BUCKET_PLAIN))) {
			// every window first meets the pre-filter: what cannot reach the minimum Tm is rejected there, the rest of the
			// bucket is compacted into qsurv (same offset) and bound by the full kernel.  (The general buckets -- windows
			// with degenerate target bases, merged short queues -- go through the instance that reads the exception lists.)
			FilterArgs2 f;
			memset(&f, 0, sizeof(f));
			f.tab = c->ftab.p;
			f.db = v;
			f.olig_q4 = b.olig_q4;
			f.olig_len = b.olig_len;
			f.slots = b.slots;
			f.thresh = b.thresh;
			f.items = b.items;
			f.index = queue;
			f.fat = direct ? c->S().qfat.p + c->h_q_base[i] : nullptr;
			f.index_base_dev = b.index_base_dev;
			f.n_items_dev = b.n_items_dev;
			f.n_items = queue_cap;
			f.survivors = survivors;
			f.n_survivors = cnt + C_BUCKET + i * BUCKET_STRIDE + 3;
			f.result = b.result;
			f.n_rejected = gcnt + C_FILTERED;
			f.audit = c->filter_audit ? c->S().audit.p : nullptr;
			f.status_exact = c->filter_status_exact ? 1 : 0;
			const int rows = c->bucket_lmax[i] & 0xFF;
			const bool degen = !(c->bucket_lmax[i] & BUCKET_PLAIN);
			// Second tier on the survivors (tnt_sbound.hip.h): plain windows of oligos of up to 40 bases in a regrouped bucket -- a
			// trace-free fill of align_dimer's own recurrence whose maximum carries the credit of the maximal-score paths; the first
			// tier hands its survivors over group by group, so that a wavefront of the second holds one oligo as well.  (Not where the
			// caller reports statuses: a window this tier rejects against the maximum dG may fail the Tm range first.)
			const bool second = regrouped && c->rg_desc_of[i] >= 0 && c->tier2 && c->S().qsurv2.p && c->S().rg_fill.p && (c->bucket_lmax[i] & BUCKET_PLAIN) && has_sbound_instance(rows) &&
			                    (rows < 17 || (c->bucket_lmax[i] & BUCKET_EXACT)) && c->h_rg_desc[size_t(c->rg_desc_of[i])].n_bins <= uint32_t(SB_MAX_BINS) && !c->filter_status_exact && !split;
			if (regrouped && c->rg_desc_of[i] >= 0) {
				const RegroupBucket &B = c->h_rg_desc[size_t(c->rg_desc_of[i])];
				UFilterArgs u;
				memset(&u, 0, sizeof(u));
				u.tab = f.tab, u.packed = v.packed, u.olig_q4 = f.olig_q4, u.olig_len = f.olig_len, u.slots = f.slots;
				u.db = v, u.items = f.items;
				u.fat = c->S().qsort.p + B.dst_base;
				u.n_items_dev = cnt + C_BUCKET + B.counter + 4;
				u.n_items = B.src_cap + uint32_t(UF_BLOCK) * B.n_bins;
				u.next_block = cnt + C_BUCKET + B.counter + 5;
				u.survivors = f.survivors, u.n_survivors = f.n_survivors, u.n_rejected = f.n_rejected;
				u.audit = f.audit, u.status_exact = f.status_exact;
				if (second) { // survivors by group, in the regrouped queue's own geometry
					u.survivors = c->S().qsurv.p + B.dst_base;
					u.slot_bin = c->slot_bin.p, u.general = B.general;
					u.bin_end = c->S().rg_cursor.p + B.bin_base, u.bin_count = c->S().rg_hist.p + B.bin_base, u.bin_fill = c->S().rg_fill.p + B.bin_base;
				}
				launch_ufilter(u, c->n_cu * 8, rows, LS(split ? 0 : lane), degen);
				if (second) {
					SBoundArgs sb;
					memset(&sb, 0, sizeof(sb));
					sb.tab = c->stab.p, sb.ftab = c->ftab.p, sb.db = v;
					sb.olig_q4 = b.olig_q4, sb.olig_len = b.olig_len, sb.slots = b.slots, sb.items = b.items;
					sb.index = u.survivors, sb.bin_end = u.bin_end, sb.bin_count = u.bin_count, sb.bin_fill = u.bin_fill, sb.n_bins = B.n_bins;
					sb.next_work = cnt + C_BUCKET + B.counter + 6;
					sb.survivors = c->S().qsurv2.p + c->h_q_base[i];
					sb.n_survivors = cnt + C_BUCKET + i * BUCKET_STRIDE + 7;
					sb.result = b.result, sb.n_rejected = gcnt + C_FILTERED, sb.audit = f.audit;
					launch_sbound(sb, c->n_cu * 3, rows, LS(lane));
					f.survivors = sb.survivors, f.n_survivors = sb.n_survivors; // what the bind kernel takes
				}
			} else {
				launch_filter(f, c->n_cu * 8, rows, LS(split ? 0 : lane), degen);
			}
			if (split) { // the filter kernels, each of which fills the device, queue up on lane 0; the bind launches follow theirs on the other lanes
				if (!c->filtered[i]) (void)hipEventCreateWithFlags(&c->filtered[i], hipEventDisableTiming);
				(void)hipEventRecord(c->filtered[i], LS(0));
				(void)hipStreamWaitEvent(LS(lane), c->filtered[i], 0);
			}
			// ... and of the general buckets of phase A (windows with IUPAC codes or inosine): the instance that resolves a column's
			// code against the query base of every row, on the first tier's dense list.  (Not in phase B: its few thousand windows are
			// one short chain of launches, and one more launch costs config 2 and 3 0.1 ms for nothing.)
			if (degen && direct && c->tier2 && c->S().qsurv2.p && has_sbound_degen_instance(rows) && !c->filter_status_exact && !split && !getenv("TNT_MI355X_NO_TIER2_GENERAL")) {
				SBoundDegenArgs sd;
				memset(&sd, 0, sizeof(sd));
				sd.tab = c->stab.p, sd.dtab = c->dstab.p, sd.ftab = c->ftab.p;
				sd.rt = reinterpret_cast<const uint32_t *>(reinterpret_cast<const char *>(c->tab.p) + offsetof(Tables, rt));
				sd.db = v;
				sd.olig_q4 = b.olig_q4, sd.olig_len = b.olig_len, sd.slots = b.slots, sd.items = b.items;
				sd.index = f.survivors, sd.index_base_dev = b.index_base_dev, sd.n_items_dev = f.n_survivors, sd.n_items = queue_cap;
				sd.survivors = c->S().qsurv2.p + (direct ? c->h_q_base[i] : 0u);
				sd.n_survivors = cnt + C_BUCKET + i * BUCKET_STRIDE + 7;
				sd.result = b.result, sd.n_rejected = gcnt + C_FILTERED, sd.audit = f.audit;
				launch_sbound_degen(sd, c->n_cu * 2, rows, LS(lane));
				f.survivors = sd.survivors, f.n_survivors = sd.n_survivors;
			}
			b.index = f.survivors;
			b.index_stride = 1;
			b.n_items_dev = f.n_survivors;
			b.audit = f.audit;
		}
		b.trace = c->trace.p + size_t(lane) * c->lane_trace;
		b.best = c->best.p + size_t(lane) * c->lane_cols;
		b.stack = c->stack.p + size_t(lane) * c->lane_cols;
		b.ties = c->ties.p + size_t(lane) * c->lane_ties;
		if (!c->opt.use_dinkelbach && ((c->bucket_lmax[i] & BUCKET_WAVE) || (c->wave_from && (c->bucket_lmax[i] & 0xFF) >= c->wave_from)))
			launch_wbind(c, b, bucket_rows(c->bucket_lmax[i]), false, LS(lane));
		else launch_bind(c, b, bind_grid(c, c->bucket_lmax[i]), c->bucket_lmax[i], LS(lane));
		if (time_chain) {
			(void)hipEventRecord(c->chain_ev[i][1], LS(lane));
			c->chain_timed[i] = true;
		}
	}
	for (int l = 0; l < c->lanes && (l < n_launch || c->split_filter); l++) {
		(void)hipEventRecord(c->lane_done[l], c->lane_stream[l]);
		(void)hipStreamWaitEvent(c->stream, c->lane_done[l], 0);
	}
	if (deferred) hipLaunchKernelGGL(bind_finish_kernel, dim3(c->n_cu * 4), dim3(FIN_THREADS), 0, c->stream, b); // what the instances parked: records and texts
}

int ensure_scratch(tnt_ctx *c, int need_lmax)
{
	if (need_lmax > MAX_LANE_OLIGO && !(need_lmax & BUCKET_WAVE)) need_lmax = BUCKET_WAVE | wb_rows_per_lane(need_lmax); // (rows of a unit-level request)
	if (c->trace.p && bucket_rows(c->scratch_lmax) >= bucket_rows(need_lmax)) return TNT_OK;
	if (bucket_rows(need_lmax) > bucket_rows(c->scratch_lmax)) c->scratch_lmax = need_lmax;
	hipDeviceProp_t prop;
	HIP_TRY(hipGetDeviceProperties(&prop, c->device), c->err);
	c->n_cu = prop.multiProcessorCount;
	c->bind_blocks = c->n_cu * 4;
	// per-workgroup slabs, sized for the most demanding instance this context launches
	std::vector<int> inst = c->bucket_lmax;
	inst.push_back(c->scratch_lmax);
	size_t n_trace = 0, n_cols = 0, n_ties = 0;
	for (int id : inst) {
		if (id & BUCKET_WAVE) { // one trace slab per wavefront of the persistent grid (rows beyond the LDS budget)
			n_trace = std::max(n_trace, size_t(bind_grid(c, id)) * size_t(wb_waves(id & 0xFF)) * wb_slab_bytes(id & 0xFF));
			continue;
		}
		const int l = id & 0xFF;
		const size_t lanes = size_t(bind_grid(c, id)) * size_t(bind_threads(id));
		const size_t lm = size_t(l), wm = lm + 2 * FLANK, nc = lm + wm + 4;
		n_trace = std::max(n_trace, lanes * lm * wm);
		n_cols = std::max(n_cols, lanes * nc);
		n_ties = std::max(n_ties, lanes * TIE_CAP);
	}
	c->lane_trace = n_trace, c->lane_cols = n_cols, c->lane_ties = n_ties;
	HIP_TRY(c->trace.alloc(n_trace * c->lanes), c->err);
	HIP_TRY(c->best.alloc(n_cols * c->lanes), c->err);
	HIP_TRY(c->stack.alloc(n_cols * c->lanes), c->err);
	HIP_TRY(c->ties.alloc(n_ties * c->lanes), c->err);
	return TNT_OK;
}

// (re)allocates buffer set k at the context's current capacities
int alloc_set(tnt_ctx *c, int k)
{
	tnt_ctx::BatchSet &W = c->set[k];
	HIP_TRY(W.items.alloc(c->item_cap), c->err);
	HIP_TRY(W.result.alloc(c->item_cap), c->err);
	HIP_TRY(W.groups.alloc(c->group_cap), c->err);
	HIP_TRY(W.active.alloc(c->group_cap), c->err);
	HIP_TRY(W.pairable.alloc(c->group_cap), c->err);
	HIP_TRY(W.qa.alloc(c->item_cap), c->err);
	HIP_TRY(W.qb.alloc(c->item_cap), c->err);
	HIP_TRY(W.qbucket.alloc(c->item_cap), c->err);
	if (c->filter_on) HIP_TRY(W.qsurv.alloc(c->item_cap + size_t(UF_BLOCK) * c->rg_total_bins), c->err); // (a regrouped bucket's survivors keep the regrouped queue's geometry)
	if (c->tier2) HIP_TRY(W.qsurv2.alloc(c->item_cap), c->err);
	if (c->seed2) HIP_TRY(W.qfat.alloc(c->item_cap), c->err);
	if (c->seed2 && c->regroup) {
		HIP_TRY(W.qsort.alloc(c->item_cap + size_t(UF_BLOCK) * c->rg_total_bins), c->err);
		HIP_TRY(W.rg_hist.alloc(std::max<size_t>(1, c->rg_total_bins)), c->err);
		HIP_TRY(W.rg_cursor.alloc(std::max<size_t>(1, c->rg_total_bins)), c->err);
		HIP_TRY(W.rg_fill.alloc(std::max<size_t>(1, c->rg_total_bins)), c->err);
	}
	if (c->filter_audit) HIP_TRY(W.audit.alloc(c->item_cap), c->err);
	if (c->dedupe) HIP_TRY(W.item_rep.alloc(c->item_cap), c->err);
	HIP_TRY(W.pool.alloc(POOL_CAP), c->err);
	if (c->defer_finish) HIP_TRY(W.pending.alloc(POOL_CAP), c->err);
	HIP_TRY(W.counters.alloc(C_COUNT), c->err);
	HIP_TRY(W.region_counts.alloc(2 * tnt_ctx::MAX_REGIONS), c->err);
	HIP_TRY(W.q_state.alloc(size_t(tnt_ctx::MAX_REGIONS) * MAX_BUCKETS * 2), c->err);
	if (!W.seeded) HIP_TRY(hipEventCreateWithFlags(&W.seeded, hipEventDisableTiming), c->err);
	if (!W.released) HIP_TRY(hipEventCreateWithFlags(&W.released, hipEventDisableTiming), c->err);
	return TNT_OK;
}

int ensure_set(tnt_ctx *c, int k) { return c->set[k].items.p ? TNT_OK : alloc_set(c, k); }

// Work buffers sized for what the context is asked to search: `db` (expected raw seeds of the whole database x 8: a batch is
// planned for every raw seed becoming an item, plan()) and / or `min_items` requests of a unit-level entry.  They grow
// (never shrink) up to ITEM_CAP_MAX / GROUP_CAP_MAX, beyond which a search runs in several batches: a context that only ever
// sees small databases -- the front end on a few records, the test-suite -- does not allocate (and clear page tables for) the
// 40 GB a million targets need.
int ensure_work_buffers(tnt_ctx *c, int need_lmax, const tnt_db *db = nullptr, size_t min_items = 0)
{
	auto pow2 = [](double x) {
		size_t p = size_t(1) << 24; // 16 M entries at least
		while (double(p) < x && p < (size_t(1) << 40)) p <<= 1;
		return p;
	};
	double want_i = double(min_items), want_g = 0.0;
	if (db) {
		const double words = double(size_t(1) << (2 * c->opt.hash_word_size));
		want_i = std::max(want_i, 8.0 * double(db->n_bases) * double(c->n_entries) / words);
		want_g = 2.0 * double(db->n_targets) * double(std::max<size_t>(1, c->h_assays.size()));
	}
	const size_t items = std::min(ITEM_CAP_MAX, pow2(want_i)), groups = std::min(GROUP_CAP_MAX, pow2(want_g));
	if (!c->set[0].items.p || items > c->item_cap || groups > c->group_cap) {
		HIP_TRY(hipDeviceSynchronize(), c->err); // nothing may still use the buffers that are replaced
		c->item_cap = std::max(c->item_cap, items);
		c->group_cap = std::max(c->group_cap, groups);
		c->h_q_base.clear(); // (the queue regions are shares of the item capacity: planned again)
		const bool second = c->set[1].items.p != nullptr;
		int rc = alloc_set(c, 0);
		if (rc == TNT_OK && second) rc = alloc_set(c, 1);
		if (rc) return rc;
		if (!c->hits.p) HIP_TRY(c->hits.alloc(c->hit_cap), c->err);
	}
	return ensure_scratch(c, need_lmax);
}

BindArgs bind_args(tnt_ctx *c, const DbView &v)
{
	BindArgs a;
	memset(&a, 0, sizeof(a));
	a.tab = c->tab.p;
	a.db = v;
	a.olig_base = c->olig_base.p;
	a.olig_q4 = c->olig_q4.p;
	a.olig_len = c->olig_len.p;
	a.slots = c->slots.p;
	a.thresh = c->thresh.p;
	a.items = c->S().items.p;
	a.result = c->S().result.p;
	a.pool = c->S().pool.p;
	a.pool_cap = uint32_t(POOL_CAP);
	a.pool_count = c->S().counters.p + C_POOL;
	a.trace = c->trace.p;
	a.best = c->best.p;
	a.stack = c->stack.p;
	a.ties = c->ties.p;
	a.err = c->G() + C_ERR;
	a.groups = c->S().groups.p;
	a.active = c->S().active.p;
	a.n_active = c->S().counters.p + C_ACTIVE;
	a.debug = c->G() + C_DEBUG;
	a.ltext = c->ltext.p;
	a.ltext_cap = uint32_t(std::min<size_t>(c->ltext.n, 0xFFFFFF00u));
	a.ltext_count = c->G() + C_LTEXT;
	return a;
}

// the long alignment texts of the work just finished -> host (tnt_long_text reads them there)
int fetch_long_texts(tnt_ctx *c, uint32_t bytes)
{
	c->h_ltext.clear();
	if (!bytes || !c->ltext.p) return TNT_OK;
	bytes = uint32_t(std::min<size_t>(bytes, c->ltext.n));
	c->h_ltext.resize(bytes);
	HIP_TRY(hipMemcpy(c->h_ltext.data(), c->ltext.p, bytes, hipMemcpyDeviceToHost), c->err);
	return TNT_OK;
}

const char *err_text(uint32_t e)
{
	if (e & ERR_SEED_OVERFLOW) return "seed list of one target overflowed the LDS buffer (target too long for this assay chunk)";
	if (e & ERR_ITEM_OVERFLOW) return "bind work-item buffer overflow";
	if (e & ERR_POOL_OVERFLOW) return "bound-oligo pool overflow";
	if (e & ERR_HIT_OVERFLOW) return "hit buffer overflow";
	if (e & ERR_GROUP_OVERFLOW) return "group buffer overflow";
	if (e & ERR_LIST_OVERFLOW) return "more than 512 seeds of one assay inside one seed window";
	if (e & ERR_STACK_OVERFLOW) return "traceback branch stack overflow";
	if (e & ERR_TIE_OVERFLOW) return "more than 256 maximal DP cells in one window";
	if (e & ERR_TEXT_OVERFLOW) return "alignment texts beyond TNT_ALIGN_TEXT do not fit the long-text arena (TNT_MI355X_LTEXT_MB)";
	if (e & ERR_FILTER_UNSOUND) return "pre-filter audit: the filter would have rejected a window the bind kernel accepts";
	return "unknown device error";
}

} // namespace

extern "C" {

int tnt_abi_version(void) { return TNT_ABI_VERSION; }

void tnt_options_default(tnt_options *o) // options.h:85-170, tntblast.h:19-76
{
	memset(o, 0, sizeof(*o));
	o->min_primer_tm = 0.0f, o->max_primer_tm = 9999.0f;
	o->min_primer_dg = -9999.0f, o->max_primer_dg = 0.0f;
	o->min_probe_tm = 0.0f, o->max_probe_tm = 9999.0f;
	o->min_probe_dg = -9999.0f, o->max_probe_dg = 0.0f;
	o->salt = 50.0e-3f;
	o->primer_strand = 9.0e-7f;
	o->probe_strand = 2.5e-7f;
	o->asymmetric_strand_ratio = 1.0f;
	o->target_t = 310.15f;
	o->primer_clamp = 0;
	o->min_max_primer_clamp = -1;
	o->max_gap = o->max_mismatch = 999;
	o->max_poly_degen = 3;
	o->max_len = 2000;
	o->assay_format = TNT_ASSAY_PCR;
	o->target_strand = TNT_STRAND_BOTH;
	o->hash_word_size = 7;
	o->single_primer_pcr = 1;
}

const char *tnt_last_error(const tnt_ctx *ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }
const char *tnt_ctx_warnings(const tnt_ctx *ctx) { return ctx ? ctx->warnings.c_str() : ""; }
int tnt_long_pair_stats(const tnt_ctx *ctx, uint64_t out[4])
{
	if (!ctx || !out) return TNT_E_ARG;
	for (int i = 0; i < 4; i++) out[i] = ctx->long_pair_stats[i];
	return TNT_OK;
}

// The pre-filter of oligos of 65 to 128 bases costs a fifth of the bind it spares (5 ns against 22 ns per window), and whether it
// spares any hangs on the thresholds: against -E 40 an unrelated 120-base window already reaches the bound (0.7 % rejected: the
// search is a fifth SLOWER with the filter), against -E 60 nine windows in ten are rejected (three times faster).  So a context
// with such oligos measures it once: the filter kernel itself on 2 048 random windows that share an 8-mer with their oligo (what
// a seed hands over), thresholds and strand concentrations of the context's own slots; the long instances are used when they
// reject a quarter of them.  TNT_MI355X_FILTER_LONG=0 / 1 overrides.
static int calibrate_long_filter(tnt_ctx *c)
{
	if (const char *e = getenv("TNT_MI355X_FILTER_LONG")) {
		c->filter_long = atoi(e) != 0;
		return TNT_OK;
	}
	std::vector<const Slot *> longs;
	for (const Slot &sl : c->h_slots)
		if (sl.oligo >= 0 && c->oligos[size_t(sl.oligo)].size() > 64 && c->oligos[size_t(sl.oligo)].size() <= size_t(MAX_LANE_OLIGO) && longs.size() < 64) longs.push_back(&sl);
	if (longs.empty() || !c->filter_on) return TNT_OK;
	const size_t N = 2048, TLEN = 320;
	std::vector<uint8_t> codes(N * TLEN);
	std::vector<uint64_t> offsets(N + 1);
	std::vector<tnt_bind_req> req(N);
	uint64_t state = 0x9E3779B97F4A7C15ull; // (fixed: the same decision for the same context every time)
	auto next = [&]() {
		state = state * 6364136223846793005ull + 1442695040888963407ull;
		return uint32_t(state >> 33);
	};
	for (size_t i = 0; i <= N; i++) offsets[i] = i * TLEN;
	for (size_t i = 0; i < N; i++) {
		const Slot &sl = *longs[i % longs.size()];
		const std::string &q = c->oligos[size_t(sl.oligo)];
		uint8_t *t = codes.data() + i * TLEN;
		for (size_t k = 0; k < TLEN; k++) t[k] = uint8_t(next() & 3u);
		// an 8-mer of the oligo, reverse-complemented, where the window's seed would sit (plus-strand binding)
		const int ql = int(next() % uint32_t(q.size() - 8)), at = 150;
		for (int k = 0; k < 8; k++) {
			int b = -1;
			switch (q[size_t(ql + 7 - k)]) {
			case 'A': case 'a': b = 3; break;
			case 'C': case 'c': b = 2; break;
			case 'G': case 'g': b = 1; break;
			case 'T': case 't': b = 0; break;
			default: break;
			}
			if (b >= 0) t[at + k] = uint8_t(b);
		}
		req[i] = tnt_bind_req{int32_t(i), sl.oligo, TNT_PLUS, std::min(ql, 127), at, sl.thresh == 0 ? 0 : 1, expf(sl.log_ct)};
	}
	tnt_db *db = nullptr;
	int rc = tnt_db_create(c, codes.data(), offsets.data(), N, 0, &db);
	if (rc) return rc;
	std::vector<int32_t> score(2 * N);
	rc = tnt_filter_bound(c, db, req.data(), N, score.data());
	tnt_db_destroy(db);
	if (rc) return rc;
	size_t rejected = 0;
	for (size_t i = 0; i < N; i++) rejected += (score[2 * i] < score[2 * i + 1] && score[2 * i] < FILT_SAT) ? 1 : 0;
	c->filter_long_rate = float(double(rejected) / double(N));
	c->filter_long = c->filter_long_rate >= 0.25f;
	return TNT_OK;
}

int tnt_ctx_create(const tnt_options *opt, const tnt_assay *assays, size_t n_assays, int device, tnt_ctx **out)
{
	if (!opt || !out || (!assays && n_assays)) {
		g_create_error = "null argument";
		return TNT_E_ARG;
	}
	if (opt->hash_word_size < 3 || opt->hash_word_size > 8) {
		g_create_error = "hash word size must be 3..8 (options.cpp -W)";
		return TNT_E_ARG;
	}
	if (!(opt->salt >= 1.0e-6f && opt->salt <= 1.0f)) { // NucCruc::salt, nuc_cruc.h:852-867
		g_create_error = "[Na+] out of range";
		return TNT_E_ARG;
	}
	int ndev = 0;
	if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev) {
		g_create_error = "no usable HIP device (this library has no CPU fallback)";
		return TNT_E_DEVICE;
	}
	tnt_ctx *c = new tnt_ctx;
	c->device = device;
	c->opt = *opt;
	memset(&c->timing, 0, sizeof(c->timing));
	std::string &err = g_create_error;
#define FAIL(code, msg)   \
	do {                  \
		err = (msg);      \
		delete c;         \
		return (code);    \
	} while (0)
	if (hipSetDevice(device) != hipSuccess) FAIL(TNT_E_DEVICE, "hipSetDevice failed");
	if (hipStreamCreate(&c->stream) != hipSuccess) FAIL(TNT_E_DEVICE, "hipStreamCreate failed");
	if (const char *e = getenv("TNT_MI355X_HIT_CAP")) // initial capacity of the hit buffer (it grows on demand)
		c->hit_cap = std::max<size_t>(16, strtoull(e, nullptr, 10));
	if (const char *e = getenv("TNT_MI355X_BIND_LANES")) c->lanes = std::max(1, std::min(int(tnt_ctx::BIND_LANES), atoi(e)));
	c->split_filter = getenv("TNT_MI355X_SPLIT_FILTER") != nullptr;
	for (int i = 0; i < c->lanes; i++)
		if (hipStreamCreateWithFlags(&c->lane_stream[i], hipStreamNonBlocking) != hipSuccess ||
		    hipEventCreateWithFlags(&c->lane_done[i], hipEventDisableTiming) != hipSuccess)
			FAIL(TNT_E_DEVICE, "hipStreamCreate failed");
	if (hipEventCreateWithFlags(&c->phase_ready, hipEventDisableTiming) != hipSuccess) FAIL(TNT_E_DEVICE, "hipEventCreate failed");

	// ---- unique oligos
	std::map<std::string, int> index;
	auto intern = [&](const char *s) -> int {
		if (!s) return -1;
		auto it = index.find(s);
		if (it != index.end()) return it->second;
		const int id = int(c->oligos.size());
		index[s] = id;
		c->oligos.push_back(s);
		return id;
	};
	// ---- slots and thresholds
	std::vector<Thresh> th(4);
	auto mk = [&](float a, float b, float cc, float d, int c5, int c3) {
		Thresh t;
		memset(&t, 0, sizeof(t));
		t.min_tm = a, t.max_tm = b, t.min_dg = cc, t.max_dg = d, t.clamp5 = c5, t.clamp3 = c3;
		t.max_gap = opt->max_gap, t.max_mm = opt->max_mismatch, t.max_poly = opt->max_poly_degen;
		return t;
	};
	th[0] = mk(opt->min_primer_tm, opt->max_primer_tm, opt->min_primer_dg, opt->max_primer_dg, 0, opt->primer_clamp);
	th[1] = mk(opt->min_probe_tm, opt->max_probe_tm, opt->min_probe_dg, opt->max_probe_dg, opt->probe_clamp_5, opt->probe_clamp_3);
	th[2] = mk(opt->min_probe_tm, opt->max_probe_tm, opt->min_probe_dg, opt->max_probe_dg, opt->probe_clamp_5, 0); // padlock upstream
	th[3] = mk(opt->min_probe_tm, opt->max_probe_tm, opt->min_probe_dg, opt->max_probe_dg, 0, opt->probe_clamp_3); // padlock downstream
	c->h_thresh = th;

	const float fwd_strand = opt->asymmetric_strand_ratio * opt->primer_strand; // tntblast_local.cpp:232-234
	const float rev_strand = opt->primer_strand;

	size_t max_L = 0;
	for (size_t i = 0; i < n_assays; i++) {
		const tnt_assay &A = assays[i];
		AssayDev d;
		memset(&d, 0, sizeof(d));
		d.id = A.id, d.degen_id = A.degen_id;
		d.slot_f = d.slot_r = d.slot_p = -1;
		const bool primers = A.forward && A.reverse;
		if (!primers && !A.probe) FAIL(TNT_E_ARG, "assay without oligos");
		if (primers && (A.forward_degen <= 0 || A.reverse_degen <= 0)) FAIL(TNT_E_ARG, "degeneracy divisor must be >= 1");
		if (A.probe && A.probe_degen <= 0) FAIL(TNT_E_ARG, "degeneracy divisor must be >= 1");
		// dispatch, tntblast_local.cpp:559-626
		if (primers) {
			if (opt->assay_format == TNT_ASSAY_PCR) d.kind = 0;
			else if (opt->assay_format == TNT_ASSAY_PADLOCK || opt->assay_format == TNT_ASSAY_MIPS) d.kind = 1;
			else d.kind = -1; // the reference's switch has no case: no results
		} else {
			d.kind = 2;
		}
		d.has_probe = (primers && A.probe && d.kind == 0) ? 1 : 0;
		// Slot table is laid out [assay][role] so the kernels address it without an indirection
		c->h_slots.resize((i + 1) * 3, Slot{-1, 0.0f, 0, int(i), 0, 0});
		auto add_slot = [&](int role, const char *s, float ct, int thr) -> int {
			Slot &sl = c->h_slots[i * 3 + role];
			sl.oligo = intern(s);
			sl.log_ct = logf(ct * 1.0f); // nuc_cruc.cpp:2291 with heterodimer_inv_alpha = 1
			sl.thresh = thr;
			sl.assay = int(i);
			return int(i * 3 + role);
		};
		if (primers && d.kind >= 0) {
			// strand(c, 0) = c - 0.5f*0 (nuc_cruc.h:890-910)
			const float fs = fwd_strand / A.forward_degen - 0.5f * 0.0f, rs = rev_strand / A.reverse_degen - 0.5f * 0.0f;
			d.slot_f = add_slot(0, A.forward, fs, d.kind == 0 ? 0 : 3);
			d.slot_r = add_slot(1, A.reverse, rs, d.kind == 0 ? 0 : 2);
		}
		if (A.probe && (d.kind == 2 || d.has_probe)) {
			const float ps = opt->probe_strand / A.probe_degen;
			d.slot_p = add_slot(2, A.probe, ps, 1);
		}
		c->h_assays.push_back(d);
	}
	c->n_input_assays = n_assays;
	c->h_segment.assign(n_assays, -1);
	{
		// Oligos beyond 128 bases are seeded in segments of 128 query words, one (virtual) assay per segment.  The probe of a
		// probe-only assay: every seed is bound on its own, the segments' hits are merged at collect time.  A primer, a TaqMan
		// probe or a padlock arm (nuc_cruc.h:80: every role may have 1 024 bases): the culls and the pairing act on the lists of
		// the whole oligo, so the segments' seeds join ONE list per (target, assay) -- assembled on the host, where seeds that
		// two segments found on one diagonal are reduced to the first (bind_oligo.cpp:98-99) -- and every target takes the path
		// of the targets that are seeded in windows (tnt_long.hip.h).
		const size_t k = size_t(opt->hash_word_size);
		c->h_assay_base.resize(n_assays);
		for (size_t i = 0; i < n_assays; i++) c->h_assay_base[i] = uint32_t(i);
		for (size_t i = 0; i < n_assays; i++) {
			const AssayDev d = c->h_assays[i];
			const int slots3[3] = {d.slot_f, d.slot_r, d.slot_p};
			size_t seg_of[3] = {0, 0, 0}, n_seg = 0; // segments of the role's word list (0: the role has no oligo of more than 128 bases)
			for (int role = 0; role < 3; role++) {
				if (slots3[role] < 0) continue;
				const size_t L = c->oligos[size_t(c->h_slots[size_t(slots3[role])].oligo)].size();
				if (L <= size_t(MAX_LANE_OLIGO)) continue;
				const size_t n_words = L >= k ? L - k + 1 : 0;
				seg_of[role] = std::max<size_t>(1, (n_words + 127) / 128);
				n_seg = std::max(n_seg, seg_of[role]);
			}
			if (n_seg == 0) continue;
			c->h_segment[i] = 0;
			if (d.kind == 2) {
				if (n_seg > 1) c->long_probes = true, c->h_assays[i].cache_replay |= 2;
			} else {
				c->long_primers = true;
			}
			for (size_t sg = 1; sg < n_seg; sg++) {
				const size_t v = c->h_assays.size();
				c->h_assays.push_back(c->h_assays[i]);
				c->h_segment.push_back(int32_t(sg));
				c->h_assay_base.push_back(uint32_t(i));
				c->h_slots.resize((v + 1) * 3, Slot{-1, 0.0f, 0, int(v), 0, 0});
				int32_t *const vslot[3] = {&c->h_assays[v].slot_f, &c->h_assays[v].slot_r, &c->h_assays[v].slot_p};
				for (int role = 0; role < 3; role++) {
					*vslot[role] = -1; // (a role whose word list ends before this segment seeds nothing here)
					if (slots3[role] < 0 || sg >= seg_of[role]) continue;
					Slot sl = c->h_slots[size_t(slots3[role])];
					sl.assay = int(v);
					sl.qloc_bias = int32_t(128 * sg);
					c->h_slots[v * 3 + size_t(role)] = sl;
					*vslot[role] = int32_t(v * 3 + size_t(role));
				}
			}
		}
	}
	for (const std::string &o : c->oligos) {
		if (o.size() > size_t(TNT_MAX_OLIGO)) FAIL(TNT_E_LIMIT, "oligo longer than TNT_MAX_OLIGO (1024) bases");
		if (o.empty()) FAIL(TNT_E_ARG, "empty oligo");
		for (char ch : o)
			if (char_to_base(ch) < 0) FAIL(TNT_E_ARG, "oligos must be expanded to A/C/G/T/I (expand_degenerate_signatures)");
		max_L = std::max(max_L, o.size());
	}
	c->lmax = (opt->use_dinkelbach && max_L <= 64) ? 64 : pick_lmax(max_L); // --dinkelbach: ONE instance binds everything (64 lane rows, or the wavefront kernel beyond)
	c->unit_rows = max_L <= 64 ? 64 : (max_L <= size_t(MAX_LANE_OLIGO) ? MAX_LANE_OLIGO : int(max_L)); // unit-level instance: 64 / 128 rows, or the wavefront kernel
	{
		// buckets: one per instance in use, in ascending length; the last one is the general instance that
		// takes the longest oligo (short queues are bound there as a whole)
		auto by_len = [](int a, int b) { return bucket_rows(a) != bucket_rows(b) ? bucket_rows(a) < bucket_rows(b) : a > b; };
		std::vector<int> used;
		for (const Slot &sl : c->h_slots)
			if (sl.oligo >= 0) {
				used.push_back(pick_instance(c->oligos[sl.oligo].size()));
				used.push_back(pick_lmax(c->oligos[sl.oligo].size()));
			}
		used.push_back(c->lmax);
		std::sort(used.begin(), used.end(), by_len);
		used.erase(std::unique(used.begin(), used.end()), used.end());
		c->bucket_lmax = used;
		auto index_of = [&](int id) { return int32_t(std::lower_bound(used.begin(), used.end(), id, by_len) - used.begin()); };
		for (Slot &sl : c->h_slots)
			if (sl.oligo >= 0) {
				sl.bucket = index_of(pick_instance(c->oligos[sl.oligo].size()));
				sl.bucket_general = index_of(pick_lmax(c->oligos[sl.oligo].size()));
			}
	}

	// ---- tables
	if (!fill_tables(c->h_tab, *opt)) FAIL(TNT_E_UNSUPPORTED, "delta_g table lacks the class structure the bind kernel relies on");
	fill_filter_tables(c->h_ftab, c->h_tab, *opt);
	c->h_dstab.reset(new SBoundTablesD);
	fill_sbound_tables(c->h_stab, *c->h_dstab, c->h_ftab, c->h_tab, *opt);
	c->filter_on = (c->h_ftab.enabled[0] || c->h_ftab.enabled[1]) && !getenv("TNT_MI355X_NO_FILTER");
	c->filter_audit = c->filter_on && getenv("TNT_MI355X_FILTER_AUDIT");
	c->filter_general = c->filter_on && !getenv("TNT_MI355X_NO_FILTER_GENERAL");
	c->tier2 = c->filter_on && (c->h_stab.enabled[0] || c->h_stab.enabled[1]);
	c->serial_bind = getenv("TNT_MI355X_SERIAL_BIND") != nullptr;
	if (const char *e = getenv("TNT_MI355X_WAVE_BIND")) c->wave_from = std::max(0, atoi(e)); // A/B switch: 1 = every bucket, 33 = oligos of 33 bases and more, ...
	std::vector<uint8_t> ob(std::max<size_t>(1, c->oligos.size()) * MAX_OLIGO, 0);
	std::vector<uint32_t> orv(std::max<size_t>(1, c->oligos.size()) * (MAX_OLIGO / 4), 0);
	std::vector<int32_t> ol(std::max<size_t>(1, c->oligos.size()), 0);
	for (size_t i = 0; i < c->oligos.size(); i++) {
		const std::string &o = c->oligos[i];
		const int L = int(o.size());
		ol[i] = L;
		for (int k = 0; k < L; k++) {
			ob[i * MAX_OLIGO + k] = uint8_t(char_to_base(o[k]));
			const uint32_t b = uint32_t(char_to_base(o[L - 1 - k])); // row k+1 of the DP uses query[L-(k+1)]
			orv[i * (MAX_OLIGO / 4) + (k >> 2)] |= (b * 4u) << ((k & 3) * 8);
		}
	}

	// ---- inverted query-word table
	const int k = opt->hash_word_size;
	const size_t n_words = size_t(1) << (2 * k);
	std::vector<uint32_t> wl_off(c->h_assays.size() * 6 + 1, 0);
	std::vector<uint16_t> wlist;
	std::vector<unsigned long long> wl_seq(c->h_assays.size() * 12 + 2, 0ull);
	c->assay_entries.assign(c->h_assays.size(), {});
	c->n_entries = 0;
	for (size_t ai = 0; ai < c->h_assays.size(); ai++) {
		const AssayDev &d = c->h_assays[ai];
		const int slots3[3] = {d.slot_f, d.slot_r, d.slot_p};
		for (int role = 0; role < 3; role++)
			for (int minus = 0; minus < 2; minus++) {
				const size_t li = (ai * 3 + role) * 2 + minus;
				wl_off[li] = uint32_t(wlist.size());
				if (slots3[role] < 0) continue;
				// PCR binds both strands; probes and padlocks honour --target-strand (probe_search.cpp:93,163,
				// padlock_search.cpp:97,238)
				if (d.kind != 0 && !(opt->target_strand & (minus ? TNT_STRAND_MINUS : TNT_STRAND_PLUS))) continue;
				const std::string &o = c->oligos[c->h_slots[slots3[role]].oligo];
				std::vector<uint16_t> w = word_list(o, minus == 0, k); // plus strand: find_complement
				if (c->h_segment[ai] >= 0 && o.size() > size_t(MAX_LANE_OLIGO)) { // an oligo of more than 128 bases: this (virtual) assay seeds one segment of its word list
					const size_t w0 = std::min(w.size(), size_t(128) * size_t(c->h_segment[ai])), w1 = std::min(w.size(), w0 + 128);
					w = std::vector<uint16_t>(w.begin() + long(w0), w.begin() + long(w1));
				}
				// the strand-specific sequence, 2 bits per base (plain A/C/G/T oligos only: word index == base offset)
				bool plain = o.size() <= 64; // (the packed form holds 64 bases; longer oligos take the word-list test)
				for (size_t x = 0; x < o.size() && plain; x++) {
					const char ch = minus ? o[x] : o[o.size() - 1 - x];
					int b2 = -1;
					switch (ch) {
					case 'A': case 'a': b2 = minus ? 0 : 3; break;
					case 'C': case 'c': b2 = minus ? 1 : 2; break;
					case 'G': case 'g': b2 = minus ? 2 : 1; break;
					case 'T': case 't': b2 = minus ? 3 : 0; break;
					default: plain = false; break;
					}
					if (b2 >= 0) wl_seq[li * 2 + (x >> 5)] |= (unsigned long long)b2 << ((x & 31) * 2);
				}
				if (w.size() > 128) FAIL(TNT_E_LIMIT, "too many query words");
				if (ai >= (size_t(1) << (32 - WT_ASSAY_SHIFT))) FAIL(TNT_E_LIMIT, "too many assays");
				for (size_t wi = 0; wi < w.size(); wi++) {
					c->assay_entries[ai].push_back(std::make_pair(
						w[wi], uint32_t(wi) | (minus ? WT_MINUS : 0u) | (uint32_t(role) << 8) | (d.kind != 0 ? WT_NONPCR : 0u) |
						           (d.has_probe ? WT_HASPROBE : 0u) | (plain ? WT_PLAIN : 0u) | (uint32_t(ai) << WT_ASSAY_SHIFT)));
					wlist.push_back(w[wi]);
				}
				c->n_entries += w.size();
			}
	}
	wl_off[c->h_assays.size() * 6] = uint32_t(wlist.size());
	(void)n_words; // the per-chunk bucket tables are built by plan() once the target lengths are known

	{
		std::map<std::tuple<int32_t, uint32_t, int32_t>, uint32_t> first;
		std::map<uint32_t, uint32_t> last; // class -> its last slot so far
		c->h_slot_canon.resize(c->h_slots.size());
		c->h_slot_next.assign(c->h_slots.size(), ITEM_NO_REP);
		for (size_t si = 0; si < c->h_slots.size(); si++) {
			const Slot &sl = c->h_slots[si];
			c->h_slot_canon[si] = uint32_t(si);
			if (sl.oligo < 0) continue;
			uint32_t ct_bits;
			memcpy(&ct_bits, &sl.log_ct, sizeof(ct_bits));
			const auto it = first.emplace(std::make_tuple(sl.oligo, ct_bits, sl.thresh), uint32_t(si));
			const uint32_t canon = it.first->second;
			c->h_slot_canon[si] = canon;
			if (!it.second) {
				c->dedupe = true;
				c->h_slot_next[last[canon]] = uint32_t(si);
			}
			last[canon] = uint32_t(si);
		}
		// Marking the repeats costs the seed kernel a walk along the class per phase-A seed; it pays when a bind is the full
		// kernel.  Behind the pre-filter a repeated window costs one cheap sweep, and the marking loses (config 4, 400 k
		// targets: seed 336 -> 261 ms, bind stage 216 -> 247 ms without it): it is used when the filter is off (dangling-end
		// options, thresholds at or below 0 C, --dinkelbach) or asked for.
		if (c->filter_on && !getenv("TNT_MI355X_DEDUPE")) c->dedupe = false;
		if (getenv("TNT_MI355X_NO_DEDUPE")) c->dedupe = false; // ablation
		// The reference memoises every evaluated (oligo string, window) per target and hands the memo out again whatever
		// strand concentration or threshold set asks for it (BindCacheKey holds neither, tntblast.h:247-270; on a hit the
		// caller's thresholds are applied to the first caller's Tm, and to zeros where that caller's cascade had stopped
		// early, bind_oligo.cpp:600-612,750-803).  Which evaluation comes first depends on the whole bind / cull history of
		// the target; that is not reproduced here, so an assay set in which it could matter is refused, loudly.
		std::map<int32_t, std::string> cls; // oligo -> (strand concentration, thresholds) as bytes
		std::vector<char> shared(c->oligos.size(), 0);
		for (const Slot &sl : c->h_slots) {
			if (sl.oligo < 0) continue;
			if (c->h_assays[size_t(sl.assay)].kind == 2) continue; // hybrid() binds with a cache of its own (probe_search.cpp:96,166)
			std::string key(reinterpret_cast<const char *>(&sl.log_ct), sizeof(sl.log_ct));
			key.append(reinterpret_cast<const char *>(&th[size_t(sl.thresh)]), sizeof(Thresh));
			const auto it = cls.emplace(sl.oligo, key);
			if (!it.second && it.first->second != key) shared[size_t(sl.oligo)] = 1;
		}
		// Such a context is searched with the memo's semantics since round 4: the assays that hold a shared oligo are paired by
		// the cache replay (first evaluation of an (oligo string, window) per target and strand wins, later callers apply their
		// thresholds to it).  TNT_MI355X_UNCACHED_SHARED_OLIGOS=1 keeps the uncached computation (every use on its own).
		if (!getenv("TNT_MI355X_UNCACHED_SHARED_OLIGOS"))
			for (const Slot &sl : c->h_slots)
				if (sl.oligo >= 0 && shared[size_t(sl.oligo)] && c->h_assays[size_t(sl.assay)].kind != 2 && !(c->h_assays[size_t(sl.assay)].cache_replay & 1)) {
					c->h_assays[size_t(sl.assay)].cache_replay |= 1;
					c->n_cache_assays++;
					c->cache_replay = true;
				}
	}
	{ // Q11: thresholds under which the reference reports hits this library does not
		bool primers = false, probes = false;
		for (const Slot &sl : c->h_slots)
			if (sl.oligo >= 0) (sl.thresh == 0 ? primers : probes) = true;
		auto open_class = [](float min_tm, float max_tm, float min_dg, float max_dg) { return min_tm <= 0.0f && max_tm >= 0.0f && min_dg <= 0.0f && max_dg >= 0.0f; };
		const bool qp = primers && open_class(opt->min_primer_tm, opt->max_primer_tm, opt->min_primer_dg, opt->max_primer_dg);
		const bool qb = probes && open_class(opt->min_probe_tm, opt->max_probe_tm, opt->min_probe_dg, opt->max_probe_dg);
		if (qp || qb)
			c->warnings = std::string("the ") + (qp && qb ? "primer and probe" : (qp ? "primer" : "probe")) +
			              " thresholds accept Tm = 0 and dG = 0 (no minimum Tm above 0 C, no maximum dG below 0): for a window without any valid "
			              "alignment the reference scores the first / last match its NucCruc object kept from an earlier window (alignment::clear() "
			              "does not reset them, nuc_cruc.h:360-371,386-387; bind_oligo.cpp:600-624) and can report it as a hit with Tm = 0 -- which "
			              "hits depends on its thread schedule; this library reports such a window as not bound, so it may list fewer hits than "
			              "tntblast does under these options";
	}
	// the round-3 seed kernel serves every search but those that mark shared oligos in the seed kernel
	c->seed2 = !c->dedupe && !getenv("TNT_MI355X_SEED_V2");
	if (const char *pe = getenv("TNT_MI355X_PIPELINE")) c->pipeline_batches = std::max(1, std::min(64, atoi(pe)));
	if (const char *pe = getenv("TNT_MI355X_SEED2_WGS")) c->seed2_wgs = std::max(1, std::min(TNT_S2_OCC, atoi(pe)));
	std::vector<uint32_t> slot_pack(std::max<size_t>(1, c->h_slots.size()), 0u);
	for (size_t si = 0; si < c->h_slots.size(); si++) {
		const Slot &sl = c->h_slots[si];
		if (sl.oligo < 0) continue;
		uint32_t b0 = uint32_t(sl.bucket), b1 = uint32_t(sl.bucket_general);
		if (opt->use_dinkelbach) b0 = b1 = uint32_t(c->bucket_lmax.size() - 1); // the Dinkelbach passes only exist in the longest general instance
		slot_pack[si] = b0 | (b1 << 8) | (uint32_t(c->oligos[size_t(sl.oligo)].size()) << 16);
	}
	if (c->bucket_lmax.size() > size_t(MAX_BUCKETS)) FAIL(TNT_E_LIMIT, "too many oligo length classes");
	hipError_t e = hipSuccess;
	if ((e = c->slot_pack.upload(slot_pack)) != hipSuccess) FAIL(TNT_E_DEVICE, std::string("table upload: ") + hipGetErrorString(e));
	{ // bins of the regrouping: the distinct (oligo, threshold class) pairs of every length bucket, in order of appearance
		c->h_slot_bin.assign(2 * std::max<size_t>(1, c->h_slots.size()), 0u);
		c->rg_bins.assign(c->bucket_lmax.size(), 0u);
		std::map<std::tuple<int32_t, int32_t, int>, uint32_t> seen;
		for (size_t si = 0; si < c->h_slots.size(); si++) {
			const Slot &sl = c->h_slots[si];
			if (sl.oligo < 0) continue;
			for (int g = 0; g < 2; g++) { // the slot's plain bucket, its general bucket (one bucket is never both)
				const int32_t b = int32_t((slot_pack[si] >> (8 * g)) & 0xFFu);
				const auto it = seen.emplace(std::make_tuple(b, sl.oligo, sl.thresh == 0 ? 0 : 1), c->rg_bins[size_t(b)]);
				if (it.second) c->rg_bins[size_t(b)]++;
				c->h_slot_bin[2 * si + size_t(g)] = it.first->second;
			}
		}
		c->rg_bin_base.assign(c->bucket_lmax.size(), 0u);
		c->rg_total_bins = 0;
		for (size_t b = 0; b < c->bucket_lmax.size(); b++) c->rg_bin_base[b] = c->rg_total_bins, c->rg_total_bins += c->rg_bins[b];
		c->regroup = c->filter_on && !opt->use_dinkelbach && !getenv("TNT_MI355X_NO_REGROUP");
		// built and measured in round 5: bind stage 8.74 ms with it against 8.72 without on config 2, 532 against 543 ms on config 5 --
		// what the bind kernels lose in lane-starved tail code the extra launch per phase (which the next phase waits for) takes
		// back.  Opt-in (TNT_MI355X_DEFER=1), like the other measured-and-dropped variants.
		c->defer_finish = !opt->use_dinkelbach && getenv("TNT_MI355X_DEFER") != nullptr;
		if ((e = c->slot_bin.upload(c->h_slot_bin)) != hipSuccess || (e = c->rg_bin_base_dev.upload(c->rg_bin_base)) != hipSuccess) FAIL(TNT_E_DEVICE, std::string("table upload: ") + hipGetErrorString(e));
		c->rg_seed_counts = c->regroup && c->rg_total_bins > 0 && c->rg_total_bins <= uint32_t(S2_BINS) && !getenv("TNT_MI355X_REGROUP_COUNT_PASS");
	}
	std::vector<Tables> tv(1, c->h_tab);
	std::vector<FilterTables> fv(1, c->h_ftab);
	std::vector<SBoundTables> sv(1, c->h_stab);
	std::vector<SBoundTablesD> dv(1, *c->h_dstab);
	if ((e = c->tab.upload(tv)) != hipSuccess || (e = c->ftab.upload(fv)) != hipSuccess || (e = c->stab.upload(sv)) != hipSuccess || (e = c->dstab.upload(dv)) != hipSuccess || (e = c->olig_base.upload(ob)) != hipSuccess ||
	    (e = c->olig_q4.upload(orv)) != hipSuccess || (e = c->olig_len.upload(ol)) != hipSuccess ||
	    (e = c->slots.upload(c->h_slots)) != hipSuccess || (e = c->thresh.upload(th)) != hipSuccess ||
	    (e = c->slot_canon.upload(c->h_slot_canon)) != hipSuccess || (e = c->slot_next.upload(c->h_slot_next)) != hipSuccess ||
	    (e = c->assays.upload(c->h_assays)) != hipSuccess || (e = c->wl_off.upload(wl_off)) != hipSuccess ||
	    (e = c->wl_seq.upload(wl_seq)) != hipSuccess || (c->h_wl_seq = wl_seq, false) ||
	    (e = c->wlist.upload(wlist)) != hipSuccess)
		FAIL(TNT_E_DEVICE, std::string("table upload: ") + hipGetErrorString(e));
	if (max_L > 64) { // alignment texts of such oligos may exceed TNT_ALIGN_TEXT: the arena they go to
		size_t mb = max_L > size_t(MAX_LANE_OLIGO) ? 256 : 32;
		if (const char *e = getenv("TNT_MI355X_LTEXT_MB")) mb = std::max<size_t>(1, strtoull(e, nullptr, 10));
		if ((e = c->ltext.alloc(mb << 20)) != hipSuccess) FAIL(TNT_E_DEVICE, std::string("long-text arena: ") + hipGetErrorString(e));
	}
	if (max_L > 64) {
		const int crc = calibrate_long_filter(c);
		if (crc) {
			const std::string why = c->err.empty() ? std::string("calibration of the long pre-filter failed") : c->err;
			FAIL(crc, why);
		}
	}
#undef FAIL
	*out = c;
	return TNT_OK;
}

void tnt_ctx_destroy(tnt_ctx *ctx)
{
	if (!ctx) return;
	(void)hipSetDevice(ctx->device);
	(void)hipStreamSynchronize(ctx->stream);
	for (tnt_db *db : ctx->dbs) db->ctx = nullptr; // a database may outlive its context: it can still be destroyed, nothing else
	delete ctx;
}

int tnt_ctx_num_oligos(const tnt_ctx *ctx) { return ctx ? int(ctx->oligos.size()) : 0; }
const char *tnt_ctx_oligo(const tnt_ctx *ctx, int i)
{
	return (ctx && i >= 0 && i < int(ctx->oligos.size())) ? ctx->oligos[i].c_str() : nullptr;
}

int tnt_ctx_delta_g(const tnt_ctx *ctx, int32_t *out2401)
{
	if (!ctx || !out2401) return TNT_E_ARG;
	// read back what the kernels actually use
	Tables *t = new Tables;
	const hipError_t e = hipMemcpy(t, ctx->tab.p, sizeof(Tables), hipMemcpyDeviceToHost);
	if (e == hipSuccess) memcpy(out2401, t->dg, sizeof(t->dg));
	delete t;
	return e == hipSuccess ? TNT_OK : TNT_E_DEVICE;
}

namespace {

// stride of the windows a long target is seeded in
constexpr uint32_t UNIT_STRIDE = 4096;

hipError_t build_units(tnt_ctx *ctx, tnt_db *db)
{
	const int k = ctx->opt.hash_word_size;
	// window length for which the busiest assay expects about 128 seeds (short hash words seed densely)
	size_t max_ent = 1;
	for (const auto &ae : ctx->assay_entries) max_ent = std::max(max_ent, ae.size());
	const double dense = 128.0 * double(size_t(1) << (2 * k)) / double(max_ent);
	uint32_t stride = uint32_t(std::max(128.0, std::min(double(UNIT_STRIDE), dense)));
	// after a seed-list overflow with single-assay chunks the windows shrink.  The split into whole targets and windowed
	// ones does NOT depend on unit_scale: search_long_targets halves it and rebuilds the units after the whole targets' hits
	// were collected, so a target must not change sides then (it would be reported twice).
	if (ctx->unit_scale < 1.0) stride = std::max<uint32_t>(64u, uint32_t(double(stride) * ctx->unit_scale));
	db->h_unit_target.clear(), db->h_unit_begin.clear(), db->h_unit_end.clear();
	db->h_unit_own_lo.clear(), db->h_unit_own_hi.clear();
	db->h_long_targets.clear();
	db->max_unit_len = 0, db->max_short_len = 0;
	db->split = false;
	// Whole targets first: they go through the device pipeline (seed -> bind -> pair per work unit).  A target is searched
	// whole whenever it can be: the reference's culls and pairing act on the lists of the whole target.  It is cut into
	// windows of k-mer positions when its seeds would not fit the LDS lists of one work unit (about 256 expected for the
	// busiest assay) or when it is genome-sized (> 32 k positions); the windows are only SEEDED and BOUND on the device,
	// their lists are merged per (target, assay) and culled / paired whole on the host (search_long_targets).
	auto is_long = [&](size_t t) {
		const uint32_t len = db->h_len[t];
		const uint32_t npos = len >= uint32_t(k) ? len - uint32_t(k) + 1u : 0u;
		const double expect = double(npos) * double(max_ent) / double(size_t(1) << (2 * k));
		return !(expect <= 256.0 && npos <= 32768u) || db->all_long || ctx->long_primers;
	};
	auto skipped = [&](size_t t) { return t < db->h_skip.size() && db->h_skip[t] != 0; };
	for (size_t t = 0; t < db->n_targets; t++) {
		if (is_long(t) || skipped(t)) continue;
		const uint32_t len = db->h_len[t];
		const uint32_t npos = len >= uint32_t(k) ? len - uint32_t(k) + 1u : 0u;
		db->h_unit_target.push_back(uint32_t(t)), db->h_unit_begin.push_back(0), db->h_unit_end.push_back(npos);
		db->h_unit_own_lo.push_back(INT32_MIN), db->h_unit_own_hi.push_back(INT32_MAX);
		db->max_unit_len = std::max(db->max_unit_len, len), db->max_short_len = std::max(db->max_short_len, len);
	}
	db->n_short_units = db->h_unit_target.size();
	for (size_t t = 0; t < db->n_targets; t++) {
		if (!is_long(t) || skipped(t)) continue;
		const uint32_t len = db->h_len[t];
		if (len < uint32_t(k)) continue; // (only reachable with all_long: nothing to seed, tntblast_local.cpp:513)
		const uint32_t npos = len - uint32_t(k) + 1u;
		db->split = true;
		db->h_long_targets.push_back(uint32_t(t));
		// windows overlap by the k-mer itself only: every position is seeded once (the diagonal de-duplication looks at the
		// packed target, not at the window's list)
		for (uint64_t b = 0; b < npos; b += stride) {
			const uint32_t e = uint32_t(std::min<uint64_t>(npos, b + stride));
			db->h_unit_target.push_back(uint32_t(t)), db->h_unit_begin.push_back(uint32_t(b)), db->h_unit_end.push_back(e);
			db->h_unit_own_lo.push_back(INT32_MIN), db->h_unit_own_hi.push_back(INT32_MAX);
			db->max_unit_len = std::max(db->max_unit_len, e - uint32_t(b) + uint32_t(k));
		}
	}
	std::vector<uint4> rec(db->n_short_units + 4, make_uint4(0, 0, 0, 0)); // (slack: the kernel fetches two units ahead)
	for (size_t u = 0; u < db->n_short_units; u++) {
		const uint32_t t = db->h_unit_target[u];
		rec[u] = make_uint4(t, db->h_len[t], uint32_t(db->h_word_off[t]), uint32_t(db->h_word_off[t] >> 32));
	}
	hipError_t e;
	if ((e = db->unit_rec.upload(rec)) != hipSuccess) return e;
	if ((e = db->unit_target.upload(db->h_unit_target)) != hipSuccess || (e = db->unit_begin.upload(db->h_unit_begin)) != hipSuccess ||
	    (e = db->unit_end.upload(db->h_unit_end)) != hipSuccess || (e = db->unit_own_lo.upload(db->h_unit_own_lo)) != hipSuccess ||
	    (e = db->unit_own_hi.upload(db->h_unit_own_hi)) != hipSuccess)
		return e;
	return hipSuccess;
}

} // namespace

int tnt_db_create(tnt_ctx *ctx, const uint8_t *codes, const uint64_t *offsets, size_t n_targets, uint32_t first_seq_index,
                  tnt_db **out)
{
	if (!ctx || !out || !offsets || (!codes && n_targets && offsets[n_targets] > 0)) return TNT_E_ARG;
	(void)hipSetDevice(ctx->device);
	tnt_db *db = new tnt_db;
	db->ctx = ctx;
	db->n_targets = n_targets;
	db->first_seq_index = first_seq_index;
	db->h_word_off.resize(n_targets + 1);
	db->h_len.resize(std::max<size_t>(1, n_targets));
	db->h_exc_off.assign(n_targets + 1, 0);
	uint64_t words = 0;
	for (size_t t = 0; t < n_targets; t++) {
		const uint64_t len = offsets[t + 1] - offsets[t];
		if (offsets[t + 1] < offsets[t] || len >= (uint64_t(1) << 31)) {
			ctx->err = "bad target offsets";
			delete db;
			return TNT_E_ARG;
		}
		db->h_word_off[t] = words;
		db->h_len[t] = uint32_t(len);
		db->max_len = std::max(db->max_len, uint32_t(len));
		db->n_bases += len;
		words += ((len + 15) / 16 + 3) / 4 * 4; // 16-byte aligned rows: a wavefront reads whole dwordx4 lines
	}
	db->h_word_off[n_targets] = words;
	db->h_packed.assign(words + PACKED_SLACK, 0u); // slack: the window loaders read up to ten words from a window's first word (the 128-row pre-filter)
	// exceptions: count first (codes > 3 keep their low two bits in the stream)
	std::vector<uint32_t> exc_count(n_targets + 1, 0);
	parallel_for(n_targets, [&](size_t t) {
		const uint8_t *s = codes + offsets[t];
		const uint32_t len = db->h_len[t];
		uint32_t *pk = db->h_packed.data() + db->h_word_off[t];
		uint32_t ne = 0;
		for (uint32_t i = 0; i < len; i++) {
			const uint8_t cde = s[i];
			pk[i >> 4] |= uint32_t(cde & 3u) << ((i & 15u) * 2u);
			ne += cde > 3;
		}
		exc_count[t] = ne;
	});
	for (size_t t = 0; t < n_targets; t++) db->h_exc_off[t + 1] = db->h_exc_off[t] + exc_count[t];
	db->n_exceptions = db->h_exc_off[n_targets];
	db->host_copy = true;
	memset(&db->ingest, 0, sizeof(db->ingest));
	db->h_exc_pos.resize(std::max<size_t>(1, db->h_exc_off[n_targets]));
	db->h_exc_code.resize(std::max<size_t>(1, db->h_exc_off[n_targets]));
	parallel_for(n_targets, [&](size_t t) {
		if (!exc_count[t]) return;
		const uint8_t *s = codes + offsets[t];
		uint32_t o = db->h_exc_off[t];
		for (uint32_t i = 0; i < db->h_len[t]; i++)
			if (s[i] > 3) {
				db->h_exc_pos[o] = i;
				db->h_exc_code[o] = s[i] > 17 ? uint8_t(17) : s[i];
				o++;
			}
	});
	hipError_t e;
	if ((e = db->packed.upload(db->h_packed)) != hipSuccess || (e = db->word_off.upload(db->h_word_off)) != hipSuccess ||
	    (e = db->len.upload(db->h_len)) != hipSuccess || (e = db->exc_off.upload(db->h_exc_off)) != hipSuccess ||
	    (e = db->exc_pos.upload(db->h_exc_pos)) != hipSuccess || (e = db->exc_code.upload(db->h_exc_code)) != hipSuccess ||
	    (e = build_units(ctx, db)) != hipSuccess) {
		ctx->err = std::string("target upload: ") + hipGetErrorString(e);
		delete db;
		return TNT_E_DEVICE;
	}
	ctx->dbs.push_back(db);
	*out = db;
	return TNT_OK;
}

void tnt_db_destroy(tnt_db *db)
{
	if (!db) return;
	if (tnt_ctx *c = db->ctx) {
		(void)hipSetDevice(c->device);
		(void)hipStreamSynchronize(c->stream);
		c->dbs.erase(std::remove(c->dbs.begin(), c->dbs.end(), db), c->dbs.end());
		if (c->pending_db == db) c->pending_db = nullptr;
	}
	delete db;
}

int tnt_db_skip_targets(tnt_ctx *ctx, tnt_db *db, const uint8_t *skip, size_t n)
{
	if (!ctx || !db || db->ctx != ctx || (!skip && n) || n > db->n_targets) return TNT_E_ARG;
	HIP_TRY(hipSetDevice(ctx->device), ctx->err);
	HIP_TRY(hipStreamSynchronize(ctx->stream), ctx->err);
	db->h_skip.assign(db->n_targets, 0);
	for (size_t t = 0; t < n; t++) db->h_skip[t] = skip[t] ? 1 : 0;
	HIP_TRY(build_units(ctx, db), ctx->err);
	return TNT_OK;
}

size_t tnt_db_num_targets(const tnt_db *db) { return db ? db->n_targets : 0; }
uint64_t tnt_db_num_bases(const tnt_db *db) { return db ? db->n_bases : 0; }

// ---------------------------------------------------------------------------------------------------------
// Target ingest: FASTA text -> packed rows, parsed on the device (tnt_ingest.hip.h)
namespace {

double now_ms()
{
	return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// ascii_to_hash_base (seq.h:148-189) behind the skip rule of read_bio_seq_fasta_slow (sequence_data_fastx.cpp:366)
void ingest_lut(uint8_t lut[256])
{
	for (int c = 0; c < 256; c++) {
		uint8_t v = TNT_DB_UNKNOWN;
		switch (toupper(c)) {
		case 'A': v = TNT_DB_A; break;
		case 'C': v = TNT_DB_C; break;
		case 'G': v = TNT_DB_G; break;
		case 'T':
		case 'U': v = TNT_DB_T; break;
		case 'I': v = TNT_DB_I; break;
		case 'M': v = TNT_DB_M; break;
		case 'R': v = TNT_DB_R; break;
		case 'S': v = TNT_DB_S; break;
		case 'V': v = TNT_DB_V; break;
		case 'W': v = TNT_DB_W; break;
		case 'Y': v = TNT_DB_Y; break;
		case 'H': v = TNT_DB_H; break;
		case 'K': v = TNT_DB_K; break;
		case 'D': v = TNT_DB_D; break;
		case 'B': v = TNT_DB_B; break;
		case 'N': v = TNT_DB_N; break;
		default: break;
		}
		// isspace() in the C locale, '*' and '-' never reach ascii_to_hash_base
		if (c == ' ' || (c >= '\t' && c <= '\r') || c == '*' || c == '-') v = INGEST_SKIP;
		lut[c] = v;
	}
}

// file_type (annotation.cpp:296-374) as far as it tells FASTQ from FASTA: the first non-space character of the first
// 511 bytes is an '@'
bool text_is_fastq(const char *text, size_t n)
{
	for (size_t i = 0; i < n && i < 511; i++)
		if (!isspace(static_cast<unsigned char>(text[i]))) return text[i] == '@';
	return false;
}

// load_fastq's record index (sequence_data_fastx.cpp:103-158): an '@' that opens a line (blanks and tabs before it do
// not count) starts a record unless the line before it was the '+' line -- then it is the first quality value.  The scan
// has to start at a record boundary (the top of the file, or where an earlier batch was cut).
void fastq_record_starts(const char *text, size_t n, std::vector<uint64_t> &rec)
{
	char last_header = '-';
	bool read_eol = true;
	for (size_t i = 0; i < n; i++) {
		const char c = text[i];
		if (c == '\n' || c == '\r') {
			read_eol = true;
		} else if (c == ' ' || c == '\t') {
			continue;
		} else if (read_eol) {
			if (c == '@') {
				if (last_header != '+') rec.push_back(i);
				last_header = '@';
			} else if (c == '+') {
				last_header = (last_header == '+') ? '-' : '+';
			} else {
				last_header = '-';
			}
			read_eol = false;
			// nothing else on this line matters
			while (i + 1 < n && text[i + 1] != '\n' && text[i + 1] != '\r') i++;
		}
	}
}

} // namespace

int tnt_db_create_from_fasta(tnt_ctx *ctx, const char *text, size_t n_bytes, uint32_t first_seq_index, tnt_db **out)
{
	if (!ctx || !out || (!text && n_bytes)) return TNT_E_ARG;
	HIP_TRY(hipSetDevice(ctx->device), ctx->err);
	std::unique_ptr<tnt_db> owner(new tnt_db); // freed on every early return below
	tnt_db *db = owner.get();
	db->ctx = ctx;
	db->first_seq_index = first_seq_index;
	memset(&db->ingest, 0, sizeof(db->ingest));
	db->ingest.text_bytes = n_bytes;
	double t0 = now_ms();

	// ---- record boundaries: a record starts at the first '>' of a line (load_fasta, sequence_data_fastx.cpp:37-52)
	// The text is cut into slices scanned by host threads: every '>' is a candidate; it opens a record unless
	// another '>' precedes it on its line (looked up backwards, across the slice boundary if need be).
	std::vector<uint64_t> rec; // offsets of the '>' + end of text
	const bool fastq = text_is_fastq(text, n_bytes);
	if (fastq) {
		fastq_record_starts(text, n_bytes, rec);
	} else {
		const size_t slice = size_t(16) << 20, n_slices = (n_bytes + slice - 1) / slice;
		std::vector<std::vector<uint64_t>> found(n_slices);
		parallel_for_each(n_slices, [&](size_t si) {
			const size_t b = si * slice, e = std::min(n_bytes, b + slice);
			for (size_t at = b; at < e;) {
				const char *gt = static_cast<const char *>(memchr(text + at, '>', e - at));
				if (!gt) break;
				const size_t g = size_t(gt - text);
				const char *nl = g ? static_cast<const char *>(memrchr(text, '\n', g)) : nullptr;
				const size_t line = nl ? size_t(nl - text) + 1 : 0;
				if (!memchr(text + line, '>', g - line)) found[si].push_back(g);
				at = g + 1;
			}
		});
		for (const auto &f : found) rec.insert(rec.end(), f.begin(), f.end());
	}
	const size_t n = rec.size();
	rec.push_back(n_bytes);
	std::vector<uint64_t> seq_begin(std::max<size_t>(1, n)), seq_end(std::max<size_t>(1, n)), def_begin(std::max<size_t>(1, n));
	db->defline_off.assign(n + 1, 0);
	std::vector<uint8_t> truncated(std::max<size_t>(1, n), 0);
	std::vector<uint64_t> def_end(std::max<size_t>(1, n));
	parallel_for(n, [&](size_t r) { // defline: read_bio_seq_fasta_slow / _fastq_slow, sequence_data_fastx.cpp:248-276,448-479
		size_t p = rec[r] + 1;
		const size_t last = rec[r + 1];
		while (p < last && isspace(static_cast<unsigned char>(text[p]))) p++;
		size_t q = p;
		while (q < last && text[q] != '\n' && text[q] != '\r') q++;
		truncated[r] = (p == last || q == last) ? 1 : 0;
		def_begin[r] = p;
		def_end[r] = q;
		seq_begin[r] = q;
		seq_end[r] = last;
		if (fastq && !truncated[r]) {
			// ONE end-of-line character is skipped, then the sequence is the rest of that line (:476-479,568-571):
			// after "\r\n" the line is empty
			size_t b = q + 1;
			if (b == last) truncated[r] = 1;
			size_t e = b;
			while (e < last && text[e] != '\n' && text[e] != '\r') e++;
			seq_begin[r] = b, seq_end[r] = e;
		}
	});
	for (size_t r = 0; r < n; r++) {
		if (truncated[r]) {
			ctx->err = fastq ? "Truncated fastq file detected!" : "Truncated fasta file detected!";
			return TNT_E_ARG;
		}
		db->defline_off[r + 1] = db->defline_off[r] + (def_end[r] - def_begin[r]);
	}
	db->record_bytes.resize(n);
	for (size_t r = 0; r < n; r++) db->record_bytes[r] = rec[r + 1] - rec[r];
	db->deflines.resize(db->defline_off[n]);
	parallel_for(n, [&](size_t r) { memcpy(&db->deflines[db->defline_off[r]], text + def_begin[r], def_end[r] - def_begin[r]); });
	db->n_targets = n;
	db->ingest.index_ms = float(now_ms() - t0);

	// ---- text to the device
	t0 = now_ms();
	DevBuf<uint8_t> d_text, d_lut;
	DevBuf<uint64_t> d_begin, d_end;
	DevBuf<uint32_t> d_nexc;
	uint8_t lut[256];
	ingest_lut(lut);
	hipError_t e;
	if ((e = d_text.alloc(n_bytes + 8)) != hipSuccess || (e = d_lut.alloc(256)) != hipSuccess || (e = d_begin.upload(seq_begin)) != hipSuccess ||
	    (e = d_end.upload(seq_end)) != hipSuccess || (e = db->len.alloc(std::max<size_t>(1, n))) != hipSuccess ||
	    (e = d_nexc.alloc(std::max<size_t>(1, n))) != hipSuccess ||
	    (n_bytes && (e = hipMemcpy(d_text.p, text, n_bytes, hipMemcpyHostToDevice)) != hipSuccess) ||
	    (e = hipMemcpy(d_lut.p, lut, 256, hipMemcpyHostToDevice)) != hipSuccess) {
		ctx->err = std::string("fasta upload: ") + hipGetErrorString(e);
		return TNT_E_DEVICE;
	}
	db->ingest.h2d_ms = float(now_ms() - t0);

	IngestArgs a;
	memset(&a, 0, sizeof(a));
	a.text = d_text.p;
	a.seq_begin = d_begin.p;
	a.seq_end = d_end.p;
	a.n_records = uint32_t(n);
	a.lut = d_lut.p;
	a.len = db->len.p;
	a.n_exc = d_nexc.p;
	const int blocks = int(std::min<size_t>(std::max<size_t>(1, (n + INGEST_WAVES - 1) / INGEST_WAVES), size_t(1) << 16));
	struct Events { // destroyed on every return
		hipEvent_t e[4] = {nullptr, nullptr, nullptr, nullptr};
		~Events()
		{
			for (hipEvent_t x : e)
				if (x) (void)hipEventDestroy(x);
		}
	} events;
	hipEvent_t *ev = events.e;
	for (int x = 0; x < 4; x++) HIP_TRY(hipEventCreate(&ev[x]), ctx->err);
	HIP_TRY(hipEventRecord(ev[0], ctx->stream), ctx->err);
	if (n) hipLaunchKernelGGL(ingest_count_kernel, dim3(blocks), dim3(INGEST_THREADS), 0, ctx->stream, a);
	HIP_TRY(hipEventRecord(ev[1], ctx->stream), ctx->err);

	// ---- layout of the packed rows (same as tnt_db_create)
	db->h_len.assign(std::max<size_t>(1, n), 0);
	std::vector<uint32_t> nexc(std::max<size_t>(1, n), 0);
	HIP_TRY(hipMemcpyAsync(db->h_len.data(), db->len.p, n * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream), ctx->err);
	HIP_TRY(hipMemcpyAsync(nexc.data(), d_nexc.p, n * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream), ctx->err);
	HIP_TRY(hipStreamSynchronize(ctx->stream), ctx->err);
	t0 = now_ms();
	db->h_word_off.resize(n + 1);
	db->h_exc_off.assign(n + 1, 0);
	uint64_t words = 0;
	for (size_t t = 0; t < n; t++) {
		const uint64_t len = db->h_len[t];
		db->h_word_off[t] = words;
		db->max_len = std::max(db->max_len, uint32_t(len));
		db->n_bases += len;
		words += ((len + 15) / 16 + 3) / 4 * 4;
		db->h_exc_off[t + 1] = db->h_exc_off[t] + nexc[t];
	}
	db->h_word_off[n] = words;
	db->n_exceptions = db->h_exc_off[n];
	if ((e = db->packed.alloc(words + PACKED_SLACK)) != hipSuccess || (e = db->word_off.upload(db->h_word_off)) != hipSuccess ||
	    (e = db->exc_off.upload(db->h_exc_off)) != hipSuccess || (e = db->exc_pos.alloc(std::max<uint64_t>(1, db->n_exceptions))) != hipSuccess ||
	    (e = db->exc_code.alloc(std::max<uint64_t>(1, db->n_exceptions))) != hipSuccess ||
	    (e = hipMemsetAsync(db->packed.p, 0, (words + PACKED_SLACK) * sizeof(uint32_t), ctx->stream)) != hipSuccess) {
		ctx->err = std::string("target allocation: ") + hipGetErrorString(e);
		return TNT_E_DEVICE;
	}
	db->ingest.layout_ms = float(now_ms() - t0);
	a.word_off = db->word_off.p;
	a.exc_off = db->exc_off.p;
	a.packed = db->packed.p;
	a.exc_pos = db->exc_pos.p;
	a.exc_code = db->exc_code.p;
	HIP_TRY(hipEventRecord(ev[2], ctx->stream), ctx->err);
	if (n) hipLaunchKernelGGL(ingest_pack_kernel, dim3(blocks), dim3(INGEST_THREADS), 0, ctx->stream, a);
	HIP_TRY(hipEventRecord(ev[3], ctx->stream), ctx->err);
	HIP_TRY(hipStreamSynchronize(ctx->stream), ctx->err);
	HIP_TRY(hipGetLastError(), ctx->err);
	(void)hipEventElapsedTime(&db->ingest.count_ms, ev[0], ev[1]);
	(void)hipEventElapsedTime(&db->ingest.pack_ms, ev[2], ev[3]);
	HIP_TRY(build_units(ctx, db), ctx->err);
	db->ingest.bases = db->n_bases;
	db->ingest.exceptions = db->n_exceptions;
	ctx->dbs.push_back(db);
	*out = owner.release();
	return TNT_OK;
}

int tnt_db_load_fasta(tnt_ctx *ctx, const char *path, uint32_t first_seq_index, tnt_db **out)
{
	if (!ctx || !path || !out) return TNT_E_ARG;
	const double t0 = now_ms();
	gzFile in = gzopen(path, "r"); // reads plain files as they are (sequence_data_fastx.cpp:16)
	if (!in) {
		ctx->err = "Unable to open fasta sequence file for reading";
		return TNT_E_ARG;
	}
	(void)gzbuffer(in, 1u << 20);
	std::vector<char> text;
	size_t used = 0;
	while (true) {
		if (text.size() - used < (size_t(1) << 24)) text.resize(std::max<size_t>(size_t(1) << 26, text.size() * 2));
		const int got = gzread(in, text.data() + used, unsigned(std::min<size_t>(text.size() - used, size_t(1) << 30)));
		if (got < 0) {
			gzclose(in);
			ctx->err = "Error while reading fasta file";
			return TNT_E_ARG;
		}
		if (got == 0) break;
		used += size_t(got);
	}
	gzclose(in);
	const float read_ms = float(now_ms() - t0);
	const int rc = tnt_db_create_from_fasta(ctx, text.data(), used, first_seq_index, out);
	if (rc == TNT_OK) (*out)->ingest.read_ms = read_ms;
	return rc;
}

// ---- streaming reader: whole records in batches, the next batch is read while the caller works on the current one
struct tnt_fasta_reader {
	gzFile in = nullptr;
	std::string err;
	std::vector<char> carry;   // bytes read beyond the last batch (they start at a record boundary, or before the first record)
	bool eof = false;
	int fastq = -1;            // unknown until the first 511 bytes are seen
	uint32_t next_index = 0;
	// prefetched batch
	std::thread worker;
	bool have_next = false;
	std::vector<char> next_text;
	size_t next_max = 0;
	bool next_failed = false;
};

namespace {

// offset of the last record start (first '>' of its line) at a position > 0, or 0 if there is none
size_t last_record_start(const char *text, size_t n)
{
	size_t end = n;
	while (end > 1) {
		const char *gt = static_cast<const char *>(memrchr(text + 1, '>', end - 1));
		if (!gt) return 0;
		const size_t g = size_t(gt - text);
		const char *nl = static_cast<const char *>(memrchr(text, '\n', g));
		const size_t line = nl ? size_t(nl - text) + 1 : 0;
		if (!memchr(text + line, '>', g - line)) return g;
		end = g;
	}
	return 0;
}

// offset of the first record start at a position >= from (> 0), or n if there is none
size_t next_record_start(const char *text, size_t n, size_t from)
{
	for (size_t at = std::max<size_t>(from, 1); at < n;) {
		const char *gt = static_cast<const char *>(memchr(text + at, '>', n - at));
		if (!gt) return n;
		const size_t g = size_t(gt - text);
		const char *nl = static_cast<const char *>(memrchr(text, '\n', g));
		const size_t line = nl ? size_t(nl - text) + 1 : 0;
		if (!memchr(text + line, '>', g - line)) return g;
		at = g + 1;
	}
	return n;
}

// fills `out` with the carried bytes plus whole records up to about max_bytes; the rest goes back to carry
void read_batch(tnt_fasta_reader *r, size_t max_bytes, std::vector<char> &out)
{
	out.swap(r->carry);
	r->carry.clear();
	while (true) {
		// enough text for a cut at a record boundary near max_bytes?
		if (r->fastq < 0 && (out.size() >= 511 || r->eof)) r->fastq = text_is_fastq(out.data(), out.size()) ? 1 : 0;
		if (out.size() > max_bytes) {
			size_t cut = 0;
			if (r->fastq == 1) { // record starts depend on the lines before them: scan from the top of the batch
				std::vector<uint64_t> starts;
				fastq_record_starts(out.data(), out.size(), starts);
				size_t next = out.size();
				for (const uint64_t st : starts) {
					if (st == 0) continue;
					if (st <= max_bytes) cut = size_t(st);
					else {
						next = size_t(st);
						break;
					}
				}
				if (!cut) cut = next;
			} else {
				cut = last_record_start(out.data(), max_bytes);
				if (!cut) cut = next_record_start(out.data(), out.size(), max_bytes); // one record longer than the batch
			}
			if (cut < out.size()) {
				r->carry.assign(out.begin() + cut, out.end());
				out.resize(cut);
				return;
			}
		}
		if (r->eof) return; // the tail of the file
		const size_t want = std::max<size_t>(size_t(1) << 22, max_bytes > out.size() ? max_bytes - out.size() + (size_t(1) << 16) : size_t(1) << 22);
		const size_t at = out.size();
		out.resize(at + want);
		const int got = gzread(r->in, out.data() + at, unsigned(std::min<size_t>(want, size_t(1) << 30)));
		if (got < 0) {
			r->next_failed = true;
			out.resize(at);
			return;
		}
		out.resize(at + size_t(got));
		if (got == 0) r->eof = true;
	}
}

void start_prefetch(tnt_fasta_reader *r, size_t max_bytes)
{
	r->next_max = max_bytes;
	r->have_next = true;
	r->worker = std::thread([r, max_bytes]() { read_batch(r, max_bytes, r->next_text); });
}

} // namespace

int tnt_fasta_open(const char *path, tnt_fasta_reader **out)
{
	if (!path || !out) return TNT_E_ARG;
	gzFile in = gzopen(path, "r");
	if (!in) return TNT_E_ARG;
	(void)gzbuffer(in, 1u << 20);
	tnt_fasta_reader *r = new tnt_fasta_reader;
	r->in = in;
	*out = r;
	return TNT_OK;
}

int tnt_fasta_next(tnt_fasta_reader *r, tnt_ctx *ctx, size_t max_bytes, tnt_db **out)
{
	if (!r || !ctx || !out) return TNT_E_ARG;
	*out = nullptr;
	max_bytes = std::max<size_t>(max_bytes, 1u << 16);
	std::vector<char> text;
	if (r->have_next) {
		r->worker.join();
		r->have_next = false;
		text.swap(r->next_text);
	} else {
		read_batch(r, max_bytes, text);
	}
	if (r->next_failed) {
		ctx->err = "Error while reading fasta file";
		return TNT_E_ARG;
	}
	if (text.empty() && r->eof) return TNT_OK;
	if (!r->eof || !r->carry.empty()) start_prefetch(r, max_bytes); // overlaps with the caller's search of this batch
	const int rc = tnt_db_create_from_fasta(ctx, text.data(), text.size(), r->next_index, out);
	if (rc != TNT_OK) return rc;
	r->next_index += uint32_t((*out)->n_targets);
	if ((*out)->n_targets == 0 && r->have_next) { // text before the first record only: move on
		tnt_db_destroy(*out);
		*out = nullptr;
		return tnt_fasta_next(r, ctx, max_bytes, out);
	}
	return TNT_OK;
}

void tnt_fasta_close(tnt_fasta_reader *r)
{
	if (!r) return;
	if (r->have_next) r->worker.join();
	if (r->in) gzclose(r->in);
	delete r;
}

int tnt_db_defline(const tnt_db *db, size_t i, const char **defline, size_t *len)
{
	if (!db || !defline || !len || i >= db->n_targets) return TNT_E_ARG;
	if (db->defline_off.empty()) {
		*defline = "";
		*len = 0;
		return TNT_OK;
	}
	*defline = db->deflines.data() + db->defline_off[i];
	*len = size_t(db->defline_off[i + 1] - db->defline_off[i]);
	return TNT_OK;
}

int tnt_db_target_info(const tnt_db *db, size_t i, uint32_t *n_bases, uint64_t *record_bytes)
{
	if (!db || i >= db->n_targets) return TNT_E_ARG;
	if (n_bases) *n_bases = db->h_len[i];
	if (record_bytes) *record_bytes = db->record_bytes.empty() ? uint64_t(db->h_len[i]) : db->record_bytes[i];
	return TNT_OK;
}

int64_t tnt_db_target_codes(tnt_ctx *ctx, tnt_db *db, size_t i, uint8_t *codes, size_t cap)
{
	if (!ctx || !db || db->ctx != ctx || i >= db->n_targets || (!codes && cap)) return TNT_E_ARG;
	const uint32_t len = db->h_len[i];
	if (len > cap) return TNT_E_LIMIT;
	if (!len) return 0;
	HIP_TRY(hipSetDevice(ctx->device), ctx->err);
	DevBuf<uint8_t> d;
	HIP_TRY(d.alloc(len), ctx->err);
	UnpackArgs u;
	u.db = make_view(db, 0, db->n_targets);
	u.target = uint32_t(i);
	u.p0 = 0;
	u.n = len;
	u.out = d.p;
	hipLaunchKernelGGL(unpack_kernel, dim3((len + 255) / 256), dim3(256), 0, ctx->stream, u);
	HIP_TRY(hipMemcpyAsync(codes, d.p, len, hipMemcpyDeviceToHost, ctx->stream), ctx->err);
	HIP_TRY(hipStreamSynchronize(ctx->stream), ctx->err);
	return int64_t(len);
}

int tnt_db_ingest_timing(const tnt_db *db, tnt_ingest_timing *out)
{
	if (!db || !out) return TNT_E_ARG;
	*out = db->ingest;
	return TNT_OK;
}

// Row a6 behind the ordinary stages of a batch: the groups of the assays that share an oligo across strand concentrations /
// threshold sets are collected, every one of their items is bound in full by the unit-level kernel instance, the groups are
// sorted by (target, assay) and replayed target by target with the reference's BindCache semantics (tnt_pair.hip.h).
static int ensure_cache_buffers(tnt_ctx *c, size_t n_units, size_t &cap_groups, size_t &cap_items, size_t &tmp_bytes)
{
	cap_groups = std::max<size_t>(1, n_units * size_t(c->n_cache_assays)); // every (target, flagged assay) at most once
	cap_items = std::min<size_t>(size_t(1) << 24, std::max<size_t>(size_t(1) << 18, 16 * cap_groups));
	if (c->ck_in.n < cap_groups) {
		HIP_TRY(c->ck_in.alloc(cap_groups), c->err);
		HIP_TRY(c->ck_out.alloc(cap_groups), c->err);
		HIP_TRY(c->cv_in.alloc(cap_groups), c->err);
		HIP_TRY(c->cv_out.alloc(cap_groups), c->err);
		HIP_TRY(c->c_recs.alloc(cap_groups), c->err);
	}
	if (c->c_items.n < cap_items) {
		HIP_TRY(c->c_items.alloc(cap_items), c->err);
		HIP_TRY(c->c_origin.alloc(cap_items), c->err);
		HIP_TRY(c->c_full.alloc(cap_items), c->err);
		HIP_TRY(c->c_pool.alloc(cap_items), c->err);
	}
	if (!c->c_counts.p) HIP_TRY(c->c_counts.alloc(4), c->err);
	tmp_bytes = 0;
	HIP_TRY(sort_pairs_u64_u32(nullptr, tmp_bytes, c->ck_in.p, c->ck_out.p, c->cv_in.p, c->cv_out.p, cap_groups, c->stream), c->err);
	if (c->c_sort_tmp.n < tmp_bytes) HIP_TRY(c->c_sort_tmp.alloc(tmp_bytes + tmp_bytes / 4 + 256), c->err);
	return TNT_OK;
}

static int enqueue_cache_replay(tnt_ctx *c, PairArgs p, bool direct, size_t cap_groups, size_t cap_items, size_t tmp_bytes, uint32_t *cnt)
{
	HIP_TRY(hipMemsetAsync(c->ck_in.p, 0xFF, cap_groups * sizeof(unsigned long long), c->stream), c->err);
	HIP_TRY(hipMemsetAsync(c->c_counts.p, 0, 4 * sizeof(uint32_t), c->stream), c->err);
	CacheCollectArgs k;
	memset(&k, 0, sizeof(k));
	k.groups = p.groups;
	if (direct) k.seg_counts = c->S().region_counts.p + 1, k.seg_count_stride = 2, k.n_segs = c->batch_regions, k.seg_size = uint32_t(c->group_cap / c->batch_regions);
	else k.seg_counts = cnt + C_GROUPS, k.seg_count_stride = SHARD_STRIDE, k.n_segs = N_SHARD, k.seg_size = uint32_t(c->group_cap / N_SHARD);
	k.assays = p.assays;
	k.items = p.items;
	k.keys = c->ck_in.p, k.vals = c->cv_in.p, k.recs = c->c_recs.p;
	k.cap_groups = uint32_t(cap_groups);
	k.n_groups = c->c_counts.p;
	k.citems = c->c_items.p, k.corigin = c->c_origin.p;
	k.cap_items = uint32_t(cap_items);
	k.n_items = c->c_counts.p + 1;
	k.n_big = c->c_counts.p + 3;
	k.err = p.err;
	hipLaunchKernelGGL(cache_collect_kernel, dim3(std::min<uint32_t>(k.n_segs, 4096u)), dim3(256), 0, c->stream, k);
	// every collected item in full: status, Tm, dH, dS, dG, anchors, counts, footprint, text -- whatever its own thresholds say
	BindArgs b = bind_args(c, p.db);
	b.items = c->c_items.p;
	b.n_items = uint32_t(cap_items);
	b.n_items_dev = c->c_counts.p + 1;
	b.index = nullptr;
	b.groups = nullptr;
	b.full = c->c_full.p;
	launch_bind(c, b, c->bind_blocks, 0);
	HIP_TRY(sort_pairs_u64_u32(c->c_sort_tmp.p, tmp_bytes, c->ck_in.p, c->ck_out.p, c->cv_in.p, c->cv_out.p, cap_groups, c->stream), c->err);
	CacheReplayArgs r;
	r.p = p;
	r.p.slots = c->slots.p, r.p.thresh = c->thresh.p;
	r.p.citems = c->c_items.p, r.p.cfull = c->c_full.p, r.p.corigin = c->c_origin.p;
	r.p.cpool = c->c_pool.p, r.p.cpool_cap = uint32_t(cap_items), r.p.cpool_count = c->c_counts.p + 2;
	r.keys = c->ck_out.p, r.vals = c->cv_out.p, r.recs = c->c_recs.p;
	r.cap_groups = uint32_t(cap_groups);
	r.n_big = c->c_counts.p + 3;
	hipLaunchKernelGGL(cache_replay_kernel<PAIR_MAX>, dim3(uint32_t(std::min<size_t>(8192, (cap_groups + PAIR_THREADS - 1) / PAIR_THREADS))), dim3(PAIR_THREADS), 0, c->stream, r);
	hipLaunchKernelGGL(cache_replay_kernel<PAIR_BIG>, dim3(uint32_t(std::min<size_t>(256, (cap_groups + PAIR_THREADS - 1) / PAIR_THREADS))), dim3(PAIR_THREADS), 0, c->stream, r);
	return TNT_OK;
}

// One internal batch: seed -> bind -> filter -> pair.  `pipelined`: the batch works on buffer set c->cur; its seed stage runs
// on the seed stream (behind the pair stage of the batch that used the set before), everything else on the context stream
// (behind its seed stage), so that the seed stage of the next batch -- issue-bound scalar / LDS work with long memory waits
// -- runs beside the bind stage of this one.  Otherwise everything runs on the context stream with set 0.
static int enqueue_batch(tnt_ctx *c, tnt_db *db, size_t u0, size_t u1, bool seeds_only, bool pipelined = false)
{
	// one internal batch = a range of seed work units; target indices stay those of the whole database
	const DbView v = make_view(db, 0, db->n_targets);
	if (!pipelined) c->cur = 0;
	tnt_ctx::BatchSet &W = c->S();
	uint32_t *cnt = W.counters.p, *gcnt = c->G();
	hipStream_t seed_stream = pipelined ? c->seed_stream : c->stream;
	if (pipelined && W.used) HIP_TRY(hipStreamWaitEvent(seed_stream, W.released, 0), c->err); // the set is free again
	// hits and the error word accumulate over batches; the rest restarts
	HIP_TRY(hipMemsetAsync(cnt + C_POOL, 0, sizeof(uint32_t) * 4, seed_stream), c->err);
	HIP_TRY(hipMemsetAsync(cnt + C_ITEMS, 0, sizeof(uint32_t) * 3 * N_SHARD * SHARD_STRIDE, seed_stream), c->err);

	hipEvent_t ev[5]; // seed start, bind start, bind end, pair end, seed end
	for (int i = 0; i < 5; i++) {
		HIP_TRY(hipEventCreate(&ev[i]), c->err);
		c->events.push_back(ev[i]);
	}
	HIP_TRY(hipEventRecord(ev[0], seed_stream), c->err);

	SeedArgs s;
	memset(&s, 0, sizeof(s));
	s.db = v;
	s.unit_target = db->unit_target.p;
	s.unit_begin = db->unit_begin.p;
	s.unit_end = db->unit_end.p;
	s.unit0 = uint32_t(u0);
	s.unit1 = uint32_t(u1);
	s.word_size = c->opt.hash_word_size;
	s.wlist = c->wlist.p;
	s.wl_off = c->wl_off.p;
	s.wl_seq = c->wl_seq.p;
	s.max_len = c->opt.max_len;
	s.single_primer_pcr = c->opt.single_primer_pcr;
	if (!seeds_only) {
		s.items = c->S().items.p;
		s.item_cap = uint32_t(c->item_cap);
		s.item_count = cnt + C_ITEMS;
		s.groups = c->S().groups.p;
		s.group_cap = uint32_t(c->group_cap);
		s.group_count = cnt + C_GROUPS;
		s.qa = c->S().qa.p;
		s.qa_count = cnt + C_QA;
		s.big_count = cnt + C_BIG;
		if (c->dedupe) s.slot_canon = c->slot_canon.p, s.slot_next = c->slot_next.p, s.item_rep = c->S().item_rep.p;
	} else {
		s.seeds_out = c->seeds_out.p;
		s.seeds_cap = uint32_t(c->seeds_out.n);
		s.seeds_count = cnt + C_SEEDS;
	}
	s.err = gcnt + C_ERR;
	s.debug = gcnt + C_DEBUG + 8;
	// the round-3 seed kernel: private regions, phase-A items straight into their buckets' queues (its queue entries hold the
	// windows' packed-word index in 32 bits)
	const bool direct = c->seed2 && !seeds_only && db->h_word_off[db->n_targets] < (uint64_t(1) << 32);
	c->batch_direct = direct;
	if (direct) {
		// four 256-thread workgroups per CU that read the chunk's bucket offsets through L1 / L2; on request the instance that
		// keeps them in LDS (hash words of up to 7 bases, one 1024-thread workgroup per CU: 16 wavefronts per CU either way) --
		// built in round 4 and measured SLOWER (config 2: seed 5.9 against 5.3 ms, config 4 per 100 k targets: 50.7 against 37.2 ms;
		// DESIGN.md 4.1), although the scan on its own runs 3.6 x faster out of LDS (tnt_seed_scan_only)
		const bool ltab = c->opt.hash_word_size <= 7 && getenv("TNT_MI355X_SEED2_LDS_TABLE");
		const size_t S2_WAVES = size_t(s2_waves(ltab));
		const bool big = c->seed2_big && !ltab;
		const uint32_t blocks = uint32_t(std::min<size_t>((u1 - u0 + S2_WAVES - 1) / S2_WAVES,
		                                                   std::min<size_t>(size_t(c->n_cu) * (ltab ? 1 : size_t(big ? std::min(c->seed2_wgs, S2_OCC_BIG) : c->seed2_wgs)), tnt_ctx::MAX_REGIONS / S2_WAVES)));
		const uint32_t regions = blocks * uint32_t(S2_WAVES);
		c->batch_regions = regions;
		HIP_TRY(hipMemsetAsync(c->S().region_counts.p, 0, sizeof(uint32_t) * 2 * regions, seed_stream), c->err);
		HIP_TRY(hipMemsetAsync(c->S().q_state.p, 0, sizeof(uint32_t) * size_t(regions) * MAX_BUCKETS * 2, seed_stream), c->err);
		HIP_TRY(hipMemsetAsync(cnt + C_BUCKET, 0, sizeof(uint32_t) * MAX_BUCKETS * BUCKET_STRIDE, seed_stream), c->err);
		if (c->regroup && c->rg_total_bins) HIP_TRY(hipMemsetAsync(c->S().rg_hist.p, 0, sizeof(uint32_t) * c->rg_total_bins, seed_stream), c->err);
		if (c->regroup && c->rg_total_bins && c->S().rg_fill.p) HIP_TRY(hipMemsetAsync(c->S().rg_fill.p, 0, sizeof(uint32_t) * c->rg_total_bins, seed_stream), c->err);
		Seed2Args q;
		memset(&q, 0, sizeof(q));
		q.db = v;
		q.unit_rec = db->unit_rec.p;
		q.unit0 = uint32_t(u0), q.unit1 = uint32_t(u1);
		q.word_size = c->opt.hash_word_size;
		q.wlist = c->wlist.p, q.wl_off = c->wl_off.p;
		q.assays = c->assays.p;
		q.max_len = c->opt.max_len;
		q.single_primer_pcr = c->opt.single_primer_pcr;
		q.slot_pack = c->slot_pack.p;
		q.has_exceptions = db->n_exceptions != 0 ? 1 : 0;
		q.items = c->S().items.p, q.item_region = uint32_t(c->item_cap / regions);
		q.groups = c->S().groups.p, q.group_region = uint32_t(c->group_cap / regions);
		q.region_counts = c->S().region_counts.p;
		q.queue = c->S().qfat.p;
		q.result = c->S().result.p;
		q.q_base = c->q_base.p, q.q_cap = c->q_cap.p;
		q.q_count = cnt + C_BUCKET;
		q.q_state = c->S().q_state.p;
		q.n_buckets = int(c->bucket_lmax.size());
		q.stat_items = cnt + C_ITEMS, q.stat_groups = cnt + C_GROUPS, q.stat_phase_a = cnt + C_QA;
		q.big_count = cnt + C_BIG;
		q.stat_cells = reinterpret_cast<unsigned long long *>(gcnt + C_TOT_CELLS_LO);
		if (c->rg_seed_counts && c->S().rg_hist.p) // the entries per (oligo, class) group, counted as they are written
			q.slot_bin = c->slot_bin.p, q.bin_base = c->rg_bin_base_dev.p, q.bin_hist = c->S().rg_hist.p, q.n_bins = c->rg_total_bins;
		q.err = gcnt + C_ERR;
		q.debug = gcnt + C_DEBUG + 8;
		const bool units_short = !getenv("TNT_MI355X_SEED2_GENERAL") && db->max_short_len <= size_t(S2_TGT_WORDS) * 16 && c->opt.max_len >= 0 &&
		                         size_t(db->max_short_len) <= size_t(c->opt.max_len) + 50 + size_t(c->opt.hash_word_size);
		const size_t n_words2 = size_t(1) << (2 * c->opt.hash_word_size);
		const size_t off_stride2 = (n_words2 + 1 + 1) & ~size_t(1);
		for (size_t ch = 0; ch + 1 < c->chunk_begin.size(); ch++) {
			q.assay_begin = c->chunk_begin[ch];
			q.assay_end = c->chunk_begin[ch + 1];
			q.wt_off = c->wt_off.p + ch * off_stride2;
			q.wt_ent = c->wt_ent.p + c->chunk_ent_base[ch];
			q.wt_seq = reinterpret_cast<const uint4 *>(c->wt_seq.p) + c->chunk_ent_base[ch];
			q.wt_rec = c->wt_rec.p + c->chunk_ent_base[ch];
			q.chunk_has_probe = 0;
			for (uint32_t ai = q.assay_begin; ai < q.assay_end; ai++)
				if (c->h_assays[ai].kind == 0 && c->h_assays[ai].has_probe) q.chunk_has_probe = 1;
			q.last_chunk = ch + 2 == c->chunk_begin.size() ? 1 : 0;
			// (an instance compiled for the default word size -W 7, one that takes it at run time)
			// ... and one for databases of short targets (SHORT: every unit in LDS and within the first cull's reach as a whole)
			// ... and the ones with the large seed lists for assay sets of many chunks (BIG, plan())
			const dim3 g2(blocks), b2(64 * s2_waves(false));
			if (q.word_size == 7 && ltab) hipLaunchKernelGGL((seed2_kernel<7, true, false>), dim3(blocks), dim3(64 * s2_waves(true)), 0, seed_stream, q);
			else if (ltab) hipLaunchKernelGGL((seed2_kernel<0, true, false>), dim3(blocks), dim3(64 * s2_waves(true)), 0, seed_stream, q);
			else if (q.word_size == 7 && units_short && big) hipLaunchKernelGGL((seed2_kernel<7, false, true, true>), g2, b2, 0, seed_stream, q);
			else if (q.word_size == 7 && units_short) hipLaunchKernelGGL((seed2_kernel<7, false, true, false>), g2, b2, 0, seed_stream, q);
			else if (q.word_size == 7 && big) hipLaunchKernelGGL((seed2_kernel<7, false, false, true>), g2, b2, 0, seed_stream, q);
			else if (q.word_size == 7) hipLaunchKernelGGL((seed2_kernel<7, false, false, false>), g2, b2, 0, seed_stream, q);
			else if (units_short && big) hipLaunchKernelGGL((seed2_kernel<0, false, true, true>), g2, b2, 0, seed_stream, q);
			else if (units_short) hipLaunchKernelGGL((seed2_kernel<0, false, true, false>), g2, b2, 0, seed_stream, q);
			else if (big) hipLaunchKernelGGL((seed2_kernel<0, false, false, true>), g2, b2, 0, seed_stream, q);
			else hipLaunchKernelGGL((seed2_kernel<0, false, false, false>), g2, b2, 0, seed_stream, q);
		}
	}
	const int seed_blocks = int(std::min<size_t>((u1 - u0 + SEED_WAVES - 1) / SEED_WAVES, size_t(c->bind_blocks) * 4));
	// the streaming instance keeps the chunk's tables in LDS: one persistent workgroup per CU
	const int lds_blocks = int(std::min<size_t>((u1 - u0 + SEED_LDS_WAVES - 1) / SEED_LDS_WAVES, size_t(c->n_cu)));
	const size_t n_words = size_t(1) << (2 * c->opt.hash_word_size);
	const size_t off_stride = (n_words + 1 + 1) & ~size_t(1); // every chunk's offset table starts on a dword
	for (size_t ch = 0; ch + 1 < c->chunk_begin.size() && !direct; ch++) {
		s.assay_begin = c->chunk_begin[ch];
		s.assay_end = c->chunk_begin[ch + 1];
		s.wt_off = c->wt_off.p + ch * off_stride;
		s.wt_ent = c->wt_ent.p + c->chunk_ent_base[ch];
		s.n_ent = c->chunk_ent_base[ch + 1] - c->chunk_ent_base[ch];
		if (c->seed_lds && s.n_ent <= uint32_t(SEED_LDS_ENT))
			launch_seed<true>(s, seeds_only ? SEED_LIST : (s.item_rep ? SEED_SEARCH_SHARED : SEED_SEARCH), lds_blocks, 64 * SEED_LDS_WAVES, seed_stream);
		else
			launch_seed<false>(s, seeds_only ? SEED_LIST : (s.item_rep ? SEED_SEARCH_SHARED : SEED_SEARCH), seed_blocks, SEED_THREADS, seed_stream);
	}
	HIP_TRY(hipEventRecord(ev[4], seed_stream), c->err);
	if (pipelined) { // the stages behind the seed stage run on the context stream
		HIP_TRY(hipEventRecord(W.seeded, seed_stream), c->err);
		HIP_TRY(hipStreamWaitEvent(c->stream, W.seeded, 0), c->err);
	}
	HIP_TRY(hipEventRecord(ev[1], c->stream), c->err);
	if (seeds_only) { // (every event of the set is recorded: the stage timings walk over all of them)
		HIP_TRY(hipEventRecord(ev[2], c->stream), c->err);
		HIP_TRY(hipEventRecord(ev[3], c->stream), c->err);
		return TNT_OK;
	}

	// phase A: first strand of the PCR groups, every seed of the probe / padlock groups
	BindArgs b = bind_args(c, v);
	launch_bind_phase(c, b, v, db->n_exceptions != 0, c->S().qa.p, nullptr, cnt + C_QA, 0, direct);
	if (c->dedupe) {
		PropagateArgs pa;
		memset(&pa, 0, sizeof(pa));
		pa.src = c->S().qa.p;
		pa.shard_counts = cnt + C_QA;
		pa.shard_region = uint32_t(c->item_cap / N_SHARD);
		pa.cap = uint32_t(c->item_cap);
		pa.items = c->S().items.p;
		pa.item_rep = c->S().item_rep.p;
		pa.result = c->S().result.p;
		pa.groups = c->S().groups.p;
		pa.active = c->S().active.p;
		pa.n_active = cnt + C_ACTIVE;
		pa.n_dups = cnt + C_DUPS;
		hipLaunchKernelGGL(propagate_kernel, dim3(1024), dim3(256), 0, c->stream, pa);
	}

	FilterArgs f;
	f.groups = c->S().groups.p;
	f.items = c->S().items.p;
	f.active = c->S().active.p;
	f.n_active = cnt + C_ACTIVE;
	f.qb = c->S().qb.p;
	f.qb_count = cnt + C_QB;
	hipLaunchKernelGGL(filter_kernel, dim3(256), dim3(256), 0, c->stream, f);

	// phase B: the other strand and the probes of the groups that are still alive
	launch_bind_phase(c, b, v, db->n_exceptions != 0, c->S().qb.p, cnt + C_QB, nullptr, getenv("TNT_MI355X_PHASE_B_BUCKETS") ? 0 : 3);
	HIP_TRY(hipEventRecord(ev[2], c->stream), c->err);

	PairArgs p;
	memset(&p, 0, sizeof(p));
	p.groups = c->S().groups.p;
	p.active = c->S().active.p;
	p.n_active = cnt + C_ACTIVE;
	p.items = c->S().items.p;
	p.result = c->S().result.p;
	p.pool = c->S().pool.p;
	p.assays = c->assays.p;
	p.slots = c->slots.p;
	p.db = v;
	p.max_len = c->opt.max_len;
	p.single_primer_pcr = c->opt.single_primer_pcr;
	p.min_max_primer_clamp = c->opt.min_max_primer_clamp;
	p.assay_format = c->opt.assay_format;
	p.target_strand = c->opt.target_strand;
	p.hits = c->hits.p;
	p.hit_cap = uint32_t(c->hit_cap);
	p.hit_count = gcnt + C_HITS;
	p.err = gcnt + C_ERR;
	p.n_big = cnt + C_BIG;
	p.unit_target = db->unit_target.p;
	p.unit_own_lo = db->unit_own_lo.p;
	p.unit_own_hi = db->unit_own_hi.p;
	// groups of more than PAIR_MAX items need the instance with the large per-lane lists; it is only launched where
	// they can be expected (dense seeding, windows of long targets) or after a search ran into one
	{
		size_t max_ent = 1;
		for (const auto &ae : c->assay_entries) max_ent = std::max(max_ent, ae.size());
		const double expect = double(db->max_unit_len) * double(max_ent) / double(size_t(1) << (2 * c->opt.hash_word_size));
		p.big_enabled = (c->force_big || expect > 16.0) ? 1 : 0;
	}
	size_t cache_groups = 0, cache_items = 0, cache_tmp = 0;
	if (c->cache_replay) { // (the pair kernels leave the groups of the flagged assays to the replay: p.citems set)
		const int crc = ensure_cache_buffers(c, u1 - u0, cache_groups, cache_items, cache_tmp);
		if (crc) return crc;
		p.citems = c->c_items.p;
	}
	{ // the pair kernels only see the groups that can report something (tnt_pair.hip.h, pair_select_kernel)
		HIP_TRY(hipMemsetAsync(cnt + C_PAIRABLE, 0, sizeof(uint32_t), c->stream), c->err);
		PairSelectArgs ps;
		ps.p = p;
		ps.selected = c->S().pairable.p;
		ps.n_selected = cnt + C_PAIRABLE;
		hipLaunchKernelGGL(pair_select_kernel, dim3(2048), dim3(256), 0, c->stream, ps);
		p.active = c->S().pairable.p;
		p.n_active = cnt + C_PAIRABLE;
	}
	hipLaunchKernelGGL(pair_kernel<PAIR_SMALL>, dim3(8192), dim3(PAIR_THREADS), 0, c->stream, p);
	hipLaunchKernelGGL(pair_kernel<PAIR_MAX>, dim3(2048), dim3(PAIR_THREADS), 0, c->stream, p);
	if (p.big_enabled) hipLaunchKernelGGL(pair_kernel<PAIR_BIG>, dim3(64), dim3(PAIR_THREADS), 0, c->stream, p);
	if (c->cache_replay) {
		const int crc = enqueue_cache_replay(c, p, direct, cache_groups, cache_items, cache_tmp, cnt);
		if (crc) return crc;
	}
	hipLaunchKernelGGL(accumulate_kernel, dim3(1), dim3(1), 0, c->stream, cnt, gcnt);
	HIP_TRY(hipEventRecord(ev[3], c->stream), c->err);
	if (pipelined) {
		HIP_TRY(hipEventRecord(W.released, c->stream), c->err);
		W.used = true;
	}
	HIP_TRY(hipGetLastError(), c->err);
	return TNT_OK;
}

// Assay chunks for the seed kernel and target batches sized so the expected number of seeds fits.
// Every chunk gets its own bucket table (16-bit offsets, < 65536 entries) so it can sit in LDS.
static int plan(tnt_ctx *c, const tnt_db *db, size_t &targets_per_batch, bool for_search = false)
{
	const size_t n_words = size_t(1) << (2 * c->opt.hash_word_size);
	const double words = double(n_words);
	const double per_base = double(c->n_entries) / words; // expected raw seeds per target base
	const double maxlen = std::max<double>(1.0, db->max_unit_len); // longest seed work unit (target or window)
	const size_t n_assays = c->h_assays.size();
	// Assay chunks: the seeds one work unit of the longest length yields for the chunk must fit the LDS seed list
	// (SEED_CAP).  For random sequence their number is a sum over the positions of m_w = entries of the position's
	// word: mean = len * sum(m_w) / 4^k, variance = len * sum(m_w^2) / 4^k (expanded degenerate assays share words, so
	// their seeds come in bursts).  A chunk grows while mean + 6 sigma stays below the list size; a target that
	// overflows it anyway makes tnt_search_collect halve chunk_scale and repeat the search.
	const bool s2_ltab = c->opt.hash_word_size <= 7 && getenv("TNT_MI355X_SEED2_LDS_TABLE");
	const double cap = double(c->seed2 ? s2_cap(s2_ltab) : SEED_CAP) * c->chunk_scale;
	// the streaming seed kernel keeps a chunk's table in LDS: hash words up to 7 bases, at most SEED_LDS_ENT entries
	// (measured slower than the instance reading them through L2 at twice the occupancy: opt-in until the stages behind
	// the scan are split off into their own launch)
	c->seed_lds = c->opt.hash_word_size <= SEED_LDS_WORD_SIZE && getenv("TNT_MI355X_SEED_LDS_TABLES");
	const size_t ent_cap = c->seed_lds ? size_t(SEED_LDS_ENT) : size_t(65535);
	std::vector<uint32_t> mult(n_words);
	auto cut = [&](const double list_cap) {
		std::vector<uint32_t> begin;
		size_t a0 = 0;
		while (a0 < n_assays) {
			size_t a1 = a0, ent = 0;
			double sum_sq = 0.0;
			std::fill(mult.begin(), mult.end(), 0u);
			while (a1 < n_assays && a1 - a0 < size_t(c->seed2 ? S2_ASSAYS : SEED_MAX_ASSAYS)) {
				const auto &ae = c->assay_entries[a1];
				double add_sq = 0.0;
				for (const auto &we : ae) add_sq += 2.0 * double(mult[we.first]++) + 1.0;
				const double mean = maxlen * double(ent + ae.size()) / words, sigma = sqrt(maxlen * (sum_sq + add_sq) / words);
				if (a1 > a0 && (mean + 6.0 * sigma > list_cap || ent + ae.size() > ent_cap)) break;
				ent += ae.size();
				sum_sq += add_sq;
				a1++;
			}
			begin.push_back(uint32_t(a0));
			a0 = a1;
		}
		begin.push_back(uint32_t(n_assays));
		return begin;
	};
	std::vector<uint32_t> chunk_begin = cut(cap);
	// Every chunk launch of the round-3 seed kernel scans the database again: an assay set that fills six chunks or more takes
	// the instances with the large seed lists when that spares three chunks in ten (three workgroups per CU instead of four;
	// tnt_seed2.hip.h, S2_CAP_BIG).  Only where every unit goes through that kernel: the windows of long targets are seeded by
	// the round-2 kernel, whose lists hold SEED_CAP.
	c->seed2_big = false;
	if (for_search && c->seed2 && !s2_ltab && !getenv("TNT_MI355X_SEED2_STD_LISTS") && db->h_long_targets.empty() && db->h_word_off[db->n_targets] < (uint64_t(1) << 32) &&
	    chunk_begin.size() - 1 >= 6) {
		std::vector<uint32_t> big = cut(double(S2_CAP_BIG) * c->chunk_scale);
		if ((big.size() - 1) * 10 <= (chunk_begin.size() - 1) * 7) chunk_begin.swap(big), c->seed2_big = true;
	}
	if (chunk_begin != c->planned_chunks) {
		std::vector<uint16_t> off;
		std::vector<uint32_t> ent;
		c->chunk_ent_base.clear();
		std::vector<std::vector<uint32_t>> buckets(n_words);
		for (size_t ch = 0; ch + 1 < chunk_begin.size(); ch++) {
			for (auto &b : buckets) b.clear();
			for (uint32_t ai = chunk_begin[ch]; ai < chunk_begin[ch + 1]; ai++)
				for (const auto &we : c->assay_entries[ai]) buckets[we.first].push_back(we.second);
			c->chunk_ent_base.push_back(uint32_t(ent.size()));
			const size_t base = ent.size();
			for (size_t w = 0; w < n_words; w++) {
				off.push_back(uint16_t(ent.size() - base));
				ent.insert(ent.end(), buckets[w].begin(), buckets[w].end());
			}
			if (ent.size() - base > 65535) {
				c->err = "one assay has more than 65535 query words";
				return TNT_E_LIMIT;
			}
			off.push_back(uint16_t(ent.size() - base));
			if (off.size() & 1) off.push_back(0); // the next chunk's table starts on a dword (the LDS instance copies dwords)
		}
		c->chunk_ent_base.push_back(uint32_t(ent.size()));
		off.push_back(0), off.push_back(0);
		HIP_TRY(hipStreamSynchronize(c->stream), c->err);
		HIP_TRY(c->wt_off.upload(off), c->err);
		HIP_TRY(c->wt_ent.upload(ent), c->err);
		{
			std::vector<unsigned long long> es(ent.size() * 2 + 2, 0ull);
			for (size_t i = 0; i < ent.size(); i++) {
				const uint32_t en = ent[i];
				const size_t li = (size_t(en >> WT_ASSAY_SHIFT) * 3u + ((en >> 8) & 3u)) * 2u + ((en & WT_MINUS) ? 1u : 0u);
				es[2 * i] = c->h_wl_seq[2 * li], es[2 * i + 1] = c->h_wl_seq[2 * li + 1];
			}
			HIP_TRY(c->wt_seq.upload(es), c->err);
			// the dense pass of the round-3 seed kernel reads an entry and the first 32 bases of its oligo as one record
			std::vector<uint4> rec(ent.size() + 1, make_uint4(0, 0, 0, 0));
			for (size_t i = 0; i < ent.size(); i++) rec[i] = make_uint4(ent[i], uint32_t(es[2 * i]), uint32_t(es[2 * i] >> 32), 0u);
			HIP_TRY(c->wt_rec.upload(rec), c->err);
		}
		c->planned_chunks = chunk_begin;
	}
	c->chunk_begin = chunk_begin;
	// batch of targets: expected kept seeds (<= raw) must stay below c->item_cap / 2
	// (units = targets, or overlapping windows of long targets: count their positions, not the bases)
	const size_t n_units = std::max<size_t>(1, db->h_unit_target.size());
	double unit_positions = 0.0;
	if (db->split) {
		for (size_t u = 0; u < db->h_unit_target.size(); u++) unit_positions += double(db->h_unit_end[u] - db->h_unit_begin[u]);
	} else {
		unit_positions = double(db->n_bases);
	}
	const double avg_len = unit_positions / double(n_units);
	const double per_target = std::max(1.0, per_base * avg_len);
	// (per_target counts raw seeds; the first cull keeps a small part of them for PCR assays, all of them for probes:
	// a batch is sized for every raw seed becoming an item, an overflow of a shard region is repaired by halving it)
	double tb = (double(c->item_cap) * 0.75) / per_target;
	const double gb = (double(c->group_cap) / 2.0) / std::max<double>(1.0, double(n_assays));
	tb = std::min(tb, gb);
	targets_per_batch = size_t(std::max(1.0, std::min(tb, double(n_units)))); // in work units
	// Regions of the phase-A queues (round-3 seed kernel): the buffer is shared out by the expected number of seeds of the
	// slots a bucket serves (query words of their oligos; windows with a degenerate target base go to the general bucket of
	// their length, so with such a database both get the share).  A region that overflows anyway raises ERR_ITEM_OVERFLOW:
	// the search is repeated with smaller batches.
	const bool exc = db->n_exceptions != 0;
	if (c->seed2 && (c->h_q_base.empty() || c->q_planned_exceptions != exc)) {
		const size_t nb = c->bucket_lmax.size();
		std::vector<double> weight(nb, 0.0);
		for (size_t si = 0; si < c->h_slots.size(); si++) {
			const Slot &sl = c->h_slots[si];
			if (sl.oligo < 0) continue;
			const double w = double(std::max<size_t>(1, c->oligos[size_t(sl.oligo)].size()));
			if (c->opt.use_dinkelbach) {
				weight[nb - 1] += w;
				continue;
			}
			weight[size_t(sl.bucket)] += w;
			if (exc) weight[size_t(sl.bucket_general)] += w;
		}
		double total = 0.0;
		for (double w : weight) total += w;
		// every wavefront may hold one open block per bucket on top of the bucket's share
		size_t nb_used = 0;
		for (size_t b = 0; b < nb; b++) nb_used += weight[b] > 0.0 ? 1 : 0;
		const uint64_t floor_slots = std::min<uint64_t>(uint64_t(tnt_ctx::MAX_REGIONS) * QUEUE_BLOCK, (uint64_t(c->item_cap) / (2 * std::max<size_t>(1, nb_used))) & ~uint64_t(QUEUE_BLOCK - 1));
		uint64_t used = 0;
		for (size_t b = 0; b < nb; b++) used += weight[b] > 0.0 ? floor_slots : 0;
		const double spare = double(c->item_cap) - double(used);
		c->h_q_base.assign(nb, 0);
		c->h_q_cap.assign(nb, 0);
		uint64_t at = 0;
		for (size_t b = 0; b < nb; b++) {
			if (!(weight[b] > 0.0)) continue;
			uint64_t cap = floor_slots + uint64_t(spare * weight[b] / total);
			// (test hook: regions of at most this many slots, so that a small workload overflows one and meets the repair)
			if (const char *e = getenv("TNT_MI355X_QUEUE_SLOTS")) cap = std::min<uint64_t>(cap, std::max<uint64_t>(QUEUE_BLOCK, strtoull(e, nullptr, 10)));
			cap &= ~uint64_t(QUEUE_BLOCK - 1);
			c->h_q_base[b] = uint32_t(at);
			c->h_q_cap[b] = uint32_t(cap);
			at += cap;
		}
		HIP_TRY(hipStreamSynchronize(c->stream), c->err);
		HIP_TRY(c->q_base.upload(c->h_q_base), c->err);
		HIP_TRY(c->q_cap.upload(c->h_q_cap), c->err);
		c->q_planned_exceptions = exc;
		// the buckets whose queue is regrouped by oligo in front of the pre-filter: those the filter serves (launch_bind_phase),
		// with an instance of ufilter_kernel and a histogram that fits LDS
		c->h_rg_desc.clear();
		c->rg_desc_of.assign(nb, -1);
		for (size_t b = 0; b < nb && c->regroup; b++) {
			const int id = c->bucket_lmax[b], rows = id & 0xFF;
			const bool plain = (id & BUCKET_PLAIN) != 0;
			if (c->h_q_cap[b] == 0 || (id & BUCKET_WAVE) || rows > 64 || !(plain || c->filter_general) || !has_filter_instance(rows, !plain) || !has_ufilter_instance(rows, !plain)) continue;
			if (c->rg_bins[b] == 0 || c->rg_bins[b] > uint32_t(RG_MAX_BINS)) continue;
			RegroupBucket B;
			memset(&B, 0, sizeof(B));
			B.src_base = c->h_q_base[b], B.src_cap = c->h_q_cap[b];
			B.dst_base = c->h_q_base[b] + uint32_t(UF_BLOCK) * c->rg_bin_base[b];
			B.bin_base = c->rg_bin_base[b], B.n_bins = c->rg_bins[b];
			B.general = plain ? 0u : 1u;
			B.counter = uint32_t(b) * BUCKET_STRIDE;
			c->rg_desc_of[b] = int(c->h_rg_desc.size());
			c->h_rg_desc.push_back(B);
		}
		if (!c->h_rg_desc.empty()) HIP_TRY(c->rg_desc.upload(c->h_rg_desc), c->err);
	}
	return TNT_OK;
}

#include "tnt_long.hip.h" // search_long_targets: the targets that are seeded in windows

static int enqueue_compaction(tnt_ctx *ctx, size_t cap_hits, const uint32_t *n_dev); // (with tnt_search_hits_compact, below)

int tnt_search_enqueue(tnt_ctx *ctx, tnt_db *db)
{
	if (ctx) ctx->dev_hits_valid = false;
	if (!ctx || !db || db->ctx != ctx) return TNT_E_ARG;
	HIP_TRY(hipSetDevice(ctx->device), ctx->err);
	int rc = ensure_work_buffers(ctx, ctx->cache_replay ? std::max(ctx->lmax, ctx->unit_rows) : ctx->lmax, db); // (the replay binds through the unit-level instance)
	if (rc) return rc;
	for (hipEvent_t e : ctx->events) (void)hipEventDestroy(e);
	ctx->events.clear();
	ctx->cur = 0;
	size_t tpb = 1;
	rc = plan(ctx, db, tpb, true); // (the search may take the seed kernel's large-list instances)
	if (rc) return rc;
	tpb = std::max<size_t>(1, size_t(double(tpb) * ctx->batch_scale));
	ctx->n_batches = 0;
	const size_t n_units = db->n_short_units; // the windows of long targets are handled by search_long_targets
	// Large searches are cut into several batches on two buffer sets so that the seed stage of one batch runs beside the bind
	// stage of the batch before it (enqueue_batch); small ones stay one batch on the context stream.
	const bool pipelined = ctx->pipeline_batches > 1 && n_units >= size_t(ctx->pipeline_batches) * 32768 && !ctx->cache_replay;
	if (pipelined) {
		tpb = std::min(tpb, (n_units + size_t(ctx->pipeline_batches) - 1) / size_t(ctx->pipeline_batches));
		rc = ensure_set(ctx, 1);
		if (rc) return rc;
		if (!ctx->seed_stream) HIP_TRY(hipStreamCreateWithFlags(&ctx->seed_stream, hipStreamNonBlocking), ctx->err);
		HIP_TRY(hipMemsetAsync(ctx->set[1].counters.p, 0, sizeof(uint32_t) * C_COUNT, ctx->stream), ctx->err);
	}
	HIP_TRY(hipMemsetAsync(ctx->set[0].counters.p, 0, sizeof(uint32_t) * C_COUNT, ctx->stream), ctx->err);
	ctx->set[0].used = ctx->set[1].used = false;
	hipEvent_t start = nullptr;
	if (pipelined) { // the seed stream starts behind whatever the context stream holds (the counters' reset above)
		HIP_TRY(hipEventCreateWithFlags(&start, hipEventDisableTiming), ctx->err);
		HIP_TRY(hipEventRecord(start, ctx->stream), ctx->err);
		HIP_TRY(hipStreamWaitEvent(ctx->seed_stream, start, 0), ctx->err);
	}
	for (size_t u0 = 0; u0 < n_units; u0 += tpb) {
		if (pipelined) ctx->cur = int(ctx->n_batches & 1);
		rc = enqueue_batch(ctx, db, u0, std::min(n_units, u0 + tpb), false, pipelined);
		if (rc) break;
		ctx->n_batches++;
	}
	ctx->cur = 0;
	if (start) (void)hipEventDestroy(start);
	if (rc) return rc;
	ctx->blob_cap_hits = 0, ctx->blob_hits = 0, ctx->blob_bytes = 0;
	// (the hits of windowed targets join the list on the host, at collect time; the segments' hits of probes beyond 128 bases are
	// merged there too -- amplicon_start carries the seed's diagonal until then -- so the blob is packed from the merged records)
	if (ctx->compact_in_search && db->h_long_targets.empty() && !ctx->long_probes) {
		const size_t cap_hits = std::min<size_t>(ctx->hit_cap, std::max<size_t>(4 * ctx->dev_hits_n, size_t(1) << 16));
		rc = enqueue_compaction(ctx, cap_hits, ctx->G() + C_HITS);
		if (rc) return rc;
		ctx->blob_cap_hits = cap_hits;
	}
	ctx->pending = true;
	ctx->pending_db = db;
	return TNT_OK;
}

// Shared by tnt_search_collect and tnt_search_collect_compact.  want_records: the 1 464-byte hit records are copied to the
// library's pinned host buffer (47 MB per search of config 2); without, they stay on the device (tnt_search_hits_device /
// _compact and the gang's gather take them from there).
static int collect_search(tnt_ctx *ctx, bool want_records, const tnt_hit **hits, size_t *n_hits)
{
	if (!ctx || !ctx->pending) return TNT_E_ARG;
	if (ctx->pending_db && !ctx->pending_db->h_long_targets.empty()) want_records = true; // (the hits of windowed targets are made on the host)
	if (ctx->long_probes) want_records = true; // (the hits of a long probe's segments are merged on the host)
	HIP_TRY(hipSetDevice(ctx->device), ctx->err);
	uint32_t cnt[C_COUNT];
	HIP_TRY(hipMemcpyAsync(cnt, ctx->S().counters.p, sizeof(cnt), hipMemcpyDeviceToHost, ctx->stream), ctx->err);
	HIP_TRY(hipStreamSynchronize(ctx->stream), ctx->err);
	ctx->pending = false;
	for (int i = 0; i < MAX_BUCKETS; i++) { // how long the buckets' chains of the last batch took: the next search deals them by these
		if (!ctx->chain_timed[i]) continue;
		float ms = 0.0f;
		if (hipEventElapsedTime(&ms, ctx->chain_ev[i][0], ctx->chain_ev[i][1]) == hipSuccess && ms > 0.0f) ctx->chain_ms[i] = double(ms);
		else (void)hipGetLastError();
		ctx->chain_timed[i] = false;
	}
#ifdef TNT_EXP_STATS
	{
		const uint32_t *d = cnt + C_DEBUG;
		const double w = d[7] ? double(d[7]) : 1.0;
		if (ctx->seed2)
			fprintf(stderr, "seed2 waves %u; kilo-cycles per wave: setup %.0f scan %.0f dense %.0f sort %.0f cull %.0f emit %.0f\n", d[15], double(d[8]) / (d[15] ? d[15] : 1),
			        double(d[9]) / (d[15] ? d[15] : 1), double(d[10]) / (d[15] ? d[15] : 1), double(d[11]) / (d[15] ? d[15] : 1), double(d[12]) / (d[15] ? d[15] : 1), double(d[13]) / (d[15] ? d[15] : 1));
		else
		fprintf(stderr, "seed waves %u; kilo-cycles per wave: scan+dedupe %.0f (dedupe %.0f) sort %.0f cull %.0f phaseB %.0f prefix+groups %.0f emit %.0f\n", d[15], double(d[8]) / (d[15] ? d[15] : 1),
		        double(d[14]) / (d[15] ? d[15] : 1), double(d[9]) / (d[15] ? d[15] : 1), double(d[10]) / (d[15] ? d[15] : 1), double(d[11]) / (d[15] ? d[15] : 1), double(d[12]) / (d[15] ? d[15] : 1), double(d[13]) / (d[15] ? d[15] : 1));
		fprintf(stderr, "wave-DPs %u; cycles per wave-DP: fill %.0f trace %.0f trim %.0f eval %.0f other-enum %.0f thresholds %.0f output %.0f\n", d[7],
		        64.0 * d[0] / w, 64.0 * d[1] / w, 64.0 * d[2] / w, 64.0 * d[3] / w, 64.0 * d[4] / w, 64.0 * d[5] / w, 64.0 * d[6] / w);
	}
#endif
	// Capacity overflows are repaired and the search repeated: the hit buffer grows to what the pair kernel counted,
	// the per-batch lists (work items, groups, bound records) get smaller batches.
	// (a (target, assay) list beyond the crowded-group instance's 512 entries: the whole database takes the route of the targets that
	// are seeded in windows, whose lists -- up to 2^20 entries -- are culled and paired on host threads)
	const bool lists_to_host = ctx->force_big && ctx->pending_db && !ctx->pending_db->all_long && !ctx->long_primers;
	const uint32_t repairable = ERR_HIT_OVERFLOW | ERR_ITEM_OVERFLOW | ERR_GROUP_OVERFLOW | ERR_POOL_OVERFLOW | ERR_SEED_OVERFLOW |
	                            ((!ctx->force_big || lists_to_host) ? ERR_LIST_OVERFLOW : 0u);
	if (cnt[C_ERR] & repairable) {
		const uint32_t fatal = cnt[C_ERR] & ~repairable;
		if (!fatal && ctx->retries < 16 && ctx->pending_db) {
			ctx->retries++;
			if (cnt[C_ERR] & ERR_HIT_OVERFLOW) {
				ctx->hit_cap = std::max<size_t>(ctx->hit_cap * 2, size_t(cnt[C_HITS]) + size_t(cnt[C_HITS]) / 4);
				HIP_TRY(ctx->hits.alloc(ctx->hit_cap), ctx->err);
			}
			if (cnt[C_ERR] & (ERR_ITEM_OVERFLOW | ERR_GROUP_OVERFLOW | ERR_POOL_OVERFLOW)) ctx->batch_scale *= 0.5;
			if (cnt[C_ERR] & ERR_SEED_OVERFLOW) {
				// fewer assays per seed pass; with one assay per pass, shorter windows (the units of this database are rebuilt)
				if (ctx->chunk_begin.size() < ctx->h_assays.size() + 1) {
					ctx->chunk_scale *= 0.5;
				} else {
					// a whole target overflowed with one assay per pass: this database is seeded in windows throughout (the
					// search restarts from scratch below, so no hit is reported twice)
					if (ctx->pending_db->all_long) ctx->unit_scale *= 0.5;
					ctx->pending_db->all_long = true;
					HIP_TRY(build_units(ctx, ctx->pending_db), ctx->err);
				}
			}
			if (cnt[C_ERR] & ERR_LIST_OVERFLOW) {
				if (lists_to_host) {
					ctx->pending_db->all_long = true;
					HIP_TRY(build_units(ctx, ctx->pending_db), ctx->err);
				}
				ctx->force_big = true;
			}
			const int rc = tnt_search_enqueue(ctx, ctx->pending_db);
			if (rc) return rc;
			return collect_search(ctx, want_records, hits, n_hits);
		}
	}
	ctx->retries = 0;
	if (cnt[C_ERR]) {
		ctx->err = err_text(cnt[C_ERR]);
		return TNT_E_LIMIT;
	}
	const size_t n = std::min<size_t>(cnt[C_HITS], ctx->hit_cap);
	if (want_records && n > ctx->h_hits_cap) {
		if (ctx->h_hits) (void)hipHostFree(ctx->h_hits);
		ctx->h_hits = nullptr;
		ctx->h_hits_cap = 0;
		const size_t cap = std::max<size_t>(n + n / 2, 4096);
		HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&ctx->h_hits), cap * sizeof(tnt_hit), hipHostMallocDefault), ctx->err);
		ctx->h_hits_cap = cap;
	}
	if (n && want_records) HIP_TRY(hipMemcpy(ctx->h_hits, ctx->hits.p, n * sizeof(tnt_hit), hipMemcpyDeviceToHost), ctx->err);
	size_t n_unique = n;
	for (int i = 0; i < 4; i++) ctx->long_pair_stats[i] = 0;
	if (ctx->pending_db && !ctx->pending_db->h_long_targets.empty()) {
		// targets seeded in windows: lists merged per (target, assay), culled and paired whole on the host
		std::vector<tnt_hit> extra;
		const int rc = search_long_targets(ctx, ctx->pending_db, extra);
		// (its device arrays are kept for the next search -- allocating 17 GB for a 250-Mb record takes 0.4 s -- up to 48 GB)
		if (ctx->long_arena.cap + ctx->long_pair_arena.cap + ctx->seeds_out.n * sizeof(SeedOut) > (size_t(48) << 30)) {
			ctx->long_arena.release();
			ctx->long_pair_arena.release();
			ctx->long_pool.release();
			ctx->seeds_out.release();
		}
		if (rc) return rc;
		if (!extra.empty()) {
			const size_t total = n + extra.size();
			if (total > ctx->h_hits_cap) {
				tnt_hit *bigger = nullptr;
				const size_t cap = total + total / 2;
				HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&bigger), cap * sizeof(tnt_hit), hipHostMallocDefault), ctx->err);
				if (n) memcpy(bigger, ctx->h_hits, n * sizeof(tnt_hit));
				if (ctx->h_hits) (void)hipHostFree(ctx->h_hits);
				ctx->h_hits = bigger;
				ctx->h_hits_cap = cap;
			}
			memcpy(ctx->h_hits + n, extra.data(), extra.size() * sizeof(tnt_hit));
			n_unique = total;
			// the device copy follows (tnt_search_hits_device / _compact: the multi-GPU gather takes the records from there)
			if (total > ctx->hit_cap) {
				ctx->hit_cap = total + total / 2;
				HIP_TRY(ctx->hits.alloc(ctx->hit_cap), ctx->err);
				HIP_TRY(hipMemcpy(ctx->hits.p, ctx->h_hits, total * sizeof(tnt_hit), hipMemcpyHostToDevice), ctx->err);
			} else {
				HIP_TRY(hipMemcpy(ctx->hits.p + n, ctx->h_hits + n, extra.size() * sizeof(tnt_hit), hipMemcpyHostToDevice), ctx->err);
			}
		}
	}
	{ // alignment texts beyond TNT_ALIGN_TEXT (oligos of more than 64 bases) follow their records to the host
		if ((ctx->pending_db && !ctx->pending_db->h_long_targets.empty()) || ctx->S().counters.p != ctx->G()) // (the binds of the windowed targets went on filling the arena; the fill is a search-wide counter)
			HIP_TRY(hipMemcpy(&cnt[C_LTEXT], ctx->G() + C_LTEXT, sizeof(uint32_t), hipMemcpyDeviceToHost), ctx->err);
		const int lrc = fetch_long_texts(ctx, cnt[C_LTEXT]);
		if (lrc) return lrc;
	}
	if (ctx->long_probes && n_unique) {
		// A probe of more than 128 bases is seeded in segments (virtual assays): a site found through two segments is reported
		// by both.  hybrid() keeps, per strand, one record per footprint -- the one first in oligo_info::operator< order, i.e.
		// with the highest Tm (bind_oligo.cpp:431-453): the same rule across the segments.
		std::vector<uint32_t> order(n_unique);
		for (size_t i = 0; i < n_unique; i++) order[i] = uint32_t(i);
		const tnt_hit *H = ctx->h_hits;
		auto key_less = [H](uint32_t x, uint32_t y) {
			const tnt_hit &a = H[x], &b = H[y];
			if (a.degen_id != b.degen_id) return a.degen_id < b.degen_id;
			if (a.seq_index != b.seq_index) return a.seq_index < b.seq_index;
			if (a.probe_strand != b.probe_strand) return a.probe_strand < b.probe_strand;
			if (a.probe_start != b.probe_start) return a.probe_start < b.probe_start;
			if (a.probe_stop != b.probe_stop) return a.probe_stop < b.probe_stop;
			if (a.probe_tm != b.probe_tm) return a.probe_tm > b.probe_tm;
			return a.amplicon_start < b.amplicon_start; // the seed's diagonal (list order of the hash overload, bind_oligo.cpp:157)
		};
		std::vector<int32_t> long_ids; // degen_id of the assays whose probe is seeded in segments
		for (size_t ai = 0; ai < ctx->n_input_assays; ai++)
			if (ctx->h_assays[ai].cache_replay & 2) long_ids.push_back(ctx->h_assays[ai].degen_id);
		std::sort(long_ids.begin(), long_ids.end());
		auto segmented = [&](const tnt_hit &h) { return !h.has_primers && std::binary_search(long_ids.begin(), long_ids.end(), h.degen_id); };
		std::stable_sort(order.begin(), order.end(), key_less);
		std::vector<char> drop(n_unique, 0);
		size_t n_drop = 0;
		for (size_t i = 1; i < n_unique; i++) {
			const tnt_hit &a = H[order[i - 1]], &b = H[order[i]];
			if (segmented(b) && a.degen_id == b.degen_id && a.seq_index == b.seq_index && a.probe_strand == b.probe_strand && a.probe_start == b.probe_start &&
			    a.probe_stop == b.probe_stop && !a.has_primers)
				drop[order[i]] = 1, n_drop++;
		}
		size_t o = 0;
		for (size_t i = 0; i < n_unique; i++)
			if (!drop[i]) {
				if (o != i) memcpy(&ctx->h_hits[o], &ctx->h_hits[i], sizeof(tnt_hit));
				if (segmented(ctx->h_hits[o])) ctx->h_hits[o].amplicon_start = 0; // (it carried the seed's diagonal up to here)
				o++;
			}
		n_unique = o;
		(void)n_drop;
		HIP_TRY(hipMemcpy(ctx->hits.p, ctx->h_hits, n_unique * sizeof(tnt_hit), hipMemcpyHostToDevice), ctx->err); // (the device view and the compact form follow)
	}
	// stage timings
	tnt_timing &T = ctx->timing;
	memset(&T, 0, sizeof(T));
	// (with the seed stage of the next batch running beside the bind stage of this one the stage times overlap: their sum
	// exceeds total_ms)
	for (size_t b = 0; b + 4 < ctx->events.size(); b += 5) {
		float ms = 0;
		if (hipEventElapsedTime(&ms, ctx->events[b], ctx->events[b + 4]) == hipSuccess) T.seed_ms += ms;
		if (hipEventElapsedTime(&ms, ctx->events[b + 1], ctx->events[b + 2]) == hipSuccess) T.bind_ms += ms;
		if (hipEventElapsedTime(&ms, ctx->events[b + 2], ctx->events[b + 3]) == hipSuccess) T.pair_ms += ms;
	}
	if (ctx->events.size() >= 5) {
		float ms = 0;
		if (hipEventElapsedTime(&ms, ctx->events.front(), ctx->events[ctx->events.size() - 2]) == hipSuccess) T.total_ms = ms;
	}
	T.n_binds = (uint64_t(cnt[C_TOT_BINDS_HI]) << 32) | cnt[C_TOT_BINDS_LO]; // DP evaluations (phase A + phase B)
	T.n_seeds = (uint64_t(cnt[C_TOT_GROUPS_HI]) << 32) | cnt[C_TOT_GROUPS_LO]; // (target, assay) groups that reached the bind stage
	T.n_bound = cnt[C_TOT_BOUND];
	T.n_hits = n_unique;
	T.n_filtered = cnt[C_FILTERED];
	T.n_cells = (uint64_t(cnt[C_TOT_CELLS_HI]) << 32) | cnt[C_TOT_CELLS_LO];
	ctx->dev_hits_valid = true; // (the hits of long targets, paired on the host, were copied behind the device's own)
	ctx->dev_hits_n = n_unique;
	if (ctx->blob_cap_hits && n_unique <= ctx->blob_cap_hits && cnt[C_BLOB + 1] == n_unique) ctx->blob_hits = n_unique, ctx->blob_bytes = cnt[C_BLOB];
	if (hits) *hits = want_records ? ctx->h_hits : nullptr;
	if (n_hits) *n_hits = n_unique;
	return TNT_OK;
}

int tnt_search_collect(tnt_ctx *ctx, const tnt_hit **hits, size_t *n_hits)
{
	return collect_search(ctx, hits != nullptr, hits, n_hits); // hits == NULL: the records stay on the device
}

int tnt_search_collect_compact(tnt_ctx *ctx, const void **host_blob, size_t *n_bytes, size_t *n_hits)
{
	if (!ctx || !host_blob || !n_bytes || !n_hits) return TNT_E_ARG;
	int rc = collect_search(ctx, false, nullptr, nullptr);
	if (rc) return rc;
	const void *dev = nullptr;
	rc = tnt_search_hits_compact(ctx, &dev, n_bytes, n_hits);
	if (rc) return rc;
	HIP_TRY(ctx->pin_blob.ensure(*n_bytes + 8), ctx->err);
	HIP_TRY(hipMemcpy(ctx->pin_blob.p, dev, *n_bytes, hipMemcpyDeviceToHost), ctx->err);
	*host_blob = ctx->pin_blob.p;
	return TNT_OK;
}

int tnt_search_hits_device(tnt_ctx *ctx, const void **device_hits, size_t *n_hits)
{
	if (!ctx || !device_hits || !n_hits) return TNT_E_ARG;
	if (ctx->pending) {
		ctx->err = "tnt_search_hits_device: collect the search first";
		return TNT_E_ARG;
	}
	if (!ctx->dev_hits_valid) {
		ctx->err = "the hits of targets that were seeded in windows only exist in the host copy";
		return TNT_E_UNSUPPORTED;
	}
	*device_hits = ctx->hits.p;
	*n_hits = ctx->dev_hits_n;
	return TNT_OK;
}

// The compaction kernels behind the pair stage of a search, on the context's stream: the hit count is read from the device
// counter, the blob's byte count lands in the counter block (C_BLOB) that tnt_search_collect fetches anyway.  Sized for
// `cap_hits` records; a search that reports more is packed again by tnt_search_hits_compact.
static int enqueue_compaction(tnt_ctx *ctx, size_t cap_hits, const uint32_t *n_dev)
{
	const size_t worst = 8 + cap_hits * (size_t(HIT_POD_BYTES) + 8 + 3 * size_t(ALIGN_TEXT));
	if (ctx->hit_blob.n < worst) HIP_TRY(ctx->hit_blob.alloc(worst + worst / 4 + 4096), ctx->err);
	if (ctx->hit_sizes.n < cap_hits + 1) HIP_TRY(ctx->hit_sizes.alloc(cap_hits + cap_hits / 4 + 1024), ctx->err);
	HitPackArgs a;
	a.hits = ctx->hits.p;
	a.n = uint32_t(cap_hits);
	a.n_dev = n_dev;
	a.sizes = ctx->hit_sizes.p;
	a.blob = ctx->hit_blob.p;
	a.total = ctx->G() + C_BLOB;
	const int blocks = int(std::min<size_t>(2048, (cap_hits + 3) / 4 + 1)); // one wavefront per hit, four per workgroup
	hipLaunchKernelGGL(hit_size_kernel, dim3(blocks), dim3(256), 0, ctx->stream, a);
	hipLaunchKernelGGL(hit_scan_kernel, dim3(1), dim3(1024), 0, ctx->stream, a);
	hipLaunchKernelGGL(hit_pack_kernel, dim3(blocks), dim3(256), 0, ctx->stream, a);
	return TNT_OK;
}

int tnt_search_hits_compact(tnt_ctx *ctx, const void **device_blob, size_t *n_bytes, size_t *n_hits)
{
	if (!ctx || !device_blob || !n_bytes || !n_hits) return TNT_E_ARG;
	if (ctx->pending) {
		ctx->err = "tnt_search_hits_compact: collect the search first";
		return TNT_E_ARG;
	}
	if (!ctx->dev_hits_valid) {
		ctx->err = "the hits of targets that were seeded in windows only exist in the host copy";
		return TNT_E_UNSUPPORTED;
	}
	const size_t n = ctx->dev_hits_n;
	if (ctx->blob_hits == n && ctx->blob_bytes) { // packed behind the search already (tnt_search_enqueue)
		*device_blob = ctx->hit_blob.p;
		*n_bytes = ctx->blob_bytes;
		*n_hits = n;
		return TNT_OK;
	}
	HIP_TRY(hipSetDevice(ctx->device), ctx->err);
	const int rc = enqueue_compaction(ctx, n, nullptr);
	if (rc) return rc;
	uint32_t total[2] = {0, 0};
	HIP_TRY(hipMemcpyAsync(total, ctx->G() + C_BLOB, sizeof(total), hipMemcpyDeviceToHost, ctx->stream), ctx->err);
	HIP_TRY(hipStreamSynchronize(ctx->stream), ctx->err);
	HIP_TRY(hipGetLastError(), ctx->err);
	ctx->compact_in_search = true; // a caller that wants the compact form once wants it after every search
	ctx->blob_hits = n, ctx->blob_bytes = total[0];
	*device_blob = ctx->hit_blob.p;
	*n_bytes = total[0];
	*n_hits = n;
	return TNT_OK;
}

int tnt_hits_unpack(const void *blob, size_t n_bytes, tnt_hit *out, size_t cap, size_t *n_hits)
{
	if (!blob || n_bytes < 8 || !n_hits) return TNT_E_ARG;
	const uint8_t *p = static_cast<const uint8_t *>(blob);
	uint32_t n = 0, total = 0;
	memcpy(&n, p, 4);
	memcpy(&total, p + 4, 4);
	*n_hits = 0;
	// a truncated or corrupt blob (a peer rank's bytes) must not make the caller allocate for a count it cannot hold
	if (total < 8 || total > n_bytes || 8 + uint64_t(n) * (uint64_t(HIT_POD_BYTES) + 8) > uint64_t(total)) return TNT_E_ARG;
	*n_hits = n;
	if (!out) return TNT_OK; // a query for the count
	if (cap < n) return TNT_E_LIMIT;
	size_t at = 8;
	for (uint32_t i = 0; i < n; i++) {
		if (at + HIT_POD_BYTES + 8 > total) return TNT_E_ARG;
		tnt_hit &h = out[i];
		memcpy(&h, p + at, HIT_POD_BYTES);
		uint32_t l01 = 0, l2 = 0;
		memcpy(&l01, p + at + HIT_POD_BYTES, 4);
		memcpy(&l2, p + at + HIT_POD_BYTES + 4, 4);
		const uint32_t len[3] = {l01 & 0xFFFFu, l01 >> 16, l2 & 0xFFFFu};
		char *dst[3] = {h.forward_align, h.reverse_align, h.probe_align};
		size_t t = at + HIT_POD_BYTES + 8;
		for (int k = 0; k < 3; k++) {
			if (len[k] >= TNT_ALIGN_TEXT || t + len[k] > total) return TNT_E_ARG;
			memcpy(dst[k], p + t, len[k]);
			memset(dst[k] + len[k], 0, TNT_ALIGN_TEXT - len[k]);
			t += len[k];
		}
		at = (t + 3) & ~size_t(3);
	}
	return TNT_OK;
}


int tnt_device_count(void)
{
	int n = 0;
	if (hipGetDeviceCount(&n) != hipSuccess) return 0;
	return n;
}

// ------------------------------------------------------------------------------------------------ multi-GPU: the gang
// One process, one context per device, one host thread per context while they search (targets are independent; the
// caller deals its batches out).  What the reference's master collects from its MPI workers (packed hit lists,
// tntblast_master.cpp:811-828; the workers send theirs, tntblast_worker.cpp:602) is collected here from the devices: the
// compact blobs (tnt_search_hits_compact) travel to the first context's device over RCCL -- ncclSend / ncclRecv in one
// group, sized from the byte counts the ranks report; an all-reduce of the hit counts rides along as the check that every
// record arrived -- and reach the host in ONE copy.  RCCL is looked up at run time (dlopen): where it is missing, fails
// to initialise, or two contexts share a device (RCCL wants one rank per device), every device copies its own blob to
// the host instead ("host" transport) -- same bytes in the same order.
struct tnt_gang {
	std::vector<tnt_ctx *> ctx;
	std::string err, transport;
	void *lib = nullptr;
	std::vector<ncclComm_t> comms;
	decltype(&ncclCommInitAll) p_init = nullptr;
	decltype(&ncclCommDestroy) p_destroy = nullptr;
	decltype(&ncclGroupStart) p_group_start = nullptr;
	decltype(&ncclGroupEnd) p_group_end = nullptr;
	decltype(&ncclSend) p_send = nullptr;
	decltype(&ncclRecv) p_recv = nullptr;
	decltype(&ncclAllReduce) p_all_reduce = nullptr;
	decltype(&ncclGetErrorString) p_error = nullptr;
	uint8_t *recv = nullptr;             // on the first context's device
	size_t recv_cap = 0;
	std::vector<unsigned long long *> count_dev; // per rank: {its hit count, the sum over the ranks}
	PinnedBuf<uint8_t> host;
	~tnt_gang()
	{
		if (!comms.empty() && p_destroy)
			for (ncclComm_t c : comms) (void)p_destroy(c);
		for (size_t r = 0; r < count_dev.size(); r++)
			if (count_dev[r] && hipSetDevice(ctx[r]->device) == hipSuccess) (void)hipFree(count_dev[r]);
		if (recv && hipSetDevice(ctx[0]->device) == hipSuccess) (void)hipFree(recv);
		if (lib) dlclose(lib);
	}
};

int tnt_gang_create(tnt_ctx *const *ctx, int n, tnt_gang **out)
{
	if (!ctx || n < 1 || !out) return TNT_E_ARG;
	for (int r = 0; r < n; r++)
		if (!ctx[r]) return TNT_E_ARG;
	std::unique_ptr<tnt_gang> g(new tnt_gang);
	g->ctx.assign(ctx, ctx + n);
	g->transport = "host";
	bool distinct = true;
	for (int a = 0; a < n; a++)
		for (int b = a + 1; b < n; b++)
			if (ctx[a]->device == ctx[b]->device) distinct = false;
	const char *forced = getenv("TNT_MI355X_GATHER"); // "host": never RCCL; "rccl": fail instead of falling back
	const bool want_rccl = !(forced && !strcmp(forced, "host"));
	std::string why;
	if (!distinct) why = "two contexts share a device";
	if (want_rccl && distinct) {
		std::vector<std::string> names = {"librccl.so.1", "/opt/rocm/lib/librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so"};
		if (const char *rp = getenv("ROCM_PATH")) names.push_back(std::string(rp) + "/lib/librccl.so.1"), names.push_back(std::string(rp) + "/lib/librccl.so");
		if (const char *lp = getenv("TNT_MI355X_RCCL_LIB")) names.insert(names.begin(), lp);
		for (const std::string &name : names)
			if ((g->lib = dlopen(name.c_str(), RTLD_NOW | RTLD_LOCAL))) break;
		if (!g->lib) why = "librccl not found";
		if (g->lib) {
			g->p_init = reinterpret_cast<decltype(g->p_init)>(dlsym(g->lib, "ncclCommInitAll"));
			g->p_destroy = reinterpret_cast<decltype(g->p_destroy)>(dlsym(g->lib, "ncclCommDestroy"));
			g->p_group_start = reinterpret_cast<decltype(g->p_group_start)>(dlsym(g->lib, "ncclGroupStart"));
			g->p_group_end = reinterpret_cast<decltype(g->p_group_end)>(dlsym(g->lib, "ncclGroupEnd"));
			g->p_send = reinterpret_cast<decltype(g->p_send)>(dlsym(g->lib, "ncclSend"));
			g->p_recv = reinterpret_cast<decltype(g->p_recv)>(dlsym(g->lib, "ncclRecv"));
			g->p_all_reduce = reinterpret_cast<decltype(g->p_all_reduce)>(dlsym(g->lib, "ncclAllReduce"));
			g->p_error = reinterpret_cast<decltype(g->p_error)>(dlsym(g->lib, "ncclGetErrorString"));
			if (!g->p_init || !g->p_destroy || !g->p_group_start || !g->p_group_end || !g->p_send || !g->p_recv || !g->p_all_reduce || !g->p_error) {
				why = "librccl lacks an entry point";
			} else {
				std::vector<int> devs(size_t(n), 0);
				for (int r = 0; r < n; r++) devs[size_t(r)] = ctx[r]->device;
				g->comms.assign(size_t(n), nullptr);
				const ncclResult_t e = g->p_init(g->comms.data(), n, devs.data());
				if (e != ncclSuccess) {
					why = std::string("ncclCommInitAll: ") + g->p_error(e);
					g->comms.clear();
				} else {
					g->transport = "rccl";
					g->count_dev.assign(size_t(n), nullptr);
					for (int r = 0; r < n; r++) {
						if (hipSetDevice(ctx[r]->device) != hipSuccess || hipMalloc(reinterpret_cast<void **>(&g->count_dev[size_t(r)]), 2 * sizeof(unsigned long long)) != hipSuccess) {
							g_create_error = "tnt_gang_create: device allocation failed";
							return TNT_E_DEVICE;
						}
					}
				}
			}
		}
	}
	if (g->transport != "rccl" && forced && !strcmp(forced, "rccl")) {
		g_create_error = "tnt_gang_create: RCCL transport asked for (TNT_MI355X_GATHER=rccl) but not available: " + why;
		return TNT_E_DEVICE;
	}
	if (g->transport != "rccl" && !why.empty()) g->transport = "host (" + why + ")";
	*out = g.release();
	return TNT_OK;
}

void tnt_gang_destroy(tnt_gang *g) { delete g; }
const char *tnt_gang_transport(const tnt_gang *g) { return g ? g->transport.c_str() : ""; }
const char *tnt_gang_last_error(const tnt_gang *g) { return g ? g->err.c_str() : g_create_error.c_str(); }

int tnt_gang_gather(tnt_gang *g, const uint8_t *has, const void **blobs, size_t *n_bytes, size_t *n_hits)
{
	if (!g || !blobs || !n_bytes || !n_hits) return TNT_E_ARG;
	const size_t n = g->ctx.size();
	std::vector<const void *> dev(n, nullptr);
	std::vector<size_t> off(n + 1, 0);
	unsigned long long sum = 0;
	for (size_t r = 0; r < n; r++) {
		n_bytes[r] = n_hits[r] = 0;
		blobs[r] = nullptr;
		if (has && !has[r]) {
			off[r + 1] = off[r];
			continue;
		}
		const int rc = tnt_search_hits_compact(g->ctx[r], &dev[r], &n_bytes[r], &n_hits[r]); // (synchronises the context's stream)
		if (rc) {
			g->err = "rank " + std::to_string(r) + ": " + g->ctx[r]->err;
			return rc;
		}
		off[r + 1] = off[r] + ((n_bytes[r] + 15) & ~size_t(15));
		sum += n_hits[r];
	}
	const size_t total = off[n];
	HIP_TRY(g->host.ensure(total + 16), g->err);
	if (g->transport == "rccl") {
		tnt_ctx *c0 = g->ctx[0];
		HIP_TRY(hipSetDevice(c0->device), g->err);
		if (total > g->recv_cap) {
			if (g->recv) (void)hipFree(g->recv);
			g->recv = nullptr, g->recv_cap = 0;
			HIP_TRY(hipMalloc(reinterpret_cast<void **>(&g->recv), total + total / 2 + 4096), g->err);
			g->recv_cap = total + total / 2 + 4096;
		}
		for (size_t r = 0; r < n; r++) { // every rank's hit count next to the place of the sum
			const unsigned long long mine[2] = {(has && !has[r]) ? 0ull : (unsigned long long)n_hits[r], 0ull};
			HIP_TRY(hipSetDevice(g->ctx[r]->device), g->err);
			HIP_TRY(hipMemcpyAsync(g->count_dev[r], mine, sizeof(mine), hipMemcpyHostToDevice, g->ctx[r]->stream), g->err);
			HIP_TRY(hipStreamSynchronize(g->ctx[r]->stream), g->err); // (`mine` leaves scope)
		}
		HIP_TRY(hipSetDevice(c0->device), g->err);
		if (dev[0] && n_bytes[0]) HIP_TRY(hipMemcpyAsync(g->recv + off[0], dev[0], n_bytes[0], hipMemcpyDeviceToDevice, c0->stream), g->err);
		ncclResult_t e = g->p_group_start();
		for (size_t r = 0; r < n && e == ncclSuccess; r++) {
			HIP_TRY(hipSetDevice(g->ctx[r]->device), g->err);
			e = g->p_all_reduce(g->count_dev[r], g->count_dev[r] + 1, 1, ncclUint64, ncclSum, g->comms[r], g->ctx[r]->stream);
		}
		for (size_t r = 1; r < n && e == ncclSuccess; r++) {
			if (!dev[r] || !n_bytes[r]) continue;
			HIP_TRY(hipSetDevice(g->ctx[r]->device), g->err);
			e = g->p_send(dev[r], n_bytes[r], ncclUint8, 0, g->comms[r], g->ctx[r]->stream);
			if (e != ncclSuccess) break;
			HIP_TRY(hipSetDevice(c0->device), g->err);
			e = g->p_recv(g->recv + off[r], n_bytes[r], ncclUint8, int(r), g->comms[0], c0->stream);
		}
		const ncclResult_t e2 = g->p_group_end();
		if (e == ncclSuccess) e = e2;
		if (e != ncclSuccess) {
			g->err = std::string("RCCL gather: ") + g->p_error(e);
			return TNT_E_DEVICE;
		}
		for (size_t r = 1; r < n; r++) {
			HIP_TRY(hipSetDevice(g->ctx[r]->device), g->err);
			HIP_TRY(hipStreamSynchronize(g->ctx[r]->stream), g->err);
		}
		HIP_TRY(hipSetDevice(c0->device), g->err);
		unsigned long long seen = 0;
		if (total) HIP_TRY(hipMemcpyAsync(g->host.p, g->recv, total, hipMemcpyDeviceToHost, c0->stream), g->err);
		HIP_TRY(hipMemcpyAsync(&seen, g->count_dev[0] + 1, sizeof(seen), hipMemcpyDeviceToHost, c0->stream), g->err);
		HIP_TRY(hipStreamSynchronize(c0->stream), g->err);
		if (seen != sum) {
			g->err = "RCCL gather: the ranks' hit counts add up to " + std::to_string(seen) + ", the blobs hold " + std::to_string(sum);
			return TNT_E_DEVICE;
		}
	} else {
		for (size_t r = 0; r < n; r++) {
			if (!dev[r] || !n_bytes[r]) continue;
			HIP_TRY(hipSetDevice(g->ctx[r]->device), g->err);
			HIP_TRY(hipMemcpy(g->host.p + off[r], dev[r], n_bytes[r], hipMemcpyDeviceToHost), g->err);
		}
	}
	for (size_t r = 0; r < n; r++)
		if (!(has && !has[r])) blobs[r] = g->host.p + off[r];
	return TNT_OK;
}

int tnt_search(tnt_ctx *ctx, tnt_db *db, const tnt_hit **hits, size_t *n_hits)
{
	const bool timing = getenv("TNT_MI355X_LONG_TIMING") != nullptr;
	const double t0 = timing ? now_ms() : 0.0;
	int rc = tnt_search_enqueue(ctx, db);
	if (rc) return rc;
	const double t1 = timing ? now_ms() : 0.0;
	rc = tnt_search_collect(ctx, hits, n_hits);
	if (timing) fprintf(stderr, "tnt_search: enqueue %.2f ms, collect %.2f ms\n", t1 - t0, now_ms() - t1);
	return rc;
}

int tnt_last_timing(tnt_ctx *ctx, tnt_timing *out)
{
	if (!ctx || !out) return TNT_E_ARG;
	*out = ctx->timing;
	return TNT_OK;
}

int tnt_seeds(tnt_ctx *ctx, tnt_db *db, const tnt_seed **seeds, size_t *n_seeds)
{
	if (!ctx || !db || db->ctx != ctx || !seeds || !n_seeds) return TNT_E_ARG;
	HIP_TRY(hipSetDevice(ctx->device), ctx->err);
	int rc = ensure_work_buffers(ctx, ctx->lmax, db);
	if (rc) return rc;
	size_t tpb = 1;
	rc = plan(ctx, db, tpb);
	if (rc) return rc;
	const double words = double(size_t(1) << (2 * ctx->opt.hash_word_size));
	const size_t cap = size_t(double(db->n_bases) * double(ctx->n_entries) / words * 2.0) + (1u << 20);
	HIP_TRY(ctx->seeds_out.alloc(cap), ctx->err);
	HIP_TRY(hipMemsetAsync(ctx->S().counters.p, 0, sizeof(uint32_t) * C_COUNT, ctx->stream), ctx->err);
	rc = enqueue_batch(ctx, db, 0, db->h_unit_target.size(), true);
	if (rc) return rc;
	uint32_t cnt[C_COUNT];
	HIP_TRY(hipMemcpyAsync(cnt, ctx->S().counters.p, sizeof(cnt), hipMemcpyDeviceToHost, ctx->stream), ctx->err);
	HIP_TRY(hipStreamSynchronize(ctx->stream), ctx->err);
	if (cnt[C_ERR]) {
		ctx->err = err_text(cnt[C_ERR]);
		return TNT_E_LIMIT;
	}
	if (cnt[C_SEEDS] > cap) {
		ctx->err = "seed output overflow";
		return TNT_E_LIMIT;
	}
	std::vector<SeedOut> raw(cnt[C_SEEDS]);
	if (!raw.empty()) HIP_TRY(hipMemcpy(raw.data(), ctx->seeds_out.p, raw.size() * sizeof(SeedOut), hipMemcpyDeviceToHost), ctx->err);
	ctx->seeds_out.release();
	// (assay, role) -> unique oligo; shared oligos yield the same seed several times
	std::vector<tnt_seed> out;
	out.reserve(raw.size());
	for (const SeedOut &s : raw) {
		const int slot = int(s.assay_role >> 2) * 3 + int(s.assay_role & 3u);
		tnt_seed o;
		o.seq_index = int(db->first_seq_index + s.target);
		o.oligo = ctx->h_slots[slot].oligo;
		o.strand = s.strand_minus ? TNT_MINUS : TNT_PLUS;
		o.query_loc = int(s.query_loc);
		o.target_loc = int(s.target_loc);
		out.push_back(o);
	}
	auto key = [](const tnt_seed &a, const tnt_seed &b) {
		if (a.seq_index != b.seq_index) return a.seq_index < b.seq_index;
		if (a.oligo != b.oligo) return a.oligo < b.oligo;
		if (a.strand != b.strand) return a.strand > b.strand; // minus first
		const int da = a.query_loc - a.target_loc, dbb = b.query_loc - b.target_loc;
		if (da != dbb) return da < dbb;
		return a.query_loc < b.query_loc; // (windows of a long target report the seeds of their overlap twice)
	};
	std::sort(out.begin(), out.end(), key);
	out.erase(std::unique(out.begin(), out.end(),
	                      [](const tnt_seed &a, const tnt_seed &b) {
		                      return a.seq_index == b.seq_index && a.oligo == b.oligo && a.strand == b.strand &&
		                             a.query_loc == b.query_loc && a.target_loc == b.target_loc;
	                      }),
	          out.end());
	ctx->h_seeds.swap(out);
	*seeds = ctx->h_seeds.data();
	*n_seeds = ctx->h_seeds.size();
	return TNT_OK;
}

int tnt_first_cull(tnt_ctx *ctx, tnt_db *db, const int32_t **rows, size_t *n_rows)
{
	if (!ctx || !db || db->ctx != ctx || !rows || !n_rows) return TNT_E_ARG;
	if (ctx->pending) {
		ctx->err = "a search is pending on this context: collect it first";
		return TNT_E_ARG;
	}
	HIP_TRY(hipSetDevice(ctx->device), ctx->err);
	int rc = ensure_work_buffers(ctx, ctx->lmax, db);
	if (rc) return rc;
	for (hipEvent_t e : ctx->events) (void)hipEventDestroy(e);
	ctx->events.clear();
	ctx->dev_hits_valid = false;
	HIP_TRY(hipMemsetAsync(ctx->S().counters.p, 0, sizeof(uint32_t) * C_COUNT, ctx->stream), ctx->err);
	size_t tpb = 1;
	rc = plan(ctx, db, tpb);
	if (rc) return rc;
	ctx->h_cull_rows.clear();
	std::vector<Item> items;
	std::vector<Group> groups;
	const size_t n_units = db->n_short_units, item_region = ctx->item_cap / N_SHARD, group_region = ctx->group_cap / N_SHARD;
	for (size_t u0 = 0; u0 < n_units; u0 += tpb) {
		// a whole batch of the search (the stages behind the seed kernel run too; their results are not looked at)
		rc = enqueue_batch(ctx, db, u0, std::min(n_units, u0 + tpb), false);
		if (rc) return rc;
		uint32_t cnt[C_COUNT];
		HIP_TRY(hipMemcpyAsync(cnt, ctx->S().counters.p, sizeof(cnt), hipMemcpyDeviceToHost, ctx->stream), ctx->err);
		HIP_TRY(hipStreamSynchronize(ctx->stream), ctx->err);
		const uint32_t err = cnt[C_ERR] & ~(ERR_HIT_OVERFLOW | ERR_POOL_OVERFLOW); // (the stages behind the seed kernel are not looked at)
		if (err) {
			ctx->err = err_text(err);
			return TNT_E_LIMIT;
		}
		// where the seed kernel left its items and groups: 32 shards (round-2 kernel), or one region per wavefront
		// (the batch says which kernel filled it: a database of 2^32 packed words or more goes through the round-2 kernel also
		// in a context that otherwise uses seed2_kernel)
		std::vector<uint32_t> region_n;
		uint32_t n_regions = N_SHARD;
		size_t ir = item_region, gr = group_region;
		const bool by_wavefront = ctx->batch_direct && ctx->batch_regions != 0;
		if (by_wavefront) {
			n_regions = ctx->batch_regions;
			ir = ctx->item_cap / n_regions, gr = ctx->group_cap / n_regions;
			region_n.resize(2 * size_t(n_regions));
			HIP_TRY(hipMemcpy(region_n.data(), ctx->S().region_counts.p, region_n.size() * sizeof(uint32_t), hipMemcpyDeviceToHost), ctx->err);
		}
		for (uint32_t r = 0; r < n_regions; r++) {
			const uint32_t ni = by_wavefront ? region_n[2 * r] : cnt[C_ITEMS + r * SHARD_STRIDE], ng = by_wavefront ? region_n[2 * r + 1] : cnt[C_GROUPS + r * SHARD_STRIDE];
			if (!ni) continue;
			const size_t item_region = ir, group_region = gr;
			items.resize(ni);
			groups.resize(ng);
			HIP_TRY(hipMemcpy(items.data(), ctx->S().items.p + r * item_region, ni * sizeof(Item), hipMemcpyDeviceToHost), ctx->err);
			HIP_TRY(hipMemcpy(groups.data(), ctx->S().groups.p + r * group_region, ng * sizeof(Group), hipMemcpyDeviceToHost), ctx->err);
			for (const Item &it : items) {
				const uint32_t slot = it.info >> ITEM_SLOT_SHIFT, assay = slot / 3u;
				if (ctx->h_assays[assay].kind != 0) continue; // probe / padlock lists are not culled (every seed is an item)
				const Group &g = groups[it.group - r * group_region];
				const int32_t row[7] = {int32_t(db->first_seq_index + it.target), int32_t(assay), int32_t(slot % 3u),
				                        (it.info & ITEM_MINUS) ? TNT_MINUS : TNT_PLUS, int32_t(it.info & 127u), int32_t(it.tloc),
				                        (g.assay & GROUP_MINUS_FIRST) ? 1 : 0};
				ctx->h_cull_rows.insert(ctx->h_cull_rows.end(), row, row + 7);
			}
		}
	}
	if (!db->h_long_targets.empty()) { // targets seeded in windows: their lists are assembled and culled whole (tnt_sort.hip)
		std::vector<tnt_hit> none;
		rc = search_long_targets(ctx, db, none, &ctx->h_cull_rows);
		if (rc) return rc;
	}
	*rows = ctx->h_cull_rows.data();
	*n_rows = ctx->h_cull_rows.size() / 7;
	return TNT_OK;
}

int tnt_seed_scan_only(tnt_ctx *ctx, tnt_db *db, int repeats, float *ms_per_pass, uint64_t *n_candidates)
{
	if (!ctx || !db || db->ctx != ctx || !ms_per_pass || !n_candidates || repeats < 1) return TNT_E_ARG;
	if (ctx->pending) {
		ctx->err = "a search is pending on this context: collect it first";
		return TNT_E_ARG;
	}
	if (db->n_short_units == 0 || !db->unit_rec.p) {
		ctx->err = "tnt_seed_scan_only: the database has no whole-target work units";
		return TNT_E_UNSUPPORTED;
	}
	HIP_TRY(hipSetDevice(ctx->device), ctx->err);
	int rc = ensure_work_buffers(ctx, ctx->lmax, db);
	if (rc) return rc;
	size_t tpb = 1;
	rc = plan(ctx, db, tpb); // the chunk tables of the context
	if (rc) return rc;
	DevBuf<unsigned long long> count;
	HIP_TRY(count.alloc(1), ctx->err);
	hipEvent_t e0 = nullptr, e1 = nullptr;
	HIP_TRY(hipEventCreate(&e0), ctx->err);
	HIP_TRY(hipEventCreate(&e1), ctx->err);
	ScanOnlyArgs a;
	memset(&a, 0, sizeof(a));
	a.db = make_view(db, 0, db->n_targets);
	a.unit_rec = db->unit_rec.p;
	a.unit0 = 0, a.unit1 = uint32_t(db->n_short_units);
	a.word_size = ctx->opt.hash_word_size;
	a.n_candidates = count.p;
	const size_t n_words = size_t(1) << (2 * ctx->opt.hash_word_size);
	const size_t off_stride = (n_words + 1 + 1) & ~size_t(1);
	const int blocks = ctx->n_cu * 2; // two 1024-thread workgroups per CU (32 KB of LDS each at k = 7)
	float total = 0.0f;
	for (int r = 0; r <= repeats; r++) { // (pass 0 warms up and is not timed)
		HIP_TRY(hipMemsetAsync(count.p, 0, sizeof(unsigned long long), ctx->stream), ctx->err);
		HIP_TRY(hipEventRecord(e0, ctx->stream), ctx->err);
		for (size_t ch = 0; ch + 1 < ctx->chunk_begin.size(); ch++) {
			a.wt_off = ctx->wt_off.p + ch * off_stride;
			hipLaunchKernelGGL(scan_only_kernel, dim3(blocks), dim3(SCAN_THREADS), n_words * sizeof(uint16_t), ctx->stream, a);
		}
		HIP_TRY(hipEventRecord(e1, ctx->stream), ctx->err);
		HIP_TRY(hipStreamSynchronize(ctx->stream), ctx->err);
		float ms = 0.0f;
		HIP_TRY(hipEventElapsedTime(&ms, e0, e1), ctx->err);
		if (r) total += ms;
	}
	(void)hipEventDestroy(e0);
	(void)hipEventDestroy(e1);
	HIP_TRY(hipGetLastError(), ctx->err);
	unsigned long long n = 0;
	HIP_TRY(hipMemcpy(&n, count.p, sizeof(n), hipMemcpyDeviceToHost), ctx->err);
	*ms_per_pass = total / float(repeats);
	*n_candidates = n;
	return TNT_OK;
}

int tnt_bind_batch(tnt_ctx *ctx, tnt_db *db, const tnt_bind_req *req, size_t n, tnt_bind *out)
{
	return tnt_bind_batch_routed(ctx, db, req, n, out, TNT_ROUTE_UNIT);
}

int tnt_bind_batch_routed(tnt_ctx *ctx, tnt_db *db, const tnt_bind_req *req, size_t n, tnt_bind *out, int route)
{
	if (!ctx || !db || db->ctx != ctx || (!req && n) || (!out && n)) return TNT_E_ARG;
	if (route != TNT_ROUTE_UNIT && route != TNT_ROUTE_BUCKETS && route != TNT_ROUTE_MERGED && route != TNT_ROUTE_WAVE && route != TNT_ROUTE_WAVE_UNIT) return TNT_E_ARG;
	if ((route == TNT_ROUTE_WAVE || route == TNT_ROUTE_WAVE_UNIT) && ctx->opt.use_dinkelbach) {
		ctx->err = "the wavefront-per-window kernel has no Dinkelbach passes";
		return TNT_E_UNSUPPORTED;
	}
	if (ctx->pending) {
		ctx->err = "a search is pending on this context: collect it first";
		return TNT_E_ARG;
	}
	if (n == 0) return TNT_OK;
	if (n > ITEM_CAP_MAX || n > (size_t(1) << (32 - ITEM_SLOT_SHIFT))) return TNT_E_LIMIT;
	HIP_TRY(hipSetDevice(ctx->device), ctx->err);
	int rc = ensure_work_buffers(ctx, ctx->unit_rows, nullptr, n); // the unit-level kernel instance is bind_kernel<64 or 128, true>
	if (rc) return rc;
	auto bucket_of = [&](int id) -> int32_t { // position of an instance in the context's bucket list (it holds the instances of every oligo)
		for (size_t i = 0; i < ctx->bucket_lmax.size(); i++)
			if (ctx->bucket_lmax[i] == id) return int32_t(i);
		return int32_t(ctx->bucket_lmax.size()) - 1;
	};
	// one temporary slot per request (arbitrary strand concentration)
	std::vector<Slot> slots(n);
	std::vector<Item> items(n);
	for (size_t i = 0; i < n; i++) {
		const tnt_bind_req &r = req[i];
		const size_t t = size_t(r.seq_index) - db->first_seq_index;
		if (t >= db->n_targets || r.oligo < 0 || r.oligo >= int(ctx->oligos.size()) || r.query_loc < 0 || r.query_loc >= TNT_MAX_OLIGO ||
		    size_t(r.query_loc) >= ctx->oligos[size_t(r.oligo)].size() || r.target_loc < 0 || !(r.ct > 0.0f)) {
			ctx->err = "bad bind request";
			return TNT_E_ARG;
		}
		slots[i].oligo = r.oligo;
		slots[i].log_ct = logf((r.ct - 0.5f * 0.0f) * 1.0f);
		slots[i].thresh = r.role == 1 ? 1 : 0;
		slots[i].assay = -1;
		slots[i].bucket = bucket_of(pick_instance(ctx->oligos[r.oligo].size()));
		slots[i].bucket_general = bucket_of(pick_lmax(ctx->oligos[r.oligo].size()));
		items[i].target = uint32_t(t);
		items[i].tloc = uint32_t(r.target_loc);
		slots[i].qloc_bias = r.query_loc & ~127; // (a work item holds seven bits of the word index)
		items[i].info = uint32_t(r.query_loc & 127) | (r.strand == TNT_MINUS ? ITEM_MINUS : 0u) | (uint32_t(i) << ITEM_SLOT_SHIFT);
		items[i].group = 0;
	}
	DevBuf<Slot> dslots;
	DevBuf<BindFull> dfull;
	HIP_TRY(dslots.upload(slots), ctx->err);
	HIP_TRY(dfull.alloc(n), ctx->err);
	HIP_TRY(hipMemcpy(ctx->S().items.p, items.data(), n * sizeof(Item), hipMemcpyHostToDevice), ctx->err);
	HIP_TRY(hipMemsetAsync(ctx->S().counters.p, 0, sizeof(uint32_t) * C_COUNT, ctx->stream), ctx->err);
	BindArgs b = bind_args(ctx, make_view(db, 0, db->n_targets));
	b.slots = dslots.p;
	b.groups = nullptr;
	b.n_items = uint32_t(n);
	b.n_items_dev = nullptr;
	if (route == TNT_ROUTE_UNIT) {
		b.full = dfull.p;
		launch_bind(ctx, b, int(std::min<size_t>(ctx->bind_blocks, (n + size_t(bind_threads(0)) - 1) / size_t(bind_threads(0)))), 0);
	} else if (route == TNT_ROUTE_WAVE_UNIT) {
		b.full = dfull.p;
		launch_wbind(ctx, b, ctx->unit_rows, true, ctx->stream, n);
	} else {
		// the path tnt_search takes: the queue is split by bucket_kernel and every request is bound by the throughput
		// instance of its oligo length (exact / plain ladder / general for windows with degenerate bases)
		std::vector<uint32_t> queue(n);
		for (size_t i = 0; i < n; i++) queue[i] = uint32_t(i);
		const uint32_t n32 = uint32_t(n);
		HIP_TRY(hipStreamSynchronize(ctx->stream), ctx->err); // the counters were cleared on the stream
		HIP_TRY(hipMemcpy(ctx->S().qb.p, queue.data(), n * sizeof(uint32_t), hipMemcpyHostToDevice), ctx->err);
		HIP_TRY(hipMemcpy(ctx->S().counters.p + C_QB, &n32, sizeof(uint32_t), hipMemcpyHostToDevice), ctx->err);
		ctx->filter_status_exact = true; // this caller reports the status of every request
		const int wave_from = ctx->wave_from;
		if (route == TNT_ROUTE_WAVE) ctx->wave_from = 1; // every length bucket through the wavefront-per-window kernel
		launch_bind_phase(ctx, b, b.db, db->n_exceptions != 0, ctx->S().qb.p, ctx->S().counters.p + C_QB, nullptr, route == TNT_ROUTE_MERGED ? 2 : 1);
		ctx->wave_from = wave_from;
		ctx->filter_status_exact = false;
	}
	HIP_TRY(hipGetLastError(), ctx->err);
	uint32_t cnt[C_COUNT];
	HIP_TRY(hipMemcpyAsync(cnt, ctx->S().counters.p, sizeof(cnt), hipMemcpyDeviceToHost, ctx->stream), ctx->err);
	HIP_TRY(hipStreamSynchronize(ctx->stream), ctx->err);
	if (cnt[C_ERR]) {
		ctx->err = err_text(cnt[C_ERR]);
		return TNT_E_LIMIT;
	}
	rc = fetch_long_texts(ctx, cnt[C_LTEXT]); // (tnt_bind::align of a long alignment refers to them: tnt_long_text)
	if (rc) return rc;
	if (route != TNT_ROUTE_UNIT && route != TNT_ROUTE_WAVE_UNIT) {
		// what the throughput instances leave behind: a status per request and, for bound ones, the pool record
		std::vector<int32_t> res(n);
		HIP_TRY(hipMemcpy(res.data(), ctx->S().result.p, n * sizeof(int32_t), hipMemcpyDeviceToHost), ctx->err);
		const size_t n_pool = std::min<size_t>(cnt[C_POOL], POOL_CAP);
		std::vector<BoundRec> recs(n_pool);
		if (n_pool) HIP_TRY(hipMemcpy(recs.data(), ctx->S().pool.p, n_pool * sizeof(BoundRec), hipMemcpyDeviceToHost), ctx->err);
		for (size_t i = 0; i < n; i++) {
			tnt_bind &o = out[i];
			memset(&o, 0, sizeof(o));
			o.q_first = o.q_last = o.t_first = o.t_last = o.poly_degen = o.dp_max = o.n_max = -1; // not kept by these instances
			if (res[i] < 0) {
				o.status = -res[i];
				continue;
			}
			if (size_t(res[i]) >= n_pool) {
				ctx->err = "bind result points outside the pool";
				return TNT_E_LIMIT;
			}
			const BoundRec &r = recs[size_t(res[i])];
			o.status = 0, o.valid = 1;
			o.tm = r.tm, o.dH = r.dH, o.dS = r.dS;
			o.dg = r.dH - ctx->opt.target_t * r.dS; // delta_G(), nuc_cruc.h:1217
			o.anchor5 = r.a5, o.anchor3 = r.a3, o.num_mm = r.mm, o.num_gap = r.gap;
			o.loc5 = r.loc5, o.loc3 = r.loc3;
			memcpy(o.align, r.text, TNT_ALIGN_TEXT);
			o.align[TNT_ALIGN_TEXT - 1] = 0;
		}
		return TNT_OK;
	}
	std::vector<BindFull> full(n);
	HIP_TRY(hipMemcpy(full.data(), dfull.p, n * sizeof(BindFull), hipMemcpyDeviceToHost), ctx->err);
	for (size_t i = 0; i < n; i++) {
		const BindFull &f = full[i];
		tnt_bind &o = out[i];
		memset(&o, 0, sizeof(o));
		o.status = f.status, o.target_start = f.target_start, o.target_stop = f.target_stop;
		o.tm = f.tm, o.dH = f.dH, o.dS = f.dS, o.dg = f.dg;
		o.anchor5 = f.a5, o.anchor3 = f.a3, o.num_mm = f.mm, o.num_gap = f.gap, o.poly_degen = f.poly;
		o.loc5 = f.loc5, o.loc3 = f.loc3;
		o.q_first = f.q_first, o.q_last = f.q_last, o.t_first = f.t_first, o.t_last = f.t_last;
		o.valid = f.valid, o.dp_max = f.dp_max, o.n_max = f.n_max;
		memcpy(o.align, f.text, TNT_ALIGN_TEXT);
		o.align[TNT_ALIGN_TEXT - 1] = 0;
	}
	return TNT_OK;
}

int tnt_filter_bound(tnt_ctx *ctx, tnt_db *db, const tnt_bind_req *req, size_t n, int32_t *score)
{
	if (!ctx || !db || db->ctx != ctx || (!req && n) || (!score && n)) return TNT_E_ARG;
	if (ctx->pending) {
		ctx->err = "a search is pending on this context: collect it first";
		return TNT_E_ARG;
	}
	if (n == 0) return TNT_OK;
	if (n > ITEM_CAP_MAX || n > (size_t(1) << (32 - ITEM_SLOT_SHIFT))) return TNT_E_LIMIT;
	HIP_TRY(hipSetDevice(ctx->device), ctx->err);
	int rc = ensure_work_buffers(ctx, 64, nullptr, n);
	if (rc) return rc;
	if (!db->host_copy) {
		ctx->err = "tnt_filter_bound needs a database packed from codes (tnt_db_create)";
		return TNT_E_UNSUPPORTED;
	}
	std::vector<Slot> slots(n);
	std::vector<Item> items(n);
	std::map<int, std::vector<uint32_t>> by_rows; // kernel instance (rows) -> requests
	std::vector<uint8_t> plain(n, 1);
	for (size_t i = 0; i < n; i++) {
		const tnt_bind_req &r = req[i];
		const size_t t = size_t(r.seq_index) - db->first_seq_index;
		if (t >= db->n_targets || r.oligo < 0 || r.oligo >= int(ctx->oligos.size()) || r.query_loc < 0 ||
		    size_t(r.query_loc) >= ctx->oligos[size_t(r.oligo)].size() || r.target_loc < 0 || !(r.ct > 0.0f)) {
			ctx->err = "bad bind request";
			return TNT_E_ARG;
		}
		const int L = int(ctx->oligos[r.oligo].size());
		// the pre-filter is a lane-per-window kernel: oligos of up to MAX_LANE_OLIGO bases, whose word index fits the seven bits
		// of a work item (longer oligos are bound by the wavefront kernel, which has no pre-filter in front of it)
		if (L > MAX_LANE_OLIGO || r.query_loc > 127) {
			ctx->err = "tnt_filter_bound: the pre-filter serves oligos of up to 128 bases";
			return TNT_E_LIMIT;
		}
		slots[i] = Slot{r.oligo, logf((r.ct - 0.5f * 0.0f) * 1.0f), r.role == 1 ? 1 : 0, -1, 0, 0};
		items[i] = Item{uint32_t(t), uint32_t(r.target_loc), uint32_t(r.query_loc) | (r.strand == TNT_MINUS ? ITEM_MINUS : 0u) | (uint32_t(i) << ITEM_SLOT_SHIFT), 0};
		// windows with a non-ACGT code meet the filter instance of the general buckets
		const int start = std::max(0, r.target_loc - (r.query_loc + FLANK));
		const uint32_t stop = std::min<uint32_t>(uint32_t(start + L + 2 * FLANK), db->h_len[t]);
		const uint32_t *eb = db->h_exc_pos.data() + db->h_exc_off[t], *ee = db->h_exc_pos.data() + db->h_exc_off[t + 1];
		const uint32_t *it = std::lower_bound(eb, ee, uint32_t(start));
		if (it != ee && *it < stop) plain[i] = 0, by_rows[1000 + (pick_lmax(size_t(L)) & 0xFF)].push_back(uint32_t(i)); // the instance of the general buckets
		else by_rows[pick_instance(size_t(L)) & 0xFF].push_back(uint32_t(i));
	}
	DevBuf<Slot> dslots;
	DevBuf<int32_t> dscore;
	DevBuf<uint32_t> dindex;
	HIP_TRY(dslots.upload(slots), ctx->err);
	HIP_TRY(dscore.alloc(2 * n), ctx->err);
	HIP_TRY(hipStreamSynchronize(ctx->stream), ctx->err);
	HIP_TRY(hipMemcpy(ctx->S().items.p, items.data(), n * sizeof(Item), hipMemcpyHostToDevice), ctx->err);
	HIP_TRY(hipMemset(dscore.p, 0xFF, 2 * n * sizeof(int32_t)), ctx->err);
	std::vector<uint32_t> order;
	for (const auto &kv : by_rows) order.insert(order.end(), kv.second.begin(), kv.second.end());
	HIP_TRY(dindex.upload(order), ctx->err);
	FilterArgs2 f;
	memset(&f, 0, sizeof(f));
	f.tab = ctx->ftab.p;
	f.db = make_view(db, 0, db->n_targets);
	f.olig_q4 = ctx->olig_q4.p;
	f.olig_len = ctx->olig_len.p;
	f.slots = dslots.p;
	f.thresh = ctx->thresh.p;
	f.items = ctx->S().items.p;
	f.result = ctx->S().result.p;
	f.score_out = dscore.p;
	size_t at = 0;
	// plain windows of oligos of up to 64 bases go the way a search sends them: regrouped by (oligo, class), then the pre-filter
	// on the regrouped queue (tnt_ufilter.hip.h; TNT_MI355X_NO_REGROUP=1: filter_kernel on the queue as it is)
	std::deque<DevBuf<uint4>> keep_fat;     // (a deque: growing it leaves the buffers where they are)
	std::deque<DevBuf<uint32_t>> keep_words;
	std::deque<DevBuf<RegroupBucket>> keep_desc;
	for (const auto &kv : by_rows) {
		const int rows = kv.first % 1000;
		const bool degen = kv.first >= 1000;
		bool regrouped = false;
		if (ctx->regroup && has_ufilter_instance(rows, degen) && db->h_word_off[db->n_targets] < (uint64_t(1) << 32)) {
			std::map<std::pair<int32_t, int32_t>, uint32_t> bins;
			std::vector<uint32_t> slot_bin(2 * n, 0u);
			std::vector<uint4> fat(kv.second.size());
			for (size_t x = 0; x < kv.second.size(); x++) {
				const uint32_t i = kv.second[x];
				const tnt_bind_req &r = req[i];
				const size_t t = size_t(r.seq_index) - db->first_seq_index;
				const int L = int(ctx->oligos[r.oligo].size());
				const int start = std::max(0, r.target_loc - (r.query_loc + FLANK));
				const uint32_t stop = std::min<uint32_t>(uint32_t(start + L + 2 * FLANK), db->h_len[t]);
				slot_bin[2 * i] = slot_bin[2 * i + 1] = bins.emplace(std::make_pair(int32_t(r.oligo), int32_t(slots[i].thresh)), uint32_t(bins.size())).first->second;
				fat[x] = make_uint4(i, uint32_t(db->h_word_off[t] + uint64_t(start >> 4)), uint32_t(start & 15) | ((stop - uint32_t(start)) << 4) | (uint32_t(L) << 12), items[i].info);
			}
			if (bins.size() <= size_t(RG_MAX_BINS)) {
				const uint32_t n_bins = uint32_t(bins.size());
				keep_fat.emplace_back(), keep_fat.emplace_back();
				keep_words.emplace_back(), keep_words.emplace_back();
				keep_desc.emplace_back();
				DevBuf<uint4> &dfat = keep_fat[keep_fat.size() - 2], &dsorted = keep_fat.back();
				DevBuf<uint32_t> &dbin = keep_words[keep_words.size() - 2], &dwork = keep_words.back();
				HIP_TRY(dfat.upload(fat), ctx->err);
				HIP_TRY(dsorted.alloc(fat.size() + size_t(UF_BLOCK) * n_bins), ctx->err);
				HIP_TRY(dbin.upload(slot_bin), ctx->err);
				std::vector<uint32_t> work(8 + 2 * size_t(n_bins), 0u); // a counter line ([0] entries, [4] regrouped entries, [5] work counter); histogram; cursors
				work[0] = uint32_t(fat.size());
				HIP_TRY(dwork.upload(work), ctx->err);
				RegroupBucket B;
				memset(&B, 0, sizeof(B));
				B.src_cap = uint32_t(fat.size()), B.n_bins = n_bins, B.general = degen ? 1u : 0u;
				HIP_TRY(keep_desc.back().upload(std::vector<RegroupBucket>(1, B)), ctx->err);
				launch_regroup(ctx, keep_desc.back().p, 1, dfat.p, dsorted.p, dwork.p, dbin.p, dwork.p + 8, dwork.p + 8 + n_bins, ctx->stream);
				UFilterArgs u;
				memset(&u, 0, sizeof(u));
				u.tab = f.tab, u.packed = f.db.packed, u.olig_q4 = f.olig_q4, u.olig_len = f.olig_len, u.slots = f.slots;
				u.db = f.db, u.items = f.items;
				u.score_out = f.score_out;
				u.fat = dsorted.p, u.n_items_dev = dwork.p + 4, u.n_items = uint32_t(fat.size()) + uint32_t(UF_BLOCK) * n_bins, u.next_block = dwork.p + 5;
				launch_ufilter(u, ctx->n_cu * 8, rows, ctx->stream, degen);
				regrouped = true;
			}
		}
		if (!regrouped) {
			f.index = dindex.p + at;
			f.n_items = uint32_t(kv.second.size());
			launch_filter(f, int(std::min<size_t>(size_t(ctx->n_cu) * 8, (kv.second.size() + 255) / 256)), rows, ctx->stream, degen);
		}
		at += kv.second.size();
	}
	HIP_TRY(hipGetLastError(), ctx->err);
	HIP_TRY(hipStreamSynchronize(ctx->stream), ctx->err);
	HIP_TRY(hipMemcpy(score, dscore.p, 2 * n * sizeof(int32_t), hipMemcpyDeviceToHost), ctx->err);
	return TNT_OK;
}

int tnt_filter_bound_tier2(tnt_ctx *ctx, tnt_db *db, const tnt_bind_req *req, size_t n, int32_t *score)
{
	if (!ctx || !db || db->ctx != ctx || (!req && n) || (!score && n)) return TNT_E_ARG;
	if (ctx->pending) {
		ctx->err = "a search is pending on this context: collect it first";
		return TNT_E_ARG;
	}
	if (n == 0) return TNT_OK;
	if (n > ITEM_CAP_MAX || n > (size_t(1) << (32 - ITEM_SLOT_SHIFT))) return TNT_E_LIMIT;
	HIP_TRY(hipSetDevice(ctx->device), ctx->err);
	int rc = ensure_work_buffers(ctx, 64, nullptr, n);
	if (rc) return rc;
	if (!db->host_copy) {
		ctx->err = "tnt_filter_bound_tier2 needs a database packed from codes (tnt_db_create)";
		return TNT_E_UNSUPPORTED;
	}
	std::vector<Slot> slots(n);
	std::vector<Item> items(n);
	std::map<std::tuple<int, int, int>, std::vector<uint32_t>> by_group; // (instance rows, oligo, threshold class) -> requests: the kernel takes one oligo per wavefront
	std::map<int, std::vector<uint32_t>> by_rows_degen; // general instance (rows) -> requests
	for (size_t i = 0; i < n; i++) {
		const tnt_bind_req &r = req[i];
		const size_t t = size_t(r.seq_index) - db->first_seq_index;
		if (t >= db->n_targets || r.oligo < 0 || r.oligo >= int(ctx->oligos.size()) || r.query_loc < 0 || r.query_loc > 127 ||
		    size_t(r.query_loc) >= ctx->oligos[size_t(r.oligo)].size() || r.target_loc < 0 || !(r.ct > 0.0f)) {
			ctx->err = "bad bind request";
			return TNT_E_ARG;
		}
		const int L = int(ctx->oligos[r.oligo].size());
		slots[i] = Slot{r.oligo, logf((r.ct - 0.5f * 0.0f) * 1.0f), r.role == 1 ? 1 : 0, -1, 0, 0};
		items[i] = Item{uint32_t(t), uint32_t(r.target_loc), uint32_t(r.query_loc) | (r.strand == TNT_MINUS ? ITEM_MINUS : 0u) | (uint32_t(i) << ITEM_SLOT_SHIFT), 0};
		score[2 * i] = INT32_MAX, score[2 * i + 1] = INT32_MIN; // (not served: passes)
		const int start = std::max(0, r.target_loc - (r.query_loc + FLANK));
		const uint32_t stop = std::min<uint32_t>(uint32_t(start + L + 2 * FLANK), db->h_len[t]);
		const uint32_t *eb = db->h_exc_pos.data() + db->h_exc_off[t], *ee = db->h_exc_pos.data() + db->h_exc_off[t + 1];
		const uint32_t *it = std::lower_bound(eb, ee, uint32_t(start));
		const int rows = pick_instance(size_t(L)) & 0xFF;
		if ((it == ee || *it >= stop) && L <= SB_MAX_ROWS && has_sbound_instance(rows)) by_group[std::make_tuple(rows, r.oligo, r.role == 1 ? 1 : 0)].push_back(uint32_t(i)); // plain windows of oligos of up to 40 bases
		else if (it != ee && *it < stop && L <= SB_MAX_ROWS && has_sbound_degen_instance(pick_lmax(size_t(L)) & 0xFF)) by_rows_degen[pick_lmax(size_t(L)) & 0xFF].push_back(uint32_t(i)); // windows with an IUPAC code or an inosine: the general instance
	}
	DevBuf<Slot> dslots;
	DevBuf<int32_t> dscore;
	DevBuf<uint32_t> dindex, dfirst, dfill, dwork;
	HIP_TRY(dslots.upload(slots), ctx->err);
	HIP_TRY(dscore.alloc(2 * n), ctx->err);
	HIP_TRY(hipStreamSynchronize(ctx->stream), ctx->err);
	HIP_TRY(hipMemcpy(ctx->S().items.p, items.data(), n * sizeof(Item), hipMemcpyHostToDevice), ctx->err);
	HIP_TRY(hipMemcpy(dscore.p, score, 2 * n * sizeof(int32_t), hipMemcpyHostToDevice), ctx->err);
	// the groups side by side, a launch per instance and SB_MAX_BINS groups
	struct Launch {
		int rows;
		uint32_t g0, n_groups;
	};
	std::vector<uint32_t> order, first, fill;
	std::vector<Launch> launches;
	for (const auto &kv : by_group) {
		const int rows = std::get<0>(kv.first);
		if (launches.empty() || launches.back().rows != rows || launches.back().n_groups == uint32_t(SB_MAX_BINS)) launches.push_back(Launch{rows, uint32_t(first.size()), 0u});
		launches.back().n_groups++;
		first.push_back(uint32_t(order.size())), fill.push_back(uint32_t(kv.second.size()));
		order.insert(order.end(), kv.second.begin(), kv.second.end());
	}
	DevBuf<uint32_t> dindex_d;
	if (!by_rows_degen.empty()) {
		std::vector<uint32_t> order_d;
		for (const auto &kv : by_rows_degen) order_d.insert(order_d.end(), kv.second.begin(), kv.second.end());
		HIP_TRY(dindex_d.upload(order_d), ctx->err);
		SBoundDegenArgs sd;
		memset(&sd, 0, sizeof(sd));
		sd.tab = ctx->stab.p, sd.dtab = ctx->dstab.p, sd.ftab = ctx->ftab.p;
		sd.rt = reinterpret_cast<const uint32_t *>(reinterpret_cast<const char *>(ctx->tab.p) + offsetof(Tables, rt));
		sd.db = make_view(db, 0, db->n_targets);
		sd.olig_q4 = ctx->olig_q4.p, sd.olig_len = ctx->olig_len.p, sd.slots = dslots.p, sd.items = ctx->S().items.p;
		sd.score_out = dscore.p;
		size_t at = 0;
		for (const auto &kv : by_rows_degen) {
			sd.index = dindex_d.p + at;
			sd.n_items = uint32_t(kv.second.size());
			launch_sbound_degen(sd, int(std::min<size_t>(size_t(ctx->n_cu) * 2, (kv.second.size() + SBD_THREADS - 1) / SBD_THREADS)), kv.first, ctx->stream);
			at += kv.second.size();
		}
	}
	if (order.empty() && by_rows_degen.empty()) return TNT_OK;
	if (!order.empty()) HIP_TRY(dindex.upload(order), ctx->err);
	if (!order.empty()) {
		HIP_TRY(dfirst.upload(first), ctx->err);
		HIP_TRY(dfill.upload(fill), ctx->err);
		HIP_TRY(dwork.upload(std::vector<uint32_t>(launches.size(), 0u)), ctx->err);
	}
	SBoundArgs sb;
	memset(&sb, 0, sizeof(sb));
	sb.tab = ctx->stab.p, sb.ftab = ctx->ftab.p, sb.db = make_view(db, 0, db->n_targets);
	sb.olig_q4 = ctx->olig_q4.p, sb.olig_len = ctx->olig_len.p, sb.slots = dslots.p, sb.items = ctx->S().items.p;
	sb.score_out = dscore.p;
	sb.index = dindex.p;
	for (size_t l = 0; l < launches.size(); l++) {
		sb.bin_end = dfirst.p + launches[l].g0, sb.bin_count = nullptr, sb.bin_fill = dfill.p + launches[l].g0, sb.n_bins = launches[l].n_groups;
		sb.next_work = dwork.p + l;
		size_t blocks = 0;
		for (uint32_t g = 0; g < launches[l].n_groups; g++) blocks += (fill[launches[l].g0 + g] + SB_BLOCK - 1) / SB_BLOCK;
		launch_sbound(sb, int(std::min<size_t>(size_t(ctx->n_cu) * 3, (blocks + SB_WAVES - 1) / SB_WAVES)), launches[l].rows, ctx->stream);
	}
	HIP_TRY(hipGetLastError(), ctx->err);
	HIP_TRY(hipStreamSynchronize(ctx->stream), ctx->err);
	HIP_TRY(hipMemcpy(score, dscore.p, 2 * n * sizeof(int32_t), hipMemcpyDeviceToHost), ctx->err);
	return TNT_OK;
}

int tnt_oligo_structure(tnt_ctx *ctx, const char *oa, const char *ob, float ct_a, float ct_b, float *hairpin_a, float *homodimer_a,
                        float *heterodimer_ab)
{
	if (!ctx || !oa || !hairpin_a || !homodimer_a || (ob && !heterodimer_ab) || !(ct_a > 0.0f) || (ob && !(ct_b > 0.0f))) return TNT_E_ARG;
	if (ctx->pending) {
		ctx->err = "a search is pending on this context: collect it first";
		return TNT_E_ARG;
	}
	HIP_TRY(hipSetDevice(ctx->device), ctx->err);
	const size_t struct_len = std::max(strlen(oa), ob ? strlen(ob) : size_t(0));
	const int struct_rows = struct_len <= 64 ? 64 : (struct_len <= size_t(MAX_LANE_OLIGO) ? MAX_LANE_OLIGO : int(std::min<size_t>(struct_len, MAX_OLIGO))); // which unit-level instance
	int rc = ensure_work_buffers(ctx, struct_rows);
	if (rc) return rc;
	const char *seqs[2] = {oa, ob};
	const int n_ol = ob ? 2 : 1;
	std::vector<uint8_t> base(2 * MAX_OLIGO, 0);
	std::vector<uint32_t> q4(2 * (MAX_OLIGO / 4), 0);
	std::vector<int32_t> len(2, 0);
	for (int o = 0; o < n_ol; o++) {
		const size_t L = strlen(seqs[o]);
		if (L == 0 || L > size_t(TNT_MAX_OLIGO)) {
			ctx->err = "oligo length must be 1..1024";
			return TNT_E_LIMIT;
		}
		len[o] = int(L);
		for (size_t k = 0; k < L; k++) {
			const int b = char_to_base(seqs[o][k]);
			if (b < 0) {
				ctx->err = "oligos must be expanded to A/C/G/T/I";
				return TNT_E_ARG;
			}
			base[o * MAX_OLIGO + k] = uint8_t(b);
			const uint32_t r = uint32_t(char_to_base(seqs[o][L - 1 - k]));
			q4[o * (MAX_OLIGO / 4) + (k >> 2)] |= (r * 4u) << ((k & 3) * 8);
		}
	}
	// strand(c_a, c_b) = excess - limit / 2 (nuc_cruc.h:890-910)
	const float ct_self = ct_a - 0.5f * ct_a;
	const float ct_pair = ob ? ((ct_a > ct_b) ? ct_a - 0.5f * ct_b : ct_b - 0.5f * ct_a) : ct_self;
	std::vector<Slot> slots(2);
	slots[0] = Slot{0, logf(ct_self * 1.0f), 0, -1, 0, 0};
	slots[1] = Slot{0, logf(ct_pair * 1.0f), 0, -1, 0, 0};
	std::vector<Item> items;
	items.push_back(Item{3u << 29, 0, 0u << ITEM_SLOT_SHIFT, 0});                  // hairpin of a
	items.push_back(Item{2u << 29, 0, 0u << ITEM_SLOT_SHIFT, 0});                  // homodimer of a
	if (ob) items.push_back(Item{(1u << 29) | 1u, 0, 1u << ITEM_SLOT_SHIFT, 0});   // query a, target b
	DevBuf<uint8_t> dbase;
	DevBuf<uint32_t> dq4;
	DevBuf<int32_t> dlen;
	DevBuf<Slot> dslots;
	DevBuf<BindFull> dfull;
	HIP_TRY(dbase.upload(base), ctx->err);
	HIP_TRY(dq4.upload(q4), ctx->err);
	HIP_TRY(dlen.upload(len), ctx->err);
	HIP_TRY(dslots.upload(slots), ctx->err);
	HIP_TRY(dfull.alloc(items.size()), ctx->err);
	HIP_TRY(hipStreamSynchronize(ctx->stream), ctx->err);
	HIP_TRY(hipMemcpy(ctx->S().items.p, items.data(), items.size() * sizeof(Item), hipMemcpyHostToDevice), ctx->err);
	DbView v;
	memset(&v, 0, sizeof(v));
	BindArgs b = bind_args(ctx, v);
	b.olig_base = dbase.p;
	b.olig_q4 = dq4.p;
	b.olig_len = dlen.p;
	b.slots = dslots.p;
	b.groups = nullptr;
	b.n_items = uint32_t(items.size());
	b.full = dfull.p;
	launch_bind(ctx, b, 1, 0, nullptr, struct_rows);
	HIP_TRY(hipGetLastError(), ctx->err);
	HIP_TRY(hipStreamSynchronize(ctx->stream), ctx->err);
	std::vector<BindFull> full(items.size());
	HIP_TRY(hipMemcpy(full.data(), dfull.p, full.size() * sizeof(BindFull), hipMemcpyDeviceToHost), ctx->err);
	*hairpin_a = full[0].tm;
	*homodimer_a = full[1].tm;
	if (ob) *heterodimer_ab = full[2].tm;
	return TNT_OK;
}

int tnt_long_text(tnt_ctx *ctx, const char *field, char *buf, size_t cap)
{
	if (!field || !buf || cap == 0) return TNT_E_ARG;
	const char *src = field;
	size_t n = 0;
	if (static_cast<unsigned char>(field[0]) == 0x7Fu && field[1] == 'L' && field[2] == 'T') { // a reference into the long-text arena
		if (!ctx) return TNT_E_ARG;
		unsigned long off = 0, len = 0;
		for (int i = 0; i < 8; i++) off = off * 16 + (unsigned long)(field[3 + i] <= '9' ? field[3 + i] - '0' : field[3 + i] - 'a' + 10);
		for (int i = 0; i < 6; i++) len = len * 16 + (unsigned long)(field[11 + i] <= '9' ? field[11 + i] - '0' : field[11 + i] - 'a' + 10);
		if (off + len > ctx->h_ltext.size()) {
			ctx->err = "tnt_long_text: the reference is older than the context's last search / bind call";
			return TNT_E_ARG;
		}
		src = ctx->h_ltext.data() + off;
		n = len;
	} else {
		n = strnlen(field, TNT_ALIGN_TEXT);
	}
	if (n + 1 > cap) return TNT_E_LIMIT;
	memcpy(buf, src, n);
	buf[n] = 0;
	return int(n);
}

int tnt_hit_sequence(tnt_ctx *ctx, tnt_db *db, const tnt_hit *hit, char *buf, size_t cap)
{
	if (!ctx || !db || !hit || !buf) return TNT_E_ARG;
	const size_t t = size_t(hit->seq_index) - db->first_seq_index;
	if (t >= db->n_targets) return TNT_E_ARG;
	if (!db->host_copy) { // database built on the device (FASTA ingest): fetch the rows once
		HIP_TRY(hipSetDevice(ctx->device), ctx->err);
		db->h_packed.resize(db->packed.n);
		db->h_exc_pos.resize(db->exc_pos.n);
		db->h_exc_code.resize(db->exc_code.n);
		HIP_TRY(hipMemcpy(db->h_packed.data(), db->packed.p, db->packed.n * sizeof(uint32_t), hipMemcpyDeviceToHost), ctx->err);
		HIP_TRY(hipMemcpy(db->h_exc_pos.data(), db->exc_pos.p, db->exc_pos.n * sizeof(uint32_t), hipMemcpyDeviceToHost), ctx->err);
		HIP_TRY(hipMemcpy(db->h_exc_code.data(), db->exc_code.p, db->exc_code.n, hipMemcpyDeviceToHost), ctx->err);
		db->host_copy = true;
	}
	const uint32_t len = db->h_len[t];
	const uint32_t *pk = db->h_packed.data() + db->h_word_off[t];
	auto code_at = [&](int pos) -> int {
		int code = int((pk[pos >> 4] >> ((pos & 15) * 2)) & 3u);
		const uint32_t *b = db->h_exc_pos.data() + db->h_exc_off[t], *e = db->h_exc_pos.data() + db->h_exc_off[t + 1];
		const uint32_t *it = std::lower_bound(b, e, uint32_t(pos));
		if (it != e && *it == uint32_t(pos)) code = db->h_exc_code[it - db->h_exc_pos.data()];
		return code;
	};
	static const char *same = "ACGTIMRSVWYHKDBN-?", *comp = "TGCAIKYSBWRDMHVN-?";
	const bool primers = hit->has_primers != 0;
	const int start = primers ? hit->amplicon_start : hit->probe_start;
	const int stop = primers ? hit->amplicon_stop : hit->probe_stop;
	if (stop < start) return TNT_E_ARG;
	const size_t n = size_t(stop - start + 1);
	if (n + 1 > cap) return TNT_E_LIMIT;
	// The text is '-' where nothing is copied.  Which positions are copied when the range sticks out of the target
	// differs between the three search functions; each is followed to the letter:
	//   amplicon()  amplicon_search.cpp:508-537   plus: target as is, minus: reverse complement; padded with '-'
	//   padlock()   padlock_search.cpp:205-218 (its MINUS products: as is, first index max(0, 1 - start)),
	//               :339-351 (its PLUS products: reverse complement, first index max(0, stop - len - 1))
	//   hybrid()    probe_search.cpp:138-152,208-222: the copy always starts at index 0 from the clamped end, i.e.
	//               the window slides into the target instead of being padded
	const bool padlock = primers && (ctx->opt.assay_format == TNT_ASSAY_PADLOCK || ctx->opt.assay_format == TNT_ASSAY_MIPS);
	const bool minus = primers ? ((hit->primer_strand == TNT_MINUS) != padlock) : (hit->probe_strand == TNT_MINUS);
	for (size_t i = 0; i < n; i++) buf[i] = '-';
	if (!minus) {
		long ptr = std::max(0, start);
		const long i0 = !primers ? 0 : (padlock ? std::max(0, 1 - start) : std::max(0, -start));
		for (long i = i0; i < long(n); i++, ptr++) {
			if (ptr >= long(len)) break;
			buf[i] = same[code_at(int(ptr))];
		}
	} else {
		long ptr = std::min<long>(stop, long(len) - 1);
		const long i0 = !primers ? 0 : (padlock ? std::max<long>(0, long(stop) - long(len) - 1) : std::max<long>(0, long(stop) - long(len) + 1));
		for (long i = i0; i < long(n); i++, ptr--) {
			if (ptr < 0) break;
			buf[i] = comp[code_at(int(ptr))];
		}
	}
	buf[n] = 0;
	return int(n);
}

} // extern "C"
