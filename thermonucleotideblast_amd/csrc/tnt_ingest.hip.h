// tnt_ingest.hip.h -- gfx950 target ingest: FASTA text -> 2-bit packed targets + exception lists.
//
// Replaces the per-record reader of the reference (sequence_data_fastx.cpp:187-382: gzseek + gzread inside an
// omp critical section, one byte at a time through ascii_to_hash_base, seq.h:148-189) and the host packer of
// tnt_db_create.  The host only finds the record boundaries (load_fasta's rule, sequence_data_fastx.cpp:13-79:
// a record starts at the first '>' of a line); the text itself is parsed on the device:
//
//   count_kernel   one wavefront per record: number of bases (everything but white space, '*', '-') and of
//                  exceptions (codes > 3) -> lengths, from which the host lays out the packed rows
//   pack_kernel    one wavefront per record: ordered compaction of the bases into 2-bit words (16 per uint32,
//                  low two bits of the code: non-ACGT codes are hashed by their low bits) and
//                  ascending (position, code) exception lists
//
// Both are streaming byte kernels, bound by HBM: 1 text byte read (twice) and 0.25 byte written per base.
#ifndef TNT_INGEST_HIP_H
#define TNT_INGEST_HIP_H

#include <hip/hip_runtime.h>
#include <stdint.h>
#include "tnt_dev.h"

namespace tnt {

constexpr int INGEST_WAVES = 4; // records in flight per workgroup
constexpr int INGEST_THREADS = 64 * INGEST_WAVES;
constexpr uint8_t INGEST_SKIP = 0xFF;

struct IngestArgs {
	const uint8_t *text;       // FASTA bytes, device resident; 4 readable bytes of slack behind the end
	const uint64_t *seq_begin; // per record: first byte after the defline
	const uint64_t *seq_end;   //             one past the record's last byte
	uint32_t n_records;
	const uint8_t *lut;        // 256 entries: DB_* code (seq.h:12-35) or INGEST_SKIP
	uint32_t *len;             // out (count) / in (pack): bases per record
	uint32_t *n_exc;           // out (count): exceptions per record
	const uint64_t *word_off;  // pack: first packed word of every record
	const uint32_t *exc_off;   // pack: first exception of every record
	uint32_t *packed;          // pack: zero-initialised
	uint32_t *exc_pos;
	uint8_t *exc_code;
};

// The four bytes a lane owns in this step: the aligned word at `base + 4 * (step * 64 + lane)`, with the bytes
// outside [b, e) turned into skips.  Returns the four codes packed in one word (INGEST_SKIP = no base).
__device__ __forceinline__ uint32_t ingest_codes(const uint8_t *text, const uint8_t *lut, uint64_t base, uint64_t b, uint64_t e, uint32_t step,
                                                 int lane)
{
	const uint64_t at = base + 4ull * (uint64_t(step) * 64ull + uint64_t(lane));
	uint32_t out = 0xFFFFFFFFu;
	if (at < e && at + 4 > b) {
		const uint32_t w = *reinterpret_cast<const uint32_t *>(text + at);
#pragma unroll
		for (int k = 0; k < 4; k++) {
			const uint64_t pos = at + uint64_t(k);
			uint32_t code = INGEST_SKIP;
			if (pos >= b && pos < e) code = lut[(w >> (8 * k)) & 255u];
			out = (out & ~(255u << (8 * k))) | (code << (8 * k));
		}
	}
	return out;
}

__global__ __launch_bounds__(INGEST_THREADS) void ingest_count_kernel(IngestArgs a)
{
	__shared__ uint8_t s_lut[256];
	s_lut[threadIdx.x & 255] = a.lut[threadIdx.x & 255];
	__syncthreads();
	const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
	for (uint32_t r = blockIdx.x * INGEST_WAVES + wv; r < a.n_records; r += gridDim.x * INGEST_WAVES) {
		const uint64_t b = a.seq_begin[r], e = a.seq_end[r];
		const uint64_t base = b & ~3ull;
		const uint32_t steps = uint32_t((e - base + 255) / 256);
		uint32_t nb = 0, ne = 0;
		for (uint32_t s = 0; s < steps; s++) {
			const uint32_t c4 = ingest_codes(a.text, s_lut, base, b, e, s, lane);
#pragma unroll
			for (int k = 0; k < 4; k++) {
				const uint32_t code = (c4 >> (8 * k)) & 255u;
				nb += code != INGEST_SKIP ? 1u : 0u;
				ne += (code != INGEST_SKIP && code > 3u) ? 1u : 0u;
			}
		}
		for (int o = 32; o > 0; o >>= 1) nb += __shfl_xor(nb, o), ne += __shfl_xor(ne, o);
		if (lane == 0) a.len[r] = nb, a.n_exc[r] = ne;
	}
}

__global__ __launch_bounds__(INGEST_THREADS) void ingest_pack_kernel(IngestArgs a)
{
	__shared__ uint8_t s_lut[256];
	__shared__ uint32_t s_stage[INGEST_WAVES][20]; // words touched by one step: <= 256 bases + the carried partial word
	s_lut[threadIdx.x & 255] = a.lut[threadIdx.x & 255];
	__syncthreads();
	const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
	uint32_t *stage = s_stage[wv];
	for (uint32_t r = blockIdx.x * INGEST_WAVES + wv; r < a.n_records; r += gridDim.x * INGEST_WAVES) {
		const uint64_t b = a.seq_begin[r], e = a.seq_end[r];
		const uint64_t base = b & ~3ull;
		const uint32_t steps = uint32_t((e - base + 255) / 256);
		uint32_t *pk = a.packed + a.word_off[r];
		uint32_t *xp = a.exc_pos + a.exc_off[r];
		uint8_t *xc = a.exc_code + a.exc_off[r];
		uint32_t nb = 0, ne = 0; // bases / exceptions written so far (wave-uniform)
		if (lane < 20) stage[lane] = 0;
		for (uint32_t s = 0; s < steps; s++) {
			const uint32_t c4 = ingest_codes(a.text, s_lut, base, b, e, s, lane);
			uint32_t cb = 0, ce = 0;
#pragma unroll
			for (int k = 0; k < 4; k++) {
				const uint32_t code = (c4 >> (8 * k)) & 255u;
				cb += code != INGEST_SKIP ? 1u : 0u;
				ce += (code != INGEST_SKIP && code > 3u) ? 1u : 0u;
			}
			// wave-wide exclusive prefix sums: the lanes hold consecutive text bytes, so this is the base order
			uint32_t ib = cb, ie = ce;
			for (int d = 1; d < 64; d <<= 1) {
				const uint32_t yb = __shfl_up(ib, d), ye = __shfl_up(ie, d);
				if (lane >= d) ib += yb, ie += ye;
			}
			const uint32_t tb = __shfl(ib, 63), te = __shfl(ie, 63);
			uint32_t pos = nb + ib - cb, xo = ne + ie - ce;
			const uint32_t w0 = nb >> 4; // first word of this step (its low part may be carried over)
#pragma unroll
			for (int k = 0; k < 4; k++) {
				const uint32_t code = (c4 >> (8 * k)) & 255u;
				if (code == INGEST_SKIP) continue;
				atomicOr(&stage[(pos >> 4) - w0], (code & 3u) << ((pos & 15u) * 2u));
				if (code > 3u) {
					xp[xo] = pos;
					xc[xo] = uint8_t(code);
					xo++;
				}
				pos++;
			}
			nb += tb, ne += te;
			// flush the completed words, carry the partial one
			__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
			__builtin_amdgcn_wave_barrier();
			__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
			const uint32_t w1 = nb >> 4;
			uint32_t keep = 0;
			if (uint32_t(lane) < w1 - w0) pk[w0 + lane] = stage[lane];
			if (lane == 0) keep = stage[w1 - w0];
			__builtin_amdgcn_wave_barrier();
			if (lane < 20) stage[lane] = 0;
			if (lane == 0) stage[0] = keep;
			__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
			__builtin_amdgcn_wave_barrier();
			__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
		}
		if (lane == 0 && (nb & 15u)) pk[nb >> 4] = stage[0];
		__builtin_amdgcn_wave_barrier();
	}
}

// 1-byte codes of target positions [p0, p0 + n) rebuilt from the packed rows and the exception lists
// (tnt_db_target_codes: debugging / parity aid, and the source of the amplicon text)
struct UnpackArgs {
	DbView db;
	uint32_t target, p0, n;
	uint8_t *out;
};

__global__ void unpack_kernel(UnpackArgs a)
{
	const uint32_t *pk = a.db.packed + a.db.word_off[a.target];
	const uint32_t e0 = a.db.exc_off[a.target], e1 = a.db.exc_off[a.target + 1];
	for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < a.n; i += gridDim.x * blockDim.x) {
		const uint32_t pos = a.p0 + i;
		uint32_t code = (pk[pos >> 4] >> ((pos & 15u) * 2u)) & 3u;
		uint32_t lo = e0, hi = e1;
		while (lo < hi) {
			const uint32_t mid = (lo + hi) >> 1;
			if (a.db.exc_pos[mid] < pos) lo = mid + 1;
			else hi = mid;
		}
		if (lo < e1 && a.db.exc_pos[lo] == pos) code = a.db.exc_code[lo];
		a.out[i] = uint8_t(code);
	}
}

} // namespace tnt

#endif
