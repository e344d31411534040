// tnt_hitrec.h -- how a hit record is filled from bound records: shared by the pair kernels (tnt_pair.hip.h) and the device
// pairing of windowed targets (tnt_sort.hip).  Product code: nothing here includes or links anything under oracle/.
#ifndef TNT_HITREC_H
#define TNT_HITREC_H

#include <hip/hip_runtime.h>
#include "tnt_dev.h"

namespace tnt {

#ifndef TNT_HD
#define TNT_HD __host__ __device__
#endif

#ifndef TNT_EXP_PAIR_SKIP
#define TNT_EXP_PAIR_SKIP 0 // timing experiments (wrong results): 1 no alignment texts in the hits, 2 no hit records at all, 4 nothing behind the gather of a group
#endif
// The alignment text of a bound record into a hit.  Both fields start on a 4-byte boundary and the bind kernel writes its
// text a dword at a time with zeros behind the last character (tnt_bind.hip.h): the copy moves whole dwords, as many as the
// record's text length says -- known up front, so the loads do not wait for each other (a byte loop that stops at the zero
// byte cost the pair kernel half of its time: 0.29 of 0.63 ms per config-2 step).  A text that lives in the long-text arena
// is LTEXT_REF_CHARS characters of reference in the field, whatever its own length.
TNT_HD inline void copy_text(char *dst, const BoundRec &rec)
{
	if (TNT_EXP_PAIR_SKIP & 1) {
		dst[0] = 0;
		return;
	}
	int chars = rec.text_len < ALIGN_TEXT ? rec.text_len : LTEXT_REF_CHARS;
	chars = chars < 0 ? 0 : (chars > ALIGN_TEXT - 1 ? ALIGN_TEXT - 1 : chars);
	const uint32_t *s = reinterpret_cast<const uint32_t *>(rec.text);
	uint32_t *d = reinterpret_cast<uint32_t *>(dst);
	const int words = chars / 4 + 1; // the characters and the zero behind them
	for (int i = 0; i < words; i++) d[i] = s[i];
}

// hybrid_sig keeps its mismatch / gap / clamp counts in int8_t (hybrid_sig.h:121-164): a count beyond 127 -- a probe of
// several hundred bases -- comes out of the reference wrapped, and so out of here
TNT_HD inline int32_t wrap8(int32_t v) { return int32_t(int8_t(uint8_t(uint32_t(v)))); }

TNT_HD inline void init_hit(Hit &h, const AssayDev &A, int seq_index)
{
	// hybrid_sig::init(), hybrid_sig.h:54-110
	h.assay_id = A.id, h.degen_id = A.degen_id, h.seq_index = seq_index;
	h.amplicon_start = h.amplicon_stop = h.probe_start = h.probe_stop = 0;
	h.forward_tm = h.reverse_tm = h.probe_tm = -1.0f;
	h.forward_dH = h.reverse_dH = h.probe_dH = 100.0f;
	h.forward_dS = h.reverse_dS = h.probe_dS = 0.0f;
	h.primer_strand = h.probe_strand = 0;
	h.forward_primer_clamp = h.reverse_primer_clamp = -1;
	h.forward_mm = h.forward_gap = h.reverse_mm = h.reverse_gap = h.probe_mm = h.probe_gap = -1;
	h.forward_oligo = 0, h.reverse_oligo = 1;
	h.has_primers = (A.slot_f >= 0 && A.slot_r >= 0) ? 1 : 0;
	h.has_probe = (A.slot_p >= 0) ? 1 : 0;
	h.forward_align[0] = h.reverse_align[0] = h.probe_align[0] = 0;
}

} // namespace tnt
#endif
