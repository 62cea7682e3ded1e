// h2g_go_kernels.h — the go() kernels (HI_Aligner::go per read / per pair).  They are instantiated in their own translation
// units (h2g_k_go_linear.hip, h2g_k_go_graph.hip) so that the three .hip files compile in parallel and the linear kernels
// never see the graph code; h2g_kernels.hip only declares them (H2G_GO_DECLARE_ONLY) and launches them.
#pragma once
#include <hip/hip_runtime.h>
#include "h2g_core.h"
#include "h2g_align.h"

using namespace h2g;

// waves per SIMD each go() kernel is compiled for (register budget = 512 / waves; measured on the bench workload)
#ifndef H2G_GRAPH_WAVES
#define H2G_GRAPH_WAVES 5
#endif
#ifndef H2G_LINEAR_WAVES
#define H2G_LINEAR_WAVES 5
#endif
#ifndef H2G_LINEAR_PE_WAVES
#define H2G_LINEAR_PE_WAVES 3
#endif
#ifndef H2G_GRAPH_PE_WAVES
#define H2G_GRAPH_PE_WAVES 3
#endif

__device__ __forceinline__ void wave_add(unsigned long long* dst, unsigned long long v) {
	for(int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
	if((threadIdx.x & 63) == 0 && v) atomicAdd(dst, v);
}

struct GraphArgs { DAlts alts; GraphWS* base; };   // graph index: ALT database + per-lane graph scratch (base == nullptr on linear)

template <int WAVES_PER_SIMD, bool GRAPH>
__global__ __launch_bounds__(256, WAVES_PER_SIMD) void k_align(DGfm g, DRef ref, DLocalSet ls, DReads rd, AlnParams P, const char* names,
                                               const uint32_t* name_offs, AlignWS* pool, ReadOut* outs, h2g_alnres* aln,
                                               unsigned long long* counters, const uint32_t* perm, unsigned long long* work,
                                               uint8_t* sw_base, size_t sw_stride, GraphArgs ga)
{
	const size_t tid = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
	const size_t stride = (size_t)gridDim.x * blockDim.x;
	AlignWS* ws = pool + tid;
	AlnCtx C; C.g = &g; C.ref = &ref; C.ls = &ls; C.P = &P;
	C.sw = sw_base ? sw_base + tid * sw_stride : nullptr;
	C.alts = &ga.alts; C.gws = ga.base ? ga.base + tid : nullptr; C.graph = GRAPH;
	unsigned long long nrank = 0, nsteps = 0, naln = 0, novf = 0, nside = 0;
	// per-lane packed copy of the current read in LDS: the byte-per-base global reads of the search / extension
	// loops become conflict-free ds_read_b32 (word k of lane t at [k][t])
	__shared__ uint32_t s_pk[(H2G_PK_WORDS + H2G_PK_WORDS / 2) * 256];
	DReads rdl = rd;
	rdl.pk = s_pk + threadIdx.x;
	rdl.pk_stride = 256;
	// scheduling knob (work != nullptr): a lane that finishes a read takes the next one of the work list instead of
	// waiting for its wave's round (measured: no gain, the kernel is issue-bound under divergence)
	for(size_t jj = tid;; jj += stride) {
		size_t j = jj;
		if(work) j = (size_t)atomicAdd(work, 1ull);
		if(j >= rd.n) break;
		const size_t i = perm ? perm[j] : j;
		ReadOut o;
		const uint32_t a = name_offs[i], b = name_offs[i + 1];
		{
			const uint32_t ro = rd.offs[i], rl = rd.offs[i + 1] - ro;
			rdl.pk_read = 0xffffffffu;
			if(rl <= H2G_PK_MAXLEN) {
				for(uint32_t w = 0; w < (rl + 15) / 16; w++) {
					uint32_t bits = 0, mask = 0;
					for(uint32_t k = 0; k < 16 && w * 16 + k < rl; k++) {
						const uint32_t c = rd.codes[ro + w * 16 + k];
						bits |= (c & 3u) << (2 * k);
						mask |= (c > 3u ? 1u : 0u) << k;
					}
					s_pk[w * 256 + threadIdx.x] = bits;
					uint32_t& mw = s_pk[(H2G_PK_WORDS + (w >> 1)) * 256 + threadIdx.x];
					mw = (w & 1) ? (mw | (mask << 16)) : mask;
				}
				rdl.pk_read = (uint32_t)i;
			}
		}
		al_read(C, rdl, (uint32_t)i, names + a, b - a, ws, &o);
		outs[i] = o;
		for(uint32_t k = 0; k < o.nselect && k < H2G_ALN_CAP; k++) {
			const AlnRec& r = ws->m[0].res[o.select[k]];
			h2g_alnres& d = aln[i * H2G_ALN_CAP + k];
			d.fw = r.fw; d.tidx = r.tidx; d.toff = r.toff; d.len = r.len; d.trim5 = r.trim5; d.trim3 = r.trim3;
			d.nedits = r.nedits; d.pad = 0; d.score = r.score;
			for(uint32_t e = 0; e < r.nedits; e++) d.edits[e] = r.edits[e];
		}
		nrank += o.nrank; nsteps += o.nsteps; naln += o.nselect > 0; novf += o.overflow != 0; nside += o.nside;
	}
	wave_add(counters + 0, nrank);
	wave_add(counters + 1, nside);
	wave_add(counters + 2, nsteps);
	wave_add(counters + 4, naln);
	wave_add(counters + 5, novf);
}

// lane = one read pair; both mates are packed into LDS (mate 2 behind mate 1)
template <bool GRAPH, int WIDE = 0>   // WIDE only tags the linear build with AL_MAX_GHITS = 20 (-k up to 10, --sensitive)
__global__ __launch_bounds__(256, GRAPH ? H2G_GRAPH_PE_WAVES : H2G_LINEAR_PE_WAVES) void k_align_pairs(DGfm g, DRef ref, DLocalSet ls, DReads rd1, DReads rd2, AlnParams P,
                                                        const char* names1, const uint32_t* noffs1, const char* names2,
                                                        const uint32_t* noffs2, AlignWS* pool, PairOut* outs, h2g_alnres* aln1,
                                                        h2g_alnres* aln2, unsigned long long* counters, uint8_t* sw_base, size_t sw_stride, GraphArgs ga)
{
	const size_t tid = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
	const size_t stride = (size_t)gridDim.x * blockDim.x;
	AlignWS* ws = pool + tid;
	AlnCtx C; C.g = &g; C.ref = &ref; C.ls = &ls; C.P = &P;
	C.sw = sw_base ? sw_base + tid * sw_stride : nullptr;
	C.alts = &ga.alts; C.gws = ga.base ? ga.base + tid : nullptr; C.graph = GRAPH;
	unsigned long long nrank = 0, nsteps = 0, npair = 0, novf = 0, nside = 0;
	__shared__ uint32_t s_pk[2 * (H2G_PK_WORDS + H2G_PK_WORDS / 2) * 256];
	DReads rl[2] = {rd1, rd2};
	for(int m = 0; m < 2; m++) { rl[m].pk = s_pk + m * (H2G_PK_WORDS + H2G_PK_WORDS / 2) * 256 + threadIdx.x; rl[m].pk_stride = 256; }
	for(size_t i = tid; i < rd1.n; i += stride) {
		for(int m = 0; m < 2; m++) {
			const DReads& rd = m == 0 ? rd1 : rd2;
			uint32_t* pk = s_pk + m * (H2G_PK_WORDS + H2G_PK_WORDS / 2) * 256;
			const uint32_t ro = rd.offs[i], rlen = rd.offs[i + 1] - ro;
			rl[m].pk_read = 0xffffffffu;
			if(rlen <= H2G_PK_MAXLEN) {
				for(uint32_t w = 0; w < (rlen + 15) / 16; w++) {
					uint32_t bits = 0, mask = 0;
					for(uint32_t k = 0; k < 16 && w * 16 + k < rlen; k++) {
						const uint32_t c = rd.codes[ro + w * 16 + k];
						bits |= (c & 3u) << (2 * k);
						mask |= (c > 3u ? 1u : 0u) << k;
					}
					pk[w * 256 + threadIdx.x] = bits;
					uint32_t& mw = pk[(H2G_PK_WORDS + (w >> 1)) * 256 + threadIdx.x];
					mw = (w & 1) ? (mw | (mask << 16)) : mask;
				}
				rl[m].pk_read = (uint32_t)i;
			}
		}
		PairOut o;
		al_pair(C, rl[0], rl[1], (uint32_t)i, names1 + noffs1[i], noffs1[i + 1] - noffs1[i], names2 + noffs2[i], noffs2[i + 1] - noffs2[i], ws, &o);
		outs[i] = o;
		for(int m = 0; m < 2; m++) {
			h2g_alnres* dst = (m == 0 ? aln1 : aln2) + i * H2G_PAIR_RES_CAP;
			const uint32_t n = o.nres[m] < H2G_PAIR_RES_CAP ? o.nres[m] : H2G_PAIR_RES_CAP;
			for(uint32_t k = 0; k < n; k++) {
				const AlnRec& r = ws->m[m].res[k];
				h2g_alnres& d = dst[k];
				d.fw = r.fw; d.tidx = r.tidx; d.toff = r.toff; d.len = r.len; d.trim5 = r.trim5; d.trim3 = r.trim3;
				d.nedits = r.nedits; d.pad = 0; d.score = r.score;
				for(uint32_t e = 0; e < r.nedits; e++) d.edits[e] = r.edits[e];
			}
		}
		nrank += o.nrank; nsteps += o.nsteps; npair += o.npairs > 0; nside += o.nside;
		novf += (o.overflow != 0 || o.nres[0] > H2G_PAIR_RES_CAP || o.nres[1] > H2G_PAIR_RES_CAP);
	}
	wave_add(counters + 0, nrank);
	wave_add(counters + 1, nside);
	wave_add(counters + 2, nsteps);
	wave_add(counters + 4, npair);
	wave_add(counters + 5, novf);
}

#if defined(H2G_GO_DECLARE_ONLY)
extern template __global__ void k_align<H2G_LINEAR_WAVES, false>(DGfm, DRef, DLocalSet, DReads, AlnParams, const char*, const uint32_t*, AlignWS*, ReadOut*, h2g_alnres*,
                                                  unsigned long long*, const uint32_t*, unsigned long long*, uint8_t*, size_t, GraphArgs);
extern template __global__ void k_align<H2G_GRAPH_WAVES, true>(DGfm, DRef, DLocalSet, DReads, AlnParams, const char*, const uint32_t*, AlignWS*, ReadOut*, h2g_alnres*,
                                                 unsigned long long*, const uint32_t*, unsigned long long*, uint8_t*, size_t, GraphArgs);
extern template __global__ void k_align_pairs<false>(DGfm, DRef, DLocalSet, DReads, DReads, AlnParams, const char*, const uint32_t*, const char*,
                                                     const uint32_t*, AlignWS*, PairOut*, h2g_alnres*, h2g_alnres*, unsigned long long*, uint8_t*, size_t, GraphArgs);
extern template __global__ void k_align<3, false>(DGfm, DRef, DLocalSet, DReads, AlnParams, const char*, const uint32_t*, AlignWS*, ReadOut*, h2g_alnres*,
                                                  unsigned long long*, const uint32_t*, unsigned long long*, uint8_t*, size_t, GraphArgs);
extern template __global__ void k_align_pairs<false, 1>(DGfm, DRef, DLocalSet, DReads, DReads, AlnParams, const char*, const uint32_t*, const char*,
                                                        const uint32_t*, AlignWS*, PairOut*, h2g_alnres*, h2g_alnres*, unsigned long long*, uint8_t*, size_t, GraphArgs);
extern template __global__ void k_align_pairs<true>(DGfm, DRef, DLocalSet, DReads, DReads, AlnParams, const char*, const uint32_t*, const char*,
                                                    const uint32_t*, AlignWS*, PairOut*, h2g_alnres*, h2g_alnres*, unsigned long long*, uint8_t*, size_t, GraphArgs);
#endif
