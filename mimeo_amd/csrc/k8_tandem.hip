// K8 — tandem-repeat scorer: the role of `trf F 2 7 7 80 10 50 50 -m -h -ngs` in mimeo map
// (reference src/mimeo/wrappers.py:120-262 trfFilter; flags src/mimeo/run_map.py:145-178).
// The reference only uses TRF's masked output to compute, per hit, the percentage of bases
// replaced by N and keeps the hit if it is below --maxtandem (wrappers.py:237-240).
//
// PARITY UNPINNED: TRF is an external heuristic program that is absent here; this is a
// specification of our own (DESIGN.md "Tandem scorer v2"), restated on the CPU in
// oracle/pipeline.py.  Like TRF it scores a candidate period p by aligning the sequence with a copy of itself p
// bases back, with TRF's three weights: +match, -mismatch, -delta per inserted or deleted base.  For every period
// p = 1 .. maxperiod the slice is aligned locally (scores restart at 0) with itself on the diagonals p - b .. p + b
// (b = 0 for p = 1, 1 for p < 5, else 2: a copy may drift by that many bases through indels):
//   H[j][d] = max(0, H[j-1][d] + s(S[j], S[j-d]), H[j-1][d-1] - delta, H[j][d+1] - delta)
// (ties: diagonal move, then the insertion, then the deletion); a path that reaches a new best >= minscore marks
// everything from the first base of its earlier copy to the base it has reached.  The masked set is the union over
// the periods.  delta <= 0 switches the indel moves off (b = 0): the gap-free scorer of round 1.  TRF's detection
// statistics (PM, PI) have no counterpart: every period is simply tried.
//
// One wavefront per slice, lane = period: each lane streams the slice 32 bases at a time with bit-parallel
// self-comparison masks per diagonal; qualified stretches are OR-ed into a per-slice bitmask.
#include "device_util.h"

namespace mimeo {

__device__ __forceinline__ void mark_range(uint32_t *__restrict__ bits, uint32_t a, uint32_t b) {
    // set bits [a, b) of the slice's mask
    while (a < b) {
        uint32_t w = a >> 5, lo = a & 31u, n = min(32u - lo, b - a);
        uint32_t m = (n == 32u ? 0xFFFFFFFFu : ((1u << n) - 1u)) << lo;
        atomicOr(&bits[w], m);
        a += n;
    }
}

constexpr int TBAND = 5;   // diagonals per period at most (b = 2)
struct TCell { int32_t h; uint32_t start, best, mend; };   // score; first base of the earlier copy; best on the path; masked up to

__global__ __launch_bounds__(256) void k8_tandem_mask(const StrandView *__restrict__ views,
                                                      const mimeo_interval *__restrict__ iv, uint64_t n,
                                                      const uint64_t *__restrict__ word_off, int match, int mismatch, int delta,
                                                      int minscore, int maxperiod, uint32_t *__restrict__ bits) {
    const uint64_t wid = ((uint64_t)blockIdx.x * 256 + threadIdx.x) >> 6;
    if (wid >= n) return;
    const uint32_t p = (threadIdx.x & 63u) + 1u;
    if ((int)p > maxperiod) return;
    const mimeo_interval v = iv[wid];
    const StrandView S = views[v.chrom];
    const uint32_t end = min(v.end, S.len);
    if (v.start >= end) return;
    const uint32_t L = end - v.start;
    const uint32_t b = delta > 0 ? (p == 1 ? 0u : (p < 5 ? 1u : 2u)) : 0u, nd = 2 * b + 1, d0 = p - b;   // diagonals d0 .. d0 + nd - 1
    if (L <= d0) return;
    uint32_t *mybits = bits + word_off[wid];
    TCell cur[TBAND], prev[TBAND];
#pragma unroll
    for (int k = 0; k < TBAND; k++) prev[k] = TCell{0, 0, 0, 0};
    for (uint32_t j0 = (d0 / 32u) * 32u; j0 < L; j0 += 32) {
        const Win32 a = win32(S, (int32_t)(v.start + j0));
        uint32_t eq[TBAND];
#pragma unroll
        for (int k = 0; k < TBAND; k++) {
            eq[k] = 0;
            if ((uint32_t)k < nd) {
                const Win32 c = win32(S, (int32_t)(v.start + j0) - (int32_t)(d0 + k));
                eq[k] = ~((a.lo ^ c.lo) | (a.hi ^ c.hi)) & ~(a.nm | c.nm);
            }
        }
        const uint32_t cnt = min(32u, L - j0);
        for (uint32_t t = 0; t < cnt; t++) {
            const uint32_t j = j0 + t;
#pragma unroll
            for (int k = TBAND - 1; k >= 0; k--) {
                if ((uint32_t)k >= nd) continue;
                const uint32_t d = d0 + (uint32_t)k;
                TCell c{0, 0, 0, 0};
                if (j >= d) {
                    // diagonal move (a fresh path starts at the earlier copy's base j - d)
                    const TCell &pd = prev[k];
                    const int32_t sc = ((eq[k] >> t) & 1u) ? match : -mismatch;
                    c.h = pd.h + sc;
                    c.start = pd.h > 0 ? pd.start : j - d;
                    c.best = pd.h > 0 ? pd.best : 0;
                    c.mend = pd.h > 0 ? pd.mend : 0;
                    if (k > 0 && prev[k - 1].h - delta > c.h) { c = prev[k - 1]; c.h -= delta; }          // insertion: (j-1, d-1)
                    if ((uint32_t)k + 1 < nd && cur[k + 1].h - delta > c.h) { c = cur[k + 1]; c.h -= delta; }  // deletion: (j, d+1)
                    if (c.h <= 0) c = TCell{0, 0, 0, 0};
                    else if ((uint32_t)c.h > c.best) {
                        c.best = (uint32_t)c.h;
                        if (c.h >= minscore) {
                            mark_range(mybits, max(c.mend, c.start), j + 1);
                            c.mend = j + 1;
                        }
                    }
                }
                cur[k] = c;
            }
#pragma unroll
            for (int k = 0; k < TBAND; k++) prev[k] = cur[k];
        }
    }
}

__global__ void k8_tandem_count(const mimeo_interval *__restrict__ iv, uint64_t n, const uint64_t *__restrict__ word_off,
                                const uint32_t *__restrict__ bits, uint32_t *__restrict__ masked) {
    const uint64_t wid = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (wid >= n) return;
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t c = 0;
    for (uint64_t w = word_off[wid] + lane; w < word_off[wid + 1]; w += 64) c += __popc(bits[w]);
    for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o);
    if (lane == 0) masked[wid] = c;
}

int tandem_masked_device(const mimeo_genome *A, const mimeo_interval *h_iv, uint64_t n, int match, int mismatch, int delta,
                         int minscore, int maxperiod, uint32_t *h_masked) {
    if (!n) return 0;
    if (maxperiod < 1 || maxperiod > 64) { set_error("tandem scorer: maxperiod must be in 1..64"); return MIMEO_ERR_LIMIT; }
    hipStream_t st = stream();
    std::vector<StrandView> views(A->scaf.size());
    for (size_t i = 0; i < views.size(); i++) views[i] = A->scaf[i].fwd.view(false);
    std::vector<uint64_t> off(n + 1, 0);
    for (uint64_t i = 0; i < n; i++) {
        if (h_iv[i].chrom >= views.size()) { set_error("tandem scorer: chromosome id out of range"); return MIMEO_ERR_ARG; }
        uint32_t e = std::min<uint32_t>(h_iv[i].end, views[h_iv[i].chrom].len);
        uint64_t L = h_iv[i].start < e ? e - h_iv[i].start : 0;
        off[i + 1] = off[i] + (L + 31) / 32;
    }
    DeviceBuf dv, di, dof, db, dm;
    int rc;
    if ((rc = dv.reserve(views.size() * sizeof(StrandView) + 16)) || (rc = di.reserve(n * sizeof(mimeo_interval))) ||
        (rc = dof.reserve((n + 1) * 8)) || (rc = db.reserve(off[n] * 4 + 16)) || (rc = dm.reserve(n * 4)))
        return rc;
    HIP_TRY(hipMemcpyAsync(dv.p, views.data(), views.size() * sizeof(StrandView), hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(di.p, h_iv, n * sizeof(mimeo_interval), hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(dof.p, off.data(), (n + 1) * 8, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemsetAsync(db.p, 0, off[n] * 4 + 16, st));
    HIP_TRY(hipMemsetAsync(dm.p, 0, n * 4, st));
    const uint32_t nb = (uint32_t)((n * 64 + 255) / 256);
    hipLaunchKernelGGL(k8_tandem_mask, dim3(nb), dim3(256), 0, st, (const StrandView *)dv.p, (const mimeo_interval *)di.p, n,
                       (const uint64_t *)dof.p, match, mismatch, delta, minscore, maxperiod, (uint32_t *)db.p);
    hipLaunchKernelGGL(k8_tandem_count, dim3(nb), dim3(256), 0, st, (const mimeo_interval *)di.p, n,
                       (const uint64_t *)dof.p, (const uint32_t *)db.p, (uint32_t *)dm.p);
    HIP_TRY(hipMemcpyAsync(h_masked, dm.p, n * 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    HIP_TRY(hipGetLastError());
    for (DeviceBuf *b : {&dv, &di, &dof, &db, &dm}) b->release();
    return 0;
}

}  // namespace mimeo
