// K8 — tandem-repeat scorer: the role of `trf F 2 7 7 80 10 50 50 -m -h -ngs` in mimeo map
// (reference src/mimeo/wrappers.py:120-262 trfFilter; flags src/mimeo/run_map.py:145-178).
// The reference only uses TRF's masked output to compute, per hit, the percentage of bases
// replaced by N and keeps the hit if it is below --maxtandem (wrappers.py:237-240).
//
// PARITY UNPINNED: TRF is an external heuristic program that is absent here; this is a
// specification of our own (DESIGN.md "Tandem scorer v1"), restated on the CPU in
// oracle/pipeline.py.  For every period p = 1..maxperiod the slice is compared with itself
// shifted by p (+match for an identical ACGT pair, -mismatch otherwise; no indels).  Scanning
// left to right with a running score that restarts when it drops to <= 0, a segment whose best
// prefix reaches minscore marks [segment start, best end + p) as tandem.  The masked set is
// the union over the periods.
//
// One wavefront per slice, lane = period: each lane streams the slice 32 bases at a time with
// bit-parallel self-comparison masks; qualified segments are OR-ed into a per-slice bitmask.
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

__global__ __launch_bounds__(256) void k8_tandem_mask(const StrandView *__restrict__ views,
                                                      const mimeo_interval *__restrict__ iv, uint64_t n,
                                                      const uint64_t *__restrict__ word_off, int match, int mismatch,
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
    if (L <= p) return;
    uint32_t *mybits = bits + word_off[wid];
    const uint32_t ncmp = L - p;  // comparable positions i in [0, L-p)
    int32_t run = 0, best = 0;
    uint32_t seg = 0, bestend = 0;
    for (uint32_t i0 = 0; i0 < ncmp; i0 += 32) {
        const Win32 a = win32(S, (int32_t)(v.start + i0)), b = win32(S, (int32_t)(v.start + i0 + p));
        uint32_t eq = ~((a.lo ^ b.lo) | (a.hi ^ b.hi)) & ~(a.nm | b.nm);
        const uint32_t cnt = min(32u, ncmp - i0);
        for (uint32_t k = 0; k < cnt; k++) {
            run += ((eq >> k) & 1u) ? match : -mismatch;
            if (run <= 0) {
                if (best >= minscore) mark_range(mybits, seg, min(L, bestend + p));
                run = 0; best = 0; seg = i0 + k + 1;
            } else if (run > best) {
                best = run; bestend = i0 + k + 1;
            }
        }
    }
    if (best >= minscore) mark_range(mybits, seg, min(L, bestend + p));
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

int tandem_masked_device(const mimeo_genome *A, const mimeo_interval *h_iv, uint64_t n, int match, int mismatch,
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
                       (const uint64_t *)dof.p, match, mismatch, minscore, maxperiod, (uint32_t *)db.p);
    hipLaunchKernelGGL(k8_tandem_count, dim3(nb), dim3(256), 0, st, (const mimeo_interval *)di.p, n,
                       (const uint64_t *)dof.p, (const uint32_t *)db.p, (uint32_t *)dm.p);
    HIP_TRY(hipMemcpyAsync(h_masked, dm.p, n * 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    HIP_TRY(hipGetLastError());
    for (DeviceBuf *b : {&dv, &di, &dof, &db, &dm}) b->release();
    return 0;
}

}  // namespace mimeo
