// K7 — coverage-depth collapse (SURVEY §8a A13 + the minLen half of A14):
//   sort -k1,1 -k2n,3n | bedtools genomecov -bg -g lens | awk '$4>=cov' | sort | bedtools merge |
//   awk '$3-$2>=minLen'                       (reference src/mimeo/wrappers.py:1128-1167)
// as pure integer work on the device: every interval becomes a +1 event at its start and a -1
// event at its (clipped) end; events are radix-sorted by (chrom, position), an inclusive scan of
// the deltas gives the depth that holds from each distinct position to the next, and a maximal
// run of positions with depth >= min_cov is one merged region (book-ended runs are contiguous by
// construction, which is what `bedtools merge -d 0` joins).
#include <cstring>

#include <rocprim/rocprim.hpp>

#include "common.h"

namespace mimeo {

__global__ void k7_events(const mimeo_interval *__restrict__ iv, uint64_t n, const uint32_t *__restrict__ chrom_len,
                          uint32_t nchrom, uint64_t *__restrict__ key, int32_t *__restrict__ delta) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    mimeo_interval v = iv[i];
    uint64_t ks = ~0ull, ke = ~0ull;  // invalid intervals sort to the very end with delta 0
    int32_t ds = 0, de = 0;
    if (v.chrom < nchrom) {
        uint32_t L = chrom_len[v.chrom], e = v.end < L ? v.end : L;
        if (v.start < e) {
            ks = ((uint64_t)v.chrom << 32) | v.start; ds = 1;
            ke = ((uint64_t)v.chrom << 32) | e; de = -1;
        }
    }
    key[2 * i] = ks; delta[2 * i] = ds;
    key[2 * i + 1] = ke; delta[2 * i + 1] = de;
}

// flag[i] = 1 for the LAST event of every distinct (chrom, position): depth[i] holds from there to the next position
__global__ void k7_last_flags(const uint64_t *__restrict__ key, uint64_t n, uint8_t *__restrict__ flag) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t k = key[i];
    flag[i] = (k != ~0ull && (i + 1 == n || key[i + 1] != k)) ? 1 : 0;
}
// On the distinct positions: a region starts where the depth reaches min_cov and ends where it falls below it.  The
// depth is 0 behind the last event of a chromosome, so a run never crosses into the next one; runs of equal or
// different depth >= min_cov that touch are one run by construction (bedtools merge -d 0 joins book-ended intervals).
__global__ void k7_edge_flags(const int32_t *__restrict__ udep, const unsigned long long *__restrict__ nu_dev, int32_t min_cov,
                              uint8_t *__restrict__ sflag, uint8_t *__restrict__ eflag, uint64_t cap) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= cap) return;
    const uint64_t nu = *nu_dev;
    bool s = false, e = false;
    if (i < nu) {
        const bool above = udep[i] >= min_cov, prev = i > 0 && udep[i - 1] >= min_cov;
        s = above && !prev;
        e = !above && prev;
    }
    sflag[i] = s ? 1 : 0;
    eflag[i] = e ? 1 : 0;
}
// starts[j] / ends[j] are the j-th rising / falling edge in (chrom, position) order: region j
__global__ void k7_regions(const uint64_t *__restrict__ starts, const uint64_t *__restrict__ ends,
                           const unsigned long long *__restrict__ nreg_dev, uint32_t min_len, mimeo_interval *__restrict__ out,
                           unsigned long long *__restrict__ nout, uint64_t cap) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= cap || i >= *nreg_dev) return;
    const uint64_t ks = starts[i], ke = ends[i];
    const uint32_t s = (uint32_t)ks, e = (uint32_t)ke;
    if (e - s >= min_len) {
        unsigned long long o = atomicAdd(nout, 1ull);
        out[o] = mimeo_interval{(uint32_t)(ks >> 32), s, e};
    }
}

int coverage_collapse_device(const mimeo_interval *h_iv, uint64_t n, const uint32_t *h_chrom_len, uint32_t nchrom,
                             uint32_t min_cov, uint32_t min_len, std::vector<mimeo_interval> &out, float *ms) {
    out.clear();
    if (!n) return 0;
    hipStream_t st = stream();
    DeviceBuf iv, cl, k1, k2, d1, d2, dep, res, cnt, tmp, fl, fl2;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    // every return path, the error ones included, gives the buffers and the event pair back
    struct Cleanup {
        DeviceBuf *bufs[12];
        hipEvent_t *a, *b;
        ~Cleanup() {
            for (DeviceBuf *x : bufs) x->release();
            if (*a) (void)hipEventDestroy(*a);
            if (*b) (void)hipEventDestroy(*b);
        }
    } cleanup{{&iv, &cl, &k1, &k2, &d1, &d2, &dep, &res, &cnt, &tmp, &fl, &fl2}, &e0, &e1};
    int rc = 0;
    uint64_t ne = 2 * n;
    if ((rc = iv.reserve(n * sizeof(mimeo_interval))) || (rc = cl.reserve((size_t)(nchrom ? nchrom : 1) * 4)) ||
        (rc = k1.reserve(ne * 8)) || (rc = k2.reserve(ne * 8)) || (rc = d1.reserve(ne * 4)) ||
        (rc = d2.reserve(ne * 4)) || (rc = dep.reserve(ne * 4)) || (rc = res.reserve(n * sizeof(mimeo_interval))) ||
        (rc = cnt.reserve(32)) || (rc = fl.reserve(ne)) || (rc = fl2.reserve(ne)))
        return rc;
    HIP_TRY(hipEventCreate(&e0));
    HIP_TRY(hipEventCreate(&e1));
    HIP_TRY(hipMemcpyAsync(iv.p, h_iv, n * sizeof(mimeo_interval), hipMemcpyHostToDevice, st));
    if (nchrom) HIP_TRY(hipMemcpyAsync(cl.p, h_chrom_len, (size_t)nchrom * 4, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemsetAsync(cnt.p, 0, 32, st));
    HIP_TRY(hipEventRecord(e0, st));
    hipLaunchKernelGGL(k7_events, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, st, (const mimeo_interval *)iv.p, n,
                       (const uint32_t *)cl.p, nchrom, (uint64_t *)k1.p, (int32_t *)d1.p);
    // cnt: [0] regions kept, [1] distinct positions, [2] rising edges, [3] falling edges
    unsigned long long *c_out = (unsigned long long *)cnt.p, *c_nu = c_out + 1, *c_ns = c_out + 2, *c_ne = c_out + 3;
    uint64_t *ukey = (uint64_t *)k1.p;                    // k1 / d1 are free again once the sort has read them
    int32_t *udep = (int32_t *)d1.p;
    size_t t1 = 0, t2 = 0, t3 = 0, t4 = 0;
    HIP_TRY(rocprim::radix_sort_pairs(nullptr, t1, (uint64_t *)k1.p, (uint64_t *)k2.p, (int32_t *)d1.p, (int32_t *)d2.p,
                                      (size_t)ne, 0, 64, st));
    HIP_TRY(rocprim::inclusive_scan(nullptr, t2, (int32_t *)d2.p, (int32_t *)dep.p, (size_t)ne, rocprim::plus<int32_t>(), st));
    HIP_TRY(rocprim::select(nullptr, t3, (uint64_t *)k2.p, (uint8_t *)fl.p, ukey, c_nu, (size_t)ne, st));
    HIP_TRY(rocprim::select(nullptr, t4, (int32_t *)dep.p, (uint8_t *)fl.p, udep, c_nu, (size_t)ne, st));
    if ((rc = tmp.reserve(std::max(std::max(t1, t2), std::max(t3, t4)) + 16))) return rc;
    HIP_TRY(rocprim::radix_sort_pairs(tmp.p, t1, (uint64_t *)k1.p, (uint64_t *)k2.p, (int32_t *)d1.p, (int32_t *)d2.p,
                                      (size_t)ne, 0, 64, st));
    HIP_TRY(rocprim::inclusive_scan(tmp.p, t2, (int32_t *)d2.p, (int32_t *)dep.p, (size_t)ne, rocprim::plus<int32_t>(), st));
    int32_t cov = min_cov < 1 ? 1 : (int32_t)min_cov;  // genomecov -bg never reports depth 0
    const dim3 grd((uint32_t)((ne + 255) / 256)), blk(256);
    // distinct positions with the depth that holds behind them; then the rising / falling edges of "depth >= cov".
    // No thread walks a run: at min_cov 1 with the trivial self alignment a run spans every event of a scaffold.
    hipLaunchKernelGGL(k7_last_flags, grd, blk, 0, st, (const uint64_t *)k2.p, ne, (uint8_t *)fl.p);
    HIP_TRY(rocprim::select(tmp.p, t3, (uint64_t *)k2.p, (uint8_t *)fl.p, ukey, c_nu, (size_t)ne, st));
    HIP_TRY(rocprim::select(tmp.p, t4, (int32_t *)dep.p, (uint8_t *)fl.p, udep, c_nu, (size_t)ne, st));
    hipLaunchKernelGGL(k7_edge_flags, grd, blk, 0, st, (const int32_t *)udep, (const unsigned long long *)c_nu, cov,
                       (uint8_t *)fl.p, (uint8_t *)fl2.p, ne);
    uint64_t *starts = (uint64_t *)k2.p, *ends = (uint64_t *)dep.p;  // dep holds ne * 4 bytes: room for ne / 2 ends (>= one per interval)
    HIP_TRY(rocprim::select(tmp.p, t3, ukey, (uint8_t *)fl.p, starts, c_ns, (size_t)ne, st));
    HIP_TRY(rocprim::select(tmp.p, t3, ukey, (uint8_t *)fl2.p, ends, c_ne, (size_t)ne, st));
    hipLaunchKernelGGL(k7_regions, grd, blk, 0, st, (const uint64_t *)starts, (const uint64_t *)ends,
                       (const unsigned long long *)c_ns, min_len, (mimeo_interval *)res.p, c_out, ne);
    HIP_TRY(hipEventRecord(e1, st));
    unsigned long long m = 0;
    HIP_TRY(hipMemcpyAsync(&m, cnt.p, 8, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    HIP_TRY(hipGetLastError());
    out.resize(m);
    if (m) HIP_TRY(hipMemcpy(out.data(), res.p, m * sizeof(mimeo_interval), hipMemcpyDeviceToHost));
    std::sort(out.begin(), out.end(), [](const mimeo_interval &a, const mimeo_interval &b) {
        return a.chrom != b.chrom ? a.chrom < b.chrom : a.start < b.start;
    });
    if (ms) {
        float t = 0;
        HIP_TRY(hipEventElapsedTime(&t, e0, e1));
        *ms += t;
    }
    return 0;
}


// ---- bedtools genomecov -bg (reference src/mimeo/wrappers.py:1131-1145): maximal runs of equal depth > 0 ----------------
// The same events, sort and scan; a run begins at every distinct position where the depth differs from the depth in
// front of it (equal depths on either side of a position — one interval ends where another begins — are one run, as
// genomecov reports per-base depth), and ends at the next such position.
__global__ void k7_change_flags(const int32_t *__restrict__ udep, const unsigned long long *__restrict__ nu_dev, uint8_t *__restrict__ flag, uint64_t cap) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= cap) return;
    const uint64_t nu = *nu_dev;
    flag[i] = (i < nu && udep[i] != (i ? udep[i - 1] : 0)) ? 1 : 0;
}
__global__ void k7_runs(const uint64_t *__restrict__ bkey, const int32_t *__restrict__ bdep, const unsigned long long *__restrict__ nb_dev,
                        mimeo_depth_run *__restrict__ out, uint8_t *__restrict__ keep, uint64_t cap) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= cap) return;
    const uint64_t nb = *nb_dev;
    const bool k = i + 1 < nb && bdep[i] > 0;   // the last change of all goes down to depth 0
    keep[i] = k ? 1 : 0;
    if (k) out[i] = mimeo_depth_run{(uint32_t)(bkey[i] >> 32), (uint32_t)bkey[i], (uint32_t)bkey[i + 1], (uint32_t)bdep[i]};
}

int coverage_bedgraph_device(const mimeo_interval *h_iv, uint64_t n, const uint32_t *h_chrom_len, uint32_t nchrom,
                             std::vector<mimeo_depth_run> &out) {
    out.clear();
    if (!n) return 0;
    hipStream_t st = stream();
    DeviceBuf iv, cl, k1, k2, d1, d2, dep, rows, rows2, cnt, tmp, fl;
    struct Cleanup {
        DeviceBuf *bufs[12];
        ~Cleanup() { for (DeviceBuf *x : bufs) x->release(); }
    } cleanup{{&iv, &cl, &k1, &k2, &d1, &d2, &dep, &rows, &rows2, &cnt, &tmp, &fl}};
    int rc = 0;
    const uint64_t ne = 2 * n;
    if ((rc = iv.reserve(n * sizeof(mimeo_interval))) || (rc = cl.reserve((size_t)(nchrom ? nchrom : 1) * 4)) ||
        (rc = k1.reserve(ne * 8)) || (rc = k2.reserve(ne * 8)) || (rc = d1.reserve(ne * 4)) || (rc = d2.reserve(ne * 4)) ||
        (rc = dep.reserve(ne * 4)) || (rc = rows.reserve(ne * sizeof(mimeo_depth_run))) || (rc = rows2.reserve(ne * sizeof(mimeo_depth_run))) ||
        (rc = cnt.reserve(32)) || (rc = fl.reserve(ne)))
        return rc;
    HIP_TRY(hipMemcpyAsync(iv.p, h_iv, n * sizeof(mimeo_interval), hipMemcpyHostToDevice, st));
    if (nchrom) HIP_TRY(hipMemcpyAsync(cl.p, h_chrom_len, (size_t)nchrom * 4, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemsetAsync(cnt.p, 0, 32, st));
    hipLaunchKernelGGL(k7_events, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, st, (const mimeo_interval *)iv.p, n,
                       (const uint32_t *)cl.p, nchrom, (uint64_t *)k1.p, (int32_t *)d1.p);
    unsigned long long *c_nu = (unsigned long long *)cnt.p, *c_nb = c_nu + 1, *c_rows = c_nu + 2;
    uint64_t *ukey = (uint64_t *)k1.p;
    int32_t *udep = (int32_t *)d1.p;
    size_t t1 = 0, t2 = 0, t3 = 0, t4 = 0, t5 = 0;
    HIP_TRY(rocprim::radix_sort_pairs(nullptr, t1, (uint64_t *)k1.p, (uint64_t *)k2.p, (int32_t *)d1.p, (int32_t *)d2.p, (size_t)ne, 0, 64, st));
    HIP_TRY(rocprim::inclusive_scan(nullptr, t2, (int32_t *)d2.p, (int32_t *)dep.p, (size_t)ne, rocprim::plus<int32_t>(), st));
    HIP_TRY(rocprim::select(nullptr, t3, (uint64_t *)k2.p, (uint8_t *)fl.p, ukey, c_nu, (size_t)ne, st));
    HIP_TRY(rocprim::select(nullptr, t4, (int32_t *)dep.p, (uint8_t *)fl.p, udep, c_nu, (size_t)ne, st));
    HIP_TRY(rocprim::select(nullptr, t5, (mimeo_depth_run *)rows.p, (uint8_t *)fl.p, (mimeo_depth_run *)rows2.p, c_rows, (size_t)ne, st));
    if ((rc = tmp.reserve(std::max(std::max(std::max(t1, t2), std::max(t3, t4)), t5) + 16))) return rc;
    HIP_TRY(rocprim::radix_sort_pairs(tmp.p, t1, (uint64_t *)k1.p, (uint64_t *)k2.p, (int32_t *)d1.p, (int32_t *)d2.p, (size_t)ne, 0, 64, st));
    HIP_TRY(rocprim::inclusive_scan(tmp.p, t2, (int32_t *)d2.p, (int32_t *)dep.p, (size_t)ne, rocprim::plus<int32_t>(), st));
    const dim3 grd((uint32_t)((ne + 255) / 256)), blk(256);
    hipLaunchKernelGGL(k7_last_flags, grd, blk, 0, st, (const uint64_t *)k2.p, ne, (uint8_t *)fl.p);
    HIP_TRY(rocprim::select(tmp.p, t3, (uint64_t *)k2.p, (uint8_t *)fl.p, ukey, c_nu, (size_t)ne, st));
    HIP_TRY(rocprim::select(tmp.p, t4, (int32_t *)dep.p, (uint8_t *)fl.p, udep, c_nu, (size_t)ne, st));
    // positions where the depth changes, with the depth behind them
    hipLaunchKernelGGL(k7_change_flags, grd, blk, 0, st, (const int32_t *)udep, (const unsigned long long *)c_nu, (uint8_t *)fl.p, ne);
    uint64_t *bkey = (uint64_t *)k2.p;
    int32_t *bdep = (int32_t *)dep.p;
    HIP_TRY(rocprim::select(tmp.p, t3, ukey, (uint8_t *)fl.p, bkey, c_nb, (size_t)ne, st));
    HIP_TRY(rocprim::select(tmp.p, t4, udep, (uint8_t *)fl.p, bdep, c_nb, (size_t)ne, st));
    hipLaunchKernelGGL(k7_runs, grd, blk, 0, st, (const uint64_t *)bkey, (const int32_t *)bdep, (const unsigned long long *)c_nb,
                       (mimeo_depth_run *)rows.p, (uint8_t *)fl.p, ne);
    HIP_TRY(rocprim::select(tmp.p, t5, (mimeo_depth_run *)rows.p, (uint8_t *)fl.p, (mimeo_depth_run *)rows2.p, c_rows, (size_t)ne, st));
    unsigned long long m = 0;
    HIP_TRY(hipMemcpyAsync(&m, c_rows, 8, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    HIP_TRY(hipGetLastError());
    out.resize(m);   // in (chrom, start) order: the select keeps the order of the sorted positions
    if (m) HIP_TRY(hipMemcpy(out.data(), rows2.p, m * sizeof(mimeo_depth_run), hipMemcpyDeviceToHost));
    return 0;
}

}  // namespace mimeo
