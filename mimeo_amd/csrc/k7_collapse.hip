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

__global__ void k7_regions(const uint64_t *__restrict__ key, const int32_t *__restrict__ depth, uint64_t n,
                           int32_t min_cov, uint32_t min_len, mimeo_interval *__restrict__ out,
                           unsigned long long *__restrict__ nout) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint64_t k = key[i];
    if (k == ~0ull) return;
    bool last = (i + 1 == n) || key[i + 1] != k;  // depth[i] holds from this position to the next one
    if (!last || depth[i] < min_cov) return;
    uint64_t g = i;  // first event of my position
    while (g > 0 && key[g - 1] == k) g--;
    if (g > 0 && depth[g - 1] >= min_cov) return;  // the run started earlier
    uint64_t j = i + 1;
    while (j < n) {
        bool lj = (j + 1 == n) || key[j + 1] != key[j];
        if (lj && depth[j] < min_cov) break;
        j++;
    }
    if (j >= n) return;  // cannot happen: depth returns to 0 at the last event of a chromosome
    uint32_t s = (uint32_t)k, e = (uint32_t)key[j];
    if (e - s >= min_len) {
        unsigned long long o = atomicAdd(nout, 1ull);
        out[o] = mimeo_interval{(uint32_t)(k >> 32), s, e};
    }
}

int coverage_collapse_device(const mimeo_interval *h_iv, uint64_t n, const uint32_t *h_chrom_len, uint32_t nchrom,
                             uint32_t min_cov, uint32_t min_len, std::vector<mimeo_interval> &out, float *ms) {
    out.clear();
    if (!n) return 0;
    hipStream_t st = stream();
    DeviceBuf iv, cl, k1, k2, d1, d2, dep, res, cnt, tmp;
    int rc = 0;
    uint64_t ne = 2 * n;
    if ((rc = iv.reserve(n * sizeof(mimeo_interval))) || (rc = cl.reserve((size_t)(nchrom ? nchrom : 1) * 4)) ||
        (rc = k1.reserve(ne * 8)) || (rc = k2.reserve(ne * 8)) || (rc = d1.reserve(ne * 4)) ||
        (rc = d2.reserve(ne * 4)) || (rc = dep.reserve(ne * 4)) || (rc = res.reserve(n * sizeof(mimeo_interval))) ||
        (rc = cnt.reserve(8)))
        return rc;
    hipEvent_t e0, e1;
    HIP_TRY(hipEventCreate(&e0));
    HIP_TRY(hipEventCreate(&e1));
    HIP_TRY(hipMemcpyAsync(iv.p, h_iv, n * sizeof(mimeo_interval), hipMemcpyHostToDevice, st));
    if (nchrom) HIP_TRY(hipMemcpyAsync(cl.p, h_chrom_len, (size_t)nchrom * 4, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemsetAsync(cnt.p, 0, 8, st));
    HIP_TRY(hipEventRecord(e0, st));
    hipLaunchKernelGGL(k7_events, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, st, (const mimeo_interval *)iv.p, n,
                       (const uint32_t *)cl.p, nchrom, (uint64_t *)k1.p, (int32_t *)d1.p);
    size_t t1 = 0, t2 = 0;
    HIP_TRY(rocprim::radix_sort_pairs(nullptr, t1, (uint64_t *)k1.p, (uint64_t *)k2.p, (int32_t *)d1.p, (int32_t *)d2.p,
                                      (size_t)ne, 0, 64, st));
    HIP_TRY(rocprim::inclusive_scan(nullptr, t2, (int32_t *)d2.p, (int32_t *)dep.p, (size_t)ne, rocprim::plus<int32_t>(), st));
    if ((rc = tmp.reserve(std::max(t1, t2) + 16))) return rc;
    HIP_TRY(rocprim::radix_sort_pairs(tmp.p, t1, (uint64_t *)k1.p, (uint64_t *)k2.p, (int32_t *)d1.p, (int32_t *)d2.p,
                                      (size_t)ne, 0, 64, st));
    HIP_TRY(rocprim::inclusive_scan(tmp.p, t2, (int32_t *)d2.p, (int32_t *)dep.p, (size_t)ne, rocprim::plus<int32_t>(), st));
    int32_t cov = min_cov < 1 ? 1 : (int32_t)min_cov;  // genomecov -bg never reports depth 0
    hipLaunchKernelGGL(k7_regions, dim3((uint32_t)((ne + 255) / 256)), dim3(256), 0, st, (const uint64_t *)k2.p,
                       (const int32_t *)dep.p, ne, cov, min_len, (mimeo_interval *)res.p, (unsigned long long *)cnt.p);
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
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    for (DeviceBuf *b : {&iv, &cl, &k1, &k2, &d1, &d2, &dep, &res, &cnt, &tmp}) b->release();
    return 0;
}

}  // namespace mimeo
