// K2 — 12of19 seed index of one scaffold strand (lastz's target seed-word
// position table, --step=1; reference call site src/mimeo/wrappers.py:1028-1031).
//
// Layout: CSR  off[2^24+1], pos[n]  keyed by  (pext12(lo) << 12) | pext12(hi)  (common.h).
// Built once per scaffold strand and reused against every partner scaffold — the reference
// rebuilds lastz's table in each of its S^2 invocations.
//
// The (key,pos) ordering uses rocPRIM's device radix sort: this is plumbing executed 2*S times
// per job against S^2*2 seed scans, not a hot kernel.
#include <cstring>
#include <mutex>
#include <utility>
#include <vector>

#include <rocprim/rocprim.hpp>

#include "common.h"

namespace mimeo {

// 12 care bits of the 19-bit window: offsets {0,1,2,4,7,8,11,13,15,16,17,18}
__device__ __forceinline__ uint32_t pext12(uint32_t x) {
    return (x & 0x7u) | ((x >> 1) & 0x8u) | ((x >> 3) & 0x30u) | ((x >> 5) & 0x40u) | ((x >> 6) & 0x80u) |
           ((x >> 7) & 0xF00u);
}

// one thread per 32 start positions: keys (1<<24 for "no seed here") + histogram
__global__ void k2_seed_keys(StrandView s, uint32_t nwords, uint32_t *__restrict__ keys,
                             uint32_t *__restrict__ posv, uint32_t *__restrict__ hist, uint32_t p0, uint32_t p1) {
    uint32_t w = blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= nwords) return;
    const uint4 a = s.pw[w], c = s.pw[w + 1];
    uint64_t lo = (uint64_t)a.x | ((uint64_t)c.x << 32);
    uint64_t hi = (uint64_t)a.y | ((uint64_t)c.y << 32);
    uint32_t sv = s.svt ? s.svt[w] : a.w;
    uint32_t base = w * 32u;
#pragma unroll 4
    for (uint32_t b = 0; b < 32; b++) {
        uint32_t p = base + b;
        if (p >= s.len) break;
        uint32_t key = NBUCKET;
        if (((sv >> b) & 1u) && p >= p0 && p < p1) {  // [p0, p1): the whole strand, or one chunk of a very large query
            uint32_t wl = (uint32_t)(lo >> b) & 0x7FFFFu, wh = (uint32_t)(hi >> b) & 0x7FFFFu;
            key = (pext12(wl) << 12) | pext12(wh);
            atomicAdd(&hist[key], 1u);
        }
        keys[p] = key;
        posv[p] = p;
    }
}

// Device allocations are slow (tens to hundreds of microseconds each) and an index build used to do
// a dozen of them: the temporaries live in a grow-only workspace, and released off/pos arrays go to
// a small free list that the next build of the same size picks up.
static DeviceBuf g_hist, g_keys_in, g_keys_out, g_pos_in, g_tmp;
static std::vector<std::pair<size_t, void *>> g_free_list;
// The free list has a lock of its own: SeedIndex::release() runs on any thread (a lane giving a chunk index
// back, the end of a call) while the index builder thread may be inside build_index() taking a buffer.  It is
// not build_index's mutex: release() is also called with that one held.
static std::mutex g_free_mu;

static int pool_alloc(void **p, size_t bytes) {
    {
        std::lock_guard<std::mutex> lk(g_free_mu);
        for (size_t i = 0; i < g_free_list.size(); i++)
            if (g_free_list[i].first == bytes) {
                *p = g_free_list[i].second;
                g_free_list.erase(g_free_list.begin() + i);
                return 0;
            }
    }
    HIP_TRY(hipMalloc(p, bytes));
    return 0;
}
static void pool_free(void *p, size_t bytes) {
    if (!p) return;
    {
        std::lock_guard<std::mutex> lk(g_free_mu);
        if (g_free_list.size() < 512) { g_free_list.emplace_back(bytes, p); return; }
    }
    (void)hipFree(p);
}

void SeedIndex::release() {
    pool_free(off, ((size_t)NBUCKET + 2) * 4);
    pool_free(pos, pos_bytes);
    off = pos = nullptr;
    n = 0;
    pos_bytes = 0;
}

int build_index(const StrandView &s, SeedIndex &out, float *ms, uint32_t p0, uint32_t p1) {
    // one build at a time: the workspace and the event pair are shared (index builder thread, lanes joining a chunked unit)
    static std::mutex mu;
    std::lock_guard<std::mutex> lk(mu);
    out.release();
    uint32_t len = s.len;
    uint32_t nwords = (len + 31) / 32;
    static hipEvent_t e0 = nullptr, e1 = nullptr;
    if (!e0) { HIP_TRY(hipEventCreate(&e0)); HIP_TRY(hipEventCreate(&e1)); }
    HIP_TRY(hipEventRecord(e0, stream()));
    size_t n = len ? len : 1;
    int rc;
    if ((rc = g_hist.reserve(((size_t)NBUCKET + 2) * 4)) || (rc = g_keys_in.reserve(n * 4)) ||
        (rc = g_keys_out.reserve(n * 4)) || (rc = g_pos_in.reserve(n * 4)))
        return rc;
    uint32_t *hist = (uint32_t *)g_hist.p, *keys_in = (uint32_t *)g_keys_in.p, *keys_out = (uint32_t *)g_keys_out.p,
             *pos_in = (uint32_t *)g_pos_in.p;
    HIP_TRY(hipMemsetAsync(hist, 0, ((size_t)NBUCKET + 2) * 4, stream()));
    if ((rc = pool_alloc((void **)&out.off, ((size_t)NBUCKET + 2) * 4))) return rc;
    out.pos_bytes = n * 4;
    if ((rc = pool_alloc((void **)&out.pos, out.pos_bytes))) return rc;
    if (nwords)
        hipLaunchKernelGGL(k2_seed_keys, dim3((nwords + 255) / 256), dim3(256), 0, stream(), s, nwords, keys_in,
                           pos_in, hist, p0, p1);
    // off = exclusive scan of hist over NBUCKET+1 entries (entry NBUCKET = total)
    size_t tmp_bytes = 0, tmp2 = 0;
    HIP_TRY(rocprim::exclusive_scan(nullptr, tmp_bytes, hist, out.off, 0u, (size_t)NBUCKET + 1, rocprim::plus<uint32_t>(),
                                    stream()));
    if (len)
        HIP_TRY(rocprim::radix_sort_pairs(nullptr, tmp2, keys_in, keys_out, pos_in, out.pos, (size_t)len, 0, 25,
                                          stream()));
    if (tmp2 > tmp_bytes) tmp_bytes = tmp2;
    if ((rc = g_tmp.reserve(tmp_bytes + 16))) return rc;
    void *tmp = g_tmp.p;
    HIP_TRY(rocprim::exclusive_scan(tmp, tmp_bytes, hist, out.off, 0u, (size_t)NBUCKET + 1, rocprim::plus<uint32_t>(),
                                    stream()));
    if (len)
        HIP_TRY(rocprim::radix_sort_pairs(tmp, tmp_bytes, keys_in, keys_out, pos_in, out.pos, (size_t)len, 0, 25,
                                          stream()));
    uint32_t total = 0;
    HIP_TRY(hipMemcpyAsync(&total, out.off + NBUCKET, 4, hipMemcpyDeviceToHost, stream()));
    HIP_TRY(hipEventRecord(e1, stream()));
    HIP_TRY(hipStreamSynchronize(stream()));
    out.n = total;
    if (ms) {
        float t = 0;
        HIP_TRY(hipEventElapsedTime(&t, e0, e1));
        *ms += t;
    }
    HIP_TRY(hipGetLastError());
    return 0;
}

}  // namespace mimeo
