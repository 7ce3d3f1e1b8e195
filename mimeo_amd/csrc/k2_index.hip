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

#include "device_util.h"

namespace mimeo {

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

// Seed frames (common.h): one thread per index entry gathers the seven words around its seed start from the
// two-plane copy, brings them into the frame alignment (bit 0 = base p - FRAME_LEFT) and stores the twelve words;
// an N anywhere in the frame sets bit 31 of pos[].  Once per strand and job: the fused seed-scan kernel then
// streams the frames of both sides of every seed hit from contiguous memory instead of gathering 13 words per HIT.
__global__ void k2_frames(StrandView s, uint32_t *__restrict__ pos, uint32_t n, uint4 *__restrict__ fr, uint32_t stride) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t p = pos[i] & POS_MASK;
    const int32_t b0 = (int32_t)p - FRAME_LEFT;   // may be negative: PLANE_PAD zero words lie in front of word 0
    const int32_t w = b0 >> 5;
    const uint32_t sh = (uint32_t)b0 & 31u;
    uint32_t lo[FRAME_WORDS + 1], hi[FRAME_WORDS + 1], nm = 0;
    if (s.has_n) {
        uint32_t z[FRAME_WORDS + 1];
#pragma unroll
        for (int k = 0; k <= FRAME_WORDS; k++) { const uint4 v = s.pw[w + k]; lo[k] = v.x; hi[k] = v.y; z[k] = v.z; }
#pragma unroll
        for (int k = 0; k < FRAME_WORDS; k++) nm |= __builtin_amdgcn_alignbit(z[k + 1], z[k], sh);
    } else {
#pragma unroll
        for (int k = 0; k <= FRAME_WORDS; k++) { const uint2 v = s.p2[w + k]; lo[k] = v.x; hi[k] = v.y; }
    }
    uint32_t fl[FRAME_WORDS], fh[FRAME_WORDS];
#pragma unroll
    for (int k = 0; k < FRAME_WORDS; k++) {
        fl[k] = __builtin_amdgcn_alignbit(lo[k + 1], lo[k], sh);
        fh[k] = __builtin_amdgcn_alignbit(hi[k + 1], hi[k], sh);
    }
    fr[i] = make_uint4(fl[0], fl[1], fl[2], fl[3]);
    fr[(size_t)stride + i] = make_uint4(fl[4], fl[5], fh[0], fh[1]);
    fr[2 * (size_t)stride + i] = make_uint4(fh[2], fh[3], fh[4], fh[5]);
    if (nm) pos[i] = p | POS_NFLAG;
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
    pool_free(fr, fr_bytes);
    off = pos = nullptr;
    fr = nullptr;
    n = fr_stride = 0;
    pos_bytes = fr_bytes = 0;
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
    out.fr_bytes = n * 48;
    out.fr_stride = (uint32_t)n;
    if ((rc = pool_alloc((void **)&out.fr, out.fr_bytes))) return rc;
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
    // the entries behind `total` (positions without a seed word, key 2^24) are never read; their frames are built
    // too rather than paying a host round trip for the count first
    if (len) hipLaunchKernelGGL(k2_frames, dim3((len + 255) / 256), dim3(256), 0, stream(), s, out.pos, len, out.fr, out.fr_stride);
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
