// K3 — seed scan as an index join (lastz's query scan + table probes with the
// default one-transition tolerance; reference call site src/mimeo/wrappers.py:1031,
// `--step=1 --strand=both`).
//
// lastz walks the query and probes a 2^24-entry hash table 13 times per base: 104 random
// bytes per query base.  Here BOTH sides are CSR seed indexes in the transition-closed key
// layout of common.h, so the 13 neighbours of a word live in the same 4096-key tile: one
// workgroup streams the two 16 KiB offset slices of a tile into LDS with coalesced 16-byte
// loads and resolves every probe from LDS.  HBM traffic per (target, query-strand) unit is the
// two offset arrays + the position lists once + the hit records.
//
// Launches: k3_join_count (per-tile hit counts) -> k3_tile_scan -> k3_join_fill.  The fill kernel
// places hits with a workgroup prefix sum, so the output order is deterministic:
// (key, target position, neighbour, query position).
#include <chrono>

#include "common.h"

namespace mimeo {

constexpr int JOIN_THREADS = 256;

__device__ __forceinline__ void load_tile_offsets(const uint32_t *__restrict__ off, uint32_t tile,
                                                  uint32_t *s) {
    const uint4 *src = reinterpret_cast<const uint4 *>(off + (size_t)tile * TILE_WORDS);
    uint4 *dst = reinterpret_cast<uint4 *>(s);
#pragma unroll
    for (int k = 0; k < (int)(TILE_WORDS / 4 / JOIN_THREADS); k++) dst[k * JOIN_THREADS + threadIdx.x] = src[k * JOIN_THREADS + threadIdx.x];
    if (threadIdx.x == 0) s[TILE_WORDS] = off[(size_t)tile * TILE_WORDS + TILE_WORDS];
}

// same sum, plus the set of non-empty neighbours: bit 0 = w itself, bit j + 1 = w ^ (1 << j)
__device__ __forceinline__ uint32_t neighbour_sum_mask(const uint32_t *sQ, uint32_t w, int transitions, uint32_t &mask) {
    uint32_t sum = sQ[w + 1] - sQ[w];
    mask = sum ? 1u : 0u;
    if (transitions) {
#pragma unroll
        for (int j = 0; j < SEED_WEIGHT; j++) {
            const uint32_t w2 = w ^ (1u << j);
            const uint32_t c = sQ[w2 + 1] - sQ[w2];
            sum += c;
            mask |= c ? (2u << j) : 0u;
        }
    }
    return sum;
}

__device__ __forceinline__ uint32_t neighbour_sum(const uint32_t *sQ, uint32_t w, int transitions) {
    uint32_t sum = sQ[w + 1] - sQ[w];
    if (transitions) {
#pragma unroll
        for (int j = 0; j < SEED_WEIGHT; j++) {
            uint32_t w2 = w ^ (1u << j);
            sum += sQ[w2 + 1] - sQ[w2];
        }
    }
    return sum;
}

__global__ __launch_bounds__(JOIN_THREADS) void k3_join_count(const uint32_t *__restrict__ offT,
                                                              const uint32_t *__restrict__ offQ, int transitions,
                                                              unsigned long long *__restrict__ tile_count) {
    __shared__ __attribute__((aligned(16))) uint32_t sT[TILE_WORDS + 4];
    __shared__ __attribute__((aligned(16))) uint32_t sQ[TILE_WORDS + 4];
    __shared__ unsigned long long red[JOIN_THREADS / 64];
    uint32_t tile = blockIdx.x;
    load_tile_offsets(offT, tile, sT);
    load_tile_offsets(offQ, tile, sQ);
    __syncthreads();
    unsigned long long cnt = 0;
#pragma unroll 4
    for (uint32_t k = 0; k < TILE_WORDS / JOIN_THREADS; k++) {
        uint32_t w = k * JOIN_THREADS + threadIdx.x;
        uint32_t nT = sT[w + 1] - sT[w];
        if (nT) cnt += (unsigned long long)nT * neighbour_sum(sQ, w, transitions);
    }
    for (int o = 32; o > 0; o >>= 1) cnt += __shfl_down(cnt, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = cnt;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long t = 0;
        for (int i = 0; i < JOIN_THREADS / 64; i++) t += red[i];
        tile_count[tile] = t;
    }
}

// single workgroup: exclusive scan of the NTILE tile counts; tile_base[NTILE] = total
__global__ __launch_bounds__(1024) void k3_tile_scan(const unsigned long long *__restrict__ cnt,
                                                     unsigned long long *__restrict__ base) {
    __shared__ unsigned long long part[1024];
    constexpr int PER = NTILE / 1024;
    unsigned long long loc[PER], s = 0;
    for (int i = 0; i < PER; i++) { loc[i] = s; s += cnt[threadIdx.x * PER + i]; }
    part[threadIdx.x] = s;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {
        unsigned long long v = threadIdx.x >= (unsigned)o ? part[threadIdx.x - o] : 0;
        __syncthreads();
        part[threadIdx.x] += v;
        __syncthreads();
    }
    unsigned long long excl = part[threadIdx.x] - s;
    for (int i = 0; i < PER; i++) base[threadIdx.x * PER + i] = excl + loc[i];
    if (threadIdx.x == 1023) base[NTILE] = part[1023];
}

// Fill: one thread per TARGET ENTRY of the tile (not per key): entry g of the CSR position list
// finds its key by a binary search over the LDS-resident offsets, counts the query entries of its
// 13 neighbour keys, and a workgroup prefix sum gives it a contiguous slot range in the hit
// array, which it fills directly (about four 8-byte records = one 32-byte sector per entry on
// random sequence).  The tile's query positions are one contiguous CSR slice; it is staged in
// LDS so that the emission loop's dependent reads cost an LDS round trip instead of an L2 one.
// Measured and rejected: staging the hits in LDS for fully coalesced stores, block-wide (0.233 vs
// 0.187 ms per 5 Mbp x 5 Mbp unit) or per wavefront (0.262 ms) — the stage costs occupancy and
// barriers and the 32-byte runs already store efficiently.
constexpr int FILL_THREADS = 512;
constexpr uint32_t FILL_QCACHE = 4096;       // query positions of a tile kept in LDS: at most (16 KiB) ...
constexpr uint32_t FILL_QCACHE_SMALL = 1984;  // ... or 7.75 KiB, which lets four workgroups share a CU instead of three

__device__ __forceinline__ void load_tile_offsets_n(const uint32_t *__restrict__ off, uint32_t tile, uint32_t *s,
                                                    int nthreads) {
    const uint4 *src = reinterpret_cast<const uint4 *>(off + (size_t)tile * TILE_WORDS);
    uint4 *dst = reinterpret_cast<uint4 *>(s);
    for (uint32_t k = threadIdx.x; k < TILE_WORDS / 4; k += nthreads) dst[k] = src[k];
    if (threadIdx.x == 0) s[TILE_WORDS] = off[(size_t)tile * TILE_WORDS + TILE_WORDS];
}

__global__ __launch_bounds__(FILL_THREADS) void k3_join_fill(const uint32_t *__restrict__ offT,
                                                             const uint32_t *__restrict__ posT,
                                                             const uint32_t *__restrict__ offQ,
                                                             const uint32_t *__restrict__ posQ, int transitions,
                                                             const unsigned long long *__restrict__ tile_base,
                                                             uint2 *__restrict__ hits, unsigned long long cap,
                                                             uint32_t qcache_n) {
    __shared__ __attribute__((aligned(16))) uint32_t sT[TILE_WORDS + 4];
    __shared__ __attribute__((aligned(16))) uint32_t sQ[TILE_WORDS + 4];
    extern __shared__ uint32_t sPQ[];  // qcache_n entries (dynamic: the size decides the occupancy)
    __shared__ uint32_t wsum[2][FILL_THREADS / 64];
    const uint32_t tile = blockIdx.x;
    if (tile_base[NTILE] > cap) return;  // speculative launch whose buffer is too small: the host redoes the unit
    unsigned long long out = tile_base[tile];
    if (tile_base[tile + 1] == out) return;  // empty tile
    load_tile_offsets_n(offT, tile, sT, FILL_THREADS);
    load_tile_offsets_n(offQ, tile, sQ, FILL_THREADS);
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t t0 = sT[0], nT = sT[TILE_WORDS] - t0;
    const uint32_t q0t = sQ[0], nQ = sQ[TILE_WORDS] - q0t;
    const bool qcached = nQ <= qcache_n;
    if (qcached) {
        for (uint32_t i = threadIdx.x; i < nQ; i += FILL_THREADS) sPQ[i] = posQ[q0t + i] & POS_MASK;
        __syncthreads();
    }
    uint32_t par = 0;
    for (uint32_t e0 = 0; e0 < nT; e0 += FILL_THREADS, par ^= 1u) {
        const uint32_t e = e0 + threadIdx.x;
        const uint32_t g = t0 + e;
        uint32_t w = 0, c = 0, nmask = 0;
        if (e < nT) {
            uint32_t lo = 0, hi = TILE_WORDS;  // largest w with sT[w] <= g
            while (hi - lo > 1) {
                uint32_t mid = (lo + hi) >> 1;
                if (sT[mid] <= g) lo = mid; else hi = mid;
            }
            w = lo;
            c = neighbour_sum_mask(sQ, w, transitions, nmask);
        }
        uint32_t inc = c;
        for (int o = 1; o < 64; o <<= 1) {
            uint32_t v = __shfl_up(inc, o);
            if (lane >= (uint32_t)o) inc += v;
        }
        if (lane == 63) wsum[par][wave] = inc;
        __syncthreads();  // the other wsum buffer is free again: one barrier per pass
        uint32_t wbase = 0, total = 0;
#pragma unroll
        for (int i = 0; i < FILL_THREADS / 64; i++) {
            uint32_t v = wsum[par][i];
            if ((uint32_t)i < wave) wbase += v;
            total += v;
        }
        if (c) {
            const uint32_t tp = posT[g] & POS_MASK;  // bit 31 flags a frame with an N (K34)
            uint2 *dst = hits + out + (wbase + inc - c);
            // only the non-empty neighbours (3-4 of the 13 on random sequence), in the same order as before
            for (uint32_t m = nmask; m; m &= m - 1u) {
                const uint32_t jj = (uint32_t)__builtin_ctz(m);
                const uint32_t w2 = jj ? (w ^ (1u << (jj - 1u))) : w;
                const uint32_t q0 = sQ[w2], q1 = sQ[w2 + 1];
                if (qcached)
                    for (uint32_t b = q0; b < q1; b++) *dst++ = make_uint2(tp, sPQ[b - q0t]);
                else
                    for (uint32_t b = q0; b < q1; b++) *dst++ = make_uint2(tp, posQ[b] & POS_MASK);
            }
        }
        out += total;
    }
}

double g_alloc_ms = 0;   // host time spent in hipMalloc / hipFree of work buffers (MIMEO_TRACE prints it per call)
struct AllocTimer {
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    ~AllocTimer() { g_alloc_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); }
};
int DeviceBuf::reserve(size_t bytes) {
    if (bytes <= cap) return 0;
    if (view) { set_error("internal: a slice of an arena cannot grow"); return MIMEO_ERR_ARG; }
    AllocTimer timer;
    if (p) HIP_TRY(hipFree(p));
    p = nullptr;
    cap = 0;
    size_t want = bytes + bytes / 4 + 4096;
    hipError_t e = hipMalloc(&p, want);
    if (e == hipErrorOutOfMemory && want > bytes + 4096) {   // no room for the slack: the bytes asked for will do
        (void)hipGetLastError();
        want = bytes + 4096;
        e = hipMalloc(&p, want);
    }
    if (e != hipSuccess) {
        p = nullptr;
        size_t free_b = 0, total_b = 0;
        (void)hipMemGetInfo(&free_b, &total_b);
        char what[160];
        snprintf(what, sizeof what, "hipMalloc of %.2f GiB (device memory free: %.2f of %.2f GiB)", (double)want / (1 << 30), (double)free_b / (1 << 30),
                 (double)total_b / (1 << 30));
        return hip_fail(e, what, __FILE__, __LINE__);
    }
    cap = want;
    return 0;
}
void DeviceBuf::release() {
    if (view) { p = nullptr; cap = 0; view = false; return; }
    AllocTimer timer;
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
}

void JoinCtx::release() {
    if (tile_count) (void)hipFree(tile_count);
    if (tile_base) (void)hipFree(tile_base);
    tile_count = tile_base = nullptr;
    for (auto &e : ev) { if (e) (void)hipEventDestroy(e); e = nullptr; }
}

// exact mode only (stage entry point mimeo_seed_hits and the A/B heavy path): one host round trip for the hit count
int join_hits(JoinCtx &ctx, const IndexView &T, const IndexView &Q, int transitions, DeviceBuf &hits, uint64_t *nhits,
              JoinTiming *tm) {
    if (!ctx.tile_count) {
        HIP_TRY(hipMalloc((void **)&ctx.tile_count, (NTILE + 2) * sizeof(unsigned long long)));
        HIP_TRY(hipMalloc((void **)&ctx.tile_base, (NTILE + 2) * sizeof(unsigned long long)));
    }
    if (!ctx.ev[0])
        for (auto &e : ctx.ev) HIP_TRY(hipEventCreate(&e));
    hipStream_t st = stream();
    HIP_TRY(hipEventRecord(ctx.ev[0], st));
    hipLaunchKernelGGL(k3_join_count, dim3(NTILE), dim3(JOIN_THREADS), 0, st, T.off, Q.off, transitions, ctx.tile_count);
    hipLaunchKernelGGL(k3_tile_scan, dim3(1), dim3(1024), 0, st, ctx.tile_count, ctx.tile_base);
    HIP_TRY(hipEventRecord(ctx.ev[1], st));
    unsigned long long total = 0;
    HIP_TRY(hipMemcpyAsync(&total, ctx.tile_base + NTILE, sizeof total, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    *nhits = total;
    if (total >= (1ull << 32)) {
        set_error("one (target, query, strand) unit yields 2^32 or more seed hits: the stand-alone seed scan cannot hold them (the fused path has no such limit)");
        return MIMEO_ERR_LIMIT;
    }
    int rc = hits.reserve((size_t)(total ? total : 1) * sizeof(uint2));
    if (rc) return rc;
    HIP_TRY(hipEventRecord(ctx.ev[2], st));
    {
        // the small cache when the average tile's query slice fits it with room to spare (tiles beyond it read
        // the positions from L2 instead): 4 instead of 3 workgroups per CU, 0.157 -> 0.145 ms on a C2 unit
        const uint32_t qn = ((uint64_t)Q.n * 3 / 2) / NTILE <= FILL_QCACHE_SMALL ? FILL_QCACHE_SMALL : FILL_QCACHE;
        if (total)
            hipLaunchKernelGGL(k3_join_fill, dim3(NTILE), dim3(FILL_THREADS), qn * sizeof(uint32_t), st, T.off, T.pos, Q.off, Q.pos,
                               transitions, ctx.tile_base, (uint2 *)hits.p, ~0ull, qn);
    }
    HIP_TRY(hipEventRecord(ctx.ev[3], st));
    HIP_TRY(hipGetLastError());
    if (tm) {
        HIP_TRY(hipEventSynchronize(ctx.ev[3]));
        float a = 0, b2 = 0;
        HIP_TRY(hipEventElapsedTime(&a, ctx.ev[0], ctx.ev[1]));
        HIP_TRY(hipEventElapsedTime(&b2, ctx.ev[2], ctx.ev[3]));
        tm->ms_count += a;
        tm->ms_fill += b2;
    }
    return 0;
}

}  // namespace mimeo
