// K1 — ASCII scaffold -> bit planes for both strands, and seed-start validity planes.
// Replaces the sequence loading lastz does for `lastz T Q` (reference call site
// src/mimeo/wrappers.py:1025-1031) and the per-scaffold split of utils.py:274-309.
// Runs once per genome; not on the timed path.
#include "common.h"

namespace mimeo {

// one thread per output word (32 bases): writes the lo/hi/nm components of pw[w]
__global__ void k1_pack_planes(const uint8_t *__restrict__ ascii, uint32_t len, int reverse, uint4 *__restrict__ pw,
                               uint2 *__restrict__ slim, uint32_t *__restrict__ lower, uint32_t nwords,
                               uint32_t *__restrict__ any_lower, uint32_t *__restrict__ any_n) {
    uint32_t w = blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= nwords) return;
    uint32_t vlo = 0, vhi = 0, vnm = 0, vlow = 0;
    uint32_t base = w * 32u;
#pragma unroll 4
    for (uint32_t b = 0; b < 32; b++) {
        uint32_t i = base + b;
        if (i >= len) break;
        uint8_t c = reverse ? ascii[len - 1 - i] : ascii[i];
        uint32_t low = (c >= 'a' && c <= 'z');
        if (low) c -= 32;
        uint32_t code = 4;
        if (c == 'A') code = 0;
        else if (c == 'C') code = 1;
        else if (c == 'G') code = 2;
        else if (c == 'T') code = 3;
        if (code == 4) {
            vnm |= 1u << b;
        } else {
            if (reverse) code = 3 - code;
            vlo |= (code & 1u) << b;
            vhi |= (code >> 1) << b;
        }
        vlow |= low << b;
    }
    pw[w] = make_uint4(vlo, vhi, vnm, 0u);
    slim[w] = make_uint2(vlo, vhi);
    if (vnm) atomicOr(any_n, 1u);
    if (lower) {
        lower[w] = vlow;
        if (vlow) atomicOr(any_lower, 1u);
    }
}

// sv bit p = 1 iff p + 19 <= len and no bad base in [p, p+19); bad = nm | lower (lower may be
// null).  Writes pw[w].w (out == null) or a plain plane out[w].
__global__ void k1_seed_valid(uint4 *__restrict__ pw, const uint32_t *__restrict__ lower, uint32_t len,
                              uint32_t nwords, uint32_t *__restrict__ out) {
    uint32_t w = blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= nwords) return;
    uint64_t bad = (uint64_t)pw[w].z | ((uint64_t)pw[w + 1].z << 32);  // word nwords is padding (zero)
    if (lower) bad |= (uint64_t)lower[w] | ((uint64_t)lower[w + 1] << 32);
    uint64_t b1 = bad | (bad >> 1);
    uint64_t b2 = b1 | (b1 >> 2);
    uint64_t b4 = b2 | (b2 >> 4);
    uint64_t b8 = b4 | (b4 >> 8);                      // offsets 0..15
    uint64_t inv = b8 | (b1 >> 16) | (bad >> 18);      // + 16,17 + 18
    uint32_t ok = ~(uint32_t)inv;
    // range: p + 19 <= len
    uint32_t base = w * 32u;
    if (len < SEED_LEN || base > len - SEED_LEN) ok = 0;
    else {
        uint32_t lastp = len - SEED_LEN;  // inclusive
        if (lastp - base < 31) ok &= (2u << (lastp - base)) - 1u;
    }
    if (out) out[w] = ok;
    else reinterpret_cast<uint32_t *>(&pw[w])[3] = ok;
}

static int alloc_strand(Strand &s, uint32_t len) {
    s.len = len;
    s.nwords = (len + 31) / 32;
    size_t per = (size_t)s.nwords + 2 * PLANE_PAD;
    HIP_TRY(hipMalloc((void **)&s.base, per * sizeof(uint4)));
    HIP_TRY(hipMemsetAsync(s.base, 0, per * sizeof(uint4), stream()));
    HIP_TRY(hipMalloc((void **)&s.slim, per * sizeof(uint2)));
    HIP_TRY(hipMemsetAsync(s.slim, 0, per * sizeof(uint2), stream()));
    return 0;
}

StrandView Strand::view(bool as_target) const {
    StrandView v;
    v.pw = base + PLANE_PAD;
    v.svt = (as_target && sv_target) ? sv_target + PLANE_PAD : nullptr;
    v.p2 = slim + PLANE_PAD;
    v.len = len;
    v.has_n = has_n ? 1u : 0u;
    return v;
}

int pack_scaffold(const uint8_t *d_ascii, uint64_t len64, Scaffold &out) {
    if (len64 > 0x7FFFFF00ull) { set_error("scaffold longer than 2^31-256 bases"); return MIMEO_ERR_LIMIT; }
    uint32_t len = (uint32_t)len64;
    out.len = len64;
    int rc;
    if ((rc = alloc_strand(out.fwd, len))) return rc;
    if ((rc = alloc_strand(out.rc, len))) return rc;
    uint32_t nwords = out.fwd.nwords;
    if (nwords == 0) return 0;
    size_t per = (size_t)nwords + 2 * PLANE_PAD;
    uint32_t *d_lower = nullptr, *d_flag = nullptr;
    HIP_TRY(hipMalloc((void **)&d_lower, per * sizeof(uint32_t) + 64));
    HIP_TRY(hipMemsetAsync(d_lower, 0, per * sizeof(uint32_t) + 64, stream()));
    d_flag = d_lower + per;
    dim3 blk(256), grd((nwords + 255) / 256);
    uint4 *f = out.fwd.base + PLANE_PAD, *r = out.rc.base + PLANE_PAD;
    hipLaunchKernelGGL(k1_pack_planes, grd, blk, 0, stream(), d_ascii, len, 0, f, out.fwd.slim + PLANE_PAD, d_lower + PLANE_PAD,
                       nwords, d_flag, d_flag + 1);
    hipLaunchKernelGGL(k1_pack_planes, grd, blk, 0, stream(), d_ascii, len, 1, r, out.rc.slim + PLANE_PAD, (uint32_t *)nullptr,
                       nwords, (uint32_t *)nullptr, d_flag + 1);
    hipLaunchKernelGGL(k1_seed_valid, grd, blk, 0, stream(), f, (const uint32_t *)nullptr, len, nwords, (uint32_t *)nullptr);
    hipLaunchKernelGGL(k1_seed_valid, grd, blk, 0, stream(), r, (const uint32_t *)nullptr, len, nwords, (uint32_t *)nullptr);
    uint32_t flag[2] = {0, 0};
    HIP_TRY(hipMemcpyAsync(flag, d_flag, 8, hipMemcpyDeviceToHost, stream()));
    HIP_TRY(hipStreamSynchronize(stream()));
    out.has_lower = flag[0] != 0;
    out.fwd.has_n = out.rc.has_n = flag[1] != 0;
    if (out.has_lower) {
        // the target role excludes soft-masked (lower-case) bases from seeding: separate sv plane
        HIP_TRY(hipMalloc((void **)&out.fwd.sv_target, per * sizeof(uint32_t)));
        HIP_TRY(hipMemsetAsync(out.fwd.sv_target, 0, per * sizeof(uint32_t), stream()));
        hipLaunchKernelGGL(k1_seed_valid, grd, blk, 0, stream(), f, (const uint32_t *)(d_lower + PLANE_PAD), len, nwords,
                           out.fwd.sv_target + PLANE_PAD);
        HIP_TRY(hipStreamSynchronize(stream()));
    }
    HIP_TRY(hipFree(d_lower));
    HIP_TRY(hipGetLastError());
    return 0;
}

void free_scaffold(Scaffold &s) {
    if (s.fwd.base) (void)hipFree(s.fwd.base);
    if (s.fwd.sv_target) (void)hipFree(s.fwd.sv_target);
    if (s.rc.base) (void)hipFree(s.rc.base);
    if (s.fwd.slim) (void)hipFree(s.fwd.slim);
    if (s.rc.slim) (void)hipFree(s.rc.slim);
    s.fwd = Strand();
    s.rc = Strand();
}

}  // namespace mimeo
