// pack.hip — super-scaffolds for fragmented assemblies (the per-pair loop of run_jobs.sh, src/mimeo/wrappers.py:1015-1059,
// runs lastz once per ordered scaffold pair whatever the scaffold sizes: S^2 invocations).
//
// A unit of this library (target scaffold, query scaffold, strand) costs two 64 MiB offset arrays read, three launches and
// a count per tile before the first seed hit is looked at — ~60 us however small the scaffolds.  A 2000-scaffold
// assembly is 8 * 10^6 units: eight minutes of nothing.  So the small scaffolds of a genome are concatenated, behind
// spacers of N, into SUPER-SCAFFOLDS of a few Mbp for the seed index and the gap-free stage (K2, K34, K4):
//   * a seed never spans an N and a gap-free walk loses 100 per N column, so with a spacer longer than x-drop / 100 no
//     walk crosses into the neighbour: the HSPs of a (super, super, strand) unit are exactly the union of the HSPs of
//     its member pairs, each inside one member on either side (checked on the device: an HSP that touches a spacer is
//     an error, not a result);
//   * every HSP is handed back to its member pair — coordinates made member-local, groups formed per (target
//     scaffold, query scaffold, strand) on the ORIGINAL strand views — and K5 (chain) / K6 (gapped extension) run on
//     those groups exactly as on the unpacked path.  The spacer therefore only has to stop the gap-free stage.
// Both strands of a member sit at the same word offset of the super-scaffold's two strands (the minus strand of the
// super is NOT the reverse complement of its plus strand: it is the members' minus strands in the same order), members
// start on 32-base boundaries, so the planes are copied word for word.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include <rocprim/rocprim.hpp>

#include "common.h"

namespace mimeo {

struct PackSrc {
    const uint4 *base;      // member planes (word 0)
    const uint32_t *svt;    // member's target-role sv plane (word 0) or null
    uint32_t w0, nwords, len, pad;
};

// one thread per word of the super-scaffold strand
__global__ void k_pack_super(const PackSrc *__restrict__ src, uint32_t nsrc, uint32_t total_words, uint4 *__restrict__ base,
                             uint2 *__restrict__ slim, uint32_t *__restrict__ svt) {
    const uint32_t w = blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= total_words) return;
    uint32_t lo = 0, hi = nsrc;   // last member with w0 <= w
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (src[mid].w0 <= w) lo = mid; else hi = mid;
    }
    const PackSrc m = src[lo];
    uint4 v = make_uint4(0u, 0u, 0xFFFFFFFFu, 0u);   // spacer: N, no seed starts
    uint32_t sv_t = 0;
    const uint32_t k = w - m.w0;
    if (w >= m.w0 && k < m.nwords) {
        v = m.base[k];
        sv_t = m.svt ? m.svt[k] : v.w;
        if (k == m.nwords - 1 && (m.len & 31u)) {   // bases beyond the member's end inside its last word
            const uint32_t valid = (1u << (m.len & 31u)) - 1u;
            v.x &= valid; v.y &= valid; v.z |= ~valid; v.w &= valid;
            sv_t &= valid;
        }
    }
    base[w] = v;
    slim[w] = make_uint2(v.x, v.y);
    if (svt) svt[w] = sv_t;
}

static int alloc_super_strand(Strand &s, uint32_t len, bool with_svt) {
    s.len = len;
    s.nwords = (len + 31) / 32;
    s.has_n = true;
    const size_t per = (size_t)s.nwords + 2 * PLANE_PAD;
    HIP_TRY(hipMalloc((void **)&s.base, per * sizeof(uint4)));
    HIP_TRY(hipMemsetAsync(s.base, 0, per * sizeof(uint4), stream()));
    HIP_TRY(hipMalloc((void **)&s.slim, per * sizeof(uint2)));
    HIP_TRY(hipMemsetAsync(s.slim, 0, per * sizeof(uint2), stream()));
    if (with_svt) {
        HIP_TRY(hipMalloc((void **)&s.sv_target, per * sizeof(uint32_t)));
        HIP_TRY(hipMemsetAsync(s.sv_target, 0, per * sizeof(uint32_t), stream()));
    }
    return 0;
}

void SuperSide::release() {
    for (size_t k = 0; k < supers.size(); k++)
        if (owned[k]) free_scaffold(supers[k]);
    supers.clear(); owned.clear(); members.clear(); super_of.clear(); start_of.clear();
}

// ids: the scaffolds of g that take part (ascending).  Scaffolds of at most member_max bases are packed, in order, into
// supers of about super_len bases; every other scaffold is a super of its own (its planes are used as they are).
int build_super_side(const mimeo_genome *g, const std::vector<uint32_t> &ids, uint32_t spacer, uint64_t member_max,
                     uint64_t super_len, SuperSide &out) {
    out.release();
    out.super_of.assign(g->scaf.size(), 0xFFFFFFFFu);
    out.start_of.assign(g->scaf.size(), 0u);
    std::vector<uint64_t> len_of(g->scaf.size());
    for (size_t i = 0; i < g->scaf.size(); i++) len_of[i] = g->scaf[i].len;
    const std::vector<std::vector<PackMember>> plan = host_plan::plan_supers(len_of, ids, spacer, member_max, super_len);
    out.supers.resize(plan.size());
    out.owned.assign(plan.size(), false);
    out.members = plan;
    hipStream_t st = stream();
    DeviceBuf tab;
    int rc = 0;
    for (size_t k = 0; k < plan.size() && !rc; k++) {
        const auto &mem = plan[k];
        for (const PackMember &m : mem) { out.super_of[m.id] = (uint32_t)k; out.start_of[m.id] = m.start; }
        if (mem.size() == 1 && mem[0].start == 0) {   // the scaffold itself
            out.supers[k] = g->scaf[mem[0].id];
            continue;
        }
        // a trailing spacer ends the last member as well (the planes' own padding looks like A, not N)
        const uint32_t total = mem.back().start + mem.back().len + spacer;
        Scaffold &S = out.supers[k];
        out.owned[k] = true;
        S.len = total;
        bool any_lower = false;
        for (const PackMember &m : mem) any_lower = any_lower || g->scaf[m.id].has_lower;
        S.has_lower = any_lower;
        if ((rc = alloc_super_strand(S.fwd, total, any_lower)) || (rc = alloc_super_strand(S.rc, total, false))) break;
        for (int minus = 0; minus < 2 && !rc; minus++) {
            std::vector<PackSrc> h(mem.size());
            for (size_t i = 0; i < mem.size(); i++) {
                const Scaffold &ms = g->scaf[mem[i].id];
                const Strand &sd = minus ? ms.rc : ms.fwd;
                h[i].base = sd.base + PLANE_PAD;
                h[i].svt = (!minus && sd.sv_target) ? sd.sv_target + PLANE_PAD : nullptr;
                h[i].w0 = mem[i].start / 32u; h[i].nwords = sd.nwords; h[i].len = sd.len; h[i].pad = 0;
            }
            if ((rc = tab.reserve(h.size() * sizeof(PackSrc)))) break;
            if (hipMemcpyAsync(tab.p, h.data(), h.size() * sizeof(PackSrc), hipMemcpyHostToDevice, st) != hipSuccess) {
                set_error("hipMemcpyAsync(super-scaffold member table) failed");
                rc = MIMEO_ERR_HIP;
                break;
            }
            Strand &D = minus ? S.rc : S.fwd;
            hipLaunchKernelGGL(k_pack_super, dim3((D.nwords + 255) / 256), dim3(256), 0, st, (const PackSrc *)tab.p, (uint32_t)h.size(),
                               D.nwords, D.base + PLANE_PAD, D.slim + PLANE_PAD, D.sv_target ? D.sv_target + PLANE_PAD : (uint32_t *)nullptr);
            if (hipStreamSynchronize(st) != hipSuccess) { set_error("super-scaffold packing failed"); rc = MIMEO_ERR_HIP; }   // h / tab are reused
        }
    }
    tab.release();
    if (rc) out.release();
    return rc;
}

// ---- HSPs of super units back to their member pairs ------------------------------------------------------------------
__device__ __forceinline__ uint32_t member_of(const uint32_t *__restrict__ starts, uint32_t n, uint32_t pos) {
    uint32_t lo = 0, hi = n;   // last member with start <= pos
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (starts[mid] <= pos) lo = mid; else hi = mid;
    }
    return lo;
}

__global__ void k_regroup_hsps(mimeo_hsp *__restrict__ hs, const uint32_t *__restrict__ hunit, uint64_t n, RegroupTables R,
                               uint32_t *__restrict__ keys, uint32_t *__restrict__ gmap, unsigned int *__restrict__ bad) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    mimeo_hsp h = hs[i];
    const uint3 ut = R.unit_tab[hunit[i]];   // target super, query super, minus
    const uint32_t tb = R.t_off[ut.x], tn = R.t_off[ut.x + 1] - tb, qb = R.q_off[ut.y], qn = R.q_off[ut.y + 1] - qb;
    const uint32_t tm = tb + member_of(R.t_start + tb, tn, h.tstart), qm = qb + member_of(R.q_start + qb, qn, h.qstart);
    const uint32_t ts = R.t_start[tm], qs = R.q_start[qm];
    if (h.tstart + h.length > ts + R.t_len[tm] || h.qstart + h.length > qs + R.q_len[qm]) atomicAdd(bad, 1u);   // touches a spacer
    h.tstart -= ts; h.qstart -= qs;
    hs[i] = h;
    const uint32_t key = R.pairidx[(size_t)R.t_rank[tm] * R.nq + R.q_rank[qm]] * 2u + ut.z;
    keys[i] = key;
    gmap[key] = 1u;
}

__global__ void k_build_groups(const uint32_t *__restrict__ gmap, const uint32_t *__restrict__ gscan, uint32_t nkeys, RegroupTables R,
                               Group *__restrict__ groups) {
    const uint32_t key = blockIdx.x * blockDim.x + threadIdx.x;
    if (key >= nkeys || !gmap[key]) return;
    const uint32_t pair = key >> 1, minus = key & 1u, t = R.pair_t[pair], q = R.pair_q[pair];
    Group G;
    memset(&G, 0, sizeof G);
    G.T = R.t_view[t];
    G.Q = minus ? R.q_view_rc[q] : R.q_view_fwd[q];
    G.tid = t; G.qid = q; G.minus = minus;
    groups[gscan[key]] = G;
}

__global__ void k_tag_hsps(const uint32_t *__restrict__ keys, const uint32_t *__restrict__ gscan, uint64_t n, uint32_t *__restrict__ tags) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) tags[i] = gscan[keys[i]];
}

// out[0] = sum of nchain, out[1] = number of groups whose gapped extension overflowed, out[2] = one such group + 1
__global__ void k_group_summary(const Group *__restrict__ groups, uint32_t ngroups, unsigned long long *__restrict__ out) {
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long nchain = g < ngroups ? groups[g].nchain : 0ull;
    for (int o = 32; o > 0; o >>= 1) nchain += __shfl_xor(nchain, o);
    if ((threadIdx.x & 63u) == 0 && nchain) atomicAdd(out, nchain);
    if (g < ngroups && groups[g].overflow) { atomicAdd(out + 1, 1ull); atomicMax(out + 2, (unsigned long long)g + 1ull); }
}

static DeviceBuf g_keys, g_gmap, g_gscan, g_tmp, g_sum;

void release_pack_buffers() {
    for (DeviceBuf *b : {&g_keys, &g_gmap, &g_gscan, &g_tmp, &g_sum}) b->release();
}

// hsps / hunit: the batch's HSPs in super coordinates with their unit numbers -> member-local coordinates, group tags in
// `tags` (may alias hunit), the groups in d_groups (grown on demand): *ngroups of them
int regroup_hsps_device(mimeo_hsp *d_hsps, uint32_t *d_hunit, uint64_t nh, const RegroupTables &R, uint32_t npairs, DeviceBuf &groups,
                        uint32_t *ngroups) {
    hipStream_t st = stream();
    const uint32_t nkeys = 2u * npairs;
    int rc;
    if ((rc = g_keys.reserve(nh * 4)) || (rc = g_gmap.reserve(((size_t)nkeys + 1) * 4)) || (rc = g_gscan.reserve(((size_t)nkeys + 1) * 4)) ||
        (rc = g_sum.reserve(64)))
        return rc;
    HIP_TRY(hipMemsetAsync(g_gmap.p, 0, ((size_t)nkeys + 1) * 4, st));
    HIP_TRY(hipMemsetAsync(g_sum.p, 0, 64, st));
    const dim3 blk(256), grd((uint32_t)((nh + 255) / 256));
    hipLaunchKernelGGL(k_regroup_hsps, grd, blk, 0, st, d_hsps, (const uint32_t *)d_hunit, nh, R, (uint32_t *)g_keys.p, (uint32_t *)g_gmap.p,
                       (unsigned int *)g_sum.p);
    size_t tb = 0;
    HIP_TRY(rocprim::exclusive_scan(nullptr, tb, (uint32_t *)g_gmap.p, (uint32_t *)g_gscan.p, 0u, (size_t)nkeys + 1, rocprim::plus<uint32_t>(), st));
    if ((rc = g_tmp.reserve(tb + 16))) return rc;
    HIP_TRY(rocprim::exclusive_scan(g_tmp.p, tb, (uint32_t *)g_gmap.p, (uint32_t *)g_gscan.p, 0u, (size_t)nkeys + 1, rocprim::plus<uint32_t>(), st));
    uint32_t h[2] = {0, 0};
    HIP_TRY(hipMemcpyAsync(&h[0], (uint32_t *)g_gscan.p + nkeys, 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(&h[1], g_sum.p, 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    if (h[1]) {
        char msg[160];
        snprintf(msg, sizeof msg, "internal error: %u HSPs of a packed unit touch a spacer between scaffolds (set MIMEO_PACK=0 and report)", h[1]);
        set_error(msg);
        return MIMEO_ERR_LIMIT;
    }
    *ngroups = h[0];
    if (!h[0]) return 0;
    if ((rc = groups.reserve((size_t)h[0] * sizeof(Group)))) return rc;
    hipLaunchKernelGGL(k_build_groups, dim3((nkeys + 255) / 256), blk, 0, st, (const uint32_t *)g_gmap.p, (const uint32_t *)g_gscan.p, nkeys, R,
                       (Group *)groups.p);
    hipLaunchKernelGGL(k_tag_hsps, grd, blk, 0, st, (const uint32_t *)g_keys.p, (const uint32_t *)g_gscan.p, nh, d_hunit);
    HIP_TRY(hipGetLastError());
    return 0;
}

// (target scaffold, query scaffold) of every group whose gapped extension hit a limit (a few at most)
__global__ void k_overflowed_groups(const Group *__restrict__ groups, uint32_t ngroups, uint2 *__restrict__ out, unsigned int *__restrict__ n, uint32_t cap) {
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g < ngroups && groups[g].overflow) {
        const unsigned int i = atomicAdd(n, 1u);
        if (i < cap) out[i] = make_uint2(groups[g].tid, groups[g].qid);
    }
}
int overflowed_groups_device(const Group *d_groups, uint32_t ngroups, uint64_t expect, std::vector<uint2> *out) {
    hipStream_t st = stream();
    int rc;
    const uint32_t cap = (uint32_t)std::min<uint64_t>(expect, 1u << 20);
    if ((rc = g_tmp.reserve((size_t)cap * 8 + 16))) return rc;
    unsigned int *n = (unsigned int *)((char *)g_tmp.p + (size_t)cap * 8);
    HIP_TRY(hipMemsetAsync(n, 0, 4, st));
    hipLaunchKernelGGL(k_overflowed_groups, dim3((ngroups + 255) / 256), dim3(256), 0, st, d_groups, ngroups, (uint2 *)g_tmp.p, n, cap);
    unsigned int got = 0;
    HIP_TRY(hipMemcpyAsync(&got, n, 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    out->resize(std::min<uint32_t>(got, cap));
    if (!out->empty()) HIP_TRY(hipMemcpy(out->data(), g_tmp.p, out->size() * 8, hipMemcpyDeviceToHost));
    return 0;
}

int group_summary_device(const Group *d_groups, uint32_t ngroups, uint64_t out[3]) {
    hipStream_t st = stream();
    int rc;
    if ((rc = g_sum.reserve(64))) return rc;
    HIP_TRY(hipMemsetAsync(g_sum.p, 0, 64, st));
    hipLaunchKernelGGL(k_group_summary, dim3((ngroups + 255) / 256), dim3(256), 0, st, d_groups, ngroups, (unsigned long long *)g_sum.p);
    HIP_TRY(hipMemcpyAsync(out, g_sum.p, 24, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    return 0;
}

}  // namespace mimeo
