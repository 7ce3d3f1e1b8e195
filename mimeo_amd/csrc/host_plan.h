// host_plan.h — HIP-free planning logic of the pipeline: which scaffolds are concatenated into which super-scaffold
// (pack.hip) and whether a pair list is a full cross product (pipeline.hip, run_packed).  Kept apart from the device
// code so that it runs under the CPU sanitizers (tests/sanitize/host_sanitize.cc, tests/test_host_sanitize.py).
#pragma once
#include <algorithm>
#include <cstdint>
#include <utility>
#include <vector>

namespace mimeo {
namespace host_plan {

struct Member { uint32_t id, start, len; };   // scaffold number, first base inside the super-scaffold, bases

// ids: the scaffolds that take part (ascending).  Scaffolds of at most member_max bases are packed, in order, into supers
// of at most about super_len bases: members start on 32-base boundaries, at least `spacer` bases behind the end of the
// member before; every other scaffold (and an empty one) is a super of its own.
inline std::vector<std::vector<Member>> plan_supers(const std::vector<uint64_t> &len_of, const std::vector<uint32_t> &ids, uint32_t spacer,
                                                    uint64_t member_max, uint64_t super_len) {
    std::vector<std::vector<Member>> plan;
    std::vector<Member> cur;
    uint64_t cur_end = 0;
    auto close = [&]() { if (!cur.empty()) { plan.push_back(cur); cur.clear(); } cur_end = 0; };
    for (uint32_t id : ids) {
        const uint64_t len = len_of[id];
        if (len > member_max || len == 0) {
            plan.push_back(std::vector<Member>{Member{id, 0u, (uint32_t)len}});
            continue;
        }
        uint64_t start = cur.empty() ? 0 : ((cur_end + spacer + 31) / 32) * 32;
        if (!cur.empty() && start + len > super_len) { close(); start = 0; }
        cur.push_back(Member{id, (uint32_t)start, (uint32_t)len});
        cur_end = start + len;
    }
    close();
    return plan;
}

// Is (pair_t[k], pair_q[k]), k < n, the full cross product of the targets and the queries it names?  Duplicates count
// once (dups: (duplicate, first occurrence)); pairidx[trank[t] * nq + qrank[q]] = first occurrence of (t, q).
struct CrossProduct {
    bool full = false;
    std::vector<uint32_t> tset, qset, trank, qrank, pairidx;
    std::vector<std::pair<uint64_t, uint64_t>> dups;
    size_t distinct = 0;
};
inline CrossProduct cross_product(const uint32_t *pair_t, const uint32_t *pair_q, uint64_t n, size_t n_targets, size_t n_queries) {
    CrossProduct c;
    c.tset.assign(pair_t, pair_t + n); c.qset.assign(pair_q, pair_q + n);
    std::sort(c.tset.begin(), c.tset.end()); c.tset.erase(std::unique(c.tset.begin(), c.tset.end()), c.tset.end());
    std::sort(c.qset.begin(), c.qset.end()); c.qset.erase(std::unique(c.qset.begin(), c.qset.end()), c.qset.end());
    c.trank.assign(n_targets, 0xFFFFFFFFu); c.qrank.assign(n_queries, 0xFFFFFFFFu);
    for (size_t i = 0; i < c.tset.size(); i++) c.trank[c.tset[i]] = (uint32_t)i;
    for (size_t i = 0; i < c.qset.size(); i++) c.qrank[c.qset[i]] = (uint32_t)i;
    const size_t nq = c.qset.size();
    if (nq && c.tset.size() > (size_t)0xFFFFFFF0u / nq) return c;   // more cells than a pair index can name: not taken
    if (c.tset.size() * nq > n) return c;   // fewer pairs than cells: cannot be the full product (a sparse list would cost |T| x |Q| cells to find out)
    c.pairidx.assign(c.tset.size() * nq, 0xFFFFFFFFu);
    for (uint64_t k = 0; k < n; k++) {
        uint32_t &slot = c.pairidx[(size_t)c.trank[pair_t[k]] * nq + c.qrank[pair_q[k]]];
        if (slot == 0xFFFFFFFFu) { slot = (uint32_t)k; c.distinct++; } else c.dups.emplace_back(k, slot);
    }
    c.full = c.distinct == c.pairidx.size();
    return c;
}

}  // namespace host_plan
}  // namespace mimeo
