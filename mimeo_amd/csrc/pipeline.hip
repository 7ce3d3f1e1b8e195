// pipeline.hip — host orchestration of the per-pair loop of run_jobs.sh
// (src/mimeo/wrappers.py:1015-1059 in the reference): mimeo_align_pairs / mimeo_align_pair.
//
// Pairs are grouped by target scaffold.  A scaffold strand's seed index (offsets, positions, seed frames) is
// built once and kept for the whole call (lastz rebuilds its table in every one of the S^2 invocations).  The
// (target, query, strand) units are worked off in BATCHES — a target row of a C4 job: 200 units — and a batch
// is one straight stream of device work with four host round trips in all:
//   heavy phase   per unit: the fused seed-scan / pre-filter kernel (K34), the sharp filter + exact walk of the hits it
//                 passes on (k4_extend_hits on the unit's walk queue), a queue reset — back to back, nothing read
//                 back, appending to batch-wide queues tagged with the unit
//   tails         walks beyond the frame, one radix sort of the followers of ALL units, segment resolution,
//                 entropy (K4; two round trips: follower count, HSP count)
//   K5 / K6       chain and gapped extension of all units of the batch (one workgroup per unit in K5,
//                 wavefronts per half extension in K6; round trips per DP round)
// Round 1 issued the tails per unit (~10 launches, 19 merge-sort passes and two host round trips each, three
// host threads to hide them); a C4 row now costs ~830 launches instead of ~6000.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <set>
#include <tuple>

#include "common.h"

namespace mimeo {

extern mimeo_stats g_stats;

// Seed indexes of one call (or of one index block of it).  key: (scaffold address, strand, target-role sv plane in
// use).  An index a genome kept from an earlier call (mimeo_genome_keep_indexes) is adopted, the others are built
// before the first batch that needs them runs.
struct IndexCache {
    typedef std::tuple<const Scaffold *, int, int> Key;
    std::map<Key, SeedIndex> m;
    std::vector<std::pair<Key, StrandView>> plan;
    std::map<Key, const mimeo_genome *> owner_of;
    std::set<Key> adopted;
    float ms = 0;

    static Key key_of(const Scaffold &s, int minus, bool as_target, StrandView *sv) {
        const bool tsv = as_target && !minus && s.fwd.sv_target != nullptr;
        *sv = (minus ? s.rc : s.fwd).view(tsv);
        return std::make_tuple(&s, minus, tsv ? 1 : 0);
    }
    static std::tuple<uint32_t, int, int> kept_key(const mimeo_genome *g, const Key &k) {
        return std::make_tuple((uint32_t)(std::get<0>(k) - g->scaf.data()), std::get<1>(k), std::get<2>(k));
    }
    void want(const mimeo_genome *owner, const Scaffold &s, int minus, bool as_target, std::set<Key> &seen) {
        StrandView sv;
        Key k = key_of(s, minus, as_target, &sv);
        if (!seen.insert(k).second) return;
        owner_of[k] = owner;
        if (owner) {
            auto it = owner->kept.find(kept_key(owner, k));
            if (it != owner->kept.end()) { m.emplace(k, it->second); adopted.insert(k); return; }
        }
        plan.emplace_back(k, sv);
    }
    int build_all() {
        for (auto &job : plan) {
            SeedIndex idx;
            int rc = build_index(job.second, idx, &ms);
            if (rc) return rc;
            m.emplace(job.first, idx);
        }
        plan.clear();
        return 0;
    }
    int get(const Scaffold &s, int minus, bool as_target, IndexView *out, StrandView *sv) {
        Key k = key_of(s, minus, as_target, sv);
        auto it = m.find(k);
        if (it == m.end()) { set_error("seed index was not planned"); return MIMEO_ERR_ARG; }
        *out = it->second.view();
        return 0;
    }
    // end of a call: indexes go to their genome when it keeps them (also after an error: they are
    // complete), else back to the pool
    ~IndexCache() { clear(); }   // error paths that leave the call early give the indexes back too
    void clear() {
        for (auto &kv : m) {
            if (adopted.count(kv.first)) continue;  // still owned by the genome
            const mimeo_genome *g = owner_of[kv.first];
            if (g && g->keep_indexes) g->kept.emplace(kept_key(g, kv.first), kv.second);
            else kv.second.release();
        }
        m.clear();
    }
};

static ExtBatch g_ext;
static DeviceBuf g_scratch, g_aln, g_dense, g_groups;
// tables of the packed path (run_packed)
static DeviceBuf d_toff, d_tstart, d_tlen, d_trank, d_qoff, d_qstart, d_qlen, d_qrank, d_pairidx, d_pt, d_pq, d_tview, d_qvf, d_qvr, d_utab;
static std::vector<std::pair<uint64_t, int>> g_failed;   // pairs of the last call that hit a limit
const std::vector<std::pair<uint64_t, int>> &failed_pairs() { return g_failed; }

constexpr uint64_t NO_PAIR = ~0ull;
struct Unit {
    uint64_t pair;          // index into pair_t / pair_q
    uint32_t tid, qid, minus;
    uint64_t mirror_pair;   // plus-strand unit of a self job: the pair (qid, tid) that receives the transposed HSPs, or NO_PAIR
};

// switches read once per call (tests change them between calls)
struct Switches {
    bool pack, mirror, no_diag0, timing, k6_stats;
    uint64_t pack_member, pack_super, index_budget_mb;
    size_t pack_min, batch_units;
    double batch_hits;
    static const char *env(const char *k) { return getenv(k); }
    Switches() {
        pack = !(env("MIMEO_PACK") && !atoi(env("MIMEO_PACK")));
        mirror = !(env("MIMEO_MIRROR") && !atoi(env("MIMEO_MIRROR")));
        no_diag0 = env("MIMEO_NO_DIAG0") != nullptr;
        timing = env("MIMEO_TIMING") != nullptr;
        k6_stats = env("MIMEO_K6_STATS") != nullptr;
        pack_member = env("MIMEO_PACK_MEMBER") ? (uint64_t)atol(env("MIMEO_PACK_MEMBER")) : (6ull << 20);
        pack_super = env("MIMEO_PACK_SUPER") ? (uint64_t)atol(env("MIMEO_PACK_SUPER")) : (20ull << 20);
        pack_min = env("MIMEO_PACK_MIN") ? (size_t)atol(env("MIMEO_PACK_MIN")) : 8;
        index_budget_mb = env("MIMEO_INDEX_BUDGET_MB") ? (uint64_t)atol(env("MIMEO_INDEX_BUDGET_MB")) : 0;
        batch_units = env("MIMEO_BATCH_UNITS") ? (size_t)std::max(1l, atol(env("MIMEO_BATCH_UNITS"))) : 0;
        batch_hits = env("MIMEO_BATCH_HITS") ? atof(env("MIMEO_BATCH_HITS")) : 2.5e10;
    }
};

// mimeo_shutdown: give the work buffers and streams back
void release_pipeline_buffers() {
    g_ext.release();
    g_scratch.release();
    g_aln.release();
    g_dense.release();
    g_groups.release();
    for (DeviceBuf *b : {&d_toff, &d_tstart, &d_tlen, &d_trank, &d_qoff, &d_qstart, &d_qlen, &d_qrank, &d_pairidx, &d_pt, &d_pq, &d_tview, &d_qvf, &d_qvr, &d_utab})
        b->release();
    release_pack_buffers();
}

// the extension stage of an arbitrary unit list, HSPs copied to the host per unit (stage entry point
// mimeo_ungapped_hsps: the same code path as mimeo_align_pairs up to K4)
int ungapped_units(const std::vector<UnitWork> &work, const mimeo_params *p, std::vector<std::vector<mimeo_hsp>> *per_unit,
                   ExtStats *st) {
    uint64_t nh = 0;
    int rc = g_ext.run(work, p, &nh, st);
    if (rc) return rc;
    per_unit->assign(work.size(), std::vector<mimeo_hsp>());
    if (!nh) return 0;
    std::vector<mimeo_hsp> h(nh);
    std::vector<uint32_t> u(nh);
    HIP_TRY(hipMemcpy(h.data(), g_ext.hsps.p, nh * sizeof(mimeo_hsp), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(u.data(), g_ext.hsp_unit.p, nh * 4, hipMemcpyDeviceToHost));
    for (uint64_t i = 0; i < nh; i++) (*per_unit)[u[i]].push_back(h[i]);
    return 0;
}

// ---- fragmented assemblies: the extension stage on super-scaffolds (pack.hip) ------------------------------------------
// Taken when the pair list is a full cross product T x Q (what every mimeo workflow asks for) with at least
// MIMEO_PACK_MIN (8) scaffolds of at most MIMEO_PACK_MEMBER (6 Mbp) bases on one side and no kept indexes; MIMEO_PACK=0
// switches it off.
// *used = false: the caller runs the unit-per-pair path.
template <typename T>
static int upload(DeviceBuf &b, const std::vector<T> &v) {
    int rc = b.reserve((v.size() ? v.size() : 1) * sizeof(T));
    if (rc) return rc;
    if (!v.empty()) HIP_TRY(hipMemcpyAsync(b.p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice, stream()));
    return 0;
}

static void record_failure(uint64_t pair, uint32_t tid, uint32_t qid, char strand, std::vector<char> &failed) {
    if (failed[pair]) return;
    failed[pair] = 1;
    g_failed.emplace_back(pair, MIMEO_ERR_LIMIT);
    char msg[256];
    snprintf(msg, sizeof msg, "gapped extension of target %u, query %u, strand %c: DP band wider than 65536 columns: "
             "not supported; the pair is left out (mimeo_get_failed_pairs)", tid, qid, strand);
    set_error(msg);   // readable through mimeo_last_error() although the call succeeds
}

static int run_packed(const mimeo_genome *A, const mimeo_genome *QG, const uint32_t *pair_t, const uint32_t *pair_q, uint64_t npairs,
                      const mimeo_params *p, const Switches &sw, std::vector<std::vector<mimeo_alignment>> &per_pair, std::vector<char> &failed,
                      ExtStats &est, float &ms_chain, float &ms_gapped, float &ms_index, bool *used) {
    *used = false;
    if (!sw.pack) return 0;
    if (npairs == 0 || npairs >= (1ull << 30)) return 0;
    // K34 works a 10 Mbp x 10 Mbp unit off at 13.5 ps per seed hit, a 5 Mbp x 5 Mbp one at 21.7, a 2 Mbp x 2 Mbp one at 117 (4096
    // tiles to stage whatever the scaffold size): scaffolds of up to 6 Mbp are packed into super-scaffolds of about 20 Mbp
    // when there are at least eight of them (C2, ten scaffolds of 5 Mbp: 144 -> 106 ms per job).
    const uint64_t member_max = sw.pack_member, super_len = sw.pack_super;
    const size_t pack_min = sw.pack_min;
    // indexes kept on a genome handle say that the caller issues the job as many calls (a row per call): they are used
    if (A->keep_indexes || QG->keep_indexes) return 0;
    {   // the cheap rejections first: too few small scaffolds among the ones named (before any |T| x |Q| table is made)
        std::vector<uint8_t> seen_t(A->scaf.size(), 0), seen_q(QG->scaf.size(), 0);
        size_t small_t = 0, small_q = 0;
        for (uint64_t k = 0; k < npairs; k++) {
            if (!seen_t[pair_t[k]]) { seen_t[pair_t[k]] = 1; small_t += A->scaf[pair_t[k]].len <= member_max; }
            if (!seen_q[pair_q[k]]) { seen_q[pair_q[k]] = 1; small_q += QG->scaf[pair_q[k]].len <= member_max; }
        }
        if (std::max(small_t, small_q) < pack_min) return 0;
    }
    host_plan::CrossProduct cp = host_plan::cross_product(pair_t, pair_q, npairs, A->scaf.size(), QG->scaf.size());
    const std::vector<uint32_t> &tset = cp.tset, &qset = cp.qset, &trank = cp.trank, &qrank = cp.qrank, &pairidx = cp.pairidx;
    const std::vector<std::pair<uint64_t, uint64_t>> &dups = cp.dups;   // (duplicate, first occurrence): answered from the first
    const size_t nq = qset.size(), distinct = cp.distinct;
    if (!cp.full) return 0;   // not the full cross product T x Q
    // one genome, the same scaffolds in both roles: the two roles share the super-scaffolds, and the main diagonals stay
    // with k4_diag0.  A subset of the targets against all scaffolds (a rank's share of a self job, dist.py) packs the two
    // roles separately: a scaffold's main diagonal is then an ordinary diagonal of its unit, whose seed hits are resolved
    // as the followers of its first one — the general rule, which k4_diag0 only short-cuts.
    const bool self = (A == QG) && tset == qset;
    const uint32_t spacer = (uint32_t)std::max(64, p->xdrop / 100 + 32);
    SuperSide side_t, side_q;
    int rc = build_super_side(A, tset, spacer, member_max, super_len, side_t);
    if (!rc && !self) rc = build_super_side(QG, qset, spacer, member_max, super_len, side_q);
    if (rc) { side_t.release(); side_q.release(); return rc; }
    SuperSide &ST = side_t, &SQ = self ? side_t : side_q;
    struct Cleanup { SuperSide &a, &b; ~Cleanup() { a.release(); b.release(); } } cleanup{side_t, side_q};
    // seed indexes of the supers: when they do not all fit (60 % of the free device memory; a 3 Gbp fragmented assembly needs
    // 330 GB of them) the super x super matrix is cut into index blocks, as on the unit-per-pair path
    uint64_t idx_need = 0, idx_budget = 0, idx_tmax = 1, idx_qmax = 1;
    {
        const int qroles = ((p->strand & MIMEO_STRAND_BOTH) == MIMEO_STRAND_BOTH) ? 2 : 1;
        for (auto &s : ST.supers) { idx_need += seed_index_bytes(s.len); idx_tmax = std::max(idx_tmax, seed_index_bytes(s.len)); }
        for (auto &s : SQ.supers) { idx_need += seed_index_bytes(s.len) * qroles; idx_qmax = std::max(idx_qmax, seed_index_bytes(s.len) * qroles); }
        size_t free_b = 0, total_b = 0;
        HIP_TRY(hipMemGetInfo(&free_b, &total_b));
        idx_budget = (uint64_t)(0.6 * (double)free_b);
        if (sw.index_budget_mb) idx_budget = sw.index_budget_mb << 20;
        if (idx_tmax + idx_qmax > idx_budget) return 0;   // not even one pair of supers: the other path cuts finer
    }
    *used = true;
    hipStream_t st = stream();
    // ---- device tables
    auto member_tables = [&](const SuperSide &S, const std::vector<uint32_t> &rank, std::vector<uint32_t> &off, std::vector<uint32_t> &start,
                             std::vector<uint32_t> &len, std::vector<uint32_t> &rk) {
        off.assign(1, 0u);
        for (auto &mem : S.members) {
            for (const PackMember &m : mem) { start.push_back(m.start); len.push_back(m.len); rk.push_back(rank[m.id]); }
            off.push_back((uint32_t)start.size());
        }
    };
    std::vector<uint32_t> h_toff, h_tstart, h_tlen, h_trank, h_qoff, h_qstart, h_qlen, h_qrank;
    member_tables(ST, trank, h_toff, h_tstart, h_tlen, h_trank);
    member_tables(SQ, qrank, h_qoff, h_qstart, h_qlen, h_qrank);
    std::vector<StrandView> h_tview(A->scaf.size()), h_qvf(QG->scaf.size()), h_qvr(QG->scaf.size());
    memset(h_tview.data(), 0, h_tview.size() * sizeof(StrandView));
    memset(h_qvf.data(), 0, h_qvf.size() * sizeof(StrandView));
    memset(h_qvr.data(), 0, h_qvr.size() * sizeof(StrandView));
    for (uint32_t t : tset) { IndexCache::key_of(A->scaf[t], 0, true, &h_tview[t]); }
    for (uint32_t q : qset) { IndexCache::key_of(QG->scaf[q], 0, false, &h_qvf[q]); IndexCache::key_of(QG->scaf[q], 1, false, &h_qvr[q]); }
    std::vector<uint32_t> h_pt(pair_t, pair_t + npairs), h_pq(pair_q, pair_q + npairs);
    if ((rc = upload(d_toff, h_toff)) || (rc = upload(d_tstart, h_tstart)) || (rc = upload(d_tlen, h_tlen)) || (rc = upload(d_trank, h_trank)) ||
        (rc = upload(d_qoff, h_qoff)) || (rc = upload(d_qstart, h_qstart)) || (rc = upload(d_qlen, h_qlen)) || (rc = upload(d_qrank, h_qrank)) ||
        (rc = upload(d_pairidx, pairidx)) || (rc = upload(d_pt, h_pt)) || (rc = upload(d_pq, h_pq)) || (rc = upload(d_tview, h_tview)) ||
        (rc = upload(d_qvf, h_qvf)) || (rc = upload(d_qvr, h_qvr)))
        return rc;
    HIP_TRY(hipStreamSynchronize(st));   // the host vectors go out of use only at the end of the call; be plain about it
    RegroupTables R;
    R.t_off = (const uint32_t *)d_toff.p; R.t_start = (const uint32_t *)d_tstart.p; R.t_len = (const uint32_t *)d_tlen.p; R.t_rank = (const uint32_t *)d_trank.p;
    R.q_off = (const uint32_t *)d_qoff.p; R.q_start = (const uint32_t *)d_qstart.p; R.q_len = (const uint32_t *)d_qlen.p; R.q_rank = (const uint32_t *)d_qrank.p;
    R.pairidx = (const uint32_t *)d_pairidx.p; R.nq = (uint32_t)nq; R.pad = 0;
    R.pair_t = (const uint32_t *)d_pt.p; R.pair_q = (const uint32_t *)d_pq.p;
    R.t_view = (const StrandView *)d_tview.p; R.q_view_fwd = (const StrandView *)d_qvf.p; R.q_view_rc = (const StrandView *)d_qvr.p;

    // ---- units: (target super, query super, strand), target-major.  Self job: the plus-strand unit (S1, S2) with S1 < S2 also
    // serves (S2, S1) — its HSPs transposed — when neither super holds soft-masked bases (shared plus strand, k4_mirror_hsps)
    struct SUnit { uint32_t ts, qs, minus, mirrored; };
    std::vector<SUnit> units;
    uint64_t max_t = 1, max_q = 1;
    const bool plus = (p->strand & MIMEO_STRAND_PLUS) != 0;
    auto can_mirror = [&](uint32_t a, uint32_t b) {
        return self && sw.mirror && plus && a != b && !ST.supers[a].fwd.sv_target && !ST.supers[b].fwd.sv_target;
    };
    for (uint32_t ts = 0; ts < ST.supers.size(); ts++)
        for (uint32_t qs = 0; qs < SQ.supers.size(); qs++)
            for (uint32_t minus = 0; minus < 2; minus++) {
                if (!(p->strand & (minus ? MIMEO_STRAND_MINUS : MIMEO_STRAND_PLUS))) continue;
                if (!minus && can_mirror(ts, qs)) {
                    if (ts > qs) continue;   // served by (qs, ts, +)
                    units.push_back(SUnit{ts, qs, 0, 1});
                } else {
                    units.push_back(SUnit{ts, qs, minus, 0});
                }
            }
    for (auto &s : ST.supers) max_t = std::max<uint64_t>(max_t, s.len);
    for (auto &s : SQ.supers) max_q = std::max<uint64_t>(max_q, s.len);
    std::vector<size_t> block_end;   // unit index where each index block ends
    if (idx_need > idx_budget && !units.empty()) {
        const uint64_t Bt = std::max<uint64_t>(1, idx_budget / 2 / idx_tmax), Bq = std::max<uint64_t>(1, idx_budget / 2 / idx_qmax);
        std::stable_sort(units.begin(), units.end(), [&](const SUnit &a, const SUnit &b) {
            const uint64_t ta = a.ts / Bt, tb = b.ts / Bt, qa = a.qs / Bq, qb = b.qs / Bq;
            if (ta != tb) return ta < tb;
            if (qa != qb) return qa < qb;
            return a.ts < b.ts;   // target-major inside a block
        });
        for (size_t i = 1; i <= units.size(); i++)
            if (i == units.size() || units[i].ts / Bt != units[i - 1].ts / Bt || units[i].qs / Bq != units[i - 1].qs / Bq) block_end.push_back(i);
    } else {
        block_end.push_back(units.size());
    }
    size_t max_units = std::min<size_t>(8192, ext_batch_max_units(max_t, max_q));
    if (sw.batch_units) max_units = std::min<size_t>(max_units, sw.batch_units);
    double max_hits = sw.batch_hits;
    const uint64_t max_groups = 1ull << 22;   // K5 names a group in 23 bits
    size_t blk_begin = 0;
    for (size_t blk = 0; blk < block_end.size() && !rc; blk_begin = block_end[blk], blk++) {
    const size_t blk_end = block_end[blk];
    IndexCache cache;
    {
        std::set<IndexCache::Key> seen;
        for (size_t i = blk_begin; i < blk_end; i++) {
            const SUnit &u = units[i];
            cache.want(nullptr, ST.supers[u.ts], 0, true, seen);
            cache.want(nullptr, SQ.supers[u.qs], (int)u.minus, false, seen);
        }
    }
    rc = cache.build_all();
    for (size_t b0 = blk_begin; b0 < blk_end && !rc;) {
        std::vector<UnitWork> work;
        std::vector<uint3> utab;
        std::vector<uint32_t> mirror_dst;
        double hits = 0;
        uint64_t tmembers = 0;
        size_t b1 = b0;
        for (; b1 < blk_end; b1++) {
            const SUnit &u = units[b1];
            if (work.size() + (u.mirrored ? 2 : 1) > max_units && !work.empty()) break;
            UnitWork w;
            memset(&w, 0, sizeof w);
            if ((rc = cache.get(ST.supers[u.ts], 0, true, &w.ti, &w.d.T)) || (rc = cache.get(SQ.supers[u.qs], (int)u.minus, false, &w.qi, &w.d.Q))) break;
            const double e = 13.0 * (double)w.ti.n * (double)w.qi.n / 16777216.0;
            // groups of the batch: at most (target members named) x nq x 2 (an upper bound; a mirror unit names the other super's)
            const uint64_t tm_new = tmembers + ST.members[u.ts].size() + (u.mirrored ? ST.members[u.qs].size() : 0);
            if (!work.empty() && (hits + e > max_hits || tm_new * nq * 2 > max_groups)) break;
            hits += e; tmembers = tm_new;
            w.d.same = (w.d.T.pw == w.d.Q.pw && w.d.T.len == w.d.Q.len && !sw.no_diag0) ? 1u : 0u;
            work.push_back(w);
            utab.push_back(make_uint3(u.ts, u.qs, u.minus));
            mirror_dst.push_back(NO_MIRROR);
            if (u.mirrored) {   // (qs, ts, +): no indexes, no heavy phase; receives the transposed HSPs
                UnitWork m;
                memset(&m, 0, sizeof m);
                StrandView sv;
                IndexCache::key_of(ST.supers[u.qs], 0, true, &sv); m.d.T = sv;
                IndexCache::key_of(ST.supers[u.ts], 0, false, &sv); m.d.Q = sv;
                mirror_dst.back() = (uint32_t)work.size();
                work.push_back(m);
                utab.push_back(make_uint3(u.qs, u.ts, 0));
                mirror_dst.push_back(NO_MIRROR);
            }
        }
        if (rc) break;
        uint64_t nh = 0;
        rc = g_ext.run(work, p, &nh, &est, &mirror_dst);
        if (rc == MIMEO_ERR_SPLIT) {   // the queues of this batch do not fit: smaller batches from here on
            if (b1 - b0 <= 1) { set_error("internal: a batch of one unit was refused"); rc = MIMEO_ERR_NOMEM; break; }
            rc = 0;
            max_hits = std::max(1.0, hits / 2);
            max_units = std::max<size_t>(1, std::min(max_units, work.size() / 2));
            continue;
        }
        if (rc) break;
        g_stats.super_units += work.size();
        g_stats.hsps += nh;
        for (size_t i = b0; i < b1; i++) g_stats.query_bases_scanned += SQ.supers[units[i].qs].len;
        g_stats.batches++;
        if (nh) {
            if ((rc = upload(d_utab, utab))) break;
            R.unit_tab = (const uint3 *)d_utab.p;
            uint32_t ngroups = 0;
            if ((rc = regroup_hsps_device((mimeo_hsp *)g_ext.hsps.p, (uint32_t *)g_ext.hsp_unit.p, nh, R, (uint32_t)npairs, g_groups, &ngroups))) break;
            if (ngroups >= (1u << 23)) { set_error("more than 2^23 scaffold pairs with HSPs in one batch"); rc = MIMEO_ERR_LIMIT; break; }
            if (!ngroups) { b0 = b1; continue; }
            if ((rc = g_aln.reserve(nh * sizeof(mimeo_alignment))) || (rc = g_dense.reserve(nh * sizeof(mimeo_alignment)))) break;
            if ((rc = chain_gapped_device((Group *)g_groups.p, ngroups, (const mimeo_hsp *)g_ext.hsps.p, (const uint32_t *)g_ext.hsp_unit.p, nh, p,
                                          g_scratch, (mimeo_alignment *)g_aln.p, &ms_chain, &ms_gapped)))
                break;
            dense_alignments_device((Group *)g_groups.p, ngroups, (const mimeo_alignment *)g_aln.p, (mimeo_alignment *)g_dense.p);
            uint64_t sum[3] = {0, 0, 0};
            if ((rc = group_summary_device((const Group *)g_groups.p, ngroups, sum))) break;
            Group last;
            HIP_TRY(hipMemcpy(&last, (const Group *)g_groups.p + (ngroups - 1), sizeof(Group), hipMemcpyDeviceToHost));
            if (sum[1]) {   // a pair that hit a limit is left out; the others go on (the reference's script has no `set -e`: utils.py:125-128)
                std::vector<uint2> bad;
                if ((rc = overflowed_groups_device((const Group *)g_groups.p, ngroups, sum[1], &bad))) break;
                for (const uint2 &tq : bad) record_failure(pairidx[(size_t)trank[tq.x] * nq + qrank[tq.y]], tq.x, tq.y, '?', failed);
            }
            g_stats.chained_hsps += sum[0];
            const uint64_t naln_total = (uint64_t)last.job0 + last.naln;
            std::vector<mimeo_alignment> host_aln(naln_total);
            if (naln_total) HIP_TRY(hipMemcpy(host_aln.data(), g_dense.p, naln_total * sizeof(mimeo_alignment), hipMemcpyDeviceToHost));
            // dense order = (pair, strand) order: a pair's plus-strand alignments come before its minus-strand ones, as on the other path
            for (const mimeo_alignment &a : host_aln) per_pair[pairidx[(size_t)trank[a.tid] * nq + qrank[a.qid]]].push_back(a);
        }
        b0 = b1;
    }
    cache.clear();
    ms_index += cache.ms;
    g_stats.index_blocks++;
    }   // index blocks
    g_stats.pair_strands += distinct * (((p->strand & MIMEO_STRAND_BOTH) == MIMEO_STRAND_BOTH) ? 2 : 1);
    if (!rc)
        for (auto &d : dups) { per_pair[d.first] = per_pair[d.second]; if (failed[d.second]) failed[d.first] = 1; }
    return rc;
}

int align_units_impl(const mimeo_genome *A, const mimeo_genome *B, const uint32_t *pair_t, const uint32_t *pair_q, const uint8_t *pair_strand,
                     uint64_t npairs, const mimeo_params *p, mimeo_alignment **out, uint64_t *nout) {
    auto t0 = std::chrono::steady_clock::now();
    memset(&g_stats, 0, sizeof g_stats);
    g_failed.clear();
    const Switches sw;
    const mimeo_genome *QG = B ? B : A;
    uint64_t max_t = 1, max_q = 1;
    bool uniform = true;   // every pair on the same strands: what run_jobs.sh asks for (wrappers.py:1031 --strand=both)
    const uint32_t all = (uint32_t)p->strand & MIMEO_STRAND_BOTH;
    auto strands_of = [&](uint64_t k) { return pair_strand ? ((uint32_t)pair_strand[k] & all) : all; };
    for (uint64_t k = 0; k < npairs; k++) {
        if (pair_t[k] >= A->scaf.size() || pair_q[k] >= QG->scaf.size()) { set_error("pair index out of range"); return MIMEO_ERR_ARG; }
        max_t = std::max(max_t, A->scaf[pair_t[k]].len);
        max_q = std::max(max_q, QG->scaf[pair_q[k]].len);
        if (strands_of(k) != strands_of(0)) uniform = false;
    }
    std::vector<std::vector<mimeo_alignment>> per_pair(npairs);
    std::vector<char> failed(npairs, 0);
    float ms_chain = 0, ms_gapped = 0, ms_index = 0;
    ExtStats est;
    int rc = 0;
    bool packed = false;
    HIP_TRY(hipStreamSynchronize(stream()));
    if (uniform && npairs) {
        mimeo_params pp = *p;
        pp.strand = (int32_t)strands_of(0);
        if (pp.strand && (rc = run_packed(A, QG, pair_t, pair_q, npairs, &pp, sw, per_pair, failed, est, ms_chain, ms_gapped, ms_index, &packed))) return rc;
    }
    if (!packed) {
    // units in target-major order (stable in the caller's pair order): neighbouring units share the target index
    std::vector<uint64_t> ord(npairs);
    for (uint64_t k = 0; k < npairs; k++) ord[k] = k;
    std::stable_sort(ord.begin(), ord.end(), [&](uint64_t a, uint64_t b) { return pair_t[a] < pair_t[b]; });
    // Shared plus strand (self jobs): when the list names (t, q) and (q, t), t != q, both on the plus strand, and neither
    // scaffold has soft-masked bases, the unit of the pair with t < q also produces the HSPs of the other one, transposed
    // (k4_mirror_hsps): one seed scan and one gap-free stage instead of two.  Chain and gapped extension run per pair as
    // ever (their tie-breaks are not symmetric).  First occurrences only; MIMEO_MIRROR=0 switches it off (tests).
    std::vector<uint64_t> mirror_of(npairs, NO_PAIR);   // canonical pair -> the pair it also serves
    std::vector<char> served(npairs, 0);                // pairs whose plus strand comes from their partner
    if (A == QG && sw.mirror && (all & MIMEO_STRAND_PLUS)) {
        struct E { uint64_t key, k; };
        std::vector<E> es;
        for (uint64_t k = 0; k < npairs; k++) {
            const uint32_t t = pair_t[k], q = pair_q[k];
            if (t == q || !(strands_of(k) & MIMEO_STRAND_PLUS)) continue;
            if (A->scaf[t].fwd.sv_target || A->scaf[q].fwd.sv_target) continue;   // soft-masked bases: target-only seeding rule
            es.push_back(E{((uint64_t)std::min(t, q) << 32) | std::max(t, q), k});
        }
        std::sort(es.begin(), es.end(), [](const E &a, const E &b) { return a.key != b.key ? a.key < b.key : a.k < b.k; });
        for (size_t i = 0; i < es.size();) {
            size_t j = i;
            uint64_t lo = NO_PAIR, hi = NO_PAIR;   // first occurrence of (min, max) and of (max, min)
            for (; j < es.size() && es[j].key == es[i].key; j++) {
                const uint64_t k = es[j].k;
                if (pair_t[k] < pair_q[k]) { if (lo == NO_PAIR) lo = k; } else if (hi == NO_PAIR) hi = k;
            }
            if (lo != NO_PAIR && hi != NO_PAIR) { mirror_of[lo] = hi; served[hi] = 1; }
            i = j;
        }
    }
    std::vector<Unit> units;
    for (uint64_t k = 0; k < npairs; k++)
        for (uint32_t minus = 0; minus < 2; minus++) {
            const uint64_t pk = ord[k];
            if (!(strands_of(pk) & (minus ? MIMEO_STRAND_MINUS : MIMEO_STRAND_PLUS))) continue;
            if (!minus && served[pk]) continue;
            units.push_back(Unit{pk, pair_t[pk], pair_q[pk], minus, minus ? NO_PAIR : mirror_of[pk]});
        }
    // Seed indexes cost 64 MiB + 52 bytes per base and strand, so a large or fragmented genome cannot keep them all
    // (a 1 Gbp genome, both strands: 117 GB; 2000 small scaffolds x 2 strands: 256 GB of offset arrays).  When the
    // indexes a call needs exceed the budget (60 % of the free device memory; MIMEO_INDEX_BUDGET_MB for tests) the
    // pair matrix is cut into blocks of Bt targets x Bq queries whose indexes fit, each block with an index cache of
    // its own: S + 2 S^2 / Bt builds instead of S + 2 S.  Results do not depend on the blocking (they are assembled
    // per pair).
    std::vector<size_t> block_end;  // unit index where each index block ends
    {
        std::map<uint32_t, uint64_t> tb, qb;  // bytes still to be built per distinct target / query scaffold
        for (const Unit &u : units) {
            const Scaffold &ts = A->scaf[u.tid], &qs = QG->scaf[u.qid];
            StrandView sv;
            if (!A->kept.count(IndexCache::kept_key(A, IndexCache::key_of(ts, 0, true, &sv)))) tb[u.tid] = seed_index_bytes(ts.len);
            if (!QG->kept.count(IndexCache::kept_key(QG, IndexCache::key_of(qs, (int)u.minus, false, &sv))))
                qb[u.qid] = std::max<uint64_t>(qb[u.qid], seed_index_bytes(qs.len) * (all == MIMEO_STRAND_BOTH ? 2 : 1));
        }
        uint64_t need = 0, tmax = 1, qmax = 1;
        for (auto &kv : tb) { need += kv.second; tmax = std::max(tmax, kv.second); }
        for (auto &kv : qb) { need += kv.second; qmax = std::max(qmax, kv.second); }
        size_t free_b = 0, total_b = 0;
        HIP_TRY(hipMemGetInfo(&free_b, &total_b));
        uint64_t budget = (uint64_t)(0.6 * (double)free_b);
        if (sw.index_budget_mb) budget = sw.index_budget_mb << 20;
        if (need > budget && !units.empty()) {
            const uint64_t Bt = std::max<uint64_t>(1, budget / 2 / tmax), Bq = std::max<uint64_t>(1, budget / 2 / qmax);
            std::map<uint32_t, uint64_t> trank, qrank;
            for (const Unit &u : units) { trank[u.tid]; qrank[u.qid]; }
            uint64_t r = 0;
            for (auto &kv : trank) kv.second = r++ / Bt;
            r = 0;
            for (auto &kv : qrank) kv.second = r++ / Bq;
            std::stable_sort(units.begin(), units.end(), [&](const Unit &a, const Unit &b) {
                const uint64_t ta = trank[a.tid], tbk = trank[b.tid], qa = qrank[a.qid], qbk = qrank[b.qid];
                if (ta != tbk) return ta < tbk;
                if (qa != qbk) return qa < qbk;
                return a.tid < b.tid;  // target-major inside a block; the two strands of a pair stay adjacent (stable)
            });
            for (size_t i = 1; i <= units.size(); i++)
                if (i == units.size() || trank[units[i].tid] != trank[units[i - 1].tid] || qrank[units[i].qid] != qrank[units[i - 1].qid])
                    block_end.push_back(i);
        } else {
            block_end.push_back(units.size());
        }
    }
    // batch limits: the follower key names at most 2^(64 - end bits - diagonal bits) units; K5/K6 take at most
    // MAX_GROUPS groups; and the queues of a batch are sized from its expected seed hits (2.5e10: a C4 row is 1.6e10)
    // — less when the free device memory says so (ExtBatch answers MIMEO_ERR_SPLIT: repeat-rich input beside resident
    // indexes, C5: half of all hits are followers inside microsatellites)
    const size_t MAX_GROUPS = 8192;
    size_t max_units = std::min<size_t>(MAX_GROUPS, ext_batch_max_units(max_t, max_q));
    if (sw.batch_units) max_units = std::min<size_t>(max_units, sw.batch_units);
    double max_hits = sw.batch_hits;
    hipStream_t st = stream();
    HIP_TRY(hipStreamSynchronize(st));
    size_t blk_begin = 0;
    for (size_t blk = 0; blk < block_end.size() && !rc; blk_begin = block_end[blk], blk++) {
        const size_t blk_end = block_end[blk];
        IndexCache cache;
        {
            std::set<IndexCache::Key> seen;
            for (size_t i = blk_begin; i < blk_end; i++) {
                const Unit &u = units[i];
                cache.want(A, A->scaf[u.tid], 0, true, seen);
                cache.want(QG, QG->scaf[u.qid], (int)u.minus, false, seen);
            }
        }
        rc = cache.build_all();
        // ---- the batches of this block, one at a time
        for (size_t b0 = blk_begin; b0 < blk_end && !rc;) {
            std::vector<UnitWork> work;
            std::vector<Group> groups;
            std::vector<uint64_t> pair_of;      // pair of every work unit (a mirror unit's is the served pair)
            std::vector<uint32_t> mirror_dst;
            double hits = 0;
            size_t b1 = b0;
            for (; b1 < blk_end; b1++) {
                const Unit &u = units[b1];
                const bool mir = u.mirror_pair != NO_PAIR;
                if (!work.empty() && work.size() + (mir ? 2 : 1) > max_units) break;
                const Scaffold &ts = A->scaf[u.tid], &qs = QG->scaf[u.qid];
                UnitWork w;
                memset(&w, 0, sizeof w);
                if ((rc = cache.get(ts, 0, true, &w.ti, &w.d.T)) || (rc = cache.get(qs, (int)u.minus, false, &w.qi, &w.d.Q))) break;
                const double e = 13.0 * (double)w.ti.n * (double)w.qi.n / 16777216.0;
                if (!work.empty() && hits + e > max_hits) break;
                hits += e;
                w.d.same = (w.d.T.pw == w.d.Q.pw && w.d.T.len == w.d.Q.len && !sw.no_diag0) ? 1u : 0u;
                Group g;
                memset(&g, 0, sizeof g);
                g.T = w.d.T; g.Q = w.d.Q; g.tid = u.tid; g.qid = u.qid; g.minus = u.minus;
                work.push_back(w); groups.push_back(g); pair_of.push_back(u.pair); mirror_dst.push_back(NO_MIRROR);
                if (mir) {   // (qid, tid, +): no indexes, no heavy phase; receives the transposed HSPs
                    UnitWork m;
                    memset(&m, 0, sizeof m);
                    IndexCache::key_of(qs, 0, true, &m.d.T);    // no soft-masked bases on either: the plain planes in both roles
                    IndexCache::key_of(ts, 0, false, &m.d.Q);
                    Group gm;
                    memset(&gm, 0, sizeof gm);
                    gm.T = m.d.T; gm.Q = m.d.Q; gm.tid = u.qid; gm.qid = u.tid; gm.minus = 0;
                    mirror_dst.back() = (uint32_t)work.size();
                    work.push_back(m); groups.push_back(gm); pair_of.push_back(u.mirror_pair); mirror_dst.push_back(NO_MIRROR);
                }
            }
            if (rc) break;
            auto tb0 = std::chrono::steady_clock::now();
            uint64_t nh = 0;
            rc = g_ext.run(work, p, &nh, &est, &mirror_dst);
            if (rc == MIMEO_ERR_SPLIT) {   // the queues of this batch do not fit beside the indexes: smaller batches from here on
                if (b1 - b0 <= 1) { set_error("internal: a batch of one unit was refused"); rc = MIMEO_ERR_NOMEM; break; }
                rc = 0;
                max_hits = std::max(1.0, hits / 2);
                max_units = std::max<size_t>(1, std::min(max_units, work.size() / 2));
                continue;
            }
            if (rc) break;
            auto tb1 = std::chrono::steady_clock::now();
            g_stats.pair_strands += work.size();
            g_stats.hsps += nh;
            for (size_t i = b0; i < b1; i++) g_stats.query_bases_scanned += QG->scaf[units[i].qid].len;
            g_stats.batches++;
            if (nh) {
                if ((rc = g_groups.reserve(groups.size() * sizeof(Group))) || (rc = g_aln.reserve(nh * sizeof(mimeo_alignment)))) break;
                if (hipMemcpyAsync(g_groups.p, groups.data(), groups.size() * sizeof(Group), hipMemcpyHostToDevice, st) != hipSuccess) {
                    set_error("hipMemcpyAsync(groups) failed");
                    rc = MIMEO_ERR_HIP;
                    break;
                }
                if ((rc = chain_gapped_device((Group *)g_groups.p, (uint32_t)groups.size(), (const mimeo_hsp *)g_ext.hsps.p,
                                              (const uint32_t *)g_ext.hsp_unit.p, nh, p, g_scratch, (mimeo_alignment *)g_aln.p, &ms_chain,
                                              &ms_gapped)))
                    break;
                // the alignments are a few thousand records in an array of one slot per HSP: packed on the device, then read
                if ((rc = g_dense.reserve(nh * sizeof(mimeo_alignment)))) break;
                dense_alignments_device((Group *)g_groups.p, (uint32_t)groups.size(), (const mimeo_alignment *)g_aln.p,
                                        (mimeo_alignment *)g_dense.p);
                if (hipMemcpyAsync(groups.data(), g_groups.p, groups.size() * sizeof(Group), hipMemcpyDeviceToHost, st) != hipSuccess ||
                    hipStreamSynchronize(st) != hipSuccess) {
                    set_error("HIP error while reading back alignments");
                    rc = MIMEO_ERR_HIP;
                    break;
                }
                const uint64_t naln_total = (uint64_t)groups.back().job0 + groups.back().naln;
                std::vector<mimeo_alignment> host_aln(naln_total);
                if (naln_total && (hipMemcpyAsync(host_aln.data(), g_dense.p, naln_total * sizeof(mimeo_alignment), hipMemcpyDeviceToHost, st) != hipSuccess ||
                                   hipStreamSynchronize(st) != hipSuccess)) {
                    set_error("HIP error while reading back alignments");
                    rc = MIMEO_ERR_HIP;
                    break;
                }
                for (size_t gi = 0; gi < groups.size(); gi++) {
                    const Group &g = groups[gi];
                    if (sw.k6_stats && gi < 24)
                        fprintf(stderr, "  [grp] t%u q%u %c hsps %llu nchain %u naln %u\n", g.tid, g.qid, g.minus ? '-' : '+',
                                (unsigned long long)(g.hsp_end - g.hsp_begin), g.nchain, g.naln);
                    if (g.overflow) {   // this pair hit a limit: it is left out, the others go on (utils.py:125-128: no `set -e`)
                        record_failure(pair_of[gi], g.tid, g.qid, g.minus ? '-' : '+', failed);
                        continue;
                    }
                    g_stats.chained_hsps += g.nchain;
                    auto &dst = per_pair[pair_of[gi]];
                    dst.insert(dst.end(), host_aln.begin() + g.job0, host_aln.begin() + g.job0 + g.naln);
                }
            }
            if (sw.timing) {
                auto tb2 = std::chrono::steady_clock::now();
                fprintf(stderr, "[timing] batch of %zu units: heavy + tails %.2f ms (device heavy %.2f + tails %.2f so far), chain + gapped + read-back %.2f ms (device %.2f + %.2f so far)\n",
                        work.size(), std::chrono::duration<double, std::milli>(tb1 - tb0).count(), est.ms_heavy, est.ms_tails,
                        std::chrono::duration<double, std::milli>(tb2 - tb1).count(), ms_chain, ms_gapped);
            }
            b0 = b1;
        }
        if (rc) (void)hipStreamSynchronize(st);   // a started batch drains
        cache.clear();
        ms_index += cache.ms;
        g_stats.index_blocks++;
    }
    }   // !packed
    if (rc) return rc;
    for (uint64_t k = 0; k < npairs; k++)
        if (failed[k]) per_pair[k].clear();   // a pair that hit a limit on one strand yields no rows at all (its lastz run failed)
    // a pair's plus-strand alignments come before its minus-strand ones whichever batch made them
    for (auto &v : per_pair)
        std::stable_sort(v.begin(), v.end(), [](const mimeo_alignment &a, const mimeo_alignment &b) { return a.qstrand < b.qstrand; });
    uint64_t total = 0;
    for (auto &v : per_pair) total += v.size();
    mimeo_alignment *res = (mimeo_alignment *)malloc((total ? total : 1) * sizeof(mimeo_alignment));
    if (!res) { set_error("host allocation failed"); return MIMEO_ERR_NOMEM; }
    uint64_t w = 0;
    for (auto &v : per_pair) { if (!v.empty()) memcpy(res + w, v.data(), v.size() * sizeof(mimeo_alignment)); w += v.size(); }
    *out = res;
    *nout = total;
    g_stats.alignments = total;
    g_stats.seed_hits = est.seed_hits;
    g_stats.scan_bytes_algorithmic = est.scan_bytes_algorithmic;
    g_stats.scan_bytes_kernel = est.scan_bytes_kernel;
    g_stats.scan_launches = est.heavy_launches;
    g_stats.scan_kernel_launches = est.heavy_kernel_launches;
    g_stats.walked_hits = est.walked;
    g_stats.followers = est.followers;
    g_stats.queue_reruns = est.reruns;
    g_stats.ms_index = ms_index;
    g_stats.ms_scan = est.ms_heavy;
    g_stats.ms_scan_fill = est.ms_k34;
    g_stats.ms_extend = est.ms_tails;
    g_stats.ms_chain = ms_chain;
    g_stats.ms_gapped = ms_gapped;
    g_stats.ms_total = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    if (getenv("MIMEO_TRACE")) {
        extern double g_alloc_ms;
        fprintf(stderr, "[trace] call: %.1f ms in all, %.1f ms of it in hipMalloc / hipFree of work buffers (since the library was loaded: cumulative)\n", g_stats.ms_total, g_alloc_ms);
    }
    return 0;
}

int build_kept_indexes(mimeo_genome *g, const uint32_t *scaf, uint64_t n) {
    const uint64_t cnt = n ? n : g->scaf.size();
    for (uint64_t i = 0; i < cnt; i++) {
        const uint32_t id = n ? scaf[i] : (uint32_t)i;
        const Scaffold &s = g->scaf[id];
        for (int role = 0; role < 3; role++) {  // query +, query -, target + (only when it differs)
            StrandView sv;
            IndexCache::Key k = IndexCache::key_of(s, role == 1, role == 2, &sv);
            auto kk = std::make_tuple(id, std::get<1>(k), std::get<2>(k));
            if (g->kept.count(kk)) continue;
            SeedIndex idx;
            int rc = build_index(sv, idx, nullptr);
            if (rc) return rc;
            g->kept.emplace(kk, idx);
        }
    }
    HIP_TRY(hipStreamSynchronize(stream()));
    return 0;
}

int chain_gapped_device(Group *d_groups, uint32_t ngroups, const mimeo_hsp *d_hsps, const uint32_t *d_hsp_unit, uint64_t nhsps,
                        const mimeo_params *p, DeviceBuf &scratch, mimeo_alignment *d_aln, float *ms_chain,
                        float *ms_gapped) {
    if (!ngroups || !nhsps) return 0;
    hipStream_t st = stream();
    // scratch: sorted HSPs | best | cand | pred | order
    size_t off_hs = 0, off_best = off_hs + nhsps * sizeof(mimeo_hsp), off_cand = off_best + nhsps * 8,
           off_pred = off_cand + nhsps * 8, off_order = off_pred + nhsps * 4, total = off_order + nhsps * 4;
    int rc = scratch.reserve(total);
    if (rc) return rc;
    char *b = (char *)scratch.p;
    static hipEvent_t e0 = nullptr, e1 = nullptr, e2 = nullptr;  // calls are blocking and single-threaded: one set will do
    if (!e0) { HIP_TRY(hipEventCreate(&e0)); HIP_TRY(hipEventCreate(&e1)); HIP_TRY(hipEventCreate(&e2)); }
    HIP_TRY(hipEventRecord(e0, st));
    if ((rc = chain_device(d_groups, ngroups, d_hsps, d_hsp_unit, nhsps, p->chain, (mimeo_hsp *)(b + off_hs), (long long *)(b + off_best),
                           (long long *)(b + off_cand), (int *)(b + off_pred), (uint32_t *)(b + off_order))))
        return rc;
    HIP_TRY(hipEventRecord(e1, st));
    if ((rc = gapped_device(d_groups, ngroups, (const mimeo_hsp *)(b + off_hs), (const uint32_t *)(b + off_order), nhsps, p,
                            d_aln)))
        return rc;
    HIP_TRY(hipEventRecord(e2, st));
    HIP_TRY(hipStreamSynchronize(st));
    HIP_TRY(hipGetLastError());
    float a = 0, c = 0;
    HIP_TRY(hipEventElapsedTime(&a, e0, e1));
    HIP_TRY(hipEventElapsedTime(&c, e1, e2));
    if (ms_chain) *ms_chain += a;
    if (ms_gapped) *ms_gapped += c;
    return 0;
}

}  // namespace mimeo
