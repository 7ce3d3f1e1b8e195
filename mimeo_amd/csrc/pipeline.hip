// pipeline.hip — host orchestration of the per-pair loop of run_jobs.sh
// (src/mimeo/wrappers.py:1015-1059 in the reference): mimeo_align_pairs / mimeo_align_pair.
//
// Pairs are grouped by target scaffold.  A scaffold strand's seed index is built once and
// kept for the whole call (lastz rebuilds its table in every one of the S^2 invocations).  Every
// (target, query, strand) unit goes through K3 (index join) and K4 (gap-free extension) on one of
// three lanes (host thread + stream each); the HSPs of up to MAX_GROUPS units are then chained and
// gap-extended together (K5/K6, one workgroup per unit in K5, wavefronts per half extension in K6).
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <condition_variable>
#include <map>
#include <mutex>
#include <set>
#include <thread>
#include <tuple>

#include "common.h"

namespace mimeo {

extern mimeo_stats g_stats;

struct IndexCache {
    // key: (scaffold address, strand, target-role sv plane in use).  A builder thread with a stream of its
    // own works through the indexes in the order the units will need them, ahead of the lanes.  It issues
    // one build each time a lane has launched a fast K4 kernel (a token): the radix sorts are bandwidth
    // work that fits beside that VALU-bound kernel (0.6 ms vs 0.9 ms), and they are over before the next
    // seed scan reads its offset arrays, so they neither share the chip with it nor evict what its count
    // pass left in the Infinity Cache.  A lane that needs an index not yet built waits (and that lets the
    // builder run at once).  build_index synchronises its stream before returning: a published index is
    // complete.
    typedef std::tuple<const Scaffold *, int, int> Key;
    std::map<Key, SeedIndex> m;
    std::mutex mu;
    std::condition_variable cv;
    std::vector<std::pair<Key, StrandView>> plan;
    std::thread builder;
    hipStream_t bstream = nullptr;
    bool stop = false;
    int rc = 0, tokens = 0, waiters = 0;
    std::string err;
    float ms = 0;

    static Key key_of(const Scaffold &s, int minus, bool as_target, StrandView *sv) {
        const bool tsv = as_target && !minus && s.fwd.sv_target != nullptr;
        *sv = (minus ? s.rc : s.fwd).view(tsv);
        return std::make_tuple(&s, minus, tsv ? 1 : 0);
    }
    // `owner` is the genome the scaffold belongs to: an index it kept from an earlier call is adopted
    // instead of being planned (mimeo_genome_keep_indexes)
    std::map<Key, const mimeo_genome *> owner_of;
    std::set<Key> adopted;
    static std::tuple<uint32_t, int, int> kept_key(const mimeo_genome *g, const Key &k) {
        return std::make_tuple((uint32_t)(std::get<0>(k) - g->scaf.data()), std::get<1>(k), std::get<2>(k));
    }
    void want(const mimeo_genome *owner, const Scaffold &s, int minus, bool as_target, std::set<Key> &seen) {
        StrandView sv;
        Key k = key_of(s, minus, as_target, &sv);
        if (!seen.insert(k).second) return;
        owner_of[k] = owner;
        auto it = owner->kept.find(kept_key(owner, k));
        if (it != owner->kept.end()) { m.emplace(k, it->second); adopted.insert(k); }
        else plan.emplace_back(k, sv);
    }
    void start(hipStream_t st) {
        bstream = st;
        builder = std::thread([this] {
            (void)hipSetDevice(device_id());
            set_thread_stream(bstream);
            for (auto &job : plan) {
                {
                    std::unique_lock<std::mutex> lk(mu);
                    cv.wait(lk, [&] { return stop || tokens > 0 || waiters > 0; });
                    if (stop) break;
                    if (tokens > 0) tokens--;
                }
                SeedIndex idx;
                float t = 0;
                int r = build_index(job.second, idx, &t);
                std::lock_guard<std::mutex> lk(mu);
                ms += t;
                if (r) { rc = r; err = last_error_copy(); cv.notify_all(); break; }
                m.emplace(job.first, idx);
                cv.notify_all();
            }
            set_thread_stream(nullptr);
        });
    }
    int get(const Scaffold &s, int minus, bool as_target, IndexView *out, StrandView *sv) {
        Key k = key_of(s, minus, as_target, sv);
        std::unique_lock<std::mutex> lk(mu);
        if (!m.count(k) && !rc && !stop) {
            waiters++;
            cv.notify_all();
            cv.wait(lk, [&] { return m.count(k) || rc || stop; });
            waiters--;
        }
        auto it = m.find(k);
        if (it == m.end()) {
            set_error(rc ? err : std::string("seed index was not planned"));
            return rc ? rc : MIMEO_ERR_ARG;
        }
        *out = it->second.view();
        return 0;
    }
    void token() {  // a fast K4 kernel has just been launched
        {
            std::lock_guard<std::mutex> lk(mu);
            if (tokens < 2) tokens++;
        }
        cv.notify_all();
    }
    void finish() {
        {
            std::lock_guard<std::mutex> lk(mu);
            stop = true;
        }
        cv.notify_all();
        if (builder.joinable()) builder.join();
    }
    // end of a call: indexes go to their genome when it keeps them (also after an error: they are
    // complete), else back to the pool
    void clear() {
        finish();
        for (auto &kv : m) {
            if (adopted.count(kv.first)) continue;  // still owned by the genome
            const mimeo_genome *g = owner_of[kv.first];
            if (g && g->keep_indexes) g->kept.emplace(kept_key(g, kv.first), kv.second);
            else kv.second.release();
        }
        m.clear();
    }
};

static uint64_t scan_bytes_algorithmic(uint64_t Lq, uint64_t H) {
    // B_scan = ceil(Lq/4) + 8*W*(Lq-18) + 4*H + 8*H, W = 13
    return (Lq + 3) / 4 + (Lq > 18 ? 8ull * 13ull * (Lq - 18) : 0) + 12ull * H;
}
static uint64_t scan_bytes_kernel(uint64_t nT, uint64_t nQ, uint64_t H) {
    // compulsory traffic of the index join: both offset arrays, both position lists once, hits out
    return 2ull * 4ull * ((uint64_t)NBUCKET + 1) + 4ull * (nT + nQ) + 8ull * H;
}

// A lane = one host thread + one stream + its own K3/K4 work buffers, taking units off a shared counter.
// Three lanes keep the GPU busy through the host round trips and the small latency-bound kernels of a
// unit (follower sort, segment resolution, entropy).  The heavy phase of a unit — K3 and the fast K4
// kernel — is serialised across lanes (HeavyGate): each one starts after the previous one's fast kernel
// has finished, so the bandwidth-bound seed scan is never sharing the chip with another scan.
// Why three: a lane's stream is in order, so its next heavy phase cannot start before the tails of its
// previous unit are through; with the pre-filtered fast kernel (1.8 ms per 10 Mbp x 10 Mbp unit) the tails,
// squeezed in beside another lane's heavy phase, take about as long as that phase, and with two lanes the
// chip idled ~0.25 ms per unit waiting for them (rocprofv3 timeline, scripts/timeline_gaps.py).
struct Lane {
    hipStream_t st = nullptr;
    hipEvent_t heavy_end = nullptr;
    JoinCtx jc;
    ExtWork ew;
    DeviceBuf hits, unit_hsps;
    JoinTiming tm;
    float ms_ext = 0;
    double excess = 1.0;  // largest hits / expected-on-random ratio of the units seen in this call
};
struct HeavyGate {
    std::mutex mu;
    hipEvent_t last = nullptr;
    // hand-over bookkeeping (own lock: read while another lane holds `mu` for its whole heavy phase)
    std::mutex hmu;
    std::condition_variable hcv;
    uint64_t seq = 0;                 // heavy phases issued so far
    hipEvent_t last_fill = nullptr;   // "seed-scan fill finished" event of the latest one (may be null)
    int active = 0;                   // lanes that may still issue a heavy phase
    bool failed = false;
    void acquire(hipStream_t st) {
        mu.lock();
        if (last) (void)hipStreamWaitEvent(st, last, 0);
    }
    // returns the sequence number of the heavy phase just issued
    uint64_t release(hipStream_t st, hipEvent_t mine, hipEvent_t fill_done) {
        (void)hipEventRecord(mine, st);
        last = mine;
        uint64_t my;
        {
            std::lock_guard<std::mutex> lk(hmu);
            my = ++seq;
            last_fill = fill_done;
        }
        mu.unlock();
        hcv.notify_all();
        return my;
    }
    // The small kernels that finish a unit should run beside the NEXT lane's fast K4 kernel, not in front of
    // its seed scan (which would have to wait for them: it runs alone).  So a lane holds them back until the
    // next heavy phase has been issued and orders them behind that phase's fill.
    void tails_after_next_fill(hipStream_t st, uint64_t my) {
        std::unique_lock<std::mutex> lk(hmu);
        hcv.wait(lk, [&] { return seq > my || active <= 1 || failed; });
        if (seq > my && last_fill) (void)hipStreamWaitEvent(st, last_fill, 0);
    }
    void lane_done(bool error) {
        {
            std::lock_guard<std::mutex> lk(hmu);
            active--;
            if (error) failed = true;
        }
        hcv.notify_all();
    }
};
constexpr int MAX_LANES = 4;
static Lane g_lane[MAX_LANES];
static DeviceBuf g_hsp_batch, g_scratch, g_aln, g_groups;

struct Unit {
    uint64_t pair;  // index into pair_t / pair_q
    uint32_t tid, qid, minus;
};

struct Batch {
    std::vector<Group> groups;
    std::vector<uint64_t> group_pair;
    uint64_t nh_total = 0;
};
struct BatchResult {
    int rc = 0;
    std::string err;
    uint64_t chained = 0;
    float ms_chain = 0, ms_gapped = 0;
};

// K5 + K6 + read-back of one batch of units whose HSPs are complete
static BatchResult run_batch(Batch &b, const mimeo_params *p, std::vector<std::vector<mimeo_alignment>> *per_pair) {
    BatchResult r;
    hipStream_t st = stream();
    auto fail = [&](int rc) { r.rc = rc; r.err = last_error_copy(); return r; };
    if (!b.nh_total || b.groups.empty()) return r;
    int rc;
    if ((rc = g_groups.reserve(b.groups.size() * sizeof(Group)))) return fail(rc);
    if ((rc = g_aln.reserve(b.nh_total * sizeof(mimeo_alignment)))) return fail(rc);
    if (hipMemcpyAsync(g_groups.p, b.groups.data(), b.groups.size() * sizeof(Group), hipMemcpyHostToDevice, st) != hipSuccess)
        return fail(MIMEO_ERR_HIP);
    if ((rc = chain_gapped_device((Group *)g_groups.p, (uint32_t)b.groups.size(), (const mimeo_hsp *)g_hsp_batch.p,
                                  b.nh_total, p, g_scratch, (mimeo_alignment *)g_aln.p, &r.ms_chain, &r.ms_gapped)))
        return fail(rc);
    std::vector<mimeo_alignment> host_aln(b.nh_total);
    if (hipMemcpyAsync(b.groups.data(), g_groups.p, b.groups.size() * sizeof(Group), hipMemcpyDeviceToHost, st) != hipSuccess ||
        hipMemcpyAsync(host_aln.data(), g_aln.p, b.nh_total * sizeof(mimeo_alignment), hipMemcpyDeviceToHost, st) != hipSuccess ||
        hipStreamSynchronize(st) != hipSuccess) {
        set_error("HIP error while reading back alignments");
        return fail(MIMEO_ERR_HIP);
    }
    if (getenv("MIMEO_K6_STATS"))
        for (size_t gi = 0; gi < b.groups.size() && gi < 24; gi++)
            fprintf(stderr, "  [grp] t%u q%u %c hsps %llu nchain %u naln %u\n", b.groups[gi].tid, b.groups[gi].qid,
                    b.groups[gi].minus ? '-' : '+', (unsigned long long)(b.groups[gi].hsp_end - b.groups[gi].hsp_begin),
                    b.groups[gi].nchain, b.groups[gi].naln);
    for (size_t gi = 0; gi < b.groups.size(); gi++) {
        const Group &g = b.groups[gi];
        if (g.overflow) {
            set_error("gapped extension: DP band wider than 65536 columns, or score beyond int32: not supported");
            return fail(MIMEO_ERR_LIMIT);
        }
        r.chained += g.nchain;
        auto &dst = (*per_pair)[b.group_pair[gi]];
        dst.insert(dst.end(), host_aln.begin() + g.hsp_begin, host_aln.begin() + g.hsp_begin + g.naln);
    }
    return r;
}

// state shared by the lanes while one batch of units goes through K3/K4
struct Shared {
    const mimeo_genome *A, *QG;
    const mimeo_params *p;
    const std::vector<Unit> *units;
    size_t begin, end;            // unit range of this batch
    std::atomic<size_t> next;
    std::atomic<int> rc{0};
    std::string err;
    std::mutex mu;                // batch append, stats, err
    IndexCache *cache;
    HeavyGate gate;
    Exclusive *excl;
    bool speculative = true, handover = true;
    uint64_t spec_shrink = 0;
    Batch *batch;
};

static void lane_main(Lane *ln, Shared *sh) {
    (void)hipSetDevice(device_id());
    set_thread_stream(ln->st);
    hipStream_t st = ln->st;
    const mimeo_params *p = sh->p;
    auto fail = [&](int rc) {
        std::lock_guard<std::mutex> lk(sh->mu);
        if (!sh->rc.load()) { sh->rc = rc; sh->err = last_error_copy(); }
    };
    for (;;) {
        size_t ui = sh->next.fetch_add(1);
        if (ui >= sh->end || sh->rc.load()) break;
        const Unit &u = (*sh->units)[ui];
        const Scaffold &ts = sh->A->scaf[u.tid], &qs = sh->QG->scaf[u.qid];
        IndexView ti, qi;
        StrandView tv, qv;
        int rc;
        if ((rc = sh->cache->get(ts, 0, true, &ti, &tv)) || (rc = sh->cache->get(qs, (int)u.minus, false, &qi, &qv))) { fail(rc); break; }
        uint64_t nhits = 0, nh = 0;
        // Speculative first: buffers sized from the expected hit count (13 probes per query word on random
        // sequence, scaled by the largest excess seen in this call), no host round trip between the seed
        // scan and the fast K4 kernel, so the lane hands the heavy phase on at once.  A unit that does not
        // fit (the kernels then do nothing) is repeated with the exact count.
        const double expect = 13.0 * (double)ti.n * (double)qi.n / 16777216.0;
        // A unit too large for one pass (2^32 hits are the limit of the hit indices, and every hit costs 36 bytes
        // of work buffers): the query is joined chunk by chunk — an index over a range of query positions per chunk,
        // hit coordinates stay global — with the followers and candidates of all chunks resolved together at the end
        // (ExtChunk, common.h).  The unit keeps the heavy-phase gate for its whole duration.
        const double chunk_hits = getenv("MIMEO_CHUNK_HITS") ? atof(getenv("MIMEO_CHUNK_HITS")) : 1.5e9;
        if (expect > chunk_hits) {
            const uint32_t nchunks = (uint32_t)std::min(4096.0, std::ceil(expect / (0.67 * chunk_hits)));
            sh->gate.acquire(st);
            ExtChunk ch{1, 0, (uint64_t)(expect / 32.0) + (1u << 20), 0, 0};
            // ranges of query positions still to do (ascending); a range whose real hit count is beyond the
            // per-chunk budget (repeats, satellites) is halved
            const uint64_t max_hits = getenv("MIMEO_CHUNK_MAX_HITS") ? (uint64_t)atof(getenv("MIMEO_CHUNK_MAX_HITS")) : (uint64_t)(2.0 * chunk_hits);
            std::vector<std::pair<uint32_t, uint32_t>> todo;
            for (uint32_t c = nchunks; c-- > 0;)
                todo.emplace_back((uint32_t)((uint64_t)qv.len * c / nchunks), (uint32_t)((uint64_t)qv.len * (c + 1) / nchunks));
            bool any = false;
            while (!todo.empty() && !rc) {
                const std::pair<uint32_t, uint32_t> r = todo.back();
                todo.pop_back();
                SeedIndex qc;
                float ms_idx = 0;
                uint64_t n = 0;
                if ((rc = build_index(qv, qc, &ms_idx, r.first, r.second))) break;
                rc = join_hits(ln->jc, ti, qc.view(), p->transitions, ln->hits, &n, &ln->tm, sh->excl, 0, max_hits);
                qc.release();
                if (rc == MIMEO_SPLIT) {
                    rc = 0;
                    if (r.second - r.first < 2) { set_error("one query position yields more seed hits than a chunk may hold"); rc = MIMEO_ERR_LIMIT; break; }
                    const uint32_t mid = r.first + (r.second - r.first) / 2;
                    todo.emplace_back(mid, r.second);
                    todo.emplace_back(r.first, mid);
                    g_stats.chunk_splits++;
                    continue;
                }
                if (rc) break;
                ch.first = !any;
                ch.last = todo.empty();
                ch.nfollow_before = ch.nfollow_after;
                rc = ungapped_hsps_device(ln->ew, tv, qv, (const uint2 *)ln->hits.p, n, p, ln->unit_hsps, &nh, &ln->ms_ext, nullptr,
                                          nullptr, nullptr, &ch);
                any = true;
                nhits += n;
            }
            (void)sh->gate.release(st, ln->heavy_end, nullptr);
            sh->cache->token();
            if (rc) { fail(rc); break; }
            std::lock_guard<std::mutex> lk(sh->mu);
            g_stats.chunked_units++;
        } else {
        uint64_t spec_cap = sh->speculative ? (uint64_t)(expect * std::max(1.5, 1.25 * ln->excess)) + (4u << 20) : 0;
        if (spec_cap && sh->spec_shrink > 0) spec_cap = spec_cap / sh->spec_shrink + 1;  // tests: force the retry path
        if (spec_cap >= (1ull << 32)) spec_cap = 0;
        for (int attempt = 0; attempt < 2; attempt++) {
            bool held = true;
            sh->gate.acquire(st);
            std::function<void()> after_fast = [&] {
                if (!held) return;
                held = false;
                hipEvent_t fill_done = nullptr;
                if (sh->excl) { const int me = sh->excl->index_of(st); if (me >= 0) fill_done = sh->excl->done[me]; }
                const uint64_t my = sh->gate.release(st, ln->heavy_end, fill_done);
                sh->cache->token();
                if (sh->handover) sh->gate.tails_after_next_fill(st, my);
            };
            rc = join_hits(ln->jc, ti, qi, p->transitions, ln->hits, &nhits, &ln->tm, sh->excl, spec_cap);
            if (!rc)
                rc = ungapped_hsps_device(ln->ew, tv, qv, (const uint2 *)ln->hits.p, spec_cap ? spec_cap : nhits, p, ln->unit_hsps, &nh,
                                          &ln->ms_ext, &after_fast, spec_cap ? ln->jc.total_dev() : nullptr, &nhits);
            after_fast();  // no hits, or an error before the fast kernel
            if (rc != MIMEO_RETRY_EXACT) break;
            spec_cap = 0;
            rc = 0;
        }
        if (!rc && expect > 0) ln->excess = std::max(ln->excess, (double)nhits / expect);
        if (rc) { fail(rc); break; }
        }  // one-pass unit
        {
            std::lock_guard<std::mutex> lk(sh->mu);
            Batch &b = *sh->batch;
            if ((b.nh_total + nh) * sizeof(mimeo_hsp) > g_hsp_batch.cap) {
                // grow: every lane's copies into the old buffer are issued under this lock, so a device-wide
                // wait makes them complete
                DeviceBuf bigger;
                if ((rc = bigger.reserve((b.nh_total + nh) * 2 * sizeof(mimeo_hsp) + 4096))) { if (!sh->rc.load()) { sh->rc = rc; sh->err = last_error_copy(); } break; }
                (void)hipDeviceSynchronize();
                if (b.nh_total) (void)hipMemcpy(bigger.p, g_hsp_batch.p, b.nh_total * sizeof(mimeo_hsp), hipMemcpyDeviceToDevice);
                g_hsp_batch.release();
                g_hsp_batch = bigger;
            }
            if (nh && hipMemcpyAsync((char *)g_hsp_batch.p + b.nh_total * sizeof(mimeo_hsp), ln->unit_hsps.p, nh * sizeof(mimeo_hsp),
                                     hipMemcpyDeviceToDevice, st) != hipSuccess) {
                if (!sh->rc.load()) { sh->rc = MIMEO_ERR_HIP; sh->err = "hipMemcpyAsync(unit HSPs) failed"; }
                break;
            }
            Group &g = b.groups[ui - sh->begin];
            memset(&g, 0, sizeof g);
            g.T = tv; g.Q = qv; g.tid = u.tid; g.qid = u.qid; g.minus = u.minus;
            g.hsp_begin = b.nh_total; g.hsp_end = b.nh_total + nh;
            b.group_pair[ui - sh->begin] = u.pair;
            b.nh_total += nh;
            g_stats.pair_strands++;
            g_stats.seed_hits += nhits;
            g_stats.hsps += nh;
            g_stats.query_bases_scanned += qs.len;
            g_stats.scan_bytes_algorithmic += scan_bytes_algorithmic(qs.len, nhits);
            g_stats.scan_bytes_kernel += scan_bytes_kernel(ti.n, qi.n, nhits);
            g_stats.scan_launches++;
        }
        // the copy reads unit_hsps, which the next unit of this lane overwrites: same stream, so ordered
    }
    sh->gate.lane_done(sh->rc.load() != 0);
    (void)hipStreamSynchronize(st);
    join_timing_flush(ln->jc);
    set_thread_stream(nullptr);
}

// mimeo_shutdown: give the lanes' work buffers and streams back
void release_pipeline_buffers() {
    for (Lane &l : g_lane) {
        if (l.st) (void)hipStreamSynchronize(l.st);
        l.jc.release();
        l.ew.release();
        l.hits.release();
        l.unit_hsps.release();
        if (l.heavy_end) { (void)hipEventDestroy(l.heavy_end); l.heavy_end = nullptr; }
        if (l.st) { (void)hipStreamDestroy(l.st); l.st = nullptr; }
    }
    g_hsp_batch.release();
    g_scratch.release();
    g_aln.release();
    g_groups.release();
}

int align_pairs_impl(const mimeo_genome *A, const mimeo_genome *B, const uint32_t *pair_t, const uint32_t *pair_q,
                     uint64_t npairs, const mimeo_params *p, mimeo_alignment **out, uint64_t *nout) {
    auto t0 = std::chrono::steady_clock::now();
    memset(&g_stats, 0, sizeof g_stats);
    const mimeo_genome *QG = B ? B : A;
    for (uint64_t k = 0; k < npairs; k++)
        if (pair_t[k] >= A->scaf.size() || pair_q[k] >= QG->scaf.size()) { set_error("pair index out of range"); return MIMEO_ERR_ARG; }
    int nlanes = getenv("MIMEO_LANES") ? atoi(getenv("MIMEO_LANES")) : 3;
    nlanes = std::max(1, std::min(MAX_LANES, nlanes));
    {
        // every lane owns hit, follower and queue buffers for its unit (36 bytes per expected seed hit, with the 1.5x
        // margin of the speculative launch): large scaffolds get fewer lanes so that the buffers stay within ~35 % of
        // the device memory (three lanes up to ~17 Mbp x 17 Mbp, one lane from ~30 Mbp x 30 Mbp)
        double worst = 0;
        for (uint64_t k = 0; k < npairs; k++)
            worst = std::max(worst, (double)A->scaf[pair_t[k]].len * (double)QG->scaf[pair_q[k]].len);
        const double per_lane = (13.0 * worst / 16777216.0 * 1.5 + 4194304.0) * 36.0;
        size_t free_b = 0, total_b = 0;
        HIP_TRY(hipMemGetInfo(&free_b, &total_b));
        double budget = 0.35 * (double)total_b;
        if (getenv("MIMEO_LANE_BUDGET_MB")) budget = (double)atol(getenv("MIMEO_LANE_BUDGET_MB")) * 1048576.0;
        nlanes = std::max(1, std::min(nlanes, (int)(budget / per_lane)));
    }
    for (int l = 0; l < nlanes; l++) {
        if (!g_lane[l].st) HIP_TRY(hipStreamCreateWithFlags(&g_lane[l].st, hipStreamNonBlocking));
        if (!g_lane[l].heavy_end) HIP_TRY(hipEventCreateWithFlags(&g_lane[l].heavy_end, hipEventDisableTiming));
        g_lane[l].tm = JoinTiming();
        g_lane[l].ms_ext = 0;
        g_lane[l].excess = 1.0;
    }
    static hipStream_t index_stream = nullptr;
    if (!index_stream) HIP_TRY(hipStreamCreateWithFlags(&index_stream, hipStreamNonBlocking));
    static Exclusive excl;  // events are created once; the lane set may change between calls
    {
        static bool ev_ok = false;
        if (!ev_ok) {
            for (int i = 0; i < 8; i++) {
                HIP_TRY(hipEventCreateWithFlags(&excl.reached[i], hipEventDisableTiming));
                HIP_TRY(hipEventCreateWithFlags(&excl.done[i], hipEventDisableTiming));
            }
            ev_ok = true;
        }
        excl.n = nlanes + 1;  // the index builder's stream also yields to the seed-scan fill
        for (int l = 0; l < nlanes; l++) excl.st[l] = g_lane[l].st;
        excl.st[nlanes] = index_stream;
    }
    const bool use_excl = !(getenv("MIMEO_NO_EXCL") && atoi(getenv("MIMEO_NO_EXCL")));
    // units in target-major order (stable in the caller's pair order): neighbouring units share the
    // target index, and the lanes work on neighbouring units
    std::vector<uint64_t> ord(npairs);
    for (uint64_t k = 0; k < npairs; k++) ord[k] = k;
    std::stable_sort(ord.begin(), ord.end(), [&](uint64_t a, uint64_t b) { return pair_t[a] < pair_t[b]; });
    std::vector<Unit> units;
    for (uint64_t k = 0; k < npairs; k++)
        for (uint32_t minus = 0; minus < 2; minus++)
            if (p->strand & (minus ? MIMEO_STRAND_MINUS : MIMEO_STRAND_PLUS))
                units.push_back(Unit{ord[k], pair_t[ord[k]], pair_q[ord[k]], minus});
    std::vector<std::vector<mimeo_alignment>> per_pair(npairs);
    // Seed indexes cost 64 MiB + 4 bytes per base and strand whatever the scaffold's length, so a genome of thousands
    // of scaffolds cannot keep them all (2000 scaffolds x 2 strands = 256 GB).  When the indexes a call needs exceed
    // the budget (60 % of the free device memory; MIMEO_INDEX_BUDGET_MB for tests) the pair matrix is cut into blocks of
    // Bt targets x Bq queries whose indexes fit, each block with an index cache of its own: S + 2 S^2 / Bt builds
    // instead of S + 2 S.  Results do not depend on the blocking (they are assembled per pair).
    std::vector<size_t> block_end;  // unit index where each index block ends
    {
        auto idx_bytes = [](uint64_t len) { return ((uint64_t)NBUCKET + 2) * 4 + len * 4; };
        std::map<uint32_t, uint64_t> tb, qb;  // bytes still to be built per distinct target / query scaffold
        for (const Unit &u : units) {
            const Scaffold &ts = A->scaf[u.tid], &qs = QG->scaf[u.qid];
            StrandView sv;
            if (!A->kept.count(IndexCache::kept_key(A, IndexCache::key_of(ts, 0, true, &sv)))) tb[u.tid] = idx_bytes(ts.len);
            if (!QG->kept.count(IndexCache::kept_key(QG, IndexCache::key_of(qs, (int)u.minus, false, &sv))))
                qb[u.qid] = std::max<uint64_t>(qb[u.qid], idx_bytes(qs.len) * ((p->strand & MIMEO_STRAND_BOTH) == MIMEO_STRAND_BOTH ? 2 : 1));
        }
        uint64_t need = 0, tmax = 1, qmax = 1;
        for (auto &kv : tb) { need += kv.second; tmax = std::max(tmax, kv.second); }
        for (auto &kv : qb) { need += kv.second; qmax = std::max(qmax, kv.second); }
        size_t free_b = 0, total_b = 0;
        HIP_TRY(hipMemGetInfo(&free_b, &total_b));
        uint64_t budget = (uint64_t)(0.6 * (double)free_b);
        if (getenv("MIMEO_INDEX_BUDGET_MB")) budget = (uint64_t)atol(getenv("MIMEO_INDEX_BUDGET_MB")) << 20;
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
    float ms_chain = 0, ms_gapped = 0, ms_index = 0;
    int rc = 0;
    const size_t MAX_GROUPS = 8192;  // units per K5/K6 batch
    HIP_TRY(hipStreamSynchronize(stream()));
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
    cache.start(index_stream);
    for (size_t b0 = blk_begin; b0 < blk_end && !rc; b0 += MAX_GROUPS) {
        size_t b1 = std::min(blk_end, b0 + MAX_GROUPS);
        Batch batch;
        batch.groups.resize(b1 - b0);
        batch.group_pair.resize(b1 - b0);
        Shared sh;
        sh.A = A; sh.QG = QG; sh.p = p; sh.units = &units; sh.begin = b0; sh.end = b1; sh.next = b0;
        sh.cache = &cache; sh.batch = &batch; sh.excl = use_excl ? &excl : nullptr;
        sh.speculative = !(getenv("MIMEO_NO_SPEC") && atoi(getenv("MIMEO_NO_SPEC")));
        sh.handover = !(getenv("MIMEO_NO_HANDOVER") && atoi(getenv("MIMEO_NO_HANDOVER")));
        sh.spec_shrink = getenv("MIMEO_SPEC_SHRINK") ? (uint64_t)atol(getenv("MIMEO_SPEC_SHRINK")) : 0;
        sh.gate.active = nlanes;
        std::vector<std::thread> th;
        for (int l = 1; l < nlanes; l++) th.emplace_back(lane_main, &g_lane[l], &sh);
        lane_main(&g_lane[0], &sh);  // the calling thread is lane 0
        for (auto &t : th) t.join();
        if (sh.rc.load()) { rc = sh.rc.load(); set_error(sh.err); break; }
        BatchResult r = run_batch(batch, p, &per_pair);
        g_stats.chained_hsps += r.chained;
        ms_chain += r.ms_chain;
        ms_gapped += r.ms_gapped;
        if (r.rc) { rc = r.rc; set_error(r.err); }
    }
    cache.clear();
    ms_index += cache.ms;
    g_stats.index_blocks++;
    g_stats.lanes = (uint64_t)nlanes;
    }  // index blocks
    if (rc) return rc;
    uint64_t total = 0;
    for (auto &v : per_pair) total += v.size();
    mimeo_alignment *res = (mimeo_alignment *)malloc((total ? total : 1) * sizeof(mimeo_alignment));
    if (!res) { set_error("host allocation failed"); return MIMEO_ERR_NOMEM; }
    uint64_t w = 0;
    for (auto &v : per_pair) { if (!v.empty()) memcpy(res + w, v.data(), v.size() * sizeof(mimeo_alignment)); w += v.size(); }
    *out = res;
    *nout = total;
    g_stats.alignments = total;
    g_stats.ms_index = ms_index;
    for (int l = 0; l < nlanes; l++) {
        g_stats.ms_scan += g_lane[l].tm.ms_count + g_lane[l].tm.ms_fill;
        g_stats.ms_scan_fill += g_lane[l].tm.ms_fill;
        g_stats.ms_extend += g_lane[l].ms_ext;
    }
    g_stats.ms_chain = ms_chain;
    g_stats.ms_gapped = ms_gapped;
    g_stats.ms_total = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
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

int chain_gapped_device(Group *d_groups, uint32_t ngroups, const mimeo_hsp *d_hsps, uint64_t nhsps,
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
    hipEvent_t e0, e1, e2;
    HIP_TRY(hipEventCreate(&e0)); HIP_TRY(hipEventCreate(&e1)); HIP_TRY(hipEventCreate(&e2));
    HIP_TRY(hipEventRecord(e0, st));
    if ((rc = chain_device(d_groups, ngroups, d_hsps, nhsps, p->chain, (mimeo_hsp *)(b + off_hs), (long long *)(b + off_best),
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
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); (void)hipEventDestroy(e2);
    return 0;
}

}  // namespace mimeo
