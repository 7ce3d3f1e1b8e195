// pipeline.hip — host orchestration of the per-pair loop of run_jobs.sh
// (src/mimeo/wrappers.py:1015-1059 in the reference): mimeo_align_pairs / mimeo_align_pair.
//
// Pairs are grouped by target scaffold.  A scaffold strand's seed index is built once and
// kept for the whole call (lastz rebuilds its table in every one of the S^2 invocations).  For
// every (target, query, strand) unit K3 (index join) and K4 (gap-free extension) run one unit at a
// time; the HSPs of up to MAX_GROUPS units are then chained and gap-extended together (K5/K6, one
// workgroup per unit in K5, one wavefront per half extension in K6).
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <future>
#include <map>
#include <tuple>

#include "common.h"

namespace mimeo {

extern mimeo_stats g_stats;

struct IndexCache {
    // key: (scaffold address, strand, target-role sv plane in use)
    std::map<std::tuple<const Scaffold *, int, int>, SeedIndex> m;
    float ms = 0;
    int get(const Scaffold &s, int minus, bool as_target, IndexView *out, StrandView *sv) {
        bool tsv = as_target && !minus && s.fwd.sv_target != nullptr;
        const Strand &st = minus ? s.rc : s.fwd;
        *sv = st.view(tsv);
        auto key = std::make_tuple(&s, minus, tsv ? 1 : 0);
        auto it = m.find(key);
        if (it == m.end()) {
            SeedIndex idx;
            int rc = build_index(*sv, idx, &ms);
            if (rc) return rc;
            it = m.emplace(key, idx).first;
        }
        *out = it->second.view();
        return 0;
    }
    void clear() {
        for (auto &kv : m) kv.second.release();
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

static DeviceBuf g_hits, g_unit_hsps, g_hsp_batch[2], g_scratch, g_aln, g_groups;
static hipStream_t g_stream2 = nullptr;
// MIMEO_OVERLAP=1 runs K5/K6 of a finished batch on a second stream while the next batch goes through
// K3/K4.  Off by default: on C2 the two phases compete for the same VALUs (step 0.585 s vs 0.561 s
// without overlap) and the co-running kernels distort the per-kernel timings.
static bool g_overlap = false;

// One batch of units whose HSPs are complete: K5 + K6 + read-back.  Runs on a worker thread with
// its own stream so that the latency-bound K6 rounds of batch b overlap K3/K4 of batch b+1.
struct Batch {
    std::vector<Group> groups;
    std::vector<uint64_t> group_pair;
    uint64_t nh_total = 0;
    int buf = 0;
};
struct BatchResult {
    int rc = 0;
    std::string err;
    uint64_t chained = 0;
    float ms_chain = 0, ms_gapped = 0;
};

static BatchResult run_batch(Batch b, const mimeo_params *p, std::vector<std::vector<mimeo_alignment>> *per_pair) {
    BatchResult r;
    (void)hipSetDevice(device_id());
    struct StreamScope {  // the worker's launches go to the second stream; restored on every return path
        StreamScope(hipStream_t s) { set_thread_stream(s); }
        ~StreamScope() { set_thread_stream(nullptr); }
    } scope(g_overlap ? g_stream2 : nullptr);
    hipStream_t st = stream();
    auto fail = [&](int rc) { r.rc = rc; r.err = last_error_copy(); return r; };
    if (!b.nh_total || b.groups.empty()) return r;
    int rc;
    if ((rc = g_groups.reserve(b.groups.size() * sizeof(Group)))) return fail(rc);
    if ((rc = g_aln.reserve(b.nh_total * sizeof(mimeo_alignment)))) return fail(rc);
    if (hipMemcpyAsync(g_groups.p, b.groups.data(), b.groups.size() * sizeof(Group), hipMemcpyHostToDevice, st) != hipSuccess)
        return fail(MIMEO_ERR_HIP);
    if ((rc = chain_gapped_device((Group *)g_groups.p, (uint32_t)b.groups.size(), (const mimeo_hsp *)g_hsp_batch[b.buf].p,
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
            set_error("gapped extension: DP band wider than 2048 columns, or score beyond int32: not supported yet");
            return fail(MIMEO_ERR_LIMIT);
        }
        r.chained += g.nchain;
        auto &dst = (*per_pair)[b.group_pair[gi]];
        dst.insert(dst.end(), host_aln.begin() + g.hsp_begin, host_aln.begin() + g.hsp_begin + g.naln);
    }
    return r;
}

int align_pairs_impl(const mimeo_genome *A, const mimeo_genome *B, const uint32_t *pair_t, const uint32_t *pair_q,
                     uint64_t npairs, const mimeo_params *p, mimeo_alignment **out, uint64_t *nout) {
    auto t0 = std::chrono::steady_clock::now();
    memset(&g_stats, 0, sizeof g_stats);
    const mimeo_genome *QG = B ? B : A;
    for (uint64_t k = 0; k < npairs; k++)
        if (pair_t[k] >= A->scaf.size() || pair_q[k] >= QG->scaf.size()) { set_error("pair index out of range"); return MIMEO_ERR_ARG; }
    if (!g_stream2) HIP_TRY(hipStreamCreateWithFlags(&g_stream2, hipStreamNonBlocking));
    g_overlap = getenv("MIMEO_OVERLAP") && atoi(getenv("MIMEO_OVERLAP")) != 0;
    // stable grouping of pair indices by target
    std::vector<uint64_t> ord(npairs);
    for (uint64_t k = 0; k < npairs; k++) ord[k] = k;
    std::stable_sort(ord.begin(), ord.end(), [&](uint64_t a, uint64_t b) { return pair_t[a] < pair_t[b]; });
    std::vector<std::vector<mimeo_alignment>> per_pair(npairs);
    IndexCache cache;
    JoinTiming tm;
    float ms_ext = 0, ms_chain = 0, ms_gapped = 0;
    hipStream_t st = stream();
    int rc = 0;
    uint64_t pos = 0;
    // units accumulate into a batch; a full batch is handed to the worker (K5/K6) while the next one
    // is being filled.  Four batches per call (at most MAX_GROUPS units each).
    uint64_t nunits = 0;
    for (int m = 0; m < 2; m++) if (p->strand & (m ? MIMEO_STRAND_MINUS : MIMEO_STRAND_PLUS)) nunits += npairs;
    const size_t MAX_GROUPS = 8192;
    const size_t batch_groups = g_overlap ? std::max<size_t>(8, std::min<size_t>(MAX_GROUPS, (nunits + 3) / 4)) : MAX_GROUPS;
    Batch cur;
    cur.buf = 0;
    std::future<BatchResult> pending[2];
    auto collect = [&](int buf) -> int {  // wait for the batch that used buffer `buf`
        if (!pending[buf].valid()) return 0;
        BatchResult r = pending[buf].get();
        g_stats.chained_hsps += r.chained;
        ms_chain += r.ms_chain;
        ms_gapped += r.ms_gapped;
        if (r.rc) { set_error(r.err); return r.rc; }
        return 0;
    };
    auto flush = [&]() -> int {
        HIP_TRY(hipStreamSynchronize(st));  // the batch's HSP copies are complete
        int buf = cur.buf;
        // one worker at a time (they share the K5/K6 work buffers); this also frees the other HSP
        // buffer, which is the one refilled next
        int prc = collect(buf ^ 1);
        if (prc) return prc;
        pending[buf] = std::async(g_overlap ? std::launch::async : std::launch::deferred, run_batch, std::move(cur), p, &per_pair);
        if (!g_overlap) { int r0 = collect(buf); if (r0) return r0; }
        cur = Batch();
        cur.buf = buf ^ 1;
        return 0;
    };
    while (pos < npairs && !rc) {
        uint32_t tid = pair_t[ord[pos]];
        uint64_t end = pos;
        while (end < npairs && pair_t[ord[end]] == tid) end++;
        const Scaffold &ts = A->scaf[tid];
        IndexView ti;
        StrandView tv;
        if ((rc = cache.get(ts, 0, true, &ti, &tv))) break;
        for (uint64_t k = pos; k < end && !rc; k++) {
            uint32_t qid = pair_q[ord[k]];
            const Scaffold &qs = QG->scaf[qid];
            for (int minus = 0; minus < 2 && !rc; minus++) {
                if (!(p->strand & (minus ? MIMEO_STRAND_MINUS : MIMEO_STRAND_PLUS))) continue;
                IndexView qi;
                StrandView qv;
                if ((rc = cache.get(qs, minus, false, &qi, &qv))) break;
                uint64_t nhits = 0, nh = 0;
                if ((rc = join_hits(ti, qi, p->transitions, g_hits, &nhits, &tm))) break;
                if ((rc = ungapped_hsps_device(tv, qv, (const uint2 *)g_hits.p, nhits, p, g_unit_hsps, &nh, &ms_ext))) break;
                // append this unit's HSPs to the batch-level array
                DeviceBuf &hb = g_hsp_batch[cur.buf];
                if ((cur.nh_total + nh) * sizeof(mimeo_hsp) > hb.cap) {
                    DeviceBuf bigger;
                    if ((rc = bigger.reserve((cur.nh_total + nh) * 2 * sizeof(mimeo_hsp) + 4096))) break;
                    if (cur.nh_total) HIP_TRY(hipMemcpyAsync(bigger.p, hb.p, cur.nh_total * sizeof(mimeo_hsp), hipMemcpyDeviceToDevice, st));
                    HIP_TRY(hipStreamSynchronize(st));
                    hb.release();
                    hb = bigger;
                }
                if (nh) HIP_TRY(hipMemcpyAsync((char *)hb.p + cur.nh_total * sizeof(mimeo_hsp), g_unit_hsps.p, nh * sizeof(mimeo_hsp), hipMemcpyDeviceToDevice, st));
                Group g;
                memset(&g, 0, sizeof g);
                g.T = tv; g.Q = qv; g.tid = tid; g.qid = qid; g.minus = (uint32_t)minus;
                g.hsp_begin = cur.nh_total; g.hsp_end = cur.nh_total + nh;
                cur.groups.push_back(g);
                cur.group_pair.push_back(ord[k]);
                cur.nh_total += nh;
                g_stats.pair_strands++;
                g_stats.seed_hits += nhits;
                g_stats.hsps += nh;
                g_stats.query_bases_scanned += qs.len;
                g_stats.scan_bytes_algorithmic += scan_bytes_algorithmic(qs.len, nhits);
                g_stats.scan_bytes_kernel += scan_bytes_kernel(ti.n, qi.n, nhits);
                g_stats.scan_launches++;
                if (cur.groups.size() >= batch_groups) rc = flush();
            }
        }
        pos = end;
    }
    if (!rc && !cur.groups.empty()) rc = flush();
    for (int b = 0; b < 2; b++) { int r2 = collect(b); if (!rc) rc = r2; }
    cache.clear();
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
    g_stats.ms_index = cache.ms;
    g_stats.ms_scan = tm.ms_count + tm.ms_fill;
    g_stats.ms_scan_fill = tm.ms_fill;
    g_stats.ms_extend = ms_ext;
    g_stats.ms_chain = ms_chain;
    g_stats.ms_gapped = ms_gapped;
    g_stats.ms_total = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
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
