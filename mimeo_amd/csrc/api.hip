// api.hip — the C-ABI of libmimeo_hip.so (include/mimeo_hip.h).  Host-side runtime only:
// device selection, streams, error strings, genome handles, stage orchestration.
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "common.h"

namespace mimeo {

static thread_local std::string g_err;
static hipStream_t g_stream = nullptr;
static bool g_init = false;
static int g_device = -1;
mimeo_stats g_stats;

void set_error(const std::string &msg) { g_err = msg; }
int hip_fail(hipError_t e, const char *what, const char *file, int line) {
    char buf[512];
    snprintf(buf, sizeof buf, "HIP error %d (%s) in %s at %s:%d", (int)e, hipGetErrorString(e), what, file, line);
    g_err = buf;
    if (e == hipErrorOutOfMemory) { (void)hipGetLastError(); return MIMEO_ERR_NOMEM; }   // not sticky: the caller may retry smaller
    return MIMEO_ERR_HIP;
}
// a worker thread of the library may run its launches on a stream of its own
static thread_local hipStream_t tls_stream = nullptr;
hipStream_t stream() { return tls_stream ? tls_stream : g_stream; }
void set_thread_stream(hipStream_t s) { tls_stream = s; }
int device_id() { return g_device; }
std::string last_error_copy() { return g_err; }
bool initialised() { return g_init; }

static int need_init() {
    if (!g_init) {
        set_error("mimeo_init() has not been called or no gfx950 device is available (no CPU fallback exists)");
        return MIMEO_ERR_NO_DEVICE;
    }
    return 0;
}

}  // namespace mimeo

using namespace mimeo;

extern "C" {

int mimeo_abi_version(void) { return MIMEO_ABI_VERSION; }

const char *mimeo_last_error(void) { return g_err.c_str(); }

int mimeo_init(int device) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        set_error("no HIP device visible: the mimeo engine has no CPU fallback");
        return MIMEO_ERR_NO_DEVICE;
    }
    if (device < 0 || device >= n) { set_error("device index out of range"); return MIMEO_ERR_ARG; }
    HIP_TRY(hipSetDevice(device));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        set_error(std::string("device is ") + prop.gcnArchName + ", this library is built for gfx950 only");
        return MIMEO_ERR_NO_DEVICE;
    }
    if (!g_stream) HIP_TRY(hipStreamCreateWithFlags(&g_stream, hipStreamNonBlocking));
    g_device = device;
    g_init = true;
    memset(&g_stats, 0, sizeof g_stats);
    return MIMEO_OK;
}

void mimeo_shutdown(void) {
    if (g_init) release_pipeline_buffers();
    if (g_stream) { (void)hipStreamSynchronize(g_stream); (void)hipStreamDestroy(g_stream); g_stream = nullptr; }
    g_init = false;
}

int mimeo_params_default(mimeo_params *p) {
    if (!p) { set_error("null params"); return MIMEO_ERR_ARG; }
    memset(p, 0, sizeof *p);
    p->hspthresh = 3000; p->xdrop = 910; p->ydrop = 9400; p->gap_open = 400; p->gap_extend = 30;
    p->transitions = 1; p->entropy = 1; p->chain = 1; p->gapped = 1; p->strand = MIMEO_STRAND_BOTH;
    return MIMEO_OK;
}

int mimeo_get_stats(mimeo_stats *out) {
    if (!out) { set_error("null stats"); return MIMEO_ERR_ARG; }
    *out = g_stats;
    return MIMEO_OK;
}

void mimeo_free(void *p) { free(p); }

int mimeo_genome_create(uint32_t nscaf, const uint8_t *bases, const uint64_t *offsets, mimeo_genome **out) {
    int rc = need_init();
    if (rc) return rc;
    if (!out || !offsets || (!bases && nscaf && offsets[nscaf] > 0)) { set_error("null argument"); return MIMEO_ERR_ARG; }
    mimeo_genome *g = new mimeo_genome();
    g->scaf.resize(nscaf);
    uint64_t maxlen = 0;
    for (uint32_t i = 0; i < nscaf; i++) {
        if (offsets[i + 1] < offsets[i]) { delete g; set_error("offsets not monotone"); return MIMEO_ERR_ARG; }
        maxlen = std::max<uint64_t>(maxlen, offsets[i + 1] - offsets[i]);
    }
    uint8_t *d_ascii = nullptr;
    hipError_t e = hipMalloc((void **)&d_ascii, maxlen ? maxlen : 1);
    if (e != hipSuccess) { delete g; return hip_fail(e, "hipMalloc(ascii)", __FILE__, __LINE__); }
    for (uint32_t i = 0; i < nscaf && !rc; i++) {
        uint64_t len = offsets[i + 1] - offsets[i];
        if (len) {
            e = hipMemcpyAsync(d_ascii, bases + offsets[i], len, hipMemcpyHostToDevice, stream());
            if (e != hipSuccess) { rc = hip_fail(e, "hipMemcpy(ascii)", __FILE__, __LINE__); break; }
        }
        rc = pack_scaffold(d_ascii, len, g->scaf[i]);
    }
    (void)hipStreamSynchronize(stream());
    (void)hipFree(d_ascii);
    if (rc) { mimeo_genome_destroy(g); return rc; }
    *out = g;
    return MIMEO_OK;
}

void mimeo_genome_destroy(mimeo_genome *g) {
    if (!g) return;
    for (auto &kv : g->kept) kv.second.release();
    g->kept.clear();
    for (auto &s : g->scaf) free_scaffold(s);
    delete g;
}

int mimeo_genome_keep_indexes(mimeo_genome *g, int keep) {
    if (!g) { set_error("null argument"); return MIMEO_ERR_ARG; }
    g->keep_indexes = keep != 0;
    return MIMEO_OK;
}

int mimeo_genome_build_indexes(mimeo_genome *g, const uint32_t *scaf, uint64_t n) {
    int rc = need_init();
    if (rc) return rc;
    if (!g || (n && !scaf)) { set_error("null argument"); return MIMEO_ERR_ARG; }
    for (uint64_t i = 0; i < n; i++)
        if (scaf[i] >= g->scaf.size()) { set_error("bad scaffold id"); return MIMEO_ERR_ARG; }
    g->keep_indexes = true;
    return build_kept_indexes(g, scaf, n);
}

int mimeo_genome_drop_indexes(mimeo_genome *g, const uint32_t *scaf, uint64_t n) {
    if (!g || (n && !scaf)) { set_error("null argument"); return MIMEO_ERR_ARG; }
    for (uint64_t i = 0; i < n; i++)
        if (scaf[i] >= g->scaf.size()) { set_error("bad scaffold id"); return MIMEO_ERR_ARG; }
    for (auto it = g->kept.begin(); it != g->kept.end();) {
        bool drop = n == 0;
        for (uint64_t i = 0; i < n && !drop; i++) drop = std::get<0>(it->first) == scaf[i];
        if (drop) { it->second.release(); it = g->kept.erase(it); }
        else ++it;
    }
    return MIMEO_OK;
}

int mimeo_genome_nscaf(const mimeo_genome *g, uint32_t *nscaf) {
    if (!g || !nscaf) { set_error("null argument"); return MIMEO_ERR_ARG; }
    *nscaf = (uint32_t)g->scaf.size();
    return MIMEO_OK;
}

int mimeo_genome_load_fasta(const char *const *paths, uint32_t npaths, const char *split_dir, mimeo_genome **out) {
    int rc = need_init();
    if (rc) return rc;
    if (!out || (!paths && npaths)) { set_error("null argument"); return MIMEO_ERR_ARG; }
    return load_fasta_impl(paths, npaths, split_dir, out);
}

int mimeo_genome_name(const mimeo_genome *g, uint32_t scaf, const char **name) {
    if (!g || !name || scaf >= g->scaf.size()) { set_error("bad scaffold id"); return MIMEO_ERR_ARG; }
    *name = scaf < g->names.size() ? g->names[scaf].c_str() : "";
    return MIMEO_OK;
}

int mimeo_genome_length(const mimeo_genome *g, uint32_t scaf, uint64_t *length) {
    if (!g || !length || scaf >= g->scaf.size()) { set_error("bad scaffold id"); return MIMEO_ERR_ARG; }
    *length = g->scaf[scaf].len;
    return MIMEO_OK;
}

// parameter domain of the kernels (include/mimeo_hip.h, mimeo_params)
static int check_params(const mimeo_params *p) {
    // the gap-free walk applies four columns at a time; that is exact only while four HOXD70 columns
    // (at most 4 * 125) cannot exceed the x-drop
    if (p->xdrop < 500) { set_error("xdrop below 500 is not supported (the extension walks four columns per step)"); return MIMEO_ERR_ARG; }
    if (p->xdrop > 30000 || p->hspthresh < 0 || p->ydrop < 0 || p->gap_open < 0 || p->gap_extend <= 0) {
        set_error("alignment parameter out of range");
        return MIMEO_ERR_ARG;
    }
    if (!(p->strand & MIMEO_STRAND_BOTH)) { set_error("strand selects nothing"); return MIMEO_ERR_ARG; }
    return 0;
}

static int check_pair(const mimeo_genome *T, uint32_t tid, const mimeo_genome *Q, uint32_t qid, const mimeo_params *p,
                      const void *out, const void *nout) {
    int rc = need_init();
    if (rc) return rc;
    if (!T || !Q || !p || !out || !nout) { set_error("null argument"); return MIMEO_ERR_ARG; }
    if ((rc = check_params(p))) return rc;
    if (tid >= T->scaf.size() || qid >= Q->scaf.size()) { set_error("scaffold id out of range"); return MIMEO_ERR_ARG; }
    return 0;
}

int mimeo_seed_hits(const mimeo_genome *T, uint32_t tid, const mimeo_genome *Q, uint32_t qid, uint32_t qstrand,
                    const mimeo_params *p, mimeo_seed_hit **out, uint64_t *nout) {
    int rc = check_pair(T, tid, Q, qid, p, out, nout);
    if (rc) return rc;
    auto t0 = std::chrono::steady_clock::now();
    memset(&g_stats, 0, sizeof g_stats);
    const Scaffold &ts = T->scaf[tid], &qs = Q->scaf[qid];
    SeedIndex it, iq;
    float ms_index = 0;
    if ((rc = build_index(ts.fwd.view(true), it, &ms_index))) return rc;
    if ((rc = build_index((qstrand ? qs.rc : qs.fwd).view(false), iq, &ms_index))) { it.release(); return rc; }
    DeviceBuf hits;
    uint64_t n = 0;
    JoinTiming tm;
    static JoinCtx jc;
    rc = join_hits(jc, it.view(), iq.view(), p->transitions, hits, &n, &tm);
    if (!rc) {
        mimeo_seed_hit *h = (mimeo_seed_hit *)malloc((n ? n : 1) * sizeof(mimeo_seed_hit));
        if (!h) { set_error("host allocation failed"); rc = MIMEO_ERR_NOMEM; }
        else {
            hipError_t e = hipSuccess;
            if (n) e = hipMemcpy(h, hits.p, n * sizeof(mimeo_seed_hit), hipMemcpyDeviceToHost);
            if (e != hipSuccess) { free(h); rc = hip_fail(e, "hipMemcpy(hits)", __FILE__, __LINE__); }
            else { *out = h; *nout = n; }
        }
    }
    hits.release();
    it.release();
    iq.release();
    g_stats.pair_strands = 1;
    g_stats.seed_hits = n;
    g_stats.query_bases_scanned = qs.len;
    g_stats.ms_index = ms_index;
    g_stats.ms_scan = tm.ms_count + tm.ms_fill;
    g_stats.ms_scan_fill = tm.ms_fill;
    g_stats.scan_launches = 1;
    g_stats.ms_total = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return rc;
}

int mimeo_chain_hsps(const mimeo_hsp *in, uint64_t n, mimeo_hsp *out) {
    int rc = need_init();
    if (rc) return rc;
    if (!n) return 0;
    if (!in || !out) { set_error("mimeo_chain_hsps: null argument"); return MIMEO_ERR_ARG; }
    if (n >= (1ull << 32)) { set_error("more than 2^32 HSPs"); return MIMEO_ERR_LIMIT; }
    DeviceBuf d_in, d_unit, d_group, d_sorted, d_best, d_cand, d_pred, d_order;
    auto done = [&](int r) {
        d_in.release(); d_unit.release(); d_group.release(); d_sorted.release(); d_best.release(); d_cand.release(); d_pred.release(); d_order.release();
        return r;
    };
    if ((rc = d_in.reserve(n * sizeof(mimeo_hsp))) || (rc = d_unit.reserve(n * 4)) || (rc = d_group.reserve(sizeof(Group))) ||
        (rc = d_sorted.reserve(n * sizeof(mimeo_hsp))) || (rc = d_best.reserve(n * 8)) || (rc = d_cand.reserve(n * 8)) ||
        (rc = d_pred.reserve(n * 4)) || (rc = d_order.reserve(n * 4)))
        return done(rc);
    hipError_t e = hipMemcpy(d_in.p, in, n * sizeof(mimeo_hsp), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemset(d_unit.p, 0, n * 4);
    if (e == hipSuccess) e = hipMemset(d_group.p, 0, sizeof(Group));
    if (e != hipSuccess) return done(hip_fail(e, "mimeo_chain_hsps: upload", __FILE__, __LINE__));
    if ((rc = chain_device((Group *)d_group.p, 1, (const mimeo_hsp *)d_in.p, (const uint32_t *)d_unit.p, n, 1, (mimeo_hsp *)d_sorted.p,
                           (long long *)d_best.p, (long long *)d_cand.p, (int *)d_pred.p, (uint32_t *)d_order.p)))
        return done(rc);
    e = hipStreamSynchronize(stream());
    if (e == hipSuccess) e = hipMemcpy(out, d_sorted.p, n * sizeof(mimeo_hsp), hipMemcpyDeviceToHost);
    if (e != hipSuccess) return done(hip_fail(e, "mimeo_chain_hsps: download", __FILE__, __LINE__));
    return done(0);
}

int mimeo_ungapped_hsps(const mimeo_genome *T, uint32_t tid, const mimeo_genome *Q, uint32_t qid, uint32_t qstrand,
                        const mimeo_params *p, mimeo_hsp **out, uint64_t *nout) {
    int rc = check_pair(T, tid, Q, qid, p, out, nout);
    if (rc) return rc;
    auto t0 = std::chrono::steady_clock::now();
    memset(&g_stats, 0, sizeof g_stats);
    const Scaffold &ts = T->scaf[tid], &qs = Q->scaf[qid];
    SeedIndex it, iq;
    float ms_index = 0;
    mimeo_hsp *h = nullptr;
    uint64_t nh = 0;
    ExtStats est;
    do {
        // the production path of mimeo_align_pairs up to K4, on a batch of one unit
        std::vector<UnitWork> work(1);
        memset(&work[0], 0, sizeof(UnitWork));
        work[0].d.T = ts.fwd.view(true);
        work[0].d.Q = (qstrand ? qs.rc : qs.fwd).view(false);
        work[0].d.same = (work[0].d.T.pw == work[0].d.Q.pw && work[0].d.T.len == work[0].d.Q.len && !getenv("MIMEO_NO_DIAG0")) ? 1u : 0u;
        if ((rc = build_index(work[0].d.T, it, &ms_index))) break;
        if ((rc = build_index(work[0].d.Q, iq, &ms_index))) break;
        work[0].ti = it.view();
        work[0].qi = iq.view();
        std::vector<std::vector<mimeo_hsp>> per_unit;
        if ((rc = ungapped_units(work, p, &per_unit, &est))) break;
        nh = per_unit[0].size();
        h = (mimeo_hsp *)malloc((nh ? nh : 1) * sizeof(mimeo_hsp));
        if (!h) { set_error("host allocation failed"); rc = MIMEO_ERR_NOMEM; break; }
        if (nh) memcpy(h, per_unit[0].data(), nh * sizeof(mimeo_hsp));
        std::sort(h, h + nh, [](const mimeo_hsp &a, const mimeo_hsp &b) {
            int64_t da = (int64_t)a.tstart - a.qstart, db = (int64_t)b.tstart - b.qstart;
            if (da != db) return da < db;
            if (a.tstart != b.tstart) return a.tstart < b.tstart;
            return a.length < b.length;
        });
    } while (0);
    it.release(); iq.release();
    if (rc) { free(h); return rc; }
    *out = h;
    *nout = nh;
    g_stats.pair_strands = 1;
    g_stats.seed_hits = est.seed_hits;
    g_stats.hsps = nh;
    g_stats.query_bases_scanned = qs.len;
    g_stats.scan_bytes_algorithmic = est.scan_bytes_algorithmic;
    g_stats.scan_bytes_kernel = est.scan_bytes_kernel;
    g_stats.walked_hits = est.walked;
    g_stats.followers = est.followers;
    g_stats.queue_reruns = est.reruns;
    g_stats.batches = 1;
    g_stats.ms_index = ms_index;
    g_stats.ms_scan = est.ms_heavy;
    g_stats.ms_scan_fill = est.ms_k34;
    g_stats.scan_launches = est.heavy_launches;
    g_stats.scan_kernel_launches = est.heavy_kernel_launches;
    g_stats.ms_extend = est.ms_tails;
    g_stats.ms_total = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return MIMEO_OK;
}

int mimeo_align_pairs(const mimeo_genome *A, const mimeo_genome *B, const uint32_t *pair_t, const uint32_t *pair_q,
                      uint64_t npairs, const mimeo_params *p, mimeo_alignment **out, uint64_t *nout) {
    int rc = need_init();
    if (rc) return rc;
    if (!A || !p || !out || !nout || (npairs && (!pair_t || !pair_q))) { set_error("null argument"); return MIMEO_ERR_ARG; }
    if ((rc = check_params(p))) return rc;
    return align_units_impl(A, B == A ? nullptr : B, pair_t, pair_q, nullptr, npairs, p, out, nout);
}

int mimeo_align_units(const mimeo_genome *A, const mimeo_genome *B, const uint32_t *pair_t, const uint32_t *pair_q,
                      const uint8_t *pair_strand, uint64_t npairs, const mimeo_params *p, mimeo_alignment **out, uint64_t *nout) {
    int rc = need_init();
    if (rc) return rc;
    if (!A || !p || !out || !nout || (npairs && (!pair_t || !pair_q))) { set_error("null argument"); return MIMEO_ERR_ARG; }
    if ((rc = check_params(p))) return rc;
    return align_units_impl(A, B == A ? nullptr : B, pair_t, pair_q, pair_strand, npairs, p, out, nout);
}

int mimeo_get_failed_pairs(uint64_t *pair_index, int32_t *code, uint64_t cap, uint64_t *n) {
    if (!n) { set_error("null argument"); return MIMEO_ERR_ARG; }
    const auto &f = failed_pairs();
    *n = f.size();
    for (uint64_t i = 0; i < f.size() && i < cap; i++) {
        if (pair_index) pair_index[i] = f[i].first;
        if (code) code[i] = f[i].second;
    }
    return MIMEO_OK;
}

int mimeo_align_pair(const mimeo_genome *T, uint32_t tid, const mimeo_genome *Q, uint32_t qid, const mimeo_params *p,
                     mimeo_alignment **out, uint64_t *nout) {
    int rc = check_pair(T, tid, Q, qid, p, out, nout);
    if (rc) return rc;
    return align_units_impl(T, Q == T ? nullptr : Q, &tid, &qid, nullptr, 1, p, out, nout);
}

int mimeo_coverage_collapse(const mimeo_interval *iv, uint64_t n, const uint32_t *chrom_len, uint32_t nchrom,
                            uint32_t min_cov, uint32_t min_len, mimeo_interval **out, uint64_t *nout) {
    int rc = need_init();
    if (rc) return rc;
    if (!out || !nout || (n && !iv) || (nchrom && !chrom_len)) { set_error("null argument"); return MIMEO_ERR_ARG; }
    std::vector<mimeo_interval> res;
    float ms = 0;
    auto t0 = std::chrono::steady_clock::now();
    if ((rc = coverage_collapse_device(iv, n, chrom_len, nchrom, min_cov, min_len, res, &ms))) return rc;
    mimeo_interval *r = (mimeo_interval *)malloc((res.size() ? res.size() : 1) * sizeof(mimeo_interval));
    if (!r) { set_error("host allocation failed"); return MIMEO_ERR_NOMEM; }
    if (!res.empty()) memcpy(r, res.data(), res.size() * sizeof(mimeo_interval));
    *out = r;
    *nout = res.size();
    g_stats.ms_collapse = ms;
    (void)t0;
    return MIMEO_OK;
}

int mimeo_coverage_bedgraph(const mimeo_interval *iv, uint64_t n, const uint32_t *chrom_len, uint32_t nchrom, mimeo_depth_run **out,
                            uint64_t *nout) {
    int rc = need_init();
    if (rc) return rc;
    if (!out || !nout || (n && !iv) || (nchrom && !chrom_len)) { set_error("null argument"); return MIMEO_ERR_ARG; }
    std::vector<mimeo_depth_run> res;
    if ((rc = coverage_bedgraph_device(iv, n, chrom_len, nchrom, res))) return rc;
    mimeo_depth_run *r = (mimeo_depth_run *)malloc((res.size() ? res.size() : 1) * sizeof(mimeo_depth_run));
    if (!r) { set_error("host allocation failed"); return MIMEO_ERR_NOMEM; }
    if (!res.empty()) memcpy(r, res.data(), res.size() * sizeof(mimeo_depth_run));
    *out = r;
    *nout = res.size();
    return MIMEO_OK;
}

int mimeo_tandem_masked(const mimeo_genome *A, const mimeo_interval *iv, uint64_t n, int32_t match, int32_t mismatch,
                        int32_t delta, int32_t minscore, int32_t maxperiod, uint32_t *masked) {
    int rc = need_init();
    if (rc) return rc;
    if (!A || (n && (!iv || !masked))) { set_error("null argument"); return MIMEO_ERR_ARG; }
    return tandem_masked_device(A, iv, n, match, mismatch, delta, minscore, maxperiod, masked);
}

}  // extern "C"
