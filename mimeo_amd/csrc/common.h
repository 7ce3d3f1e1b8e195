// common.h — shared declarations of the mimeo HIP engine (gfx950 only).
//
// Data layout in HBM:
//   a scaffold strand is stored as four 1-bit planes, interleaved word by word: pw[w] is a uint4
//   holding 32 bases (base i at bit (i & 31) of word (i >> 5)) of
//     .x lo — bit 0 of the 2-bit code (A0 C1 G2 T3)
//     .y hi — bit 1 of the 2-bit code; a transition (A<->G, C<->T) flips exactly this bit
//     .z nm — 1 where the base is not ACGT (scored as N)
//     .w sv — 1 where a valid 12of19 seed word starts in the QUERY role (no N inside the 19-base
//             window, and the window fits)
//   so one 64-byte cache line carries all planes of 128 consecutive bases: an extension touches
//   one or two lines per sequence instead of one per plane.  A scaffold with lower-case bases has
//   an extra plane svt = seed validity in the TARGET role (lastz excludes soft-masked target
//   bases from seeding).  PLANE_PAD zero words lie in front of word 0 and behind the last word
//   so that window loads never need a bounds check.
//   A second copy holds the lo / hi planes alone (uint2 per 32 bases, same padding): the K4 pre-filter gathers
//   ~200 bases around every seed hit and is bound by the bytes it pulls through L2, so it reads this
//   copy (6.4 + 4 = 10.4 bits per base and strand in all).
//   The seed index of a strand is CSR: off[2^24 + 1] (u32) and pos[nvalid] (u32, ascending
//   inside a bucket).  Key = (pext12(lo window) << 12) | pext12(hi window): the low 12 bits are
//   the transition bits, so the 13 words within one transition of a key differ only in the low
//   12 bits and a block of 4096 consecutive keys ("tile") is closed under them.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <functional>
#include <map>
#include <tuple>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/mimeo_hip.h"

namespace mimeo {

constexpr int SEED_LEN = 19;
constexpr int SEED_WEIGHT = 12;
constexpr uint32_t NBUCKET = 1u << 24;
constexpr uint32_t TILE_WORDS = 4096;  // keys per join tile
constexpr uint32_t NTILE = NBUCKET / TILE_WORDS;
constexpr int PLANE_PAD = 8;  // u32 words (256 bases) of zero padding on both sides

void set_error(const std::string &msg);
int hip_fail(hipError_t e, const char *what, const char *file, int line);

#define HIP_TRY(call)                                                         \
    do {                                                                      \
        hipError_t e__ = (call);                                              \
        if (e__ != hipSuccess) return mimeo::hip_fail(e__, #call, __FILE__, __LINE__); \
    } while (0)

// device views ---------------------------------------------------------------------------
struct StrandView {
    const uint4 *pw;       // interleaved planes {lo, hi, nm, sv}, pointer to word 0
    const uint32_t *svt;   // target-role seed validity plane (word 0), or null: use pw[].w
    const uint2 *p2;       // the same lo / hi planes alone, 8 bytes per 32 bases (K4 pre-filter: half the bytes to gather)
    uint32_t len;          // bases
    uint32_t has_n;        // the strand holds at least one non-ACGT base
};

struct IndexView {
    const uint32_t *off;  // NBUCKET + 1
    const uint32_t *pos;
    uint32_t n;  // number of indexed positions
};

// host-side owner of one strand's planes
struct Strand {
    uint4 *base = nullptr;          // nwords + 2*PLANE_PAD interleaved words
    uint2 *slim = nullptr;          // nwords + 2*PLANE_PAD words of {lo, hi} only
    bool has_n = false;
    uint32_t *sv_target = nullptr;  // separate sv plane for the target role (soft-mask aware); null = same as sv
    uint32_t nwords = 0;
    uint32_t len = 0;
    StrandView view(bool as_target) const;
};

struct SeedIndex {
    uint32_t *off = nullptr;
    uint32_t *pos = nullptr;
    uint32_t n = 0;
    size_t pos_bytes = 0;  // allocation size of pos (free-list key)
    IndexView view() const { return IndexView{off, pos, n}; }
    void release();
};

struct Scaffold {
    uint64_t len = 0;
    bool has_lower = false;
    Strand fwd, rc;
};

}  // namespace mimeo

struct mimeo_genome {
    std::vector<mimeo::Scaffold> scaf;
    std::vector<std::string> names;  // record ids when loaded from FASTA (ingest.hip); else empty
    // seed indexes kept across mimeo_align_pairs calls (mimeo_genome_keep_indexes); key =
    // (scaffold number, strand, target-role sv plane in use).  Only touched between calls and by the
    // calling thread at the start / end of a call.
    bool keep_indexes = false;
    mutable std::map<std::tuple<uint32_t, int, int>, mimeo::SeedIndex> kept;
    bool owns(const mimeo::Scaffold *s) const { return !scaf.empty() && s >= scaf.data() && s < scaf.data() + scaf.size(); }
};

namespace mimeo {

// stream used by every kernel launch of the library (created by mimeo_init)
hipStream_t stream();
void set_thread_stream(hipStream_t s);  // per-thread override (worker threads)
int device_id();
std::string last_error_copy();
bool initialised();

// K1: ASCII -> planes for both strands (k1_pack.hip)
int pack_scaffold(const uint8_t *d_ascii, uint64_t len, Scaffold &out);
void free_scaffold(Scaffold &s);

// streaming FASTA ingest (ingest.hip)
int load_fasta_impl(const char *const *paths, uint32_t npaths, const char *split_dir, mimeo_genome **out);

// K2: seed index of one strand (k2_index.hip)
// [p0, p1): index only the seed words that start in that range (chunks of a very large query; default = all)
int build_index(const StrandView &s, SeedIndex &out, float *ms, uint32_t p0 = 0, uint32_t p1 = 0xFFFFFFFFu);

// K3: index join (k3_join.hip).  Produces all seed hits of (T, Q) into a device buffer the
// caller owns (grown on demand).
struct DeviceBuf {
    void *p = nullptr;
    size_t cap = 0;
    int reserve(size_t bytes);
    void release();
};
struct JoinTiming { float ms_count = 0, ms_fill = 0; };
// per-lane work state of K3 (a lane = one host thread + one stream working through units)
struct JoinCtx {
    unsigned long long *tile_count = nullptr, *tile_base = nullptr;
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
    JoinTiming *pending_tm = nullptr;  // a speculative launch whose event times have not been read yet
    const unsigned long long *total_dev() const { return tile_base + NTILE; }  // hit count of the last join
    void release();
};
// Lets one kernel run alone on the device while several lanes (streams) are active: begin() makes
// `mine` wait for everything the other streams have been given so far, end() makes the other
// streams wait for what `mine` was given in between.  Used for the bandwidth-bound seed-scan fill.
struct Exclusive {
    std::mutex mu;
    int n = 0;
    hipStream_t st[8];
    hipEvent_t reached[8], done[8];
    int index_of(hipStream_t s) const { for (int i = 0; i < n; i++) if (st[i] == s) return i; return -1; }
    void begin(hipStream_t mine) {
        mu.lock();
        for (int i = 0; i < n; i++)
            if (st[i] != mine) { (void)hipEventRecord(reached[i], st[i]); (void)hipStreamWaitEvent(mine, reached[i], 0); }
    }
    void end(hipStream_t mine) {
        int me = index_of(mine);
        if (me >= 0) {
            (void)hipEventRecord(done[me], mine);
            for (int i = 0; i < n; i++)
                if (st[i] != mine) (void)hipStreamWaitEvent(st[i], done[me], 0);
        }
        mu.unlock();
    }
};
// spec_cap = 0: exact (one host round trip for the hit count).  spec_cap > 0: speculative — the buffer holds
// spec_cap hits, nothing is read back, *nhits is 0 and the count stays on the device (ctx.total_dev()).
// max_hits (exact mode only): a count beyond it returns MIMEO_SPLIT with *nhits = the count and nothing joined
int join_hits(JoinCtx &ctx, const IndexView &T, const IndexView &Q, int transitions, DeviceBuf &hits, uint64_t *nhits,
              JoinTiming *tm, Exclusive *ex = nullptr, uint64_t spec_cap = 0, uint64_t max_hits = ~0ull);
constexpr int MIMEO_SPLIT = 2;  // internal: a chunk of a chunked unit holds too many hits, the caller halves it
void join_timing_flush(JoinCtx &ctx);
constexpr int MIMEO_RETRY_EXACT = 1;  // internal: a speculative unit did not fit its buffers

// K4: seed hits -> HSPs (k4_extend.hip); out_hsps holds mimeo_hsp records on the device
struct ExtCounters;
struct ExtWork {  // per-lane work state of K4
    ExtCounters *ctr = nullptr;  // device
    DeviceBuf cand, fkey, fkey2, fprev, fprev2, longq, medq, flags, segs, tmp, nsel, bigseg;
    void release();
};
// `after_fast` (optional) is called once, right after the fast kernel of the first attempt has been
// launched: the point where a concurrent lane may start its own heavy phase.
// d_nhits != null: speculative — `nhits` is the capacity of `hits`, the real count is read by the kernels
// from *d_nhits and returned in *nhits_out; MIMEO_RETRY_EXACT if it exceeded the capacity.
// `chunk` (optional): the unit's hits arrive in several calls (a very large unit whose query is joined chunk by chunk,
// pipeline.hip).  Heads are decided from the sequences alone and are finished per call; followers and candidates
// accumulate in W and are resolved in the call flagged `last` — a follower chain may cross a chunk border.
struct ExtChunk {
    int first, last;
    uint64_t cand_cap;        // candidate capacity of the whole unit (no rerun in chunked mode)
    uint64_t nfollow_before;  // followers gathered by the earlier chunks (in)
    uint64_t nfollow_after;   // ... including this one (out)
};
int ungapped_hsps_device(ExtWork &W, const StrandView &T, const StrandView &Q, const uint2 *hits, uint64_t nhits,
                         const mimeo_params *p, DeviceBuf &out_hsps, uint64_t *nhsp, float *ms,
                         const std::function<void()> *after_fast = nullptr, const unsigned long long *d_nhits = nullptr,
                         uint64_t *nhits_out = nullptr, ExtChunk *chunk = nullptr);

// K5/K6: one group = one (target scaffold, query scaffold, strand) with its HSP range
constexpr int MAX_BATCH = 32;
struct Group {
    StrandView T, Q;
    uint32_t tid, qid, minus, nchain;
    uint64_t hsp_begin, hsp_end;
    uint32_t naln, overflow;
    // K6 round state: anchors are taken in order, up to nbatch per round
    uint32_t next, nacc, nbatch, job0;
    uint32_t batch[MAX_BATCH];
};
// K5 (k5_chain.hip): sort + chain + anchor order for every group
int chain_device(Group *d_groups, uint32_t ngroups, const mimeo_hsp *d_hsps, uint64_t nhsps, int do_chain,
                 mimeo_hsp *d_sorted, long long *d_best, long long *d_cand, int *d_pred, uint32_t *d_order);
// K6 (k6_gapped.hip): anchors -> gapped alignments; alignments of group g land at
// d_aln[hsp_begin .. hsp_begin + naln)
int gapped_device(Group *d_groups, uint32_t ngroups, const mimeo_hsp *d_sorted, const uint32_t *d_order,
                  uint64_t nhsps, const mimeo_params *p, mimeo_alignment *d_aln);
// chain + gapped extension of every group (pipeline.hip)
int chain_gapped_device(Group *d_groups, uint32_t ngroups, const mimeo_hsp *d_hsps, uint64_t nhsps,
                        const mimeo_params *p, DeviceBuf &scratch, mimeo_alignment *d_aln, float *ms_chain,
                        float *ms_gapped);

// K7: coverage collapse (k7_collapse.hip); host in, host out (sorted by chrom, start)
int coverage_collapse_device(const mimeo_interval *h_iv, uint64_t n, const uint32_t *h_chrom_len, uint32_t nchrom,
                             uint32_t min_cov, uint32_t min_len, std::vector<mimeo_interval> &out, float *ms);

// whole-job loop (pipeline.hip)
int align_pairs_impl(const mimeo_genome *A, const mimeo_genome *B, const uint32_t *pair_t, const uint32_t *pair_q,
                     uint64_t npairs, const mimeo_params *p, mimeo_alignment **out, uint64_t *nout);

void release_pipeline_buffers();  // pipeline.hip
int build_kept_indexes(mimeo_genome *g, const uint32_t *scaf, uint64_t n);  // pipeline.hip

// K8: tandem scorer (k8_tandem.hip); host in, host out
int tandem_masked_device(const mimeo_genome *A, const mimeo_interval *h_iv, uint64_t n, int match, int mismatch,
                         int minscore, int maxperiod, uint32_t *h_masked);

}  // namespace mimeo
