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
#include "host_plan.h"

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

// Seed frames (K2, read by the fused seed-scan / extension kernel K34): every index entry carries the 192 bases
// [p - FRAME_LEFT, p - FRAME_LEFT + 192) around its seed start p as two bit planes brought into ONE alignment (bit 0 =
// base p - FRAME_LEFT), 12 words = three uint4 {lo0..lo3} {lo4,lo5,hi0,hi1} {hi2..hi5}.  Target and query frames of
// a seed hit then XOR word by word with no shifting, and every shift in the pre-filter is a compile-time constant.
// The seed end p + 19 sits at frame bit 128: the left walk (through the seed) owns bits 127 .. 32 (96 steps, six
// half-word blocks), the right walk bits 128 .. 191 (64 steps), and bits 13 .. 31 only serve the seed windows that
// END where the last left steps arrive.  Bit 31 of pos[] is set for an entry whose frame holds a non-ACGT base (the
// frame shows it as A): such a hit always goes to the exact walk.
constexpr int FRAME_LEFT = 109;            // frame bit 0 = base p - 109
constexpr int FRAME_WORDS = 6;             // per plane
constexpr uint32_t POS_MASK = 0x7FFFFFFFu;
constexpr uint32_t POS_NFLAG = 0x80000000u;

struct IndexView {
    const uint32_t *off;  // NBUCKET + 1
    const uint32_t *pos;  // bit 31: the entry's frame holds an N (POS_NFLAG)
    const uint4 *fr;      // 3 parts per entry, part k of entry i at fr[k * fr_stride + i]
    uint32_t fr_stride;
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
    uint4 *fr = nullptr;
    uint32_t n = 0, fr_stride = 0;
    size_t pos_bytes = 0, fr_bytes = 0;  // allocation sizes (free-list keys)
    IndexView view() const { return IndexView{off, pos, fr, fr_stride, n}; }
    void release();
};
// device bytes of the seed index of one strand of `len` bases: offsets, positions, frames
inline uint64_t seed_index_bytes(uint64_t len) { return ((uint64_t)NBUCKET + 2) * 4 + len * (4 + 48); }

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
    bool view = false;     // a slice of an arena (another DeviceBuf owns the memory): never freed or grown through this handle
    int reserve(size_t bytes);
    void release();
    void set_view(void *ptr, size_t bytes) { p = ptr; cap = bytes; view = true; }
};
struct JoinTiming { float ms_count = 0, ms_fill = 0; };
// per-lane work state of K3 (a lane = one host thread + one stream working through units)
struct JoinCtx {
    unsigned long long *tile_count = nullptr, *tile_base = nullptr;
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
    void release();
};
// exact mode: one host round trip for the hit count, then the fill (stage entry point, A/B heavy path)
int join_hits(JoinCtx &ctx, const IndexView &T, const IndexView &Q, int transitions, DeviceBuf &hits, uint64_t *nhits,
              JoinTiming *tm);

// K4: seed hits -> HSPs for a BATCH of units (k4_extend.hip, k34_fused.hip).  One unit = one (target scaffold,
// query scaffold, strand); the device table of UnitDesc is what every K4 kernel looks its two strands up in.
struct UnitDesc {
    StrandView T, Q;
    uint32_t same;  // target and query are the same strand of the same scaffold: diagonal 0 belongs to k4_diag0
    uint32_t pad[3];
};
struct UnitWork {  // host side: a unit with the seed indexes its heavy kernel reads
    UnitDesc d;
    IndexView ti, qi;
};
// a unit of the batch as the heavy kernels see it (K34 over all units in one launch; the walk-queue kernel likewise)
struct FusedUnit {
    IndexView T, Q;
    StrandView Tv, Qv;
    uint32_t unit;        // number in the batch (index into the UnitDesc table, tag of every record)
    uint32_t same;        // target and query are the same strand: the main diagonal belongs to k4_diag0
    uint64_t walk_base;   // this unit's region of the walk queue: eight shards of walk_cap entries from here
    uint64_t walk_cap;
};

struct ExtStats {
    uint64_t seed_hits = 0, walked = 0, walk_queue = 0, followers = 0, candidates = 0, reruns = 0;
    uint64_t scan_bytes_algorithmic = 0, scan_bytes_kernel = 0, heavy_launches = 0, heavy_kernel_launches = 0;
    float ms_heavy = 0, ms_k34 = 0, ms_walk = 0, ms_tails = 0;  // ms_k34: the K34 launches alone (event pair per launch)  // ms_walk: the exact walks of the walk queue (part of ms_tails)
};
// The extension stage of one batch: heavy kernel per unit (K34 fused seed scan + pre-filter + exact walks; or, for
// A/B checks, the stand-alone K3 join + K4 fast kernel of round 1) appending to batch-wide queues, then the tails ONCE
// per batch: walks beyond the frame, k4_diag0 of the self units, ONE radix sort of the followers of all units,
// segment resolution, entropy.  Two host synchronisations per batch.  Leaves nhsp HSPs in hsps / hsp_unit (device).
struct ExtBatch {
    DeviceBuf units, ctr, cand, fkey, fkey2, fprev, fprev2, medq, medu, longq, longu, walkq, flags, segs, tmp, nsel, bigseg, hsps,
        hsp_unit, unit_hits, tile_hits, selfs, hits, bigcand, bigacc, heavy;
    JoinCtx jc;
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
    std::vector<hipEvent_t> kev;     // an event pair per unit around its K34 launch
    hipStream_t side = nullptr;      // k4_diag0 of the self units runs beside the heavy kernels
    hipEvent_t side_done = nullptr;
    std::vector<unsigned long long> h_unit_hits;
    // queue sizing: largest excess over the random-sequence shares seen so far, per queue (followers, walks beyond the frame,
    // long walks, candidates, walk queue): microsatellites flood the follower queue alone (C5: half of all hits)
    double boost_f = 1.0, boost_m = 1.0, boost_l = 1.0, boost_c = 1.0, boost_w = 1.0;
    // run = start (the heavy phase of the batch is enqueued on the calling thread's stream, nothing is waited for) + finish
    // (tails on the calling thread's stream at that time, which may be another one: two host round trips; a batch whose queues
    // overflowed is repeated there with room).  The pipeline starts the next batch before it finishes this one.
    // mirror_dst (optional, one entry per unit): unit u's HSPs are also emitted transposed (target <-> query) for unit
    // mirror_dst[u] (NO_MIRROR: none).  The plus-strand units (A, B) and (B, A) of a self job have transposed HSP sets —
    // seeds, x-drop walks, the per-diagonal rule, entropy and HOXD70 are all symmetric — so one heavy phase serves both
    // (DESIGN.md "Shared plus strand"); the mirror unit itself comes with empty indexes and launches nothing.
    // MIMEO_ERR_SPLIT (internal): the queues of this batch do not fit the free device memory — the caller cuts it in two.
    int run(const std::vector<UnitWork> &work, const mimeo_params *p, uint64_t *nhsp, ExtStats *st, const std::vector<uint32_t> *mirror_dst = nullptr);
    int start(const std::vector<UnitWork> &work, const mimeo_params *p, const std::vector<uint32_t> *mirror_dst = nullptr);
    int finish(uint64_t *nhsp, ExtStats *st);
    void release();
    // state of the batch between start and finish
    std::vector<UnitWork> w_;
    mimeo_params p_;
    std::vector<uint32_t> h_selfs_, mirror_dst_;
    std::vector<UnitDesc> h_units_;
    std::vector<FusedUnit> h_funits_;
    DeviceBuf mirror;     // mirror_dst_ on the device + one counter
    // hipMalloc / hipFree of tens of GB cost seconds (C5 at full size spent 80 % of its wall time there): the queues of a batch are
    // slices of ONE allocation that only ever grows (arena_q, carved per batch), the follower sort's buffers of another (arena_s)
    DeviceBuf arena_q, arena_s;
    DeviceBuf plan_buf;          // the split pass's plan (k34_plan)
    DeviceBuf funits, nwalk_u;   // per-unit table of the heavy kernels (FusedUnit); walk-queue counters per unit and shard + split-pass tile counts
    std::vector<uint64_t> walk_cap_u_;   // walk-queue capacity per unit and shard
    uint64_t walk_entries_ = 0;          // ... all regions together
    uint32_t nactive_ = 0;               // units of the batch that launch the heavy kernels
    const unsigned int *plan_ctr_ = nullptr;   // the split pass plan's counters on the device (statistics)
    uint64_t cap_f_ = 0, cap_m_ = 0, cap_l_ = 0, cap_c_ = 0;
    double expect_hits_ = 0, shrink_ = 1.0;
    uint32_t ebits_ = 0, dbits_ = 0, key_bits_ = 0;
    bool v1_ = false, started_ = false, k4_stats_ = false, splittable_ = false;
    uint32_t k34_dbg_ = 0, qw_blocks_ = 256;
    int k4_variant_ = 0;
    void *q_ = nullptr;   // ExtQueues of the batch (k4_device.h), owned
    int enqueue_heavy();
    void release_queues();          // the big queue buffers back to the device pool
    uint64_t arena_bytes() const;   // device bytes of the batch's queues at the current capacities
    uint64_t queue_bytes() const;   // device bytes the queues of the batch take at the current capacities
    uint64_t held_bytes() const;    // ... and what its buffers hold already
};
constexpr uint32_t NO_MIRROR = 0xFFFFFFFFu;
constexpr int MIMEO_ERR_SPLIT = -100;   // internal: never crosses the C-ABI
// largest batch the follower key can name for scaffolds of these lengths (unit bits = 64 - end bits - diagonal bits)
uint32_t ext_batch_max_units(uint64_t max_tlen, uint64_t max_qlen);

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
int chain_device(Group *d_groups, uint32_t ngroups, const mimeo_hsp *d_hsps, const uint32_t *d_hsp_unit, uint64_t nhsps,
                 int do_chain, mimeo_hsp *d_sorted, long long *d_best, long long *d_cand, int *d_pred, uint32_t *d_order);
// K6 (k6_gapped.hip): anchors -> gapped alignments; alignments of group g land at
// d_aln[hsp_begin .. hsp_begin + naln)
int gapped_device(Group *d_groups, uint32_t ngroups, const mimeo_hsp *d_sorted, const uint32_t *d_order,
                  uint64_t nhsps, const mimeo_params *p, mimeo_alignment *d_aln);
// alignments of all groups packed densely (group g: d_dense[job0 .. job0 + naln)); same stream as gapped_device
void dense_alignments_device(Group *d_groups, uint32_t ngroups, const mimeo_alignment *d_aln, mimeo_alignment *d_dense);
// ---- super-scaffolds (pack.hip): small scaffolds concatenated behind spacers of N for K2 / K34 / K4 -------------------
typedef host_plan::Member PackMember;   // scaffold number, first base inside the super-scaffold, bases (host_plan.h)
struct SuperSide {
    std::vector<Scaffold> supers;                   // a super of ONE member at offset 0 is the scaffold itself (not owned)
    std::vector<bool> owned;
    std::vector<std::vector<PackMember>> members;
    std::vector<uint32_t> super_of, start_of;       // per scaffold number of the genome (0xFFFFFFFF: takes no part)
    void release();
};
int build_super_side(const mimeo_genome *g, const std::vector<uint32_t> &ids, uint32_t spacer, uint64_t member_max,
                     uint64_t super_len, SuperSide &out);
struct RegroupTables {   // device pointers
    const uint3 *unit_tab;                                // per unit of the batch: target super, query super, minus
    const uint32_t *t_off, *t_start, *t_len, *t_rank;     // members of target super s: slots t_off[s] .. t_off[s + 1]; rank in the target set
    const uint32_t *q_off, *q_start, *q_len, *q_rank;
    const uint32_t *pairidx;                              // [t_rank * nq + q_rank] -> index of the pair in pair_t / pair_q
    uint32_t nq, pad;
    const uint32_t *pair_t, *pair_q;                      // scaffold numbers of the pairs
    const StrandView *t_view, *q_view_fwd, *q_view_rc;    // per scaffold number
};
int regroup_hsps_device(mimeo_hsp *d_hsps, uint32_t *d_hunit, uint64_t nh, const RegroupTables &R, uint32_t npairs, DeviceBuf &groups,
                        uint32_t *ngroups);
int group_summary_device(const Group *d_groups, uint32_t ngroups, uint64_t out[3]);
int overflowed_groups_device(const Group *d_groups, uint32_t ngroups, uint64_t expect, std::vector<uint2> *out);
void release_pack_buffers();
// chain + gapped extension of every group (pipeline.hip)
int chain_gapped_device(Group *d_groups, uint32_t ngroups, const mimeo_hsp *d_hsps, const uint32_t *d_hsp_unit, uint64_t nhsps,
                        const mimeo_params *p, DeviceBuf &scratch, mimeo_alignment *d_aln, float *ms_chain,
                        float *ms_gapped);

// K7: depth runs (bedtools genomecov -bg); host in, host out (in chrom, start order)
int coverage_bedgraph_device(const mimeo_interval *h_iv, uint64_t n, const uint32_t *h_chrom_len, uint32_t nchrom,
                             std::vector<mimeo_depth_run> &out);
// K7: coverage collapse (k7_collapse.hip); host in, host out (sorted by chrom, start)
int coverage_collapse_device(const mimeo_interval *h_iv, uint64_t n, const uint32_t *h_chrom_len, uint32_t nchrom,
                             uint32_t min_cov, uint32_t min_len, std::vector<mimeo_interval> &out, float *ms);

// whole-job loop (pipeline.hip): pair k on the strands pair_strand[k] (null: p->strand for every pair)
int align_units_impl(const mimeo_genome *A, const mimeo_genome *B, const uint32_t *pair_t, const uint32_t *pair_q,
                     const uint8_t *pair_strand, uint64_t npairs, const mimeo_params *p, mimeo_alignment **out, uint64_t *nout);
// pairs of the last align call that hit a documented limit (their rows are left out, the call goes on): (pair index, code)
const std::vector<std::pair<uint64_t, int>> &failed_pairs();

void release_pipeline_buffers();  // pipeline.hip
int ungapped_units(const std::vector<UnitWork> &work, const mimeo_params *p, std::vector<std::vector<mimeo_hsp>> *per_unit,
                   ExtStats *st);  // pipeline.hip: the extension stage of a batch, HSPs per unit on the host
int build_kept_indexes(mimeo_genome *g, const uint32_t *scaf, uint64_t n);  // pipeline.hip

// K8: tandem scorer (k8_tandem.hip); host in, host out
int tandem_masked_device(const mimeo_genome *A, const mimeo_interval *h_iv, uint64_t n, int match, int mismatch, int delta,
                         int minscore, int maxperiod, uint32_t *h_masked);

}  // namespace mimeo
