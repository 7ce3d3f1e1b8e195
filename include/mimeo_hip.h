/*
 * mimeo_hip.h — C-ABI of libmimeo_hip.so, the MI355X (gfx950) engine behind the
 * `mimeo self | x | map` hot path.
 *
 * What this boundary replaces in the reference (Adamtaranto/mimeo):
 *   the reference has no library API for its hot path; it builds a shell script
 *   (src/mimeo/wrappers.py:899-1271 self_LZ_cmds, :683-896 xspecies_LZ_cmds,
 *   :525-680 map_LZ_cmds) and runs it with src/mimeo/utils.py:213-254 run_cmd.
 *   Every entry point below cites the command line(s) of that script whose work
 *   it performs.  The Python host (mimeo_amd/) binds these with ctypes; see
 *   INTEGRATION.md for the stub a reference maintainer would add.
 *
 * Conventions
 *   - plain pointers and sizes only; no C++/torch types cross the boundary;
 *   - every function returns 0 on success, <0 on error; the message is read with
 *     mimeo_last_error() (reference convention: non-zero exit of run_jobs.sh ->
 *     RuntimeError, utils.py:194-210);
 *   - buffers returned through `**out` are allocated by the library and released
 *     with mimeo_free(); inputs are borrowed for the duration of the call;
 *   - calls are blocking; a handle must not be used from two threads at once;
 *   - coordinates are 0-based half-open inside the ABI.  The origin-one `start1`
 *     / `start2+` columns of LASTZ's `general` format (wrappers.py:1031) are
 *     produced by the host writer, not here.
 *   - there is NO CPU fallback: every entry point that computes fails with
 *     MIMEO_ERR_NO_DEVICE when no gfx950 device is usable.
 */
#ifndef MIMEO_HIP_H
#define MIMEO_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MIMEO_ABI_VERSION 3

enum {
    MIMEO_OK = 0,
    MIMEO_ERR_ARG = -1,        /* bad argument (null pointer, id out of range, ...) */
    MIMEO_ERR_NO_DEVICE = -2,  /* no usable HIP device / library not initialised   */
    MIMEO_ERR_HIP = -3,        /* a HIP runtime call failed                         */
    MIMEO_ERR_NOMEM = -4,      /* host or device allocation failed                  */
    MIMEO_ERR_LIMIT = -5       /* input exceeds a documented limit                  */
};

/* Strand selector == lastz --strand=plus|minus|both (wrappers.py:1031 passes both). */
enum { MIMEO_STRAND_PLUS = 1, MIMEO_STRAND_MINUS = 2, MIMEO_STRAND_BOTH = 3 };

/*
 * Alignment parameters == the lastz flags the reference passes
 * (wrappers.py:1025-1037 / 786-798 / 607-653) plus the LASTZ defaults they imply.
 * mimeo_params_default() fills the values in brackets.
 */
typedef struct mimeo_params {
    int32_t hspthresh;    /* --hspthresh [3000]; also the gapped threshold (lastz default)  */
    int32_t xdrop;        /* ungapped x-drop [910 = 10*sub[A][A]]; >= 500 (MIMEO_ERR_ARG below)  */
    int32_t ydrop;        /* gapped y-drop [9400 = open + 300*extend]                      */
    int32_t gap_open;     /* [400]                                                        */
    int32_t gap_extend;   /* [30]                                                         */
    int32_t transitions;  /* seed tolerates one transition [1] (lastz default)            */
    int32_t entropy;      /* --entropy [1]                                                */
    int32_t chain;        /* --chain [1]                                                  */
    int32_t gapped;       /* --gapped [1]                                                 */
    int32_t strand;       /* --strand [MIMEO_STRAND_BOTH]                                 */
    int32_t reserved[6];
} mimeo_params;

/* One raw seed hit: 0-based starts of the 19-base seed window (12of19, lastz default seed). */
typedef struct mimeo_seed_hit {
    uint32_t tpos;
    uint32_t qpos; /* on the strand being scanned (reverse-complement coordinates for '-') */
} mimeo_seed_hit;

/* One gap-free HSP (lastz --gfextend output, after --entropy and --hspthresh). */
typedef struct mimeo_hsp {
    uint32_t tstart;
    uint32_t qstart; /* strand coordinates, as above */
    uint32_t length;
    uint32_t flags;  /* bit0: survives --chain (set by mimeo_chain_hsps / align) */
    int64_t score;   /* entropy-adjusted score (== raw when entropy is off) */
    int64_t raw_score;
} mimeo_hsp;

/*
 * One gapped alignment == one row of lastz --format=general:name1,strand1,start1,
 * end1,length1,name2,strand2,start2+,end2+,length2,score,identity (wrappers.py:1031).
 * qstart/qend are on the query PLUS strand (like start2+/end2+), 0-based half-open.
 */
typedef struct mimeo_alignment {
    uint32_t tid;    /* index of the target scaffold in genome A */
    uint32_t qid;    /* index of the query scaffold in genome B (or A for self) */
    uint32_t tstart, tend;
    uint32_t qstart, qend;
    int64_t score;
    uint32_t id_n;   /* identity numerator: matched columns                         */
    uint32_t id_d;   /* identity denominator: matched + mismatched columns (no gaps) */
    uint32_t qstrand; /* 0 '+', 1 '-' */
    uint32_t reserved;
} mimeo_alignment;

/* Half-open interval on a chromosome, BED-interpreted (wrappers.py:1120). */
typedef struct mimeo_interval {
    uint32_t chrom;
    uint32_t start;
    uint32_t end;
} mimeo_interval;

/* Per-call statistics of the last mimeo_align_pairs / stage call (metrics). */
typedef struct mimeo_stats {
    uint64_t pair_strands;        /* (target, query, strand) units processed              */
    uint64_t seed_hits;           /* raw seed hits produced by the seed-scan kernel        */
    uint64_t hsps;                /* HSPs >= hspthresh                                    */
    uint64_t chained_hsps;
    uint64_t alignments;
    uint64_t query_bases_scanned; /* sum over pair-strands of Lq                           */
    uint64_t scan_bytes_algorithmic; /* SURVEY §8(d) B_scan summed over pair-strands       */
    uint64_t scan_bytes_kernel;   /* compulsory bytes of the seed-scan kernel   */
    double ms_index;              /* HIP-event time in the index-build kernels              */
    double ms_scan;               /* HIP-event time of the heavy phase: the fused seed-scan / pre-filter / walk kernels (K34) */
    double ms_extend;             /* the tails of the gap-free stage, once per batch: long walks, follower sort + resolution, entropy */
    double ms_chain;
    double ms_gapped;
    double ms_collapse;
    double ms_total;              /* host wall time of the call                            */
    uint64_t scan_launches;       /* units (target, query, strand) the seed-scan kernel worked off in ms_scan_fill (one launch per unit until ABI 2) */
    double ms_scan_fill;          /* HIP-event time of the seed-scan kernels (K34, the roofline kernel), back to back per batch */
    uint64_t index_blocks;        /* blocks the pair matrix was cut into so that the seed indexes fit in memory (1 = none) */
    uint64_t batches;             /* batches of units the call was worked off in (tails, chain and gapped stage run once per batch) */
    uint64_t queue_reruns;        /* batches repeated because a queue sized for random sequence overflowed (repeat-rich units) */
    uint64_t walked_hits;         /* seed hits the pre-filter could not dismiss: walked exactly */
    uint64_t followers;           /* hits with an earlier seed hit of their diagonal in reach: resolved after one sort per batch */
    uint64_t super_units;         /* units of super-scaffolds (small scaffolds packed behind spacers: fragmented assemblies); 0 = the call ran one unit per scaffold pair and strand */
    uint64_t scan_kernel_launches; /* launches of the seed-scan kernel K34 (first pass) in ms_scan_fill: one per BATCH of units (ABI 3) */
} mimeo_stats;

typedef struct mimeo_genome mimeo_genome; /* opaque: device-resident packed scaffolds */

/* ---- lifecycle ------------------------------------------------------------------ */

/* ABI version of the loaded library (compare with MIMEO_ABI_VERSION). Never fails. */
int mimeo_abi_version(void);

/* Select HIP device `device` for this process and create the library's streams.
 * One process drives one GPU (multi-GPU = one process per GPU). */
int mimeo_init(int device);
void mimeo_shutdown(void);

/* Message of the last failing call on this thread ("" if none). */
const char *mimeo_last_error(void);

/* Fill *p with the LASTZ defaults listed at mimeo_params. */
int mimeo_params_default(mimeo_params *p);

int mimeo_get_stats(mimeo_stats *out);

void mimeo_free(void *p);

/* ---- genome ingest (replaces utils.py:274-309 splitFasta + lastz's own file read) --- */

/*
 * Upload `nscaf` scaffolds given as concatenated ASCII bases (`bases`, any case,
 * anything outside ACGTacgt is treated as N) with `offsets[nscaf+1]` delimiting
 * them, and pack them on the device into 2-bit bit-planes + N / soft-mask planes
 * for both strands (kernel K1).  Lower-case target bases are excluded from seeding
 * and N from seeding on both sides, as lastz does by default.
 */
int mimeo_genome_create(uint32_t nscaf, const uint8_t *bases, const uint64_t *offsets,
                        mimeo_genome **out);
void mimeo_genome_destroy(mimeo_genome *g);
int mimeo_genome_nscaf(const mimeo_genome *g, uint32_t *nscaf);
int mimeo_genome_length(const mimeo_genome *g, uint32_t scaf, uint64_t *length);

/*
 * Streaming FASTA ingest: replaces splitFasta + chromlens + the Biopython parse
 * (src/mimeo/utils.py:274-309, 502-557; call sites run_self.py:199-224).  Every
 * record of every file in `paths` (in order) becomes one scaffold; the record id is
 * the first word of the header line, as Biopython's rec.id.  A host thread parses
 * record i+1 into pinned memory while record i is copied and packed (K1).  A
 * duplicate id is MIMEO_ERR_ARG with the reference's message (utils.py:300-306).
 * If `split_dir` is non-NULL one `<id>.fa` per record is also written there — the
 * side effect of the reference's --adir/--bdir.  `mimeo_genome_name` returns a
 * pointer owned by the genome ("" for genomes made by mimeo_genome_create).
 */
int mimeo_genome_load_fasta(const char *const *paths, uint32_t npaths, const char *split_dir,
                            mimeo_genome **out);
int mimeo_genome_name(const mimeo_genome *g, uint32_t scaf, const char **name);

/*
 * Seed-index residency.  lastz rebuilds the target's seed table in every one of the S^2 runs
 * of run_jobs.sh (call site wrappers.py:1028-1031); mimeo_align_pairs builds the index of a
 * scaffold strand once per call and, by default, gives it back when the call returns.  With
 * keep != 0 the indexes that later calls build for scaffolds of `g` stay attached to the handle
 * (64 MiB + 52 bytes per base and strand — offsets, positions and the seed frames the fused seed-scan
 * kernel streams: a 1 Gbp genome, both strands, is 117 GB of the 288 GB)
 * and are reused by every later call, so a job may be issued as several calls (one per target
 * scaffold, say) without paying for the tables again.  mimeo_genome_drop_indexes releases the
 * indexes (both strands) of the listed scaffolds, or all of them when n == 0;
 * mimeo_genome_destroy releases them too.  Results never depend on this setting.
 */
int mimeo_genome_keep_indexes(mimeo_genome *g, int keep);
/* Build (and keep: implies keep != 0) the indexes of the listed scaffolds now — both strands, and the
 * soft-mask-aware target variant where the scaffold has lower-case bases; n == 0: every scaffold. */
int mimeo_genome_build_indexes(mimeo_genome *g, const uint32_t *scaf, uint64_t n);
int mimeo_genome_drop_indexes(mimeo_genome *g, const uint32_t *scaf, uint64_t n);

/* ---- stage entry points (parity tests, profiling) ----------------------------- */

/*
 * lastz seed stage (SURVEY §8a A6+A7): build the 12of19 seed index of target
 * scaffold `tid` of T and of query scaffold `qid` of Q on strand `qstrand`
 * (0 '+', 1 '-'), and return every seed hit.  Order of the returned hits is
 * unspecified; the multiset is exact.
 */
int mimeo_seed_hits(const mimeo_genome *T, uint32_t tid, const mimeo_genome *Q, uint32_t qid,
                    uint32_t qstrand, const mimeo_params *p, mimeo_seed_hit **out, uint64_t *nout);

/*
 * lastz --gfextend --entropy --hspthresh (A8): seed hits -> gap-free HSPs with the
 * per-diagonal "already extended" suppression.  Returned sorted by
 * (tstart-qstart, tstart, length).
 */
int mimeo_ungapped_hsps(const mimeo_genome *T, uint32_t tid, const mimeo_genome *Q, uint32_t qid,
                        uint32_t qstrand, const mimeo_params *p, mimeo_hsp **out, uint64_t *nout);

/*
 * lastz --chain alone (A9; wrappers.py:1031), a stage inside lastz exposed for parity tests like the two above:
 * the n HSPs of ONE (target, query, strand) are copied to out[0 .. n) in (tstart, qstart, length) order with bit 0
 * of `flags` set on the members of the best chain (ties: earliest predecessor, earliest end) and cleared on the
 * others.  Runs K5 as mimeo_align_pairs does (the kernel is chosen by the number of HSPs).  out: n records, the caller's.
 */
int mimeo_chain_hsps(const mimeo_hsp *in, uint64_t n, mimeo_hsp *out);

/*
 * One full `lastz t.fa q.fa ...` invocation (A6-A10; wrappers.py:1025-1037): all
 * requested strands of one (target, query) pair -> gapped alignments.
 */
int mimeo_align_pair(const mimeo_genome *T, uint32_t tid, const mimeo_genome *Q, uint32_t qid,
                     const mimeo_params *p, mimeo_alignment **out, uint64_t *nout);

/* ---- whole-job entry points ---------------------------------------------------- */

/*
 * The per-pair loop of run_jobs.sh (wrappers.py:1015-1059): for k in [0,npairs)
 * align target scaffold pair_t[k] of A against query scaffold pair_q[k] of B
 * (B == NULL: of A, i.e. `mimeo self`).  Results are concatenated in pair order;
 * the host applies the awk/sort filter (A11) when it writes the TAB.
 *
 * How the pairs are worked off is the library's business and changes no record: units
 * (target, query, strand) in batches; when the list is a full cross product with at least eight
 * scaffolds of at most 6 Mbp on one side (and no kept indexes), they are concatenated behind spacers
 * of N into super-scaffolds for the seed index and the gap-free stage, and every HSP is handed
 * back to its scaffold pair before chaining and gapped extension (mimeo_stats.super_units > 0;
 * DESIGN.md "Fragmented assemblies"); in a self job (B == NULL) the plus-strand units of (t, q) and
 * (q, t) share one seed scan and gap-free stage when neither scaffold has soft-masked bases (DESIGN.md
 * "Shared plus strand").  Limits: scaffolds below 2^31 - 256 bases.  A pair whose gapped extension
 * needs a DP band beyond 65 536 columns is LEFT OUT — no row of it is returned,
 * the other pairs are, the call returns 0 and mimeo_get_failed_pairs names the pair: the reference's
 * run_jobs.sh has no `set -e`, a failing lastz run costs its own pair's rows only (utils.py:125-128,
 * :194-210 look at the last command's status).
 */
int mimeo_align_pairs(const mimeo_genome *A, const mimeo_genome *B, const uint32_t *pair_t,
                      const uint32_t *pair_q, uint64_t npairs, const mimeo_params *p,
                      mimeo_alignment **out, uint64_t *nout);

/*
 * The same loop with the strands chosen per pair: pair k is aligned on the strands pair_strand[k]
 * (MIMEO_STRAND_PLUS | MIMEO_STRAND_MINUS, masked with p->strand; pair_strand == NULL: p->strand for every
 * pair, i.e. mimeo_align_pairs).  This is what a rank's share of a sharded self job looks like
 * (mimeo_amd/dist.py deal_units): all minus-strand units of its target rows, and the plus-strand units
 * of the unordered scaffold pairs dealt to it in BOTH orders, so that the shared plus strand applies.
 * Results are concatenated in pair order, a pair's plus-strand rows first.
 */
int mimeo_align_units(const mimeo_genome *A, const mimeo_genome *B, const uint32_t *pair_t,
                      const uint32_t *pair_q, const uint8_t *pair_strand, uint64_t npairs,
                      const mimeo_params *p, mimeo_alignment **out, uint64_t *nout);

/*
 * Pairs of the last mimeo_align_pairs / mimeo_align_units call that hit a documented limit and were left
 * out: *n of them; the first min(*n, cap) are written to pair_index[] (index into the call's pair list)
 * and code[] (MIMEO_ERR_LIMIT).  Either array may be NULL.  mimeo_last_error() describes the last one.
 */
int mimeo_get_failed_pairs(uint64_t *pair_index, int32_t *code, uint64_t cap, uint64_t *n);

/*
 * bedtools genomecov -bg | awk $4>=min_cov | sort | bedtools merge | awk len>=min_len
 * (A13+A14; wrappers.py:1131-1177) as one integer kernel pipeline (K7).  Intervals
 * with start >= end are ignored, ends are clipped to chrom_len[chrom].  Output is
 * sorted by (chrom, start).
 */
int mimeo_coverage_collapse(const mimeo_interval *iv, uint64_t n, const uint32_t *chrom_len,
                            uint32_t nchrom, uint32_t min_cov, uint32_t min_len,
                            mimeo_interval **out, uint64_t *nout);

/*
 * bedtools genomecov -bg alone (wrappers.py:1131-1145): the maximal runs of equal depth > 0 of the intervals, in
 * (chrom, start) order — what `bedtools genomecov -bg -i sorted.bed -g lens` prints, as records.  Serves
 * scripts/bedtools, the drop-in for the reference's --bedtools option; the library's own workflows
 * use mimeo_coverage_collapse, which never materialises the runs.
 */
typedef struct mimeo_depth_run {
    uint32_t chrom, start, end, depth;
} mimeo_depth_run;
int mimeo_coverage_bedgraph(const mimeo_interval *iv, uint64_t n, const uint32_t *chrom_len, uint32_t nchrom,
                            mimeo_depth_run **out, uint64_t *nout);

/*
 * Tandem-repeat content of genome slices: the role of `trf F 2 7 7 80 10 50 50 -m -h -ngs` in
 * wrappers.py:120-262 trfFilter (flags run_map.py:145-178).  masked[k] = number of bases of
 * iv[k] = [start, end) on scaffold `chrom` of A that the tandem scorer (K8, DESIGN.md "Tandem
 * scorer v2") marks; the host keeps a hit if 100*masked/len < maxtandem (wrappers.py:237-240).
 * match / mismatch / delta / minscore / maxperiod are TRF's weights and thresholds of the same
 * names (delta = indel penalty; delta <= 0: gap-free comparison); TRF's detection statistics PM and
 * PI have no counterpart.  `masked` is a caller-owned array of n entries.  maxperiod <= 64.
 */
int mimeo_tandem_masked(const mimeo_genome *A, const mimeo_interval *iv, uint64_t n, int32_t match,
                        int32_t mismatch, int32_t delta, int32_t minscore, int32_t maxperiod, uint32_t *masked);

#ifdef __cplusplus
}
#endif
#endif /* MIMEO_HIP_H */
