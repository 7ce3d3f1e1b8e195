// ingest.hip — streaming FASTA ingest: replaces splitFasta + chromlens + the Biopython
// parse of the reference (src/mimeo/utils.py:274-309, 502-557).  A parser thread strips headers and line
// ends into pinned staging buffers, one record at a time; the calling thread uploads each finished record
// and runs K1 on it, so parsing record i+1 overlaps the copy and packing of record i.  Optionally the
// per-record `<id>.fa` files the reference's --adir/--bdir leave behind are written as a side effect.
// The threaded host logic lives in ingest_host.h (HIP-free: it runs under the CPU sanitizers in
// tests/test_host_sanitize.py); this file supplies pinned memory and the device-side consumer.
#include <set>

#include "common.h"
#include "ingest_host.h"

namespace mimeo {

static_assert(ingest_host::ERR_ARG == MIMEO_ERR_ARG && ingest_host::ERR_NOMEM == MIMEO_ERR_NOMEM, "error codes of ingest_host.h");

namespace {
struct PinnedMem {
    static void *alloc(size_t n) {
        void *p = nullptr;
        return hipHostMalloc(&p, n, hipHostMallocDefault) == hipSuccess ? p : nullptr;
    }
    static void release(void *p) { (void)hipHostFree(p); }
    static void thread_init(int dev) { (void)hipSetDevice(dev); }
};
}  // namespace

int load_fasta_impl(const char *const *paths, uint32_t npaths, const char *split_dir, mimeo_genome **out) {
    ingest_host::Ingest<PinnedMem> in;
    in.dev = device_id();
    if (split_dir && *split_dir) in.split_dir = split_dir;
    std::vector<std::string> pv;
    for (uint32_t i = 0; i < npaths; i++) {
        if (!paths[i]) { set_error("null path"); return MIMEO_ERR_ARG; }
        pv.push_back(paths[i]);
    }
    mimeo_genome *g = new mimeo_genome();
    std::set<std::string> seen;
    uint8_t *d_ascii = nullptr;
    size_t d_cap = 0;
    int rc = in.run(pv, [&](ingest_host::Slot &s) -> int {
        if (!seen.insert(s.name).second) {  // utils.py:300-306
            set_error("Non-unique name in genome: " + s.name + ". Quitting.");
            return MIMEO_ERR_ARG;
        }
        if (s.len > d_cap) {
            if (d_ascii) (void)hipFree(d_ascii);
            d_cap = s.len + (s.len >> 2) + 4096;
            if (hipMalloc((void **)&d_ascii, d_cap) != hipSuccess) { set_error("device allocation failed (FASTA staging)"); d_ascii = nullptr; d_cap = 0; return MIMEO_ERR_NOMEM; }
        }
        if (s.len && hipMemcpyAsync(d_ascii, s.buf, s.len, hipMemcpyHostToDevice, stream()) != hipSuccess) {
            set_error("hipMemcpyAsync(FASTA record) failed");
            return MIMEO_ERR_HIP;
        }
        g->scaf.emplace_back();
        g->names.push_back(s.name);
        return pack_scaffold(d_ascii, s.len, g->scaf.back());  // synchronises the stream: the slot is reusable
    });
    if (d_ascii) (void)hipFree(d_ascii);
    if (rc && in.rc == rc && !in.err.empty()) set_error(in.err);   // the parser's own failure
    if (rc) {
        for (auto &s : g->scaf) free_scaffold(s);
        delete g;
        return rc;
    }
    *out = g;
    return 0;
}

}  // namespace mimeo
