// ingest.hip — streaming FASTA ingest: replaces splitFasta + chromlens + the Biopython
// parse of the reference (src/mimeo/utils.py:274-309, 502-557).  A parser thread strips headers and line
// ends into pinned staging buffers, one record at a time; the calling thread uploads each finished record
// and runs K1 on it, so parsing record i+1 overlaps the copy and packing of record i.  Optionally the
// per-record `<id>.fa` files the reference's --adir/--bdir leave behind are written as a side effect.
#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

#include <condition_variable>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <set>
#include <thread>

#include "common.h"

namespace mimeo {

namespace {

struct Slot {
    uint8_t *buf = nullptr;  // pinned
    size_t cap = 0, len = 0;
    std::string name, header;
    bool full = false;
};

struct Ingest {
    Slot slot[2];
    std::mutex mu;
    std::condition_variable cv;
    bool done = false, abort = false;
    int rc = 0;
    std::string err;
    int dev = 0;
    std::string split_dir;

    int fail(int code, const std::string &msg) {
        std::lock_guard<std::mutex> lk(mu);
        if (!rc) { rc = code; err = msg; }
        return code;
    }
    int grow(Slot &s, size_t need) {
        if (need <= s.cap) return 0;
        size_t ncap = std::max<size_t>(need, std::max<size_t>(s.cap * 2, (size_t)16 << 20));
        uint8_t *nb = nullptr;
        if (hipHostMalloc((void **)&nb, ncap, hipHostMallocDefault) != hipSuccess)
            return fail(MIMEO_ERR_NOMEM, "pinned staging allocation failed");
        if (s.len) memcpy(nb, s.buf, s.len);
        if (s.buf) (void)hipHostFree(s.buf);
        s.buf = nb;
        s.cap = ncap;
        return 0;
    }
    // append one line's bytes: line ends are already cut off; blanks inside a line are dropped the way
    // Biopython's FASTA parser drops them
    int push(Slot &s, const uint8_t *p, size_t n) {
        while (n && (p[n - 1] == '\r' || p[n - 1] == ' ' || p[n - 1] == '\t')) n--;
        if (!n) return 0;
        int r = grow(s, s.len + n);
        if (r) return r;
        if (!memchr(p, ' ', n) && !memchr(p, '\t', n) && !memchr(p, '\r', n)) {
            memcpy(s.buf + s.len, p, n);
            s.len += n;
        } else {
            for (size_t i = 0; i < n; i++)
                if (p[i] != ' ' && p[i] != '\t' && p[i] != '\r') s.buf[s.len++] = p[i];
        }
        return 0;
    }
    int write_split(const Slot &s) {
        if (split_dir.empty()) return 0;
        std::string path = split_dir + "/" + s.name + ".fa";
        FILE *f = fopen(path.c_str(), "wb");
        if (!f) return fail(MIMEO_ERR_ARG, "cannot write " + path);
        fprintf(f, ">%s\n", s.header.c_str());
        for (size_t i = 0; i < s.len; i += 60) {  // SeqIO.write wraps at 60 columns
            fwrite(s.buf + i, 1, std::min<size_t>(60, s.len - i), f);
            fputc('\n', f);
        }
        fclose(f);
        return 0;
    }
};

// parser thread: all files, in order
void parse_files(Ingest *in, std::vector<std::string> paths) {
    (void)hipSetDevice(in->dev);
    int cur = 0;
    bool rec_open = false;
    std::vector<uint8_t> chunk((size_t)8 << 20);
    std::string carry;  // an unfinished line from the previous chunk
    auto begin = [&](const std::string &hdr) -> bool {
        std::unique_lock<std::mutex> lk(in->mu);
        in->cv.wait(lk, [&] { return !in->slot[cur].full || in->abort; });
        if (in->abort) return false;
        Slot &s = in->slot[cur];
        s.len = 0;
        s.header = hdr;
        size_t a = 0;
        while (a < hdr.size() && (hdr[a] == ' ' || hdr[a] == '\t')) a++;
        size_t b = a;
        while (b < hdr.size() && hdr[b] != ' ' && hdr[b] != '\t') b++;
        s.name = hdr.substr(a, b - a);
        rec_open = true;
        return true;
    };
    auto finish = [&]() -> bool {
        if (!rec_open) return true;
        if (in->write_split(in->slot[cur])) return false;
        {
            std::lock_guard<std::mutex> lk(in->mu);
            in->slot[cur].full = true;
        }
        in->cv.notify_all();
        cur ^= 1;
        rec_open = false;
        return true;
    };
    auto line = [&](const uint8_t *p, size_t n) -> bool {  // one complete line without its '\n'
        if (n && p[0] == '>') {
            if (!finish()) return false;
            size_t m = n - 1;
            while (m && (p[m] == '\r')) m--;
            return begin(std::string((const char *)p + 1, m));
        }
        if (!rec_open) return true;  // text before the first header is ignored
        return in->push(in->slot[cur], p, n) == 0;
    };
    bool ok = true;
    for (size_t fi = 0; fi < paths.size() && ok; fi++) {
        int fd = ::open(paths[fi].c_str(), O_RDONLY);
        if (fd < 0) { in->fail(MIMEO_ERR_ARG, "cannot open FASTA file " + paths[fi]); ok = false; break; }
        carry.clear();
        for (;;) {
            ssize_t got = ::read(fd, chunk.data(), chunk.size());
            if (got < 0) { in->fail(MIMEO_ERR_ARG, "read error on " + paths[fi]); ok = false; break; }
            if (got == 0) break;
            const uint8_t *p = chunk.data(), *end = p + got;
            while (p < end && ok) {
                const uint8_t *nl = (const uint8_t *)memchr(p, '\n', end - p);
                if (!nl) { carry.append((const char *)p, end - p); break; }
                if (!carry.empty()) {
                    carry.append((const char *)p, nl - p);
                    ok = line((const uint8_t *)carry.data(), carry.size());
                    carry.clear();
                } else {
                    ok = line(p, nl - p);
                }
                p = nl + 1;
            }
            if (!ok) break;
        }
        ::close(fd);
        if (ok && !carry.empty()) { ok = line((const uint8_t *)carry.data(), carry.size()); carry.clear(); }
        if (ok) ok = finish();  // a record never spans files
    }
    {
        std::lock_guard<std::mutex> lk(in->mu);
        in->done = true;
    }
    in->cv.notify_all();
}

}  // namespace

int load_fasta_impl(const char *const *paths, uint32_t npaths, const char *split_dir, mimeo_genome **out) {
    Ingest in;
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
    std::thread parser(parse_files, &in, pv);
    int rc = 0, idx = 0;
    for (;;) {
        {
            std::unique_lock<std::mutex> lk(in.mu);
            in.cv.wait(lk, [&] { return in.slot[idx].full || in.done; });
            if (!in.slot[idx].full) break;  // done and drained
        }
        Slot &s = in.slot[idx];
        if (!seen.insert(s.name).second) {  // utils.py:300-306
            rc = MIMEO_ERR_ARG;
            set_error("Non-unique name in genome: " + s.name + ". Quitting.");
        }
        if (!rc && s.len > d_cap) {
            if (d_ascii) (void)hipFree(d_ascii);
            d_cap = s.len + (s.len >> 2) + 4096;
            if (hipMalloc((void **)&d_ascii, d_cap) != hipSuccess) { rc = MIMEO_ERR_NOMEM; set_error("device allocation failed (FASTA staging)"); d_ascii = nullptr; d_cap = 0; }
        }
        if (!rc && s.len && hipMemcpyAsync(d_ascii, s.buf, s.len, hipMemcpyHostToDevice, stream()) != hipSuccess) {
            rc = MIMEO_ERR_HIP;
            set_error("hipMemcpyAsync(FASTA record) failed");
        }
        if (!rc) {
            g->scaf.emplace_back();
            g->names.push_back(s.name);
            rc = pack_scaffold(d_ascii, s.len, g->scaf.back());  // synchronises the stream: the slot is reusable
        }
        {
            std::lock_guard<std::mutex> lk(in.mu);
            s.full = false;
            if (rc) in.abort = true;
        }
        in.cv.notify_all();
        if (rc) break;
        idx ^= 1;
    }
    parser.join();
    if (d_ascii) (void)hipFree(d_ascii);
    for (auto &s : in.slot) if (s.buf) (void)hipHostFree(s.buf);
    if (!rc && in.rc) { rc = in.rc; set_error(in.err); }
    if (rc) {
        for (auto &s : g->scaf) free_scaffold(s);
        delete g;
        return rc;
    }
    *out = g;
    return 0;
}

}  // namespace mimeo
