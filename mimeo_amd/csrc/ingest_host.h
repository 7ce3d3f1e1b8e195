// ingest_host.h — the host side of the streaming FASTA ingest (ingest.hip), free of HIP: the parser thread, the two
// staging slots and their hand-over protocol.  ingest.hip instantiates it with pinned host memory and a consumer that
// uploads and packs each record; tests/sanitize/host_sanitize.cc instantiates it with malloc and a consumer that
// keeps the bytes, and runs it under AddressSanitizer / UBSan / ThreadSanitizer on the CPU (GPU sanitizers are not
// available on the pool).  Replaces splitFasta + chromlens + the Biopython parse of the reference
// (src/mimeo/utils.py:274-309, 502-557).
#pragma once
#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace mimeo {
namespace ingest_host {

constexpr int ERR_ARG = -1, ERR_NOMEM = -4;   // == MIMEO_ERR_ARG / MIMEO_ERR_NOMEM (include/mimeo_hip.h; checked in ingest.hip)

struct Slot {
    uint8_t *buf = nullptr;  // staging memory of the policy
    size_t cap = 0, len = 0;
    std::string name, header;
    bool full = false;
};

// Mem: static void *alloc(size_t); static void release(void *); static void thread_init(int dev);
template <class Mem>
struct Ingest {
    Slot slot[2];
    std::mutex mu;
    std::condition_variable cv;
    bool done = false, abort = false;
    int rc = 0;
    std::string err;
    int dev = 0;
    std::string split_dir;
    size_t chunk_bytes = (size_t)8 << 20, first_cap = (size_t)16 << 20;

    ~Ingest() { for (auto &s : slot) if (s.buf) Mem::release(s.buf); }
    int fail(int code, const std::string &msg) {
        std::lock_guard<std::mutex> lk(mu);
        if (!rc) { rc = code; err = msg; }
        return code;
    }
    int grow(Slot &s, size_t need) {
        if (need <= s.cap) return 0;
        size_t ncap = std::max<size_t>(need, std::max<size_t>(s.cap * 2, first_cap));
        uint8_t *nb = (uint8_t *)Mem::alloc(ncap);
        if (!nb) return fail(ERR_NOMEM, "pinned staging allocation failed");
        if (s.len) memcpy(nb, s.buf, s.len);
        if (s.buf) Mem::release(s.buf);
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
        if (!f) return fail(ERR_ARG, "cannot write " + path);
        fprintf(f, ">%s\n", s.header.c_str());
        for (size_t i = 0; i < s.len; i += 60) {  // SeqIO.write wraps at 60 columns
            fwrite(s.buf + i, 1, std::min<size_t>(60, s.len - i), f);
            fputc('\n', f);
        }
        fclose(f);
        return 0;
    }

    // parser thread: all files, in order
    void parse_files(std::vector<std::string> paths) {
        Mem::thread_init(dev);
        int cur = 0;
        bool rec_open = false;
        std::vector<uint8_t> chunk(chunk_bytes);
        std::string carry;  // an unfinished line from the previous chunk
        auto begin = [&](const std::string &hdr) -> bool {
            std::unique_lock<std::mutex> lk(mu);
            cv.wait(lk, [&] { return !slot[cur].full || abort; });
            if (abort) return false;
            Slot &s = slot[cur];
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
            if (write_split(slot[cur])) return false;
            {
                std::lock_guard<std::mutex> lk(mu);
                slot[cur].full = true;
            }
            cv.notify_all();
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
            return push(slot[cur], p, n) == 0;
        };
        bool ok = true;
        for (size_t fi = 0; fi < paths.size() && ok; fi++) {
            int fd = ::open(paths[fi].c_str(), O_RDONLY);
            if (fd < 0) { fail(ERR_ARG, "cannot open FASTA file " + paths[fi]); ok = false; break; }
            carry.clear();
            for (;;) {
                ssize_t got = ::read(fd, chunk.data(), chunk.size());
                if (got < 0) { fail(ERR_ARG, "read error on " + paths[fi]); ok = false; break; }
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
            std::lock_guard<std::mutex> lk(mu);
            done = true;
        }
        cv.notify_all();
    }

    // calling thread: consume(slot) -> 0 or an error code for every record in order (the slot's memory is the
    // consumer's until it returns); returns 0, the consumer's code, or the parser's (message in err)
    template <class Consume>
    int run(const std::vector<std::string> &paths, Consume &&consume) {
        std::thread parser([this, paths] { parse_files(paths); });
        int rcc = 0, idx = 0;
        for (;;) {
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return slot[idx].full || done; });
                if (!slot[idx].full) break;  // done and drained
            }
            Slot &s = slot[idx];
            rcc = consume(s);
            {
                std::lock_guard<std::mutex> lk(mu);
                s.full = false;
                if (rcc) abort = true;
            }
            cv.notify_all();
            if (rcc) break;
            idx ^= 1;
        }
        parser.join();
        if (!rcc && rc) rcc = rc;
        return rcc;
    }
};

}  // namespace ingest_host
}  // namespace mimeo
