// Host-side logic of the library under the CPU sanitizers (GPU AddressSanitizer is not available on the pool):
//   * the threaded FASTA ingest (mimeo_amd/csrc/ingest_host.h: parser thread, two staging slots, hand-over, abort)
//     with malloc in the place of pinned memory and a consumer that keeps the bytes;
//   * the planning code of the pipeline (mimeo_amd/csrc/host_plan.h: super-scaffold plan, cross-product test).
// Built and run by tests/test_host_sanitize.py with -fsanitize=address,undefined and with -fsanitize=thread.
#include <cassert>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <string>
#include <vector>

#include "../../mimeo_amd/csrc/host_plan.h"
#include "../../mimeo_amd/csrc/ingest_host.h"

using namespace mimeo;

struct MallocMem {
    static void *alloc(size_t n) { return malloc(n); }
    static void release(void *p) { free(p); }
    static void thread_init(int) {}
};

#define CHECK(c) do { if (!(c)) { fprintf(stderr, "CHECK failed: %s (%s:%d)\n", #c, __FILE__, __LINE__); exit(1); } } while (0)

struct Rec { std::string name, header, seq; };

// the plain restatement the ingest is checked against (Biopython SimpleFastaParser semantics: text before the first header
// ignored, blanks and line ends inside a record dropped, id = first word of the header)
static std::vector<Rec> reference_parse(const std::vector<std::string> &texts) {
    std::vector<Rec> out;
    for (const std::string &t : texts) {
        bool open = false;
        size_t p = 0;
        while (p <= t.size()) {
            size_t nl = t.find('\n', p);
            if (nl == std::string::npos) nl = t.size();
            std::string line = t.substr(p, nl - p);
            p = nl + 1;
            if (!line.empty() && line[0] == '>') {
                std::string h = line.substr(1);
                while (!h.empty() && h.back() == '\r') h.pop_back();
                Rec r;
                r.header = h;
                size_t a = 0;
                while (a < h.size() && (h[a] == ' ' || h[a] == '\t')) a++;
                size_t b = a;
                while (b < h.size() && h[b] != ' ' && h[b] != '\t') b++;
                r.name = h.substr(a, b - a);
                out.push_back(r);
                open = true;
            } else if (open) {
                for (char c : line) if (c != ' ' && c != '\t' && c != '\r') out.back().seq.push_back(c);
            }
            if (nl == t.size()) break;
        }
    }
    return out;
}

static std::string random_fasta(std::mt19937 &rng, int nrec, bool crlf, bool trailing_newline) {
    std::string t;
    if (rng() % 2) t += "this text precedes the first header\n";
    const char *alpha = "ACGTacgtNn";
    for (int r = 0; r < nrec; r++) {
        t += ">rec" + std::to_string(rng() % 100000) + "_" + std::to_string(r) + (rng() % 2 ? " some description" : "") + (crlf ? "\r\n" : "\n");
        size_t len = (rng() % 8 == 0) ? 0 : rng() % 5000;
        if (rng() % 16 == 0) len = 300000 + rng() % 50000;   // longer than the chunk and the first staging capacity of the test
        size_t col = 0, width = 1 + rng() % 120;
        for (size_t i = 0; i < len; i++) {
            t.push_back(alpha[rng() % 10]);
            if (rng() % 997 == 0) t.push_back(' ');
            if (++col == width) { t += crlf ? "\r\n" : "\n"; col = 0; if (rng() % 50 == 0) t += "\n"; }
        }
        if (col && (r + 1 < nrec || trailing_newline)) t += crlf ? "\r\n" : "\n";
    }
    return t;
}

static void write_file(const std::string &path, const std::string &text) {
    FILE *f = fopen(path.c_str(), "wb");
    CHECK(f);
    fwrite(text.data(), 1, text.size(), f);
    fclose(f);
}

static void test_ingest(const std::string &dir) {
    std::mt19937 rng(12345);
    for (int round = 0; round < 12; round++) {
        std::vector<std::string> texts, paths;
        const int nfiles = 1 + rng() % 3;
        for (int f = 0; f < nfiles; f++) {
            texts.push_back(random_fasta(rng, 1 + rng() % 12, rng() % 3 == 0, rng() % 2));
            paths.push_back(dir + "/in_" + std::to_string(round) + "_" + std::to_string(f) + ".fa");
            write_file(paths.back(), texts.back());
        }
        const std::vector<Rec> want = reference_parse(texts);
        ingest_host::Ingest<MallocMem> in;
        in.chunk_bytes = 1 + rng() % 70000;   // lines straddle chunks
        in.first_cap = 64;                    // the staging buffers grow many times
        std::vector<Rec> got;
        int rc = in.run(paths, [&](ingest_host::Slot &s) -> int {
            got.push_back(Rec{s.name, s.header, std::string((const char *)s.buf, s.len)});
            return 0;
        });
        CHECK(rc == 0);
        CHECK(got.size() == want.size());
        for (size_t i = 0; i < got.size(); i++) {
            CHECK(got[i].name == want[i].name);
            CHECK(got[i].header == want[i].header);
            CHECK(got[i].seq == want[i].seq);
        }
        // the consumer gives up half way: the parser thread must come home
        if (want.size() >= 2) {
            ingest_host::Ingest<MallocMem> in2;
            in2.chunk_bytes = 4096;
            in2.first_cap = 64;
            size_t seen = 0;
            rc = in2.run(paths, [&](ingest_host::Slot &) -> int { return ++seen == want.size() / 2 + 1 ? -7 : 0; });
            CHECK(rc == -7);
        }
    }
    {   // a file that does not exist: the parser's error comes back
        ingest_host::Ingest<MallocMem> in;
        int rc = in.run({dir + "/no_such_file.fa"}, [&](ingest_host::Slot &) -> int { return 0; });
        CHECK(rc == ingest_host::ERR_ARG);
        CHECK(in.err.find("cannot open") != std::string::npos);
    }
    {   // split files as a side effect (60 columns per line)
        const std::string t = ">a first\nACGTACGTAC\nGT\n>b\n" + std::string(130, 'C') + "\n";
        write_file(dir + "/split_in.fa", t);
        ingest_host::Ingest<MallocMem> in;
        in.split_dir = dir;
        int n = 0;
        CHECK(in.run({dir + "/split_in.fa"}, [&](ingest_host::Slot &) -> int { n++; return 0; }) == 0);
        CHECK(n == 2);
        FILE *f = fopen((dir + "/b.fa").c_str(), "rb");
        CHECK(f);
        char buf[512];
        size_t got = fread(buf, 1, sizeof buf, f);
        fclose(f);
        CHECK(std::string(buf, got) == ">b\n" + std::string(60, 'C') + "\n" + std::string(60, 'C') + "\n" + std::string(10, 'C') + "\n");
    }
}

static void test_plan() {
    std::mt19937 rng(99);
    for (int round = 0; round < 200; round++) {
        const size_t n = 1 + rng() % 300;
        std::vector<uint64_t> len(n);
        for (auto &l : len) l = (rng() % 10 == 0) ? 0 : (rng() % 7 == 0 ? 3000000 + rng() % 1000000 : 1 + rng() % 200000);
        std::vector<uint32_t> ids;
        for (uint32_t i = 0; i < n; i++) if (rng() % 4) ids.push_back(i);
        const uint32_t spacer = 42 + rng() % 400;
        const uint64_t member_max = 2u << 20, super_len = 100000 + rng() % 8000000;
        auto plan = host_plan::plan_supers(len, ids, spacer, member_max, super_len);
        std::vector<uint32_t> seen;
        for (auto &mem : plan) {
            CHECK(!mem.empty());
            for (size_t i = 0; i < mem.size(); i++) {
                seen.push_back(mem[i].id);
                CHECK(mem[i].len == len[mem[i].id]);
                CHECK(mem[i].start % 32 == 0);
                if (i) CHECK((uint64_t)mem[i].start >= (uint64_t)mem[i - 1].start + mem[i - 1].len + spacer);
                if (mem.size() > 1) { CHECK(mem[i].len <= member_max && mem[i].len > 0); CHECK((uint64_t)mem[i].start + mem[i].len <= super_len || i == 0); }
            }
            CHECK(mem[0].start == 0);
        }
        for (auto &mem : plan) for (size_t i = 1; i < mem.size(); i++) CHECK(mem[i].id > mem[i - 1].id);   // members in scaffold order
        std::sort(seen.begin(), seen.end());
        CHECK(seen == ids);   // every scaffold once
    }
    // cross product
    for (int round = 0; round < 100; round++) {
        const size_t nt = 1 + rng() % 12, nq = 1 + rng() % 12, NA = 40, NQ = 50;
        std::vector<uint32_t> ts, qs;
        while (ts.size() < nt) { uint32_t v = rng() % NA; if (std::find(ts.begin(), ts.end(), v) == ts.end()) ts.push_back(v); }
        while (qs.size() < nq) { uint32_t v = rng() % NQ; if (std::find(qs.begin(), qs.end(), v) == qs.end()) qs.push_back(v); }
        std::vector<uint32_t> pt, pq;
        for (uint32_t t : ts) for (uint32_t q : qs) { pt.push_back(t); pq.push_back(q); }
        std::shuffle(pt.begin(), pt.end(), std::mt19937(round));
        std::shuffle(pq.begin(), pq.end(), std::mt19937(round));   // the same permutation: pairs stay pairs
        auto c = host_plan::cross_product(pt.data(), pq.data(), pt.size(), NA, NQ);
        CHECK(c.full && c.distinct == nt * nq && c.dups.empty());
        for (size_t k = 0; k < pt.size(); k++) CHECK(c.pairidx[(size_t)c.trank[pt[k]] * c.qset.size() + c.qrank[pq[k]]] == k);
        pt.push_back(pt[0]); pq.push_back(pq[0]);   // a duplicate
        c = host_plan::cross_product(pt.data(), pq.data(), pt.size(), NA, NQ);
        CHECK(c.full && c.dups.size() == 1 && c.dups[0].first == pt.size() - 1 && c.dups[0].second == 0);
        if (nt > 1 && nq > 1) {   // one cell missing
            pt.pop_back(); pq.pop_back(); pt.pop_back(); pq.pop_back();
            c = host_plan::cross_product(pt.data(), pq.data(), pt.size(), NA, NQ);
            CHECK(!c.full);
        }
    }
}

int main(int argc, char **argv) {
    CHECK(argc == 2);
    test_plan();
    test_ingest(argv[1]);
    printf("host_sanitize: ok\n");
    return 0;
}
