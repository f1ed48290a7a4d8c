// vlr_ingest.cpp — the process boundary of `call variants` in native code:
//   observation files in  (BCF2 in BGZF blocks, or text VCF)  ->  the SoA columns of vlr_batch, page-locked
//   calls file out        (results + observations -> BCF2 / text VCF records)
// What it replaces in the reference (file:line under /root/reference/src):
//   calling/variants/calling.rs:306-318, 324-339   bcf::Reader over the observation files, format-version check
//   calling/variants/preprocessing/mod.rs:818-919  read_observations: INFO integer vectors -> bincode -> ReadObservation
//   utils/mod.rs:449-474                           MiniLogProb {F16, F32}
//   calling/variants/calling.rs:517-598            WorkItem.check_* flags, is_snv_or_mnv, remove_nonstandard_alignments
//   variants/model/mod.rs:87-133                   HaplotypeIdentifier (EVENT / MATEID pairs)
//   calling/variants/mod.rs:178-600                Call::write_final_record (PROB_*, DP, AF, SAOBS/SROBS/OBS, bias symbols, AFD)
//   calling/variants/preprocessing/mod.rs:921-1038 write_observations (used here by the synthetic-workload writer of bench.py and
//                                                  by the round-trip tests; the BAM side of `preprocess` is out of scope)
// No htslib: BGZF members are located through their BSIZE fields and inflated in parallel with zlib, BCF records are decoded in
// place, every stage runs on a pool of std::threads over contiguous record ranges.  Pure host code; the only device-related call
// is vlr_host_alloc (page-locked result arrays so that vlr_batch_run_host copies them by direct DMA).
#include <dlfcn.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <sched.h>
#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <condition_variable>
#include <deque>
#include <functional>
#include <limits>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <unordered_map>
#include <unordered_set>
#include <vector>

#include "../../include/vlr.h"
#include "vlr_gpuio.h"

extern "C" void vlr_set_error(const char* msg);  // vlr_host.cpp: the text behind vlr_last_error()

#include <chrono>
namespace {

// stage timings of the last vlr_obs_read / vlr_calls_write of this process (vlr_ingest_last_timings): measurement aid
double g_ingest_t[16] = {0};
double g_ingest_total[16] = {0};  // the same stages summed over every call since the last reset (streaming reader / writer)
double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int ifail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    vlr_set_error(buf);
    return code;
}

// ------------------------------------------------------------------------------------------------ threads
// CPUs this process may actually use: the affinity mask and the cgroup CPU quota (containers report the host's hardware threads)
int effective_cpus() {
    unsigned h = std::thread::hardware_concurrency();
    int n = (int)(h ? h : 1u);
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof set, &set) == 0) n = std::min(n, std::max(1, CPU_COUNT(&set)));
    if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) {  // cgroup v2: "<quota> <period>" or "max <period>"
        char q[64];
        long period = 0;
        if (fscanf(f, "%63s %ld", q, &period) == 2 && strcmp(q, "max") != 0 && period > 0) n = std::min(n, (int)std::max(1L, (atol(q) + period - 1) / period));
        fclose(f);
    } else {
        long quota = -1, period = 0;
        if (FILE* g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) { if (fscanf(g, "%ld", &quota) != 1) quota = -1; fclose(g); }
        if (FILE* g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) { if (fscanf(g, "%ld", &period) != 1) period = 0; fclose(g); }
        if (quota > 0 && period > 0) n = std::min(n, (int)std::max(1L, (quota + period - 1) / period));
    }
    return std::max(1, n);
}
int pick_threads(int n) {
    if (n > 0) return n;
    if (const char* ev = getenv("VLR_INGEST_THREADS")) { const int v = atoi(ev); if (v > 0) return v; }
    static const int eff = effective_cpus();
    return std::max(1, std::min(eff, 256));
}
// fn(begin, end, worker) over [0, n) in contiguous ranges
template <typename F>
void parallel_ranges(int64_t n, int n_threads, F&& fn) {
    if (n <= 0) return;
    const int T = (int)std::max<int64_t>(1, std::min<int64_t>(n_threads, n));
    if (T == 1) { fn((int64_t)0, n, 0); return; }
    std::vector<std::thread> th;
    th.reserve(T);
    for (int t = 0; t < T; ++t) {
        const int64_t b = n * t / T, e = n * (t + 1) / T;
        th.emplace_back([&fn, b, e, t] { fn(b, e, t); });
    }
    for (auto& x : th) x.join();
}
// dynamic work items (blocks of very different cost)
template <typename F>
void parallel_items(int64_t n, int n_threads, F&& fn) {
    if (n <= 0) return;
    const int T = (int)std::max<int64_t>(1, std::min<int64_t>(n_threads, n));
    std::atomic<int64_t> next{0};
    auto body = [&](int t) {
        for (;;) {
            const int64_t i = next.fetch_add(1);
            if (i >= n) break;
            fn(i, t);
        }
    };
    if (T == 1) { body(0); return; }
    std::vector<std::thread> th;
    for (int t = 0; t < T; ++t) th.emplace_back(body, t);
    for (auto& x : th) x.join();
}

// ------------------------------------------------------------------------------------------------ files, BGZF
// an uninitialised byte buffer (std::vector would zero gigabytes on one thread before the workers overwrite them)
struct Blob {
    uint8_t* p = nullptr;
    size_t n = 0;
    Blob() = default;
    Blob(const Blob&) = delete;
    Blob& operator=(const Blob&) = delete;
    bool mapped = false;
    // (unmapping a file of gigabytes walks all its touched pages — 65 ms for the 2 GB of a million-record run: off the caller's thread)
    void release() {
        if (p) {
            if (mapped && n >= ((size_t)64 << 20)) { uint8_t* q = p; const size_t m = n; std::thread([q, m] { munmap(q, m); }).detach(); }
            else if (mapped) munmap(p, n ? n : 1);
            else free(p);
        }
        p = nullptr; n = 0; mapped = false;
    }
    ~Blob() { release(); }
    bool alloc(size_t bytes) {
        release();
        if (bytes >= ((size_t)8 << 20)) {  // gigabytes touched for the first time by all workers at once: 2 MB pages, 512 x fewer faults
            void* q = nullptr;
            if (posix_memalign(&q, (size_t)2 << 20, (bytes + (((size_t)2 << 20) - 1)) & ~(((size_t)2 << 20) - 1)) == 0) {
                (void)madvise(q, bytes, MADV_HUGEPAGE);
                p = (uint8_t*)q; n = bytes;
                return true;
            }
        }
        p = (uint8_t*)malloc(bytes ? bytes : 1); n = p ? bytes : 0; return p != nullptr;
    }
    // the file's pages straight from the page cache (no copy; the inflate workers fault them in side by side)
    bool map_file(const char* path) {
        release();
        const int fd = open(path, O_RDONLY);
        if (fd < 0) return false;
        struct stat st;
        if (fstat(fd, &st) != 0 || st.st_size <= 0) { close(fd); return false; }
        void* m = mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
        close(fd);
        if (m == MAP_FAILED) return false;
        (void)madvise(m, (size_t)st.st_size, MADV_WILLNEED);
        p = (uint8_t*)m; n = (size_t)st.st_size; mapped = true;
        return true;
    }
    const uint8_t* data() const { return p; }
    size_t size() const { return n; }
    uint8_t operator[](size_t i) const { return p[i]; }
};

bool read_whole_file(const char* path, Blob& out, std::string& err) {
    if (out.map_file(path)) return true;
    FILE* f = fopen(path, "rb");
    if (!f) { err = std::string("cannot open ") + path; return false; }
    fseek(f, 0, SEEK_END);
    const long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    if (!out.alloc((size_t)std::max(0L, n))) { fclose(f); err = "out of memory"; return false; }
    const size_t got = n > 0 ? fread(out.p, 1, (size_t)n, f) : 0;
    fclose(f);
    if (got != (size_t)std::max(0L, n)) { err = std::string("short read on ") + path; return false; }
    return true;
}

struct BgzfBlock { size_t off, clen; uint32_t isize; size_t out_off; uint32_t crc; };   // crc: CRC32 of the inflated bytes (member trailer)

// all gzip members carry the BGZF extra field `BC` with their total size: index them without inflating
bool bgzf_index(const Blob& raw, std::vector<BgzfBlock>& blocks) {
    size_t p = 0;
    size_t total = 0;
    while (p < raw.size()) {
        if (p + 18 > raw.size() || raw[p] != 0x1f || raw[p + 1] != 0x8b || raw[p + 2] != 8 || !(raw[p + 3] & 4)) return false;
        const unsigned xlen = raw[p + 10] | (raw[p + 11] << 8);
        size_t q = p + 12;
        const size_t xend = q + xlen;
        if (xend > raw.size()) return false;
        long bsize = -1;
        while (q + 4 <= xend) {
            const unsigned slen = raw[q + 2] | (raw[q + 3] << 8);
            if (raw[q] == 'B' && raw[q + 1] == 'C' && slen == 2 && q + 6 <= xend) bsize = (raw[q + 4] | (raw[q + 5] << 8)) + 1;
            q += 4 + slen;
        }
        if (bsize < 0 || p + (size_t)bsize > raw.size() || (size_t)bsize < xlen + 20) return false;
        const size_t cdata = xend, cend = p + (size_t)bsize - 8;
        const uint32_t isize = raw[cend + 4] | (raw[cend + 5] << 8) | (raw[cend + 6] << 16) | ((uint32_t)raw[cend + 7] << 24);
        const uint32_t crc = raw[cend] | (raw[cend + 1] << 8) | (raw[cend + 2] << 16) | ((uint32_t)raw[cend + 3] << 24);
        blocks.push_back({cdata, cend - cdata, isize, total, crc});
        total += isize;
        p += (size_t)bsize;
    }
    return true;
}

// libdeflate (a system library next to zlib, about twice zlib's inflate rate) is taken through dlopen when it is installed — the
// image carries the runtime library without its header; its C API (v1.x) is declared here.  zlib is the fallback.
struct LibDeflate {
    void* (*alloc_d)() = nullptr;
    int (*decompress)(void*, const void*, size_t, void*, size_t, size_t*) = nullptr;
    void (*free_d)(void*) = nullptr;
    void* (*alloc_c)(int) = nullptr;
    size_t (*compress)(void*, const void*, size_t, void*, size_t) = nullptr;
    size_t (*bound)(void*, size_t) = nullptr;
    void (*free_c)(void*) = nullptr;
    uint32_t (*crc32)(uint32_t, const void*, size_t) = nullptr;
    bool ok = false;
    LibDeflate() {
        if (getenv("VLR_NO_LIBDEFLATE")) return;
        void* h = dlopen("libdeflate.so.0", RTLD_NOW | RTLD_LOCAL);
        if (!h) return;
        alloc_d = (void* (*)())dlsym(h, "libdeflate_alloc_decompressor");
        decompress = (int (*)(void*, const void*, size_t, void*, size_t, size_t*))dlsym(h, "libdeflate_deflate_decompress");
        free_d = (void (*)(void*))dlsym(h, "libdeflate_free_decompressor");
        alloc_c = (void* (*)(int))dlsym(h, "libdeflate_alloc_compressor");
        compress = (size_t (*)(void*, const void*, size_t, void*, size_t))dlsym(h, "libdeflate_deflate_compress");
        bound = (size_t (*)(void*, size_t))dlsym(h, "libdeflate_deflate_compress_bound");
        free_c = (void (*)(void*))dlsym(h, "libdeflate_free_compressor");
        crc32 = (uint32_t (*)(uint32_t, const void*, size_t))dlsym(h, "libdeflate_crc32");
        ok = alloc_d && decompress && free_d && alloc_c && compress && bound && free_c && crc32;
    }
};
const LibDeflate& libdeflate() { static LibDeflate L; return L; }

// CRC32 of an inflated member against the value in its trailer (RFC 1952; htslib's bgzf.c checks it for every block it reads)
bool member_crc_ok(const uint8_t* data, size_t n, uint32_t want) {
    const LibDeflate& ld = libdeflate();
    const uint32_t got = ld.ok ? ld.crc32(0, data, n) : (uint32_t)crc32(crc32(0L, Z_NULL, 0), data, (uInt)n);
    return got == want;
}
// one BGZF member: raw DEFLATE -> dlen bytes whose CRC32 is `crc`
bool inflate_raw(const uint8_t* src, size_t clen, uint8_t* dst, size_t dlen, uint32_t crc) {
    const LibDeflate& ld = libdeflate();
    if (ld.ok) {
        thread_local void* d = ld.alloc_d();
        size_t got = 0;
        if (d && ld.decompress(d, src, clen, dst, dlen, &got) == 0 && got == dlen) return member_crc_ok(dst, dlen, crc);
        // fall through to zlib on any doubt
    }
    z_stream zs;
    memset(&zs, 0, sizeof zs);
    if (inflateInit2(&zs, -15) != Z_OK) return false;
    zs.next_in = const_cast<Bytef*>(src); zs.avail_in = (uInt)clen;
    zs.next_out = dst; zs.avail_out = (uInt)dlen;
    const int rc = inflate(&zs, Z_FINISH);
    const bool ok = (rc == Z_STREAM_END) && zs.total_out == dlen;
    inflateEnd(&zs);
    return ok && member_crc_ok(dst, dlen, crc);
}

// file contents, decompressed: BGZF (parallel), plain gzip (sequential), or as is
bool load_inflated(const char* path, Blob& out_blob, int n_threads, std::string& err) {
    Blob raw;
    const double t0 = now_s();
    if (!read_whole_file(path, raw, err)) return false;
    g_ingest_t[0] += now_s() - t0;
    std::vector<uint8_t> out;
    auto finish = [&](std::vector<uint8_t>& v) { if (!out_blob.alloc(v.size())) { err = "out of memory"; return false; } memcpy(out_blob.p, v.data(), v.size()); return true; };
    if (raw.size() < 2 || raw[0] != 0x1f || raw[1] != 0x8b) { std::swap(out_blob.p, raw.p); std::swap(out_blob.n, raw.n); std::swap(out_blob.mapped, raw.mapped); return true; }
    std::vector<BgzfBlock> blocks;
    if (bgzf_index(raw, blocks)) {
        size_t total = 0;
        for (auto& b : blocks) total += b.isize;
        if (!out_blob.alloc(total)) { err = "out of memory"; return false; }
        std::atomic<bool> bad{false};
        const double t1 = now_s();
        parallel_items((int64_t)blocks.size(), n_threads, [&](int64_t i, int) {
            const BgzfBlock& b = blocks[(size_t)i];
            if (b.isize == 0) return;
            if (!inflate_raw(raw.data() + b.off, b.clen, out_blob.p + b.out_off, b.isize, b.crc)) bad = true;
        });
        g_ingest_t[1] += now_s() - t1;
        if (bad) { err = std::string("corrupt BGZF block in ") + path; return false; }
        return true;
    }
    // generic (multi-member) gzip
    z_stream zs;
    memset(&zs, 0, sizeof zs);
    if (inflateInit2(&zs, 15 + 32) != Z_OK) { err = "zlib init failed"; return false; }
    zs.next_in = raw.p; zs.avail_in = 0;
    const uint8_t* const raw_end = raw.p + raw.size();
    out.resize(std::max<size_t>(raw.size() * 4, 1 << 16));
    size_t have = 0;
    for (;;) {
        if (have == out.size()) out.resize(out.size() * 2);
        // zlib counts in 32 bits: hand the input over in pieces (a plain-gzip file above 2 GiB used to end at the first piece)
        if (zs.avail_in == 0 && zs.next_in < raw_end) zs.avail_in = (uInt)std::min<size_t>((size_t)(raw_end - zs.next_in), 0x40000000);
        zs.next_out = out.data() + have; zs.avail_out = (uInt)std::min<size_t>(out.size() - have, 0x7fffffff);
        const int rc = inflate(&zs, Z_NO_FLUSH);
        have = (size_t)(zs.next_out - out.data());
        if (rc == Z_STREAM_END) {
            if (zs.avail_in == 0 && zs.next_in >= raw_end) break;
            if (inflateReset(&zs) != Z_OK) { inflateEnd(&zs); err = "zlib reset failed"; return false; }
            continue;
        }
        if (rc != Z_OK) { inflateEnd(&zs); err = std::string("corrupt gzip stream in ") + path; return false; }
    }
    inflateEnd(&zs);
    out.resize(have);
    return finish(out);
}

// one BGZF member holding `n` bytes (n <= 0xff00)
void bgzf_deflate_block(const uint8_t* data, size_t n, int level, std::vector<uint8_t>& out) {
    uLong bound = compressBound((uLong)n) + 64;
    const size_t at = out.size();
    size_t clen = 0;
    const LibDeflate& ld = libdeflate();
    if (ld.ok) {
        thread_local int c_level = -1;
        thread_local void* c = nullptr;
        if (c_level != level) { if (c) ld.free_c(c); c = ld.alloc_c(level); c_level = level; }
        if (c) {
            bound = (uLong)ld.bound(c, n) + 64;
            out.resize(at + 18 + bound + 8);
            clen = ld.compress(c, data, n, out.data() + at + 18, bound);
        }
    }
    if (clen == 0 || clen + 26 > 0xffff) {
        bound = compressBound((uLong)n) + 64;
        out.resize(at + 18 + bound + 8);
        z_stream zs;
        memset(&zs, 0, sizeof zs);
        deflateInit2(&zs, level, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY);
        zs.next_in = const_cast<Bytef*>(data); zs.avail_in = (uInt)n;
        zs.next_out = out.data() + at + 18; zs.avail_out = (uInt)bound;
        deflate(&zs, Z_FINISH);
        clen = zs.total_out;
        deflateEnd(&zs);
    }
    static const uint8_t head[16] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0};
    memcpy(out.data() + at, head, 16);
    const unsigned bsize = (unsigned)(clen + 25);
    out[at + 16] = (uint8_t)(bsize & 0xff); out[at + 17] = (uint8_t)(bsize >> 8);
    const uint32_t crc = ld.ok ? ld.crc32(0, data, n) : (uint32_t)crc32(crc32(0L, Z_NULL, 0), data, (uInt)n), isz = (uint32_t)n;
    uint8_t* t = out.data() + at + 18 + clen;
    for (int i = 0; i < 4; ++i) { t[i] = (uint8_t)(crc >> (8 * i)); t[4 + i] = (uint8_t)(isz >> (8 * i)); }
    out.resize(at + 18 + clen + 8);
}
const uint8_t kBgzfEof[28] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 0x42, 0x43, 2, 0, 0x1b, 0, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0};

// compress `data` into BGZF members of 0xff00 bytes in parallel and write the file
// File output behind the streaming calls writer: the compressed members of a chunk are handed to one I/O thread, so that copying
// them into the page cache (tens of milliseconds per hundred megabytes, serial) runs beside the encoding of the next chunk.
struct AsyncSink {
    FILE* f = nullptr;
    std::thread th;
    std::mutex mu;
    std::condition_variable cv;
    std::deque<std::vector<std::vector<uint8_t>>> q;
    bool closing = false, failed = false;
    explicit AsyncSink(FILE* file) : f(file) {
        th = std::thread([this] {
            for (;;) {
                std::vector<std::vector<uint8_t>> job;
                {
                    std::unique_lock<std::mutex> lk(mu);
                    cv.wait(lk, [this] { return closing || !q.empty(); });
                    if (q.empty()) return;
                    job = std::move(q.front());
                    q.pop_front();
                }
                bool ok = true;
                for (auto& c : job) ok = ok && fwrite(c.data(), 1, c.size(), f) == c.size();
                if (!ok) { std::lock_guard<std::mutex> g(mu); failed = true; }
                cv.notify_all();
            }
        });
    }
    void push(std::vector<std::vector<uint8_t>>&& job) {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [this] { return q.size() < 3; });   // (bounded: at most three chunks of compressed members in memory)
        q.push_back(std::move(job));
        cv.notify_all();
    }
    bool finish() {   // everything handed over is in the file (or failed)
        { std::lock_guard<std::mutex> g(mu); closing = true; }
        cv.notify_all();
        if (th.joinable()) th.join();
        return !failed;
    }
    ~AsyncSink() { (void)finish(); }
};
bool write_bgzf_stream(FILE* f, const std::vector<const std::vector<uint8_t>*>& parts, int n_threads, int level, bool with_eof, std::string& err, AsyncSink* sink = nullptr);
bool write_bgzf_file(const char* path, const std::vector<const std::vector<uint8_t>*>& parts, int n_threads, int level, std::string& err) {
    FILE* f = fopen(path, "wb");
    if (!f) { err = std::string("cannot create ") + path; return false; }
    bool ok = write_bgzf_stream(f, parts, n_threads, level, true, err);
    ok = (fclose(f) == 0) && ok;
    if (!ok && err.empty()) err = std::string("write failed on ") + path;
    return ok;
}
// the parts, logically concatenated, as BGZF members appended to an open file (a short last member is legal BGZF)
bool write_bgzf_stream(FILE* f, const std::vector<const std::vector<uint8_t>*>& parts, int n_threads, int level, bool with_eof, std::string& err, AsyncSink* sink) {
    // the parts are logically concatenated; cut into blocks without copying more than one block at a time
    size_t total = 0;
    std::vector<size_t> start;
    for (auto* p : parts) { start.push_back(total); total += p->size(); }
    const size_t B = 0xff00;
    const int64_t nb = (int64_t)((total + B - 1) / B);
    std::vector<std::vector<uint8_t>> comp((size_t)nb);
    parallel_items(nb, n_threads, [&](int64_t bi, int) {
        const size_t b0 = (size_t)bi * B, b1 = std::min(total, b0 + B);
        std::vector<uint8_t> tmp;
        tmp.reserve(b1 - b0);
        size_t k = (size_t)(std::upper_bound(start.begin(), start.end(), b0) - start.begin()) - 1;
        size_t pos = b0;
        while (pos < b1) {
            const std::vector<uint8_t>& P = *parts[k];
            const size_t o = pos - start[k], n = std::min(P.size() - o, b1 - pos);
            tmp.insert(tmp.end(), P.begin() + (long)o, P.begin() + (long)(o + n));
            pos += n;
            ++k;
        }
        bgzf_deflate_block(tmp.data(), tmp.size(), level, comp[(size_t)bi]);
    });
    if (sink) {   // (the streaming writer: the end-of-file member is written by its close, after the sink has drained)
        sink->push(std::move(comp));
        return true;
    }
    bool ok = true;
    for (auto& c : comp) ok = ok && fwrite(c.data(), 1, c.size(), f) == c.size();
    if (with_eof) ok = ok && fwrite(kBgzfEof, 1, 28, f) == 28;
    if (!ok) err = "write failed";
    return ok;
}

// ------------------------------------------------------------------------------------------------ f16
float half_to_float(uint16_t h) {
    const uint32_t s = (uint32_t)(h >> 15) << 31, e = (h >> 10) & 0x1f, m = h & 0x3ff;
    uint32_t bits;
    if (e == 0) {
        if (m == 0) bits = s;
        else {  // subnormal: normalise
            int k = 0;
            uint32_t mm = m;
            while (!(mm & 0x400)) { mm <<= 1; ++k; }
            bits = s | ((uint32_t)(127 - 15 - k + 1) << 23) | ((mm & 0x3ff) << 13);
        }
    } else if (e == 31) bits = s | 0x7f800000u | (m << 13);
    else bits = s | ((e + 112) << 23) | (m << 13);
    float f;
    memcpy(&f, &bits, 4);
    return f;
}
uint16_t float_to_half(float f) {  // round to nearest even (half::f16::from_f64 of an f32 value rounds once, as here)
    uint32_t x;
    memcpy(&x, &f, 4);
    const uint32_t s = (x >> 16) & 0x8000u;
    const int32_t e = (int32_t)((x >> 23) & 0xff) - 127 + 15;
    uint32_t m = x & 0x7fffffu;
    if (((x >> 23) & 0xff) == 0xff) return (uint16_t)(s | 0x7c00u | (m ? 0x200u : 0));
    if (e >= 31) return (uint16_t)(s | 0x7c00u);
    if (e <= 0) {
        if (e < -10) return (uint16_t)s;
        m |= 0x800000u;
        const int shift = 14 - e;
        uint32_t hm = m >> shift;
        const uint32_t rem = m & ((1u << shift) - 1), halfway = 1u << (shift - 1);
        if (rem > halfway || (rem == halfway && (hm & 1))) ++hm;
        return (uint16_t)(s | hm);
    }
    uint32_t hm = m >> 13;
    const uint32_t rem = m & 0x1fffu;
    uint32_t out = s | ((uint32_t)e << 10) | hm;
    if (rem > 0x1000u || (rem == 0x1000u && (hm & 1))) ++out;  // carries into the exponent correctly
    return (uint16_t)out;
}

// ------------------------------------------------------------------------------------------------ header
struct Header {
    std::string text;
    std::vector<std::string> dict;     // FILTER/INFO/FORMAT ids by dictionary index
    std::vector<std::string> contigs;  // by index
    bool version_ok = false;
};
std::string attr(const std::string& body, const char* key) {
    const std::string k = std::string(key) + "=";
    size_t p = 0;
    while ((p = body.find(k, p)) != std::string::npos) {
        if (p == 0 || body[p - 1] == ',' || body[p - 1] == '<') {
            size_t q = p + k.size(), e = q;
            while (e < body.size() && body[e] != ',' && body[e] != '>') ++e;
            return body.substr(q, e - q);
        }
        p += k.size();
    }
    return "";
}
void parse_header(const std::string& text, Header& h) {
    h.text = text;
    h.dict.assign(1, "PASS");
    std::unordered_map<std::string, int> seen{{"PASS", 0}};
    size_t p = 0;
    int next = 1, cnext = 0;
    while (p < text.size()) {
        size_t e = text.find('\n', p);
        if (e == std::string::npos) e = text.size();
        const std::string line = text.substr(p, e - p);
        p = e + 1;
        if (line.compare(0, 2, "##") != 0) continue;
        if (line == "##varlociraptor_observation_format_version=15") h.version_ok = true;  // preprocessing/mod.rs:810, calling.rs:324-339
        const bool dictline = line.compare(0, 9, "##FILTER=") == 0 || line.compare(0, 7, "##INFO=") == 0 || line.compare(0, 9, "##FORMAT=") == 0;
        if (dictline) {
            const std::string body = line.substr(line.find('<'));
            const std::string id = attr(body, "ID"), idx = attr(body, "IDX");
            if (id.empty() || seen.count(id)) continue;
            const int i = idx.empty() ? next : atoi(idx.c_str());
            seen[id] = i;
            if ((int)h.dict.size() <= i) h.dict.resize((size_t)i + 1);
            h.dict[(size_t)i] = id;
            next = std::max(next, i + 1);
        } else if (line.compare(0, 9, "##contig=") == 0) {
            const std::string body = line.substr(line.find('<'));
            const std::string id = attr(body, "ID"), idx = attr(body, "IDX");
            const int i = idx.empty() ? cnext : atoi(idx.c_str());
            if ((int)h.contigs.size() <= i) h.contigs.resize((size_t)i + 1);
            h.contigs[(size_t)i] = id;
            cnext = std::max(cnext, i + 1);
        }
    }
}

// ------------------------------------------------------------------------------------------------ observation records
// (enum VlrObsField: vlr_gpuio.h — the device decoder indexes the same table)
const char* const kFieldName[FD_N] = {
    "PROB_MAPPING", "PROB_REF", "PROB_ALT", "PROB_MISSED_ALLELE", "PROB_SAMPLE_ALT", "PROB_DOUBLE_OVERLAP", "PROB_HIT_BASE",
    "STRAND", "READ_ORIENTATION", "READ_POSITION", "ALT_LOCUS", "SOFTCLIPPED", "PAIRED", "IS_MAX_MAPQ",
    "PROB_HOMOPOLYMER_ARTIFACT_OBSERVABLE", "PROB_HOMOPOLYMER_VARIANT_OBSERVABLE", "HOMOPOLYMER_INDEL_LEN", "THIRD_ALLELE_EVIDENCE",
    "IMPRECISE", "EVENT", "MATEID", "HETEROZYGOSITY", "SOMATIC_EFFECTIVE_MUTATION_RATE"};

// one decoded record of one sample file, before it is placed into the batch
struct RecView {
    int32_t contig = -1;          // index into the file's contig dictionary (BCF) / interned name (text)
    int64_t pos = 0;              // 1-based
    std::string id, ref, alt, event, mateid;
    bool imprecise = false;
    double het_ln = NAN, som_ln = NAN;
    std::vector<uint8_t> vec[FD_N_VEC];  // bincode bytes of every vector field (LE u16 words of the INFO integers)
    // the MiniLogProb and enum vectors of a BCF record (the bulk of it) are decoded straight from the typed integers:
    // word k of the field = low 16 bits of element k; wstride = bytes per element (1: int8, sign-extended; 2; 4), 0 = use vec
    const uint8_t* wsrc[FD_N_VEC] = {};
    uint32_t wn[FD_N_VEC] = {};
    int wstride[FD_N_VEC] = {};
    bool present[FD_N_VEC] = {};
    void reset() {
        id.clear(); ref.clear(); alt.clear(); event.clear(); mateid.clear();
        imprecise = false; het_ln = NAN; som_ln = NAN;
        for (int i = 0; i < FD_N_VEC; ++i) { vec[i].clear(); present[i] = false; wsrc[i] = nullptr; wn[i] = 0; wstride[i] = 0; }
    }
};
inline bool hot_field(int f) { return f <= FD_PROB_HIT_BASE || (f >= FD_STRAND && f <= FD_ALTLOCUS); }
template <int ST>
inline uint32_t word16(const uint8_t* p, size_t k) {
    if (ST == 4) return (uint32_t)p[4 * k] | ((uint32_t)p[4 * k + 1] << 8);
    if (ST == 2) return (uint32_t)p[2 * k] | ((uint32_t)p[2 * k + 1] << 8);
    return (uint32_t)(uint16_t)(int16_t)(int8_t)p[k];
}
// Vec<MiniLogProb>: u64 length (4 words), then per element a u32 tag (2 words) and an f16 (1 word) or an f32 (2 words)
// f16 -> f32 bit patterns of all 65 536 halves (256 kB, built once): the decoder's inner loop selects between the table entry and
// the f32 payload with a mask instead of branching on the element's tag — which of the two a MiniLogProb is (utils/mod.rs:449-474:
// f16 below -10 when that loses nothing) varies from element to element and is not predictable
const uint32_t* half_bits_table() {
    static const std::vector<uint32_t> tab = [] {
        std::vector<uint32_t> t(65536);
        for (uint32_t h = 0; h < 65536; ++h) { const float f = half_to_float((uint16_t)h); memcpy(&t[h], &f, 4); }
        return t;
    }();
    return tab.data();
}
template <int ST>
bool mini_words(const uint8_t* p, uint32_t nw, uint64_t n, float* dst) {
    size_t k = 4;
    uint64_t i = 0;
    {
        const uint32_t* hb = half_bits_table();
        uint32_t bad = 0;
        uint32_t* out = reinterpret_cast<uint32_t*>(dst);
        for (; i < n && k + 4 <= nw; ++i) {  // all four words of the widest element are readable: no bounds test per variant
            const uint32_t tag = word16<ST>(p, k) | (word16<ST>(p, k + 1) << 16);
            const uint32_t lo = word16<ST>(p, k + 2), hi = word16<ST>(p, k + 3);
            const uint32_t m32 = 0u - (tag & 1u);               // all ones for an f32 payload
            out[i] = ((lo | (hi << 16)) & m32) | (hb[lo] & ~m32);
            bad |= tag >> 1;
            k += 3 + (tag & 1u);
        }
        if (bad) return false;
    }
    for (; i < n; ++i) {
        if (k + 3 > nw) return false;
        const uint32_t tag = word16<ST>(p, k) | (word16<ST>(p, k + 1) << 16);
        if (tag == 0) { dst[i] = half_to_float((uint16_t)word16<ST>(p, k + 2)); k += 3; }
        else if (tag == 1) {
            if (k + 4 > nw) return false;
            const uint32_t bits = word16<ST>(p, k + 2) | (word16<ST>(p, k + 3) << 16);
            float f; memcpy(&f, &bits, 4); dst[i] = f; k += 4;
        } else return false;
    }
    return true;
}
// The seven mandatory MiniLogProb vectors of a record side by side: where element i + 1 of a vector starts depends on the tag of its
// element i, a chain of dependent loads per vector — seven vectors give the core seven independent chains to overlap.  Elements
// [0, returned count) of every vector are decoded; the caller finishes the (short) remainders with mini_words' own loop.
template <int ST>
uint64_t mini_words_x7(const uint8_t* const* p, const uint32_t* nw, uint64_t n, float* const* dst, size_t* k_out, bool* bad_out) {
    const uint32_t* hb = half_bits_table();
    size_t k[7];
    uint32_t bad = 0;
    for (int c = 0; c < 7; ++c) k[c] = 4;
    uint64_t i = 0;
    for (; i < n; ++i) {
        bool room = true;
        for (int c = 0; c < 7; ++c) room = room && (k[c] + 4 <= nw[c]);
        if (!room) break;
        for (int c = 0; c < 7; ++c) {
            const uint8_t* q = p[c];
            const uint32_t tag = word16<ST>(q, k[c]) | (word16<ST>(q, k[c] + 1) << 16);
            const uint32_t lo = word16<ST>(q, k[c] + 2), hi = word16<ST>(q, k[c] + 3);
            const uint32_t m32 = 0u - (tag & 1u);
            reinterpret_cast<uint32_t*>(dst[c])[i] = ((lo | (hi << 16)) & m32) | (hb[lo] & ~m32);
            bad |= tag >> 1;
            k[c] += 3 + (tag & 1u);
        }
    }
    for (int c = 0; c < 7; ++c) k_out[c] = k[c];
    *bad_out = bad != 0;
    return i;
}
// the remainder of one vector from word k on (elements i0 .. n)
template <int ST>
bool mini_words_from(const uint8_t* p, uint32_t nw, uint64_t n, float* dst, size_t k, uint64_t i0) {
    for (uint64_t i = i0; i < n; ++i) {
        if (k + 3 > nw) return false;
        const uint32_t tag = word16<ST>(p, k) | (word16<ST>(p, k + 1) << 16);
        if (tag == 0) { dst[i] = half_to_float((uint16_t)word16<ST>(p, k + 2)); k += 3; }
        else if (tag == 1) {
            if (k + 4 > nw) return false;
            const uint32_t bits = word16<ST>(p, k + 2) | (word16<ST>(p, k + 3) << 16);
            float f; memcpy(&f, &bits, 4); dst[i] = f; k += 4;
        } else return false;
    }
    return true;
}
template <int ST>
uint64_t len_words(const uint8_t* p, uint32_t nw) {
    if (nw < 4) return ~0ull;
    return (uint64_t)word16<ST>(p, 0) | ((uint64_t)word16<ST>(p, 1) << 16) | ((uint64_t)word16<ST>(p, 2) << 32) | ((uint64_t)word16<ST>(p, 3) << 48);
}
template <int ST, typename Put>
bool enum_words(const uint8_t* p, uint32_t nw, uint64_t n, Put&& put) {
    if ((uint64_t)nw < 4 + 2 * n) return false;
    for (uint64_t i = 0; i < n; ++i) put(i, word16<ST>(p, 4 + 2 * i) | (word16<ST>(p, 5 + 2 * i) << 16));
    return true;
}

struct Cursor {  // bincode reader (little-endian, u64 lengths, u32 enum tags, u8 Option tags)
    const uint8_t* p; const uint8_t* e; bool bad = false;
    bool need(size_t n) { if ((size_t)(e - p) < n) { bad = true; return false; } return true; }
    uint8_t u8() { if (!need(1)) return 0; return *p++; }
    uint32_t u32() { if (!need(4)) return 0; uint32_t v; memcpy(&v, p, 4); p += 4; return v; }
    uint64_t u64() { if (!need(8)) return 0; uint64_t v; memcpy(&v, p, 8); p += 8; return v; }
    float mini() {  // MiniLogProb (utils/mod.rs:449-474)
        const uint32_t tag = u32();
        if (tag == 0) { if (!need(2)) return 0; uint16_t h; memcpy(&h, p, 2); p += 2; return half_to_float(h); }
        if (tag == 1) { if (!need(4)) return 0; float f; memcpy(&f, p, 4); p += 4; return f; }
        bad = true; return 0;
    }
};

// growable array of a trivially copyable type WITHOUT value initialisation (std::vector::resize would write every element once
// before the decoder writes it again) and with 2 MB pages for large capacities
template <typename T>
struct RawVec {
    T* p = nullptr;
    size_t n = 0, cap = 0;
    RawVec() = default;
    RawVec(const RawVec&) = delete;
    RawVec& operator=(const RawVec&) = delete;
    RawVec(RawVec&& o) noexcept : p(o.p), n(o.n), cap(o.cap) { o.p = nullptr; o.n = o.cap = 0; }
    RawVec& operator=(RawVec&& o) noexcept { if (this != &o) { free(p); p = o.p; n = o.n; cap = o.cap; o.p = nullptr; o.n = o.cap = 0; } return *this; }
    ~RawVec() { free(p); }
    void reserve(size_t c) {
        if (c <= cap) return;
        void* q = nullptr;
        const size_t bytes = c * sizeof(T);
        if (bytes >= ((size_t)8 << 20) && posix_memalign(&q, (size_t)2 << 20, (bytes + (((size_t)2 << 20) - 1)) & ~(((size_t)2 << 20) - 1)) == 0) (void)madvise(q, bytes, MADV_HUGEPAGE);
        else q = malloc(bytes ? bytes : 1);
        if (n) memcpy(q, p, n * sizeof(T));
        free(p);
        p = (T*)q; cap = c;
    }
    void resize(size_t m) { if (m > cap) reserve(std::max(m, cap + cap / 2 + 1024)); n = m; }
    size_t size() const { return n; }
    T* data() { return p; }
    const T* data() const { return p; }
};

// columns of one sample file for a contiguous range of its records
struct Chunk {
    std::vector<uint32_t> n_obs;  // per record
    RawVec<float> col[9];
    RawVec<uint32_t> flags;
    RawVec<int32_t> third;
    std::vector<uint8_t> is_hp, imprecise;
    std::vector<int32_t> contig;
    std::vector<int64_t> pos;
    std::vector<double> het, som;
    std::string pool;                       // NUL-terminated strings
    std::vector<uint32_t> id, ref, alt, hap;  // offsets into pool (hap: 0xffffffff = none)
    std::string error;
};

uint32_t pool_add(std::string& pool, const std::string& s) {
    const uint32_t at = (uint32_t)pool.size();
    pool.append(s);
    pool.push_back('\0');
    return at;
}

// the per-record (cold) fields of a decoded record: site, strings, haplotype identifier, priors
bool push_cold(const RecView& r, Chunk& c, uint32_t n, bool is_hp, std::string& err) {
    c.n_obs.push_back(n);
    c.is_hp.push_back(is_hp ? 1 : 0);
    c.imprecise.push_back(r.imprecise ? 1 : 0);
    c.contig.push_back(r.contig);
    c.pos.push_back(r.pos);
    c.het.push_back(r.het_ln);
    c.som.push_back(r.som_ln);
    c.id.push_back(pool_add(c.pool, r.id.empty() ? std::string(".") : r.id));
    c.ref.push_back(pool_add(c.pool, r.ref));
    c.alt.push_back(pool_add(c.pool, r.alt));
    // HaplotypeIdentifier::from (variants/model/mod.rs:87-133): EVENT, else the sorted pair (record id, MATEID)
    if (!r.event.empty()) c.hap.push_back(pool_add(c.pool, r.event.substr(0, r.event.find(','))));
    else if (!r.mateid.empty()) {
        if (r.id.empty() || r.id == ".") { err = "breakend with MATEID but without record ID"; return false; }
        std::string a = r.id, b = r.mateid.substr(0, r.mateid.find(','));
        if (b < a) std::swap(a, b);
        c.hap.push_back(pool_add(c.pool, a + "-" + b));
    } else c.hap.push_back(0xffffffffu);
    return true;
}

// read_observations (preprocessing/mod.rs:818-919) of one record into the chunk
bool decode_into(const RecView& r, Chunk& c, std::string& err) {
    static const int kMini[7] = {FD_PROB_MAPPING, FD_PROB_ALT, FD_PROB_REF, FD_PROB_MISSED, FD_PROB_SAMPLE_ALT, FD_PROB_DOUBLE_OVERLAP, FD_PROB_HIT_BASE};
    for (int k = 0; k < 7; ++k)
        if (!r.present[kMini[k]]) { err = std::string("No varlociraptor observations found in record (") + kFieldName[kMini[k]] + ")"; return false; }
    for (int f : {FD_STRAND, FD_ORIENT, FD_READPOS, FD_ALTLOCUS, FD_SOFTCLIPPED, FD_PAIRED, FD_MAX_MAPQ})
        if (!r.present[f]) { err = std::string("No varlociraptor observations found in record (") + kFieldName[f] + ")"; return false; }
    uint64_t n = 0;
    const size_t base = c.flags.size();
    bool all7 = false;   // the seven vectors decoded side by side (all int32-typed, the rule for values up to 65535)
    {
        bool same = true;
        for (int k = 0; k < 7; ++k) same = same && r.wstride[kMini[k]] == 4;
        if (same) {
            const uint8_t* ps[7]; uint32_t nws[7]; float* ds[7];
            for (int k = 0; k < 7; ++k) { ps[k] = r.wsrc[kMini[k]]; nws[k] = r.wn[kMini[k]]; }
            n = len_words<4>(ps[0], nws[0]);
            bool ok = n <= (1u << 28);
            for (int k = 1; k < 7 && ok; ++k) ok = len_words<4>(ps[k], nws[k]) == n;
            if (!ok) { err = "inconsistent observation vector lengths"; return false; }
            for (int k = 0; k < 7; ++k) { c.col[k].resize(base + n); ds[k] = c.col[k].data() + base; }
            size_t kk[7];
            bool bad = false;
            const uint64_t done = mini_words_x7<4>(ps, nws, n, ds, kk, &bad);
            if (bad) { err = "truncated PROB vector"; return false; }
            for (int k = 0; k < 7; ++k)
                if (!mini_words_from<4>(ps[k], nws[k], n, ds[k], kk[k], done)) { err = std::string("truncated ") + kFieldName[kMini[k]]; return false; }
            all7 = true;
        }
    }
    for (int k = 0; k < 7 && !all7; ++k) {  // column order of vlr_batch: pm, pa, pr, miss, psa, pdo, phb
        const int f = kMini[k];
        auto& col = c.col[k];
        if (r.wstride[f]) {
            const uint8_t* p = r.wsrc[f];
            const uint32_t nw = r.wn[f];
            const int st = r.wstride[f];
            const uint64_t m = st == 4 ? len_words<4>(p, nw) : st == 2 ? len_words<2>(p, nw) : len_words<1>(p, nw);
            if (k == 0) n = m;
            if (m != n || m > (1u << 28)) { err = "inconsistent observation vector lengths"; return false; }
            col.resize(base + n);
            float* dst = col.data() + base;
            const bool ok = st == 4 ? mini_words<4>(p, nw, n, dst) : st == 2 ? mini_words<2>(p, nw, n, dst) : mini_words<1>(p, nw, n, dst);
            if (!ok) { err = std::string("truncated ") + kFieldName[f]; return false; }
            continue;
        }
        const auto& v = r.vec[f];
        Cursor cu{v.data(), v.data() + v.size()};
        const uint64_t m = cu.u64();
        if (k == 0) n = m;
        if (m != n || m > (1u << 28)) { err = "inconsistent observation vector lengths"; return false; }
        col.resize(base + n);
        float* dst = col.data() + base;
        for (uint64_t i = 0; i < n; ++i) dst[i] = cu.mini();
        if (cu.bad) { err = std::string("truncated ") + kFieldName[f]; return false; }
    }
    c.flags.resize(base + n);
    uint32_t* fl = c.flags.data() + base;
    for (uint64_t i = 0; i < n; ++i) fl[i] = 0;
    auto enums = [&](int field, auto&& put) -> bool {
        if (r.wstride[field]) {
            const uint8_t* p = r.wsrc[field];
            const uint32_t nw = r.wn[field];
            const int st = r.wstride[field];
            const uint64_t m = st == 4 ? len_words<4>(p, nw) : st == 2 ? len_words<2>(p, nw) : len_words<1>(p, nw);
            if (m != n) { err = std::string("length of ") + kFieldName[field]; return false; }
            const bool ok = st == 4 ? enum_words<4>(p, nw, n, put) : st == 2 ? enum_words<2>(p, nw, n, put) : enum_words<1>(p, nw, n, put);
            if (!ok) { err = std::string("truncated ") + kFieldName[field]; return false; }
            return true;
        }
        const auto& v = r.vec[field];
        Cursor cu{v.data(), v.data() + v.size()};
        if (cu.u64() != n) { err = std::string("length of ") + kFieldName[field]; return false; }
        for (uint64_t i = 0; i < n; ++i) put(i, cu.u32());
        if (cu.bad) { err = std::string("truncated ") + kFieldName[field]; return false; }
        return true;
    };
    if (!enums(FD_STRAND, [&](uint64_t i, uint32_t v) { fl[i] |= (v & 3u) << VLR_F_STRAND_SHIFT; })) return false;
    // bio_types SequenceReadPairOrientation: F1R2 0, F2R1 1, ..., None 8 (the reference fixture pins None = 8)
    if (!enums(FD_ORIENT, [&](uint64_t i, uint32_t v) {
            const uint32_t o = v == 0 ? VLR_ORIENT_F1R2 : v == 1 ? VLR_ORIENT_F2R1 : v == 8 ? VLR_ORIENT_NONE : VLR_ORIENT_OTHER;
            fl[i] |= o << VLR_F_ORIENT_SHIFT;
        })) return false;
    if (!enums(FD_READPOS, [&](uint64_t i, uint32_t v) { if (v == 0) fl[i] |= VLR_F_READPOS_MAJOR; })) return false;
    if (!enums(FD_ALTLOCUS, [&](uint64_t i, uint32_t v) { fl[i] |= (v & 3u) << VLR_F_ALTLOCUS_SHIFT; })) return false;
    auto bits = [&](int field, uint32_t flag) -> bool {  // bv::BitVec<u8>: Option tag, u64 blocks, bytes, u64 bits
        const auto& v = r.vec[field];
        Cursor cu{v.data(), v.data() + v.size()};
        if (!cu.u8()) { const uint64_t nb = cu.u64(); if (nb != n && !(nb == 0 && n == 0)) { err = std::string("length of ") + kFieldName[field]; return false; } return !cu.bad; }
        const uint64_t nblocks = cu.u64();
        if (!cu.need(nblocks)) { err = std::string("truncated ") + kFieldName[field]; return false; }
        const uint8_t* blocks = cu.p;
        cu.p += nblocks;
        const uint64_t nbits = cu.u64();
        if (cu.bad || nbits != n || nblocks * 8 < nbits) { err = std::string("length of ") + kFieldName[field]; return false; }
        for (uint64_t i = 0; i < n; ++i)
            if ((blocks[i >> 3] >> (i & 7)) & 1) fl[i] |= flag;
        return true;
    };
    if (!bits(FD_SOFTCLIPPED, VLR_F_SOFTCLIPPED) || !bits(FD_PAIRED, VLR_F_PAIRED) || !bits(FD_MAX_MAPQ, VLR_F_MAX_MAPQ)) return false;
    // homopolymer fields: present for the whole record or not at all (mod.rs:867: is_homopolymer_indel)
    bool is_hp = false;
    for (int k = 0; k < 2; ++k) {
        auto& col = c.col[7 + k];
        col.resize(base + n);
        float* dst = col.data() + base;
        const int field = k == 0 ? FD_HP_ART : FD_HP_VAR;
        if (!r.present[field]) { for (uint64_t i = 0; i < n; ++i) dst[i] = NAN; continue; }
        const auto& v = r.vec[field];
        Cursor cu{v.data(), v.data() + v.size()};
        const uint64_t m = cu.u64();
        if (m != n) { err = std::string("length of ") + kFieldName[field]; return false; }
        if (k == 0 && m > 0) is_hp = true;
        for (uint64_t i = 0; i < n; ++i) dst[i] = cu.u8() ? cu.mini() : NAN;
        if (cu.bad) { err = std::string("truncated ") + kFieldName[field]; return false; }
    }
    if (r.present[FD_HP_LEN]) {
        const auto& v = r.vec[FD_HP_LEN];
        Cursor cu{v.data(), v.data() + v.size()};
        if (cu.u64() != n) { err = "length of HOMOPOLYMER_INDEL_LEN"; return false; }
        for (uint64_t i = 0; i < n; ++i)
            if (cu.u8()) fl[i] |= VLR_F_HP_LEN_VALID | ((uint32_t)cu.u8() << VLR_F_HP_LEN_SHIFT);
        if (cu.bad) { err = "truncated HOMOPOLYMER_INDEL_LEN"; return false; }
    }
    c.third.resize(base + n);
    int32_t* th = c.third.data() + base;
    if (r.present[FD_THIRD]) {
        const auto& v = r.vec[FD_THIRD];
        Cursor cu{v.data(), v.data() + v.size()};
        if (cu.u64() != n) { err = "length of THIRD_ALLELE_EVIDENCE"; return false; }
        for (uint64_t i = 0; i < n; ++i) th[i] = cu.u8() ? (int32_t)cu.u32() : -1;
        if (cu.bad) { err = "truncated THIRD_ALLELE_EVIDENCE"; return false; }
    } else
        for (uint64_t i = 0; i < n; ++i) th[i] = -1;
    return push_cold(r, c, (uint32_t)n, is_hp, err);
}

// ---- BCF2 typed values
struct Typed { int type; uint32_t n; const uint8_t* data; };
bool bcf_typed(const uint8_t*& p, const uint8_t* e, Typed& t) {
    if (p >= e) return false;
    const uint8_t b = *p++;
    t.type = b & 0xf;
    t.n = b >> 4;
    if (t.n == 15) {
        Typed l;
        if (!bcf_typed(p, e, l) || l.n != 1) return false;
        if (l.type == 1) t.n = (uint32_t)(int8_t)l.data[0];
        else if (l.type == 2) { int16_t v; memcpy(&v, l.data, 2); t.n = (uint32_t)v; }
        else if (l.type == 3) { int32_t v; memcpy(&v, l.data, 4); t.n = (uint32_t)v; }
        else return false;
    }
    static const int size[16] = {0, 1, 2, 4, 0, 4, 0, 1, 0, 0, 0, 0, 0, 0, 0, 0};
    const size_t bytes = (size_t)t.n * size[t.type];
    if ((size_t)(e - p) < bytes) return false;
    t.data = p;
    p += bytes;
    return true;
}
int32_t typed_int(const Typed& t, uint32_t i) {
    if (t.type == 1) return (int8_t)t.data[i];
    if (t.type == 2) { int16_t v; memcpy(&v, t.data + 2 * i, 2); return v; }
    int32_t v; memcpy(&v, t.data + 4 * (size_t)i, 4); return v;
}
double phred_to_ln(float x) { return x != x ? NAN : -(double)x * std::log(10.0) / 10.0; }  // calling.rs:470-494

bool parse_bcf_record(const uint8_t* rec, const uint8_t* end, const std::vector<int8_t>& field_of_key, RecView& r, std::string& err) {
    if (end - rec < 32) { err = "truncated BCF record"; return false; }
    uint32_t l_shared;
    memcpy(&l_shared, rec, 4);
    const uint8_t* p = rec + 8;
    const uint8_t* se = p + l_shared;
    if (se > end) { err = "truncated BCF record"; return false; }
    int32_t chrom, pos;
    uint32_t nai;
    memcpy(&chrom, p, 4); memcpy(&pos, p + 4, 4); memcpy(&nai, p + 16, 4);
    p += 24;
    r.contig = chrom; r.pos = (int64_t)pos + 1;
    const uint32_t n_allele = nai >> 16, n_info = nai & 0xffff;
    Typed t;
    if (!bcf_typed(p, se, t)) { err = "bad ID"; return false; }
    r.id.assign((const char*)t.data, t.type == 7 ? t.n : 0);
    for (uint32_t a = 0; a < n_allele; ++a) {
        if (!bcf_typed(p, se, t) || (t.type != 7 && t.n != 0)) { err = "bad allele"; return false; }
        if (a == 0) r.ref.assign((const char*)t.data, t.n);
        else { if (a > 1) r.alt.push_back(','); r.alt.append((const char*)t.data, t.n); }
    }
    if (n_allele < 2) r.alt = ".";
    if (!bcf_typed(p, se, t)) { err = "bad FILTER"; return false; }
    for (uint32_t k = 0; k < n_info; ++k) {
        Typed key, val;
        if (!bcf_typed(p, se, key) || key.n != 1 || key.type < 1 || key.type > 3 || !bcf_typed(p, se, val)) { err = "bad INFO"; return false; }
        const int32_t ki = typed_int(key, 0);
        const int f = (ki >= 0 && (size_t)ki < field_of_key.size()) ? field_of_key[(size_t)ki] : -1;
        if (f < 0) continue;
        if (f < FD_N_VEC) {
            if (val.type < 1 || val.type > 3) continue;  // read_values: info(tag).integer()
            if (hot_field(f)) { r.wsrc[f] = val.data; r.wn[f] = val.n; r.wstride[f] = val.type == 3 ? 4 : val.type; r.present[f] = true; continue; }
            auto& out = r.vec[f];
            out.resize((size_t)val.n * 2);
            // i32 -> u16 -> 2 bytes LE (mod.rs:836-842); vector-end padding cannot occur inside an INFO vector
            if (val.type == 3) for (uint32_t i = 0; i < val.n; ++i) { out[2 * i] = val.data[4 * (size_t)i]; out[2 * i + 1] = val.data[4 * (size_t)i + 1]; }
            else if (val.type == 2) memcpy(out.data(), val.data, (size_t)val.n * 2);
            else for (uint32_t i = 0; i < val.n; ++i) { const int v = (int8_t)val.data[i]; out[2 * i] = (uint8_t)(v & 0xff); out[2 * i + 1] = (uint8_t)((v >> 8) & 0xff); }
            r.present[f] = true;
        } else if (f == FD_IMPRECISE) r.imprecise = true;
        else if (f == FD_EVENT && val.type == 7) r.event.assign((const char*)val.data, val.n);
        else if (f == FD_MATEID && val.type == 7) r.mateid.assign((const char*)val.data, val.n);
        else if ((f == FD_HET || f == FD_SOM) && val.type == 5 && val.n >= 1) {
            uint32_t bits;
            memcpy(&bits, val.data, 4);
            float x;
            memcpy(&x, &bits, 4);
            const bool missing = bits == 0x7F800001u || bits == 0x7F800002u;
            (f == FD_HET ? r.het_ln : r.som_ln) = missing ? NAN : phred_to_ln(x);
        }
    }
    while (!r.event.empty() && r.event.back() == '\0') r.event.pop_back();
    while (!r.mateid.empty() && r.mateid.back() == '\0') r.mateid.pop_back();
    return true;
}

bool parse_text_record(const char* line, const char* le, std::unordered_map<std::string, int>& contig_ids, std::mutex& contig_mu,
                       std::vector<std::string>& contig_names, RecView& r, std::string& err) {
    const char* f[9];
    const char* fe[9];
    int nf = 0;
    const char* p = line;
    while (nf < 9) {
        const char* t = (const char*)memchr(p, '\t', (size_t)(le - p));
        f[nf] = p; fe[nf] = t ? t : le;
        ++nf;
        if (!t) break;
        p = t + 1;
    }
    if (nf < 8) { err = "VCF line with fewer than 8 columns"; return false; }
    const std::string chrom(f[0], fe[0]);
    {
        std::lock_guard<std::mutex> g(contig_mu);
        auto it = contig_ids.find(chrom);
        if (it == contig_ids.end()) { it = contig_ids.emplace(chrom, (int)contig_names.size()).first; contig_names.push_back(chrom); }
        r.contig = it->second;
    }
    r.pos = strtoll(std::string(f[1], fe[1]).c_str(), nullptr, 10);
    r.id.assign(f[2], fe[2]); r.ref.assign(f[3], fe[3]); r.alt.assign(f[4], fe[4]);
    const char* q = f[7];
    const char* qe = fe[7];
    while (q < qe) {
        const char* semi = (const char*)memchr(q, ';', (size_t)(qe - q));
        const char* ke = semi ? semi : qe;
        const char* eq = (const char*)memchr(q, '=', (size_t)(ke - q));
        const size_t klen = (size_t)((eq ? eq : ke) - q);
        int fld = -1;
        for (int i = 0; i < FD_N; ++i)
            if (strlen(kFieldName[i]) == klen && memcmp(kFieldName[i], q, klen) == 0) { fld = i; break; }
        if (fld >= 0) {
            const char* v = eq ? eq + 1 : ke;
            if (fld < FD_N_VEC) {
                auto& out = r.vec[fld];
                while (v < ke) {
                    char* endp;
                    const long x = strtol(v, &endp, 10);
                    if (endp == v) break;
                    out.push_back((uint8_t)(x & 0xff));
                    out.push_back((uint8_t)((x >> 8) & 0xff));
                    v = endp;
                    if (v < ke && *v == ',') ++v;
                }
                r.present[fld] = true;
            } else if (fld == FD_IMPRECISE) r.imprecise = true;
            else if (fld == FD_EVENT) r.event.assign(v, ke);
            else if (fld == FD_MATEID) r.mateid.assign(v, ke);
            else {
                const std::string s(v, ke);
                const std::string first = s.substr(0, s.find(','));
                if (!first.empty() && first != ".") {
                    const float x = strtof(first.c_str(), nullptr);
                    (fld == FD_HET ? r.het_ln : r.som_ln) = phred_to_ln(x);
                }
            }
        }
        q = semi ? semi + 1 : qe;
    }
    return true;
}

// all records of one observation file, decoded by `n_threads` workers into chunks of consecutive records
struct SampleFile {
    std::vector<Chunk> chunks;
    std::vector<std::string> contig_names;
    int64_t n_rec = 0;
};

bool decode_records(const uint8_t* base, size_t size, const std::vector<size_t>& starts, bool is_bcf, const std::vector<int8_t>& field_of_key,
                    std::unordered_map<std::string, int>& contig_ids, std::mutex& contig_mu, const char* path, int64_t first_record_no, int n_threads,
                    SampleFile& sf, std::string& err);

bool read_sample_file(const char* path, int n_threads, SampleFile& sf, std::string& err) {
    Blob data;
    if (!load_inflated(path, data, n_threads, err)) return false;
    const double t_parse0 = now_s();
    const bool is_bcf = data.size() >= 9 && memcmp(data.data(), "BCF\2\2", 5) == 0;
    Header h;
    std::vector<size_t> starts;  // record starts (+ end sentinel)
    if (is_bcf) {
        uint32_t l_text;
        memcpy(&l_text, data.data() + 5, 4);
        if (9 + (size_t)l_text > data.size()) { err = std::string("truncated BCF header in ") + path; return false; }
        std::string text((const char*)data.data() + 9, l_text);
        while (!text.empty() && text.back() == '\0') text.pop_back();
        parse_header(text, h);
        size_t p = 9 + (size_t)l_text;
        while (p + 8 <= data.size()) {
            uint32_t ls, li;
            memcpy(&ls, data.data() + p, 4); memcpy(&li, data.data() + p + 4, 4);
            starts.push_back(p);
            p += 8 + (size_t)ls + li;
        }
        if (p != data.size()) { err = std::string("truncated BCF record in ") + path; return false; }
        starts.push_back(p);
        sf.contig_names = h.contigs;
    } else {
        size_t p = 0;
        std::string text;
        while (p < data.size()) {
            const uint8_t* nl = (const uint8_t*)memchr(data.data() + p, '\n', data.size() - p);
            const size_t e = nl ? (size_t)(nl - data.data()) : data.size();
            if (e > p && data[p] == '#') text.append((const char*)data.data() + p, e - p + 1);
            else if (e > p) starts.push_back(p);
            p = e + 1;
        }
        starts.push_back(data.size() + 1);
        parse_header(text, h);
    }
    if (!h.version_ok) { err = std::string("invalid observation format in ") + path + " (calling.rs:324-339: varlociraptor_observation_format_version=15 expected)"; return false; }
    std::vector<int8_t> field_of_key(h.dict.size(), -1);
    for (size_t i = 0; i < h.dict.size(); ++i)
        for (int f = 0; f < FD_N; ++f)
            if (h.dict[i] == kFieldName[f]) field_of_key[i] = (int8_t)f;
    std::unordered_map<std::string, int> contig_ids;
    std::mutex contig_mu;
    if (!decode_records(data.data(), data.size(), starts, is_bcf, field_of_key, contig_ids, contig_mu, path, 0, n_threads, sf, err)) return false;
    g_ingest_t[2] += now_s() - t_parse0;
    return true;
}

// records [starts[i], starts[i+1]) of `base` -> the chunks of sf, on n_threads workers over contiguous record ranges
bool decode_records(const uint8_t* base, size_t size, const std::vector<size_t>& starts, bool is_bcf, const std::vector<int8_t>& field_of_key,
                    std::unordered_map<std::string, int>& contig_ids, std::mutex& contig_mu, const char* path, int64_t first_record_no, int n_threads,
                    SampleFile& sf, std::string& err) {
    struct View { const uint8_t* p; size_t n; const uint8_t* data() const { return p; } size_t size() const { return n; } } data{base, size};
    const int64_t n = (int64_t)starts.size() - 1;
    sf.n_rec = n;
    const int T = (int)std::max<int64_t>(1, std::min<int64_t>(pick_threads(n_threads), (n + 255) / 256));
    sf.chunks = std::vector<Chunk>((size_t)T);
    parallel_ranges(n, T, [&](int64_t b, int64_t e, int t) {
        Chunk& c = sf.chunks[(size_t)t];
        {   // one allocation per column instead of doubling (a v15 observation takes 100-160 bytes of an uncompressed BCF record)
            const size_t est = is_bcf ? (starts[(size_t)e] - starts[(size_t)b]) / 100 + 1024 : 0;
            if (est) { for (int k = 0; k < 9; ++k) c.col[k].reserve(est); c.flags.reserve(est); c.third.reserve(est); }
            c.n_obs.reserve((size_t)(e - b));
        }
        RecView r;
        for (int64_t i = b; i < e && c.error.empty(); ++i) {
            r.reset();
            std::string er;
            bool ok;
            if (is_bcf) ok = parse_bcf_record(data.data() + starts[(size_t)i], data.data() + starts[(size_t)i + 1], field_of_key, r, er);
            else {
                const char* ls = (const char*)data.data() + starts[(size_t)i];
                const char* le = (const char*)memchr(ls, '\n', data.size() - starts[(size_t)i]);
                if (!le) le = (const char*)data.data() + data.size();
                if (le > ls && le[-1] == '\r') --le;
                ok = parse_text_record(ls, le, contig_ids, contig_mu, sf.contig_names, r, er);
            }
            ok = ok && decode_into(r, c, er);
            if (!ok) c.error = er + " (record " + std::to_string(first_record_no + i + 1) + " of " + path + ")";
        }
    });
    for (auto& c : sf.chunks)
        if (!c.error.empty()) { err = c.error; return false; }
    return true;
}

// _variant_class of obsfmt.py / calling.rs:517-534: (vlr_variant_type, is_snv_or_mnv, has_snv)
void variant_class(const char* ref, const char* alt, int& vt, bool& snv_or_mnv, bool& has_snv) {
    const size_t lr = strlen(ref), la = strlen(alt);
    has_snv = false;
    if (alt[0] == '<') {
        std::string t(alt);
        while (!t.empty() && (t.front() == '<')) t.erase(t.begin());
        while (!t.empty() && (t.back() == '>')) t.pop_back();
        vt = (t == "DEL" || t == "INS") ? VLR_VT_INDEL : (t == "INV" || t == "DUP" || t == "BND") ? VLR_VT_SV : VLR_VT_OTHER;
        snv_or_mnv = lr == la;  // calling.rs compares allele byte lengths
        return;
    }
    if (strchr(alt, '[') || strchr(alt, ']')) { vt = VLR_VT_SV; snv_or_mnv = lr == la; return; }
    if (lr == 1 && la == 1) { vt = VLR_VT_SNV; snv_or_mnv = true; has_snv = true; return; }
    if (lr == la) { vt = VLR_VT_MNV; snv_or_mnv = true; return; }
    vt = VLR_VT_INDEL; snv_or_mnv = false;
}

void* table_alloc(size_t bytes, bool& pinned) {
    if (bytes == 0) bytes = 8;
    // page-locking gigabytes costs more than the bounce copies it saves on a one-shot CLI run: opt-in (VLR_INGEST_PINNED=1)
    static const bool want_pinned = getenv("VLR_INGEST_PINNED") && atoi(getenv("VLR_INGEST_PINNED")) != 0;
    void* p = want_pinned ? vlr_host_alloc(bytes) : nullptr;
    pinned = p != nullptr;
    if (!p) {
        if (bytes >= ((size_t)8 << 20)) {
            void* q = nullptr;
            if (posix_memalign(&q, (size_t)2 << 20, (bytes + (((size_t)2 << 20) - 1)) & ~(((size_t)2 << 20) - 1)) == 0) { (void)madvise(q, bytes, MADV_HUGEPAGE); p = q; }
        }
        if (!p) p = aligned_alloc(64, (bytes + 63) & ~(size_t)63);
    }
    return p;
}

// column storage of the tables of a device reader: one device allocation and one page-locked host allocation of the same layout,
// recycled (hipMalloc / hipHostMalloc synchronise the device and take milliseconds per hundred megabytes)
struct DevSlab { void* d = nullptr; void* h = nullptr; size_t cap = 0; };
struct DevPool {
    int device = 0;
    std::mutex mu;
    std::vector<DevSlab> free_list;
    ~DevPool() { for (auto& s : free_list) vlr_dev_slab_free(device, s.d, s.h); }
    int acquire(size_t bytes, DevSlab& out) {
        {
            std::lock_guard<std::mutex> g(mu);
            int best = -1;
            for (int i = 0; i < (int)free_list.size(); ++i)
                if (free_list[(size_t)i].cap >= bytes && (best < 0 || free_list[(size_t)i].cap < free_list[(size_t)best].cap)) best = i;
            if (best >= 0) { out = free_list[(size_t)best]; free_list.erase(free_list.begin() + best); return VLR_OK; }
            if (free_list.size() >= 8) { vlr_dev_slab_free(device, free_list[0].d, free_list[0].h); free_list.erase(free_list.begin()); }  // (too small ones do not pile up)
        }
        const size_t cap = bytes + bytes / 8 + (1u << 20);
        const int rc = vlr_dev_slab_alloc(device, cap, &out.d, &out.h);
        if (rc != VLR_OK) return rc;
        out.cap = cap;
        return VLR_OK;
    }
    void release(DevSlab& s) {
        if (!s.d && !s.h) return;
        std::lock_guard<std::mutex> g(mu);
        free_list.push_back(s);
        s = DevSlab();
    }
};
// one pool per device for the whole process: page-locking a slab costs tens of milliseconds, and a caller that opens one reader per
// file (or per bench step) would pay it again for every table in flight
std::shared_ptr<DevPool> shared_dev_pool(int device) {
    static std::mutex mu;
    static auto* pools = new std::map<int, std::shared_ptr<DevPool>>();   // (never destroyed: no HIP calls during static destruction)
    std::lock_guard<std::mutex> g(mu);
    auto& p = (*pools)[device];
    if (!p) { p = std::make_shared<DevPool>(); p->device = device; }
    return p;
}
// byte offsets of the arrays of a table inside its slab (same on both sides)
struct DevLayout {
    size_t off_obs, off_col[9], off_flags, off_third, off_lflags, off_vt, off_ref, off_alt, bytes;
    static size_t up(size_t x) { return (x + 255) & ~(size_t)255; }
    DevLayout(int64_t L, int S, uint64_t total) {
        size_t at = 0;
        off_obs = at; at = up(at + ((size_t)(L * S) + 1) * 4);
        for (int k = 0; k < 9; ++k) { off_col[k] = at; at = up(at + (size_t)total * 4); }
        off_flags = at; at = up(at + (size_t)total * 4);
        off_third = at; at = up(at + (size_t)total * 4);
        off_lflags = at; at = up(at + (size_t)L);
        off_vt = at; at = up(at + (size_t)L);
        off_ref = at; at = up(at + (size_t)L);
        off_alt = at; at = up(at + (size_t)L);
        bytes = at;
    }
};

}  // namespace

// ================================================================================================ the observation table
struct vlr_obs_table {
    int32_t n_samples = 0;
    int64_t n_loci = 0, n_obs = 0;
    struct Arr { void* p = nullptr; bool pinned = false; };
    Arr a_off, a_col[9], a_flags, a_lflags, a_vt, a_ref, a_alt, a_third;
    uint32_t* obs_offset = nullptr;
    float* col[9] = {};
    uint32_t* flags = nullptr;
    int32_t* third = nullptr;
    uint8_t *locus_flags = nullptr, *variant_type = nullptr, *ref_base = nullptr, *alt_base = nullptr;
    std::vector<int32_t> contig;
    std::vector<int64_t> pos, hap_rep;
    std::vector<uint64_t> hap_key;  // 64-bit hash of the haplotype identifier of a grouped record (0: ungrouped): lets a driver that reads
                                    // the file in chunks recognise an event whose first record sat in an earlier chunk
    std::vector<double> het, som;
    std::vector<uint8_t> imprecise;
    std::vector<std::string> contig_names;
    std::vector<const char*> contig_ptrs;
    std::string pool;
    std::vector<uint64_t> id_off, ref_off, alt_off;
    // a table of the device reader: every array above points into one page-locked slab, `dev` holds the same layout in device
    // memory; both go back to the reader's pool when the table is freed
    std::shared_ptr<DevPool> dev_pool;
    DevSlab slab;
    vlr_batch dev_batch;
    bool has_dev = false;
    int device = 0;
    bool cols_on_host = true;      // false: the observation columns were not copied down (vlr_obs_table_fetch_columns does it on demand)
    void* cols_event = nullptr;    // the copy of the columns is still in flight (reader option async columns): wait_columns() before reading them
    std::mutex cols_mu;
    size_t col_region_off = 0, col_region_bytes = 0;   // the column arrays inside the slab (both sides)
    // observation summaries of the device reader (vlr::PileSum): what the calls writer needs per pileup instead of the columns.
    // They live in the host side of the column region until the columns are fetched over them.
    bool has_summary = false;
    const vlr::PileSum* sum_hdr = nullptr;
    const uint8_t* sum_text = nullptr;
    const float* sum_run_pm = nullptr;
    const uint32_t* sum_run_len = nullptr;
    int wait_columns() {
        std::lock_guard<std::mutex> g(cols_mu);
        if (!cols_event) return VLR_OK;
        const int rc = vlr_dev_event_wait(device, cols_event);
        vlr_dev_event_destroy(device, cols_event);
        cols_event = nullptr;
        return rc;
    }
    ~vlr_obs_table() {
        if (cols_event) vlr_dev_event_destroy(device, cols_event);   // (waits: the slab must not be recycled under a copy)
        if (dev_pool) dev_pool->release(slab);
        Arr* all[] = {&a_off, &a_col[0], &a_col[1], &a_col[2], &a_col[3], &a_col[4], &a_col[5], &a_col[6], &a_col[7], &a_col[8],
                      &a_flags, &a_lflags, &a_vt, &a_ref, &a_alt, &a_third};
        for (Arr* a : all)
            if (a->p) { if (a->pinned) vlr_host_free(a->p); else free(a->p); }
    }
    template <typename T>
    T* make(Arr& a, size_t n) { a.p = table_alloc(n * sizeof(T), a.pinned); return (T*)a.p; }
};

static int build_table(std::vector<SampleFile>& files, const char* const* paths, uint32_t omit_bias_mask, int n_threads, vlr_obs_table** out,
                       const DevLayout* dl = nullptr, std::shared_ptr<DevPool> pool = nullptr, DevSlab* slab = nullptr);

extern "C" {

int vlr_obs_read(int n_samples, const char* const* paths, uint32_t omit_bias_mask, int n_threads, vlr_obs_table** out) {
    if (!out || !paths || n_samples < 1 || n_samples > VLR_MAX_SAMPLES) return ifail(VLR_ERR_INVALID_ARGUMENT, "vlr_obs_read: bad argument");
    *out = nullptr;
    n_threads = pick_threads(n_threads);
    std::vector<SampleFile> files((size_t)n_samples);
    for (int i = 0; i < 16; ++i) g_ingest_t[i] = 0.0;
    const double t_all0 = now_s();
    {   // the sample files side by side, each with its share of the threads
        std::vector<std::string> errs((size_t)n_samples);
        std::vector<char> oks((size_t)n_samples, 0);
        std::vector<std::thread> th;
        const int per = std::max(1, n_threads / n_samples);
        for (int s = 0; s < n_samples; ++s)
            th.emplace_back([&, s] { oks[(size_t)s] = read_sample_file(paths[s], per, files[(size_t)s], errs[(size_t)s]) ? 1 : 0; });
        for (auto& x : th) x.join();
        for (int s = 0; s < n_samples; ++s)
            if (!oks[(size_t)s]) return ifail(VLR_ERR_INVALID_ARGUMENT, "%s", errs[(size_t)s].c_str());
    }
    g_ingest_t[3] = now_s() - t_all0;
    const int rc = build_table(files, paths, omit_bias_mask, n_threads, out);
    g_ingest_t[6] = now_s() - t_all0;
    return rc;
}
}  // extern "C"

// the decoded chunks of the sample files -> one table in the locus x sample order of vlr_batch
// (dl != nullptr: a table of the device reader — the columns are already in `slab` (device side decoded them, host side copied), only
// the per-locus arrays and the site data are filled here)
static int build_table(std::vector<SampleFile>& files, const char* const* paths, uint32_t omit_bias_mask, int n_threads, vlr_obs_table** out,
                       const DevLayout* dl, std::shared_ptr<DevPool> pool, DevSlab* slab) {
    const int n_samples = (int)files.size();
    const double t_all0 = now_s();
    const double t_merge0 = now_s();
    for (int s = 0; s < n_samples; ++s) {
        if (files[(size_t)s].n_rec != files[0].n_rec) return ifail(VLR_ERR_INVALID_ARGUMENT, "inconsistent observations: %s holds %lld records, %s %lld (calling.rs:369-371)",
                                                                    paths[s], (long long)files[(size_t)s].n_rec, paths[0], (long long)files[0].n_rec);
    }
    const int S = n_samples;
    const int64_t L = files[0].n_rec;
    std::unique_ptr<vlr_obs_table> t(new vlr_obs_table());
    t->n_samples = S; t->n_loci = L;
    // where record l of sample s lives: (chunk, index in chunk, observation base in chunk)
    struct Loc { const Chunk* c; uint32_t i; uint64_t base; };
    std::vector<Loc> loc((size_t)(L * S));
    for (int s = 0; s < S; ++s) {
        int64_t l = 0;
        for (const Chunk& c : files[(size_t)s].chunks) {
            uint64_t base = 0;
            for (uint32_t i = 0; i < c.n_obs.size(); ++i) { loc[(size_t)(l * S + s)] = {&c, i, base}; base += c.n_obs[i]; ++l; }
        }
    }
    if (dl) {
        t->dev_pool = pool; t->slab = *slab; *slab = DevSlab(); t->has_dev = true;
        uint8_t* h = (uint8_t*)t->slab.h;
        t->obs_offset = (uint32_t*)(h + dl->off_obs);
    } else t->obs_offset = t->make<uint32_t>(t->a_off, (size_t)(L * S + 1));
    uint64_t total = 0;
    for (int64_t p = 0; p < L * S; ++p) { t->obs_offset[p] = (uint32_t)total; total += loc[(size_t)p].c->n_obs[loc[(size_t)p].i]; }
    if (total > 0xffffffffull) return ifail(VLR_ERR_INVALID_ARGUMENT, "more than 2^32 observations in one table");
    t->obs_offset[L * S] = (uint32_t)total;
    t->n_obs = (int64_t)total;
    if (dl) {
        uint8_t* h = (uint8_t*)t->slab.h;
        uint8_t* d = (uint8_t*)t->slab.d;
        for (int k = 0; k < 9; ++k) t->col[k] = (float*)(h + dl->off_col[k]);
        t->flags = (uint32_t*)(h + dl->off_flags);
        t->third = (int32_t*)(h + dl->off_third);
        t->locus_flags = h + dl->off_lflags; t->variant_type = h + dl->off_vt; t->ref_base = h + dl->off_ref; t->alt_base = h + dl->off_alt;
        t->col_region_off = dl->off_col[0]; t->col_region_bytes = dl->off_lflags - dl->off_col[0];
        vlr_batch& b = t->dev_batch;
        memset(&b, 0, sizeof b);
        b.n_loci = L; b.n_samples = S; b.n_obs = (int64_t)total;
        b.obs_offset = (const uint32_t*)(d + dl->off_obs);
        b.prob_mapping = (const float*)(d + dl->off_col[0]); b.prob_alt = (const float*)(d + dl->off_col[1]); b.prob_ref = (const float*)(d + dl->off_col[2]);
        b.prob_missed_allele = (const float*)(d + dl->off_col[3]); b.prob_sample_alt = (const float*)(d + dl->off_col[4]);
        b.prob_double_overlap = (const float*)(d + dl->off_col[5]); b.prob_hit_base = (const float*)(d + dl->off_col[6]);
        b.prob_hp_artifact = (const float*)(d + dl->off_col[7]); b.prob_hp_variant = (const float*)(d + dl->off_col[8]);
        b.flags = (const uint32_t*)(d + dl->off_flags);
        b.locus_flags = d + dl->off_lflags; b.variant_type = d + dl->off_vt; b.ref_base = d + dl->off_ref; b.alt_base = d + dl->off_alt;
    } else {
        for (int k = 0; k < 9; ++k) t->col[k] = t->make<float>(t->a_col[k], (size_t)total);
        t->flags = t->make<uint32_t>(t->a_flags, (size_t)total);
        t->third = t->make<int32_t>(t->a_third, (size_t)total);
        t->locus_flags = t->make<uint8_t>(t->a_lflags, (size_t)L);
        t->variant_type = t->make<uint8_t>(t->a_vt, (size_t)L);
        t->ref_base = t->make<uint8_t>(t->a_ref, (size_t)L);
        t->alt_base = t->make<uint8_t>(t->a_alt, (size_t)L);
    }
    t->contig.resize((size_t)L); t->pos.resize((size_t)L); t->het.resize((size_t)L); t->som.resize((size_t)L); t->imprecise.resize((size_t)L);
    t->id_off.resize((size_t)L); t->ref_off.resize((size_t)L); t->alt_off.resize((size_t)L); t->hap_rep.resize((size_t)L); t->hap_key.resize((size_t)L);
    // contigs: names of sample 0's file; the other files must name the same contig at every record
    t->contig_names = files[0].contig_names;
    std::atomic<int64_t> bad_rec{-1};
    parallel_ranges(L, n_threads, [&](int64_t b, int64_t e, int) {
        for (int64_t l = b; l < e; ++l) {
            const Loc& a = loc[(size_t)(l * S)];
            const char* ref0 = a.c->pool.c_str() + a.c->ref[a.i];
            const char* alt0 = a.c->pool.c_str() + a.c->alt[a.i];
            bool any_hp = false;
            for (int s = 0; s < S; ++s) {
                const Loc& x = loc[(size_t)(l * S + s)];
                const uint32_t n = x.c->n_obs[x.i];
                const uint32_t dst = t->obs_offset[l * S + s];
                if (!dl) {
                    for (int k = 0; k < 9; ++k) memcpy(t->col[k] + dst, x.c->col[k].data() + x.base, (size_t)n * 4);
                    memcpy(t->flags + dst, x.c->flags.data() + x.base, (size_t)n * 4);
                    memcpy(t->third + dst, x.c->third.data() + x.base, (size_t)n * 4);
                }
                any_hp = any_hp || x.c->is_hp[x.i];
                if (s > 0) {  // calling.rs:379-390: same site in every sample
                    const std::string& n0 = files[0].contig_names.size() > (size_t)a.c->contig[a.i] && a.c->contig[a.i] >= 0 ? files[0].contig_names[(size_t)a.c->contig[a.i]] : std::string();
                    const auto& cn = files[(size_t)s].contig_names;
                    const std::string& ns = cn.size() > (size_t)x.c->contig[x.i] && x.c->contig[x.i] >= 0 ? cn[(size_t)x.c->contig[x.i]] : std::string();
                    if (n0 != ns || x.c->pos[x.i] != a.c->pos[a.i] || strcmp(x.c->pool.c_str() + x.c->ref[x.i], ref0) != 0 ||
                        strcmp(x.c->pool.c_str() + x.c->alt[x.i], alt0) != 0) {
                        int64_t exp = -1;
                        bad_rec.compare_exchange_strong(exp, l);
                    }
                }
            }
            int vt;
            bool snv_or_mnv, has_snv;
            variant_class(ref0, alt0, vt, snv_or_mnv, has_snv);
            const bool precise = !a.c->imprecise[a.i];
            // WorkItem.check_* (calling.rs:557-567), remove_nonstandard_alignments (590-598)
            unsigned m = 0;
            if (snv_or_mnv && precise && !(omit_bias_mask & VLR_BIAS_ORIENTATION)) m |= VLR_BIAS_ORIENTATION;
            if (precise && !(omit_bias_mask & VLR_BIAS_STRAND)) m |= VLR_BIAS_STRAND;
            if (snv_or_mnv && precise && !(omit_bias_mask & VLR_BIAS_POSITION)) m |= VLR_BIAS_POSITION;
            if (snv_or_mnv && precise && !(omit_bias_mask & VLR_BIAS_SOFTCLIP)) m |= VLR_BIAS_SOFTCLIP;
            if (any_hp && !(omit_bias_mask & VLR_BIAS_HOMOPOLYMER)) m |= VLR_BIAS_HOMOPOLYMER;
            if (!(omit_bias_mask & VLR_BIAS_ALTLOCUS)) m |= VLR_BIAS_ALTLOCUS;
            if (snv_or_mnv && !(omit_bias_mask & VLR_BIAS_ORIENTATION)) m |= VLR_LOCUS_REMOVE_NONSTANDARD;
            if (has_snv) m |= VLR_LOCUS_HAS_SNV;
            t->locus_flags[l] = (uint8_t)m;
            t->variant_type[l] = (uint8_t)vt;
            t->ref_base[l] = has_snv ? (uint8_t)ref0[0] : 0;
            t->alt_base[l] = has_snv ? (uint8_t)alt0[0] : 0;
            t->contig[(size_t)l] = a.c->contig[a.i];
            t->pos[(size_t)l] = a.c->pos[a.i];
            t->het[(size_t)l] = a.c->het[a.i];
            t->som[(size_t)l] = a.c->som[a.i];
            t->imprecise[(size_t)l] = a.c->imprecise[a.i];
        }
    });
    if (bad_rec >= 0) return ifail(VLR_ERR_INVALID_ARGUMENT, "inconsistent observations: record %lld differs between the sample files (calling.rs:379-390)", (long long)bad_rec + 1);
    g_ingest_t[4] = now_s() - t_merge0;
    const double t_str0 = now_s();
    // strings and breakend groups (sequential: first occurrence of a haplotype identifier is the representative)
    std::unordered_map<std::string, int64_t> first;
    {
        size_t need = 0;
        for (const Chunk& c : files[0].chunks) need += c.pool.size();
        t->pool.reserve(need + 16);
    }
    for (int64_t l = 0; l < L; ++l) {
        const Loc& a = loc[(size_t)(l * S)];
        const char* base = a.c->pool.c_str();
        t->id_off[(size_t)l] = t->pool.size(); t->pool.append(base + a.c->id[a.i]); t->pool.push_back('\0');
        t->ref_off[(size_t)l] = t->pool.size(); t->pool.append(base + a.c->ref[a.i]); t->pool.push_back('\0');
        t->alt_off[(size_t)l] = t->pool.size(); t->pool.append(base + a.c->alt[a.i]); t->pool.push_back('\0');
        int64_t rep = l;
        uint64_t hk = 0;
        if (a.c->hap[a.i] != 0xffffffffu) {
            auto ins = first.emplace(std::string(base + a.c->hap[a.i]), l);
            rep = ins.first->second;
            hk = 0xcbf29ce484222325ull;  // FNV-1a
            for (const char* q = base + a.c->hap[a.i]; *q; ++q) { hk ^= (unsigned char)*q; hk *= 0x100000001b3ull; }
            hk |= 1ull;
        }
        t->hap_rep[(size_t)l] = rep;
        t->hap_key[(size_t)l] = hk;
    }
    for (auto& n : t->contig_names) t->contig_ptrs.push_back(n.c_str());
    g_ingest_t[5] = now_s() - t_str0;
    (void)t_all0;
    *out = t.release();
    return VLR_OK;
}

extern "C" {

// measurement aid: seconds of the stages of the last vlr_obs_read — [0] file reads, [1] BGZF inflate, [2] record parse + decode
// (each summed over the sample files, which run side by side), [3] wall time of all files, [4] merge into the table, [5] strings and
// breakend groups, [6] total — and of the last vlr_calls_write — [8] record encoding, [9] BGZF deflate + file write, [10] total.
void vlr_ingest_last_timings(double* out16) { if (out16) for (int i = 0; i < 16; ++i) out16[i] = g_ingest_t[i]; }
// the same indices summed over all vlr_obs_reader_next / vlr_calls_writer_append calls since the last reset (the streaming front
// door makes dozens of calls per file: the last one — the empty chunk at the end — says nothing)
void vlr_ingest_total_timings(double* out16, int reset) {
    if (out16) for (int i = 0; i < 16; ++i) out16[i] = g_ingest_total[i];
    if (reset) for (int i = 0; i < 16; ++i) g_ingest_total[i] = 0.0;
}

void vlr_obs_table_free(vlr_obs_table* t) { delete t; }

int vlr_obs_table_batch(const vlr_obs_table* t, vlr_batch* b) {
    if (!t || !b) return ifail(VLR_ERR_INVALID_ARGUMENT, "null argument");
    memset(b, 0, sizeof *b);
    b->n_loci = t->n_loci; b->n_samples = t->n_samples; b->n_obs = t->n_obs;
    b->obs_offset = t->obs_offset;
    b->prob_mapping = t->col[0]; b->prob_alt = t->col[1]; b->prob_ref = t->col[2]; b->prob_missed_allele = t->col[3];
    b->prob_sample_alt = t->col[4]; b->prob_double_overlap = t->col[5]; b->prob_hit_base = t->col[6];
    b->prob_hp_artifact = t->col[7]; b->prob_hp_variant = t->col[8];
    b->flags = t->flags;
    b->locus_flags = t->locus_flags; b->variant_type = t->variant_type; b->ref_base = t->ref_base; b->alt_base = t->alt_base;
    return VLR_OK;
}

int vlr_obs_table_sites(const vlr_obs_table* t, vlr_obs_sites* s) {
    if (!t || !s) return ifail(VLR_ERR_INVALID_ARGUMENT, "null argument");
    s->n_loci = t->n_loci;
    s->n_contigs = (int32_t)t->contig_names.size();
    s->contig_names = t->contig_ptrs.data();
    s->contig = t->contig.data(); s->pos = t->pos.data();
    s->strings = t->pool.c_str();
    s->id_offset = t->id_off.data(); s->ref_offset = t->ref_off.data(); s->alt_offset = t->alt_off.data();
    s->group_representative = t->hap_rep.data();
    s->heterozygosity_ln = t->het.data(); s->somatic_effective_mutation_rate_ln = t->som.data();
    s->third_allele_evidence = t->third;
    s->imprecise = t->imprecise.data();
    s->group_key = t->hap_key.data();
    return VLR_OK;
}

}  // extern "C"


// ================================================================================================ streaming reader
// The same files, a bounded number of records at a time (bounded memory, and the caller can overlap the read of the next chunk
// with the evaluation and emission of the previous one).  Every sample file keeps its position: the BGZF block index, the next
// block, and the inflated bytes behind the last record it delivered.  All files advance by the same number of records per call.
namespace {
// std::vector storage from vlr_host_alloc (page-locked when a device is present)
template <typename T>
struct PinnedAlloc {
    using value_type = T;
    PinnedAlloc() = default;
    template <typename U> PinnedAlloc(const PinnedAlloc<U>&) {}
    T* allocate(size_t n) { void* p = vlr_host_alloc(n * sizeof(T)); if (!p) p = malloc(n * sizeof(T)), tag(p, false); else tag(p, true); if (!p) throw std::bad_alloc(); return (T*)p; }
    void deallocate(T* p, size_t) { if (was_pinned(p)) vlr_host_free(p); else free(p); }
    template <typename U> bool operator==(const PinnedAlloc<U>&) const { return true; }
    template <typename U> bool operator!=(const PinnedAlloc<U>&) const { return false; }
    static std::mutex& mu() { static std::mutex m; return m; }
    static std::unordered_set<void*>& set() { static std::unordered_set<void*> s; return s; }
    static void tag(void* p, bool pinned) { if (p && pinned) { std::lock_guard<std::mutex> g(mu()); set().insert(p); } }
    static bool was_pinned(void* p) { std::lock_guard<std::mutex> g(mu()); return set().erase(p) != 0; }
};

struct FileStream {
    std::string path;
    Blob raw;                          // the mapped file
    std::vector<BgzfBlock> blocks;     // BGZF members (empty: other containers — everything is inflated at open)
    size_t next_block = 0;
    RawVec<uint8_t> win;               // inflated, not yet delivered bytes (BGZF): uninitialised, 2 MB pages ...
    double bytes_per_record = 32768.0; // running estimate: how many blocks a request for max_records needs
    Blob whole;                        // ... or the whole inflated file
    size_t pos = 0;                    // first undelivered byte of win / whole
    bool use_whole = false, is_bcf = false, header_done = false;
    Header h;
    std::vector<int8_t> field_of_key;
    std::vector<std::string> contig_names;
    std::unordered_map<std::string, int> contig_ids;
    std::mutex contig_mu;
    int64_t delivered = 0;
    const uint8_t* base() const { return use_whole ? whole.p : win.p; }
    size_t avail() const { return use_whole ? whole.n : win.n; }
    bool more_blocks() const { return !use_whole && next_block < blocks.size(); }
};

bool stream_open(FileStream& f, const char* path, int n_threads, std::string& err) {
    f.path = path;
    if (!read_whole_file(path, f.raw, err)) return false;
    const bool gz = f.raw.size() >= 2 && f.raw[0] == 0x1f && f.raw[1] == 0x8b;
    if (!(gz && bgzf_index(f.raw, f.blocks))) {
        f.blocks.clear();
        f.use_whole = true;
        f.raw.release();
        if (!load_inflated(path, f.whole, n_threads, err)) return false;
    }
    return true;
}
// inflate the next `count` blocks behind the undelivered bytes of the window
bool stream_fill(FileStream& f, size_t count, int n_threads, std::string& err) {
    if (f.pos > 0) {  // drop what was delivered: the undelivered tail moves to the front
        const size_t keep = f.win.n - f.pos;
        if (keep) memmove(f.win.p, f.win.p + f.pos, keep);
        f.win.n = keep; f.pos = 0;
    }
    const size_t b0 = f.next_block, b1 = std::min(f.blocks.size(), b0 + count);
    size_t add = 0;
    std::vector<size_t> off(b1 - b0);
    for (size_t b = b0; b < b1; ++b) { off[b - b0] = add; add += f.blocks[b].isize; }
    const size_t old = f.win.n;
    f.win.resize(old + add);
    std::atomic<bool> bad{false};
    const double t_inf0 = now_s();
    parallel_items((int64_t)(b1 - b0), n_threads, [&](int64_t i, int) {
        const BgzfBlock& b = f.blocks[b0 + (size_t)i];
        if (b.isize && !inflate_raw(f.raw.p + b.off, b.clen, f.win.p + old + off[(size_t)i], b.isize, b.crc)) bad = true;
    });
    g_ingest_t[1] += now_s() - t_inf0;  // (summed over the sample files, which run side by side)
    f.next_block = b1;
    if (bad) { err = std::string("corrupt BGZF block in ") + f.path; return false; }
    return true;
}
// header of the stream (BCF: magic + text; text VCF: the leading '#' lines)
bool stream_header(FileStream& f, int n_threads, std::string& err) {
    for (;;) {
        const uint8_t* d = f.base();
        const size_t n = f.avail();
        if (n >= 9 && memcmp(d, "BCF\2\2", 5) == 0) {
            uint32_t l_text;
            memcpy(&l_text, d + 5, 4);
            if (9 + (size_t)l_text <= n) {
                std::string text((const char*)d + 9, l_text);
                while (!text.empty() && text.back() == '\0') text.pop_back();
                parse_header(text, f.h);
                f.is_bcf = true; f.pos = 9 + (size_t)l_text; f.contig_names = f.h.contigs;
                break;
            }
        } else if (n > 0 && d[0] == '#') {
            // all header lines must be in the window: the first line that does not start with '#' ends it
            size_t p = 0;
            std::string text;
            bool complete = false;
            while (p < n) {
                const uint8_t* nl = (const uint8_t*)memchr(d + p, '\n', n - p);
                if (!nl) break;
                const size_t e = (size_t)(nl - d);
                if (d[p] != '#') { complete = true; break; }
                text.append((const char*)d + p, e - p + 1);
                p = e + 1;
            }
            if (complete || !f.more_blocks()) { parse_header(text, f.h); f.is_bcf = false; f.pos = p; break; }
        } else if (n > 0 && !f.more_blocks()) { err = std::string("not a BCF or VCF file: ") + f.path; return false; }
        if (!f.more_blocks()) { if (n == 0) { parse_header("", f.h); break; } err = std::string("truncated header in ") + f.path; return false; }
        if (!stream_fill(f, 64, n_threads, err)) return false;
    }
    if (!f.h.version_ok) { err = std::string("invalid observation format in ") + f.path + " (calling.rs:324-339: varlociraptor_observation_format_version=15 expected)"; return false; }
    f.field_of_key.assign(f.h.dict.size(), -1);
    for (size_t i = 0; i < f.h.dict.size(); ++i)
        for (int k = 0; k < FD_N; ++k)
            if (f.h.dict[i] == kFieldName[k]) f.field_of_key[i] = (int8_t)k;
    f.header_done = true;
    return true;
}
// up to max_records records of the stream into sf (fewer only at the end of the file)
bool stream_next(FileStream& f, int64_t max_records, int n_threads, SampleFile& sf, std::string& err) {
    if (!f.header_done && !stream_header(f, n_threads, err)) return false;
    std::vector<size_t> starts;
    size_t scan = f.pos;
    for (;;) {
        const uint8_t* d = f.base();
        const size_t n = f.avail();
        if (f.is_bcf) {
            while ((int64_t)starts.size() < max_records && scan + 8 <= n) {
                uint32_t ls, li;
                memcpy(&ls, d + scan, 4); memcpy(&li, d + scan + 4, 4);
                const size_t e = scan + 8 + (size_t)ls + li;
                if (e > n) break;
                starts.push_back(scan);
                scan = e;
            }
        } else {
            while ((int64_t)starts.size() < max_records && scan < n) {
                const uint8_t* nl = (const uint8_t*)memchr(d + scan, '\n', n - scan);
                if (!nl && f.more_blocks()) break;
                const size_t e = nl ? (size_t)(nl - d) : n;
                if (e > scan && d[scan] != '#') starts.push_back(scan);
                scan = nl ? e + 1 : n;
            }
        }
        if ((int64_t)starts.size() >= max_records || !f.more_blocks()) break;
        // more input needed: as many blocks as the missing records are expected to take (one round for a typical request); the
        // window is compacted by stream_fill, so the offsets collected so far move with it
        const size_t shift = f.pos;
        const double missing = (double)(max_records - (int64_t)starts.size());
        const size_t blocks_needed = (size_t)std::min(1e9, std::max(64.0, missing * f.bytes_per_record * 1.05 / 65280.0 + 8.0));
        if (!stream_fill(f, blocks_needed, n_threads, err)) return false;
        for (auto& x : starts) x -= shift;
        scan -= shift;
    }
    if (f.is_bcf && !f.more_blocks() && (int64_t)starts.size() < max_records && scan != f.avail()) { err = std::string("truncated BCF record in ") + f.path; return false; }
    starts.push_back(scan);
    sf = SampleFile();
    sf.contig_names = f.contig_names;
    const bool ok = decode_records(f.base(), f.avail(), starts, f.is_bcf, f.field_of_key, f.contig_ids, f.contig_mu, f.path.c_str(), f.delivered, n_threads, sf, err);
    f.contig_names = sf.contig_names;
    f.delivered += sf.n_rec;
    if (sf.n_rec > 0) f.bytes_per_record = 0.5 * f.bytes_per_record + 0.5 * (double)(scan - f.pos) / (double)sf.n_rec;
    f.pos = scan;
    return ok;
}
}  // namespace

// Page-locked staging of a mapped file for the uploads of the device reader (round 5).  hipMemcpyAsync from the mapping is staged by the
// CALLING thread — 0.095 s of the reader's 0.14 s per 200 000 records went into that call —, so a helper thread copies the file, ahead of
// the reader and in file order, into a ring of page-locked segments; a feed uploads its byte range from the ring as two or three DMA
// copies and returns at once.  Segment k (file bytes [k, k + 1) * kSeg from `base`) lives in slot k mod kSlots; a slot is rewritten once
// everything below the segment that comes to it has been released (the reader releases what its waited-for feeds covered).
struct StageRing {
    // twenty segments of 8 MB (VLR_INGEST_STAGE_SEG_KB: other segment size — the tests run rings of 1.25 MB so that their small files
    // wrap around them many times)
    static size_t seg_bytes() { const char* e = getenv("VLR_INGEST_STAGE_SEG_KB"); const long k = e ? atol(e) : 0; return k >= 16 ? (size_t)k << 10 : (size_t)8 << 20; }
    const size_t kSeg = seg_bytes();
    static constexpr int kSlots = 20;
    const size_t kMaxRange = 6 * kSeg - 2;   // the largest byte range of one feed the ring serves (three of them fit beside each other)
    static size_t max_range() { return 6 * seg_bytes() - 2; }
    const uint8_t* src = nullptr;       // a mapped file ...
    int fd = -1;                        // ... or a descriptor: pread straight into the ring, the file is never mapped (see index below)
    size_t size = 0, base = 0;          // the file bytes, where staging starts
    uint8_t* ring = nullptr;
    // descriptor mode: the member chain (BSIZE fields) is walked in the ring as the bytes land, so that nobody touches the file's pages
    // a second time — mapping a 2 GB file and walking its 345 000 members cost 0.07 s at open and the unmap 0.06 s at close
    bool indexing = false, idx_done = false;
    size_t idx_pos = 0, idx_out = 0;
    std::vector<BgzfBlock> found;       // members indexed since the reader's last harvest (under mu)
    std::string idx_err;
    std::thread th;
    std::mutex mu;
    std::condition_variable cv;
    size_t hi = 0, released = 0;        // [base, hi) staged so far; bytes below `released` are no longer read
    bool stop = false;
    static std::mutex& pool_mu() { static std::mutex m; return m; }
    static std::vector<uint8_t*>& pool() { static auto* v = new std::vector<uint8_t*>(); return *v; }   // (page-locking 160 MB costs tens of milliseconds: rings outlive their reader)
    bool start(const uint8_t* file, size_t file_size, size_t from) {
        {
            std::lock_guard<std::mutex> g(pool_mu());
            if (!pool().empty() && kSeg == ((size_t)8 << 20)) { ring = pool().back(); pool().pop_back(); }
        }
        if (!ring) ring = (uint8_t*)vlr_host_alloc(kSeg * kSlots);
        if (!ring) return false;
        src = file; size = file_size; base = from - from % kSeg; hi = base; released = base;
        th = std::thread([this] { run(); });
        return true;
    }
    // descriptor mode: the whole file from offset 0, members indexed on the way (idx_from = offset of the first member to index)
    bool start_fd(int file, size_t file_size) {
        {
            std::lock_guard<std::mutex> g(pool_mu());
            if (!pool().empty() && kSeg == ((size_t)8 << 20)) { ring = pool().back(); pool().pop_back(); }
        }
        if (!ring) ring = (uint8_t*)vlr_host_alloc(kSeg * kSlots);
        if (!ring) return false;
        fd = file; size = file_size; base = 0; hi = 0; released = 0; indexing = true; idx_pos = 0; idx_out = 0;
        th = std::thread([this] { run(); });
        return true;
    }
    uint8_t byte_at(size_t off) const { return ring[((off - base) / kSeg % kSlots) * kSeg + (off - base) % kSeg]; }
    // members whose bytes are all staged (stager thread; `upto` = hi)
    void index_more(size_t upto) {
        std::vector<BgzfBlock> add;
        std::string err;
        bool done = false;
        size_t p = idx_pos, out = idx_out;
        for (;;) {
            if (p == size) { done = true; break; }
            if (p + 18 > upto) { if (upto == size) err = "truncated BGZF member"; break; }
            if (byte_at(p) != 0x1f || byte_at(p + 1) != 0x8b || byte_at(p + 2) != 8 || !(byte_at(p + 3) & 4)) { err = "not a BGZF member"; break; }
            const size_t xlen = (size_t)byte_at(p + 10) | ((size_t)byte_at(p + 11) << 8);
            size_t q = p + 12;
            const size_t xend = q + xlen;
            if (xend > upto) { if (upto == size) err = "truncated BGZF member"; break; }
            long bsize = -1;
            while (q + 4 <= xend) {
                const size_t slen = (size_t)byte_at(q + 2) | ((size_t)byte_at(q + 3) << 8);
                if (byte_at(q) == 'B' && byte_at(q + 1) == 'C' && slen == 2 && q + 6 <= xend) bsize = (long)((size_t)byte_at(q + 4) | ((size_t)byte_at(q + 5) << 8)) + 1;
                q += 4 + slen;
            }
            if (bsize < 0 || (size_t)bsize < xlen + 20) { err = "not a BGZF member"; break; }
            const size_t end = p + (size_t)bsize;
            if (end > size) { err = "truncated BGZF member"; break; }
            if (end > upto) break;
            const size_t cend = end - 8;
            const uint32_t crc = (uint32_t)byte_at(cend) | ((uint32_t)byte_at(cend + 1) << 8) | ((uint32_t)byte_at(cend + 2) << 16) | ((uint32_t)byte_at(cend + 3) << 24);
            const uint32_t isize = (uint32_t)byte_at(cend + 4) | ((uint32_t)byte_at(cend + 5) << 8) | ((uint32_t)byte_at(cend + 6) << 16) | ((uint32_t)byte_at(cend + 7) << 24);
            add.push_back({xend, cend - xend, isize, out, crc});
            out += isize;
            p = end;
        }
        std::lock_guard<std::mutex> lk(mu);
        found.insert(found.end(), add.begin(), add.end());
        idx_pos = p; idx_out = out;
        if (!err.empty()) { idx_err = err; idx_done = true; }
        if (done) idx_done = true;
    }
    void run() {
        for (;;) {
            size_t at;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return stop || (hi < size && hi + kSeg <= released - released % kSeg + kSeg * kSlots); });
                if (stop || hi >= size) return;
                at = hi;
            }
            const size_t n = std::min(kSeg, size - at);
            uint8_t* dst = ring + ((at - base) / kSeg % kSlots) * kSeg;
            if (fd >= 0) {
                size_t got = 0;
                while (got < n) {
                    const ssize_t r = pread(fd, dst + got, n - got, (off_t)(at + got));
                    if (r <= 0) break;
                    got += (size_t)r;
                }
                if (got < n) {   // (the file shrank or cannot be read: what is there is zero-filled, the index reports it)
                    memset(dst + got, 0, n - got);
                    std::lock_guard<std::mutex> lk(mu);
                    if (idx_err.empty()) idx_err = "read error";
                    idx_done = true;
                }
            } else memcpy(dst, src + at, n);
            if (indexing && !idx_done) index_more(at + n);
            {
                std::lock_guard<std::mutex> lk(mu);
                hi = at + n;
            }
            cv.notify_all();
        }
    }
    // the pieces of file bytes [a, b) in the ring (at most 7: see fits); waits for the stager
    int pieces(size_t a, size_t b, const uint8_t** ptr, size_t* len) {
        {
            std::unique_lock<std::mutex> lk(mu);
            cv.wait(lk, [&] { return hi >= b || stop; });
        }
        int n = 0;
        for (size_t p = a; p < b;) {
            const size_t k = (p - base) / kSeg, in = (p - base) % kSeg, take = std::min(kSeg - in, b - p);
            ptr[n] = ring + (k % kSlots) * kSeg + in; len[n] = take; ++n;
            p += take;
        }
        return n;
    }
    bool fits(size_t a, size_t b) const { return ring && a >= base && b > a && b - a <= kMaxRange; }
    // staging up to file offset b overwrites nothing at or above `released`
    bool can_stage(size_t b) {
        std::lock_guard<std::mutex> lk(mu);
        const size_t top = (b - base + kSeg - 1) / kSeg * kSeg + base;
        return top <= released - released % kSeg + kSeg * kSlots;
    }
    void release(size_t upto) {
        { std::lock_guard<std::mutex> lk(mu); if (upto > released) released = upto; }
        cv.notify_all();
    }
    void shutdown() {
        if (th.joinable()) {
            { std::lock_guard<std::mutex> lk(mu); stop = true; }
            cv.notify_all();
            th.join();
        }
        if (ring) {
            std::lock_guard<std::mutex> g(pool_mu());
            if (pool().size() < 8 && kSeg == ((size_t)8 << 20)) pool().push_back(ring);
            else vlr_host_free(ring);
            ring = nullptr;
        }
    }
    ~StageRing() { shutdown(); }
};

// one sample file of the device reader: the compressed file on the host, its inflated records on the device (vlr_decode.hip)
struct DevFileStream {
    std::string path;
    Blob raw;
    std::vector<BgzfBlock> blocks;
    size_t next_block = 0;
    vlr_dev_file* dev = nullptr;
    Header h;
    std::vector<int8_t> field_of_key;
    size_t header_bytes = 0;           // "BCF\2\2" + l_text + text: skipped behind the first members
    bool header_skipped = false;
    int n_hdr_samples = 0;
    double bytes_per_record = 12288.0;
    int64_t delivered = 0;
    std::vector<uint8_t, PinnedAlloc<uint8_t>> cold;   // cold records of the current chunk (page-locked: asynchronous D2H)
    std::vector<uint64_t> cold_off;
    std::vector<vlr::InflateBlock> ib;   // members of the feed in flight (read by the asynchronous copy)
    // ---- sharded reader (vlr_obs_reader_open_device_shard): this reader's window of the file and what its scan found
    size_t block_limit = (size_t)-1;   // members at and behind this index are not fed
    size_t w0 = 0, mb0 = 0, mb1 = 0;   // window start, own members [mb0, mb1)
    int64_t sh_lead = 0, sh_own = 0, sh_total = 0;   // complete records of the window: in front of the own share, starting in it, in all
    uint64_t sh_first[2] = {0, 0}, sh_land[2] = {0, 0};   // (member, byte in the member) of the first own record's start / of the start behind the last own record
    // page-locked staging of the uploads (plain readers; a sharded reader's window is one large feed from the mapping)
    std::unique_ptr<StageRing> stage;
    std::deque<size_t> fed_begin;       // file offsets where the feeds still in flight begin (what the ring must keep)
    // descriptor mode (plain readers of large files): the file is read through `fd` into the ring and never mapped; `blocks` grows as
    // the stager indexes members (harvest)
    int fd = -1;
    bool streaming = false, index_done = false;
    std::string index_err;
    // members the stager has indexed since the last call -> blocks; wait: until there is at least one more (or the index is complete)
    void harvest(bool wait) {
        if (!streaming || index_done) return;
        std::unique_lock<std::mutex> lk(stage->mu);
        if (wait) stage->cv.wait(lk, [&] { return !stage->found.empty() || stage->idx_done; });
        blocks.insert(blocks.end(), stage->found.begin(), stage->found.end());
        stage->found.clear();
        if (stage->idx_done) { index_done = true; index_err = stage->idx_err; }
    }
    bool more_blocks() const { return (next_block < blocks.size() && next_block < block_limit) || (streaming && !index_done); }
    ~DevFileStream() {
        if (dev) vlr_dev_file_destroy(dev);   // (waits for the uploads in flight: the ring goes after them)
        stage.reset();
        if (fd >= 0) close(fd);
    }
};

struct vlr_obs_reader {
    std::vector<std::unique_ptr<FileStream>> files;
    std::vector<std::string> paths;
    uint32_t omit = 0;
    int n_threads = 0;
    bool done = false;
    // device reader (vlr_obs_reader_open_device)
    bool on_device = false;
    int device = 0;
    std::vector<std::unique_ptr<DevFileStream>> dfiles;
    std::shared_ptr<DevPool> pool;
    bool host_columns = true;       // vlr_obs_reader_set_host_columns
    bool async_columns = false;     // vlr_obs_reader_set_async_columns: next() returns while the column copy to the host is in flight
    bool summaries_off = false;     // a chunk had pileups with more distinct observation keys than the summary kernel keeps: columns from then on
    // sharded reader: this reader delivers `remaining` records of its shard (set by vlr_obs_reader_shard_assign)
    bool sharded = false, assigned = false;
    int shard = 0, n_shards = 1;
    int64_t remaining = 0;
    DevSlab sum_scratch;            // device side only in use: summary headers, entries, runs, cursors of the chunk being built
    void* sum_copy_mark = nullptr;  // the detached copies of the last table's summaries: the next summary kernels wait for them (they reuse the scratch)
};

namespace { int dev_reader_next(vlr_obs_reader* r, int64_t max_records, vlr_obs_table** out); }

extern "C" {

int vlr_obs_reader_open(int n_samples, const char* const* paths, uint32_t omit_bias_mask, int n_threads, vlr_obs_reader** out) {
    if (!out || !paths || n_samples < 1 || n_samples > VLR_MAX_SAMPLES) return ifail(VLR_ERR_INVALID_ARGUMENT, "vlr_obs_reader_open: bad argument");
    *out = nullptr;
    std::unique_ptr<vlr_obs_reader> r(new vlr_obs_reader());
    r->omit = omit_bias_mask;
    r->n_threads = pick_threads(n_threads);
    for (int s = 0; s < n_samples; ++s) {
        r->paths.push_back(paths[s]);
        r->files.emplace_back(new FileStream());
        std::string err;
        if (!stream_open(*r->files.back(), paths[s], r->n_threads, err)) return ifail(VLR_ERR_INVALID_ARGUMENT, "%s", err.c_str());
    }
    *out = r.release();
    return VLR_OK;
}

// The next max_records records (fewer at the end) of every file as one table; *out = NULL once the files are exhausted.
int vlr_obs_reader_next(vlr_obs_reader* r, int64_t max_records, vlr_obs_table** out) {
    if (!r || !out || max_records < 1) return ifail(VLR_ERR_INVALID_ARGUMENT, "vlr_obs_reader_next: bad argument");
    *out = nullptr;
    if (r->done) return VLR_OK;
    if (r->on_device) return dev_reader_next(r, max_records, out);
    const int S = (int)r->files.size();
    for (int i = 0; i < 16; ++i) g_ingest_t[i] = 0.0;
    const double t0 = now_s();
    std::vector<SampleFile> files((size_t)S);
    std::vector<std::string> errs((size_t)S);
    std::vector<char> oks((size_t)S, 0);
    {
        std::vector<std::thread> th;
        const int per = std::max(1, r->n_threads / S);
        for (int s = 0; s < S; ++s)
            th.emplace_back([&, s] { oks[(size_t)s] = stream_next(*r->files[(size_t)s], max_records, per, files[(size_t)s], errs[(size_t)s]) ? 1 : 0; });
        for (auto& x : th) x.join();
    }
    for (int s = 0; s < S; ++s)
        if (!oks[(size_t)s]) return ifail(VLR_ERR_INVALID_ARGUMENT, "%s", errs[(size_t)s].c_str());
    g_ingest_t[3] = now_s() - t0;
    if (files[0].n_rec == 0) {
        for (int s = 1; s < S; ++s)
            if (files[(size_t)s].n_rec != 0) return ifail(VLR_ERR_INVALID_ARGUMENT, "inconsistent observations: %s holds more records than %s (calling.rs:369-371)", r->paths[(size_t)s].c_str(), r->paths[0].c_str());
        r->done = true;
        return VLR_OK;
    }
    std::vector<const char*> pp;
    for (auto& p : r->paths) pp.push_back(p.c_str());
    const int rc = build_table(files, pp.data(), r->omit, r->n_threads, out);
    g_ingest_t[6] = now_s() - t0;
    for (int i = 0; i < 7; ++i) g_ingest_total[i] += g_ingest_t[i];
    return rc;
}

// the summary scratch of a closed device reader waits for the next one (hipFree synchronises the device: milliseconds per run of the CLI);
// vlr_ingest_device_trim returns it
namespace {
std::mutex g_scratch_mu;
struct ParkedScratch { int device; DevSlab slab; };
std::vector<ParkedScratch>& parked_scratch() { static auto* v = new std::vector<ParkedScratch>(); return *v; }
}
void vlr_obs_reader_close(vlr_obs_reader* r) {
    if (r && r->sum_copy_mark) { vlr_dev_event_destroy(r->device, r->sum_copy_mark); r->sum_copy_mark = nullptr; }   // (waits: the scratch is parked or freed below)
    if (r && r->sum_scratch.d) {
        std::lock_guard<std::mutex> g(g_scratch_mu);
        auto& pk = parked_scratch();
        size_t k = 0;
        while (k < pk.size() && pk[k].device != r->device) ++k;
        if (k == pk.size()) pk.push_back({r->device, r->sum_scratch});
        else if (pk[k].slab.cap < r->sum_scratch.cap) { vlr_dev_slab_free(pk[k].device, pk[k].slab.d, pk[k].slab.h); pk[k].slab = r->sum_scratch; }
        else vlr_dev_slab_free(r->device, r->sum_scratch.d, r->sum_scratch.h);
        r->sum_scratch = DevSlab();
    }
    delete r;
}

}  // extern "C"

// ================================================================================================ device reader
// The same reader with inflate, record split and v15 decode on the device (vlr_inflate.hip, vlr_decode.hip).  This file keeps what
// is per-file and per-record bookkeeping: the BGZF member index, the BCF header (inflated here: kilobytes), the merge of the sample
// files into one table, the strings.
namespace {
double g_dev_t[16] = {0};

int count_header_samples(const std::string& text) {
    const size_t p = text.rfind("#CHROM");
    if (p == std::string::npos) return 0;
    size_t e = text.find('\n', p);
    if (e == std::string::npos) e = text.size();
    int tabs = 0;
    for (size_t i = p; i < e; ++i) tabs += text[i] == '\t';
    return tabs >= 9 ? tabs - 8 : 0;  // CHROM POS ID REF ALT QUAL FILTER INFO [FORMAT sample...]
}

int dev_stream_header(DevFileStream& f, const char* path, const std::vector<uint8_t>& head, size_t need, int device);

// Descriptor mode (plain readers of files above 32 MB, VLR_INGEST_STAGE=0: off): the file is opened, its first megabyte read for the BCF
// header, and everything else — bytes into the page-locked ring, member chain — is the stager's work beside the reader.  Returns
// VLR_ERR_UNSUPPORTED for what it does not take (small files, a header beyond the first megabyte): the caller maps the file then.
int dev_stream_open_fd(DevFileStream& f, const char* path, int device) {
    const int fd = open(path, O_RDONLY);
    if (fd < 0) return VLR_ERR_UNSUPPORTED;
    struct stat st;
    // (VLR_INGEST_STAGE_MIN_MB: files below it are mapped — default 32; the tests run their small files through this path with 0)
    size_t min_bytes = (size_t)32 << 20;
    if (const char* e = getenv("VLR_INGEST_STAGE_MIN_MB")) min_bytes = (size_t)std::max(0L, atol(e)) << 20;
    if (fstat(fd, &st) != 0 || st.st_size <= 0 || (size_t)st.st_size < min_bytes) { close(fd); return VLR_ERR_UNSUPPORTED; }
    std::vector<uint8_t> first(std::min<size_t>((size_t)1 << 20, (size_t)st.st_size));
    size_t got = 0;
    while (got < first.size()) { const ssize_t r = pread(fd, first.data() + got, first.size() - got, (off_t)got); if (r <= 0) break; got += (size_t)r; }
    if (got < first.size()) { close(fd); return VLR_ERR_UNSUPPORTED; }
    // the members of the first megabyte, inflated until the header is complete
    std::vector<uint8_t> head;
    size_t need = 9, p = 0;
    bool ok = true;
    while (head.size() < need && ok) {
        if (p + 18 > got || first[p] != 0x1f || first[p + 1] != 0x8b || first[p + 2] != 8 || !(first[p + 3] & 4)) { ok = false; break; }
        const size_t xlen = first[p + 10] | ((size_t)first[p + 11] << 8);
        size_t q = p + 12;
        const size_t xend = q + xlen;
        if (xend > got) { ok = false; break; }
        long bsize = -1;
        while (q + 4 <= xend) {
            const size_t slen = first[q + 2] | ((size_t)first[q + 3] << 8);
            if (first[q] == 'B' && first[q + 1] == 'C' && slen == 2 && q + 6 <= xend) bsize = (long)(first[q + 4] | ((size_t)first[q + 5] << 8)) + 1;
            q += 4 + slen;
        }
        if (bsize < 0 || (size_t)bsize < xlen + 20 || p + (size_t)bsize > got) { ok = false; break; }
        const size_t cend = p + (size_t)bsize - 8;
        uint32_t crc, isize;
        memcpy(&crc, first.data() + cend, 4); memcpy(&isize, first.data() + cend + 4, 4);
        const size_t old = head.size();
        head.resize(old + isize);
        if (isize && !inflate_raw(first.data() + xend, cend - xend, head.data() + old, isize, crc)) { close(fd); return ifail(VLR_ERR_INVALID_ARGUMENT, "corrupt BGZF block in %s", path); }
        if (head.size() >= 9 && need == 9) {
            if (memcmp(head.data(), "BCF\2\2", 5) != 0) { close(fd); return ifail(VLR_ERR_UNSUPPORTED, "device reader: %s is not a BCF2 file (use vlr_obs_reader_open)", path); }
            uint32_t l_text;
            memcpy(&l_text, head.data() + 5, 4);
            need = 9 + (size_t)l_text;
        }
        p += (size_t)bsize;
    }
    if (!ok || head.size() < need || need == 9) { close(fd); return VLR_ERR_UNSUPPORTED; }   // (not BGZF, or a header beyond the first megabyte: the mapped path decides)
    f.path = path;
    const int rc = dev_stream_header(f, path, head, need, device);
    if (rc != VLR_OK) { close(fd); return rc; }
    f.fd = fd;
    f.stage.reset(new StageRing());
    if (!f.stage->start_fd(fd, (size_t)st.st_size)) {
        f.stage.reset(); f.fd = -1; close(fd);
        vlr_dev_file_destroy(f.dev); f.dev = nullptr;
        return VLR_ERR_UNSUPPORTED;
    }
    f.streaming = true;
    return VLR_OK;
}

// the BCF header of a BGZF file, inflated on the host from the first members
int dev_stream_open(DevFileStream& f, const char* path, int device) {
    f.path = path;
    std::string err;
    if (!read_whole_file(path, f.raw, err)) return ifail(VLR_ERR_INVALID_ARGUMENT, "%s", err.c_str());
    const bool gz = f.raw.size() >= 2 && f.raw[0] == 0x1f && f.raw[1] == 0x8b;
    if (!gz || !bgzf_index(f.raw, f.blocks)) return ifail(VLR_ERR_UNSUPPORTED, "device reader: %s is not a BGZF file (use vlr_obs_reader_open)", path);
    std::vector<uint8_t> head;
    size_t b = 0;
    size_t need = 9;
    while (head.size() < need && b < f.blocks.size()) {
        const BgzfBlock& k = f.blocks[b++];
        const size_t old = head.size();
        head.resize(old + k.isize);
        if (k.isize && !inflate_raw(f.raw.p + k.off, k.clen, head.data() + old, k.isize, k.crc)) return ifail(VLR_ERR_INVALID_ARGUMENT, "corrupt BGZF block in %s", path);
        if (head.size() >= 9 && need == 9) {
            if (memcmp(head.data(), "BCF\2\2", 5) != 0) return ifail(VLR_ERR_UNSUPPORTED, "device reader: %s is not a BCF2 file (use vlr_obs_reader_open)", path);
            uint32_t l_text;
            memcpy(&l_text, head.data() + 5, 4);
            need = 9 + (size_t)l_text;
        }
    }
    if (head.size() < need || need == 9) return ifail(VLR_ERR_INVALID_ARGUMENT, "truncated header in %s", path);
    return dev_stream_header(f, path, head, need, device);
}

// the parsed header, the field table and the device side of a file whose inflated header bytes are `head`
int dev_stream_header(DevFileStream& f, const char* path, const std::vector<uint8_t>& head, size_t need, int device) {
    std::string text((const char*)head.data() + 9, need - 9);
    while (!text.empty() && text.back() == '\0') text.pop_back();
    parse_header(text, f.h);
    if (!f.h.version_ok) return ifail(VLR_ERR_INVALID_ARGUMENT, "invalid observation format in %s (calling.rs:324-339: varlociraptor_observation_format_version=15 expected)", path);
    f.header_bytes = need;
    f.n_hdr_samples = count_header_samples(text);
    f.field_of_key.assign(f.h.dict.size(), -1);
    for (size_t i = 0; i < f.h.dict.size(); ++i)
        for (int k = 0; k < FD_N; ++k)
            if (f.h.dict[i] == kFieldName[k]) f.field_of_key[i] = (int8_t)k;
    return vlr_dev_file_create(device, &f.dev);
}

// members up to `want` buffered bytes onto the device, in feeds of at most `piece` inflated bytes each (a feed is what the reader waits
// for: request-sized pieces let it take the oldest while the later ones still inflate)
int dev_stream_feed(DevFileStream& f, uint64_t want, uint64_t piece = ~0ull) {
    // (the header bytes in front of the first record are buffered like record bytes until they are skipped)
    const auto goal = [&] { return want + (f.header_skipped ? 0 : (uint64_t)f.header_bytes); };
    while (f.more_blocks() && vlr_dev_file_buffered(f.dev) < goal()) {
        f.harvest(false);
        const uint64_t have = vlr_dev_file_buffered(f.dev);
        const size_t b0 = f.next_block;
        size_t b1 = b0;
        uint64_t add = 0;
        f.ib.clear();
        for (;;) {
            while (b1 < f.blocks.size() && b1 < f.block_limit && ((have + add < goal() && add < piece) || b1 == b0) && b1 - b0 < (1u << 20)) {
                const BgzfBlock& k = f.blocks[b1];
                // (a feed from the staging ring is one byte range the ring holds at once)
                if (f.stage && b1 > b0 && (k.off + k.clen) - f.blocks[b0].off > StageRing::max_range()) break;
                vlr::InflateBlock x;
                x.src = k.off - f.blocks[b0].off; x.dst = add; x.clen = (uint32_t)k.clen; x.isize = k.isize; x.crc = k.crc; x.pad = 0;
                f.ib.push_back(x);
                add += k.isize;
                ++b1;
            }
            // descriptor mode: the members behind the indexed ones.  With members in hand the feed goes with them (the next one takes the
            // rest); with none the stager has to get on: it may be waiting for room in the ring, which only completed feeds give back
            if (f.streaming && !f.index_done && b1 == f.blocks.size() && b1 == b0) {
                f.harvest(false);
                if (b1 < f.blocks.size() || f.index_done) continue;
                if (vlr_dev_file_feeds_in_flight(f.dev) > 0) { const int rcw = vlr_dev_file_feed_wait_oldest(f.dev); if (rcw != VLR_OK) return rcw; }
                while ((int)f.fed_begin.size() > vlr_dev_file_feeds_in_flight(f.dev)) f.fed_begin.pop_front();
                f.stage->release(!f.fed_begin.empty() ? f.fed_begin.front() : f.blocks.empty() ? 0 : f.blocks.back().off + f.blocks.back().clen);
                if (vlr_dev_file_feeds_in_flight(f.dev) == 0) f.harvest(true);
                continue;
            }
            break;
        }
        if (f.streaming && f.index_done && !f.index_err.empty() && b1 == f.blocks.size())
            return ifail(VLR_ERR_INVALID_ARGUMENT, "corrupt BGZF block in %s (%s)", f.path.c_str(), f.index_err.c_str());
        if (b1 == b0) break;   // (the index ended: nothing left to feed)
        // one contiguous piece of the file: from the first member's DEFLATE payload to the end of the last one's
        const uint8_t* comp = f.raw.p ? f.raw.p + f.blocks[b0].off : nullptr;
        const size_t comp_bytes = (f.blocks[b1 - 1].off + f.blocks[b1 - 1].clen) - f.blocks[b0].off;
        const double t_up = now_s();
        int rc;
        const size_t fa = f.blocks[b0].off, fb = fa + comp_bytes;
        if (f.stage && f.stage->fits(fa, fb)) {
            // what the feeds that are no longer in flight covered may be overwritten in the ring
            while ((int)f.fed_begin.size() > vlr_dev_file_feeds_in_flight(f.dev)) f.fed_begin.pop_front();
            f.stage->release(f.fed_begin.empty() ? fa : f.fed_begin.front());
            // the ring holds three ranges: with more feeds in flight (a request of several feeds) the oldest ones are waited for until the
            // stager may write up to the end of this one — the reader would otherwise wait for bytes the stager has no room for
            while (!f.stage->can_stage(fb) && vlr_dev_file_feeds_in_flight(f.dev) > 0) {
                const int rcw = vlr_dev_file_feed_wait_oldest(f.dev);
                if (rcw != VLR_OK) return rcw;
                while ((int)f.fed_begin.size() > vlr_dev_file_feeds_in_flight(f.dev)) f.fed_begin.pop_front();
                f.stage->release(f.fed_begin.empty() ? fa : f.fed_begin.front());
            }
            const uint8_t* pp[8]; size_t pl[8];
            const int np = f.stage->pieces(fa, fb, pp, pl);
            rc = vlr_dev_file_feed_pieces(f.dev, pp, pl, np, f.ib.data(), (int)f.ib.size(), add);
            f.fed_begin.push_back(fa);
        } else if (f.streaming) {
            return ifail(VLR_ERR_INVALID_ARGUMENT, "device reader: a member range of %s does not fit the staging ring", f.path.c_str());   // (cannot happen: feeds are cut to the ring)
        } else {
            if (f.stage) { (void)vlr_dev_file_feed_wait(f.dev); f.stage.reset(); f.fed_begin.clear(); }   // (a range the ring does not hold: from the mapping from here on)
            rc = vlr_dev_file_feed(f.dev, comp, comp_bytes, f.ib.data(), (int)f.ib.size(), add);
        }
        g_dev_t[14] += now_s() - t_up;   // (inside the enqueueing call: an upload from pageable memory is staged by the calling thread)
        g_dev_t[9] += (double)add; g_dev_t[10] += (double)comp_bytes;
        if (rc != VLR_OK) return rc;
        f.next_block = b1;
        if (!f.header_skipped && vlr_dev_file_buffered(f.dev) >= f.header_bytes) {
            { const int rcw = vlr_dev_file_feed_wait(f.dev); if (rcw != VLR_OK) return rcw; }   // (once per file: the header's members are checked before they are skipped)
            const int rs = vlr_dev_file_skip(f.dev, f.header_bytes);
            if (rs != VLR_OK) return rs;
            f.header_skipped = true;
        }
    }
    if (!f.header_skipped) return ifail(VLR_ERR_INVALID_ARGUMENT, "truncated header in %s", f.path.c_str());
    return VLR_OK;
}

const char* rec_status_text(uint32_t st) {
    if (st & vlr::REC_TRUNCATED) return "truncated BCF record";
    if (st & vlr::REC_BAD_ID) return "bad ID";
    if (st & vlr::REC_BAD_ALLELE) return "bad allele";
    if (st & vlr::REC_BAD_FILTER) return "bad FILTER";
    if (st & vlr::REC_BAD_INFO) return "bad INFO";
    if (st & vlr::REC_MISSING_FIELD) return "No varlociraptor observations found in record";
    if (st & vlr::REC_BAD_LENGTHS) return "inconsistent observation vector lengths";
    return "truncated observation vector";
}

int dev_reader_next(vlr_obs_reader* r, int64_t max_records, vlr_obs_table** out) {
    const int S = (int)r->dfiles.size();
    if (r->sharded) {
        if (!r->assigned) return ifail(VLR_ERR_INVALID_ARGUMENT, "sharded reader: vlr_obs_reader_shard_assign comes before the first vlr_obs_reader_next");
        if (r->remaining <= 0) { r->done = true; return VLR_OK; }
        max_records = std::min(max_records, r->remaining);
    }
    max_records = std::min<int64_t>(max_records, (int64_t)1 << 22);   // (per-request device buffers are sized by it; the caller just asks again)
    const double t_all0 = now_s();
    std::vector<int64_t> n_rec((size_t)S, 0);
    std::vector<const vlr::RecHost*> rh((size_t)S, nullptr);
    int64_t n = 0;
    double scale = 1.0;
    // requests ahead of the one being delivered whose members are uploaded and inflating (VLR_INGEST_PREFETCH; the reader waits for the
    // OLDEST feed only).  Default 1: with two, the inflate waves of two requests sit on the CUs the decode and the evaluation of the current
    // one want (stream priorities order dispatch, they do not preempt) — 838 k against 928 k records/s end to end when the enqueueing call still waited for the inflate; level with it since.
    static const int depth = [] { const char* e = getenv("VLR_INGEST_PREFETCH"); const int v = e ? atoi(e) : 1; return v < 1 ? 1 : v > 2 ? 2 : v; }();   // (the staging ring holds three feed ranges: two in flight and the one being built)
    auto split_all = [&](int64_t& n_min) -> int {
        n_min = max_records;
        for (int s = 0; s < S; ++s) {
            DevFileStream& f = *r->dfiles[(size_t)s];
            int serial = 0;
            const int rc = vlr_dev_file_split(f.dev, max_records, (int)f.h.contigs.size(), f.n_hdr_samples, f.field_of_key.data(), (int)f.field_of_key.size(), &n_rec[(size_t)s], &rh[(size_t)s], &serial);
            if (rc != VLR_OK) return rc;
            if (serial) g_dev_t[12] += 1.0;
            n_min = std::min(n_min, n_rec[(size_t)s]);
        }
        return VLR_OK;
    };
    for (int round = 0;; ++round) {
        for (int s = 0; s < S; ++s) {
            DevFileStream& f = *r->dfiles[(size_t)s];
            const uint64_t want = (uint64_t)((double)max_records * f.bytes_per_record * 1.04 * scale) + 131072;
            const int rc = dev_stream_feed(f, want * (uint64_t)depth, want);
            if (rc != VLR_OK) return rc;
        }
        {
            const double tw = now_s();
            for (int s = 0; s < S; ++s) {
                DevFileStream& f = *r->dfiles[(size_t)s];
                const uint64_t want = (uint64_t)((double)max_records * f.bytes_per_record * 1.04 * scale) + 131072;
                const int rc = round == 0 ? vlr_dev_file_wait_ready(f.dev, want) : vlr_dev_file_feed_wait(f.dev);
                if (rc != VLR_OK) return rc;
            }
            g_dev_t[2] += now_s() - tw;
        }
        const double t0 = now_s();
        { const int rc = split_all(n); if (rc != VLR_OK) return rc; }
        if (n == 0) {   // nothing complete in the inflated bytes, but feeds are still in flight: all of them, then again
            bool in_flight = false;
            for (int s = 0; s < S; ++s) in_flight = in_flight || vlr_dev_file_feeds_in_flight(r->dfiles[(size_t)s]->dev) > 0;
            if (in_flight) {
                for (int s = 0; s < S; ++s) { const int rc = vlr_dev_file_feed_wait(r->dfiles[(size_t)s]->dev); if (rc != VLR_OK) return rc; }
                const int rc = split_all(n);
                if (rc != VLR_OK) return rc;
            }
        }
        g_dev_t[3] += now_s() - t0;
        if (n > 0) break;
        bool any_more = false;
        for (int s = 0; s < S; ++s) any_more = any_more || r->dfiles[(size_t)s]->more_blocks();
        if (!any_more) {
            if (r->sharded) return ifail(VLR_ERR_INVALID_ARGUMENT, "sharded reader: the window of %s ends before the shard's records do", r->paths[0].c_str());
            for (int s = 0; s < S; ++s) {
                if (n_rec[(size_t)s] > 0) return ifail(VLR_ERR_INVALID_ARGUMENT, "inconsistent observations: %s holds more records than the other files (calling.rs:369-371)", r->paths[(size_t)s].c_str());
                if (vlr_dev_file_buffered(r->dfiles[(size_t)s]->dev) > 0) return ifail(VLR_ERR_INVALID_ARGUMENT, "truncated BCF record in %s", r->paths[(size_t)s].c_str());
            }
            r->done = true;
            return VLR_OK;
        }
        scale *= 2.0;   // a record larger than the request's estimate: buffer more
        if (round > 40) return ifail(VLR_ERR_INVALID_ARGUMENT, "device reader: record larger than the device buffer in %s", r->paths[0].c_str());
    }
    const int64_t L = n;
    // records with scan errors
    for (int s = 0; s < S; ++s)
        for (int64_t i = 0; i < L; ++i)
            if (rh[(size_t)s][i].status) return ifail(VLR_ERR_INVALID_ARGUMENT, "%s (record %lld of %s)", rec_status_text(rh[(size_t)s][i].status), (long long)(r->dfiles[(size_t)s]->delivered + i + 1), r->paths[(size_t)s].c_str());
    // merged observation offsets: pileup p = locus * S + sample
    uint64_t total = 0;
    std::vector<uint32_t> obs_offset((size_t)(L * S) + 1);
    for (int64_t l = 0; l < L; ++l)
        for (int s = 0; s < S; ++s) { obs_offset[(size_t)(l * S + s)] = (uint32_t)total; total += rh[(size_t)s][l].n_obs; }
    if (total > 0xffffffffull) return ifail(VLR_ERR_INVALID_ARGUMENT, "more than 2^32 observations in one table");
    obs_offset[(size_t)(L * S)] = (uint32_t)total;
    const DevLayout dl(L, S, total);
    DevSlab slab;
    {
        const int rc = r->pool->acquire(dl.bytes, slab);
        if (rc != VLR_OK) return rc;
    }
    uint8_t* d = (uint8_t*)slab.d;
    uint8_t* h = (uint8_t*)slab.h;
    void* cols_event = nullptr;   // (detached copy of the columns, async_columns)
    auto drop_event = [&] { if (cols_event) { vlr_dev_event_destroy(r->device, cols_event); cols_event = nullptr; } };
    auto fail = [&](int rc) { drop_event(); r->pool->release(slab); return rc; };
    const double t_dec0 = now_s();
    {   // offsets up (every decode kernel reads them), then the files side by side on their own streams
        DevFileStream& f0 = *r->dfiles[0];
        int rc = vlr_dev_file_copy(f0.dev, d + dl.off_obs, obs_offset.data(), obs_offset.size() * 4, 1);
        if (rc == VLR_OK) rc = vlr_dev_file_sync(f0.dev);
        if (rc != VLR_OK) return fail(rc);
    }
    g_dev_t[0] += now_s() - t_dec0;   // (decode_offsets_up)
    const double t_launch0 = now_s();
    vlr::DeviceCols cols;
    for (int k = 0; k < 9; ++k) cols.col[k] = (float*)(d + dl.off_col[k]);
    cols.flags = (uint32_t*)(d + dl.off_flags);
    cols.third = (int32_t*)(d + dl.off_third);
    for (int s = 0; s < S; ++s) {
        DevFileStream& f = *r->dfiles[(size_t)s];
        int rc = vlr_dev_file_decode(f.dev, L, (const uint32_t*)(d + dl.off_obs), S, s, &cols);
        if (rc != VLR_OK) return fail(rc);
        f.cold_off.resize((size_t)L + 1);
        uint64_t at = 0;
        for (int64_t i = 0; i < L; ++i) { f.cold_off[(size_t)i] = at; at += rh[(size_t)s][i].cold_bytes; }
        f.cold_off[(size_t)L] = at;
        f.cold.resize((size_t)at + 64);
        rc = vlr_dev_file_cold(f.dev, L, f.cold_off.data(), f.cold.data());
        if (rc != VLR_OK) return fail(rc);
    }
    g_dev_t[1] += now_s() - t_launch0;   // (decode_launch: decode + cold kernels and copies enqueued)
    // the per-record flags the table needs from the scan (before the decode's error pass overwrites the host copy)
    std::vector<std::vector<uint32_t>> n_obs_of((size_t)S), flags_of((size_t)S);
    for (int s = 0; s < S; ++s) {
        n_obs_of[(size_t)s].resize((size_t)L); flags_of[(size_t)s].resize((size_t)L);
        for (int64_t i = 0; i < L; ++i) { n_obs_of[(size_t)s][(size_t)i] = rh[(size_t)s][i].n_obs; flags_of[(size_t)s][(size_t)i] = rh[(size_t)s][i].flags; }
    }
    // the records are consumed (the kernels above hold their own pointers) and the members of the NEXT request go up and inflate on
    // the feed streams, beside the decode kernels, the copies and the host work below — and beside the caller's evaluation of this chunk
    for (int s = 0; s < S; ++s) {
        DevFileStream& f = *r->dfiles[(size_t)s];
        const uint64_t before = vlr_dev_file_buffered(f.dev);
        const int rc = vlr_dev_file_consume(f.dev, L);
        if (rc != VLR_OK) return fail(rc);
        const double used = (double)(before - vlr_dev_file_buffered(f.dev)) / (double)L;
        f.bytes_per_record = 0.5 * f.bytes_per_record + 0.5 * used;
    }
    {
        const double tf = now_s();
        for (int s = 0; s < S; ++s) {
            DevFileStream& f = *r->dfiles[(size_t)s];
            const uint64_t want = (uint64_t)((double)max_records * f.bytes_per_record * 1.04) + 131072;
            const int rc = dev_stream_feed(f, want * (uint64_t)depth, want);
            if (rc != VLR_OK) return fail(rc);
        }
        g_dev_t[2] += now_s() - tf;
    }
    const double t_err0 = now_s();
    for (int s = 0; s < S; ++s) {
        uint32_t st = 0;
        int64_t bad = -1;
        const int rc = vlr_dev_file_errors(r->dfiles[(size_t)s]->dev, L, &st, &bad);   // (waits for the file's stream)
        if (rc != VLR_OK) return fail(rc);
        if (st) return fail(ifail(VLR_ERR_INVALID_ARGUMENT, "%s (record %lld of %s)", rec_status_text(st), (long long)(r->dfiles[(size_t)s]->delivered + bad + 1), r->paths[(size_t)s].c_str()));
    }
    g_dev_t[15] += now_s() - t_err0;   // (decode_wait: the decode and cold kernels of both files)
    g_dev_t[5] += now_s() - t_dec0;
    const double t_d2h0 = now_s();
    // the per-pileup summaries — P headers, the OBS text, at most one run per observation — are brought down INTO the host side of the
    // column region (44 bytes per observation), so they must fit there: the text buffer of the kernel is what the region leaves behind
    // the headers and the runs (an OBS item is 12 to 23 bytes per DISTINCT observation key; a pileup whose text does not fit any more is
    // marked and the table takes the columns after all)
    const int64_t P = L * S;
    const auto up64 = [](size_t x) { return (x + 63) & ~(size_t)63; };
    const size_t sum_fixed = up64((size_t)P * sizeof(vlr::PileSum)) + 2 * up64((size_t)total * 4) + 64;
    const size_t col_region = dl.off_lflags - dl.off_col[0];
    const size_t text_cap = col_region > sum_fixed ? std::min<size_t>((col_region - sum_fixed) & ~(size_t)63, (size_t)0xffffff00u) : 0;
    const bool summaries = !r->host_columns && !r->summaries_off && text_cap >= (size_t)total * 8;
    if (!summaries) {   // columns down for the calls writer (one copy: the column arrays are contiguous in the slab)
        DevFileStream& f0 = *r->dfiles[0];
        const int rc = r->async_columns ? vlr_dev_file_copy_detached(f0.dev, h + dl.off_col[0], d + dl.off_col[0], dl.off_lflags - dl.off_col[0], &cols_event)
                                        : vlr_dev_file_copy(f0.dev, h + dl.off_col[0], d + dl.off_col[0], dl.off_lflags - dl.off_col[0], 0);
        if (rc != VLR_OK) return fail(rc);
    }
    // ---- host side, while the columns come down: the cold records -> chunks -> table
    const double t_host0 = now_s();
    std::vector<SampleFile> files((size_t)S);
    for (int s = 0; s < S; ++s) {
        DevFileStream& f = *r->dfiles[(size_t)s];
        SampleFile& sf = files[(size_t)s];
        sf.n_rec = L;
        sf.contig_names = f.h.contigs;
        const int T = (int)std::max<int64_t>(1, std::min<int64_t>(r->n_threads, (L + 1023) / 1024));
        sf.chunks = std::vector<Chunk>((size_t)T);
        parallel_ranges(L, T, [&](int64_t b, int64_t e, int t) {
            Chunk& c = sf.chunks[(size_t)t];
            RecView rv;
            for (int64_t i = b; i < e && c.error.empty(); ++i) {
                rv.reset();
                std::string er;
                const uint8_t* p = f.cold.data() + f.cold_off[(size_t)i];
                const uint8_t* pe = f.cold.data() + f.cold_off[(size_t)i + 1];
                bool ok = parse_bcf_record(p, pe, f.field_of_key, rv, er);
                ok = ok && push_cold(rv, c, n_obs_of[(size_t)s][(size_t)i], (flags_of[(size_t)s][(size_t)i] & 1u) != 0, er);
                if (!ok) c.error = er + " (record " + std::to_string(f.delivered + i + 1) + " of " + f.path + ")";
            }
        });
        for (auto& c : sf.chunks)
            if (!c.error.empty()) return fail(ifail(VLR_ERR_INVALID_ARGUMENT, "%s", c.error.c_str()));
    }
    std::vector<const char*> pp;
    for (auto& p : r->paths) pp.push_back(p.c_str());
    // (build_table writes obs_offset and the per-locus arrays into the host side of the slab)
    int rc = build_table(files, pp.data(), r->omit, r->n_threads, out, &dl, r->pool, &slab);
    if (rc != VLR_OK) { drop_event(); if (slab.d) r->pool->release(slab); return rc; }
    (*out)->device = r->device;
    (*out)->cols_event = cols_event;   // (the table waits for it where its columns are read, and before its slab goes back to the pool)
    cols_event = nullptr;
    g_dev_t[7] += now_s() - t_host0;
    {
        DevFileStream& f0 = *r->dfiles[0];
        vlr_obs_table* t = *out;
        rc = vlr_dev_file_copy(f0.dev, (uint8_t*)t->slab.d + dl.off_lflags, (uint8_t*)t->slab.h + dl.off_lflags, dl.bytes - dl.off_lflags, 1);
        if (rc == VLR_OK) rc = vlr_dev_file_sync(f0.dev);
        if (rc != VLR_OK) { vlr_obs_table_free(t); *out = nullptr; return rc; }
    }
    (*out)->device = r->device;
    if (summaries) {
        // ---- observation summaries (the locus flags they need are on the device now); the columns stay in device memory
        vlr_obs_table* t = *out;
        DevFileStream& f0 = *r->dfiles[0];
        auto up = [](size_t x) { return (x + 63) & ~(size_t)63; };
        // device scratch: headers, run arrays, the distinct keys of every pileup in output order (one slot per observation), text
        // lengths and offsets, cursors, text
        const size_t o_hdr = 0, o_rpm = up((size_t)P * sizeof(vlr::PileSum)), o_rln = o_rpm + up((size_t)total * 4), o_ik = o_rln + up((size_t)total * 4),
                     o_ic = o_ik + up((size_t)total * 8), o_tl = o_ic + up((size_t)total * 4), o_to = o_tl + up((size_t)P * 4), o_cur = o_to + up((size_t)P * 4),
                     o_txt = o_cur + 64, need = o_txt + text_cap;
        if (r->sum_scratch.cap < need) {   // (the scratch a closed reader of this device left behind, if it is large enough)
            std::lock_guard<std::mutex> g(g_scratch_mu);
            auto& pk = parked_scratch();
            for (size_t k = 0; k < pk.size(); ++k)
                if (pk[k].device == r->device && pk[k].slab.cap >= need && !r->sum_scratch.d) { r->sum_scratch = pk[k].slab; pk.erase(pk.begin() + (long)k); break; }
        }
        if (r->sum_scratch.cap < need) {
            if (r->sum_scratch.d) vlr_dev_slab_free(r->device, r->sum_scratch.d, r->sum_scratch.h);
            r->sum_scratch = DevSlab();
            rc = vlr_dev_slab_alloc(r->device, need + need / 4, &r->sum_scratch.d, nullptr);
            if (rc != VLR_OK) { vlr_obs_table_free(t); *out = nullptr; return rc; }
            r->sum_scratch.cap = need + need / 4;
        }
        uint8_t* sd = (uint8_t*)r->sum_scratch.d;
        static const vlr::SumConsts kc = {std::log(3.0), std::log(20.0), std::log(150.0), std::numeric_limits<double>::epsilon()};
        vlr::DeviceCols dc;
        for (int k = 0; k < 9; ++k) dc.col[k] = (float*)((uint8_t*)t->slab.d + dl.off_col[k]);
        dc.flags = (uint32_t*)((uint8_t*)t->slab.d + dl.off_flags);
        dc.third = (int32_t*)((uint8_t*)t->slab.d + dl.off_third);
        uint32_t max_pile = 0;
        for (int64_t q = 0; q < P; ++q) max_pile = std::max(max_pile, obs_offset[(size_t)q + 1] - obs_offset[(size_t)q]);
        if (r->sum_copy_mark) rc = vlr_dev_file_wait_mark(f0.dev, r->sum_copy_mark);   // (the previous table's summaries may still be on their way out of the scratch)
        if (rc == VLR_OK) rc = vlr_dev_file_summaries(f0.dev, &dc, (const uint32_t*)((uint8_t*)t->slab.d + dl.off_obs), (const uint8_t*)t->slab.d + dl.off_lflags, L, S, max_pile, &kc,
                                    (vlr::PileSum*)(sd + o_hdr), sd + o_txt, (uint32_t)text_cap, (float*)(sd + o_rpm), (uint32_t*)(sd + o_rln), (uint64_t*)(sd + o_ik),
                                    (uint32_t*)(sd + o_ic), (uint32_t*)(sd + o_tl), (uint32_t*)(sd + o_to), (uint32_t*)(sd + o_cur));
        uint32_t cur[4] = {0, 0, 0, 0};
        if (rc == VLR_OK) rc = vlr_dev_file_copy(f0.dev, cur, sd + o_cur, 16, 0);
        if (rc == VLR_OK) rc = vlr_dev_file_sync(f0.dev);
        // host layout inside the column region: headers, run values, run lengths, text (pileups that found no room left their bytes
        // out: the cursor may stand beyond the buffer)
        uint8_t* hb = (uint8_t*)t->slab.h + dl.off_col[0];
        const size_t text_used = std::min<size_t>((size_t)cur[0], text_cap);
        const size_t h_rpm = up((size_t)P * sizeof(vlr::PileSum)), h_rln = h_rpm + up((size_t)cur[1] * 4), h_txt = h_rln + up((size_t)cur[1] * 4);
        if (r->async_columns) {
            // like the columns (vlr_obs_reader_set_async_columns): the summaries travel on the copy stream while the reader goes on — the
            // writer and everything else that reads them waits for the table's event; the next summary kernels wait for the mark
            if (rc == VLR_OK) rc = vlr_dev_file_copy_more(f0.dev, hb, sd + o_hdr, (size_t)P * sizeof(vlr::PileSum));
            if (rc == VLR_OK) rc = vlr_dev_file_copy_more(f0.dev, hb + h_rpm, sd + o_rpm, (size_t)cur[1] * 4);
            if (rc == VLR_OK) rc = vlr_dev_file_copy_more(f0.dev, hb + h_rln, sd + o_rln, (size_t)cur[1] * 4);
            void* ev = nullptr;
            if (rc == VLR_OK) rc = vlr_dev_file_copy_detached(f0.dev, hb + h_txt, sd + o_txt, text_used, &ev);
            if (rc == VLR_OK) rc = vlr_dev_file_copy_mark(f0.dev, &r->sum_copy_mark);
            if (rc == VLR_OK) { std::lock_guard<std::mutex> g(t->cols_mu); t->cols_event = ev; }
            else if (ev) vlr_dev_event_destroy(r->device, ev);
        } else {
            if (rc == VLR_OK) rc = vlr_dev_file_copy(f0.dev, hb, sd + o_hdr, (size_t)P * sizeof(vlr::PileSum), 0);
            if (rc == VLR_OK) rc = vlr_dev_file_copy(f0.dev, hb + h_rpm, sd + o_rpm, (size_t)cur[1] * 4, 0);
            if (rc == VLR_OK) rc = vlr_dev_file_copy(f0.dev, hb + h_rln, sd + o_rln, (size_t)cur[1] * 4, 0);
            if (rc == VLR_OK) rc = vlr_dev_file_copy(f0.dev, hb + h_txt, sd + o_txt, text_used, 0);
            if (rc == VLR_OK) rc = vlr_dev_file_sync(f0.dev);
        }
        if (rc != VLR_OK) { vlr_obs_table_free(t); *out = nullptr; return rc; }
        t->cols_on_host = false;
        t->has_summary = true;
        {   // pileups the kernel could not summarise (more than kSumMaxObs observations, or no room left for their text) need the columns
            // after all: when they are more than a few (the kernels count them: cursor[2]), fetch them now and stop summarising this file
            const int64_t n_over = (int64_t)cur[2];
            if (n_over * 50 > P) {
                r->summaries_off = true;
                rc = vlr_obs_table_fetch_columns(t);
                if (rc != VLR_OK) { vlr_obs_table_free(t); *out = nullptr; return rc; }
            }
        }
        t->sum_hdr = (const vlr::PileSum*)hb; t->sum_text = hb + h_txt;
        t->sum_run_pm = (const float*)(hb + h_rpm); t->sum_run_len = (const uint32_t*)(hb + h_rln);
    }
    g_dev_t[6] += now_s() - t_d2h0;
    for (int s = 0; s < S; ++s) r->dfiles[(size_t)s]->delivered += L;
    if (r->sharded) r->remaining -= L;
    g_dev_t[8] += now_s() - t_all0;
    g_dev_t[11] += (double)L;
    for (int s = 0; s < S; ++s) g_dev_t[13] += vlr_dev_file_inflate_seconds(r->dfiles[(size_t)s]->dev, 1);
    return VLR_OK;
}
}  // namespace

extern "C" {

// allow_stage: page-locked staging of the uploads (descriptor mode for large files).  A sharded reader maps its files: its window is one
// large feed anywhere in the file and it needs the whole member index up front.
static int open_device_impl(int device, int n_samples, const char* const* paths, uint32_t omit_bias_mask, int n_threads, bool allow_stage, vlr_obs_reader** out);
int vlr_obs_reader_open_device(int device, int n_samples, const char* const* paths, uint32_t omit_bias_mask, int n_threads, vlr_obs_reader** out) {
    return open_device_impl(device, n_samples, paths, omit_bias_mask, n_threads, true, out);
}
static int open_device_impl(int device, int n_samples, const char* const* paths, uint32_t omit_bias_mask, int n_threads, bool allow_stage, vlr_obs_reader** out) {
    if (!out || !paths || n_samples < 1 || n_samples > VLR_MAX_SAMPLES) return ifail(VLR_ERR_INVALID_ARGUMENT, "vlr_obs_reader_open_device: bad argument");
    *out = nullptr;
    std::unique_ptr<vlr_obs_reader> r(new vlr_obs_reader());
    r->omit = omit_bias_mask;
    r->n_threads = pick_threads(n_threads);
    r->on_device = true;
    r->device = device;
    r->pool = shared_dev_pool(device);
    for (int s = 0; s < n_samples; ++s) {
        r->paths.push_back(paths[s]);
        r->dfiles.emplace_back(new DevFileStream());
    }
    // the files side by side: mapping one and walking its member chain (two page touches per member, 345 000 members in the file of a
    // million tumor-normal records) is 45 ms of one thread
    std::vector<int> rcs((size_t)n_samples, VLR_OK);
    std::vector<std::string> errs((size_t)n_samples);
    {
        std::vector<std::thread> th;
        const char* e = getenv("VLR_INGEST_STAGE");
        const bool staged = allow_stage && !(e && atoi(e) == 0);
        auto open_one = [&, staged](int s) {
            int rc = staged ? dev_stream_open_fd(*r->dfiles[(size_t)s], paths[s], device) : VLR_ERR_UNSUPPORTED;
            if (rc == VLR_ERR_UNSUPPORTED && !r->dfiles[(size_t)s]->streaming) {
                r->dfiles[(size_t)s].reset(new DevFileStream());   // (whatever the first attempt parsed is dropped with it)
                rc = dev_stream_open(*r->dfiles[(size_t)s], paths[s], device);
            }
            rcs[(size_t)s] = rc;
            if (rc != VLR_OK) errs[(size_t)s] = vlr_last_error();
        };
        for (int s = 1; s < n_samples; ++s) th.emplace_back(open_one, s);
        open_one(0);
        for (auto& t : th) t.join();
    }
    for (int s = 0; s < n_samples; ++s)
        if (rcs[(size_t)s] != VLR_OK) return ifail(rcs[(size_t)s], "%s", errs[(size_t)s].c_str());   // (the message is per thread)
    {   // page-locked staging of the uploads (VLR_INGEST_STAGE=0: uploads from the mapping, staged by the reader's thread)
        const char* e = getenv("VLR_INGEST_STAGE");
        if (allow_stage && !(e && atoi(e) == 0))
            for (int s = 0; s < n_samples; ++s) {
                DevFileStream& f = *r->dfiles[(size_t)s];
                if (f.streaming || f.blocks.empty() || f.raw.size() < ((size_t)32 << 20)) continue;   // (descriptor mode has its ring; small files: nothing to gain)
                f.stage.reset(new StageRing());
                if (!f.stage->start(f.raw.p, f.raw.size(), f.blocks[0].off)) f.stage.reset();
            }
    }
    *out = r.release();
    return VLR_OK;
}

int vlr_obs_table_device_batch(const vlr_obs_table* t, vlr_batch* b) {
    if (!t || !b) return ifail(VLR_ERR_INVALID_ARGUMENT, "null argument");
    if (!t->has_dev) return ifail(VLR_ERR_INVALID_ARGUMENT, "vlr_obs_table_device_batch: the table was not read by a device reader");
    *b = t->dev_batch;
    return VLR_OK;
}

int vlr_obs_reader_set_host_columns(vlr_obs_reader* r, int keep) {
    if (!r) return ifail(VLR_ERR_INVALID_ARGUMENT, "null reader");
    r->host_columns = keep != 0;
    return VLR_OK;
}

int vlr_obs_reader_set_async_columns(vlr_obs_reader* r, int on) {
    if (!r) return ifail(VLR_ERR_INVALID_ARGUMENT, "null reader");
    r->async_columns = on != 0;
    return VLR_OK;
}

// ---- sharded device reader: N readers (ranks of a torchrun job, devices of a node) each inflate and decode about 1 / N of every file.
// The reference reads every record once (calling.rs:306-339, 357-367); the loci then shard across the GPUs (north star).  A shard is
// a contiguous range of RECORDS; where it lies in the compressed file is only known after inflating, so:
//   open   every reader takes the members of its byte share of every file, plus a few members of lead-in and tail, inflates them, finds
//          the record starts (first start: a guess, confirmed by the walk and by the neighbour) and counts the records that start in
//          its own members;
//   counts (vlr_obs_reader_shard_counts) are exchanged by the caller — one small all-gather, or plain memory inside one process;
//   assign (vlr_obs_reader_shard_assign) checks that consecutive shards meet (landing of shard k - 1 = first start of shard k in
//          every file), numbers the records, takes the record range of the FIRST file's share as this reader's range in every file
//          (the shares of the other files differ by a few records: that is what lead-in and tail are for) and positions the files.
// vlr_obs_reader_next then hands out the shard's records like any reader.
namespace {
struct ShardRow { int64_t own, first_m, first_o, land_m, land_o, lead, total; };
constexpr int kShardRow = 7;

// window offset -> (member index, byte in the member); `pre` = inflated bytes in front of member w0 + j, j = 0 .. n
void member_of(const DevFileStream& f, const std::vector<uint64_t>& pre, uint64_t woff, uint64_t out[2]) {
    size_t j = (size_t)(std::upper_bound(pre.begin(), pre.end(), woff) - pre.begin());
    j = j == 0 ? 0 : j - 1;
    // (the LAST member that starts at or in front of the offset: a position on a member boundary belongs to the member behind it, and a
    //  position at the window's end to the member behind the window — the same pair whichever shard's window it is computed in)
    if (j + 1 >= pre.size()) { out[0] = f.w0 + (pre.size() - 1); out[1] = 0; return; }
    out[0] = f.w0 + j; out[1] = woff - pre[j];
}

int shard_scan_file(vlr_obs_reader* r, DevFileStream& f) {
    const int N = r->n_shards, k = r->shard;
    const size_t nb = f.blocks.size();
    const uint64_t total_c = nb ? (uint64_t)(f.blocks.back().off + f.blocks.back().clen) : 0;
    auto share_begin = [&](int q) -> size_t {
        if (q <= 0) return 0;
        if (q >= N) return nb;
        const uint64_t at = total_c / (uint64_t)N * (uint64_t)q;
        size_t lo = 0, hi = nb;
        while (lo < hi) { const size_t mid = (lo + hi) / 2; if ((uint64_t)f.blocks[mid].off < at) lo = mid + 1; else hi = mid; }
        return lo;
    };
    f.mb0 = share_begin(k); f.mb1 = share_begin(k + 1);
    double slack = 1.0 / 32.0;
    if (const char* ev = getenv("VLR_INGEST_SHARD_SLACK")) slack = std::max(0.0, atof(ev));
    const size_t extra = std::max<size_t>(8, (size_t)((double)(f.mb1 - f.mb0) * slack));
    f.w0 = k > 0 ? (f.mb0 > extra ? f.mb0 - extra : 0) : 0;
    f.block_limit = k + 1 < N ? std::min(nb, f.mb1 + extra) : nb;
    // the header is only in the window that starts at the file's first member
    size_t hdr_members = 0;
    { uint64_t acc = 0; while (hdr_members < nb && acc < f.header_bytes) acc += f.blocks[hdr_members++].isize; }
    if (f.w0 < hdr_members) f.w0 = 0;   // (a window that would begin inside the header begins at the file's start)
    f.next_block = f.w0;
    f.header_skipped = f.w0 != 0;
    int rc = dev_stream_feed(f, ~0ull >> 2);
    if (rc != VLR_OK) return rc;
    rc = vlr_dev_file_feed_wait(f.dev);
    if (rc != VLR_OK) return rc;
    uint64_t skipped = f.w0 == 0 ? f.header_bytes : 0;
    if (f.w0 != 0) {
        uint64_t sk = 0;
        rc = vlr_dev_file_anchor_first(f.dev, (int)f.h.contigs.size(), f.n_hdr_samples, &sk);
        if (rc != VLR_OK) return rc;
        skipped = sk;
    }
    // every record of the window (the estimate of their number is doubled until it was large enough)
    int64_t n = 0;
    const vlr::RecHost* rh = nullptr;
    int64_t cap = (int64_t)(vlr_dev_file_buffered(f.dev) / 4096) + 4096;
    for (;;) {
        int serial = 0;
        rc = vlr_dev_file_split(f.dev, cap, (int)f.h.contigs.size(), f.n_hdr_samples, f.field_of_key.data(), (int)f.field_of_key.size(), &n, &rh, &serial);
        if (rc != VLR_OK) return rc;
        if (n < cap) break;
        cap *= 2;
    }
    int64_t ns = 0;
    const uint64_t* st = vlr_dev_file_starts(f.dev, &ns);
    std::vector<uint64_t> pre(1, 0);
    for (size_t m = f.w0; m < f.block_limit && m < nb; ++m) pre.push_back(pre.back() + f.blocks[m].isize);
    const uint64_t U0 = pre[std::min(f.mb0 - f.w0, pre.size() - 1)], U1 = pre[std::min(f.mb1 - f.w0, pre.size() - 1)];
    int64_t lead = 0, own = 0;
    for (int64_t q = 0; q < n; ++q) {
        const uint64_t w = skipped + st[q];
        if (w < U0) ++lead; else if (w < U1 || k + 1 == N) ++own;
    }
    f.sh_lead = lead; f.sh_own = own; f.sh_total = n;
    if (k + 1 < N && f.block_limit < nb && skipped + (n ? st[n] : 0) < U1)   // (the record behind the last complete one starts in the own share)
        return ifail(VLR_ERR_INVALID_ARGUMENT, "sharded reader: the last record of the shard of %s is not complete in its window (VLR_INGEST_SHARD_SLACK)", f.path.c_str());
    member_of(f, pre, skipped + (n ? st[lead] : 0), f.sh_first);
    member_of(f, pre, skipped + (n ? st[lead + own] : 0), f.sh_land);
    if (n == 0) { f.sh_first[0] = f.sh_land[0] = f.mb0; f.sh_first[1] = f.sh_land[1] = 0; }
    return VLR_OK;
}
}  // namespace

int vlr_obs_reader_open_device_shard(int device, int n_samples, const char* const* paths, uint32_t omit_bias_mask, int n_threads, int shard, int n_shards, vlr_obs_reader** out) {
    if (!out || n_shards < 1 || shard < 0 || shard >= n_shards) return ifail(VLR_ERR_INVALID_ARGUMENT, "vlr_obs_reader_open_device_shard: bad argument");
    vlr_obs_reader* r = nullptr;
    int rc = open_device_impl(device, n_samples, paths, omit_bias_mask, n_threads, false, &r);
    if (rc != VLR_OK) return rc;
    r->sharded = true; r->shard = shard; r->n_shards = n_shards;
    for (auto& f : r->dfiles) {
        rc = shard_scan_file(r, *f);
        if (rc != VLR_OK) { vlr_obs_reader_close(r); return rc; }
    }
    *out = r;
    return VLR_OK;
}

int vlr_obs_reader_shard_row_size(void) { return kShardRow; }

int vlr_obs_reader_shard_counts(const vlr_obs_reader* r, int64_t* out) {
    if (!r || !out || !r->sharded) return ifail(VLR_ERR_INVALID_ARGUMENT, "vlr_obs_reader_shard_counts: not a sharded reader");
    for (size_t s = 0; s < r->dfiles.size(); ++s) {
        const DevFileStream& f = *r->dfiles[s];
        int64_t* o = out + (size_t)kShardRow * s;
        o[0] = f.sh_own; o[1] = (int64_t)f.sh_first[0]; o[2] = (int64_t)f.sh_first[1]; o[3] = (int64_t)f.sh_land[0]; o[4] = (int64_t)f.sh_land[1];
        o[5] = f.sh_lead; o[6] = f.sh_total;
    }
    return VLR_OK;
}

int vlr_obs_reader_shard_assign(vlr_obs_reader* r, const int64_t* all, int64_t* first_record, int64_t* n_records) {
    if (!r || !all || !r->sharded) return ifail(VLR_ERR_INVALID_ARGUMENT, "vlr_obs_reader_shard_assign: not a sharded reader");
    const int N = r->n_shards, S = (int)r->dfiles.size(), k = r->shard;
    auto row = [&](int q, int s) { return all + ((size_t)q * (size_t)S + (size_t)s) * kShardRow; };
    // consecutive shards meet: the start behind the last own record of shard q - 1 is the first own start of shard q
    for (int s = 0; s < S; ++s)
        for (int q = 1; q < N; ++q) {
            const int64_t* a = row(q - 1, s);
            const int64_t* b = row(q, s);
            if (b[0] == 0 && a[0] == 0) continue;
            if (a[3] != b[1] || a[4] != b[2])
                return ifail(VLR_ERR_INVALID_ARGUMENT, "sharded reader: shards %d and %d of %s do not meet (a guessed record start was wrong); read the file unsharded", q - 1, q, r->paths[(size_t)s].c_str());
        }
    // records in front of shard q of file s
    std::vector<int64_t> C((size_t)(N + 1) * (size_t)S, 0);
    for (int s = 0; s < S; ++s)
        for (int q = 0; q < N; ++q) C[(size_t)(q + 1) * S + s] = C[(size_t)q * S + s] + row(q, s)[0];
    for (int s = 1; s < S; ++s)
        if (C[(size_t)N * S + s] != C[(size_t)N * S])
            return ifail(VLR_ERR_INVALID_ARGUMENT, "inconsistent observations: %s holds %lld records, %s %lld (calling.rs:369-371)", r->paths[(size_t)s].c_str(),
                         (long long)C[(size_t)N * S + s], r->paths[0].c_str(), (long long)C[(size_t)N * S]);
    const int64_t B0 = C[(size_t)k * S], B1 = C[(size_t)(k + 1) * S];   // this reader's records: those of the first file's share
    for (int s = 0; s < S; ++s) {
        DevFileStream& f = *r->dfiles[(size_t)s];
        const int64_t skip = f.sh_lead + (B0 - C[(size_t)k * S + s]);
        if (skip < 0 || skip + (B1 - B0) > f.sh_total)
            return ifail(VLR_ERR_INVALID_ARGUMENT, "sharded reader: the window of %s does not hold records %lld .. %lld (it starts %lld records before its share and holds %lld): raise VLR_INGEST_SHARD_SLACK",
                         f.path.c_str(), (long long)B0, (long long)B1, (long long)f.sh_lead, (long long)f.sh_total);
        const int rc = vlr_dev_file_consume(f.dev, skip);
        if (rc != VLR_OK) return rc;
        f.delivered = B0;
    }
    r->remaining = B1 - B0;
    r->assigned = true;
    if (first_record) *first_record = B0;
    if (n_records) *n_records = B1 - B0;
    return VLR_OK;
}

// The same for the devices of a node driven from ONE process (vlr_node_*): one sharded reader per device, opened side by side on threads
// of their own, the counts exchanged in memory.  out[vlr_node_n_devices(node)]; reader r delivers the records of shard r on
// vlr_node_device(node, r) — the tables of its vlr_obs_reader_next go to vlr_batch_run_device_in on vlr_node_plan(node, r).
int vlr_node_obs_readers_open(vlr_gpu_node* node, int n_samples, const char* const* paths, uint32_t omit_bias_mask, int n_threads, vlr_obs_reader** out) {
    if (!node || !out || !paths || n_samples < 1) return ifail(VLR_ERR_INVALID_ARGUMENT, "vlr_node_obs_readers_open: bad argument");
    const int N = vlr_node_n_devices(node);
    if (N < 1) return ifail(VLR_ERR_INVALID_ARGUMENT, "vlr_node_obs_readers_open: empty node");
    std::vector<vlr_obs_reader*> rd((size_t)N, nullptr);
    std::vector<int> rcs((size_t)N, VLR_OK);
    std::vector<std::string> errs((size_t)N);
    const int per = std::max(1, pick_threads(n_threads) / N);
    {
        std::vector<std::thread> th;
        for (int r = 0; r < N; ++r)
            th.emplace_back([&, r] {
                rcs[(size_t)r] = vlr_obs_reader_open_device_shard(vlr_node_device(node, r), n_samples, paths, omit_bias_mask, per, r, N, &rd[(size_t)r]);
                if (rcs[(size_t)r] != VLR_OK) errs[(size_t)r] = vlr_last_error();   // (the message is per thread)
            });
        for (auto& t : th) t.join();
    }
    auto close_all = [&] { for (auto* q : rd) if (q) vlr_obs_reader_close(q); };
    for (int r = 0; r < N; ++r)
        if (rcs[(size_t)r] != VLR_OK) { close_all(); return ifail(rcs[(size_t)r], "%s", errs[(size_t)r].c_str()); }
    std::vector<int64_t> all((size_t)N * (size_t)n_samples * kShardRow);
    for (int r = 0; r < N; ++r) {
        const int rc = vlr_obs_reader_shard_counts(rd[(size_t)r], all.data() + (size_t)r * (size_t)n_samples * kShardRow);
        if (rc != VLR_OK) { close_all(); return rc; }
    }
    for (int r = 0; r < N; ++r) {
        const int rc = vlr_obs_reader_shard_assign(rd[(size_t)r], all.data(), nullptr, nullptr);
        if (rc != VLR_OK) { close_all(); return rc; }
    }
    for (int r = 0; r < N; ++r) out[r] = rd[(size_t)r];
    return VLR_OK;
}

int vlr_obs_table_summaries(const vlr_obs_table* t, int64_t* n_overflow) {
    if (n_overflow) *n_overflow = 0;
    if (!t || !t->has_summary) return 0;
    if (n_overflow && const_cast<vlr_obs_table*>(t)->wait_columns() != VLR_OK) return 0;   // (the headers may still be on their way)
    if (n_overflow)
        for (int64_t p = 0; p < t->n_loci * t->n_samples; ++p) *n_overflow += t->sum_hdr[p].overflow != 0;
    return 1;
}

int vlr_obs_table_fetch_columns(vlr_obs_table* t) {
    if (!t) return ifail(VLR_ERR_INVALID_ARGUMENT, "null table");
    { const int rcw = t->wait_columns(); if (rcw != VLR_OK) return rcw; }
    if (t->cols_on_host) return VLR_OK;
    // (the summaries live where the columns go: they are gone afterwards, the writer then counts from the columns)
    t->has_summary = false;
    const int rc = vlr_dev_copy_to_host(t->device, (uint8_t*)t->slab.h + t->col_region_off, (const uint8_t*)t->slab.d + t->col_region_off, t->col_region_bytes);
    if (rc != VLR_OK) return rc;
    t->cols_on_host = true;
    return VLR_OK;
}

void vlr_ingest_device_timings(double* out16, int reset) {
    if (out16) for (int i = 0; i < 16; ++i) out16[i] = g_dev_t[i];
    if (reset) for (int i = 0; i < 16; ++i) g_dev_t[i] = 0.0;
}

void vlr_ingest_device_trim(void) {
    vlr_dev_file_trim();
    {   // the page-locked staging rings closed readers left behind
        std::lock_guard<std::mutex> g(StageRing::pool_mu());
        for (uint8_t* q : StageRing::pool()) vlr_host_free(q);
        StageRing::pool().clear();
    }
    std::lock_guard<std::mutex> g(g_scratch_mu);
    for (auto& q : parked_scratch()) vlr_dev_slab_free(q.device, q.slab.d, q.slab.h);
    parked_scratch().clear();
}

}  // extern "C"

// ================================================================================================ writers
namespace {

// ---- BCF2 encoding helpers
void put_u32(std::vector<uint8_t>& o, uint32_t v) { for (int i = 0; i < 4; ++i) o.push_back((uint8_t)(v >> (8 * i))); }
void put_typed_int_scalar(std::vector<uint8_t>& o, int32_t v) {
    if (v >= -120 && v <= 127) { o.push_back(0x11); o.push_back((uint8_t)(int8_t)v); }
    else if (v >= -32760 && v <= 32767) { o.push_back(0x12); o.push_back((uint8_t)(v & 0xff)); o.push_back((uint8_t)((v >> 8) & 0xff)); }
    else { o.push_back(0x13); put_u32(o, (uint32_t)v); }
}
void put_desc(std::vector<uint8_t>& o, uint32_t n, int type) {
    if (n < 15) o.push_back((uint8_t)((n << 4) | type));
    else { o.push_back((uint8_t)(0xF0 | type)); put_typed_int_scalar(o, (int32_t)n); }
}
void put_str(std::vector<uint8_t>& o, const char* s, size_t n) { put_desc(o, (uint32_t)n, 7); o.insert(o.end(), s, s + n); }
int int_type_for(int32_t lo, int32_t hi) { return (lo >= -120 && hi <= 127) ? 1 : (lo >= -32760 && hi <= 32767) ? 2 : 3; }
void put_int_vec(std::vector<uint8_t>& o, const int32_t* v, uint32_t n) {  // typed vector, smallest width
    int32_t lo = 0, hi = 0;
    for (uint32_t i = 0; i < n; ++i) { lo = std::min(lo, v[i]); hi = std::max(hi, v[i]); }
    const int t = int_type_for(lo, hi);
    put_desc(o, n, t);
    for (uint32_t i = 0; i < n; ++i) {
        if (t == 1) o.push_back((uint8_t)(int8_t)v[i]);
        else if (t == 2) { o.push_back((uint8_t)(v[i] & 0xff)); o.push_back((uint8_t)((v[i] >> 8) & 0xff)); }
        else put_u32(o, (uint32_t)v[i]);
    }
}

struct OutHeader {
    std::string text;                              // with the PASS filter line, ends with '\n'
    std::unordered_map<std::string, int> dict, contigs;
};
void build_out_header(const std::string& header_text, OutHeader& h) {
    std::vector<std::string> meta;
    std::string chrom_line;
    size_t p = 0;
    while (p < header_text.size()) {
        size_t e = header_text.find('\n', p);
        if (e == std::string::npos) e = header_text.size();
        std::string line = header_text.substr(p, e - p);
        p = e + 1;
        if (line.empty()) continue;
        if (line.compare(0, 2, "##") == 0) meta.push_back(line);
        else if (line.compare(0, 6, "#CHROM") == 0) chrom_line = line;
    }
    bool has_pass = false;
    for (auto& l : meta) has_pass = has_pass || l.compare(0, 17, "##FILTER=<ID=PASS") == 0;
    if (!has_pass) meta.insert(meta.begin() + std::min<size_t>(1, meta.size()), "##FILTER=<ID=PASS,Description=\"All filters passed\">");
    h.dict["PASS"] = 0;
    int next = 1, cnext = 0;
    for (auto& l : meta) {
        const bool d = l.compare(0, 9, "##FILTER=") == 0 || l.compare(0, 7, "##INFO=") == 0 || l.compare(0, 9, "##FORMAT=") == 0;
        if (d) {
            const std::string body = l.substr(l.find('<'));
            const std::string id = attr(body, "ID"), idx = attr(body, "IDX");
            if (!h.dict.count(id)) { const int i = idx.empty() ? next : atoi(idx.c_str()); h.dict[id] = i; next = std::max(next, i + 1); }
        } else if (l.compare(0, 9, "##contig=") == 0) {
            const std::string body = l.substr(l.find('<'));
            const std::string id = attr(body, "ID"), idx = attr(body, "IDX");
            if (!h.contigs.count(id)) { const int i = idx.empty() ? cnext : atoi(idx.c_str()); h.contigs[id] = i; cnext = std::max(cnext, i + 1); }
        }
    }
    for (auto& l : meta) { h.text += l; h.text.push_back('\n'); }
    h.text += chrom_line;
    h.text.push_back('\n');
}

// htslib's %g of an f32 (callsfmt.fmt_float)
std::string fmt_g(float x) {
    if (x != x) return ".";
    if (std::isinf(x)) return x > 0 ? "inf" : "-inf";
    char b[48];
    snprintf(b, sizeof b, "%g", (double)x);
    return b;
}
bool relative_eq(double a, double b) {
    if (a == b) return true;
    if (std::isinf(a) || std::isinf(b)) return false;  // (approx::RelativeEq: infinities are equal only to themselves)
    const double d = std::fabs(a - b), eps = std::numeric_limits<double>::epsilon();
    return d <= eps || d <= std::max(std::fabs(a), std::fabs(b)) * eps;
}
char kr_letter(double bf) {  // utils/mod.rs:158-167 over bio's Kass-Raftery scale
    if (bf <= 1.0) return relative_eq(bf, 1.0) ? 'E' : 'N';
    if (bf <= 3.0) return 'B';
    if (bf <= 20.0) return 'P';
    if (bf <= 150.0) return 'S';
    return 'V';
}
// utils/mod.rs:122-156 generalized_cigar, keep_order = false: counts in first-appearance order, stable by count desc, stable by aux
template <typename Aux>
std::string cigar_of_counts(std::vector<std::pair<std::string, int>>& cnt, Aux aux) {
    std::stable_sort(cnt.begin(), cnt.end(), [](const auto& a, const auto& b) { return a.second > b.second; });
    std::stable_sort(cnt.begin(), cnt.end(), [&](const auto& a, const auto& b) { return aux(a.first) < aux(b.first); });
    std::string out;
    for (auto& kv : cnt) { out += std::to_string(kv.second); out += kv.first; }
    return out;
}
template <typename Aux>
std::string generalized_cigar(const std::vector<std::string>& items, Aux aux) {
    std::vector<std::pair<std::string, int>> cnt;  // a handful of distinct items: a linear search beats hashing
    for (auto& it : items) {
        size_t k = 0;
        while (k < cnt.size() && cnt[k].first != it) ++k;
        if (k == cnt.size()) cnt.emplace_back(it, 1);
        else cnt[k].second++;
    }
    std::stable_sort(cnt.begin(), cnt.end(), [](const auto& a, const auto& b) { return a.second > b.second; });
    std::stable_sort(cnt.begin(), cnt.end(), [&](const auto& a, const auto& b) { return aux(a.first) < aux(b.first); });
    std::string out;
    for (auto& kv : cnt) { out += std::to_string(kv.second); out += kv.first; }
    return out;
}

// "%.<digits>f" of a finite double below 1e12 in magnitude, correctly rounded like printf (exact decimal expansion of the binary
// value, ties to even): x = v * 10^digits with its exact rounding error from an fma, so a product that lands on k + 1/2 is decided
// by the side the true value lies on.  Appends to `out`; anything else (inf, nan, huge) goes through snprintf.
inline void append_fixed(std::string& out, double v, int digits) {
    static const double p10[4] = {1.0, 10.0, 100.0, 1000.0};
    if (!(std::fabs(v) < 1e12) || digits < 0 || digits > 3) {   // (v * 10^digits stays below 2^53: floor and the tie test are exact)
        char b[64];
        snprintf(b, sizeof b, "%.*f", digits, v);
        out += b;
        return;
    }
    const bool neg = std::signbit(v);
    const double a = std::fabs(v), sc = p10[digits];
    const double x = a * sc, err = std::fma(a, sc, -x);   // a * sc = x + err exactly
    double k = std::floor(x);
    const double frac = x - k;                             // exact
    bool up;
    if (frac > 0.5) up = true;
    else if (frac < 0.5) up = false;
    else up = err > 0.0 || (err == 0.0 && std::fmod(k, 2.0) == 1.0);   // on the tie: the side of the exact value, then half to even
    // (frac just below / above 1/2 by less than |err| cannot happen: |err| <= ulp(x)/2 and frac is a multiple of ulp(x))
    if (up) k += 1.0;
    uint64_t n = (uint64_t)k;
    char buf[40];
    int pos = 40;
    for (int d = 0; d < digits; ++d) { buf[--pos] = (char)('0' + n % 10); n /= 10; }
    if (digits) buf[--pos] = '.';
    do { buf[--pos] = (char)('0' + n % 10); n /= 10; } while (n);
    if (neg) buf[--pos] = '-';   // (printf prints -0.000 for a negative value that rounds to zero)
    out.append(buf + pos, (size_t)(40 - pos));
}

#ifdef VLR_WRITER_PROF
#include <x86intrin.h>
static std::atomic<unsigned long long> g_wprof[8];
struct WProf { int k; unsigned long long t0; WProf(int k_) : k(k_), t0(__rdtsc()) {} ~WProf() { g_wprof[k] += __rdtsc() - t0; } };
#define WPROF(k) WProf wprof_##k(k)
extern "C" void vlr_writer_prof(unsigned long long* out8, int reset) { for (int i = 0; i < 8; ++i) { out8[i] = g_wprof[i]; if (reset) g_wprof[i] = 0; } }
#else
#define WPROF(k)
#endif
struct SampleFields { int32_t dp = 0, oobs = 0; float af = NAN; std::string saobs, srobs, obs, sym[6], afd; bool has_afd = false; };

// Call::write_final_record, per sample (calling/variants/mod.rs:233-360, 473-559); mirrors callsfmt.sample_fields
void sample_fields(const vlr_obs_table* t, const vlr_results* r, int64_t l, int s, SampleFields& o) {
    const int S = t->n_samples;
    const uint32_t b = t->obs_offset[l * S + s], e = t->obs_offset[l * S + s + 1];
    const bool drop_nonstd = t->locus_flags[l] & VLR_LOCUS_REMOVE_NONSTANDARD;  // pileup.rs:26-43
    // observations are counted by a packed key (two score characters, the flag characters, third-allele evidence); the strings
    // are only built for the distinct keys, in first-appearance order (what Counter::most_common sees)
    static thread_local std::vector<std::pair<uint64_t, int>> obs_cnt;
    static thread_local std::vector<int> slots;
    obs_cnt.clear(); slots.clear();
    std::vector<std::pair<std::string, int>> alt_cnt, ref_cnt;  // one-letter items: at most twelve distinct
    auto count_letter = [](std::vector<std::pair<std::string, int>>& v, char ch) {
        for (auto& kv : v)
            if (kv.first[0] == ch) { kv.second++; return; }
        v.emplace_back(std::string(1, ch), 1);
    };
    double depth = 0.0;
    int kept = 0;
    double last_pm = NAN, last_w = 0.0;
    static const double kLn3 = std::log(3.0), kLn20 = std::log(20.0), kLn150 = std::log(150.0);
    const bool from_summary = t->has_summary;
    if (from_summary) {
        // the device counted per observation (vlr_decode.hip obs_text_kernel: this loop and the OBS text below, one wave per pileup); what is left is per pileup
        const vlr::PileSum& h = t->sum_hdr[l * S + s];
        kept = (int)h.kept;
        o.obs.assign((const char*)t->sum_text + h.obs_off, (size_t)h.obs_len);   // (counted, ordered and written by obs_text_kernel)
        for (uint32_t q = 0; q < h.alt_n; ++q) alt_cnt.emplace_back(std::string(1, (char)h.alt_letter[q]), (int)h.alt_cnt[q]);
        for (uint32_t q = 0; q < h.ref_n; ++q) ref_cnt.emplace_back(std::string(1, (char)h.ref_letter[q]), (int)h.ref_cnt[q]);
        for (uint32_t q = 0; q < h.n_run; ++q) {   // the same sequence of additions as the loop below (run 0 sits in the header)
            const double w = std::exp((double)(q ? t->sum_run_pm[h.run_off + q - 1] : h.run0_pm));
            const uint32_t len = q ? t->sum_run_len[h.run_off + q - 1] : h.run0_len;
            for (uint32_t j = 0; j < len; ++j) depth += w;
        }
    }
    { WPROF(0);
    for (uint32_t i = b; i < e && !from_summary; ++i) {
        const uint32_t f = t->flags[i];
        const unsigned orient = (f >> VLR_F_ORIENT_SHIFT) & 3;
        if (drop_nonstd && orient == VLR_ORIENT_OTHER) continue;
        ++kept;
        const double pa = t->col[1][i], pr = t->col[2][i], pm = t->col[0][i];
        if (pm != last_pm) { last_pm = pm; last_w = std::exp(pm); }   // (the MAPQ-adjusted mean: one value per pileup as a rule)
        depth += last_w;
        // Bayes factors exp(pa - pr) / exp(pr - pa) against the Kass-Raftery bounds 1, 3, 20, 150 in log space: d is a difference of
        // two f32 values — a dyadic rational that cannot sit within rounding distance of ln 3, ln 20 or ln 150 — so the comparisons
        // are those of the exponentials; only differences the exponential rounds to 1 (|d| < 1e-15) take the exponentials themselves
        const double d = pa - pr;
        double bf_alt, bf_ref;
        char kl_alt, kl_ref;
        if (std::fabs(d) >= 1e-15 && std::fabs(d) < 700.0) {
            const double ad = std::fabs(d);
            const char k = ad <= kLn3 ? 'B' : ad <= kLn20 ? 'P' : ad <= kLn150 ? 'S' : 'V';
            bf_alt = d > 0 ? 2.0 : 0.5; bf_ref = d > 0 ? 0.5 : 2.0;   // (only their order is used below)
            kl_alt = d > 0 ? k : 'N'; kl_ref = d > 0 ? 'N' : k;
        } else {
            bf_alt = std::exp(d); bf_ref = std::exp(-d);
            kl_alt = kr_letter(bf_alt); kl_ref = kr_letter(bf_ref);
        }
        const bool maxq = f & VLR_F_MAX_MAPQ;
        char s0, s1 = 0;
        if (bf_alt > bf_ref) { s0 = 'A'; s1 = kl_alt; }
        else if (bf_ref > bf_alt) { s0 = 'R'; s1 = kl_ref; }
        else s0 = 'E';
        if (!maxq) { s0 = (char)tolower(s0); if (s1) s1 = (char)tolower(s1); }
        const unsigned strand = (f >> VLR_F_STRAND_SHIFT) & 3, altloc = (f >> VLR_F_ALTLOCUS_SHIFT) & 3;
        const bool hp_err = (f & VLR_F_HP_LEN_VALID) && ((f >> VLR_F_HP_LEN_SHIFT) & 0xff) != 0;
        const uint64_t key = (uint64_t)(uint8_t)s0 | ((uint64_t)(uint8_t)s1 << 8) | ((uint64_t)((f & VLR_F_PAIRED) ? 1 : 0) << 16) |
                             ((uint64_t)(altloc > 2 ? 2 : altloc) << 17) | ((uint64_t)strand << 19) | ((uint64_t)orient << 21) |
                             ((uint64_t)((f & VLR_F_READPOS_MAJOR) ? 1 : 0) << 23) | ((uint64_t)((f & VLR_F_SOFTCLIPPED) ? 1 : 0) << 24) |
                             ((uint64_t)(hp_err ? 1 : 0) << 25) | ((uint64_t)(uint32_t)(t->third[i] + 1) << 32);
        {   // first-appearance counting through a small open-addressing table (synthetic pileups have almost as many distinct keys
            // as observations: the linear search was quadratic)
            if (slots.empty() || obs_cnt.size() * 2 >= slots.size()) {
                slots.assign(slots.empty() ? 64 : slots.size() * 2, -1);
                for (size_t q = 0; q < obs_cnt.size(); ++q) {
                    size_t hq = (size_t)((obs_cnt[q].first * 0x9E3779B97F4A7C15ull) >> 32) & (slots.size() - 1);
                    while (slots[hq] >= 0) hq = (hq + 1) & (slots.size() - 1);
                    slots[hq] = (int)q;
                }
            }
            size_t hq = (size_t)((key * 0x9E3779B97F4A7C15ull) >> 32) & (slots.size() - 1);
            while (slots[hq] >= 0 && obs_cnt[(size_t)slots[hq]].first != key) hq = (hq + 1) & (slots.size() - 1);
            if (slots[hq] < 0) { slots[hq] = (int)obs_cnt.size(); obs_cnt.emplace_back(key, 1); }
            else obs_cnt[(size_t)slots[hq]].second++;
        }
        if (pa > pr) { const char c = kl_alt; count_letter(alt_cnt, maxq ? (char)toupper(c) : (char)tolower(c)); }
        else { const char c = kl_ref; count_letter(ref_cnt, maxq ? (char)toupper(c) : (char)tolower(c)); }
    }
    }
    WPROF(1);
    // generalized_cigar(keep_order = false) of the items (utils/mod.rs:122-156): stable by count descending, then stable by the
    // auxiliary class — one sort of (class, -count, first appearance) — and the item text is written once, in output order
    static thread_local std::vector<uint64_t> ord;
    ord.clear();
    for (size_t q = 0; q < obs_cnt.size() && !from_summary; ++q) {
        const char s0 = (char)(obs_cnt[q].first & 0xff);
        const uint64_t aux = s0 == 'N' ? 2 : s0 == 'E' ? 1 : 0;
        ord.push_back((aux << 56) | ((uint64_t)(0xffffffu - (uint32_t)std::min(obs_cnt[q].second, 0xffffff)) << 32) | (uint64_t)q);
    }
    std::sort(ord.begin(), ord.end());
    if (!from_summary) o.obs.clear();
    for (uint64_t v : ord) {
        const auto& kc = obs_cnt[(size_t)(v & 0xffffffffu)];
        const uint64_t key = kc.first;
        char item[40];
        int n = 0;
        {   // count
            char d[12]; int m = 0; unsigned c = (unsigned)kc.second;
            do { d[m++] = (char)('0' + c % 10); c /= 10; } while (c);
            while (m) item[n++] = d[--m];
        }
        item[n++] = (char)(key & 0xff);
        if ((key >> 8) & 0xff) item[n++] = (char)((key >> 8) & 0xff);
        const uint32_t th = (uint32_t)(key >> 32);
        if (th) { char d[12]; int m = 0; uint32_t c = th - 1; do { d[m++] = (char)('0' + c % 10); c /= 10; } while (c); while (m) item[n++] = d[--m]; }
        else item[n++] = '.';
        item[n++] = ((key >> 16) & 1) ? 'p' : 's';
        item[n++] = "#*."[(key >> 17) & 3];
        item[n++] = "+-*."[(key >> 19) & 3];
        item[n++] = "><*!"[(key >> 21) & 3];
        item[n++] = ((key >> 23) & 1) ? '^' : '*';
        item[n++] = ((key >> 24) & 1) ? '$' : '.';
        item[n++] = ((key >> 25) & 1) ? '*' : '.';
        o.obs.append(item, (size_t)n);
    }
    o.dp = kept ? (int32_t)std::floor(depth + 0.5) : 0;  // expected_depth (read_observation.rs:43-47)
    o.oobs = (int32_t)(e - b) - kept;
    auto simple = [](const std::string& k) { return k[0] == 'R' ? 2 : (k.back() == 'E' ? 1 : 0); };
    o.saobs = cigar_of_counts(alt_cnt, simple);
    o.srobs = cigar_of_counts(ref_cnt, simple);
    const uint8_t* mb = r->map_bias ? r->map_bias + l * VLR_N_BIAS : nullptr;
    static const char* sy[6] = {".+-", ".><", ".^", ".$", ".*", ".*"};
    bool any_bias = false;
    for (int k = 0; k < 6; ++k) { const int v = mb ? mb[k] : 0; o.sym[k] = std::string(1, sy[k][v < (int)strlen(sy[k]) ? v : 0]); any_bias = any_bias || v; }
    o.af = (float)r->map_vaf[l * S + s];
    o.afd = ".";
    o.has_afd = false;
    const uint32_t* tspan = (r->afd_count && r->afd_text && r->afd_text_span) ? r->afd_text_span + 2 * (size_t)(l * S + s) : nullptr;
    if (tspan && !any_bias && tspan[1] != 0xffffffffu) {   // formatted on the device (afd_text_kernel of vlr_decode.hip: the loop below)
        o.afd.assign((const char*)r->afd_text + tspan[0], (size_t)tspan[1]);
        o.has_afd = true;
    } else if (r->afd_count && !any_bias) {
        WPROF(2);
        const int n = std::min(r->afd_count[l * S + s], r->afd_capacity);
        const double* v = r->afd_vaf + (size_t)(l * S + s) * r->afd_capacity;
        const double* p = r->afd_lnprob + (size_t)(l * S + s) * r->afd_capacity;
        // stable order by allele frequency: the lists are short (<= capacity) and mostly ascending already — an insertion sort on
        // the stack instead of std::stable_sort's heap buffer per call
        int order_buf[256];
        std::vector<int> order_big;
        int* order = order_buf;
        if (n > 256) { order_big.resize((size_t)n); order = order_big.data(); }
        for (int i = 0; i < n; ++i) {
            int j = i;
            while (j > 0 && v[i] < v[order[j - 1]]) { order[j] = order[j - 1]; --j; }
            order[j] = i;
        }
        std::string out;
        out.reserve((size_t)n * 12);
        static const double kLn10 = std::log(10.0);
        for (int k = 0; k < n; ++k) {
            const int i = order[k];
            const double ph = -10.0 * p[i] / kLn10 + 0.0;
            if (k) out.push_back(',');
            append_fixed(out, v[i], 3);
            out.push_back('=');
            append_fixed(out, ph, 2);
        }
        o.afd = out;
        o.has_afd = true;
    }
}

}  // namespace

extern "C" {

// diagnostics: the writer's "%.<digits>f" (append_fixed) for one value, NUL-terminated into out[cap] — compared with printf by the tests
int vlr_selftest_format_fixed(double v, int digits, char* out, int cap) {
    std::string s;
    append_fixed(s, v, digits);
    if (!out || cap < (int)s.size() + 1) return VLR_ERR_INVALID_ARGUMENT;
    memcpy(out, s.c_str(), s.size() + 1);
    return VLR_OK;
}

// The calls file (calling.rs:296-304 bcf::Writer, mod.rs:178-600): one record per locus of the table.  `header_text`: the VCF
// header (## lines and #CHROM line with the sample names); out_names[n_out]: names of the columns of ln_posterior
// ("absent", events..., "artifact").  `path` ending in ".bcf" -> BCF2 in BGZF blocks, otherwise text VCF.
static int calls_write_impl(FILE* out_file, bool bcf, bool with_header, bool with_eof, const char* header_text, const vlr_obs_table* t, const vlr_results* r,
                            const char* const* out_names, int n_threads, AsyncSink* sink = nullptr);

int vlr_calls_write(const char* path, const char* header_text, const vlr_obs_table* t, const vlr_results* r, const char* const* out_names, int n_threads) {
    if (!path || !header_text || !t || !r || !out_names) return ifail(VLR_ERR_INVALID_ARGUMENT, "null argument");
    const size_t plen = strlen(path);
    const bool bcf = plen > 4 && strcmp(path + plen - 4, ".bcf") == 0;
    FILE* f = fopen(path, "wb");
    if (!f) return ifail(VLR_ERR_INVALID_ARGUMENT, "cannot create %s", path);
    int rc = calls_write_impl(f, bcf, true, true, header_text, t, r, out_names, n_threads);
    if (fclose(f) != 0 && rc == VLR_OK) rc = ifail(VLR_ERR_INVALID_ARGUMENT, "write failed on %s", path);
    return rc;
}

// The same file written in pieces (the streaming CLI: one append per chunk of records): header with the first append (or at close
// if nothing was appended), BGZF end-of-file marker at close.
struct vlr_calls_writer { FILE* f = nullptr; bool bcf = false, first = true, eof = true; std::string header; std::unique_ptr<AsyncSink> sink; };

int vlr_calls_writer_open(const char* path, const char* header_text, vlr_calls_writer** out) {
    if (!path || !header_text || !out) return ifail(VLR_ERR_INVALID_ARGUMENT, "null argument");
    *out = nullptr;
    FILE* f = fopen(path, "wb");
    if (!f) return ifail(VLR_ERR_INVALID_ARGUMENT, "cannot create %s", path);
    vlr_calls_writer* w = new vlr_calls_writer();
    const size_t plen = strlen(path);
    w->f = f; w->bcf = plen > 4 && strcmp(path + plen - 4, ".bcf") == 0; w->header = header_text;
    if (w->bcf) w->sink.reset(new AsyncSink(f));
    *out = w;
    return VLR_OK;
}
// A part of a calls file written by one of several writers (the shards of a sharded run): BGZF members concatenate, so part k > 0
// carries records only (with_header = 0) and no part but the assembled file an end-of-file member (with_eof = 0);
// vlr_calls_concat_parts puts the parts together.  Before the first append.
int vlr_calls_writer_set_part(vlr_calls_writer* w, int with_header, int with_eof) {
    if (!w || !w->f) return ifail(VLR_ERR_INVALID_ARGUMENT, "null argument");
    if (!w->bcf) return ifail(VLR_ERR_UNSUPPORTED, "parts are written as BCF (BGZF members concatenate; text VCF takes one writer)");
    if (!with_header) w->first = false;
    w->eof = with_eof != 0;
    return VLR_OK;
}
// the parts in order, then the BGZF end-of-file member; the part files are removed
int vlr_calls_concat_parts(const char* path, const char* const* parts, int n_parts) {
    if (!path || !parts || n_parts < 1) return ifail(VLR_ERR_INVALID_ARGUMENT, "null argument");
    FILE* o = fopen(path, "wb");
    if (!o) return ifail(VLR_ERR_INVALID_ARGUMENT, "cannot create %s", path);
    std::vector<uint8_t> buf((size_t)4 << 20);
    bool ok = true;
    for (int k = 0; k < n_parts && ok; ++k) {
        FILE* in = fopen(parts[k], "rb");
        if (!in) { ok = false; break; }
        size_t got;
        while ((got = fread(buf.data(), 1, buf.size(), in)) > 0) ok = ok && fwrite(buf.data(), 1, got, o) == got;
        ok = ok && !ferror(in);
        fclose(in);
    }
    static const uint8_t kEof[28] = {0x1f, 0x8b, 0x08, 0x04, 0, 0, 0, 0, 0, 0xff, 0x06, 0, 0x42, 0x43, 0x02, 0, 0x1b, 0, 0x03, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    ok = ok && fwrite(kEof, 1, sizeof kEof, o) == sizeof kEof;
    ok = (fclose(o) == 0) && ok;
    if (!ok) return ifail(VLR_ERR_INVALID_ARGUMENT, "assembling %s from its parts failed", path);
    for (int k = 0; k < n_parts; ++k) (void)remove(parts[k]);
    return VLR_OK;
}
int vlr_calls_writer_append(vlr_calls_writer* w, const vlr_obs_table* t, const vlr_results* r, const char* const* out_names, int n_threads) {
    if (!w || !w->f || !t || !r || !out_names) return ifail(VLR_ERR_INVALID_ARGUMENT, "null argument");
    const int rc = calls_write_impl(w->f, w->bcf, w->first, false, w->header.c_str(), t, r, out_names, n_threads, w->sink.get());
    w->first = false;
    if (rc == VLR_OK && w->sink) { std::lock_guard<std::mutex> g(w->sink->mu); if (w->sink->failed) return ifail(VLR_ERR_INVALID_ARGUMENT, "write failed"); }
    return rc;
}
int vlr_calls_writer_close(vlr_calls_writer* w) {
    if (!w) return VLR_OK;
    int rc = VLR_OK;
    if (w->f) {
        if (w->sink && !w->sink->finish()) rc = ifail(VLR_ERR_INVALID_ARGUMENT, "write failed");   // (every member handed over is in the file now)
        w->sink.reset();
        if (w->first || (w->bcf && w->eof)) {  // header of an empty file and / or the end-of-file member
            vlr_obs_table* empty = nullptr;
            (void)empty;
            OutHeader h;
            build_out_header(w->header, h);
            std::vector<uint8_t> p0;
            if (w->first) {
                if (w->bcf) { p0.insert(p0.end(), {'B', 'C', 'F', 2, 2}); put_u32(p0, (uint32_t)h.text.size() + 1); p0.insert(p0.end(), h.text.begin(), h.text.end()); p0.push_back(0); }
                else { std::string ht = w->header; if (ht.empty() || ht.back() != '\n') ht.push_back('\n'); p0.assign(ht.begin(), ht.end()); }
            }
            std::string err;
            bool ok = true;
            if (w->bcf) { std::vector<const std::vector<uint8_t>*> ps{&p0}; ok = write_bgzf_stream(w->f, ps, 1, 4, w->eof, err); }
            else if (!p0.empty()) ok = fwrite(p0.data(), 1, p0.size(), w->f) == p0.size();
            if (!ok) rc = ifail(VLR_ERR_INVALID_ARGUMENT, "write failed");
        }
        if (fclose(w->f) != 0 && rc == VLR_OK) rc = ifail(VLR_ERR_INVALID_ARGUMENT, "write failed");
    }
    delete w;
    return rc;
}

static int calls_write_impl(FILE* out_file, bool bcf, bool with_header, bool with_eof, const char* header_text, const vlr_obs_table* t, const vlr_results* r,
                            const char* const* out_names, int n_threads, AsyncSink* sink) {
    if (r->n_loci < t->n_loci || r->n_samples != t->n_samples || !r->ln_posterior || !r->map_vaf || !r->status)
        return ifail(VLR_ERR_INVALID_ARGUMENT, "results do not match the table");
    n_threads = pick_threads(n_threads);
    const int S = t->n_samples, n_out = r->n_out;
    const int64_t L = t->n_loci;
    { const int rcw = const_cast<vlr_obs_table*>(t)->wait_columns(); if (rcw != VLR_OK) return rcw; }   // (a device reader with async columns)
    if (t->has_summary) {   // a pileup with more distinct observation keys than the kernel keeps: count from the columns instead
        bool over = false;
        for (int64_t p = 0; p < L * S && !over; ++p) over = t->sum_hdr[p].overflow != 0;
        if (over) { const int rc = vlr_obs_table_fetch_columns(const_cast<vlr_obs_table*>(t)); if (rc != VLR_OK) return rc; }
    } else if (!t->cols_on_host) {
        const int rc = vlr_obs_table_fetch_columns(const_cast<vlr_obs_table*>(t));
        if (rc != VLR_OK) return rc;
    }
    OutHeader h;
    build_out_header(header_text, h);
    static const char* kFmt[13] = {"DP", "AF", "SAOBS", "SROBS", "OBS", "OOBS", "SB", "ROB", "RPB", "SCB", "HE", "ALB", "AFD"};
    std::vector<int> tag_key((size_t)n_out), fmt_key(13);
    std::vector<std::string> tags((size_t)n_out);
    for (int i = 0; i < n_out; ++i) {
        std::string u = out_names[i];
        for (auto& c : u) c = (char)toupper(c);
        tags[(size_t)i] = "PROB_" + u;
        auto it = h.dict.find(tags[(size_t)i]);
        if (bcf && it == h.dict.end()) return ifail(VLR_ERR_INVALID_ARGUMENT, "INFO key %s not in the header", tags[(size_t)i].c_str());
        tag_key[(size_t)i] = bcf ? it->second : 0;
    }
    for (int k = 0; k < 13; ++k) {
        auto it = h.dict.find(kFmt[k]);
        if (bcf && it == h.dict.end()) return ifail(VLR_ERR_INVALID_ARGUMENT, "FORMAT key %s not in the header", kFmt[k]);
        fmt_key[(size_t)k] = bcf ? it->second : 0;
    }
    std::vector<int> contig_idx(t->contig_names.size(), -1);
    for (size_t i = 0; i < t->contig_names.size(); ++i) {
        auto it = h.contigs.find(t->contig_names[i]);
        if (it != h.contigs.end()) contig_idx[i] = it->second;
    }
    // records are encoded in blocks of 1 024 handed out dynamically (pileups differ in what they cost: contiguous shares per thread
    // left the threads waiting for the slowest one); every block is its own part of the output, in order
    const int T = (int)std::max<int64_t>(1, std::min<int64_t>(n_threads, (L + 63) / 64));
    const int64_t kBlk = 1024, n_blk = (L + kBlk - 1) / kBlk;
    std::vector<std::vector<uint8_t>> parts((size_t)n_blk + 1);
    std::vector<std::string> errs((size_t)std::max<int64_t>(n_blk, 1));
    if (with_header && bcf) {
        auto& p0 = parts[0];
        p0.insert(p0.end(), {'B', 'C', 'F', 2, 2});
        put_u32(p0, (uint32_t)h.text.size() + 1);
        p0.insert(p0.end(), h.text.begin(), h.text.end());
        p0.push_back(0);
    } else if (with_header) {
        std::string ht;  // text output carries the caller's header verbatim (no PASS line added)
        ht = header_text;
        if (ht.empty() || ht.back() != '\n') ht.push_back('\n');
        parts[0].assign(ht.begin(), ht.end());
    }
    const double LN10 = std::log(10.0);
    const double t_w0 = now_s();
    parallel_items(n_blk, T, [&](int64_t w, int) {
        const int64_t b = w * kBlk, e = std::min<int64_t>(L, b + kBlk);
        std::vector<uint8_t>& out = parts[(size_t)w + 1];
        out.reserve((size_t)(e - b) * (size_t)(200 + 260 * S));  // one allocation instead of doubling through tens of megabytes
        std::vector<SampleFields> sf((size_t)S);
        std::vector<uint8_t> shared, indiv;
        std::vector<int> order((size_t)n_out);
        for (int64_t l = b; l < e; ++l) {
            const bool missing = r->status[l] & VLR_LOCUS_MISSING_DATA;
            const double* lp = r->ln_posterior + l * n_out;
            for (int i = 0; i < n_out; ++i) order[(size_t)i] = i;
            // mod.rs:231: sorted by descending probability (NaN last)
            std::stable_sort(order.begin(), order.end(), [&](int a, int c) {
                const double ka = lp[a] == lp[a] ? -lp[a] : INFINITY, kc = lp[c] == lp[c] ? -lp[c] : INFINITY;
                return ka < kc;
            });
            if (!missing) { WPROF(3); for (int s = 0; s < S; ++s) sample_fields(t, r, l, s, sf[(size_t)s]); }
            WPROF(4);
            const char* cid = t->pool.c_str() + t->id_off[(size_t)l];
            const char* ref = t->pool.c_str() + t->ref_off[(size_t)l];
            const char* alt = t->pool.c_str() + t->alt_off[(size_t)l];
            const int ci = t->contig[(size_t)l];
            if (!bcf) {
                std::string line = (ci >= 0 && (size_t)ci < t->contig_names.size()) ? t->contig_names[(size_t)ci] : std::to_string(ci);
                line += "\t" + std::to_string(t->pos[(size_t)l]) + "\t.\t" + ref + "\t" + alt + "\t.\t.\t";
                for (int k = 0; k < n_out; ++k) {
                    const int i = order[(size_t)k];
                    if (k) line.push_back(';');
                    line += tags[(size_t)i] + "=" + (missing ? std::string(".") : fmt_g((float)std::fabs(-10.0 * lp[i] / LN10)));
                }
                line += "\tDP:AF:SAOBS:SROBS:OBS:OOBS:SB:ROB:RPB:SCB:HE:ALB:AFD";
                for (int s = 0; s < S; ++s) {
                    line.push_back('\t');
                    if (missing) { line += "0:.:.:.:.:.:.:.:.:.:.:.:."; continue; }
                    const SampleFields& f = sf[(size_t)s];
                    line += std::to_string(f.dp) + ":" + fmt_g(f.af) + ":" + (f.saobs.empty() ? "." : f.saobs) + ":" + (f.srobs.empty() ? "." : f.srobs) + ":" +
                            (f.obs.empty() ? "." : f.obs) + ":" + std::to_string(f.oobs);
                    for (int k = 0; k < 6; ++k) line += ":" + f.sym[k];
                    line += ":" + f.afd;
                }
                line.push_back('\n');
                out.insert(out.end(), line.begin(), line.end());
                (void)cid;
                continue;
            }
            if (ci < 0 || (size_t)ci >= contig_idx.size() || contig_idx[(size_t)ci] < 0) { errs[(size_t)w] = "contig of record " + std::to_string(l + 1) + " not in the header"; return; }
            shared.clear(); indiv.clear();
            put_str(shared, "", 0);  // ID "."
            put_str(shared, ref, strlen(ref));
            uint32_t n_allele = 1;
            if (strcmp(alt, ".") != 0) {
                const char* a = alt;
                while (true) {
                    const char* comma = strchr(a, ',');
                    put_str(shared, a, comma ? (size_t)(comma - a) : strlen(a));
                    ++n_allele;
                    if (!comma) break;
                    a = comma + 1;
                }
            }
            shared.push_back(0x00);  // FILTER: none
            for (int k = 0; k < n_out; ++k) {
                const int i = order[(size_t)k];
                put_typed_int_scalar(shared, tag_key[(size_t)i]);
                shared.push_back(0x15);
                uint32_t bits = 0x7F800001u;  // missing
                if (!missing && lp[i] == lp[i]) { const float v = (float)std::fabs(-10.0 * lp[i] / LN10); memcpy(&bits, &v, 4); }
                put_u32(shared, bits);
            }
            // FORMAT
            auto fmt_ints = [&](int key, auto get) {
                put_typed_int_scalar(indiv, fmt_key[(size_t)key]);
                int32_t lo = 0, hi = 0;
                for (int s = 0; s < S; ++s) { lo = std::min(lo, get(s)); hi = std::max(hi, get(s)); }
                const int ty = int_type_for(lo, hi);
                put_desc(indiv, 1, ty);
                for (int s = 0; s < S; ++s) {
                    const int32_t v = get(s);
                    if (ty == 1) indiv.push_back((uint8_t)(int8_t)v);
                    else if (ty == 2) { indiv.push_back((uint8_t)(v & 0xff)); indiv.push_back((uint8_t)((v >> 8) & 0xff)); }
                    else put_u32(indiv, (uint32_t)v);
                }
            };
            auto fmt_strs = [&](int key, auto get) {
                put_typed_int_scalar(indiv, fmt_key[(size_t)key]);
                size_t n = 0;
                for (int s = 0; s < S; ++s) n = std::max(n, get(s).size());
                put_desc(indiv, (uint32_t)n, 7);
                for (int s = 0; s < S; ++s) { const std::string& v = get(s); indiv.insert(indiv.end(), v.begin(), v.end()); indiv.insert(indiv.end(), n - v.size(), (uint8_t)0); }
            };
            const std::string dot(".");
            fmt_ints(0, [&](int s) { return missing ? 0 : sf[(size_t)s].dp; });
            put_typed_int_scalar(indiv, fmt_key[1]);
            put_desc(indiv, 1, 5);
            for (int s = 0; s < S; ++s) {
                uint32_t bits = 0x7F800001u;
                if (!missing && sf[(size_t)s].af == sf[(size_t)s].af) memcpy(&bits, &sf[(size_t)s].af, 4);
                put_u32(indiv, bits);
            }
            // (the getters hand out references: an OBS or AFD string is a kilobyte, and every field asks twice)
            fmt_strs(2, [&](int s) -> const std::string& { return (missing || sf[(size_t)s].saobs.empty()) ? dot : sf[(size_t)s].saobs; });
            fmt_strs(3, [&](int s) -> const std::string& { return (missing || sf[(size_t)s].srobs.empty()) ? dot : sf[(size_t)s].srobs; });
            fmt_strs(4, [&](int s) -> const std::string& { return (missing || sf[(size_t)s].obs.empty()) ? dot : sf[(size_t)s].obs; });
            if (missing) fmt_strs(5, [&](int) -> const std::string& { return dot; });  // "." in an Integer field: callsfmt writes the missing value
            else fmt_ints(5, [&](int s) { return sf[(size_t)s].oobs; });
            for (int k = 0; k < 6; ++k) fmt_strs(6 + k, [&](int s) -> const std::string& { return missing ? dot : sf[(size_t)s].sym[k]; });
            fmt_strs(12, [&](int s) -> const std::string& { return missing ? dot : sf[(size_t)s].afd; });
            const uint32_t l_shared = 24 + (uint32_t)shared.size(), l_indiv = (uint32_t)indiv.size();
            put_u32(out, l_shared); put_u32(out, l_indiv);
            put_u32(out, (uint32_t)contig_idx[(size_t)ci]);
            put_u32(out, (uint32_t)(t->pos[(size_t)l] - 1));
            put_u32(out, (uint32_t)strlen(ref));
            put_u32(out, 0x7F800001u);  // QUAL missing
            put_u32(out, (n_allele << 16) | (uint32_t)n_out);
            put_u32(out, (13u << 24) | (uint32_t)S);
            out.insert(out.end(), shared.begin(), shared.end());
            out.insert(out.end(), indiv.begin(), indiv.end());
        }
    });
    for (auto& e : errs)
        if (!e.empty()) return ifail(VLR_ERR_INVALID_ARGUMENT, "%s", e.c_str());
    g_ingest_t[8] = now_s() - t_w0;
    std::string err;
    if (bcf) {
        std::vector<const std::vector<uint8_t>*> ps;
        for (auto& p : parts) ps.push_back(&p);
        const char* lv = getenv("VLR_BGZF_LEVEL");
        // level 1 by default: the calls file is written once and read once; deflate at level 4 was 60 % of the writer's time for 12 %
        // smaller files (VLR_BGZF_LEVEL selects another level)
        if (!write_bgzf_stream(out_file, ps, n_threads, lv ? atoi(lv) : 1, with_eof, err, sink)) return ifail(VLR_ERR_INVALID_ARGUMENT, "%s", err.c_str());
        g_ingest_t[9] = now_s() - t_w0 - g_ingest_t[8];
        g_ingest_t[10] = now_s() - t_w0;
        for (int i = 8; i <= 10; ++i) g_ingest_total[i] += g_ingest_t[i];
        return VLR_OK;
    }
    bool ok = true;
    for (auto& p : parts) ok = ok && fwrite(p.data(), 1, p.size(), out_file) == p.size();
    return ok ? VLR_OK : ifail(VLR_ERR_INVALID_ARGUMENT, "write failed");
}

// write_observations (preprocessing/mod.rs:921-1038) for one sample of a host batch: the observation BCF that `call variants`
// reads.  Used by bench.py --workload cli (synthetic pileups -> files) and by the round-trip tests; alleles are synthesised from
// variant_type / ref_base / alt_base when `sites` is NULL (SNV: the bases; MNV: AC>GT; indel: A>AT; SV: N><DUP>; other: N><METH>).
int vlr_obs_write(const char* path, const vlr_batch* in, int sample, const vlr_obs_sites* sites, const int32_t* third_allele_evidence, int n_threads) {
    if (!path || !in || sample < 0 || sample >= in->n_samples) return ifail(VLR_ERR_INVALID_ARGUMENT, "bad argument");
    n_threads = pick_threads(n_threads);
    const int S = in->n_samples;
    const int64_t L = in->n_loci;
    static const char* kInfo[FD_N_VEC + 1] = {"FRAGMENT_ID", "PROB_MAPPING", "PROB_REF", "PROB_ALT", "PROB_MISSED_ALLELE", "PROB_SAMPLE_ALT", "PROB_DOUBLE_OVERLAP",
                                              "STRAND", "READ_ORIENTATION", "SOFTCLIPPED", "PAIRED", "READ_POSITION", "PROB_HIT_BASE", "IS_MAX_MAPQ", "ALT_LOCUS",
                                              "THIRD_ALLELE_EVIDENCE", "PROB_HOMOPOLYMER_ARTIFACT_OBSERVABLE", "PROB_HOMOPOLYMER_VARIANT_OBSERVABLE", "HOMOPOLYMER_INDEL_LEN"};
    std::string text = "##fileformat=VCFv4.2\n##FILTER=<ID=PASS,Description=\"All filters passed\">\n";
    std::vector<std::string> contigs;
    if (sites) for (int i = 0; i < sites->n_contigs; ++i) contigs.push_back(sites->contig_names[i]);
    else contigs.push_back("1");
    for (auto& c : contigs) text += "##contig=<ID=" + c + ">\n";
    text += "##varlociraptor_observation_format_version=15\n";
    for (int k = 0; k < FD_N_VEC + 1; ++k) text += std::string("##INFO=<ID=") + kInfo[k] + ",Number=.,Type=Integer,Description=\"Varlociraptor observations (binary encoded, meant for internal use only).\">\n";
    text += "##INFO=<ID=IMPRECISE,Number=0,Type=Flag,Description=\"Imprecise structural variation\">\n";
    text += "##INFO=<ID=EVENT,Number=1,Type=String,Description=\"ID of event associated to breakend\">\n";
    text += "##INFO=<ID=MATEID,Number=1,Type=String,Description=\"ID of mate breakend\">\n";
    text += "##INFO=<ID=HETEROZYGOSITY,Number=A,Type=Float,Description=\"PHRED scaled expected heterozygosity\">\n";
    text += "##INFO=<ID=SOMATIC_EFFECTIVE_MUTATION_RATE,Number=A,Type=Float,Description=\"PHRED scaled expected somatic effective mutation rate\">\n";
    text += "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\n";
    const int key0 = 1;  // PASS = 0, then the INFO lines in order
    const int key_imprecise = key0 + FD_N_VEC + 1, key_het = key_imprecise + 3, key_som = key_imprecise + 4;
    const int T = (int)std::max<int64_t>(1, std::min<int64_t>(n_threads, (L + 63) / 64));
    std::vector<std::vector<uint8_t>> parts((size_t)T + 1);
    {
        auto& p0 = parts[0];
        p0.insert(p0.end(), {'B', 'C', 'F', 2, 2});
        put_u32(p0, (uint32_t)text.size() + 1);
        p0.insert(p0.end(), text.begin(), text.end());
        p0.push_back(0);
    }
    parallel_ranges(L, T, [&](int64_t b, int64_t e, int w) {
        std::vector<uint8_t>& out = parts[(size_t)w + 1];
        std::vector<uint8_t> enc, shared;
        std::vector<int32_t> words;
        for (int64_t l = b; l < e; ++l) {
            const uint32_t o0 = in->obs_offset[l * S + sample], o1 = in->obs_offset[l * S + sample + 1];
            const uint32_t n = o1 - o0;
            shared.clear();
            std::string ref, alt, id = "";
            int32_t contig = 0;
            int64_t pos = l + 1;
            if (sites) {
                contig = sites->contig[l]; pos = sites->pos[l];
                ref = sites->strings + sites->ref_offset[l]; alt = sites->strings + sites->alt_offset[l];
                id = sites->strings + sites->id_offset[l];
                if (id == ".") id.clear();
            } else {
                const int vt = in->variant_type ? in->variant_type[l] : VLR_VT_SNV;
                if (vt == VLR_VT_SNV) { ref = std::string(1, (char)(in->ref_base && in->ref_base[l] ? in->ref_base[l] : 'A')); alt = std::string(1, (char)(in->alt_base && in->alt_base[l] ? in->alt_base[l] : 'C')); }
                else if (vt == VLR_VT_MNV) { ref = "AC"; alt = "GT"; }
                else if (vt == VLR_VT_INDEL) { ref = "A"; alt = "AT"; }
                else if (vt == VLR_VT_SV) { ref = "N"; alt = "<DUP>"; }
                else { ref = "N"; alt = "<METH>"; }
            }
            put_str(shared, id.data(), id.size());
            put_str(shared, ref.data(), ref.size());
            uint32_t n_allele = 1;
            if (alt != ".") { put_str(shared, alt.data(), alt.size()); n_allele = 2; }
            shared.push_back(0x00);
            uint32_t n_info = 0;
            auto push = [&](int field_index) {  // push_values: bytes -> LE u16 words as i32 (mod.rs:978-1000)
                if (enc.size() & 1) enc.push_back(0);
                words.resize(enc.size() / 2);
                for (size_t i = 0; i < words.size(); ++i) words[i] = enc[2 * i] | (enc[2 * i + 1] << 8);
                put_typed_int_scalar(shared, key0 + field_index);
                put_int_vec(shared, words.data(), (uint32_t)words.size());
                ++n_info;
            };
            auto u64le = [&](uint64_t v) { for (int i = 0; i < 8; ++i) enc.push_back((uint8_t)(v >> (8 * i))); };
            auto u32le = [&](uint32_t v) { for (int i = 0; i < 4; ++i) enc.push_back((uint8_t)(v >> (8 * i))); };
            auto mini = [&](float v) {  // MiniLogProb::new (utils/mod.rs:455-463)
                const uint16_t hbits = float_to_half(v);
                const double proj = half_to_float(hbits);
                if ((double)v < -10.0 && std::floor(proj) == std::floor((double)v)) { u32le(0); enc.push_back((uint8_t)(hbits & 0xff)); enc.push_back((uint8_t)(hbits >> 8)); }
                else { u32le(1); uint32_t bits; memcpy(&bits, &v, 4); u32le(bits); }
            };
            auto minicol = [&](const float* col, int field_index) { enc.clear(); u64le(n); for (uint32_t i = 0; i < n; ++i) mini(col[o0 + i]); push(field_index); };
            auto enumcol = [&](int field_index, auto get) { enc.clear(); u64le(n); for (uint32_t i = 0; i < n; ++i) u32le(get(in->flags[o0 + i])); push(field_index); };
            auto bitcol = [&](int field_index, uint32_t flag) {  // bv::BitVec<u8>
                enc.clear();
                if (n == 0) { enc.push_back(0); u64le(0); push(field_index); return; }
                enc.push_back(1);
                const uint64_t nblocks = ((uint64_t)n + 7) / 8;
                u64le(nblocks);
                const size_t at = enc.size();
                enc.resize(at + nblocks, 0);
                for (uint32_t i = 0; i < n; ++i)
                    if (in->flags[o0 + i] & flag) enc[at + (i >> 3)] |= (uint8_t)(1u << (i & 7));
                u64le(n);
                push(field_index);
            };
            enc.clear(); u64le(n); for (uint32_t i = 0; i < n; ++i) enc.push_back(0); push(0);  // FRAGMENT_ID: None
            minicol(in->prob_mapping, 1); minicol(in->prob_ref, 2); minicol(in->prob_alt, 3); minicol(in->prob_missed_allele, 4);
            minicol(in->prob_sample_alt, 5); minicol(in->prob_double_overlap, 6);
            enumcol(7, [](uint32_t f) { return (f >> VLR_F_STRAND_SHIFT) & 3u; });
            enumcol(8, [](uint32_t f) { const uint32_t o = (f >> VLR_F_ORIENT_SHIFT) & 3u; return o == VLR_ORIENT_F1R2 ? 0u : o == VLR_ORIENT_F2R1 ? 1u : o == VLR_ORIENT_NONE ? 8u : 2u; });
            bitcol(9, VLR_F_SOFTCLIPPED); bitcol(10, VLR_F_PAIRED);
            enumcol(11, [](uint32_t f) { return (f & VLR_F_READPOS_MAJOR) ? 0u : 1u; });
            minicol(in->prob_hit_base, 12);
            bitcol(13, VLR_F_MAX_MAPQ);
            enumcol(14, [](uint32_t f) { return (f >> VLR_F_ALTLOCUS_SHIFT) & 3u; });
            enc.clear(); u64le(n);
            for (uint32_t i = 0; i < n; ++i) {
                const int32_t tv = third_allele_evidence ? third_allele_evidence[o0 + i] : -1;
                if (tv >= 0) { enc.push_back(1); u32le((uint32_t)tv); } else enc.push_back(0);
            }
            push(15);
            bool any_hp = false;
            if (in->prob_hp_artifact)
                for (uint32_t i = 0; i < n; ++i) any_hp = any_hp || in->prob_hp_artifact[o0 + i] == in->prob_hp_artifact[o0 + i];
            if (any_hp) {  // mod.rs:1018-1034: only if any observation carries homopolymer information
                for (int k = 0; k < 2; ++k) {
                    const float* col = k == 0 ? in->prob_hp_artifact : in->prob_hp_variant;
                    enc.clear(); u64le(n);
                    for (uint32_t i = 0; i < n; ++i) { const float v = col ? col[o0 + i] : NAN; if (v == v) { enc.push_back(1); mini(v); } else enc.push_back(0); }
                    push(16 + k);
                }
                enc.clear(); u64le(n);
                for (uint32_t i = 0; i < n; ++i) {
                    const uint32_t f = in->flags[o0 + i];
                    if (f & VLR_F_HP_LEN_VALID) { enc.push_back(1); enc.push_back((uint8_t)((f >> VLR_F_HP_LEN_SHIFT) & 0xff)); } else enc.push_back(0);
                }
                push(18);
            }
            // without site data the precision of a record is what its bias mask says (check_strand_bias follows is_precise, calling.rs:559)
            const bool imprecise = sites ? (sites->imprecise && sites->imprecise[l]) : (in->locus_flags && !(in->locus_flags[l] & VLR_BIAS_STRAND) && (in->locus_flags[l] & VLR_BIAS_ALTLOCUS));
            if (imprecise) { put_typed_int_scalar(shared, key_imprecise); shared.push_back(0x11); shared.push_back(1); ++n_info; }
            if (sites && sites->heterozygosity_ln && sites->heterozygosity_ln[l] == sites->heterozygosity_ln[l]) {
                const float ph = (float)(-10.0 * sites->heterozygosity_ln[l] / std::log(10.0));
                uint32_t bits; memcpy(&bits, &ph, 4);
                put_typed_int_scalar(shared, key_het); shared.push_back(0x15); put_u32(shared, bits); ++n_info;
            }
            if (sites && sites->somatic_effective_mutation_rate_ln && sites->somatic_effective_mutation_rate_ln[l] == sites->somatic_effective_mutation_rate_ln[l]) {
                const float ph = (float)(-10.0 * sites->somatic_effective_mutation_rate_ln[l] / std::log(10.0));
                uint32_t bits; memcpy(&bits, &ph, 4);
                put_typed_int_scalar(shared, key_som); shared.push_back(0x15); put_u32(shared, bits); ++n_info;
            }
            put_u32(out, 24 + (uint32_t)shared.size()); put_u32(out, 0);
            put_u32(out, (uint32_t)contig); put_u32(out, (uint32_t)(pos - 1)); put_u32(out, (uint32_t)ref.size());
            put_u32(out, 0x7F800001u);
            put_u32(out, (n_allele << 16) | n_info);
            put_u32(out, 0);
            out.insert(out.end(), shared.begin(), shared.end());
        }
    });
    std::vector<const std::vector<uint8_t>*> ps;
    for (auto& p : parts) ps.push_back(&p);
    std::string err;
    // observation files are what `preprocess` writes through rust-htslib's bcf::Writer (preprocessing/mod.rs:921-1038): BGZF at htslib's
    // default level 6 (bgzf.c: Z_DEFAULT_COMPRESSION -> 6).  Until the middle of round 5 this writer used level 1 like the calls writer —
    // 11 % larger files with a third more symbols per member than the reader meets in files of the reference.  VLR_OBS_BGZF_LEVEL: other level.
    const char* lv = getenv("VLR_OBS_BGZF_LEVEL");
    if (!write_bgzf_file(path, ps, n_threads, lv ? atoi(lv) : 6, err)) return ifail(VLR_ERR_INVALID_ARGUMENT, "%s", err.c_str());
    return VLR_OK;
}

}  // extern "C"

// ================================================================================================ filter-calls control-fdr
// The whole of `varlociraptor filter-calls control-fdr` behind the ABI (reference src/filtration/fdr.rs:36-158,
// src/utils/mod.rs:169-374, record typing src/utils/collect_variants.rs:44-304): calls BCF in -> kept records out, the threshold
// search on the device (vlr_fdr_threshold).  The Python restatement the tests compare with: varlociraptor_amd/fdr.py.
extern "C" int vlr_fdr_threshold(int device, const double* ln_prob, int64_t n, int smart, double alpha_ln, double* threshold, int* status);
namespace {
constexpr double kNumericalEpsilon = 1e-3;  // utils/mod.rs:40
struct CallRec {
    const uint8_t* raw = nullptr;
    size_t raw_len = 0;
    int64_t pos0 = 0;
    std::string ref, event, svtype;
    std::vector<std::string> alts;
    bool has_event = false, has_svlen = false, has_end = false;
    std::vector<int64_t> svlen;      // |SVLEN| per entry, -1 = missing
    int64_t end0 = 0;                // END - 1
    std::vector<std::vector<double>> prob;  // per wanted tag: PHRED values per ALT (NaN = missing), empty = tag absent
};
struct VarType { int kind = -1; int64_t len = 0; };  // kind < 0: skipped by collect_variants
enum { VK_SNV, VK_MNV, VK_INS, VK_DEL, VK_BND, VK_INV, VK_DUP, VK_REP, VK_REF, VK_METH };
int vk_of(const std::string& s) {
    static const char* const n[] = {"SNV", "MNV", "INS", "DEL", "BND", "INV", "DUP", "REP", "REF", "METH"};
    for (int i = 0; i < 10; ++i) if (s == n[i]) return i;
    return -1;
}
bool valid_del(const std::string& r, const std::string& a) { return a == "<DEL>" || (r.size() > a.size() && r.compare(0, a.size(), a) == 0 && a.size() == 1); }
bool valid_ins(const std::string& r, const std::string& a) { return a == "<INS>" || (r.size() < a.size() && a.compare(0, r.size(), r) == 0 && r.size() == 1); }
// (type, length) per ALT allele as collect_variants types it (collect_variants.rs:44-304)
bool variant_types(const CallRec& r, std::vector<VarType>& out, std::string& err) {
    out.clear();
    if (!r.svtype.empty()) {
        const std::string& sv = r.svtype;
        if (sv == "INV" || sv == "DUP") {
            VarType v;
            if (r.alts.size() == 1 && r.has_end) { v.kind = sv == "INV" ? VK_INV : VK_DUP; v.len = r.end0 + 1 - r.pos0; }
            out.push_back(v);
        } else if (sv == "BND") {
            for (size_t i = 0; i < r.alts.size(); ++i) out.push_back(VarType{VK_BND, 0});
        } else if (sv == "INS") {
            VarType v;
            if (!r.alts.empty() && r.alts[0] != "<INS>" && valid_ins(r.ref, r.alts[0])) { v.kind = VK_INS; v.len = (int64_t)r.alts[0].size() - (int64_t)r.ref.size(); }
            out.push_back(v);
        } else if (sv == "DEL") {
            int64_t svlen;
            if (r.has_svlen && !r.svlen.empty() && r.svlen[0] >= 0) svlen = r.svlen[0];
            else if (!r.has_svlen && r.has_end) svlen = r.end0 - (r.pos0 + 1);  // collect_variants.rs:196-199
            else { err = "missing SVLEN or END"; return false; }
            VarType v;
            if (!r.alts.empty() && valid_del(r.ref, r.alts[0])) { v.kind = VK_DEL; v.len = svlen; }
            out.push_back(v);
        }
        return true;
    }
    for (size_t i = 0; i < r.alts.size(); ++i) {
        const std::string& a = r.alts[i];
        VarType v;
        if (a == "<*>") v = VarType{VK_REF, 0};
        else if (a == "<DEL>") { if (r.has_svlen && i < r.svlen.size() && r.svlen[i] >= 0) v = VarType{VK_DEL, r.svlen[i]}; }
        else if (a == "<METH>") v = VarType{VK_METH, 0};
        else if (!a.empty() && a[0] == '<') {}
        else if (a.size() == 1 && r.ref.size() == 1) v = VarType{VK_SNV, 1};
        else if (a.size() == r.ref.size()) v = VarType{VK_MNV, (int64_t)a.size()};
        else if (valid_del(r.ref, a)) v = VarType{VK_DEL, (int64_t)r.ref.size() - (int64_t)a.size()};
        else if (valid_ins(r.ref, a)) v = VarType{VK_INS, (int64_t)a.size() - (int64_t)r.ref.size()};
        else v = VarType{VK_REP, 0};
        out.push_back(v);
    }
    return true;
}
struct TypeFilter { int kind = -1; bool has_range = false; int64_t lo = 0, hi = 0; };
bool is_type(const VarType& v, const TypeFilter& f) {  // Variant::is_type (variants/model/mod.rs)
    if (v.kind < 0) return false;
    if (f.kind < 0) return true;
    if (v.kind != f.kind) return false;
    return !f.has_range || (f.lo <= v.len && v.len < f.hi);
}
double lse(const std::vector<double>& v) {  // bio LogProb::ln_sum_exp: the maximum apart, ln_1p of the others
    if (v.empty()) return -INFINITY;
    size_t im = 0;
    for (size_t i = 1; i < v.size(); ++i)
        if (v[i] > v[im]) im = i;
    const double m = v[im];
    if (m == -INFINITY) return m;
    double s = 0.0;
    for (size_t i = 0; i < v.size(); ++i)
        if (i != im && v[i] != -INFINITY) s += std::exp(v[i] - m);
    return m + std::log1p(s);
}
// utils/mod.rs:177-212: ln-sum over the given PROB_* tags per (typed) variant; NaN = None
void tags_prob_sum(const CallRec& r, const std::vector<VarType>& types, const std::vector<int>& tags, const TypeFilter& tf, std::vector<double>& out) {
    std::vector<const VarType*> variants;
    for (auto& v : types) if (v.kind >= 0) variants.push_back(&v);
    std::vector<std::vector<double>> acc(variants.size());
    for (int t : tags) {
        const std::vector<double>& vals = r.prob[(size_t)t];
        for (size_t i = 0; i < variants.size() && i < vals.size(); ++i) {
            const double p = vals[i];
            if (p != p || !is_type(*variants[i], tf)) continue;
            acc[i].push_back(-p * std::log(10.0) / 10.0);
        }
    }
    out.clear();
    for (auto& probs : acc) {
        if (probs.empty()) { out.push_back(NAN); continue; }
        double s = lse(probs);
        if (0.0 < s && s <= kNumericalEpsilon) s = 0.0;  // cap_numerical_overshoot
        out.push_back(s);
    }
}
double ln_one_minus_exp(double p) { return p < 0.0 ? (p < -0.693 ? std::log1p(-std::exp(p)) : std::log(-std::expm1(p))) : -INFINITY; }
}  // namespace

extern "C" int vlr_calls_filter_fdr(const char* in_path, const char* out_path, int n_events, const char* const* events, double alpha, uint32_t mode,
                                    const char* vartype, int64_t minlen, int64_t maxlen, int device, int n_threads, int64_t* n_kept, int64_t* n_total) {
    if (!in_path || !out_path || n_events <= 0 || !events) return ifail(VLR_ERR_INVALID_ARGUMENT, "null argument");
    if (!(alpha > 0.0)) return ifail(VLR_ERR_INVALID_ARGUMENT, "alpha must be positive");
    const bool local = (mode & VLR_FDR_MODE_LOCAL) != 0, smart = (mode & VLR_FDR_MODE_SMART) != 0, retain = (mode & VLR_FDR_MODE_RETAIN_ARTIFACTS) != 0;
    n_threads = pick_threads(n_threads);
    TypeFilter tf;
    if (vartype && *vartype) {
        tf.kind = vk_of(vartype);
        if (tf.kind < 0) return ifail(VLR_ERR_INVALID_ARGUMENT, "unknown variant type %s", vartype);
        if (minlen >= 0 || maxlen >= 0) { tf.has_range = true; tf.lo = minlen >= 0 ? minlen : 0; tf.hi = maxlen >= 0 ? maxlen : ((int64_t)1 << 62); }
    }
    std::string err;
    Blob data;
    if (!load_inflated(in_path, data, n_threads, err)) return ifail(VLR_ERR_INVALID_ARGUMENT, "%s", err.c_str());
    if (data.size() < 9 || memcmp(data.data(), "BCF\2\2", 5) != 0) return ifail(VLR_ERR_INVALID_ARGUMENT, "%s is not a BCF file", in_path);
    uint32_t l_text;
    memcpy(&l_text, data.data() + 5, 4);
    if (9 + (size_t)l_text > data.size()) return ifail(VLR_ERR_INVALID_ARGUMENT, "truncated BCF header in %s", in_path);
    Header h;
    {
        std::string text((const char*)data.data() + 9, l_text);
        while (!text.empty() && text.back() == '\0') text.pop_back();
        parse_header(text, h);
    }
    // wanted INFO tags: the events' PROB_* (those the header declares, fdr.rs:46-60), PROB_ABSENT, PROB_ARTIFACT
    std::vector<std::string> tag_names;
    auto tag_id = [&](const std::string& name) {
        for (size_t i = 0; i < tag_names.size(); ++i) if (tag_names[i] == name) return (int)i;
        tag_names.push_back(name);
        return (int)tag_names.size() - 1;
    };
    auto declared = [&](const std::string& name) { return std::find(h.dict.begin(), h.dict.end(), name) != h.dict.end() && h.text.find("##INFO=<ID=" + name + ",") != std::string::npos; };
    std::vector<int> ev_tags;
    for (int i = 0; i < n_events; ++i) {
        std::string t = "PROB_";
        for (const char* q = events[i]; *q; ++q) t.push_back((char)toupper((unsigned char)*q));
        if (declared(t)) ev_tags.push_back(tag_id(t));
    }
    if (ev_tags.empty()) return ifail(VLR_ERR_INVALID_ARGUMENT, "invalid FDR control events");  // errors::Error::InvalidFDRControlEvents
    const int t_absent = tag_id("PROB_ABSENT"), t_artifact = tag_id("PROB_ARTIFACT");
    std::vector<int> key_tag(h.dict.size(), -1);  // dictionary index -> 0.. wanted tag, -2 EVENT, -3 SVTYPE, -4 SVLEN, -5 END
    for (size_t k = 0; k < h.dict.size(); ++k) {
        for (size_t t = 0; t < tag_names.size(); ++t) if (h.dict[k] == tag_names[t]) key_tag[k] = (int)t;
        if (h.dict[k] == "EVENT") key_tag[k] = -2;
        else if (h.dict[k] == "SVTYPE") key_tag[k] = -3;
        else if (h.dict[k] == "SVLEN") key_tag[k] = -4;
        else if (h.dict[k] == "END") key_tag[k] = -5;
    }
    // records
    std::vector<CallRec> recs;
    {
        const uint8_t* p = data.data() + 9 + l_text;
        const uint8_t* const e = data.data() + data.size();
        while (p + 8 <= e) {
            uint32_t ls, li;
            memcpy(&ls, p, 4); memcpy(&li, p + 4, 4);
            const size_t len = 8 + (size_t)ls + li;
            if ((size_t)(e - p) < len || ls < 24) return ifail(VLR_ERR_INVALID_ARGUMENT, "truncated BCF record in %s", in_path);
            CallRec r;
            r.raw = p; r.raw_len = len;
            r.prob.resize(tag_names.size());
            const uint8_t* q = p + 8;
            const uint8_t* se = q + ls;
            int32_t pos;
            uint32_t nai;
            memcpy(&pos, q + 4, 4); memcpy(&nai, q + 16, 4);
            q += 24;
            r.pos0 = pos;
            const uint32_t n_allele = nai >> 16, n_info = nai & 0xffff;
            Typed t;
            if (!bcf_typed(q, se, t)) return ifail(VLR_ERR_INVALID_ARGUMENT, "bad ID field in %s", in_path);
            for (uint32_t a = 0; a < n_allele; ++a) {
                if (!bcf_typed(q, se, t) || (t.type != 7 && t.n != 0)) return ifail(VLR_ERR_INVALID_ARGUMENT, "bad allele in %s", in_path);
                std::string s((const char*)t.data, t.n);
                if (a == 0) r.ref = s; else r.alts.push_back(s);
            }
            if (!bcf_typed(q, se, t)) return ifail(VLR_ERR_INVALID_ARGUMENT, "bad FILTER field in %s", in_path);
            for (uint32_t k = 0; k < n_info; ++k) {
                Typed key, val;
                if (!bcf_typed(q, se, key) || key.n != 1 || key.type < 1 || key.type > 3 || !bcf_typed(q, se, val)) return ifail(VLR_ERR_INVALID_ARGUMENT, "bad INFO field in %s", in_path);
                const int32_t ki = typed_int(key, 0);
                const int kt = (ki >= 0 && (size_t)ki < key_tag.size()) ? key_tag[(size_t)ki] : -1;
                if (kt >= 0 && val.type == 5) {
                    auto& v = r.prob[(size_t)kt];
                    for (uint32_t i = 0; i < val.n; ++i) {
                        uint32_t bits;
                        memcpy(&bits, val.data + 4 * (size_t)i, 4);
                        if (bits == 0x7F800002u) break;  // end of vector
                        float x;
                        memcpy(&x, &bits, 4);
                        v.push_back(bits == 0x7F800001u ? NAN : (double)x);
                    }
                    if (v.empty()) v.push_back(NAN);  // (a present tag is a list: keeps "tag absent" apart)
                } else if (kt == -2 && val.type == 7) {
                    r.event.assign((const char*)val.data, val.n);
                    while (!r.event.empty() && r.event.back() == '\0') r.event.pop_back();
                    r.has_event = true;
                } else if (kt == -3 && val.type == 7) {
                    r.svtype.assign((const char*)val.data, val.n);
                    while (!r.svtype.empty() && r.svtype.back() == '\0') r.svtype.pop_back();
                } else if ((kt == -4 || kt == -5) && val.type >= 1 && val.type <= 3) {
                    static const int32_t miss[4] = {0, -128, -32768, (int32_t)0x80000000}, eov[4] = {0, -127, -32767, (int32_t)0x80000001};
                    std::vector<int64_t> vals;
                    for (uint32_t i = 0; i < val.n; ++i) {
                        const int32_t x = typed_int(val, i);
                        if (x == eov[val.type]) continue;
                        vals.push_back(x == miss[val.type] ? -1 : (kt == -4 ? std::llabs((long long)x) : (int64_t)x));
                    }
                    if (kt == -4) { r.has_svlen = true; r.svlen = vals; }
                    else if (!vals.empty()) { r.has_end = true; r.end0 = vals[0] - 1; }
                }
            }
            recs.push_back(std::move(r));
            p += len;
        }
    }
    const int64_t N = (int64_t)recs.size();
    std::vector<std::vector<VarType>> types((size_t)N);
    for (int64_t i = 0; i < N; ++i)
        if (!variant_types(recs[(size_t)i], types[(size_t)i], err)) return ifail(VLR_ERR_INVALID_ARGUMENT, "%s (record %lld of %s)", err.c_str(), (long long)i, in_path);
    // threshold (fdr.rs:62-141)
    const double alpha_ln = std::log(alpha);
    bool have_thr = false;
    double thr = 0.0;
    if (local) { have_thr = true; thr = alpha < 1.0 ? std::log1p(-alpha) : -INFINITY; }
    else if (alpha != 1.0) {
        std::vector<int> dist_tags;
        if (smart) { dist_tags.push_back(t_absent); if (!retain) dist_tags.push_back(t_artifact); }
        else dist_tags = ev_tags;
        std::vector<double> dist, sums;
        std::unordered_set<std::string> seen;
        for (int64_t i = 0; i < N; ++i) {  // utils/mod.rs:236-270: one entry per breakend event
            const CallRec& r = recs[(size_t)i];
            if (r.has_event) { if (seen.count(r.event)) continue; seen.insert(r.event); }
            tags_prob_sum(r, types[(size_t)i], dist_tags, tf, sums);
            for (double s : sums) if (s == s) dist.push_back(s);
        }
        int status = 0;
        const int rc = vlr_fdr_threshold(device, dist.data(), (int64_t)dist.size(), smart ? 1 : 0, alpha_ln, &thr, &status);
        if (rc != 0) return rc;
        if (status == VLR_FDR_VALUE) have_thr = true;
        else if (status == VLR_FDR_LN_ONE) { have_thr = true; thr = 0.0; }
    }
    // filter_by_threshold (utils/mod.rs:288-374)
    std::vector<int> ftags = ev_tags, absent_tags{t_absent};
    if (smart && retain) ftags.push_back(t_artifact); else absent_tags.push_back(t_artifact);
    std::unordered_map<std::string, bool> decisions;
    std::vector<uint8_t> header_bytes(data.data(), data.data() + 9 + l_text);
    std::vector<uint8_t> body;
    int64_t kept = 0;
    std::vector<double> pe, pa;
    for (int64_t i = 0; i < N; ++i) {
        const CallRec& r = recs[(size_t)i];
        tags_prob_sum(r, types[(size_t)i], ftags, tf, pe);
        if (smart) tags_prob_sum(r, types[(size_t)i], absent_tags, tf, pa);
        bool keep_any = false;
        for (size_t v = 0; v < pe.size(); ++v) {
            bool keep;
            auto it = r.has_event ? decisions.find(r.event) : decisions.end();
            if (r.has_event && it != decisions.end()) keep = it->second;
            else {
                const double prob_events = pe[v];
                double p = prob_events;
                if (smart) { const double a = v < pa.size() ? pa[v] : NAN; p = (a == a) ? ln_one_minus_exp(a) : NAN; }
                if (p == p && have_thr) keep = p > thr || relative_eq(p, thr);
                else if (p == p) keep = true;
                else keep = false;
                if (smart) keep = keep && (prob_events == prob_events && prob_events > std::log(0.5));
                if (r.has_event) decisions[r.event] = keep;
            }
            keep_any = keep_any || keep;
        }
        if (keep_any) {
            // The reference removes the alleles that did not pass (filter_calls, utils/mod.rs:382-417: record.remove_alleles) and
            // asserts one filter decision per ALT.  This entry writes kept records byte for byte, which is the same thing for the
            // single-ALT records `call variants` emits; anything else is refused rather than emitted untrimmed.
            if (r.alts.size() > 1 || pe.size() != r.alts.size())
                return ifail(VLR_ERR_UNSUPPORTED, "record %lld of %s has %zu ALT alleles (%zu typed variants): multi-allelic calls files are not supported by control-fdr here",
                             (long long)i, in_path, r.alts.size(), pe.size());
            body.insert(body.end(), r.raw, r.raw + r.raw_len); ++kept;
        }
    }
    if (!write_bgzf_file(out_path, {&header_bytes, &body}, n_threads, 6, err)) return ifail(VLR_ERR_INVALID_ARGUMENT, "%s", err.c_str());
    if (n_kept) *n_kept = kept;
    if (n_total) *n_total = N;
    return VLR_OK;
}
