// bigsi_bdb.hpp -- BerkeleyDB HASH files read without libdb (header-only host C++; no HIP): the on-disk format of the reference's
// default backend (`db.DB().open(filename, None, db.DB_HASH, db.DB_CREATE)`, bigsi/storage/berkeleydb.py:12-19).  Used by
// libbigsi_hip.so (bigsi_hip_load_rows_file on a store, bigsi_hip_bdb_small_records) and by libbigsi_cpu.so (bigsi_cpu_open_bdb: the
// CPU baseline with its rows served from such a file).  Layout per Berkeley DB's public db_page.h (hash versions 7-10): a 26-byte
// page header, item offsets growing up from it, items growing down from the page end; item type H_KEYDATA (1) = inline bytes,
// H_OFFPAGE (3) = {first page, length} of an overflow chain (every row of an index with more than a few thousand samples).  Every
// key / data pair lives on exactly one hash page, so one scan of all pages finds each record once.  bigsi_amd/bdb.py is the same
// reader in Python: the definition of the format here and this reader's test oracle (tests/test_bdb_reader.py).
// Errors: 0 or an errno-style code; `error` holds the message.
#pragma once
#include <atomic>
#include <cerrno>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <sys/stat.h>
#include <thread>
#include <unistd.h>
#include <vector>

static inline uint64_t bigsi_bdb_ceil_div(uint64_t x, uint64_t a) { return (x + a - 1) / a; }

struct BigsiBdb {
    struct Loc {               // where a record's value is
        uint64_t at = 0;       // inline: byte offset in the file; overflow: first page of the chain
        uint32_t len = 0;
        uint8_t kind = 0;      // 0 absent, 1 inline, 3 overflow chain
    };
    int fd = -1;
    bool swap = false;         // the file's byte order is not the host's
    uint32_t pagesize = 0;
    uint64_t n_pages = 0;
    std::string error;
    static bool is_bdb(int fd);
    int open_fd(int fd_);                                                      // 0, or EINVAL with `error` set
    // one pass over all hash pages with `threads` threads; on_item(thread, key, key_len, loc) for every record whose key is inline
    // (overflow keys -- longer than a page -- are no index records and are skipped)
    template <typename F> int scan(unsigned threads, F on_item);
    int read_value(const Loc &l, uint8_t *dst, uint32_t want, std::vector<uint8_t> &page) const;      // first `want` bytes; 0 or errno
    uint16_t u16(const uint8_t *p) const { uint16_t v; memcpy(&v, p, 2); return swap ? (uint16_t)((v >> 8) | (v << 8)) : v; }
    uint32_t u32(const uint8_t *p) const { uint32_t v; memcpy(&v, p, 4); return swap ? __builtin_bswap32(v) : v; }
    int set_error(const char *fmt, ...) __attribute__((format(printf, 2, 3)))
    {
        char buf[512];
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(buf, sizeof buf, fmt, ap);
        va_end(ap);
        error = buf;
        return EINVAL;
    }
};

static constexpr uint32_t kBdbHashMagic = 0x061561;
enum { kBdbPageOverflow = 7, kBdbPageHashMeta = 8, kBdbPageHash = 13, kBdbPageHashUnsorted = 2, kBdbHdr = 26, kBdbKeyData = 1, kBdbOffPage = 3 };

inline bool BigsiBdb::is_bdb(int fd)
{
    uint8_t head[32];
    if (pread(fd, head, sizeof head, 0) != (ssize_t)sizeof head) return false;
    uint32_t magic;
    memcpy(&magic, head + 12, 4);
    return (magic == kBdbHashMagic || __builtin_bswap32(magic) == kBdbHashMagic) && head[25] == kBdbPageHashMeta;
}

inline int BigsiBdb::open_fd(int fd_)
{
    fd = fd_;
    uint8_t head[72];
    if (pread(fd, head, sizeof head, 0) != (ssize_t)sizeof head) return set_error("too short for a BerkeleyDB file");
    uint32_t magic;
    memcpy(&magic, head + 12, 4);
    swap = magic != kBdbHashMagic;
    if (u32(head + 12) != kBdbHashMagic) return set_error("not a BerkeleyDB hash file");
    pagesize = u32(head + 20);
    if (head[24] != 0) return set_error("encrypted BerkeleyDB files are not supported");
    if (pagesize < 512 || pagesize > 65536 || (pagesize & (pagesize - 1))) return set_error("BerkeleyDB page size %u", pagesize);
    struct stat sb;
    if (fstat(fd, &sb) != 0) return set_error("fstat: %s", strerror(errno));
    n_pages = (uint64_t)sb.st_size / pagesize;
    return 0;
}

template <typename F> int BigsiBdb::scan(unsigned threads, F on_item)
{
    const uint64_t pages_per_read = std::max<uint64_t>(1, (4ull << 20) / pagesize);
    const uint64_t n_blocks = bigsi_bdb_ceil_div(n_pages, pages_per_read);
    threads = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>(threads, n_blocks));
    std::atomic<uint64_t> next{0};
    std::atomic<int> err{0};
    auto work = [&](unsigned tid) {
        std::vector<uint8_t> buf(pages_per_read * pagesize);
        for (;;) {
            const uint64_t b = next.fetch_add(1);
            if (b >= n_blocks || err.load()) return;
            const uint64_t p0 = b * pages_per_read, np = std::min(pages_per_read, n_pages - p0);
            uint64_t got = 0;
            while (got < np * pagesize) {
                const ssize_t r = pread(fd, buf.data() + got, (size_t)(np * pagesize - got), (off_t)(p0 * pagesize + got));
                if (r < 0) { if (errno == EINTR) continue; err.store(errno); return; }
                if (r == 0) { err.store(ENODATA); return; }
                got += (uint64_t)r;
            }
            for (uint64_t i = 0; i < np; i++) {
                const uint64_t pgno = p0 + i;
                if (pgno == 0) continue;
                const uint8_t *p = buf.data() + i * pagesize;
                if (p[25] != kBdbPageHash && p[25] != kBdbPageHashUnsorted) continue;
                const uint32_t n = u16(p + 20);
                if (n == 0 || kBdbHdr + 2ull * n > pagesize) continue;
                uint32_t end = pagesize;
                for (uint32_t it = 0; it + 1 < n; it += 2) {
                    const uint32_t ks = u16(p + kBdbHdr + 2 * it), vs = u16(p + kBdbHdr + 2 * (it + 1));
                    const uint32_t kend = end, vend = ks;
                    end = vs;
                    if (ks >= kend || vs >= vend || kend > pagesize) { err.store(EILSEQ); return; }
                    if (p[ks] != kBdbKeyData) continue;               // an overflow key (longer than a page) is no index record
                    Loc l;
                    if (p[vs] == kBdbKeyData) { l.kind = 1; l.at = pgno * pagesize + vs + 1; l.len = vend - vs - 1; }
                    else if (p[vs] == kBdbOffPage) { if (vs + 12 > vend) { err.store(EILSEQ); return; } l.kind = 3; l.at = u32(p + vs + 4); l.len = u32(p + vs + 8); }
                    else { err.store(ENOTSUP); return; }            // duplicate sets do not occur in BIGSI stores
                    on_item(tid, p + ks + 1, kend - ks - 1, l);
                }
            }
        }
    };
    if (threads == 1) work(0);
    else {
        std::vector<std::thread> pool;
        for (unsigned t = 0; t < threads; t++) pool.emplace_back(work, t);
        for (auto &th : pool) th.join();
    }
    const int e = err.load();
    if (e) return set_error("BerkeleyDB file: %s", e == EILSEQ ? "corrupt hash page" : e == ENOTSUP ? "duplicate items are not supported" : e == ENODATA ? "file shorter than its page count" : strerror(e));
    return 0;
}

inline int BigsiBdb::read_value(const Loc &l, uint8_t *dst, uint32_t want, std::vector<uint8_t> &page) const
{
    want = std::min(want, l.len);
    if (l.kind == 1) {
        uint32_t got = 0;
        while (got < want) {
            const ssize_t r = pread(fd, dst + got, want - got, (off_t)(l.at + got));
            if (r < 0) { if (errno == EINTR) continue; return errno; }
            if (r == 0) return ENODATA;
            got += (uint32_t)r;
        }
        return 0;
    }
    // an overflow chain: every page carries hf_offset (bytes 22-23) bytes after the header, next_pgno (bytes 16-19) links the chain
    // libdb allocates the pages of a chain one after the other when it writes a large value into a growing file: a window of up to
    // 16 pages is read at once and walked for as long as next_pgno is the page that follows (one system call per 64 KB instead of one
    // per 4 KB page); a chain that jumps simply starts a new window
    const uint32_t kWindow = 16;
    page.resize((size_t)kWindow * pagesize);
    uint64_t pgno = l.at;
    uint32_t got = 0;
    while (got < want) {
        if (pgno == 0 || pgno >= n_pages) return EILSEQ;
        const uint32_t per_page = pagesize - kBdbHdr;
        const uint64_t pages_left = bigsi_bdb_ceil_div(want - got, per_page);
        const uint32_t win = (uint32_t)std::min<uint64_t>(std::min<uint64_t>(kWindow, pages_left), n_pages - pgno);
        // (of the last page only as much as is needed)
        const uint64_t need = (uint64_t)(win - 1) * pagesize + std::min<uint64_t>(pagesize, kBdbHdr + ((uint64_t)(want - got) - std::min<uint64_t>(want - got, (uint64_t)(win - 1) * per_page)));
        uint64_t have = 0;
        while (have < need) {
            const ssize_t r = pread(fd, page.data() + have, (size_t)(need - have), (off_t)(pgno * pagesize + have));
            if (r < 0) { if (errno == EINTR) continue; return errno; }
            if (r == 0) return ENODATA;
            have += (uint64_t)r;
        }
        for (uint32_t i = 0; i < win && got < want; i++) {
            const uint8_t *pg = page.data() + (size_t)i * pagesize;
            if (pg[25] != kBdbPageOverflow) return EILSEQ;
            const uint32_t used = std::min<uint32_t>(u16(pg + 22), per_page), take = std::min(used, want - got);
            if (take == 0) return EILSEQ;
            if ((uint64_t)i * pagesize + kBdbHdr + take > need) { pgno += i; goto next_window; }      // (a short page inside the window: the tail was not read)
            memcpy(dst + got, pg + kBdbHdr, take);
            got += take;
            const uint64_t nxt = u32(pg + 16);
            if (got < want && nxt != pgno + i + 1) { pgno = nxt; goto next_window; }
            if (i + 1 == win) pgno = nxt;
        }
    next_window:;
    }
    return 0;
}

// "<digits>:bitarray" -> row id (bigsi/storage/base.py:29-36); false for any other key
inline bool bigsi_bdb_row_key(const uint8_t *key, uint32_t len, uint64_t *row)
{
    static const char tail[] = ":bitarray";
    if (len < 10 || len > 29 || memcmp(key + len - 9, tail, 9) != 0) return false;
    uint64_t r = 0;
    for (uint32_t i = 0; i + 9 < len; i++) {
        if (key[i] < '0' || key[i] > '9') return false;
        r = r * 10 + (key[i] - '0');
    }
    *row = r;
    return true;
}

