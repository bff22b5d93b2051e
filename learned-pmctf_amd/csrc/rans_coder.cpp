// rans_coder.cpp — host range coder of the pMCTF encode path (C ABI: include/pmctf_rans.h).
//
// Produces byte-identical streams to the reference's RansEncoder (pMCTF/cpp/rans/rans.cpp,
// pMCTF/cpp/py_rans/py_rans.cpp) built on the 64-bit rANS of rygorous/ryg_rans (rans64.h): state in
// [2^31, 2^63), renormalisation by 32-bit little-endian words, symbols pushed in coding order and
// folded into the state in reverse at flush so the decoder reads forward.  Out-of-table values use
// the reference's escape: the row's last symbol as sentinel followed by 4-bit bypass digits.
//
// Layout choices (not the reference's): a push only records WHERE its symbols are (the int16 arrays are copied, or
// borrowed when the caller guarantees their lifetime, pmctf_rans_encoder_set_borrow) — nothing is expanded into
// per-symbol records.  flush() walks the pushes backwards ONCE, mapping (symbol, CDF row) to its interval on the
// fly and folding it into the state: a full-frame stream (~9 M symbols at 1080p) costs one sequential read of
// 4 bytes per symbol.  Division by the interval width is a multiply with a per-interval reciprocal (exact for the
// states that occur, x < 2^63); the state update is x + q * (2^16 - range) + start, which equals
// ((x / range) << 16) + x % range + start.  The (CDF row 0, symbol 0) interval — three quarters of a four-step
// subband stream are such off-mask positions — has its constants in registers.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <new>
#include <thread>
#include <vector>

#include "../../include/pmctf_rans.h"

namespace {

constexpr uint32_t kPrecision = 16;
constexpr uint32_t kBypassPrecision = 4;
constexpr uint32_t kMaxBypassVal = (1u << kBypassPrecision) - 1;
constexpr uint64_t kRansL = 1ull << 31;
// Rans64EncPutBits(.., 4): renormalise when x >= ((L >> 16) << 32) << (16 - 4)
constexpr uint64_t kBypassXMax = ((kRansL >> 16) << 32) * (uint64_t)(1u << (16 - kBypassPrecision));

// One CDF interval with the constants of its state update.  q = x / range for x < 2^63:
// q = mulhi(x, magic) >> sh with l = ceil(log2 range), magic = ceil(2^(63+l) / range) (< 2^64), sh = l - 1
// (error of the product < 2^-l <= 1/range, so the floor is exact); range == 1: q = x.
struct Entry {
    uint64_t magic;
    uint64_t x_max;    // ((L >> 16) << 32) * range: renormalise first when x >= x_max
    uint32_t start;
    uint32_t cmpl;     // 2^16 - range
    uint32_t sh;
    uint32_t range;
};

struct Table {          // a registered (cdfs, sizes, offsets) triple; entries[row * cols + value]
    int rows, cols;
    uint64_t checksum;
    uint32_t base;      // first index of this table in the encoder's flat entry array
    std::vector<int32_t> max_value, offset;     // per row: sizes[row] - 2, offsets[row]
};

struct Push {           // one encode_with_indexes call
    const int16_t *sym, *idx;
    int64_t n;
    int table;
    std::unique_ptr<int16_t[]> owned;           // copy of both arrays unless borrowed
};

struct Part {
    std::vector<uint8_t> stream;
    std::unique_ptr<uint32_t[]> words;          // scratch of flush(), grown on demand, never initialised
    size_t cap = 0;
};

inline Entry make_entry(uint32_t start, uint32_t range) {
    Entry e;
    e.start = start; e.range = range; e.cmpl = (1u << kPrecision) - range; e.magic = 0; e.sh = 0;
    e.x_max = ((kRansL >> kPrecision) << 32) * (uint64_t)range;
    if (range > 1) {
        uint32_t l = 0;
        while ((1ull << l) < range) ++l;
        const unsigned __int128 num = (unsigned __int128)1 << (63 + l);
        e.magic = (uint64_t)((num + range - 1) / range);
        e.sh = l - 1;
    }
    return e;
}

inline uint64_t put(uint64_t x, const Entry &e, uint32_t *&ptr) {
    if (x >= e.x_max) { *--ptr = (uint32_t)x; x >>= 32; }
    const uint64_t q = e.range == 1 ? x : (uint64_t)(((unsigned __int128)x * e.magic) >> 64) >> e.sh;
    return x + q * e.cmpl + e.start;
}

inline uint64_t put_bits(uint64_t x, uint32_t val, uint32_t *&ptr) {
    if (x >= kBypassXMax) { *--ptr = (uint32_t)x; x >>= 32; }
    return (x << kBypassPrecision) | val;
}

// the escape of rans.cpp:100-137 in REVERSE coding order: raw digits high to low, the unary digit count, the sentinel
inline uint64_t put_escape(uint64_t x, const Entry &sentinel, uint32_t raw_val, uint32_t *&ptr) {
    int32_t n_bypass = 0;
    while ((raw_val >> (n_bypass * kBypassPrecision)) != 0) ++n_bypass;
    for (int32_t j = n_bypass - 1; j >= 0; --j) x = put_bits(x, (raw_val >> (j * kBypassPrecision)) & kMaxBypassVal, ptr);
    // forward order: (15)*, then the remainder < 15 -> reversed: remainder first
    int32_t val = n_bypass;
    int32_t fifteens = 0;
    while (val >= (int32_t)kMaxBypassVal) { ++fifteens; val -= kMaxBypassVal; }
    x = put_bits(x, (uint32_t)val, ptr);
    for (int32_t j = 0; j < fifteens; ++j) x = put_bits(x, kMaxBypassVal, ptr);
    return put(x, sentinel, ptr);
}

}  // namespace

struct pmctf_rans_encoder {
    std::vector<Part> parts;
    std::vector<Push> pushes;
    std::vector<uint8_t> stream;  // assembled by flush()
    bool multi_thread = false;    // flush() runs in the background (parts in parallel); readers join first
    bool borrow = false;
    std::thread worker;
    void join() { if (worker.joinable()) worker.join(); }
    ~pmctf_rans_encoder() { join(); }

    std::vector<Entry> entries;   // flat interval table of every registered CDF table
    std::vector<Table> tables;

    void flush_part(size_t pi) {
        Part &p = parts[pi];
        const int64_t nparts = (int64_t)parts.size();
        // at most 16 bits per symbol plus 6 bypass digits: 40 bits -> 5/4 words
        size_t total = 0;
        for (const Push &u : pushes) {
            const int64_t each = u.n / nparts;
            total += (size_t)(pi + 1 < (size_t)nparts ? each : u.n - each * (nparts - 1));
        }
        const size_t need = total + total / 4 + 16;
        if (p.cap < need) { p.words.reset(new uint32_t[need]); p.cap = need; }
        uint32_t *const end = p.words.get() + p.cap;
        uint32_t *ptr = end;
        uint64_t x = kRansL;
        const Entry *const E = entries.data();
        for (size_t ui = pushes.size(); ui-- > 0;) {
            const Push &u = pushes[ui];
            const Table &t = tables[(size_t)u.table];
            const int64_t each = u.n / nparts;
            const int64_t b = (int64_t)pi * each;
            const int64_t cnt = (int64_t)pi + 1 < nparts ? each : u.n - each * (nparts - 1);
            const int16_t *sym = u.sym + b, *idx = u.idx + b;
            const int32_t *maxv = t.max_value.data(), *offs = t.offset.data();
            const Entry *const T = E + t.base;
            const int cols = t.cols;
            // (row 0, symbol 0): constants in registers when that value is a regular interval of row 0
            const int32_t v0 = 0 - offs[0];
            const bool fast0 = v0 >= 0 && v0 < maxv[0] && T[v0].range > 1;
            const Entry e0 = fast0 ? T[v0] : Entry{};
            for (int64_t i = cnt; i-- > 0;) {
                const int32_t row = idx[i];
                const int32_t s = sym[i];
                if (fast0 && (row | s) == 0) {
                    if (x >= e0.x_max) { *--ptr = (uint32_t)x; x >>= 32; }
                    const uint64_t q = (uint64_t)(((unsigned __int128)x * e0.magic) >> 64) >> e0.sh;
                    x = x + q * e0.cmpl + e0.start;
                    continue;
                }
                if (row < 0) continue;
                const int32_t max_value = maxv[row];
                const int32_t value = s - offs[row];
                const Entry *R = T + (size_t)row * cols;
                if (value >= 0 && value < max_value) {
                    x = put(x, R[value], ptr);
                } else {
                    const uint32_t raw = value < 0 ? (uint32_t)(-2 * value - 1) : (uint32_t)(2 * (value - max_value));
                    x = put_escape(x, R[max_value], raw, ptr);
                }
            }
        }
        ptr -= 2;
        ptr[0] = (uint32_t)x;
        ptr[1] = (uint32_t)(x >> 32);
        const size_t nbytes = (size_t)(end - ptr) * sizeof(uint32_t);
        p.stream.resize(nbytes);
        memcpy(p.stream.data(), ptr, nbytes);
    }

    void assemble(bool parallel) {
        const size_t np = parts.size();
        if (parallel && np > 1) {
            std::vector<std::thread> th;
            for (size_t i = 1; i < np; ++i) th.emplace_back([this, i] { flush_part(i); });
            flush_part(0);
            for (auto &t : th) t.join();
        } else {
            for (size_t i = 0; i < np; ++i) flush_part(i);
        }
        pushes.clear();
        size_t total = 0, max_size = 0;
        for (size_t i = 0; i < np; ++i) {
            const size_t sz = parts[i].stream.size();
            total += sz;
            if (i + 1 < np && sz > max_size) max_size = sz;
        }
        const size_t per_hdr = max_size > 65535 ? 4 : 2;
        const size_t overhead = 1 + (np > 1 ? (np - 1) * per_hdr : 0);
        stream.resize(total + overhead);
        uint8_t *o = stream.data();
        o[0] = (uint8_t)(((np - 1) << 4) + (per_hdr == 2 ? 1 : 0));
        for (size_t i = 0; i + 1 < np; ++i) {
            if (per_hdr == 2) { const uint16_t v = (uint16_t)parts[i].stream.size(); memcpy(o + 1 + 2 * i, &v, 2); }
            else { const uint32_t v = (uint32_t)parts[i].stream.size(); memcpy(o + 1 + 4 * i, &v, 4); }
        }
        size_t off = overhead;
        for (size_t i = 0; i < np; ++i) {
            memcpy(o + off, parts[i].stream.data(), parts[i].stream.size());
            off += parts[i].stream.size();
        }
    }

    // find or register (cdfs, sizes, offsets); tables are tiny (<= 26 k ints) so the checksum is recomputed per call
    int table_index(const int32_t *cdfs, int rows, int cols, const int32_t *sizes, const int32_t *offsets) {
        uint64_t h = 1469598103934665603ull;
        for (long i = 0; i < (long)rows * cols; ++i) h = (h ^ (uint32_t)cdfs[i]) * 1099511628211ull;
        for (int i = 0; i < rows; ++i) h = (h ^ (uint32_t)sizes[i]) * 1099511628211ull;
        for (int i = 0; i < rows; ++i) h = (h ^ (uint32_t)offsets[i]) * 1099511628211ull;
        for (size_t k = 0; k < tables.size(); ++k)
            if (tables[k].rows == rows && tables[k].cols == cols && tables[k].checksum == h) return (int)k;
        Table t;
        t.rows = rows; t.cols = cols; t.checksum = h; t.base = (uint32_t)entries.size();
        t.max_value.resize((size_t)rows); t.offset.assign(offsets, offsets + rows);
        entries.resize(entries.size() + (size_t)rows * cols);
        for (int r = 0; r < rows; ++r) {
            const int n = sizes[r] - 1;          // number of symbols in this row (incl. the escape symbol)
            t.max_value[(size_t)r] = sizes[r] - 2;
            for (int v = 0; v < cols; ++v) {
                Entry e = make_entry(0, 1);
                if (v < n && v + 1 < cols) e = make_entry((uint32_t)cdfs[(size_t)r * cols + v],
                                                          (uint32_t)(cdfs[(size_t)r * cols + v + 1] - cdfs[(size_t)r * cols + v]));
                entries[t.base + (size_t)r * cols + v] = e;
            }
        }
        tables.push_back(std::move(t));
        return (int)tables.size() - 1;
    }
};

struct pmctf_rans_decoder {
    struct D { std::vector<uint32_t> buf; const uint32_t *ptr = nullptr; const uint32_t *end = nullptr; uint64_t x = 0; };
    std::vector<D> parts;
};

extern "C" {

pmctf_rans_encoder *pmctf_rans_encoder_create(int multi_thread, int stream_part) {
    if (stream_part < 1 || stream_part > 16) return nullptr;
    auto *e = new (std::nothrow) pmctf_rans_encoder();
    if (e) {
        e->parts.resize((size_t)stream_part);
        e->multi_thread = multi_thread != 0;
    }
    return e;
}

void pmctf_rans_encoder_destroy(pmctf_rans_encoder *e) { delete e; }

int pmctf_rans_encoder_reset(pmctf_rans_encoder *e) {
    if (!e) return PMCTF_RANS_EINVAL;
    e->join();
    e->pushes.clear();
    return PMCTF_RANS_OK;
}

int pmctf_rans_encoder_set_borrow(pmctf_rans_encoder *e, int borrow) {
    if (!e) return PMCTF_RANS_EINVAL;
    e->borrow = borrow != 0;
    return PMCTF_RANS_OK;
}

int pmctf_rans_encoder_encode_with_indexes(pmctf_rans_encoder *e, const int16_t *symbols, const int16_t *indexes,
                                           int64_t n, const int32_t *cdfs, int cdf_rows, int cdf_cols,
                                           const int32_t *cdf_sizes, const int32_t *offsets) {
    if (!e || !symbols || !indexes || n < 0 || !cdfs || !cdf_sizes || !offsets || cdf_rows <= 0 || cdf_cols <= 2)
        return PMCTF_RANS_EINVAL;
    for (int r = 0; r < cdf_rows; ++r)
        if (cdf_sizes[r] < 2 || cdf_sizes[r] > cdf_cols) return PMCTF_RANS_EINVAL;
    int32_t hi = -1;
    for (int64_t i = 0; i < n; ++i) hi = indexes[i] > hi ? indexes[i] : hi;      // rows are checked now, coded at flush
    if (hi >= cdf_rows) return PMCTF_RANS_EINVAL;
    e->join();
    Push u;
    u.table = e->table_index(cdfs, cdf_rows, cdf_cols, cdf_sizes, offsets);
    u.n = n;
    if (e->borrow) {
        u.sym = symbols; u.idx = indexes;
    } else {
        u.owned.reset(new int16_t[2 * (size_t)n + 2]);
        memcpy(u.owned.get(), symbols, (size_t)n * sizeof(int16_t));
        memcpy(u.owned.get() + n, indexes, (size_t)n * sizeof(int16_t));
        u.sym = u.owned.get(); u.idx = u.owned.get() + n;
    }
    e->pushes.push_back(std::move(u));
    return PMCTF_RANS_OK;
}

int pmctf_rans_encoder_flush(pmctf_rans_encoder *e) {
    if (!e) return PMCTF_RANS_EINVAL;
    e->join();
    if (e->multi_thread) e->worker = std::thread([e] { e->assemble(true); });
    else e->assemble(false);
    return PMCTF_RANS_OK;
}

int64_t pmctf_rans_encoder_stream_size(const pmctf_rans_encoder *e) {
    if (!e) return PMCTF_RANS_EINVAL;
    const_cast<pmctf_rans_encoder *>(e)->join();
    return (int64_t)e->stream.size();
}

int pmctf_rans_encoder_get_encoded_stream(const pmctf_rans_encoder *e, uint8_t *out, int64_t capacity) {
    if (e) const_cast<pmctf_rans_encoder *>(e)->join();
    if (!e || !out || capacity < (int64_t)e->stream.size()) return PMCTF_RANS_EINVAL;
    memcpy(out, e->stream.data(), e->stream.size());
    return PMCTF_RANS_OK;
}

int64_t pmctf_rans_encoder_write_file(const pmctf_rans_encoder *e, const uint8_t *header, int64_t header_len,
                                      const char *path) {
    if (!e || !path || header_len < 0 || (header_len > 0 && !header)) return PMCTF_RANS_EINVAL;
    const_cast<pmctf_rans_encoder *>(e)->join();
    FILE *f = fopen(path, "wb");
    if (!f) return PMCTF_RANS_EIO;
    bool ok = true;
    if (header_len > 0) ok = fwrite(header, 1, (size_t)header_len, f) == (size_t)header_len;
    if (ok && !e->stream.empty()) ok = fwrite(e->stream.data(), 1, e->stream.size(), f) == e->stream.size();
    ok = (fclose(f) == 0) && ok;
    return ok ? header_len + (int64_t)e->stream.size() : PMCTF_RANS_EIO;
}

pmctf_rans_decoder *pmctf_rans_decoder_create(int stream_part) {
    if (stream_part < 1 || stream_part > 16) return nullptr;
    auto *d = new (std::nothrow) pmctf_rans_decoder();
    if (d) d->parts.resize((size_t)stream_part);
    return d;
}

void pmctf_rans_decoder_destroy(pmctf_rans_decoder *d) { delete d; }

int pmctf_rans_decoder_set_stream(pmctf_rans_decoder *d, const uint8_t *s, int64_t n) {
    if (!d || !s || n < 1) return PMCTF_RANS_EINVAL;
    const uint8_t flag = s[0];
    const size_t np = (size_t)(flag >> 4) + 1;
    if (np != d->parts.size()) return PMCTF_RANS_ESTREAM;
    const size_t len_bytes = (flag & 0x0f) == 1 ? 2 : 4;
    size_t off = 1, total = 0;
    std::vector<size_t> sizes;
    for (size_t i = 0; i + 1 < np; ++i) {
        if (off + len_bytes > (size_t)n) return PMCTF_RANS_ESTREAM;
        size_t v = 0;
        if (len_bytes == 2) { uint16_t t; memcpy(&t, s + off, 2); v = t; } else { uint32_t t; memcpy(&t, s + off, 4); v = t; }
        off += len_bytes;
        sizes.push_back(v);
        total += v;
    }
    if (off + total > (size_t)n) return PMCTF_RANS_ESTREAM;
    sizes.push_back((size_t)n - off - total);
    for (size_t i = 0; i < np; ++i) {
        auto &p = d->parts[i];
        if (sizes[i] < 8 || (sizes[i] & 3)) return PMCTF_RANS_ESTREAM;
        p.buf.assign(sizes[i] / 4 + 2, 0u);
        memcpy(p.buf.data(), s + off, sizes[i]);
        off += sizes[i];
        p.ptr = p.buf.data();
        p.end = p.buf.data() + sizes[i] / 4;
        p.x = (uint64_t)p.ptr[0] | ((uint64_t)p.ptr[1] << 32);
        p.ptr += 2;
    }
    return PMCTF_RANS_OK;
}

int pmctf_rans_decoder_decode_stream(pmctf_rans_decoder *d, const int16_t *indexes, int64_t n, const int32_t *cdfs,
                                     int cdf_rows, int cdf_cols, const int32_t *cdf_sizes, const int32_t *offsets,
                                     int16_t *out) {
    if (!d || !indexes || n < 0 || !cdfs || !cdf_sizes || !offsets || !out || cdf_rows <= 0) return PMCTF_RANS_EINVAL;
    const int64_t nparts = (int64_t)d->parts.size();
    const int64_t each = n / nparts;
    for (int64_t pi = 0; pi < nparts; ++pi) {
        auto &p = d->parts[(size_t)pi];
        const int64_t b = pi * each;
        const int64_t cnt = pi < nparts - 1 ? each : n - each * (nparts - 1);
        // the coder state in locals for the whole run (through `p` every symbol would store and reload it: the int16 /
        // uint32 accesses of the loop may alias the struct as far as the compiler can tell)
        uint64_t x = p.x;
        const uint32_t *ptr = p.ptr;
        const uint32_t *const end = p.end;
        auto get_bits = [&](uint32_t nbits) -> uint32_t {
            const uint32_t val = (uint32_t)(x & ((1u << nbits) - 1));
            x >>= nbits;
            if (x < kRansL && ptr <= end) { x = (x << 32) | *ptr; ptr += 1; }
            return val;
        };
        int rc = PMCTF_RANS_OK;
        for (int64_t i = b; i < b + cnt; ++i) {
            const int32_t row = indexes[i];
            if (row < 0 || row >= cdf_rows) { rc = PMCTF_RANS_EINVAL; break; }
            const int32_t offset = offsets[row];
            const int32_t *cdf = cdfs + (size_t)row * cdf_cols;
            const int32_t size = cdf_sizes[row];
            const int32_t max_value = size - 2;
            const uint32_t cum = (uint32_t)(x & ((1u << kPrecision) - 1));
            // s = (number of leading entries <= cum) - 1; the rows are non-decreasing: narrow rows by a short scan, wide
            // rows (large scales: ~100 entries) by bisection
            int32_t s;
            if (size <= 8) {
                s = 0;
                while (s < size && (uint32_t)cdf[s] <= cum) ++s;
            } else {
                int32_t lo = 0, len = size;               // first index in [0, size) whose entry is > cum
                while (len > 0) {
                    const int32_t half = len >> 1;
                    const bool le = (uint32_t)cdf[lo + half] <= cum;
                    lo = le ? lo + half + 1 : lo;
                    len = le ? len - half - 1 : half;
                }
                s = lo;
            }
            s -= 1;
            if (s < 0 || s + 1 >= cdf_cols) { rc = PMCTF_RANS_ESTREAM; break; }
            {
                const uint32_t start = (uint32_t)cdf[s], freq = (uint32_t)(cdf[s + 1] - cdf[s]);
                x = (uint64_t)freq * (x >> kPrecision) + (x & ((1ull << kPrecision) - 1)) - start;
                if (x < kRansL && ptr <= end) { x = (x << 32) | *ptr; ptr += 1; }
            }
            int32_t value = s;
            if (value == max_value) {
                int32_t val = (int32_t)get_bits(kBypassPrecision);
                int32_t n_bypass = val;
                while (val == (int32_t)kMaxBypassVal) {
                    val = (int32_t)get_bits(kBypassPrecision);
                    n_bypass += val;
                }
                int32_t raw_val = 0;
                for (int32_t j = 0; j < n_bypass; ++j) {
                    val = (int32_t)get_bits(kBypassPrecision);
                    raw_val |= val << (j * kBypassPrecision);
                }
                value = raw_val >> 1;
                if (raw_val & 1) value = -value - 1; else value += max_value;
            }
            out[i] = (int16_t)(value + offset);
        }
        p.x = x;
        p.ptr = const_cast<decltype(p.ptr)>(ptr);
        if (rc != PMCTF_RANS_OK) return rc;
    }
    return PMCTF_RANS_OK;
}

int pmctf_rans_decoder_get_state(const pmctf_rans_decoder *d, uint64_t *x, int64_t *word_pos) {
    if (!d || !x || !word_pos || d->parts.size() != 1 || !d->parts[0].ptr) return PMCTF_RANS_EINVAL;
    *x = d->parts[0].x;
    *word_pos = (int64_t)(d->parts[0].ptr - d->parts[0].buf.data());
    return PMCTF_RANS_OK;
}

int pmctf_rans_decoder_set_state(pmctf_rans_decoder *d, uint64_t x, int64_t word_pos) {
    if (!d || d->parts.size() != 1 || !d->parts[0].ptr || word_pos < 2 ||
        word_pos > (int64_t)(d->parts[0].end - d->parts[0].buf.data()) + 1)
        return PMCTF_RANS_EINVAL;
    d->parts[0].x = x;
    d->parts[0].ptr = d->parts[0].buf.data() + word_pos;
    return PMCTF_RANS_OK;
}

int pmctf_pmf_to_quantized_cdf(const float *pmf, int n, int precision, uint32_t *cdf) {
    if (!pmf || !cdf || n <= 0 || precision <= 0 || precision > 16) return PMCTF_RANS_EINVAL;
    cdf[0] = 0;
    for (int i = 0; i < n; ++i) cdf[i + 1] = (uint32_t)((double)std::round(pmf[i] * (float)(1 << precision)) + 0.5);
    uint32_t total = 0;
    for (int i = 0; i <= n; ++i) total += cdf[i];
    if (total == 0) return PMCTF_RANS_EINVAL;
    for (int i = 0; i <= n; ++i) cdf[i] = (uint32_t)((((uint64_t)1 << precision) * cdf[i]) / total);
    for (int i = 1; i <= n; ++i) cdf[i] += cdf[i - 1];
    cdf[n] = 1u << precision;
    for (int i = 0; i < n; ++i) {
        if (cdf[i] == cdf[i + 1]) {
            uint32_t best_freq = ~0u;
            int best_steal = -1;
            for (int j = 0; j < n; ++j) {
                const uint32_t freq = cdf[j + 1] - cdf[j];
                if (freq > 1 && freq < best_freq) { best_freq = freq; best_steal = j; }
            }
            if (best_steal == -1) return PMCTF_RANS_EINVAL;
            if (best_steal < i) { for (int j = best_steal + 1; j <= i; ++j) cdf[j]--; }
            else { for (int j = i + 1; j <= best_steal; ++j) cdf[j]++; }
        }
    }
    return PMCTF_RANS_OK;
}

}  // extern "C"
