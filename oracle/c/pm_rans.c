/* ORACLE — test infrastructure only (see pm_math.h header note).
 *
 * pm_rans.c: plain-C restatement of the reference's CPU range coder.
 *   symbol mapping + bypass coding   pMCTF/cpp/rans/rans.cpp:76-139
 *   reverse-order flush              pMCTF/cpp/rans/rans.cpp:141-168
 *   decoder                          pMCTF/cpp/rans/rans.cpp:265-331
 *   1-byte stream header (1 stream)  pMCTF/cpp/py_rans/py_rans.cpp:74-119,133-164
 *   N-part container (pm_rans_menc / pm_rans_mdec): per-push partition py_rans.cpp:52-65,196-209,
 *   flag byte + 2/4-byte part sizes py_rans.cpp:74-119, parsing :133-164
 *   pmf -> quantised CDF             pMCTF/cpp/ops/ops.cpp:24-82
 * The 64-bit rANS state primitives come from a dependency that is NOT in
 * /root/reference: rygorous/ryg_rans @ c9d162d996fd600315af9ae8eb89d832576cb32d,
 * file rans64.h (fetched by pMCTF/cpp/3rdparty/ryg_rans/CMakeLists.txt.in:8-9).
 * Its published algorithm is restated below (state in [2^31, 2^63), 32-bit
 * little-endian renormalisation words, encoder writes backwards).  Those
 * primitives are pinned only through the end-to-end known-answer stream in
 * SURVEY.md §8c (tests/test_oracle_kat.py::test_rans_known_answer_stream_and_roundtrip).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define PM_PRECISION 16
#define PM_BYPASS_PRECISION 4
#define PM_MAX_BYPASS_VAL ((1 << PM_BYPASS_PRECISION) - 1)
#define PM_RANS64_L (1ull << 31)

typedef struct { uint16_t start, range; uint8_t bypass; } pm_sym;

typedef struct pm_rans_enc {
    pm_sym *syms; long n, cap;
    uint8_t *stream; long stream_len;
} pm_rans_enc;

pm_rans_enc *pm_rans_enc_new(void) { return (pm_rans_enc *)calloc(1, sizeof(pm_rans_enc)); }
void pm_rans_enc_free(pm_rans_enc *e) { if (e) { free(e->syms); free(e->stream); free(e); } }
void pm_rans_enc_reset(pm_rans_enc *e) { e->n = 0; }

static void pm_push(pm_rans_enc *e, uint16_t start, uint16_t range, uint8_t bypass) {
    if (e->n == e->cap) {
        e->cap = e->cap ? e->cap * 2 : 1 << 16;
        e->syms = (pm_sym *)realloc(e->syms, (size_t)e->cap * sizeof(pm_sym));
    }
    e->syms[e->n].start = start; e->syms[e->n].range = range; e->syms[e->n].bypass = bypass;
    e->n++;
}

/* rans.cpp:76-139 */
void pm_rans_enc_encode_with_indexes(pm_rans_enc *e, const int16_t *symbols, const int16_t *indexes,
                                     long n, const int32_t *cdfs, int cdf_cols,
                                     const int32_t *cdf_sizes, const int32_t *offsets) {
    for (long i = 0; i < n; ++i) {
        const int32_t idx = indexes[i];
        if (idx < 0) continue;
        const int32_t *cdf = cdfs + (long)idx * cdf_cols;
        const int32_t max_value = cdf_sizes[idx] - 2;
        int32_t value = (int32_t)symbols[i] - offsets[idx];
        uint32_t raw_val = 0;
        if (value < 0) { raw_val = (uint32_t)(-2 * value - 1); value = max_value; }
        else if (value >= max_value) { raw_val = (uint32_t)(2 * (value - max_value)); value = max_value; }
        pm_push(e, (uint16_t)cdf[value], (uint16_t)(cdf[value + 1] - cdf[value]), 0);
        if (value == max_value) {
            int32_t n_bypass = 0;
            while ((raw_val >> (n_bypass * PM_BYPASS_PRECISION)) != 0) ++n_bypass;
            int32_t val = n_bypass;
            while (val >= PM_MAX_BYPASS_VAL) {
                pm_push(e, PM_MAX_BYPASS_VAL, PM_MAX_BYPASS_VAL + 1, 1);
                val -= PM_MAX_BYPASS_VAL;
            }
            pm_push(e, (uint16_t)val, (uint16_t)(val + 1), 1);
            for (int32_t j = 0; j < n_bypass; ++j) {
                const int32_t v1 = (raw_val >> (j * PM_BYPASS_PRECISION)) & PM_MAX_BYPASS_VAL;
                pm_push(e, (uint16_t)v1, (uint16_t)(v1 + 1), 1);
            }
        }
    }
}

/* ryg_rans rans64.h: Rans64EncPut */
static inline void pm_enc_put(uint64_t *r, uint32_t **pp, uint32_t start, uint32_t freq, uint32_t bits) {
    uint64_t x = *r;
    const uint64_t x_max = ((PM_RANS64_L >> bits) << 32) * freq;
    if (x >= x_max) { *pp -= 1; **pp = (uint32_t)x; x >>= 32; }
    *r = ((x / freq) << bits) + (x % freq) + start;
}
/* rans.cpp:36-56 Rans64EncPutBits */
static inline void pm_enc_put_bits(uint64_t *r, uint32_t **pp, uint32_t val, uint32_t nbits) {
    uint64_t x = *r;
    const uint32_t freq = 1u << (16 - nbits);
    const uint64_t x_max = ((PM_RANS64_L >> 16) << 32) * freq;
    if (x >= x_max) { *pp -= 1; **pp = (uint32_t)x; x >>= 32; }
    *r = (x << nbits) | val;
}

/* rans.cpp:141-168 + py_rans.cpp:74-119 (single stream: flag byte 0x01) */
void pm_rans_enc_flush(pm_rans_enc *e) {
    uint64_t rans = PM_RANS64_L; /* Rans64EncInit */
    const long words = e->n + 2;
    uint32_t *out = (uint32_t *)malloc((size_t)words * 4);
    uint32_t *ptr = out + words;
    for (long i = e->n - 1; i >= 0; --i) {
        const pm_sym s = e->syms[i];
        if (!s.bypass) pm_enc_put(&rans, &ptr, s.start, s.range, PM_PRECISION);
        else pm_enc_put_bits(&rans, &ptr, s.start, PM_BYPASS_PRECISION);
    }
    e->n = 0;
    /* Rans64EncFlush */
    ptr -= 2; ptr[0] = (uint32_t)rans; ptr[1] = (uint32_t)(rans >> 32);
    const long nbytes = (long)((out + words) - ptr) * 4;
    free(e->stream);
    e->stream = (uint8_t *)malloc((size_t)nbytes + 1);
    e->stream[0] = 0x01; /* ((1-1)<<4) + (perStreamHeader==2) */
    memcpy(e->stream + 1, ptr, (size_t)nbytes);
    e->stream_len = nbytes + 1;
    free(out);
}
long pm_rans_enc_stream_size(const pm_rans_enc *e) { return e->stream_len; }
void pm_rans_enc_get_stream(const pm_rans_enc *e, uint8_t *dst) { memcpy(dst, e->stream, (size_t)e->stream_len); }

/* ------------------------------ decoder ---------------------------------- */
typedef struct pm_rans_dec { uint64_t rans; uint32_t *buf; uint32_t *ptr; } pm_rans_dec;

pm_rans_dec *pm_rans_dec_new(void) { return (pm_rans_dec *)calloc(1, sizeof(pm_rans_dec)); }
void pm_rans_dec_free(pm_rans_dec *d) { if (d) { free(d->buf); free(d); } }

/* py_rans.cpp:133-164 (one stream) + rans.cpp:265-270 */
int pm_rans_dec_set_stream(pm_rans_dec *d, const uint8_t *bytes, long n) {
    if (n < 9 || (bytes[0] >> 4) != 0) return -1;
    free(d->buf);
    d->buf = (uint32_t *)malloc((size_t)(n - 1) + 16);
    memset(d->buf, 0, (size_t)(n - 1) + 16);
    memcpy(d->buf, bytes + 1, (size_t)(n - 1));
    d->ptr = d->buf;
    d->rans = (uint64_t)d->ptr[0] | ((uint64_t)d->ptr[1] << 32); /* Rans64DecInit */
    d->ptr += 2;
    return 0;
}

static inline uint32_t pm_dec_get_bits(pm_rans_dec *d, uint32_t nbits) { /* rans.cpp:58-73 */
    uint64_t x = d->rans;
    const uint32_t val = (uint32_t)(x & ((1u << nbits) - 1));
    x >>= nbits;
    if (x < PM_RANS64_L) { x = (x << 32) | *d->ptr; d->ptr += 1; }
    d->rans = x;
    return val;
}

/* rans.cpp:272-331 */
void pm_rans_dec_decode_stream(pm_rans_dec *d, const int16_t *indexes, long n, const int32_t *cdfs,
                               int cdf_cols, const int32_t *cdf_sizes, const int32_t *offsets,
                               int16_t *out) {
    for (long i = 0; i < n; ++i) {
        const int32_t idx = indexes[i];
        if (idx < 0) { out[i] = (int16_t)offsets[0]; continue; }
        const int32_t offset = offsets[idx];
        const int32_t *cdf = cdfs + (long)idx * cdf_cols;
        const int32_t size = cdf_sizes[idx];
        const int32_t max_value = size - 2;
        const uint32_t cum = (uint32_t)(d->rans & ((1u << PM_PRECISION) - 1)); /* Rans64DecGet */
        int32_t s = 0;
        while (s < size && (uint32_t)cdf[s] <= cum) ++s;
        s -= 1;
        { /* Rans64DecAdvance */
            const uint64_t mask = (1ull << PM_PRECISION) - 1;
            const uint32_t start = (uint32_t)cdf[s], freq = (uint32_t)(cdf[s + 1] - cdf[s]);
            uint64_t x = d->rans;
            x = freq * (x >> PM_PRECISION) + (x & mask) - start;
            if (x < PM_RANS64_L) { x = (x << 32) | *d->ptr; d->ptr += 1; }
            d->rans = x;
        }
        int32_t value = s;
        if (value == max_value) {
            int32_t val = (int32_t)pm_dec_get_bits(d, PM_BYPASS_PRECISION);
            int32_t n_bypass = val;
            while (val == PM_MAX_BYPASS_VAL) {
                val = (int32_t)pm_dec_get_bits(d, PM_BYPASS_PRECISION);
                n_bypass += val;
            }
            int32_t raw_val = 0;
            for (int32_t j = 0; j < n_bypass; ++j) {
                val = (int32_t)pm_dec_get_bits(d, PM_BYPASS_PRECISION);
                raw_val |= val << (j * PM_BYPASS_PRECISION);
            }
            value = raw_val >> 1;
            if (raw_val & 1) value = -value - 1; else value += max_value;
        }
        out[i] = (int16_t)(value + offset);
    }
}

/* ------------------- N-part container (stream_part > 1) ------------------ */
/* py_rans.cpp: RansEncoder(multiThread, streamPart) holds streamPart RansEncoderLib objects; EVERY
 * encode_with_indexes call cuts its symbols into streamPart runs of n/streamPart (the last takes the
 * remainder, :52-65); get_encoded_stream (:74-119) writes  flag = ((N-1)<<4) | (sizes are 2-byte ? 1 : 0),
 * then the sizes of the first N-1 part streams (2 bytes each if the largest of THEM is <= 65535, else 4),
 * then the part streams back to back.  The decoder (:133-164, :196-224) undoes exactly that. */
typedef struct pm_rans_menc { int parts; pm_rans_enc **e; uint8_t *stream; long stream_len; } pm_rans_menc;

pm_rans_menc *pm_rans_menc_new(int parts) {
    if (parts < 1 || parts > 16) return NULL;
    pm_rans_menc *m = (pm_rans_menc *)calloc(1, sizeof(pm_rans_menc));
    m->parts = parts;
    m->e = (pm_rans_enc **)calloc((size_t)parts, sizeof(pm_rans_enc *));
    for (int i = 0; i < parts; ++i) m->e[i] = pm_rans_enc_new();
    return m;
}
void pm_rans_menc_free(pm_rans_menc *m) {
    if (!m) return;
    for (int i = 0; i < m->parts; ++i) pm_rans_enc_free(m->e[i]);
    free(m->e); free(m->stream); free(m);
}
void pm_rans_menc_reset(pm_rans_menc *m) { for (int i = 0; i < m->parts; ++i) pm_rans_enc_reset(m->e[i]); }
void pm_rans_menc_encode_with_indexes(pm_rans_menc *m, const int16_t *symbols, const int16_t *indexes, long n,
                                      const int32_t *cdfs, int cdf_cols, const int32_t *cdf_sizes,
                                      const int32_t *offsets) {
    const long each = n / m->parts, last = n - each * (m->parts - 1);
    for (int i = 0; i < m->parts; ++i)
        pm_rans_enc_encode_with_indexes(m->e[i], symbols + i * each, indexes + i * each,
                                        i < m->parts - 1 ? each : last, cdfs, cdf_cols, cdf_sizes, offsets);
}
void pm_rans_menc_flush(pm_rans_menc *m) {
    long total = 0, maximum = 0;
    for (int i = 0; i < m->parts; ++i) {
        pm_rans_enc_flush(m->e[i]);                  /* e->stream = 1 flag byte + payload */
        const long nb = m->e[i]->stream_len - 1;
        if (i < m->parts - 1 && nb > maximum) maximum = nb;
        total += nb;
    }
    const int hdr = maximum > 65535 ? 4 : 2;
    const long overhead = 1 + (long)(m->parts - 1) * hdr;
    free(m->stream);
    m->stream = (uint8_t *)malloc((size_t)(total + overhead));
    m->stream[0] = (uint8_t)(((m->parts - 1) << 4) + (hdr == 2 ? 1 : 0));
    for (int i = 0; i < m->parts - 1; ++i) {
        const long nb = m->e[i]->stream_len - 1;
        if (hdr == 2) { const uint16_t v = (uint16_t)nb; memcpy(m->stream + 1 + 2 * i, &v, 2); }
        else { const uint32_t v = (uint32_t)nb; memcpy(m->stream + 1 + 4 * i, &v, 4); }
    }
    long o = overhead;
    for (int i = 0; i < m->parts; ++i) {
        const long nb = m->e[i]->stream_len - 1;
        memcpy(m->stream + o, m->e[i]->stream + 1, (size_t)nb);
        o += nb;
    }
    m->stream_len = total + overhead;
}
long pm_rans_menc_stream_size(const pm_rans_menc *m) { return m->stream_len; }
void pm_rans_menc_get_stream(const pm_rans_menc *m, uint8_t *dst) { memcpy(dst, m->stream, (size_t)m->stream_len); }

typedef struct pm_rans_mdec { int parts; pm_rans_dec **d; } pm_rans_mdec;
pm_rans_mdec *pm_rans_mdec_new(int parts) {
    if (parts < 1 || parts > 16) return NULL;
    pm_rans_mdec *m = (pm_rans_mdec *)calloc(1, sizeof(pm_rans_mdec));
    m->parts = parts;
    m->d = (pm_rans_dec **)calloc((size_t)parts, sizeof(pm_rans_dec *));
    for (int i = 0; i < parts; ++i) m->d[i] = pm_rans_dec_new();
    return m;
}
void pm_rans_mdec_free(pm_rans_mdec *m) {
    if (!m) return;
    for (int i = 0; i < m->parts; ++i) pm_rans_dec_free(m->d[i]);
    free(m->d); free(m);
}
/* py_rans.cpp:133-164 */
int pm_rans_mdec_set_stream(pm_rans_mdec *m, const uint8_t *bytes, long n) {
    if (n < 1) return -1;
    const int streams = (bytes[0] >> 4) + 1;
    const int hdr = (bytes[0] & 0x0f) == 1 ? 2 : 4;
    if (streams != m->parts) return -1;
    long offset = 1, sizes[16], total = 0;
    for (int i = 0; i < streams - 1; ++i) {
        if (offset + hdr > n) return -1;
        if (hdr == 2) { uint16_t v; memcpy(&v, bytes + offset, 2); sizes[i] = v; }
        else { uint32_t v; memcpy(&v, bytes + offset, 4); sizes[i] = v; }
        offset += hdr; total += sizes[i];
    }
    sizes[streams - 1] = n - offset - total;
    for (int i = 0; i < streams; ++i) {
        if (sizes[i] < 8 || offset + sizes[i] > n) return -1;
        /* pm_rans_dec_set_stream wants the single-stream framing: one flag byte in front of the payload */
        uint8_t *tmp = (uint8_t *)malloc((size_t)sizes[i] + 1);
        tmp[0] = 0x01;
        memcpy(tmp + 1, bytes + offset, (size_t)sizes[i]);
        const int rc = pm_rans_dec_set_stream(m->d[i], tmp, sizes[i] + 1);
        free(tmp);
        if (rc != 0) return -1;
        offset += sizes[i];
    }
    return 0;
}
/* py_rans.cpp:196-224: the index array of ONE decode_stream call is cut like the symbols of one push */
void pm_rans_mdec_decode_stream(pm_rans_mdec *m, const int16_t *indexes, long n, const int32_t *cdfs, int cdf_cols,
                                const int32_t *cdf_sizes, const int32_t *offsets, int16_t *out) {
    const long each = n / m->parts, last = n - each * (m->parts - 1);
    for (int i = 0; i < m->parts; ++i)
        pm_rans_dec_decode_stream(m->d[i], indexes + i * each, i < m->parts - 1 ? each : last, cdfs, cdf_cols,
                                  cdf_sizes, offsets, out + i * each);
}

/* ops.cpp:24-82.  cdf has n+1 entries.  Returns 0, or -1 if no frequency can be stolen. */
int pm_pmf_to_quantized_cdf(const float *pmf, int n, int precision, uint32_t *cdf) {
    cdf[0] = 0;
    for (int i = 0; i < n; ++i) {
        /* static_cast<uint32_t>(std::round(p * (1 << precision)) + 0.5): float product, float round, +0.5 in double */
        const float scaled = pmf[i] * (float)(1 << precision);
        const float r = roundf(scaled); /* std::round: half away from zero */
        cdf[i + 1] = (uint32_t)((double)r + 0.5);
    }
    uint32_t total = 0; /* std::accumulate with int init 0 -> wraps like uint32 */
    for (int i = 0; i <= n; ++i) total += cdf[i];
    for (int i = 0; i <= n; ++i) cdf[i] = (uint32_t)((((uint64_t)1 << precision) * cdf[i]) / total);
    for (int i = 1; i <= n; ++i) cdf[i] += cdf[i - 1];
    cdf[n] = 1u << precision;
    for (int i = 0; i < n; ++i) {
        if (cdf[i] == cdf[i + 1]) {
            uint32_t best_freq = ~0u; int best_steal = -1;
            for (int j = 0; j < n; ++j) {
                const uint32_t freq = cdf[j + 1] - cdf[j];
                if (freq > 1 && freq < best_freq) { best_freq = freq; best_steal = j; }
            }
            if (best_steal == -1) return -1;
            if (best_steal < i) { for (int j = best_steal + 1; j <= i; ++j) cdf[j]--; }
            else { for (int j = i + 1; j <= best_steal; ++j) cdf[j]++; }
        }
    }
    return 0;
}
