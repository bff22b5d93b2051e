/* pmctf_rans.h — C ABI of libpmctf_rans.so: the host-side range coder of the pMCTF encode path.
 *
 * These entry points are what the reference binds through pybind11 as
 * pMCTF.models.MLCodec_rans / pMCTF.models.MLCodec_CXX (pMCTF/cpp/py_rans/py_rans.cpp:227-243,
 * pMCTF/cpp/ops/ops.cpp:84-91; used from pMCTF/entropy_models/entropy_models.py:9-55): same operations,
 * same argument meaning, raw pointers + lengths and an opaque handle instead of numpy arrays.
 * All pointers are HOST pointers.  Return value 0 = ok, negative = error (PMCTF_RANS_E*).
 * A handle is not thread-safe; different handles may be used from different threads.
 */
#ifndef PMCTF_RANS_H
#define PMCTF_RANS_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define PMCTF_RANS_OK 0
#define PMCTF_RANS_EINVAL (-1)   /* bad argument / out-of-range CDF row */
#define PMCTF_RANS_ESTREAM (-2)  /* malformed stream on decode */
#define PMCTF_RANS_EIO (-3)      /* file write failed */

typedef struct pmctf_rans_encoder pmctf_rans_encoder;
typedef struct pmctf_rans_decoder pmctf_rans_decoder;

/* RansEncoder(bool multiThread, int streamPart)                      py_rans.cpp:11-20, rans.cpp:174-263
 * multi_thread != 0: flush() returns at once and the stream is coded on a background thread (the stream_part parts in
 * parallel); stream_size / get_encoded_stream / write_file / reset / encode_with_indexes wait for it.  The bytes are
 * the same as with multi_thread == 0. */
pmctf_rans_encoder *pmctf_rans_encoder_create(int multi_thread, int stream_part);
void pmctf_rans_encoder_destroy(pmctf_rans_encoder *e);
/* RansEncoder.reset()                                                 py_rans.cpp:121-125 */
int pmctf_rans_encoder_reset(pmctf_rans_encoder *e);
/* RansEncoder.encode_with_indexes(symbols, indexes, cdfs, cdfs_sizes, offsets)
 *                                                  py_rans.cpp:22-66 + rans.cpp:76-139
 * symbols/indexes: n int16 each; cdfs: [cdf_rows][cdf_cols] int32; cdf_sizes, offsets: [cdf_rows].
 * The two int16 arrays are copied before the call returns (as the pybind wrapper copies its numpy arguments,
 * py_rans.cpp:36-66) and coded at flush(); CDF rows are range-checked here. */
int pmctf_rans_encoder_encode_with_indexes(pmctf_rans_encoder *e, const int16_t *symbols, const int16_t *indexes,
                                           int64_t n, const int32_t *cdfs, int cdf_rows, int cdf_cols,
                                           const int32_t *cdf_sizes, const int32_t *offsets);
/* Not in the reference: borrow != 0 makes the following encode_with_indexes calls keep the caller's pointers instead
 * of copying (the arrays must stay valid and unchanged until flush() has completed — for multi_thread encoders until
 * stream_size / get_encoded_stream / write_file has returned).  Used by the product's writer threads, whose symbol
 * buffers are pinned host memory owned by the job. */
int pmctf_rans_encoder_set_borrow(pmctf_rans_encoder *e, int borrow);
/* RansEncoder.flush()                                                 py_rans.cpp:68-72 + rans.cpp:141-168 */
int pmctf_rans_encoder_flush(pmctf_rans_encoder *e);
/* RansEncoder.get_encoded_stream(): flag byte + per-stream sizes + payloads   py_rans.cpp:74-119 */
int64_t pmctf_rans_encoder_stream_size(const pmctf_rans_encoder *e);
int pmctf_rans_encoder_get_encoded_stream(const pmctf_rans_encoder *e, uint8_t *out, int64_t capacity);
/* Convenience used by the product's writer threads: header bytes + encoded stream -> file in one call
 * (what encode_p / encode_image do with the stream, pMCTF/utils/stream_helper.py:181-207).
 * Returns the file size in bytes, or a negative error. */
int64_t pmctf_rans_encoder_write_file(const pmctf_rans_encoder *e, const uint8_t *header, int64_t header_len,
                                      const char *path);

/* RansDecoder(int streamPart), set_stream, decode_stream              py_rans.cpp:127-225 + rans.cpp:265-331 */
pmctf_rans_decoder *pmctf_rans_decoder_create(int stream_part);
void pmctf_rans_decoder_destroy(pmctf_rans_decoder *d);
int pmctf_rans_decoder_set_stream(pmctf_rans_decoder *d, const uint8_t *stream, int64_t n);
int pmctf_rans_decoder_decode_stream(pmctf_rans_decoder *d, const int16_t *indexes, int64_t n, const int32_t *cdfs,
                                     int cdf_rows, int cdf_cols, const int32_t *cdf_sizes, const int32_t *offsets,
                                     int16_t *out);

/* Hand-over of a single-stream decoder's state to/from the in-kernel LL decoder (include/pmctf_hip.h,
 * pmctf_ll_ar_decode_f32): Rans64 state x and the index of the next 32-bit payload word. */
int pmctf_rans_decoder_get_state(const pmctf_rans_decoder *d, uint64_t *x, int64_t *word_pos);
int pmctf_rans_decoder_set_state(pmctf_rans_decoder *d, uint64_t x, int64_t word_pos);

/* pmf_to_quantized_cdf(pmf, precision) -> cdf[n+1]                    ops.cpp:24-82 */
int pmctf_pmf_to_quantized_cdf(const float *pmf, int n, int precision, uint32_t *cdf);

#ifdef __cplusplus
}
#endif
#endif
