/*
 * vsearch_hip.h -- C ABI of libvsearch_hip.so: the MI355X (gfx950) implementation of vsearch's
 * vocabulary-space retrieval hot path.
 *
 * The reference (jzhoubu/vsearch @ 2024-12-18) is pure Python and has no FFI layer; the boundary it
 * offers is its Python API (src/ir/retriever/index.py, retriever.py, src/ir/utils/sparse.py,
 * src/ir/encoder/vdr.py).  This header is the boundary *below* that API: each entry point names the
 * reference call site (file:line under the reference root) whose arithmetic it replaces.  The
 * Python facade in vsearch_amd/ir/ (same class / method names as src/ir) binds these symbols
 * through ctypes; INTEGRATION.md shows the stub a maintainer of the reference would add.
 *
 * Conventions
 *   - Plain C: pointers + explicit sizes, no C++ or torch types.  Returns 0 (VS_OK) or a negative
 *     VS_E* code; vs_last_error() gives the thread-local message.
 *   - Data pointers may be host or device (HIP) pointers; the library detects which
 *     (hipPointerGetAttributes).  Device pointers must belong to the index's device.
 *   - `stream` is a hipStream_t (e.g. torch.cuda.current_stream().cuda_stream).  NULL = the
 *     device's null stream AND a blocking call: it returns after the work has completed.  With a
 *     non-NULL stream (hipStreamLegacy = (hipStream_t)1 names the null stream without blocking) and
 *     device pointers for every input and output, vs_index_search on the blocked-postings filter
 *     path (vs_index_info_t.last_path == 3: the default for large sparse / bag-of-token indexes)
 *     only enqueues kernels: tile plan, candidate proof and the exact pass for unproven queries are
 *     decided on the device.  The other paths (8-query CSR scan, fp64 walk, one-query scan)
 *     synchronise once per call to read the tile plan; host pointers are copied and block.
 *     vs_index_info() reads device-side statistics of the last search and therefore synchronises.
 *   - ONE search at a time per vs_index handle, from ONE host thread, on ONE stream at a time: the
 *     handle and a few per-device buffers (vs_merge_topk, the sparsify entry points) own grow-only
 *     scratch memory that consecutive calls re-use in stream order.  Calls on a different stream
 *     must be ordered after the previous call's work by the caller (event / synchronise).
 *   - Top-k order is canonical: score descending, then id ascending (torch.topk leaves ties
 *     unspecified, index.py:92).  fp32 accumulation order differs from MKL/cuSPARSE: scores agree to
 *     ~1e-6 relative; on binary index x dyadic query weights they are bit-exact.
 */
#ifndef VSEARCH_HIP_H
#define VSEARCH_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VS_API __attribute__((visibility("default")))

/* error codes */
#define VS_OK            0
#define VS_EINVAL       -1   /* bad argument (Python facade: ValueError / TypeError)                   */
#define VS_ERANGE       -2   /* k > number of rows: torch.topk's RuntimeError (index.py:92)           */
#define VS_ENOMEM       -3   /* host or device allocation failed                                      */
#define VS_EHIP         -4   /* HIP runtime error (message has the hipError string)                   */
#define VS_EUNSUPPORTED -5   /* e.g. n_cols > 65535                                                   */
#define VS_ENODEVICE    -6   /* no usable gfx950 device: the product path never falls back to the CPU */

/* element types */
#define VS_F32   0
#define VS_F16   1
#define VS_I32   2
#define VS_I64   3
#define VS_U16   4
#define VS_U8    5
#define VS_NONE -1           /* "no values": binary (bag-of-token) index                              */

/* index kinds (IndexType, index.py:20-23) */
#define VS_KIND_DENSE 0
#define VS_KIND_CSR   1      /* SparseIndex / BoTIndex                                                */

typedef struct vs_index vs_index;

typedef struct vs_index_info_t {
    int32_t kind;            /* VS_KIND_*                                                             */
    int32_t store_dtype;     /* VS_F32 | VS_F16 | VS_NONE (binary)                                    */
    int64_t n_rows;
    int32_t n_cols;
    int32_t device;
    int64_t nnz;             /* logical non-zeros (CSR) or n_rows*n_cols (dense)                      */
    int64_t n_packets;       /* CSR device format: 8-nnz packets incl. row padding                    */
    int64_t device_bytes;    /* bytes of the device-resident index                                    */
    int64_t bytes_per_pass;  /* algorithmic HBM bytes one scoring pass streams      */
    int32_t lanes_per_row;   /* CSR scan geometry                                                     */
    int32_t queries_per_pass;/* Qt of the most recent search() (the planned default before any search)   */
    int64_t last_scan_bytes; /* bytes the scan kernels of the most recent search() had to read: passes x
                              * bytes_per_pass on the CSR paths; on the blocked-postings path the posting lists of
                              * the queries' columns (document ids + values) + their directory entries            */
    int64_t aux_bytes;       /* bytes of the blocked-postings copy (0 when absent)                      */
    int32_t last_path;       /* most recent search(): 0 = one query per pass, 1 = 8-query CSR scan, 2 = blocked postings
                              * (fp64 walk), 3 = blocked postings, fixed-point filter walk + exact refine               */
    int32_t last_fallbacks;  /* path 3: queries of the most recent search() whose top k the refine step could not prove from
                              * the filter's candidates and that were re-run on the exact walk (reading it synchronises)  */
    int64_t last_walk_postings; /* score terms (query weight x document value) the most recent walk accumulated: scatter-adds
                              * of posting lists + multiply-adds on the dense head strips                                 */
    int32_t head_columns;    /* columns of the blocked-postings copy kept as dense strips (option "postings_head")         */
    int32_t postings_state;  /* the blocked-postings copy: 0 = not attempted yet, 1 = built, 2 = NOT built: no room in HBM (sparse
                              * queries take the ~10x slower CSR scan), 3 = NOT built: a block holds more records than a directory
                              * word addresses, 4 = not wanted (small / short-row index, or option "blocked_postings" = 0)          */
    int32_t postings_walk;   /* which walk serves the filter on the copy that was built: 0 = list walk over 8-posting records,
                              * 4 = quad walk over 64-cell chunks (the default of a valued index), 5 = bag-of-token walk;
                              * -1 = no copy                                                                                     */
    int32_t last_packed_tiles; /* path 3 on bag-of-token chunks: query tiles of the most recent search() that took the packed walk (four slots
                              * on 16-bit sums, option "postings_packed"); the other tiles ran two int32 slots each                     */
} vs_index_info_t;

/* ---- library ------------------------------------------------------------------------------- */
VS_API int          vs_version(void);
VS_API const char*  vs_last_error(void);
VS_API int          vs_device_count(int32_t* out);

/* ---- index containers: Index / SparseIndex / BoTIndex (index.py:25-218) ---------------------- */

/* Builds the device-resident CSR index from a torch/scipy-style CSR (what SparseIndex.init_index
 * assembles at index.py:163-179 and build_index at retriever.py:299-312).  Inputs are copied; the
 * library owns its own device format (row-padded 8-nnz packets: uint16 column ids, fp32/fp16/no
 * values, uint32 packet row pointers).
 *   rowptr_dtype, col_dtype: VS_I32 | VS_I64.   val_dtype: VS_F32 | VS_F16; values == NULL -> binary.
 *   store_dtype: VS_F32 | VS_F16 | VS_NONE -- dtype streamed by search (fp16 = the reference's
 *   `fp16=True` load default, index.py:135,176; VS_NONE asserts every value == 1).              */
VS_API int vs_index_create_csr(const void* rowptr, int rowptr_dtype, const void* colidx, int col_dtype,
                               const void* values, int val_dtype, int store_dtype,
                               int64_t n_rows, int32_t n_cols, int device, vs_index** out);

/* Shard-by-shard construction: what SparseIndex.init_index does with `vstack(shards)` (index.py:172-175),
 * without ever holding the concatenation on the host.  Reserve capacity (rows, 8-nnz packets: a row of
 * `len` entries takes ceil(len / 8)), then append CSR row blocks in order; rows become searchable as they
 * are appended.                                                                                       */
VS_API int vs_index_create_reserved(int64_t rows_cap, int64_t packets_cap, int32_t n_cols, int store_dtype, int device, vs_index** out);
VS_API int vs_index_append_csr(vs_index* index, const void* rowptr, int rowptr_dtype, const void* colidx, int col_dtype,
                               const void* values, int val_dtype, int64_t n_rows);

/* Row-range sharding ("per-shard npz or row ranges"): rows [row0, row0 + n_rows) of a CSR index as a NEW
 * index on GPU `device` -- the packets are copied device to device (a peer copy across GPUs).  The reference joins its shard files
 * with vstack (index.py:172-175) and keeps ONE matrix on one device; SparseIndex(index_file=<one file>, devices=N) and
 * Retriever.build_index(..., devices=N) deal that one matrix out in contiguous row ranges with this call.                      */
VS_API int vs_index_slice_rows(const vs_index* index, int64_t row0, int64_t n_rows, int device, vs_index** out);

/* Native shard files (".vsx"): the device format written / read verbatim.  SparseIndex.init_index re-parses,
 * slices and vstacks scipy .npz shards on every load (index.py:172-176); a .vsx file is a header + the three
 * device arrays, so a 97 GB index loads at storage speed.                                                */
/* scipy.sparse.save_npz shards read natively (SparseIndex.init_index: load_npz(f)[:, shift:] per shard, then
 * vstack, index.py:172-175).  A .npz is a ZIP archive (stored or deflated members, ZIP64 for members beyond 4 GB) of .npy arrays
 * indptr / indices / data / shape / format; only format "csr" is read (anything else: VS_EUNSUPPORTED -- the Python facade then
 * falls back to scipy).  No GPU is touched by vs_npz_inspect.
 *   vs_npz_inspect: shape of the shard and what remains after the column shift -- columns below `shift` are dropped, n_cols
 *     is reported after the shift; `packets` = 8-nnz packets of the device format (what vs_index_create_reserved needs).
 *   vs_index_append_npz: appends the shard's rows to a reserved CSR index (columns shifted and sorted within a row; a binary
 *     index -- store VS_NONE -- requires every stored value == 1).                                                            */
VS_API int vs_npz_inspect(const char* path, int32_t shift, int64_t* n_rows, int64_t* n_cols, int64_t* nnz, int64_t* packets);
VS_API int vs_index_append_npz(vs_index* index, const char* path, int32_t shift);
/* SparseIndex.save (index.py:181-202): the index as a scipy.sparse.save_npz file that scipy.sparse.load_npz reads back -- CSR,
 * int64 indptr / indices (the reference's torch CSR has int64 ids), fp32 data (all 1 for a binary index).  compressed = 0:
 * stored members; 1: deflate (level 1).  ZIP64 records are written where a member or an offset passes 4 GB.                  */
VS_API int vs_index_save_npz(const vs_index* index, const char* path, int compressed);
VS_API int vs_index_save_native(const vs_index* index, const char* path);
VS_API int vs_index_load_native(const char* path, int device, vs_index** out);

/* Dense index (Index.vector = [n_rows, n_cols], index.py:25-44, retriever.py:292-297). */
VS_API int vs_index_create_dense(const void* mat, int dtype, int store_dtype, int64_t n_rows, int32_t n_cols,
                                 int64_t ld, int device, vs_index** out);

/* Sparsity-aware variant: when the matrix density is <= max_density (a dense index of VDR embeddings holds
 * <= 768 non-zeros per 29 523-wide row) the rows are stored as CSR packets and searched by the CSR scan --
 * identical dot products, ~2.6 % of the bytes, none of the dense flops.  vs_index_info still reports
 * VS_KIND_DENSE and vs_index_export_dense works.  max_density <= 0 == vs_index_create_dense.             */
VS_API int vs_index_create_dense_auto(const void* mat, int dtype, int store_dtype, int64_t n_rows, int32_t n_cols,
                                      int64_t ld, double max_density, int device, vs_index** out);

/* Synthetic corpus generated straight into the device format (bench / tests; no reference
 * counterpart).  Rows are the pure function of (seed, global row id) defined in
 * vsearch_amd/synth.py; this shard holds rows [row0, row0 + n_rows).
 *   kind: 0 = fixed `nnz` per row with values, uniform columns; 1 = bag-of-token lengths (binary);
 *         2 = fixed `nnz`, column popularity ~ 1 / rank (Zipf s = 1, the most popular ~127 columns in every row).  */
VS_API int vs_index_create_synthetic(uint64_t seed, int64_t row0, int64_t n_rows, int32_t n_cols, int32_t nnz,
                                     int kind, int val_law, int store_dtype, int device, vs_index** out);

/* Index.search (index.py:88-94):  q.to(device).type(vector.dtype); matmul(q, vector.t()); topk(k).
 *   q: dense [B, n_cols] row-major with leading dimension ldq (elements), q_dtype VS_F32 | VS_F16.
 *   out_ids [B,k] int64 (row ids + id_offset), out_scores [B,k] fp32, descending.
 *   Returns VS_ERANGE when k > n_rows (the reference raises RuntimeError there).                 */
VS_API int vs_index_search(vs_index* index, const void* q, int q_dtype, int64_t ldq, int32_t B, int32_t k,
                           int64_t id_offset, int64_t* out_ids, float* out_scores, void* stream);

/* Dense score matrix [B, n_rows] fp32 -- the intermediate index.py:91 materialises.  Used by the
 * parity tests to check every score, not just the top-k.                                         */
VS_API int vs_index_scores(vs_index* index, const void* q, int q_dtype, int64_t ldq, int32_t B,
                           float* out_scores, void* stream);

VS_API int  vs_index_info(const vs_index* index, vs_index_info_t* out);

/* Scan selection (tuning / tests; no reference counterpart): 0 = auto -- score tiles of 8 sparse queries
 * per pass over the index when every query is sparse enough for the LDS tile tables (512 ranks per
 * pass; larger k takes several passes), else one query per pass with a dense fp32 query image;
 * 1 = always the latter.                                                                          */
VS_API int  vs_index_set_queries_per_pass(vs_index* index, int qt);

/* Builds NOW what the first sparse search would otherwise build inside the call (the blocked-postings copy: 0.5 s at 21 M docs):
 * Index.move_to_device / load_index / build_index of the Python facade call it, so a user's first retrieve() pays nothing extra.
 * Idempotent; indexes that get no copy return at once.  vs_index_info_t.postings_state tells what happened.  stream as in
 * vs_index_search (NULL = blocking).                                                                                           */
VS_API int  vs_index_prepare(vs_index* index, void* stream);

/* Tuning / test options by name (no reference counterpart):
 *   "queries_per_pass"  as above
 *   "blocked_postings"  -1 = auto (long-row valued indexes, when HBM has room for the second copy), 0 = off, 1 = on:
 *                       sparse queries are scored from a row-blocked, column-grouped copy of the index that is built on
 *                       first use -- a query tile reads only the posting lists of its own columns
 *   "postings_rows"     0 = auto (a column's list in a block averages ~50 postings, at most 2048 documents; the bag-of-token chunks of a
 *                       binary index: ~18 postings, at most 8192), else documents per block of the copy (a multiple of 64 in 256..8192,
 *                       clipped to what the index's walk holds)
 *   "postings_chunks"   0 = auto, else the number of block runs the postings scan cuts the index into (work items = tiles x runs)
 *   "postings_filter"   1 (default) = the postings walk accumulates int32 fixed-point sums (3.4x the LDS atomic rate of fp64 on
 *                       MI355X) and returns k + max(28, k/4) candidates per query, which are re-scored with the exact numerics
 *                       and PROVEN to contain the top k; unproven queries re-run on the exact one-query scan of the CSR packets
 *                       (exact_scan_topk_kernel).  Results are identical to 0 = fp64 walk only
 *   "postings_quant"    1 (default) = an fp32 index keeps fp16-rounded values in the postings copy (the filter only ranks
 *                       candidates; the refine step re-scores them from the fp32 CSR), 0 = fp32 values there too
 *   "postings_walk"     which copy / kernel serves the filter: -1 = auto -- a valued index gets QUAD CHUNKS (bp_quad.h: 256-byte chunks of 64
 *                       postings per (block, column), walked by a generated asm loop) while they stay within 3 x the CSR bytes, else the
 *                       list walk over records; a binary index gets BAG-OF-TOKEN CHUNKS (bp_bq.h: 64-byte chunks of 32 document ids);
 *                       0 = the list walk over records (bp_walk.h: a list per 8-lane group), 4 = quad chunks whatever their size,
 *                       5 = the record walk of a binary index (bp_bin.h), 6 = bag-of-token chunks; 1 - 3 = the experimental walks of round 3
 *                       (flat worklists, two accumulator sets, streamed records: only in a `make EXPERIMENTAL=1` build).  A copy that does
 *                       not fit HBM falls back to the records, then to the CSR scan.  All return identical results; a change rebuilds the copy.
 *   "postings_head_gemm" -1 / 1 = the head columns' part of the filter sums comes from the head pre-pass (bp_head.h: one MFMA product per
 *                       pass of query tiles, columns in >= 1/8 of the documents, up to 1024; from 1 M documents on, HBM permitting: >= 1/16, up to 1536), 0 = multiplied inside the walk (round 4)
 *   "postings_head_tiles" 0 = auto, else query tiles per pass of the head pre-pass (its scratch: 32 KB per tile and block)
 *   "postings_head_product" the pre-pass's kernel: 1 = workgroups of 2 x 2 waves share a k-step's operands through an LDS ring (LDS-DMA,
 *                       four k-steps in flight), 0 = every wave loads its own, -1 = auto (the ring from 32 tiles a pass on).  Identical results
 *   "postings_packed"   -1 / 1 = the bag-of-token chunk walk puts FOUR query slots on the two sum planes of a tile (16-bit sums, two to a dword)
 *                       for the queries whose weights allow it (integer at a small power-of-two scale, longest row x largest weight < 65 536:
 *                       checked per query on the device); 0 = two int32 slots a tile.  Identical results
 *   "postings_pace"     lock-step window of the walk's work items in blocks (-1 / 0 = free running, the default)
 *   "postings_arrange"  1 = bank-aware order inside each posting list at build time (off by default: no measured gain)
 *   "postings_lanes"    0 = auto; valued index: lanes per posting list (4 | 8, auto 8); binary index: records in flight per lane (4 | 8, auto 8)
 *   "postings_align"    1 = posting lists start on whole 128-byte lines (16 % more bytes, ~2 % less walk time); 0 / -1 = packed
 *   "postings_head"     -1 = auto (4), 0 = off, N in 2..64: columns present in >= 1/N of the documents (at most 512, the most
 *                       frequent first) leave the posting lists and are kept as dense fp16 strips [block][column][document];
 *                       a query tile scores them with multiply-adds in registers (skewed vocabularies: a Zipf corpus has
 *                       3/4 of its non-zeros there).  Valued, non-negative indexes with the filter search; results unchanged
 *   "mq_variant"        -1 = auto (from the batch's query overlap), 0 = plain, 1 = shared-column variant of the 8-query scan */
VS_API int  vs_index_set_option(vs_index* index, const char* name, int value);

/* ---- row-sharded search in one process: one vs_index per GPU (the reference searches one index on one device,
 * its only sharding precedent is the per-shard index build, examples/inference_sparse/README.md:90-107, index.py:172-175) -----
 * `shards` are consecutive row ranges of one corpus (shard i holds rows [sum of n_rows of the shards before it, ...)), each on
 * its own device; the group does not own them.  vs_shard_group_search scores the batch on every GPU concurrently, moves the
 * B * k (id, score) pairs of each shard to the first shard's device with peer copies (xGMI), and merges there: the result is
 * identical to searching the unsharded index (global ids, canonical order).  q: host pointer or device pointer on any GPU;
 * outputs: host pointers or device pointers on the first shard's device.  Blocking.  The shards run on streams of the group: with a
 * device-resident q (or device outputs) the call first synchronises that device, so that work the caller has queued on ANY of its
 * streams (the encoder writing q, a kernel still reading a recycled output buffer) is done before the shards touch the buffers.
 * Fewer than 2^32 - 1 documents in total.
 * (One process per GPU with torch.distributed / RCCL uses vs_index_search's id_offset + vs_merge_topk instead:
 * vsearch_amd/distributed.py.)                                                                                                */
typedef struct vs_shard_group vs_shard_group;
VS_API int  vs_shard_group_create(vs_index* const* shards, int32_t n_shards, vs_shard_group** out);
VS_API int  vs_shard_group_search(vs_shard_group* group, const void* q, int q_dtype, int64_t ldq, int32_t B, int32_t k,
                                  int64_t* out_ids, float* out_scores);
VS_API void vs_shard_group_destroy(vs_shard_group* group);

/* SparseIndex.save (index.py:181-202) needs crow/col/values back: int64 rowptr [n_rows+1], int64
 * colidx [nnz], values [nnz] as val_dtype (VS_F32 | VS_F16).  Pass NULL colidx/values to get rowptr
 * only (to size the other two).                                                                  */
VS_API int  vs_index_export_csr(const vs_index* index, int64_t* rowptr, int64_t* colidx, void* values, int val_dtype);
VS_API int  vs_index_export_dense(const vs_index* index, void* mat, int dtype, int64_t ld);
VS_API void vs_index_destroy(vs_index* index);

/* Row-sharded search, merge step (new in this build): candidates gathered from
 * all shards ([B, n_cand] global ids + scores, e.g. after an RCCL all-gather) -> canonical top-k. */
/* (ids must be in [0, 2^32 - 1): the merge keys hold 32-bit ids; a candidate outside that range is dropped, never aliased) */
VS_API int vs_merge_topk(const int64_t* cand_ids, const float* cand_scores, int32_t B, int64_t n_cand, int32_t k,
                         int64_t* out_ids, float* out_scores, int device, void* stream);

/* ---- sparsify: src/ir/utils/sparse.py + the tail of VDREncoder.embed (vdr.py:152-169) -------- */

/* build_topk_mask (sparse.py:8-14): mask[b, c] = 1 iff x[b, c] is among the k largest of row b
 * (ties at the k-th value: lower column wins).  x: [B, V] fp32 device/host, mask: [B, V] uint8.    */
VS_API int vs_topk_mask(const float* x, int32_t B, int32_t V, int64_t ld, int32_t k, uint8_t* mask, int device, void* stream);

/* build_bow_mask (sparse.py:21-29): multi-hot of token ids over `vocab`, first `shift` columns
 * dropped, optional L2 row norm.  ids: [B, L] int64; out: [B, vocab - shift] fp32.               */
VS_API int vs_bow_mask(const int64_t* ids, int32_t B, int32_t L, int32_t vocab, int32_t shift, int norm,
                       float* out, int device, void* stream);

/* embed()'s mask stage (vdr.py:152-169), in place on emb [B, V = vocab - shift] (leading dim ld):
 *   bow != 0          -> emb = bow_mask(ids)
 *   topk == 0         -> keep only lexical dims;  topk < 0 -> keep all;  else keep top-k
 *   activate_lexical  -> mask |= bow_mask(ids)                                                   */
VS_API int vs_embed_mask(float* emb, int64_t ld, const int64_t* ids, int32_t B, int32_t L, int32_t vocab, int32_t shift,
                         int32_t topk, int activate_lexical, int bow, int device, void* stream);

/* The mask stage of VDREncoder.embed FUSED with Tensor.to_sparse_csr() (vdr.py:152-169 + retriever.py:304; SURVEY.md 8(f1) "write CSR
 * rows directly"): x [B, V] fp32 (device; NOT modified) -> the CSR of x * (topk_mask | lexical_mask): int64 rowptr [B + 1], int32 cols /
 * fp32 vals [cap] (device; cap >= B * (topk + L) always suffices; nnz = rowptr[B]).  One read of [B, V] instead of the five passes of
 * vs_embed_mask + vs_dense_to_csr.  VS_EUNSUPPORTED outside the fused kernel's range (topk <= 0, V > 32 Ki, topk + L > 8192): use
 * those two.                                                                                      */
VS_API int vs_embed_mask_to_csr(const float* x, int64_t ld, const int64_t* ids, int32_t B, int32_t L, int32_t vocab, int32_t shift,
                                int32_t topk, int activate_lexical, int64_t* rowptr, int32_t* cols, float* vals, int64_t cap,
                                int device, void* stream);

/* Tensor.to_sparse_csr() (retriever.py:304): non-zeros of dense x [B, V] as CSR -- int64 rowptr
 * [B+1], int32 cols / fp32 vals.  Two-call protocol: (1) cols == vals == NULL: rowptr is WRITTEN
 * (sizes the outputs: nnz = rowptr[B]); (2) cols / vals with cap >= nnz: the same rowptr is READ and
 * the non-zeros are written in column order.  build_index emits CSR batch by batch this way and
 * never holds the dense [N, V] matrix (retriever.py:281).                                        */
VS_API int vs_dense_to_csr(const float* x, int32_t B, int32_t V, int64_t ld, int64_t* rowptr, int32_t* cols, float* vals,
                           int64_t cap, int device, void* stream);

/* Encoder head tail (vdr.py:73-75): elu1p then max over L of logits [B, L, V] -> out [B, V]
 * (computed as elu1p(max) -- elu1p is monotone; pad positions are pooled like the reference).    */
VS_API int vs_head_pool(const float* logits, int32_t B, int32_t L, int32_t V, float* out, int device, void* stream);

/* Encoder head, fused (vdr.py:72-75): out[b, v] = elu1p(max_l hidden[b, l, :] . W[v, :]) on the fp32 matrix cores;
 * hidden [B, L, H] (LayerNorm'ed), W [V, H] (= word_embeddings[shift:]), out [B, V]: device pointers, H % 32 == 0.
 * The reference's [B, L, V] logits tensor is never materialised.                                          */
VS_API int vs_head_project_pool(const float* hidden, const float* W, int32_t B, int32_t L, int32_t H, int32_t V, float* out,
                                int device, void* stream);

/* VDREncoder.forward with pooling = "mean" and pooling_topk = t (vdr.py:76-79):
 * out[b, c] = mean over the t largest elu1p(logits[b, l, c]), l < L.  logits [B, L, V], out [B, V]: device pointers, t <= 32. */
VS_API int vs_head_pool_mean_topk(const float* logits, int32_t B, int32_t L, int32_t V, int32_t topk, float* out, int device, void* stream);

/* ---- rerank of bag-of-token hits: Retriever.retrieve(rerank=True) (retriever.py:137-147) --------
 * The reference re-embeds the B * k hit texts with encoder_p into a dense [B * k, V] tensor, takes torch.bmm against the query
 * embeddings, and topk(k) re-sorts every row.  Here the two steps are separate entry points so that the re-embedding can be
 * streamed in batches (the dense tensor is 12 GB at B = 1024, k = 100): vs_rerank_scores fills scores[row0 .. row0 + n_rows) of
 * the flat [B * k] score array from one batch of passage embeddings (row r of the batch is hit (row0 + r) % k of query
 * (row0 + r) / k; fp32 products, fp64 sums), vs_rerank_topk then orders every query's k hits by (score descending,
 * first-stage rank ascending) and gathers their ids.  Device pointers only; k <= 2048.                                    */
VS_API int vs_rerank_scores(const void* p_emb, int p_dtype, int64_t ldp, int64_t n_rows, int64_t row0, const float* q, int64_t ldq, int32_t B,
                            int32_t k, int32_t n_cols, float* scores, int device, void* stream);
VS_API int vs_rerank_topk(const float* scores, const int64_t* hit_ids, int32_t B, int32_t k, int64_t* out_ids, float* out_scores, int device,
                          void* stream);

/* elu1p (sparse.py:6) elementwise. */
VS_API int vs_elu1p(const float* x, int64_t n, float* out, int device, void* stream);

/* ---- bag-of-token builder: Retriever._build_bot_vectors (retriever.py:208-253) --------------- */
/* tokens: flat int32 token ids of all docs (already truncated by the tokenizer), offsets [n_docs+1].
 * Per doc: first-`max_token`-unique cap (0 = off; counts every id incl. [CLS], index_utils.py:11-21),
 * multi-hot, drop ids < shift, sorted columns.  Two-call protocol: out_cols == NULL fills only
 * out_rowptr (int64 [n_docs+1]).  Host pointers only (tokenisation is host work in the reference). */
VS_API int vs_bot_build(const int32_t* tokens, const int64_t* offsets, int64_t n_docs, int32_t vocab, int32_t shift,
                        int32_t max_token, int64_t* out_rowptr, int32_t* out_cols);

/* ---- measurement hooks (bench.py) ----------------------------------------------------------- */
/* When enabled, every launch of the scoring kernels is bracketed by hipEvents on the launch
 * stream; vs_profile_read returns accumulated device time and launch count per kernel name.      */
VS_API int vs_profile_enable(int on);
VS_API int vs_profile_reset(void);
VS_API int vs_profile_read(const char* kernel, double* total_ms, int64_t* launches);

#ifdef __cplusplus
}
#endif
#endif /* VSEARCH_HIP_H */
