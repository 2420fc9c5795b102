/* clair_host.h -- C ABI of the host-side helpers around the MI355X forward pass (libclair_host.so, plain C++, no HIP).
 *
 * These are the SURVEY.md section 8(f) "next" rows: the callers either side of the hot path that cap end-to-end
 * throughput once the network runs at millions of candidates per second.  Every function restates a piece of the
 * reference's Python (file:line below) and is pinned, byte for byte, against the build's Python counterpart in
 * clair_amd/ (which is itself pinned against fixtures minted from the real reference, tests/golden/).
 *
 * Convention: functions return 0 on success, non-zero on failure with a message in clair_host_last_error()
 * (thread-local).  Plain pointers and sizes only. */
#ifndef CLAIR_HOST_H
#define CLAIR_HOST_H
#include <stdint.h>
#include "clair_call.h"
#ifdef __cplusplus
extern "C" {
#endif

#define CLAIR_HOST_ABI_VERSION 6
#define CLAIR_HOST_VALUES 1056      /* 33 positions x 8 rows x 4 channels (shared/param.py:9-13) */

int clair_host_abi_version(void);
const char *clair_host_last_error(void);
/* worker threads a call over `work_items` lines / candidates uses: CLAIR_HOST_THREADS from the environment, else up to 16
 * hardware threads, one per 128 items (results do not depend on it: rows keep their input order) */
int clair_host_threads(int work_items);
/* CRC32C (Castagnoli) of a byte range -- the checksum of TensorFlow's bundle format (clair_amd/tf_bundle.py); "123456789" -> 0xE3069283 */
uint32_t clair_host_crc32c(const uint8_t *data, int64_t n);

/* raw pileup counts -> network input (clair/utils.py:96-98: X[:,:,:,1:] -= X[:,:,:,0:1] after the float32 conversion of :81-83):
 * n_quads groups of four channels (33 x 8 per candidate); x[4q] = c[4q], x[4q+k] = c[4q+k] - c[4q].  int16 and int32 counts. */
int clair_host_counts_to_input_i16(const int16_t *counts, int64_t n_quads, float *x);
int clair_host_counts_to_input_i32(const int32_t *counts, int64_t n_quads, float *x);

/* -- ingest: the parsing work of clair/utils.py:72-109 (tensor_generator_from) for one chunk of text ---------------
 * Record format (dataPrepScripts/CreateTensor.py:60-65): "ctg pos refseq33 v0 ... v1055", whitespace separated.
 * buf[0..len) is consumed line by line ('\n'; a last line without '\n' counts only when `final` is non-zero) until
 * max_rows lines have been TAKEN (utils.py:79 counts dropped rows too) or the complete lines run out.  Per line:
 *   - the last 1056 columns -> float32 (utils.py:81-83); exactly three columns must precede them (utils.py:86);
 *   - rows whose centre base refseq[16] is not an IUPAC code are dropped (utils.py:90-91, shared/utils.py:24-27);
 *   - kept row k: x[k*1056 ..] = values with channels 1..3 -= channel 0 (utils.py:96-98);
 *     tok[k*6 ..] = (offset, length) pairs of ctg, pos, refseq inside buf.
 * Outputs: *rows_taken, *rows_kept, *bytes_consumed (start of the first line not taken).
 * Errors (malformed line: too few columns, not three leading columns, refseq shorter than 17, unparsable value) name
 * the 0-based line index inside this chunk; the reference raises a Python exception at the same places. */
int clair_host_parse_tensors(const char *buf, int64_t len, int final, int max_rows,
                             float *x, int32_t *tok, int *rows_taken, int *rows_kept, int64_t *bytes_consumed);

/* -- decode: probabilities -> VCF rows, the work of clair/call_var.py:589-1236 (possible_outcome_probabilites_from, output_from,
 *    output_with, batch_output) for one batch, in the configuration the GPU pipeline runs in: no BAM look-ups available (every
 *    look-up answers "", the reference's own fall-back to tensor-inferred bases, :520-524, :562-564), no --debug, no
 *    --output_for_ensemble.  x [n][1056] (channels 1..3 already minus channel 0), the four softmax arrays [n][21|3|33|33];
 *    meta + meta_tok: per candidate the (offset, length) pairs of ctg, pos, refseq inside meta (the layout
 *    clair_host_parse_tensors produces).  show_reference / haploid_* = the CLI flags (:1402-1429); qual_threshold < 0 = no
 *    --qual (FILTER "."); arith_numpy2 selects how QUAL and AF are rounded (clair_amd/call_var.py "Arithmetic mode").
 *    Rows are appended to out, each terminated by '\n', in input order; candidates that produce no row are skipped exactly as
 *    the reference skips them.  Byte-identical to clair_amd.call_var.VariantDecoder (tests/test_host.py). */
int clair_host_decode_rows(const float *x, const float *gt21, const float *genotype, const float *len1, const float *len2,
                           const char *meta, const int32_t *meta_tok, int n, int show_reference, int haploid_precision,
                           int haploid_sensitive, int qual_threshold, int arith_numpy2, char *out, int64_t out_cap,
                           int64_t *out_len, int *n_rows);
/* The same with a per-candidate status byte (status[n], may be NULL): bit 0 = the candidate produced a row, bit 1 = its
 * resolution passed a point where the reference consults the BAM when it has one (an indel of 16 bases or more,
 * call_var.py:498-524, 540-565; the second allele of an Ins/Ins call, :805-823).  A caller that holds a BAM decodes exactly the
 * candidates with bit 1 on its own look-up path and splices their rows in (clair_amd/call_var.py: VariantDecoder.decode_batch);
 * for every other candidate the BAM cannot change the row. */
int clair_host_decode_rows_ex(const float *x, const float *gt21, const float *genotype, const float *len1, const float *len2,
                              const char *meta, const int32_t *meta_tok, int n, int show_reference, int haploid_precision,
                              int haploid_sensitive, int qual_threshold, int arith_numpy2, char *out, int64_t out_cap,
                              int64_t *out_len, int *n_rows, uint8_t *status);

/* The two halves of that decode, apart (include/clair_call.h describes the 32-byte record between them).
 * clair_host_resolve_calls: everything arithmetic -- the ten outcome families, the iterative arg-max, genotype, depth, supporting
 *   reads, the probability QUAL starts from, the tensor's vote on inserted bases -- for n candidates; centre[2i] = the reference
 *   window's centre character refseq[16], centre[2i + 1] = min(length of refseq, 255).  It is the CPU twin of the GPU decode kernel
 *   (include/clair_amd.h: clair_submit_ex) and the yardstick that kernel is compared with bit for bit.
 * clair_host_format_calls: records + the candidates' text -> rows, exactly the rows clair_host_decode_rows_ex writes for the same
 *   candidates (that function IS resolve + format per candidate); status as there.
 * clair_host_centre_bytes: the centre[] array of a batch from its meta table. */
int clair_host_resolve_calls(const float *x, const float *gt21, const float *genotype, const float *len1, const float *len2,
                             const uint8_t *centre, int n, clair_call_t *calls);
int clair_host_format_calls(const clair_call_t *calls, const char *meta, const int32_t *meta_tok, int n, int show_reference,
                            int haploid_precision, int haploid_sensitive, int qual_threshold, int arith_numpy2, char *out,
                            int64_t out_cap, int64_t *out_len, int *n_rows, uint8_t *status);
int clair_host_centre_bytes(const char *meta, const int32_t *meta_tok, int n, uint8_t *centre);
/* clair_host_format_calls from the columns of binary tensor records (clair_amd/tensor_binary.py): contig names [n][37] with their
 * lengths, positions, reference windows [n][33] with their lengths -- no text table is built for the batch.  Same rows. */
int clair_host_format_calls_records(const clair_call_t *calls, const char *ctg, const uint8_t *ctg_len, const int64_t *pos, const char *seq,
                                    const uint8_t *seq_len, int n, int show_reference, int haploid_precision, int haploid_sensitive,
                                    int qual_threshold, int arith_numpy2, char *out, int64_t out_cap, int64_t *out_len, int *n_rows,
                                    uint8_t *status);

/* -- pileup: alignments -> [33][8][4] count windows, the work of dataPrepScripts/CreateTensor.py:179-394 (OutputAlnTensor) and
 *    :29-65 (generate_tensor) as a streaming builder.  The caller supplies what the reference obtains from its sub-processes:
 *    the reference slice `samtools faidx` printed (upper-cased, :137; reference_start_0_based = 0 or region start - 1, :217), the
 *    candidate positions (column 2 of the candidate rows, 1-based, those outside [ctgStart, ctgEnd] already dropped, :86-92, in
 *    stream order) and the text `samtools view` prints.  The builder replays the reference's rules: candidates become known
 *    100 000 bp ahead of the reads (:274-275); mapping-quality filter (:268-269); at most dcov reads per start position
 *    (:277-284); windows open / close as the CIGAR walk passes centre-17 / centre+17 (:286-371; consider_left_edge: at any
 *    walked position inside, :95-100); a window is finished when a read with a new start position begins beyond it (:373-386)
 *    or at clair_host_pileup_finish (:388-394), in first-touch order, and dropped when its centre depth is below min_coverage or
 *    it would start before the loaded reference (:58-59); available_slots is the reference's budget of outstanding
 *    (window, base) tuples (5 000 000, :181): bases are dropped once it is used up, exactly where the reference drops them
 *    (within one reference position the windows are served in the order clair_host_pileup_set_order selects).  force_general_path != 0 selects the hash-map twin of the sorted-candidates fast path (tests).
 *    Pinned byte for byte against records minted from the real script (tests/golden/pileup_ct_*.json.gz). */
typedef struct clair_pileup clair_pileup_t;
int clair_host_pileup_create(const char *ref_seq, int64_t ref_len, int64_t reference_start_0_based, const int64_t *candidates,
                             int64_t n_candidates, int consider_left_edge, int dcov, int min_coverage, int min_mq,
                             int64_t available_slots, int force_general_path, clair_pileup_t **out);
void clair_host_pileup_destroy(clair_pileup_t *p);
/* The order in which a read base is offered to the windows open over it (CreateTensor.py:296-310), before the first alignment is fed.  It shows
 * in the records only where the budget of outstanding tuples runs out in the MIDDLE of one base.  0 (default): the order the windows were opened
 * in -- what an insertion-ordered set gives, i.e. the script under PyPy, the interpreter clair/callVarBam.py runs it with by default (--pypy);
 * 1: the iteration order of CPython's hash set, restated in host_pileup.cpp and pinned against records minted from the real script under
 * CPython with the budget binding (tests/golden/pileup_ct_budget_binds.json.gz). */
int clair_host_pileup_set_order(clair_pileup_t *p, int cpython_set);
/* test hook for that restatement: replay set operations (key >= 0: add(key); -(key + 1): remove(key)) -> the keys in iteration order */
int clair_host_pyset_order(const int64_t *ops, int64_t n_ops, int64_t *keys, int64_t capacity, int64_t *n_keys);
/* Consume SAM text line by line ('\n'; a last line without '\n' only when `final`): header lines ('@') are skipped, columns
 * FLAG, POS, MAPQ, CIGAR, SEQ are used (:252-263).  *bytes_consumed = start of the first line not consumed.  Errors (too few
 * columns, non-integer column, CIGAR longer than SEQ, position outside the loaded reference) name the 0-based line index since
 * creation; the reference raises a Python exception at the same places. */
int clair_host_pileup_feed(clair_pileup_t *p, const char *sam, int64_t len, int final, int64_t *bytes_consumed);
int clair_host_pileup_finish(clair_pileup_t *p);
int64_t clair_host_pileup_pending(const clair_pileup_t *p);   /* finished windows waiting to be taken */
/* Take up to max_rows finished windows: centres[k] (1-based), refseq[k*34 ..] (NUL-padded; 33 bases unless the loaded reference
 * ends inside the window, as the reference's slice :63), counts[k*1056 ..] = [33][8][4] int32. */
int clair_host_pileup_take(clair_pileup_t *p, int64_t max_rows, int64_t *centres, char *refseq, int32_t *counts, int64_t *n_taken);
/* Take finished windows as the reference's text records "ctg centre refseq v0 ... v1055\n" (:60-65), as many as fit in cap. */
int clair_host_pileup_take_text(clair_pileup_t *p, const char *ctg_name, char *out, int64_t cap, int64_t *out_len, int64_t *n_taken);
/* stats[0..3] = reads walked, windows open, slots left, 1 if the sorted-candidates path is in use */
int clair_host_pileup_stats(const clair_pileup_t *p, int64_t *stats);

/* -- candidates: alignments -> candidate sites, the work of dataPrepScripts/ExtractVariantCandidates.py:160-393 (make_candidates)
 *    in inference mode, streaming.  Per alignment of contig ctg_name (:279-281): mapping quality >= min_mq (:291), CIGAR not "*"
 *    and at least 55 % aligned (:143-157, 293); M/=/X bases are tallied per reference position under their IUPAC_base_to_ACGT base
 *    (N kept), an insertion / deletion counts once at the position before it (:298-316).  Positions before the current read's
 *    start are complete (:319): a position is a candidate when it lies in [ctg_start, ctg_end] (1-based; -1, -1 = no range) and in
 *    the bed intervals (0-based half-open, start == end widened by one, shared/interval_tree.py:30-32; n_bed = -1: no bed file),
 *    its reference base is an IUPAC code, depth = bases - I - D >= min_coverage, and either the most frequent tally is not the
 *    reference base or the second one reaches `threshold` of the depth (:357-371; ties keep the order A C G T I D N).
 *    Rows: "ctg pos refbase depth X n X n ... (7 pairs, descending)" (:379-383).  The training-set switches (--gen4Training,
 *    --var_fn, --outputProb) sample with Python's random module and are not restated. */
typedef struct clair_evc clair_evc_t;
int clair_host_evc_create(const char *ctg_name, const char *ref_seq, int64_t ref_len, int64_t reference_start_0_based,
                          int64_t ctg_start, int64_t ctg_end, const int64_t *bed_start, const int64_t *bed_end, int64_t n_bed,
                          double min_coverage, double threshold, int min_mq, clair_evc_t **out);
void clair_host_evc_destroy(clair_evc_t *e);
int clair_host_evc_feed(clair_evc_t *e, const char *sam, int64_t len, int final, int64_t *bytes_consumed);
int clair_host_evc_finish(clair_evc_t *e);
int64_t clair_host_evc_pending(const clair_evc_t *e);
int64_t clair_host_evc_reads(const clair_evc_t *e);      /* alignments that passed the filters (:295) */
int clair_host_evc_take(clair_evc_t *e, int64_t max_rows, int64_t *positions, int64_t *n_taken);   /* 1-based positions only */
int clair_host_evc_take_text(clair_evc_t *e, char *out, int64_t cap, int64_t *out_len, int64_t *n_taken);

/* -- packed alignments for the device front end (include/clair_reads.h; include/clair_amd.h: clair_frontend_*).  One pass over
 *    `samtools view` text replaces the line handling of BOTH reference stages (ExtractVariantCandidates.py:266-295 and
 *    CreateTensor.py:251-287): an alignment is kept when either stage would walk it and carries one flag bit per stage --
 *    the candidate search wants RNAME == ctg_name, MAPQ >= evc_min_mq, CIGAR != "*" and >= 55 % aligned; the pileup wants
 *    MAPQ >= pile_min_mq, at most dcov alignments per start position (:277-287) and, when pile_start/pile_end (1-based inclusive,
 *    -1 -1 = none) are given, an overlap with that range as `samtools view ctg:start-end` computes it (the reference gives the two
 *    stages different regions, callVarBam.py:124-199).  Errors as clair_host_pileup_feed.  The slab grows until it is taken:
 *    ..._slab lends the arrays (valid until the next feed / reset), ..._reset starts the next slab; the dcov and sortedness state
 *    runs on across slabs.  stats[0..7] = alignments, operations, elements, SEQ bytes in the slab; CLAIR_FE_* bits seen so far;
 *    lines, candidate-search alignments, pileup alignments since creation. */
typedef struct clair_sampack clair_sampack_t;
struct clair_read;
struct clair_op;
int clair_host_sampack_create(const char *ctg_name, int dcov, int evc_min_mq, int pile_min_mq, int64_t pile_start, int64_t pile_end,
                              clair_sampack_t **out);
void clair_host_sampack_destroy(clair_sampack_t *p);
int clair_host_sampack_feed(clair_sampack_t *p, const char *sam, int64_t len, int final, int64_t *bytes_consumed);
int clair_host_sampack_stats(const clair_sampack_t *p, int64_t *stats);
int clair_host_sampack_slab(const clair_sampack_t *p, const struct clair_read **reads, const struct clair_op **ops, const uint32_t **op_elem,
                            const uint8_t **seq);
int clair_host_sampack_reset(clair_sampack_t *p);
/* CreateTensor.py's count of free tuple slots (:181, 283-289, 369-373) replayed from what the device counted: alignments in stream
 * order with the tuples each appended, candidate centres ascending with the tuples their windows held when released.  state[0] =
 * free slots (start: 5 000 000), state[1] = first centre not released yet (start: 0); carried from slab to slab.  *binds = 1 when
 * the count would have reached zero: the reference then drops bases in offer order and only the sequential code reproduces it. */
int clair_host_tuple_budget_binds(const struct clair_read *reads, const uint64_t *tuples, int64_t n_reads, const int64_t *centres,
                                  const uint64_t *window_tuples, int64_t n_centres, int64_t *state, int *binds);

#ifdef __cplusplus
}
#endif
#endif
