/*
 * clair_amd.h -- C ABI of the MI355X (gfx950) engine for Clair's call_var forward pass.
 *
 * The reference has no FFI: its boundary for this path is the Python class
 * clair.model.Clair (/root/reference/clair/model.py:24) as driven by
 * clair/call_var.py:213-215, 1337, 1343.  Each entry point below names the reference
 * member it stands behind; clair_amd/model.py is the ctypes shim that re-creates the
 * Python interface on top of it (INTEGRATION.md shows the binding).
 *
 * Conventions: plain pointers and sizes only; every function returns 0 on success and a
 * non-zero code on failure, with a message available from clair_last_error(); float32
 * everywhere (clair/model.py:167-168 forces tf.float32 for the LSTM structure).
 * Functions taking an engine are not re-entrant on one engine (the reference never calls
 * predict concurrently with itself, clair/call_var.py:1349-1352) but may be called from
 * any thread: the HIP device is selected on every call.
 */
#ifndef CLAIR_AMD_H
#define CLAIR_AMD_H

#include <stdint.h>
#include "clair_call.h"

#ifdef __cplusplus
extern "C" {
#endif

/* 1: round-1 surface.  2: + clair_slot_input, clair_submit_counts, clair_kernel_workgroups (added late in round 1 without a
 * bump), the clair_comm_* communicator (round 2).  3: + clair_engine_counter, clair_comm_preflight, clair_submit_ex, clair_decode, clair_pinned_alloc / _free and kernel id CLAIR_K_DECODE (round 3).
 * 4: + the clair_frontend_* device front end; clair_submit_ex takes device pointers (round 3).
 * 5: + clair_device_pci_bus_id (round 5: a rank finds the NUMA node of ITS GPU, clair_amd/shard.py), clair_comm_abort.
 * 6: + clair_comm_create_timed (round 6: a hung RCCL bring-up ends at a deadline, not at the job's). */
#define CLAIR_ABI_VERSION 6

/* geometry: shared/param.py:9-11 (33 x 8 x 4 input), clair/task/main.py:10-29 (head sizes) */
#define CLAIR_POSITIONS 33
#define CLAIR_FEATURES 32
#define CLAIR_INPUT_FLOATS (CLAIR_POSITIONS * CLAIR_FEATURES) /* 1056 */
#define CLAIR_GT21 21
#define CLAIR_GENOTYPE 3
#define CLAIR_INDEL_LEN 33
#define CLAIR_OUTPUT_FLOATS 90

/* Weight tensors of the inference graph (clair/model.py:400-620), row-major, float32.
 * TF variable names: clair_amd/weights.py:tf_variable_names(). */
enum clair_tensor_id {
    CLAIR_T_LSTM1_FW_KERNEL = 0, /* [160,512]  rows 0..31 multiply x, 32..159 multiply h; cols i|c~|f|o */
    CLAIR_T_LSTM1_FW_BIAS = 1,   /* [512] */
    CLAIR_T_LSTM1_BW_KERNEL = 2,
    CLAIR_T_LSTM1_BW_BIAS = 3,
    CLAIR_T_LSTM2_FW_KERNEL = 4, /* [384,512] */
    CLAIR_T_LSTM2_FW_BIAS = 5,
    CLAIR_T_LSTM2_BW_KERNEL = 6,
    CLAIR_T_LSTM2_BW_BIAS = 7,
    CLAIR_T_L3_KERNEL = 8,       /* [256,33,30]  L3/Unit_c/kernel stacked over c */
    CLAIR_T_L3_BIAS = 9,         /* [256,30] */
    CLAIR_T_L4_KERNEL = 10,      /* [7680,192]  input index u*256+c */
    CLAIR_T_L4_BIAS = 11,        /* [192] */
    CLAIR_T_L5_KERNEL = 12,      /* [4,192,96]  L5_1..L5_4 */
    CLAIR_T_L5_BIAS = 13,        /* [4,96] */
    CLAIR_T_HEAD_GT21_KERNEL = 14,     /* [96,21] */
    CLAIR_T_HEAD_GT21_BIAS = 15,
    CLAIR_T_HEAD_GENOTYPE_KERNEL = 16, /* [96,3] */
    CLAIR_T_HEAD_GENOTYPE_BIAS = 17,
    CLAIR_T_HEAD_LEN1_KERNEL = 18,     /* [96,33] */
    CLAIR_T_HEAD_LEN1_BIAS = 19,
    CLAIR_T_HEAD_LEN2_KERNEL = 20,     /* [96,33] */
    CLAIR_T_HEAD_LEN2_BIAS = 21,
    CLAIR_T_COUNT = 22
};

/* kernels of one forward pass, in launch order (index into clair_kernel_times) */
enum clair_kernel_id {
    CLAIR_K_PROJ1 = 0,  /* (unused since the LSTM1 input projection is fused into CLAIR_K_LSTM1) */
    CLAIR_K_LSTM1 = 1,  /* LSTM1: input projection + recurrence, both directions */
    CLAIR_K_PROJ2 = 2,  /* LSTM2 input projection GEMM  [33n,256]x[256,1024] */
    CLAIR_K_LSTM2 = 3,  /* LSTM2 recurrence                                  */
    CLAIR_K_L3 = 4,     /* (unused: slice dense is fused into CLAIR_K_L4)    */
    CLAIR_K_L4 = 5,     /* slice dense 256 x (33->30) + selu, split-K GEMM 7680->192 */
    CLAIR_K_TAIL = 6,   /* L4 reduce+selu, L5 x4, heads, selu, softmax       */
    CLAIR_K_DECODE = 7, /* probabilities -> call records (clair_submit_ex only) */
    CLAIR_K_COUNT = 8
};

typedef struct clair_engine clair_engine_t;

/* -- lifetime: Clair() + Clair.init()  (clair/model.py:58-192, 807-813) ------------------------
 * device: HIP device ordinal.  max_batch: largest n accepted by one predict/submit.
 * n_slots: submits that may be pending at once (clair_submit* .. clair_wait); 1 is enough for the synchronous predict.  A slot owns the
 * input and output buffers of its batch on both sides of the host link.  The forward passes themselves run on compute LANES (a HIP stream
 * + the inter-kernel workspaces each): one per slot up to four slots (the process has four hardware queues), three for a handle with more
 * slots (the fourth queue is the incoming copy stream's), slot s on lane s % lanes; with twice as
 * many slots as lanes a lane computes one batch of its slots while the copy engine brings the other one in, which is what takes the
 * host-array boundary from half of the HBM-resident rate to within a few per cent of it.  clair_run_resident(slot)
 * runs on the slot's lane. */
int clair_engine_create(int device, int max_batch, int n_slots, clair_engine_t **out);
/* Clair.close() / __del__  (clair/model.py:872-876, 1149-1152) */
void clair_engine_destroy(clair_engine_t *e);
/* message of the last failure on this engine (e may be NULL: failure of clair_engine_create) */
const char *clair_last_error(const clair_engine_t *e);
int clair_abi_version(void);
/* number of HIP devices visible (0 when there is none); negative never */
int clair_device_count(void);
/* "dddd:bb:dd.f" of HIP device `device` into buf (NUL-terminated; len >= 16): the name of its directory under
 * /sys/bus/pci/devices, where numa_node and local_cpulist say which host cores sit next to it.  The reference pins its
 * stages to cores with taskset (clair/callVarBam.py:103-115); a rank of this library pins itself to the cores of its GPU's
 * NUMA node (clair_amd/shard.py: bind_to_gpu).  0 on success; 1 when there is no such device (buf[0] = 0). */
int clair_device_pci_bus_id(int device, char *buf, int len);

/* -- weights: Clair.restore_parameters()  (clair/model.py:1016-1020) ----------------------------
 * Hand over one tensor (host pointer, `count` floats, shape as in enum clair_tensor_id);
 * after all CLAIR_T_COUNT tensors are set, clair_finalize_weights packs them into the
 * device layouts the kernels read.  The host pointers are not retained. */
int clair_set_tensor(clair_engine_t *e, int tensor_id, const float *host, int64_t count);
int clair_finalize_weights(clair_engine_t *e);

/* -- Clair.predict(batchX)  (clair/model.py:946-966; called at clair/call_var.py:1343) ----------
 * x: host, C-contiguous [n,33,8,4] float32 (channels 1..3 already minus channel 0,
 * clair/utils.py:96-98); 1 <= n <= max_batch.  Outputs: caller-allocated host arrays
 * [n,21] [n,3] [n,33] [n,33].  Synchronous; x is not retained. */
int clair_predict(clair_engine_t *e, const float *x, int n, float *gt21, float *genotype,
                  float *indel_len1, float *indel_len2);

/* -- pipelined form of the same call (what call_var's load/predict/output threads overlap,
 *    clair/call_var.py:1331-1352): submit copies x to the device and enqueues the forward pass on
 *    slot `slot`; wait blocks until that slot's outputs are in the caller's arrays.
 *    x and the output arrays must stay valid until wait returns. */
int clair_submit(clair_engine_t *e, int slot, const float *x, int n, float *gt21, float *genotype,
                 float *indel_len1, float *indel_len2);
/* (A batch in pageable memory is copied to the slot's page-locked buffer, and its transfers and kernels are enqueued, by a staging
 * thread of the engine: clair_submit returns at once and clair_wait reports a failure of that work.  This is why x must stay valid
 * until clair_wait, as the reference's predict thread leaves loading and output to two others, clair/call_var.py:1331-1352.) */
int clair_wait(clair_engine_t *e, int slot);
/* The slot's own page-locked input buffer, [max_batch][33][8][4] float32.  A producer that writes its batch there and passes
 * this pointer as `x` to clair_submit gets a direct DMA transfer (pageable memory goes through the runtime's staging copies at
 * about a third of the PCIe rate).  The buffer belongs to the handle; it may be refilled once clair_wait(slot) has returned. */
int clair_slot_input(clair_engine_t *e, int slot, float **x_pinned);
/* The pipelined call for a producer that holds the RAW pileup counts (dataPrepScripts/CreateTensor.py:29-65: what the text
 * records carry before clair/utils.py:96-98 subtracts channel 0 from channels 1..3): counts [n][33][8][4] int16, half the bytes
 * of the float32 tensor on the host link, which is what bounds the host-buffer boundary.  The subtraction
 * and the conversion run on the device; the results are bit-identical to clair_submit on the float32 tensor utils.py would
 * build from the same counts (every count is exactly representable).  Counts above 32767 do not fit: the caller checks. */
int clair_submit_counts(clair_engine_t *e, int slot, const int16_t *counts, int n, float *gt21, float *genotype,
                        float *indel_len1, float *indel_len2);

/* The pipelined call with the DECODE on the device.  What call_var does with the probabilities of a batch -- the ten outcome families
 * (1 179 float32 products, clair/call_var.py:589-690), the iterative arg-max with exact-equality membership tests (:693-947), genotype,
 * depth, supporting reads and the probability QUAL starts from (:1002-1166) -- runs as one more kernel behind the forward pass, and what
 * comes back per candidate is the 32-byte call record of include/clair_call.h instead of (or besides) the 360 bytes of probabilities.
 * clair_host_format_calls (include/clair_host.h) turns records into VCF rows; clair_host_resolve_calls is the CPU twin of the kernel
 * (bit-identical records).  input: [n][33][8][4] float32 as for clair_submit (input_is_counts == 0) or raw int16 counts as for
 * clair_submit_counts (!= 0); input_stride_bytes = distance between consecutive candidates in the caller's buffer (0 = dense), so that the
 * counts can be taken straight out of an array of binary tensor records (clair_amd/tensor_binary.py: 2 192-byte records) without a
 * host-side copy to make them contiguous.  centre: [n][2] bytes per candidate -- the centre character of its reference window (refseq[16],
 * clair/call_var.py:1015) and min(length of refseq, 255); required when calls != NULL.  calls: caller's array of n records, or NULL.
 * gt21 / genotype / indel_len1 / indel_len2: all four or all NULL (NULL: the probabilities stay on the device).  Pair with
 * clair_wait(slot); buffers must stay valid until it returns.  `input` may also be a DEVICE address of int16 counts (the windows
 * clair_frontend_build_windows leaves in HBM, clair_frontend_counts_device): nothing is copied then. */
int clair_submit_ex(clair_engine_t *e, int slot, const void *input, int input_is_counts, int64_t input_stride_bytes, int n,
                    const uint8_t *centre, clair_call_t *calls, float *gt21, float *genotype, float *indel_len1, float *indel_len2);

/* Page-locked host memory the DMA engine can read in place.  A producer that fills such a buffer -- e.g. reads binary tensor records
 * from a file straight into it -- and passes a pointer INTO it as `input` of clair_submit_ex (with the records' stride) gets the
 * batch to the GPU without any pass over it on the submitting thread: the transfer is a strided 2-D copy from where the data lies.
 * The buffer must not be rewritten before clair_wait of the submit that read it has returned.  Freed by clair_pinned_free (which
 * waits for the handle's streams) or with the engine. */
int clair_pinned_alloc(clair_engine_t *e, int64_t bytes, void **ptr);
int clair_pinned_free(clair_engine_t *e, void *ptr);

/* The decode alone, on probabilities the caller already holds (call_var --input_probabilities, clair/call_var.py:1276-1309, and the
 * tests that feed the kernel crafted probabilities: exact ties, exact zeros, products that underflow).  Synchronous; does not need
 * weights.  x: the candidates' network input [n][33][8][4] float32 (the decode reads depth, supporting reads and the votes on
 * inserted bases from it). */
int clair_decode(clair_engine_t *e, int slot, const float *x, const float *gt21, const float *genotype, const float *indel_len1,
                 const float *indel_len2, int n, const uint8_t *centre, clair_call_t *calls);

/* -- device-resident candidate sets (benchmark / multi-GPU shard driver) ------------------------
 * The candidate set lives in HBM: x_dev [N,33,8,4]; outputs out_dev [N,90] rows laid out
 * gt21(21) | genotype(3) | len1(33) | len2(33).  clair_run_resident enqueues the forward pass
 * for candidates [first, first+n) on slot `slot` (no host copies) and returns immediately;
 * clair_sync waits for all slots. */
int clair_dataset_alloc(clair_engine_t *e, int64_t n_candidates, void **x_dev, void **out_dev);
int clair_dataset_free(clair_engine_t *e, void *x_dev, void *out_dev);
int clair_dataset_upload(clair_engine_t *e, void *x_dev, int64_t first, const float *x_host, int64_t n);
int clair_dataset_download(clair_engine_t *e, const void *out_dev, int64_t first, float *out_host, int64_t n);
int clair_run_resident(clair_engine_t *e, int slot, const void *x_dev, void *out_dev, int64_t first, int n);
int clair_sync(clair_engine_t *e);

/* -- measurement ---------------------------------------------------------------------------------
 * When enabled, every kernel launch is bracketed by HIP events on the slot's own stream.
 * clair_kernel_times returns, per kernel id, the summed duration in milliseconds and the number
 * of launches since the last reset (it synchronises first).
 * on: 0 = off, 1 = every kernel, any other value = bit mask over enum clair_kernel_id (bit k = kernel k; bit 0 names an
 * unused id, so 1 is unambiguous): a pass that times ONE kernel adds two marker packets per forward pass instead of ten. */
int clair_timing_enable(clair_engine_t *e, int on);
int clair_kernel_times(clair_engine_t *e, double *ms_sum /*[CLAIR_K_COUNT]*/, int64_t *launches /*[CLAIR_K_COUNT]*/);
int clair_timing_reset(clair_engine_t *e);
/* Launch geometry: the number of 256-thread workgroups each kernel id is launched with for a batch of n candidates on this
 * handle (0 for ids that launch nothing).  The recurrent kernels and, on handles with several slots, the projection GEMM are
 * sized to PART of the chip so that the batches in flight on other slots run beside them; a per-kernel roofline needs that share. */
int clair_kernel_workgroups(clair_engine_t *e, int n, int *workgroups /*[CLAIR_K_COUNT]*/);

/* Event counters of the handle.  which: 0 = forward passes launched with the fused layer-2 kernel (one launch for the LSTM2
 * projection and recurrence, used on handles with one or two slots), 1 = forward passes that were RE-RUN on the two-launch path
 * because a fused launch reported that its workgroups were not placed as it assumes.  A re-run is invisible to the caller
 * (same arithmetic, same outputs; the reference never drops a batch, clair/call_var.py:1331-1352) except through this counter
 * and one line on stderr; after the first one the handle stays on the two-launch path. */
int clair_engine_counter(clair_engine_t *e, int which, int64_t *value);

/* -- layer taps for parity tests: copy an intermediate of the LAST forward pass run on `slot`
 *    to the host.  which: 1 = LSTM1 output [33,n_pad,256], 2 = LSTM2 output [33,n_pad,256],
 *    3 = split-K partials of the L4 product [8,n_pad,192], 4 = L3 output [n_pad,7680] (only when the engine was created
 *    with CLAIR_AMD_TAP_L3=1 in the environment)
 *    (L3/L4 activations only ever exist in LDS / split-K partials).  n_pad = n rounded up to 32. */
int clair_debug_read(clair_engine_t *e, int slot, int which, float *host, int64_t count);

/* -- multi-GPU: one process per GPU, candidates shard in contiguous blocks of whole batches -------------------------------
 * The reference scales out by running one callVarBam per 10 Mbp chunk under GNU parallel and concatenating the chunk VCFs
 * (clair/callVarBamParallel.py:90-119, README.md:297-303); candidates are classified independently
 * (docs/POST_PROCESSING.md:17).  Here rank r owns a contiguous block of the candidate stream (clair_amd/shard.py) and the
 * forward pass needs NO collective.  This communicator is a thin binding of RCCL (xGMI between the GPUs of a node) for
 * the trivial parts only: the weight blob from rank 0 (9.5 MB, once), the gather of per-rank output rows, counters and timers.
 * librccl.so is loaded on the first call.  Rendezvous: rank 0 calls clair_comm_unique_id and hands the 128 bytes to the
 * other ranks out of band (clair_amd/shard.py uses a socket on 127.0.0.1); every rank then calls clair_comm_create
 * (collective: returns once all `world` ranks have joined).  All calls are blocking; host buffers are staged through HBM. */
#define CLAIR_COMM_ID_BYTES 128
enum clair_comm_op { CLAIR_COMM_SUM = 0, CLAIR_COMM_MAX = 1, CLAIR_COMM_MIN = 2 };
typedef struct clair_comm clair_comm_t;
/* Everything clair_comm_create needs that can be checked WITHOUT the other ranks: a HIP device with this ordinal that accepts an
 * allocation, librccl.so loadable with the entry points bound.  The ranks exchange the result out of band before any of them
 * enters the collective ncclCommInitRank, so one rank's failure is an error on every rank instead of a hang on the others. */
int clair_comm_preflight(int device);
int clair_comm_unique_id(uint8_t *id /*[CLAIR_COMM_ID_BYTES]*/);                 /* ncclGetUniqueId */
int clair_comm_create(int device, int rank, int world, const uint8_t *id, clair_comm_t **out);   /* ncclCommInitRank */
void clair_comm_destroy(clair_comm_t *c);
/* Tear-down that does not wait for the peers (ncclCommAbort): for a communicator that came up on this rank while a peer's
 * clair_comm_create failed, and for the helper thread of clair_comm_create_timed when RCCL returns after its deadline. */
void clair_comm_abort(clair_comm_t *c);
/* clair_comm_create with a deadline (round 6; ABI 6): ncclCommInitRank AND the first collective on the new communicator (one
 * all-reduce: RCCL connects its transports lazily) run on a helper thread; 0 = up and proven, 1 = failed (clair_comm_last_error(NULL)),
 * CLAIR_COMM_TIMED_OUT = neither returned within `timeout_ms`.  On a time-out no communicator exists for the caller: the helper thread
 * is abandoned and aborts its communicator should RCCL ever return (until then a thread of this process sits inside librccl: leave the
 * process with _exit once the results are written -- exit() would run librccl's static destructors under it).  clair_amd/shard.py then tells the peers over the bootstrap
 * sockets and every rank goes on over the socket transport -- a hung RCCL bring-up still yields the N-rank line (`rccl_failure`). */
#define CLAIR_COMM_TIMED_OUT 2
int clair_comm_create_timed(int device, int rank, int world, const uint8_t *id, int timeout_ms, clair_comm_t **out);
const char *clair_comm_last_error(const clair_comm_t *c);                         /* c may be NULL: failure of create / unique_id */
int clair_comm_barrier(clair_comm_t *c);
int clair_comm_allreduce_f64(clair_comm_t *c, double *values /*in place*/, int count, int op /*enum clair_comm_op*/);
int clair_comm_broadcast(clair_comm_t *c, void *host, int64_t bytes, int root);  /* ncclBroadcast of a host buffer (the weight blob) */
int clair_comm_allgather(clair_comm_t *c, const void *send_host, void *recv_host /*[world][bytes_per_rank]*/, int64_t bytes_per_rank);
/* the same on HBM-resident buffers (e.g. the out_dev rows of clair_dataset_alloc), no host staging */
int clair_comm_allgather_device(clair_comm_t *c, const void *send_dev, void *recv_dev, int64_t bytes_per_rank);

/* ---- front end on the device: alignments -> candidate sites -> pileup windows that stay in HBM -----------------------------------
 *
 * Stands behind the two pypy stages callVarBam pipes into call_var (clair/callVarBam.py:124-199):
 *   dataPrepScripts/ExtractVariantCandidates.py:160-393 (make_candidates)   -> clair_frontend_find_candidates
 *   dataPrepScripts/CreateTensor.py:173-388 (OutputAlnTensor), :29-65       -> clair_frontend_build_windows
 *   clair/utils.py:90-98 (centre-base filter; the channel subtraction is clair_submit_ex's, input_is_counts)
 * Input: slabs of packed alignments (include/clair_reads.h, produced from `samtools view` text by clair_host_sampack_*), the
 * reference bases both stages index (`samtools faidx` of the region widened by 1 Mbp, upper-cased; reference_start_0_based = 0-based
 * position of its first base), and the span of 0-based reference positions [span_lo, span_hi) the per-position tables cover
 * (alignment bases outside it are ignored: make it the region plus a margin of >= 64).  Output: n windows of [33][8][4] int16 counts
 * in device memory (clair_frontend_counts_device -> clair_submit_ex with input_is_counts = 1), their centres (1-based) and the 33
 * reference bases under them.  Windows are the ones the sequential stages write, in ascending order, bit for bit, PROVIDED no
 * CLAIR_FE_* bit (clair_reads.h) is set in stats[0] and the tuple budget did not bind (clair_frontend_budget_inputs ->
 * clair_host_tuple_budget_binds); otherwise the caller runs clair_host_evc_* / clair_host_pileup_*, which reproduce the reference
 * in every regime.
 * Calls on one handle are not concurrent; all are synchronous. */
typedef struct clair_frontend clair_frontend_t;
struct clair_read;
struct clair_op;
int clair_frontend_create(int device, const char *ref_seq, int64_t ref_len, int64_t reference_start_0_based, int64_t span_lo, int64_t span_hi,
                          clair_frontend_t **out);
void clair_frontend_destroy(clair_frontend_t *f);
const char *clair_frontend_last_error(const clair_frontend_t *f);      /* f may be NULL: failure of create */
/* Copy one slab to the device (it stays there) and add its bases to the per-position tables.  The arrays may be reused on return.
 * The alignments of a slab are expected in ascending order of pos0 (what `samtools view` of a sorted BAM prints and both packers
 * keep); a slab whose starts decrease is still tallied correctly (by the order-independent per-base kernel) and sets
 * CLAIR_FE_UNSORTED in stats[0] of clair_frontend_stats, like the packers do: the reference's scripts would have stopped there. */
int clair_frontend_add_reads(clair_frontend_t *f, const struct clair_read *reads, int64_t n_reads, const struct clair_op *ops, int64_t n_ops,
                             const uint32_t *op_elem, const uint8_t *seq, int64_t seq_bytes);
/* ... or hand over the `samtools view` TEXT and let the device do the packing as well: the line handling of both scripts
 * (ExtractVariantCandidates.py:266-295, CreateTensor.py:251-287) one thread per line, exactly what clair_host_sampack_* produces
 * (arguments of _text_options as clair_host_sampack_create); of the text only the SEQ columns of the kept alignments stay on the device.  `sam` holds whole
 * lines (the caller keeps an unfinished last line for the next call), at most 2 GB at a time; the --dcov and sortedness state runs on
 * across calls.  Returns 2 when a line is malformed (too few columns, a non-integer FLAG / POS / MAPQ): clair_host_sampack_feed on
 * the same text reports it the way the host path does.  _text_stats: stats[0..3] = lines, candidate-search alignments, pileup
 * alignments since _text_options, CLAIR_FE_* bits of the line handling (also folded into clair_frontend_stats).  _slab_reads copies
 * a slab's alignment records back (the budget replay walks them); reads == NULL only asks for the count. */
int clair_frontend_text_options(clair_frontend_t *f, const char *ctg_name, int dcov, int evc_min_mq, int pile_min_mq, int64_t pile_start, int64_t pile_end);
int clair_frontend_add_text(clair_frontend_t *f, const char *sam, int64_t len);
int clair_frontend_text_stats(clair_frontend_t *f, int64_t *stats);
int clair_frontend_slab_reads(clair_frontend_t *f, int64_t slab, struct clair_read *reads, int64_t capacity, int64_t *n_reads);
/* The candidate filter over the tallies: arguments as clair_host_evc_create (include/clair_host.h). */
int clair_frontend_find_candidates(clair_frontend_t *f, double min_coverage, double threshold, int64_t ctg_start, int64_t ctg_end,
                                   const int64_t *bed_start, const int64_t *bed_end, int64_t n_bed, int64_t *n_candidates);
/* ... or a given list (--vcf_fn; 1-based, strictly ascending, else CLAIR_FE_CANDIDATES); positions outside the span are dropped. */
int clair_frontend_set_candidates(clair_frontend_t *f, const int64_t *positions, int64_t n_positions, int64_t *n_candidates);
int clair_frontend_get_candidates(clair_frontend_t *f, int64_t *positions /*[n_candidates]*/);
/* Second pass over the resident alignments + assembly.  min_coverage: CreateTensor's --minCoverage (depth at the centre);
 * drop_non_iupac_centre != 0 applies clair/utils.py:90-91 as well. */
int clair_frontend_build_windows(clair_frontend_t *f, int min_coverage, int drop_non_iupac_centre, int64_t *n_windows);
/* The same with CreateTensor's --stop_consider_left_edge (consider_left_edge = 0, CreateTensor.py:103-104): a read opens a window only
 * by walking its first column, so a read that starts inside a window adds nothing to it; the per-position tables hold such reads too
 * and what they added is collected per window and taken out again. */
int clair_frontend_build_windows_ex(clair_frontend_t *f, int min_coverage, int drop_non_iupac_centre, int consider_left_edge, int64_t *n_windows);
int clair_frontend_window_info(clair_frontend_t *f, int64_t first, int64_t n, int64_t *centres, char *refseq /*[n][34], NUL-padded*/);
int clair_frontend_window_counts(clair_frontend_t *f, int64_t first, int64_t n, int16_t *counts /*[n][33][8][4], host*/);
const int16_t *clair_frontend_counts_device(clair_frontend_t *f, int64_t first);   /* device address of window `first`; NULL before build_windows */
/* What the budget replay needs: tuples appended per alignment of slab `slab` (read_tuples, may be NULL), all candidate centres and
 * the tuples each window held (0 for a window no alignment opened); either pair may be NULL. */
int clair_frontend_budget_inputs(clair_frontend_t *f, int64_t slab, uint64_t *read_tuples, int64_t *centres, uint64_t *window_tuples);
/* stats[0..5] = CLAIR_FE_* bits seen on the device, slabs, alignments, elements, candidates (-1: not yet), windows (-1: not yet) */
int clair_frontend_stats(clair_frontend_t *f, int64_t *stats);

#ifdef __cplusplus
}
#endif
#endif /* CLAIR_AMD_H */
