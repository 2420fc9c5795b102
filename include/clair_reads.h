/* clair_reads.h -- packed alignments: what the host packer (include/clair_host.h: clair_host_sampack_*) hands the device front
 * end (include/clair_amd.h: clair_frontend_*).
 *
 * The reference's two pileup stages (dataPrepScripts/ExtractVariantCandidates.py:259-345, CreateTensor.py:251-373) each read
 * `samtools view` text and walk CIGAR and SEQ character by character.  Here the text is read ONCE; a slab holds, for the alignments
 * either stage would use, the columns both walk (POS, the strand bit of FLAG, SEQ upper-cased) and the CIGAR as a list of the
 * operations that touch a reference position or a read base in those loops: M/=/X, I, D.  S is folded into the read offset of the
 * next operation; H, N, P and anything else move neither cursor there (:296-316, :289-365) and are dropped.
 *
 * One "element" is one iteration of the innermost loops: a matched, inserted or deleted base.  Element e of a slab belongs to the
 * operation j with op_elem[j] <= e < op_elem[j+1] (exclusive prefix sum of the operation lengths, n_ops + 1 entries). */
#ifndef CLAIR_READS_H
#define CLAIR_READS_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct clair_read {
    int64_t pos0;        /* POS - 1 (CreateTensor.py:258) */
    uint32_t seq0;       /* first byte of SEQ in the slab's base array */
    uint32_t seq_len;
    uint32_t op0;        /* first operation in the slab's operation array */
    uint32_t n_ops;
    uint32_t flags;      /* CLAIR_READ_* */
    uint32_t reserved;
} clair_read_t;          /* 32 bytes */

enum {
    CLAIR_READ_REVERSE = 1,   /* FLAG & 16 (CreateTensor.py:264) */
    CLAIR_READ_EVC = 2,       /* passes the candidate search's filters: RNAME, MQ, CIGAR != "*", >= 55 % aligned (EVC :279-293) */
    CLAIR_READ_PILE = 4,      /* walked by the pileup: in the pileup's region, MQ, not beyond --dcov at its start (CT :266-287) */
    CLAIR_READ_FLUSH = 8      /* first read the pileup walks at a new start position: windows left of it are complete (CT :369) */
};

typedef struct clair_op {
    uint32_t read;       /* index of the alignment in the slab */
    uint32_t code_len;   /* length << 2 | code; lengths are > 0 */
    int32_t ref_off;     /* reference offset of the operation from pos0 */
    uint32_t q_off;      /* offset of its first base in the alignment's SEQ */
} clair_op_t;            /* 16 bytes */

enum { CLAIR_OP_M = 0, CLAIR_OP_I = 1, CLAIR_OP_D = 2 };

/* What takes a run out of the regime the device front end reproduces exactly (LABNOTES.md part B 6b); the caller then runs the sequential
 * host code (clair_host_evc_*, clair_host_pileup_*), which reproduces the reference there too, errors included. */
enum {
    CLAIR_FE_UNSORTED = 1,        /* start positions decrease */
    CLAIR_FE_ZERO_INDEL = 2,      /* "0I" / "0D": the candidate search counts the operation, there is no base to hang it on */
    CLAIR_FE_LONG_SPAN = 4,       /* reference span beyond len(SEQ) + 100 000: the reference has not loaded those candidates yet (CT :274) */
    CLAIR_FE_SEQ_OVERRUN = 8,     /* CIGAR walks past the end of SEQ: an IndexError in the reference */
    CLAIR_FE_BAD_BASE = 16,       /* a read base that is not an IUPAC code: KeyError in the candidate search, a skipped tuple in the pileup */
    CLAIR_FE_BAD_REF = 32,        /* reference base missing or not an IUPAC code under a walked position */
    CLAIR_FE_OVERFLOW = 64,       /* a count beyond int16 */
    CLAIR_FE_BUDGET = 128,        /* the budget of 5 M outstanding tuples would have run out (CT :181, 289): results depend on offer order */
    CLAIR_FE_CANDIDATES = 256,    /* a given candidate list is not strictly ascending */
    CLAIR_FE_LEAD_INDEL = 512     /* an alignment the candidate search accepts begins with an I or D (tallied at POS - 1) and an earlier accepted alignment has
                                   * the same POS: the reference flushed POS - 1 after that one (EVC :316-345: positions < POS after EVERY alignment) and
                                   * evaluates the late tally on its own, which one sum per position cannot reproduce */
};

#ifdef __cplusplus
}
#endif
#endif
