/* clair_call.h -- one candidate's resolved variant call, as the decode stage leaves it.
 *
 * The reference turns the four softmax vectors of a candidate into a VCF row in two steps (clair/call_var.py): it forms the ten
 * outcome families (1 179 float32 products, possible_outcome_probabilites_from :589-690) and picks the best outcome by an iterative
 * arg-max with exact-equality membership tests (output_from :693-947), then derives genotype, allele frequency and quality from the
 * pick (output_with :1002-1196).  Everything in that chain that is ARITHMETIC -- the products, the arg-max, the tie rules, the
 * fall-through of unresolvable outcomes, depth and supporting reads, the probability the quality is computed from, the tensor's
 * vote on inserted bases -- ends in the 32 bytes below; what is left is text (REF / ALT strings from the reference window, the
 * logarithm of QUAL, printf).  Two producers write this record bit for bit alike:
 *   - clair_host_resolve_calls   (include/clair_host.h, clair_amd/hostsrc/host_decode.cpp): scalar C++, -ffp-contract=off;
 *   - clair_submit_ex            (include/clair_amd.h, clair_amd/csrc/decode.hip.h): one wavefront per candidate on the GPU, straight
 *     from the probabilities in HBM, so that the 360 bytes of probabilities per candidate need not cross the host link at all;
 * and one consumer turns it into the row: clair_host_format_calls.
 */
#ifndef CLAIR_CALL_H
#define CLAIR_CALL_H
#include <stdint.h>

/* outcome families in the reference's order of precedence (call_var.py:733-762 tests them in this order) and their sizes */
enum clair_family {
    CLAIR_F_REF = 0,       /* 1   reference call                                             */
    CLAIR_F_HOMO_SNP = 1,  /* 4   AA CC GG TT                                                */
    CLAIR_F_HET_SNP = 2,   /* 6   AC AG AT CG CT GT                                          */
    CLAIR_F_HOMO_INS = 3,  /* 16  length 1..16                                               */
    CLAIR_F_ACGT_INS = 4,  /* 64  (length - 1) * 4 + base                                    */
    CLAIR_F_INSINS = 5,    /* 256 (length1 - 1) * 16 + (length2 - 1)                         */
    CLAIR_F_HOMO_DEL = 6,  /* 16                                                             */
    CLAIR_F_ACGT_DEL = 7,  /* 64                                                             */
    CLAIR_F_DELDEL = 8,    /* 240 pairs (i, j), j != i, in list order                        */
    CLAIR_F_INSDEL = 9,    /* 512 ((i - 1) * 16 + (j - 1)) * 2 + {0: ins i / del j, 1: del i / ins j} */
    CLAIR_F_COUNT = 10
};

#define CLAIR_CALL_RESOLVED 1u   /* centre base in ACGTU, depth > 0: the arg-max ran and picked an outcome              */
#define CLAIR_CALL_CONSULTED 2u  /* the pick passed a point where the reference asks the BAM when it has one (an indel of 16+
                                    bases, call_var.py:498-524, 540-565; the second allele of an Ins/Ins call, :805-823)      */
#define CLAIR_CALL_MULTI 4u      /* two alternative alleles ("1/2")                                                       */
#define CLAIR_CALL_SAME 8u       /* REF == ALT for a non-reference pick (homozygous "SNP" to the reference base): no row     */

typedef struct clair_call {
    uint8_t status;      /* CLAIR_CALL_* bits                                                                               */
    uint8_t family;      /* the family that produced REF / ALT (enum clair_family)                                          */
    uint16_t index;      /* first index, in that family, of the value equal to the best (indel families)                     */
    uint16_t flags;      /* bit k: family k holds a value equal to the best in the deciding round (ties across families count:
                            genotype and supporting reads follow ALL flags, as the reference's is_* variables do)            */
    uint8_t gt;          /* 0 "0/0", 1 "1/1", 2 "0/1", 3 "1/2"                                                              */
    uint8_t gi;          /* gt21 class of the call (task/gt21.py:60-110): what QUAL's probability is read at                  */
    uint8_t alt_b0;      /* SNP families: first ALT base (0..3 = ACGT); ACGT_INS / ACGT_DEL: the outcome's base; else 255     */
    uint8_t alt_b1;      /* heterozygous SNP with two non-reference bases: the second; else 255                              */
    uint8_t ins_avail;   /* bases the tensor supports for an insertion of 16 or more (15 or 16, call_var.py:487-497)          */
    uint8_t reserved0;
    uint32_t ins_code;   /* the tensor's vote for the inserted base at window positions 17..32, two bits each (:428-447)      */
    float depth;         /* sum of the centre position's deletion + reference channels (:1022-1024)                          */
    float support;       /* reads supporting the call (:1096-1150), float32 like the NumPy scalars                            */
    float p_call;        /* gt21[gi] * genotype[zygosity of gt]: the probability quality_score_from starts from (:568-586)     */
    uint32_t rounds;     /* rounds the arg-max took (1 unless outcomes fell through)                                          */
} clair_call_t;

#endif /* CLAIR_CALL_H */
