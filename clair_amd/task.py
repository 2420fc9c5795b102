"""Output-task layout of the four heads and the label algebra the VCF decode needs.

Counterpart of /root/reference/clair/task/{main,gt21,genotype,variant_length}.py, inference
side only (label encoders for training are out of scope):

  head sizes 21 / 3 / 33 / 33 and their slices in a packed 90-vector   task/main.py:10-29
  the 21 genotype labels and their indices                              task/gt21.py:3-50
  genotype classes 0/0, 1/1, 0/1, 1/2                                   task/genotype.py:3-10
  indel length range -16..+16, index offset 16                          task/variant_length.py:6-12
  gt21 label of a (REF, ALT, genotype) triple                           task/gt21.py:60-110
"""

GT21_LABELS = ("AA", "AC", "AG", "AT", "CC", "CG", "CT", "GG", "GT", "TT",
               "DelDel", "ADel", "CDel", "GDel", "TDel",
               "InsIns", "AIns", "CIns", "GIns", "TIns", "InsDel")
GT21_INDEX = {label: i for i, label in enumerate(GT21_LABELS)}

HOMO_SNP = ("AA", "CC", "GG", "TT")                      # task/gt21.py:112-113
HETERO_SNP = ("AC", "AG", "AT", "CG", "CT", "GT")        # task/gt21.py:115-116
HOMO_SNP_IDX = tuple(GT21_INDEX[s] for s in HOMO_SNP)
HETERO_SNP_IDX = tuple(GT21_INDEX[s] for s in HETERO_SNP)
INS_BASE_IDX = tuple(GT21_INDEX[b + "Ins"] for b in "ACGT")
DEL_BASE_IDX = tuple(GT21_INDEX[b + "Del"] for b in "ACGT")
IDX_INSINS, IDX_DELDEL, IDX_INSDEL = GT21_INDEX["InsIns"], GT21_INDEX["DelDel"], GT21_INDEX["InsDel"]

GENOTYPE_STRINGS = ("0/0", "1/1", "0/1", "1/2")          # task/genotype.py:3
HOMO_REFERENCE, HOMO_VARIANT, HETERO_VARIANT, HETERO_VARIANT_MULTI = 0, 1, 2, 3

LENGTH_OFFSET = 16                                        # task/variant_length.py:6
LENGTH_MAX = 16
N_LENGTH = 2 * LENGTH_OFFSET + 1

SLICES = {"gt21": (0, 21), "genotype": (21, 24), "len1": (24, 57), "len2": (57, 90)}

# shared/utils.py:19-29
IUPAC_TO_NUM = dict(zip("ACGTURYSWKMBDHVN", (0, 1, 2, 3, 3, 0, 1, 1, 0, 2, 0, 1, 0, 0, 0, 0)))
IUPAC_TO_ACGT = dict(zip("ACGTURYSWKMBDHVN", "ACGTTACCAGACAAAA"))
BASIC_BASES = frozenset("ACGTU")


def _allele_kind(ref, alt):
    """task/gt21.py:60-65 -- how one allele differs from REF: 'Del', 'Ins' or its first base."""
    if len(ref) > len(alt):
        return "Del"
    if len(ref) < len(alt):
        return "Ins"
    return alt[0]


def gt21_index_of_call(ref, alt, g1, g2):
    """Index of the gt21 class a VCF call (REF, ALT string, genotype digits) belongs to.

    Follows task/gt21.py:68-110: a single ALT is paired with REF when either genotype digit
    is 0, else with itself; two base alleles sort alphabetically; base+indel gives e.g. 'AIns';
    equal indel kinds give 'InsIns'/'DelDel'; mixed gives 'InsDel'.
    """
    alts = alt.split(",")
    if len(alts) == 1:
        alts = [ref if (g1 == 0 or g2 == 0) else alts[0]] + alts
    a, b = _allele_kind(ref, alts[0]), _allele_kind(ref, alts[1])
    if len(a) == 1 and len(b) == 1:
        label = a + b if a <= b else b + a
    elif len(a) == 1 or len(b) == 1:
        label = (a + b) if len(a) == 1 else (b + a)
    elif a == b:
        label = a + b
    else:
        label = "InsDel"
    return GT21_INDEX[label]


def genotype_class_of(g1, g2):
    """task/genotype.py:19-33: class index used by the genotype head (1/2 folds into 0/1)."""
    if g1 == 0 and g2 == 0:
        return HOMO_REFERENCE
    if g1 == g2:
        return HOMO_VARIANT
    return HETERO_VARIANT
