"""`python -m clair_amd <submodule> [options]` -- the reference's `python clair.py <submodule> [options]` (clair.py:60-86) for the
submodules this build covers.  Submodule names are the reference's."""
import sys
from importlib import import_module

SUBMODULES = {
    "call_var": "clair_amd.call_var",
    "callVarBam": "clair_amd.callVarBam",
    "callVarBamParallel": "clair_amd.callVarBamParallel",
    "CreateTensor": "clair_amd.create_tensor",
    "ExtractVariantCandidates": "clair_amd.extract_variant_candidates",
}
NOT_COVERED = ("evaluate", "plot_tensor", "train", "train_clr", "GetTruth", "PairWithNonVariants", "Tensor2Bin", "CombineBins",
               "Bin2To3", "ensemble", "overlap_variant")


def main():
    if len(sys.argv) <= 1 or sys.argv[1] in ("-h", "--help"):
        print("clair_amd submodule invocator:\n    Usage: python -m clair_amd [submodule] [options of the submodule]\n\n"
              "Available submodules:\n%s" % "\n".join("      - %s" % k for k in SUBMODULES))
        sys.exit(0)
    name = sys.argv[1]
    if name in NOT_COVERED:
        sys.exit("[ERROR] Submodule %s is outside this build (variant calling only: %s)." % (name, ", ".join(SUBMODULES)))
    if name not in SUBMODULES:
        sys.exit("[ERROR] Submodule %s not found." % name)
    sys.argv = sys.argv[1:]          # the submodule parses its own options, as under the reference's dispatcher
    import_module(SUBMODULES[name]).main()


if __name__ == "__main__":
    main()
