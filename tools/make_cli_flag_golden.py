"""Mint tests/golden/cli_flags.json: option strings, defaults, types of the reference command lines (build container only)."""
import argparse, json, sys, os, importlib, tempfile
sys.path.insert(0, "/root/reference")
stub = tempfile.mkdtemp(); os.makedirs(os.path.join(stub, "intervaltree"))
sys.path.insert(0, "/root/repo/tools"); import make_pileup_goldens as g
open(os.path.join(stub, "intervaltree", "__init__.py"), "w").write(g.INTERVALTREE_STUB)
sys.path.insert(0, stub)
class Captured(Exception): pass
out = {}
orig = argparse.ArgumentParser.parse_args
def capture(self, *a, **k):
    acts = []
    for act in self._actions:
        if not act.option_strings or "-h" in act.option_strings: continue
        acts.append({"flags": act.option_strings, "default": act.default, "type": getattr(act.type, "__name__", None),
                     "nargs": act.nargs, "action": type(act).__name__})
    raise Captured(acts)
argparse.ArgumentParser.parse_args = capture
for name, mod in (("CreateTensor", "dataPrepScripts.CreateTensor"), ("ExtractVariantCandidates", "dataPrepScripts.ExtractVariantCandidates"),
                  ("callVarBam", "clair.callVarBam"), ("callVarBamParallel", "clair.callVarBamParallel")):
    m = importlib.import_module(mod)
    try:
        m.main()
    except Captured as c:
        out[name] = c.args[0]
json.dump(out, open("/root/repo/tests/golden/cli_flags.json", "w"), indent=1)
print({k: len(v) for k, v in out.items()})
