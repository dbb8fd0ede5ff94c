"""Informational: how far has each diagnostic copy under scripts/diag/ drifted from the product file it was forked from?
    python scripts/diag/check_drift.py
Prints, per copy, the product file, the similarity of the two texts (difflib ratio over non-blank lines; 1.0 = identical apart from
the diagnostic switches) and the number of product lines that no longer occur in the copy.  Nothing fails: the copies carry timing
switches / stamps on purpose, the number tells a reader whether a copy still describes the product kernel."""
import difflib, os, re

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "..", "..", "pesr_amd", "csrc")
FORKS = {"conv3x3_wgrad_wino4_diag.hip": "conv3x3_wgrad_wino4.hip", "conv3x3_wino4_diag.hip": "conv3x3_wino4.hip",
         "conv3x3_wino4_stag_diag.hip": "conv3x3_wino4.hip", "conv3x3_wino4_persist.hip": "conv3x3_wino4.hip", "conv3x3_wino4_vstore_diag.hip": "conv3x3_wino4.hip",
         "conv3x3_bf16_diag.hip": "conv3x3_bf16.hip", "conv3x3_mfma_diag.hip": "conv3x3_mfma.hip", "conv3x3_wgrad_bf16_ldsdma.hip": "conv3x3_wgrad_bf16.hip", "linear_diag.hip": "linear.hip"}


def lines(p):
    return [l.rstrip() for l in open(p).read().split("\n") if l.strip()]


for copy, prod in sorted(FORKS.items()):
    a, b = os.path.join(HERE, copy), os.path.join(CSRC, prod)
    if not (os.path.exists(a) and os.path.exists(b)):
        print(f"{copy:38s} (missing: {'copy' if not os.path.exists(a) else prod})")
        continue
    la, lb = lines(a), lines(b)
    ratio = difflib.SequenceMatcher(None, la, lb, autojunk=False).ratio()
    sa = set(la)
    gone = sum(1 for l in lb if l not in sa and not re.match(r"\s*//", l))
    print(f"{copy:38s} <- {prod:28s} similarity {ratio:.3f}   product code lines not in the copy: {gone} of {len(lb)}")
