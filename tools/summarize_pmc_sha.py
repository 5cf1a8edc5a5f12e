"""Content hash of the sources behind a measured traffic figure (shared by tools/summarize_pmc.py and bench.py)."""
import hashlib
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNEL_SOURCES = ("rslmtoasa_amd/csrc/kernels_spmm5.hpp",)   # the dominant kernel (its launch geometry in rsrec.hip is option-driven: tuned runs print no traffic)


def kernel_sha():
    """Identity of the code a traffic figure was measured on (the GPU box has no .git: a content hash stands in for the commit);
    bench.py refuses a profiles/traffic.json entry whose hash differs from the kernel source it runs."""
    h = hashlib.sha256()
    for f in KERNEL_SOURCES:
        with open(os.path.join(ROOT, f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


