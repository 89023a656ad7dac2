"""Digest of the sources whose change makes hardware counters stale: the HIP kernels and the host schedule (not docs, tests or tools).
Recorded in profiles/pmc*.json at profiling time (`_meta.sources_sha`), recomputed by bench.py when it quotes those counters."""
import hashlib
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def sources_sha():
    h = hashlib.sha1()
    pkg = os.path.join(ROOT, "rnd_semantic_segmentation_amd")
    files = []
    for sub, exts in (("csrc", (".hip", ".h")), ("host", (".py",)), ("", (".py",))):
        d = os.path.join(pkg, sub)
        files += [os.path.join(d, f) for f in sorted(os.listdir(d)) if f.endswith(exts) and os.path.isfile(os.path.join(d, f))]
    for f in files:
        h.update(os.path.relpath(f, ROOT).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]
