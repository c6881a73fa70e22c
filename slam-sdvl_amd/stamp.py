"""Which kernel sources a committed profile belongs to.  tools/summarize_profiles.py writes kernel_source_stamp() of the tree it
profiled into profiles/rNN/source_stamp.json; bench.py compares it with the tree it runs from and says `traffic_stale` when the
counters it quotes (roofline.traffic, roofline.valu) were taken on other kernels.  The stamp is git's own blob hash
(`git hash-object`) of every file under csrc/, computed here without git: the GPU boxes have no .git."""
import glob
import hashlib
import os

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")


def blob_hash(path):
    data = open(path, "rb").read()
    return hashlib.sha1(b"blob %d\0" % len(data) + data).hexdigest()


def kernel_source_files():
    files = []
    for pat in ("*.hip", "*.h", "*.inc"):
        files += glob.glob(os.path.join(CSRC, pat))
    return sorted(files)


def kernel_source_stamp():
    """{"files": {name: git blob hash}, "combined": sha1 over the sorted "name hash" lines}"""
    files = {os.path.basename(f): blob_hash(f) for f in kernel_source_files()}
    combined = hashlib.sha1("".join("%s %s\n" % kv for kv in sorted(files.items())).encode()).hexdigest()
    return {"files": files, "combined": combined}
