"""Identity of the kernel sources a measurement was taken on.
    python tools/source_stamp.py            -> one JSON object on stdout
    python tools/source_stamp.py --write    -> also stored in tools/bin/stamp.json (before a run on the GPU box,
                                               which receives the tree without its git history)
csrc_sha256: digest over the library's sources (csrc/*.h, *.inl, *.hip, the Makefile, the public header) -- what
bench.py compares with the running tree to tell whether a committed profile is stale.  git_head / dirty: the commit
the tree was at, from git where there is one, else from tools/bin/stamp.json if its digest matches the tree."""
import glob
import hashlib
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "image-compression_amd", "csrc")
STORE = os.path.join(ROOT, "tools", "bin", "stamp.json")


def csrc_digest():
    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(CSRC, "*.inl")) +
                   glob.glob(os.path.join(CSRC, "*.hip")) + [os.path.join(CSRC, "Makefile"),
                                                            os.path.join(ROOT, "include", "khugle_hip.h")])
    for f in files:
        h.update(os.path.basename(f).encode() + b"\0")
        h.update(open(f, "rb").read())
    return h.hexdigest()


def stamp():
    s = {"csrc_sha256": csrc_digest(), "git_head": None, "dirty": None}
    try:
        head = subprocess.run(["git", "-C", ROOT, "rev-parse", "HEAD"], capture_output=True, text=True, timeout=10)
        if head.returncode == 0:
            s["git_head"] = head.stdout.strip()
            st = subprocess.run(["git", "-C", ROOT, "status", "--porcelain", "--", "image-compression_amd/csrc", "include"],
                                capture_output=True, text=True, timeout=10)
            s["dirty"] = bool(st.stdout.strip())
            return s
    except (OSError, subprocess.SubprocessError):
        pass
    try:
        kept = json.load(open(STORE))
        if kept.get("csrc_sha256") == s["csrc_sha256"]:
            s["git_head"], s["dirty"] = kept.get("git_head"), kept.get("dirty")
    except (OSError, ValueError):
        pass
    return s


if __name__ == "__main__":
    s = stamp()
    if "--write" in sys.argv:
        os.makedirs(os.path.dirname(STORE), exist_ok=True)
        json.dump(s, open(STORE, "w"))
    print(json.dumps(s))
