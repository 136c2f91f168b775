"""Identity of the kernel sources a profile was collected on: sha256 over the files of csrc/ (names and contents, sorted), first 16 hex digits.
tools/collect_profiles_r05.sh stamps every profiles/r05_*.json with it; bench.py quotes counter-derived numbers (roofline.traffic, the
conv stack's HBM GB/s) only from files whose stamp equals the hash of the sources the running library was built from.
usage: python tools/csrc_hash.py            -> prints the hash
       python tools/csrc_hash.py FILE...    -> adds / replaces the key "csrc_sha16" in those JSON files"""
import hashlib, json, os, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def csrc_sha16(root=ROOT):
    d = os.path.join(root, "superpoint-stereo-visual-odometry_amd", "csrc")
    h = hashlib.sha256()
    for name in sorted(os.listdir(d)):
        if name.endswith((".hip", ".h")):
            h.update(name.encode())
            h.update(open(os.path.join(d, name), "rb").read())
    return h.hexdigest()[:16]


if __name__ == "__main__":
    tag = csrc_sha16()
    for path in sys.argv[1:]:
        j = json.load(open(path))
        j["csrc_sha16"] = tag
        json.dump(j, open(path, "w"), indent=1)
    print(tag)
