"""Content hash of the kernel sources (brisk_amd/csrc/* and include/brisk_hip.h): what a committed PMC profile is tied to.
    python tools/src_hash.py   -> prints the sha256"""
import hashlib
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def kernel_source_hash(root: str = ROOT) -> str:
    h = hashlib.sha256()
    files = [os.path.join("brisk_amd", "csrc", f) for f in sorted(os.listdir(os.path.join(root, "brisk_amd", "csrc"))) if f.endswith((".hip", ".h"))]
    files.append(os.path.join("include", "brisk_hip.h"))
    for rel in files:
        h.update(rel.encode())
        h.update(open(os.path.join(root, rel), "rb").read())
    return h.hexdigest()


if __name__ == "__main__":
    print(kernel_source_hash())
