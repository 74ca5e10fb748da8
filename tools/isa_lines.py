"""Static per-source-line instruction counts of one kernel, from a `hipcc -gline-tables-only -save-temps` .s file.

    python tools/isa_lines.py FILE.s KERNEL_SUBSTRING [min_count]

Every instruction is attributed to the innermost `.loc` line in effect (inlined helpers count where they are
defined).  Counts are static (one per instruction in the binary, not per execution): a guide to where a kernel's
vector instructions come from, to be read next to rocprofv3's SQ_INSTS_VALU for the executed totals.
"""
import collections
import re
import sys


def main():
    path, want = sys.argv[1], sys.argv[2]
    min_count = int(sys.argv[3]) if len(sys.argv) > 3 else 8
    files = {}
    per = collections.defaultdict(lambda: collections.Counter())
    in_kernel = False
    cur = ("?", 0)
    totals = collections.Counter()
    for ln in open(path, errors="replace"):
        s = ln.strip()
        m = re.match(r'\.file\s+(\d+)\s+"([^"]*)"\s+"([^"]*)"', s)
        if m:
            files[int(m.group(1))] = m.group(3)
            continue
        if re.match(r"^_Z\w+:", ln) or re.match(r"^\w+:\s*; @", ln):
            in_kernel = want in ln
            continue
        if not in_kernel:
            continue
        if s.startswith(".loc"):
            p = s.split()
            cur = (files.get(int(p[1]), p[1]), int(p[2]))
            continue
        if s.startswith("s_endpgm"):
            in_kernel = False
            continue
        if not s or s[0] in ".;" or s.endswith(":"):
            continue
        op = s.split()[0]
        if op.startswith("v_"):
            kind = "valu"
        elif op.startswith("s_"):
            kind = "salu"
        elif op.startswith("ds_"):
            kind = "lds"
        elif op.startswith(("global_", "buffer_", "flat_", "scratch_")):
            kind = "vmem"
        else:
            kind = "other"
        per[cur][kind] += 1
        totals[kind] += 1
        if op in ("v_readlane_b32", "v_writelane_b32"):
            per[cur]["lane"] += 1
            totals["lane"] += 1
    print("totals:", dict(totals))
    rows = sorted(per.items(), key=lambda kv: -kv[1]["valu"])
    for (f, line), c in rows:
        if c["valu"] + c["salu"] < min_count:
            continue
        print(f"{f}:{line:<5d} valu={c['valu']:<5d} salu={c['salu']:<5d} lds={c['lds']:<4d} vmem={c['vmem']:<4d} lane={c['lane']}")


if __name__ == "__main__":
    main()
