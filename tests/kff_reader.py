"""In-tree reader of the KFF 1.0 subset that brisk_amd/include/brisk_kff.hpp writes (header, 'v', 'r' and 'm' sections,
footer): test infrastructure for the KFF writer.  Written from the published format description, independently of the
emitter's code paths where it matters (bit unpacking by string slicing, not by shifts).  Parity with the reference's
kff-cpp-api output is unpinned (the library is an empty submodule in the reference tree)."""
import math
import struct

NT = "ACTG"  # the index's 2-bit codes: A0 C1 T2 G3 (reference encoding 0,1,3,2 for A,C,G,T, brisk/writer.hpp:26)


def _field_bytes(max_value: int) -> int:
    bits = math.ceil(math.log2(max_value)) if max_value > 1 else 0
    return (bits + 7) // 8


def _nts(raw: bytes, n: int, code2nt: str) -> str:
    bits = "".join(f"{b:08b}" for b in raw)
    bits = bits[len(bits) - 2 * n:]
    return "".join(code2nt[int(bits[2 * i:2 * i + 2], 2)] for i in range(n))


def read_kff(path):
    """-> (header dict, list of (k-mer string, minimizer_idx or None, data bytes))"""
    buf = open(path, "rb").read()
    assert buf[:3] == b"KFF" and buf[-3:] == b"KFF", "signature"
    major, minor, enc, uniq, canon = buf[3], buf[4], buf[5], buf[6], buf[7]
    assert (major, minor) == (1, 0)
    code2nt = [None] * 4
    for nt, shift in zip("ACGT", (6, 4, 2, 0)):
        code2nt[(enc >> shift) & 3] = nt
    assert None not in code2nt, "encoding is not a permutation"
    (meta_len,) = struct.unpack(">I", buf[8:12])
    pos = 12 + meta_len
    hdr = {"version": (major, minor), "encoding": "".join(code2nt), "metadata": buf[12:pos].decode(), "uniq": uniq, "canon": canon, "sections": []}
    var = {}
    out = []
    end = len(buf) - 3
    while pos < end:
        kind = chr(buf[pos])
        pos += 1
        hdr["sections"].append(kind)
        if kind == "v":
            (n,) = struct.unpack(">Q", buf[pos:pos + 8])
            pos += 8
            for _ in range(n):
                z = buf.index(0, pos)
                name = buf[pos:z].decode()
                (val,) = struct.unpack(">Q", buf[z + 1:z + 9])
                pos = z + 9
                var[name] = val
            hdr.setdefault("vars", []).append(dict(var))
        elif kind in "rm":
            k, ds, mx = var["k"], var["data_size"], var["max"]
            mini = None
            if kind == "m":
                m = var["m"]
                nb = (m + 3) // 4
                mini = _nts(buf[pos:pos + nb], m, code2nt)
                pos += nb
            (n_blocks,) = struct.unpack(">Q", buf[pos:pos + 8])
            pos += 8
            for _ in range(n_blocks):
                nk = 1
                fb = _field_bytes(mx)
                if mx > 1:
                    nk = int.from_bytes(buf[pos:pos + fb], "big")
                    pos += fb
                assert 1 <= nk <= mx
                mp = None
                if kind == "m":
                    pb = _field_bytes(k + mx - 1)
                    mp = int.from_bytes(buf[pos:pos + pb], "big")
                    pos += pb
                nts = k + nk - 1 - (var["m"] if kind == "m" else 0)
                sb = (nts + 3) // 4
                raw = buf[pos:pos + sb]
                pad = 8 * sb - 2 * nts
                assert pad == 0 or raw[0] >> (8 - pad) == 0, "unused high bits of the first byte must be zero"
                seq = _nts(raw, nts, code2nt)
                pos += sb
                data = buf[pos:pos + nk * ds]
                pos += nk * ds
                if kind == "m":
                    assert 0 <= mp <= nts
                    seq = seq[:mp] + mini + seq[mp:]
                for j in range(nk):
                    km = seq[j:j + k]
                    idx = None
                    if kind == "m":  # nts of this k-mer behind the minimizer
                        idx = k - var["m"] - (mp - j)
                        assert 0 <= idx <= k - var["m"], "minimizer outside the k-mer"
                    out.append((km, idx, bytes(data[j * ds:(j + 1) * ds])))
        else:
            raise AssertionError(f"unknown section {kind!r} at {pos - 1}")
    assert pos == end
    assert hdr["sections"][-1] == "v" and var.get("footer_size") is not None, "footer"
    return hdr, out
