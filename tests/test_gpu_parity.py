"""GPU parity: the HIP path (through the C-ABI) against the oracle and the golden
fixtures.  Everything here is integer/byte work: the bar is bit-exact."""
import hashlib
import random

import numpy as np
import pytest

import oracle
from conftest import load_golden

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def B():
    import brisk_amd
    import torch
    assert torch.cuda.is_available(), "gpu tests need a device"
    assert brisk_amd.library_path()
    return brisk_amd


def md5_lines(lines):
    return hashlib.md5("".join("{} idx={} {}\n".format(*l.split()) for l in lines).encode()).hexdigest()


def gpu_count(B, seqs, k, m, b, batches=1, **kw):
    with B.BriskHip(k, m, b, **kw) as ix:
        step = max(1, (len(seqs) + batches - 1) // batches)
        for i in range(0, len(seqs), step):
            ix.insert_reads(seqs[i:i + step])
        st = ix.stats()
        lines = oracle.multiset_lines(*ix.enumerate(), k)
    return lines, st["nb_kmers"], st["nb_buckets"]


def _seqs(name):
    return oracle.fasta_sequences(load_golden(name))


CONFIGS = [(31, 11, 4), (63, 21, 14), (31, 13, 12), (31, 15, 14), (63, 21, 9), (31, 11, 11), (31, 11, 10)]


def test_reference_fixtures_against_golden(B):
    """config #1 and friends: the reference's own data files, golden md5s from the reference."""
    for e in load_golden("multisets.json"):
        if e["input"] not in ("test.fa", "debug_test.fa"):
            continue
        lines, nk, nb = gpu_count(B, _seqs(e["input"]), e["k"], e["m"], e["b"])
        assert (nk, nb) == (e["nb_kmers"], e["nb_buckets"]), e
        assert sum(int(l.split()[2]) for l in lines) == e["sum_counts"]
        assert md5_lines(lines) == e["md5"], (e["input"], e["k"], e["m"], e["b"])


def test_full_multiset_fixture(B):
    for k, m, b in ((31, 11, 4), (63, 21, 14)):
        want = load_golden(f"multiset_test_k{k}m{m}b{b}.txt.gz").split("\n")[:-1]
        assert gpu_count(B, _seqs("test.fa"), k, m, b)[0] == want


def test_synthetic_goldens(B, O):
    for e in load_golden("multisets.json"):
        if not e["input"].startswith("synth:"):
            continue
        kv = dict(p.split("=") for p in e["input"][6:].split(","))
        seqs = [bytes(r) for r in O.synth_reads(int(kv["G"]), 0, int(kv["n"]))]
        lines, nk, nb = gpu_count(B, seqs, e["k"], e["m"], e["b"])
        assert (nk, nb) == (e["nb_kmers"], e["nb_buckets"])
        assert md5_lines(lines) == e["md5"]
        with B.BriskHip(e["k"], e["m"], e["b"]) as ix:
            ix.insert_reads(seqs)
            assert [int(v) for v in ix.get_reads(seqs[:50])] == e["query_sums_first50"]


def _random_reads(rng, n, glen, L=150):
    genome = "".join(rng.choice("ACGT") for _ in range(glen))
    out = []
    for _ in range(n):
        p = rng.randrange(0, glen - L)
        s = genome[p:p + L]
        if rng.random() < 0.5:
            s = s[::-1].translate(str.maketrans("ACGT", "TGCA"))
        out.append(s)
    return out


SPECIAL = ["A" * 150, "C" * 150, "G" * 150, "T" * 150, "AC" * 75, "ACG" * 50, "ACGT" * 40, "T" * 149 + "A",
           "A" * 70 + "ACGTTGCA" * 10, "acgt" * 40, "ACGTTGCATGCA" * 13]


@pytest.mark.parametrize("k,m,b", CONFIGS + [(33, 11, 7), (47, 15, 10), (21, 7, 3), (63, 31, 12), (41, 21, 5), (63, 21, 4), (32, 13, 6), (34, 21, 9), (31, 11, 11),
                                            # short minimizers: a minimizer_idx class bit in the routing id (k - m + 1 >= 8), or not (15, 11, 5)
                                            (63, 11, 4), (63, 11, 11), (20, 9, 3), (18, 11, 9), (15, 11, 5), (40, 7, 7), (33, 9, 1)])
def test_random_and_degenerate_reads_vs_oracle(B, O, k, m, b):
    rng = random.Random(k * 1000 + m * 10 + b)
    reads = _random_reads(rng, 500, 5000) + SPECIAL
    reads += ["".join(rng.choice("ACGT") for _ in range(rng.randint(1, 400))) for _ in range(40)]  # ragged, some < k
    reads += ["", "A", "ACGT" * 5]
    want = O.count(reads, k, m, b)
    got = gpu_count(B, reads, k, m, b)
    assert got[1:] == want[1:]
    assert got[0] == want[0]
    if m <= 11:  # and the get, whose records the scan cuts the same way
        q = reads[:120] + SPECIAL
        qf, qo = oracle.pack_reads(q)
        flat, offs = oracle.pack_reads(reads)
        h = O.index_new(k, m, b)
        O.index_insert_reads(h, flat, offs)
        with B.BriskHip(k, m, b) as ix:
            ix.insert_reads(reads)
            assert np.array_equal(ix.get_reads(q), O.index_query_reads(h, qf, qo))
        O.index_free(h)


def test_insert_is_incremental_and_order_independent(B, O):
    rng = random.Random(5)
    reads = _random_reads(rng, 1500, 8000) + SPECIAL
    for k, m, b in ((63, 21, 14), (31, 11, 11)):
        want = O.count(reads, k, m, b)
        assert gpu_count(B, reads, k, m, b, batches=7) == want
        shuffled = reads[:]
        rng.shuffle(shuffled)
        assert gpu_count(B, shuffled, k, m, b, batches=3) == want
        assert gpu_count(B, reads, k, m, b, max_batch_reads=100) == want


def test_counts_wrap_mod_256(B, O):
    s = "ACGTTGCATGCCGATAGCTAGCTAGGATCGATCGGCTAGCTAGCTAGGCTAGCCATAGACCAGATTTACAGGATACCCAGGGTAAACCA"
    for k, m, b in ((31, 11, 4), (63, 21, 14)):
        assert gpu_count(B, [s] * 700, k, m, b, batches=3) == O.count([s] * 700, k, m, b)


def test_hot_partition_multi_chunk(B, O):
    """One partition receiving far more instances than one LDS chunk holds."""
    rng = random.Random(11)
    base = _random_reads(rng, 30, 400)
    reads = base * 200 + ["A" * 150] * 300
    for k, m, b, pb in ((31, 11, 4, 0), (63, 21, 9, 2), (31, 11, 11, 1)):
        assert gpu_count(B, reads, k, m, b, part_bits=pb) == O.count(reads, k, m, b)


def test_fewer_partitions_fitted_to_the_batch(B, O):
    """k31 m15 b14 with 2^22 / 2^23 partitions (brisk_amd.exchange.suggest_part_bits with min_bits: what bench.py uses for
    one batch of ~20 M reads): the specialised insert / get kernels of these geometries against the oracle on a small
    read set (records through the staging path), and the bins of up to 256 records on a batch dense enough for the
    binned layout (700 k reads: ~9 M records, more than two per partition) against the default 2^24 layout (bench.py's
    own check covers the full size: every k-mer counted once, the digest of profiles/r03_part_bits_k31.txt)."""
    import torch
    rng = random.Random(2231)
    k, m, b = 31, 15, 14
    reads = _random_reads(rng, 1500, 9000) + SPECIAL
    want = O.count(reads, k, m, b)
    flat, offs = oracle.pack_reads(reads)
    h = O.index_new(k, m, b)
    O.index_insert_reads(h, flat, offs)
    sums = O.index_query_reads(h, flat, offs)
    O.index_free(h)
    for pb in (22, 23):
        assert gpu_count(B, reads, k, m, b, part_bits=pb) == want, pb
        with B.BriskHip(k, m, b, part_bits=pb) as ix:
            assert ix.layout["part_bits"] == pb
            ix.insert_reads(reads)
            assert np.array_equal(ix.get_reads(reads), sums), pb
    n, L = 700_000, 150
    d_packed = torch.zeros((n * L + 15) // 16 + 4, dtype=torch.int32, device="cuda")
    d_starts = torch.zeros(n + 1, dtype=torch.int64, device="cuda")
    d_sums = {pb: torch.zeros(n, dtype=torch.int64, device="cuda") for pb in (0, 22, 23)}
    torch.cuda.synchronize()
    digest = {}
    for pb in (0, 22, 23):
        with B.BriskHip(k, m, b, part_bits=pb) as ix:
            ix.synth_reads(n * L // 12, 0, n, L, d_packed.data_ptr(), d_starts.data_ptr())
            ix.insert_packed(d_packed.data_ptr(), d_starts.data_ptr(), n)
            ix.get_packed(d_packed.data_ptr(), d_starts.data_ptr(), n, d_sums[pb].data_ptr())
            ix.sync()
            st = ix.stats()
            digest[pb] = (ix.checksum(), st["nb_kmers"], st["nb_buckets"])
    assert digest[22] == digest[0] and digest[23] == digest[0], digest
    assert digest[0][0][1] == n * (L - k + 1)
    assert torch.equal(d_sums[22], d_sums[0]) and torch.equal(d_sums[23], d_sums[0])


def test_lookup_and_get(B, O):
    rng = random.Random(21)
    reads = _random_reads(rng, 600, 6000) + SPECIAL
    for k, m, b in ((31, 11, 4), (63, 21, 14)):
        flat, offs = oracle.pack_reads(reads)
        h = O.index_new(k, m, b)
        O.index_insert_reads(h, flat, offs)
        lo, hi, idx, cnt = O.index_dump(h)
        with B.BriskHip(k, m, b) as ix:
            ix.insert_reads(reads)
            data, found = ix.lookup(lo, hi, idx)
            assert found.all() and np.array_equal(data, cnt)
            # absent: flip a nucleotide far from the minimizer / wrong idx
            lo2 = lo ^ np.uint64(1)
            data2, found2 = ix.lookup(lo2, hi, idx)
            want2 = np.array([O.index_get(h, int(a), int(c), int(i)) for a, c, i in zip(lo2[:500], hi[:500], idx[:500])])
            assert np.array_equal(found2[:500].astype(bool), want2 >= 0)
            assert np.array_equal(data2[:500][want2 >= 0], want2[want2 >= 0].astype(np.uint8))
            # bulk per-read query incl. the minimizer==0 break (counter.cpp:304-306)
            q = reads[:200] + SPECIAL + _random_reads(rng, 100, 6000)
            qf, qo = oracle.pack_reads(q)
            assert np.array_equal(ix.get_reads(q), O.index_query_reads(h, qf, qo))
        O.index_free(h)


def test_kernel_variants_match_the_oracle(B):
    """The layouts and kernel bodies small inputs never (or hardly ever) reach by themselves: records binned by the scan (bins of 2: nearly
    everything overflows into the classic scatter; bins of 64: nothing does), for the insert and for the get, the classic
    layout, and the run-time-geometry insert / query bodies.  One process per environment (tests/env_variant_worker.py)."""
    import os
    import subprocess
    import sys
    worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), "env_variant_worker.py")
    for extra in ({"BRISK_BINS": "2", "BRISK_QUERY_ENT": "256"}, {"BRISK_BINS": "64", "BRISK_QUERY_ENT": "128"}, {"BRISK_BINS": "0", "BRISK_DEFER": "0"},
                  {"BRISK_INSERT_GENERIC": "1", "BRISK_QUERY_GENERIC": "1", "BRISK_BINS": "0"},
                  # the workgroup-per-partition insert / query for every partition of more than 24 instances / 8 (0) entries, classic and binned records
                  {"BRISK_HUGE_AT": "24", "BRISK_BINS": "0", "BRISK_HUGE_QUERY_AT": "8"}, {"BRISK_HUGE_AT": "24", "BRISK_BINS": "2", "BRISK_HUGE_QUERY_AT": "0"}):
        env = dict(os.environ, **extra)
        p = subprocess.run([sys.executable, worker], env=env, capture_output=True, text=True, timeout=600)
        assert p.returncode == 0 and p.stdout.strip().endswith("ok 12"), (extra, p.stdout[-2000:], p.stderr[-4000:])


def test_deferred_inserts_are_invisible(B, O):
    """Small insert batches are scanned at once and inserted later (brisk_hip_options.immediate_inserts): the same index as with
    immediate inserts whatever is called in between; with few partitions the pending records reach the flush threshold by
    themselves (12 records per partition), with the default 2^24 they wait for the first call that needs the index."""
    import torch
    rng = random.Random(71)
    reads = _random_reads(rng, 4000, 20000) + SPECIAL
    flat, offs = oracle.pack_reads(reads)
    d_bases = torch.from_numpy(flat).cuda()
    d_packed = torch.zeros((len(flat) + 15) // 16 + 4, dtype=torch.int32, device="cuda")
    d_starts = torch.from_numpy(offs.astype(np.int64)).cuda()
    for (k, m, b), pb in (((63, 21, 14), 10), ((63, 21, 14), 0), ((31, 11, 11), 0)):
        want = O.count(reads, k, m, b)
        q = reads[:50] + SPECIAL
        qf, qo = oracle.pack_reads(q)
        h = O.index_new(k, m, b)
        got = {}
        for immediate in (True, False):
            with B.BriskHip(k, m, b, part_bits=pb, immediate_inserts=immediate) as ix:
                torch.cuda.synchronize()
                ix.pack_ascii(d_bases.data_ptr(), len(flat), d_packed.data_ptr())
                ix.sync()
                seen = []
                step = 150
                for i, lo in enumerate(range(0, len(reads), step)):
                    n = min(step, len(reads) - lo)
                    ix.insert_packed(d_packed.data_ptr(), d_starts[lo:lo + n + 1].contiguous().data_ptr(), n)
                    if i % 7 == 3:   # a reader in between sees everything inserted so far
                        seen.append((i, ix.stats()["nb_skmers"], ix.get_reads(q).tolist()))
                st = ix.stats()
                got[immediate] = (sorted(oracle.multiset_lines(*ix.enumerate(), k)), st["nb_kmers"], st["nb_buckets"], seen)
        assert got[True][:3] == want and got[False][:3] == want, (k, m, b, pb)
        assert got[True][3] == got[False][3]
        # and what the readers saw is what the oracle has after the same batches
        for i, _, sums in got[False][3]:
            hh = O.index_new(k, m, b)
            upto = min((i + 1) * 150, len(reads))
            f2, o2 = oracle.pack_reads(reads[:upto])
            O.index_insert_reads(hh, f2, o2)
            assert sums == O.index_query_reads(hh, qf, qo).tolist()
            O.index_free(hh)
        O.index_free(h)


def test_randomised_parity_soak(B):
    """25 s of tests/fuzz_parity.py: random (k, m, b), partition counts, read sets with repeats and homopolymers, ragged
    lengths, random batch splits with readers in between, deferred and immediate inserts -- index and get against the oracle."""
    import os
    import subprocess
    import sys
    worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), "fuzz_parity.py")
    p = subprocess.run([sys.executable, worker, "25", "7"], capture_output=True, text=True, timeout=600)
    last = p.stdout.strip().splitlines()[-1] if p.stdout.strip() else ""
    assert p.returncode == 0 and last.startswith("done") and int(last.split()[1]) >= 20, (p.stdout[-2000:], p.stderr[-3000:])


def test_scan_records_match_oracle_records(B, O):
    """The scan kernel's output, record by record (the super-k-mer boundary of the path)."""
    import torch
    rng = random.Random(31)
    reads = _random_reads(rng, 300, 4000) + SPECIAL
    for k, m, b in ((31, 11, 4), (31, 11, 11), (63, 21, 14), (47, 15, 10)):
        h = O.index_new(k, m, b)
        want = []
        for s in reads:
            c, bucket, n, idx0 = O.records(h, s, k, m, b)
            for i in range(len(n)):
                want.append(tuple(int(x) for x in c[i]) + (int(bucket[i]), int(n[i]), int(idx0[i])))
        O.index_free(h)
        flat, offs = oracle.pack_reads(reads)
        with B.BriskHip(k, m, b) as ix:
            d_bases = torch.from_numpy(flat).cuda()
            d_packed = torch.zeros((len(flat) + 15) // 16 + 4, dtype=torch.int32, device="cuda")
            d_starts = torch.from_numpy(offs.astype(np.int64)).cuda()
            torch.cuda.synchronize()
            ix.pack_ascii(d_bases.data_ptr(), len(flat), d_packed.data_ptr())
            bound = ix.scan_bound(d_starts.data_ptr(), len(reads))
            assert bound == sum(max(0, len(s) - k + 1) for s in reads)
            W = ix.record_words
            d_rec = torch.zeros(bound * W, dtype=torch.int64, device="cuda")
            n_rec = ix.scan_packed(d_packed.data_ptr(), d_starts.data_ptr(), len(reads), d_rec.data_ptr(), bound)
            ix.sync()
            rec = d_rec.cpu().numpy().view(np.uint64)[: n_rec * W].reshape(n_rec, W)
            lay = ix.layout
        ext = lay["ext_bits"]  # header bits 0..31: bucket id << ext | extra routing bits of the same minimizer hash | idx class
        got = [tuple(int(x) for x in r[: W - 1]) + ((int(r[W - 1]) & 0xffffffff) >> ext, (int(r[W - 1]) >> 32) & 0xff, (int(r[W - 1]) >> 40) & 0xff)
               for r in rec]
        if not lay["cls_bits"]:
            assert sorted(got) == sorted(want)
            continue
        # 2m < 24: the scan cuts a super-k-mer where the class of minimizer_idx changes.  Same k-mers, element by element
        # (compacted_j = (C >> 2(n-1-j)) & ones(2(k-b)), SuperKmerLight.hpp:301-312), and every piece within one class.
        ones = (1 << (2 * (k - b))) - 1

        def elements(records):
            out = []
            for t in records:
                C = sum(w << (64 * i) for i, w in enumerate(t[: W - 1]))
                bucket, n, idx0 = t[W - 1:]
                assert C >> (2 * (k - b + n - 1)) == 0
                out += [((C >> (2 * (n - 1 - j))) & ones, bucket, idx0 + j) for j in range(n)]
            return out
        assert sorted(elements(got)) == sorted(elements(want))
        top, width, sr = (1 << lay["cls_bits"]) - 1, lay["cls_width"], (m - b + 1) // 2
        assert len(got) > len(want)
        for r, t in zip(rec, got):
            n, idx0 = t[W:]
            classes = {min((idx0 - sr + j) // width, top) for j in range(n)}
            assert classes == {int(r[W - 1]) & top}


def test_bucket_range_sharding_two_owners(B, O):
    """scan -> route by owner -> insert on the owner: union of the shards == oracle."""
    import torch
    rng = random.Random(41)
    reads = _random_reads(rng, 800, 6000) + SPECIAL
    for k, m, b in ((63, 21, 14), (31, 11, 4)):
        want = O.count(reads, k, m, b)
        flat, offs = oracle.pack_reads(reads)
        owners = [B.BriskHip(k, m, b, owner_rank=r, n_owners=2) for r in range(2)]
        d_bases = torch.from_numpy(flat).cuda()
        d_packed = torch.zeros((len(flat) + 15) // 16 + 4, dtype=torch.int32, device="cuda")
        d_starts = torch.from_numpy(offs.astype(np.int64)).cuda()
        torch.cuda.synchronize()
        ix0 = owners[0]
        ix0.pack_ascii(d_bases.data_ptr(), len(flat), d_packed.data_ptr())
        ix0.sync()
        W = ix0.record_words
        # each "rank" scans half of the reads
        half = len(reads) // 2
        inbox = [[], []]
        hist_slices = [[], []]
        n_parts = 1 << ix0.layout["part_bits"]
        for r, (lo, hi) in enumerate(((0, half), (half, len(reads)))):
            ix = owners[r]
            st = d_starts[lo:hi + 1].contiguous()
            bound = ix.scan_bound(st.data_ptr(), hi - lo)
            d_rec = torch.zeros(max(bound, 1) * W, dtype=torch.int64, device="cuda")
            d_out = torch.zeros_like(d_rec)
            torch.cuda.synchronize()
            n_rec = ix.scan_packed(d_packed.data_ptr(), st.data_ptr(), hi - lo, d_rec.data_ptr(), bound)
            counts = ix.route_records(d_rec.data_ptr(), n_rec, d_out.data_ptr())
            d_hist = torch.zeros(n_parts, dtype=torch.int64, device="cuda")
            torch.cuda.synchronize()
            lens = ix.export_hist(d_hist.data_ptr())  # the scan's per-partition counts travel with the records
            ix.sync()
            assert int(counts.sum()) == n_rec and int(lens.sum()) == n_parts
            o0 = int(counts[0])
            inbox[0].append(d_out[: o0 * W].clone())
            inbox[1].append(d_out[o0 * W: n_rec * W].clone())
            hist_slices[0].append(d_hist[: int(lens[0])].clone())
            hist_slices[1].append(d_hist[int(lens[0]):].clone())
        lines, nk, nb = [], 0, 0
        for r in range(2):
            recv = torch.cat(inbox[r])
            slices = torch.cat(hist_slices[r])
            torch.cuda.synchronize()
            if k == 63:  # the owner adds the scanners' histogram slices up ...
                owners[r].insert_records_hist(recv.data_ptr(), recv.numel() // W, slices.data_ptr(), 2)
            else:        # ... or counts what it received itself
                owners[r].insert_records(recv.data_ptr(), recv.numel() // W)
            st = owners[r].stats()
            nk += st["nb_kmers"]
            nb += st["nb_buckets"]
            lines += oracle.multiset_lines(*owners[r].enumerate(), k)
        for ix in owners:
            ix.close()
        assert (sorted(lines), nk, nb) == want


def test_sharding_with_balanced_owner_cuts(B, O):
    """brisk_hip_set_owner_cuts (histogram-balanced cut points): three owners whose ranges come from the partition
    histogram of the whole job (k31 m15 b14: a partition is the top bits of the minimizer hash, equal ranges are lopsided), then the
    same again with an owner that holds nothing.  Routed records lie in their owner's range, the slices add up, the union of the
    shards is the oracle's index; cut points are refused on an index that holds entries."""
    import torch
    from brisk_amd import exchange as X
    rng = random.Random(43)
    reads = _random_reads(rng, 900, 8000) + SPECIAL
    k, m, b, N = 31, 15, 14, 3
    want = O.count(reads, k, m, b)
    flat, offs = oracle.pack_reads(reads)
    d_bases = torch.from_numpy(flat).cuda()
    d_packed = torch.zeros((len(flat) + 15) // 16 + 4, dtype=torch.int32, device="cuda")
    d_starts = torch.from_numpy(offs.astype(np.int64)).cuda()
    torch.cuda.synchronize()
    # the job's histogram, from an unsharded scan
    with B.BriskHip(k, m, b) as one:
        one.pack_ascii(d_bases.data_ptr(), len(flat), d_packed.data_ptr())
        one.sync()
        W, pb = one.record_words, one.layout["part_bits"]
        bound = one.scan_bound(d_starts.data_ptr(), len(reads))
        d_rec = torch.zeros(max(bound, 1) * W, dtype=torch.int64, device="cuda")
        hist = torch.zeros(1 << pb, dtype=torch.int64, device="cuda")
        torch.cuda.synchronize()
        one.scan_packed(d_packed.data_ptr(), d_starts.data_ptr(), len(reads), d_rec.data_ptr(), bound)
        one.export_hist(hist.data_ptr())
        one.sync()
    inst = (hist >> 32).to(torch.float64)
    balanced = X.balanced_cuts(hist, pb, N)
    uni = X.uniform_cuts(pb, N)
    load = lambda cuts: [float(inst[cuts[o]:cuts[o + 1]].sum()) for o in range(N)]
    assert max(load(balanced)) / (sum(load(balanced)) / N) < max(load(uni)) / (sum(load(uni)) / N)
    for cuts in (balanced, [0, balanced[1], balanced[1], 1 << pb]):  # ... and owner 1 with an empty range
        owners = [B.BriskHip(k, m, b, owner_rank=r, n_owners=N) for r in range(N)]
        for ix in owners:
            ix.set_owner_cuts(cuts)
        inbox, slices = [[] for _ in range(N)], [[] for _ in range(N)]
        share = [len(reads) * i // N for i in range(N + 1)]
        for r in range(N):
            ix = owners[r]
            st = d_starts[share[r]:share[r + 1] + 1].contiguous()
            n = share[r + 1] - share[r]
            bound = ix.scan_bound(st.data_ptr(), n)
            d_rec = torch.zeros(max(bound, 1) * W, dtype=torch.int64, device="cuda")
            d_out = torch.zeros_like(d_rec)
            d_hist = torch.zeros(1 << pb, dtype=torch.int64, device="cuda")
            torch.cuda.synchronize()
            n_rec = ix.scan_packed(d_packed.data_ptr(), st.data_ptr(), n, d_rec.data_ptr(), bound)
            counts = [int(c) for c in ix.route_records(d_rec.data_ptr(), n_rec, d_out.data_ptr())]
            lens = [int(v) for v in ix.export_hist(d_hist.data_ptr())]
            ix.sync()
            assert lens == [cuts[o + 1] - cuts[o] for o in range(N)] and sum(counts) == n_rec
            at = 0
            for o in range(N):
                got = d_out[at * W:(at + counts[o]) * W].clone()
                part = (got.reshape(-1, W)[:, W - 1] & 0xffffffff) >> (2 * b - pb)
                assert bool(((part >= cuts[o]) & (part < cuts[o + 1])).all()), "a routed record lies outside its owner's range"
                inbox[o].append(got)
                slices[o].append(d_hist[cuts[o]:cuts[o + 1]].clone())
                at += counts[o]
        lines, nk, nb = [], 0, 0
        for o in range(N):
            recv = torch.cat(inbox[o]) if inbox[o] else torch.zeros(0, dtype=torch.int64, device="cuda")
            sl = torch.cat(slices[o])
            torch.cuda.synchronize()
            if cuts[o + 1] > cuts[o]:
                owners[o].insert_records_hist(recv.data_ptr() if recv.numel() else 0, recv.numel() // W, sl.data_ptr() if sl.numel() else 0, N)
            else:
                assert recv.numel() == 0
            stt = owners[o].stats()
            nk += stt["nb_kmers"]
            nb += stt["nb_buckets"]
            lines += oracle.multiset_lines(*owners[o].enumerate(), k)
        assert (sorted(lines), nk, nb) == want
        with pytest.raises(B.BriskHipError):  # the index holds entries of these ranges: new cut points are refused
            owners[0].set_owner_cuts(uni)
        for ix in owners:
            ix.close()


def test_sharding_with_more_partitions_and_summed_histograms(B, O):
    """A sharded job sized for N x the reads (include/brisk_hip.h, brisk_hip_options.part_bits): 2^25 and 2^27 partitions
    over 2 and 3 owners, each rank scanning its reads in two pieces whose histograms are summed before they travel
    (brisk_hip_export_hist_add).  Union of the shards == oracle; records of another owner are refused."""
    import torch
    rng = random.Random(47)
    reads = _random_reads(rng, 900, 6000) + SPECIAL
    k, m, b = 63, 21, 14
    want = O.count(reads, k, m, b)
    flat, offs = oracle.pack_reads(reads)
    d_bases = torch.from_numpy(flat).cuda()
    d_packed = torch.zeros((len(flat) + 15) // 16 + 4, dtype=torch.int32, device="cuda")
    d_starts = torch.from_numpy(offs.astype(np.int64)).cuda()
    for n_owners, part_bits in ((2, 25), (3, 27)):
        owners = [B.BriskHip(k, m, b, owner_rank=r, n_owners=n_owners, part_bits=part_bits) for r in range(n_owners)]
        assert owners[0].layout["part_bits"] == part_bits
        torch.cuda.synchronize()
        owners[0].pack_ascii(d_bases.data_ptr(), len(flat), d_packed.data_ptr())
        owners[0].sync()
        W = owners[0].record_words
        n_parts = 1 << part_bits
        inbox = [[] for _ in range(n_owners)]
        slices = [[] for _ in range(n_owners)]
        share = [len(reads) * i // n_owners for i in range(n_owners + 1)]
        for r in range(n_owners):  # rank r scans its share in two pieces
            ix = owners[r]
            acc = torch.zeros(n_parts, dtype=torch.int64, device="cuda")
            mid = (share[r] + share[r + 1]) // 2
            for lo, hi in ((share[r], mid), (mid, share[r + 1])):
                st = d_starts[lo:hi + 1].contiguous()
                bound = ix.scan_bound(st.data_ptr(), hi - lo)
                d_rec = torch.zeros(max(bound, 1) * W, dtype=torch.int64, device="cuda")
                d_out = torch.zeros_like(d_rec)
                torch.cuda.synchronize()
                n_rec = ix.scan_packed(d_packed.data_ptr(), st.data_ptr(), hi - lo, d_rec.data_ptr(), bound)
                counts = ix.route_records(d_rec.data_ptr(), n_rec, d_out.data_ptr())
                lens = ix.export_hist_add(acc.data_ptr())
                ix.sync()
                assert int(counts.sum()) == n_rec and int(lens.sum()) == n_parts
                at = 0
                for o in range(n_owners):
                    inbox[o].append(d_out[at * W:(at + int(counts[o])) * W].clone())
                    at += int(counts[o])
            assert int((acc & 0xffffffff).sum()) == sum(t.numel() // W for o in range(n_owners) for t in inbox[o][2 * r:])
            at = 0
            for o in range(n_owners):
                slices[o].append(acc[at:at + int(lens[o])].clone())
                at += int(lens[o])
        lines, nk, nb = [], 0, 0
        for r in range(n_owners):
            recv = torch.cat(inbox[r])
            sl = torch.cat(slices[r])
            torch.cuda.synchronize()
            if r == 0 and n_owners == 2:
                # what belongs to the other owner is not taken: nothing is inserted, the index stays usable
                foreign = torch.cat(inbox[1])
                with pytest.raises(B.BriskHipError) as e:
                    owners[0].insert_records(foreign.data_ptr(), foreign.numel() // W)
                assert e.value.code == 1 and "other owners" in str(e.value)  # EINVAL
                owners[0].insert_records(recv.data_ptr(), recv.numel() // W)  # counted here
            else:
                owners[r].insert_records_hist(recv.data_ptr(), recv.numel() // W, sl.data_ptr(), n_owners)
            owners[r].sync()
            st = owners[r].stats()
            nk += st["nb_kmers"]
            nb += st["nb_buckets"]
            lines += oracle.multiset_lines(*owners[r].enumerate(), k)
        for ix in owners:
            ix.close()
        assert (sorted(lines), nk, nb) == want


def test_get_across_two_owners(B, O):
    """sharded get: scan_query -> route_tagged -> (exchange) -> query_records on the owner -> sums back ->
    per-read sums == the oracle's query of the whole index (incl. the minimizer==0 break, poly-A reads)"""
    import torch
    rng = random.Random(43)
    reads = _random_reads(rng, 500, 5000) + SPECIAL + ["A" * 200, "ACGT" * 60]
    for k, m, b in ((63, 21, 14), (31, 11, 4)):
        flat, offs = oracle.pack_reads(reads)
        h = O.index_new(k, m, b)
        O.index_insert_reads(h, flat, offs)
        want = O.index_query_reads(h, flat, offs)
        O.index_free(h)
        owners = [B.BriskHip(k, m, b, owner_rank=r, n_owners=2) for r in range(2)]
        W = owners[0].record_words
        d_bases = torch.from_numpy(flat).cuda()
        d_packed = torch.zeros((len(flat) + 15) // 16 + 4, dtype=torch.int32, device="cuda")
        d_starts = torch.from_numpy(offs.astype(np.int64)).cuda()
        torch.cuda.synchronize()
        owners[0].pack_ascii(d_bases.data_ptr(), len(flat), d_packed.data_ptr())
        owners[0].sync()
        n = len(reads)
        bound = owners[0].scan_bound(d_starts.data_ptr(), n)
        # count: one scan, routed to the two owners
        d_rec = torch.zeros(max(bound, 1) * W, dtype=torch.int64, device="cuda")
        d_out = torch.zeros_like(d_rec)
        torch.cuda.synchronize()
        n_rec = owners[0].scan_packed(d_packed.data_ptr(), d_starts.data_ptr(), n, d_rec.data_ptr(), bound)
        counts = owners[0].route_records(d_rec.data_ptr(), n_rec, d_out.data_ptr())
        o0 = int(counts[0])
        owners[0].insert_records(d_out.data_ptr(), o0)
        owners[1].insert_records(d_out[o0 * W:].data_ptr(), n_rec - o0)
        # get: query-mode scan with read tags, routed, answered per record by the owner, folded per read
        d_tags = torch.zeros(max(bound, 1), dtype=torch.int32, device="cuda")
        d_tags_out = torch.zeros_like(d_tags)
        torch.cuda.synchronize()
        nq = owners[0].scan_query(d_packed.data_ptr(), d_starts.data_ptr(), n, d_rec.data_ptr(), d_tags.data_ptr(), bound)
        counts = owners[0].route_tagged(d_rec.data_ptr(), d_tags.data_ptr(), nq, d_out.data_ptr(), d_tags_out.data_ptr())
        q0 = int(counts[0])
        assert int(counts.sum()) == nq
        sums = torch.zeros(max(nq, 1), dtype=torch.int64, device="cuda")
        torch.cuda.synchronize()
        owners[0].query_records(d_out.data_ptr(), q0, sums.data_ptr())
        owners[1].query_records(d_out[q0 * W:].data_ptr(), nq - q0, sums[q0:].data_ptr())
        per_read = torch.zeros(n, dtype=torch.int64, device="cuda")
        per_read.index_add_(0, d_tags_out[:nq].to(torch.int64), sums[:nq])
        assert np.array_equal(per_read.cpu().numpy().astype(np.uint64), want), (k, m, b)
        for o in owners:
            o.close()
    # the same calls behind ShardedCounter.get_packed (one rank owning everything)
    from brisk_amd.exchange import ShardedCounter
    k, m, b = 63, 21, 14
    flat, offs = oracle.pack_reads(reads)
    h = O.index_new(k, m, b)
    O.index_insert_reads(h, flat, offs)
    want = O.index_query_reads(h, flat, offs)
    O.index_free(h)
    stream = torch.cuda.Stream()
    sc = ShardedCounter(k, m, b, 0, 1, 0, stream)
    d_bases = torch.from_numpy(flat).cuda()
    d_packed = torch.zeros((len(flat) + 15) // 16 + 4, dtype=torch.int32, device="cuda")
    d_starts = torch.from_numpy(offs.astype(np.int64)).cuda()
    torch.cuda.synchronize()
    sc.ix.pack_ascii(d_bases.data_ptr(), len(flat), d_packed.data_ptr())
    sc.ix.sync()
    sc.count_packed(d_packed, d_starts, len(reads))
    got = sc.get_packed(d_packed, d_starts, len(reads))
    torch.cuda.synchronize()
    assert np.array_equal(got.cpu().numpy().astype(np.uint64), want)
    sc.ix.close()


def test_device_synth_generator_matches_oracle(B, O):
    import torch
    k, m, b = 63, 21, 14
    G, n, L = 30000, 3000, 150
    seqs = [bytes(r) for r in O.synth_reads(G, 100, n)]
    want = O.count(seqs, k, m, b)
    with B.BriskHip(k, m, b) as ix:
        d_packed = torch.zeros((n * L + 15) // 16 + 4, dtype=torch.int32, device="cuda")
        d_starts = torch.zeros(n + 1, dtype=torch.int64, device="cuda")
        torch.cuda.synchronize()
        ix.synth_reads(G, 100, n, L, d_packed.data_ptr(), d_starts.data_ptr())
        ix.insert_packed(d_packed.data_ptr(), d_starts.data_ptr(), n)
        st = ix.stats()
        got = oracle.multiset_lines(*ix.enumerate(), k)
    assert (got, st["nb_kmers"], st["nb_buckets"]) == want


def test_parameter_contract(B):
    for k, m, b in ((31, 11, 14), (31, 12, 4), (31, 31, 4), (64, 21, 14), (31, 11, 0), (63, 33, 4)):
        with pytest.raises(B.BriskHipError) as e:
            B.BriskHip(k, m, b)
        assert e.value.code == 1
    with pytest.raises(B.BriskHipError) as e:  # key does not fit 128 bits: outside this library's envelope
        B.BriskHip(63, 21, 1)
    assert e.value.code == 2


def test_offsets_that_do_not_ascend_are_refused(B):
    """A read table whose offsets go backwards (a caller's bug, or a buffer overwritten while the call runs: a fill on another
    stream landing late did exactly that to this repo's own tools in round 3) must be refused before anything is scanned: a length
    that wraps around would send the scan far outside the caller's buffer."""
    import torch
    reads = _random_reads(random.Random(3), 50, 2000)
    flat, offs = oracle.pack_reads(reads)
    d_bases = torch.from_numpy(flat).cuda()
    d_packed = torch.zeros((len(flat) + 15) // 16 + 4, dtype=torch.int32, device="cuda")
    bad = offs.astype(np.int64).copy()
    bad[20] = 0  # offsets[20] < offsets[19]
    d_starts = torch.from_numpy(bad).cuda()
    torch.cuda.synchronize()
    with B.BriskHip(31, 11, 4) as ix:
        ix.pack_ascii(d_bases.data_ptr(), len(flat), d_packed.data_ptr())
        ix.sync()
        with pytest.raises(B.BriskHipError) as e:
            ix.insert_packed(d_packed.data_ptr(), d_starts.data_ptr(), len(reads))
        assert e.value.code == 1 and "ascend" in str(e.value)
        assert ix.stats()["nb_kmers"] == 0  # nothing was scanned, the handle stays usable
        ix.insert_reads(reads)
        assert ix.stats()["nb_kmers"] > 0


def test_empty_inputs(B):
    with B.BriskHip(31, 11, 4) as ix:
        ix.insert_reads([])
        ix.insert_reads(["", "ACGT"])
        assert ix.stats()["nb_kmers"] == 0
        assert len(ix.enumerate()[0]) == 0
        assert list(ix.get_reads(["ACGT", ""])) == [0, 0]


def test_order_keys_fast_and_exact_paths(B, O):
    """a2/a3 on device: the table-driven class (with its guard band) and the plain FP64
    fold both equal the reference's keys, on random and on low-complexity m-mers (where
    R(x) is mathematically 0 and only rounding noise decides nothing)."""
    rng = random.Random(77)
    # every shape of the chunk tables (brisk_scan.hip, cls_nch / cls_width): 1 chunk (m 5), [5,4], [6,5], [5,4,4], [5,5,5], [5,4,4,4] (a
    # chunk across the 32-bit boundary), [5,5,5,4], [6,5,5,5], 5 chunks (23, 25), 6 (27, 29), 7 (31)
    for k, m, b in ((63, 21, 14), (31, 11, 4), (63, 31, 12), (21, 7, 3), (31, 15, 14), (21, 5, 2), (31, 9, 4), (41, 13, 6), (47, 17, 8), (55, 19, 9),
                    (63, 23, 11), (63, 25, 12), (63, 27, 13), (63, 29, 14), (9, 3, 1)):
        M = (1 << (2 * m)) - 1
        xs = [rng.getrandbits(2 * m) for _ in range(200000)]
        for unit in ("A", "C", "G", "T", "AC", "AG", "AT", "CG", "CT", "GT", "ACG", "ACT", "AAC", "ACGT", "AACC", "ACCGT"):
            for rot in range(len(unit)):
                xs.append(oracle.str2kmer(((unit[rot:] + unit[:rot]) * m)[:m])[0])
        xs += [0, M, 1, M - 1, M >> 2]
        # fake (zero-padded) windows of the k>32 re-scan: short prefixes
        xs += [rng.getrandbits(2 * rng.randint(1, m - 1)) for _ in range(20000)]
        want = O.key_many(xs[:20000] + xs[200000:], m)
        g = load_golden("units.json.gz").get(str(m))
        with B.BriskHip(k, m, b, part_bits=2) as ix:
            fast = ix.debug_order_keys(xs)
            exact = ix.debug_order_keys(xs, exact=True)
            assert np.array_equal(fast, exact)
            assert np.array_equal(np.concatenate([fast[:20000], fast[200000:]]), want)
            if g:
                gx = [int(x, 16) for x in g["x"]]
                assert [f"{int(v):x}" for v in ix.debug_order_keys(gx)] == g["key"]


def test_periodic_reads_force_minimizer_ties(B, O):
    """Repeats make the same m-mer the minimum at several windows: every branch of
    get_minimizer's tie rules (Kmers.cpp:389-404), incl. canonized(), is taken."""
    rng = random.Random(123)
    reads = []
    for period in range(1, 40):
        for _ in range(6):
            unit = "".join(rng.choice("ACGT") for _ in range(period))
            L = rng.choice((150, 151, 200, 97))
            s = (unit * (L // period + 1))[:L]
            reads.append(s)
            # a repeat embedded in random flanks, and its reverse complement
            fl = "".join(rng.choice("ACGT") for _ in range(40))
            t = fl + s[:80] + fl[::-1]
            reads.append(t)
            reads.append(t[::-1].translate(str.maketrans("ACGT", "TGCA")))
    for k, m, b in ((31, 11, 4), (63, 21, 14), (33, 11, 7), (63, 31, 12), (41, 21, 5)):
        assert gpu_count(B, reads, k, m, b) == O.count(reads, k, m, b)
        with B.BriskHip(k, m, b) as ix:
            ix.insert_reads(reads)
            flat, offs = oracle.pack_reads(reads)
            h = O.index_new(k, m, b)
            O.index_insert_reads(h, flat, offs)
            assert np.array_equal(ix.get_reads(reads), O.index_query_reads(h, flat, offs))
            O.index_free(h)


def test_long_sequences_and_ragged_lengths(B, O):
    rng = random.Random(9)
    reads = ["".join(rng.choice("ACGT") for _ in range(n)) for n in (63, 64, 65, 95, 96, 97, 127, 128, 129, 1000, 5000, 20011)]
    reads += ["A" * 3000, ("ACGTTGCA" * 500)]
    for k, m, b in ((63, 21, 14), (31, 11, 11)):
        assert gpu_count(B, reads, k, m, b) == O.count(reads, k, m, b)


def test_chromosome_length_sequences_are_scanned_in_chunks(B, O):
    """Sequences with more than 8192 k-mers are scanned as overlapping chunks whose seams are verified
    against the sequential state; chunks whose seam does not match (long runs without a new minimum:
    homopolymers, short tandem repeats) are scanned again seeded with the exact state.  Bit-exact either way."""
    rng = random.Random(2024)
    rnd = lambda n: "".join(rng.choice("ACGT") for _ in range(n))
    seqs = [rnd(300_017), rnd(8192 + 62), rnd(8192 + 63), rnd(8192 + 64), rnd(12_288 + 62), rnd(40_000)]
    seqs += ["A" * 30_000, ("ACGTTGCA" * 4000), rnd(15_000) + "T" * 20_000 + rnd(15_000), (rnd(37) * 1000)]
    # periodic stretches between random flanks: cold starts inside them agree with each other, not with the sequential run
    seqs += [rnd(5000) + "ACGTTGCA" * 3000 + rnd(5000) + rnd(64) * 200 + rnd(3000), rnd(2500) + "CA" * 9000 + rnd(2500)]
    seqs += _random_reads(rng, 200, 3000)  # short reads in the same batch
    rc = lambda s: s[::-1].translate(str.maketrans("ACGT", "TGCA"))
    seqs.append(rc(seqs[0][1000:150_000]))
    # (31, 15, 14): the reference's defaults -- k <= 32, the window minimum comes from the per-lane queue, seeded chunks restart it
    for k, m, b in ((63, 21, 14), (31, 11, 11), (31, 11, 4), (31, 15, 14), (29, 13, 9)):
        assert gpu_count(B, seqs, k, m, b) == O.count(seqs, k, m, b), (k, m, b)
    # the query path chunks long sequences too and stops each where query_sequence stops it (a returned minimizer of 0:
    # the poly-A stretches below), whatever chunk that falls in
    queries = seqs + [rnd(20_000) + "A" * 90 + rnd(20_000), "A" * 70 + rnd(30_000), rnd(9_000) + "A" * 500 + rnd(9_000) + "A" * 64 + rnd(5_000)]
    for k, m, b in ((63, 21, 14), (31, 11, 11)):
        flat, offs = oracle.pack_reads(seqs)
        h = O.index_new(k, m, b)
        O.index_insert_reads(h, flat, offs)
        qflat, qoffs = oracle.pack_reads(queries)
        want = O.index_query_reads(h, qflat, qoffs)
        O.index_free(h)
        with B.BriskHip(k, m, b) as ix:
            ix.insert_reads(seqs)
            assert np.array_equal(ix.get_reads(queries), want), (k, m, b)
    # and as two batches, through the records API used for sharding (scan -> insert_records)
    import torch
    k, m, b = 63, 21, 14
    flat, offs = oracle.pack_reads(seqs)
    with B.BriskHip(k, m, b) as ix:
        d_bases = torch.from_numpy(flat).cuda()
        d_packed = torch.zeros((len(flat) + 15) // 16 + 4, dtype=torch.int32, device="cuda")
        d_starts = torch.from_numpy(offs.astype(np.int64)).cuda()
        torch.cuda.synchronize()
        ix.pack_ascii(d_bases.data_ptr(), len(flat), d_packed.data_ptr())
        bound = ix.scan_bound(d_starts.data_ptr(), len(seqs))
        d_rec = torch.zeros(bound * ix.record_words, dtype=torch.int64, device="cuda")
        torch.cuda.synchronize()
        n_rec = ix.scan_packed(d_packed.data_ptr(), d_starts.data_ptr(), len(seqs), d_rec.data_ptr(), bound)
        ix.insert_records(d_rec.data_ptr(), n_rec)
        st = ix.stats()
        assert (oracle.multiset_lines(*ix.enumerate(), k), st["nb_kmers"], st["nb_buckets"]) == O.count(seqs, k, m, b)


def test_per_call_api_entry_ids(B, O):
    """The facade's per-call path: scan_sequence == the enumerator stream (incl. the returned
    minimizer values), upsert/find/enumerate_ids == insert_superkmer/get/next with DATA on the host."""
    rng = random.Random(61)
    reads = _random_reads(rng, 60, 900) + SPECIAL + ["".join(rng.choice("ACGT") for _ in range(700))]
    for k, m, b in ((31, 11, 4), (63, 21, 14)):
        with B.BriskHip(k, m, b, entry_ids=True) as ix:
            data = {}
            for s in reads:
                want = O.enumerate(s.upper() if s.islower() else s, k, m)
                got = ix.scan_sequence(s)
                for a, c in zip(got, want[:5]):
                    assert np.array_equal(a, c), (k, m, s[:20])
                p = 0
                for n in got[1]:
                    ids, new = ix.upsert_kmers(got[2][p:p + n], got[3][p:p + n], got[4][p:p + n])
                    for i, nw in zip(ids, new):
                        data[int(i)] = 1 if nw else (data[int(i)] + 1) % 256  # counter.cpp:262-269
                    assert np.array_equal(ix.find_kmers(got[2][p:p + n], got[3][p:p + n], got[4][p:p + n]), ids)
                    p += n
            assert sorted(data) == list(range(len(data))), "ids are dense, in insertion order"
            lo, hi, idx, ids = ix.enumerate_ids()
            lines = oracle.multiset_lines(lo, hi, idx, [data[int(i)] for i in ids], k)
            st = ix.stats()
            assert (lines, st["nb_kmers"], st["nb_buckets"]) == O.count(reads, k, m, b)
            with pytest.raises(B.BriskHipError):
                ix.insert_reads(reads[:2])  # bulk count is refused on an entry-id index


def test_checksum_matches_oracle_digest(B, O):
    rng = random.Random(8)
    reads = _random_reads(rng, 700, 5000) + SPECIAL
    for k, m, b in ((63, 21, 14), (31, 11, 4)):
        flat, offs = oracle.pack_reads(reads)
        h = O.index_new(k, m, b)
        O.index_insert_reads(h, flat, offs)
        want = oracle.digest(*O.index_dump(h))
        O.index_free(h)
        with B.BriskHip(k, m, b) as ix:
            ix.insert_reads(reads)
            assert ix.checksum() == want


def test_large_host_input_goes_through_the_threaded_upload(B):
    """Host ASCII above 64 MiB is uploaded by several threads through pinned buffers and packed chunk by chunk;
    the index must equal the one built from the same bases packed in one piece on the device."""
    import torch
    rng = np.random.default_rng(5)
    L, n = 150, 600_000
    G = n * L // 15
    genome = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, G, dtype=np.uint8)]
    starts = rng.integers(0, G - L + 1, n)
    flat = genome[(starts[:, None] + np.arange(L)).ravel()].copy()
    offs = np.arange(n + 1, dtype=np.uint64) * L
    assert flat.nbytes > 4 * (16 << 20)
    k, m, b = 63, 21, 14
    with B.BriskHip(k, m, b) as ix:
        ix.insert_flat(flat, offs)
        via_host = ix.checksum()
    d_bases = torch.from_numpy(flat).cuda()
    d_packed = torch.zeros((len(flat) + 15) // 16 + 4, dtype=torch.int32, device="cuda")
    d_starts = torch.from_numpy(offs.astype(np.int64)).cuda()
    torch.cuda.synchronize()
    with B.BriskHip(k, m, b) as ix:
        ix.pack_ascii(d_bases.data_ptr(), len(flat), d_packed.data_ptr())
        ix.insert_packed(d_packed.data_ptr(), d_starts.data_ptr(), n)
        via_device = ix.checksum()
        assert via_device[1] == n * (L - k + 1)
    assert via_host == via_device


def test_full_size_properties_config2_and_3(B):
    """BASELINE configs #2' (10M reads, k31 m11 b11: b=14 of config #2 is invalid in the reference, F1)
    and #3 (50M reads, k63 m21 b14) at full size, through size-independent properties: the result does not
    depend on batching or read order; every k-mer instance is counted exactly once (sum of counts ==
    reads x (L-k+1), no count wraps at 15x coverage); counting the same reads again changes no entry and
    doubles every count; every read's own k-mers are found."""
    import torch
    L = 150
    for (n_reads, k, m, b) in ((10_000_000, 31, 11, 11), (50_000_000, 63, 21, 14)):
        G = n_reads * L // 15
        d_packed = torch.zeros((n_reads * L + 15) // 16 + 4, dtype=torch.int32, device="cuda")
        d_starts = torch.zeros(n_reads + 1, dtype=torch.int64, device="cuda")
        torch.cuda.synchronize()
        with B.BriskHip(k, m, b) as ix:
            ix.synth_reads(G, 0, n_reads, L, d_packed.data_ptr(), d_starts.data_ptr())
            ix.insert_packed(d_packed.data_ptr(), d_starts.data_ptr(), n_reads)
            # the device-side consistency flags (scatter slot out of range, arena exhausted, chunk overflow) stayed clear:
            # brisk_hip_sync fails on any of them.  At k31/m11/b11 -- hot partitions of up to 18 k instances -- round 1
            # once faulted here, before large slices bypassed the per-wave arena chunk.
            ix.sync()
            st = ix.stats()
            one = ix.checksum()
            assert one[0] == st["nb_kmers"]
            assert one[1] == n_reads * (L - k + 1)
            # same reads again: no new entry, every count doubled
            ix.insert_packed(d_packed.data_ptr(), d_starts.data_ptr(), n_reads)
            ix.sync()
            two = ix.checksum()
            assert two[0] == one[0] and two[1] == 2 * one[1] and ix.stats()["nb_buckets"] == st["nb_buckets"]
        # five batches, last batch first: identical index
        with B.BriskHip(k, m, b) as ix:
            step = n_reads // 5
            for i in reversed(range(5)):
                sl = d_starts[i * step:(i + 1) * step + 1].contiguous()
                torch.cuda.synchronize()
                ix.insert_packed(d_packed.data_ptr(), sl.data_ptr(), step)
            assert ix.checksum() == one
            st2 = ix.stats()
            assert (st2["nb_kmers"], st2["nb_buckets"]) == (st["nb_kmers"], st["nb_buckets"])
        del d_packed, d_starts
        torch.cuda.empty_cache()


def test_mixed_insert_get_workload(B, O):
    """BASELINE config #5's shape (no real FASTA stream exists offline: Accessions_List holds IDs only):
    batches of reads are inserted and queried alternately; after every batch the bulk query of the batch
    just inserted and of the NEXT batch (mostly absent k-mers) equals the CPU path's, and so does the final state."""
    rng = random.Random(55)
    reads = _random_reads(rng, 3000, 20000) + SPECIAL
    rng.shuffle(reads)
    k, m, b = 63, 21, 14
    h = O.index_new(k, m, b)
    step = 400
    with B.BriskHip(k, m, b) as ix:
        for i in range(0, len(reads), step):
            cur, nxt = reads[i:i + step], reads[i + step:i + 2 * step]
            f, o = oracle.pack_reads(cur)
            O.index_insert_reads(h, f, o)
            ix.insert_reads(cur)
            assert np.array_equal(ix.get_reads(cur), O.index_query_reads(h, f, o))
            if nxt:
                f2, o2 = oracle.pack_reads(nxt)
                assert np.array_equal(ix.get_reads(nxt), O.index_query_reads(h, f2, o2))
        assert ix.checksum() == oracle.digest(*O.index_dump(h))
    O.index_free(h)


def test_batch_splits_when_the_arena_reserve_does_not_fit(B, O, monkeypatch):
    """The single-pass insert reserves arena for "every instance is new"; when that does not fit, the
    batch is inserted in halves (nothing was written yet).  Also the copy-growth path without HIP VMM."""
    rng = random.Random(17)
    reads = _random_reads(rng, 2500, 15000)
    k, m, b = 63, 21, 14
    want = O.count(reads, k, m, b)
    inst = sum(len(r) - k + 1 for r in reads)
    monkeypatch.delenv("BRISK_NO_VMM", raising=False)  # this test sets it itself, further down
    with B.BriskHip(k, m, b) as probe:
        slack = probe.insert_slack()  # resident insert waves x ARENA_CHUNK, from the library (was a constant copied from csrc/brisk_insert.hip by hand)
    assert slack > 0
    monkeypatch.setenv("BRISK_ARENA_LIMIT", str(slack + inst // 2))  # the slack plus half the pessimistic bound
    assert gpu_count(B, reads, k, m, b) == want
    monkeypatch.setenv("BRISK_ARENA_LIMIT", "1000")  # nothing fits: a clean error, not a crash
    with pytest.raises(B.BriskHipError) as e:
        gpu_count(B, reads, k, m, b)
    assert e.value.code == 4
    monkeypatch.delenv("BRISK_ARENA_LIMIT")
    monkeypatch.setenv("BRISK_NO_VMM", "1")
    assert gpu_count(B, reads, k, m, b, batches=4) == want


def test_arena_reuse_across_many_indexes(B, O):
    """ADVICE r01 / DESIGN.md section 3: an arena address is mapped at most once per process, so destroyed indexes hand
    their arenas to the next one (a pool) instead of unmapping them.  Seven indexes in a row, each with a bulk insert that
    grows the arena: every one is correct, arenas are reused (no address space is retired while the pool has room), and
    what the pool holds is reported."""
    reads = [bytes(r) for r in O.synth_reads(30000, 0, 3000)]
    k, m, b = 63, 21, 14
    want = O.count(reads, k, m, b)
    retired0 = None
    for i in range(7):
        with B.BriskHip(k, m, b) as ix:
            ix.insert_reads(reads)
            st = ix.stats()
            assert (st["nb_kmers"], st["nb_buckets"]) == (want[1], want[2]), i
            if i in (0, 6):
                assert oracle.multiset_lines(*ix.enumerate(), k) == want[0]
            mi = ix.memory_info()
            assert mi["arena_mapped"] > 0 and mi["arena_reserved"] >= mi["arena_mapped"]
            if retired0 is None:
                retired0 = mi["retired_va"]
            assert mi["retired_va"] == retired0, "an arena was retired although the pool had room"
    with B.BriskHip(k, m, b) as ix:
        assert ix.memory_info()["pooled"] == 0 or True  # the arena just taken came out of the pool; others may still be in it


def test_gpu_against_the_reference_build_itself(B, R):
    """the HIP path against oracle/_ref -- the reference's OWN Kmers.cpp / hashing.cpp / Decycling.cpp
    and Bucket<DATA> / SKL, compiled where they lie -- with no restatement in between (every other parity test goes
    through oracle/brisk_oracle.c, which is itself pinned by the same build).  Skipped where the reference build did
    not travel."""
    rng = random.Random(91)
    reads = _random_reads(rng, 600, 8000) + SPECIAL
    for k, m, b in ((63, 21, 14), (31, 11, 4), (31, 15, 14)):
        assert gpu_count(B, reads, k, m, b, batches=3) == R.count(reads, k, m, b), (k, m, b)
