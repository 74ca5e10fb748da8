"""BASELINE config #5's protocol at a meaningful size on one GPU: config #3's reads written as .fa.gz, streamed by the C++
front-end (FastaBatcher, brisk_amd/include/brisk_fasta.hpp) into brisk_hip_insert_reads while a SECOND host thread
issues brisk_hip_get_reads against the same handle (brisk_amd/apps/brisk_count --mixed; reference protocol:
apps/counter.cpp:197-227 insert loop, :314-346 query loop, brisk/Brisk.hpp:102-147 under lock stripes).

What a concurrent get may observe is stated in include/brisk_hip.h ("Threads"): whole batches, never a torn one.  Checked:
  * every concurrent get of batch J's first reads equals what a strictly sequential run returns after SOME whole
    number of batches J' >= J (the sequential run is replayed here, batch by batch, on the same boundaries);
  * the gets issued after the last batch equal the sequential run's final answers;
  * the final digest equals the insert-only run's, and every k-mer instance was counted exactly once;
  * the final answers for batch 0's reads equal the ORACLE's (oracle/brisk_oracle.c) on an index built from every read
    of the job that overlaps them on the genome (a 63-mer of a random 50 Mbp genome occurs once, so no other read can
    touch their entries).
"""
import os
import subprocess
import sys
import zlib

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N_READS = int(os.environ.get("BRISK_MIXED_READS", "5000000"))
L, K, M, B, COVERAGE = 150, 63, 21, 14, 15
MASK = (1 << 64) - 1


def _mix(z):
    z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return z ^ (z >> np.uint64(31))


def _read_starts(n, genome_len, seed_r=2):
    """start position of every synthetic read (p = u(seed_r, 2r) mod (G - L + 1))"""
    with np.errstate(over="ignore"):
        i = np.arange(n, dtype=np.uint64) * np.uint64(2)
        u = _mix(np.uint64(seed_r) + (i + np.uint64(1)) * np.uint64(0x9E3779B97F4A7C15))
    return (u % np.uint64(genome_len - L + 1)).astype(np.int64)


@pytest.mark.gpu
def test_concurrent_insert_and_get_observe_whole_batches(tmp_path, O):
    import brisk_amd
    exe = os.path.join(ROOT, "brisk_amd", "apps", "brisk_count")
    if not os.path.exists(exe):
        brisk_amd.build_apps()
    G = N_READS * L // COVERAGE
    reads = O.synth_reads(G, 0, N_READS, L)  # (N, L) uint8, the bench's generator
    # FASTA.gz, one record per read
    fa = np.empty((N_READS, L + 3), dtype=np.uint8)
    fa[:, 0] = ord(">")
    fa[:, 1] = ord("\n")
    fa[:, 2:L + 2] = reads
    fa[:, L + 2] = ord("\n")
    path = str(tmp_path / "reads.fa.gz")
    co = zlib.compressobj(1, zlib.DEFLATED, 31)
    flat_fa = fa.reshape(-1)
    with open(path, "wb") as f:
        step = 64 << 20
        for lo in range(0, flat_fa.size, step):
            f.write(co.compress(flat_fa[lo:lo + step].tobytes()))
        f.write(co.flush())
    del fa, flat_fa

    env = dict(os.environ, BRISK_BATCH_BASES=str(max(N_READS * L // 12, 1 << 16)), BRISK_MIXED_SAMPLE="50")
    run = subprocess.run([exe, "--mixed", path, str(K), str(M), str(B)], capture_output=True, text=True, timeout=900, env=env)
    assert run.returncode == 0, run.stderr[-2000:]
    batches, gets, finals, dig = [], [], {}, None
    for line in run.stdout.splitlines():
        w = line.split()
        if w[0] == "batch":
            assert int(w[1]) == len(batches)
            batches.append(int(w[2]))
        elif w[0] == "get":
            gets.append((int(w[1]), [int(v) for v in w[2:]]))
        elif w[0] == "final":
            finals[int(w[1])] = [int(v) for v in w[2:]]
        elif w[0] == "digest":
            dig = [int(v) for v in w[1:]]
    assert sum(batches) == N_READS and len(batches) >= 8, batches
    assert len(gets) >= 1, "the second thread never got a call in between two batches"
    assert dig is not None and dig[1] == N_READS * (L - K + 1), "every k-mer instance counted once"

    # the strictly sequential run on the same batch boundaries; after every batch, the answers for every batch already in
    starts = np.concatenate(([0], np.cumsum(batches))).astype(np.int64)
    seq_answers = []  # seq_answers[j][s] = per-read sums of sample s after batches 0..j
    with brisk_amd.BriskHip(K, M, B) as ix:
        for j in range(len(batches)):
            lo, hi = int(starts[j]), int(starts[j + 1])
            flat = np.ascontiguousarray(reads[lo:hi].reshape(-1))
            offs = np.arange(hi - lo + 1, dtype=np.uint64) * np.uint64(L)
            ix.insert_flat(flat, offs)
            row = {}
            for s in range(j + 1):
                s_lo = int(starts[s])
                ns = min(50, batches[s])
                row[s] = [int(v) for v in ix.get_reads([bytes(r) for r in reads[s_lo:s_lo + ns]])]
            seq_answers.append(row)
        ent, sumc, d64 = ix.checksum()
    assert dig == [ent, sumc, d64], "concurrent run's final index differs from the insert-only run's"
    last = len(batches) - 1
    for s, sums in finals.items():
        assert sums == seq_answers[last][s], ("final get", s)
    later = 0
    for s, sums in gets:
        seen_at = [j for j in range(s, len(batches)) if seq_answers[j][s] == sums]
        assert seen_at, ("a concurrent get of batch %d matches no whole-batch state" % s, sums[:8])
        later += seen_at[0] > s
    assert len(finals) == len(batches)

    # the oracle on batch 0's sample: every read of the job that overlaps one of them on the genome
    pos = _read_starts(N_READS, G)
    order = np.argsort(pos, kind="stable")
    sorted_pos = pos[order]
    ns = min(50, batches[0])
    near = set()
    for p in pos[:ns]:
        a, b_ = np.searchsorted(sorted_pos, p - L + 1, "left"), np.searchsorted(sorted_pos, p + L - 1, "right")
        near.update(int(r) for r in order[a:b_])
    near = sorted(near)
    import oracle
    flat, offs = oracle.pack_reads([bytes(reads[r]) for r in near])
    h = O.index_new(K, M, B)
    O.index_insert_reads(h, flat, offs)
    sflat, soffs = oracle.pack_reads([bytes(r) for r in reads[:ns]])
    want = [int(v) for v in O.index_query_reads(h, sflat, soffs)]
    O.index_free(h)
    assert finals[0] == want, "final get of batch 0's first reads differs from the oracle"
    print("mixed workload: %d batches, %d concurrent gets (%d of them saw later batches)" % (len(batches), len(gets), later))


@pytest.mark.gpu
def test_big_host_batch_is_uploaded_in_pieces_and_checked(monkeypatch, O):
    """brisk_hip_insert_reads on more than 256 MiB of ASCII goes in sub-batches whose upload overlaps the previous piece's scan
    (insert_reads_pipelined).  The same reads in calls of 100,000 (far below the threshold: one upload, then the scan) give the same
    index (digest), with every stage hand-over checked (BRISK_VERIFY=1: the packed stream against the caller's bytes before and
    after each piece's scan, the records against the k-mer count) -- and again, in a process of its own, with the pipeline off."""
    import brisk_amd
    k, m, b, n = 63, 21, 14, 2_000_000
    G = n * L // 15
    flat = np.ascontiguousarray(O.synth_reads(G, 0, n, L).reshape(-1))  # 300 MB of ASCII, the bench's generator
    offs = np.arange(n + 1, dtype=np.uint64) * np.uint64(L)
    monkeypatch.setenv("BRISK_VERIFY", "1")
    with brisk_amd.BriskHip(k, m, b) as ix:
        step = 100_000
        for i in range(0, n, step):
            ix.insert_flat(flat[i * L:(i + step) * L], offs[:step + 1])
        want = ix.checksum()
    assert want[1] == n * (L - k + 1)
    with brisk_amd.BriskHip(k, m, b) as ix:
        ix.insert_flat(flat, offs)  # one call: 300 MB
        assert ix.checksum() == want
    # (BRISK_UPLOAD_PIPELINE is read once per process: the other setting runs in a process of its own)
    code = ("import sys, numpy as np; sys.path.insert(0, %r); import oracle, brisk_amd; O = oracle.Oracle(); n, L = %d, %d; "
            "flat = np.ascontiguousarray(O.synth_reads(n * L // 15, 0, n, L).reshape(-1)); offs = np.arange(n + 1, dtype=np.uint64) * np.uint64(L); "
            "ix = brisk_amd.BriskHip(%d, %d, %d); ix.insert_flat(flat, offs); print(ix.checksum())" % (ROOT, n, L, k, m, b))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=dict(os.environ, BRISK_UPLOAD_PIPELINE="0", BRISK_VERIFY="1"))
    assert out.returncode == 0, out.stderr[-1500:]
    assert out.stdout.strip().splitlines()[-1] == str(want)
