"""Host FASTA / FASTA.gz front-end (brisk_amd/include/brisk_fasta.hpp) against the
segmentation rules of the reference's harness (apps/counter.cpp:130-190), restated in
oracle.fasta_sequences.  CPU only."""
import gzip
import os
import random
import subprocess

import pytest

import brisk_amd
import oracle
from conftest import GOLDEN, ROOT


@pytest.fixture(scope="module")
def dumper(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("bin") / "fasta_dump")
    subprocess.check_call(["g++", "-std=gnu++17", "-O2", "-pthread", "-I" + os.path.join(os.path.dirname(brisk_amd.__file__), "include"),
                           os.path.join(ROOT, "tests", "cpp", "fasta_dump.cpp"), "-lz", "-o", exe])
    return exe


def run(dumper, path, batch, threads=0):
    out = subprocess.run([dumper, path, str(batch), str(threads)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    return [l for l in out.stdout.split("\n") if l]


def test_reference_fixtures_plain_and_gz(dumper, tmp_path):
    for name in ("test.fa", "debug_test.fa"):
        text = open(os.path.join(GOLDEN, name)).read()
        want = oracle.fasta_sequences(text)
        assert run(dumper, os.path.join(GOLDEN, name), 1 << 20) == want
        gz = str(tmp_path / (name + ".gz"))
        with gzip.open(gz, "wt") as f:
            f.write(text)
        for batch in (1 << 20, 1000, 1):  # tiny batches: a batch never ends inside a sequence
            assert run(dumper, gz, batch) == want


def test_ragged_input(dumper, tmp_path):
    rng = random.Random(4)
    recs = []
    for i in range(200):
        body = "".join(rng.choice("ACGTacgtNnRY-") if rng.random() < 0.1 else rng.choice("ACGT") for _ in range(rng.randint(0, 400)))
        lines = [body[j:j + 60] for j in range(0, len(body), 60)]
        if rng.random() < 0.2:
            lines.insert(rng.randint(0, len(lines)), "")
        recs.append(">r%d some description\n" % i + "\n".join(lines))
    text = "\n".join(recs)  # no trailing newline
    p = str(tmp_path / "ragged.fa")
    open(p, "w").write(text)
    want = oracle.fasta_sequences(text)
    assert want and run(dumper, p, 4096) == want
    # CRLF line ends: '\r' is not a base, so it cuts sequences exactly as it does in the reference's reader
    p2 = str(tmp_path / "crlf.fa")
    open(p2, "w", newline="").write(">x\r\nACGT\r\nACGT\r\n>y\r\nTTTT")
    assert run(dumper, p2, 100) == ["ACGT", "ACGT", "TTTT"]
    # the first line of a file is a header whatever it holds (getLineFasta, counter.cpp:173-178)
    p3 = str(tmp_path / "nohdr.fa")
    open(p3, "w").write("ACGTACGT\nGGGG\n>z\nCCCC\n")
    assert run(dumper, p3, 100) == ["GGGG", "CCCC"]


def test_plain_files_are_parsed_by_several_threads(dumper, tmp_path):
    """A plain file is mapped and each batch is cut into runs of whole records, one per thread: the sequences and
    their order do not depend on the number of threads or on the batch size (records of very different lengths,
    some longer than a batch, headers with '>' inside, empty records)."""
    rng = random.Random(12)
    recs = []
    for i in range(400):
        n = rng.choice([0, 5, 80, 300, 3000, 20000]) if rng.random() < 0.3 else rng.randint(1, 500)
        body = "".join(rng.choice("ACGTNacgt") if rng.random() < 0.02 else rng.choice("ACGT") for _ in range(n))
        lines = [body[j:j + 70] for j in range(0, len(body), 70)]
        recs.append(">r%d a>b|c\n" % i + "\n".join(lines))
    text = "\n".join(recs) + "\n"
    p = str(tmp_path / "many.fa")
    open(p, "w").write(text)
    want = oracle.fasta_sequences(text)
    assert want
    for batch in (200, 7000, 1 << 20):
        for threads in (1, 2, 5, 8):
            assert run(dumper, p, batch, threads) == want, (batch, threads)


def test_batches_of_one_line_records_end_between_records(dumper, tmp_path):
    """A file of reads has ONE sequence line per record: the running sequence is open at the end of every line, and a
    batch must still end in front of the next record's header once it is full -- not at the end of the file (a 5 M-read
    .fa.gz used to arrive as one batch).  gz (streamed) and plain (mapped) paths."""
    rng = random.Random(9)
    reads = ["".join(rng.choice("ACGT") for _ in range(150)) for _ in range(400)]
    text = "".join(">\n%s\n" % r for r in reads)
    gz = str(tmp_path / "reads.fa.gz")
    with gzip.open(gz, "wt") as f:
        f.write(text)
    plain = str(tmp_path / "reads.fa")
    open(plain, "w").write(text)
    for path in (gz, plain):
        out = subprocess.run([dumper, path, "3000", "0", "marks"], capture_output=True, text=True, timeout=120)
        assert out.returncode == 0, out.stderr
        lines = [l for l in out.stdout.split("\n") if l]
        sizes = [int(l.split()[1]) for l in lines if l.startswith("#batch")]
        assert [l for l in lines if not l.startswith("#")] == reads
        assert sum(sizes) == len(reads) and len(sizes) >= 15, sizes  # 60 kb in batches of >= 3 kb: 20 whole reads each
        assert max(sizes) <= 40, sizes


def test_clean_ranges_are_copied_in_place_and_dirty_ones_parsed(dumper, tmp_path):
    """Mapped files take two passes per batch: a thread's run of records is counted first and, if every sequence line holds
    [ACGTacgt] only, copied straight to its place in the batch (upper-cased, one sequence per record); a run with anything
    else goes through the general parser.  Clean multi-line records with lower case and empty records, no trailing newline;
    then the same file with one dirty stretch in the middle, so that some threads' runs are clean and others are not."""
    rng = random.Random(33)

    def rec(i, alphabet, n):
        body = "".join(rng.choice(alphabet) for _ in range(n))
        lines = [body[j:j + 61] for j in range(0, len(body), 61)]
        return ">c%d\n" % i + "\n".join(lines)

    clean = [rec(i, "ACGTacgt", rng.choice([0, 1, 31, 32, 33, 64, 150, 1000])) for i in range(600)]
    dirty = [rec(i, "ACGTNn-", rng.randint(1, 300)) for i in range(40)]
    for name, recs in (("clean.fa", clean), ("mixed.fa", clean[:300] + dirty + clean[300:])):
        text = "\n".join(recs)  # no trailing newline
        p = str(tmp_path / name)
        open(p, "w").write(text)
        want = oracle.fasta_sequences(text)
        assert want and all(s == s.upper() for s in want)
        for batch in (500, 20000, 1 << 20):
            for threads in (1, 3, 8):
                assert run(dumper, p, batch, threads) == want, (name, batch, threads)
