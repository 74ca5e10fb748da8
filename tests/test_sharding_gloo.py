"""world_size-2 gloo test of the N>1 path (runs on CPU): the owner arithmetic, the
record exchange (counts, then payload, one all-to-all) and the record format.  The
per-rank device work is replaced here by the oracle's record generator -- the same
records the scan kernel is checked against in tests/test_gpu_parity.py."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _records_for(O, h, reads, k, m, b):
    W = int(O.lib.bo_record_words(k, m, b)) + 1
    rows = []
    for s in reads:
        c, bucket, n, idx0 = O.records(h, s, k, m, b)
        for i in range(len(n)):
            hdr = int(bucket[i]) | (int(n[i]) << 32) | (int(idx0[i]) << 40)
            rows.append([int(x) for x in c[i]] + [hdr])
    return np.array(rows, dtype=np.uint64).reshape(-1, W), W


def _expand(rec, W, k, b):
    """records -> set-like list of (bucket, compacted k-mer, idx') with multiplicity"""
    out = []
    kb = k - b
    for r in rec:
        big = sum(int(w) << (64 * i) for i, w in enumerate(r[: W - 1]))
        hdr = int(r[W - 1])
        bucket, n, idx0 = hdr & 0xffffffff, (hdr >> 32) & 0xff, (hdr >> 40) & 0xff
        for j in range(n):
            out.append((bucket, (big >> (2 * (n - 1 - j))) & ((1 << (2 * kb)) - 1), idx0 + j))
    return out


def _records_with_tags(O, h, reads, k, m, b):
    W = int(O.lib.bo_record_words(k, m, b)) + 1
    rows, tags = [], []
    for t, s in enumerate(reads):
        c, bucket, n, idx0 = O.records(h, s, k, m, b)
        for i in range(len(n)):
            hdr = int(bucket[i]) | (int(n[i]) << 32) | (int(idx0[i]) << 40)
            rows.append([int(x) for x in c[i]] + [hdr])
            tags.append(t)
    return np.array(rows, dtype=np.uint64).reshape(-1, W), np.array(tags, dtype=np.int64), W


def _get_worker(rank, world, port, k, m, b, part_bits, q):
    """sharded get on CPU: the owners' device work (insert, per-record sums) is done with python dicts
    over the oracle's records; what is under test is the route, the two all-to-alls and return_sums"""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle
    from collections import Counter
    from brisk_amd.exchange import exchange_records, owner_of_bucket, return_sums
    O = oracle.Oracle()
    reads = [bytes(r) for r in O.synth_reads(5000, 0, 400)]
    mine = reads[rank::world]
    h = O.index_new(k, m, b)
    rec, tags, W = _records_with_tags(O, h, mine, k, m, b)
    O.index_free(h)
    owner = owner_of_bucket((rec[:, W - 1] & np.uint64(0xffffffff)).astype(np.int64), b, part_bits, world)
    order = np.argsort(owner, kind="stable")
    counts = np.bincount(owner, minlength=world)
    send = torch.from_numpy(rec[order].astype(np.int64).reshape(-1))
    tags_out = torch.from_numpy(tags[order])
    # count: the owner's share of the index
    inbox, recv_counts = exchange_records(send, counts, W)
    got = inbox[: sum(recv_counts) * W].numpy().view(np.uint64).reshape(-1, W)
    mine_counts = Counter(_expand(got, W, k, b))
    # get: the same records travel again, the owner answers one sum per record
    inbox, recv_counts = exchange_records(send, counts, W)
    got = inbox[: sum(recv_counts) * W].numpy().view(np.uint64).reshape(-1, W)
    sums = torch.tensor([sum(mine_counts[key] % 256 for key in _expand(got[i:i + 1], W, k, b)) for i in range(len(got))], dtype=torch.int64)
    per_read = return_sums(sums, recv_counts, [int(c) for c in counts], tags_out, len(mine))
    q.put((rank, [int(v) for v in per_read.tolist()]))
    dist.barrier()
    dist.destroy_process_group()


def _worker(rank, world, port, k, m, b, part_bits, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle
    from brisk_amd.exchange import exchange_records, owner_of_bucket
    O = oracle.Oracle()
    reads = [bytes(r) for r in O.synth_reads(6000, 0, 600)]
    mine = reads[rank::world]
    h = O.index_new(k, m, b)
    rec, W = _records_for(O, h, mine, k, m, b)
    O.index_free(h)
    owner = owner_of_bucket((rec[:, W - 1] & np.uint64(0xffffffff)).astype(np.int64), b, part_bits, world)
    order = np.argsort(owner, kind="stable")
    counts = np.bincount(owner, minlength=world)
    send = torch.from_numpy(rec[order].astype(np.int64).reshape(-1))
    inbox, recv_counts = exchange_records(send, counts, W)
    got = inbox[: sum(recv_counts) * W].numpy().view(np.uint64).reshape(-1, W)
    got_owner = owner_of_bucket((got[:, W - 1] & np.uint64(0xffffffff)).astype(np.int64), b, part_bits, world)
    q.put((rank, len(rec), [int(c) for c in counts], recv_counts, bool((got_owner == rank).all()), _expand(got, W, k, b)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("k,m,b,part_bits", [(63, 21, 14, 24), (31, 11, 4, 8)])
def test_two_rank_exchange_matches_oracle(O, k, m, b, part_bits):
    import oracle
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, world, port, k, m, b, part_bits, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # conservation: what rank s sent to rank r is what r received from s
    for r in range(world):
        assert res[r][3] == [res[s][2][r] for s in range(world)]
        assert res[r][4], "a rank received a record it does not own"
    assert sum(sum(x[3]) for x in res) == sum(x[1] for x in res)
    # the union of the shards' k-mers, counted, equals the oracle's whole-job multiset
    from collections import Counter
    shards = [Counter(x[5]) for x in res]
    assert not (set(shards[0]) & set(shards[1])), "a k-mer landed on two owners"
    total = shards[0] + shards[1]
    reads = [bytes(r) for r in O.synth_reads(6000, 0, 600)]
    lines, nk, nb = O.count(reads, k, m, b)
    assert len(total) == nk
    assert sorted(c % 256 for c in total.values()) == sorted(int(l.split()[2]) for l in lines)
    assert len({key[0] for key in total}) == nb


@pytest.mark.parametrize("k,m,b,part_bits", [(63, 21, 14, 24), (31, 11, 4, 8)])
def test_two_rank_get_return_trip_matches_oracle(O, k, m, b, part_bits):
    """get across bucket-range shards: records out, per-record sums back, folded per read"""
    import oracle
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_get_worker, args=(r, world, port, k, m, b, part_bits, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=180) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    reads = [bytes(r) for r in O.synth_reads(5000, 0, 400)]
    flat, offs = oracle.pack_reads(reads)
    h = O.index_new(k, m, b)
    O.index_insert_reads(h, flat, offs)
    want = [int(v) for v in O.index_query_reads(h, flat, offs)]
    O.index_free(h)
    for r in range(world):
        assert res[r] == want[r::world]


def _pieces_worker(rank, world, port, shares, four_from, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from brisk_amd.exchange import agree_pieces
    got = [agree_pieces(n[rank], torch.device("cpu"), None, four_from) for n in shares]
    q.put((rank, got))
    dist.destroy_process_group()


def test_piece_count_is_collective_with_unequal_shares():
    """the number of pieces (and with it the number of all-to-alls) of a sharded batch must not depend on a
    rank's own share: ranks on different sides of the thresholds, or with no reads at all, agree on one count."""
    world = 2
    shares = [(0, 0), (1, 0), (0, 5), (3, 99), (100, 7), (100, 0), (99, 99)]  # (rank 0, rank 1) reads per call
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 33500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_pieces_worker, args=(r, world, port, shares, 100, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0] == res[1] == [1, 1, 1, 1, 4, 4, 1]  # four pieces from the threshold up, else one (two collectives)


def _balanced_worker(rank, world, port, k, m, b, part_bits, q):
    """SURVEY.md 8(e) with histogram-balanced cut points and ONE payload collective: every rank builds its records and their
    partition histogram (here: from the oracle's records, as the scan's export_hist delivers it), the ranks agree on cut points
    from the all-reduced block sums, route by them, and exchange counts + one payload that carries records AND histogram slices"""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle
    from brisk_amd import exchange as X
    O = oracle.Oracle()
    reads = [bytes(r) for r in O.synth_reads(6000, 0, 600)]
    mine = reads[rank::world]
    h = O.index_new(k, m, b)
    rec, W = _records_for(O, h, mine, k, m, b)
    O.index_free(h)
    shift = 2 * b - part_bits
    hdr = rec[:, W - 1]
    part = ((hdr & np.uint64(0xffffffff)) >> np.uint64(shift)).astype(np.int64)
    n_k = ((hdr >> np.uint64(32)) & np.uint64(0xff)).astype(np.int64)
    hist = np.zeros(1 << part_bits, dtype=np.int64)
    np.add.at(hist, part, 1 + (n_k << 32))
    hist_t = torch.from_numpy(hist)
    blocks = X.coarse_sums(hist_t, part_bits).to(torch.int64)
    dist.all_reduce(blocks)  # 2^14 block sums: all the ranks have to agree on
    cuts = X.cuts_from_coarse(blocks, part_bits, world)
    uni = X.uniform_cuts(part_bits, world)
    owner = X.owner_of_partition(part, cuts)
    order = np.argsort(owner, kind="stable")
    counts = np.bincount(owner, minlength=world)
    lens = [cuts[o + 1] - cuts[o] for o in range(world)]
    out = torch.from_numpy(rec[order].astype(np.int64).reshape(-1))
    recv_counts = X.exchange_counts(counts, torch.device("cpu"))           # collective 1: the counts
    pay, send_words = X.pack_payload(out, counts, hist_t, lens, W)
    my_len = lens[rank]
    recv_words = [c * W + my_len for c in recv_counts]
    stage = torch.empty(max(sum(recv_words), 1), dtype=torch.int64)
    wk = X.exchange_payload_async(pay, send_words, recv_words, 1, stage, 0)  # collective 2: records + histogram slices
    if wk is not None:
        wk.wait()
    inbox = torch.empty(max(sum(recv_counts), 1) * W, dtype=torch.int64)
    slices = torch.empty(world * my_len, dtype=torch.int64)
    n_in = X.unpack_payload(stage, recv_counts, my_len, W, inbox, 0, slices)
    got = inbox[: n_in * W].numpy().view(np.uint64).reshape(-1, W)
    got_part = ((got[:, W - 1] & np.uint64(0xffffffff)) >> np.uint64(shift)).astype(np.int64)
    # the slices, added up, are the histogram of what arrived (what brisk_hip_insert_records_hist relies on)
    want_hist = np.zeros(my_len, dtype=np.int64)
    np.add.at(want_hist, got_part - cuts[rank], 1 + (((got[:, W - 1] >> np.uint64(32)) & np.uint64(0xff)).astype(np.int64) << 32))
    slices_ok = bool((slices.reshape(world, my_len).sum(dim=0).numpy() == want_hist).all())
    mine_ok = bool(((got_part >= cuts[rank]) & (got_part < cuts[rank + 1])).all())
    inst = (blocks.to(torch.float64))
    q.put((rank, cuts, uni, n_in, int(len(rec)), slices_ok, mine_ok, _expand(got, W, k, b)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("k,m,b,part_bits", [(31, 15, 14, 24), (63, 21, 14, 24)])
def test_two_rank_balanced_cuts_and_single_payload(O, k, m, b, part_bits):
    from collections import Counter
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 35500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_balanced_worker, args=(r, world, port, k, m, b, part_bits, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0][1] == res[1][1], "the ranks must install the same cut points"
    cuts = res[0][1]
    assert cuts[0] == 0 and cuts[-1] == 1 << part_bits and cuts == sorted(cuts)
    assert all(x[5] and x[6] for x in res), "slices must add up to the received records' histogram, records must lie in the owner's range"
    assert sum(x[3] for x in res) == sum(x[4] for x in res)
    # balanced: the owners' shares of the k-mer instances are within a few percent of each other (equal ranges need not be:
    # at k31 m15 a partition is the TOP bits of a hash whose minima the minimizers are)
    loads = [sum(1 for _ in x[7]) for x in res]
    assert max(loads) / (sum(loads) / world) < 1.06, loads
    shards = [Counter(x[7]) for x in res]
    assert not (set(shards[0]) & set(shards[1]))
    total = shards[0] + shards[1]
    reads = [bytes(r) for r in O.synth_reads(6000, 0, 600)]
    lines, nk, nb = O.count(reads, k, m, b)
    assert len(total) == nk and sorted(c % 256 for c in total.values()) == sorted(int(l.split()[2]) for l in lines)


def test_cut_points_from_a_skewed_histogram():
    from brisk_amd import exchange as X
    pb = 16
    w = torch.arange(1 << pb, dtype=torch.float64)
    inst = ((1 << pb) - w).to(torch.int64)          # load falls linearly with the partition index: the first half holds 3/4 of it
    hist = (inst << 32) | 1
    for n in (2, 4, 8):
        cuts = X.balanced_cuts(hist, pb, n)
        loads = [int(inst[cuts[o]:cuts[o + 1]].sum()) for o in range(n)]
        assert cuts[0] == 0 and cuts[-1] == 1 << pb and max(loads) / (sum(loads) / n) < 1.01
        uni = X.uniform_cuts(pb, n)
        assert max(int(inst[uni[o]:uni[o + 1]].sum()) for o in range(n)) / (sum(loads) / n) > 1.4
    assert X.owner_of_partition(np.array([0, 9, 10, 99]), [0, 10, 50, 100]).tolist() == [0, 0, 1, 2]
    assert X.uniform_cuts(4, 3) == [0, 6, 11, 16]  # the smallest p with p * 3 >> 4 == o


def test_owner_of_bucket_takes_the_routing_id():
    from brisk_amd.exchange import owner_of_bucket
    # b = 4 with the default layout: 8 bucket bits + 16 ext bits = 24 routing bits = part_bits; owner = top of the routing id
    rid = np.array([0, (1 << 23) - 1, 1 << 23, (1 << 24) - 1], dtype=np.int64)
    assert owner_of_bucket(rid, 4, 24, 2, ext_bits=16).tolist() == [0, 0, 1, 1]
    assert owner_of_bucket(rid >> 16, 4, 8, 2).tolist() == [0, 0, 1, 1]  # explicit part_bits: plain bucket ranges
    with pytest.raises(ValueError):
        owner_of_bucket(rid, 4, 24, 2)


def test_partition_count_follows_the_batch():
    """brisk_hip_options.part_bits for sharded jobs: ~3 reads per partition and batch, never below the default, at most 2b"""
    from brisk_amd.exchange import suggest_part_bits
    assert suggest_part_bits(14, 50_000_000) == 0      # one GPU's batch: the library default (2^24)
    assert suggest_part_bits(14, 100_000_000) == 25
    assert suggest_part_bits(14, 200_000_000) == 26
    assert suggest_part_bits(14, 400_000_000) == 27    # BASELINE config #4
    assert suggest_part_bits(14, 10**10) == 28         # bucket ranges: at most 2b bits
    assert suggest_part_bits(11, 400_000_000) == 0     # 2b = 22 < 24: nothing to scale
    assert suggest_part_bits(14, 0) == 0
    # stated in k-mer instances (<= 512 per partition and call): the same answers at k63 / 150 bp ...
    for reads, want in ((50_000_000, 0), (100_000_000, 25), (200_000_000, 26), (400_000_000, 27), (10**10, 28)):
        assert suggest_part_bits(14, reads, 88) == want
    # ... and fewer partitions than the default only where the caller allows it (k31 m15 b14, 20 M reads: 2^23)
    assert suggest_part_bits(14, 20_000_000, 120) == 0
    assert suggest_part_bits(14, 20_000_000, 120, min_bits=22) == 23
    assert suggest_part_bits(14, 20_000_000, 120, min_bits=22, per_partition=1024) == 22   # what bench.py asks for at k <= 32
    assert suggest_part_bits(14, 50_000_000, 120, min_bits=22, per_partition=1024) == 23
    assert suggest_part_bits(14, 1000, 120, min_bits=22) == 22
    assert suggest_part_bits(11, 20_000_000, 120, min_bits=22) == 0
