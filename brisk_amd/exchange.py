"""Bucket-range sharding across the GPUs of one node.

One process per GPU.  Every rank scans its own reads; super-k-mer records are
binned by the owner of their bucket range and exchanged with ONE all-to-all
(counts first, then the payload) over torch.distributed -- backend "nccl" is RCCL
over xGMI on ROCm, "gloo" on CPU for tests.  The scan's per-partition counts ride
inside the same payload (behind the last piece's records), so a batch that goes in
one piece is two collectives: the counts, the payload.  Ownership: equal partition
ranges, or ranges balanced on the partition histogram of a first scan
(balanced_cuts / ShardedCounter.balance).  The reference is single-process; nothing
here has a counterpart in it.
"""
from __future__ import annotations

from typing import List, Optional, Tuple

import numpy as np
import torch
import torch.distributed as dist


def owner_of_bucket(rid, b: int, part_bits: int, n_owners: int, ext_bits: int = 0):
    """Owner rank of a routing id (the bucket id followed by the layout's `ext_bits` extra minimizer-hash bits; the
    bucket id itself when ext_bits == 0): partitions are dealt in contiguous blocks, owner = partition * N >> part_bits
    with partition = rid >> (2b + ext_bits - part_bits).  Mirrors owner_of_record in csrc/brisk_partition.hip; works on
    ints and numpy arrays.  part_bits and ext_bits come from BriskHip.layout."""
    shift = 2 * b + ext_bits - part_bits
    if shift < 0:
        raise ValueError("part_bits exceeds the routing id's 2b + ext_bits bits")
    part = rid >> shift
    return (part * n_owners) >> part_bits


def uniform_cuts(part_bits: int, n_owners: int) -> List[int]:
    """first partition of every owner (+ the end) under equal ranges: the smallest p with p * N >> part_bits == o"""
    return [((o << part_bits) + n_owners - 1) // n_owners for o in range(n_owners)] + [1 << part_bits]


def balanced_cuts(hist: torch.Tensor, part_bits: int, n_owners: int, coarse_bits: int = 14) -> List[int]:
    """Cut points that give every owner the same share of the WORK in `hist` (a scan's partition histogram as export_hist
    delivers it: records in the low, k-mer instances in the high 32 bits of each word; the insert's time follows the
    instances).  Cuts fall on multiples of 2^(part_bits - coarse_bits) partitions: 2^coarse_bits block sums are all the
    ranks have to agree on (ShardedCounter.balance all-reduces exactly those)."""
    blocks = coarse_sums(hist, part_bits, coarse_bits)
    return cuts_from_coarse(blocks, part_bits, n_owners, coarse_bits)


def coarse_sums(hist: torch.Tensor, part_bits: int, coarse_bits: int = 14) -> torch.Tensor:
    cb = min(coarse_bits, part_bits)
    return (hist >> 32).reshape(1 << cb, -1).sum(dim=1)


def cuts_from_coarse(blocks: torch.Tensor, part_bits: int, n_owners: int, coarse_bits: int = 14) -> List[int]:
    cb = min(coarse_bits, part_bits)
    cum = torch.cumsum(blocks.to(torch.float64), 0).cpu().numpy()
    total = float(cum[-1]) if len(cum) else 0.0
    if total <= 0:
        return uniform_cuts(part_bits, n_owners)
    cuts = [0]
    for o in range(1, n_owners):
        # the first block boundary at which the cumulated work reaches o / N of the total (closest boundary)
        i = int(np.searchsorted(cum, total * o / n_owners))
        if i + 1 < len(cum) and i >= 0 and abs(cum[i] - total * o / n_owners) < abs((cum[i - 1] if i else 0.0) - total * o / n_owners):
            i += 1
        cuts.append(max(cuts[-1], min(i, len(cum)) << (part_bits - cb)))
    return cuts + [1 << part_bits]


def owner_of_partition(part, cuts):
    """owner of partition(s) `part` under cut points (first partition per owner + the end): ints or numpy arrays"""
    return np.searchsorted(np.asarray(cuts[1:-1], dtype=np.int64), part, side="right")


def pack_payload(out: torch.Tensor, counts, hist: torch.Tensor, lens, words: int, pay: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, List[int]]:
    """One send buffer for records AND histogram slices: to owner d its counts[d] records (of `words` int64 each, `out` holds
    them grouped by owner) followed by the lens[d] histogram words of d's partition range (`hist` is in partition order =
    owner order).  Returns (buffer, words per destination)."""
    send_words = [int(c) * words + int(l) for c, l in zip(counts, lens)]
    total = sum(send_words)
    if pay is None or pay.numel() < total:
        pay = torch.empty(total + (total >> 3) + 64, dtype=out.dtype, device=out.device)
    at = ro = ho = 0
    for c, l in zip(counts, lens):
        c, l = int(c), int(l)
        pay[at: at + c * words].copy_(out[ro * words: (ro + c) * words])
        pay[at + c * words: at + c * words + l].copy_(hist[ho: ho + l])
        at, ro, ho = at + c * words + l, ro + c, ho + l
    return pay, send_words


def unpack_payload(stage: torch.Tensor, recv_counts, my_len: int, words: int, inbox: torch.Tensor, inbox_offset: int, slices: torch.Tensor) -> int:
    """what pack_payload's buffers look like on arrival: from every source its records, then its slice of MY range.  Records go
    to inbox[inbox_offset ...] back to back, slices side by side into `slices`; returns the records unpacked."""
    at = n = 0
    for s_, c in enumerate(recv_counts):
        c = int(c)
        inbox[(inbox_offset + n) * words: (inbox_offset + n + c) * words].copy_(stage[at: at + c * words])
        slices[s_ * my_len: (s_ + 1) * my_len].copy_(stage[at + c * words: at + c * words + my_len])
        at += c * words + my_len
        n += c
    return n


def exchange_records(send: torch.Tensor, send_counts, words: int, group=None,
                     inbox: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, List[int]]:
    """send: int64 tensor of records grouped by destination rank (rank 0 first),
    send_counts[r] records of `words` int64 each for rank r.  Returns (inbox, recv_counts):
    the records this rank owns, grouped by source rank."""
    world = dist.get_world_size(group)
    if send.is_cuda and dist.get_backend(group) == "gloo":
        # rehearsal only (several ranks sharing one GPU, where RCCL refuses duplicate devices):
        # stage through host memory; the production path below never leaves the device
        n_out = int(sum(int(c) for c in send_counts))
        host_in, recv_counts = exchange_records(send[: n_out * words].cpu(), send_counts, words, group, None)
        n_in = sum(recv_counts)
        if inbox is None or inbox.numel() < n_in * words:
            inbox = torch.empty(max(n_in, 1) * words, dtype=torch.int64, device=send.device)
        inbox[: n_in * words].copy_(host_in[: n_in * words])
        return inbox, recv_counts
    sc = torch.as_tensor(np.asarray(send_counts, dtype=np.int64), device=send.device)
    assert sc.numel() == world
    rc = torch.empty_like(sc)
    dist.all_to_all_single(rc, sc, group=group)  # counts first
    recv_counts = [int(v) for v in rc.cpu().tolist()]
    n_in = sum(recv_counts)
    if inbox is None or inbox.numel() < n_in * words:
        inbox = torch.empty(max(n_in, 1) * words, dtype=torch.int64, device=send.device)
    view = inbox[: n_in * words]
    n_out = int(sum(int(c) for c in send_counts))
    dist.all_to_all_single(view, send[: n_out * words],
                           output_split_sizes=[c * words for c in recv_counts],
                           input_split_sizes=[int(c) * words for c in send_counts], group=group)  # then the payload
    return inbox, recv_counts


def exchange_counts(send_counts, device, group=None) -> List[int]:
    """the small all-to-all that goes first: how many records every rank is about to send me"""
    sc = torch.as_tensor(np.asarray(send_counts, dtype=np.int64), device=device)
    if sc.is_cuda and dist.get_backend(group) == "gloo":
        sc = sc.cpu()
    rc = torch.empty_like(sc)
    dist.all_to_all_single(rc, sc, group=group)
    return [int(v) for v in rc.cpu().tolist()]


def exchange_payload_async(send: torch.Tensor, send_counts, recv_counts, words: int, out: torch.Tensor, out_offset: int, group=None):
    """The payload all-to-all into out[out_offset*words : ...], not waited for: returns a work handle (None when the
    transfer is already complete, as on the gloo rehearsal path).  `send` and `out` must stay untouched until wait()."""
    n_in, n_out = int(sum(recv_counts)), int(sum(int(c) for c in send_counts))
    view = out[out_offset * words: (out_offset + n_in) * words]
    if send.is_cuda and dist.get_backend(group) == "gloo":  # rehearsal: through host memory, synchronously
        host = torch.empty(max(n_in, 1) * words, dtype=send.dtype)
        dist.all_to_all_single(host[: n_in * words], send[: n_out * words].cpu(), output_split_sizes=[c * words for c in recv_counts],
                               input_split_sizes=[int(c) * words for c in send_counts], group=group)
        view.copy_(host[: n_in * words])
        return None
    return dist.all_to_all_single(view, send[: n_out * words], output_split_sizes=[c * words for c in recv_counts],
                                  input_split_sizes=[int(c) * words for c in send_counts], group=group, async_op=True)


def return_sums(sums: torch.Tensor, recv_counts, send_counts, tags: torch.Tensor, n_reads: int, group=None) -> torch.Tensor:
    """Return trip of a sharded get.  `sums` (int64, one per record this rank answered, in the order the
    records arrived: grouped by source rank) goes back to the ranks the records came from; what comes
    back is in the order this rank sent its own records, so `tags` (read index per sent record) folds it
    into per-read sums."""
    back, _ = exchange_records(sums, recv_counts, 1, group, None)
    n_out = int(sum(int(c) for c in send_counts))
    per_read = torch.zeros(n_reads, dtype=torch.int64, device=sums.device)
    if n_out:
        per_read.index_add_(0, tags[:n_out].to(torch.int64), back[:n_out])
    return per_read


def agree_pieces(n_reads: int, device, group=None, four_from: int = 1 << 22) -> int:
    """The piece count of one ShardedCounter.count_packed call, the same on every rank: every piece issues the same
    all-to-alls on every rank, so the count follows the LARGEST share (one all-reduce(MAX)), not the local one.
    Only the last piece's all-to-all is exposed: four pieces for a large batch, one for a small one."""
    gloo = dist.get_backend(group) == "gloo"
    most = torch.tensor([int(n_reads)], dtype=torch.int64, device=torch.device("cpu") if gloo else device)
    dist.all_reduce(most, op=dist.ReduceOp.MAX, group=group)
    most = int(most.item())
    return 4 if most >= four_from else 1  # (a small batch in one piece: two collectives, nothing to overlap them with)


def suggest_part_bits(b: int, reads_per_batch: int, kmers_per_read: int = 0, min_bits: int = 24, per_partition: int = 512) -> int:
    """log2(#partitions) for a job that counts `reads_per_batch` reads per count_packed call, all ranks together
    (0: the library default, 2^24).  The insert wants 10-20 records per partition and call (include/brisk_hip.h,
    brisk_hip_options.part_bits): 50 M reads bring 209 M records, 12 per partition at 2^24.  N owners that each receive
    that much (N x 50 M reads per batch) keep the same density with N times the partitions -- each still walks only its
    own N-th of them; with 2^24 for 400 M reads the partitions hold 100 records each and the insert takes 2.5x as long
    (77 against 31 ms per owner and batch, tools/owner_emulation.py on one MI355X).  Explicit partition counts are bucket ranges:
    at most 2b bits.
    With `kmers_per_read` (read length - k + 1) the rule is stated in k-mer instances -- at most `per_partition` per partition
    and call (512: the insert's chunk is 256 after its record-level dedupe) -- and also goes BELOW 2^24 for a batch that would
    leave the default's partitions nearly empty: k31 m15 b14 brings 9 k-mers per record, and 20 M reads in 2^24 partitions
    are 143 instances = 12 entries each; the same job takes 33.0 ms with 2^24, 29.3 with 2^23 and 26.1 with 2^22 partitions
    (profiles/r03_part_bits_k31.txt: fewer histogram lines under the scan's atomics, which bound that scan, and fewer
    partitions for the insert to walk; 2^21 doubles the insert).  k63 does not follow: its scan is bound by instruction
    issue whatever the histogram's size and its insert loses 4.5 ms at 2^23, so 512 stays the default and bench.py passes
    1024 for k <= 32 (two-word records, one-word keys).  For 150 bp reads at k = 63 both rules give the same numbers.
    `min_bits` bounds the way down (default: never below 2^24): only 2^22 and 2^23 at k31 m15 b14 have specialised
    insert/get kernels and measurements behind them, so bench.py passes 22 for k <= 32."""
    import math
    if reads_per_batch <= 0 or 2 * b < 24:
        return 0
    if kmers_per_read > 0:
        bits = int(math.ceil(math.log2(max(reads_per_batch * kmers_per_read / float(per_partition), 1.0))))
        bits = max(min(min_bits, 24), min(bits, 2 * b))
        return 0 if bits == 24 else bits
    bits = min(int(round(math.log2(max(reads_per_batch, 3) / 3.0))), 2 * b)
    return bits if bits > 24 else 0


class ShardedCounter:
    """A rank's share of a k-mer counting job: owns the buckets of its partition range."""

    def __init__(self, k: int, m: int, b: int, rank: int, world: int, device: int, stream: torch.cuda.Stream,
                 part_bits: int = 0, group=None):
        import brisk_amd
        self.rank, self.world, self.group, self.stream = rank, world, group, stream
        self.dev = torch.device("cuda", device)
        self.ix = brisk_amd.BriskHip(k, m, b, device=device, stream=stream.cuda_stream, owner_rank=rank, n_owners=world,
                                     part_bits=part_bits)
        self.W = self.ix.record_words
        self._rec = self._out = self._inbox = self._hist = self._slices = self._pay = self._stage = None
        self._cap = 0

    # reads per rank from which a batch goes in four pieces (patchable: the tests lower it)
    PIECES4_MIN_READS = 1 << 22

    def count_packed(self, d_packed: torch.Tensor, d_starts: torch.Tensor, n_reads: int, pieces: Optional[int] = None) -> None:
        """Count this rank's reads into the sharded index.  The reads go in pieces (four for a large batch) so that the
        all-to-all of one piece's records runs while the next piece is scanned; the owner inserts everything it received
        at once.

        COLLECTIVE: every rank of the group must call this the same number of times.  Ranks may hold different
        numbers of reads (0 included: the last batch of a sharded FASTA): the piece count -- every piece issues the
        same two all-to-alls (counts, payload) on every rank -- is agreed on with one all-reduce(MAX) of n_reads, or given
        explicitly (the same value on every rank) as `pieces`; a rank with fewer reads than pieces runs empty pieces."""
        ix, W = self.ix, self.W
        if self.world == 1:
            ix.insert_packed(d_packed.data_ptr(), d_starts.data_ptr(), n_reads)
            return
        import brisk_amd
        if pieces is None:
            pieces = agree_pieces(n_reads, self.dev, self.group, self.PIECES4_MIN_READS)
        cuts = [n_reads * i // pieces for i in range(pieces + 1)]
        halves = list(zip(cuts[:-1], cuts[1:]))
        n_parts = 1 << ix.layout["part_bits"]
        with torch.cuda.stream(self.stream):
            # record capacity of one piece (scan output, routed copy); the inbox takes what all pieces bring, with room for skew
            cap = self._cap or (max(hi - lo for lo, hi in halves) * 6 + 4096)
            if self._rec is None or self._rec.numel() < cap * W or len(self._out) != len(halves):
                self._rec = torch.empty(cap * W, dtype=torch.int64, device=self.dev)
                self._out = [torch.empty(cap * W, dtype=torch.int64, device=self.dev) for _ in halves]
                self._inbox = torch.empty(len(halves) * (cap + cap // 4) * W, dtype=torch.int64, device=self.dev)
            # the scan's per-partition counts of ALL pieces, summed here: one slice per owner travels per batch
            if self._hist is None:
                self._hist = torch.empty(n_parts, dtype=torch.int64, device=self.dev)
            self._hist.zero_()
            lens = None
            works, n_in_total = [], 0
            for hi_, (lo, hi) in enumerate(halves):
                starts_ptr = d_starts.data_ptr() + lo * 8
                while True:
                    try:
                        n_rec = ix.scan_packed(d_packed.data_ptr(), starts_ptr, hi - lo, self._rec.data_ptr(), self._rec.numel() // W)
                        break
                    except brisk_amd.BriskHipError as e:
                        if e.code != brisk_amd.hipapi.ECAPACITY:
                            raise
                        for w in works:  # the buffers are about to be replaced: nothing may still be reading them
                            if w is not None:
                                w.wait()
                        works = []
                        self.stream.synchronize()
                        cap = max(cap * 2, ix.scan_bound(starts_ptr, hi - lo))
                        self._rec = torch.empty(cap * W, dtype=torch.int64, device=self.dev)
                        new_out = [torch.empty(cap * W, dtype=torch.int64, device=self.dev) for _ in halves]
                        for a_, b_ in zip(new_out, self._out):
                            a_[: b_.numel()].copy_(b_)
                        self._out = new_out
                out = self._out[hi_]
                counts = [int(c) for c in ix.route_records(self._rec.data_ptr(), n_rec, out.data_ptr())]
                # the scan counted its records per partition: each owner gets the slice of its range and adds the
                # slices up instead of counting the records it receives again (209 M random atomics per 50 M reads)
                lens = [int(v) for v in ix.export_hist_add(self._hist.data_ptr())]
                last = hi_ == len(halves) - 1
                my_len = lens[self.rank]
                recv_counts = exchange_counts(counts, self.dev, self.group)
                n_in = sum(recv_counts)
                if (n_in_total + n_in) * W > self._inbox.numel():  # skewed ownership: make room (what arrived is kept)
                    for w in works:
                        if w is not None:
                            w.wait()
                    works = []
                    self.stream.synchronize()
                    bigger = torch.empty((n_in_total + n_in) * W * 2, dtype=torch.int64, device=self.dev)
                    bigger[: n_in_total * W].copy_(self._inbox[: n_in_total * W])
                    self._inbox = bigger
                if not last:
                    works.append(exchange_payload_async(out, counts, recv_counts, W, self._inbox, n_in_total, self.group))
                    n_in_total += n_in
                    continue
                # The last piece's payload carries the histogram slices too -- to owner d: its records, then the slice of d's
                # partition range (summed over this batch's pieces) -- so that a batch is one counts exchange and ONE payload
                # exchange per piece, nothing else (the slices used to travel in a collective of their own).
                self._pay, send_words = pack_payload(out, counts, self._hist, lens, W, self._pay)
                recv_words = [c * W + my_len for c in recv_counts]
                if self._stage is None or self._stage.numel() < sum(recv_words):
                    self._stage = torch.empty(sum(recv_words) + (sum(recv_words) >> 3) + 64, dtype=torch.int64, device=self.dev)
                self.stream.synchronize()  # (the packed buffer is complete before the collective reads it)
                w_last = exchange_payload_async(self._pay, send_words, recv_words, 1, self._stage, 0, self.group)
                if self._slices is None or self._slices.numel() != self.world * my_len:
                    self._slices = torch.empty(self.world * my_len, dtype=torch.int64, device=self.dev)
                for w in works + [w_last]:
                    if w is not None:
                        w.wait()
                works = []
                n_in_total += unpack_payload(self._stage, recv_counts, my_len, W, self._inbox, n_in_total, self._slices)
            slices, n_slices = self._slices, self.world
            self._cap = cap
            self.stream.synchronize()
            ix.insert_records_hist(self._inbox.data_ptr(), n_in_total, slices.data_ptr(), n_slices)

    def balance(self, d_packed: torch.Tensor, d_starts: torch.Tensor, n_reads: int, threshold: float = 1.3) -> Optional[List[int]]:
        """Histogram-balanced ownership, from a scan of (a sample of) this rank's reads BEFORE anything is counted:
        every rank's partition histogram is reduced to 2^14 block sums of k-mer instances, the sums are all-reduced (128 KB), and
        if the most loaded owner under equal ranges carries more than `threshold` times the mean, every rank installs the same
        balanced cut points (brisk_hip_set_owner_cuts).  COLLECTIVE.  Returns the cut points installed, or None (equal ranges kept)."""
        ix, W = self.ix, self.W
        if self.world == 1:
            return None
        pb = ix.layout["part_bits"]
        with torch.cuda.stream(self.stream):
            cap = max(ix.scan_bound(d_starts.data_ptr(), n_reads), 1)
            rec = torch.empty(cap * W, dtype=torch.int64, device=self.dev)
            hist = torch.empty(1 << pb, dtype=torch.int64, device=self.dev)
            self.stream.synchronize()
            ix.scan_packed(d_packed.data_ptr(), d_starts.data_ptr(), n_reads, rec.data_ptr(), cap)
            ix.export_hist(hist.data_ptr())
            blocks = coarse_sums(hist, pb).to(torch.int64)
            del rec, hist
        gloo = dist.get_backend(self.group) == "gloo"
        red = blocks.cpu() if gloo else blocks
        dist.all_reduce(red, op=dist.ReduceOp.SUM, group=self.group)
        blocks = red.to(torch.float64).cpu()
        uni = uniform_cuts(pb, self.world)
        cb = min(14, pb)
        cum = torch.cat([torch.zeros(1, dtype=torch.float64), torch.cumsum(blocks, 0)])
        load = lambda cuts: [float(cum[cuts[o + 1] >> (pb - cb)] - cum[cuts[o] >> (pb - cb)]) for o in range(self.world)]
        total = float(cum[-1])
        self.owner_load = {"uniform_max_over_mean": (max(load(uni)) * self.world / total) if total else 1.0}
        if not total or self.owner_load["uniform_max_over_mean"] <= threshold:
            return None
        cuts = cuts_from_coarse(blocks, pb, self.world)
        self.owner_load["balanced_max_over_mean"] = max(load(cuts)) * self.world / total
        ix.set_owner_cuts(cuts)
        return cuts

    def stats(self) -> dict:
        """Brisk::stats of the whole sharded index: buckets, super-k-mers, entries and memory add up over the
        owners (bucket ranges are disjoint), the largest bucket is the maximum."""
        st = self.ix.stats()
        if self.world == 1:
            return st
        keys = ("nb_buckets", "nb_skmers", "nb_kmers", "memory_bytes")
        dev = self.dev if dist.get_backend(self.group) != "gloo" else torch.device("cpu")
        add = torch.tensor([int(st[k]) for k in keys], dtype=torch.int64, device=dev)
        big = torch.tensor([int(st["largest_bucket"])], dtype=torch.int64, device=dev)
        dist.all_reduce(add, op=dist.ReduceOp.SUM, group=self.group)
        dist.all_reduce(big, op=dist.ReduceOp.MAX, group=self.group)
        out = {k: int(v) for k, v in zip(keys, add.cpu().tolist())}
        out["largest_bucket"] = int(big.item())
        return out

    def get_packed(self, d_packed: torch.Tensor, d_starts: torch.Tensor, n_reads: int) -> torch.Tensor:
        """Per-read sum of the counts of the read's k-mers (query_sequence, apps/counter.cpp:281-310), with the
        buckets spread over the ranks: scan in query mode here, route the records (and the index of the
        read each came from) to the owners, the owners answer per record, the answers travel back."""
        ix, W = self.ix, self.W
        import brisk_amd
        with torch.cuda.stream(self.stream):
            cap = ix.scan_bound(d_starts.data_ptr(), n_reads)
            rec = torch.empty(max(cap, 1) * W, dtype=torch.int64, device=self.dev)
            tags = torch.empty(max(cap, 1), dtype=torch.int32, device=self.dev)
            self.stream.synchronize()
            n_rec = ix.scan_query(d_packed.data_ptr(), d_starts.data_ptr(), n_reads, rec.data_ptr(), tags.data_ptr(), cap)
            out = torch.empty(max(n_rec, 1) * W, dtype=torch.int64, device=self.dev)
            tags_out = torch.empty(max(n_rec, 1), dtype=torch.int32, device=self.dev)
            self.stream.synchronize()
            counts = ix.route_tagged(rec.data_ptr(), tags.data_ptr(), n_rec, out.data_ptr(), tags_out.data_ptr())
            if self.world == 1:
                inbox, recv_counts = out, [n_rec]
            else:
                inbox, recv_counts = exchange_records(out, counts, W, self.group, None)
            n_in = sum(recv_counts)
            sums = torch.zeros(max(n_in, 1), dtype=torch.int64, device=self.dev)
            self.stream.synchronize()
            ix.query_records(inbox.data_ptr(), n_in, sums.data_ptr())
            if self.world == 1:
                per_read = torch.zeros(n_reads, dtype=torch.int64, device=self.dev)
                if n_rec:
                    per_read.index_add_(0, tags_out[:n_rec].to(torch.int64), sums[:n_rec])
                return per_read
            return return_sums(sums[:max(n_in, 1)], recv_counts, counts, tags_out, n_reads, self.group)
