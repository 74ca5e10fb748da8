"""Multi-GPU readiness on ONE MI355X (no 8-GPU node is ours to use): what one rank of N does, measured piece by piece.

  python tools/owner_emulation.py weak   [N] [part_bits]          rank 0 of N, N x 50 M reads per batch (BASELINE config #4 for N = 8):
                                                                  its own scan + route, then the insert of what N peers send it
  python tools/owner_emulation.py strong [N] [total_reads] [k m b]  rank 0 of N on ONE job of total_reads (default 50 M, k63 m21 b14): the critical
                                                                  path of the strong-scaling target -- scan of its N-th of the reads, route by
                                                                  owner, histogram export, [exchange: bytes only], insert of the N-th of the
                                                                  job's records it owns -- with every call's wall time (host synchronisations
                                                                  included) next to the kernels' own HIP-event times
  python tools/owner_emulation.py balance [total_reads] [k m b]   max / mean owner load for N in {2, 4, 8} from the N = 1 partition histogram,
                                                                  with uniform partition ranges and with histogram-balanced cut points
                                                                  (brisk_amd.exchange.balanced_cuts)
The exchange itself cannot be measured here; its payload is printed (bytes out per rank) with the time 7 xGMI links of ~153 GB/s
would need for it, a lower bound."""
import sys, time
import torch
sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
import brisk_amd
from brisk_amd import exchange

mode = sys.argv[1] if len(sys.argv) > 1 else "weak"
args = sys.argv[2:]
L = 150
dev = torch.device("cuda", 0)


def synth(ix, G, first, n):
    d_packed = torch.zeros((n * L + 15) // 16 + 4, dtype=torch.int32, device=dev)
    d_starts = torch.zeros(n + 1, dtype=torch.int64, device=dev)
    torch.cuda.synchronize()  # (torch's fill runs on torch's stream, the library on its own: the fill must not land on top of the reads)
    ix.synth_reads(G, first, n, L, d_packed.data_ptr(), d_starts.data_ptr())
    ix.sync()
    return d_packed, d_starts


class Timer:
    def __init__(self):
        self.rows = []

    def __call__(self, name, fn):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r = fn()
        torch.cuda.synchronize()
        self.rows.append((name, (time.perf_counter() - t0) * 1e3))
        return r


if mode == "balance":
    total = int(args[0]) if args else 50_000_000
    k, m, b = (int(a) for a in args[1:4]) if len(args) >= 4 else (63, 21, 14)
    G = total * L // 15
    ix = brisk_amd.BriskHip(k, m, b)
    W, pb = ix.record_words, ix.layout["part_bits"]
    d_packed, d_starts = synth(ix, G, 0, total)
    cap = ix.scan_bound(d_starts.data_ptr(), total)
    cap = min(cap, total * (16 if k < 40 else 6) + 4096)
    rec = torch.empty(cap * W, dtype=torch.int64, device=dev)
    hist = torch.empty(1 << pb, dtype=torch.int64, device=dev)
    n_rec = ix.scan_packed(d_packed.data_ptr(), d_starts.data_ptr(), total, rec.data_ptr(), cap)
    ix.export_hist(hist.data_ptr())
    recs = (hist & 0xffffffff).to(torch.float64)
    inst = (hist >> 32).to(torch.float64)
    print(f"k{k} m{m} b{b}, {total} reads: {n_rec} records in 2^{pb} partitions; instances {int(inst.sum().item())}")
    for N in (2, 4, 8):
        for what, cuts in (("uniform ranges", exchange.uniform_cuts(pb, N)), ("balanced cuts ", exchange.balanced_cuts(hist, pb, N))):
            loads_r = [float(recs[cuts[o]:cuts[o + 1]].sum().item()) for o in range(N)]
            loads_i = [float(inst[cuts[o]:cuts[o + 1]].sum().item()) for o in range(N)]
            print(f"  N={N} {what}: records max/mean {max(loads_r) / (sum(loads_r) / N):.3f}   k-mer instances max/mean {max(loads_i) / (sum(loads_i) / N):.3f}"
                  f"   first partitions {[int(c) for c in cuts[:-1]]}")
    sys.exit(0)

if mode == "strong":
    N = int(args[0]) if args else 8
    total = int(args[1]) if len(args) > 1 else 50_000_000
    k, m, b = (int(a) for a in args[2:5]) if len(args) >= 5 else (63, 21, 14)
    G = total * L // 15
    pb = exchange.suggest_part_bits(b, total)
    ix = brisk_amd.BriskHip(k, m, b, owner_rank=0, n_owners=N, part_bits=pb)
    print("layout", ix.layout)
    W, n_parts = ix.record_words, 1 << ix.layout["part_bits"]
    share = total // N
    cap = share * 6 + 4096
    rec = torch.empty(cap * W, dtype=torch.int64, device=dev)
    out = torch.empty(cap * W, dtype=torch.int64, device=dev)
    hist = torch.empty(n_parts, dtype=torch.int64, device=dev)
    inbox = torch.empty((cap + cap // 2) * W, dtype=torch.int64, device=dev)
    slices, n_in, sent = None, 0, 0
    mine_timing = None
    for peer in range(N):  # every peer's share is scanned here, in turn; what each routes to owner 0 is collected as rank 0's inbox
        d_packed, d_starts = synth(ix, G, peer * share, share)
        for rep in range(2 if peer == 0 else 1):  # (rank 0's own share twice: the second time with warm buffers, as in a running job)
            T = Timer()
            ix.profile_reset(); ix.profile_enable(True)
            n_rec = T("scan_packed", lambda: ix.scan_packed(d_packed.data_ptr(), d_starts.data_ptr(), share, rec.data_ptr(), cap))
            counts = T("route_records", lambda: ix.route_records(rec.data_ptr(), n_rec, out.data_ptr()))
            lens = T("export_hist", lambda: ix.export_hist(hist.data_ptr()))
            if peer == 0:
                mine_timing = (T.rows, {n: round(v["ms"], 3) for n, v in ix.profile_read().items() if v["launches"]})
                sent = int(sum(int(c) for c in counts[1:]))
        mine, ml = int(counts[0]), int(lens[0])
        if slices is None:
            slices = torch.empty(N * ml, dtype=torch.int64, device=dev)
        inbox[n_in * W:(n_in + mine) * W].copy_(out[: mine * W])
        slices[peer * ml:(peer + 1) * ml].copy_(hist[:ml])
        n_in += mine
        print(f"  peer {peer}: {n_rec} records, {mine} of them for owner 0 (inbox now {n_in})", flush=True)
        del d_packed, d_starts
    rows, kern = mine_timing
    print(f"rank 0 of {N}, strong scaling, {total} reads in all ({share} scanned here):")
    for name, ms in rows:
        print(f"  {name:24s} {ms:8.3f} ms wall")
    print("   kernels of these calls (HIP events):", kern)
    out_bytes = sent * W * 8 + (N - 1) * ml * 8
    print(f"  exchange (not measured)   {out_bytes / 1e6:8.1f} MB out of this rank: {sent} records to {N - 1} peers + their histogram slices; >= {out_bytes / (7 * 153e9) * 1e3:.3f} ms at 7 x 153 GB/s")
    best = None
    for rep in range(3):
        ix.clear(); ix.profile_reset(); ix.profile_enable(True)
        T = Timer()
        T("insert_records_hist", lambda: ix.insert_records_hist(inbox.data_ptr(), n_in, slices.data_ptr(), N))
        best = (T.rows[0][1], {n: round(v["ms"], 3) for n, v in ix.profile_read().items() if v["launches"]})
    print(f"  {'insert_records_hist':24s} {best[0]:8.3f} ms wall  ({n_in} records from {N} peers)")
    print("   kernels of this call (HIP events):", best[1])
    crit = sum(ms for _, ms in rows) + best[0]
    print(f"  critical path without the exchange: {crit:.3f} ms  (one GPU, whole job: see bench.py; ideal N-th of it: {52.4 / N:.2f} ms at k63)")
    print(ix.stats())
    sys.exit(0)

# ---- weak: what rank 0 of N does per step of the weak-scaling job (N x 50 M reads)
N = int(args[0]) if args else 8
PB = int(args[1]) if len(args) > 1 else 0
n_reads, k, m, b = 50_000_000, 63, 21, 14
G = N * n_reads * L // 15
ix = brisk_amd.BriskHip(k, m, b, owner_rank=0, n_owners=N, part_bits=PB)
print("layout", ix.layout)
W = ix.record_words
n_parts = 1 << ix.layout["part_bits"]
cap = n_reads * 6
rec = torch.empty(cap * W, dtype=torch.int64, device=dev)
out = torch.empty(cap * W, dtype=torch.int64, device=dev)
hist = torch.empty(n_parts, dtype=torch.int64, device=dev)
inbox = torch.empty((cap + cap // 4) * W, dtype=torch.int64, device=dev)
slices = None
torch.cuda.synchronize()
n_in = 0
for peer in range(N):
    d_packed, d_starts = synth(ix, G, peer * n_reads, n_reads)
    ix.profile_reset(); ix.profile_enable(True)
    t0 = time.perf_counter()
    n_rec = ix.scan_packed(d_packed.data_ptr(), d_starts.data_ptr(), n_reads, rec.data_ptr(), cap)
    counts = ix.route_records(rec.data_ptr(), n_rec, out.data_ptr())
    lens = ix.export_hist(hist.data_ptr())
    ix.sync()
    t1 = time.perf_counter()
    if peer == 0:
        print("scan+route+export %.1f ms" % ((t1 - t0) * 1e3), {n: round(v["ms"], 2) for n, v in ix.profile_read().items() if v["launches"]})
    mine, ml = int(counts[0]), int(lens[0])
    if slices is None:
        slices = torch.empty(N * ml, dtype=torch.int64, device=dev)
    inbox[n_in * W:(n_in + mine) * W].copy_(out[: mine * W])
    slices[peer * ml:(peer + 1) * ml].copy_(hist[:ml])
    n_in += mine
    del d_packed, d_starts
torch.cuda.synchronize()
print("records for owner 0:", n_in, "slice len", ml)
for rep in range(2):
    ix.clear(); ix.profile_reset(); ix.profile_enable(True)
    t0 = time.perf_counter()
    ix.insert_records_hist(inbox.data_ptr(), n_in, slices.data_ptr(), N)
    ix.sync()
    print("insert_records_hist %.1f ms" % ((time.perf_counter() - t0) * 1e3), {n: round(v["ms"], 2) for n, v in ix.profile_read().items() if v["launches"]})
print(ix.stats())
