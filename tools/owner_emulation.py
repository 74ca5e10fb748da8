# what rank 0 of N does per step of the weak-scaling job (N x 50 M reads): its own scan + route, then the insert of what N peers send it
import sys, time, torch
sys.path.insert(0, '/root/repo')
import brisk_amd
N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
PB = int(sys.argv[2]) if len(sys.argv) > 2 else 0
PIECES = int(sys.argv[3]) if len(sys.argv) > 3 else 1
n_reads, L, k, m, b = 50_000_000, 150, 63, 21, 14
G = N * n_reads * L // 15
dev = torch.device("cuda", 0)
d_packed = torch.zeros((n_reads * L + 15) // 16 + 4, dtype=torch.int32, device=dev)
d_starts = torch.zeros(n_reads + 1, dtype=torch.int64, device=dev)
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
    ix.synth_reads(G, peer * n_reads, n_reads, L, d_packed.data_ptr(), d_starts.data_ptr())
    ix.sync()
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
torch.cuda.synchronize()
print("records for owner 0:", n_in, "slice len", ml)
for rep in range(2):
    ix.clear(); ix.profile_reset(); ix.profile_enable(True)
    t0 = time.perf_counter()
    ix.insert_records_hist(inbox.data_ptr(), n_in, slices.data_ptr(), N)
    ix.sync()
    print("insert_records_hist %.1f ms" % ((time.perf_counter() - t0) * 1e3), {n: round(v["ms"], 2) for n, v in ix.profile_read().items() if v["launches"]})
st = ix.stats()
print(st)
