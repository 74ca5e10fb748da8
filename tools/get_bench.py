import sys, time, torch
sys.path.insert(0, '/root/repo')
import brisk_amd
k, m, b = (int(x) for x in sys.argv[1:4]) if len(sys.argv) > 3 else (63, 21, 14)
n_reads = int(sys.argv[4]) if len(sys.argv) > 4 else 50_000_000
L = 150
G = n_reads * L // 15
dev = torch.device("cuda", 0)
d_packed = torch.zeros((n_reads * L + 15) // 16 + 4, dtype=torch.int32, device=dev)
d_starts = torch.zeros(n_reads + 1, dtype=torch.int64, device=dev)
torch.cuda.synchronize()  # torch fills on its own stream, the library works on another: the fill must have landed
sums = torch.zeros(n_reads, dtype=torch.int64, device=dev)
ix = brisk_amd.BriskHip(k, m, b)
ix.synth_reads(G, 0, n_reads, L, d_packed.data_ptr(), d_starts.data_ptr())
ix.sync()
ix.insert_packed(d_packed.data_ptr(), d_starts.data_ptr(), n_reads)
ix.sync()
print(ix.stats())
for rep in range(3):
    ix.profile_reset(); ix.profile_enable(True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ix.get_packed(d_packed.data_ptr(), d_starts.data_ptr(), n_reads, sums.data_ptr())
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("get_packed %.1f ms  (%.2f G k-mers/s)  sum %d" % (dt * 1e3, n_reads * (L - k + 1) / dt / 1e9, int(sums.sum())), {n: round(v["ms"], 2) for n, v in ix.profile_read().items() if v["launches"]})
