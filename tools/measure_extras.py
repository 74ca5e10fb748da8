"""Side measurements quoted in DESIGN.md (run on the GPU box): the PCIe-inclusive rate of the host-ASCII
entry point, and the scan of one chromosome-length sequence.  Synthetic input made here with numpy."""
import sys, time
import numpy as np

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import torch
from brisk_amd import hipapi as B

k, m, b, L = 63, 21, 14, 150
rng = np.random.default_rng(1)


def ascii_reads(n_reads, G):
    genome = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, G, dtype=np.uint8)]
    starts = rng.integers(0, G - L + 1, n_reads)
    flat = np.empty(n_reads * L, dtype=np.uint8)
    step = 1 << 20
    ar = np.arange(L)
    for i in range(0, n_reads, step):
        s = starts[i:i + step]
        flat[i * L:(i + len(s)) * L] = genome[(s[:, None] + ar).ravel()]
    offs = (np.arange(n_reads + 1, dtype=np.uint64) * L)
    return flat, offs


def main():
    n_reads = 10_000_000
    flat, offs = ascii_reads(n_reads, n_reads * L // 15)
    with B.BriskHip(k, m, b) as ix:
        ix.insert_flat(flat, offs)  # warm-up: allocations, arena mapping
        ix.clear()
        ix.sync()
        t0 = time.perf_counter()
        ix.insert_flat(flat, offs)
        ix.sync()
        dt = time.perf_counter() - t0
        n = ix.stats()["nb_kmers"]
        print("host ASCII -> index (PCIe + pack + scan + insert): %d reads, %d entries in %.1f ms = %.2f G entries/s, %.1f GB/s of ASCII"
              % (n_reads, n, dt * 1e3, n / dt / 1e9, flat.nbytes / dt / 1e9))
    # one chromosome-length sequence, resident on the device
    G = 200_000_000
    genome = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, G, dtype=np.uint8)]
    d_bases = torch.from_numpy(genome).cuda()
    d_packed = torch.zeros((G + 15) // 16 + 4, dtype=torch.int32, device="cuda")
    d_starts = torch.tensor([0, G], dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    with B.BriskHip(k, m, b) as ix:
        ix.pack_ascii(d_bases.data_ptr(), G, d_packed.data_ptr())
        ix.sync()
        ix.insert_packed(d_packed.data_ptr(), d_starts.data_ptr(), 1)  # warm-up
        ix.sync()
        ix.clear()
        ix.profile_enable(True)
        ix.profile_reset()
        t0 = time.perf_counter()
        ix.insert_packed(d_packed.data_ptr(), d_starts.data_ptr(), 1)
        ix.sync()
        dt = time.perf_counter() - t0
        prof = ix.profile_read()
        n = ix.stats()["nb_kmers"]
        print("one %d Mbp sequence (chunked scan): %d entries in %.1f ms = %.2f G k-mers/s; kernels %s"
              % (G // 1_000_000, n, dt * 1e3, (G - k + 1) / dt / 1e9, {a: round(v["ms"], 2) for a, v in prof.items() if v["launches"]}))
        # the same sequence queried back: query-mode scan (chunked too), then one sum per record
        cap = ix.scan_bound(d_starts.data_ptr(), 1) // 8 + 4096
        d_rec = torch.empty(cap * ix.record_words, dtype=torch.int64, device="cuda")
        d_tags = torch.empty(cap, dtype=torch.int32, device="cuda")
        d_sums = torch.empty(cap, dtype=torch.int64, device="cuda")
        torch.cuda.synchronize()
        for _ in range(2):
            t0 = time.perf_counter()
            n_rec = ix.scan_query(d_packed.data_ptr(), d_starts.data_ptr(), 1, d_rec.data_ptr(), d_tags.data_ptr(), cap)
            ix.query_records(d_rec.data_ptr(), n_rec, d_sums.data_ptr())
            total = int(d_sums[:n_rec].sum().item())
            dt = time.perf_counter() - t0
        print("queried back: %d records, sum of counts %d (k-mers %d) in %.1f ms" % (n_rec, total, G - k + 1, dt * 1e3))


if __name__ == "__main__":
    main()
