"""Worker of test_kernel_variants_match_the_oracle: one process per environment (the library reads BRISK_BINS,
BRISK_INSERT_GENERIC, BRISK_QUERY_GENERIC once), the device paths against the oracle.  Prints "ok <n checks>"."""
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch

import brisk_amd
import oracle
from test_gpu_parity import SPECIAL, _random_reads

oracle.build(ref=False)
O = oracle.Oracle()
rng = random.Random(61)
reads = _random_reads(rng, 1500, 300) + SPECIAL + ["A" * 150] * 40 + ["ACGT" * 40] * 3
queries = reads[:300] + SPECIAL + _random_reads(rng, 200, 300)
checks = 0
for k, m, b in ((63, 21, 14), (31, 15, 14), (31, 11, 11), (47, 15, 10)):
    want = O.count(reads, k, m, b)
    qf, qo = oracle.pack_reads(queries)
    h = O.index_new(k, m, b)
    flat, offs = oracle.pack_reads(reads)
    O.index_insert_reads(h, flat, offs)
    want_q = O.index_query_reads(h, qf, qo)
    O.index_free(h)
    with brisk_amd.BriskHip(k, m, b) as ix:
        def to_device(fl, of):
            d_bases = torch.from_numpy(fl).cuda()
            d_packed = torch.zeros((len(fl) + 15) // 16 + 4, dtype=torch.int32, device="cuda")
            d_starts = torch.from_numpy(of.astype(np.int64)).cuda()
            torch.cuda.synchronize()
            ix.pack_ascii(d_bases.data_ptr(), len(fl), d_packed.data_ptr())
            ix.sync()
            return d_packed, d_starts
        d_packed, d_starts = to_device(flat, offs)
        half = len(reads) // 2
        ix.insert_packed(d_packed.data_ptr(), d_starts.data_ptr(), half)  # two batches: the second meets existing entries
        ix.insert_packed(d_packed.data_ptr(), d_starts[half:].contiguous().data_ptr(), len(reads) - half)
        ix.sync()
        st = ix.stats()
        got = (sorted(oracle.multiset_lines(*ix.enumerate(), k)), st["nb_kmers"], st["nb_buckets"])
        assert got == want, (k, m, b, "index", os.environ.get("BRISK_BINS"))
        q_packed, q_starts = to_device(qf, qo)
        sums = torch.full((len(queries),), -1, dtype=torch.int64, device="cuda")
        torch.cuda.synchronize()
        ix.get_packed(q_packed.data_ptr(), q_starts.data_ptr(), len(queries), sums.data_ptr())
        assert np.array_equal(sums.cpu().numpy().astype(np.uint64), want_q), (k, m, b, "get_packed")
        assert np.array_equal(ix.get_reads(queries), want_q), (k, m, b, "get_reads")
        checks += 3
        if (k, m, b) in ((63, 21, 14), (31, 11, 11)):  # re-bucketing: records that carry multiplicities (tests/test_reallocate.py)
            from test_reallocate import _expected, _lines_to_counter
            before = oracle.multiset_lines(*ix.enumerate(), k)
            with brisk_amd.BriskHip(k, m + 2, b + 2) as new:
                ix.reallocate_into(new)
                after = oracle.multiset_lines(*new.enumerate(), k)
            assert _lines_to_counter(after) == _expected(O, before, k, m + 2), (k, m, b, "reallocate")
print("ok", checks)
