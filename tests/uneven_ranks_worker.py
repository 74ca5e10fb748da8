"""Worker of test_sharded_count_with_unequal_and_empty_shares (run under torch.distributed.run, 2 ranks, gloo, one GPU)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torch.distributed as dist

import brisk_amd
from brisk_amd import exchange
from brisk_amd.exchange import ShardedCounter

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo")
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
k, m, b, L = 63, 21, 14, 150
# two batches; shares per rank: batch 0 = (900, 30), batch 1 = (4000, 0); four pieces from 2000 reads on
shares = [(900, 30), (4000, 0)]
ShardedCounter.PIECES4_MIN_READS = 2000
total = sum(sum(sh) for sh in shares)
G = total * L // 15
stream = torch.cuda.Stream(device=dev)
sc = ShardedCounter(k, m, b, rank, world, 0, stream)
seen_pieces = []
orig = exchange.agree_pieces


def spy(*a, **kw):
    p = orig(*a, **kw)
    seen_pieces.append(p)
    return p


exchange.agree_pieces = spy
first = 0
for sh in shares:
    mine_first = first + (sh[0] if rank == 1 else 0)
    n = sh[rank]
    with torch.cuda.stream(stream):
        d_packed = torch.zeros((max(n, 1) * L + 15) // 16 + 4, dtype=torch.int32, device=dev)
        d_starts = torch.zeros(max(n, 1) + 1, dtype=torch.int64, device=dev)
    stream.synchronize()
    if n:
        sc.ix.synth_reads(G, mine_first, n, L, d_packed.data_ptr(), d_starts.data_ptr())
        sc.ix.sync()
    sc.count_packed(d_packed, d_starts, n)
    sc.ix.sync()
    first += sum(sh)
ent, sumc, dig = sc.ix.checksum()
t = torch.tensor([ent, sumc, dig & ((1 << 31) - 1), (dig >> 31) & ((1 << 31) - 1), dig >> 62], dtype=torch.int64)
dist.all_reduce(t)
if rank == 0:
    sharded = [int(t[0]), int(t[1]), (int(t[2]) + (int(t[3]) << 31) + (int(t[4]) << 62)) % (1 << 64)]
    with brisk_amd.BriskHip(k, m, b) as one:
        d_packed = torch.zeros((total * L + 15) // 16 + 4, dtype=torch.int32, device=dev)
        d_starts = torch.zeros(total + 1, dtype=torch.int64, device=dev)
        torch.cuda.synchronize()
        one.synth_reads(G, 0, total, L, d_packed.data_ptr(), d_starts.data_ptr())
        one.sync()
        one.insert_packed(d_packed.data_ptr(), d_starts.data_ptr(), total)
        e1, s1, d1 = one.checksum()
    print(json.dumps({"pieces": seen_pieces, "sharded": sharded, "single": [e1, s1, d1 % (1 << 64)]}))
sc.ix.close()
dist.destroy_process_group()
