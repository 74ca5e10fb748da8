"""The multi-GPU counting job driven from C++ (brisk_amd/apps/brisk_shard.cpp; north_star: "host code stays C++", "a single
RCCL all-to-all over xGMI"): scan -> route by owner -> counts, records and histogram slices exchanged -> insert on the owner.
On the one-GPU box: the RCCL transport with a world of one (communicator, grouped ncclSend/ncclRecv to self), and the
rehearsal of worlds of two and three through the "files" transport (every rank on device 0; RCCL refuses two ranks on one
device).  Shard digests add up to the digest of one index over the same reads.  No 8-GPU run is made here."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _single_index(total, k, m, b):
    import torch
    import brisk_amd
    L = 150
    G = max(int(total * L / 15.0), L + 1)
    d_packed = torch.zeros((total * L + 15) // 16 + 4, dtype=torch.int32, device="cuda")
    d_starts = torch.zeros(total + 1, dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    with brisk_amd.BriskHip(k, m, b) as ix:
        ix.synth_reads(G, 0, total, L, d_packed.data_ptr(), d_starts.data_ptr())
        ix.insert_packed(d_packed.data_ptr(), d_starts.data_ptr(), total)
        return list(ix.checksum())


def _run_world(exe, world, transport, total, k, m, b, tmp):
    d = str(tmp / ("x%d%s%d" % (world, transport, k)))
    os.makedirs(d)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([exe, str(r), str(world), d, transport, str(total), str(k), str(m), str(b)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env)
             for r in range(world)]
    ent = sumc = dig = 0
    for p in procs:
        out, err = p.communicate(timeout=600)
        assert p.returncode == 0, err[-2000:]
        w = out.split()
        ent += int(w[w.index("entries") + 1])
        sumc += int(w[w.index("sum_counts") + 1])
        dig = (dig + int(w[w.index("digest") + 1])) % (1 << 64)
    return [ent, sumc, dig]


@pytest.mark.gpu
def test_cpp_sharded_count_matches_one_index(tmp_path):
    import brisk_amd
    exe = os.path.join(ROOT, "brisk_amd", "apps", "brisk_shard")
    if not os.path.exists(exe):
        brisk_amd.build_apps()
    k, m, b, total = 63, 21, 14, 400_001
    want = _single_index(total, k, m, b)
    assert want[1] == total * (150 - k + 1)
    assert _run_world(exe, 1, "rccl", total, k, m, b, tmp_path) == want     # RCCL communicator of one rank
    assert _run_world(exe, 2, "files", total, k, m, b, tmp_path) == want    # two owners, one GPU
    assert _run_world(exe, 3, "files", total, k, m, b, tmp_path) == want    # three owners, shares of unequal size
    k, m, b, total = 31, 11, 11, 100_000                                    # config #2': routing ids == bucket ids, big partitions
    assert _run_world(exe, 2, "files", total, k, m, b, tmp_path) == _single_index(total, k, m, b)
    k, m, b, total = 31, 15, 14, 300_000   # the reference's defaults: equal ranges are lopsided there, brisk_shard installs balanced cut points
    assert _run_world(exe, 3, "files", total, k, m, b, tmp_path) == _single_index(total, k, m, b)
