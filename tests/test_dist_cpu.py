"""world_size-2 CPU test (gloo) of the multi-GPU start-up path: rank 0 reads the weights file, preps and packs;
the packed bytes travel by a torch.distributed broadcast; rank 1 imports them into a network parsed from the cfg
alone and must end up with byte-identical packed state and identical per-channel integers.  Plus the image sharding."""
import hashlib
import os
import sys

import numpy as np
import pytest

ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def _worker(rank, world, port, cfg, wts, out_dir):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from yolo_quantization_amd import binding
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    if rank == 0:
        net = binding.Net(cfg, wts)
        net.prepare_host_only(1.0 / 255.0, 0)
        packed = net.export_packed()
        size = torch.tensor([packed.size], dtype=torch.int64)
    else:
        net = binding.Net(cfg, None)
        size = torch.zeros(1, dtype=torch.int64)
    dist.broadcast(size, 0)
    blob = torch.from_numpy(packed.copy()) if rank == 0 else torch.empty(int(size.item()), dtype=torch.uint8)
    dist.broadcast(blob, 0)
    if rank != 0:
        net.import_packed_host(blob.numpy())
    again = net.export_packed()
    digest = hashlib.sha256(again.tobytes()).hexdigest()
    zps = [net.prep(i)["zp_act"] for i in range(net.n)]
    with open(os.path.join(out_dir, f"rank{rank}.txt"), "w") as f:
        f.write(digest + "\n" + ",".join(map(str, zps)) + "\n")
    dist.barrier()
    dist.destroy_process_group()
    net.close()


def test_packed_weights_broadcast_world2(cfg_dir, tmp_path):
    import torch.multiprocessing as mp
    from yolo_quantization_amd import synth
    cfg = os.path.join(cfg_dir, "tiny_unit.cfg")
    wts = str(tmp_path / "w.weights")
    synth.synth_weights(cfg, wts, seed=1)
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port, cfg, wts, str(tmp_path)), nprocs=2, join=True)
    r0 = open(tmp_path / "rank0.txt").read()
    r1 = open(tmp_path / "rank1.txt").read()
    assert r0 == r1 and len(r0.split("\n")[0]) == 64


def test_shard_ranges_cover_and_are_disjoint():
    from yolo_quantization_amd.sharding import shard_range
    for total in (512, 64, 7, 1, 0):
        for world in (1, 2, 4, 8):
            seen = []
            for r in range(world):
                a, b = shard_range(total, r, world)
                assert 0 <= a <= b <= total
                seen.extend(range(a, b))
            assert seen == list(range(total))
            sizes = [shard_range(total, r, world)[1] - shard_range(total, r, world)[0] for r in range(world)]
            assert max(sizes) - min(sizes) <= 1


def test_requant_identities():
    """The two algebraic rewrites the device epilogue relies on (csrc/common.h), checked in float64 numpy:
    (1) trunc(trunc(fl(a*M)) * 2^-s) == trunc(fl(a * (M*2^-s)))   (single folded multiply)
    (2) round(q*0.1) == -((|q|+5)//10) for negative q, incl. the 24-bit multiply form for |q|+5 < 2^16."""
    rng = np.random.default_rng(0)
    n = 2_000_000
    a = rng.integers(-2 ** 31, 2 ** 31, n, dtype=np.int64)
    a[: n // 4] = rng.integers(-2 ** 24, 2 ** 24, n // 4)
    m0 = rng.integers(2 ** 30, 2 ** 31, n, dtype=np.int64)
    s = rng.integers(0, 32, n)
    M = m0.astype(np.float64) * 2.0 ** -31
    S = np.exp2(-s.astype(np.float64))
    ref = np.trunc(np.trunc(a.astype(np.float64) * M) * S)
    fast = np.trunc(a.astype(np.float64) * (M * S))
    assert np.array_equal(ref, fast)
    # adversarial: products next to integer multiples of 2^s
    k = rng.integers(-2 ** 20, 2 ** 20, 200_000); s2 = rng.integers(1, 20, 200_000)
    M2 = rng.integers(2 ** 30, 2 ** 31, 200_000).astype(np.float64) * 2.0 ** -31
    a2 = np.round(k * np.exp2(s2.astype(float)) / M2).astype(np.int64)
    for da in (-1, 0, 1):
        aa = np.clip(a2 + da, -2 ** 31 + 1, 2 ** 31 - 1).astype(np.float64)
        assert np.array_equal(np.trunc(np.trunc(aa * M2) * np.exp2(-s2.astype(float))),
                              np.trunc(aa * (M2 * np.exp2(-s2.astype(float)))))
    q = np.concatenate([-np.arange(1, 200_000, dtype=np.int64), -rng.integers(1, 2 ** 31, 500_000),
                        np.array([-2 ** 31, -2 ** 31 + 1, -5, -15, -25, -4, -6])])
    want = np.round(q.astype(np.float64) * 0.1)
    want = np.where(np.abs(q.astype(np.float64) * 0.1 - np.trunc(q.astype(np.float64) * 0.1)) == 0.5,
                    np.trunc(q.astype(np.float64) * 0.1) - 1, want)  # C round(): half away from zero
    got = -((np.abs(q) + 5) // 10)
    assert np.array_equal(got.astype(np.float64), want)
    x = np.arange(0, 65536, dtype=np.uint64)
    assert np.array_equal((x * 0xCCCD) >> 19, x // 10)


def test_bench_self_launch_dry_dist_world2(cfg_dir):
    """`python bench.py --gpus 2 --dry-dist` with no launcher around it (what the round driver runs for N > 1, plus the dry switch):
    bench.py re-runs itself under torch.distributed.run with two ranks and drives ITS OWN rendezvous, packed-weights broadcast,
    import, image sharding and max-over-ranks timing code on gloo with host-only prep.  One JSON line, rc 0, identical packed
    state on both ranks."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-dist", "--steps", "3", "--cfg",
                        os.path.join(cfg_dir, "tiny_unit.cfg")], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["dry_dist"] and d["n_gpus"] == 2 and d["packed_state_identical_on_all_ranks"] and d["shards_cover_global_batch"]
    assert d["image_shards"] == [[0, 64], [64, 128]]


def test_bench_self_launch_dry_dist_world8(cfg_dir):
    """the same with eight ranks -- BASELINE config[3]'s shape (512 images sharded 64 per GPU over 8 GPUs) on gloo: one JSON line, the eight
    shards are 8 x 64 covering 0..512, every rank ends up with rank 0's packed bytes"""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--dry-dist", "--steps", "3", "--cfg",
                        os.path.join(cfg_dir, "tiny_unit.cfg")], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["dry_dist"] and d["n_gpus"] == 8 and d["packed_state_identical_on_all_ranks"] and d["shards_cover_global_batch"]
    assert d["image_shards"] == [[64 * k, 64 * k + 64] for k in range(8)]


def test_bench_rejects_a_world_that_is_not_gpus():
    """a launcher that started another number of ranks than --gpus says is an error (rc 2), not a silent re-launch"""
    import subprocess
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-dist"], capture_output=True, text=True, timeout=120, env=env)
    assert r.returncode == 2
