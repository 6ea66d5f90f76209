"""World-size-2 gloo test of the multi-GPU plumbing: contiguous image shards, one config
broadcast at start, one gather of per-image results at the end; no collective in between."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n_images, out_q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from materialist_amd import synthetic
        from materialist_amd.dist import broadcast_config, broadcast_tensor, gather_results, shard_range

        cfg = broadcast_config({"spp": 16, "size": 16, "n_images": n_images} if rank == 0 else None)
        assert cfg == {"spp": 16, "size": 16, "n_images": n_images}
        t = broadcast_tensor(torch.arange(5, dtype=torch.float32) if rank == 0 else torch.zeros(5))
        assert t.tolist() == [0, 1, 2, 3, 4]
        lo, hi = shard_range(cfg["n_images"], world, rank)
        # per-image "result" that depends only on the image id (a stand-in for the per-image loss of an optimisation)
        rows = []
        for i in range(lo, hi):
            sc = synthetic.make_scene(i, cfg["size"], cfg["size"])
            rows.append([float(i), float(sc.albedo.mean()), float(sc.light[1, 0])])
        local = torch.tensor(rows, dtype=torch.float64).reshape(-1, 3)
        got = gather_results(local, dst=0)
        if rank == 0:
            out_q.put(torch.cat(got).numpy())
        else:
            assert got == []
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_images", [5, 8])
def test_two_rank_shard_and_gather(n_images):
    from materialist_amd import synthetic

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_images, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    # N ranks give exactly what one rank computes for the same images, in image order
    want = []
    for i in range(n_images):
        sc = synthetic.make_scene(i, 16, 16)
        want.append([float(i), float(sc.albedo.mean()), float(sc.light[1, 0])])
    np.testing.assert_array_equal(res, np.array(want))


def _batch_worker(rank, world, port, n_images, out_q):
    os.environ.update({"MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "RANK": str(rank), "WORLD_SIZE": str(world), "LOCAL_RANK": str(rank)})
    from materialist_amd import batch

    r, w, _ = batch.init_distributed("gloo")
    assert (r, w) == (rank, world)
    seen = []

    def process(i, path, cfg):
        seen.append(i)
        return [cfg["scale"] * i, float(len(path))]

    rows = batch.run_batch([f"img_{k}.png" for k in range(n_images)] if rank == 0 else [], {"scale": 0.5} if rank == 0 else {}, process)
    out_q.put((rank, seen, rows))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_images", [1, 5])
def test_run_batch_shards_and_gathers(n_images):
    """materialist_amd.batch: config comes from rank 0 only, each image is processed exactly once by its owner, rank 0 gets
    every row in image order (also when a rank's shard is empty)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_batch_worker, args=(r, 2, port, n_images, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted([q.get(timeout=120) for _ in range(2)], key=lambda x: x[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    (r0, seen0, rows0), (r1, seen1, rows1) = got
    assert sorted(seen0 + seen1) == list(range(n_images)) and not (set(seen0) & set(seen1))
    assert rows1 == []
    assert [r["image_id"] for r in rows0] == list(range(n_images))
    assert all(r["values"] == [0.5 * r["image_id"], float(len(r["path"]))] for r in rows0)
    assert all(r["rank"] == (0 if r["image_id"] in seen0 else 1) for r in rows0)


def _shard_worker(rank, world, port, n_images, fail_rank, out_q):
    os.environ.update({"MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "RANK": str(rank), "WORLD_SIZE": str(world), "LOCAL_RANK": str(rank)})
    from materialist_amd import batch

    batch.init_distributed("gloo")
    calls = []

    def process_shard(ids, paths, cfg):
        calls.append(list(ids))
        if rank == fail_rank:
            raise RuntimeError("boom")
        return [[cfg["scale"] * i, float(len(p))] for i, p in zip(ids, paths)]

    rows = batch.run_batch([f"synthetic:{k}" for k in range(n_images)] if rank == 0 else [], {"scale": 2.0} if rank == 0 else {}, None,
                           process_shard=process_shard)
    out_q.put((rank, calls, rows))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("fail_rank", [-1, 1])
def test_run_batch_hands_a_rank_its_whole_shard_and_survives_a_failing_rank(fail_rank):
    """BASELINE configs[2]: a rank's contiguous shard is processed as ONE batch (process_shard called once per rank); an
    exception on one rank yields NaN rows + an error message for its images while every rank still reaches the collectives."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_shard_worker, args=(r, 2, port, 5, fail_rank, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted([q.get(timeout=120) for _ in range(2)], key=lambda x: x[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    (_, calls0, rows0), (_, calls1, rows1) = got
    assert calls0 == [[0, 1, 2]] and calls1 == [[3, 4]] and rows1 == []
    assert [r["image_id"] for r in rows0] == [0, 1, 2, 3, 4]
    for r in rows0:
        if fail_rank == 1 and r["image_id"] >= 3:
            assert r["error"] == "RuntimeError: boom" and all(np.isnan(v) for v in r["values"])
        else:
            assert r["error"] is None and r["values"] == [2.0 * r["image_id"], float(len(r["path"]))]


def _weights_worker(rank, world, port, path, out_q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from materialist_amd.dist import broadcast_state_dict

        # only rank 0 reads the file (the other ranks are handed a path that does not exist)
        sd = torch.load(path, map_location="cpu", weights_only=True) if rank == 0 else None
        got = broadcast_state_dict(sd, torch.device("cpu"))
        out_q.put((rank, {k: (tuple(v.shape), str(v.dtype), float(v.double().sum())) for k, v in got.items()}))
    finally:
        dist.destroy_process_group()


def test_network_weights_leave_rank_0_once_as_a_flat_buffer(tmp_path):
    """rank 0 loads MaterialNet / PosMLP weights and broadcasts ONE flat buffer; every rank ends with the same state_dict."""
    torch.manual_seed(3)
    # every dtype arrives exactly: an integer buffer beyond 2^24 (BatchNorm's num_batches_tracked), a float64 entry with more digits than
    # fp32 holds, a half-precision tensor and a bool mask travel in buffers of their own type
    sd = {"pretrained.blocks.0.attn.qkv.weight": torch.randn(12, 4), "pretrained.blocks.0.attn.qkv.bias": torch.randn(12),
          "depth_head.scratch.output_conv2.0.weight": torch.randn(2, 3, 3, 3), "empty": torch.zeros(0),
          "bn.num_batches_tracked": torch.tensor(2 ** 24 + 1, dtype=torch.int64), "table": torch.tensor([2 ** 40 + 3, -7], dtype=torch.int64),
          "stats.mean64": torch.tensor([1.0 + 2.0 ** -40, 3.141592653589793], dtype=torch.float64),
          "half.w": torch.randn(5, 3).half(), "keep": torch.tensor([True, False, True])}
    path = str(tmp_path / "w.pth")
    torch.save(sd, path)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_weights_worker, args=(r, 2, port, path, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    want = {k: (tuple(v.shape), str(v.dtype), float(v.double().sum())) for k, v in sd.items()}
    assert res[0] == want and res[1] == want


def test_ranks_of_a_node_take_disjoint_core_sets():
    """`dist.pin_rank_cores` gives the LOCAL_RANK-th slice of the cores a process may run on; the slices of a node's ranks
    are disjoint and cover no core twice (each rank's iterations are enqueued by one Python thread)."""
    import json
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    n_all = len(os.sched_getaffinity(0))
    if n_all < 2:
        pytest.skip("one core")
    seen = []
    for lr in range(2):
        env = dict(os.environ, LOCAL_WORLD_SIZE="2", LOCAL_RANK=str(lr), WORLD_SIZE="2", PYTHONPATH=root)
        out = subprocess.run([sys.executable, "-c", "import os, json; from materialist_amd.dist import pin_rank_cores; r = pin_rank_cores(); "
                              "print(json.dumps([r, sorted(os.sched_getaffinity(0))]))"], env=env, capture_output=True, text=True, timeout=120)
        assert out.returncode == 0, out.stderr[-1000:]
        info, cores = json.loads(out.stdout.strip().splitlines()[-1])
        assert info["cores_of_this_rank"] == len(cores) == n_all // 2
        seen.append(set(cores))
    assert not (seen[0] & seen[1])
