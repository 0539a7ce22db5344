"""CPU tests of the multi-GPU path: chunk sharding, frame ownership and the blend-time overlap exchange run with
world_size 2 and 3 over gloo (one process per rank) must reproduce the single-process result BIT FOR BIT.
The decoded chunks are stand-ins (seeded by chunk index) and the blend arithmetic is a torch restatement of
vv_decode_blend -- no HIP compute here; the real kernels are covered by the -m gpu tests."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as td
import torch.multiprocessing as mp

from videovanish_amd import pipeline as PL


def _cpu_blend(dec, w, acc):
    """restatement of vv_decode_blend (separately rounded fp32 products, chunk order)."""
    pix = (dec / 2.0 + 0.5).clamp(0, 1)
    ww = w[:, None, None, None]
    acc.copy_(acc * (1.0 - ww) + pix * ww)
    return acc


def _fake_decoded(ci, n, H, W):
    g = torch.Generator().manual_seed(1000 + ci)
    return torch.randn(n, H, W, 3, generator=g)


def _run_rank(rank, world, T, chunk, overlap, H, W):
    plan = PL.chunk_plan(T, chunk, overlap)
    wts = PL.blend_weights(plan)
    shards = PL.shard_chunks(len(plan), world)
    owner, chunk_rank = PL.frame_owner(plan, shards)
    pending = {ci: _fake_decoded(ci, plan[ci][1] - plan[ci][0], H, W) for ci in shards[rank]}
    acc, (lo, hi) = PL.exchange_and_blend(plan, wts, owner, chunk_rank, rank, world, pending, (H, W), torch.device("cpu"), blend_fn=_cpu_blend)
    return acc, lo, hi


def _worker(rank, world, port, T, chunk, overlap, H, W, q):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    td.init_process_group("gloo", rank=rank, world_size=world)
    try:
        acc, lo, hi = _run_rank(rank, world, T, chunk, overlap, H, W)
        # collection step of DiffuEraserHIP.forward: u8 frames to rank 0 (send/recv) and to every rank (broadcasts), no pickling
        plan = PL.chunk_plan(T, chunk, overlap)
        owner, _ = PL.frame_owner(plan, PL.shard_chunks(len(plan), world))
        ranges = PL.owned_ranges(owner, world)
        assert ranges[rank] == (lo, hi)
        u8 = None if acc is None else (acc.clamp(0, 1) * 255).round().to(torch.uint8)
        g0 = PL.gather_frames(u8, ranges, rank, world, T, (H, W, 3), torch.uint8, torch.device("cpu"), to="rank0")
        ga = PL.gather_frames(u8, ranges, rank, world, T, (H, W, 3), torch.uint8, torch.device("cpu"), to="all")
        assert (g0 is not None) == (rank == 0) and ga.shape == (T, H, W, 3)
        if rank == 0:
            assert torch.equal(g0, ga)
        q.put((rank, lo, hi, None if acc is None else acc.numpy(), ga.numpy()))
        td.barrier()
    finally:
        td.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("world,T", [(2, 60), (2, 104), (3, 60), (2, 33), (3, 40)])
def test_sharded_blend_equals_single_process(world, T):
    chunk, overlap, H, W = 32, 8, 6, 5
    ref, lo, hi = _run_rank(0, 1, T, chunk, overlap, H, W)
    assert (lo, hi) == (0, T)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, T, chunk, overlap, H, W, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    out = np.zeros_like(ref.numpy())
    seen = np.zeros(T, bool)
    for rank, lo, hi, acc, ga in res:
        if acc is not None:
            assert not seen[lo:hi].any()
            out[lo:hi] = acc
            seen[lo:hi] = True
    assert seen.all()
    assert np.array_equal(out, ref.numpy())          # bit for bit, independent of world size
    ref_u8 = (ref.clamp(0, 1) * 255).round().to(torch.uint8).numpy()
    for rank, lo, hi, acc, ga in res:
        assert np.array_equal(ga, ref_u8)            # every rank holds all frames after the "all" gather


def test_plan_and_ownership_properties():
    for T in (8, 32, 33, 60, 256, 1024):
        plan = PL.chunk_plan(T, 32, 8)
        for world in (1, 2, 4, 8):
            shards = PL.shard_chunks(len(plan), world)
            assert sorted(c for s in shards for c in s) == list(range(len(plan)))
            assert max(len(s) for s in shards) - min(len(s) for s in shards) <= 1
            owner, chunk_rank = PL.frame_owner(plan, shards)
            assert (owner >= 0).all() and (np.diff(owner) >= 0).all()      # contiguous, monotone ownership
            for r in range(world):                                          # owned frames lie inside the rank's own chunks
                idx = np.nonzero(owner == r)[0]
                if len(idx) and shards[r]:
                    assert plan[shards[r][0]][0] <= idx[0] and idx[-1] < plan[shards[r][-1]][1]
    assert [len(s) for s in PL.shard_chunks(11, 4)] == [3, 3, 3, 2]
    assert [len(s) for s in PL.shard_chunks(43, 8)] == [6, 6, 6, 5, 5, 5, 5, 5]


def test_host_plan_matches_oracle():
    from oracle import pipeline_ref as R
    for T in (5, 32, 60, 100, 256):
        assert PL.chunk_plan(T, 32, 8) == R.chunk_plan(T, 32, 8)
        for a, b in zip(PL.blend_weights(PL.chunk_plan(T, 32, 8)), R.blend_weights(R.chunk_plan(T, 32, 8))):
            assert np.array_equal(a, b)
    for (H0, W0, mx) in [(720, 1280, 960), (1080, 1920, 960), (37, 53, 32), (256, 256, 960)]:
        assert PL.model_size(H0, W0, mx) == R.model_size(H0, W0, mx)
    assert PL.ddim_timesteps(50) == R.M.ddim_timesteps(50) and PL.tcd_timesteps(2) == R.M.tcd_timesteps(2)
    assert torch.equal(PL.alphas_cumprod(), R.M.alphas_cumprod())
    assert torch.equal(PL.chunk_noise(42, 3, (2, 4, 5, 6)), R.chunk_noise(42, 3, (2, 4, 5, 6)))


def test_streaming_blend_holds_only_chunks_in_flight():
    """StreamingBlend (round 5): chunks are consumed in canonical order as soon as their predecessors exist and dropped -- fed in order the blender never
    holds more than ONE decoded chunk, fed in the worst order two lanes can produce (k + 1 before k) never more than two; the result is bit for bit
    the one-shot blend of the complete set."""
    T, chunk, overlap, H, W = 200, 32, 8, 4, 5
    plan = PL.chunk_plan(T, chunk, overlap)
    wts = PL.blend_weights(plan)
    owner, chunk_rank = PL.frame_owner(plan, PL.shard_chunks(len(plan), 1))
    dec = {ci: _fake_decoded(ci, plan[ci][1] - plan[ci][0], H, W) for ci in range(len(plan))}
    ref, _ = PL.exchange_and_blend(plan, wts, owner, chunk_rank, 0, 1, dict(dec), (H, W), torch.device("cpu"), blend_fn=_cpu_blend)
    for order, bound in ((list(range(len(plan))), 1), ([i ^ 1 if (i ^ 1) < len(plan) else i for i in range(len(plan))], 2)):
        sb = PL.StreamingBlend(plan, wts, owner, chunk_rank, 0, 1, list(range(len(plan))), (H, W), torch.device("cpu"), blend_fn=_cpu_blend)
        for ci in order:
            sb.add(ci, dec[ci])
        acc, (lo, hi) = sb.finish()
        assert (lo, hi) == (0, T) and sb.max_pending <= bound and not sb.pending
        assert torch.equal(acc, ref)
    with pytest.raises(AssertionError):
        sb = PL.StreamingBlend(plan, wts, owner, chunk_rank, 0, 1, list(range(len(plan))), (H, W), torch.device("cpu"), blend_fn=_cpu_blend)
        sb.add(1, dec[1])
        sb.finish()


def _bench_dry_run(world, T, H=16, W=24):
    """bench.py --dry-run --frames T launched as the driver launches the N-GPU lines (one process per rank, RANK / WORLD_SIZE / MASTER_* in the
    environment), over gloo with the CPU model stub: returns rank 0's JSON line."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, os.path.join(root, "bench.py"), "--gpus", str(world), "--dry-run", "--frames", str(T), "--height", str(H),
                                       "--width", str(W), "--steps", "1", "--warmup", "1"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=600) for p in procs]
    for p, (_, err) in zip(procs, outs):
        assert p.returncode == 0, err[-2000:]
    assert all(not any(ln.startswith("{") for ln in o.splitlines()) for o, _ in outs[1:])      # only rank 0 prints a result line (gloo prints its own banner)
    return json.loads([ln for ln in outs[0][0].splitlines() if ln.startswith("{")][-1])


@pytest.mark.parametrize("world,T,shards", [(4, 256, [3, 3, 3, 2]), (8, 1024, [6, 6, 6, 5, 5, 5, 5, 5])])
def test_bench_dry_run_multi_rank_equals_one_rank(world, T, shards):
    """The fixed-clip bench lines of BASELINE configs 3 / 4 (256 frames over 4 ranks, 1024 over 8) through bench.py ITSELF -- chunk plan, per-rank inputs,
    streaming blend + overlap exchange, gather to rank 0, max-over-ranks timing all-reduce, `ranks_seen` -- under gloo: the collected frames are bit for
    bit those of the one-rank run.  Unmeasured on hardware: no multi-GPU box in the build pool."""
    one = _bench_dry_run(1, T)
    many = _bench_dry_run(world, T)
    assert one["dry_run"] and many["dry_run"] and one["n_gpus"] == 1 and many["n_gpus"] == world
    assert many["ranks_seen"] == world and many["collective_backend"] == "gloo" and len(many["per_rank_seconds"]) == world
    assert f"sharded {shards}" in many["config"]["workload"] and many["config"]["chunks"] == sum(shards)
    assert many["output_sha256"] == one["output_sha256"]
    assert all("exchange_blend_s" in t and "upload_s" in t for t in many["per_rank_seconds"])
