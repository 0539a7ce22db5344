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


# ---- round 6: the exchange under RENDEZVOUS semantics (RCCL point-to-point), and bench.py without a launcher ----------------------------------
class _RendezvousFabric:
    """An in-process stand-in for RCCL's point-to-point semantics, which gloo does not have: a send COMPLETES only once the matching receive has
    been posted (gloo buffers eagerly, so an early send costs nothing there and the gloo tests cannot see it).  Ranks are threads; every rank keeps a
    logical clock (chunks computed so far); for every transfer the fabric records the clock of each side when its half was posted and the wall time
    between the first and the second half."""

    def __init__(self, world):
        import threading
        self.cv = threading.Condition()
        self.clock = [0] * world
        self.open = {}           # (src, dst) -> list of half-posted transfers in posting order
        self.log = []            # (src, dst, clock at send post, clock at recv post, seconds the first half waited for the second)

    def comm(self, rank):
        return _RendezvousComm(self, rank)

    def tick(self, rank):
        with self.cv:
            self.clock[rank] += 1

    def _post(self, kind, rank, tensor, peer):
        import time
        key = (rank, peer) if kind == "send" else (peer, rank)
        with self.cv:
            q = self.open.setdefault(key, [])
            other = "recv" if kind == "send" else "send"
            half = next((h for h in q if other in h and kind not in h), None)
            if half is None:
                half = {}
                q.append(half)
            half[kind] = (tensor, self.clock[rank], time.time())
            if "send" in half and "recv" in half:
                half["recv"][0].copy_(half["send"][0])
                half["done"] = True
                q.remove(half)
                self.log.append((key[0], key[1], half["send"][1], half["recv"][1], abs(half["send"][2] - half["recv"][2])))
                self.cv.notify_all()
        return half

    def _wait(self, half):
        with self.cv:
            assert self.cv.wait_for(lambda: half.get("done"), timeout=60), "rendezvous never completed: a send or a receive has no partner"


class _RendezvousComm:
    def __init__(self, fabric, rank):
        self.f, self.rank = fabric, rank

    def post(self, ops):
        import types
        halves = [self.f._post(kind, self.rank, t, peer) for kind, t, peer in ops]
        return [types.SimpleNamespace(wait=lambda h=h: self.f._wait(h)) for h in halves]


@pytest.mark.parametrize("world,T", [(4, 256), (8, 1024), (3, 40), (2, 33)])
def test_overlap_exchange_under_rendezvous_semantics(world, T):
    """with a backend whose send blocks until the peer's receive is posted, (i) the exchange completes (no deadlock: a rank's sends
    and receives are posted as one group), (ii) NO transfer is posted before its rank has computed all its chunks -- so nothing sits on a GPU waiting
    for a peer that is still denoising; the wait is bounded by the rank skew (block sizes differ by at most one chunk), which the recorded clocks show,
    (iii) the result is bit for bit the single-process blend."""
    import threading
    import time
    chunk, overlap, H, W = 32, 8, 3, 4
    plan = PL.chunk_plan(T, chunk, overlap)
    wts = PL.blend_weights(plan)
    shards = PL.shard_chunks(len(plan), world)
    owner, chunk_rank = PL.frame_owner(plan, shards)
    ref, _, _ = _run_rank(0, 1, T, chunk, overlap, H, W)
    fabric = _RendezvousFabric(world)
    results, errors = {}, []
    chunk_s = 0.01

    def rank_main(r):
        try:
            sb = PL.StreamingBlend(plan, wts, owner, chunk_rank, r, world, shards[r], (H, W), torch.device("cpu"), blend_fn=_cpu_blend, comm=fabric.comm(r))
            for ci in shards[r]:
                time.sleep(chunk_s)                                   # "denoise + decode" of one chunk
                fabric.tick(r)
                sb.add(ci, _fake_decoded(ci, plan[ci][1] - plan[ci][0], H, W))
                assert not any("send" in h for h in fabric.open.get((r, r - 1), [])), "a send was posted before the rank had finished its chunks"
            results[r] = sb.finish()
        except BaseException as exc:
            errors.append(exc)

    threads = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=120)
    assert not errors, errors
    out = np.zeros_like(ref.numpy())
    for r, (acc, (lo, hi)) in results.items():
        if acc is not None:
            out[lo:hi] = acc.numpy()
    assert np.array_equal(out, ref.numpy())
    n_boundaries = sum(1 for r in range(1, world) if shards[r] and shards[r - 1])
    assert len(fabric.log) >= n_boundaries and not any(fabric.open.values())
    for src, dst, c_send, c_recv, waited in fabric.log:
        assert c_send == len(shards[src]) and c_recv == len(shards[dst])      # both halves posted in finish(), after the rank's last chunk
        assert waited < (abs(len(shards[src]) - len(shards[dst])) + 1) * chunk_s + 0.5      # outstanding for the rank skew only (<= one chunk) + thread jitter


def _bench_no_launcher(argv, **extra_env):
    """`python bench.py --gpus N ...` exactly as the driver types it for N = 1: NO torch.distributed.run, no RANK / WORLD_SIZE in the environment."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(extra_env)
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py")] + argv, env=env, capture_output=True, text=True, timeout=900)
    return p, [json.loads(ln) for ln in p.stdout.splitlines() if ln.startswith("{")]


def test_bench_spawns_its_own_ranks_without_a_launcher():
    """bench.py --gpus 4 --dry-run --frames 256 with no launcher starts its own 4 ranks (fresh child processes), prints exactly
    ONE JSON line (rank 0's) and exits 0; the frames are those of the one-rank run."""
    p, lines = _bench_no_launcher(["--gpus", "4", "--dry-run", "--frames", "256", "--height", "16", "--width", "24", "--steps", "1", "--warmup", "1"])
    assert p.returncode == 0, p.stderr[-2000:]
    assert len(lines) == 1
    many = lines[0]
    assert many["n_gpus"] == 4 and many["ranks_seen"] == 4 and many["collective_backend"] == "gloo" and "sharded [3, 3, 3, 2]" in many["config"]["workload"]
    assert many["output_sha256"] == _bench_dry_run(1, 256)["output_sha256"]


def test_bench_default_weak_line_reports_the_ranks_it_saw():
    """the DEFAULT line (weak scaling: every rank runs K chunks of one long video, the form the driver's SCALE run uses) carries
    ranks_seen / collective_backend / chunks_per_rank / per-rank seconds, and -- dry run -- the frames of the 2-rank x 3-chunk job are bit for bit those
    of the 1-rank x 6-chunk job (the overlap behind the rank boundary crossed the backend: overlap_bytes_sent)."""
    p2, l2 = _bench_no_launcher(["--gpus", "2", "--dry-run", "--height", "16", "--width", "24", "--steps", "3", "--warmup", "1"])
    p1, l1 = _bench_no_launcher(["--gpus", "1", "--dry-run", "--height", "16", "--width", "24", "--steps", "6", "--warmup", "1"])
    assert p2.returncode == 0 and p1.returncode == 0, (p2.stderr[-1500:], p1.stderr[-1500:])
    two, one = l2[-1], l1[-1]
    assert len(l2) == 1 and two["scaling"] == "weak" and two["n_gpus"] == 2 and two["ranks_seen"] == 2 and two["collective_backend"] == "gloo"
    assert two["config"]["chunks_per_rank"] == 3 and two["config"]["chunks_total"] == 6 and one["ranks_seen"] == 1
    assert [r["rank"] for r in two["per_rank"]] == [0, 1] and all(r["chunks"] == 3 and r["seconds"] >= 0 for r in two["per_rank"])
    assert two["per_rank"][0]["owned_frames"] + two["per_rank"][1]["owned_frames"] == 24 * 6 + 8
    assert two["per_rank"][1]["overlap_bytes_sent"] == 8 * 16 * 24 * 3 * 4 and two["per_rank"][0]["overlap_bytes_sent"] == 0
    assert two["frame_sha256_16"] == one["frame_sha256_16"] and len(one["frame_sha256_16"]) == 152


def test_bench_launcher_reports_a_failing_rank():
    """A rank that dies after the rendezvous must not leave the launcher (or the ranks waiting for it in a collective) hanging: the launcher ends the
    survivors by PID, prints no result line and exits non-zero -- within seconds, not at a watchdog's timeout."""
    import time
    t0 = time.time()
    p, lines = _bench_no_launcher(["--gpus", "3", "--dry-run", "--frames", "100", "--height", "16", "--width", "24"], VV_DRYRUN_FAIL_RANK="1")
    assert p.returncode != 0 and not lines and "rank exit codes" in p.stderr
    assert time.time() - t0 < 120


def test_bench_refuses_a_world_size_that_is_not_gpus():
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1")
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--dry-run", "--frames", "64"], env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode != 0 and "WORLD_SIZE=1" in p.stderr


def test_bench_under_torch_distributed_run():
    """The driver's own N > 1 form: `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...`
    (dry run, gloo): the launcher's environment (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*) is honoured as is -- bench.py does NOT spawn ranks of its own
    under it -- one JSON line from rank 0, the frames of the one-rank line."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "1"
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
                        os.path.join(root, "bench.py"), "--gpus", "2", "--dry-run", "--height", "16", "--width", "24", "--steps", "2", "--warmup", "1"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [json.loads(ln) for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    two = lines[0]
    assert two["n_gpus"] == 2 and two["ranks_seen"] == 2 and two["collective_backend"] == "gloo" and two["scaling"] == "weak" and len(two["per_rank"]) == 2
    _, one = _bench_no_launcher(["--gpus", "1", "--dry-run", "--height", "16", "--width", "24", "--steps", "4", "--warmup", "1"])
    assert two["frame_sha256_16"] == one[-1]["frame_sha256_16"]
