"""N>1 path on CPU: two `gloo` ranks shard the utterance batch, run the (simulated) kernels on their
shard, all-gather the mels -- the gathered result must be bit-identical to the single-process run of
the whole batch.  On the MI355X node the same code runs with backend "nccl" (= RCCL)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, hint, out_path):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WAVESIM_THREADS="2")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from tests import helpers as H
    from tests.simlib import use_sim
    from efficientspeech_amd.sharded import sharded_forward, shard_batch
    from efficientspeech_amd.synth import synth_phonemes
    net, cfg, sd = H.make_net("tiny", "cpu")
    B, T = 4, 20
    ids, mask = synth_phonemes(B, T, 21, [20, 9, 17, 13])
    x = {"phoneme": torch.from_numpy(ids), "phoneme_mask": torch.from_numpy(mask)}
    if hint:
        x["max_mel_len"] = 110
    assert shard_batch(x, rank, world)["phoneme"].shape[0] == B // world
    with use_sim(), torch.no_grad():
        mel, mel_len, dur = sharded_forward(net, x)
        if rank == 0:
            ref_mel, ref_len, ref_dur = net(x)            # whole batch in one process
            ok = torch.equal(mel, ref_mel) and torch.equal(mel_len, ref_len) and torch.equal(dur, ref_dur)
            np.save(out_path, np.array([int(ok), mel.shape[0], mel.shape[1]]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("hint", [False, True])
def test_two_rank_shard_and_gather_is_bit_identical(tmp_path, hint):
    out = str(tmp_path / "res.npy")
    mp.spawn(_worker, args=(2, _free_port(), hint, out), nprocs=2, join=True)
    ok, B, L = np.load(out)
    assert ok == 1 and B == 4
    assert L == (110 if hint else L)


def test_shard_batch_splits_every_batch_leading_tensor():
    from efficientspeech_amd.sharded import shard_batch
    x = {"phoneme": torch.arange(12).reshape(6, 2), "phoneme_mask": torch.zeros(6, 2, dtype=torch.bool), "max_mel_len": 7}
    s = shard_batch(x, 2, 3)
    assert s["phoneme"].tolist() == [[8, 9], [10, 11]] and s["phoneme_mask"].shape == (2, 2) and s["max_mel_len"] == 7


def test_shard_batch_pads_a_ragged_batch_and_unpad_drops_the_copies():
    """world does not divide B: equal shards of ceil(B / world), copies of real utterances behind the real ones (a
    ragged last batch must not kill a multi-GPU serving loop)."""
    from efficientspeech_amd.sharded import shard_batch, shard_rows, unpad_gathered
    x = {"phoneme": torch.arange(10).reshape(5, 2), "max_mel_len": 7}
    shards = [shard_batch(x, r, 4)["phoneme"] for r in range(4)]
    assert [tuple(t.shape) for t in shards] == [(2, 2)] * 4
    assert shards[0].tolist() == [[0, 1], [2, 3]] and shards[1].tolist() == [[4, 5], [6, 7]]
    assert shards[2].tolist() == [[8, 9], [8, 9]]          # one real utterance + its copy
    assert shards[3].tolist() == [[8, 9], [8, 9]]          # owns none: copies of the batch's last utterance
    assert [shard_rows(5, r, 4) for r in range(4)] == [(0, 2, 2), (2, 4, 2), (4, 5, 2), (5, 5, 2)]
    assert torch.equal(unpad_gathered(torch.cat(shards), 5), x["phoneme"])


def _ragged_worker(rank, world, port, out_path):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WAVESIM_THREADS="2")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from tests import helpers as H
    from tests.simlib import use_sim
    from efficientspeech_amd.sharded import sharded_forward
    from efficientspeech_amd.synth import synth_phonemes
    net, cfg, sd = H.make_net("tiny", "cpu")
    ids, mask = synth_phonemes(3, 18, 33, [18, 7, 12])
    x = {"phoneme": torch.from_numpy(ids), "phoneme_mask": torch.from_numpy(mask)}
    with use_sim(), torch.no_grad():
        mel, mel_len, dur = sharded_forward(net, x)
        if rank == 0:
            ref_mel, ref_len, ref_dur = net(x)
            ok = mel.shape == ref_mel.shape and torch.equal(mel, ref_mel) and torch.equal(mel_len, ref_len) and torch.equal(dur, ref_dur)
            np.save(out_path, np.array([int(ok), mel.shape[0]]))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_ragged_batch_is_bit_identical(tmp_path):
    """B = 3 on two ranks: shards of 2 (rank 1: one real utterance + its copy); the gathered result is the 3 real utterances,
    bit-identical to the single-process forward."""
    out = str(tmp_path / "res.npy")
    mp.spawn(_ragged_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    ok, B = np.load(out)
    assert ok == 1 and B == 3


# ------------------------------------------------------------------------------------------------ the serving loop, N > 1, on one GPU
def _pipeline_worker(rank, world, port, out_path, backend="gloo"):
    """Two ranks share cuda:0 (process group `gloo`: RCCL refuses two ranks on one device; gloo carries CUDA tensors) and run
    `ShardedMelPipeline` -- the loop `bench.py --gpus N` times -- with a different batch every step: side-stream all-gather,
    MAX-reduce of the padded length, `_masked_path_inputs`, the two-stream hand-over.  Rank 0 compares every gathered step with the
    single-process forward of the whole batch."""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    if backend == "nccl":          # RCCL: one rank per GPU (the form the 8-GPU node runs)
        torch.cuda.set_device(rank)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
    else:
        torch.cuda.set_device(0)
        dist.init_process_group("gloo", rank=rank, world_size=world)
    from tests import helpers as H
    from efficientspeech_amd.sharded import ShardedMelPipeline, shard_batch
    from efficientspeech_amd.synth import synth_phonemes
    net, cfg, sd = H.make_net("tiny", "cuda")
    results = {}
    rng = np.random.default_rng(5)
    # (name, B, T, lengths, forced durations?, two_stream)
    cases = [("plain", 6, 40, [40, 33, 21, 8, 40, 17], False, False),
             ("two_stream", 6, 40, [40, 33, 21, 8, 40, 17], False, True),
             ("one_utterance_shard", 2, 23, [23, 11], False, False),
             ("one_utterance_shard_two_stream", 2, 23, [23, 11], False, True),
             ("forced_exact_length", 4, 32, [32, 32, 32, 32], True, False),
             ("forced_exact_length_two_stream", 4, 32, [32, 32, 32, 32], True, True)]
    with torch.no_grad():
        for name, B, T, lens, forced, two in cases:
            pipe = ShardedMelPipeline(net, world_size=world, gather=True, two_stream=two)
            steps, fulls = [], []
            for s in range(3):                                   # a different batch every step
                ids, mask = synth_phonemes(B, T, 100 + 7 * s + len(name), lens)
                x = {"phoneme": torch.from_numpy(ids).cuda(), "phoneme_mask": torch.from_numpy(mask).cuda()}
                if forced:
                    x.update(duration_forced=torch.full((B, T), 3 + s, dtype=torch.int32, device="cuda"), max_mel_len=T * (3 + s),
                             max_mel_len_exact=True)
                steps.append(x)
                full, lens_g = pipe.step(shard_batch(x, rank, world))
                fulls.append((full, lens_g, pipe.last_ready))    # (kept alive; read after the flush)
            pipe.flush()
            torch.cuda.synchronize()
            if rank == 0:
                ok = True
                for x, (full, lens_g, _) in zip(steps, fulls):
                    ref_mel, ref_len, _ = net(x)
                    ok = ok and full.shape == ref_mel.shape and torch.equal(full, ref_mel) and torch.equal(lens_g, ref_len)
                results[name] = int(ok)
            dist.barrier()
    if rank == 0:
        np.save(out_path, np.array([results[c[0]] for c in cases]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
def test_gpu_two_rank_serving_loop_on_one_device(tmp_path):
    """`ShardedMelPipeline` with world_size 2 (both ranks on cuda:0): every gathered step bit-identical to the whole batch in one
    process -- plain, two-stream, 1-utterance shards (duplicated to stay on the reference's masked B > 1 path), caller-vouched length."""
    out = str(tmp_path / "pipe.npy")
    mp.spawn(_pipeline_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    assert np.load(out).tolist() == [1, 1, 1, 1, 1, 1]


@pytest.mark.gpu
@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="RCCL needs one device per rank: runs on the first multi-GPU lease")
def test_gpu_rccl_serving_loop_is_bit_identical(tmp_path):
    """The same six serving-loop cases with backend "nccl" (= RCCL over xGMI), one rank per GPU: the MAX all-reduce of the padded
    length, the side-stream `all_gather_into_tensor` of mel + mel_len one step behind, every gathered step bit-identical to the
    whole batch in one process.  Auto-skips on a 1-GPU box (every lease of rounds 1-6); the 8-GPU driver run executes it."""
    out = str(tmp_path / "pipe_rccl.npy")
    mp.spawn(_pipeline_worker, args=(2, _free_port(), out, "nccl"), nprocs=2, join=True)
    assert np.load(out).tolist() == [1, 1, 1, 1, 1, 1]


@pytest.mark.gpu
def test_gpu_two_stream_loop_bounds_its_run_ahead_and_matches_the_plain_loop():
    """The single-GPU two-stream pipeline (encoder side of step i+1 beside the decoder of step i): every step's result equals the
    plain forward's bit for bit, and the host never has more than depth + 1 steps in flight (unbounded, the allocator had to find
    fresh blocks for every step: 22-36 ms per base-ES step instead of 9, profiles/r03_probes/two_stream_runahead.md)."""
    from tests import helpers as H
    from efficientspeech_amd.sharded import ShardedMelPipeline
    from efficientspeech_amd.synth import synth_phonemes
    net, cfg, sd = H.make_net("tiny", "cuda:0")
    pipe = ShardedMelPipeline(net, world_size=1, gather=False, two_stream=True)
    with torch.no_grad():
        for step in range(12):
            ids, mask = synth_phonemes(6, 40, 100 + step, [40, 33, 21, 40, 9, 17])
            x = {"phoneme": torch.from_numpy(ids).cuda(), "phoneme_mask": torch.from_numpy(mask).cuda(),
                 "duration_forced": torch.full((6, 40), 3 + step % 3, dtype=torch.int32, device="cuda"),
                 "max_mel_len": 40 * 5, "max_mel_len_exact": False}
            mel, mel_len = pipe.step(x)
            assert len(pipe.inflight) <= pipe.depth + 1
            pipe.wait_last()
            ref, ref_len, _ = net(x)
            torch.cuda.synchronize()
            assert torch.equal(mel_len, ref_len) and torch.equal(mel, ref), step
    pipe.flush()
    assert not pipe.inflight


@pytest.mark.gpu
def test_gpu_bench_self_launches_two_ranks_on_one_device():
    """`python bench.py --gpus 2` with no launcher in front of it: it re-executes itself under torch.distributed.run (one rank
    per GPU; here both ranks on cuda:0 over gloo -- RCCL refuses two ranks on one device), runs the sharded serving loop with the
    side-stream all-gather and prints the one JSON line the driver parses (the plain form used to die on an assert)."""
    import json
    import subprocess
    env = dict(os.environ, ESMI_BENCH_ONE_DEVICE="1", ESMI_BENCH_BACKEND="gloo")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2", "--batch", "8",
                        "--phonemes", "32", "--event-every", "1", "--no-extras", "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 4 and out["scaling"] == "weak"
    assert out["config"]["global_batch"] == 16 and out["config"]["frames_per_step"] == 16 * 32 * 6
    assert out["value"] > 0 and abs(out["value"] - out["config"]["frames_per_step"] / (out["ms_per_step"] * 1e-3)) < 1e-6 * out["value"]
    assert out["allgather"]["bytes_received_per_rank"] == 8 * 32 * 6 * 80 * 4 and out["allgather"]["GBps_received_per_rank"] > 0
    assert out["without_allgather"]["value"] > 0 and out["roofline"]["kernel_ms"] > 0
    # strong scaling with a ragged split: 7 utterances over 2 ranks -> shards of 4, 7 real utterances counted
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--batch", "7",
                        "--phonemes", "32", "--scaling", "strong", "--no-extras", "--no-cpu-baseline"], env=env, capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert out["scaling"] == "strong" and out["config"]["per_gpu_batch"] == 4 and out["config"]["frames_per_step"] == 7 * 32 * 6
