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


# ------------------------------------------------------------------------------------------------ the serving loop, N > 1, on one GPU
def _pipeline_worker(rank, world, port, out_path):
    """Two ranks share cuda:0 (process group `gloo`: RCCL refuses two ranks on one device; gloo carries CUDA tensors) and run
    `ShardedMelPipeline` -- the loop `bench.py --gpus N` times -- with a different batch every step: side-stream all-gather,
    MAX-reduce of the padded length, `_masked_path_inputs`, the two-stream hand-over.  Rank 0 compares every gathered step with the
    single-process forward of the whole batch."""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
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
