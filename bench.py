#!/usr/bin/env python3
"""bench.py -- mel-frames/sec of the EfficientSpeech acoustic-model forward path on MI355X.

One "step" = one full Phoneme2Mel inference forward (phoneme ids -> mel) over one synthetic batch on every rank: by default
tiny ES, B=256 utterances x T=128 phonemes per GPU, injected durations D-const = 6 (BASELINE.json configs[1]; SURVEY.md §8d;
random-init durations round to 0, fact 6), inputs already resident in HBM.  With N > 1 ranks the utterance batch is sharded
(`--scaling weak`: B per GPU, the default; `--scaling strong`: B is the GLOBAL batch, B/N per GPU) and each step ends with the
RCCL all-gather of the mel shards over xGMI, pipelined one step behind the compute on a side stream.

Prints ONE JSON line (rank 0):
  value / ms_per_step    whole-job frames/s of the timed region (barrier + synchronize on both sides, max over ranks)
  roofline               the dominant kernel (fused mel decoder), timed live with HIP events on the launch stream
  exact_fp32             (N=1) the same step on efficientspeech_amd/libesmi_fp32mfma.so: every contraction on
                         v_mfma_f32_32x32x2_f32, with its own kernel time and fraction of the 157.3 TFLOP/s fp32-MFMA peak
  decoder_only           (N=1) MelDecoder.forward alone on frame-rate features ~ N(0,1)
  allgather              (N>1) the mel all-gather timed by itself: GB/s received per rank vs 7 xGMI links x 76.8 GB/s
  strong                 (N>1) the same model with the GLOBAL batch of the headline (256 utterances) sharded over the ranks
  base_strong            (N>1) BASELINE configs[3]: base ES, global B=512 x T=256, batch-sharded over the ranks, mel all-gather
  base                   (N=1) base ES B=512 T=256 on one GPU: ms/step, decoder kernel roofline, D-rand, exact-fp32 build
  without_allgather      (N>1) the same sharded steps with the exchange switched off (compute scaling next to the link-bound value)
  pytorch_rocm_ops       (N=1) the same forward written with stock PyTorch-ROCm operators, same inputs / weights / GPU
  b1_fox_gpu             (N=1) single-utterance latency of the forward (BASELINE configs[0] shape) with a sync per call
  vocoder                (N=1) the HiFi-GAN v2 generator on the mel the forward produced: mel-frames/s, TFLOP/s
  train_step             the training step (BASELINE configs[4]): ms/step at the reference's batch size per GPU
  cpu_baseline           (N=1) the C oracle (oracle/, fp32 accumulation, OpenMP) on this box's host cores: all cores and
                         n=24 (the reference's --threads default), plus the B=1 fox-sentence latency (BASELINE configs[0]) --
                         a reported baseline, not the target.  `torch_ops`: the same forward as PyTorch CPU ops (SURVEY 8d's own
                         definition of the baseline) at torch.set_num_threads(24) and at all physical cores
  small                  (N=1) BASELINE configs[2]: small ES B=256 T=256 on one GPU, like `base`
  encoder_side           (N=1) stage 1 of the forward alone (everything in front of the mel decoder): us, FLOPs, fraction of the bound
  roofline.clock/.pipes  the shader clock measured inside the decoder kernel (+ frac re-priced at it) and the per-pipe busy fractions
                         of the committed PMC pass
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

FP32_PEAK_TFLOPS = 157.3     # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 == fp32 vector peak
F16_PEAK_TFLOPS = 2500.0     # dense 16-bit MFMA peak
HBM_PEAK_GBS = 8000.0
XGMI_LINK_GBS, XGMI_LINKS = 76.8, 7      # per direction and link; 7 peers on an 8-GPU node
# SURVEY.md §8d algorithmic work per valid mel frame, decoder only: (flops, bytes)
DECODER_WORK = {"tiny": (189_440, 832), "small": (973_824, 1344), "base": (1_505_792, 2368)}
DEFAULT_BATCH = {"tiny": 256, "small": 256, "base": 512}
DEFAULT_PHONEMES = {"tiny": 128, "small": 256, "base": 256}
FOX_IDS = [92, 74, 117, 145, 110, 117, 89, 131, 83, 120, 105, 67, 117, 132, 116, 75, 119, 130, 132, 124, 144, 98, 92, 74, 118, 103,
           147, 113, 91, 79, 106]     # "the quick brown fox ..." as ARPAbet ids


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", default="tiny", choices=["tiny", "small", "base"])
    ap.add_argument("--batch", type=int, default=None, help="utterances per GPU (weak) or in total (strong); default: BASELINE config")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak: --batch utterances per GPU; strong: --batch utterances in total, sharded over the ranks")
    ap.add_argument("--phonemes", type=int, default=None)
    ap.add_argument("--dur", type=int, default=6, help="injected frames per phoneme (D-const)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="N=1: skip the exact-fp32 and decoder-only legs")
    ap.add_argument("--no-gather", action="store_true", help="skip the mel all-gather (N>1 debugging)")
    ap.add_argument("--two-stream", action="store_true",
                    help="software-pipeline consecutive steps: encoder side of step i+1 concurrent with the decoder of step i")
    ap.add_argument("--exact-fp32", action="store_true",
                    help="run the MAIN measurement on efficientspeech_amd/libesmi_fp32mfma.so (all contractions on "
                         "v_mfma_f32_32x32x2_f32) instead of the split-f16 default build")
    ap.add_argument("--no-auto-launch", action="store_true",
                    help="N=1 only: do not try the two-stream pipeline during warm-up (default: warm up both launch modes and "
                         "keep two-stream only if it is >= 3 %% faster -- it is on boxes whose GPU drops to a low sclk state "
                         "during the light encoder-side kernels, and ~3 %% slower elsewhere)")
    ap.add_argument("--graph", action="store_true",
                    help="replay the forward as two hipGraphs per step instead of eager launches")
    ap.add_argument("--event-every", type=int, default=5,
                    help="bracket every n-th mel-decoder launch of the timed region with HIP events for roofline.kernel_ms "
                         "(each event pair drains the queue: ~12 us per step when placed on every launch)")
    return ap.parse_args()


def _omp_threads(n):
    import ctypes
    try:
        ctypes.CDLL("libgomp.so.1").omp_set_num_threads(int(n))
        return True
    except OSError:
        return False


def cpu_baseline(cfg, sd, T, dur):
    """Time the oracle (fp32-accumulating build) on bounded samples of the same workload: n = 24 threads (the reference's
    --threads default, utils/tools.py:324 -- and the fastest setting for this port) and the B = 1 fox sentence (BASELINE
    configs[0]).  (Round 2 also printed an all-hardware-threads figure: 1.3e3 frames/s at 256 threads, a fork/join pathology of
    the port's one-parallel-region-per-layer structure and not a baseline of anything; it is gone.)"""
    from oracle import oracle
    from efficientspeech_amd.synth import synth_phonemes
    w = oracle.Weights(sd)
    cores = len(os.sched_getaffinity(0))

    def run(b, t=T, ids=None):
        if ids is None:
            ids, mask = synth_phonemes(b, t, 99)
        else:
            mask = None
        d = np.full(ids.shape, dur, np.int32)
        t0 = time.perf_counter()
        # the same data flow as the timed GPU step: eval path (predicted pitch / energy bucketised), injected durations
        o = oracle.phoneme2mel(cfg, w, ids, mask, pitch=None, energy=None, duration=d, f32=True)
        return int(o.mel_len.sum()), time.perf_counter() - t0

    def sample(threads, b, reps, seconds=8.0):
        """at least `reps` runs, and as many as fit into ~`seconds` of wall time (x threads = the CPU work the number rests on)"""
        _omp_threads(threads)
        run(4)                                   # spin up the OpenMP pool at this width
        r = []
        while len(r) < reps or (sum(x[1] for x in r) < seconds and len(r) < 200):
            r.append(run(b))
        return r[0][0], sum(x[1] for x in r) / len(r), len(r)
    # n = 24: the reference's --threads default -- and the fastest setting for this port (its OpenMP regions are one layer
    # each: beyond ~32 threads the fork/join cost dominates; measured on the MI355X box's 256 hardware threads, B = 64:
    # 24 threads 8.7e5 frames/s, 64: 4.0e5, 128: 1.6e5, 256: 5e3).  `value` is the n = 24 number.
    n24 = min(24, cores)
    b24 = 256 if cfg.name == "tiny" else 64
    frames, dt, n_runs = sample(n24, b24, 3)
    out = {"value": frames / dt, "unit": "mel-frames/s", "cores": n24, "kind": "port",
           "sample": f"oracle/es_oracle.c (fp32 accumulate, OpenMP {n24} threads = the reference's --threads default), {cfg.name} ES "
                     f"full forward, B={b24} T={T} D-const {dur}: {frames} frames in {dt:.2f} s (mean of {n_runs} runs = {n_runs * dt:.1f} s x {n24} threads of CPU work)"}
    # n = all physical cores.  The port's OpenMP regions are one per layer, so its intra-op scaling stops at a few
    # dozen threads; utterances are independent, though, so the all-cores figure shards the BATCH instead: one worker process
    # per core, each running the whole forward of its utterances with single-threaded layers.  Same arithmetic, same total work, one "parallel region" per forward.
    try:
        phys = _physical_cores()
        nall = max(1, min(phys, cores, b24))
        out["all_cores"] = _utterance_parallel_sample(run_chunk=lambda ids_, mask_: oracle.phoneme2mel(
            cfg, w, ids_, mask_, pitch=None, energy=None, duration=np.full(ids_.shape, dur, np.int32), f32=True),
            b=b24, t=T, threads=nall)
        out["all_cores"]["sample"] = (f"the same forward, B={b24} sharded by utterance over {nall} forked worker processes (one per physical core: "
                                      f"{phys} cores / {cores} hardware threads visible), layers single-threaded inside a worker")
    except Exception as e:                         # noqa: BLE001
        out["all_cores"] = {"error": repr(e)}
    # SURVEY 8d's own definition of the CPU baseline: the path restated with PyTorch ops (the nearest thing to the reference's
    # `--infer-device cpu`, demo.py:130-135, that can travel: tests/torch_mirror.py, pinned to the reference's fixtures) under
    # torch.set_num_threads(n) at n = 24 (the reference's --threads default, utils/tools.py:324) and n = all physical cores
    try:
        out["torch_ops"] = _torch_ops_cpu(cfg, sd, T, dur, n24, min(_physical_cores(), cores))
    except Exception as e:                         # noqa: BLE001
        out["torch_ops"] = {"error": repr(e)}
    fox = np.asarray([FOX_IDS], np.int32)
    for threads, key in ((n24, "b1_fox_latency_ms_n24"), (1, "b1_fox_latency_ms_n1")):
        _omp_threads(threads)
        run(1, ids=fox)
        ts = [run(1, ids=fox)[1] for _ in range(5)]
        out[key] = float(np.median(ts) * 1e3)
    out["b1_fox_note"] = (f"B=1, T={len(FOX_IDS)} fox sentence (BASELINE configs[0] shape), D-const {dur}: {len(FOX_IDS) * dur} frames; "
                          "median of 5")
    _omp_threads(cores)
    return out


def _torch_ops_cpu(cfg, sd, T, dur, n24, nall, b=256, runs=3):
    """The forward as PyTorch CPU ops (oneDNN convolutions, torch softmax / layer_norm / repeat_interleave in the reference's
    structure: tests/torch_mirror.eval_forward) on the host cores: >= `runs` timed forwards at B = 256 per thread setting."""
    from tests import torch_mirror as _mirror
    from efficientspeech_amd.synth import synth_phonemes
    net = make_net(cfg, sd, "cpu").eval()
    ids, mask = synth_phonemes(b, T, 99)
    x = {"phoneme": torch.from_numpy(ids), "phoneme_mask": torch.from_numpy(mask),
         "duration_forced": torch.full((b, T), dur, dtype=torch.int32)}
    res, old = {}, torch.get_num_threads()
    try:
        for key, n in (("n24", n24), ("all_physical_cores", nall)):
            torch.set_num_threads(int(n))
            with torch.no_grad():
                _mirror.eval_forward(net, x)                       # warm the thread pool and the primitive cache at this width
                ts, frames = [], 0
                while len(ts) < runs or (sum(ts) < 6.0 and len(ts) < 20):
                    t0 = time.perf_counter()
                    _, mel_len, _ = _mirror.eval_forward(net, x)
                    ts.append(time.perf_counter() - t0)
                    frames = int(mel_len.sum())
            res[key] = {"value": frames / (sum(ts) / len(ts)), "unit": "mel-frames/s", "threads": int(n), "runs": len(ts),
                        "seconds_per_run": sum(ts) / len(ts), "best_seconds": min(ts)}
    finally:
        torch.set_num_threads(old)
    res["kind"] = "port (PyTorch CPU ops)"
    res["sample"] = (f"tests/torch_mirror.eval_forward (plain PyTorch ops on CPU tensors, torch {torch.__version__}), {cfg.name} ES full forward, "
                     f"B={b} T={T} D-const {dur}")
    return res


def _physical_cores():
    """Physical cores among the CPUs this process may run on (unique (package, core) pairs of /proc/cpuinfo)."""
    allowed = os.sched_getaffinity(0)
    seen, cpu, pkg = set(), None, 0
    try:
        for line in open("/proc/cpuinfo"):
            k, _, v = line.partition(":")
            k = k.strip()
            if k == "processor":
                cpu = int(v)
            elif k == "physical id":
                pkg = int(v)
            elif k == "core id" and cpu in allowed:
                seen.add((pkg, int(v)))
    except OSError:
        pass
    return len(seen) or max(1, len(allowed) // 2)


_UP_STATE = {}


def _up_work(i):
    """One worker's chunk (forked process: the batch and the closure are inherited through _UP_STATE)."""
    run_chunk, chunks = _UP_STATE["run"], _UP_STATE["chunks"]
    _omp_threads(1)                                     # every worker's layers run on its own core only
    c = chunks[i]
    if c[0].shape[0] == 1:                              # a 1-utterance chunk must stay on the masked (B > 1) path like its batch
        o = run_chunk(np.concatenate([c[0], c[0]]), np.concatenate([c[1], c[1]]))
        return int(o.mel_len[0])
    return int(run_chunk(c[0], c[1]).mel_len.sum())


def _utterance_parallel_sample(run_chunk, b, t, threads, seconds=6.0):
    """frames/s of `run_chunk` over a batch of b utterances split into `threads` contiguous chunks, one forked worker process each
    (Python threads would serialise on the GIL in the glue between the C calls: 128 threads measured 1.14x of 24)."""
    import multiprocessing as mp
    from efficientspeech_amd.synth import synth_phonemes
    ids, mask = synth_phonemes(b, t, 99)
    bounds = np.linspace(0, b, threads + 1).astype(int)
    chunks = [(ids[lo:hi], mask[lo:hi]) for lo, hi in zip(bounds[:-1], bounds[1:]) if hi > lo]
    _UP_STATE.update(run=run_chunk, chunks=chunks)
    idx = list(range(len(chunks)))
    with mp.get_context("fork").Pool(len(chunks)) as pool:     # (the children never touch the GPU runtime)
        pool.map(_up_work, idx, chunksize=1)            # warm the workers
        times, frames = [], 0
        while len(times) < 2 or (sum(times) < seconds and len(times) < 50):
            t0 = time.perf_counter()
            frames = sum(pool.map(_up_work, idx, chunksize=1))
            times.append(time.perf_counter() - t0)
    dt = sum(times) / len(times)
    return {"value": frames / dt, "unit": "mel-frames/s", "cores": len(chunks), "runs": len(times), "seconds_per_run": dt}


def pmc_traffic(config, B, T, dur):
    """HBM bytes per mel_decoder launch from the committed rocprofv3 PMC passes (profiles/*pmc_counters.json:
    FETCH_SIZE and WRITE_SIZE collected in separate passes, read side doubled per the gfx950 note in
    MI355X_MICROARCH.md).  Only valid for the workload the counters were collected on; otherwise null."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*pmc_counters.json")), reverse=True):
        try:
            d = json.load(open(path))
        except Exception:
            continue
        if d.get("workload", "").startswith(f"{config} ES B={B} T={T} D-const {dur} "):
            m = d["mel_decoder"]
            return m["hbm_traffic_bytes_corrected"], os.path.basename(path), m.get("mfma_pipe_utilisation")
    return None, None, None


def pmc_pipes(config, B, T, dur):
    """Per-pipe busy fractions of the mel decoder and the shader clock under the tracer, from the newest committed PMC summary of
    this workload that has them (profiles/*pmc_counters.json `mel_decoder.pipes`, tools/prof_summary.py) -> (dict, file) or (None, None)."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*pmc_counters.json")), reverse=True):
        try:
            d = json.load(open(path))
        except Exception:
            continue
        if d.get("workload", "").startswith(f"{config} ES B={B} T={T} D-const {dur} ") and "pipes" in d.get("mel_decoder", {}):
            return d["mel_decoder"]["pipes"], os.path.basename(path)
    return None, None


def decoder_clock(net, x, dev, steps=20):
    """Shader clock during the mel decoder kernel, measured INSIDE the kernel (esmi_mel_decoder_clock_probe: s_memtime / s_memrealtime
    stamps by the launch's first workgroup: at its start and at the start of its last stage / chunk).
    -> dict or None.  Its own short run outside the timed region; the probe is disarmed again before returning."""
    from efficientspeech_amd import _lib
    lib = _lib.load()
    slots = torch.zeros(4, dtype=torch.int64, device=dev)
    ghz, span = [], []
    try:
        lib.esmi_mel_decoder_clock_probe(slots.data_ptr())
        with torch.no_grad():
            for _ in range(steps):
                net(x)
                torch.cuda.synchronize(dev)
                s0, r0, s1, r1 = (int(v) for v in slots.cpu().tolist())
                if r1 > r0 and s1 > s0:
                    ghz.append((s1 - s0) / (r1 - r0) * 0.1)
                    span.append((r1 - r0) * 0.01)
    finally:
        lib.esmi_mel_decoder_clock_probe(None)
    if not ghz:
        return None
    return {"shader_ghz": float(np.median(ghz)), "shader_ghz_min_max": [float(min(ghz)), float(max(ghz))], "launches": len(ghz),
            "span_us": float(np.median(span)),
            "how": "s_memtime (shader clock) over s_memrealtime (100 MHz) between the start of the launch's first workgroup and the start "
                   "of its last stage (dx2 = 256: of its last chunk): span_us of that workgroup's life"}


def encoder_side(net, x, dev, cfg, B, T, peak_tf, steps=50):
    """The encoder side by itself (stage 1 of the one-call forward: encoder blocks, Fuse, variance adaptor, length-regulator scan,
    the decoder's phoneme-rate first stage), back to back on the launch stream, HIP events around the whole loop."""
    enc_flops = {"tiny": 262_336, "small": 737_664, "base": 4_489_984}[cfg.name]          # SURVEY 8d, per phoneme
    with torch.no_grad():
        for _ in range(5):
            net._launch(x, stage=1)
        torch.cuda.synchronize(dev)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(steps):
            net._launch(x, stage=1)
        e1.record()
        torch.cuda.synchronize(dev)
    us = e0.elapsed_time(e1) * 1e3 / steps
    ach = enc_flops * B * T / (us * 1e-6) / 1e12
    return {"us": us, "steps": steps, "algorithmic_flops_per_phoneme": enc_flops, "achieved_tflops": ach, "peak": peak_tf,
            "frac": ach / peak_tf,
            "note": "stage 1 of esmi_phoneme2mel_forward_f32 alone, back-to-back launches, HIP events around the loop (launch gaps "
                    "included); FLOPs = SURVEY 8d's matmul-class FLOPs per phoneme (encoder + fuse + 3 predictors), NOT counting "
                    "the decoder's first stage that also runs here"}


def parity_spot(net, x, cfg, sd, sel=None, decoder_feats=None):
    """Parity gate of a timed leg (BASELINE.md 4 "parity gate before any timing counts"), OUTSIDE every timed region: the batch the leg
    just timed goes through the path once more and `sel` utterances of its output (default: the first and the last one) are held
    against the CPU oracle (oracle/: the checker, never the thing measured) run on the same inputs -- durations as injected,
    pitch / energy teacher-forced with the path's own predictions so that a bucket decision cannot flip (the predictions themselves
    are compared at 2e-5), padded to the length the path derived.  `decoder_feats`: MelDecoder.forward alone on frame-rate features.
    -> {"utterances", "max_abs_err", "mel_len_ok", "ok"}; a leg whose gate fails must report "error" instead of a number."""
    from oracle import oracle
    w = oracle.Weights(sd)
    with torch.no_grad():
        if decoder_feats is not None:
            sel = sel or [0, decoder_feats.shape[0] - 1]
            mel = net.decoder(decoder_feats)[sel].cpu().numpy()         # the batch that was timed; two of its utterances are compared
            ref = oracle.mel_decoder(cfg, w, decoder_feats[sel].cpu().numpy())
            err = float(np.abs(mel - ref).max())
            return {"utterances": sel, "max_abs_err": err, "mel_len_ok": True, "tol": 1e-4, "ok": bool(err < 1e-4)}
        enc = net.encoder._encode(x)
        mel, mel_len, _ = net(x)
    B = mel.shape[0]
    sel = sel or [0, B - 1]
    ids, mask = x["phoneme"][sel].cpu().numpy(), x["phoneme_mask"][sel].cpu().numpy()
    dur = x["duration_forced"][sel].cpu().numpy().astype(np.int32) if "duration_forced" in x else None
    L = int(mel_len.max())
    o = oracle.phoneme2mel(cfg, w, ids, mask, pitch=enc["pitch"][sel, :, 0].cpu().numpy(), energy=enc["energy"][sel, :, 0].cpu().numpy(),
                           duration=dur if dur is not None else enc["dur"][sel].cpu().numpy(), max_mel_len=L)
    pred_err = max(float(np.abs(enc[k][sel].cpu().numpy() - getattr(o, k)).max()) for k in ("pitch", "energy", "duration"))
    len_ok = bool(np.array_equal(mel_len[sel].cpu().numpy(), o.mel_len))
    if dur is not None:
        len_ok = len_ok and bool(np.array_equal(mel_len.cpu().numpy(), x["duration_forced"].sum(1).cpu().numpy()))
    m = mel[sel].cpu().numpy()
    err = float(np.abs(m[:, :L] - o.mel).max()) if L else 0.0
    tail_zero = not m[:, L:].any() and all(not m[j, int(o.mel_len[j]):].any() for j in range(len(sel)))
    return {"utterances": [int(v) for v in sel], "max_abs_err": err, "predictor_max_abs_err": pred_err, "mel_len_ok": len_ok,
            "rows_beyond_mel_len_zero": bool(tail_zero), "tol": 1e-4,
            "ok": bool(err < 1e-4 and pred_err < 2e-5 and len_ok and tail_zero)}


def gate(res, spot):
    """attach a leg's parity gate; a failed gate replaces the leg's numbers by an error (its timing does not count)"""
    if spot.get("ok"):
        res["parity_spot"] = spot
        return res
    return {"error": "parity gate failed: the timed batch does not match the oracle", "parity_spot": spot}


def make_net(cfg, sd, dev):
    from efficientspeech_amd import build_phoneme2mel, load_numpy_state_dict
    net = build_phoneme2mel(cfg)
    load_numpy_state_dict(net, sd)
    return net.to(dev)


def timed_steps(pipe, net, x, steps, warmup, sync_all, event_every, graph):
    """warm-up, then exactly `steps` steps bracketed by sync_all(); -> (seconds, mean decoder kernel ms, samples)"""
    with torch.no_grad():
        for _ in range(warmup):
            pipe.step(x)
        pipe.flush()
        net.decoder.timing = []
        net.decoder.timing_every = event_every
        pipe.dec_events = [] if graph else None
        sync_all()
        t0 = time.perf_counter()
        for _ in range(steps):
            pipe.step(x)
        pipe.flush()
        sync_all()
        dt = time.perf_counter() - t0
    ev = pipe.dec_events if pipe.dec_events else net.decoder.timing
    net.decoder.timing = None
    dec_ms = float(np.mean([s.elapsed_time(e) for s, e in ev])) if ev else float("nan")
    return dt, dec_ms, len(ev)


def _self_launch(n):
    """`python bench.py --gpus N` without a launcher (no WORLD_SIZE in the environment): become
    `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py <same arguments>` -- one rank per
    GPU over RCCL, like the reference's single flag (`train.py --devices N`)."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")        # dmabuf IPC (the host driver's only mode): RCCL needs it
    os.execv(sys.executable, [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
                              "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:])


def trace_kernel_us(config):
    """Average duration (us) of the mel decoder kernel in the newest committed `rocprofv3 --kernel-trace --stats` summary of this
    config (profiles/*kernel_stats.md; tiny has no config tag in the file name) -> (us, file) or (None, None)."""
    import glob
    import re
    tag = "" if config == "tiny" else config + "_"
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", f"r??_?_{tag}kernel_stats.md")), reverse=True):
        for line in open(path):
            m = re.match(r"\| `void esmi::mel_decoder_kernel<.*\| (\d+) \| ([\d.]+) \| ([\d.]+) \|", line)
            if m:
                return float(m.group(3)), os.path.basename(path)
    return None, None


def main():
    a = parse()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        _self_launch(a.gpus)
    fp32_lib = os.path.join(ROOT, "efficientspeech_amd", "libesmi_fp32mfma.so")
    if a.exact_fp32:
        os.environ["ESMI_LIB"] = fp32_lib
    rank = int(os.environ.get("RANK", 0))
    local = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    assert world == a.gpus, f"--gpus {a.gpus} but WORLD_SIZE={world}"
    if os.environ.get("ESMI_BENCH_ONE_DEVICE"):            # development: exercise the N>1 code path on a 1-GPU box
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        import torch.distributed as dist
        backend = os.environ.get("ESMI_BENCH_BACKEND", "nccl")     # "nccl" is RCCL on ROCm; "gloo" only for the 1-GPU dry run
        dist.init_process_group(backend, **({"device_id": dev} if backend == "nccl" else {}))

    from efficientspeech_amd import CONFIGS, _lib
    from efficientspeech_amd.synth import synth_state_dict, synth_phonemes
    from efficientspeech_amd.sharded import ShardedMelPipeline
    cfg = CONFIGS[a.config]
    B_arg = a.batch or DEFAULT_BATCH[a.config]
    if a.scaling == "strong":
        B = -(-B_arg // world)      # equal shards of ceil(B / N): a ragged split computes copies of real utterances (sharded.shard_batch)
    else:
        B = B_arg
    T = a.phonemes or DEFAULT_PHONEMES[a.config]
    L = T * a.dur
    sd = synth_state_dict(cfg, 1234)
    net = make_net(cfg, sd, dev)
    ids, mask = synth_phonemes(B, T, 1234 + rank)
    # D-const: the padded length is known on the host and exact -> no device->host sync and (N > 1) no MAX all-reduce of it
    x = {"phoneme": torch.from_numpy(ids).to(dev), "phoneme_mask": torch.from_numpy(mask).to(dev),
         "duration_forced": torch.full((B, T), a.dur, dtype=torch.int32, device=dev), "max_mel_len": L, "max_mel_len_exact": True}
    pipe = ShardedMelPipeline(net, world_size=world, gather=(world > 1 and not a.no_gather), use_graph=a.graph, two_stream=a.two_stream)

    def sync_all():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize(dev)

    launch_note = ""
    if world == 1 and not (a.graph or a.two_stream or a.no_auto_launch):
        # untimed: which launch mode is faster on THIS box?  (same kernels, same work; only the stream schedule differs)
        alt = ShardedMelPipeline(net, world_size=1, gather=False, two_stream=True)

        def trial(pl, n):
            with torch.no_grad():
                for _ in range(5):
                    pl.step(x)
                pl.flush(); torch.cuda.synchronize(dev)
                t = time.perf_counter()
                for _ in range(n):
                    pl.step(x)
                pl.flush(); torch.cuda.synchronize(dev)
            return (time.perf_counter() - t) / n
        n_try = max(10, a.warmup)
        # two alternating rounds, the better of each mode: the first trial on a cold box (clocks, page cache) is not the mode's speed
        t_eager, t_two = trial(pipe, n_try), trial(alt, n_try)
        t_eager, t_two = min(t_eager, trial(pipe, n_try)), min(t_two, trial(alt, n_try))
        if t_two < 0.97 * t_eager:
            pipe, a.two_stream = alt, True
        launch_note = f"; warm-up trial: eager {t_eager * 1e3:.3f} ms/step, two-stream {t_two * 1e3:.3f} ms/step"
    dt, dec_ms, n_ev = timed_steps(pipe, net, x, a.steps, a.warmup, sync_all, a.event_every, a.graph)
    if a.two_stream and world == 1:
        # with the encoder side of step i+1 running beside it, the decoder's event-timed duration is not the kernel's own: time
        # the kernel for the roofline in a short single-stream run (the step time above stays the two-stream one)
        solo = ShardedMelPipeline(net, world_size=1, gather=False)
        _, dec_ms, n_ev = timed_steps(solo, net, x, max(10, a.steps // 5), 3, sync_all, a.event_every, False)
        launch_note += "; roofline kernel_ms from a single-stream run"
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    # every frame valid under D-const, no padding; a ragged strong split counts the real utterances only
    frames_per_step = (B_arg if a.scaling == "strong" else B * world) * L
    value = frames_per_step * a.steps / dt
    flops, nbytes = DECODER_WORK[a.config]
    # MelDecoder's first stage (proj Linear + Tanh + LN) is row-wise and runs at PHONEME rate on the encoder side: the decoder
    # kernel itself then executes this much less
    # (round 4: every size -- inside the fused variance-adaptor kernel for tiny at T <= 128, else as one phoneme-rate GEMM launch)
    head_moved = os.environ.get("ESMI_HEAD_GEMM", "1") != "0" or (cfg.d4 == 128 and cfg.dx2 == 128 and T <= 128)
    kernel_flops = flops - (2 * cfg.d4 * cfg.dx2 if head_moved else 0)
    build_cfg = _lib.load().esmi_build_config().decode()
    split = 3 if build_cfg.startswith("dec_gemm=split-f16x2") else 0      # 16-bit MFMA products per fp32-accurate product
    split_txt = ("fp32 operands split into 2 f16 pieces (22 significand bits, weights pre-scaled 2^8), 3 f16-MFMA products, fp32 "
                 "accumulate") if split else ""
    # roofline peak of the contraction actually executed: an fp32-accurate product costs `split` products on the 16-bit matrix
    # pipe (dense peak 2500 TFLOP/s, MI355X_MICROARCH.md), or one on the fp32 MFMA path (157.3)
    peak_tf = F16_PEAK_TFLOPS / split if split else FP32_PEAK_TFLOPS
    ach_tf = flops * B * L / (dec_ms * 1e-3) / 1e12
    # algorithmic HBM bytes per valid frame of the variant that RUNS: the decoder gathers its input rows at phoneme rate
    # through the duration scan -- dx2 floats of h0 per phoneme when its first stage ran in the variance-adaptor kernel, else d4
    # floats of features -- and writes 80 floats of mel per frame: 405 / 491 / 661 B per frame at D-const 6 (tiny / small / base).
    # (832 / 1344 / 2368 B is the stand-alone MelDecoder.forward on frame-rate features: the `decoder_only` leg.)
    exec_bytes = 4.0 * cfg.n_mel_channels + 4.0 * (cfg.dx2 if head_moved else cfg.d4) / a.dur
    traffic, traffic_src, mfma_util = (None, None, None) if a.exact_fp32 else pmc_traffic(a.config, B, T, a.dur)
    tr_us, tr_src = (None, None) if a.exact_fp32 or (B, T) != (DEFAULT_BATCH[a.config], DEFAULT_PHONEMES[a.config]) else trace_kernel_us(a.config)
    out = {
        "metric": "mel-frames/sec (whole node), full Phoneme2Mel forward", "value": value, "unit": "mel-frames/s",
        "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": dt / a.steps * 1e3,
        "higher_is_better": True, "scaling": a.scaling, "vs_baseline": None,
        "dtype": "f32" + (f" (weight GEMMs: {split_txt})" if split else " (every contraction on v_mfma_f32_32x32x2_f32)"),
        "data": "synthetic",
        "mRTF": value * 256 / 22050,
        "config": {"workload": f"{a.config} ES ({sum(v.size for v in sd.values())} params, seeded random weights), "
                               f"synthetic phoneme batch B={B} T={T} per GPU, injected durations D-const {a.dur} "
                               f"(L={L}), eval path, mel all-gather over RCCL when N>1",
                   "global_batch": (B_arg if a.scaling == "strong" else B * world), "per_gpu_batch": B, "phonemes": T, "frames_per_step": frames_per_step,
                   "parallelism": f"batch-shard x{world}",
                   "launch": ("hipGraph replay (encoder graph + decoder graph per step)" if a.graph else "eager")
                             + (", encoder/decoder on two streams (steps software-pipelined)" if a.two_stream else "") + launch_note},
        "roofline": {"bound": "mfma", "kernel": "mel_decoder_kernel", "achieved": ach_tf, "peak": peak_tf,
                     "unit": "TFLOP/s", "frac": ach_tf / peak_tf,
                     "peak_is": (f"dense 16-bit MFMA peak {F16_PEAK_TFLOPS:.0f} TFLOP/s / {split} products per fp32-accurate product"
                                 if split else "v_mfma_f32_32x32x2_f32 peak"),
                     "traffic": traffic, "traffic_unit": "HBM bytes per launch (rocprofv3 PMC FETCH_SIZE x2 + WRITE_SIZE; both "
                                                         "calibrated on known byte counts in this kernel's access pattern: profiles/r03_probes/pmc_calibration.md)",
                     "traffic_source": traffic_src, "mfma_pipe_utilisation_pmc": mfma_util,
                     "algorithmic_bytes_per_launch": exec_bytes * B * L, "algorithmic_bytes_per_frame": exec_bytes,
                     "traffic_ratio": (traffic / (exec_bytes * B * L)) if traffic else None,
                     "algorithmic_bytes_note": "the executed variant: input rows gathered at phoneme rate (SURVEY 8d: 405 / 491 / 661 B per "
                                               "frame); the frame-rate-input figure (832 / 1344 / 2368) belongs to `decoder_only`",
                     "algorithmic_flops_per_frame": flops,
                     "kernel_ms": dec_ms, "kernel_ms_samples": n_ev, "build_config": build_cfg,
                     "frac_is": "event-timed (kernel_ms: live HIP events on the launch stream, this run)",
                     "trace_kernel_us": tr_us, "trace_source": tr_src,
                     "frac_trace": (flops * B * L / (tr_us * 1e-6) / 1e12 / peak_tf) if tr_us else None,
                     "frac_trace_is": "the same FLOPs over the committed rocprofv3 --kernel-trace --stats average of this kernel (tracer "
                                      "overhead included; a different run and box than kernel_ms)",
                     "contraction": ((f"{split_txt}; measured error <= that of an fp32 FMA chain (HISTORY.md 3); `peak` is the 16-bit "
                                      "matrix-pipe peak divided by the products per fp32-accurate product; `exact_fp32` below is the "
                                      "same step on the v_mfma_f32_32x32x2_f32 build") if split else "v_mfma_f32_32x32x2_f32 (exact fp32)"),
                     "proj_stage_at_phoneme_rate": head_moved, "kernel_flops_per_frame": kernel_flops,
                     "frac_kernel_flops": kernel_flops * B * L / (dec_ms * 1e-3) / 1e12 / peak_tf,
                     "hbm_frac": exec_bytes * B * L / (dec_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                     "note": "MFMA bound (228 FLOP/B >> machine balance); hbm_frac reported because north_star quotes the HBM "
                             "roofline.  achieved/frac use SURVEY 8d's algorithmic FLOP per frame; frac_kernel_flops counts only what "
                             "the decoder kernel still computes per frame (its row-wise first stage runs once per phoneme in "
                             "enc_fuse_va_kernel).  Round 6: a time budget of this kernel that sums to its duration (profiles/r06_dec_budget.md: K loop "
                             "19 %, LayerNorm 18 %, barrier wait 16 %, depthwise 15 %, prologue 14 %, tanh 13 %, mel 5 % of a workgroup's life) and the ONE "
                             "structural experiment chosen from it (frame -> phoneme search in LDS instead of 7 dependent L2 round trips): 175.6 vs 175.5 us "
                             "= 0.0 %, below the 8 % bar: stopped"},
    }

    if rank == 0:
        # ---- parity gate of the headline (outside the timed region): two utterances of the timed batch against the CPU oracle
        try:
            out["parity_spot"] = parity_spot(net, x, cfg, sd)
            if not out["parity_spot"]["ok"]:
                out["error"] = "parity gate failed: the timed batch does not match the oracle; value withdrawn"
                out["value_unverified"], out["value"] = out["value"], None
        except Exception as e:                     # noqa: BLE001
            out["parity_spot"] = {"error": repr(e)}
    if rank == 0 and world == 1 and not a.exact_fp32:
        # ---- what the decoder kernel's time consists of, in the line itself: the clock the chip actually ran it
        # at (measured inside the kernel), `frac` re-priced at that clock, and the busy fraction of each pipe from the committed PMC pass
        try:
            clk = decoder_clock(net, x, dev)
            if clk:
                clk["peak_clock_ghz"] = 2.4
                clk["frac_at_clock"] = out["roofline"]["frac"] * 2.4 / clk["shader_ghz"]
                clk["note"] = ("roofline.peak assumes the 2.4 GHz peak engine clock; the chip runs this kernel at shader_ghz (power "
                               "management), so the matrix pipe's own ceiling during the kernel is peak x shader_ghz / 2.4")
                out["roofline"]["clock"] = clk
        except Exception as e:                     # noqa: BLE001
            out["roofline"]["clock"] = {"error": repr(e)}
        pipes, pipes_src = pmc_pipes(a.config, B, T, a.dur)
        if pipes:
            out["roofline"]["pipes"] = dict(pipes, source=pipes_src)
        try:
            out["encoder_side"] = encoder_side(net, x, dev, cfg, B, T, peak_tf)
        except Exception as e:                     # noqa: BLE001
            out["encoder_side"] = {"error": repr(e)}

    if world > 1 and pipe.gather:
        # the exchange by itself: all-gather of one step's mel shards (what every step hides behind the next step's compute)
        mel_shard = torch.randn((B, L, cfg.n_mel_channels), device=dev)
        full = torch.empty((B * world, L, cfg.n_mel_channels), device=dev)
        for _ in range(3):
            dist.all_gather_into_tensor(full, mel_shard)
        sync_all()
        t0 = time.perf_counter()
        n_g = 10
        for _ in range(n_g):
            dist.all_gather_into_tensor(full, mel_shard)
        sync_all()
        tg = (time.perf_counter() - t0) / n_g
        recv = mel_shard.numel() * 4 * (world - 1)
        out["allgather"] = {"ms": tg * 1e3, "bytes_received_per_rank": recv, "GBps_received_per_rank": recv / tg / 1e9,
                            "xgmi_peak_GBps": XGMI_LINKS * XGMI_LINK_GBS, "frac_of_7_links": recv / tg / 1e9 / (XGMI_LINKS * XGMI_LINK_GBS),
                            "frac_of_links_in_use": recv / tg / 1e9 / (min(world - 1, XGMI_LINKS) * XGMI_LINK_GBS),
                            "note": "all_gather_into_tensor of the (B, L, 80) fp32 mel shards alone, blocking; in the timed steps it "
                                    "runs on a side stream under the next step's compute"}

    if world > 1 and pipe.gather:
        # the same steps with the exchange switched off: what the shards compute when nobody collects the mels.  Every rank
        # RECEIVES (N-1) x 62.9 MB per step in the timed region above, so its rate is capped by its xGMI ingress
        # (7 links x 76.8 GB/s / 320 B per frame = 1.7e9 frames/s per rank whatever N is); this line separates the two.
        steps2, warm2 = max(10, a.steps // 2), max(5, a.warmup // 2)
        pipe_ng = ShardedMelPipeline(net, world_size=world, gather=False)
        dt2, _, _ = timed_steps(pipe_ng, net, x, steps2, warm2, sync_all, 1 << 30, False)
        tt = torch.tensor([dt2], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt2 = float(tt.item())
        out["without_allgather"] = {"value": frames_per_step * steps2 / dt2, "ms_per_step": dt2 / steps2 * 1e3, "steps": steps2,
                                    "note": "same shards, same barrier/max-over-ranks timing, mels left on the rank that made them"}
        del pipe_ng


    def _forward_leg(config, b_global, t_ph, steps_, warm_, gather, lib=None, events=False, spot=True):
        """The same timed loop for another model / batch: `b_global` utterances sharded over the ranks (strong), D-const durations.
        -> dict(ms_per_step, value [real frames/s, whole job], per_gpu_batch, kernel_ms)"""
        cfg2 = CONFIGS[config]
        b_rank = -(-b_global // world)
        sd2 = synth_state_dict(cfg2, 1234)
        ctx = _lib.use_library(lib) if lib else None
        if ctx:
            ctx.__enter__()
        try:
            net2 = make_net(cfg2, sd2, dev)
            ids2, mask2 = synth_phonemes(b_rank, t_ph, 4321 + rank)
            l2 = t_ph * a.dur
            x2 = {"phoneme": torch.from_numpy(ids2).to(dev), "phoneme_mask": torch.from_numpy(mask2).to(dev),
                  "duration_forced": torch.full((b_rank, t_ph), a.dur, dtype=torch.int32, device=dev), "max_mel_len": l2,
                  "max_mel_len_exact": True}
            pipe2 = ShardedMelPipeline(net2, world_size=world, gather=(gather and world > 1))
            dt2, dec2, n2 = timed_steps(pipe2, net2, x2, steps_, warm_, sync_all, a.event_every if events else 1 << 30, False)
            if world > 1:
                t2 = torch.tensor([dt2], dtype=torch.float64, device=dev)
                dist.all_reduce(t2, op=dist.ReduceOp.MAX)
                dt2 = float(t2.item())
            spot2 = parity_spot(net2, x2, cfg2, sd2) if (spot and rank == 0) else None     # (inside the library context: the build that was timed)
        finally:
            if ctx:
                ctx.__exit__(None, None, None)
        res = {"ms_per_step": dt2 / steps_ * 1e3, "value": b_global * l2 * steps_ / dt2, "steps": steps_, "global_batch": b_global,
               "per_gpu_batch": b_rank, "phonemes": t_ph, "frames_per_step": b_global * l2}
        if events and n2:
            f2 = DECODER_WORK[config][0]
            res.update(kernel_ms=dec2, kernel_ms_samples=n2, achieved_tflops=f2 * b_rank * l2 / (dec2 * 1e-3) / 1e12)
        if spot2 is not None:
            res = gate(res, spot2)
        return res, net2, x2, cfg2, sd2

    if world > 1 and not a.no_extras and a.config == "tiny" and a.scaling == "weak":
        # north_star asks for both: the headline above is weak scaling (B per GPU fixed); here the headline's GLOBAL batch is split
        # over the ranks (32 utterances per GPU at N = 8: bounded by the encoder side's latency chain), and BASELINE configs[3]
        # (base ES, B = 512 x T = 256 over the node).  Collectives inside: every rank runs them or none does.
        for key, (c2, b2, t2) in (("strong", ("tiny", DEFAULT_BATCH["tiny"], DEFAULT_PHONEMES["tiny"])),
                                  ("base_strong", ("base", DEFAULT_BATCH["base"], DEFAULT_PHONEMES["base"]))):
            ok = torch.ones(1, device=dev)
            try:
                res, n2_, x2_, _, _ = _forward_leg(c2, b2, t2, max(10, a.steps // 2), max(5, a.warmup // 2), gather=pipe.gather)
                del n2_, x2_
            except Exception as e:          # noqa: BLE001
                res, ok = {"error": repr(e)}, torch.zeros(1, device=dev)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)          # a rank-local failure must not leave the others in a collective
            if float(ok.item()) == 0.0 and "error" not in res:
                res = {"error": "another rank failed"}
            res["note"] = (f"{c2} ES, GLOBAL batch {b2} x T={t2} sharded over {world} ranks (strong scaling), D-const {a.dur}, "
                           "mel all-gather one step behind; frames/s counts the real utterances")
            out[key] = res

    def _model_leg(name):
        # ---- BASELINE configs[2] / configs[3] on ONE GPU (the 8-GPU run of configs[3] shards exactly this batch): small ES B = 256 x
        # T = 256, base ES B = 512 x T = 256
        steps2, warm2 = max(10, a.steps // 5), max(3, a.warmup // 4)
        res, netb, xb, cfgb, sdb = _forward_leg(name, DEFAULT_BATCH[name], DEFAULT_PHONEMES[name], steps2, warm2, gather=False, events=True)
        fb = DECODER_WORK[name][0]
        bb, tb_ = DEFAULT_BATCH[name], DEFAULT_PHONEMES[name]
        lb = tb_ * a.dur
        if "achieved_tflops" in res:
            res["roofline"] = {"kernel": "mel_decoder_kernel<256,5,8>", "traffic_ratio": None, "bound": "mfma", "achieved": res["achieved_tflops"], "peak": peak_tf,
                               "unit": "TFLOP/s", "frac": res["achieved_tflops"] / peak_tf, "algorithmic_flops_per_frame": fb}
            tr_b, tr_bsrc = trace_kernel_us(name)
            if tr_b:
                res["roofline"].update(trace_kernel_us=tr_b, trace_source=tr_bsrc, frac_trace=fb * bb * lb / (tr_b * 1e-6) / 1e12 / peak_tf)
            trf, trf_src, _ = pmc_traffic(name, bb, tb_, a.dur)
            if trf:
                eb = 4.0 * cfgb.n_mel_channels + 4.0 * cfgb.dx2 / a.dur       # h0 rows (dx2 floats per phoneme) in, mel rows out
                res["roofline"].update(traffic=trf, traffic_source=trf_src, traffic_ratio=trf / (eb * bb * lb))
        try:
            ck = decoder_clock(netb, xb, dev, steps=5)
            if ck and "roofline" in res:
                ck["frac_at_clock"] = res["roofline"]["frac"] * 2.4 / ck["shader_ghz"]
                res["roofline"]["clock"] = ck
            res["encoder_side"] = encoder_side(netb, xb, dev, cfgb, bb, tb_, peak_tf, steps=10)
        except Exception as e:                     # noqa: BLE001
            res["encoder_side"] = {"error": repr(e)}
        # D-rand on this model
        rng = np.random.default_rng(1234)
        d_rand = rng.integers(1, 12, size=(bb, tb_)).astype(np.int32)
        l_rand = int(d_rand.sum(1).max())
        xr = {"phoneme": xb["phoneme"], "phoneme_mask": xb["phoneme_mask"], "duration_forced": torch.from_numpy(d_rand).to(dev),
              "max_mel_len": l_rand}
        with torch.no_grad():
            for _ in range(3):
                netb(xr)
            tr_ = float("inf")
            for _ in range(3):     # best of three blocks: one allocator stall inside a 10-step block once read 8.0 ms for a 1.7 ms step
                torch.cuda.synchronize(dev)
                t0 = time.perf_counter()
                for _ in range(10):
                    _, len_r, _ = netb(xr)
                torch.cuda.synchronize(dev)
                tr_ = min(tr_, (time.perf_counter() - t0) / 10)
        res["d_rand"] = gate({"ms_per_step": tr_ * 1e3, "valid_frames_per_step": int(d_rand.sum()), "padded_length": l_rand,
                              "value": int(d_rand.sum()) / tr_, "mel_len_matches": bool(np.array_equal(len_r.cpu().numpy(), d_rand.sum(1)))},
                             parity_spot(netb, xr, cfgb, sdb, sel=[int(d_rand.sum(1).argmin()), int(d_rand.sum(1).argmax())]))
        del netb
        if not a.exact_fp32 and os.path.exists(fp32_lib):
            r32, n32_, _, _, _ = _forward_leg(name, bb, tb_, max(5, steps2 // 2), 2, gather=False, lib=fp32_lib, events=True)
            del n32_
            res["exact_fp32"] = {k: r32[k] for k in ("ms_per_step", "value", "steps", "kernel_ms", "achieved_tflops", "parity_spot", "error") if k in r32}
            if "achieved_tflops" in r32:
                res["exact_fp32"]["frac_of_157.3"] = r32["achieved_tflops"] / FP32_PEAK_TFLOPS
        res["note"] = (f"BASELINE configs[{2 if name == 'small' else 3}] on one GPU: {name} ES ({sum(v.size for v in sdb.values())} params), "
                       f"B={bb} T={tb_} D-const, full Phoneme2Mel forward")
        out[name] = res

    def _single_gpu_extras():
        steps2, warm2 = max(10, a.steps // 2), max(5, a.warmup // 2)
        # ---- decoder only: MelDecoder.forward on frame-rate features ~ N(0,1), in-kernel proj stage
        g = torch.Generator(device=dev).manual_seed(7)
        feats = torch.randn((B, L, cfg.d4), device=dev, generator=g)
        with torch.no_grad():
            for _ in range(warm2):
                net.decoder(feats)
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            for _ in range(steps2):
                net.decoder(feats)
            torch.cuda.synchronize(dev)
            td = (time.perf_counter() - t0) / steps2
        spot_d = parity_spot(net, None, cfg, sd, decoder_feats=feats)
        del feats
        out["decoder_only"] = gate({"frames_per_s": B * L / td, "ms": td * 1e3, "steps": steps2,
                               "achieved_tflops": flops * B * L / td / 1e12, "frac": flops * B * L / td / 1e12 / peak_tf,
                               "hbm_frac": nbytes * B * L / td / 1e9 / HBM_PEAK_GBS,
                               "note": f"MelDecoder.forward alone, features ~ N(0,1) (B={B}, L={L}, {cfg.d4}) resident in HBM, "
                                       "wall clock over back-to-back launches (includes the proj stage the full forward runs at phoneme rate)"}, spot_d)
        # ---- the same full step with every contraction on the exact-fp32 MFMA instruction
        if not a.exact_fp32 and os.path.exists(fp32_lib):
            with _lib.use_library(fp32_lib):
                net32 = make_net(cfg, sd, dev)
                pipe32 = ShardedMelPipeline(net32, world_size=1, gather=False)
                dt32, dec32, n32 = timed_steps(pipe32, net32, x, steps2, warm2, sync_all, a.event_every, False)
                cfg32 = _lib.load().esmi_build_config().decode()
                spot32 = parity_spot(net32, x, cfg, sd)
            ach32 = flops * B * L / (dec32 * 1e-3) / 1e12
            out["exact_fp32"] = gate({"ms_per_step": dt32 / steps2 * 1e3, "value": frames_per_step * steps2 / dt32, "steps": steps2,
                                 "kernel_ms": dec32, "kernel_ms_samples": n32, "achieved_tflops": ach32,
                                 "frac_of_157.3": ach32 / FP32_PEAK_TFLOPS, "build_config": cfg32,
                                 "library": "efficientspeech_amd/libesmi_fp32mfma.so"}, spot32)
            del net32, pipe32
        # ---- the same forward written with stock PyTorch-ROCm operators (tests/torch_mirror.py: MIOpen / hipBLASLt convolutions,
        # torch softmax / layer_norm / repeat_interleave -- the reference's structure incl. its per-utterance upsampling loop), on
        # the same inputs and weights: what `--infer-device cuda` of the reference costs on this GPU, next to the CPU baseline
        try:
            from tests import torch_mirror as _mirror
            with torch.no_grad():
                m_mel = _mirror.eval_forward(net, x)[0]
                h_mel = net(x)[0]
                per_utt = (m_mel - h_mel).abs().amax(dim=(1, 2))
                n_flip = int((per_utt > 1e-3).sum())         # a predicted pitch / energy within 1e-7 of a bucket edge lands in the other
                diff = float(per_utt[per_utt <= 1e-3].max()) if n_flip < per_utt.numel() else float("nan")   # bucket: a discrete flip
                for _ in range(2):
                    _mirror.eval_forward(net, x)
                torch.cuda.synchronize(dev)
                t0 = time.perf_counter()
                n_m = 5
                for _ in range(n_m):
                    _mirror.eval_forward(net, x)
                torch.cuda.synchronize(dev)
                tm = (time.perf_counter() - t0) / n_m
            out["pytorch_rocm_ops"] = {"ms_per_step": tm * 1e3, "value": B * L / tm, "speedup_of_this_path": tm / (dt / a.steps),
                                       "max_abs_diff_vs_this_path": diff, "utterances_with_a_bucket_edge_flip": n_flip, "steps": n_m,
                                       "note": "same weights, same batch, same GPU; stock PyTorch-ROCm ops (tests/torch_mirror.py)"}
            del m_mel, h_mel
        except Exception as e:                     # noqa: BLE001
            out["pytorch_rocm_ops"] = {"error": repr(e)}
        # ---- BASELINE configs[0] shape on the GPU: one utterance (the 31-phoneme fox sentence), predicted durations replaced by
        # D-const (random-init predictors give 0), synchronous calls: what demo.py's loop sees per sentence
        fox = torch.tensor([FOX_IDS], dtype=torch.int32, device=dev)
        xf = {"phoneme": fox, "duration_forced": torch.full((1, len(FOX_IDS)), a.dur, dtype=torch.int32, device=dev)}
        with torch.no_grad():
            for _ in range(5):
                net(xf)
            torch.cuda.synchronize(dev)
            ts_ = []
            for _ in range(50):
                t0 = time.perf_counter()
                net(xf)
                torch.cuda.synchronize(dev)
                ts_.append(time.perf_counter() - t0)
        out["b1_fox_gpu"] = {"latency_ms_median": float(np.median(ts_) * 1e3), "frames": len(FOX_IDS) * a.dur,
                             "mRTF": len(FOX_IDS) * a.dur * 256 / 22050 / float(np.median(ts_)),
                             "note": "Phoneme2Mel.forward, B=1, T=31, synchronised after every call (host enqueue + kernels)"}
        # ---- robustness workload D-rand: durations uniform in [1, 11] (seed 1234, mean 6), ragged mel lengths, the
        # padded length derived on the device (no caller-vouched L): frame -> phoneme search, padding frames and the final mask all
        # take part.  Valid frames only are counted.
        try:
            rng = np.random.default_rng(1234)
            d_rand = rng.integers(1, 12, size=(B, T)).astype(np.int32)
            L_rand = int(d_rand.sum(1).max())
            xr = {"phoneme": x["phoneme"], "phoneme_mask": x["phoneme_mask"], "duration_forced": torch.from_numpy(d_rand).to(dev),
                  "max_mel_len": L_rand}
            with torch.no_grad():
                for _ in range(5):
                    net(xr)
                torch.cuda.synchronize(dev)
                t0 = time.perf_counter()
                n_r = 30
                for _ in range(n_r):
                    mel_r, len_r, _ = net(xr)
                torch.cuda.synchronize(dev)
                tr_ = (time.perf_counter() - t0) / n_r
            valid = int(d_rand.sum())
            out["d_rand"] = gate({"ms_per_step": tr_ * 1e3, "valid_frames_per_step": valid, "padded_length": L_rand,
                             "value": valid / tr_, "padding_fraction": 1.0 - valid / float(B * L_rand),
                             "mel_len_matches": bool(np.array_equal(len_r.cpu().numpy(), d_rand.sum(1))),
                             "note": "SURVEY 8d robustness workload: durations U[1, 11] seed 1234, same phoneme batch; frames/s counts "
                                     "valid frames only (windows that are all padding are skipped by the decoder, partly padded ones "
                                     "are computed as the reference computes them)"},
                                 parity_spot(net, xr, cfg, sd, sel=[int(d_rand.sum(1).argmin()), int(d_rand.sum(1).argmax())]))
        except Exception as e:                     # noqa: BLE001
            out["d_rand"] = {"error": repr(e)}
        # ---- the step after the path: HiFi-GAN v2 generator on the mel the forward just produced
        if not a.exact_fp32:
            from efficientspeech_amd.hifigan import HIFIGAN_CONFIGS, Generator, synth_hifigan_state_dict, flops_per_mel_frame
            hcfg = HIFIGAN_CONFIGS["v2"]
            voc = Generator(hcfg)
            voc.load_state_dict({k: torch.from_numpy(v) for k, v in synth_hifigan_state_dict(hcfg, 1234).items()})
            voc = voc.to(dev).eval()
            vb = min(B, 32)
            with torch.no_grad():
                mel_v = net(x)[0][:vb].contiguous()                      # (vb, L, 80): the acoustic model's own output
                for _ in range(2):
                    wav = voc(mel_v.transpose(1, 2))
                torch.cuda.synchronize(dev)
                t0 = time.perf_counter()
                for _ in range(5):
                    wav = voc(mel_v.transpose(1, 2))
                torch.cuda.synchronize(dev)
                tv = (time.perf_counter() - t0) / 5
                for _ in range(3):
                    voc(net(xf)[0].transpose(1, 2))
                torch.cuda.synchronize(dev)
                tw = []
                for _ in range(20):
                    t0 = time.perf_counter()
                    wav1 = voc(net(xf)[0].transpose(1, 2))
                    torch.cuda.synchronize(dev)
                    tw.append(time.perf_counter() - t0)
            out["b1_fox_gpu"]["text_to_wav_ms_median"] = float(np.median(tw) * 1e3)
            out["b1_fox_gpu"]["text_to_wav_x_realtime"] = wav1.shape[-1] / 22050.0 / float(np.median(tw))
            vf = flops_per_mel_frame(hcfg) * vb * L / tv / 1e12
            voc_fps, am_fps = vb * L / tv, out["value"]
            out["vocoder"] = {"workload": f"hifigan_v2 generator, B={vb} x L={L} mel frames -> {L * hcfg.hop} samples each",
                              "ms": tv * 1e3, "mel_frames_per_s": voc_fps, "x_realtime": voc_fps * hcfg.hop / 22050.0,
                              "achieved_tflops": vf, "frac_of_833": vf / (F16_PEAK_TFLOPS / 3.0),
                              "text_to_wav_frames_per_s": 1.0 / (1.0 / am_fps + 1.0 / voc_fps),
                              "finite": bool(torch.isfinite(wav).all()),
                              "note": "synthetic seeded generator weights; 20 launches (one per ResBlock, csrc/hifigan_resblock.h); "
                                      "text_to_wav = acoustic model and vocoder back to back on one GPU"}
            del voc, wav, mel_v
    if rank == 0 and world == 1 and not a.no_extras:
        try:                                   # an optional leg must never cost the headline line
            _single_gpu_extras()
        except Exception as e:                 # noqa: BLE001
            out["extras_error"] = repr(e)
        if a.config == "tiny" and not a.exact_fp32:
            for leg in ("small", "base"):
                try:
                    _model_leg(leg)
                except Exception as e:         # noqa: BLE001
                    out[leg] = {"error": repr(e)}
    def _train_leg():
        # ---- BASELINE configs[4]: the training step (forward + loss + backward + AdamW; N > 1: one RCCL all-reduce of the flat
        # gradient buffer per step), tiny-ES-shaped synthetic teacher-forced batch of the reference's default batch size per GPU
        from efficientspeech_amd import train as _train
        tb, tt = 128, 100
        tnet = make_net(cfg, sd, dev).train()
        tx, ty = _train.synthetic_batch(tb, tt, a.dur, dev, seed=77 + rank)
        # N = 1: hipGraph replay of the whole step (194 launches; the replay takes the host out of the loop).  N > 1: eager launches --
        # `TrainStep` can replay forward + loss + backward around an eager all-reduce (tested with two gloo ranks on one device), but
        # graph capture next to a live RCCL communicator has never run on hardware, and this leg must not put the headline line at risk
        ts = _train.TrainStep(tnet, world_size=world, graph=(world == 1))
        for _ in range(3):
            tl0 = ts.step(tx, ty)
        sync_all()
        t0 = time.perf_counter()
        n_tr = 10
        for _ in range(n_tr):
            tl1 = ts.step(tx, ty)
        sync_all()
        ttr = torch.tensor([(time.perf_counter() - t0) / n_tr], dtype=torch.float64, device=dev)
        if world > 1:
            dist.all_reduce(ttr, op=dist.ReduceOp.MAX)
        ttr = float(ttr.item())
        # algorithmic work of one step: the forward's matmul-class FLOPs (decoder per frame + encoder side per phoneme) once
        # forward, twice backward (data gradient + weight gradient of every contraction); the decoder runs at frame rate in train mode
        enc_flops = {"tiny": 262_336, "small": 737_664, "base": 4_489_984}[a.config]
        step_flops = 3.0 * tb * tt * (a.dur * DECODER_WORK[a.config][0] + enc_flops)
        t16 = t_eager = None
        if world == 1:
            tse = _train.TrainStep(make_net(cfg, sd, dev).train())
            for _ in range(3):
                tse.step(tx, ty)
            sync_all()
            t0 = time.perf_counter()
            for _ in range(n_tr):
                tse.step(tx, ty)
            sync_all()
            t_eager = (time.perf_counter() - t0) / n_tr
            del tse
            ts16 = _train.TrainStep(make_net(cfg, sd, dev).train(), precision=16, init_scale=2048.0, graph=True)
            for _ in range(3):
                ts16.step(tx, ty)
            sync_all()
            t0 = time.perf_counter()
            for _ in range(n_tr):
                ts16.step(tx, ty)
            sync_all()
            t16 = (time.perf_counter() - t0) / n_tr
            del ts16
        out["train_step"] = {"ms_per_step": ttr * 1e3, "mel_frames_per_s": tb * tt * a.dur * world / ttr, "steps": n_tr,
                             "per_gpu_batch": tb, "phonemes": tt, "frames_per_utterance": tt * a.dur,
                             "precision16_ms_per_step": None if t16 is None else t16 * 1e3,
                             "eager_ms_per_step": None if t_eager is None else t_eager * 1e3,
                             "launch": "hipGraph replay per batch shape" if world == 1 else "eager launches; one all-reduce of the flat gradient buffer per step",
                             "roofline": {"bound": "mfma", "algorithmic_flops_per_step": step_flops,
                                          "achieved": step_flops / ttr / 1e12, "peak": F16_PEAK_TFLOPS / 3.0, "unit": "TFLOP/s",
                                          "frac": step_flops / ttr / 1e12 / (F16_PEAK_TFLOPS / 3.0),
                                          "note": "whole step (194 launches, no dominant kernel; per-kernel table: profiles/r05_f_train_kernel_stats.md): 3 x the forward's "
                                                  "contraction FLOPs over the step time against the split-f16 bound; the step is "
                                                  "bound by launch count and by the partial-sum traffic of the deterministic "
                                                  "weight-gradient reductions, not by the matrix pipe (HISTORY.md 3.6)"},
                             "loss_first_last": [float(tl0[4]), float(tl1[4])],
                             "allreduce_bytes_per_step": int(ts.flat.grad.numel() * 4) if world > 1 else 0,
                             "note": "SURVEY 8f-2 / BASELINE configs[4]: train=True forward, masked L1 + 3 MSE loss, backward, AdamW "
                                     "(efficientspeech_amd/train.py; esmi_train_* kernels); synthetic teacher-forced batch, D-const "
                                     "durations; data-parallel: one all-reduce of the flat fp32 gradient buffer"}
        del tnet, ts
    if not a.no_extras and not a.exact_fp32:
        if world == 1:
            try:
                _train_leg()
            except Exception as e:             # noqa: BLE001
                out["train_step_error"] = repr(e)
        else:
            # collectives inside.  A failure in the step CONSTRUCTION is the same on every rank; a rank-local one inside the timed
            # loop (OOM, a RCCL error) would leave the others waiting in an all-reduce -- so the leg runs under a watchdog thread
            # that ends the process group's wait by aborting this rank's process after a generous bound instead of hanging the
            # benchmark, and the ranks agree on an ok-flag before anyone goes on
            import threading
            done_evt = threading.Event()

            def _watchdog():
                if not done_evt.wait(600.0):
                    if rank == 0:
                        out["train_step_error"] = "train leg timed out (a rank failed inside a collective?)"
                        print(json.dumps(out), flush=True)
                    os._exit(3)
            threading.Thread(target=_watchdog, daemon=True).start()
            ok = torch.ones(1, device=dev)
            try:
                _train_leg()
            except Exception as e:             # noqa: BLE001
                out["train_step_error"] = repr(e)
                ok.zero_()
            try:
                dist.all_reduce(ok, op=dist.ReduceOp.MIN)
                if float(ok.item()) == 0.0 and "train_step_error" not in out:
                    out["train_step_error"] = "another rank failed"
            finally:
                done_evt.set()
    if rank == 0:
        if world == 1 and not a.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(cfg, sd, T, a.dur)
            except Exception as e:                 # noqa: BLE001  (e.g. no C compiler on the box: the headline line still prints)
                out["cpu_baseline"] = {"error": repr(e)}
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
