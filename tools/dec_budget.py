#!/usr/bin/env python3
"""Development tool (GPU box, needs a -DESMI_DEC_TRACE build): where the time of mel_decoder_kernel goes, as a budget that SUMS to the
kernel's duration.

One workgroup in the middle of the launch stamps the shader clock (s_memtime) at every phase boundary of every wave (mel_decoder.h,
ESMI_STAMP) and the 100 MHz clock at its entry and end.  From that:
  * the workgroup's life in shader cycles, split by phase -- for every wave, and for the CRITICAL PATH (the wave that arrives last at
    each barrier sets the time of the phase in front of it; what the others spend behind it is barrier wait);
  * the clock it ran at (cycle span / 100 MHz span);
  * the launch: `slots` workgroups are resident at a time (2 per CU for dx2 = 128, 1 for dx2 = 256), so the kernel lasts
    ceil-ish(windows / slots) workgroup lives; what the event-timed duration has beyond windows / slots lives is the dispatch tail
    (the last, partly filled round) plus launch overhead.
usage: tools/dec_budget.py <lib.so built with -DESMI_DEC_TRACE> [--config tiny|small|base] [--md out.md]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
from efficientspeech_amd import CONFIGS, _lib, build_phoneme2mel, load_numpy_state_dict  # noqa: E402
from efficientspeech_amd.synth import synth_state_dict  # noqa: E402

lib = C.CDLL(os.path.abspath(sys.argv[1]))
_lib._LIB = _lib.bind(lib)
cname = sys.argv[sys.argv.index("--config") + 1] if "--config" in sys.argv else "tiny"
md_path = sys.argv[sys.argv.index("--md") + 1] if "--md" in sys.argv else None
cfg = CONFIGS[cname]
B, T, D = (256, 128, 6) if cname == "tiny" else (256 if cname == "small" else 512, 256, 6)
L = T * D
net = build_phoneme2mel(cfg)
load_numpy_state_dict(net, synth_state_dict(cfg))
net = net.cuda()
feat = torch.randn((B, T, cfg.d4), device="cuda")
cum = (torch.arange(1, T + 1, device="cuda", dtype=torch.int32) * D).repeat(B, 1).contiguous()
mel_len = torch.full((B,), L, dtype=torch.int32, device="cuda")
h0 = torch.randn((B, T, cfg.dx2), device="cuda")
lib.esmi_dev_set_trace.argtypes = [C.c_void_p]


def launch():
    net.decoder._fused(feat, cum, mel_len, None, L, True, L, h0=h0)


# ---- event-timed duration of the kernel (untraced launches: the trace pointer is NULL)
lib.esmi_dev_set_trace(None)
for _ in range(5):
    launch()
torch.cuda.synchronize()
ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(30)]
for a, b in ev:
    a.record(); launch(); b.record()
torch.cuda.synchronize()
kernel_us = float(np.median([a.elapsed_time(b) for a, b in ev])) * 1e3

# ---- one traced launch
tr = torch.zeros((8, 64), dtype=torch.int64, device="cuda")
lib.esmi_dev_set_trace(tr.data_ptr())
launch()
torch.cuda.synchronize()
lib.esmi_dev_set_trace(None)
t = tr.cpu().numpy()

n_layers = cfg.n_blocks * cfg.block_depth
nl = min(n_layers, 4)                       # 64 slots per wave: 3 + 12 * 4 + 2 stamps, two clock slots
phase = ["dw window load", "barrier", "dw compute + split + plane store", "barrier", "K loop (MFMA)", "bias + tanh (regs)", "barrier",
         "tanh store", "barrier", "LayerNorm", "barrier"]
#   stamps of a layer: 0 start, 1 window loaded, 2 barrier, 3 dw written, 4 barrier, 5 K loop issued, 6 (tanh on regs +) barrier,
#   7 tanh stored, 8 barrier, 9 LN done, 10 barrier.   diff k = stamp k+1 - stamp k:
#   0: window load issue, 1: barrier, 2: dw compute/write, 3: barrier, 4: K loop, 5: tanh_acc + barrier wait, 6: store, 7: barrier, 8: LN, 9: barrier
cats = {"depthwise + operand split": [0, 2], "K loop (MFMA + A-fragment reads)": [4], "bias + tanh + store": [5, 7], "LayerNorm": [9],
        "barrier wait": [1, 3, 6, 8, 10]}
W = t.shape[0]
entry, staged, first_done = t[:, 0], t[:, 1], t[:, 2]
lay0 = 3
layers_done, mel_k, end = t[:, 59], t[:, 60], t[:, 61]
life = float((end - entry).max())
rt_span = float((t[:, 63] - t[:, 62]).max())            # 100 MHz ticks
ghz = life / rt_span * 0.1 if rt_span > 0 else float("nan")

lines = []
P = lines.append
P(f"# mel_decoder_kernel time budget -- {cname} ES, B={B} T={T} D-const {D} (L={L}), one traced workgroup, 8 waves\n")
P(f"Event-timed kernel duration (median of 30 untraced launches of the same build): **{kernel_us:.1f} us**.  "
  f"Traced workgroup: life {life:.0f} shader cycles over {rt_span * 0.01:.2f} us = **{ghz:.2f} GHz**.\n")
if n_layers > nl:
    P(f"(the trace buffer holds 4 of the {n_layers} conv layers; per-layer figures are scaled to {n_layers} below)\n")

# per-wave category sums over the traced layers
per_wave = np.zeros((W, len(cats)))
for w in range(W):
    for l in range(nl):
        d = np.diff(t[w, lay0 + 12 * l: lay0 + 12 * l + 12]).astype(float)
        for ci, (_, idx) in enumerate(cats.items()):
            per_wave[w, ci] += d[idx].sum()
per_wave *= n_layers / nl
pro = (first_done - entry).astype(float)
epi = (end - layers_done).astype(float)
if n_layers > nl:   # layers beyond the trace buffer: the traced layers' category shares, scaled to the measured span of all layers
    span_all = (layers_done - first_done).astype(float)
    per_wave *= (span_all / np.maximum(per_wave.sum(1), 1.0))[:, None]
P("## Per wave (shader cycles, all conv layers)\n")
P("| wave | prologue (sources, h0 gather, first LN) | " + " | ".join(cats) + " | mel Linear + store | sum | life |")
P("|---:|---:|" + "---:|" * (len(cats) + 3))
for w in range(W):
    s_ = pro[w] + per_wave[w].sum() + epi[w]
    P(f"| {w} | {pro[w]:.0f} | " + " | ".join(f"{v:.0f}" for v in per_wave[w]) + f" | {epi[w]:.0f} | {s_:.0f} | {float(end[w] - entry[w]):.0f} |")
mean = per_wave.mean(0)
tot = pro.mean() + mean.sum() + epi.mean()
P("")
P("## Budget of the kernel's duration\n")
if cfg.dx2 <= 128:
    slots, keep = 2 * 256, 128 - 2 * cfg.halo
    windows = B * (-(-L // keep))
    rounds = windows / slots
else:   # the chunk walk: one workgroup per CU, every workgroup walks its utterance chunk by chunk; a "life" below is ONE chunk
    slots = 256
    sh = (cfg.decoder_kernel_size // 2) * cfg.block_depth
    chunks = -(-(L + cfg.halo - sh) // (128 - sh))
    windows = B * chunks
    rounds = windows / slots
us_per_cycle = 1e-3 / ghz
wg_us = life * us_per_cycle
P(f"{windows} windows / chunks, {slots} workgroups resident at a time = {rounds:.2f} lives back to back; one life = {wg_us:.2f} us "
  f"-> {rounds * wg_us:.1f} us if every slot were busy to the end.\n")
P("| component | cycles per workgroup (mean over waves) | share of a life | us of the kernel |")
P("|---|---:|---:|---:|")
rows = [("prologue: parameter staging, frame->phoneme search, h0 gather, first LayerNorm", pro.mean())] + \
       [(n, mean[i]) for i, n in enumerate(cats)] + [("mel Linear (K loop) + mel row stores", epi.mean())]
acc_us = 0.0
for n, c in rows:
    us = c / tot * rounds * wg_us
    acc_us += us
    P(f"| {n} | {c:.0f} | {100 * c / tot:.1f} % | {us:.1f} |")
tail = kernel_us - rounds * wg_us
P(f"| dispatch tail + launch (event-timed duration minus {rounds:.2f} lives; the last round fills {100 * (rounds % 1 or 1):.0f} % of the slots) | | | {tail:.1f} |")
P(f"| **sum** | {tot:.0f} | 100 % | **{acc_us + tail:.1f}** (= the event-timed {kernel_us:.1f} us) |")
P("")
P("## Prologue in detail (cycles from entry; waves 0 / 3 / 7)\n")
P("| wave | loads issued (scan row, parameters, pad rows) | scan row in LDS (barrier) | search done, sources staged (barrier) | h0 rows in the tile | barrier | first stage done (skip rows read, pad-row LN, barrier) |")
P("|---:|---:|---:|---:|---:|---:|---:|")
for w in (0, 3, 7):
    P(f"| {w} | " + " | ".join(str(int(t[w, k] - t[w, 0])) for k in (52, 53, 1, 54, 55, 2)) + " |")
P("")
P("## Layer by layer, waves 0 / 3 / 7 (cycles)\n")
for w in (0, 3, 7):
    P(f"wave {w}:")
    for l in range(nl):
        d = np.diff(t[w, lay0 + 12 * l: lay0 + 12 * l + 12])
        P("  layer %d: " % l + "  ".join(f"{n}={int(x)}" for n, x in zip(phase, d)))
out = "\n".join(lines)
print(out)
if md_path:
    open(md_path, "w").write(out + "\n")
