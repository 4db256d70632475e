"""Batch-sharded multi-GPU inference: one process per GPU, utterances split across ranks, one
all-gather of the mel shards at the end (RCCL over xGMI).

The reference has no multi-GPU inference; utterances are independent (no cross-batch op anywhere on
the path), so the only exchange is
  1. all_reduce(MAX) of the local padded length, so that every rank pads -- and zero-pads its
     convolutions -- at the same L as a single-GPU run of the whole batch would (skipped when the
     caller supplies `max_mel_len`), and
  2. all_gather of mel (B/G, L, 80) fp32 and mel_len (B/G,) int32.
The gathered result is bit-identical to the single-GPU result (tests/test_sharded.py).

`ShardedMelPipeline` overlaps step i's all-gather (side stream) with step i+1's compute: on xGMI the
gather of a tiny-ES batch costs about as much as computing it, so serialising them would halve the
throughput.
"""
import torch
import torch.distributed as dist



def _encode_with_head(net, x, need_lmax=True):
    """Encoder side of the forward, including the decoder's row-wise first stage at phoneme rate when the fused kernel
    can produce it (networks.MelDecoder._head)."""
    from . import networks
    lib, stream = networks._runtime(net.decoder.mel_linear.weight)
    return net.encoder._encode(x, train=False, need_lmax=need_lmax, head=net.decoder._head(lib, stream))


def shard_rows(B, rank, world):
    """Rows [lo, hi) of a batch of B utterances that rank `rank` of `world` owns, and the common shard size: contiguous
    shards of ceil(B / world) utterances -- the last ranks own fewer (possibly none) when world does not divide B."""
    per = -(-B // world)
    lo = min(rank * per, B)
    return lo, min(lo + per, B), per


def shard_batch(x, rank, world):
    """Contiguous utterance shard of every batch-leading tensor in the input dict.  A ragged last batch (world does not divide
    B) is padded, not refused: every rank gets ceil(B / world) utterances (equal shards are what all_gather_into_tensor needs),
    the missing ones being copies of the shard's last utterance -- or of the batch's last one for a rank that owns none.  The
    copies only cost their compute: they are duplicates of real utterances, so they cannot change the batch's padded length, and
    `unpad_gathered` drops their rows (they sit behind the B real ones in the gathered tensors)."""
    B = x["phoneme"].shape[0]
    lo, hi, per = shard_rows(B, rank, world)
    out = {}
    for k, v in x.items():
        if torch.is_tensor(v) and v.dim() > 0 and v.shape[0] == B:
            part = v[lo:hi]
            if hi - lo < per:
                fill = (part[-1:] if hi > lo else v[B - 1:B]).expand((per - (hi - lo),) + tuple(v.shape[1:]))
                part = torch.cat([part, fill])
            out[k] = part
        else:
            out[k] = v
    return out


def unpad_gathered(t, B):
    """Rows of the real utterances of an all-gathered tensor: with contiguous shards the padding copies are the rows behind B."""
    return t if t.shape[0] == B else t[:B]


def _masked_path_inputs(x):
    """A 1-utterance shard of a B > 1 batch must still take the masked (B > 1) code path of the reference
    (networks.py:338: the mask is dropped for B == 1): duplicate the utterance; the caller drops the copy afterwards.
    -> (x, duplicated?)"""
    if x["phoneme"].shape[0] != 1:
        return x, False
    return {k: (torch.cat([v, v]) if torch.is_tensor(v) and v.dim() > 0 and v.shape[0] == 1 else v) for k, v in x.items()}, True


def _exact_len(x):
    """The caller states that x['max_mel_len'] IS the batch's padded length (e.g. forced durations known on the host):
    the 4-byte MAX all-reduce of the padded length is then unnecessary (extension key `max_mel_len_exact`)."""
    return "max_mel_len" in x and bool(x.get("max_mel_len_exact", False))


def sharded_forward(net, x_full, group=None):
    """Run Phoneme2Mel inference on this rank's shard of `x_full`, return the full
    (mel (B,L,80), mel_len (B,), duration (B,T,1)) on every rank."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    if world == 1:
        return net(x_full)
    x = shard_batch(x_full, rank, world)
    if "phoneme_mask" not in x_full and x_full["phoneme"].shape[0] > 1:
        raise KeyError("phoneme_mask")                     # same contract as the reference for B > 1
    x, dup = _masked_path_inputs(x)
    # The padded length L is a property of the WHOLE batch (the reference zero-pads its convolutions at the batch max), so the
    # local maxima are MAX-reduced on the device between the encoder side and the decoder -- unless the caller vouches for
    # `max_mel_len` (max_mel_len_exact).  With a caller-supplied bound the output is allocated at it and no host sync happens.
    st = net._launch(x, stage=1)
    if st.lmax is not None:
        dist.all_reduce(st.lmax, op=dist.ReduceOp.MAX, group=group)
    st = net._launch(None, stage=2, state=st)
    mel, mel_len, dur = st.mel, st.mel_len, st.duration
    if dup:
        mel, mel_len, dur = mel[:1], mel_len[:1], dur[:1]
    outs = []
    for t in (mel.contiguous(), mel_len.contiguous(), dur.contiguous()):
        full = torch.empty((t.shape[0] * world,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
        dist.all_gather_into_tensor(full, t, group=group)
        outs.append(unpad_gathered(full, x_full["phoneme"].shape[0]))
    return tuple(outs)


class GraphedForward:
    """The inference forward of one fixed shape as two hipGraphs (encoder side, mel decoder) so that a step costs
    two graph launches instead of ~10 kernel launches + the Python between them.  Needs the `max_mel_len` bound
    (static output size, no host sync).  Between the graphs the caller may MAX-reduce `lmax` across ranks.
    Inputs are copied into static buffers; the mel output alternates between `nbuf` buffers so that a result can
    still be in flight (all-gather on a side stream) while the next step computes."""

    def __init__(self, net, x, nbuf=3, warmup=3):
        assert "max_mel_len" in x, "graph replay needs a static output length (x['max_mel_len'])"
        self.net = net
        self.L = int(x["max_mel_len"])
        self.x = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in x.items()}
        B = self.x["phoneme"].shape[0]
        self.apply_mask = ("phoneme_mask" in self.x) and B > 1
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side), torch.no_grad():       # warm-up off the capture: weight packing, attributes
            for _ in range(warmup):
                enc = _encode_with_head(net, self.x)
                net.decoder._fused(enc["feat"], enc["cum"], enc["mel_len"], enc["lmax"], self.L, self.apply_mask, self.L, h0=enc["h0"])
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.g_enc = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.g_enc), torch.no_grad():
            self.enc = _encode_with_head(net, self.x)
        self.g_dec, self.mels = [], []
        for _ in range(nbuf):
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g), torch.no_grad():
                mel = net.decoder._fused(self.enc["feat"], self.enc["cum"], self.enc["mel_len"], self.enc["lmax"],
                                         self.L, self.apply_mask, self.L, h0=self.enc["h0"])
            self.g_dec.append(g)
            self.mels.append(mel)
        self.i = 0

    def load(self, x):
        for k, v in x.items():
            if torch.is_tensor(v) and v.data_ptr() != self.x[k].data_ptr():
                self.x[k].copy_(v, non_blocking=True)

    def encode(self):
        self.g_enc.replay()
        return self.enc

    def decode(self):
        j = self.i % len(self.g_dec)
        self.i += 1
        self.g_dec[j].replay()
        return self.mels[j]


class ShardedMelPipeline:
    """Steady-state serving loop: step(x) computes this rank's shard; its all-gather runs on a side
    stream while the next step computes.  With use_graph (and a `max_mel_len` bound) the compute is two
    hipGraph replays per step."""

    def __init__(self, net, world_size=1, gather=True, group=None, depth=2, use_graph=False, two_stream=False):
        self.net, self.world, self.group = net, world_size, group
        self.gather = gather and world_size > 1
        self.depth = depth
        self.comm = None
        self.inflight = []       # (done_event, gathered mel, gathered mel_len)
        self.last = None
        self.use_graph = use_graph
        self.graphed = None
        self.dec_events = None   # bench.py: list collecting (start, end) events around the decoder launch
        # two_stream: the encoder side of step i+1 runs on its own stream while the mel decoder of step i is
        # still running -- the encoder kernels are latency-bound one-wave chains that fit beside the decoder's
        # workgroups on every CU (needs the <=168-VGPR builds of both).
        self.two_stream = two_stream
        self.s_enc = self.s_dec = None
        self.last_ready = None   # event after which `last` may be read (None: it was produced on the caller's stream)

    def _compute(self, x):
        """-> (mel, mel_len) of this rank's shard; global padded length MAX-reduced when world > 1."""
        if self.use_graph and "max_mel_len" in x:
            if self.graphed is None:
                self.graphed = GraphedForward(self.net, x, nbuf=self.depth + 1)
            g = self.graphed
            g.load(x)
            enc = g.encode()
            if self.world > 1:
                dist.all_reduce(enc["lmax"], op=dist.ReduceOp.MAX, group=self.group)
            if self.dec_events is not None:
                ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                ev[0].record()
                mel = g.decode()
                ev[1].record()
                self.dec_events.append(ev)
            else:
                mel = g.decode()
            return mel, enc["mel_len"]
        if self.world == 1:
            mel, mel_len, _ = self.net(x)
            return mel, mel_len
        # global padded length: 4-byte MAX all-reduce on the compute stream (see sharded_forward) -- unless the caller
        # vouches for it (max_mel_len_exact): then the whole forward is one C-ABI call with nothing in between
        x, dup = _masked_path_inputs(x)
        if _exact_len(x):
            st = self.net._launch(x)
        else:
            st = self.net._launch(x, stage=1)
            dist.all_reduce(st.lmax, op=dist.ReduceOp.MAX, group=self.group)
            st = self.net._launch(None, stage=2, state=st)
        mel, mel_len = st.mel, st.mel_len
        if dup:
            mel, mel_len = mel[:1], mel_len[:1]
        return mel, mel_len

    def _compute_two_stream(self, x):
        """Encoder side on s_enc, decoder on s_dec.  -> (mel, mel_len, done): `done` is recorded on s_dec after the decoder;
        the CALLER's stream is deliberately not made to wait for it here (it would serialise step i+1's encoder side behind
        step i's decoder through the caller's stream): consumers order themselves after `done` (step() does for the
        all-gather; `wait_last()` for code that reads `last` on the current stream)."""
        dev = x["phoneme"].device
        if self.s_enc is None:
            # the encoder side is three short latency-bound launches: high priority, so that its few workgroups take the
            # first slot a finishing decoder workgroup frees on each CU instead of queueing behind the decoder's own
            import os
            prio = int(os.environ.get("ESMI_ENC_STREAM_PRIORITY", "-1"))
            self.s_enc, self.s_dec = torch.cuda.Stream(device=dev, priority=prio), torch.cuda.Stream(device=dev)
        cur = torch.cuda.current_stream(dev)
        self.s_enc.wait_stream(cur)                       # inputs produced on the caller's stream
        x, dup = _masked_path_inputs(x) if self.world > 1 else (x, False)
        with torch.cuda.stream(self.s_enc):
            st = self.net._launch(x, stage=1)
            if self.world > 1 and st.lmax is not None and not _exact_len(x):
                dist.all_reduce(st.lmax, op=dist.ReduceOp.MAX, group=self.group)
            ready = torch.cuda.Event()
            ready.record()
        if st.L_out is None:
            ready.synchronize()                           # no caller-supplied output length: the host needs the padded length
        with torch.cuda.stream(self.s_dec):
            self.s_dec.wait_event(ready)
            # every tensor the decoder reads was allocated on s_enc: tell the allocator s_dec uses it too, or step i+1's
            # encoder side (running while this decoder still reads) may be handed the same blocks (h0 was missing;
            # everything the encoder side hands over now lives in ONE arena)
            for t in (st.arena, st.mel_len, st.lmax, st.ids, st.m8, st.dur_t):
                if t is not None:
                    t.record_stream(self.s_dec)
            st = self.net._launch(None, stage=2, state=st)
            mel = st.mel
            done = torch.cuda.Event()
            done.record()
        mel_len = st.mel_len
        if dup:
            mel, mel_len = mel[:1], mel_len[:1]
        mel.record_stream(cur)                            # the caller will read it on its own stream (after `done`)
        mel_len.record_stream(cur)
        return mel, mel_len, done

    def wait_last(self, stream=None):
        """Order `stream` (default: the current one) after the producer of `last` (two-stream mode: the decoder stream)."""
        if self.last_ready is not None:
            (stream or torch.cuda.current_stream()).wait_event(self.last_ready)

    def step(self, x):
        done = None
        if self.two_stream and not (self.use_graph and "max_mel_len" in x):
            mel, mel_len, done = self._compute_two_stream(x)
        else:
            mel, mel_len = self._compute(x)
        self.last_ready = done
        if not self.gather:
            if done is not None:
                # bound the host's run-ahead in two-stream mode: every step's buffers are shared between two streams
                # (record_stream), so the allocator cannot hand them out again until the GPU has passed them -- a host 100 steps
                # ahead of a 9 ms step (base ES: 252 MB of mel per step) was found allocating fresh blocks for every step,
                # 22-36 ms per step instead of 9.1 (profiles/r03_probes/two_stream_runahead.md)
                self.inflight.append((done, None, None))
                while len(self.inflight) > self.depth + 1:
                    self.inflight.pop(0)[0].synchronize()
            self.last = (mel, mel_len)
            return self.last
        if self.comm is None:
            self.comm = torch.cuda.Stream(device=mel.device)
        while len(self.inflight) >= self.depth:            # bound queue depth / memory
            self.inflight.pop(0)[0].synchronize()
        if done is None:
            done = torch.cuda.Event()
            done.record()                                   # compute stream: mel is complete here
        if self.graphed is None:                            # graph buffers are static: nothing to protect
            mel.record_stream(self.comm)
            mel_len.record_stream(self.comm)
        with torch.cuda.stream(self.comm):
            self.comm.wait_event(done)
            full = torch.empty((mel.shape[0] * self.world,) + tuple(mel.shape[1:]), dtype=mel.dtype, device=mel.device)
            lens = torch.empty((mel_len.shape[0] * self.world,), dtype=mel_len.dtype, device=mel.device)
            dist.all_gather_into_tensor(full, mel, group=self.group)
            dist.all_gather_into_tensor(lens, mel_len, group=self.group)
            gathered = torch.cuda.Event()
            gathered.record()
        self.inflight.append((gathered, full, lens))
        self.last = (full, lens)
        self.last_ready = gathered
        return self.last

    def flush(self):
        if self.two_stream and self.s_dec is not None:
            self.s_enc.synchronize()
            self.s_dec.synchronize()
        for done, _, _ in self.inflight:
            done.synchronize()
        self.inflight.clear()
