"""Length-bucketed batching of ragged inference requests.

The reference pads a batch to its longest utterance and -- a quirk kept on purpose -- lets padded key
positions take part in attention, so every padded phoneme is pure waste on the device AND changes the numbers of the
utterances it is batched with.  `BucketedSynthesizer` sorts requests by length and cuts them into batches of at most
`max_batch` utterances whose lengths fall into one bucket of `granularity` phonemes: granularity 1 never pads (only
equal-length utterances share a batch), a larger one trades a bounded amount of padding (< granularity phonemes per
utterance) for fuller batches.  Each batch is one `Phoneme2Mel` forward; results come back in request order.
"""
import numpy as np
import torch

from .networks import get_mask_from_lengths


class BucketedSynthesizer:
    def __init__(self, net, max_batch=256, granularity=8, pad_id=0):
        assert max_batch >= 1 and granularity >= 1
        self.net, self.max_batch, self.granularity, self.pad_id = net, int(max_batch), int(granularity), int(pad_id)

    def plan(self, lengths):
        """-> list of (request indices, padded length T) covering every request once, longest bucket first."""
        lengths = np.asarray(lengths, dtype=np.int64)
        order = np.argsort(-lengths, kind="stable")
        g = self.granularity
        batches, cur, cur_bucket = [], [], None
        for i in order:
            bucket = (int(lengths[i]) + g - 1) // g
            if cur and (bucket != cur_bucket or len(cur) == self.max_batch):
                batches.append((cur, int(lengths[cur[0]])))
                cur = []
            cur.append(int(i))
            cur_bucket = bucket
        if cur:
            batches.append((cur, int(lengths[cur[0]])))
        return batches

    @staticmethod
    def padding_waste(lengths, batches):
        """fraction of phoneme slots that are padding under `batches`"""
        lengths = np.asarray(lengths, dtype=np.int64)
        slots = sum(len(idx) * T for idx, T in batches)
        return 1.0 - float(lengths.sum()) / max(slots, 1)

    @torch.no_grad()
    def __call__(self, sequences, extra=None):
        """sequences: list of 1-D integer phoneme id sequences.  -> list (request order) of (mel (L_i, n_mel), duration (T_i,)).
        `extra(indices, T)` may return additional input-dict entries for a batch (e.g. forced durations)."""
        dev = self.net.decoder.mel_linear.weight.device
        lengths = [int(len(s)) for s in sequences]
        out = [None] * len(sequences)
        for idx, T in self.plan(lengths):
            ids = np.full((len(idx), T), self.pad_id, np.int32)
            for r, i in enumerate(idx):
                ids[r, :lengths[i]] = np.asarray(sequences[i], dtype=np.int32)
            x = {"phoneme": torch.from_numpy(ids).to(dev)}
            if len(idx) > 1:                                   # the reference's B == 1 path takes no mask (networks.py:338)
                x["phoneme_mask"] = get_mask_from_lengths(torch.tensor([lengths[i] for i in idx], device=dev), T)
            if extra is not None:
                x.update(extra(idx, T))
            mel, mel_len, dur = self.net(x)
            ml = mel_len.cpu().numpy()
            for r, i in enumerate(idx):
                out[i] = (mel[r, :int(ml[r])], dur[r, :lengths[i], 0])
        return out
