"""On-disk formats + loader of the reference's training / evaluation data.

`/root/reference/datamodule.py:113-186` reads a *preprocessed* LJSpeech tree (written by `preprocessor/preprocessor.py:155-251`):

    <preprocessed_path>/train.txt, val.txt     one utterance per line:  basename|speaker|{ARPAbet phones}|raw text
    <preprocessed_path>/{mel,pitch,energy,duration}/<speaker>-<kind>-<basename>.npy
                                               mel (L, 80) f32, pitch (T,) f32, energy (T,) f32, duration (T,) int
    <preprocessed_path>/speakers.json, stats.json

and batches it with `LJSpeechDataModule.collate_fn` (`datamodule.py:29-79`): utterances sorted by phoneme length (longest
first), zero-padded, with `phoneme_mask` / `mel_mask` from `get_mask_from_lengths` -- the dict `Phoneme2Mel.forward(x, train=True)`
consumes.  This module restates that host-side logic (no Lightning, no text-normalisation dependencies): the phone strings of the
preprocessed metadata are already ARPAbet in curly braces, which `text_to_sequence` maps through the reference's symbol table
(`text/symbols.py`, `text/__init__.py:15-41`).
"""
import json
import os
import re

import numpy as np
import torch

from .networks import get_mask_from_lengths

# ---- text/symbols.py: [_pad] + _special + _punctuation + _letters + ["@" + arpabet] + _silences  (152 symbols; ids 0..151)
_PAD = "_"
_SPECIAL = "-/"
_PUNCTUATION = "!'(),.:;? "
_LETTERS = "ABCDEFGHIJKLMNOPQRSTUVWXYZabcdefghijklmnopqrstuvwxyz"
_VOWELS = ("AA", "AE", "AH", "AO", "AW", "AY", "EH", "ER", "EY", "IH", "IY", "OW", "OY", "UH", "UW")
_CONSONANTS = ("B", "CH", "D", "DH", "F", "G", "HH", "JH", "K", "L", "M", "N", "NG", "P", "R", "S", "SH", "T", "TH", "V", "W",
               "Y", "Z", "ZH")
# text/cmudict.py valid_symbols: the CMUdict phone set, every vowel plain and with stress 0/1/2, in alphabetical order
ARPABET = sorted(list(_CONSONANTS) + [v + s for v in _VOWELS for s in ("", "0", "1", "2")])
_SILENCES = ["@sp", "@spn", "@sil"]
SYMBOLS = [_PAD] + list(_SPECIAL) + list(_PUNCTUATION) + list(_LETTERS) + ["@" + s for s in ARPABET] + _SILENCES
_SYMBOL_TO_ID = {s: i for i, s in enumerate(SYMBOLS)}
_CURLY = re.compile(r"(.*?)\{(.+?)\}(.*)")


def _keep(s):
    return s in _SYMBOL_TO_ID and s not in ("_", "~")


def text_to_sequence(text, cleaner_names=None):
    """text/__init__.py:15-41: characters outside curly braces are lower-cased, whitespace-collapsed (= `basic_cleaners`) and
    mapped one to one (the reference's `english_cleaners` additionally expand numbers / abbreviations through unidecode +
    inflect, which the preprocessed metadata never needs: its phone field is entirely inside braces); `{...}` contents are
    ARPAbet symbols."""
    seq = []
    while len(text):
        m = _CURLY.match(text)
        if not m:
            seq += [_SYMBOL_TO_ID[c] for c in re.sub(r"\s+", " ", text.lower()) if _keep(c)]
            break
        seq += [_SYMBOL_TO_ID[c] for c in re.sub(r"\s+", " ", m.group(1).lower()) if _keep(c)]
        seq += [_SYMBOL_TO_ID["@" + s] for s in m.group(2).split() if _keep("@" + s)]
        text = m.group(3)
    return seq


def pad_1D(inputs, PAD=0):
    """utils/tools.py:262-272"""
    n = max(len(x) for x in inputs)
    return np.stack([np.pad(x, (0, n - x.shape[0]), mode="constant", constant_values=PAD) for x in inputs])


def pad_2D(inputs, maxlen=None):
    """utils/tools.py:275-293: (L_i, C) -> (B, max L, C), zero padded"""
    n = maxlen or max(np.shape(x)[0] for x in inputs)
    out = []
    for x in inputs:
        if np.shape(x)[0] > n:
            raise ValueError("not max_len")
        out.append(np.pad(x, ((0, n - np.shape(x)[0]), (0, 0)), mode="constant", constant_values=0))
    return np.stack(out)


class LJSpeechDataset(torch.utils.data.Dataset):
    """datamodule.py:113-186: one item = ({phoneme, text, pitch, energy, duration}, {mel}) as NumPy arrays."""

    def __init__(self, filename, preprocess_config):
        self.preprocessed_path = preprocess_config["path"]["preprocessed_path"]
        self.cleaners = preprocess_config["preprocessing"]["text"]["text_cleaners"]
        self.max_text_length = preprocess_config["preprocessing"]["text"]["max_length"]
        self.basename, self.speaker, self.text, self.raw_text = self.process_meta(filename)
        with open(os.path.join(self.preprocessed_path, "speakers.json")) as f:
            self.speaker_map = json.load(f)

    def __len__(self):
        return len(self.text)

    def _load(self, kind, idx):
        return np.load(os.path.join(self.preprocessed_path, kind, f"{self.speaker[idx]}-{kind}-{self.basename[idx]}.npy"))

    def __getitem__(self, idx):
        x = {"phoneme": np.array(text_to_sequence(self.text[idx], self.cleaners)), "text": self.raw_text[idx],
             "pitch": self._load("pitch", idx), "energy": self._load("energy", idx), "duration": self._load("duration", idx)}
        return x, {"mel": self._load("mel", idx)}

    def process_meta(self, filename):
        """The metadata file: one utterance per line, four fields separated by the pipe character -- basename, speaker, phone string,
        raw text.  Utterances whose raw text is longer than `max_length` characters are left out (datamodule.py:172-186).
        -> four parallel lists (basenames, speakers, phone strings, raw texts)."""
        path = os.path.join(self.preprocessed_path, filename)
        with open(path, "r", encoding="utf-8") as f:
            records = [line.rstrip("\n").split("|") for line in f]
        for rec in records:
            if len(rec) != 4:
                raise ValueError(f"{path}: expected 4 fields per line, got {len(rec)}")       # (the reference's unpacking raises too)
        kept = [rec for rec in records if len(rec[3]) <= self.max_text_length]
        columns = list(zip(*kept)) if kept else [(), (), (), ()]
        return tuple(list(c) for c in columns)


def collate_fn(batch):
    """A list of dataset items -> the padded training batch (x, y) of datamodule.py:29-79: utterances ordered by phoneme count,
    longest first -- the order NumPy's default argsort gives the negated counts, which is what decides between utterances of
    equal length -- every per-utterance array zero-padded to the batch maximum, boolean masks that are True on the padding."""
    xs = [item[0] for item in batch]
    ys = [item[1] for item in batch]
    order = np.argsort(-np.array([x["phoneme"].shape[0] for x in xs])).tolist()
    xs, ys = [xs[j] for j in order], [ys[j] for j in order]

    def lengths(arrays):
        return torch.from_numpy(np.array([a.shape[0] for a in arrays])).int()

    def padded(key):
        return torch.from_numpy(pad_1D([x[key] for x in xs]))

    phoneme_len, mel_len = lengths([x["phoneme"] for x in xs]), lengths([y["mel"] for y in ys])
    x = {"phoneme": padded("phoneme").int(),
         "phoneme_len": phoneme_len,
         "phoneme_mask": get_mask_from_lengths(phoneme_len, int(phoneme_len.max())),
         "text": [x["text"] for x in xs],
         "mel_len": mel_len,
         "mel_mask": get_mask_from_lengths(mel_len, int(mel_len.max())),
         "pitch": padded("pitch").float(),
         "energy": padded("energy").float(),
         "duration": padded("duration").int()}
    return x, {"mel": torch.from_numpy(pad_2D([y["mel"] for y in ys])).float()}


class LJSpeechDataModule:
    """datamodule.py:19-110 without Lightning: `setup()`, then `train_dataloader()` / `test_dataloader()` / `val_dataloader()`."""

    def __init__(self, preprocess_config, batch_size=64, num_workers=4):
        self.preprocess_config, self.batch_size, self.num_workers = preprocess_config, batch_size, num_workers
        self.collate_fn = collate_fn

    def prepare_data(self):
        self.train_dataset = LJSpeechDataset("train.txt", self.preprocess_config)
        self.test_dataset = LJSpeechDataset("val.txt", self.preprocess_config)

    def setup(self, stage=None):
        self.prepare_data()

    def train_dataloader(self):
        return torch.utils.data.DataLoader(self.train_dataset, shuffle=True, batch_size=self.batch_size, collate_fn=collate_fn,
                                           num_workers=self.num_workers)

    def test_dataloader(self):
        return torch.utils.data.DataLoader(self.test_dataset, shuffle=False, batch_size=self.batch_size, collate_fn=collate_fn,
                                           num_workers=self.num_workers)

    def val_dataloader(self):
        return self.test_dataloader()
