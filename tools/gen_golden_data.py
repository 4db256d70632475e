"""Golden vectors of the reference's loader and text front-end, generated in the build container by importing
/root/reference/datamodule.py and /root/reference/text (datamodule.py:29-79 collate_fn, :113-186 LJSpeechDataset;
text/__init__.py:15-41 text_to_sequence) with one-class stand-ins for the absent `lightning`, `unidecode`, `inflect` modules.

    python tools/gen_golden_data.py          -> tests/golden/data_loader.npz, tests/golden/text_ids.json

data_loader.npz holds the INPUT tree (metadata lines and every .npy array of a seven-utterance preprocessed_data directory in the
reference's on-disk format -- equal-length utterances, `sp` tokens, an over-long line that max_length drops) and what the
reference's `LJSpeechDataset.__getitem__` and `collate_fn` return for it; tests/test_data_loader.py rebuilds the tree from the
fixture on any box and compares efficientspeech_amd.data against those outputs.  The reference never leaves this container."""
import json
import os
import sys
import tempfile
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, "/root/reference")
u = types.ModuleType("unidecode"); u.unidecode = lambda s: s; sys.modules["unidecode"] = u          # noqa: E702
class _E:                                                                                            # noqa: E302
    def number_to_words(self, *a, **k): return ""                                                    # noqa: E704
i = types.ModuleType("inflect"); i.engine = lambda: _E(); sys.modules["inflect"] = i                 # noqa: E702
li = types.ModuleType("lightning"); li.LightningDataModule = object; sys.modules["lightning"] = li   # noqa: E702
import datamodule as ref_dm                                                                          # noqa: E402
from text import text_to_sequence as ref_text_to_sequence                                            # noqa: E402
from text.symbols import symbols as ref_symbols                                                      # noqa: E402

TEXT_CASES = ["{HH AW1 S S T AH0 N}", "{sp}", "{DH AH0 sp K AE1 T spn sil}", "Turn left on {HH AW1 S S T AH0 N} Street.",
              "{AA AA0 ZH Y OY2 UW1}", "a  b\tc!",
              "{DH AH0 K W IH1 K B R AW1 N F AA1 K S JH AH1 M P S OW1 V ER0 DH AH0 L EY1 Z IY0 D AO1 G}"]


def make_tree(root, rng):
    """-> metadata lines; arrays {relative path: ndarray}"""
    arpabet = [s[1:] for s in ref_symbols if s.startswith("@") and s[1:] not in ("sp", "spn", "sil")]
    lens = [9, 14, 9, 3, 14, 6, 11]                         # ties: the collate sort is NumPy's argsort of the negated lengths
    lines, arrays = [], {}
    for k, T in enumerate(lens):
        base = f"LJ001-{k:04d}"
        phones = [arpabet[int(j)] for j in rng.integers(0, len(arpabet), size=T)]
        if k in (1, 5):
            phones[T // 2] = "sp"
        dur = rng.integers(0, 4, size=T).astype(np.int64)
        dur[0] = max(dur[0], 1)
        arr = {"mel": rng.standard_normal((int(dur.sum()), 80)).astype(np.float32), "pitch": rng.standard_normal(T).astype(np.float32),
               "energy": rng.standard_normal(T).astype(np.float64 if k == 2 else np.float32),      # (a float64 file: collate casts)
               "duration": dur}
        for kind, a in arr.items():
            arrays[f"{kind}/LJSpeech-{kind}-{base}.npy"] = a
        raw = f"utterance number {k}, printed." if k != 3 else "x" * 400                           # k = 3: dropped by max_length
        lines.append(f"{base}|LJSpeech|{{{' '.join(phones)}}}|{raw}")
    for d in ("mel", "pitch", "energy", "duration"):
        os.makedirs(os.path.join(root, d), exist_ok=True)
    for rel, a in arrays.items():
        np.save(os.path.join(root, rel), a)
    open(os.path.join(root, "train.txt"), "w", encoding="utf-8").write("\n".join(lines) + "\n")
    json.dump({"LJSpeech": 0}, open(os.path.join(root, "speakers.json"), "w"))
    return lines, arrays


def main():
    rng = np.random.default_rng(20240928)
    out = {}
    with tempfile.TemporaryDirectory() as root:
        lines, arrays = make_tree(root, rng)
        cfg = {"dataset": "LJSpeech", "path": {"preprocessed_path": root},
               "preprocessing": {"text": {"text_cleaners": ["english_cleaners"], "max_length": 100}}}
        ds = ref_dm.LJSpeechDataset("train.txt", cfg)
        dm = ref_dm.LJSpeechDataModule(cfg, batch_size=4, num_workers=0)
        out["meta_lines"] = np.array(lines)
        for rel, a in arrays.items():
            out["tree/" + rel] = a
        out["n_items"] = np.array(len(ds))
        out["basenames"] = np.array(ds.basename)
        items = [ds[j] for j in range(len(ds))]
        for j, (x, y) in enumerate(items):
            out[f"item{j}/phoneme"] = np.asarray(x["phoneme"])
            out[f"item{j}/text"] = np.array(x["text"])
            for k in ("pitch", "energy", "duration"):
                out[f"item{j}/{k}"] = x[k]
            out[f"item{j}/mel"] = y["mel"]
        for name, idxs in (("all", list(range(len(ds)))), ("first4", [0, 1, 2, 3]), ("ties", [2, 0, 4, 1])):
            x, y = dm.collate_fn([items[j] for j in idxs])
            out[f"batch_{name}/indices"] = np.array(idxs)
            out[f"batch_{name}/text"] = np.array(x["text"])
            for k, v in x.items():
                if k != "text":
                    out[f"batch_{name}/{k}"] = v.numpy()
                    out[f"batch_{name}/{k}.dtype"] = np.array(str(v.dtype))
            out[f"batch_{name}/mel"] = y["mel"].numpy()
    path = os.path.join(ROOT, "tests", "golden", "data_loader.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")
    cases = {t: [int(v) for v in ref_text_to_sequence(t, ["english_cleaners"] if t.startswith("{DH AH0 K W") else ["basic_cleaners"])] for t in TEXT_CASES}
    tpath = os.path.join(ROOT, "tests", "golden", "text_ids.json")
    json.dump({"generator": "tools/gen_golden_data.py: the reference's text.text_to_sequence run in the build container (basic_cleaners; "
                            "the fox sentence with english_cleaners)", "n_symbols": len(ref_symbols), "cases": cases}, open(tpath, "w"), indent=1)
    print("wrote", tpath)


if __name__ == "__main__":
    main()
