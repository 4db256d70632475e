"""On-disk formats + loader: a synthetic preprocessed tree in the reference's layout
(datamodule.py:113-186, preprocessor/preprocessor.py:155-251) read back through efficientspeech_amd.data."""
import json
import os

import numpy as np
import torch

from efficientspeech_amd.data import (ARPABET, SYMBOLS, LJSpeechDataModule, LJSpeechDataset, collate_fn, pad_1D, pad_2D,
                                      text_to_sequence)

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def test_symbol_table_and_text_ids_match_reference_vectors():
    """ids produced by the reference's text.text_to_sequence (committed as data by the build container)."""
    assert len(SYMBOLS) == 152 and len(ARPABET) == 84 and SYMBOLS[0] == "_" and SYMBOLS[-3:] == ["@sp", "@spn", "@sil"]
    cases = json.load(open(os.path.join(GOLD, "text_ids.json")))["cases"]
    assert len(cases) >= 6
    for text, ids in cases.items():
        assert text_to_sequence(text) == ids, text


def _fake_tree(root, rng, n_train=7, n_val=3):
    for d in ("mel", "pitch", "energy", "duration"):
        os.makedirs(os.path.join(root, d))
    json.dump({"LJSpeech": 0}, open(os.path.join(root, "speakers.json"), "w"))
    json.dump({"pitch": [-2.9, 11.4, 0.0, 1.0], "energy": [-1.4, 8.2, 0.0, 1.0]}, open(os.path.join(root, "stats.json"), "w"))
    items = {}
    for split, n in (("train.txt", n_train), ("val.txt", n_val)):
        lines = []
        for k in range(n):
            base = f"LJ00{1 if split == 'train.txt' else 2}-{k:04d}"
            T = int(rng.integers(3, 20))
            phones = [ARPABET[int(j)] for j in rng.integers(0, len(ARPABET), size=T)]
            dur = rng.integers(1, 7, size=T).astype(np.int64)
            L = int(dur.sum())
            arrs = {"mel": rng.standard_normal((L, 80)).astype(np.float32), "pitch": rng.standard_normal(T).astype(np.float32),
                    "energy": rng.standard_normal(T).astype(np.float32), "duration": dur}
            for kind, a in arrs.items():
                np.save(os.path.join(root, kind, f"LJSpeech-{kind}-{base}.npy"), a)
            raw = "x" * (30 if k != 1 else 500)                      # one over-long line per split: dropped by max_length
            lines.append(f"{base}|LJSpeech|{{{' '.join(phones)}}}|{raw}")
            items[base] = (phones, arrs, raw)
        open(os.path.join(root, split), "w", encoding="utf-8").write("\n".join(lines) + "\n")
    return items


def test_dataset_and_collate_follow_the_reference_layout(tmp_path):
    rng = np.random.default_rng(5)
    root = str(tmp_path / "preprocessed_data" / "LJSpeech")
    items = _fake_tree(root, rng)
    cfg = {"dataset": "LJSpeech", "path": {"preprocessed_path": root},
           "preprocessing": {"text": {"text_cleaners": ["english_cleaners"], "max_length": 100}}}
    ds = LJSpeechDataset("train.txt", cfg)
    assert len(ds) == 6                                               # 7 lines, one longer than max_length
    for i in range(len(ds)):
        x, y = ds[i]
        phones, arrs, raw = items[ds.basename[i]]
        assert x["text"] == raw and x["phoneme"].tolist() == [SYMBOLS.index("@" + p) for p in phones]
        for kind in ("pitch", "energy", "duration"):
            assert np.array_equal(x[kind], arrs[kind])
        assert np.array_equal(y["mel"], arrs["mel"]) and y["mel"].shape[0] == int(arrs["duration"].sum())
    batch = [ds[i] for i in range(len(ds))]
    X, Y = collate_fn(batch)
    lens = np.array([b[0]["phoneme"].shape[0] for b in batch])
    order = np.argsort(-lens).tolist()                               # datamodule.py:31-32
    B, T, L = len(batch), int(lens.max()), max(b[1]["mel"].shape[0] for b in batch)
    assert X["phoneme"].dtype == torch.int32 and X["phoneme"].shape == (B, T)
    assert X["duration"].dtype == torch.int32 and X["pitch"].dtype == torch.float32 and Y["mel"].shape == (B, L, 80)
    assert X["phoneme_len"].tolist() == lens[order].tolist() and X["text"] == [batch[i][0]["text"] for i in order]
    for r, i in enumerate(order):
        x, y = batch[i]
        n, m = x["phoneme"].shape[0], y["mel"].shape[0]
        assert X["phoneme"][r, :n].tolist() == x["phoneme"].tolist() and not X["phoneme"][r, n:].any()
        assert X["phoneme_mask"][r].tolist() == [False] * n + [True] * (T - n)
        assert X["mel_mask"][r].tolist() == [False] * m + [True] * (L - m) and int(X["mel_len"][r]) == m
        assert torch.equal(Y["mel"][r, :m], torch.from_numpy(y["mel"])) and not Y["mel"][r, m:].any()
        assert torch.equal(X["duration"][r, :n], torch.from_numpy(x["duration"]).int()) and int(X["duration"][r].sum()) == m
    dm = LJSpeechDataModule(cfg, batch_size=4, num_workers=0)
    dm.setup()
    xb, yb = next(iter(dm.val_dataloader()))
    assert xb["phoneme"].shape[0] == 2 and yb["mel"].shape[0] == 2    # 3 val lines, one dropped
    assert pad_1D([np.arange(3), np.arange(1)]).tolist() == [[0, 1, 2], [0, 0, 0]]
    assert pad_2D([np.ones((2, 3)), np.ones((1, 3))]).shape == (2, 2, 3)


def test_collated_batch_is_the_training_input_dict():
    """The keys Phoneme2Mel.forward(x, train=True) reads (networks.py:340-344) are all there with the right types."""
    rng = np.random.default_rng(1)
    batch = []
    for T in (5, 9, 2):
        dur = rng.integers(1, 4, size=T)
        batch.append(({"phoneme": rng.integers(1, 150, size=T), "text": "t", "pitch": rng.standard_normal(T).astype(np.float32),
                       "energy": rng.standard_normal(T).astype(np.float32), "duration": dur},
                      {"mel": rng.standard_normal((int(dur.sum()), 80)).astype(np.float32)}))
    X, Y = collate_fn(batch)
    for k in ("phoneme", "phoneme_mask", "pitch", "energy", "duration", "mel_len"):
        assert k in X
    assert X["phoneme_mask"].dtype == torch.bool and X["mel_len"].dtype == torch.int32
    assert int(X["mel_len"].max()) == Y["mel"].shape[1]


def _tree_from_fixture(root):
    """Rebuild the preprocessed_data directory the fixture was generated from (tools/gen_golden_data.py stores the tree itself)."""
    g = np.load(os.path.join(GOLD, "data_loader.npz"))
    for d in ("mel", "pitch", "energy", "duration"):
        os.makedirs(os.path.join(root, d))
    for k in g.files:
        if k.startswith("tree/"):
            np.save(os.path.join(root, k[5:]), g[k])
    open(os.path.join(root, "train.txt"), "w", encoding="utf-8").write("\n".join(str(v) for v in g["meta_lines"]) + "\n")
    json.dump({"LJSpeech": 0}, open(os.path.join(root, "speakers.json"), "w"))
    cfg = {"dataset": "LJSpeech", "path": {"preprocessed_path": root},
           "preprocessing": {"text": {"text_cleaners": ["english_cleaners"], "max_length": 100}}}
    return g, cfg


def test_dataset_and_collate_match_reference_generated_fixture(tmp_path):
    """What the reference's own LJSpeechDataset.__getitem__ and collate_fn returned for this tree (datamodule.py:29-79, 113-186;
    generated by tools/gen_golden_data.py in the build container): item arrays, the max_length filter, the sort order of
    equal-length utterances, padding, masks, dtypes -- all exact."""
    g, cfg = _tree_from_fixture(str(tmp_path / "pre"))
    ds = LJSpeechDataset("train.txt", cfg)
    assert len(ds) == int(g["n_items"]) == 6 and ds.basename == [str(v) for v in g["basenames"]]
    items = [ds[j] for j in range(len(ds))]
    for j, (x, y) in enumerate(items):
        assert x["phoneme"].tolist() == g[f"item{j}/phoneme"].tolist() and x["text"] == str(g[f"item{j}/text"])
        for k in ("pitch", "energy", "duration"):
            assert np.array_equal(x[k], g[f"item{j}/{k}"]) and x[k].dtype == g[f"item{j}/{k}"].dtype, (j, k)
        assert np.array_equal(y["mel"], g[f"item{j}/mel"])
    for name in ("all", "first4", "ties"):
        idxs = g[f"batch_{name}/indices"].tolist()
        X, Y = collate_fn([items[j] for j in idxs])
        assert X["text"] == [str(v) for v in g[f"batch_{name}/text"]]
        for k in ("phoneme", "phoneme_len", "phoneme_mask", "mel_len", "mel_mask", "pitch", "energy", "duration"):
            ref = g[f"batch_{name}/{k}"]
            assert str(X[k].dtype) == str(g[f"batch_{name}/{k}.dtype"]), (name, k, X[k].dtype)
            assert X[k].shape == ref.shape and np.array_equal(X[k].numpy(), ref), (name, k)
        assert np.array_equal(Y["mel"].numpy(), g[f"batch_{name}/mel"]) and Y["mel"].dtype == torch.float32
        assert set(X) == {"phoneme", "phoneme_len", "phoneme_mask", "text", "mel_len", "mel_mask", "pitch", "energy", "duration"}


def check_loader_batch_trains(dev, tmp_path):
    """Loader -> training step on the device: a batch collated from the reference-format tree goes through TrainStep.step."""
    from efficientspeech_amd import CONFIGS, build_phoneme2mel, train
    from efficientspeech_amd.synth import synth_state_dict
    g, cfg = _tree_from_fixture(str(tmp_path / "pre"))
    ds = LJSpeechDataset("train.txt", cfg)
    loader = torch.utils.data.DataLoader(ds, batch_size=3, shuffle=False, collate_fn=collate_fn)
    mcfg = CONFIGS["tiny"]
    net = build_phoneme2mel(mcfg)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in synth_state_dict(mcfg, 1234).items()}, strict=True)
    net = net.to(dev).train()
    step = train.TrainStep(net, lr=1e-3)
    losses = []
    for epoch in range(3):
        for x, y in loader:
            losses.append(step.step(train.to_device(x, dev), train.to_device(y, dev)).cpu().numpy())
    losses = np.array(losses)
    assert np.isfinite(losses).all() and losses.shape == (6, 5)
    assert losses[4:, 4].mean() < losses[:2, 4].mean()                # the same two batches, two epochs later: the loss fell
    return losses


import pytest  # noqa: E402


@pytest.mark.gpu
def test_gpu_loader_batch_through_the_training_step(tmp_path):
    check_loader_batch_trains("cuda", tmp_path)


def test_simulated_loader_batch_through_the_training_step(tmp_path):
    from tests.simlib import use_sim
    with use_sim():
        check_loader_batch_trains("cpu", tmp_path)
