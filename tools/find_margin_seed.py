"""Pick the seeds of tests/test_train_step.py's at-size cases: the first seed whose fp64 reference run keeps every kink input
(predictor ReLUs, duration ReLU, L1 |.|) at least `--margin` from zero.  CPU only, no kernels involved.

    python tools/find_margin_seed.py [--margin 3e-5] [--max 400]
    python tools/find_margin_seed.py --errors      # per-parameter errors of the simulated kernels vs the fp64 reference
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import test_train_step as T      # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--margin", type=float, default=3e-5)
    ap.add_argument("--max", type=int, default=400)
    ap.add_argument("--errors", action="store_true")
    a = ap.parse_args()
    if a.errors:
        from tests.simlib import use_sim
        with use_sim():
            print(T.check_gradients_at_size("cpu", "tiny"))
        return
    for name in T.AT_SIZE:
        for seed in range(a.max):
            cfg, sd, x, y = T.at_size_case(name, seed)
            _, _, taps = T.fp64_reference(cfg, sd, x, y, backward=False)
            m = min(taps.values())
            if m > a.margin:
                print(f"{name}: seed {seed} min margin {m:.3e} ({min(taps, key=taps.get)})")
                break
        else:
            print(f"{name}: no seed below {a.max} reaches {a.margin}")


if __name__ == "__main__":
    main()
