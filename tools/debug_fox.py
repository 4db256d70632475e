import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from tests import helpers as H
from efficientspeech_amd import _lib
if len(sys.argv) > 1:
    _lib._LIB = _lib.bind(C.CDLL(os.path.abspath(sys.argv[1])))
g = np.load(os.path.join(ROOT, "tests/golden/tiny_eval_b1_fox.npz"))
net, cfg, sd = H.make_net("tiny", "cuda", golden=g)
x = H.to_x(g, "cuda")
for plan in (31, 7, 0):
    with torch.no_grad(), _lib.launch_plan(plan):
        enc = net.encoder._encode(x)
    torch.cuda.synchronize()
    d = np.abs(enc["duration"].cpu().numpy() - g["duration"])[0, :, 0]
    f0 = np.abs(enc["feats"][0].cpu().numpy() - g["f0"]).max(-1)[0]
    f1 = np.abs(enc["feats"][1].cpu().numpy() - g["f1"]).max(-1)[0]
    ft = np.abs(enc["feat"].cpu().numpy() - g["feat"]).max(-1)[0]
    print("plan", plan, "dur bad rows", np.nonzero(d > 1e-4)[0], "f0 bad", np.nonzero(f0 > 1e-4)[0], "f1 bad", np.nonzero(f1 > 1e-4)[0], "feat bad", np.nonzero(ft > 1e-4)[0])
