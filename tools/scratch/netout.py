"""A/B helper (GPU only): runs the three networks of the mobile session on seeded inputs and saves the outputs, so that two
processes started with different RT_* switches can be compared bit for bit:
    RT_LC_WAVE=0 python tools/scratch/netout.py /tmp/a.npz; python tools/scratch/netout.py /tmp/b.npz; python tools/scratch/netout.py --cmp /tmp/a.npz /tmp/b.npz"""
import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
if sys.argv[1] == "--cmp":
    a, b = np.load(sys.argv[2]), np.load(sys.argv[3])
    same = [bool(np.array_equal(a[k].view(np.uint32), b[k].view(np.uint32))) for k in a.files]
    print("bit-identical:", same, "max |diff|:", [float(np.abs(a[k].astype(np.float64) - b[k]).max()) for k in a.files])
    sys.exit(0 if all(same) else 1)
import retto_amd
s = retto_amd.RettoSession(retto_amd.synthetic_session_config(0))
outs = []
for n, h, w in ((2, 960, 640), (3, 352, 416), (1, 96, 1248), (5, 64, 32), (1, 32, 32)):
    x = np.random.default_rng(w).uniform(-1, 1, (n, 3, h, w)).astype(np.float32)
    outs.append(s.worker.det(x))
for n, w in ((40, 336), (3, 24), (129, 344), (300, 640)):
    x = np.random.default_rng(n).uniform(-1, 1, (n, 3, 48, w)).astype(np.float32)
    r = s.worker.rec(x)
    outs.append(r.max(-1)); outs.append(r.argmax(-1).astype(np.float32))
outs.append(s.worker.cls(np.random.default_rng(6).uniform(-1, 1, (7, 3, 48, 192)).astype(np.float32)))
np.savez(sys.argv[1], *outs)
