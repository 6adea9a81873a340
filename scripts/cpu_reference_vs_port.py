"""Build container only (needs /root/reference): the UNMODIFIED reference Se3TrackNet on torch-CPU next to the oracle port that
bench.py's `cpu_baseline` times on the GPU box (where the reference tree does not exist) -- same weights, same inputs, same thread
count.  Shows once that `cpu_baseline.kind = "port"` stands for the reference's speed.
    python scripts/cpu_reference_vs_port.py > profiles/r05_cpu_reference_vs_port.txt"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import fixtures as Fx, ref_shims, se3_oracle as O
from oracle.make_golden import ref_model

torch.set_num_threads(os.cpu_count() or 1)
ref = ref_shims.load()
sd = O.make_state_dict(0)
model = ref_model(ref, sd)
print("host: %d logical cores, torch %s, %d threads; model %s" % (os.cpu_count(), torch.__version__, torch.get_num_threads(),
      [l.split(":")[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")][0]))
for n in (1, 16, 64):
    A, B = Fx.net_inputs(3, n)
    def t(fn, reps=5):
        fn(); ts = []
        for _ in range(reps):
            t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
        return float(np.median(ts))
    with torch.no_grad():
        tr = t(lambda: model(A, B))
        r = model(A, B)
    tp = t(lambda: O.forward(sd, A, B))
    o = O.forward(sd, A, B)
    d = max(float((r["trans"] - o["trans"]).abs().max()), float((r["rot"] - o["rot"]).abs().max()))
    print("batch %2d: reference Se3TrackNet %8.1f ms = %6.1f pairs/s | oracle port %8.1f ms = %6.1f pairs/s | port / reference time %.3f | "
          "max |d(trans, rot)| %.1e" % (n, tr * 1e3, n / tr, tp * 1e3, n / tp, tp / tr, d))
