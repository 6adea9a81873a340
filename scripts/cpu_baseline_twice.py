import sys, json, time
sys.path.insert(0, '/root/repo')
import bench
from oracle import se3_oracle as O
if __name__ == "__main__":
    sd = O.make_state_dict(0)
    for i in range(2):
        t = time.time(); r = bench.cpu_baseline(O, sd, 64)
        print(i, r["value"], r["configuration"], {k: v["value"] for k, v in r["configurations_pairs_per_s"].items()}, r.get("configurations_failed"), round(time.time() - t, 1))
