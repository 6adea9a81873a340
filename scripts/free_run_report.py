#!/usr/bin/env python3
"""Free-running two-track comparison (oracle/free_run.py) as a stand-alone report: the HIP tracker's own closed loop vs the oracle's
own closed loop + the oracle-vs-itself control, on the synthetic tracking problem with trained stand-in weights and on the random-init
stand-in (bench.py's `track.free_running` is the same call).
    python scripts/free_run_report.py [frames] [seeds] [frames of the random-init part] > report.json"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

if __name__ == "__main__":
    import se3tracknet_amd as se3
    from oracle import free_run
    frames = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
    seeds = tuple(range(int(sys.argv[2]))) if len(sys.argv) > 2 else (0, 1, 2)
    frames_random = int(sys.argv[3]) if len(sys.argv) > 3 else frames
    r = free_run.run_report(se3, frames, frames_random, seeds=seeds, control_seeds=seeds)
    print(json.dumps(r))
