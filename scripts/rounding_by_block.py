#!/usr/bin/env python3
"""Which Winograd block costs the rounding?  (GPU box.)  The batched 30-degree closed loop of oracle/closed_loop.py (64 independent
tracks per engine call, every pair of every frame against the CPU oracle) with SE3TN_WINOGRAD_TILE_AUTO picking, per block,
F(6x6) | F(4x4) for the 256-channel block (convAB2) and the 512-channel heads -- the library's developer switches
SE3TN_WINOGRAD_AUTO_TILE_AB2 / _HEADS, read at se3tn_create.  Prints max |d logit|, max |d(trans, rot)|, max |d pose| per setting.

    python scripts/rounding_by_block.py [frames] [tracks]
"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import closed_loop   # noqa: E402  (TEST INFRASTRUCTURE: this script is a study, not product code)


def main():
    import se3tracknet_amd as se3
    frames = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    tracks = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    rows = []
    for ab2, heads in ((6, 6), (4, 6), (6, 4), (4, 4)):
        os.environ["SE3TN_WINOGRAD_AUTO_TILE_AB2"] = str(ab2)
        os.environ["SE3TN_WINOGRAD_AUTO_TILE_HEADS"] = str(heads)
        r = closed_loop.run_regime_batch(se3, "ycbineoat_30deg", tracks, frames=frames)
        blocks = [nm for nm in r["launches"] if "[F(" in nm]
        rows.append({"AB2": ab2, "heads": heads, "pairs": r["pairs_checked"], "max_abs_logit_diff": r["max_abs_logit_diff"],
                     "max_abs_trans_rot": r["max_abs_trans_rot"], "max_abs_pose": r["max_abs_pose"], "launches": blocks})
        print("AB2 F(%dx%d)  heads F(%dx%d):  %d pairs  max |d logit| %.2e  |d(trans,rot)| %.2e  |d pose| %.2e   %s" % (
            ab2, ab2, heads, heads, r["pairs_checked"], r["max_abs_logit_diff"], r["max_abs_trans_rot"], r["max_abs_pose"],
            [b.split(" [")[1] for b in blocks]), flush=True)
    print(json.dumps(rows))


if __name__ == "__main__":
    main()
