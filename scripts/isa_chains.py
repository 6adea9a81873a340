#!/usr/bin/env python3
"""Static look at how many memory latencies a wave of each kernel waits out one after the other
(profiles/EXPERIMENTS.md item 38: the Winograd out-transform / tail kernels issued 36 `read -> s_waitcnt vmcnt(0) -> store`
groups per thread; nothing in a kernel trace or a PMC pass points at that, the disassembly shows it at a glance).

For every gfx950 kernel in the library: the sequence of global reads (L), global stores (S), LDS-DMA reads (D), waits
(`w<N>` = s_waitcnt vmcnt(N)), barriers (|) and branches (b), and from it
  * chains  = number of vmcnt waits that have at least one read issued since the previous vmcnt wait, i.e. an upper bound on the
              memory round trips a wave serialises in ONE pass over its code (loops count once),
  * inflight = the largest number of reads issued between two vmcnt waits.
Kernels with many chains and a small inflight figure are latency chains.  Matrix kernels are double-buffered loops (one wait
per K-step by design) and show up with chains of 2-10; the figure matters for the streaming passes.

    python scripts/isa_chains.py [path/to/libse3tracknet.so] [-v substring]     (-v: print the sequence of matching kernels)"""
import os
import re
import subprocess
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from isa_lint import LLVM, code_objects  # noqa: E402


def demangle(names):
    for tool in (os.path.join(LLVM, "llvm-cxxfilt"), "c++filt"):
        try:
            out = subprocess.run([tool], input="\n".join(names) + "\n", capture_output=True, text=True).stdout.splitlines()
            if len(out) == len(names):
                return dict(zip(names, out))
        except OSError:
            pass
    return {n: n for n in names}


def sequences(lib):
    seqs = {}
    with tempfile.TemporaryDirectory() as tmp:
        for co in code_objects(lib, tmp):
            dis = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "--mcpu=gfx950", co], capture_output=True, text=True).stdout
            cur = None
            for line in dis.splitlines():
                m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
                if m:
                    cur = m.group(1)
                    seqs[cur] = []
                    continue
                if cur is None:
                    continue
                t = line.split("//")[0].split()
                if not t:
                    continue
                op = t[0]
                if op.startswith("global_load_lds") or op.startswith("buffer_load") and "lds" in line:
                    seqs[cur].append("D")
                elif op.startswith(("global_load", "buffer_load", "flat_load")):
                    seqs[cur].append("L")
                elif op.startswith(("global_store", "buffer_store", "flat_store", "global_atomic")):
                    seqs[cur].append("S")
                elif op == "s_waitcnt":
                    m = re.search(r"vmcnt\((\d+)\)", line)
                    if m:
                        seqs[cur].append("w%s" % m.group(1))
                elif op == "s_barrier":
                    seqs[cur].append("|")
                elif op.startswith("s_cbranch"):
                    seqs[cur].append("b")
    return seqs


def figures(seq):
    chains = inflight = since = 0
    stores_waited = 0
    prev_store = False
    for s in seq:
        if s in ("L", "D"):
            since += 1
        elif s.startswith("w"):
            if since:
                chains += 1
                inflight = max(inflight, since)
            elif prev_store and s == "w0":
                stores_waited += 1   # a wait with nothing but stores before it: store -> wait -> store
            since = 0
            prev_store = False
        elif s == "S":
            prev_store = True
    return chains, max(inflight, since), stores_waited


if __name__ == "__main__":
    here = os.path.dirname(os.path.abspath(__file__))
    args = [a for a in sys.argv[1:]]
    show = None
    if "-v" in args:
        i = args.index("-v")
        show = args[i + 1]
        del args[i:i + 2]
    lib = args[0] if args else os.path.join(here, "..", "iros20-6d-pose-tracking_amd", "libse3tracknet.so")
    seqs = sequences(lib)
    names = demangle(list(seqs))
    rows = []
    for k, seq in seqs.items():
        c, f, sw = figures(seq)
        rows.append((c, f, sw, seq.count("L") + seq.count("D"), seq.count("S"), names[k].replace("se3tn::", "").split("(")[0][:78], k))
    rows.sort(key=lambda r: (-r[0], r[5]))
    print("%-80s %7s %8s %6s %6s %12s" % ("kernel", "chains", "inflight", "reads", "stores", "store-waits"))
    for c, f, sw, nl, ns, n, k in rows:
        print("%-80s %7d %8d %6d %6d %12d" % (n, c, f, nl, ns, sw))
        if show and show in n:
            print("    " + " ".join(seqs[k]))
