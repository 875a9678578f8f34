#!/usr/bin/env python3
"""Timeline of the API-inclusive bootstrap on a fresh handle (bench.py's `standalone_ms` harness): run under
`rocprofv3 --kernel-trace --hip-trace --memory-copy-trace --output-format csv -d DIR -o tl -- python tools/api_timeline.py run`, then
`python tools/api_timeline.py read DIR` prints, for the last calls, every HIP call / kernel / copy between the enqueue and the return of the
summary with its offset from the enqueue."""
import csv
import glob
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "plspm-python_amd")); sys.path.insert(0, os.path.join(ROOT, "tools"))


def run():
    import numpy as np
    import synthetic
    from plspm import _native
    C = synthetic.satisfaction_C()
    X, blocks = synthetic.synth(10000, C, 10, seed=0)
    boff = np.arange(0, 61, 10).astype(np.int32)
    keep, walls = None, []
    for k in range(8):
        h = _native.NativeModel(boff, C.astype(np.uint8), np.zeros(6, dtype=np.int32), 2, True, 100, 1e-6, 0)
        h.upload(X); h.prepare_bootstrap(); h.fit(want_scores=True, want_cov=True)
        time.sleep(0.002)
        t0 = time.perf_counter()
        h.bootstrap_device(5000, seed=1)
        h.summary(5000, np.ones(h.row_width))
        walls.append(round((time.perf_counter() - t0) * 1e3, 4))
        keep = h
        time.sleep(0.002)
    print(json.dumps({"standalone_ms": walls}))


def read(d):
    ev = []
    for f in glob.glob(os.path.join(d, "**", "*.csv"), recursive=True):
        base = os.path.basename(f)
        with open(f) as fh:
            for r in csv.DictReader(fh):
                s, e = r.get("Start_Timestamp"), r.get("End_Timestamp")
                if s is None or e is None:
                    continue
                if "kernel_trace" in base:
                    ev.append((int(s), int(e), "K  " + r["Kernel_Name"].split("(")[0][:60]))
                elif "memory_copy" in base:
                    ev.append((int(s), int(e), "C  %s %s B" % (r.get("Direction", ""), r.get("Size", r.get("Bytes", "?")))))
                elif "hip_api" in base:
                    ev.append((int(s), int(e), "A  " + r["Function"]))
    ev.sort()
    # the calls of interest: from the resample kernel's enqueue to the end of the summary's stream synchronise
    starts = [i for i, x in enumerate(ev) if x[2].startswith("K  ") and "resample_i8" in x[2]]
    for i0 in starts[-2:]:
        # back up to the first API call of this bootstrap_device (the launch of the resample kernel)
        j = i0
        while j > 0 and ev[j][0] > ev[i0][0] - 60000:
            j -= 1
        t0 = None
        for x in ev[j:]:
            if t0 is None:
                if x[2].startswith("A  ") and "Launch" in x[2]:
                    t0 = x[0]
                else:
                    continue
            if x[0] - t0 > 1500000:
                break
            print("%9.1f us  +%7.1f  %s" % ((x[0] - t0) / 1e3, (x[1] - x[0]) / 1e3, x[2]))
        print("-" * 60)


if __name__ == "__main__":
    run() if sys.argv[1] == "run" else read(sys.argv[2])
