#!/usr/bin/env python3
"""Summarise rocprofv3 rocpd (.db) outputs into profiles/: kernel time table (--kernel-trace --stats run) and per-kernel
HBM counters (separate --pmc FETCH_SIZE / --pmc WRITE_SIZE runs).

gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE reports exactly 1/2 of the bytes of a wide coalesced
read, so fetched bytes = 2 * FETCH_SIZE KiB; checked here on colsum_rowmajor_kernel, which reads a known N*P*8 bytes once.
WRITE_SIZE is used as reported.
usage: rocprof_summary.py <round-tag> <stats.db> [<fetch.db> <write.db> [<log of the traced run>]]"""
import json
import os
import sqlite3
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def short(name):
    return name.split("(")[0].replace("void ", "")


def main():
    tag, stats = sys.argv[1], sys.argv[2]
    out = {"round": tag, "kernels": [], "counters": {}}
    cur = sqlite3.connect(stats).cursor()
    lines = ["# rocprofv3 --kernel-trace --stats  (%s)" % tag, "", "| kernel | calls | total us | avg us | % |", "|---|---:|---:|---:|---:|"]
    # full-size launches only: bench.py ends with a few tiny validation launches of the same kernels
    q = ("select name, count(*), sum(duration)/1e3, avg(duration)/1e3 from kernels k where grid_x*grid_y*grid_z = "
         "(select max(grid_x*grid_y*grid_z) from kernels k2 where k2.name = k.name) group by name order by 3 desc")
    rows = list(cur.execute(q))
    tot_all = sum(r[2] for r in rows) or 1.0
    for name, calls, total, avg in rows:
        pct = 100.0 * total / tot_all
        out["kernels"].append({"kernel": short(name), "calls": calls, "total_us": round(total, 3), "avg_us": round(avg, 3), "pct": round(pct, 2)})
        lines.append("| %s | %d | %.1f | %.2f | %.2f |" % (short(name), calls, total, avg, pct))
    # the dominant kernel by phase of the bench command (bench.py: W + K cold steps, 200 spin-up steps, W warm-up steps, the K TIMED
    # steps, then the profiled / validation launches): the timed-region average is the one the bench line's roofline.avg_launch_ms --
    # HIP events on every 10th timed step of its own run -- has to agree with; the log of the traced run (argv[5]) carries that line
    W, K, SPIN = 5, 50, 200
    dom = rows[0][0] if rows else None
    if dom and rows[0][1] >= 2 * (W + K) + SPIN:
        d = [r[0] / 1e3 for r in cur.execute("select duration from kernels k where name = ? and grid_x*grid_y*grid_z = (select max(grid_x*grid_y*grid_z) "
                                             "from kernels k2 where k2.name = k.name) order by start", (dom,))]
        cuts = [("cold: the first W + K = %d steps (device coming out of idle)" % (W + K), 0, W + K), ("spin-up (%d steps)" % SPIN, W + K, W + K + SPIN),
                ("warm-up (W = %d)" % W, W + K + SPIN, 2 * W + K + SPIN), ("TIMED region (K = %d)" % K, 2 * W + K + SPIN, 2 * (W + K) + SPIN), ("behind the timed region", 2 * (W + K) + SPIN, len(d))]
        lines += ["", "# %s by phase of the command (`bench.py --steps %d --warmup %d`)" % (short(dom), K, W), "", "| phase | launches | avg us |", "|---|---:|---:|"]
        out["dominant_by_phase"] = []
        for label, a, b in cuts:
            if b > a:
                avg = sum(d[a:b]) / (b - a)
                lines.append("| %s | %d | %.2f |" % (label, b - a, avg))
                out["dominant_by_phase"].append({"phase": label, "launches": b - a, "avg_us": round(avg, 3)})
        if len(sys.argv) >= 6 and os.path.exists(sys.argv[5]):
            bl = [x for x in open(sys.argv[5]) if x.startswith("{")]
            if bl:
                b = json.loads(bl[-1])
                lines += ["", "Bench line of the same (traced) run: value %.4g %s, ms_per_step %.4f, roofline.avg_launch_ms %.4f over %d bracketed launches of the timed region (HIP events; "
                          "an event pair around a launch reads ~2 %% longer than the trace's begin / end of the same kernel)." % (b["value"], b["unit"], b["ms_per_step"], b["roofline"]["avg_launch_ms"], b["roofline"]["launches"])]
                out["traced_run_bench_line"] = {"value": b["value"], "ms_per_step": b["ms_per_step"], "avg_launch_ms": b["roofline"]["avg_launch_ms"]}
    if len(sys.argv) >= 5:
        for key, db in (("FETCH_SIZE", sys.argv[3]), ("WRITE_SIZE", sys.argv[4])):
            c2 = sqlite3.connect(db).cursor()
            q = ("select kernel_name, count(*), avg(value) from counters_collection c where counter_name=? and grid_size = "
                 "(select max(grid_size) from counters_collection c2 where c2.kernel_name = c.kernel_name) group by kernel_name")
            for name, n, avg in c2.execute(q, (key,)):
                out["counters"].setdefault(short(name), {})[key + "_KiB_avg"] = round(avg, 2)
                out["counters"][short(name)]["dispatches_" + key] = n
        lines += ["", "# HBM counters per dispatch (separate --pmc passes)", "",
                  "| kernel | FETCH_SIZE KiB (raw) | fetched MB (x2 gfx950 correction) | WRITE_SIZE KiB | HBM MB / dispatch |", "|---|---:|---:|---:|---:|"]
        for k, v in out["counters"].items():
            f, w = v.get("FETCH_SIZE_KiB_avg", 0.0), v.get("WRITE_SIZE_KiB_avg", 0.0)
            v["hbm_bytes_per_dispatch"] = int((2 * f + w) * 1024)
            lines.append("| %s | %.1f | %.3f | %.1f | %.3f |" % (k, f, 2 * f * 1024 / 1e6, w, v["hbm_bytes_per_dispatch"] / 1e6))
    os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
    with open(os.path.join(ROOT, "profiles", "%s_rocprof_summary.md" % tag), "w") as fh:
        fh.write("\n".join(lines) + "\n")
    with open(os.path.join(ROOT, "profiles", "%s_rocprof_summary.json" % tag), "w") as fh:
        json.dump(out, fh, indent=1)
    for prefix, fname in (("gram_i8_kernel", "%s_gram_i8_traffic.json"), ("gram_i8p_kernel", "%s_gram_i8_traffic.json"), ("gram_rows_kernel", "%s_gram_traffic.json")):
        # the instantiation that dominates the traced run's time (a run also launches other instantiations of the same template: the sub-batches of
        # the host-buffer leg, the seven-plane leg) -- never "the first name that matches"
        order = [k["kernel"] for k in out["kernels"]]
        gram = sorted((k for k in out["counters"] if k.startswith(prefix + "<") or k == prefix), key=lambda k: order.index(k) if k in order else len(order))
        if gram:
            with open(os.path.join(ROOT, "profiles", fname % tag), "w") as fh:
                json.dump({"kernel": gram[0], "hbm_bytes_per_launch": out["counters"][gram[0]]["hbm_bytes_per_dispatch"],
                           "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), FETCH_SIZE doubled per MI355X_MICROARCH.md"}, fh, indent=1)
    print("\n".join(lines))


if __name__ == "__main__":
    main()
