"""profiles/<tag>_pmc.md: the counter rows of both int8 Gram kernels (tools/gpu_profiles.sh -> <tag>_pmc_rows.json) as one A/B table, the ablation
probes of the release kernel (<tag>_i8p_ablate.jsonl) and the kernel-trace summary.  usage: pmc_md.py r04"""
import collections, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r04"
P = lambda name: os.path.join(ROOT, "profiles", "%s_%s" % (tag, name))
rows = json.load(open(P("pmc_rows.json")))
t = collections.defaultdict(dict)
for r in rows:
    if r["run"].startswith("prof_i8"):
        t[r["kernel"]][r["counter"]] = (r["avg"], r["avg_duration_ns"], r["dispatches"])
new = next(v for k, v in t.items() if k.startswith("gram_i8p_kernel<6"))
old = next(v for k, v in t.items() if k.startswith("gram_i8_kernel<6"))
nname = next(k for k in t if k.startswith("gram_i8p_kernel<6")); oname = next(k for k in t if k.startswith("gram_i8_kernel<6"))
g = lambda d, k: d[k][0]
pct = lambda d, k: 100.0 * g(d, k) / g(d, "SQ_WAVE_CYCLES")
busy = lambda d: 100 * g(d, "SQ_VALU_MFMA_BUSY_CYCLES") / g(d, "GRBM_GUI_ACTIVE") / 128
hit = lambda d: 100 * g(d, "TCC_HIT_sum") / (g(d, "TCC_HIT_sum") + g(d, "TCC_MISS_sum"))
L = ["# PMC counters of the round-4 Gram kernel beside the round-3 one (rocprofv3 --pmc, separate passes; raw rows: %s_pmc_rows.json)" % tag, "",
     "Command: `bash tools/gpu_profiles.sh %s`; the counter passes run `tools/i8p_counters.py` -- full-size launches (10k x 60 x 6, 5,000 replicates, six digit planes, the" % tag,
     "automatic cut 12 tall + 5 short tile rows x 60 pair tiles = 1,020 workgroups) of BOTH kernels in one process, so every row below is the same pass on the same box.",
     "SQ cycle counters are in units of 4 clocks; the counters of one XCD are reported (128 SIMDs).  Durations under the counters are ~8 % above the un-profiled ones.", "",
     "| quantity | `%s` (round 4: 4 waves, count fragments from global memory) | `%s` (round 3: 8 waves, both operands through LDS) |" % (nname, oname), "|---|---:|---:|",
     "| duration under the counters | %.1f us | %.1f us |" % (new["GRBM_GUI_ACTIVE"][1] / 1e3, old["GRBM_GUI_ACTIVE"][1] / 1e3),
     "| matrix pipe busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE x 128) | %.3e / %.3e: **%.1f %%** | %.3e / %.3e: %.1f %% |" % (g(new, "SQ_VALU_MFMA_BUSY_CYCLES"), g(new, "GRBM_GUI_ACTIVE"), busy(new), g(old, "SQ_VALU_MFMA_BUSY_CYCLES"), g(old, "GRBM_GUI_ACTIVE"), busy(old)),
     "| executed int8 ops = SQ_INSTS_VALU_MFMA_MOPS_I8 x 512 | %.4e | %.4e (the same tile cut: the same padding) |" % (g(new, "SQ_INSTS_VALU_MFMA_MOPS_I8") * 512, g(old, "SQ_INSTS_VALU_MFMA_MOPS_I8") * 512),
     "| SQ_WAVE_CYCLES | %.3e (4 waves per workgroup) | %.3e (8 waves) |" % (g(new, "SQ_WAVE_CYCLES"), g(old, "SQ_WAVE_CYCLES")),
     "| SQ_WAIT_ANY / SQ_WAIT_INST_ANY / SQ_ACTIVE_INST_ANY (share of the wave cycles) | %.1f %% / %.1f %% / %.1f %% | %.1f %% / %.1f %% / %.1f %% |" % (pct(new, "SQ_WAIT_ANY"), pct(new, "SQ_WAIT_INST_ANY"), pct(new, "SQ_ACTIVE_INST_ANY"), pct(old, "SQ_WAIT_ANY"), pct(old, "SQ_WAIT_INST_ANY"), pct(old, "SQ_ACTIVE_INST_ANY")),
     "| wave instructions per launch: VALU (of them 36.4 M MFMA) / SALU / LDS / VMEM reads / branches | %.1f M / %.1f M / %.2f M / %.2f M / %.2f M | %.1f M / %.1f M / %.2f M / %.2f M / %.2f M |" % tuple(g(d, k) / 1e6 for d in (new, old) for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_BRANCH")),
     "| LDS: SQ_LDS_IDX_ACTIVE (array cycles) / SQ_WAIT_INST_LDS / bank conflicts | %.3e / %.3e / %d | %.3e / %.3e / %d |" % tuple(g(d, k) for d in (new, old) for k in ("SQ_LDS_IDX_ACTIVE", "SQ_WAIT_INST_LDS", "SQ_LDS_BANK_CONFLICT")),
     "| L2: hit rate, EA read requests | %.1f %%, %.3e | %.1f %%, %.3e |" % (hit(new), g(new, "TCC_EA0_RDREQ_sum"), hit(old), g(old, "TCC_EA0_RDREQ_sum")),
     "| instruction cache requests / misses | %.2e / %d | %.2e / %d |" % (g(new, "SQC_ICACHE_REQ"), g(new, "SQC_ICACHE_MISSES"), g(old, "SQC_ICACHE_REQ"), g(old, "SQC_ICACHE_MISSES")), "",
     "Reading.  The same 36.4 M MFMAs and the same 5.06 M one-KB VMEM instructions per launch (32 per CU and k-step in both kernels: 20 count + 12 digit blocks) --",
     "what changed is what they cost the matrix pipe.  In the round-3 kernel all 32 are LDS-DMA instructions and every wave reads 11 fragments back; here 20 of the",
     "32 are plain `global_load_dwordx4` into the wave's own registers and the four waves read the 12 digit blocks (LDS instructions 14.3 M -> 7.8 M, LDS-array",
     "cycles 5.8e7 -> 3.2e7, scalar instructions 38 M -> 10 M: one M0 write per k-step instead of one per block, no per-block address arithmetic).  The waves of",
     "the round-3 kernel sat parked at `s_waitcnt` / the barrier 30 % of their cycles and stalled at issue 44 %; the single wave per SIMD of this kernel is parked",
     "9 % and stalled at issue 54 % -- that stall IS the matrix pipe now (a lone wave waits for the pipe to take its next MFMA), hence 76 % pipe busy against 65 %.", ""]
if os.path.exists(P("i8p_ablate.jsonl")) and os.path.getsize(P("i8p_ablate.jsonl")):
    abl = [json.loads(l) for l in open(P("i8p_ablate.jsonl")) if l.startswith("{")]
    L += ["Where the remaining 24 %% go -- ablation probes (`%s_i8p_ablate.jsonl`, experiments build: the release kernel with one ingredient taken out; results garbage, only the" % tag,
          "kernel time is read; B = 5,120 on the all-tall grid of 960 tiles unless the row says otherwise):", "", "| probe | Gram ms (min of 4 alternating rounds) |", "|---|---:|"]
    for a in abl:
        L.append("| %s (%d short rows) | %.4f |" % (a["config"], a["short_rows"], a["gram_ms_min"]))
    L.append("")
L += ["`tools/ubench/mfma_i8_fillers.hip` (`%s_ubench_mfma_i8_fillers.jsonl`) prices the same fillers between bare MFMAs of one wave per SIMD: 60 MFMAs 1,003 clocks; + 12" % tag,
      "ds_read_b128 +20-40; + 5 global_load_dwordx4 +20; + 3 global_load_lds_dwordx4 +50-60; all three together +150-220 (more than their sum: VMEM and LDS instructions queue",
      "behind each other); a 32x32x32 MFMA stream pays 28 clocks per ds_read_b128 (not an option); the digit blocks through registers (`global_load` + `ds_write_b128`) cost 150",
      "clocks per three blocks against 55 for three LDS-DMAs; staggering the four waves' fillers changes nothing.  What was tried on the kernel itself (`%s_i8p_variants.jsonl`):" % tag,
      "digit blocks through staging registers 0.364 against 0.352 ms (automatic cut); filler schedules strided / one-per-gap with the DMAs last / VMEM evenly spaced 0.340 / 0.335 /",
      "**0.333**; one barrier per two k-steps on a five-stage ring 0.334 (no gain).", ""]
if os.path.exists(P("rocprof_summary.json")):
    summ = json.load(open(P("rocprof_summary.json")))
    L.append("Kernel-trace of the bench command (`%s_rocprof_summary.md`): " % tag + ", ".join("%s %.1f us" % (k["kernel"].split("<")[0], k["avg_us"]) for k in summ["kernels"][:6]) + ".")
    bp = summ.get("dominant_by_phase", [])
    if bp:
        L.append("Dominant kernel by phase: " + "; ".join("%s %.1f us" % (b["phase"].split(":")[0].split("(")[0].strip(), b["avg_us"]) for b in bp) + ".")
if os.path.exists(P("bench_n1.json")):
    b = json.loads(open(P("bench_n1.json")).read().strip().splitlines()[-1])
    L.append("Bench line of the same box (`%s_bench_n1.json`): value %.4g replicates/s, ms_per_step %.4f, roofline.avg_launch_ms %.4f over %d launches, frac %.3f of nominal / %.3f of the measured ceiling; the round-3 kernel in the same line: %.4f ms." %
             (tag, b["value"], b["ms_per_step"], b["roofline"]["avg_launch_ms"], b["roofline"]["launches"], b["roofline"]["frac"], b["roofline"]["frac_of_measured_ceiling"], b.get("round3_gram_kernel", {}).get("gram_avg_launch_ms", float("nan"))))
open(P("pmc.md"), "w").write("\n".join(L) + "\n")
print("\n".join(L))
