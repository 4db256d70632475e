#!/usr/bin/env python3
"""Summarise what tools/collect_profiles.sh wrote (rocprofv3 csv output) into the two small files committed
under profiles/:  <dir>/kernel_stats.md  (per-kernel call count / total / average from the kernel trace) and
<dir>/pmc_counters.json (mean counter value per launch and kernel, plus the derived mel-decoder figures).
usage: prof_summary.py gpurun_out/<tag>"""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict


def short(name):
    m = re.search(r"esmi::(\w+(?:<[^>]*>)?)", name)
    return m.group(1) if m else name[:60]


def find(d, pat):
    r = sorted(glob.glob(os.path.join(d, "**", pat), recursive=True))
    return r[0] if r else None


def kernel_stats(d):
    path = find(os.path.join(d, "stats"), "*kernel_trace.csv")
    if not path:
        return None
    agg = defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        a = agg[r["Kernel_Name"]]
        a[0] += 1
        a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    tot = sum(v[1] for v in agg.values())
    rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
    lines = ["| kernel | calls | total (us) | avg (us) | % |", "|---|---:|---:|---:|---:|"]
    for name, (n, t) in rows:
        lines.append(f"| `{name[:100]}` | {n} | {t:.1f} | {t / n:.2f} | {100 * t / tot:.2f} |")
    return "\n".join(lines)


def pmc(d):
    per = defaultdict(dict)
    for cdir in sorted(glob.glob(os.path.join(d, "pmc_*"))):
        if not os.path.isdir(cdir):
            continue
        path = find(cdir, "*counter_collection.csv")
        if not path:
            continue
        acc = defaultdict(lambda: defaultdict(float))
        for r in csv.DictReader(open(path)):
            acc[(r["Kernel_Name"], r["Dispatch_Id"])][r["Counter_Name"]] += float(r["Counter_Value"])
        byk = defaultdict(lambda: defaultdict(list))
        for (k, _), cs in acc.items():
            if "esmi::" in k:
                for c, v in cs.items():
                    byk[short(k)][c].append(v)
        for k, cs in byk.items():
            for c, vs in cs.items():
                per[k][c] = sum(vs) / len(vs)
    return per


def workload_tag(bench):
    """Same tag bench.py's pmc_traffic() matches on: '<cfg> ES B=.. T=.. D-const .. (...)'."""
    if not bench:
        return ""
    c = bench["config"]
    B, T = c["global_batch"] // bench["n_gpus"], c["phonemes"]
    dur = c["frames_per_step"] // (c["global_batch"] * T)
    return (f"{c['workload'].split()[0]} ES B={B} T={T} D-const {dur} ({B * T * dur} frames per launch), "
            f"{bench['n_gpus']}x MI355X")


def trace_avg_us(d, kernel_prefix):
    path = find(os.path.join(d, "stats"), "*kernel_trace.csv")
    if not path:
        return None
    ts = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in csv.DictReader(open(path)) if kernel_prefix in r["Kernel_Name"]]
    return sum(ts) / len(ts) if ts else None


def scratch_bytes(d, kernel_prefix):
    path = find(os.path.join(d, "stats"), "*kernel_trace.csv")
    if path:
        for r in csv.DictReader(open(path)):
            if kernel_prefix in r["Kernel_Name"]:
                return int(r["Scratch_Size"])
    return None


def main():
    d = sys.argv[1]
    bench = {}
    try:
        bench = json.loads(open(os.path.join(d, "bench_unprofiled.json")).read().strip().splitlines()[-1])
    except Exception:
        pass
    ks = kernel_stats(d)
    if ks:
        with open(os.path.join(d, "kernel_stats.md"), "w") as f:
            f.write("# rocprofv3 --kernel-trace --stats (csv) summary\n\n"
                    "`rocprofv3 --kernel-trace --stats --output-format csv -- python bench.py --steps 10 --warmup 3 "
                    "--event-every 1 --no-cpu-baseline` (tools/collect_profiles.sh), 1x MI355X; 13 forward steps.\n")
            if bench:
                f.write(f"Un-profiled `bench.py` of the same build on the same box: {bench['ms_per_step']:.3f} ms/step = "
                        f"{bench['value']:.3e} frames/s, {bench['roofline']['kernel']} {bench['roofline']['kernel_ms']:.3f} ms "
                        f"by live HIP events ({100 * bench['roofline']['frac']:.1f} % of roofline.peak = {bench['roofline']['peak']:.0f} TFLOP/s).\n")
            try:   # the live HIP-event timing printed by bench.py INSIDE the profiled run: must agree with the trace average
                prof = json.loads([ln for ln in open(os.path.join(d, "stats.log")).read().splitlines() if ln.startswith("{")][-1])
                f.write(f"Inside the profiled run, bench.py's live HIP events (every launch of the 10 timed steps) gave "
                        f"{prof['roofline']['kernel_ms']:.4f} ms for {prof['roofline']['kernel']}: compare with the trace average below "
                        f"(13 launches incl. 3 warm-up).\n")
            except Exception:
                pass
            f.write("\n" + ks + "\n")
        print(ks)
    per = pmc(d)
    if per:
        out = {"command": "rocprofv3 --pmc <group> --kernel-trace --output-format csv -- python bench.py --steps 3 --warmup 2 "
                          "--no-cpu-baseline; one pass per counter group (tools/collect_profiles.sh)",
               "workload": workload_tag(bench),
               "per_kernel_mean_per_launch": per}
        dec = next((k for k in per if k.startswith("mel_decoder_kernel")), None)
        if dec and "FETCH_SIZE" in per[dec] and "WRITE_SIZE" in per[dec]:
            m = per[dec]
            fr, wr = m["FETCH_SIZE"] * 1024, m["WRITE_SIZE"] * 1024
            o = {"fetch_bytes_raw": fr, "fetch_bytes_corrected_x2": 2 * fr, "write_bytes": wr,
                 "hbm_traffic_bytes_corrected": 2 * fr + wr,
                 "note": "gfx950 rocprofv3 reports FETCH_SIZE at 1/2 for wide coalesced 16 B/lane reads "
                         "(MI355X_MICROARCH.md, HBM section); the decoder's global reads are 16 B/lane row gathers and "
                         "weight-slice loads, so the read side is doubled."}
            if "SQ_VALU_MFMA_BUSY_CYCLES" in m and "GRBM_GUI_ACTIVE" in m:
                # GRBM_GUI_ACTIVE is summed over the 8 XCDs; 1024 SIMDs = 128 per XCD
                o["mfma_pipe_utilisation"] = m["SQ_VALU_MFMA_BUSY_CYCLES"] / (m["GRBM_GUI_ACTIVE"] * 128)
            if "SQ_INSTS_MFMA" in m:
                o["mfma_insts"] = m["SQ_INSTS_MFMA"]
            if "SQ_LDS_BANK_CONFLICT" in m and "SQ_ACTIVE_INST_LDS" in m:
                o["lds_bank_conflict_over_active"] = m["SQ_LDS_BANK_CONFLICT"] / m["SQ_ACTIVE_INST_LDS"]
            # per-pipe busy fractions with rocprofiler-sdk's own derived-metric definitions for gfx950 (counter_defs.yaml: MfmaUtil,
            # VALUBusy, LdsUtil); GRBM_GUI_ACTIVE is summed over the 8 XCDs here (the definitions take its max = sum / 8),
            # CU_NUM = 256, SIMD_NUM = 1024
            if "GRBM_GUI_ACTIVE" in m:
                gui = m["GRBM_GUI_ACTIVE"] / 8.0
                pipes = {}
                if "SQ_VALU_MFMA_BUSY_CYCLES" in m:
                    pipes["MfmaUtil"] = m["SQ_VALU_MFMA_BUSY_CYCLES"] / (gui * 1024)
                if "SQ_ACTIVE_INST_VALU" in m:
                    pipes["VALUBusy"] = m["SQ_ACTIVE_INST_VALU"] / 256 / gui
                if "SQ_LDS_IDX_ACTIVE" in m:
                    pipes["LdsUtil"] = m["SQ_LDS_IDX_ACTIVE"] / (gui * 256)
                if "SQ_VALU_MFMA_COEXEC_CYCLES" in m:
                    pipes["valu_mfma_coexec_over_gui_simd"] = m["SQ_VALU_MFMA_COEXEC_CYCLES"] / (gui * 1024)
                if "SQ_WAIT_INST_ANY" in m and "SQ_WAVE_CYCLES" in m:
                    pipes["wait_inst_any_over_wave_cycles"] = m["SQ_WAIT_INST_ANY"] / m["SQ_WAVE_CYCLES"]
                tr = trace_avg_us(d, "mel_decoder_kernel")
                if tr:
                    pipes["shader_ghz_under_tracer"] = gui / tr / 1e3      # busy cycles of one XCD / trace-average duration
                    pipes["trace_avg_us"] = tr
                pipes["definitions"] = "rocprofiler-sdk counter_defs.yaml (gfx950): MfmaUtil = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE x SIMD_NUM), VALUBusy = SQ_ACTIVE_INST_VALU / CU_NUM / GRBM_GUI_ACTIVE, LdsUtil = SQ_LDS_IDX_ACTIVE / (GRBM_GUI_ACTIVE x CU_NUM)"
                o["pipes"] = pipes
            sc = scratch_bytes(d, "mel_decoder_kernel")
            if sc is not None:
                o["scratch_bytes_per_lane"] = sc
            out["mel_decoder"] = o
        json.dump(out, open(os.path.join(d, "pmc_counters.json"), "w"), indent=1)
        print(json.dumps(out.get("mel_decoder", {}), indent=1))


if __name__ == "__main__":
    main()
