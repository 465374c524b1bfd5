#!/usr/bin/env python3
"""Turns gpurun_out/r06c/ (written by profiles/collect_r06.sh on the GPU box) into the tracked r06_* files of profiles/.
Fails loudly when a pass is missing: a table that was not collected is an error, never a row of zeros."""
import glob, json, os, shutil, sys
O, P = 'gpurun_out/r06c', 'profiles'
missing = []


def need(path):
    if not os.path.exists(path) or not open(path).read().strip():
        missing.append(path)
        return False
    return True


if need(f'{O}/bench_line.json'):
    line = json.loads(open(f'{O}/bench_line.json').read().strip().splitlines()[-1])
    json.dump(line, open(f'{P}/r06_bench_line.json', 'w'), indent=1)
lines = {}
for f in sorted(glob.glob(f'{O}/line_*.json')):
    tag = os.path.basename(f)[5:-5]
    if not need(f):
        continue
    lines[tag] = json.loads(open(f).read())
    if tag.endswith('_sf4'):
        continue
    if tag == 'demo12':
        if need(f'{O}/kernel_stats_demo12.csv'):
            shutil.copy(f'{O}/kernel_stats_demo12.csv', f'{P}/r06_demo12_kernel_stats.csv')
        continue
    if need(f'{O}/kernel_stats_{tag}.csv'):
        shutil.copy(f'{O}/kernel_stats_{tag}.csv', f'{P}/r06_{tag}_kernel_stats.csv')
json.dump(lines, open(f'{P}/r06_lines.json', 'w'), indent=1)
curve = []
for f in sorted(glob.glob(f'{O}/sf_*.json'), key=lambda f: float(os.path.basename(f)[3:-5])):
    if not need(f):
        continue
    d = json.loads(open(f).read().strip())
    curve.append({"scale_factor": float(os.path.basename(f)[3:-5]), "rows": d["config"]["rows"], "stripes": d["config"]["stripes"], "decoded_GBps": d["value"],
                  "mrows_per_s": d["mrows_per_s"], "ms_per_step": d["ms_per_step"], "phase_ms": d["phase_ms"], "roofline_kernel": (d["roofline"]["kernel"] or "-").split(" ")[0],
                  "roofline_frac": d["roofline"]["frac"], "whole_step_frac": d["roofline"]["whole_step_frac"], "generate_s": d["setup"]["generate_s"],
                  "hbm_bytes_in_use": d.get("hbm_bytes_in_use")})
sf100 = open(f'{O}/sf100_decision.txt').read().strip() if os.path.exists(f'{O}/sf100_decision.txt') else "not attempted"
json.dump({"command": "python bench.py --sf <SF> --steps 5 --warmup 2 --no-cpu --no-e2e (SF 100: --steps 3 --warmup 1 --skip-check)",
           "sf100_on_one_gpu": sf100 + " (twice the device memory in use at SF 50, against 250 GB of the 288)", "curve": curve}, open(f'{P}/r06_sf_curve.json', 'w'), indent=1)
for name in ("select_cost", "reader_rate", "encode_rate"):
    if need(f'{O}/{name}.json'):
        json.dump(json.loads(open(f'{O}/{name}.json').read()), open(f'{P}/r06_{name}.json', 'w'), indent=1)
if need(f'{O}/kernel_stats_encode.csv'):
    shutil.copy(f'{O}/kernel_stats_encode.csv', f'{P}/r06_encode_kernel_stats.csv')
for tl in ("timeline_lineitem_zstd", "timeline_lineitem_zstd_one_lane"):
    if need(f'{O}/{tl}.txt'):
        shutil.copy(f'{O}/{tl}.txt', f'{P}/r06_{tl}.txt')
# run-to-run spread of the headline
rep = []
for f in sorted(glob.glob(f'{O}/repeat_*.json')):
    if need(f):
        d = json.loads(open(f).read().strip())
        rep.append({"ms_per_step": d["ms_per_step"], "value": d["value"], "roofline_frac": d["roofline"]["frac"], "roofline_kernel": d["roofline"]["kernel"].split(" ")[0],
                    "lanes": d.get("lanes")})
if rep:
    ms = [r["ms_per_step"] for r in rep]
    json.dump({"command": "python bench.py --no-cpu --no-e2e --skip-check (five runs, one box)", "ms_per_step_min": min(ms), "ms_per_step_max": max(ms),
               "spread_pct": round(100 * (max(ms) - min(ms)) / min(ms), 2), "runs": rep}, open(f'{P}/r06_repeat.json', 'w'), indent=1)
if need(f'{O}/pmc_per_kernel.json'):
    pmc = json.load(open(f'{O}/pmc_per_kernel.json'))
    out = {"source_digest": open(f'{O}/source_digest.txt').read().strip() if os.path.exists(f'{O}/source_digest.txt') else None,
           "command": "rocprofv3 --kernel-trace --pmc <FETCH_SIZE|WRITE_SIZE> (a pass each, no other trace domain) -- python3 bench.py --workload W "
                      "--compression C --steps 2 --warmup 1 --no-cpu --skip-check --no-e2e (lineitem: the headline's own size, SF 12.5; SF 4 when that pass did not finish -- `pass` says which)",
           "unit": "KB per launch (rocprofv3's unit), average over the launches of the run; per decode call: the sum over all launches of the run divided by its decode calls",
           "note": "gfx950: FETCH_SIZE tallies 128-byte requests at 64 bytes for wide coalesced reads (MI355X_MICROARCH.md, HBM): hbm_bytes_x2 doubles it, "
                   "hbm_bytes_raw does not; kernels that read 1-8 bytes per lane (the entropy / token / execution kernels) sit between the two. "
                   "WRITE_SIZE is exact for 16-byte per-lane stores.  A run makes 4 decode calls of the whole workload (the one behind staging, one warm-up, two timed) plus the call of the "
                   "copy-back sample (up to 4 single stripes): `calls` counts the launches of decompress_finalize_kernel (compressed workloads: one per column lane "
                   "and call) or summary_to_host_kernel; the per-call totals divide by the 4 full calls + the sample's share of a call (its rows).",
           "workloads": {}}
    big = pmc.get("lineitem_zstd_sf12_FETCH_SIZE") and pmc.get("lineitem_zstd_sf12_WRITE_SIZE")
    for wl, line_tag in ((("lineitem_zstd_sf12", "lineitem_zstd") if big else ("lineitem_zstd", "lineitem_zstd_sf4")), ("c3_none", "c3_none"), ("c2", "c2")):
        fe, wr = pmc.get(wl + "_FETCH_SIZE"), pmc.get(wl + "_WRITE_SIZE")
        out_name = "lineitem_zstd" if wl.startswith("lineitem_zstd") else wl
        if not fe or not wr:
            missing.append(f"PMC pass of {wl}: " + ("FETCH_SIZE " if not fe else "") + ("WRITE_SIZE" if not wr else ""))
            continue
        if line_tag not in lines:
            missing.append(f"bench line {line_tag}")
            continue
        L = lines[line_tag]
        ks = {}
        for k in sorted(set(fe) | set(wr), key=lambda k: -(fe.get(k, {}).get("sum", 0) + wr.get(k, {}).get("sum", 0))):
            f_, w_ = fe.get(k, {}), wr.get(k, {})
            ks[k] = {"launches": f_.get("launches", w_.get("launches")), "FETCH_SIZE_KB_avg": round(f_.get("avg", 0.0), 1), "WRITE_SIZE_KB_avg": round(w_.get("avg", 0.0), 1),
                     "FETCH_KB_run": round(f_.get("sum", 0.0), 1), "WRITE_KB_run": round(w_.get("sum", 0.0), 1)}
        if all(v["WRITE_KB_run"] == 0 for v in ks.values()) or all(v["FETCH_KB_run"] == 0 for v in ks.values()):
            missing.append(f"PMC pass of {wl}: a counter is zero for every kernel (the pass did not run to its end)")
            continue
        rows_full, fetched = L["config"]["rows"], 0
        # the run: 4 full calls (the first decode, 1 warm-up, 2 steps) + the copy-back sample; weight of the sample = its rows over a full call's rows
        sample = 0.0
        if L.get("d2h_fetch_sample"):
            n_s, n_all = [int(x) for x in L["d2h_fetch_sample"].replace(" stripes", "").split(" of ")]
            sample = n_s / max(1, n_all)
        calls = 4 + sample  # bench.py --steps 2 --warmup 1: the decode behind staging (the checks' input), one warm-up step, two timed steps -- FOUR full calls -- + the copy-back sample.  (summarise_r04.py divided by 3 + sample: its ratios are 4.36 / 3.36 = 1.30 x too high)
        tot_f = sum(v["FETCH_KB_run"] for v in ks.values()) * 1024 / calls
        tot_w = sum(v["WRITE_KB_run"] for v in ks.values()) * 1024 / calls
        algo = L["stream_bytes_in"] + L["arrow_bytes_out"]
        out["workloads"][out_name] = {"pass": wl, "bench_line": line_tag, "algorithmic_bytes_per_step": algo, "stream_bytes_in": L["stream_bytes_in"], "arrow_bytes_out": L["arrow_bytes_out"],
                                "decode_calls_in_run": round(calls, 3), "hbm_read_bytes_per_step_raw": int(tot_f), "hbm_read_bytes_per_step_x2": int(2 * tot_f),
                                "hbm_write_bytes_per_step": int(tot_w),
                                "traffic_over_algorithmic_raw": round((tot_f + tot_w) / algo, 2), "traffic_over_algorithmic_x2": round((2 * tot_f + tot_w) / algo, 2),
                                "kernels": ks}
    json.dump(out, open(f'{P}/r06_pmc_traffic.json', 'w'), indent=1)
    for wl, v in out["workloads"].items():
        print(wl, "algorithmic", v["algorithmic_bytes_per_step"], "read raw", v["hbm_read_bytes_per_step_raw"], "write", v["hbm_write_bytes_per_step"],
              "ratio raw / x2", v["traffic_over_algorithmic_raw"], v["traffic_over_algorithmic_x2"])
if missing:
    print("MISSING -- collect again (profiles/collect_one.sh runs one workload):", file=sys.stderr)
    for m in missing:
        print("   ", m, file=sys.stderr)
    sys.exit(1)
print(json.dumps({k: line[k] for k in ("value", "ms_per_step", "roofline")})[:700])
