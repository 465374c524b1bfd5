#!/usr/bin/env python3
"""Turns gpurun_out/r03c/ (written by profiles/collect_r03.sh on the GPU box) into the tracked r03_* files of profiles/."""
import glob, json, os, shutil
O, P = 'gpurun_out/r03c', 'profiles'
line = json.loads(open(f'{O}/bench_line.json').read().strip().splitlines()[-1])
json.dump(line, open(f'{P}/r03_bench_line.json', 'w'), indent=1)
lines = {}
for f in sorted(glob.glob(f'{O}/line_*.json')):
    tag = os.path.basename(f)[5:-5]
    lines[tag] = json.loads(open(f).read())
    if os.path.exists(f'{O}/kernel_stats_{tag}.csv'):  # (a profiler pass that ran out of time leaves none)
        shutil.copy(f'{O}/kernel_stats_{tag}.csv', f'{P}/r03_{tag}_kernel_stats.csv')
json.dump(lines, open(f'{P}/r03_lines.json', 'w'), indent=1)
# the scale-factor curve of the headline workload
curve = []
for f in sorted(glob.glob(f'{O}/sf_*.json'), key=lambda f: float(os.path.basename(f)[3:-5])):
    t = open(f).read().strip()
    if not t:
        continue
    d = json.loads(t)
    curve.append({"scale_factor": float(os.path.basename(f)[3:-5]), "rows": d["config"]["rows"], "stripes": d["config"]["stripes"], "decoded_GBps": d["value"],
                  "mrows_per_s": d["mrows_per_s"], "ms_per_step": d["ms_per_step"], "phase_ms": d["phase_ms"], "roofline_frac": d["roofline"]["frac"],
                  "whole_step_frac": d["roofline"]["whole_step_frac"], "generate_s": d["setup"]["generate_s"]})
json.dump({"command": "python bench.py --sf <SF> --steps 5 --warmup 2 --no-cpu --no-e2e", "curve": curve}, open(f'{P}/r03_sf_curve.json', 'w'), indent=1)
if os.path.exists(f'{O}/select_cost.json') and open(f'{O}/select_cost.json').read().strip():
    json.dump(json.loads(open(f'{O}/select_cost.json').read()), open(f'{P}/r03_select_cost.json', 'w'), indent=1)
pmc = json.load(open(f'{O}/pmc_per_kernel.json'))
out = {"command": "rocprofv3 --kernel-trace --pmc <FETCH_SIZE|WRITE_SIZE> (a pass each, no other trace domain) -- python3 bench.py --workload W "
                  "--compression C --steps 2 --warmup 1 --no-cpu --skip-check --no-e2e (lineitem: --sf 1)",
       "unit": "KB per launch (rocprofv3's unit), average over the launches of the run",
       "note": "gfx950: FETCH_SIZE tallies 128-byte requests at 64 bytes for wide coalesced reads (MI355X_MICROARCH.md, HBM): hbm_bytes_x2 doubles it, "
               "hbm_bytes_raw does not; kernels that read 1-8 bytes per lane (the entropy / token / execution kernels) sit between the two. "
               "WRITE_SIZE is exact for 16-byte per-lane stores.",
       "workloads": {}}
for wl in ("lineitem_zstd", "c3_none", "c2"):
    fe, wr = pmc.get(wl + "_FETCH_SIZE", {}), pmc.get(wl + "_WRITE_SIZE", {})
    ks = {}
    for k in sorted(set(fe) | set(wr), key=lambda k: -(fe.get(k, {}).get("avg", 0) + wr.get(k, {}).get("avg", 0))):
        f_, w_ = fe.get(k, {}).get("avg", 0.0), wr.get(k, {}).get("avg", 0.0)
        ks[k] = {"launches": fe.get(k, wr.get(k))["launches"], "FETCH_SIZE_KB": round(f_, 1), "WRITE_SIZE_KB": round(w_, 1),
                 "hbm_bytes_raw": int((f_ + w_) * 1024), "hbm_bytes_x2": int((2 * f_ + w_) * 1024)}
    L = lines[wl] if wl != "lineitem_zstd" else json.loads(open(f'{O}/sf_1.json').read())  # (the counters of the headline were taken at SF 1)
    out["workloads"][wl] = {"algorithmic_bytes_per_step": L["stream_bytes_in"] + L["arrow_bytes_out"], "stream_bytes_in": L["stream_bytes_in"],
                            "arrow_bytes_out": L["arrow_bytes_out"],
                            "kernels": ks}
    # per decode call every kernel of the table runs `launches / steps` times; the sum over kernels of (bytes per launch x launches per step):
    steps = ks["rle2_expand_kernel"]["launches"]  # decode calls of the run (rle2_expand_kernel is launched once per call)
    tot_raw = sum(v["hbm_bytes_raw"] * v["launches"] / steps for v in ks.values())
    tot_x2 = sum(v["hbm_bytes_x2"] * v["launches"] / steps for v in ks.values())
    out["workloads"][wl]["all_kernels_hbm_bytes_raw"] = int(tot_raw)
    out["workloads"][wl]["all_kernels_hbm_bytes_x2"] = int(tot_x2)
    out["workloads"][wl]["traffic_over_algorithmic_raw"] = round(tot_raw / out["workloads"][wl]["algorithmic_bytes_per_step"], 2)
    out["workloads"][wl]["traffic_over_algorithmic_x2"] = round(tot_x2 / out["workloads"][wl]["algorithmic_bytes_per_step"], 2)
json.dump(out, open(f'{P}/r03_pmc_traffic.json', 'w'), indent=1)
for wl, v in out["workloads"].items():
    print(wl, "algorithmic", v["algorithmic_bytes_per_step"], "traffic raw", v["all_kernels_hbm_bytes_raw"], "x2", v["all_kernels_hbm_bytes_x2"],
          v["traffic_over_algorithmic_raw"], v["traffic_over_algorithmic_x2"])
print(json.dumps({k: line[k] for k in ("value", "ms_per_step", "roofline")})[:600])
