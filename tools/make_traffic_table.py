#!/usr/bin/env python3
"""profiles/traffic_table.json (what bench.py's `roofline.traffic` reads) from a round's PMC pass:  python tools/make_traffic_table.py profiles/r6_final_pmc_traffic.json r6
Maps the kernels that can be a bench line's dominant kernel to their bench keys; HBM bytes per launch = FETCH_SIZE x 2 (gfx950 correction for 16 B / lane streams,
MI355X_MICROARCH.md) + WRITE_SIZE of the grid with the most traffic (the 720p / 32-frame launch), as tools/pmc_summary.py aggregates the separate rocprofv3 --pmc passes."""
import json, os, sys
src, tag = sys.argv[1], sys.argv[2]
t = json.load(open(src))
KEYS = {"attention[spatial,d40]": ("attn40q2_kernel", 1.18e9, "Q, K, V read 0.885 GB, O written 0.295 GB"),
        "attention[spatial,d80]": ("attn80_kernel", 0.59e9, "Q, K, V 0.442 GB, O 0.147 GB"),
        "spatial_chain_fused[c320]": ("chain_rs_c320_kernel", 460800 * 320 * (2 + 4 + 4 + 4), "o h16, t_in / x / out fp32"),
        "spatial_chain_front_fused[c320]": ("chain_front_rs_c320_kernel", 460800 * 320 * (4 + 4 + 6), "x / t fp32, qkv h16"),
        "motion:motion_module_fused[c320]": ("motion_c320_kernel", 460800 * 320 * (4 + 4 + 4), "x, res1 / out fp32")}
out = {}
for key, (kname, alg, what) in KEYS.items():
    hits = [(k, v) for k, v in t.items() if kname in k and "F16" in k]
    if not hits:
        continue
    k, v = hits[0]
    e = v[0]
    tot = e["fetch_corrected_bytes_per_launch"] + e["write_bytes_per_launch"]
    out[key] = {"hbm_bytes_per_launch": tot, "fetch_x2_bytes": e["fetch_corrected_bytes_per_launch"], "write_bytes": e["write_bytes_per_launch"], "algorithmic_bytes": alg,
                "note": f"round {tag} final tree (kernel {k.split('::')[-1].split('(')[0]}, grid {e['grid']}, {e['launches']} launches): rocprofv3 --pmc FETCH_SIZE and WRITE_SIZE in separate passes on "
                        f"`bench.py --steps 1 --warmup 0 --denoise-steps 2 --no-cpu-baseline --no-kernel-events --no-power-trace` (tools/profile_round.sh -> {os.path.basename(src)}): raw FETCH x2 "
                        f"{e['fetch_corrected_bytes_per_launch'] / 1e9:.3f} GB + WRITE {e['write_bytes_per_launch'] / 1e9:.3f} GB per launch against {alg / 1e9:.3f} GB algorithmic ({what}) = {tot / alg:.2f}x"}
json.dump(out, open(os.path.join(os.path.dirname(src), "traffic_table.json"), "w"), indent=1)
for k, v in out.items():
    print(f"{k:36s} {v['hbm_bytes_per_launch'] / 1e9:.3f} GB  fetch x2 {v['fetch_x2_bytes'] / 1e9:.3f}  write {v['write_bytes'] / 1e9:.3f}  alg {v['algorithmic_bytes'] / 1e9:.3f}  = {v['hbm_bytes_per_launch'] / v['algorithmic_bytes']:.2f}x")
