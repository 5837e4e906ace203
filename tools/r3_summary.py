#!/usr/bin/env python
"""profiles/r3_bench_*.json -> the numbers block at the top of profiles/r3_summary.txt (the file list below it is kept)."""
import json, os
P = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")
L = lambda n: json.load(open(os.path.join(P, n)))
names = {"B": "plane-sweep grid 256x192x64cand (image 1024x768)      ", "S": "ScanNet demo 384x256 image, grid 96x64x64            ",
         "K": "KITTI 768x256 image, grid 192x64x64                   ", "H": "480x640 image, grid 160x120x128                       "}
out = ["Round 3 — what is in profiles/ and the numbers that matter (one MI355X; files written by tools/final_check.sh, last run at the end of the round)",
       "=" * 142, ""]
for c in "BSKH":
    d = L("r3_bench_%s.json" % c)
    cb, rf, rm, p = d["cpu_baseline"], d["roofline"], d["roofline_mfma"], d["parity"]
    out.append("r3_bench_%s.json   %s %7.2f frames/s  %6.2f ms/frame   CPU (oracle, %d threads) %.4f fps = %.0fx" %
               (c, names[c], d["value"], d["ms_per_step"], cb["cores"], cb["value"], d["value"] / cb["value"]))
    out.append("                   sampling kernel %.1f us = %.1f GB/s algorithmic = %.2f %% of 8 TB/s; HBM-side traffic %s B/launch = %.1fx algorithmic (%s)" %
               (rf["kernel_ms"] * 1e3, rf["achieved"], 100 * rf["frac"], rf["traffic"], (rf["traffic"] or 0) / rf["algorithmic_bytes"],
                "measured in the bench run by two rocprofv3 --pmc child passes" if "measured in this run" in rf["traffic_source"] else "committed PMC file"))
    out.append("                   K-Net layer (wino_dw plain) %.3f ms = %.1f TFLOP/s issued = %.1f %% of the fp32 matrix peak (%.0f TFLOP/s direct-conv equivalent)" %
               (rm["kernel_ms"], rm["achieved"], 100 * rm["frac"], rm["direct_conv_equivalent_tflops"]))
    out.append("                   parity vs oracle: L1 refined/DPV/BV_cur/BV_predict %.1e / %.1e / %.1e / %.1e; max DPV %.1e, BV_predict %.1e; arg-max flips %d/%d/%d (beyond a tie: %d); pass %s, pass_strict %s" %
               (p["refined"]["mean"], p["dpv"]["mean"], p["bv_cur"]["mean"], p["bv_predict"]["mean"], p["dpv"]["max"], p["bv_predict"]["max"],
                p["refined"]["argmax_mismatch"], p["dpv"]["argmax_mismatch"], p["bv_cur"]["argmax_mismatch"],
                sum(p[k]["argmax_mismatch_beyond_tie_1e-3"] for k in ("refined", "dpv", "bv_cur")), p["pass"], p["pass_strict"]))
d = L("r3_bench_H_300frames.json")
out.append("r3_bench_H_300frames.json  BASELINE config 5 (480x640, D=128, 300 consecutive frames of one stream): %.2f frames/s sustained, %.2f ms/frame, %.1f GB peak HBM" %
           (d["value"], d["ms_per_step"], d["config"].get("peak_hbm_gb", 0)))
d = L("r3_bench_train.json"); cb = d["cpu_baseline"]
out.append("r3_bench_train.json        training iteration at the config-T grid (%s): %.2f ms (%.1f windows/s); CPU (oracle/train_oracle.py, %d threads) %.3f windows/s = %.0fx" %
           (d["config"].get("launch"), d["ms_per_step"], d["value"], cb["cores"], cb["value"], d["value"] / cb["value"]))
out.append("")
path = os.path.join(P, "r3_summary.txt")
old = open(path).read()
tail = old[old.index("r3_bench_{B,S,K}_kernel_stats.csv"):]
open(path, "w").write("\n".join(out) + "\n" + tail)
print("\n".join(out[3:]))
