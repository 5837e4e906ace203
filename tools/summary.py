#!/usr/bin/env python
"""profiles/<round>_bench_*.json -> profiles/<round>_summary.txt (python tools/summary.py r4): the numbers that matter + the file list."""
import json, os, sys
R = sys.argv[1] if len(sys.argv) > 1 else "r4"
P = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")
L = lambda n: json.load(open(os.path.join(P, n)))
names = {"B": "plane-sweep grid 256x192x64cand (image 1024x768)      ", "S": "ScanNet demo 384x256 image, grid 96x64x64            ",
         "K": "KITTI 768x256 image, grid 192x64x64                   ", "H": "480x640 image, grid 160x120x128                       "}
out = ["Round %s — what is in profiles/ and the numbers that matter (one MI355X; files written by tools/final_check.sh, last run at the end of the round)" % R[1:],
       "=" * 142, ""]
for c in "BSKH":
    d = L("%s_bench_%s.json" % (R, c))
    cb, rf, rm, p = d["cpu_baseline"], d["roofline"], d["roofline_mfma"], d["parity"]
    out.append(R + "_bench_%s.json   %s %7.2f frames/s  %6.2f ms/frame   CPU (oracle, %d threads) %.4f fps = %.0fx" %
               (c, names[c], d["value"], d["ms_per_step"], cb["cores"], cb["value"], d["value"] / cb["value"]))
    out.append("                   sampling kernel %.1f us = %.1f GB/s algorithmic = %.2f %% of 8 TB/s; HBM-side traffic %s B/launch = %.1fx algorithmic (%s)" %
               (rf["kernel_ms"] * 1e3, rf["achieved"], 100 * rf["frac"], rf["traffic"], (rf["traffic"] or 0) / rf["algorithmic_bytes"],
                "measured in the bench run by rocprofv3 --pmc child passes" if "measured in this run" in rf["traffic_source"] else "committed PMC file"))
    if "valu_frac" in rf:
        out.append("                   its SQ counters: VALU issue busy %.0f %%, LDS array busy %.0f %%, %.2f of 3 waves resident per SIMD, %.0f %% of the wave-cycles parked" %
                   (100 * rf["valu_frac"], 100 * rf["lds_frac"], rf["mean_waves_per_simd"], 100 * rf["waves_waiting_frac"]))
    out.append("                   K-Net layer (wino_dw4 clamped-FMA form) %.3f ms = %.1f TFLOP/s issued = %.1f %% of the fp32 matrix peak (%.0f TFLOP/s direct-conv equivalent)" %
               (rm["kernel_ms"], rm["achieved"], 100 * rm["frac"], rm["direct_conv_equivalent_tflops"]))
    out.append("                   parity vs oracle: L1 refined/DPV/BV_cur/BV_predict %.1e / %.1e / %.1e / %.1e; max DPV %.1e, BV_predict %.1e; arg-max flips %d/%d/%d (beyond a tie: %d); pass %s, pass_strict %s" %
               (p["refined"]["mean"], p["dpv"]["mean"], p["bv_cur"]["mean"], p["bv_predict"]["mean"], p["dpv"]["max"], p["bv_predict"]["max"],
                p["refined"]["argmax_mismatch"], p["dpv"]["argmax_mismatch"], p["bv_cur"]["argmax_mismatch"],
                sum(p[k]["argmax_mismatch_beyond_tie_1e-3"] for k in ("refined", "dpv", "bv_cur")), p["pass"], p["pass_strict"]))
    f64 = p.get("fp64", {}).get("dpv_f2")
    if f64:
        out.append("                   float64 yardstick (fixed two-frame sequence): DPV |GPU - fp64| mean %.2e max %.2e   |fp32 CPU oracle - fp64| mean %.2e" %
                   (f64["gpu_minus_fp64_mean"], f64["gpu_minus_fp64_max"], f64.get("oracle_minus_fp64_mean", float("nan"))))
d = L(R + "_bench_H_300frames.json")
out.append(R + "_bench_H_300frames.json  BASELINE config 5 (480x640, D=128, 300 consecutive frames of one stream): %.2f frames/s sustained, %.2f ms/frame, %.1f GB peak HBM" %
           (d["value"], d["ms_per_step"], d["config"].get("peak_hbm_gb", 0)))
d = L(R + "_bench_train.json"); cb = d["cpu_baseline"]
out.append(R + "_bench_train.json        training step at the config-T grid, %d windows per step (%s): %.2f ms per step = %.2f ms per window (%.1f windows/s); CPU (oracle/train_oracle.py, %d threads) %.3f windows/s = %.0fx" %
           (d["config"].get("accum_steps", 1), d["config"].get("launch"), d["ms_per_step"], d.get("ms_per_window", d["ms_per_step"]), d["value"], cb["cores"], cb["value"], d["value"] / cb["value"]))
if os.path.exists(os.path.join(P, R + "_bench_train_accum1.json")):
    d = L(R + "_bench_train_accum1.json")
    out.append(R + "_bench_train_accum1.json the same with one window per step (the shape of rounds 2-3's number): %.2f ms" % d["ms_per_step"])
out.append("")
tail = ""
path = os.path.join(P, R + "_summary.txt")
notes = os.path.join(P, R + "_files.txt")          # hand-written list of the other files of the round, appended verbatim
if os.path.exists(notes):
    tail = open(notes).read()
open(path, "w").write("\n".join(out) + "\n" + tail)
print("\n".join(out[3:]))
