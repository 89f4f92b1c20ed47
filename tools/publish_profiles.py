"""Copy the judged parts of a tools/profile_round.sh run from gpurun_out/ into profiles/ (tracked).
usage: python tools/publish_profiles.py <tag in gpurun_out> <name in profiles, e.g. r02>"""
import csv, glob, json, os, shutil, sys
tag, name = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, dst = os.path.join(root, "gpurun_out"), os.path.join(root, "profiles")
for wl, kern in (("c2", "k_selfplay_queue"), ("dc", "k_dc_selfplay_fused")):
    d = os.path.join(src, f"{tag}_prof_{wl}")
    stats = glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True)[0]
    trace = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)[0]
    label = "bench_default" if wl == "c2" else "bench_dc"
    shutil.copy(stats, os.path.join(dst, f"{name}_{label}_kernel_stats.csv"))
    rows = [r for r in csv.DictReader(open(trace)) if kern in r["Kernel_Name"]]
    with open(os.path.join(dst, f"{name}_{label}_dispatches.csv"), "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Kernel_Name", "Dispatch_Id", "Start_Timestamp", "End_Timestamp", "Duration_ms"])
        for r in rows:
            w.writerow([r["Kernel_Name"], r["Dispatch_Id"], r["Start_Timestamp"], r["End_Timestamp"],
                        "%.3f" % ((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)])
    line = [l for l in open(os.path.join(src, f"{tag}_prof_{wl}.log")) if l.startswith('{"metric"')][-1]
    with open(os.path.join(dst, f"{name}_{label}_line.json"), "w") as f:
        f.write(line)
    j = json.loads(line)
    durs = [float((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6) for r in rows]
    print(wl, "value %.1f %s, ms/step %.2f, launch_ms_mean (HIP events) %.2f, trace dispatches of the timed region %s, frac %.3f"
          % (j["value"], j["unit"], j["ms_per_step"], j["roofline"]["launch_ms_mean"], [round(x, 2) for x in durs[-2:]], j["roofline"]["frac"]))
shutil.copy(os.path.join(src, f"{tag}_queue_pmc_summary.json"), os.path.join(dst, f"{name}_queue_pmc_summary.json"))
shutil.copy(os.path.join(src, f"{tag}_dc_pmc_summary.json"), os.path.join(dst, f"{name}_dc_pmc_summary.json"))
for f in (f"{name}_queue_pmc_summary.json", f"{name}_dc_pmc_summary.json"):
    p = json.load(open(os.path.join(dst, f)))
    print(f, {k: (round(v, 3) if isinstance(v, float) else v) for k, v in p.items() if not k.startswith("SQ_") and not k.startswith("dur_") and not k.startswith("GRBM") and k not in ("FETCH_SIZE", "WRITE_SIZE", "command")})
