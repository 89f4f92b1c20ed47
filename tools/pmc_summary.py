"""Sum rocprofv3 --pmc counter CSVs per kernel.  usage: pmc_summary.py <dir> <kernel-substring> [out.json]
Takes the LONGEST dispatch of the matching kernel in each pass (the timed region) and sums each counter over its
rows (rocprofv3 writes one row per counter per dispatch, already summed over XCDs/SEs dimensions or split by them)."""
import sys, os, csv, json, glob, collections

def main():
    root, pat = sys.argv[1], sys.argv[2]
    out = {}
    for f in sorted(glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True)):
        rows = [r for r in csv.DictReader(open(f)) if pat in r["Kernel_Name"]]
        if not rows:
            continue
        per = collections.defaultdict(lambda: collections.defaultdict(float))
        for r in rows:
            per[r["Dispatch_Id"]][r["Counter_Name"]] += float(r["Counter_Value"])
        # longest dispatch = the one with the largest first counter value
        key = max(per, key=lambda d: max(per[d].values()))
        dur = None
        kt = f.replace("counter_collection", "kernel_trace")
        if os.path.exists(kt):
            for r in csv.DictReader(open(kt)):
                if r.get("Dispatch_Id") == key:
                    dur = (float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e6
        tag = os.path.basename(os.path.dirname(f)) or os.path.basename(f)
        for k, v in per[key].items():
            out[k] = v
        if dur is not None:
            out["dur_ms_" + "_".join(sorted(per[key].keys()))[:40]] = dur
    print(json.dumps(out, indent=1))
    if len(sys.argv) > 3:
        json.dump(out, open(sys.argv[3], "w"), indent=1)

main()
