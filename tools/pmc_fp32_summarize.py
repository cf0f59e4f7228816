"""fp32-mode (1e-3) step: HBM bytes per launch of every kernel of the LAST step, from the FETCH_SIZE / WRITE_SIZE passes
written by tools/pmc_fp32_mode.sh (same corrections as pmc_summarize.py: separate passes, FETCH_SIZE doubled on gfx950, KB).
Writes profiles/<ROUND>_pmc_fp32_mode_traffic.json: per kernel the per-slot bytes, and the step total."""
import collections, csv, glob, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ROUND = os.environ.get("ROUND", "r03")


def slots(pattern):
    fs = glob.glob(os.path.join(ROOT, "gpurun_out", pattern))
    if not fs:
        sys.exit("no pass under gpurun_out/" + pattern)
    rows = sorted(csv.DictReader(open(max(fs, key=os.path.getmtime))), key=lambda r: int(r["Start_Timestamp"]))
    adam = [i for i, r in enumerate(rows) if "adam_kernel" in r["Kernel_Name"]]
    out = collections.defaultdict(list)
    for r in rows[adam[-2] + 1:adam[-1] + 1]:
        out[r["Kernel_Name"].split("(")[0].replace("void ", "")].append(float(r["Counter_Value"]))
    return out


f, w = slots("pmc32_fetch/*/*counter_collection.csv"), slots("pmc32_write/*/*counter_collection.csv")
res, total = {}, 0
for k in f:
    if k in w and len(f[k]) == len(w[k]):
        b = [int((2 * a + c) * 1024) for a, c in zip(f[k], w[k])]
        total += sum(b)
        res[k] = {"launches": len(b), "bytes_per_step": sum(b), "largest_launches_MB": [round(x / 1e6, 1) for x in sorted(b, reverse=True)[:6]]}
res = dict(sorted(res.items(), key=lambda kv: -kv[1]["bytes_per_step"]))
out = {"_meta": {"command": "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace -- python3 bench.py --precision fp32 --steps 1 --warmup 1",
                 "note": "bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 per launch of the last step; FETCH_SIZE doubled on gfx950",
                 "step_total_GB": round(total / 1e9, 2)}, "kernels": res}
json.dump(out, open(os.path.join(ROOT, "profiles", f"{ROUND}_pmc_fp32_mode_traffic.json"), "w"), indent=1)
print("step total %.1f GB" % (total / 1e9))
for k, v in list(res.items())[:8]:
    print(f"{k[:60]:60s} {v['launches']:3d}  {v['bytes_per_step']/1e9:6.2f} GB  largest {v['largest_launches_MB'][:3]}")
