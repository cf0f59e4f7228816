"""Summarise the rocprofv3 passes of tools/gpu_profile.sh into profiles/ (ROUND env, default r02):
  rNN_kernel_stats_bench_steps5.csv   (copy of the newest --kernel-trace --stats summary)
  rNN_pmc_sq_dominant_kernel.json     SQ counters of the dominant kernel's launches (MFMA busy, LDS conflicts / waits)
  rNN_pmc_hbm_traffic.json            HBM bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) KB * 1024, per (kernel, grid);
                                      "_meta" records the sha256 of csrc/conv_mfma.hip and the git head the passes were
                                      taken from (bench.py reports `traffic` only while that sha matches the source):
                                      separate --pmc FETCH_SIZE / --pmc WRITE_SIZE passes, FETCH_SIZE doubled on gfx950 as
                                      MI355X_MICROARCH.md prescribes.
The dominant kernel bench.py reports is the 32 -> 32 channel specialisation in its forward variant (conv32_mfma_kernel<4>:
fused GroupNorm statistics); in the bf16 step of bench.py all its launches are the 32->32 @128^3 layers."""
import csv, glob, hashlib, json, os, shutil, subprocess, sys, collections

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "gpurun_out")
ROUND = os.environ.get("ROUND", "r02")
DOM = "mednet::conv32_mfma_kernel<4> conv3d 32->32@128^3 (forward launches with fused GroupNorm statistics, grid=65536)"


def newest(pattern):
    fs = glob.glob(os.path.join(G, pattern))
    return max(fs, key=os.path.getmtime) if fs else None


def is_dominant(r):
    return "conv32_mfma_kernel<4>" in r["Kernel_Name"] and "mednet_f16" not in r["Kernel_Name"]


def per_kernel(path):
    agg = collections.defaultdict(list)
    rows = list(csv.DictReader(open(path)))
    big = {id(r) for r in rows if is_dominant(r)}
    for r in rows:
        name = r["Kernel_Name"].split("(")[0]
        key = f"{name} grid={r['Grid_Size']}"
        if id(r) in big:
            key = DOM
        agg[key].append(float(r["Counter_Value"]))
    return agg


def per_slot(path):
    """{kernel: [counter value of its i-th launch in the LAST full step]} -- the launch sequence of a step is deterministic,
    so the i-th launch of a kernel is the same layer in every run (trace, FETCH_SIZE pass, WRITE_SIZE pass)."""
    rows = sorted(csv.DictReader(open(path)), key=lambda r: int(r["Start_Timestamp"]))
    adam = [i for i, r in enumerate(rows) if "adam_kernel" in r["Kernel_Name"]]
    step = rows[adam[-2] + 1:adam[-1] + 1]
    slots = collections.defaultdict(list)
    for r in step:
        slots[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
    return slots


def main():
    st = newest("prof_stats/*/*kernel_stats.csv")
    if st:
        shutil.copy(st, os.path.join(ROOT, "profiles", f"{ROUND}_kernel_stats_bench_steps5.csv"))
    f, w = newest("pmc_fetch/*/*counter_collection.csv"), newest("pmc_write/*/*counter_collection.csv")
    if not (f and w):
        sys.exit("no PMC passes under gpurun_out/")
    fa, wa = per_kernel(f), per_kernel(w)
    out = {}
    for k in sorted(set(fa) & set(wa)):
        if "mednet" not in k:
            continue
        fk, wk = sum(fa[k]) / len(fa[k]), sum(wa[k]) / len(wa[k])
        out[k] = {"FETCH_SIZE_KB_avg": round(fk, 1), "WRITE_SIZE_KB_avg": round(wk, 1),
                  "hbm_bytes_per_launch": int((2 * fk + wk) * 1024), "launches_fetch": len(fa[k]), "launches_write": len(wa[k])}
    try:
        head = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
    except Exception:
        head = "?"
    src = os.path.join(ROOT, "torch-mednet_amd", "csrc", "conv_mfma.hip")
    out["_meta"] = {"kernel_source": "torch-mednet_amd/csrc/conv_mfma.hip",
                    "kernel_source_sha256": hashlib.sha256(open(src, "rb").read()).hexdigest(), "git_head": head,
                    "command": "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace -- python3 bench.py --steps 1 --warmup 1 (separate passes)",
                    "note": "hbm_bytes_per_launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: FETCH_SIZE counts half of a wide coalesced read on gfx950"}
    json.dump(out, open(os.path.join(ROOT, "profiles", f"{ROUND}_pmc_hbm_traffic.json"), "w"), indent=1)
    sq = newest("pmc_sq/*/*counter_collection.csv")
    if sq:  # SQ counters of the dominant kernel's launches
        rows = list(csv.DictReader(open(sq)))
        conv = [r for r in rows if is_dominant(r)]
        byd = collections.defaultdict(dict)
        for r in conv:
            byd[r["Dispatch_Id"]][r["Counter_Name"]] = float(r["Counter_Value"])
            byd[r["Dispatch_Id"]]["_dur"] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        dmax = max((d["_dur"] for d in byd.values()), default=0)
        big = [d for d in byd.values() if d["_dur"] >= 0.6 * dmax]
        if big:
            avg = {k: sum(d[k] for d in big) / len(big) for k in big[0]}
            mf = 2 * 27 * 32 * 32 * 4 * 128 ** 3 / (2 * 32 * 32 * 16)  # MFMA instructions of one launch
            res = {"kernel": DOM, "launches": len(big), "avg_duration_us_under_pmc": round(avg.pop("_dur") / 1e3, 1),
                   "counters_avg_per_launch": {k: round(v, 1) for k, v in sorted(avg.items())},
                   "mfma_instructions_per_launch": mf,
                   "SQ_VALU_MFMA_BUSY_CYCLES_over_32x_mfma_count": round(avg.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (32 * mf), 3),
                   "_meta": out["_meta"]}
            if avg.get("SQ_BUSY_CYCLES"):
                res["mfma_busy_fraction_of_SQ_BUSY_CYCLES_x_simds"] = "see DESIGN.md: busy cycles / (kernel cycles x 1024 SIMDs)"
            json.dump(res, open(os.path.join(ROOT, "profiles", f"{ROUND}_pmc_sq_dominant_kernel.json"), "w"), indent=1)
            print("SQ:", res["counters_avg_per_launch"])
    if sq:  # the same for the weight gradient of the same layers (round 5: wgrad_mfma4_kernel; the longest launches = 32 -> 32 @128^3)
        rows = list(csv.DictReader(open(sq)))
        wg = [r for r in rows if "wgrad_mfma4_kernel" in r["Kernel_Name"] and "mednet_f16" not in r["Kernel_Name"]]
        byd = collections.defaultdict(dict)
        for r in wg:
            byd[r["Dispatch_Id"]][r["Counter_Name"]] = float(r["Counter_Value"])
            byd[r["Dispatch_Id"]]["_dur"] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        dmax = max((d["_dur"] for d in byd.values()), default=0)
        big = [d for d in byd.values() if d["_dur"] >= 0.6 * dmax]
        if big:
            avg = {k: sum(d[k] for d in big) / len(big) for k in big[0]}
            mf = 2 * 27 * 32 * 32 * 4 * 128 ** 3 / (2 * 32 * 32 * 16) * 28 / 27  # (a wave with 6 taps issues a 7th, discarded)
            res = {"kernel": "mednet::wgrad_mfma4_kernel weight gradient of conv3d 32->32@128^3 (the step's longest launches; beside the "
                             "main stream's GroupNorm-backward passes unless MEDNET_SIDE_STREAM=0)",
                   "launches": len(big), "avg_duration_us_under_pmc": round(avg.pop("_dur") / 1e3, 1),
                   "counters_avg_per_launch": {k: round(v, 1) for k, v in sorted(avg.items())},
                   "mfma_instructions_per_launch": mf, "_meta": out["_meta"]}
            json.dump(res, open(os.path.join(ROOT, "profiles", f"{ROUND}_pmc_sq_wgrad_kernel.json"), "w"), indent=1)
            print("SQ wgrad:", res["counters_avg_per_launch"])
    fs, wsl = per_slot(f), per_slot(w)
    slots = {k: [int((2 * a + b) * 1024) for a, b in zip(fs[k], wsl[k])] for k in fs if k in wsl and len(fs[k]) == len(wsl[k])}
    json.dump(slots, open(os.path.join(ROOT, "profiles", f"{ROUND}_pmc_hbm_slots.json"), "w"))
    print(DOM, out.get(DOM))


if __name__ == "__main__":
    main()
