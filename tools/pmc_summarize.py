"""Summarise the rocprofv3 passes of tools/gpu_profile.sh into profiles/:
  r01_kernel_stats_bench_steps5.csv   (copy of the newest --kernel-trace --stats summary)
  r01_pmc_hbm_traffic.json            HBM bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) KB * 1024, per (kernel, grid):
                                      separate --pmc FETCH_SIZE / --pmc WRITE_SIZE passes, FETCH_SIZE doubled on gfx950 as
                                      MI355X_MICROARCH.md prescribes.
The persistent conv kernel launches every large layer with the same grid (2 workgroups per CU), so its launches are split
by duration: the 32->32 @128^3 layers (the dominant kernel bench.py reports) are the ones within 40 % of the longest."""
import csv, glob, json, os, shutil, sys, collections

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "gpurun_out")


def newest(pattern):
    fs = glob.glob(os.path.join(G, pattern))
    return max(fs, key=os.path.getmtime) if fs else None


def per_kernel(path):
    agg = collections.defaultdict(list)
    rows = list(csv.DictReader(open(path)))
    # the persistent conv kernel launches every large layer with the same grid; the 32->32 @128^3 launches are the ones that
    # move the most bytes (537 MB in, 537 MB out; the next largest layer moves a quarter of that)
    conv = [r for r in rows if "conv_mfma_kernel<1>" in r["Kernel_Name"] and r["Grid_Size"] == "131072"]
    vmax = max((float(r["Counter_Value"]) for r in conv), default=0.0)
    big = {id(r) for r in conv if float(r["Counter_Value"]) >= 0.5 * vmax}
    for r in rows:
        name = r["Kernel_Name"].split("(")[0]
        key = f"{name} grid={r['Grid_Size']}"
        if id(r) in big:
            key = "mednet::conv_mfma_kernel<1> 32->32@128^3 (persistent grid=131072, longest launches)"
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
        shutil.copy(st, os.path.join(ROOT, "profiles", "r01_kernel_stats_bench_steps5.csv"))
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
    json.dump(out, open(os.path.join(ROOT, "profiles", "r01_pmc_hbm_traffic.json"), "w"), indent=1)
    fs, wsl = per_slot(f), per_slot(w)
    slots = {k: [int((2 * a + b) * 1024) for a, b in zip(fs[k], wsl[k])] for k in fs if k in wsl and len(fs[k]) == len(wsl[k])}
    json.dump(slots, open(os.path.join(ROOT, "profiles", "r01_pmc_hbm_slots.json"), "w"))
    k = "mednet::conv_mfma_kernel<1> 32->32@128^3 (persistent grid=131072, longest launches)"
    print(k, out.get(k))


if __name__ == "__main__":
    main()
