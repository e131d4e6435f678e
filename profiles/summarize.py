"""Turns the CSVs of profiles/collect.sh into the small summaries committed under profiles/:
<tag>_kernel_stats.csv (rocprofv3 --stats rows of our kernels), <tag>_pmc.json (per-launch counter
means) and an entry in hbm_traffic.json (HBM bytes per launch of the dominant kernel, FETCH_SIZE
doubled as MI355X_MICROARCH.md prescribes for gfx950 streaming reads - calibrated on this kernel,
see DESIGN.md section 6 - plus WRITE_SIZE).

    python profiles/summarize.py gpurun_out/prof_<tag> <tag> <variant> <nsrc> <rows_per_call>
"""
import collections
import csv
import glob
import json
import os
import sys

src, tag, variant, nsrc, rows = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4]), int(sys.argv[5])
here = os.path.dirname(os.path.abspath(__file__))

stats = os.path.join(src, "trace", "t_kernel_stats.csv")
with open(stats) as f, open(os.path.join(here, "%s_kernel_stats.csv" % tag), "w") as g:
    for i, line in enumerate(f):
        if i == 0 or "lf::" in line:
            g.write(line)

pmc = collections.defaultdict(lambda: collections.defaultdict(list))
for path in glob.glob(os.path.join(src, "pmc_*", "p_counter_collection.csv")):
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"]
        if "lf::" in k:
            name = k.split("(")[0].replace("void ", "")
            pmc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
summary = {k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in pmc.items()}
for k, cs in summary.items():
    cs["_launches_sampled"] = len(next(iter(pmc[k].values())))
json.dump(summary, open(os.path.join(here, "%s_pmc.json" % tag), "w"), indent=1, sort_keys=True)

# the dominant kernel = the direct lf_main / lf_free (the bench also runs a census instantiation and, with extras, the
# compressed-catalogue one, whose launches read a few hundred KB): the one with the largest fetch
def product(k):                # lf_main, or lf_free<ST, false, .>: not the census instantiation lf_free<ST, true, false>
    import re
    return "lf_main" in k or re.search(r"lf_free<\d+, false", k) is not None or re.search(r"lf_pers<\d+, true", k) is not None


dom = sorted([k for k in summary if product(k) and "FETCH_SIZE" in summary[k]], key=lambda k: -summary[k]["_launches_sampled"])
if dom:
    fetch_kb, write_kb = summary[dom[0]]["FETCH_SIZE"], summary[dom[0]].get("WRITE_SIZE", 0.0)
    tf = os.path.join(here, "hbm_traffic.json")
    d = json.load(open(tf)) if os.path.exists(tf) else {}
    # rocprofv3 --stats average of the same kernel (bench.py prints it next to its own event timing)
    avg_ns = None
    for r in csv.DictReader(open(os.path.join(here, "%s_kernel_stats.csv" % tag))):
        if r["Name"].split("(")[0].replace("void ", "") == dom[0]:
            avg_ns = float(r["AverageNs"])
    suffix = sys.argv[6] if len(sys.argv) > 6 else ""          # (A/B runs of the same shape: their own entry)
    d["%s_n%d_b%d%s" % (variant, nsrc, rows, suffix)] = {
        "kernel": dom[0], "FETCH_SIZE_KB": fetch_kb, "WRITE_SIZE_KB": write_kb,
        "fetch_correction": 2.0,
        "hbm_bytes_per_launch": 2.0 * fetch_kb * 1024 + write_kb * 1024, "source": "profiles/%s_pmc.json" % tag,
        "rocprofv3_avg_launch_ns": avg_ns, "stats": "profiles/%s_kernel_stats.csv" % tag}
    json.dump(d, open(tf, "w"), indent=1, sort_keys=True)
print(json.dumps(summary, indent=1)[:1500])
