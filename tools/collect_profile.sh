#!/bin/bash
# Copy what tools/profile_bench.sh <tag> left under gpurun_out/prof_<tag>/ into profiles/<tag>/ (the tracked, judged
# summaries): SUMMARY.md, both bench lines, pmc_traffic.json, the kernel-trace stats, the two PMC passes and
# decode_dispatches.csv (every k_decode* / k_slotmap dispatch of the traced process in start order).
set -e
for t in "$@"; do   # <tag>, or <tag>:<directory under profiles/> to keep a recording under another name
    src=gpurun_out/prof_${t%%:*}; dst=profiles/${t##*:}; mkdir -p $dst
    cp $src/SUMMARY.md $src/bench_plain.json $src/bench_under_rocprof.json $src/pmc_traffic.json $dst/
    cp $src/trace/bench_kernel_stats.csv $dst/kernel_stats.csv
    cp $src/pmc_fetch/bench_counter_collection.csv $dst/pmc_fetch_size.csv
    cp $src/pmc_write/bench_counter_collection.csv $dst/pmc_write_size.csv
    python - "$src" "$dst" <<'PY'
import csv, sys
src, dst = sys.argv[1:3]
rows = [r for r in csv.DictReader(open(src + "/trace/bench_kernel_trace.csv"))
        if "k_decode" in r["Kernel_Name"] or "k_slotmap" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
with open(dst + "/decode_dispatches.csv", "w") as f:
    f.write("kernel,start_ns,duration_us,vgpr,sgpr,lds\n")
    for r in rows:
        f.write("%s,%s,%.1f,%s,%s,%s\n" % (r["Kernel_Name"].split("(")[0][-60:].replace(",", ";"), r["Start_Timestamp"],
                                           (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3,
                                           r.get("VGPR_Count", ""), r.get("SGPR_Count", ""), r.get("LDS_Block_Size", "")))
print(dst, len(rows), "decode dispatches")
PY
done
