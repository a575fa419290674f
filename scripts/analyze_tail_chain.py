"""Reads a rocprofv3 --kernel-trace CSV of scripts/prof_dense.py --phases prefill and prints, for the LAST prefill pass, how the tail chain (second stream)
sits against the main chain: per queue the span and busy time, and how long after the main chain's last kernel the tail chain ends."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last pass: from the last add_rows2_kernel launch on
starts = [i for i, r in enumerate(rows) if "add_rows2_kernel" in r["Kernel_Name"]]
i0 = starts[-1]
seg = rows[i0:]
# ends at the lm_head gemv (first gemv_kernel after)
for j, r in enumerate(seg):
    if "gemv_kernel" in r["Kernel_Name"]:
        seg = seg[:j + 1]; break
t0 = int(seg[0]["Start_Timestamp"])
q = collections.defaultdict(list)
for r in seg:
    q[r["Queue_Id"]].append(r)
print(f"pass: {len(seg)} dispatches, {(int(seg[-1]['End_Timestamp']) - t0) / 1e3:.1f} us")
for k, v in q.items():
    busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in v)
    print(f" queue {k}: {len(v)} dispatches, first start {(int(v[0]['Start_Timestamp']) - t0) / 1e3:.1f} us, last end {(int(v[-1]['End_Timestamp']) - t0) / 1e3:.1f} us, busy {busy / 1e3:.1f} us")
    names = collections.Counter()
    dur = collections.Counter()
    for r in v:
        n = r["Kernel_Name"].split("(")[0][-60:]
        names[n] += 1; dur[n] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    for n, c in names.most_common(12):
        print(f"    {c:4d} x {dur[n] / c / 1e3:7.2f} us  {n}")
