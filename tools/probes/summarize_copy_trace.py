"""rocprofv3 --kernel-trace --memory-copy-trace CSVs of tools/probes/chunked_frame_trace.py -> the last frame as a timeline (us from its
first upload): python tools/probes/summarize_copy_trace.py <dir with *_memory_copy_trace.csv and *_kernel_trace.csv>"""
import csv, glob, os, sys
d = sys.argv[1]
mc = list(csv.DictReader(open(glob.glob(os.path.join(d, "**", "*_memory_copy_trace.csv"), recursive=True)[0])))
kt = list(csv.DictReader(open(glob.glob(os.path.join(d, "**", "*_kernel_trace.csv"), recursive=True)[0])))
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "copy " + r["Direction"][12:].replace("_TO_", " -> ").lower()) for r in mc]
ev += [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "").replace("mi::", "")[:44]) for r in kt]
ev.sort()
frames = [i for i, e in enumerate(ev) if e[2].startswith("k_frame")]
last = frames[-1]
start = last
while start > 0 and ev[last][0] - ev[start - 1][0] < 2_500_000 and not ev[start - 1][2].startswith("k_pack"):
    start -= 1
t0 = ev[start][0]
end = last
while end + 1 < len(ev) and ev[end + 1][0] - t0 < 3_000_000:
    end += 1
print("| start us | end us | us | what |\n|---|---|---|---|")
busy = {"host -> device": 0.0, "device -> host": 0.0}
for s, e, name in ev[start:end + 1]:
    print(f"| {(s - t0) / 1e3:.1f} | {(e - t0) / 1e3:.1f} | {(e - s) / 1e3:.1f} | {name} |")
    for k in busy:
        if k in name:
            busy[k] += (e - s) / 1e3
print(f"\nlast event ends at {(ev[end][1] - t0) / 1e3:.1f} us; copy engines busy: " + ", ".join(f"{k} {v:.0f} us" for k, v in busy.items()))
