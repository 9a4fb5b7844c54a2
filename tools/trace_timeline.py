"""Prints a slice of a rocprofv3 --kernel-trace CSV as a timeline: start / end (us, relative) and queue of each kernel.
usage: trace_timeline.py <dir with *_kernel_trace.csv> [first_row] [rows]"""
import csv, glob, sys
d = sys.argv[1]; first = int(sys.argv[2]) if len(sys.argv) > 2 else 2000; rows = int(sys.argv[3]) if len(sys.argv) > 3 else 40
f = sorted(glob.glob(d + "/**/*kernel_trace.csv", recursive=True))[0]
r = list(csv.DictReader(open(f)))
r.sort(key=lambda x: int(x["Start_Timestamp"]))
sl = r[first:first + rows]
t0 = int(sl[0]["Start_Timestamp"])
for x in sl:
    s, e = int(x["Start_Timestamp"]) - t0, int(x["End_Timestamp"]) - t0
    print(f"{s / 1e3:9.1f} -> {e / 1e3:9.1f} us  ({(e - s) / 1e3:7.1f})  queue {x.get('Queue_Id', '?'):>3}  {x['Kernel_Name'][:90]}")
