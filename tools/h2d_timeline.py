"""Timeline of one host-buffer batch registration from a rocprofv3 --kernel-trace --memory-copy-trace run:
python tools/h2d_timeline.py <dir with *_kernel_trace.csv and *_memory_copy_trace.csv>"""
import csv, glob, sys
d = sys.argv[1]
ev = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "K " + r["Kernel_Name"][:40]))
for f in glob.glob(d + "/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "C %s %s B" % (r.get("Direction", ""), r.get("Bytes", r.get("Size", "")))))
ev.sort()
# last occurrence of the big-copy pattern: print the final 40 events relative to the first of them
tail = ev[-int(sys.argv[2]) if len(sys.argv) > 2 else -45:]
t0 = tail[0][0]
for s, e, n in tail:
    print("%9.3f %9.3f  %8.3f  %s" % ((s - t0) / 1e6, (e - t0) / 1e6, (e - s) / 1e6, n))
