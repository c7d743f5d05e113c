import re, sys, statistics as st
L = [l for l in open(sys.argv[1]) if l.startswith("[sa_tracker]")]
names = {"up to the launches": ["before the jobs", "assemble", "longest assemble job", "epochs", "stage + evict", "enqueue"],
         "behind the launches": ["deferred", "wait for the association", "merges", "longest merge job", "all merge jobs", "wait for the Kalman dispatch",
                                 "tables + results", "longest job", "all jobs", "minor faults"]}
for key, nm in names.items():
    rows = [[float(x) for x in re.findall(r"[-+]?\d+\.\d+", l)] for l in L if key in l][10:75]
    rows = [r for r in rows if len(r) == len(nm)]
    if rows: print("   ", key + ":", dict(zip(nm, [round(st.median(c), 1) for c in zip(*rows)])))
