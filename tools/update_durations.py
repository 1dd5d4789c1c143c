"""Durations of the trailing updates of the LAST dense factorisation in rocprofv3 kernel traces (rocpd sqlite), side by side.
usage: update_durations.py a.db b.db ..."""
import sqlite3, sys
cols = []
for f in sys.argv[1:]:
    db = sqlite3.connect(f)
    rows = list(db.execute("select name, start, end from kernels where name like '%k_ldl_%' order by start"))
    upd = [(s, e) for n, s, e in rows if 'k_ldl_update' in n]
    # last factorisation: the last run of strictly shrinking... simply the last 62 (Venice) launches
    n = int(len(upd) / max(1, round(len(upd) / 62)))
    cols.append([(e - s) / 1e3 for s, e in upd[-n:]])
print("launch " + " ".join("%10s" % f.split('/')[-2][-10:] for f in sys.argv[1:]))
for i in range(max(len(c) for c in cols)):
    print("%6d " % i + " ".join("%10.1f" % c[i] if i < len(c) else " " * 10 for c in cols))
print("sum    " + " ".join("%10.1f" % sum(c) for c in cols))
