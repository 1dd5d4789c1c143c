"""Timeline of the last dense factorisation in a rocprofv3 kernel trace (rocpd sqlite): per-kernel start / duration.
usage: trace_timeline.py results.db [max_rows]"""
import re, sqlite3, sys
from collections import defaultdict
db = sqlite3.connect(sys.argv[1])
rows = list(db.execute("select name, start, end, queue_id from kernels order by start"))
def short(n):
    m = re.search(r'k_ldl_\w+?(?=I[df])|k_\w+', n)
    return (m.group(0) if m else n)[:28]
names = [short(r[0]) for r in rows]
starts = [i for i, n in enumerate(names) if 'ldl' in n and (i == 0 or 'ldl' not in names[i - 1])]
i = starts[-1]
j = i
while j + 1 < len(names) and 'ldl' in names[j + 1]:
    j += 1
seq = rows[i:j + 1]
t0 = seq[0][1]
print("kernels:", len(seq), "span ms", (max(r[2] for r in seq) - t0) / 1e6)
tot = defaultdict(lambda: [0, 0.0])
for r in seq:
    tot[short(r[0])][0] += 1
    tot[short(r[0])][1] += (r[2] - r[1]) / 1e3
for k, v in tot.items():
    print("  %-24s n %4d  sum %9.1f us  avg %7.1f" % (k, v[0], v[1], v[1] / v[0]))
# critical-path view: main queue only (the queue of the first kernel)
q0 = seq[0][3]
main = [r for r in seq if r[3] == q0]
busy = sum(r[2] - r[1] for r in main) / 1e3
print("main queue: busy %.1f us, gaps %.1f us" % (busy, (main[-1][2] - main[0][1]) / 1e3 - busy))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
for r in seq[:n]:
    print("%-22s start %8.1f end %8.1f dur %7.1f us q%s" % (short(r[0]), (r[1] - t0) / 1e3, (r[2] - t0) / 1e3, (r[2] - r[1]) / 1e3, r[3]))
