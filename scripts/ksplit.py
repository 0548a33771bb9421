"""python scripts/ksplit.py <results.db> <substring>: launches of one kernel in start order, averaged per position in the
step (the discrete and the cast pass launch the same kernels in turn): which pass a kernel's time belongs to."""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
rows = c.execute("select end - start from kernels where name like ? order by start", ("%" + sys.argv[2] + "%",)).fetchall()
d = [r[0] for r in rows][4:]   # (warm-up launches first)
for ph in range(2):
    v = d[ph::2]
    print("%s launch %d of 2: n=%d avg=%.1f us" % (sys.argv[2], ph, len(v), sum(v) / max(len(v), 1) / 1e3))
