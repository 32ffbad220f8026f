import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
rows = c.execute("select name, start, end from kernels order by start").fetchall()
t0 = rows[0][1]
# find gaps > 300us in last 40% of trace
lo = t0 + 0.6 * (rows[-1][2] - t0)
busy_end = rows[0][2]
for i in range(1, len(rows)):
    n, s, e = rows[i]
    if s > busy_end and s >= lo and (s - busy_end) > 300e3:
        print("---- gap %.1f us at t=%.2f ms" % ((s - busy_end) / 1e3, (s - t0) / 1e6))
        for j in range(max(0, i - 8), min(len(rows), i + 8)):
            nn, ss, ee = rows[j]
            print("   %s %10.3f ms  dur %8.1f us  %s" % ("*" if j == i else " ", (ss - t0) / 1e6, (ee - ss) / 1e3, nn[:90]))
    busy_end = max(busy_end, e)
