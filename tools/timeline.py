"""Per-launch timeline (us) of the last render step in a rocprofv3 kernel-trace database: start offset, duration, stream order."""
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
rows = c.execute("select name, start, end from kernels order by start").fetchall()
last = max(i for i, r in enumerate(rows) if r[0].startswith("k_init"))
t0 = rows[last][1]
for n, s, e in rows[last:]:
    short = 'ext' if 'k_trace<0' in n else 'vis' if 'k_trace<1' in n else 'own' if 'k_trace<2' in n else n.split('(')[0][:10]
    print(f"{short:10s} start {(s - t0) / 1000:9.1f}  dur {(e - s) / 1000:8.1f}  end {(e - t0) / 1000:9.1f}")
