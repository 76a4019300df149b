"""rocprofv3 (ROCm 7.2) writes its --kernel-trace --stats result as a rocpd SQLite database; this prints the per-kernel summary
(the same columns as the CSV that older rocprofv3 builds wrote) so that it can be committed under profiles/ as text."""
import csv, sqlite3, sys
db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) from kernels group by name "
                  "order by sum(duration) desc").fetchall()
tot = sum(r[2] for r in rows)
w = csv.writer(open(sys.argv[2], 'w') if len(sys.argv) > 2 else sys.stdout, quoting=csv.QUOTE_NONNUMERIC)
w.writerow(['Name', 'Calls', 'TotalDurationNs', 'AverageNs', 'Percentage', 'MinNs', 'MaxNs'])
for n, c, t, a, lo, hi in rows:
    w.writerow([n, c, t, round(a, 1), round(100.0 * t / tot, 4), lo, hi])
