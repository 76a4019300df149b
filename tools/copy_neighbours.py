"""Which kernels run right before / after each __amd_rocclr_copyBuffer of a rocprofv3 --kernel-trace database (who issues the copies?)."""
import collections, sqlite3, sys
db = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
rows = db.execute("select name, start, end from kernels order by start").fetchall()
c = collections.Counter()
for i, (n, s, e) in enumerate(rows):
    if 'copyBuffer' in n:
        prev = rows[i - 1][0][:60] if i else ''
        nxt = rows[i + 1][0][:60] if i + 1 < len(rows) else ''
        c[(prev, nxt)] += 1
for (p, n), k in c.most_common(40):
    print(f'{k:5d}  after {p:60s}  before {n}')
