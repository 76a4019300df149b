"""HBM traffic per kernel from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; rocpd SQLite results) -> JSON for profiles/.
    python3 tools/pmc_summary.py fetch_results.db write_results.db out.json
Units and corrections as MI355X_MICROARCH.md prescribes: both counters are KiB, FETCH_SIZE is doubled on gfx950."""
import json, sqlite3, sys, collections
GEMM = ('gemm_nt_', 'gemm_tn_', 'gemm_kernel')


def per_kernel(path, counter):
    db = sqlite3.connect(path)
    acc = collections.defaultdict(lambda: [0, 0.0])
    for name, val in db.execute("select kernel_name, value from counters_collection where counter_name = ?", (counter,)):
        acc[name][0] += 1
        acc[name][1] += val
    return acc


f, w = per_kernel(sys.argv[1], 'FETCH_SIZE'), per_kernel(sys.argv[2], 'WRITE_SIZE')
out = {'source': 'rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over `python3 bench.py --steps 1 --warmup 1 '
                 '--no-cpu-baseline --no-roofline`, MI355X',
       'corrections': 'FETCH_SIZE (KiB) x 2 on gfx950 (MI355X_MICROARCH.md, HBM section); WRITE_SIZE (KiB) as reported',
       'per_kernel': {}}
fam = {'kernels': [], 'launches': 0, 'fetch': 0.0, 'write': 0.0}
for name in sorted(f, key=lambda k: -(f[k][1] * 2 + w.get(k, [0, 0.0])[1])):
    n, fk = f[name]
    wk = w.get(name, [n, 0.0])[1]
    if any(g in name for g in GEMM):
        fam['kernels'].append(name)
        fam['launches'] += n
        fam['fetch'] += fk * 2 * 1024
        fam['write'] += wk * 1024
    if len(out['per_kernel']) < 24:
        out['per_kernel'][name] = {'launches': n, 'fetch_bytes_per_launch': fk * 2 * 1024 / n, 'write_bytes_per_launch': wk * 1024 / n}
out['gemm_family'] = {'kernels': sorted(fam['kernels']), 'launches': fam['launches'],
                      'hbm_bytes_per_launch': (fam['fetch'] + fam['write']) / max(fam['launches'], 1),
                      'fetch_bytes_per_launch': fam['fetch'] / max(fam['launches'], 1),
                      'write_bytes_per_launch': fam['write'] / max(fam['launches'], 1)}
json.dump(out, open(sys.argv[3], 'w'), indent=1)
print(json.dumps(out['gemm_family'], indent=1))
