"""Per-kernel averages of the counters of one or more rocprofv3 --pmc passes (rocpd SQLite results) as a text table / JSON.
    python3 tools/pmc_table.py out.json pass1_results.db [pass2_results.db ...] [--match substr]"""
import collections, json, sqlite3, sys
args = sys.argv[1:]
match = None
if '--match' in args:
    i = args.index('--match'); match = args[i + 1]; del args[i:i + 2]
out_path, dbs = args[0], args[1:]
tab = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for path in dbs:
    db = sqlite3.connect(path)
    for name, cname, val in db.execute("select kernel_name, counter_name, value from counters_collection"):
        if match and match not in name:
            continue
        e = tab[name][cname]
        e[0] += 1; e[1] += val
res = {}
for name, cs in tab.items():
    res[name] = {c: v[1] / v[0] for c, v in cs.items()}
    res[name]['launches'] = max(v[0] for v in cs.values())
json.dump(res, open(out_path, 'w'), indent=1, sort_keys=True)
for name, cs in sorted(res.items()):
    print(name[:110])
    print('   ' + '  '.join(f'{c}={v:.4g}' for c, v in sorted(cs.items())))
