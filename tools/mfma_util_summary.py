"""Matrix-core utilisation per kernel from rocprofv3 --pmc passes over the headline step (rocpd SQLite results) -> JSON for profiles/.
    python3 tools/mfma_util_summary.py sq_results.db grbm_results.db kernel_stats.csv out.json
Pass 1: SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES; pass 2: GRBM_GUI_ACTIVE (a run of its own).
  mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE x 128)      busy cycles are summed over the chip's 1024 SIMDs, GUI_ACTIVE
                                                                       over its 8 XCDs: 1024 / 8 = 128 SIMD-cycles per counted cycle
  flops     = SQ_INSTS_VALU_MFMA_MOPS_BF16 x 512                       (checked on a 262144 x 512 x 512 GEMM: 2.684e8 x 512 = 2 M N K)
  tflops    = flops / the kernel's average duration in the UNDISTURBED --kernel-trace --stats run (the csv)."""
import collections, csv, json, sqlite3, sys
WANT = ('gemm', 'attn', 'mfma')


def counters(path):
    db = sqlite3.connect(path)
    acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
    for name, cname, val in db.execute("select kernel_name, counter_name, value from counters_collection"):
        e = acc[name][cname]
        e[0] += 1; e[1] += val
    return acc


sq, gr = counters(sys.argv[1]), counters(sys.argv[2])
dur = {}
for row in csv.DictReader(open(sys.argv[3])):
    dur[row['Name']] = (int(float(row['Calls'])), float(row['AverageNs']), float(row['TotalDurationNs']))
out = {'source': 'rocprofv3 --kernel-trace --pmc <pass> -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline (two passes: SQ counters, '
                 'GRBM_GUI_ACTIVE), durations from the --kernel-trace --stats run of the same command; tools/profile_r02.sh',
       'definitions': {'mfma_busy': 'SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE * 128)', 'tflops': 'SQ_INSTS_VALU_MFMA_MOPS_BF16 * 512 / average duration',
                       'frac_of_peak': 'tflops / 2500 (dense bf16 MFMA peak)'},
       'kernels': {}}
rows = []
for name, cs in sq.items():
    if not any(w in name for w in WANT) or 'SQ_VALU_MFMA_BUSY_CYCLES' not in cs or name not in gr or name not in dur:
        continue
    n = cs['SQ_VALU_MFMA_BUSY_CYCLES'][0]
    busy = cs['SQ_VALU_MFMA_BUSY_CYCLES'][1] / n
    mops = cs['SQ_INSTS_VALU_MFMA_MOPS_BF16'][1] / n
    gui = gr[name]['GRBM_GUI_ACTIVE'][1] / gr[name]['GRBM_GUI_ACTIVE'][0]
    calls, avg_ns, tot_ns = dur[name]
    if mops == 0:
        continue
    tf = mops * 512 / (avg_ns * 1e-9) / 1e12
    rows.append((tot_ns, name, {'launches_in_pass': n, 'avg_us': avg_ns / 1e3, 'total_ms_in_stats_run': tot_ns / 1e6, 'mfma_busy': busy / (gui * 128),
                                'tflops': tf, 'frac_of_peak': tf / 2500.0}))
for _, name, e in sorted(rows, reverse=True):
    out['kernels'][name] = e
json.dump(out, open(sys.argv[4], 'w'), indent=1)
for name, e in out['kernels'].items():
    print(f"{name[:70]:70s} avg {e['avg_us']:8.1f} us  mfma_busy {e['mfma_busy']:.3f}  {e['tflops']:7.1f} TFLOP/s")
