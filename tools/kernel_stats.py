"""Summaries of rocprofv3 output under gpurun_out/ (written by tools/gpu.sh).

    kernel_stats.py TAG [STEPS]   prof_TAG/*_kernel_stats.csv -> TAG_kernel_stats.csv (copy) and
                                  a per-kernel table (us per step when STEPS is given)
    kernel_stats.py --pmc TAG     pmc_TAG/*counter_collection.csv -> mean counter values per
                                  spt:: kernel"""
import collections
import csv
import glob
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, 'gpurun_out')


def newest(pattern):
    files = glob.glob(pattern)
    files.sort(key=os.path.getmtime)          # gpurun merges runs: keep the newest pass
    return files[-1] if files else None


def stats(tag, steps=None):
    f = newest(os.path.join(OUT, 'prof_' + tag, '*', '*_kernel_stats.csv'))
    if f is None:
        print('no kernel_stats.csv for', tag)
        return 1
    shutil.copy(f, os.path.join(OUT, tag + '_kernel_stats.csv'))
    rows = list(csv.DictReader(open(f)))
    total = sum(float(r['TotalDurationNs']) for r in rows)
    calls = sum(int(r['Calls']) for r in rows)
    lines = ['total kernel ms {:.3f}  launches {}'.format(total / 1e6, calls)]
    if steps:
        lines[0] += '  per step ({} steps): {:.3f} ms, {:.1f} launches'.format(
            steps, total / 1e6 / steps, calls / steps)
    for r in rows[:40]:
        t, c = float(r['TotalDurationNs']), int(r['Calls'])
        head = '{:9.1f} us/step {:6.1f} x'.format(t / 1e3 / steps, c / steps) if steps else \
            '{:10.1f} us total {:6d} x'.format(t / 1e3, c)
        lines.append('{}  avg {:8.1f} us  {:5.1f} %  {}'.format(
            head, float(r['AverageNs']) / 1e3, 100 * t / total, r['Name'][:100]))
    text = '\n'.join(lines)
    open(os.path.join(OUT, tag + '_kernels.txt'), 'w').write(text + '\n')
    print(text)
    return 0


def pmc(tag):
    f = newest(os.path.join(OUT, 'pmc_' + tag, '*', '*counter_collection.csv'))
    if f is None:
        print('no counter_collection.csv for', tag)
        return 1
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        name = r['Kernel_Name'].split('(')[0].replace('void ', '')[:80]
        if 'spt::' in name:
            agg[name][r['Counter_Name']].append(float(r['Counter_Value']))
    for k, v in sorted(agg.items()):
        print(k, ' launches', len(next(iter(v.values()))))
        for c, vals in sorted(v.items()):
            print('    {:32s} {:16.1f}'.format(c, sum(vals) / len(vals)))
    return 0


if __name__ == '__main__':
    if sys.argv[1] == '--pmc':
        sys.exit(pmc(sys.argv[2]))
    sys.exit(stats(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else None))
