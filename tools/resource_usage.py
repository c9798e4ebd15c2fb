"""Per-kernel register / scratch / LDS table of one csrc/*.hip unit, from hipcc's
-Rpass-analysis=kernel-resource-usage remarks (cross-compiles; no GPU needed).

    python tools/resource_usage.py mfma_attention.hip [-DMA_E_VALUE=128 ...] [--grep rows]
"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, 'spt-proto_amd', 'csrc')
args = sys.argv[1:]
pat = None
if '--grep' in args:
    i = args.index('--grep'); pat = args[i + 1]; del args[i:i + 2]
src, extra = args[0], args[1:]
cmd = ['/opt/rocm/bin/hipcc', '-O3', '-std=c++17', '-fPIC', '--offload-arch=gfx950', '-munsafe-fp-atomics',
       '-fno-fast-math', '-fno-slp-vectorize', '-Rpass-analysis=kernel-resource-usage', '-c',
       os.path.join(CSRC, src), '-o', '/dev/null'] + extra
err = subprocess.run(cmd, capture_output=True, text=True).stderr
rows, cur = [], None
for line in err.splitlines():
    m = re.search(r'remark: [^:]+:\d+:\d+:\s+(.*?) \[-Rpass', line) or re.search(r':\d+:\d+: remark:\s+(.*?) \[-Rpass', line)
    if not m:
        if 'error' in line: print(line)
        continue
    text = m.group(1).strip()
    if text.startswith('Function Name:') or text.startswith('Name:'):
        cur = {'name': text.split(':', 1)[1].strip()}; rows.append(cur)
    elif cur is not None and ':' in text:
        k, v = text.split(':', 1); cur[k.strip()] = v.strip()
for r in rows:
    name = subprocess.run(['c++filt', r['name']], capture_output=True, text=True).stdout.strip()
    name = re.sub(r'\(.*', '', name).replace('void spt::', '')
    if pat and pat not in name: continue
    print('{:<70} vgpr {:>4} agpr {:>4} sgpr {:>4} scratch {:>5} occ {:>2} lds {:>7}'.format(
        name[:70], r.get('VGPRs', '?'), r.get('AGPRs', '?'), r.get('SGPRs', '?'), r.get('ScratchSize [bytes/lane]', '?'),
        r.get('Occupancy [waves/SIMD]', '?'), r.get('LDS Size [bytes/block]', '?')))
