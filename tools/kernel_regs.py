"""Register / occupancy table of the kernels of one .hip file (cross-compiles for gfx950, no GPU needed).

    python tools/kernel_regs.py detzero_amd/csrc/sparse_conv_h.hip [filter]
"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    src = sys.argv[1]
    flt = sys.argv[2] if len(sys.argv) > 2 else ''
    cmd = ['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-c', src, '-I', ROOT + '/include',
           '-I', ROOT + '/detzero_amd/csrc', '-o', '/tmp/_regs.o', '-Rpass-analysis=kernel-resource-usage']
    err = subprocess.run(cmd, capture_output=True, text=True).stderr
    cur = None
    rows = []
    for line in err.splitlines():
        m = re.search(r'remark: +([A-Za-z \[\]/]+): +(\S+)', line)
        if not m:
            continue
        k, v = m.group(1).strip(), m.group(2)
        if k == 'Function Name':
            name = subprocess.run(['c++filt', v], capture_output=True, text=True).stdout.strip()
            cur = {'name': name}
            rows.append(cur)
        elif cur is not None:
            cur[k] = v
    for r in rows:
        if flt in r['name']:
            print('%-4s v %-4s a %-3s occ %-2s spill %-3s lds  %s' % (
                r.get('VGPRs'), r.get('AGPRs'), r.get('Occupancy [waves/SIMD]'), r.get('VGPRs Spill'),
                r.get('LDS Size [bytes/block]'), re.sub(r'^void dz::|\(dz::\w+\)$|dz::', '', r['name'])))


if __name__ == '__main__':
    main()
