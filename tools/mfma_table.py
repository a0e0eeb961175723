"""Per-kernel matrix-pipe accounting of a round's evidence set (the table of the round-5 review, weak 5):
  useful MFMA     = algorithmic TF/s / 838.9 (pair16 peak: 2516.6 / 3)                              [bench line `kernels`, HIP events]
  busy            = SQ_VALU_MFMA_BUSY_CYCLES / (4 x SQ_BUSY_CU_CYCLES)   (pipe busy while the CU is busy)   [rocprofv3 --pmc, SQ pass]
  useful / issued = 3 x algorithmic FLOP / (SQ_VALU_MFMA_BUSY_CYCLES / 32 x 32768)   (count based, clock independent)
  traffic         = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 / algorithmic bytes                          [pmc_traffic.json]
usage: python tools/mfma_table.py <bench_graph.json> <pmc_SQ_bench_eager3.txt> [pmc_traffic.json]"""
import json
import re
import sys


def main():
    bench = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    sq = {}
    for line in open(sys.argv[2]):
        m = re.match(r'(\S.*?)\s+(SQ_\w+|GRBM_\w+)\s+calls\s+(\d+)\s+sum\s+([\d.]+)', line)
        if m:
            sq.setdefault(m.group(1).strip(), {})[m.group(2)] = (int(m.group(3)), float(m.group(4)))
    traffic = json.load(open(sys.argv[3])) if len(sys.argv) > 3 else {}

    def find(table, name):
        key = re.sub(r'[<>]', ' ', name).split()          # k_spconv_x<128> -> ['k_spconv_x', '128']
        for k in table:
            kk = k.replace(' ', '')
            if kk.startswith(key[0]) and all(('<' + t in kk or ',' + t in kk or t in kk) for t in key[1:]):
                if key[0] == 'k_spconv_x' and not kk.startswith('k_spconv_x<XCfg<%s,' % key[1]):
                    continue
                return k
        return None
    # passes of the detector inside a PMC run: from a kernel whose launches per pass are known - the 128-channel x-run kernel runs 4 times
    # per pass and per concurrent sub-pass (the PMC command runs the bench's default: config.concurrent_sub_passes)
    ways = int(bench['config'].get('concurrent_sub_passes', 1) or 1)
    # `kernels` describes ONE profiled pass of profiled_pass_frames frames; a step of the PMC command covers frames_per_step_per_gpu
    scale = float(bench['config'].get('frames_per_step_per_gpu', 1)) / float(bench.get('profiled_pass_frames', bench['config'].get('frames_per_step_per_gpu', 1)))
    def passes_of(table, field):
        k = find(table, 'k_spconv_x<128>')
        if not k:
            return None
        calls = table[k][field][0] if isinstance(table[k][field], tuple) else table[k][field]['calls']
        return calls / (4.0 * ways)
    p_sq = passes_of(sq, 'SQ_VALU_MFMA_BUSY_CYCLES')
    p_tr = passes_of(traffic, 'FETCH_SIZE') if traffic else None
    print('# passes in the SQ run: %s, in the traffic runs: %s (concurrent sub-passes: %d)' % (p_sq, p_tr, ways))
    print('%-26s %14s %8s %7s %9s %9s %8s' % ('kernel', 'launches x us', 'useful', 'busy', 'use/issue', 'HBM frac', 'traffic'))
    for k in bench['kernels']:
        name = k['kernel']
        flops_step = scale * k['tflops'] * 1e12 * k['ms_per_step'] * 1e-3
        bytes_step = scale * k['algorithmic_gbs'] * 1e9 * k['ms_per_step'] * 1e-3
        row = '%-26s %6.0f x %5.0f %8.3f' % (name, k['launches_per_step'], k['avg_us'], k['tflops'] / 838.9)
        s = find(sq, name)
        busy = ui = float('nan')
        if s and 'SQ_VALU_MFMA_BUSY_CYCLES' in sq[s] and 'SQ_BUSY_CU_CYCLES' in sq[s]:
            calls, mf = sq[s]['SQ_VALU_MFMA_BUSY_CYCLES']
            busy = mf / (4.0 * sq[s]['SQ_BUSY_CU_CYCLES'][1])
            # per-pass share: the PMC pass may split a step's launches differently (concurrent sub-passes) - totals per pass are equal
            issued = mf / max(p_sq or 1.0, 1e-9) / 32.0 * 32768.0
            ui = 3.0 * flops_step / issued
        tr = float('nan')
        t = find(traffic, name)
        if t:
            v = traffic[t]
            calls = v.get('FETCH_SIZE', v.get('WRITE_SIZE'))['calls']
            moved = (2048.0 * v.get('FETCH_SIZE', {'per_call': 0})['per_call'] + 1024.0 * v.get('WRITE_SIZE', {'per_call': 0})['per_call']) * calls / max(p_tr or 1.0, 1e-9)
            tr = moved / max(bytes_step, 1.0)
        print('%s %7.2f %9.2f %9.3f %7.2fx' % (row, busy, ui, k['algorithmic_gbs'] / 8000.0, tr))


if __name__ == '__main__':
    main()
