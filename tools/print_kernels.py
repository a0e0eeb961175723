import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['config']['frames_per_step_per_gpu'])
tot = 0
for k in d['kernels']:
    print('%-46s x%3d  %8.1f us  %7.3f ms/step' % (k['kernel'][:46], k['launches_per_step'], k['avg_us'], k['ms_per_step']))
    tot += k['ms_per_step']
print('sum convs', tot)
if 'stages' in d: print([(s['stage'], s['ms_per_step']) for s in d['stages']])
